// ff_fused_kernel.h — FeedForward(GEGLU) of the C = 320 level as ONE kernel: LayerNorm'd tokens in, residual stream out;
// the 1280-wide hidden tensor (value * gelu(gate), one fp16 rounding — the same rounding point as the two-GEMM path) never
// leaves the CU.  Reference: sgm/modules/attention.py:87-121 (GEGLU, FeedForward), video_attention.py:109-140 (ff_in / ff of
// the temporal block and their residuals), util.py:364-368 (AlphaBlender on the last one).
//
// Shape of the work (round 6): 4 waves = ONE wave per SIMD, 512 registers each.  A wave owns T
// blocks of 16 tokens across the FULL output width, so nothing is exchanged between waves; they only share the weight
// stream in LDS.  Per wave:
//   X        T x 10 B-fragments (16 tokens x 32 channels each) held in VGPRs for the whole tile,
//   out      T x 20 accumulator blocks (16 tokens x 16 output channels), fp32, in the ACCUMULATOR file, initialised with the
//            residual R1 (loaded straight into AGPRs: the epilogue of a plain residual FeedForward has no loads at all),
//   chunk j  = 32 hidden units: GEMM1^T  val|gate[hidden][token] = W1_j X^T   (A = W1 fragment from LDS, B = X)
//              -> h = val * gelu(gate) in the accumulator layout, which IS the B-fragment layout of
//              GEMM2^T  out^T[channel][token] += W2_j^T-block h   (A = W2 fragment from LDS) once W2's hidden index is
//              permuted at pack time (position p of lane group g <-> hidden 4 g + p, p < 4; 16 + 4 g + p - 4 otherwise).
// Weights are packed ONCE per parameter version in FRAGMENT ORDER (ff_pack_kernel: per chunk 40 W1 fragments + 20 W2
// fragments of 1 KB, lane-linear), so the LDS-DMA copies straight 1 KB pieces and every ds_read_b128 is lane-linear:
// no swizzle, no bank conflicts, one address register.
// Schedule of iteration i (60 slots of T MFMAs, one weight fragment each):
//   S1 slots  0-19  GEMM1(i) value / gate rows of hidden 0-15 (alternating, so that dependent MFMAs are 2 T apart)
//   S2 slots 20-39  GEMM1(i) value / gate rows of hidden 16-31
//   S3 slots 40-59  GEMM2(i-1)
// and the GELU of chunk i as single VALU operations placed one by one behind the 60 T MFMAs that follow MFMA 22 T of
// iteration i (~2 per MFMA): a wave's own VALU work issues in the shadow of its own MFMAs, nothing else does on gfx950
// (tools/issue_probe).  Every MFMA and every GELU operation is pinned (volatile asm): program order is issue order —
// left to hipcc the MFMA results land in the accumulator file and come back through v_accvgpr_read, and the polynomial
// is emitted in clumps of 10-18 instructions between MFMA pairs.  LDS-DMA of W1(i+1) and W2(i): one piece every second
// slot; ONE `vmcnt(0)` + `s_barrier` per iteration; fragments are read in batches of FB slots, one batch ahead, with ONE
// wait per batch.  Measured: tools/ff_fused_probe, profiles/r06_ff_fused_probe.txt.
#pragma once
#include <type_traits>

#include "common.h"

template <int I, int N, class F>
__device__ __forceinline__ void ff_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    ff_static_for<I + 1, N>(f);
  }
}

struct FfK {
  const f16* X;        // [M][320] fp16 (LayerNorm output), leading dimension ldx — the LN = false kernels
  int64_t ldx;
  // the LN = true kernels (FeedForward WITH the LayerNorm in front of it, attention.py:566-572 / video_attention.py:109-140:
  // x = ff(norm(x)) + x): z = x32 [+ addvec[m / rows_per_vec]] is read ONCE, in the accumulator layout, and serves as the
  // residual (it initialises the accumulators: R1 is not read) AND, normalised over its 320 channels and rounded to fp16, as
  // GEMM1's operand — neither the LayerNorm kernel's 330 MB read + 165 MB write nor this kernel's 165 MB read of them happen
  const float* x32;
  int64_t ldx32;
  const float* ln_gamma;
  const float* ln_beta;
  float ln_eps;
  const float* addvec;     // optional per-row-block vector (the frame position embedding of video_attention.py:283-284)
  int64_t ld_addvec;
  int rows_per_vec;
  const f16* Wp;       // ff_pack_kernel output: [41][60][64][8] fp16 (chunk 40 = a copy of chunk 39: never consumed)
  const float* b1;     // [2560] fp32 in pack_geglu order (16 value / 16 gate)
  const float* b2;     // [320]
  const float* R1;     // fp32 [M][320] residual folded into the accumulators (weight 1 relative to the GEMM), or nullptr
  int64_t ldr1;
  const float* R2;     // second residual of the AlphaBlender form, or nullptr; out may alias R1 or R2
  int64_t ldr2;
  void* out;           // fp32 (or fp16 when out_f16) [M][320]
  int64_t ldo;
  int out_f16;
  // out = sa (acc + b2 + R1) + sr2 R2;  sa = s_acc, sr2 = s_r2, or per `rows_per_alpha` rows sa = 1 - alpha, sr2 = alpha
  // (util.py:364-368 with the residual inside the blended term: the host checks that this is the form it needs)
  const float* frame_alpha;
  int rows_per_alpha;
  float s_acc, s_r2;
  int M;
  int sched;           // bit 0: walk the tiles from the end
  unsigned long long* dbg;   // -DFF_TIMING: s_memtime stamps of workgroup 0, wave 0 (64 per tile)
};

constexpr int FF_C = 320, FF_HID = 1280, FF_NCH = 40;
constexpr int FF_W1_BYTES = 40960, FF_W2_BYTES = 20480, FF_CHUNK_BYTES = FF_W1_BYTES + FF_W2_BYTES;
constexpr int FF_OFF_W1 = 0, FF_OFF_W2 = 2 * FF_W1_BYTES, FF_OFF_B1 = FF_OFF_W2 + 2 * FF_W2_BYTES;
constexpr int FF_OFF_B2 = FF_OFF_B1 + 2560 * 4, FF_OFF_LN = FF_OFF_B2 + 320 * 4, FF_SMEM = FF_OFF_LN + 2 * 320 * 4;     // 136 960 B

// Probe knobs (tools/ff_fused_probe builds one binary per setting).  FF_ABL bits give WRONG results by design: 1 no LDS-DMA
// in the loop, 2 no GELU operations, 4 no vmcnt / barrier in the loop, 8 no GEMM1 MFMAs, 16 no GEMM2 MFMAs, 32 no epilogue
// stores, 64 no fragment reads in the loop, 128 / 256 / 512 lane-linear (fully coalesced, wrong) addresses for the stores /
// the residual loads / the X loads: what the 16-rows-x-64-B access shape of the accumulator layout costs.  FF_FB: slots per fragment batch (3, 5 or 6).  FF_DMA_EVERY: a DMA piece every
// n-th slot from slot 1 (3: the last one at slot 43, eleven slots before the wait; 2, the default: at slot 29 — -1 %).
// FF_KERNEL: the kernel's name — tools/ff_fused_probe builds its own instantiations beside the library's and must not
// share their symbols (a second registration of the same host stub shadows the first).
#ifndef FF_KERNEL
#define FF_KERNEL ff_fused_kernel
#endif
#ifndef FF_ABL
#define FF_ABL 0
#endif
#ifndef FF_FB
#define FF_FB 5
#endif
// FF_R1MOD: cache-policy modifiers of the residual loads (probe)
#if !defined(FF_R1MODE) || FF_R1MODE == 0
#define FF_R1MOD ""
#elif FF_R1MODE == 1
#define FF_R1MOD " nt"
#elif FF_R1MODE == 2
#define FF_R1MOD " sc1"
#else
#define FF_R1MOD " sc0 sc1"
#endif
#ifndef FF_DMA_EVERY
#define FF_DMA_EVERY 2
#endif
// FF_ZPRE: LayerNorm form, 1 = the next tile's residual rows are requested during this tile's last iteration where no second
// residual stream needs the registers (EPI 0), 2 = in every form, 0 = at the tile's top (the default: measured NEUTRAL, 718-734
// us either way at M = 258 048 — the loads then share the CU's memory pipe with the stores of the same iteration, and that
// iteration has only 40 MFMAs to hide anything under: the boundary's memory time is a sum of bytes, not of phases)
#ifndef FF_ZPRE
#define FF_ZPRE 0
#endif
// FF_CHAIN_NOP n: wait states carried inside the last MFMA of every GEMM1 accumulator chain (s_nop n; -1: none).  A 4-pass
// MFMA's result needs 8 before a VALU instruction may read it on gfx950; tools/ff_isa_audit.py rule 3 checks whatever is left.
#ifndef FF_CHAIN_NOP
#define FF_CHAIN_NOP 7
#endif
#define FF_STR2(x) #x
#define FF_STR(x) FF_STR2(x)
#if FF_CHAIN_NOP >= 0
#define FF_CHAIN_TAIL "\n\ts_nop " FF_STR(FF_CHAIN_NOP)
#else
#define FF_CHAIN_TAIL ""
#endif
// FF_STAGGER n: workgroup b starts ((b >> 3) & 7) * n * 8128 cycles late (b & 7 is its XCD: neighbours within an XCD are spread) (probe: do the tile-boundary HBM bursts of the CUs coincide?)
#ifndef FF_STAGGER
#define FF_STAGGER 0
#endif
#ifdef FF_TIMING
#define FF_STAMP(idx) do { if (blockIdx.x == 0 && t == 0 && p.dbg) p.dbg[(tile0 / gridDim.x) * 64 + (idx)] = __builtin_readcyclecounter(); } while (0)
#elif defined(FF_STAMP_MODE) && FF_STAMP_MODE >= 10
// (bisecting: real stamps for a subset of the indices)
#define FF_STAMP(idx) do { if (((FF_STAMP_MODE == 10 && (idx) == 0) || (FF_STAMP_MODE == 11 && (idx) == 1) || (FF_STAMP_MODE == 12 && (idx) >= 2 && (idx) <= 4) || (FF_STAMP_MODE == 13 && (idx) == 5) || (FF_STAMP_MODE == 14 && (idx) >= 10)) && blockIdx.x == 0 && t == 0 && p.dbg) p.dbg[(tile0 / gridDim.x) * 64 + (idx)] = __builtin_readcyclecounter(); } while (0)
#elif defined(FF_STAMP_MODE) && FF_STAMP_MODE == 1
#define FF_STAMP(idx) asm volatile("; stamp" ::: "memory")
#elif defined(FF_STAMP_MODE) && FF_STAMP_MODE == 2
#define FF_STAMP(idx) do { if (blockIdx.x == 0 && t == 0 && p.dbg) asm volatile("s_nop 0"); } while (0)
#else
#define FF_STAMP(idx) do {} while (0)
#endif

// Phi(x) - 1/2 = xc P(u), xc = clamp(x, -R, R), u = 2 xc^2 / R^2 - 1 (tools/fit_gelu.py).  DEG 19: common.h's gelu_fast
// (R = 4.5, |Phi error| <= 7.1e-6: what the GEGLU GEMM computes); DEG 15: R = 4.25, <= 5.3e-5 — a fifth of the 2.4e-4
// half-ulp of the fp16 rounding the product takes next — and two operations fewer per element.
template <int DEG>
struct FfGelu;
template <>
struct FfGelu<19> {
  static constexpr int N = 10;
  static constexpr float R = 4.5f;
  static constexpr float c[10] = {1.569060299e-01f, -7.718616753e-02f, 5.463833202e-02f, -4.023398011e-02f, 2.885118603e-02f,
                                  -1.849089463e-02f, 9.667675117e-03f, -6.067269522e-03f, 4.953202455e-03f, -1.928577467e-03f};
};
template <>
struct FfGelu<15> {
  static constexpr int N = 8;
  static constexpr float R = 4.25f;
  static constexpr float c[8] = {1.659238913e-01f, -8.081913157e-02f, 5.612811356e-02f, -3.876255066e-02f,
                                 2.314932873e-02f, -1.554679539e-02f, 1.299527435e-02f, -5.433510091e-03f};
};

// sum over the four lanes of a token's row (lane bits 4 and 5), result in all of them: two VALU cross-lane swaps
// (v_permlane16_swap, v_permlane32_swap) — no trip through the LDS crossbar
__device__ __forceinline__ float ff_sum_lane_bits_45(float a) {
  {
    const unsigned u = __float_as_uint(a);
    const auto sw = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    a = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
  }
  {
    const unsigned u = __float_as_uint(a);
    const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    a = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
  }
  return a;
}

// 16 bytes of the residual straight into the accumulator file; hipcc does not count an asm load (the caller's vmcnt(0) does)
template <int OFF>
__device__ __forceinline__ void ff_load_acc(f32x4& dst, const float* src) {
  asm volatile("global_load_dwordx4 %0, %1, off offset:%2" FF_R1MOD : "=a"(dst) : "v"(src), "n"(OFF) : "memory");
}

// EPI: 0 = out = sa (acc + b2 + R1), fp32; 1 = ... + sr2 R2 (AlphaBlender), fp32; 2 = the same with an fp16 result
// LN: the LayerNorm in front of the FeedForward is computed here (FfK.x32 ...); the weights then come from ff_pack_kernel
// with kperm = 1 (GEMM1's K order follows the accumulator layout's channel strips)
template <int T, int DEG, int EPI, bool LN = false>
__global__ __launch_bounds__(256) void FF_KERNEL(const FfK p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int g = lane >> 4, r = lane & 15;
  constexpr int TILE = 64 * T;     // tokens per workgroup tile
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  using GL = FfGelu<DEG>;
  constexpr bool has2 = EPI >= 1, OUT16 = EPI == 2;

  for (int i = t; i < 640; i += 256) ((f32x4*)(smem + FF_OFF_B1))[i] = ((const f32x4*)p.b1)[i];
  if (t < 80) ((f32x4*)(smem + FF_OFF_B2))[t] = ((const f32x4*)p.b2)[t];
  if constexpr (LN) {
    if (t >= 128 && t < 208) ((f32x4*)(smem + FF_OFF_LN))[t - 128] = ((const f32x4*)p.ln_gamma)[t - 128];
    if (t >= 64 && t < 128) ((f32x4*)(smem + FF_OFF_LN + 1280))[t - 64] = ((const f32x4*)p.ln_beta)[t - 64];
    if (t < 16) ((f32x4*)(smem + FF_OFF_LN + 1280))[64 + t] = ((const f32x4*)p.ln_beta)[64 + t];
  }
  __syncthreads();

  const __amdgpu_buffer_rsrc_t rsrcW =
      __builtin_amdgcn_make_buffer_rsrc((void*)p.Wp, 0, (FF_NCH + 1) * FF_CHUNK_BYTES, 0x00020000);
  const unsigned lane16 = (unsigned)lane * 16u;
  // the result leaves through a range-checked buffer descriptor: rows >= M are dropped by the hardware — no branch, no
  // exec mask, every wave issues exactly 20 T stores per tile (the counted wait at the tile boundary relies on that, and so
  // does hipcc's own count for the loads it tracks across them)
  const __amdgpu_buffer_rsrc_t rsrcO = __builtin_amdgcn_make_buffer_rsrc(
      p.out, 0, (int)((int64_t)p.M * p.ldo * (OUT16 ? 2 : 4)), 0x00020000);
  // piece n of a run of pieces that lie 1 KB apart in memory AND in LDS: pieces 4 q .. 4 q + 3 share one scalar offset pair
  // and differ in the instruction's 12-bit offset, which the hardware adds to both addresses — a quarter of the scalar
  // registers (the first build kept one SGPR pair per piece: the kernel sat at the SGPR limit)
  auto dma = [&](int goff, int lds_off, auto Nn) {
    constexpr int n = decltype(Nn)::value;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcW, (GCD_AS3 void*)(smem + lds_off + (n >> 2) * 4096), 16, (int)lane16,
                                             goff + (n >> 2) * 4096, (n & 3) * 1024, 0);
  };
  // slot s of an iteration -> fragment of the chunk: S1 / S2 alternate value and gate row blocks (rb = 2 (s / 20) + (s & 1),
  // k-step (s % 20) / 2), S3 walks the 20 channel blocks of W2
  auto frag_off = [](const int s) { return s < 40 ? (10 * (2 * (s / 20) + (s & 1)) + (s % 20) / 2) * 1024 : (s - 40) * 1024; };

  const int ntiles = (p.M + TILE - 1) / TILE;
  auto tile_of = [&](int tl0) { return (p.sched & 1) ? ntiles - 1 - tl0 : tl0; };
  f16x8 X[T][10];
  auto load_x = [&](f16x8 (&dst)[T][10], int mb) {
#pragma unroll
    for (int tb = 0; tb < T; ++tb) {
      const f16* src = (FF_ABL & 512) ? p.X + (int64_t)min(mb + 16 * tb, p.M - 16) * p.ldx + lane * 8
                                      : p.X + (int64_t)min(mb + 16 * tb + r, p.M - 1) * p.ldx + 8 * g;
#pragma unroll
      for (int ks = 0; ks < 10; ++ks) dst[tb][ks] = *(const f16x8*)(src + ((FF_ABL & 512) ? 512 : 32) * ks);
    }
  };
  // LDS buffer parity: chunk i of the current tile sits in W1 buffer (i + par) & 1, W2 buffer (i + par) & 1; a tile has 41
  // iterations, so the next tile's W1(0) — DMA'd during this tile's last iteration — lands in buffer (41 + par) & 1 = par ^ 1
  if constexpr (FF_STAGGER > 0) {
    for (int k = 0; k < (int)((blockIdx.x >> 3) & 7) * FF_STAGGER; ++k) __builtin_amdgcn_s_sleep(127);
  }
  int par = 0;
  if ((int)blockIdx.x < ntiles) {
    // ---- first tile of this workgroup: W1(0) -> buffer 0, X fragments ----
    ff_static_for<0, 10>([&](auto Nn) { dma(wave * 10240, FF_OFF_W1 + wave * 10240, Nn); });
    if constexpr (!LN) load_x(X, tile_of(blockIdx.x) * TILE + wave * (16 * T));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  // hipcc counts the X loads but not the asm loads the first iteration issues between its MFMAs: left to itself it waits for
  // X inside that iteration with counts (vmcnt(9) ...) that also drain the residual loads and the previous tile's stores —
  // 9 memory instructions in flight per wave instead of 60 (measured: 17 000 cycles for the first 20 slots of every tile).
  // The pin makes it wait HERE, where the data is in (the vmcnt(0) above; the counted wait at every tile boundary below).
  auto pin_x = [&]() {
#pragma unroll
    for (int tb = 0; tb < T; ++tb)
      asm volatile("" : "+v"(X[tb][0]), "+v"(X[tb][1]), "+v"(X[tb][2]), "+v"(X[tb][3]), "+v"(X[tb][4]), "+v"(X[tb][5]),
                   "+v"(X[tb][6]), "+v"(X[tb][7]), "+v"(X[tb][8]), "+v"(X[tb][9]));
  };
  if constexpr (!LN) pin_x();
  // LayerNorm form: z = the fp32 residual rows of a tile in the ACCUMULATOR layout (lane (r, g): token r, channels 16 cb + 4 g
  // .. + 3 of every block cb).  With FF_ZPRE the NEXT tile's z is requested at the start of this tile's last iteration — the
  // GELU state, the GEMM1 accumulators and the fragment batches are dead there: 160 registers are free — so that its 164 KB
  // per CU travel under that iteration's MFMAs and stores instead of alone at the tile's top.
  constexpr bool ZPRE = LN && FF_ZPRE != 0 && (FF_ZPRE == 2 || !has2);
  f32x4 zn[LN ? T : 1][LN ? 20 : 1];
  auto load_z = [&](int m0) {
    if constexpr (LN) {
#pragma unroll
      for (int tb = 0; tb < T; ++tb) {
        const float* src = p.x32 + (int64_t)min(m0 + 16 * tb + r, p.M - 1) * p.ldx32 + 4 * g;
#pragma unroll
        for (int cb = 0; cb < 20; ++cb) zn[tb][cb] = *(const f32x4*)(src + 16 * cb);
      }
      asm volatile("" ::: "memory");      // (requested HERE: the loads may not sink below the MFMAs that follow)
    }
  };
  if constexpr (ZPRE) {
    if ((int)blockIdx.x < ntiles) load_z(tile_of(blockIdx.x) * TILE + wave * (16 * T));
  }

  for (int tile0 = blockIdx.x; tile0 < ntiles; tile0 += gridDim.x) {
    const int m_base = tile_of(tile0) * TILE + wave * (16 * T);
    const bool has_next = tile0 + (int)gridDim.x < ntiles;
    const int m_next = has_next ? tile_of(tile0 + gridDim.x) * TILE + wave * (16 * T) : m_base;
    FF_STAMP(0);
    f32x4 out[20][T];
    if constexpr (LN) {
      // z = x32 [+ addvec] in the ACCUMULATOR layout (lane (r, g): token r, channels 16 cb + 4 g .. + 3 of every block cb):
      // the residual the accumulators start from and, normalised (two-pass statistics over the 320 channels of a token =
      // the 80 values of a lane + the lanes r + 16, r + 32, r + 48; the formula of norm.hip's layernorm16_kernel) and rounded
      // to fp16, GEMM1's B fragments — the packer's K order (kperm) makes k-step ks of a lane exactly blocks 2 ks, 2 ks + 1.
      if constexpr (!ZPRE) load_z(m_base);
      f32x4 z[T][20];
#pragma unroll
      for (int tb = 0; tb < T; ++tb)
#pragma unroll
        for (int cb = 0; cb < 20; ++cb) z[tb][cb] = zn[tb][cb];
      if (p.addvec) {
        const float* av = p.addvec + (int64_t)(min(m_base, p.M - 1) / p.rows_per_vec) * p.ld_addvec + 4 * g;
#pragma unroll
        for (int cb = 0; cb < 20; ++cb) {
          const f32x4 a = *(const f32x4*)(av + 16 * cb);
#pragma unroll
          for (int tb = 0; tb < T; ++tb) z[tb][cb] += a;
        }
      }
#pragma unroll
      for (int tb = 0; tb < T; ++tb) {
        float sm = 0.f;
#pragma unroll
        for (int cb = 0; cb < 20; ++cb) sm += (z[tb][cb][0] + z[tb][cb][1]) + (z[tb][cb][2] + z[tb][cb][3]);
        const float mean = ff_sum_lane_bits_45(sm) * (1.0f / 320.0f);
        float q = 0.f;
#pragma unroll
        for (int cb = 0; cb < 20; ++cb)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float d = z[tb][cb][e] - mean;
            q = fmaf(d, d, q);
          }
        const float rstd = rsqrtf(ff_sum_lane_bits_45(q) * (1.0f / 320.0f) + p.ln_eps);
#pragma unroll
        for (int cb = 0; cb < 20; ++cb) {
          const f32x4 ga = *(const f32x4*)(smem + FF_OFF_LN + (16 * cb + 4 * g) * 4);
          const f32x4 be = *(const f32x4*)(smem + FF_OFF_LN + 1280 + (16 * cb + 4 * g) * 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) X[tb][cb >> 1][4 * (cb & 1) + e] = (f16)((z[tb][cb][e] - mean) * rstd * ga[e] + be[e]);
          out[cb][tb] = z[tb][cb];
          asm volatile("" : "+a"(out[cb][tb]));
        }
      }
      pin_x();      // (made by VALU conversions: behind the pin before the first MFMA reads them)
#if defined(FF_DEBUGX) && FF_DEBUGX == 1
      if (p.dbg && tile0 == (int)blockIdx.x) {      // probe: this workgroup's first tile's operand fragments and statistics
        f16x8* d = (f16x8*)p.dbg + ((size_t)blockIdx.x * 256 + t) * (T * 10);
#pragma unroll
        for (int tb = 0; tb < T; ++tb)
#pragma unroll
          for (int ks = 0; ks < 10; ++ks) d[tb * 10 + ks] = X[tb][ks];
      }
#endif
#if defined(FF_LNFIX) && FF_LNFIX == 1
      __builtin_amdgcn_sched_barrier(0);
#elif defined(FF_LNFIX) && FF_LNFIX == 2
      asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
#elif defined(FF_LNFIX) && FF_LNFIX == 3
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
#endif
    }
    f32x4 aG[4][T];
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
      for (int tb = 0; tb < T; ++tb) aG[rb][tb] = f32x4{0.f, 0.f, 0.f, 0.f};
    // H of the chunk whose GEMM2 comes next (Hcur) and of the chunk whose GELU is running (Hnext): per token block four
    // dwords = the B fragment of GEMM2 (dword 2 pair + (e >> 1) = packed fp16 of registers e, e + 1 of hidden half `pair`)
    u32x4 Hcur[T], Hnext[T];
#pragma unroll
    for (int tb = 0; tb < T; ++tb) Hcur[tb] = Hnext[tb] = u32x4{0u, 0u, 0u, 0u};
    // GELU state of the 8 T elements of a chunk (a handful live at any time)
    float e_va[8 * T], e_g[8 * T], e_xc[8 * T], e_u[8 * T], e_p[8 * T];
#pragma unroll
    for (int k = 0; k < 8 * T; ++k) e_va[k] = e_g[k] = e_xc[k] = e_u[k] = e_p[k] = 0.f;
    const float* r1src[T] = {};
    if constexpr (!LN) {
#pragma unroll
      for (int tb = 0; tb < T; ++tb)
        r1src[tb] = (FF_ABL & 256) ? p.R1 + (int64_t)min(m_base + 16 * tb, p.M - 16) * p.ldr1 + lane * 4
                                   : p.R1 + (int64_t)min(m_base + 16 * tb + r, p.M - 1) * p.ldr1 + 4 * g;
    }

    // ---- the GELU of a chunk as single VALU operations placed one by one behind the MFMAs ----
    // Element k (0 .. 8T-1): hidden half k / (4T), token block (k >> 2) % T, accumulator register k & 3.  Elements are
    // evaluated two at a time (two independent dependency chains); a group = NST steps x 2 elements + one packed conversion;
    // the V operations of a chunk are spread evenly over the 60 T MFMAs that follow MFMA number 22 T of the chunk's own
    // iteration (two slots after GEMM1's first half is complete — a 4-pass MFMA's result needs 7 wait states before a VALU
    // reads it, nothing in the asm pads that — the second half's elements start 30 slots later).
    constexpr int NE = 8 * T, NST = GL::N + 5, GOPS = 2 * NST + 1, V = GOPS * (NE / 2), MT = 60 * T, ORG = 22 * T;
    auto gelu_op = [&](auto Oo) {
      constexpr int o = decltype(Oo)::value;
      constexpr int q = o / GOPS, w = o % GOPS;
      if constexpr (w == GOPS - 1) {
        constexpr int k = 2 * q, pair = k / (4 * T), tb = (k >> 2) % T, e = k & 3;
        const f16x2 pk = f16x2{(f16)e_p[k], (f16)e_p[k + 1]};
        unsigned bits = __builtin_bit_cast(unsigned, pk);
        asm volatile("" : "+v"(bits));
        Hnext[tb][2 * pair + (e >> 1)] = bits;
      } else {
        constexpr int k = 2 * q + (w & 1), st = w >> 1;
        constexpr int pair = k / (4 * T), tb = (k >> 2) % T, e = k & 3;
        float res;
        if constexpr (st == 0) {
          // The value / gate accumulators of a (hidden half, token block) are "redefined" IN PLACE by an empty asm when
          // their first element is taken (volatile asm statements keep their order: this one sits behind the MFMAs issued
          // so far, two slots after the block's last one), and every element is read from the redefined vectors.  Not a
          // per-element "+v" pin on a scalar copy: hipcc materialised that copy with a v_mov it was free to hoist right
          // behind the producing MFMA — a VALU read of a matrix-pipe result that is not there yet (nothing pads an asm
          // MFMA's hazards): two registers of one token block wrong, in the builds where the allocator chose to copy.
          if constexpr (e == 0) asm volatile("" : "+v"(aG[2 * pair][tb]), "+v"(aG[2 * pair + 1][tb]));
          const float va = aG[2 * pair][tb][e], ga = aG[2 * pair + 1][tb][e];
          e_va[k] = va;
          e_g[k] = ga;
          res = __builtin_amdgcn_fmed3f(ga, -GL::R, GL::R);
        } else if constexpr (st == 1) res = e_xc[k] * e_xc[k];
        else if constexpr (st == 2) res = fmaf(e_u[k], 2.0f / (GL::R * GL::R), -1.0f);
        else if constexpr (st == 3) res = fmaf(GL::c[GL::N - 1], e_u[k], GL::c[GL::N - 2]);
        else if constexpr (st < GL::N + 2) res = fmaf(e_p[k], e_u[k], GL::c[GL::N + 1 - st]);     // st = 4 .. N + 1 -> c[N-3] .. c[0]
        else if constexpr (st == GL::N + 2) res = fmaf(e_xc[k], e_p[k], 0.5f);
        else if constexpr (st == GL::N + 3) res = e_g[k] * e_p[k];
        else res = e_va[k] * e_p[k];
        asm volatile("" : "+v"(res));
        if constexpr (st == 0) e_xc[k] = res;
        else if constexpr (st == 1 || st == 2) e_u[k] = res;
        else e_p[k] = res;
      }
    };
    // behind MFMA number m of an iteration: at the GELU origin the finished H of chunk i-1 becomes GEMM2's operand; then
    // the operations of chunk i (m >= ORG; not in the last iteration) or the tail of chunk i-1's (m < ORG; not in the first)
    auto after_mfma = [&](auto Kind, auto Mm) {
      constexpr int KIND = decltype(Kind)::value, m = decltype(Mm)::value;
      if constexpr (m == ORG) {
        // (pinned: the copies are made HERE, 18 slots before GEMM2 reads them — a VALU write of an MFMA operand right in
        //  front of the MFMA is a hazard nothing pads inside asm; hipcc would be free to sink the moves to their first use)
#pragma unroll
        for (int tb = 0; tb < T; ++tb) {
          Hcur[tb] = Hnext[tb];
          asm volatile("" : "+v"(Hcur[tb]));
        }
      }
      if constexpr ((FF_ABL & 2) == 0 && (m >= ORG ? KIND != 2 : KIND != 0)) {
        constexpr int rel = m >= ORG ? m - ORG : m + MT - ORG;      // MFMAs since the chunk's GELU origin
        constexpr int o0 = (rel * V + MT - 1) / MT, o1 = ((rel + 1) * V + MT - 1) / MT;    // ops with floor(o MT / V) == rel
        ff_static_for<o0, o1>(gelu_op);
      }
    };

    // epilogue of one channel block: out = sa (acc + b2) + sr2 R2, in the accumulator layout (16 tokens x 64 B per access)
    float sa = p.s_acc, sr2 = p.s_r2;
    if (p.frame_alpha) {
      const float al = p.frame_alpha[min(m_base, p.M - 1) / p.rows_per_alpha];
      sa = 1.0f - al;
      sr2 = al;
    }
    f32x4 q2[8][T];      // R2 vectors of channel blocks cb - 1 .. cb + 6 while block cb - 2 is stored (a ring of eight:
                         // the loads are six slots old when hipcc's counted wait for them comes)
    auto fetch_r2 = [&](const int cb) {
#pragma unroll
      for (int tb = 0; tb < T; ++tb)
        q2[cb % 8][tb] = *(const f32x4*)(p.R2 + (int64_t)min(m_base + 16 * tb + r, p.M - 1) * p.ldr2 + 16 * cb + 4 * g);
    };
    auto store_cb = [&](auto Cb) {
      constexpr int cb = decltype(Cb)::value;
      if constexpr ((FF_ABL & 32) != 0) {
#pragma unroll
        for (int tb = 0; tb < T; ++tb) asm volatile("" ::"a"(out[cb][tb]));
        return;
      }
      const f32x4 bv = *(const f32x4*)(smem + FF_OFF_B2 + (16 * cb + 4 * g) * 4);
#pragma unroll
      for (int tb = 0; tb < T; ++tb) {
        const int m = m_base + 16 * tb + r;
        asm volatile("" : "+a"(out[cb][tb]));      // read it HERE, 2 T MFMAs behind its last one, not right behind that
        f32x4 v = (out[cb][tb] + bv) * sa;
        if constexpr (has2) v += sr2 * q2[cb % 8][tb];
        if constexpr ((FF_ABL & 128) != 0) {
          *(f32x4*)((float*)p.out + (int64_t)(m_base + 16 * tb) * p.ldo + cb * 256 + lane * 4) = v;
        } else if constexpr (OUT16) {
          f16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (f16)v[e];
          typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o), rsrcO,
                                                    (int)((m * (int)p.ldo + 16 * cb + 4 * g) * 2), 0, 0);
        } else {
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsrcO,
                                                     (int)((m * (int)p.ldo + 16 * cb + 4 * g) * 4), 0, 0);
        }
      }
    };

    // fragment batches: batch b = slots FB b .. FB b + FB - 1, read one batch ahead into fr[(b + 1) & 1]
    constexpr int FB = FF_FB, NBATCH = 60 / FB;
    static_assert(60 % FB == 0 && NBATCH % 2 == 0, "FB must divide 60 into an even number of batches");
    constexpr int BSLOT = 60 - FB - 1;      // vmcnt(0) + barrier: the slot before the last batch starts (and reads ahead)
    f16x8 fr[2][FB];
    f32x4 bias[4];
    {
      const int w1rd = FF_OFF_W1 + (par & 1) * FF_W1_BYTES + (int)lane16, b1rd = FF_OFF_B1 + 4 * g * 4;
#pragma unroll
      for (int j = 0; j < FB; ++j) fr[0][j] = *(const f16x8*)(smem + w1rd + frag_off(j));
      bias[0] = *(const f32x4*)(smem + b1rd);
      bias[1] = *(const f32x4*)(smem + b1rd + 64);
    }
    FF_STAMP(1);

    // One iteration.  KIND 0: the first (no GEMM2 yet; the residual is loaded into the accumulators, one channel block per
    // slot), 1: steady state, 2: the last (GEMM2 of chunk 39 only; the NEXT tile's X fragments and W1(0) are requested first,
    // every channel block is stored two slots after its last MFMA).
    f16x8 Xn[T][10];
    auto iteration = [&](auto Kind, const int i) {
      constexpr int KIND = decltype(Kind)::value;
      const int w1rd = FF_OFF_W1 + ((i + par) & 1) * FF_W1_BYTES + (int)lane16;
      const int w1rd_next = FF_OFF_W1 + ((i + 1 + par) & 1) * FF_W1_BYTES + (int)lane16;
      const int w2rd = FF_OFF_W2 + ((i + 1 + par) & 1) * FF_W2_BYTES + (int)lane16;     // W2(i-1)
      const int b1rd = FF_OFF_B1 + (64 * i + 4 * g) * 4;
      const int b1rd_next = FF_OFF_B1 + (64 * (i + 1) + 4 * g) * 4;
      const int w1dst = FF_OFF_W1 + ((i + 1 + par) & 1) * FF_W1_BYTES, w2dst = FF_OFF_W2 + ((i + par) & 1) * FF_W2_BYTES;
      const int gsrc1 = (KIND == 2 ? 0 : i + 1) * FF_CHUNK_BYTES, gsrc2 = i * FF_CHUNK_BYTES + FF_W1_BYTES;
      if constexpr (KIND == 2) {
        if (has_next) {
          if constexpr ((FF_ABL & 1) == 0) {
            ff_static_for<0, 10>([&](auto Nn) { dma(gsrc1 + wave * 10240, w1dst + wave * 10240, Nn); });
          }
          if constexpr (!LN) load_x(Xn, m_next);
        }
        if constexpr (ZPRE) load_z(m_next);      // (unconditional: the last tile re-reads its own rows; no branch around loads)
        if constexpr (has2) {
          ff_static_for<0, 6>([&](auto Cb) { fetch_r2(decltype(Cb)::value); });
        }
      }
      if constexpr (KIND == 2) {      // the tail of chunk 39's GELU (no MFMAs left to put it behind), then H(39) -> GEMM2's operand
        ff_static_for<0, ORG + 1>([&](auto Mm) { after_mfma(Kind, Mm); });
        asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");      // H(39) was just written by VALU moves; GEMM2 reads it next
        FF_STAMP(20);
      }
      ff_static_for<(KIND == 2 ? 40 - FB : 0), 60>([&](auto S) {
        constexpr int s = decltype(S)::value;
        constexpr int b = s / FB, j = s % FB;
        // Keep-alive pins.  hipcc considers an asm statement's inputs dead at the statement and hands their registers to the
        // next VALU result — while the matrix pipe is still reading them (sources are read over several cycles; C over all
        // passes).  Every MFMA source is therefore named once more, as an input of an empty asm, at a point that is at least
        // one MFMA later: the fragment of slot s - 1 behind the first MFMA of slot s, the bias vectors two slots after
        // their MFMAs; H's last readers (slot 59) carry their wait states inside the asm string.  tools/ff_isa_audit.py checks
        // the generated code for it.
        auto behind_first_mfma = [&]() {
          if constexpr (s > (KIND == 2 ? 40 : 0)) asm volatile("" ::"v"(fr[((s - 1) / FB) & 1][(s - 1) % FB]));
          else if constexpr (KIND != 0 && s == 0) asm volatile("" ::"v"(fr[((60 - 1) / FB) & 1][(60 - 1) % FB]));
          if constexpr (j == 0) {      // the next batch's reads go into the buffer the previous slot's fragment sat in: after its pin
            // (first iteration: no W2 reads; last: no W1 reads, and nothing is read ahead for the next tile before its barrier)
            if constexpr ((FF_ABL & 64) == 0 && !(KIND == 0 && s + FB >= 40 && s + FB < 60) && !(KIND == 2 && s + FB >= 60)) {
#pragma unroll
              for (int jj = 0; jj < FB; ++jj) {
                const int sn = (s + FB + jj) % 60;      // (the last batch of an iteration reads batch 0 of the next)
                const int base = s + FB >= 60 ? w1rd_next : (sn < 40 ? w1rd : w2rd);
                fr[(b + 1) & 1][jj] = *(const f16x8*)(smem + base + frag_off(sn));
              }
            }
          }
        };
        if constexpr (j == 0) {
          // one wait for the whole batch (the pin names all of its fragments), then the next batch's reads
          if constexpr (KIND != 2 || s >= 40) {
            if constexpr (FB == 3) asm volatile("" : "+v"(fr[b & 1][0]), "+v"(fr[b & 1][1]), "+v"(fr[b & 1][2]));
            else if constexpr (FB == 5)
              asm volatile("" : "+v"(fr[b & 1][0]), "+v"(fr[b & 1][1]), "+v"(fr[b & 1][2]), "+v"(fr[b & 1][3]), "+v"(fr[b & 1][4]));
            else
              asm volatile("" : "+v"(fr[b & 1][0]), "+v"(fr[b & 1][1]), "+v"(fr[b & 1][2]), "+v"(fr[b & 1][3]),
                           "+v"(fr[b & 1][4]), "+v"(fr[b & 1][5]));
          }
        }
        if constexpr (KIND == 2 && s < 40) {
          behind_first_mfma();      // (no MFMA in these slots: the read-ahead of the first W2 batch)
          return;
        }
        if constexpr (KIND == 0 && s >= 40) behind_first_mfma();      // (no GEMM2 in the first iteration)
        if constexpr (KIND != 2 && (s == 3 || s == 23)) {
          asm volatile("" ::"v"(bias[s == 3 ? 0 : 2]), "v"(bias[s == 3 ? 1 : 3]));      // C operands of the k-step-0 MFMAs two slots back
        }
        if constexpr (s == 14 && KIND != 2) {
          bias[2] = *(const f32x4*)(smem + b1rd + 128);
          bias[3] = *(const f32x4*)(smem + b1rd + 192);
        }
        if constexpr (s == 57 && KIND != 2) {
          bias[0] = *(const f32x4*)(smem + b1rd_next);
          bias[1] = *(const f32x4*)(smem + b1rd_next + 64);
        }
        const f16x8 a = fr[b & 1][j];
        if constexpr (s < 40) {
          constexpr int rb = 2 * (s / 20) + (s & 1), ks = (s % 20) / 2;
          if constexpr (KIND == 0 && s < 20 && !LN) {      // the residual of channel block s -> accumulators (landed by this iteration's vmcnt(0))
#pragma unroll
            for (int tb = 0; tb < T; ++tb) {
              if constexpr ((FF_ABL & 1024) != 0) out[s][tb] = f32x4{0.f, 0.f, 0.f, 0.f};      // (probe: no residual loads)
              else if constexpr ((FF_ABL & 256) != 0) ff_load_acc<0>(out[s][tb], r1src[tb] + s * 256);
              else ff_load_acc<s * 64>(out[s][tb], r1src[tb]);
            }
          }
          ff_static_for<0, T>([&](auto Tb) {
            constexpr int tb = decltype(Tb)::value;
            if constexpr ((FF_ABL & 8) != 0) {
              if (ks == 0) aG[rb][tb] = bias[rb] + __builtin_bit_cast(f32x4, a);
            } else if constexpr (ks == 0) {
              asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %3" : "=&v"(aG[rb][tb]) : "v"(a), "v"(X[tb][ks]), "v"(bias[rb]));
            } else if constexpr (ks == 9) {
              // the LAST MFMA of an accumulator's chain carries the result's wait states inside the string: from here on the
              // value is a finished one for hipcc, whose register allocator may copy it (live-range splitting) in the very
              // next instruction — it did, and read two registers the matrix pipe had not written yet
              asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" FF_CHAIN_TAIL : "+v"(aG[rb][tb]) : "v"(a), "v"(X[tb][ks]));
            } else {
              asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(aG[rb][tb]) : "v"(a), "v"(X[tb][ks]));
            }
            if constexpr (tb == 0) behind_first_mfma();
            after_mfma(Kind, std::integral_constant<int, s * T + tb>{});
          });
        } else {
          constexpr int cb = s - 40;
          ff_static_for<0, T>([&](auto Tb) {
            constexpr int tb = decltype(Tb)::value;
            // (hazards: A comes from a ds_read the compiler waits for, B = Hcur was completed 18 slots earlier, C = D is the
            //  accumulate chain: no wait states needed; a channel block is read two slots = 2 T MFMAs after its last one)
            if constexpr (KIND != 0) {
              if constexpr ((FF_ABL & 16) != 0) asm volatile("" : "+a"(out[cb][tb]) : "v"(a), "v"(Hcur[tb]));
              else if constexpr (s == 59 || KIND == 2)
                // the LAST readers of H(i-1) and of this iteration's last fragment: the wait states that keep the next VALU
                // write out of the registers the matrix pipe is reading go INTO the string (no pin survives the loop's
                // back edge: hipcc places the copies that start the next H right behind this statement)
                // (and, in the last iteration, every one: it is the last writer of its channel block)
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n\ts_nop 7" : "+a"(out[cb][tb]) : "v"(a), "v"(Hcur[tb]));
              else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(out[cb][tb]) : "v"(a), "v"(Hcur[tb]));
              if constexpr (tb == 0) behind_first_mfma();
            }
            after_mfma(Kind, std::integral_constant<int, s * T + tb>{});
          });
          if constexpr (KIND == 2) {
            if constexpr (cb >= 2) store_cb(std::integral_constant<int, cb - 2>{});
            if constexpr (cb + 6 < 20 && has2) fetch_r2(cb + 6);      // into the ring slot the store above just freed
          }
        }
        // LDS-DMA: the 10 W1(i+1) pieces, then the 5 W2(i) pieces of this wave, one every FF_DMA_EVERY slots from slot 1
        if constexpr ((FF_ABL & 1) != 0 || KIND == 2) {
        } else if constexpr (s % FF_DMA_EVERY == 1 && s < 10 * FF_DMA_EVERY) {
          constexpr int n = s / FF_DMA_EVERY;
          dma(gsrc1 + wave * 10240, w1dst + wave * 10240, std::integral_constant<int, n>{});
        } else if constexpr (s % FF_DMA_EVERY == 1 && s < 15 * FF_DMA_EVERY) {
          constexpr int n = s / FF_DMA_EVERY - 10;
          dma(gsrc2 + wave * 5120, w2dst + wave * 5120, std::integral_constant<int, n>{});
        }
        if constexpr (KIND == 0 && s == 20) FF_STAMP(10);
        if constexpr (KIND == 0 && s == 40) FF_STAMP(11);
        if constexpr (KIND == 2 && s == 40) FF_STAMP(21);
        if constexpr (KIND == 2 && s == 50) FF_STAMP(22);
        if constexpr (s == BSLOT && (FF_ABL & 4) == 0 && KIND != 2) {
          if constexpr (KIND == 0) FF_STAMP(12);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          if constexpr (KIND == 0) FF_STAMP(13);
          __builtin_amdgcn_s_barrier();
          if constexpr (KIND == 0) FF_STAMP(14);
        }
      });
      if constexpr (KIND == 2) {
        store_cb(std::integral_constant<int, 18>{});
        store_cb(std::integral_constant<int, 19>{});
      }
    };

    iteration(std::integral_constant<int, 0>{}, 0);
    FF_STAMP(2);
#if defined(FF_DEBUGX) && FF_DEBUGX == 3
    if (p.dbg && tile0 == (int)blockIdx.x) {      // probe: the accumulators after the first iteration (= the residual they started from)
      f32x4* d = (f32x4*)p.dbg + ((size_t)blockIdx.x * 256 + t) * (T * 20);
#pragma unroll
      for (int tb = 0; tb < T; ++tb)
#pragma unroll
        for (int cb = 0; cb < 20; ++cb) {
          asm volatile("" : "+a"(out[cb][tb]));
          d[tb * 20 + cb] = out[cb][tb];
        }
    }
#endif
#if defined(FF_DEBUGX) && FF_DEBUGX == 2
    if (p.dbg && tile0 == (int)blockIdx.x) {      // probe: the operand fragments again, after the first iteration
      f16x8* d = (f16x8*)p.dbg + ((size_t)blockIdx.x * 256 + t) * (T * 10);
#pragma unroll
      for (int tb = 0; tb < T; ++tb)
#pragma unroll
        for (int ks = 0; ks < 10; ++ks) d[tb * 10 + ks] = X[tb][ks];
    }
#endif
    for (int i = 1; i < FF_NCH; ++i) {
      iteration(std::integral_constant<int, 1>{}, i);
#if defined(FF_DEBUGX) && FF_DEBUGX == 4
      if (i == 1 && p.dbg && tile0 == (int)blockIdx.x) {      // probe: H(0) as GEMM2 received it
        u32x4* d = (u32x4*)p.dbg + ((size_t)blockIdx.x * 256 + t) * T;
#pragma unroll
        for (int tb = 0; tb < T; ++tb) d[tb] = Hcur[tb];
      }
#endif
    }
    FF_STAMP(3);
#if defined(FF_DEBUGX) && FF_DEBUGX == 5
    if (p.dbg && tile0 == (int)blockIdx.x) {      // probe: H(38) as GEMM2 received it in iteration 39
      u32x4* d = (u32x4*)p.dbg + ((size_t)blockIdx.x * 256 + t) * T;
#pragma unroll
      for (int tb = 0; tb < T; ++tb) d[tb] = Hcur[tb];
    }
#endif
    iteration(std::integral_constant<int, 2>{}, FF_NCH);
    FF_STAMP(4);
    // Tile boundary: the next tile's W1(0) pieces and X fragments were requested BEFORE this tile's stores; loads return in
    // order and stores share the counter, so "at most the newest 20 T stores outstanding" = everything older has
    // landed (the stores are range-checked buffer stores: every wave issues all of them).  Then the barrier: every wave's pieces are in, and every
    // wave is past its last read of the W2 buffer the next tile's first iteration overwrites.
    if constexpr (T == 2) asm volatile("s_waitcnt vmcnt(40)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(60)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    FF_STAMP(5);
    if constexpr (!LN) {
#pragma unroll
      for (int tb = 0; tb < T; ++tb)
#pragma unroll
        for (int ks = 0; ks < 10; ++ks) X[tb][ks] = Xn[tb][ks];
      pin_x();
    }
    par ^= 1;
  }
}

// Fragment-order packing of one FeedForward's weights (once per parameter version).
//   w1: [2560][320] fp16 in pack_geglu row order (32-row blocks of 16 value + 16 gate rows), w2: [320][1280] fp16
//   Wp: [41][60][64][8]: chunk j, fragment f, lane l:
//     f < 40:  rb = f / 10, ks = f % 10:  w1[64 j + 16 rb + (l & 15)][32 ks + 8 (l >> 4) + 0..7]   (kperm = 0)
//                                         w1[..][32 ks + 16 (q >> 2) + 4 (l >> 4) + (q & 3)], q = 0..7  (kperm = 1: the LN kernels)
//     f >= 40: cb = f - 40:               w2[16 cb + (l & 15)][32 j + hp(l >> 4, 0..7)],  hp(g, p) = p < 4 ? 4 g + p : 12 + 4 g + p
__global__ void ff_pack_kernel(const f16* __restrict__ w1, const f16* __restrict__ w2, f16* __restrict__ Wp, int kperm) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;      // one 16-byte lane entry each
  if (idx >= (FF_NCH + 1) * 60 * 64) return;
  const int l = idx & 63, f = (idx >> 6) % 60, jj = idx / (60 * 64);
  const int j = min(jj, FF_NCH - 1);
  const int g = l >> 4, r = l & 15;
  f16x8 v;
  if (f < 40) {
    const int rb = f / 10, ks = f % 10;
    const f16* row = w1 + (int64_t)(64 * j + 16 * rb + r) * FF_C + 32 * ks;
    if (kperm) {      // position q of lane group g <-> channel 16 (q >> 2) + 4 g + (q & 3) of the k-step: the accumulator layout's strips
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = row[16 * (q >> 2) + 4 * g + (q & 3)];
    } else {
      v = *(const f16x8*)(row + 8 * g);
    }
  } else {
    const int cb = f - 40;
    const f16* src = w2 + (int64_t)(16 * cb + r) * FF_HID + 32 * j;
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = src[q < 4 ? 4 * g + q : 12 + 4 * g + q];
  }
  *(f16x8*)(Wp + (int64_t)idx * 8) = v;
}
