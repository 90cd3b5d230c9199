// lnqkv.hip — LayerNorm + the q | k | v projection of an attention block as ONE launch at model width 320
// (attention.py:519-521: `x = self.attn1(self.norm1(x), ...) + x` with to_q / to_k / to_v, attention.py:300-316;
// video_attention.py:90-93 for the temporal blocks).  At 72 x 128 (M = 258 048 tokens) the pair it replaces is a LayerNorm
// launch (330 MB fp32 in, 165 MB fp16 out: 82 us) and a K = 320, N = 960 GEMM whose 256 x 320 tiles spend more time in their
// fp16 epilogue than in the K loop (213 us = 744 TF/s).  What has to happen is: read the fp32 residual rows once (330 MB),
// write q | k | v once (495 MB), 158 GFLOP in between.
//
// Same family as ff_fused_kernel.h, without its constraints (no accumulator set that outlives a chunk, no GELU):
//   * a wave owns 2 blocks of 16 tokens; their normalised rows are built ONCE per tile from the fp32 stream in the MFMA
//     accumulator layout (lane (r, g): token r, channels 16 cb + 4 g .. + 3 of every block cb: 64 contiguous bytes per token
//     and instruction) and stay in registers as the B fragments of all 15 x 40 MFMA steps (two-pass statistics over the lanes
//     r, r + 16, r + 32, r + 48: the formula of norm.hip's layernorm16_kernel, one fp16 rounding — the same operand the
//     LayerNorm kernel writes);
//   * the 960 output features stream through in 15 chunks of 64: a chunk's weights (40 KB, packed once per parameter version
//     in MFMA fragment order with the K order of the accumulator layout, gcd_lnqkv_pack_f16) are double-buffered in LDS by
//     LDS-DMA, one counted wait + one barrier per chunk; out^T[feature][token] = W X^T with v_mfma_f32_16x16x32_f16, so a lane
//     ends with 4 consecutive features of one token per 16-row block — the packer orders the rows so that two blocks give 8
//     consecutive features: one 16-byte store, four lanes = 64 contiguous bytes per token;
//   * 8 waves = 2 per SIMD (32 accumulator + 80 operand registers): one wave's loads, statistics and stores run under the
//     other's MFMAs.  MFMAs are compiler builtins here (hipcc pads their hazards).
// Forms measured (tools/lnqkv_bench.py, cold caches, M = 258 048): 8 waves x 32 tokens 271-278 us (the default; 222 us inside
// the sampler step, where the rows were just written); with a chunk's stores under the next chunk's MFMAs (second accumulator
// set) 264-275; 4 waves x 64 tokens (one wave per SIMD, half the LDS fragment reads per MFMA) 297-306 — LDS is not what it
// waits for.  GCD_LNQKV_FORM selects them (development).
// Rounding points are those of the two launches it replaces: LayerNorm output to fp16, fp32 accumulation, q | k | v to fp16.
#include "common.h"

namespace {

constexpr int LQ_C = 320, LQ_CHUNK = 64, LQ_CHUNK_BYTES = LQ_CHUNK * LQ_C * 2;      // 40 960
constexpr int LQ_TILE = 256;      // tokens per workgroup tile = waves x token blocks of 16 per wave
constexpr int LQ_OFF_LN = 2 * LQ_CHUNK_BYTES, LQ_SMEM = LQ_OFF_LN + 2 * LQ_C * 4;    // 84 480 B

struct LnQkvK {
  const float* x32;
  int64_t ldx32;
  const float* gamma;
  const float* beta;
  float eps;
  const f16* Wp;      // [nchunks][40][64][8] fp16
  f16* out;
  int64_t ldo;
  int M, nchunks, sched;
};

__device__ __forceinline__ float lq_sum_lane_bits_45(float a) {
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

template <int LQ_WAVES, int LQ_T, bool PIPE>
__global__ __launch_bounds__(LQ_WAVES * 64) void lnqkv_kernel(const LnQkvK p) {
  static_assert(LQ_WAVES * LQ_T * 16 == LQ_TILE && 40 % LQ_WAVES == 0, "tile shape");
  constexpr int NP = 40 / LQ_WAVES;      // DMA pieces per wave and chunk
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);      // (scalar: DMA operands are SGPRs)
  const int r = lane & 15, g = lane >> 4;
  const int ntiles = (p.M + LQ_TILE - 1) / LQ_TILE;
  for (int i = t; i < 2 * LQ_C; i += LQ_WAVES * 64) ((float*)(smem + LQ_OFF_LN))[i] = i < LQ_C ? p.gamma[i] : p.beta[i - LQ_C];
  const __amdgpu_buffer_rsrc_t rsrcW =
      __builtin_amdgcn_make_buffer_rsrc((void*)p.Wp, 0, p.nchunks * LQ_CHUNK_BYTES, 0x00020000);
  // a chunk = 40 pieces of 1 KB; wave w copies pieces NP w .. NP w + NP - 1
  auto dma = [&](int chunk, int buf) {
#pragma unroll
    for (int n = 0; n < NP; ++n)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcW, (GCD_AS3 void*)(smem + buf * LQ_CHUNK_BYTES + (wave * NP + n) * 1024), 16,
                                               lane * 16, chunk * LQ_CHUNK_BYTES + (wave * NP + n) * 1024, 0, 0);
  };
  // stores are range-checked buffer stores, issued by every lane of every wave (rows past M carry an out-of-range offset):
  // the counted wait below relies on their number
  const __amdgpu_buffer_rsrc_t rsrcO =
      __builtin_amdgcn_make_buffer_rsrc((void*)p.out, 0, (int)((int64_t)p.M * p.ldo * 2), 0x00020000);
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  auto tile_of = [&](int i) { return (p.sched & 1) ? ntiles - 1 - i : i; };
  if ((int)blockIdx.x < ntiles) dma(0, 0);
  __syncthreads();      // (the LayerNorm affine is in LDS)
  int it = 0;           // chunks consumed so far by this workgroup: chunk `it` sits in buffer it & 1
  for (int tile0 = blockIdx.x; tile0 < ntiles; tile0 += gridDim.x) {
    const int m_base = tile_of(tile0) * LQ_TILE + wave * (16 * LQ_T);
    const bool has_next = tile0 + (int)gridDim.x < ntiles;
    // ---- normalised operand rows: X[tb][ks] = k-step ks of token block tb (k = channels 32 ks + 4 g + j, 32 ks + 16 + 4 g + j) ----
    f16x8 X[LQ_T][10];
#pragma unroll
    for (int tb = 0; tb < LQ_T; ++tb) {
      const float* src = p.x32 + (int64_t)min(m_base + 16 * tb + r, p.M - 1) * p.ldx32 + 4 * g;
      f32x4 z[20];
#pragma unroll
      for (int cb = 0; cb < 20; ++cb) z[cb] = *(const f32x4*)(src + 16 * cb);
      float sm = 0.f;
#pragma unroll
      for (int cb = 0; cb < 20; ++cb) sm += (z[cb][0] + z[cb][1]) + (z[cb][2] + z[cb][3]);
      float mean = lq_sum_lane_bits_45(sm) * (1.0f / 320.0f);
      asm volatile("" : "+v"(mean));      // (the passes stay in program order: interleaved by hipcc they spill the rows)
      float q = 0.f;
#pragma unroll
      for (int cb = 0; cb < 20; ++cb)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float d = z[cb][e] - mean;
          q = fmaf(d, d, q);
        }
      float rstd = rsqrtf(lq_sum_lane_bits_45(q) * (1.0f / 320.0f) + p.eps);
      asm volatile("" : "+v"(rstd));
      int aff = LQ_OFF_LN + 16 * g;      // opaque per token block: the affine is RE-READ from LDS, not kept in 160 registers across blocks
      asm volatile("" : "+v"(aff));
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);      // (the affine reads stay BEHIND the statistics: hoisted, they spill the rows)
#pragma unroll
      for (int cq = 0; cq < 5; ++cq) {        // four channel blocks at a time
#pragma unroll
        for (int cb = 4 * cq; cb < 4 * cq + 4; ++cb) {
          const f32x4 ga = *(const f32x4*)(smem + aff + 64 * cb);
          const f32x4 be = *(const f32x4*)(smem + aff + 1280 + 64 * cb);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            // (z - mean) rstd ga + be as z a + (be - mean a): nothing shared with the variance pass, whose 80 differences
            // hipcc would otherwise keep alive beside the 80 rows
            const float a = rstd * ga[e];
            X[tb][cb >> 1][4 * (cb & 1) + e] = (f16)fmaf(z[cb][e], a, fmaf(-mean, a, be[e]));
          }
        }
        // (these two fragments are MADE here — hipcc otherwise sinks the arithmetic behind all 40 affine reads and spills
        //  them — and the next four blocks' affine is read after them)
        asm volatile("" : "+v"(X[tb][2 * cq]), "+v"(X[tb][2 * cq + 1]) : : "memory");
        __builtin_amdgcn_sched_barrier(0);
      }
      __builtin_amdgcn_sched_barrier(0);      // (one token block's 80 row registers at a time)
    }
    // ---- 15 chunks of 64 features ----
    // lane (r, g) holds features 64 c + 32 pr + 8 g + 4 (rb & 1) + e of token r for rb = 2 pr, 2 pr + 1: 16 bytes per pair
    auto store_acc = [&](const f32x4 (&a)[4][LQ_T], int c) {
#pragma unroll
      for (int tb = 0; tb < LQ_T; ++tb) {
        const int m = m_base + 16 * tb + r;
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
          f16x8 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            o[e] = (f16)a[2 * pr][tb][e];
            o[4 + e] = (f16)a[2 * pr + 1][tb][e];
          }
          const int off = m < p.M ? (int)(((int64_t)m * p.ldo + 64 * c + 32 * pr + 8 * g) * 2) : (int)0x7fffffff;
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rsrcO, off, 0, 0);
        }
      }
    };
    // PIPE: the results of chunk c - 1 (second accumulator set) are converted and stored in the middle of chunk c's MFMAs
    auto chunk = [&](int c, f32x4 (&acc)[4][LQ_T], const f32x4 (&prev)[4][LQ_T], bool prev_valid) {
      // chunk `it` was requested one iteration ago, in front of that iteration's 2 LQ_T stores: "at most the newest 2 LQ_T memory
      // operations outstanding" = its pieces have landed (loads return in order, stores share the counter).  (PIPE: a tile's
      // first iteration issues no stores, so its second one waits for everything.)  After the barrier every wave's pieces
      // are in and every wave is past its reads of the other buffer.
      if (PIPE && c == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if constexpr (LQ_T == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (c + 1 < p.nchunks) dma(c + 1, (it + 1) & 1);
      else if (has_next) dma(0, (it + 1) & 1);
      asm volatile("" ::: "memory");
      const unsigned char* wl = smem + (it & 1) * LQ_CHUNK_BYTES + lane * 16;
#pragma unroll
      for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int tb = 0; tb < LQ_T; ++tb) acc[rb][tb] = f32x4{0.f, 0.f, 0.f, 0.f};
      // fragments in batches of 8 (two k-steps), one batch read ahead: hipcc otherwise hoists all 40 reads (160 registers)
      f16x8 wf[2][8];
#pragma unroll
      for (int j = 0; j < 8; ++j) wf[0][j] = *(const f16x8*)(wl + j * 1024);
#pragma unroll
      for (int kb = 0; kb < 5; ++kb) {
        if (kb + 1 < 5) {
#pragma unroll
          for (int j = 0; j < 8; ++j) wf[(kb + 1) & 1][j] = *(const f16x8*)(wl + ((kb + 1) * 8 + j) * 1024);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int ks = 2 * kb + (j >> 2), rb = j & 3;
#pragma unroll
          for (int tb = 0; tb < LQ_T; ++tb)
            acc[rb][tb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[kb & 1][j], X[tb][ks], acc[rb][tb], 0, 0, 0);
        }
        if (PIPE && kb == 1 && prev_valid) store_acc(prev, c - 1);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (!PIPE) store_acc(acc, c);
      ++it;
    };
    f32x4 accA[4][LQ_T], accB[4][LQ_T];
    int c = 0;
    for (; c + 1 < p.nchunks; c += 2) {
      chunk(c, accA, accB, c > 0);
      chunk(c + 1, accB, accA, true);
    }
    if (c < p.nchunks) {
      chunk(c, accA, accB, c > 0);
      if (PIPE) store_acc(accA, c);
    } else if (PIPE) {
      store_acc(accB, c - 1);
    }
  }
}

// W [N][320] fp16 row-major (the engine's packed q | k | v weight: softmax scale folded into the q rows) -> fragment image:
// chunk c, fragment (ks, rb), lane (r, g): the 8 halves W[64 c + 32 (rb / 2) + 8 (r / 4) + 4 (rb & 1) + (r & 3)][k(ks, g, j)],
// k(ks, g, j) = 32 ks + 4 g + j for j < 4, 32 ks + 16 + 4 g + (j - 4) for j >= 4
__global__ void lnqkv_pack_kernel(const f16* __restrict__ W, f16* __restrict__ Wp, int nchunks) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;      // one (chunk, fragment, lane)
  if (idx >= nchunks * 40 * 64) return;
  const int lane = idx & 63, frag = (idx >> 6) % 40, c = idx / (40 * 64);
  const int ks = frag >> 2, rb = frag & 3, r = lane & 15, g = lane >> 4;
  const int row = 64 * c + 32 * (rb >> 1) + 8 * (r >> 2) + 4 * (rb & 1) + (r & 3);
  f16x8 v;
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = W[(int64_t)row * LQ_C + 32 * ks + (j < 4 ? 4 * g + j : 16 + 4 * g + (j - 4))];
  *(f16x8*)(Wp + (int64_t)idx * 8) = v;
}

#ifndef LQ_DEFAULT_FORM
#define LQ_DEFAULT_FORM 0
#endif

int lq_cu_count() {
  int d = 0, n = 0;
  if (hipGetDevice(&d) != hipSuccess) return 256;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess || n <= 0) return 256;
  return n;
}

}      // namespace

extern "C" int64_t gcd_lnqkv_packed_bytes(int N) { return N > 0 && N % 64 == 0 ? (int64_t)(N / 64) * LQ_CHUNK_BYTES : 0; }

extern "C" int gcd_lnqkv_supported(int C, int N) { return C == LQ_C && N > 0 && N % 64 == 0 && N <= 4096; }

extern "C" int gcd_lnqkv_pack_f16(const void* W, int N, void* wp, void* stream) {
  GCD_CHECK_ARG(W && wp, "gcd_lnqkv_pack_f16: null pointer");
  GCD_CHECK_ARG(gcd_lnqkv_supported(LQ_C, N), "gcd_lnqkv_pack_f16: N=%d (a multiple of 64, <= 4096)", N);
  GCD_CHECK_ARG(((uintptr_t)W & 15) == 0 && ((uintptr_t)wp & 15) == 0, "gcd_lnqkv_pack_f16: operands must be 16-byte aligned");
  const int n = (N / 64) * 40 * 64;
  hipLaunchKernelGGL(lnqkv_pack_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const f16*)W, (f16*)wp, N / 64);
  GCD_CHECK_LAUNCH();
  return 0;
}

extern "C" int gcd_lnqkv_f16(const float* x32, int64_t ldx32, const float* gamma, const float* beta, float eps, const void* wp,
                             void* out16, int64_t ldo, int M, int C, int N, int sched, void* stream) {
  GCD_CHECK_ARG(x32 && gamma && beta && wp && out16, "gcd_lnqkv_f16: null pointer");
  GCD_CHECK_ARG(gcd_lnqkv_supported(C, N), "gcd_lnqkv_f16: C=%d N=%d (C = 320, N a multiple of 64)", C, N);
  GCD_CHECK_ARG(M >= 1, "gcd_lnqkv_f16: M=%d", M);
  GCD_CHECK_ARG(ldx32 >= C && ldx32 % 4 == 0 && ((uintptr_t)x32 & 15) == 0, "gcd_lnqkv_f16: x32 rows must be 16-byte aligned");
  GCD_CHECK_ARG(ldo >= N && ldo % 8 == 0 && ((uintptr_t)out16 & 15) == 0, "gcd_lnqkv_f16: out rows must be 16-byte aligned");
  GCD_CHECK_ARG(((uintptr_t)wp & 15) == 0, "gcd_lnqkv_f16: wp must be 16-byte aligned");
  GCD_CHECK_ARG((int64_t)M * ldo * 2 < (int64_t)0x7fffffff, "gcd_lnqkv_f16: output of %lld bytes exceeds the 32-bit buffer offsets",
                (long long)((int64_t)M * ldo * 2));
  LnQkvK k;
  k.x32 = x32;
  k.ldx32 = ldx32;
  k.gamma = gamma;
  k.beta = beta;
  k.eps = eps;
  k.Wp = (const f16*)wp;
  k.out = (f16*)out16;
  k.ldo = ldo;
  k.M = M;
  k.nchunks = N / 64;
  k.sched = sched;
  const int ntiles = (M + LQ_TILE - 1) / LQ_TILE;
  // (development: A/B of the forms — bit 0: a chunk's stores under the next chunk's MFMAs; bit 1: 4 waves x 64 tokens instead
  //  of 8 waves x 32)
  static const int form = getenv("GCD_LNQKV_FORM") ? atoi(getenv("GCD_LNQKV_FORM")) : LQ_DEFAULT_FORM;
  const dim3 grid(std::min(ntiles, lq_cu_count()));
#define LQ_LAUNCH(W, T, P)                                                                                        \
  do {                                                                                                            \
    static GcdPerDeviceOnce once;                                                                                 \
    GCD_CHECK_HIP(once.opt_in((const void*)lnqkv_kernel<W, T, P>, LQ_SMEM));                                      \
    hipLaunchKernelGGL((lnqkv_kernel<W, T, P>), grid, dim3(W * 64), LQ_SMEM, (hipStream_t)stream, k);             \
  } while (0)
  switch (form & 3) {
    case 0: LQ_LAUNCH(8, 2, false); break;
    case 1: LQ_LAUNCH(8, 2, true); break;
    case 2: LQ_LAUNCH(4, 4, false); break;
    default: LQ_LAUNCH(4, 4, true); break;
  }
#undef LQ_LAUNCH
  GCD_CHECK_LAUNCH();
  return 0;
}
