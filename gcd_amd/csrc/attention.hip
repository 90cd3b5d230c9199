// attention.hip — self-attention kernels of the SVD VideoUNet for gfx950 (head dim 64, fp16
// operands, fp32 softmax / accumulation).
//
//  * gcd_attn_spatial_f16: flash-style attention over the H*W tokens of one frame (sequence up to
//    9216), 19.4 % of the step's FLOPs.  4 waves x 32 queries per workgroup,
//    64-key K / V^T tiles staged with LDS-DMA into swizzled LDS (double-buffered, one barrier per
//    tile).  QK^T is computed *swapped* (S^T = K Q^T, v_mfma_f32_32x32x16_f16) so that a lane owns
//    one query column: the row max / row sum are in-register reductions plus ONE cross-half shuffle,
//    and the exponentiated tile is already laid out as the B operand of O^T += V^T P^T — the key
//    order (r&3) + 8(r>>2) + 4(lane>>5) of the accumulator registers is reused as the MFMA k-slot
//    order and V^T is fetched from LDS with the matching pattern, so P never moves between lanes.
//  * gcd_attn_transpose_v: builds the V^T[head][d][token] operand (zero padded to 64 keys).
//  * gcd_attn_temporal_f16: attention over the T <= 16 frames of one pixel; 0.05 % of the FLOPs and
//    HBM-bound, so it runs on the VALU (v_dot2_f32_f16 / v_fma_mix) with 16 lanes per problem.
#include "gemm_common.h"

#define LOG2E_F 1.4426950408889634f

// max(a, b, c) as ONE instruction.  fmaxf on MFMA outputs makes hipcc emit a canonicalising
// v_max_f32 x, x per operand (IEEE maxNum quieting), which tripled the row-max cost of a VALU-bound
// kernel; v_max3_f32 quiets NaNs in hardware.
__device__ __forceinline__ float max3_f(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

// ------------------------------------------------------------------------------------------------
// Per-head transpose: src[(f*S+s)*ld + col0 + h*64 + d] -> vt[((f*heads+h)*64 + d)*S_pad + perm(s)]
// (col0 = 2C: the V third of q|k|v for the forward's P V product; the backward kernels of attn_bwd.hip
// transpose K, Q and dO the same way), perm swaps the two middle quads of every 16-key group (see the
// kernel).  grid (S_pad/64, heads, frames), block 256
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_transpose_v_kernel(const f16* __restrict__ qkv,
                                                               int64_t ld, int col0, int S, int heads,
                                                               f16* __restrict__ vt, int S_pad) {
  __shared__ __attribute__((aligned(16))) f16 tile[64][72];
  const int t = threadIdx.x;
  const int s0 = blockIdx.x * 64, h = blockIdx.y, f = blockIdx.z;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int p = i * 256 + t;
    const int tok = p >> 3, ch = p & 7;
    f16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (s0 + tok < S) v = *(const f16x8*)(qkv + ((int64_t)f * S + s0 + tok) * ld + col0 + h * 64 + ch * 8);
    *(f16x8*)(&tile[tok][ch * 8]) = v;
  }
  __syncthreads();
  const int d = t >> 2, tg = (t & 3) * 16;
  f16x8 o0, o1;
  // within each group of 16 keys the two middle quads are swapped (keys 0-3, 8-11, 4-7, 12-15):
  // the 8 keys a lane of the P V MFMA needs, (r&3) + 8(r>>2) + 4(lane>>5), are then 16 contiguous bytes
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    o0[e] = tile[tg + (e < 4 ? e : e + 4)][d];
    o1[e] = tile[tg + (e < 4 ? e + 4 : e + 8)][d];
  }
  f16* dst = vt + (((int64_t)f * heads + h) * 64 + d) * S_pad + s0 + tg;
  *(f16x8*)dst = o0;
  *(f16x8*)(dst + 8) = o1;
}

extern "C" int gcd_attn_transpose_v(const void* qkv, int64_t ld, int frames, int S, int heads,
                                    void* vt, int S_pad, void* stream) {
  GCD_CHECK_ARG(qkv && vt, "gcd_attn_transpose_v: null pointer");
  GCD_CHECK_ARG(frames > 0 && S > 0 && heads > 0, "gcd_attn_transpose_v: empty problem");
  GCD_CHECK_ARG(S_pad % 64 == 0 && S_pad >= S, "gcd_attn_transpose_v: S_pad=%d for S=%d", S_pad, S);
  GCD_CHECK_ARG(ld % 8 == 0 && ld >= 3 * heads * 64, "gcd_attn_transpose_v: ld=%lld", (long long)ld);
  GCD_CHECK_ARG(frames <= 65535 && heads <= 65535, "gcd_attn_transpose_v: grid too large");
  hipLaunchKernelGGL(attn_transpose_v_kernel, dim3(S_pad / 64, heads, frames), dim3(256), 0,
                     (hipStream_t)stream, (const f16*)qkv, ld, 2 * heads * 64, S, heads, (f16*)vt, S_pad);
  GCD_CHECK_LAUNCH();
  return 0;
}

// the same transpose for any 64-wide head slice (attn_bwd.hip: K, Q of q|k|v, dO); arguments validated by the caller
int gcd_attn_transpose_heads_launch(const f16* src, int64_t ld, int col0, int frames, int S, int heads, f16* out,
                                    int S_pad, hipStream_t s) {
  hipLaunchKernelGGL(attn_transpose_v_kernel, dim3(S_pad / 64, heads, frames), dim3(256), 0, s, src, ld, col0, S,
                     heads, out, S_pad);
  GCD_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Spatial flash attention.  grid (ceil(S/128), heads, frames) remapped XCD-aware, block 256.
//
// Measured facts that shaped the loop (tools/attn_bench ablations, profiles/r01_attention_*): a SIMD
// issues roughly one instruction per 4 cycles whatever the mix and a transcendental keeps its unit
// for 16, so with 3 waves per SIMD the kernel is bound by INSTRUCTION COUNT per 32 x 64 score block
// (~240 before), not by MFMA, LDS or HBM.  Hence:
//   * PRESCALED: the softmax scale times log2(e) is folded into W_q when the weights are packed
//     (fp32 multiply before the one fp16 rounding), so a score is already an exp2 argument;
//   * the running reference max is subtracted BY THE MATRIX PIPE: Q K^T accumulates onto a register
//     block holding -m_ref, and m_ref only moves when a tile's max exceeds it by more than
//     GCD_ATTN_THR (P <= 2^THR, exact in fp16's range; O and the denominator are rescaled by the same
//     factor on that rare path), so the steady state is max3 + exp2 + cvt per score — no subtract;
//   * the denominator comes from 4 extra MFMAs against a block of ones instead of 32 adds;
//   * 3-stage K / V^T ring filled by LDS-DMA two tiles ahead, counted vmcnt(4), one barrier per tile.
// ------------------------------------------------------------------------------------------------
#define GCD_ATTN_THR 8.0f

template <bool PRESCALED>
__global__ __launch_bounds__(256, 2) void attn_spatial_kernel(const f16* __restrict__ qkv,
                                                              int64_t ld, const f16* __restrict__ vt,
                                                              int S_pad, f16* __restrict__ out,
                                                              int64_t ldo, int S, int heads,
                                                              int nqb, float c /* scale * log2(e) */) {
  // stage = [K tile 64 keys x 64 d | V^T tile 64 d x 64 keys], 8 KB each, three stages
  __shared__ __attribute__((aligned(16))) char smem[3 * 16384];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  // XCD-aware block order: every XCD owns a contiguous range of (frame, head, q-block) triples, so the
  // q-blocks of one (frame, head) share that XCD's L2 copy of its K / V (2.4 MB at 72 x 128 tokens).
  int qb, head, frame;
  {
    const int nblk = gridDim.x, bid = blockIdx.x;
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    qb = L % nqb;
    const int fh = L / nqb;
    head = fh % heads;
    frame = fh / heads;
  }
  const int C = heads * 64;
  const int q0 = qb * 128 + wave * 32;

  // Q^T fragments (B operand): query l31, d = 16 ks + 8 half + j
  f16x8 qf[4];
  {
    int qi = q0 + l31;
    qi = qi < S ? qi : S - 1;
    const f16* qp = qkv + ((int64_t)frame * S + qi) * ld + head * 64 + half * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const f16x8*)(qp + ks * 16);
  }

  // staging bookkeeping: piece p = i*256 + t -> row i*32 + (t>>3), physical chunk t&7
  const int prow = t >> 3;
  const int cl = (t & 7) ^ ((prow >> 1) & 7);
  const f16* kbase = qkv + (int64_t)frame * S * ld + C + head * 64 + cl * 8;
  const f16* vbase = vt + (((int64_t)frame * heads + head) * 64) * S_pad + cl * 8;
  const int ntiles = (S + 63) >> 6;

  auto stage = [&](int kt) {
    if (kt < ntiles) {
      const int kv0 = kt << 6;
      char* Ks = smem + (kt % 3) * 16384;
      char* Vs = Ks + 8192;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = prow + 32 * i;
        int key = kv0 + row;
        key = key < S ? key : S - 1;
        glds16(kbase + (int64_t)key * ld, Ks + (i * 256 + wave * 64) * 16);
        glds16(vbase + (int64_t)row * S_pad + kv0, Vs + (i * 256 + wave * 64) * 16);
      }
    }
  };

  // o0 / o1: O^T rows d = 0-31 / 32-63; o2: the softmax denominator (V^T row block of ones).
  f32x16 o0, o1, o2, nm;   // nm: -m_ref of this lane's query in all 16 registers (MFMA C operand)
#pragma unroll
  for (int r = 0; r < 16; ++r) o0[r] = o1[r] = o2[r] = nm[r] = 0.f;
  const f16x8 ones = {(f16)1.f, (f16)1.f, (f16)1.f, (f16)1.f, (f16)1.f, (f16)1.f, (f16)1.f, (f16)1.f};
  // per-lane LDS offsets of the K / V^T fragments within a stage (V^T = +8192, rows 32-63 = +4096)
  int foff[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) foff[ks] = lds_tile_off(l31, 2 * ks + half);

  stage(0);
  stage(1);
  for (int kt = 0; kt < ntiles; ++kt) {
    // tile kt has landed (my pieces; tile kt+1's 4 may still fly) -> barrier -> everybody's have,
    // and every wave is done reading tile kt-1, whose stage takes tile kt+2
    if (kt + 1 < ntiles) {
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    stage(kt + 2);
    const char* Ks = smem + (kt % 3) * 16384;
    const char* Vs = Ks + 8192;

    f16x8 kf[8];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      kf[2 * ks] = *(const f16x8*)(Ks + foff[ks]);
      kf[2 * ks + 1] = *(const f16x8*)(Ks + foff[ks] + 4096);
    }
    // ---- S^T - m_ref = K Q^T + nm : two 32-key sub-tiles ----
    f32x16 s0, s1;
    s0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[0], qf[0], nm, 0, 0, 0);
    s1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[1], qf[0], nm, 0, 0, 0);
#pragma unroll
    for (int ks = 1; ks < 4; ++ks) {
      s0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[2 * ks], qf[ks], s0, 0, 0, 0);
      s1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[2 * ks + 1], qf[ks], s1, 0, 0, 0);
    }
    // ---- V^T fragments: issued now, consumed after the softmax (their latency hides under it) ----
    f16x8 vf[8];   // vf[2*c2 + dt]: d rows 32 dt + l31, keys 16 c2 + {0-3, 8-11} + 4 half
#pragma unroll
    for (int c2 = 0; c2 < 4; ++c2) {
      vf[2 * c2] = *(const f16x8*)(Vs + foff[c2]);
      vf[2 * c2 + 1] = *(const f16x8*)(Vs + foff[c2] + 4096);
    }
    if (!PRESCALED) {
      // scores arrive in natural units: s <- s*c + nm*(1 - c)   (nm is already in exp2 units)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s0[r] = fmaf(s0[r] - nm[0], c, nm[0]);
        s1[r] = fmaf(s1[r] - nm[0], c, nm[0]);
      }
    }
    // register r of a sub-tile holds key (r&3) + 8*(r>>2) + 4*half
    if ((kt << 6) + 64 > S) {
      int kb = (kt << 6) + 4 * half;
      asm volatile("" : "+v"(kb));   // keeps the key-index adds inside this (last-tile-only) branch
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kb + (r & 3) + 8 * (r >> 2);
        if (key >= S) s0[r] = -INFINITY;
        if (key + 32 >= S) s1[r] = -INFINITY;
      }
    }
    // ---- tile max relative to m_ref (per query = per lane column, both halves) ----
    float mx = max3_f(s0[0], s1[0], s0[1]);
    mx = max3_f(mx, s1[1], s0[2]);
#pragma unroll
    for (int r = 2; r < 15; ++r) mx = max3_f(mx, s1[r], s0[r + 1]);
    mx = fmaxf(mx, s1[15]);
    {
      const unsigned mu = __float_as_uint(mx);
      const auto sw = __builtin_amdgcn_permlane32_swap(mu, mu, false, false);
      mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
    }
    // ---- rare path: move the reference (always on the first tile, which defines it) ----
    if (kt == 0 || __any(mx > GCD_ATTN_THR)) {
      const float delta = (kt == 0 || mx > GCD_ATTN_THR) ? mx : 0.f;
      const float alpha = kt == 0 ? 1.0f : __builtin_amdgcn_exp2f(-delta);   // O is still 0 on tile 0
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s0[r] -= delta;
        s1[r] -= delta;
        o0[r] *= alpha;
        o1[r] *= alpha;
        o2[r] *= alpha;
        nm[r] -= delta;
      }
    }
    f16x8 pf[4];  // pf[c2] = P^T fragment (B operand) for the 16 keys 16 c2 .. 16 c2 + 15
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      pf[r >> 3][r & 7] = (f16)__builtin_amdgcn_exp2f(s0[r]);
      pf[2 + (r >> 3)][r & 7] = (f16)__builtin_amdgcn_exp2f(s1[r]);
    }
    // ---- O^T += V^T P^T, denominator += 1 P^T ----
#pragma unroll
    for (int c2 = 0; c2 < 4; ++c2) {
      o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[2 * c2], pf[c2], o0, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[2 * c2 + 1], pf[c2], o1, 0, 0, 0);
      o2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ones, pf[c2], o2, 0, 0, 0);
    }
  }

  // ---- normalise and store: o{0,1}[r] is O[query l31][d = 32 dt + (r&3) + 8 (r>>2) + 4 half] ----
  const float inv = 1.0f / o2[0];
  const int qi = q0 + l31;
  if (qi < S) {
    f16* op = out + ((int64_t)frame * S + qi) * ldo + head * 64 + 4 * half;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f16x4 v0, v1;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v0[e] = (f16)(o0[4 * g + e] * inv);
        v1[e] = (f16)(o1[4 * g + e] * inv);
      }
      *(f16x4*)(op + 8 * g) = v0;
      *(f16x4*)(op + 32 + 8 * g) = v1;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Spatial flash attention, 64 queries per wave (GCD_TUNE_ATTN_IMPL = 2 / automatic for S >= 1024).
//
// Same arithmetic as attn_spatial_kernel, twice the queries per wave: the kernel above spends more
// issue slots on LDS fragment reads, LDS-DMA pieces, the barrier and the loop skeleton than on MFMA
// (profiles/r01_attention_ablation.txt), and all of those are per wave and tile, not per score — a
// wave that owns two 32-query blocks reads each K / V^T fragment once for both, issues the same four
// DMA pieces per tile for twice the scores and passes half as many barriers per score.  (Eight such
// waves per workgroup — 512 queries, half the DMA pieces per wave again — measured slower: 3484 vs
// 3334 us at 72 x 128 tokens; the barrier across 8 waves costs more than the DMA issue saves.)
// Register budget (256 at 2 waves per SIMD): O^T of both query blocks is 64 registers, Q 32, the
// scores of a tile 64.  The 16-register -m_ref and denominator blocks of the kernel above do not fit:
//   * the reference max is subtracted by a FIFTH k-step instead:  S^T += Kones (keys x 16, column 0
//     = 1)  x  Qm (16 x queries, row 0 = -m_ref), with m_ref kept exactly representable in fp16 (it
//     is only a reference point: it moves by the rounded-up tile excess, and O / the denominator are
//     rescaled by exactly the amount it moved);
//   * the denominator is summed from the fp16 P values the P V product consumes (v_dot2_f32_f16, two
//     per instruction) — and doubles as the overflow check: all p >= 0, so a per-lane tile sum <= 128
//     proves every p <= 2^7.  Only when some lane's sum is larger (or on the first tile, which defines
//     the reference) is the tile maximum computed at all; the steady state has no max pass.
// ------------------------------------------------------------------------------------------------
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

template <bool PRESCALED>
__global__ __launch_bounds__(256, 2) void attn_spatial64_kernel(const f16* __restrict__ qkv, int64_t ld,
                                                                const f16* __restrict__ vt, int S_pad,
                                                                f16* __restrict__ out, int64_t ldo,
                                                                int S, int heads, int nqb,
                                                                float c /* scale * log2(e) */) {
  __shared__ __attribute__((aligned(16))) char smem[3 * 16384];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  int qb, head, frame;
  {
    const int nblk = gridDim.x, bid = blockIdx.x;
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    qb = L % nqb;
    const int fh = L / nqb;
    head = fh % heads;
    frame = fh / heads;
  }
  const int C = heads * 64;
  const int q0 = qb * 256 + wave * 64;

  // Q^T fragments (B operand) of the two query blocks: query 32 b + l31, d = 16 ks + 8 half + j
  f16x8 qf[2][4];
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    int qi = q0 + 32 * b + l31;
    qi = qi < S ? qi : S - 1;
    const f16* qp = qkv + ((int64_t)frame * S + qi) * ld + head * 64 + half * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[b][ks] = *(const f16x8*)(qp + ks * 16);
  }
  const f16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
  f16x8 kones = zero8;            // A operand of the fifth k-step: k = 0 column of ones
  if (half == 0) kones[0] = (f16)1.f;
  f16x8 qm[2] = {zero8, zero8};   // B operand of the fifth k-step: row k = 0 holds -m_ref
  float mref[2] = {0.f, 0.f};
  float lsum[2] = {0.f, 0.f};     // this lane's share of the softmax denominators

  const int prow = t >> 3;
  const int cl = (t & 7) ^ ((prow >> 1) & 7);
  const f16* kbase = qkv + (int64_t)frame * S * ld + C + head * 64 + cl * 8;
  const f16* vbase = vt + (((int64_t)frame * heads + head) * 64) * S_pad + cl * 8;
  const int ntiles = (S + 63) >> 6;
  auto stage = [&](int kt) {
    if (kt < ntiles) {
      const int kv0 = kt << 6;
      char* Ks = smem + (kt % 3) * 16384;
      char* Vs = Ks + 8192;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = prow + 32 * i;
        int key = kv0 + row;
        key = key < S ? key : S - 1;
        glds16(kbase + (int64_t)key * ld, Ks + (i * 256 + wave * 64) * 16);
        glds16(vbase + (int64_t)row * S_pad + kv0, Vs + (i * 256 + wave * 64) * 16);
      }
    }
  };

  f32x16 o0[2], o1[2];
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) o0[b][r] = o1[b][r] = 0.f;
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const f16x2 one2 = {(f16)1.f, (f16)1.f};
  int foff[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) foff[ks] = lds_tile_off(l31, 2 * ks + half);

  stage(0);
  stage(1);
  for (int kt = 0; kt < ntiles; ++kt) {
    if (kt + 1 < ntiles) {
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    stage(kt + 2);
    // fragment addresses of this stage: four VALU adds, everything else is an immediate ds_read offset
    // (V^T = +8192, rows 32-63 = +4096); written with a pointer the compiler re-added the stage base to
    // every one of the 16 reads
    const int sbase = (kt % 3) * 16384;
    int fo[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) fo[ks] = foff[ks] + sbase;

    // ---- S^T - m_ref for both query blocks; K fragments of one 32-key half at a time ----
    f32x16 sc[2][2];   // [query block][key half]
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      f16x8 kf[4];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) kf[ks] = *(const f16x8*)(smem + fo[ks] + 4096 * kb);
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        sc[b][kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kones, qm[b], zero16, 0, 0, 0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
          sc[b][kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[ks], qf[b][ks], sc[b][kb], 0, 0, 0);
      }
    }
    f16x8 pfb[2][4];  // pfb[b][c2] = P^T fragment (B operand) of query block b, keys 16 c2 .. 16 c2 + 15
    f16x8 vf[8];      // vf[2*c2 + dt]: d rows 32 dt + l31, keys 16 c2 + {0-3, 8-11} + 4 half
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      f32x16& s0 = sc[b][0];
      f32x16& s1 = sc[b][1];
      if (!PRESCALED) {
        // scores arrive in natural units with -m_ref (exp2 units) already added
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          s0[r] = fmaf(s0[r] + mref[b], c, -mref[b]);
          s1[r] = fmaf(s1[r] + mref[b], c, -mref[b]);
        }
      }
      // register r of a key half holds key (r&3) + 8*(r>>2) + 4*half
      if ((kt << 6) + 64 > S) {
        int kb0 = (kt << 6) + 4 * half;
        asm volatile("" : "+v"(kb0));   // keeps the 32 key-index adds inside this (last-tile-only) branch
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kb0 + (r & 3) + 8 * (r >> 2);
          if (key >= S) s0[r] = -INFINITY;
          if (key + 32 >= S) s1[r] = -INFINITY;
        }
      }
      f16x8 (&pf)[4] = pfb[b];
      float lt;
      // block 0's P V product goes to the matrix pipe behind block 1's exponentials (its V^T fragments,
      // requested after softmax 0, have arrived by then); splitting the exponentials around it
      // measured no better in the full step (17.8 vs 17.7 ms of attention per step)
      auto pv0 = [&]() {
#pragma unroll
        for (int c2 = 0; c2 < 4; ++c2) {
          o0[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[2 * c2], pfb[0][c2], o0[0], 0, 0, 0);
          o1[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[2 * c2 + 1], pfb[0][c2], o1[0], 0, 0, 0);
        }
      };
      auto exponentiate = [&]() {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          pf[r >> 3][r & 7] = (f16)__builtin_amdgcn_exp2f(s0[r]);
          pf[2 + (r >> 3)][r & 7] = (f16)__builtin_amdgcn_exp2f(s1[r]);
        }
        lt = 0.f;
#pragma unroll
        for (int c2 = 0; c2 < 4; ++c2)
#pragma unroll
          for (int jj = 0; jj < 4; ++jj)
            lt = __builtin_amdgcn_fdot2(f16x2{pf[c2][2 * jj], pf[c2][2 * jj + 1]}, one2, lt, false);
      };
      exponentiate();
      if (b == 1) pv0();
      // ---- rare path: move the reference (always on the first tile, which defines it) ----
      if (kt == 0 || __any(!(lt <= 128.f))) {
        float mx = max3_f(s0[0], s1[0], s0[1]);
        mx = max3_f(mx, s1[1], s0[2]);
#pragma unroll
        for (int r = 2; r < 15; ++r) mx = max3_f(mx, s1[r], s0[r + 1]);
        mx = fmaxf(mx, s1[15]);
        {
          const unsigned mu = __float_as_uint(mx);
          const auto sw = __builtin_amdgcn_permlane32_swap(mu, mu, false, false);
          mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        }
        const bool mv = kt == 0 || mx > 7.0f;
        // new reference = fp16-exact value at or above the tile max; delta = how far it really moved
        const float want = mv ? mref[b] + ceilf(mx) : mref[b];
        const float nref = (float)(f16)want;
        const float delta = nref - mref[b];
        const float alpha = kt == 0 ? 1.0f : __builtin_amdgcn_exp2f(-delta);   // O, l still 0 on tile 0
        mref[b] = nref;
        if (half == 0) qm[b][0] = (f16)(-nref);
        lsum[b] *= alpha;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          s0[r] -= delta;
          s1[r] -= delta;
          o0[b][r] *= alpha;
          o1[b][r] *= alpha;
        }
        exponentiate();
      }
      lsum[b] += lt;
      if (b == 0) {
        // V^T fragments: issued between the two softmax passes, their latency hides under the second
#pragma unroll
        for (int c2 = 0; c2 < 4; ++c2) {
          vf[2 * c2] = *(const f16x8*)(smem + fo[c2] + 8192);
          vf[2 * c2 + 1] = *(const f16x8*)(smem + fo[c2] + 8192 + 4096);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- O^T += V^T P^T, query block 1 ----
#pragma unroll
    for (int c2 = 0; c2 < 4; ++c2) {
      o0[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[2 * c2], pfb[1][c2], o0[1], 0, 0, 0);
      o1[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[2 * c2 + 1], pfb[1][c2], o1[1], 0, 0, 0);
    }
  }

  // ---- normalise and store: o{0,1}[r] is O[query l31][d = 32 dt + (r&3) + 8 (r>>2) + 4 half] ----
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    float l = lsum[b];   // the two halves of a query column hold disjoint keys
    {
      const unsigned lu = __float_as_uint(l);
      const auto sw = __builtin_amdgcn_permlane32_swap(lu, lu, false, false);
      l = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    }
    const float inv = 1.0f / l;
    const int qi = q0 + 32 * b + l31;
    if (qi < S) {
      f16* op = out + ((int64_t)frame * S + qi) * ldo + head * 64 + 4 * half;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f16x4 v0, v1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v0[e] = (f16)(o0[b][4 * g + e] * inv);
          v1[e] = (f16)(o1[b][4 * g + e] * inv);
        }
        *(f16x4*)(op + 8 * g) = v0;
        *(f16x4*)(op + 32 + 8 * g) = v1;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Spatial flash attention, 64 queries per wave, SOFTWARE-PIPELINED (the product kernel for S >= 1024).
//
// What gfx950 overlaps (tools/issue_probe.cpp, profiles/r02k_issue_probe.txt; cycles per instruction of
// a wave64: MFMA 32x32x16 f16 32, v_exp_f32 8-9, plain VALU 4-5):
//   * a wave's OWN plain VALU / transcendental instructions issue in the shadow of its MFMA: 16 x {MFMA,
//     exp, exp} costs what 16 MFMAs cost, {MFMA, cvt, add, add, exp, exp} 33 cycles per group;
//   * ANOTHER wave's do not while the MFMA wave issues MFMAs back to back: an MFMA stream next to an
//     exp / fma / cvt stream on the same SIMD costs (nearly) the sum of the two;
//   * v_dot2(c)_f32_f16 and v_pk_*_f32 do not run under an MFMA at all (+18 cycles per group each).
// attn_spatial64_kernel above issues each wave's work as  S^T MFMAs | softmax VALU | P V MFMAs: by the
// rules above nothing overlaps inside a wave and little between the two waves of a SIMD (which belong to
// different workgroups): 2315 SIMD cycles per wave and tile for 1152 of MFMA + ~1400 of softmax — the
// serial sum.  Here the 18 MFMAs a softmax pass can cover are issued INSIDE it, one per two v_exp_f32:
//     softmax(block 0, tile kt)  covers  P V (block 1, tile kt-1)  +  S^T (block 1, tile kt)
//     softmax(block 1, tile kt)  covers  P V (block 0, tile kt)    +  S^T (block 0, tile kt+1)
// and the denominator is summed from the fp32 exponentials with v_add_f32 (two chains) instead of
// v_dot2 over the packed fp16 values (so it differs from the kernel above by the fp16 rounding of P:
// <= 2^-12 relative, unbiased).  The K and V^T fragments are read from LDS once per query block
// instead of once per tile (their registers are live for half a pass each), the ring has four stages and
// the barrier sits between the two passes (tile kt+1's K is needed from the second one on); the rare
// "reference moved" path is ONE branch (no short-circuit) laid out as unlikely: in its inline two-branch
// form the check cost a lone wave 27 % of its time (a taken branch by itself is ~44 cycles; the rest was
// scheduling the inline rare block took away from the common path).
// Measured (tools/attn_bench, same box): 72x128 tokens 3187 -> 3068 us, 36x64 495 -> 465 us; SQ counters
// of the new kernel: VALU issue 37 % of wave cycles (x 2 waves = 73 % of the SIMD), MFMA busy 51 %, 27 %
// parked on s_waitcnt / s_barrier.  The 64 v_exp_f32 per wave and tile (16 VALU-port cycles each by the
// SQ's accounting) are 60 % of that VALU time: the kernel is transcendental-bound, not matrix-bound.
// ------------------------------------------------------------------------------------------------
template <bool PRESCALED>
__global__ __launch_bounds__(256, 2) void attn_spatial64p_kernel(const f16* __restrict__ qkv, int64_t ld,
                                                                 const f16* __restrict__ vt, int S_pad,
                                                                 f16* __restrict__ out, int64_t ldo,
                                                                 int S, int heads, int nqb,
                                                                 float c /* scale * log2(e) */) {
  __shared__ __attribute__((aligned(16))) char smem[4 * 16384];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  int qb, head, frame;
  {
    const int nblk = gridDim.x, bid = blockIdx.x;
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    qb = L % nqb;
    const int fh = L / nqb;
    head = fh % heads;
    frame = fh / heads;
  }
  const int C = heads * 64;
  const int q0 = qb * 256 + wave * 64;

  f16x8 qf[2][4];
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    int qi = q0 + 32 * b + l31;
    qi = qi < S ? qi : S - 1;
    const f16* qp = qkv + ((int64_t)frame * S + qi) * ld + head * 64 + half * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[b][ks] = *(const f16x8*)(qp + ks * 16);
  }
  const f16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
  f16x8 kones = zero8;
  if (half == 0) kones[0] = (f16)1.f;
  f16x8 qm[2] = {zero8, zero8};
  float mref[2] = {0.f, 0.f};
  float lsum[2] = {0.f, 0.f};

  const int prow = t >> 3;
  const int cl = (t & 7) ^ ((prow >> 1) & 7);
  const f16* kbase = qkv + (int64_t)frame * S * ld + C + head * 64 + cl * 8;
  const f16* vbase = vt + (((int64_t)frame * heads + head) * 64) * S_pad + cl * 8;
  const int ntiles = (S + 63) >> 6;
  auto stage = [&](int kt) {
    if (kt < ntiles) {
      const int kv0 = kt << 6;
      char* Ks = smem + (kt & 3) * 16384;
      char* Vs = Ks + 8192;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = prow + 32 * i;
        int key = kv0 + row;
        key = key < S ? key : S - 1;
        glds16(kbase + (int64_t)key * ld, Ks + (i * 256 + wave * 64) * 16);
        glds16(vbase + (int64_t)row * S_pad + kv0, Vs + (i * 256 + wave * 64) * 16);
      }
    }
  };

  f32x16 o0[2], o1[2];
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) o0[b][r] = o1[b][r] = 0.f;
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int foff[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) foff[ks] = lds_tile_off(l31, 2 * ks + half);

  f32x16 sc[2][2];   // [query block][key half]: S^T - m_ref of the tile the block works on
  f16x8 pfb[2][4];   // P^T fragments (B operand) per query block; zero = "no tile yet" for the first P V
  f16x8 vf[8];       // V^T fragments of the tile whose P V products are being issued
  f16x8 kf[2][4];    // K fragments of the tile whose S^T products are being issued
#pragma unroll
  for (int i = 0; i < 4; ++i) pfb[0][i] = pfb[1][i] = zero8;
#pragma unroll
  for (int i = 0; i < 8; ++i) vf[i] = zero8;

  auto load_k = [&](int kt) {
    const int sbase = (kt & 3) * 16384;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) kf[kb][ks] = *(const f16x8*)(smem + foff[ks] + sbase + 4096 * kb);
  };
  auto load_v = [&](int kt) {
    const int sbase = (kt & 3) * 16384 + 8192;
#pragma unroll
    for (int c2 = 0; c2 < 4; ++c2) {
      vf[2 * c2] = *(const f16x8*)(smem + foff[c2] + sbase);
      vf[2 * c2 + 1] = *(const f16x8*)(smem + foff[c2] + sbase + 4096);
    }
  };
  // MFMA i (0..9) of S^T - m_ref for query block b: the two key halves alternate (dependent MFMAs are
  // two issues apart); step 0 of a half is the reference-max k-step
  auto s_mfma = [&](const int b, const int i) {
    const int kb = i & 1, st = i >> 1;
    if (st == 0) sc[b][kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kones, qm[b], zero16, 0, 0, 0);
    else sc[b][kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[kb][st - 1], qf[b][st - 1], sc[b][kb], 0, 0, 0);
  };
  // MFMA i (0..7) of O^T += V^T P^T for query block b
  auto pv_mfma = [&](const int b, const int i) {
    const int c2 = i >> 1;
    if (i & 1) o1[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[2 * c2 + 1], pfb[b][c2], o1[b], 0, 0, 0);
    else o0[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[2 * c2], pfb[b][c2], o0[b], 0, 0, 0);
  };
  // Covered MFMA i (0..17) of a softmax pass over block b: P V of the OTHER block (0..7), then its S^T
  // (8..17).  The empty volatile asm "rewrites" an operand of that MFMA together with t0 / t1 /
  // t2 (the scores the next two exponentials read and the last exponential's result): the MFMA and those
  // exponentials can only be emitted after it, and it only after the previous exponentials.  Without
  // the pin LLVM hoists the 18 MFMAs of a pass into one cluster in front of the exponentials (they do
  // not depend on each other) — the un-pipelined kernel again.
#define ATT_COVERED(b, i, t0, t1, t2, t3)                                                            \
  do {                                                                                             \
    if ((i) < 8) {                                                                                 \
      asm volatile("" : "+v"(vf[(i)&7]), "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3));                            \
      pv_mfma(1 - (b), (i));                                                                       \
    } else if ((i) < 10) {                                                                         \
      asm volatile("" : "+v"(qm[1 - (b)]), "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3));                          \
      s_mfma(1 - (b), (i) - 8);                                                                    \
    } else {                                                                                       \
      asm volatile("" : "+v"(kf[(i) & 1][((((i) - 8) >> 1) - 1) & 3]), "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3)); \
      s_mfma(1 - (b), (i) - 8);                                                                    \
    }                                                                                              \
  } while (0)

  // softmax numerators of query block b for tile kt (scores in sc[b]) -> pfb[b], lsum[b], with the 18
  // covered MFMAs issued between the exponentials; `extra(i)` runs behind MFMA i (LDS fragment reads)
  auto softmax = [&](const int b, const int kt, auto&& extra) {
    f32x16& s0 = sc[b][0];
    f32x16& s1 = sc[b][1];
    if (!PRESCALED) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s0[r] = fmaf(s0[r] + mref[b], c, -mref[b]);
        s1[r] = fmaf(s1[r] + mref[b], c, -mref[b]);
      }
    }
    if (__builtin_expect((kt << 6) + 64 > S, 0)) {
      int kb0 = (kt << 6) + 4 * half;
      asm volatile("" : "+v"(kb0));
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kb0 + (r & 3) + 8 * (r >> 2);
        if (key >= S) s0[r] = -INFINITY;
        if (key + 32 >= S) s1[r] = -INFINITY;
      }
    }
    f16x8 (&pf)[4] = pfb[b];
    // Every covered MFMA gets its share of the pass's VALU work issued right behind it: two
    // exponentials (step r), two adds into the denominator (step r - 1) and one fp16 pack (steps r - 2 /
    // r - 3) — 28 cycles of plain VALU in the 32-cycle shadow of the MFMA.  Measured rules of gfx950
    // (tools/issue_probe.cpp): a wave's own plain VALU / transcendental instructions run under its
    // MFMA, ANOTHER wave's do not (two waves of a SIMD cost the sum of their streams), and v_dot2 /
    // v_pk_*_f32 do not run under an MFMA at all — so the denominator is summed from the fp32
    // exponentials with v_add_f32 (two chains), not from the packed fp16 values with v_dot2.
    float lt0 = 0.f, lt1 = 0.f;
    float e0[16], e1[16];   // exponentials of key half 0 / 1 (a handful live at any time)
    float dm = 0.f;         // filler for the pins that have fewer than four values to tie
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < 18; ++r) {
      if (r == 0) {
        ATT_COVERED(b, r, s0[0], s1[0], lt0, dm);
      } else if (r < 16) {
        ATT_COVERED(b, r, s0[r], s1[r], e1[r - 1], lt0);
      } else if (r == 16) {
        ATT_COVERED(b, r, e0[14], e0[15], e1[15], lt0);
      } else {
        ATT_COVERED(b, r, e1[14], e1[15], lt0, dm);
      }
      extra(r);
      if (r >= 1 && r <= 16) {
        lt0 += e0[r - 1];
        lt1 += e1[r - 1];
      }
      // steps (q, q + 1), q even, are complete after step q + 1: their key-half-0 pair is packed behind
      // MFMA q + 2, their key-half-1 pair behind MFMA q + 3
      if (r >= 2 && !(r & 1)) {
        const int q = r - 2;
        pf[q >> 3][q & 7] = (f16)e0[q];
        pf[q >> 3][(q & 7) + 1] = (f16)e0[q + 1];
      }
      if (r >= 3 && (r & 1)) {
        const int q = r - 3;
        pf[2 + (q >> 3)][q & 7] = (f16)e1[q];
        pf[2 + (q >> 3)][(q & 7) + 1] = (f16)e1[q + 1];
      }
      if (r < 16) {
        e0[r] = __builtin_amdgcn_exp2f(s0[r]);
        e1[r] = __builtin_amdgcn_exp2f(s1[r]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    float lt = lt0 + lt1;
    // ---- rare path: move the reference (always on the first tile, which defines it) ----
    // (one branch, laid out as unlikely: the common path falls through; see the kernel's header)
    const bool moved = (kt == 0) | (__builtin_amdgcn_ballot_w64(!(lt <= 128.f)) != 0);
    if (__builtin_expect(moved, 0)) {
      float mx = max3_f(s0[0], s1[0], s0[1]);
      mx = max3_f(mx, s1[1], s0[2]);
#pragma unroll
      for (int r = 2; r < 15; ++r) mx = max3_f(mx, s1[r], s0[r + 1]);
      mx = fmaxf(mx, s1[15]);
      {
        const unsigned mu = __float_as_uint(mx);
        const auto sw = __builtin_amdgcn_permlane32_swap(mu, mu, false, false);
        mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
      }
      const bool mv = kt == 0 || mx > 7.0f;
      const float want = mv ? mref[b] + ceilf(mx) : mref[b];
      const float nref = (float)(f16)want;
      const float delta = nref - mref[b];
      const float alpha = kt == 0 ? 1.0f : __builtin_amdgcn_exp2f(-delta);
      mref[b] = nref;
      if (half == 0) qm[b][0] = (f16)(-nref);
      lsum[b] *= alpha;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s0[r] -= delta;
        s1[r] -= delta;
        o0[b][r] *= alpha;
        o1[b][r] *= alpha;
      }
      lt = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float x0 = __builtin_amdgcn_exp2f(s0[r]), x1 = __builtin_amdgcn_exp2f(s1[r]);
        pf[r >> 3][r & 7] = (f16)x0;
        pf[2 + (r >> 3)][r & 7] = (f16)x1;
        lt += x0 + x1;
      }
    }
    lsum[b] += lt;
    __builtin_amdgcn_sched_barrier(0);
  };

  stage(0);
  stage(1);
  stage(2);
  if (ntiles >= 3) {
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  } else if (ntiles == 2) {
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  load_k(0);
#pragma unroll
  for (int i = 0; i < 10; ++i) s_mfma(0, i);   // S^T of block 0, tile 0: nothing to hide it under yet

  for (int kt = 0; kt < ntiles; ++kt) {
    // ---- softmax (0, kt) over: P V (1, kt-1) [vf = V^T of tile kt-1, zero P on the first tile], S^T (1, kt) ----
    softmax(0, kt, [&](const int i) {
      if (i == 7) load_k(kt);
    });
    load_v(kt);   // for P V (0, kt) below (after the pass: its rare path needs the registers)
    // tile kt+1 has landed for every wave (its K is read below); the stage tile kt+3 goes to held
    // tile kt-1, whose last reader (the V^T fragments of the pass above) is behind this barrier
    if (kt + 2 < ntiles) {
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    stage(kt + 3);
    // ---- softmax (1, kt) over: P V (0, kt), S^T (0, kt+1) ----
    softmax(1, kt, [&](const int i) {
      if (i == 7) load_k(kt + 1);   // (past the last tile: stale LDS, those scores are never used)
    });
    load_v(kt);   // again, for P V (1, kt) in the next pass / after the loop
  }
  // P V (1, last tile)
#pragma unroll
  for (int i = 0; i < 8; ++i) pv_mfma(1, i);

  // ---- normalise and store: o{0,1}[r] is O[query l31][d = 32 dt + (r&3) + 8 (r>>2) + 4 half] ----
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    float l = lsum[b];
    {
      const unsigned lu = __float_as_uint(l);
      const auto sw = __builtin_amdgcn_permlane32_swap(lu, lu, false, false);
      l = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    }
    const float inv = 1.0f / l;
    const int qi = q0 + 32 * b + l31;
    if (qi < S) {
      f16* op = out + ((int64_t)frame * S + qi) * ldo + head * 64 + 4 * half;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f16x4 v0, v1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v0[e] = (f16)(o0[b][4 * g + e] * inv);
          v1[e] = (f16)(o1[b][4 * g + e] * inv);
        }
        *(f16x4*)(op + 8 * g) = v0;
        *(f16x4*)(op + 32 + 8 * g) = v1;
      }
    }
  }
}

extern "C" int gcd_attn_spatial_f16(const void* qkv, int64_t ld, const void* vt, int S_pad,
                                    void* out, int64_t ldo, int frames, int S, int heads,
                                    int q_prescaled, void* stream) {
  GCD_CHECK_ARG(qkv && vt && out, "gcd_attn_spatial_f16: null pointer");
  GCD_CHECK_ARG(frames > 0 && S > 0 && heads > 0, "gcd_attn_spatial_f16: empty problem");
  GCD_CHECK_ARG(S_pad % 64 == 0 && S_pad >= S, "gcd_attn_spatial_f16: S_pad=%d for S=%d", S_pad, S);
  GCD_CHECK_ARG(ld % 8 == 0 && ld >= 3 * heads * 64 && ldo % 4 == 0 && ldo >= heads * 64,
                "gcd_attn_spatial_f16: ld=%lld ldo=%lld", (long long)ld, (long long)ldo);
  const float c = 0.125f * LOG2E_F;  // head dim 64 -> scale 1/8, in exp2 units
  hipStream_t st = (hipStream_t)stream;
  // 64 queries per wave (256 per workgroup) unless the frame is too small to fill such blocks
  const int impl = gcd_tune_get(GCD_TUNE_ATTN_IMPL);
  // (measured: 72x128 tokens 4245 -> 3483 us, 36x64 553 -> 513 us; 18x32 = 576 tokens would waste a
  //  quarter of its 256-query blocks and runs 92 vs 109 us on the 128-query kernel)
  // (impl 2 = the un-pipelined 64-query kernel, kept for A/B runs of tools/attn_bench; 3 / automatic = the
  //  software-pipelined one: 72x128 tokens 3187 -> 3068 us, 36x64 495 -> 465 us on the same box)
  const bool wide = impl == 2 || impl == 3 || (impl != 1 && S >= 1024);
  const int qpb = wide ? 256 : 128;
  const int nqb = (S + qpb - 1) / qpb;
  const int64_t nblk = (int64_t)nqb * heads * frames;
  GCD_CHECK_ARG(nblk < (1ll << 31), "gcd_attn_spatial_f16: grid too large");
  const dim3 grid((unsigned)nblk), block(256);
  if (wide && impl != 2) {
    if (q_prescaled)
      hipLaunchKernelGGL(attn_spatial64p_kernel<true>, grid, block, 0, st, (const f16*)qkv, ld,
                         (const f16*)vt, S_pad, (f16*)out, ldo, S, heads, nqb, c);
    else
      hipLaunchKernelGGL(attn_spatial64p_kernel<false>, grid, block, 0, st, (const f16*)qkv, ld,
                         (const f16*)vt, S_pad, (f16*)out, ldo, S, heads, nqb, c);
  } else if (wide) {
    if (q_prescaled)
      hipLaunchKernelGGL(attn_spatial64_kernel<true>, grid, block, 0, st, (const f16*)qkv, ld,
                         (const f16*)vt, S_pad, (f16*)out, ldo, S, heads, nqb, c);
    else
      hipLaunchKernelGGL(attn_spatial64_kernel<false>, grid, block, 0, st, (const f16*)qkv, ld,
                         (const f16*)vt, S_pad, (f16*)out, ldo, S, heads, nqb, c);
  } else if (q_prescaled)
    hipLaunchKernelGGL(attn_spatial_kernel<true>, grid, block, 0, st, (const f16*)qkv, ld,
                       (const f16*)vt, S_pad, (f16*)out, ldo, S, heads, nqb, c);
  else
    hipLaunchKernelGGL(attn_spatial_kernel<false>, grid, block, 0, st, (const f16*)qkv, ld,
                       (const f16*)vt, S_pad, (f16*)out, ldo, S, heads, nqb, c);
  GCD_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Temporal attention: one (clip, pixel, head) problem per 16 lanes; lane i owns query frame i.
// ------------------------------------------------------------------------------------------------
#define TPROB 16                     // problems per workgroup
#define TP_STRIDE(T) ((T) * 128 + 16)  // bytes per problem in LDS (+16: spread problems over banks)

__global__ __launch_bounds__(256) void attn_temporal_kernel(const f16* __restrict__ qkv, int64_t ld,
                                                            f16* __restrict__ out, int64_t ldo,
                                                            int64_t nprob, int T, int HW, int heads) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x;
  const int p = t >> 4, i = t & 15;
  const int pstride = TP_STRIDE(T);
  char* Kp = smem + p * pstride;
  char* Vp = smem + TPROB * pstride + p * pstride;
  const int64_t pid = (int64_t)blockIdx.x * TPROB + p;
  const bool valid = pid < nprob && i < T;
  const int C = heads * 64;
  int h = 0;
  int64_t row = 0;
  if (valid) {
    h = (int)(pid % heads);
    const int64_t ps = pid / heads;
    const int s = (int)(ps % HW);
    const int64_t b = ps / HW;
    row = (b * T + i) * HW + s;
  }
  // K / V rows -> LDS with a cooperative, coalesced map: thread t always serves problem (t >> 3) & 15
  // and 16-byte chunk t & 7, and walks the 2 T (k|v, frame) rows with stride 2 — consecutive
  // threads read consecutive 16-byte chunks of one row, and consecutive problems (= heads of the same
  // pixel) are adjacent in memory, so a wave-instruction covers whole 128-byte lines instead of 64
  // scattered 16-byte pieces.
  {
    const int lp = (t >> 3) & 15, cc = t & 7;
    const int64_t lpid = (int64_t)blockIdx.x * TPROB + lp;
    if (lpid < nprob) {
      const int lh = (int)(lpid % heads);
      const int64_t lps = lpid / heads;
      const int ls = (int)(lps % HW);
      const int64_t lb = lps / HW;
      const f16* base = qkv + ((lb * T) * HW + ls) * ld + lh * 64 + cc * 8;
      char* kdst = smem + lp * pstride + cc * 16;
      char* vdst = smem + TPROB * pstride + lp * pstride + cc * 16;
      f16x8 tmp[16];   // all loads in flight before the first LDS write (T <= 16 -> <= 16 rows each)
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int wi = (t >> 7) + 2 * j;
        const int which = wi >= T ? 1 : 0;          // 0: k rows, 1: v rows
        const int fi = wi - which * T;
        if (wi < 2 * T) tmp[j] = *(const f16x8*)(base + (int64_t)fi * HW * ld + (which + 1) * C);
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int wi = (t >> 7) + 2 * j;
        const int which = wi >= T ? 1 : 0;
        const int fi = wi - which * T;
        if (wi < 2 * T) *(f16x8*)((which ? vdst : kdst) + fi * 128) = tmp[j];
      }
    }
  }
  f16x8 q[8];
  if (valid) {
    const f16* src = qkv + row * ld + h * 64;
#pragma unroll
    for (int cc = 0; cc < 8; ++cc) q[cc] = *(const f16x8*)(src + cc * 8);
  } else {
#pragma unroll
    for (int cc = 0; cc < 8; ++cc) q[cc] = (f16x8){0, 0, 0, 0, 0, 0, 0, 0};
  }
  __syncthreads();
  if (!valid) return;

  float sc[16];
  float mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    sc[j] = -INFINITY;
    if (j < T) {
      float a = 0.f;
#pragma unroll
      for (int cc = 0; cc < 8; ++cc) {
        const f16x8 kk = *(const f16x8*)(Kp + j * 128 + cc * 16);
#pragma unroll
        for (int e = 0; e < 4; ++e)
          a = __builtin_amdgcn_fdot2((f16x2){q[cc][2 * e], q[cc][2 * e + 1]},
                                     (f16x2){kk[2 * e], kk[2 * e + 1]}, a, false);
      }
      sc[j] = a * 0.125f;
      mx = fmaxf(mx, sc[j]);
    }
  }
  float l = 0.f;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const float e = (j < T) ? __expf(sc[j] - mx) : 0.f;
    sc[j] = e;
    l += e;
  }
  const float inv = 1.0f / l;
  float o[64];
#pragma unroll
  for (int d = 0; d < 64; ++d) o[d] = 0.f;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    if (j < T) {
      const float pj = sc[j] * inv;
#pragma unroll
      for (int cc = 0; cc < 8; ++cc) {
        const f16x8 vv = *(const f16x8*)(Vp + j * 128 + cc * 16);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[cc * 8 + e] = fmaf(pj, (float)vv[e], o[cc * 8 + e]);
      }
    }
  }
  f16* dst = out + row * ldo + h * 64;
#pragma unroll
  for (int cc = 0; cc < 8; ++cc) {
    f16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (f16)o[cc * 8 + e];
    *(f16x8*)(dst + cc * 8) = v;
  }
}

// ------------------------------------------------------------------------------------------------
// Temporal attention on the matrix pipe (round 4): one (clip, pixel, head) problem per WAVE — T <= 16 frames are one
// 16 x 16 MFMA tile.  The kernel above does the two products of a problem with 16 lanes of v_dot2 / v_fma reading K and
// V from LDS once per query (575 VALU + 56 ds_read_b128 wave-instructions per problem: VALU- and LDS-bound at 3 TB/s
// on a 660 MB launch); here a problem costs ~100 wave-instructions and the launch streams:
//   S^T = K Q^T   2 x v_mfma_f32_16x16x32_f16, operands straight from global memory: lane l holds row l & 15 (a frame),
//                 head channels 8 (l >> 4) .. + 7 (+ 32 for the second K step) of K resp. Q — the MFMA A / B layouts ARE
//                 16-byte row pieces, no LDS;
//   softmax       S^T arrives as [key 4 (l >> 4) + e][query l & 15]: 4 scores per lane, the rest of a query's column sits
//                 in lanes l ^ 16, l ^ 32, l ^ 48 (v_permlane16_swap / v_permlane32_swap); 4 v_exp_f32 per lane;
//   O^T = V^T P^T 4 (x 2) x v_mfma_f32_16x16x16_f16: P^T in the accumulator layout IS the B operand; V^T (A operand:
//                 lane = channel, 4 consecutive keys) is the one transposed read — V rows go through a wave-private
//                 2.3 KB LDS tile (144-byte rows) and come back with ds_read_u16.  P enters as fp16 hi + fp16 lo
//                 (two MFMAs per channel block, the pipe is idle anyway): the fp32 probabilities of the kernel above
//                 to 2^-22, no new rounding on the path;
//   store         O^T is [channel 16 b + 4 (l >> 4) + e][query l & 15]: four 8-byte pieces per lane, 32 contiguous bytes
//                 per (frame row, 16-channel block).
// A wave walks problems pid, pid + #waves, ... with the next problem's six 16-byte loads issued before the current
// one's arithmetic.  Rows >= T (T = 14: two of sixteen) load a clamped row and are masked.
// ------------------------------------------------------------------------------------------------
#define TM_ROW 144   // bytes per V row in LDS: 128 + 16 pad (the four key quads of a ds_read_u16 land 16 banks apart)

// PF: problems of prefetch (the loads of problem p + PF are issued before the arithmetic of problem p).  Measured and
// not used: PF = 2 (101.19 vs 101.16 ms per step, profiles/r04p_ab_sweep_nontemporal.txt) and non-temporal loads / stores
// (+0.6 ms per step, profiles/r04q_*: the q|k|v tensor was just written by the GEMM before and partly sits in the caches).
template <int PF>
__global__ __launch_bounds__(256) void attn_temporal_mfma_kernel(const f16* __restrict__ qkv, int64_t ld,
                                                                 f16* __restrict__ out, int64_t ldo,
                                                                 int64_t nprob, int T, int HW, int heads) {
  __shared__ __attribute__((aligned(16))) char vs_all[4 * 16 * TM_ROW];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  char* const vs = vs_all + wave * 16 * TM_ROW;
  const int r15 = lane & 15, q = lane >> 4;
  const int C = heads * 64;
  const int fr = r15 < T ? r15 : T - 1;           // the frame this lane loads (clamped: always a valid row)
  const int nwaves = (int)gridDim.x * 4;
  const int pid0 = __builtin_amdgcn_readfirstlane((int)blockIdx.x * 4 + wave);
  if (pid0 >= nprob) return;
  // problem = (clip b, pixel s, head h), h fastest: walked as a mixed-radix counter (one division set per wave, none per
  // problem): (h, s, b) of pid0 and the digits of the stride
  int ph = pid0 % heads, ps = (pid0 / heads) % HW, pb = pid0 / heads / HW;
  const int dh = nwaves % heads, dsx = (nwaves / heads) % HW, db = nwaves / heads / HW;
  const int64_t frow = (int64_t)fr * HW;
  const f16* const lane_base = qkv + 8 * q;
  auto advance = [&](int& h, int& s2, int& b) {
    h += dh;
    int c = h >= heads ? 1 : 0;
    h -= c * heads;
    s2 += dsx + c;
    c = s2 >= HW ? 1 : 0;
    s2 -= c * HW;
    b += db + c;
  };

  f16x8 kf[PF][2], qf[PF][2], vf[PF][2];   // slot 0 = the current problem, slot j = j problems ahead
  auto load = [&](int h, int s2, int b, f16x8 (&kk)[2], f16x8 (&qq)[2], f16x8 (&vv)[2]) {
    const int64_t row = (int64_t)b * T * HW + s2 + frow;
    const f16* src = lane_base + row * ld + h * 64;
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2) {
      qq[k2] = *(const f16x8*)(src + 32 * k2);
      kk[k2] = *(const f16x8*)(src + C + 32 * k2);
      vv[k2] = *(const f16x8*)(src + 2 * C + 32 * k2);
    }
  };
  // (fh, fs, fb): the problem PF ahead of the current one — the next to load
  int fh = ph, fs = ps, fb = pb;
#pragma unroll
  for (int j = 0; j < PF; ++j) {
    if ((int64_t)pid0 + (int64_t)j * nwaves < nprob) load(fh, fs, fb, kf[j], qf[j], vf[j]);
    advance(fh, fs, fb);
  }
  const float cs = 0.125f * 1.4426950408889634f;   // 1 / sqrt(64) in exp2 units
  for (int64_t pid = pid0; pid < nprob; pid += nwaves) {
    f16x8 kn[2], qn[2], vn[2];
    const bool more = pid + (int64_t)PF * nwaves < nprob;
    if (more) load(fh, fs, fb, kn, qn, vn);
    advance(fh, fs, fb);
    f16x8 (&kc)[2] = kf[0];
    f16x8 (&qc)[2] = qf[0];
    f16x8 (&vc)[2] = vf[0];
    // ---- V rows -> the wave's LDS tile (row = frame, 128 B + pad) ----
    *(f16x8*)(vs + r15 * TM_ROW + 16 * q) = vc[0];
    *(f16x8*)(vs + r15 * TM_ROW + 64 + 16 * q) = vc[1];
    // (other lanes read these rows back below: pin the LDS store -> load order inside the wave for any compiler
    //  version — no instruction is emitted for either)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- S^T = K Q^T ----
    f32x4 st = {0.f, 0.f, 0.f, 0.f};
    st = __builtin_amdgcn_mfma_f32_16x16x32_f16(kc[0], qc[0], st, 0, 0, 0);
    st = __builtin_amdgcn_mfma_f32_16x16x32_f16(kc[1], qc[1], st, 0, 0, 0);
    // ---- softmax over the keys 4 q + e of query r15 ----
    float sc[4];
    float mx = -INFINITY;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      sc[e] = (4 * q + e < T) ? st[e] : -INFINITY;
      mx = fmaxf(mx, sc[e]);
    }
    {
      const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
      mx = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
      const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
      mx = fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
    }
    float pe[4], l = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      pe[e] = __builtin_amdgcn_exp2f((sc[e] - mx) * cs);   // exp2(-inf) = 0 for the masked keys
      l += pe[e];
    }
    {
      const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(l), __float_as_uint(l), false, false);
      l = __uint_as_float(a[0]) + __uint_as_float(a[1]);
      const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(l), __float_as_uint(l), false, false);
      l = __uint_as_float(b[0]) + __uint_as_float(b[1]);
    }
    const float inv = 1.0f / l;
    f16x4 phi, plo;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float pv = pe[e] * inv;
      phi[e] = (f16)pv;
      plo[e] = (f16)(pv - (float)phi[e]);
    }
    // ---- O^T = V^T P^T, 16 channels at a time ----
    f16* const dst = out + ((int64_t)pb * T * HW + ps + frow) * ldo + ph * 64 + 4 * q;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      f16x4 va;
#pragma unroll
      for (int e = 0; e < 4; ++e) va[e] = *(const f16*)(vs + (4 * q + e) * TM_ROW + (16 * b + r15) * 2);
      f32x4 o = {0.f, 0.f, 0.f, 0.f};
      o = __builtin_amdgcn_mfma_f32_16x16x16f16(va, phi, o, 0, 0, 0);
      o = __builtin_amdgcn_mfma_f32_16x16x16f16(va, plo, o, 0, 0, 0);
      if (r15 < T) {
        f16x4 ov;
#pragma unroll
        for (int e = 0; e < 4; ++e) ov[e] = (f16)o[e];
        *(f16x4*)(dst + 16 * b) = ov;
      }
    }
    advance(ph, ps, pb);
    // the tile's reads are done before the next problem's V rows overwrite it
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // rotate the slots: j + 1 -> j, the fresh loads -> PF - 1
#pragma unroll
    for (int j = 0; j + 1 < PF; ++j)
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) {
        kf[j][k2] = kf[j + 1][k2];
        qf[j][k2] = qf[j + 1][k2];
        vf[j][k2] = vf[j + 1][k2];
      }
    if (more) {
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) {
        kf[PF - 1][k2] = kn[k2];
        qf[PF - 1][k2] = qn[k2];
        vf[PF - 1][k2] = vn[k2];
      }
    }
  }
}

extern "C" int gcd_attn_temporal_f16(const void* qkv, int64_t ld, void* out, int64_t ldo, int clips,
                                     int T, int HW, int heads, void* stream) {
  GCD_CHECK_ARG(qkv && out, "gcd_attn_temporal_f16: null pointer");
  GCD_CHECK_ARG(clips > 0 && HW > 0 && heads > 0, "gcd_attn_temporal_f16: empty problem");
  GCD_CHECK_ARG(T >= 1 && T <= 16, "gcd_attn_temporal_f16: T=%d (supported: 1..16 frames)", T);
  GCD_CHECK_ARG(ld % 8 == 0 && ld >= 3 * heads * 64 && ldo % 8 == 0 && ldo >= heads * 64,
                "gcd_attn_temporal_f16: ld=%lld ldo=%lld", (long long)ld, (long long)ldo);
  const int64_t nprob = (int64_t)clips * HW * heads;
  if (gcd_tune_get(GCD_TUNE_ATTN_IMPL) != 16 && nprob < (1ll << 31) - 8192) {
    // one problem per wave on the matrix pipe; 7 workgroups (28 waves of 72 registers) per CU walk the problems.
    // GCD_TUNE_ATTN_IMPL = 16 keeps the 16-lanes-per-problem VALU kernel below (A/B, tests).
    int64_t mblocks = (nprob + 3) / 4;
    if (mblocks > 1792) mblocks = 1792;
    hipLaunchKernelGGL(attn_temporal_mfma_kernel<1>, dim3((unsigned)mblocks), dim3(256), 0, (hipStream_t)stream,
                       (const f16*)qkv, ld, (f16*)out, ldo, nprob, T, HW, heads);
    GCD_CHECK_LAUNCH();
    return 0;
  }
  const int64_t blocks = (nprob + TPROB - 1) / TPROB;
  GCD_CHECK_ARG(blocks < (1ll << 31), "gcd_attn_temporal_f16: grid too large");
  const int smem = 2 * TPROB * TP_STRIDE(T);
  static GcdPerDeviceOnce attr_once;
  GCD_CHECK_HIP(attr_once.opt_in((const void*)attn_temporal_kernel, 2 * TPROB * TP_STRIDE(16)));
  hipLaunchKernelGGL(attn_temporal_kernel, dim3((unsigned)blocks), dim3(256), smem,
                     (hipStream_t)stream, (const f16*)qkv, ld, (f16*)out, ldo, nprob, T, HW, heads);
  GCD_CHECK_LAUNCH();
  return 0;
}
