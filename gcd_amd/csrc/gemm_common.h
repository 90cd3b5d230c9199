// gemm_common.h — kernel-side view of gcd_gemm_desc shared by the GEMM kernels of libgcd_amd
// (gemm.hip: general 128-row tiles; gemm_pp.hip: 256x320 ping-pong tiles).
#pragma once
#include "common.h"

struct GemmK {
  const f16* A;
  const f16* W;
  void* out;
  int64_t lda, ldo;
  int M, N, K;
  int Cin, Hi, Wi, Ho, Wo, stride, up, T, HW;
  int asym;   // CONV3X3 stride 2 with (0,1,0,1) padding: taps at 2y + ky, ky = 0..2
  const float* bias;
  const float* rowvec;
  int64_t ld_rowvec;
  int rows_per_vec;
  const float* R1;
  int64_t ldr1;
  const float* R2;
  int64_t ldr2;
  float s_acc, s_r1, s_r2;
  const float* frame_alpha;
  int rows_per_alpha;
  int r1_blend;
  int out_kind;
  const f16* zero;
  int tiles_m, tiles_n;
  // fused LayerNorm of the output rows (gemm_pp.hip, N == 320)
  f16* ln_out;
  int64_t ld_ln_out;
  const float* ln_gamma;
  const float* ln_beta;
  float ln_eps;
  int ln_rows_per_vec;
  const float* ln_addvec;
  int64_t ld_ln_addvec;
  float* ln_sum_out;
  int64_t ld_ln_sum;
  // split-K (gemm_pp.hip): workgroup L works on tile L / splitk, K slice L % splitk, and stores raw
  // fp32 partial sums at out + slice * split_stride (elements)
  int splitk;
  int64_t split_stride;
  // per-64-row column sums / sums of squares of the fp32 output for the next GroupNorm
  // (gcd_gemm_desc.colstats): [2 * M / 64, N]
  float* colstats;
  int out_blocked, a_blocked;   // tile-blocked GEGLU hidden tensor (gcd_gemm_desc.out_blocked / a_blocked)
  int operand_bf16;             // A and W are bfloat16 (general kernel, PLAIN mode, fp32 output)
  // gemm_p8.hip: operand extents in bytes (buffer descriptors), frames of a conv input
  uint32_t a_bytes, w_bytes;
  int a_frames;
  // gcd_gemm_desc.sched (gemm_p8.hip): bit 0 = every XCD walks its share of the tiles from the END (pure scheduling)
  int sched;
};

// Global stores of the row-major fast epilogues, plain or write-through by a BUILD switch (-DGCD_EPI_WT=mask: 1 the
// GEGLU path, 2 the fp16 path, 4 the fp32 / residual path; 0 = plain stores, the default).  sc1 stores do not keep the
// written line in the XCD's L2 (MI355X_MICROARCH.md, "stores of each flavour"): a tile's output is never read again by
// its launch, the A / W panels its neighbours re-read are.  A build switch, not a descriptor bit, so that the A/B
// (tools/ab_sweep.sh: a second library, GCD_AMD_LIB) leaves the code generation of the product kernels untouched.
// Inline asm because clang has no builtin for a flat global store with a cache policy; the trailing s_nop covers the
// "VMEM store of more than 8 bytes -> VALU write of the data registers" hazard the compiler cannot see through the asm
// (the in-order vmcnt the compiler derives stays a safe over-estimate: an unknown extra store in the queue only makes a
// counted wait stricter).
#ifndef GCD_EPI_WT
#define GCD_EPI_WT 0
#endif
// -DGCD_EPI_NT=mask: the same paths with NON-TEMPORAL stores (bits 1 / 2 / 4 as above) and, bit 8, non-temporal loads of
// the fp32 residuals (read once per launch) — the hint that moved the LayerNorm / GroupNorm streamers (norm.hip).
#ifndef GCD_EPI_NT
#define GCD_EPI_NT 0
#endif
template <bool WT, bool NT, typename V>
__device__ __forceinline__ void gcd_store16(void* ptr, const V& v) {
  static_assert(sizeof(V) == 16, "16-byte vector");
  if constexpr (WT) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(ptr), "v"(v));
  else if constexpr (NT) __builtin_nontemporal_store(v, (V*)ptr);
  else *(V*)ptr = v;
}
template <bool WT, bool NT, typename V>
__device__ __forceinline__ void gcd_store8(void* ptr, const V& v) {
  static_assert(sizeof(V) == 8, "8-byte vector");
  if constexpr (WT) asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(ptr), "v"(v));
  else if constexpr (NT) __builtin_nontemporal_store(v, (V*)ptr);
  else *(V*)ptr = v;
}
__device__ __forceinline__ f32x4 gcd_load_res(const float* p) {
  if constexpr ((GCD_EPI_NT & 8) != 0) return __builtin_nontemporal_load((const f32x4*)p);
  else return *(const f32x4*)p;
}

// gemm_pp.hip: the 256 x 320 ping-pong kernel.
bool gcd_gemm_pp_supported(const GemmK& k, int mode);
int gcd_gemm_pp_launch(const GemmK& k, int mode, hipStream_t s);

// gemm_pp.hip: split-K launch = partial sums into `ws` + reduce-and-epilogue kernel.
int gcd_gemm_pp_launch_splitk(const GemmK& k, int mode, int splitk, float* ws, hipStream_t s);

// runtime.hip: tuning knobs (gcd_tune_set / environment), see include/gcd_amd.h
int gcd_tune_get(int knob);
// conv_narrow.hip: the 3x3 convolution with N == 16 output columns (the UNet's 320 -> 4 output head)
bool gcd_conv3x3_narrow_supported(const GemmK& k, int mode);
int gcd_conv3x3_narrow_launch(const GemmK& k, hipStream_t s);

// ------------------------------------------------------------------------------------------------
// Coalesced epilogue of a wave that owns 64 tokens x 160 channels as acc[5][2] tiles of
// v_mfma_f32_32x32x16 (gemm_pp.hip):  acc[i][j][4g + e] = C[m_base + 32 j + l31]
// [n_base + 32 i + 8 g + 4 hh + e].  In that layout a store instruction touches 32 rows with 16-32
// bytes each, which makes the small-K GEMMs store-issue bound.  The tiles are therefore transposed
// through a wave-private LDS region (`stage`, >= GCD_EPI_STAGE_BYTES, 16-byte aligned) so that every
// global access of the epilogue — residual loads included — is 8 rows x 128 contiguous bytes (fp32)
// per instruction.  No workgroup barrier inside: a wave only reads back what it wrote itself.
// ------------------------------------------------------------------------------------------------
#define GCD_EPI_ROW_F32 144                       /* 32 floats + 16 B pad: conflict-free b128 writes */
#define GCD_EPI_TILE_F32 (32 * GCD_EPI_ROW_F32)   /* one 32 x 32 fp32 tile */
#define GCD_EPI_ROW_F16 176                       /* GEGLU: 80 fp16 + 16 B pad */
#define GCD_EPI_STAGE_BYTES 10752                 /* max(2 * 4608, 32 * 176, 32 * 336) */

// Measured on MI355X (tools/gemm_bench, profiles/r01_gemm_epilogue_ablation.txt): the transposed
// path pays for fp16 outputs (8-byte pieces per row become 64-byte rows: q|k|v projections -7..-12 %)
// but not for fp32 outputs with a residual (their cost is the HBM read+write mix, not the store
// shape: stores alone +40 us, residual loads alone +100 us, both +215 us on 258048 x 320), so fp32
// and GEGLU outputs are written straight from the accumulator layout.
// EV: experiment switches for tools/gemm_bench (0 = product): 1 skip residual loads, 2 skip stores,
// 4 nontemporal stores, 8 force the direct path, 16 force the transposed path, 32 no epilogue at all
// Fused LayerNorm (p.ln_out != nullptr, N == 320 == the tile width, fp32 out): the two waves that
// share a 64-token row block (wn = 0 / 1, 160 channels each) exchange per-row sum and sum of squares
// through `red` (LDS, >= 4 KB, workgroup-shared; ONE __syncthreads, so every thread of the workgroup
// must call this), then each normalises its own 160 channels from the registers that still hold the
// fp32 row it just stored.  Single-pass variance in fp32 over 320 values (error ~1e-6 * E[x^2]).
__device__ __forceinline__ void gcd_epilogue_64x160_ln(const GemmK& p, f32x16 (&acc)[5][2],
                                                       int m_base, int n_base, int lane, int wm,
                                                       int wn, float* red) {
  const int l31 = lane & 31, hh = lane >> 5;
  float rs[2] = {0.f, 0.f}, rq[2] = {0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int m = m_base + 32 * j + l31;
    const bool mok = m < p.M;
    const int mc = mok ? m : p.M - 1;
    float sa = p.s_acc, sr1 = p.s_r1, sr2 = p.s_r2;
    if (p.frame_alpha) {
      const float al = p.frame_alpha[mc / p.rows_per_alpha];
      sa = 1.0f - al;
      sr2 = al;
      if (p.r1_blend) sr1 *= 1.0f - al;
    }
    const float* rv = p.rowvec ? p.rowvec + (int64_t)(mc / p.rows_per_vec) * p.ld_rowvec : nullptr;
    const float* av = p.ln_addvec ? p.ln_addvec + (int64_t)(mc / p.ln_rows_per_vec) * p.ld_ln_addvec
                                  : nullptr;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      __builtin_amdgcn_sched_barrier(0);   // keep the loads of one column block together (registers)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = n_base + 32 * i + 8 * g + 4 * hh;
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g + e];
        if (p.bias) v += *(const f32x4*)(p.bias + n);
        if (rv) v += *(const f32x4*)(rv + n);
        v *= sa;
        if (p.R1) v += sr1 * *(const f32x4*)(p.R1 + (int64_t)mc * p.ldr1 + n);
        if (p.R2) v += sr2 * *(const f32x4*)(p.R2 + (int64_t)mc * p.ldr2 + n);
        if (mok) *(f32x4*)((float*)p.out + (int64_t)m * p.ldo + n) = v;
        if (av) {
          v += *(const f32x4*)(av + n);
          if (p.ln_sum_out && mok) *(f32x4*)(p.ln_sum_out + (int64_t)m * p.ld_ln_sum + n) = v;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          acc[i][j][4 * g + e] = v[e];
          rs[j] += v[e];
          rq[j] = fmaf(v[e], v[e], rq[j]);
        }
      }
    }
  }
  // lane ^ 32 holds the other half of this wave's 160 channels of the same row
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(rs[j]), __float_as_uint(rs[j]), false, false);
    rs[j] = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(rq[j]), __float_as_uint(rq[j]), false, false);
    rq[j] = __uint_as_float(b[0]) + __uint_as_float(b[1]);
    if (hh == 0) {
      float* dst = red + ((wn * 4 + wm) * 64 + 32 * j + l31) * 2;
      dst[0] = rs[j];
      dst[1] = rq[j];
    }
  }
  __syncthreads();
  float mean[2], rstd[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const float* src = red + (((1 - wn) * 4 + wm) * 64 + 32 * j + l31) * 2;
    const float s = rs[j] + src[0], q = rq[j] + src[1];
    mean[j] = s * (1.0f / 320.0f);
    const float var = fmaxf(q * (1.0f / 320.0f) - mean[j] * mean[j], 0.f);
    rstd[j] = rsqrtf(var + p.ln_eps);
  }
#pragma unroll
  for (int i = 0; i < 5; ++i)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      __builtin_amdgcn_sched_barrier(0);
      const int n = n_base + 32 * i + 8 * g + 4 * hh;
      const f32x4 ga = *(const f32x4*)(p.ln_gamma + n), be = *(const f32x4*)(p.ln_beta + n);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int m = m_base + 32 * j + l31;
        f16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          o[e] = (f16)fmaf((acc[i][j][4 * g + e] - mean[j]) * rstd[j], ga[e], be[e]);
        if (m < p.M) *(f16x4*)(p.ln_out + (int64_t)m * p.ld_ln_out + n) = o;
      }
    }
}

// ------------------------------------------------------------------------------------------------
// The two accumulator layouts of a wave's 64 x 160 tile, and the helpers that hide them from the
// row-major epilogues below (the read-back side of every staged path is layout independent):
//   GcdAcc32 (gemm_pp.hip, v_mfma_f32_32x32x16):  acc[i][j][4g + e] = C[32 j + l31][32 i + 8 g + 4 hh + e]
//   GcdAcc16 (gemm_p8.hip, v_mfma_f32_16x16x32):  acc[c][t][e]      = C[16 t + (lane & 15)][16 c + 4 (lane >> 4) + e]
// GEGLU weight rows are interleaved 16 value / 16 gate per 32-row block (packing.pack_geglu): value and gate
// of one output live in the same lane in both layouts (registers 4g+e / 8+4g+e, blocks 2i / 2i+1).
// ------------------------------------------------------------------------------------------------
typedef f32x16 GcdAcc32[5][2];
typedef f32x4 GcdAcc16[10][4];

// 32 x 32 sub-tile (column block i, token half j) -> `stage` as row-major fp32 rows of GCD_EPI_ROW_F32 bytes
__device__ __forceinline__ void gcd_stage_tile32(const GcdAcc32& acc, int i, int j, char* stage, int lane) {
  char* const wr = stage + (lane & 31) * GCD_EPI_ROW_F32 + (lane >> 5) * 16;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g + e];
    *(f32x4*)(wr + g * 32) = v;
  }
}
__device__ __forceinline__ void gcd_stage_tile32(const GcdAcc16& acc, int i, int j, char* stage, int lane) {
  char* const wr = stage + (lane & 15) * GCD_EPI_ROW_F32 + (lane >> 4) * 16;
#pragma unroll
  for (int th = 0; th < 2; ++th)
#pragma unroll
    for (int ch = 0; ch < 2; ++ch) *(f32x4*)(wr + th * 16 * GCD_EPI_ROW_F32 + ch * 64) = acc[2 * i + ch][2 * j + th];
}

// GEGLU: token half j of the wave tile -> `stage` as fp16 [32][80] rows of GCD_EPI_ROW_F16 bytes
// (lbl = the wave's 160 staged per-column addends)
__device__ __forceinline__ void gcd_stage_geglu_half(const GcdAcc32& acc, int j, const float* lb, char* stage, int lane) {
  const int l31 = lane & 31, hh = lane >> 5;
  const float* lbl = lb + 4 * hh;
#pragma unroll
  for (int i = 0; i < 5; ++i)
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const f32x4 ba = *(const f32x4*)(lbl + 32 * i + 8 * g);
      const f32x4 bg = *(const f32x4*)(lbl + 32 * i + 16 + 8 * g);
      f16x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        o[e] = (f16)((acc[i][j][4 * g + e] + ba[e]) * gelu_fast(acc[i][j][8 + 4 * g + e] + bg[e]));
      *(f16x4*)(stage + l31 * GCD_EPI_ROW_F16 + (16 * i + 8 * g + 4 * hh) * 2) = o;
    }
}
__device__ __forceinline__ void gcd_stage_geglu_half(const GcdAcc16& acc, int j, const float* lb, char* stage, int lane) {
  const int r15 = lane & 15, q = lane >> 4;
  const float* lbl = lb + 4 * q;
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const f32x4 ba = *(const f32x4*)(lbl + 32 * i);
    const f32x4 bg = *(const f32x4*)(lbl + 32 * i + 16);
#pragma unroll
    for (int th = 0; th < 2; ++th) {
      f16x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        o[e] = (f16)((acc[2 * i][2 * j + th][e] + ba[e]) * gelu_fast(acc[2 * i + 1][2 * j + th][e] + bg[e]));
      *(f16x4*)(stage + (16 * th + r15) * GCD_EPI_ROW_F16 + (16 * i + 4 * q) * 2) = o;
    }
  }
}

#ifdef GCD_ABLATION_BUILD
// Ablation forms of the GEGLU staging (gemm_p8 VAR bits 128 / 256, tools/gemm_bench): ABL 1 = value * gate without the GELU
// polynomial, ABL 2 = no arithmetic at all (the value, converted) — how much of the serial tile epilogue is VALU.
template <int ABL>
__device__ __forceinline__ void gcd_stage_geglu_half_abl(const GcdAcc16& acc, int j, const float* lb, char* stage, int lane) {
  const int r15 = lane & 15, q = lane >> 4;
  const float* lbl = lb + 4 * q;
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const f32x4 ba = *(const f32x4*)(lbl + 32 * i);
    const f32x4 bg = *(const f32x4*)(lbl + 32 * i + 16);
#pragma unroll
    for (int th = 0; th < 2; ++th) {
      f16x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        o[e] = ABL == 2 ? (f16)acc[2 * i][2 * j + th][e]
                        : (f16)((acc[2 * i][2 * j + th][e] + ba[e]) * (acc[2 * i + 1][2 * j + th][e] + bg[e]));
      *(f16x4*)(stage + (16 * th + r15) * GCD_EPI_ROW_F16 + (16 * i + 4 * q) * 2) = o;
    }
  }
}
#endif

// fp16 output without residuals: token half j -> `stage` as fp16 [32][160] rows of GCD_EPI_ROW_H160 bytes
#define GCD_EPI_ROW_H160 336
__device__ __forceinline__ void gcd_stage_f16_half(const GcdAcc32& acc, int j, const float* lb, float sa, char* stage,
                                                   int lane) {
  const int l31 = lane & 31, hh = lane >> 5;
  const float* lbl = lb + 4 * hh;
#pragma unroll
  for (int i = 0; i < 5; ++i)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 bv = *(const f32x4*)(lbl + 32 * i + 8 * g);
      f16x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (f16)((acc[i][j][4 * g + e] + bv[e]) * sa);
      *(f16x4*)(stage + l31 * GCD_EPI_ROW_H160 + (32 * i + 8 * g + 4 * hh) * 2) = o;
    }
}
__device__ __forceinline__ void gcd_stage_f16_half(const GcdAcc16& acc, int j, const float* lb, float sa, char* stage,
                                                   int lane) {
  const int r15 = lane & 15, q = lane >> 4;
  const float* lbl = lb + 4 * q;
#pragma unroll
  for (int c = 0; c < 10; ++c) {
    const f32x4 bv = *(const f32x4*)(lbl + 16 * c);
#pragma unroll
    for (int th = 0; th < 2; ++th) {
      f16x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (f16)((acc[c][2 * j + th][e] + bv[e]) * sa);
      *(f16x4*)(stage + (16 * th + r15) * GCD_EPI_ROW_H160 + (16 * c + 4 * q) * 2) = o;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Pipelined epilogues for a FULL 64 x 160 wave tile (every row < M, every column < N).
//
// The generic epilogue further below issues one residual load, waits for it (s_waitcnt vmcnt(0),
// which on gfx9-family hardware also waits for every earlier STORE: loads and stores share the
// in-order vmcnt counter), stores, and repeats — the output may alias the residual, so the compiler
// cannot hoist a load over the previous store.  That serialises 40 HBM round trips per wave and tile.
// Here the residual vectors are fetched in batches of 4 (one 32 x 32 tile) that run D batches ahead
// of their use, so the counted waits the compiler derives only cover stores issued D batches earlier
// and 4-8 KB of loads per wave (32-64 KB per CU) stay in flight.  Aliasing stays legal: every element
// is loaded by the lane that later stores it, before that store.
//
// lb: this wave's 160 per-column addends in LDS (bias + the rowvec row when it is uniform over the
// tile), staged by the caller — bias loads would otherwise sit in vmcnt between the stores.
// ------------------------------------------------------------------------------------------------
// fp32 outputs, row-major.  tools/hbm_epi (the same read-modify-write without the GEMM,
// 330 MB in place, 256 x 8 waves): accumulator-layout accesses (32 rows x 32 B per instruction) run
// at 3.8 TB/s, row-contiguous ones (8 rows x 128 B) at 5.6 TB/s — the vector-memory path wants whole
// 128-byte lines per instruction.  Each 32 x 32 accumulator tile is therefore transposed through a
// wave-private LDS tile (`stage`, >= 4608 B): lane -> row 8 qq + (lane >> 3), channels 4 (lane & 7),
// and the residual vectors are fetched in that layout, D tiles ahead.  sa / sr1 / sr2 are
// wave-uniform here (the caller checked that one frame_alpha entry serves the whole tile).
// Sum over the 8 lanes that differ in lane bits 3, 4, 5 (same lane & 7), result in all of them — three
// VALU cross-lane ops (DPP row rotate by 8, v_permlane16_swap, v_permlane32_swap), no LDS round trip:
// eight dependent ds_bpermute chains per column block made the statistics cost ~3 % of a conv launch.
__device__ __forceinline__ float gcd_sum_lane_bits_345(float a) {
  const int v = __float_as_int(a);
  a += __int_as_float(__builtin_amdgcn_update_dpp(v, v, 0x128 /* row_ror:8 */, 0xf, 0xf, false));
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

// STATS: also reduce the stored values per column over the wave's 64 rows (sum, sum of squares) and
// write them to p.colstats — a lane owns 4 fixed channels of every 32-column block and 8 of its rows
// (rr + 8 qq + 32 j), so one block costs 8 adds + 8 FMAs per lane on values it holds anyway, then an
// 8-lane tree over the lanes that share those channels (lane bits 3-5).
template <bool HAS_R1, bool HAS_R2, bool OUT16 = false, bool STATS = false, typename ACC = GcdAcc32>
__device__ __forceinline__ void gcd_epi_f32_rows_full(const GemmK& p, ACC& acc, int m_base,
                                                     int n_base, int lane, const float* lb, char* stage,
                                                     float sa, float sr1, float sr2) {
  constexpr int D = HAS_R2 ? 1 : 2;
  const int rr = lane >> 3, cc = (lane & 7) * 4;
  const int64_t col = n_base + cc;
  // OUT16: the same pipeline with an fp16 result (the last temporal FF of a transformer, whose blended
  // output only feeds proj_out): 8 rows x 64 B per store instruction
  float* const op = (float*)p.out + (int64_t)(m_base + rr) * p.ldo + col;
  f16* const op16 = (f16*)p.out + (int64_t)(m_base + rr) * p.ldo + col;
  const float* const r1p = HAS_R1 ? p.R1 + (int64_t)(m_base + rr) * p.ldr1 + col : nullptr;
  const float* const r2p = HAS_R2 ? p.R2 + (int64_t)(m_base + rr) * p.ldr2 + col : nullptr;
  const int64_t so = 8 * p.ldo, s1 = 8 * p.ldr1, s2 = 8 * p.ldr2;
  const char* const rd = stage + rr * GCD_EPI_ROW_F32 + cc * 4;
  f32x4 q1[D][4], q2[D][4];
  auto fetch = [&](int b, int slot) {   // tile b = (column block b >> 1, token half b & 1)
    const int i = b >> 1, j = b & 1;
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) {
      if (HAS_R1) q1[slot][qq] = gcd_load_res(r1p + (4 * j + qq) * s1 + 32 * i);
      if (HAS_R2) q2[slot][qq] = gcd_load_res(r2p + (4 * j + qq) * s2 + 32 * i);
    }
  };
  if (HAS_R1 || HAS_R2) {
#pragma unroll
    for (int b = 0; b < D; ++b) fetch(b, b);
  }
  f32x4 cs = {0.f, 0.f, 0.f, 0.f}, cq = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int b = 0; b < 10; ++b) {
    const int i = b >> 1, j = b & 1;
    gcd_stage_tile32(acc, i, j, stage, lane);
    const f32x4 bv = *(const f32x4*)(lb + 32 * i + cc);
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) {
      f32x4 v = *(const f32x4*)(rd + qq * 8 * GCD_EPI_ROW_F32);
      v = (v + bv) * sa;
      if (HAS_R1) v += sr1 * q1[b % D][qq];
      if (HAS_R2) v += sr2 * q2[b % D][qq];
      if (STATS) {
        cs += v;
#pragma unroll
        for (int e = 0; e < 4; ++e) cq[e] = fmaf(v[e], v[e], cq[e]);
      }
      if (OUT16) {
        f16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (f16)v[e];
        gcd_store8<(GCD_EPI_WT & 4) != 0, (GCD_EPI_NT & 4) != 0>(op16 + (4 * j + qq) * so + 32 * i, o);
      } else {
        gcd_store16<(GCD_EPI_WT & 4) != 0, (GCD_EPI_NT & 4) != 0>(op + (4 * j + qq) * so + 32 * i, v);
      }
    }
    if ((HAS_R1 || HAS_R2) && b + D < 10) fetch(b + D, b % D);
    if (STATS && j == 1) {
      // both token halves of column block i are in: fold the 8 lanes (rr = 0..7) that hold the same
      // 4 channels, lanes 0-7 write [sum | sumsq] of the wave's 64 rows
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        cs[e] = gcd_sum_lane_bits_345(cs[e]);
        cq[e] = gcd_sum_lane_bits_345(cq[e]);
      }
      if (rr == 0) {
        float* dst = p.colstats + (int64_t)(m_base >> 6) * 2 * p.N + col + 32 * i;
        *(f32x4*)dst = cs;
        *(f32x4*)(dst + p.N) = cq;
      }
      cs = f32x4{0.f, 0.f, 0.f, 0.f};
      cq = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __builtin_amdgcn_sched_barrier(0);   // keep the pipeline order (and the register budget) as written
  }
}

// a * gelu(g) of a full wave tile (value / gate rows interleaved in 16s by packing.pack_geglu, so both
// live in the same lane; lb as above).  The wave's 64 x 80 fp16 results go through its LDS stage
// ([32][80] + 16 B row pad, twice) and leave as 16-byte pieces of 160-byte row segments: 6.4 rows per store
// instruction instead of 32 rows x 16 B straight from the accumulator layout, and half as many store
// instructions (measured -3..-7 % per GEGLU launch, profiles/r01q_gemm_bench_rows.txt).
template <typename ACC, int ABL = 0>
__device__ __forceinline__ void gcd_epi_geglu_rows_full(const GemmK& p, ACC& acc, int m_base,
                                                        int n_base, int lane, const float* lb, char* stage) {
  // row-major [M, N/2], or tile-blocked: tile (tm, tn) = one contiguous [256][160] block (out_blocked)
  f16* outp = (f16*)p.out + (int64_t)m_base * p.ldo + (n_base >> 1);
  int64_t rs = p.ldo;
  if (p.out_blocked) {
    const int tn = n_base / 320;
    const int64_t blk = (int64_t)(m_base >> 8) * (p.N / 320) + tn;
    outp = (f16*)p.out + (blk * 256 + (m_base & 255)) * 160 + ((n_base - tn * 320) >> 1);
    rs = 160;
  }
  // one 32-token half at a time (5.5 KB of stage): the stores of half 0 drain under the GELU of half 1
#pragma unroll
  for (int j = 0; j < 2; ++j) {
#ifdef GCD_ABLATION_BUILD
    if constexpr (ABL != 0) gcd_stage_geglu_half_abl<ABL>(acc, j, lb, stage, lane);
    else
#endif
      gcd_stage_geglu_half(acc, j, lb, stage, lane);
#pragma unroll
    for (int it = 0; it < 5; ++it) {
      const int tt = it * 64 + lane;
      const int row = tt / 10, ch = tt - row * 10;
      const f16x8 v = *(const f16x8*)(stage + row * GCD_EPI_ROW_F16 + ch * 16);
      gcd_store16<(GCD_EPI_WT & 1) != 0, (GCD_EPI_NT & 1) != 0>(outp + (int64_t)(32 * j + row) * rs + ch * 8, v);
    }
  }
}

// fp16 outputs without residuals (the q|k|v projections): each 32-token half of the wave tile is
// staged as fp16 [32][160] (+16 B row pad) and leaves as 16-byte pieces of 320-byte row segments — 3.2
// rows per store instruction, against 8 rows x 64 B through the generic fp32-staged path below.
template <typename ACC>
__device__ __forceinline__ void gcd_epi_f16_rows_full(const GemmK& p, ACC& acc, int m_base,
                                                      int n_base, int lane, const float* lb, char* stage) {
  const float sa = p.s_acc;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    gcd_stage_f16_half(acc, j, lb, sa, stage, lane);
    f16* const outp = (f16*)p.out + (int64_t)(m_base + 32 * j) * p.ldo + n_base;
#pragma unroll
    for (int it = 0; it < 10; ++it) {
      const int tt = it * 64 + lane;
      const int row = tt / 20, ch = tt - row * 20;
      const f16x8 v = *(const f16x8*)(stage + row * GCD_EPI_ROW_H160 + ch * 16);
      gcd_store16<(GCD_EPI_WT & 2) != 0, (GCD_EPI_NT & 2) != 0>(outp + (int64_t)row * p.ldo + ch * 8, v);
    }
  }
}

template <int EV = 0>
__device__ __forceinline__ void gcd_epilogue_64x160(const GemmK& p, f32x16 (&acc)[5][2], int m_base,
                                                    int n_base, int lane, char* stage) {
  const int l31 = lane & 31, hh = lane >> 5;
  if (EV & 32) {   // experiment: no epilogue at all (mainloop-only timing); never set by VAR decoding of
                   // the product kernels (gemm_pp.hip maps VAR bit 32768 to it)
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) asm volatile("" ::"v"(acc[i][j]));
    return;
  }
  if (p.out_kind == GCD_OUT_GEGLU) {   // ragged N: direct stores from the accumulator layout
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int m = m_base + 32 * j + l31;
      if (m >= p.M) continue;
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        const int nb = n_base + 32 * i;
        if (nb >= p.N) continue;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          const int c = 8 * g + 4 * hh;
          f32x4 a, gt;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            a[e] = acc[i][j][4 * g + e];
            gt[e] = acc[i][j][8 + 4 * g + e];
          }
          if (p.bias) {
            a += *(const f32x4*)(p.bias + nb + c);
            gt += *(const f32x4*)(p.bias + nb + 16 + c);
          }
          f16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (f16)((EV & 4) ? a[e] * gt[e] : a[e] * gelu_fast(gt[e]));
          if (EV & 2) asm volatile("" ::"v"(o));
          else *(f16x4*)((f16*)p.out + (int64_t)m * p.ldo + (nb >> 1) + c) = o;
        }
      }
    }
    return;
  }
  if (((EV & 8) || p.out_kind == GCD_OUT_F32) && !(EV & 16)) {
    // direct stores from the accumulator layout (32 rows x 32 B per instruction)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int m = m_base + 32 * j + l31;
      if (m >= p.M) continue;
      float sa = p.s_acc, sr1 = p.s_r1, sr2 = p.s_r2;
      if (p.frame_alpha) {
        const float al = p.frame_alpha[m / p.rows_per_alpha];
        sa = 1.0f - al;
        sr2 = al;
        if (p.r1_blend) sr1 *= 1.0f - al;
      }
      const float* rv = p.rowvec ? p.rowvec + (int64_t)(m / p.rows_per_vec) * p.ld_rowvec : nullptr;
#pragma unroll
      for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = n_base + 32 * i + 8 * g + 4 * hh;
          if (n >= p.N) continue;
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g + e];
          if (p.bias) v += *(const f32x4*)(p.bias + n);
          if (rv) v += *(const f32x4*)(rv + n);
          v *= sa;
          if (p.R1 && !(EV & 1)) v += sr1 * *(const f32x4*)(p.R1 + (int64_t)m * p.ldr1 + n);
          if (p.R2 && !(EV & 1)) v += sr2 * *(const f32x4*)(p.R2 + (int64_t)m * p.ldr2 + n);
          if (EV & 2) {
            asm volatile("" ::"v"(v));
          } else if (p.out_kind == GCD_OUT_F32) {
            f32x4* dst = (f32x4*)((float*)p.out + (int64_t)m * p.ldo + n);
            if (EV & 4) __builtin_nontemporal_store(v, dst);
            else *dst = v;
          } else {
            f16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (f16)v[e];
            *(f16x4*)((f16*)p.out + (int64_t)m * p.ldo + n) = o;
          }
        }
    }
    return;
  }
  // fp32 / fp16 outputs: per 32-channel column block, both 32-token tiles go through LDS and come
  // back row-major: lane -> row 8 q + (lane >> 3), channels 4 (lane & 7) .. +3.
  const int rr = lane >> 3, cc = (lane & 7) * 4;
#pragma unroll
  for (int i = 0; i < 5; ++i) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g + e];
        *(f32x4*)(stage + j * GCD_EPI_TILE_F32 + l31 * GCD_EPI_ROW_F32 + (8 * g + 4 * hh) * 4) = v;
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int n = n_base + 32 * i + cc;
    const bool n_ok = n < p.N;
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (p.bias && n_ok) bv = *(const f32x4*)(p.bias + n);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int row = 8 * q + rr;
      const int m = m_base + row;
      f32x4 v = *(const f32x4*)(stage + (row >> 5) * GCD_EPI_TILE_F32 + (row & 31) * GCD_EPI_ROW_F32 +
                                cc * 4);
      if (m < p.M && n_ok) {
        float sa = p.s_acc, sr1 = p.s_r1, sr2 = p.s_r2;
        if (p.frame_alpha) {
          const float al = p.frame_alpha[m / p.rows_per_alpha];
          sa = 1.0f - al;
          sr2 = al;
          if (p.r1_blend) sr1 *= 1.0f - al;
        }
        v += bv;
        if (p.rowvec) v += *(const f32x4*)(p.rowvec + (int64_t)(m / p.rows_per_vec) * p.ld_rowvec + n);
        v *= sa;
        if (p.R1 && !(EV & 1)) v += sr1 * *(const f32x4*)(p.R1 + (int64_t)m * p.ldr1 + n);
        if (p.R2 && !(EV & 1)) v += sr2 * *(const f32x4*)(p.R2 + (int64_t)m * p.ldr2 + n);
        if (EV & 2) {
          asm volatile("" ::"v"(v));
        } else if (p.out_kind == GCD_OUT_F32) {
          f32x4* dst = (f32x4*)((float*)p.out + (int64_t)m * p.ldo + n);
          if (EV & 4) __builtin_nontemporal_store(v, dst);
          else *dst = v;
        } else {
          f16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (f16)v[e];
          *(f16x4*)((f16*)p.out + (int64_t)m * p.ldo + n) = o;
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // read-back done before the next overwrite
  }
}

// The generic epilogue for the 16x16x32 accumulator layout (gemm_p8.hip): ragged edge tiles, a rowvec / alpha that
// changes inside a tile, fp16 outputs with residuals, split-K partial sums — direct stores from the accumulator layout
// (16 rows x 64 B fp32 per instruction).  Full tiles take the row-major fast paths above.
__device__ __forceinline__ void gcd_epilogue_64x160(const GemmK& p, GcdAcc16& acc, int m_base, int n_base, int lane) {
  const int r15 = lane & 15, q = lane >> 4;
  if (p.out_kind == GCD_OUT_GEGLU) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int m = m_base + 16 * t + r15;
      if (m >= p.M) continue;
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        const int nb = n_base + 32 * i;
        if (nb >= p.N) continue;
        f32x4 a = acc[2 * i][t], gt = acc[2 * i + 1][t];
        if (p.bias) {
          a += *(const f32x4*)(p.bias + nb + 4 * q);
          gt += *(const f32x4*)(p.bias + nb + 16 + 4 * q);
        }
        f16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (f16)(a[e] * gelu_fast(gt[e]));
        *(f16x4*)((f16*)p.out + (int64_t)m * p.ldo + (nb >> 1) + 4 * q) = o;
      }
    }
    return;
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int m = m_base + 16 * t + r15;
    if (m >= p.M) continue;
    float sa = p.s_acc, sr1 = p.s_r1, sr2 = p.s_r2;
    if (p.frame_alpha) {
      const float al = p.frame_alpha[m / p.rows_per_alpha];
      sa = 1.0f - al;
      sr2 = al;
      if (p.r1_blend) sr1 *= 1.0f - al;
    }
    const float* rv = p.rowvec ? p.rowvec + (int64_t)(m / p.rows_per_vec) * p.ld_rowvec : nullptr;
#pragma unroll
    for (int c = 0; c < 10; ++c) {
      const int n = n_base + 16 * c + 4 * q;
      if (n >= p.N) continue;
      f32x4 v = acc[c][t];
      if (p.bias) v += *(const f32x4*)(p.bias + n);
      if (rv) v += *(const f32x4*)(rv + n);
      v *= sa;
      if (p.R1) v += sr1 * *(const f32x4*)(p.R1 + (int64_t)m * p.ldr1 + n);
      if (p.R2) v += sr2 * *(const f32x4*)(p.R2 + (int64_t)m * p.ldr2 + n);
      if (p.out_kind == GCD_OUT_F32) {
        *(f32x4*)((float*)p.out + (int64_t)m * p.ldo + n) = v;
      } else {
        f16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (f16)v[e];
        *(f16x4*)((f16*)p.out + (int64_t)m * p.ldo + n) = o;
      }
    }
  }
}
