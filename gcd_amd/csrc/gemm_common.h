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
};

// gemm_pp.hip: the 256 x 320 ping-pong kernel.
bool gcd_gemm_pp_supported(const GemmK& k, int mode);
int gcd_gemm_pp_launch(const GemmK& k, int mode, hipStream_t s);

// gemm_pp.hip: split-K launch = partial sums into `ws` + reduce-and-epilogue kernel.
int gcd_gemm_pp_launch_splitk(const GemmK& k, int mode, int splitk, float* ws, hipStream_t s);

// runtime.hip: tuning knobs (gcd_tune_set / environment), see include/gcd_amd.h
int gcd_tune_get(int knob);

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
#define GCD_EPI_STAGE_BYTES 11264                 /* max(2 * 4608, 64 * 176) */

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
  if ((EV & 16) && p.out_kind == GCD_OUT_GEGLU && p.N % 320 == 0 && (p.ldo & 7) == 0) {
    // a * gelu(g) in the accumulator layout (value / gate live in the same lane), staged as fp16
    // [64 rows][80 hidden columns], written out as 160 contiguous bytes per row.
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        const int nb = n_base + 32 * i;
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
          for (int e = 0; e < 4; ++e) o[e] = (f16)(a[e] * gelu_fast(gt[e]));
          *(f16x4*)(stage + (32 * j + l31) * GCD_EPI_ROW_F16 + (16 * i + c) * 2) = o;
        }
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    f16* outp = (f16*)p.out + (n_base >> 1);
#pragma unroll
    for (int it = 0; it < 10; ++it) {
      const int tt = it * 64 + lane;
      const int row = tt / 10, ch = tt - row * 10;
      const int m = m_base + row;
      const f16x8 v = *(const f16x8*)(stage + row * GCD_EPI_ROW_F16 + ch * 16);
      if (m < p.M) *(f16x8*)(outp + (int64_t)m * p.ldo + ch * 8) = v;
    }
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
