// attn_bwd.hip — backward of the spatial self-attention (d = 64 per head) for the fine-tune step
// (BASELINE.json cfg4; reference: torch.autograd through F.scaled_dot_product_attention,
// sgm/modules/attention.py:281-344).  Flash-style: nothing S x S ever reaches memory.
//
//   forward (attention.hip) kept qkv (fp16) and O (fp16).  With s = scale * q.k, P = softmax(s):
//     delta_i = sum_d dO_id O_id                     dV = P^T dO
//     dP = dO V^T        dS = P o (dP - delta)       dQ = scale * dS K       dK = scale * dS^T Q
//
// Two kernels, both recomputing P tile by tile from a per-query log-sum-exp:
//   * attn_bwd_dq_kernel   (query-stationary): pass 1 over the keys -> lse (and delta), pass 2 -> dQ;
//   * attn_bwd_dkdv_kernel (key-stationary):   one pass over the queries -> dK, dV.
// Every product is v_mfma_f32_32x32x16_f16 with the operands swapped so that the score tile comes out
// with ONE stationary index per lane:  S^T[key][query] in the dQ kernel, S[query][key] in the dK/dV
// kernel.  The exponentiated / differentiated tile is then already the B operand of the next product
// (register r of a lane holds row (r&3) + 8 (r>>2) + 4 (lane>>5) — the k-slot order of the MFMA), and
// the matching A operands (K^T, Q^T, dO^T) are read from copies transposed once per call with the SAME
// 16-row permutation the forward's V^T uses (gcd_attn_transpose_heads).  No cross-lane traffic.
//
// Tiles: 4 waves per workgroup, 32 stationary rows per wave, streamed tiles of 32 rows through a
// register-prefetched double buffer in LDS (one barrier per tile).  Not tuned beyond that: at the
// fine-tune shapes (S <= 1536) attention is a few percent of the step.
#include "common.h"

#define LOG2E_F 1.4426950408889634f

namespace {

// [64 d][32 rows-permuted] fp16 tile: 64-byte rows of four 16-B chunks, chunk c of row d at c ^ ((d >> 2) & 3)
__device__ __forceinline__ int ldsT_off(int d, int chunk) { return d * 64 + ((chunk ^ ((d >> 2) & 3)) << 4); }

constexpr int NAT_BYTES = 32 * 128;   // natural tile: 32 rows x 64 d
constexpr int TR_BYTES = 64 * 64;     // transposed tile: 64 d x 32 rows

// ------------------------------------------------------------------------------------------------
// dQ (+ lse, delta).  grid (ceil(S/128), heads, frames), block 256.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(
    const f16* __restrict__ qkv, int64_t ld, const f16* __restrict__ kT, const f16* __restrict__ dO,
    int64_t lddo, const f16* __restrict__ O, int64_t ldoo, float* __restrict__ dqkv, int64_t ldg,
    float* __restrict__ lse, float* __restrict__ delta, int S, int S_pad, int heads, float scale) {
  // per buffer: K natural | V natural | K^T
  __shared__ __attribute__((aligned(16))) char smem[2 * (2 * NAT_BYTES + TR_BYTES)];
  constexpr int BUF = 2 * NAT_BYTES + TR_BYTES;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int head = blockIdx.y, frame = blockIdx.z;
  const int C = heads * 64;
  const int q0 = blockIdx.x * 128 + wave * 32;
  const float c = scale * LOG2E_F;

  // stationary operands (B): query l31, d = 16 ks + 8 half + j
  int qi = q0 + l31;
  const bool q_ok = qi < S;
  qi = q_ok ? qi : S - 1;
  f16x8 qf[4], dof[4];
  float dl = 0.f;
  {
    const f16* qp = qkv + ((int64_t)frame * S + qi) * ld + head * 64 + half * 8;
    const f16* dp = dO + ((int64_t)frame * S + qi) * lddo + head * 64 + half * 8;
    const f16* op = O + ((int64_t)frame * S + qi) * ldoo + head * 64 + half * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      qf[ks] = *(const f16x8*)(qp + ks * 16);
      dof[ks] = *(const f16x8*)(dp + ks * 16);
      const f16x8 of = *(const f16x8*)(op + ks * 16);
#pragma unroll
      for (int j = 0; j < 8; ++j) dl = fmaf((float)dof[ks][j], (float)of[j], dl);
    }
    dl += __shfl_xor(dl, 32);
  }

  // staging: natural tiles row t>>3, chunk t&7; transposed tile d = t>>2, chunk t&3
  const int nrow = t >> 3, nch = t & 7, td = t >> 2, tch = t & 3;
  const f16* kbase = qkv + (int64_t)frame * S * ld + C + head * 64 + nch * 8;
  const f16* vbase = kbase + C;
  const f16* ktbase = kT + (((int64_t)frame * heads + head) * 64 + td) * S_pad + tch * 8;
  const int ntiles = (S + 31) >> 5;
  f16x8 rk, rv, rt;
  auto fetch = [&](int kt, bool full) {
    int key = kt * 32 + nrow;
    key = key < S ? key : S - 1;
    rk = *(const f16x8*)(kbase + (int64_t)key * ld);
    if (full) {
      rv = *(const f16x8*)(vbase + (int64_t)key * ld);
      rt = *(const f16x8*)(ktbase + kt * 32);
    }
  };
  auto put = [&](int b, bool full) {
    char* base = smem + b * BUF;
    *(f16x8*)(base + lds_tile_off(nrow, nch)) = rk;
    if (full) {
      *(f16x8*)(base + NAT_BYTES + lds_tile_off(nrow, nch)) = rv;
      *(f16x8*)(base + 2 * NAT_BYTES + ldsT_off(td, tch)) = rt;
    }
  };
  int foff[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) foff[ks] = lds_tile_off(l31, 2 * ks + half);

  // ---- pass 1: log2-domain log-sum-exp of this lane's query over all keys ----
  float m = -INFINITY, l = 0.f;
  fetch(0, false);
  put(0, false);
  __syncthreads();
  for (int kt = 0; kt < ntiles; ++kt) {
    if (kt + 1 < ntiles) fetch(kt + 1, false);
    const char* Ks = smem + (kt & 1) * BUF;
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const f16x8 kf = *(const f16x8*)(Ks + foff[ks]);
      s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], s, 0, 0, 0);
    }
    const int kb = kt * 32 + 4 * half;
    float tmax = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = kb + (r & 3) + 8 * (r >> 2);
      s[r] = key < S ? s[r] * c : -INFINITY;
      tmax = fmaxf(tmax, s[r]);
    }
    if (tmax > -INFINITY) {
      const float mn = fmaxf(m, tmax);
      float acc = l * __builtin_amdgcn_exp2f(m - mn);
#pragma unroll
      for (int r = 0; r < 16; ++r) acc += __builtin_amdgcn_exp2f(s[r] - mn);
      l = acc;
      m = mn;
    }
    if (kt + 1 < ntiles) put((kt + 1) & 1, false);
    __syncthreads();
  }
  {
    const float mo = __shfl_xor(m, 32), lo = __shfl_xor(l, 32);
    const float M = fmaxf(m, mo);     // finite: every query sees key 0
    l = l * __builtin_amdgcn_exp2f(m - M) + lo * __builtin_amdgcn_exp2f(mo - M);
    m = M + __builtin_amdgcn_logf(l);   // v_log_f32 = log2
  }
  const float lse_q = m;
  if (q_ok && half == 0) {
    const int64_t o = ((int64_t)frame * heads + head) * S + qi;
    lse[o] = lse_q;
    delta[o] = dl;
  }

  // ---- pass 2: dQ^T[d][query] += K^T[d][key] dS^T[key][query] ----
  f32x16 dq[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) dq[0][r] = dq[1][r] = 0.f;
  fetch(0, true);
  put(0, true);
  __syncthreads();
  for (int kt = 0; kt < ntiles; ++kt) {
    if (kt + 1 < ntiles) fetch(kt + 1, true);
    const char* Ks = smem + (kt & 1) * BUF;
    const char* Vs = Ks + NAT_BYTES;
    const char* KTs = Ks + 2 * NAT_BYTES;
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = dp[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const f16x8 kf = *(const f16x8*)(Ks + foff[ks]);
      const f16x8 vf = *(const f16x8*)(Vs + foff[ks]);
      s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], s, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, dof[ks], dp, 0, 0, 0);
    }
    const int kb = kt * 32 + 4 * half;
    f16x8 dsf[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = kb + (r & 3) + 8 * (r >> 2);
      const float p = key < S ? __builtin_amdgcn_exp2f(fmaf(s[r], c, -lse_q)) : 0.f;
      dsf[r >> 3][r & 7] = (f16)(p * (dp[r] - dl));
    }
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const f16x8 ktf = *(const f16x8*)(KTs + ldsT_off(32 * dt + l31, 2 * s2 + half));
        dq[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ktf, dsf[s2], dq[dt], 0, 0, 0);
      }
    if (kt + 1 < ntiles) put((kt + 1) & 1, true);
    __syncthreads();
  }
  if (q_ok) {
    float* gp = dqkv + ((int64_t)frame * S + qi) * ldg + head * 64 + 4 * half;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = dq[dt][4 * g + e] * scale;
        *(f32x4*)(gp + 32 * dt + 8 * g) = v;
      }
  }
}

// ------------------------------------------------------------------------------------------------
// dK, dV.  grid (ceil(S/128), heads, frames), block 256.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_bwd_dkdv_kernel(
    const f16* __restrict__ qkv, int64_t ld, const f16* __restrict__ qT, const f16* __restrict__ dO,
    int64_t lddo, const f16* __restrict__ dOT, const float* __restrict__ lse, const float* __restrict__ delta,
    float* __restrict__ dqkv, int64_t ldg, int S, int S_pad, int heads, float scale) {
  // per buffer: Q natural | dO natural | Q^T | dO^T | lse[32] | delta[32]
  constexpr int BUF = 2 * NAT_BYTES + 2 * TR_BYTES + 256;
  __shared__ __attribute__((aligned(16))) char smem[2 * BUF];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int head = blockIdx.y, frame = blockIdx.z;
  const int C = heads * 64;
  const int k0 = blockIdx.x * 128 + wave * 32;
  const float c = scale * LOG2E_F;

  // stationary operands (B): key l31, d = 16 ks + 8 half + j
  int ki = k0 + l31;
  const bool k_ok = ki < S;
  ki = k_ok ? ki : S - 1;
  f16x8 kf[4], vf[4];
  {
    const f16* kp = qkv + ((int64_t)frame * S + ki) * ld + C + head * 64 + half * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      kf[ks] = *(const f16x8*)(kp + ks * 16);
      vf[ks] = *(const f16x8*)(kp + C + ks * 16);
    }
  }
  const int nrow = t >> 3, nch = t & 7, td = t >> 2, tch = t & 3;
  const f16* qbase = qkv + (int64_t)frame * S * ld + head * 64 + nch * 8;
  const f16* dobase = dO + (int64_t)frame * S * lddo + head * 64 + nch * 8;
  const int64_t fh = (int64_t)frame * heads + head;
  const f16* qtbase = qT + (fh * 64 + td) * S_pad + tch * 8;
  const f16* dotbase = dOT + (fh * 64 + td) * S_pad + tch * 8;
  const float* lbase = lse + fh * S;
  const float* dbase = delta + fh * S;
  const int ntiles = (S + 31) >> 5;
  f16x8 rq, rdo, rqt, rdot;
  float rl = 0.f, rd = 0.f;
  auto fetch = [&](int qt) {
    int q = qt * 32 + nrow;
    q = q < S ? q : S - 1;
    rq = *(const f16x8*)(qbase + (int64_t)q * ld);
    rdo = *(const f16x8*)(dobase + (int64_t)q * lddo);
    rqt = *(const f16x8*)(qtbase + qt * 32);
    rdot = *(const f16x8*)(dotbase + qt * 32);
    if (t < 32) {
      const int qq = qt * 32 + t;
      rl = qq < S ? lbase[qq] : INFINITY;   // exp2(s - inf) = 0: queries past the end contribute nothing
      rd = qq < S ? dbase[qq] : 0.f;
    }
  };
  auto put = [&](int b) {
    char* base = smem + b * BUF;
    *(f16x8*)(base + lds_tile_off(nrow, nch)) = rq;
    *(f16x8*)(base + NAT_BYTES + lds_tile_off(nrow, nch)) = rdo;
    *(f16x8*)(base + 2 * NAT_BYTES + ldsT_off(td, tch)) = rqt;
    *(f16x8*)(base + 2 * NAT_BYTES + TR_BYTES + ldsT_off(td, tch)) = rdot;
    if (t < 32) {
      ((float*)(base + 2 * NAT_BYTES + 2 * TR_BYTES))[t] = rl;
      ((float*)(base + 2 * NAT_BYTES + 2 * TR_BYTES + 128))[t] = rd;
    }
  };
  int foff[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) foff[ks] = lds_tile_off(l31, 2 * ks + half);

  f32x16 dk[2], dv[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) dk[0][r] = dk[1][r] = dv[0][r] = dv[1][r] = 0.f;
  fetch(0);
  put(0);
  __syncthreads();
  for (int qt = 0; qt < ntiles; ++qt) {
    if (qt + 1 < ntiles) fetch(qt + 1);
    const char* Qs = smem + (qt & 1) * BUF;
    const char* dOs = Qs + NAT_BYTES;
    const char* QTs = Qs + 2 * NAT_BYTES;
    const char* dOTs = QTs + TR_BYTES;
    const float* ls = (const float*)(dOTs + TR_BYTES);
    const float* ds_ = ls + 32;
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = dp[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const f16x8 qa = *(const f16x8*)(Qs + foff[ks]);
      const f16x8 da = *(const f16x8*)(dOs + foff[ks]);
      s = __builtin_amdgcn_mfma_f32_32x32x16_f16(qa, kf[ks], s, 0, 0, 0);      // S[query][key]
      dp = __builtin_amdgcn_mfma_f32_32x32x16_f16(da, vf[ks], dp, 0, 0, 0);    // dP[query][key]
    }
    f16x8 pf[2], dsf[2];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 l4 = *(const f32x4*)(ls + 8 * g + 4 * half);
      const f32x4 d4 = *(const f32x4*)(ds_ + 8 * g + 4 * half);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = 4 * g + e;
        const float p = __builtin_amdgcn_exp2f(fmaf(s[r], c, -l4[e]));
        pf[r >> 3][r & 7] = (f16)p;
        dsf[r >> 3][r & 7] = (f16)(p * (dp[r] - d4[e]));
      }
    }
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const int off = ldsT_off(32 * dt + l31, 2 * s2 + half);
        const f16x8 dot = *(const f16x8*)(dOTs + off);
        const f16x8 qtf = *(const f16x8*)(QTs + off);
        dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(dot, pf[s2], dv[dt], 0, 0, 0);    // dV^T[d][key]
        dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(qtf, dsf[s2], dk[dt], 0, 0, 0);   // dK^T[d][key]
      }
    if (qt + 1 < ntiles) put((qt + 1) & 1);
    __syncthreads();
  }
  if (k_ok) {
    float* gk = dqkv + ((int64_t)frame * S + ki) * ldg + C + head * 64 + 4 * half;
    float* gv = gk + C;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 a, b;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          a[e] = dk[dt][4 * g + e] * scale;
          b[e] = dv[dt][4 * g + e];
        }
        *(f32x4*)(gk + 32 * dt + 8 * g) = a;
        *(f32x4*)(gv + 32 * dt + 8 * g) = b;
      }
  }
}

}  // namespace

int gcd_attn_transpose_heads_launch(const f16* src, int64_t ld, int col0, int frames, int S, int heads, f16* out,
                                    int S_pad, hipStream_t s);   // attention.hip

extern "C" int64_t gcd_attn_spatial_bwd_ws_bytes(int frames, int S, int heads) {
  if (frames <= 0 || S <= 0 || heads <= 0) return 0;
  const int64_t S_pad = (S + 63) / 64 * 64;
  const int64_t fh = (int64_t)frames * heads;
  return 3 * fh * 64 * S_pad * 2 + 2 * fh * S * 4 + 256;
}

extern "C" int gcd_attn_spatial_bwd(const void* qkv16, int64_t ld, const void* out16, int64_t ldo,
                                    const void* dout16, int64_t lddo, void* dqkv32, int64_t ldg, void* ws,
                                    int64_t ws_bytes, int frames, int S, int heads, float scale, void* stream) {
  GCD_CHECK_ARG(qkv16 && out16 && dout16 && dqkv32 && ws, "gcd_attn_spatial_bwd: null pointer");
  GCD_CHECK_ARG(frames > 0 && S > 0 && heads > 0, "gcd_attn_spatial_bwd: empty problem");
  GCD_CHECK_ARG(frames <= 65535 && heads <= 65535, "gcd_attn_spatial_bwd: grid too large");
  const int C = heads * 64;
  GCD_CHECK_ARG(ld % 8 == 0 && ld >= 3 * C && ldo % 8 == 0 && ldo >= C && lddo % 8 == 0 && lddo >= C &&
                    ldg % 4 == 0 && ldg >= 3 * C,
                "gcd_attn_spatial_bwd: bad leading dimensions (ld=%lld ldo=%lld lddo=%lld ldg=%lld for C=%d)",
                (long long)ld, (long long)ldo, (long long)lddo, (long long)ldg, C);
  GCD_CHECK_ARG((((uintptr_t)qkv16 | (uintptr_t)out16 | (uintptr_t)dout16 | (uintptr_t)dqkv32 | (uintptr_t)ws) & 15) == 0,
                "gcd_attn_spatial_bwd: pointers must be 16-byte aligned");
  GCD_CHECK_ARG(ws_bytes >= gcd_attn_spatial_bwd_ws_bytes(frames, S, heads),
                "gcd_attn_spatial_bwd: workspace of %lld bytes, need %lld", (long long)ws_bytes,
                (long long)gcd_attn_spatial_bwd_ws_bytes(frames, S, heads));
  hipStream_t s = (hipStream_t)stream;
  const int S_pad = (S + 63) / 64 * 64;
  const int64_t fh = (int64_t)frames * heads;
  f16* kT = (f16*)ws;
  f16* qT = kT + fh * 64 * S_pad;
  f16* dOT = qT + fh * 64 * S_pad;
  float* lse = (float*)(dOT + fh * 64 * S_pad);
  float* delta = lse + fh * S;
  if (int rc = gcd_attn_transpose_heads_launch((const f16*)qkv16, ld, C, frames, S, heads, kT, S_pad, s)) return rc;
  if (int rc = gcd_attn_transpose_heads_launch((const f16*)qkv16, ld, 0, frames, S, heads, qT, S_pad, s)) return rc;
  if (int rc = gcd_attn_transpose_heads_launch((const f16*)dout16, lddo, 0, frames, S, heads, dOT, S_pad, s)) return rc;
  const dim3 grid((S + 127) / 128, heads, frames);
  hipLaunchKernelGGL(attn_bwd_dq_kernel, grid, dim3(256), 0, s, (const f16*)qkv16, ld, (const f16*)kT,
                     (const f16*)dout16, lddo, (const f16*)out16, ldo, (float*)dqkv32, ldg, lse, delta, S, S_pad,
                     heads, scale);
  GCD_CHECK_LAUNCH();
  hipLaunchKernelGGL(attn_bwd_dkdv_kernel, grid, dim3(256), 0, s, (const f16*)qkv16, ld, (const f16*)qT,
                     (const f16*)dout16, lddo, (const f16*)dOT, (const float*)lse, (const float*)delta,
                     (float*)dqkv32, ldg, S, S_pad, heads, scale);
  GCD_CHECK_LAUNCH();
  return 0;
}
