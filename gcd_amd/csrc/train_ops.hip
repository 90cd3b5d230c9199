// train_ops.hip — libgcd_amd_train.so, the fine-tune step's helper kernels for the PLANNED engine (gcd_amd/train_plan.py,
// round 5): what the round-4 step did with thousands of torch fills / copies / casts / transposes / flips / lerps per
// step runs here as a handful of launches.  C ABI: include/gcd_amd_train.h.  All HBM-bound.
//
//   gcd_train_pack_weights   ONE launch turns every fp32 parameter of the network into the 16-bit operand forms the
//                            forward AND backward GEMMs read (forward form, transposed form, tap-mirrored dgrad form):
//                            replaces ~900 casts + 870 transposes + flips + permuted copies per step
//   gcd_blend_fwd / _bwd     AlphaBlender (util.py:358-369) and its backward incl. the mix-factor gradient partials
//   gcd_smallm_*             the network's few-row fp32 Linears (emb_layers of the 44 ResBlocks, the one-key
//                            cross-attention chains, time_pos_embed, the embedding MLPs) GROUPED: one launch runs a table
//                            of independent y = act(x) W^T + b problems of <= 32 rows; dgrad and wgrad likewise
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/gcd_amd_train.h"

typedef _Float16 f16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

extern "C" const char* gcd_train_last_error(void);
void gcd_train_set_error(const char* fmt, ...);

#define T_CHECK_ARG(cond, ...)          \
  do {                                  \
    if (!(cond)) {                      \
      gcd_train_set_error(__VA_ARGS__); \
      return 2;                         \
    }                                   \
  } while (0)
#define T_CHECK_LAUNCH(what)                                                        \
  do {                                                                              \
    hipError_t e_ = hipGetLastError();                                              \
    if (e_ != hipSuccess) {                                                         \
      gcd_train_set_error("%s: launch failed: %s", what, hipGetErrorString(e_));    \
      return 1;                                                                     \
    }                                                                               \
  } while (0)

namespace {

__device__ __forceinline__ unsigned short to16(float v, bool bf16) {
  if (bf16) {
    unsigned u = __float_as_uint(v);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);   // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);                                                  // round to nearest even
    return (unsigned short)(u >> 16);
  }
  const f16 h = (f16)v;
  return __builtin_bit_cast(unsigned short, h);
}

// ---------------------------------------------------------------------------------------------------------------------
// Multi-tensor weight pack.  Entry e: fp32 parameter viewed as [N][C][taps] (Linear / 1x1: taps = 1; Conv2d 3x3: 9;
// Conv3d (3,1,1): 3).  One workgroup = a 32 (n) x 32 (c) x taps tile: read once (coalesced along (c, tap)), rounded once,
// written in up to two layouts from LDS.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int PT = 32, PPAD = PT + 1, PMAXT = 9;

__global__ __launch_bounds__(256) void pack_weights_kernel(const gcd_pack_entry* __restrict__ tab, int n_entries, int bf16) {
  __shared__ unsigned short lds[PMAXT * PT * PPAD];
  // entry of this workgroup: the last one whose first tile is <= blockIdx.x
  int lo = 0, hi = n_entries - 1;
  const int b = (int)blockIdx.x;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (tab[mid].tile0 <= b) lo = mid;
    else hi = mid - 1;
  }
  const gcd_pack_entry e = tab[lo];
  const int tile = b - e.tile0;
  const int tn = tile / e.tiles_c, tc = tile - tn * e.tiles_c;
  const int n0 = tn * PT, c0 = tc * PT;
  const int taps = e.taps;
  const int seg = PT * taps;                       // floats per n of this tile's source segment
  const int cw = min(PT, e.C - c0), nw = min(PT, e.N - n0);
  const int t = threadIdx.x;
  for (int i = t; i < PT * seg; i += 256) {
    const int nl = i / seg, r = i - nl * seg;
    const int cl = r / taps, tap = r - cl * taps;
    float v = 0.f;
    if (nl < nw && cl < cw) v = e.src[((int64_t)(n0 + nl) * e.C + c0) * taps + r];
    lds[(tap * PT + nl) * PPAD + cl] = to16(v, bf16 != 0);
  }
  __syncthreads();
  // two 16-bit elements (4 bytes) per store where the pair is inside the tile and its address is 4-byte aligned (every
  // full tile of the network's shapes); single elements at ragged edges / odd strides
  if (e.dst_f) {
    unsigned short* d = (unsigned short*)e.dst_f;
    const bool pair_ok = ((e.f_ns | e.f_ts | c0) & 1) == 0 && (((uintptr_t)d) & 3) == 0 && (cw & 1) == 0;
    for (int i = t; i < taps * PT * (PT / 2); i += 256) {
      const int cl = (i & (PT / 2 - 1)) * 2, nl = (i >> 4) & (PT - 1), tap = i >> 9;
      if (nl < nw && cl < cw) {
        unsigned short* q = d + (int64_t)(n0 + nl) * e.f_ns + (int64_t)tap * e.f_ts + c0 + cl;
        const unsigned short v0 = lds[(tap * PT + nl) * PPAD + cl], v1 = lds[(tap * PT + nl) * PPAD + cl + 1];
        if (pair_ok) {
          *(unsigned*)q = (unsigned)v0 | ((unsigned)v1 << 16);
        } else {
          q[0] = v0;
          if (cl + 1 < cw) q[1] = v1;
        }
      }
    }
  }
  if (e.dst_t) {
    unsigned short* d = (unsigned short*)e.dst_t;
    const bool pair_ok = ((e.t_cs | e.t_ts | n0) & 1) == 0 && (((uintptr_t)d) & 3) == 0 && (nw & 1) == 0;
    for (int i = t; i < taps * PT * (PT / 2); i += 256) {
      const int nl = (i & (PT / 2 - 1)) * 2, cl = (i >> 4) & (PT - 1), tap = i >> 9;
      const int tx = e.mirror ? taps - 1 - tap : tap;
      if (nl < nw && cl < cw) {
        unsigned short* q = d + (int64_t)(c0 + cl) * e.t_cs + (int64_t)tx * e.t_ts + n0 + nl;
        const unsigned short v0 = lds[(tap * PT + nl) * PPAD + cl], v1 = lds[(tap * PT + nl + 1) * PPAD + cl];
        if (pair_ok) {
          *(unsigned*)q = (unsigned)v0 | ((unsigned)v1 << 16);
        } else {
          q[0] = v0;
          if (nl + 1 < nw) q[1] = v1;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// AlphaBlender: y = a xs + (1 - a) xt, a per frame (rows_per_frame tokens each); C % 4 == 0, 16-byte aligned rows
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void blend_fwd_kernel(const float* __restrict__ xs, int64_t lds_, const float* __restrict__ xt,
                                                        int64_t ldt, const float* __restrict__ alpha, int64_t M, int C4,
                                                        int64_t rows, float* __restrict__ y, int64_t ldy) {
  const int64_t total = M * C4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t m = i / C4;
    const int c = (int)(i - m * C4) * 4;
    const float a = alpha[m / rows];
    const f32x4 s = *(const f32x4*)(xs + m * lds_ + c), u = *(const f32x4*)(xt + m * ldt + c);
    *(f32x4*)(y + m * ldy + c) = a * s + (1.0f - a) * u;
  }
}

// d_xs (+)= a dy, d_xt = (1 - a) dy, d_alpha[frame] += sum dy (xs - xt)   (d_alpha zeroed by the caller)
// grid: (chunks per frame, frames)
__global__ __launch_bounds__(256) void blend_bwd_kernel(const float* __restrict__ dy, int64_t lddy, const float* __restrict__ xs,
                                                        int64_t lds_, const float* __restrict__ xt, int64_t ldt,
                                                        const float* __restrict__ alpha, int C4, int64_t rows,
                                                        float* __restrict__ dxs, int64_t lddxs, int acc_xs,
                                                        float* __restrict__ dxt, int64_t lddxt, float* __restrict__ dalpha) {
  const int frame = blockIdx.y;
  const float a = alpha[frame];
  const int64_t total = rows * C4;
  const int64_t m0 = (int64_t)frame * rows;
  float part = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / C4;
    const int c = (int)(i - r * C4) * 4;
    const int64_t m = m0 + r;
    const f32x4 g = *(const f32x4*)(dy + m * lddy + c);
    if (dalpha) {
      const f32x4 d = *(const f32x4*)(xs + m * lds_ + c) - *(const f32x4*)(xt + m * ldt + c);
      part += g[0] * d[0] + g[1] * d[1] + g[2] * d[2] + g[3] * d[3];
    }
    if (dxs) {
      f32x4 o = a * g;
      if (acc_xs) o += *(const f32x4*)(dxs + m * lddxs + c);
      *(f32x4*)(dxs + m * lddxs + c) = o;
    }
    *(f32x4*)(dxt + m * lddxt + c) = (1.0f - a) * g;
  }
  if (dalpha) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off, 64);
    __shared__ float ws[4];
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = part;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(dalpha + frame, ws[0] + ws[1] + ws[2] + ws[3]);
  }
}

// GroupNorm affine gradients from gcd_groupnorm_bwd's AB[inst][c][2] (fp64: sum dz, sum dz xhat): dbeta[c] = sum_inst AB[.][c][0],
// dgamma[c] = sum_inst AB[.][c][1]
__global__ __launch_bounds__(256) void gn_affine_grads_kernel(const double* __restrict__ AB, int ninst, int C,
                                                             float* __restrict__ dgamma, float* __restrict__ dbeta, int acc) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  double a = 0.0, b = 0.0;
  for (int i = 0; i < ninst; ++i) {
    a += AB[((int64_t)i * C + c) * 2];
    b += AB[((int64_t)i * C + c) * 2 + 1];
  }
  dbeta[c] = acc ? dbeta[c] + (float)a : (float)a;
  dgamma[c] = acc ? dgamma[c] + (float)b : (float)b;
}

// ---------------------------------------------------------------------------------------------------------------------
// Grouped few-row Linears, fp32 (weights are streamed once; nothing here is matrix-pipe work: <= 32 rows).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int SM_MAXM = 32;

__device__ __forceinline__ float silu_f(float v) { return v / (1.0f + __expf(-v)); }
__device__ __forceinline__ float dsilu_f(float v) {
  const float s = 1.0f / (1.0f + __expf(-v));
  return s * (1.0f + v * (1.0f - s));
}

__device__ __forceinline__ gcd_smallm_problem smallm_find(const gcd_smallm_problem* __restrict__ tab, int n_prob, int b) {
  int lo = 0, hi = n_prob - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (tab[mid].block0 <= b) lo = mid;
    else hi = mid - 1;
  }
  return tab[lo];
}

// forward: y[m][n] = sum_k act(x[m][k]) W[n][k] + b[n].  A workgroup owns 16 output columns (4 per wave) and walks K in
// chunks of 256: the chunk of x (activation applied once) sits in LDS, a lane owns 4 consecutive k of the chunk, so every
// weight is read exactly once, 16 bytes per lane, coalesced.  blocks of a problem: ceil(N / 16).
constexpr int SM_KC = 256, SM_NB = 16;
__global__ __launch_bounds__(256) void smallm_fwd_kernel(const gcd_smallm_problem* __restrict__ tab, int n_prob) {
  __shared__ float xs[SM_MAXM * SM_KC];
  const gcd_smallm_problem p = smallm_find(tab, n_prob, (int)blockIdx.x);
  const int M = p.M, K = p.K;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int nb = ((int)blockIdx.x - p.block0) * SM_NB + 4 * wave;
  float acc[4][SM_MAXM];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int m = 0; m < SM_MAXM; ++m) acc[j][m] = 0.f;
  for (int kc = 0; kc < K; kc += SM_KC) {
    __syncthreads();
    for (int i = threadIdx.x; i < M * SM_KC; i += 256) {
      const int m = i >> 8, k = kc + (i & 255);
      float v = 0.f;
      if (k < K) {
        v = p.x[(int64_t)m * p.ldx + k];
        if (p.flags & 1) v = silu_f(v);
      }
      xs[i] = v;
    }
    __syncthreads();
    const int k = kc + 4 * lane;
    if (k < K) {          // K % 4 == 0
      f32x4 wv[4];
#pragma unroll
      for (int j = 0; j < 4; ++j)
        wv[j] = nb + j < p.N ? *(const f32x4*)(p.W + (int64_t)(nb + j) * K + k) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int m = 0; m < SM_MAXM; ++m)
        if (m < M) {
          const f32x4 xv = *(const f32x4*)(xs + m * SM_KC + 4 * lane);
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[j][m] += wv[j][0] * xv[0] + wv[j][1] * xv[1] + wv[j][2] * xv[2] + wv[j][3] * xv[3];
        }
    }
  }
  // 128 per-lane partial sums (4 columns x 32 rows) -> 128 totals over the 64 lanes by a butterfly REDUCE-SCATTER: at
  // every step a lane keeps one half of its values and sends the other half to its partner (126 cross-lane moves instead
  // of the 768 of 128 independent butterflies); lane l ends with the totals of indices 2 l and 2 l + 1, index = 32 j + m.
  float v[128];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int m = 0; m < SM_MAXM; ++m) v[32 * j + m] = acc[j][m];
#define SM_RS_STEP(N2, MASK)                                     \
  {                                                              \
    const bool up = (lane & (MASK)) != 0;                        \
    _Pragma("unroll") for (int i = 0; i < (N2); ++i) {           \
      const float keep = up ? v[i + (N2)] : v[i];                \
      const float send = up ? v[i] : v[i + (N2)];                \
      v[i] = keep + __shfl_xor(send, (MASK), 64);                \
    }                                                            \
  }
  SM_RS_STEP(64, 32)
  SM_RS_STEP(32, 16)
  SM_RS_STEP(16, 8)
  SM_RS_STEP(8, 4)
  SM_RS_STEP(4, 2)
  SM_RS_STEP(2, 1)
#undef SM_RS_STEP
  {
    const int j = lane >> 4;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int m = (2 * lane + e) & 31;
      if (m < M && nb + j < p.N) {
        const float r = v[e] + (p.b ? p.b[nb + j] : 0.f);
        float* dst = p.y + (int64_t)m * p.ldy + nb + j;
        *dst = (p.flags & 4) ? *dst + r : r;
      }
    }
  }
}

// dgrad: dx[m][k] (+)= dact(x[m][k]) * sum_n dy[m][n] W[n][k].  Thread per k (every W row read coalesced along k), a
// slice of 64 columns of dy in LDS (broadcast reads); workgroup = (k chunk of 256, n slice of 64).  flags: 1 = multiply by
// silu'(x) (x = the forward's PRE-activation input), 4 = atomicAdd into dx (several n slices, or several problems, share
// one dx: zeroed — or holding the value to add onto — by the caller) instead of a plain store (only legal when the problem
// has ONE n slice).  blocks of a problem: ceil(K / 256) * ceil(N / 64).
constexpr int SM_NS = 64;
__global__ __launch_bounds__(256) void smallm_dgrad_kernel(const gcd_smallm_problem* __restrict__ tab, int n_prob) {
  __shared__ float dys[SM_MAXM * SM_NS];
  const gcd_smallm_problem p = smallm_find(tab, n_prob, (int)blockIdx.x);
  const int M = p.M, K = p.K, N = p.N;
  const int rel = (int)blockIdx.x - p.block0;
  const int kblocks = (K + 255) / 256;
  const int ns = rel / kblocks, kb = rel - ns * kblocks;
  const int n0 = ns * SM_NS, nn = min(N - n0, SM_NS);
  for (int i = threadIdx.x; i < M * SM_NS; i += 256) {
    const int m = i >> 6, n = i & 63;
    dys[i] = n < nn ? p.y[(int64_t)m * p.ldy + n0 + n] : 0.f;        // (p.y = dy here)
  }
  __syncthreads();
  const int k = kb * 256 + threadIdx.x;
  if (k >= K) return;
  float acc[SM_MAXM];
#pragma unroll
  for (int m = 0; m < SM_MAXM; ++m) acc[m] = 0.f;
  for (int n = 0; n < nn; ++n) {
    const float w = p.W[(int64_t)(n0 + n) * K + k];
#pragma unroll
    for (int m = 0; m < SM_MAXM; ++m)
      if (m < M) acc[m] += dys[m * SM_NS + n] * w;
  }
#pragma unroll
  for (int m = 0; m < SM_MAXM; ++m)
    if (m < M) {
      float v = acc[m];
      if (p.flags & 1) v *= dsilu_f(p.x[(int64_t)m * p.ldx + k]);
      float* dst = p.dx + (int64_t)m * p.lddx + k;
      if (p.flags & 4) atomicAdd(dst, v);
      else *dst = v;
    }
}

// wgrad: dW[n][k] (+)= sum_m dy[m][n] act(x[m][k]);  db[n] (+)= sum_m dy[m][n].  Thread per k with its column of x in
// registers, a slice of 64 rows n of dy in LDS; every dW row is written coalesced, once.  flags: 1 = act is SiLU,
// 8 = accumulate onto dW / db.  blocks of a problem: ceil(K / 256) * ceil(N / 64).
__global__ __launch_bounds__(256) void smallm_wgrad_kernel(const gcd_smallm_problem* __restrict__ tab, int n_prob) {
  __shared__ float dys[SM_MAXM * SM_NS];
  const gcd_smallm_problem p = smallm_find(tab, n_prob, (int)blockIdx.x);
  const int M = p.M, K = p.K, N = p.N;       // M: any number of rows, walked 32 at a time
  const int rel = (int)blockIdx.x - p.block0;
  const int kblocks = (K + 255) / 256;
  const int ns = rel / kblocks, kb = rel - ns * kblocks;
  const int n0 = ns * SM_NS, nn = min(N - n0, SM_NS);
  const int k = kb * 256 + threadIdx.x;
  for (int m0 = 0; m0 < M; m0 += SM_MAXM) {
    const int mm = min(SM_MAXM, M - m0);
    const bool add = (p.flags & 8) || m0 > 0;
    __syncthreads();
    for (int i = threadIdx.x; i < SM_MAXM * SM_NS; i += 256) {
      const int m = i >> 6, n = i & 63;
      dys[i] = (n < nn && m < mm) ? p.y[(int64_t)(m0 + m) * p.ldy + n0 + n] : 0.f;
    }
    __syncthreads();
    if (kb == 0 && p.db && (int)threadIdx.x < nn) {
      float sdb = 0.f;
      for (int m = 0; m < mm; ++m) sdb += dys[m * SM_NS + threadIdx.x];
      float* dst = p.db + n0 + threadIdx.x;
      *dst = add ? *dst + sdb : sdb;
    }
    if (k < K) {
      float xr[SM_MAXM];
#pragma unroll
      for (int m = 0; m < SM_MAXM; ++m) {
        float v = 0.f;
        if (m < mm) {
          v = p.x[(int64_t)(m0 + m) * p.ldx + k];
          if (p.flags & 1) v = silu_f(v);
        }
        xr[m] = v;
      }
      for (int n = 0; n < nn; ++n) {
        float a = 0.f;
#pragma unroll
        for (int m = 0; m < SM_MAXM; ++m) a += dys[m * SM_NS + n] * xr[m];      // (rows >= mm are zeros on both sides)
        float* dst = p.dW + (int64_t)(n0 + n) * K + k;
        *dst = add ? *dst + a : a;
      }
    }
  }
}

}  // namespace

extern "C" int gcd_train_pack_weights(const gcd_pack_entry* table_dev, int n_entries, int total_tiles, int bf16,
                                      void* stream) {
  T_CHECK_ARG(table_dev && n_entries > 0 && total_tiles > 0, "gcd_train_pack_weights: empty table");
  hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)total_tiles), dim3(256), 0, (hipStream_t)stream, table_dev,
                     n_entries, bf16);
  T_CHECK_LAUNCH("gcd_train_pack_weights");
  return 0;
}

extern "C" int gcd_blend_fwd_f32(const float* xs, int64_t ld_s, const float* xt, int64_t ld_t, const float* alpha, int64_t M,
                                 int C, int64_t rows_per_frame, float* y, int64_t ld_y, void* stream) {
  T_CHECK_ARG(xs && xt && alpha && y, "gcd_blend_fwd_f32: null pointer");
  T_CHECK_ARG(M > 0 && C > 0 && C % 4 == 0 && rows_per_frame > 0 && M % rows_per_frame == 0 && ld_s % 4 == 0 &&
                  ld_t % 4 == 0 && ld_y % 4 == 0,
              "gcd_blend_fwd_f32: M=%lld C=%d rows=%lld", (long long)M, C, (long long)rows_per_frame);
  int64_t blocks = (M * (C / 4) + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(blend_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, xs, ld_s, xt, ld_t, alpha,
                     M, C / 4, rows_per_frame, y, ld_y);
  T_CHECK_LAUNCH("gcd_blend_fwd_f32");
  return 0;
}

extern "C" int gcd_blend_bwd_f32(const float* dy, int64_t ld_dy, const float* xs, int64_t ld_s, const float* xt, int64_t ld_t,
                                 const float* alpha, int64_t M, int C, int64_t rows_per_frame, float* d_xs, int64_t ld_dxs,
                                 int accumulate_xs, float* d_xt, int64_t ld_dxt, float* d_alpha_zeroed, void* stream) {
  T_CHECK_ARG(dy && alpha && d_xt && (!d_alpha_zeroed || (xs && xt)), "gcd_blend_bwd_f32: null pointer");   // (d_xs optional)
  T_CHECK_ARG(M > 0 && C > 0 && C % 4 == 0 && rows_per_frame > 0 && M % rows_per_frame == 0 && ld_dy % 4 == 0 &&
                  ld_dxs % 4 == 0 && ld_dxt % 4 == 0 && (!d_alpha_zeroed || (ld_s % 4 == 0 && ld_t % 4 == 0)),
              "gcd_blend_bwd_f32: M=%lld C=%d rows=%lld", (long long)M, C, (long long)rows_per_frame);
  const int64_t frames = M / rows_per_frame;
  T_CHECK_ARG(frames < 65536, "gcd_blend_bwd_f32: %lld frames", (long long)frames);
  int64_t chunks = (rows_per_frame * (C / 4) + 255) / 256;
  const int64_t cap = frames >= 512 ? 1 : (4096 + frames - 1) / frames;
  if (chunks > cap) chunks = cap;
  hipLaunchKernelGGL(blend_bwd_kernel, dim3((unsigned)chunks, (unsigned)frames), dim3(256), 0, (hipStream_t)stream, dy, ld_dy,
                     xs, ld_s, xt, ld_t, alpha, C / 4, rows_per_frame, d_xs, ld_dxs, accumulate_xs, d_xt, ld_dxt,
                     d_alpha_zeroed);
  T_CHECK_LAUNCH("gcd_blend_bwd_f32");
  return 0;
}

extern "C" int gcd_gn_affine_grads(const double* AB, int ninst, int C, float* dgamma, float* dbeta, int accumulate,
                                   void* stream) {
  T_CHECK_ARG(AB && dgamma && dbeta && ninst > 0 && C > 0, "gcd_gn_affine_grads: bad arguments");
  hipLaunchKernelGGL(gn_affine_grads_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, (hipStream_t)stream, AB, ninst, C,
                     dgamma, dbeta, accumulate);
  T_CHECK_LAUNCH("gcd_gn_affine_grads");
  return 0;
}

static int smallm_launch(const char* what, void (*fn)(const gcd_smallm_problem*, int), const gcd_smallm_problem* tab,
                         int n_prob, int total_blocks, void* stream) {
  if (!(tab && n_prob > 0 && total_blocks > 0)) {
    gcd_train_set_error("%s: empty table", what);
    return 2;
  }
  hipLaunchKernelGGL(fn, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, tab, n_prob);
  T_CHECK_LAUNCH(what);
  return 0;
}

extern "C" int gcd_smallm_fwd(const gcd_smallm_problem* table_dev, int n_prob, int total_blocks, void* stream) {
  return smallm_launch("gcd_smallm_fwd", smallm_fwd_kernel, table_dev, n_prob, total_blocks, stream);
}
extern "C" int gcd_smallm_dgrad(const gcd_smallm_problem* table_dev, int n_prob, int total_blocks, void* stream) {
  return smallm_launch("gcd_smallm_dgrad", smallm_dgrad_kernel, table_dev, n_prob, total_blocks, stream);
}
extern "C" int gcd_smallm_wgrad(const gcd_smallm_problem* table_dev, int n_prob, int total_blocks, void* stream) {
  return smallm_launch("gcd_smallm_wgrad", smallm_wgrad_kernel, table_dev, n_prob, total_blocks, stream);
}
