// backward.hip — kernels of the fine-tune step's backward pass (BASELINE.json cfg4; SURVEY.md §8a a23,
// §8(f)-2: StandardDiffusionLoss forward + backward of the VideoUNet, loss.py:115-273,
// openaimodel.py:326-357, attention.py:544-572, video_attention.py:104-140, diffusion.py:412-431).
//
// The contractions of the backward pass (dgrad / wgrad of every Linear and convolution, and the
// S x S products of the attention backward) run on the same MFMA GEMM kernels as the forward
// (gcd_gemm_f16 with transposed operands); this file holds what surrounds them:
//   * im2col / col2im for the 3x3 and (3,1,1) convolutions (wgrad operand, dgrad scatter as a gather),
//   * GroupNorm(+SiLU) and LayerNorm backward, GEGLU forward / backward on the fp32 pre-activations,
//   * softmax backward over score rows, the temporal (T <= 16 tokens) attention backward,
//   * row-block sums (bias / per-frame vector gradients), scaled fp32 -> fp16 casts, fused Adam.
// All are HBM-bound streaming kernels: 16-byte accesses, fp32 (fp64 where sums are long) accumulation.
#include "common.h"

namespace {

__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + __expf(-x)); }
// d/dz silu(z) = s (1 + z (1 - s)),  s = sigmoid(z)
__device__ __forceinline__ float dsilu_f(float z) {
  const float s = sigmoid_f(z);
  return s * (1.0f + z * (1.0f - s));
}
// exact-erf GELU and its derivative: Phi(g) + g phi(g)
__device__ __forceinline__ float dgelu_f(float g) {
  const float cdf = 0.5f * (1.0f + erff(g * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * __expf(-0.5f * g * g);
  return cdf + g * pdf;
}

// ------------------------------------------------------------------------------------------------
// im2col / col2im, 3x3 taps, K order (kh, kw, cin) = packing.pack_conv3x3.
// Output pixel (y, x), tap (kh, kw) reads input pixel
//   upsample:  u = (y + kh - 1, x + kw - 1) in the 2x grid -> (u >> 1)        (stride 1)
//   otherwise: (y * stride + kh - 1 + asym, x * stride + kw - 1 + asym)
// ------------------------------------------------------------------------------------------------
struct ConvGeo {
  int Cin, Hi, Wi, Ho, Wo, stride, up, asym;
};

__global__ __launch_bounds__(256) void im2col3x3_kernel(const f16* __restrict__ x, int64_t ldx,
                                                        f16* __restrict__ col, ConvGeo g, int64_t M) {
  const int cv = g.Cin >> 3;
  const int64_t total = M * 9 * cv;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int c8 = (int)(idx % cv);
    const int64_t r = idx / cv;
    const int tap = (int)(r % 9);
    const int64_t m = r / 9;
    const int hw = g.Ho * g.Wo;
    const int n = (int)(m / hw);
    const int rem = (int)(m - (int64_t)n * hw);
    const int y = rem / g.Wo, xo = rem - y * g.Wo;
    const int kh = tap / 3, kw = tap - 3 * kh;
    int iy, ix;
    bool ok;
    if (g.up) {
      const int uy = y + kh - 1, ux = xo + kw - 1;
      ok = uy >= 0 && uy < g.Ho && ux >= 0 && ux < g.Wo;
      iy = uy >> 1;
      ix = ux >> 1;
    } else {
      iy = y * g.stride + kh - 1 + g.asym;
      ix = xo * g.stride + kw - 1 + g.asym;
      ok = iy >= 0 && iy < g.Hi && ix >= 0 && ix < g.Wi;
    }
    f16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (ok) v = *(const f16x8*)(x + ((int64_t)n * g.Hi * g.Wi + (int64_t)iy * g.Wi + ix) * ldx + c8 * 8);
    *(f16x8*)(col + (m * 9 + tap) * g.Cin + c8 * 8) = v;
  }
}

// dx[pixel][c] = sum over the (output pixel, tap) pairs that read this input pixel of dcol (a gather:
// deterministic, no atomics).
__global__ __launch_bounds__(256) void col2im3x3_kernel(const float* __restrict__ dcol,
                                                        float* __restrict__ dx, int64_t lddx, ConvGeo g,
                                                        int64_t Min) {
  const int cv = g.Cin >> 2;
  const int64_t total = Min * cv;
  const int K9 = 9 * g.Cin;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int c4 = (int)(idx % cv);
    const int64_t pix = idx / cv;
    const int hwi = g.Hi * g.Wi;
    const int n = (int)(pix / hwi);
    const int rem = (int)(pix - (int64_t)n * hwi);
    const int iy = rem / g.Wi, ix = rem - iy * g.Wi;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int kh = 0; kh < 3; ++kh) {
      // output rows y whose tap kh reads input row iy
      int ys[2], ny = 0;
      if (g.up) {
        for (int d = 0; d < 2; ++d) {
          const int y = 2 * iy + d - kh + 1;
          if (y >= 0 && y < g.Ho) ys[ny++] = y;
        }
      } else {
        const int t = iy - kh + 1 - g.asym;
        if (t >= 0 && t % g.stride == 0 && t / g.stride < g.Ho) ys[ny++] = t / g.stride;
      }
      for (int kw = 0; kw < 3; ++kw) {
        int xs[2], nx = 0;
        if (g.up) {
          for (int d = 0; d < 2; ++d) {
            const int xx = 2 * ix + d - kw + 1;
            if (xx >= 0 && xx < g.Wo) xs[nx++] = xx;
          }
        } else {
          const int t = ix - kw + 1 - g.asym;
          if (t >= 0 && t % g.stride == 0 && t / g.stride < g.Wo) xs[nx++] = t / g.stride;
        }
        for (int a = 0; a < ny; ++a)
          for (int b = 0; b < nx; ++b) {
            const int64_t m = (int64_t)n * g.Ho * g.Wo + (int64_t)ys[a] * g.Wo + xs[b];
            acc += *(const f32x4*)(dcol + m * K9 + (kh * 3 + kw) * g.Cin + c4 * 4);
          }
      }
    }
    *(f32x4*)(dx + pix * lddx + c4 * 4) = acc;
  }
}

// (3,1,1) temporal taps, K order (kt, cin) = packing.pack_conv_t3; rows are (clip, t, hw).
__global__ __launch_bounds__(256) void im2col_t3_kernel(const f16* __restrict__ x, int64_t ldx,
                                                        f16* __restrict__ col, int C, int T, int HW,
                                                        int64_t M) {
  const int cv = C >> 3;
  const int64_t total = M * 3 * cv;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int c8 = (int)(idx % cv);
    const int64_t r = idx / cv;
    const int kt = (int)(r % 3);
    const int64_t m = r / 3;
    const int t = (int)((m / HW) % T);
    const int tt = t + kt - 1;
    f16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (tt >= 0 && tt < T) v = *(const f16x8*)(x + (m + (int64_t)(kt - 1) * HW) * ldx + c8 * 8);
    *(f16x8*)(col + (m * 3 + kt) * C + c8 * 8) = v;
  }
}

__global__ __launch_bounds__(256) void col2im_t3_kernel(const float* __restrict__ dcol,
                                                        float* __restrict__ dx, int64_t lddx, int C,
                                                        int T, int HW, int64_t M) {
  const int cv = C >> 2;
  const int64_t total = M * cv;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int c4 = (int)(idx % cv);
    const int64_t m = idx / cv;
    const int t = (int)((m / HW) % T);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int kt = 0; kt < 3; ++kt) {
      const int to = t - (kt - 1);          // the output frame whose tap kt reads frame t
      if (to >= 0 && to < T)
        acc += *(const f32x4*)(dcol + ((m - (int64_t)(kt - 1) * HW) * 3 + kt) * C + c4 * 4);
    }
    *(f32x4*)(dx + m * lddx + c4 * 4) = acc;
  }
}

// ------------------------------------------------------------------------------------------------
// out[b][n] += sum over rows b*rows .. b*rows+rows-1 of x[row][n]  (out zeroed by the caller).
// grid (ceil(N/64), nblk, nsplit), block 256 = 64 columns x 4 row lanes; fp64 in registers, one fp32
// atomic per (block, column).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rowblock_sum_kernel(const float* __restrict__ x, int64_t ldx,
                                                           int64_t rows, int N, float* __restrict__ out) {
  __shared__ double red[4][256];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int col = (blockIdx.x * 64 + cl) * 4;          // four columns per lane: 1 KB per wave and row
  const int64_t base = (int64_t)blockIdx.y * rows;
  double s[4] = {0.0, 0.0, 0.0, 0.0};
  if (col < N)
    for (int64_t r = (int64_t)blockIdx.z * 4 + rl; r < rows; r += (int64_t)gridDim.z * 4) {
      const f32x4 v = *(const f32x4*)(x + (base + r) * ldx + col);
#pragma unroll
      for (int e = 0; e < 4; ++e) s[e] += (double)v[e];
    }
#pragma unroll
  for (int e = 0; e < 4; ++e) red[rl][4 * cl + e] = s[e];
  __syncthreads();
  const int c = threadIdx.x, oc = blockIdx.x * 256 + c;
  if (oc < N) atomicAdd(out + (int64_t)blockIdx.y * N + oc, (float)(red[0][c] + red[1][c] + red[2][c] + red[3][c]));
}

// ------------------------------------------------------------------------------------------------
// GroupNorm (+SiLU) backward.   xhat = (x - mean_g) rstd_g,  z = gamma xhat + beta,  y = [silu](z)
//   dz = dy [silu'(z)],  A_c = sum dz,  B_c = sum dz xhat   per (instance, channel)   -> kernel 1
//   dx = rstd_g (gamma_c dz - (s1 + xhat s2) / n),  s1 = sum_{c in g} gamma_c A_c,  s2 likewise B,
//   n = rows_per_inst * C / 32                                                        -> kernel 2
//   dgamma_c = sum_inst B_c,  dbeta_c = sum_inst A_c                                  (host, tiny)
// ------------------------------------------------------------------------------------------------
// Round 3: per-channel constants live in LDS tables built once per workgroup (the first version re-derived group
// index, statistics and affine per ELEMENT and ran its reduction on 168 workgroups with fp64 chains and 8 fp64 atomics
// per thread: 0.6 TB/s); the reduction accumulates in fp32 per thread (<= 64 rows), combines the row lanes of a
// workgroup in LDS and issues one fp64 atomic per (channel, workgroup).
__global__ __launch_bounds__(256) void gn_bwd_reduce_kernel(
    const float* __restrict__ x, int64_t ldx, const float* __restrict__ dy, int64_t lddy, int C,
    int64_t rows_per_inst, int rows_per_chunk, int txw, int kpass, const float* __restrict__ stats,
    const float* __restrict__ gamma, const float* __restrict__ beta, int silu, float* __restrict__ partial) {
  // Round 4: no atomics.  The first form folded every workgroup's (A_c, B_c) into AB with 8 fp64 device-scope
  // atomics per column thread — ~1 M atomics on ~2 K cache lines per launch, which set the kernel's time (115 us
  // for a 110 MB pass: 1 TB/s).  A workgroup now WRITES its chunk's sums, partial[inst][chunk][2][C] fp32, and
  // gn_bwd_finalize_kernel folds the chunks in fp64; the row loop keeps four rows (8 x 16 B) per thread in flight.
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* red = (float*)smem_raw;   // [rpp][txw * 8]
  const int t = threadIdx.x;
  const int cg = C / 32;
  const int inst = blockIdx.y, chunk = blockIdx.x;
  const int64_t r0 = (int64_t)chunk * rows_per_chunk;
  int64_t r1 = r0 + rows_per_chunk;
  if (r1 > rows_per_inst) r1 = rows_per_inst;
  const int64_t base = (int64_t)inst * rows_per_inst;
  const int rpp = blockDim.x / txw;
  const int tx = t % txw, ty = t / txw;
  const bool on = ty < rpp;
  float* const pout = partial + ((int64_t)inst * gridDim.x + chunk) * 2 * C;
  for (int kp = 0; kp < kpass; ++kp) {
    const int c = (tx + kp * txw) * 4;
    float mu[4], rs[4], ga[4], be[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int g = (c + e) / cg;
      mu[e] = stats[inst * 64 + 2 * g];
      rs[e] = stats[inst * 64 + 2 * g + 1];
      ga[e] = gamma[c + e];
      be[e] = beta[c + e];
    }
    float a[4] = {0.f, 0.f, 0.f, 0.f}, b[4] = {0.f, 0.f, 0.f, 0.f};
    auto acc = [&](const f32x4 xv, const f32x4 dv) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float xh = (xv[e] - mu[e]) * rs[e];
        float dz = dv[e];
        if (silu) dz *= dsilu_f(fmaf(ga[e], xh, be[e]));
        a[e] += dz;
        b[e] = fmaf(dz, xh, b[e]);
      }
    };
    if (on) {
      const float* xp = x + base * ldx + c;
      const float* dp = dy + base * lddy + c;
      int64_t r = r0 + ty;
      for (; r + 3 * rpp < r1; r += 4 * rpp) {
        f32x4 xv[4], dv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          xv[k] = *(const f32x4*)(xp + (r + k * rpp) * ldx);
          dv[k] = *(const f32x4*)(dp + (r + k * rpp) * lddy);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) acc(xv[k], dv[k]);
      }
      for (; r < r1; r += rpp) acc(*(const f32x4*)(xp + r * ldx), *(const f32x4*)(dp + r * lddy));
      float* dst = red + ((int64_t)ty * txw + tx) * 8;
      *(f32x4*)dst = f32x4{a[0], a[1], a[2], a[3]};
      *(f32x4*)(dst + 4) = f32x4{b[0], b[1], b[2], b[3]};
    }
    __syncthreads();
    if (on && ty == 0) {
      f32x4 sa = {0.f, 0.f, 0.f, 0.f}, sb = {0.f, 0.f, 0.f, 0.f};
      for (int y = 0; y < rpp; ++y) {
        const float* src = red + ((int64_t)y * txw + tx) * 8;
        sa += *(const f32x4*)src;
        sb += *(const f32x4*)(src + 4);
      }
      *(f32x4*)(pout + c) = sa;
      *(f32x4*)(pout + C + c) = sb;
    }
    __syncthreads();
  }
}

// AB[inst][c] = (sum over the chunks of A_c, of B_c) in fp64.  grid = (ceil(C / 64), ninst), block 256: thread
// (channel t & 63, part t >> 6) folds every 4th chunk, the four parts meet in LDS.
__global__ __launch_bounds__(256) void gn_bwd_finalize_kernel(const float* __restrict__ partial, int nchunks, int C,
                                                             double* __restrict__ AB) {
  __shared__ double red[2][4][64];
  const int inst = blockIdx.y, cl = threadIdx.x & 63, part = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  double a = 0.0, b = 0.0;
  if (c < C) {
    const float* src = partial + (int64_t)inst * nchunks * 2 * C + c;
    for (int k = part; k < nchunks; k += 4) {
      a += (double)src[(int64_t)k * 2 * C];
      b += (double)src[(int64_t)k * 2 * C + C];
    }
  }
  red[0][part][cl] = a;
  red[1][part][cl] = b;
  __syncthreads();
  if (part == 0 && c < C) {
    double* dst = AB + ((int64_t)inst * C + c) * 2;
    dst[0] = red[0][0][cl] + red[0][1][cl] + red[0][2][cl] + red[0][3][cl];
    dst[1] = red[1][0][cl] + red[1][1][cl] + red[1][2][cl] + red[1][3][cl];
  }
}

__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(
    const float* __restrict__ x, int64_t ldx, const float* __restrict__ dy, int64_t lddy, int C,
    int64_t rows_per_inst, int rows_per_chunk, const float* __restrict__ stats,
    const float* __restrict__ gamma, const float* __restrict__ beta, int silu,
    const double* __restrict__ AB, float* __restrict__ dx, int64_t lddx, const float* __restrict__ dx_add, int64_t ldadd) {
  // per-channel tables: mean, rstd, gamma, beta, s1 (group), s2 (group)
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* tmu = (float*)smem_raw;
  float* trs = tmu + C;
  float* tga = trs + C;
  float* tbe = tga + C;
  float* ts1 = tbe + C;
  float* ts2 = ts1 + C;
  __shared__ float s1[32], s2[32];
  const int inst = blockIdx.y, t = threadIdx.x;
  const int cg = C / 32;
  if (t < 32) {
    double a = 0.0, b = 0.0;
    for (int c = t * cg; c < (t + 1) * cg; ++c) {
      a += (double)gamma[c] * AB[((int64_t)inst * C + c) * 2];
      b += (double)gamma[c] * AB[((int64_t)inst * C + c) * 2 + 1];
    }
    const double n = (double)rows_per_inst * (double)cg;
    s1[t] = (float)(a / n);
    s2[t] = (float)(b / n);
  }
  __syncthreads();
  for (int c = t; c < C; c += 256) {
    const int g = c / cg;
    tmu[c] = stats[inst * 64 + 2 * g];
    trs[c] = stats[inst * 64 + 2 * g + 1];
    tga[c] = gamma[c];
    tbe[c] = beta[c];
    ts1[c] = s1[g];
    ts2[c] = s2[g];
  }
  __syncthreads();
  const int cv4 = C >> 2;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_chunk;
  int64_t nrows = rows_per_inst - r0;
  if (nrows > rows_per_chunk) nrows = rows_per_chunk;
  const int64_t base = (int64_t)inst * rows_per_inst + r0;
  const int64_t total = nrows * cv4;
  for (int64_t idx = t; idx < total; idx += 256) {
    const int64_t r = idx / cv4;
    const int c = (int)(idx - r * cv4) * 4;
    const f32x4 xv = *(const f32x4*)(x + (base + r) * ldx + c);
    const f32x4 dv = *(const f32x4*)(dy + (base + r) * lddy + c);
    const f32x4 mu = *(const f32x4*)(tmu + c), rs = *(const f32x4*)(trs + c), ga = *(const f32x4*)(tga + c);
    const f32x4 be = *(const f32x4*)(tbe + c), g1 = *(const f32x4*)(ts1 + c), g2 = *(const f32x4*)(ts2 + c);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float xh = (xv[e] - mu[e]) * rs[e];
      float dz = dv[e];
      if (silu) dz *= dsilu_f(fmaf(ga[e], xh, be[e]));
      o[e] = rs[e] * (ga[e] * dz - g1[e] - xh * g2[e]);
    }
    if (dx_add) o += *(const f32x4*)(dx_add + (base + r) * ldadd + c);      // (ABI v7: the other gradient path's addend)
    *(f32x4*)(dx + (base + r) * lddx + c) = o;
  }
}

// ------------------------------------------------------------------------------------------------
// LayerNorm backward: one wave per row (C <= 1280: <= 5 float4 per lane), rows strided over the grid.
//   dxhat = dy gamma,  dx = rstd (dxhat - mean(dxhat) - xhat mean(dxhat xhat)),
//   dgamma += dy xhat, dbeta += dy  (per-lane column accumulators, one fp32 atomic per column and wave)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ x, int64_t ldx,
                                                     const float* __restrict__ dy, int64_t lddy,
                                                     int64_t M, int C, const float* __restrict__ gamma,
                                                     float eps, float* __restrict__ dx, int64_t lddx,
                                                     float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                     const float* __restrict__ dx_add, int64_t ldadd) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nwave = (int64_t)gridDim.x * 4;
  constexpr int KM = 5;
  f32x4 gv[KM], ag[KM], ab[KM];
#pragma unroll
  for (int k = 0; k < KM; ++k) {
    const int c = 256 * k + 4 * lane;
    gv[k] = c < C ? *(const f32x4*)(gamma + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    ag[k] = ab[k] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const float invC = 1.0f / (float)C;
  for (int64_t m = wave; m < M; m += nwave) {
    f32x4 xv[KM], dv[KM];
    float s = 0.f, q = 0.f;
#pragma unroll
    for (int k = 0; k < KM; ++k) {
      const int c = 256 * k + 4 * lane;
      const bool ok = c < C;
      xv[k] = ok ? *(const f32x4*)(x + m * ldx + c) : f32x4{0.f, 0.f, 0.f, 0.f};
      dv[k] = ok ? *(const f32x4*)(dy + m * lddy + c) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int e = 0; e < 4; ++e) s += xv[k][e];
    }
    const float mean = wave_sum(s) * invC;
#pragma unroll
    for (int k = 0; k < KM; ++k) {
      const int c = 256 * k + 4 * lane;
      if (c < C) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float d = xv[k][e] - mean;
          q = fmaf(d, d, q);
        }
      }
    }
    const float rstd = rsqrtf(wave_sum(q) * invC + eps);
    float m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int k = 0; k < KM; ++k) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float xh = (xv[k][e] - mean) * rstd;
        const float dh = dv[k][e] * gv[k][e];
        xv[k][e] = xh;
        m1 += dh;
        m2 = fmaf(dh, xh, m2);
        ag[k][e] = fmaf(dv[k][e], xh, ag[k][e]);
        ab[k][e] += dv[k][e];
      }
    }
    m1 = wave_sum(m1) * invC;
    m2 = wave_sum(m2) * invC;
#pragma unroll
    for (int k = 0; k < KM; ++k) {
      const int c = 256 * k + 4 * lane;
      if (c < C) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = rstd * (dv[k][e] * gv[k][e] - m1 - xv[k][e] * m2);
        if (dx_add) o += *(const f32x4*)(dx_add + m * ldadd + c);      // (ABI v7: the residual path's gradient)
        *(f32x4*)(dx + m * lddx + c) = o;
      }
    }
  }
  // the four waves of a workgroup combine their column sums in LDS: one fp32 atomic per column and WORKGROUP
  // (at one per wave, 8192 waves x 2 C atomics on 2 C addresses made this kernel 10x slower than its traffic)
  __shared__ float red[4][2 * 1280];
  const int w = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < KM; ++k) {
    const int c = 256 * k + 4 * lane;
    if (c < C) {
      *(f32x4*)(&red[w][c]) = ag[k];
      *(f32x4*)(&red[w][1280 + c]) = ab[k];
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    atomicAdd(dgamma + c, red[0][c] + red[1][c] + red[2][c] + red[3][c]);
    atomicAdd(dbeta + c, red[0][1280 + c] + red[1][1280 + c] + red[2][1280 + c] + red[3][1280 + c]);
  }
}

// ------------------------------------------------------------------------------------------------
// GEGLU on the fp32 projection h = [value | gate] ([M, 2H], attention.py:87-97):
//   out = value * gelu(gate);   dvalue = dout gelu(gate),  dgate = dout value gelu'(gate)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void geglu_fwd_kernel(const float* __restrict__ h, int64_t ldh,
                                                        float* __restrict__ out, int64_t ldo, int64_t M,
                                                        int H) {
  const int hv = H >> 2;
  const int64_t total = M * hv;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int64_t m = idx / hv;
    const int c = (int)(idx - m * hv) * 4;
    const f32x4 a = *(const f32x4*)(h + m * ldh + c), g = *(const f32x4*)(h + m * ldh + H + c);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = a[e] * gelu_f(g[e]);
    *(f32x4*)(out + m * ldo + c) = o;
  }
}

// the same with the result rounded to fp16: the operand of the FeedForward's second Linear, written once
__global__ __launch_bounds__(256) void geglu_fwd_f16_kernel(const float* __restrict__ h, int64_t ldh,
                                                            f16* __restrict__ out, int64_t ldo, int64_t M, int H,
                                                            int bf16) {
  const int hv = H >> 3;
  const int64_t total = M * hv;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int64_t m = idx / hv;
    const int c = (int)(idx - m * hv) * 8;
    const float* row = h + m * ldh + c;
    const f32x4 a0 = *(const f32x4*)row, a1 = *(const f32x4*)(row + 4);
    const f32x4 g0 = *(const f32x4*)(row + H), g1 = *(const f32x4*)(row + H + 4);
    f16x8 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      o[e] = gcd_cvt16(a0[e] * gelu_f(g0[e]), bf16 != 0);
      o[e + 4] = gcd_cvt16(a1[e] * gelu_f(g1[e]), bf16 != 0);
    }
    *(f16x8*)(out + m * ldo + c) = o;
  }
}

__global__ __launch_bounds__(256) void geglu_bwd_kernel(const float* __restrict__ h, int64_t ldh,
                                                        const float* __restrict__ dout, int64_t lddo,
                                                        float* __restrict__ dh, int64_t lddh, int64_t M,
                                                        int H) {
  const int hv = H >> 2;
  const int64_t total = M * hv;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int64_t m = idx / hv;
    const int c = (int)(idx - m * hv) * 4;
    const f32x4 a = *(const f32x4*)(h + m * ldh + c), g = *(const f32x4*)(h + m * ldh + H + c);
    const f32x4 d = *(const f32x4*)(dout + m * lddo + c);
    f32x4 da, dg;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      da[e] = d[e] * gelu_f(g[e]);
      dg[e] = d[e] * a[e] * dgelu_f(g[e]);
    }
    *(f32x4*)(dh + m * lddh + c) = da;
    *(f32x4*)(dh + m * lddh + H + c) = dg;
  }
}

// ------------------------------------------------------------------------------------------------
// softmax backward over rows:  dS = P (dP - sum_j P_j dP_j) * scale   (P fp16, dP fp32, dS fp16)
// one block per row.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void softmax_bwd_rows_kernel(const f16* __restrict__ P, int64_t ldp,
                                                               const float* __restrict__ dP, int64_t lddp,
                                                               f16* __restrict__ dS, int64_t ldds, int S,
                                                               float scale) {
  __shared__ float red[4];
  const int64_t r = blockIdx.x;
  const int t = threadIdx.x;
  float acc = 0.f;
  for (int j = t; j < S; j += 256) acc = fmaf((float)P[r * ldp + j], dP[r * lddp + j], acc);
  acc = wave_sum(acc);
  if ((t & 63) == 0) red[t >> 6] = acc;
  __syncthreads();
  const float D = red[0] + red[1] + red[2] + red[3];
  for (int j = t; j < S; j += 256)
    dS[r * ldds + j] = (f16)((float)P[r * ldp + j] * (dP[r * lddp + j] - D) * scale);
}

// ------------------------------------------------------------------------------------------------
// Temporal self-attention backward (T <= 16 tokens per problem, d = 64): one wave per
// (clip, pixel, head).  qkv fp16 [clips*T*HW, 3C] as the forward kernel reads it, dO fp32 [M, C];
// dqkv fp32 [M, 3C].  Everything of a problem lives in this wave's LDS slice.
// ------------------------------------------------------------------------------------------------
constexpr int TB_T = 16;
__global__ __launch_bounds__(128) void attn_temporal_bwd_kernel(const f16* __restrict__ qkv, int64_t ld,
                                                                const float* __restrict__ dO, int64_t lddo,
                                                                float* __restrict__ dqkv, int64_t lddq,
                                                                int64_t nprob, int T, int HW, int heads,
                                                                float scale) {
  __shared__ float sm[2][4 * TB_T * 64 + 3 * TB_T * TB_T];   // 2 waves x 19 KB
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float* Q = sm[w];
  float* K = Q + TB_T * 64;
  float* V = K + TB_T * 64;
  float* G = V + TB_T * 64;           // dO
  float* Pm = G + TB_T * 64;          // [T][T] probabilities
  float* dPm = Pm + TB_T * TB_T;      // [T][T]
  float* dSm = dPm + TB_T * TB_T;     // [T][T]
  const int C = heads * 64;
  for (int64_t prob = (int64_t)blockIdx.x * 2 + w; prob < nprob; prob += (int64_t)gridDim.x * 2) {
    const int head = (int)(prob % heads);
    const int64_t ph = prob / heads;
    const int hw = (int)(ph % HW);
    const int64_t clip = ph / HW;
    const int64_t row0 = clip * T * HW + hw;
    for (int t = 0; t < T; ++t) {
      const int64_t row = row0 + (int64_t)t * HW;
      Q[t * 64 + lane] = (float)qkv[row * ld + head * 64 + lane];
      K[t * 64 + lane] = (float)qkv[row * ld + C + head * 64 + lane];
      V[t * 64 + lane] = (float)qkv[row * ld + 2 * C + head * 64 + lane];
      G[t * 64 + lane] = dO[row * lddo + head * 64 + lane];
    }
    __builtin_amdgcn_wave_barrier();
    for (int p = lane; p < T * T; p += 64) {
      const int t = p / T, s = p - t * T;
      float a = 0.f, b = 0.f;
      for (int d = 0; d < 64; ++d) {
        const int dd = (d + lane) & 63;   // rotate: spreads the LDS banks
        a = fmaf(Q[t * 64 + dd], K[s * 64 + dd], a);
        b = fmaf(G[t * 64 + dd], V[s * 64 + dd], b);
      }
      Pm[t * TB_T + s] = a * scale;
      dPm[t * TB_T + s] = b;
    }
    __builtin_amdgcn_wave_barrier();
    if (lane < T) {
      float mx = -INFINITY;
      for (int s = 0; s < T; ++s) mx = fmaxf(mx, Pm[lane * TB_T + s]);
      float sum = 0.f;
      for (int s = 0; s < T; ++s) {
        const float e = __expf(Pm[lane * TB_T + s] - mx);
        Pm[lane * TB_T + s] = e;
        sum += e;
      }
      const float inv = 1.0f / sum;
      float D = 0.f;
      for (int s = 0; s < T; ++s) {
        const float pv = Pm[lane * TB_T + s] * inv;
        Pm[lane * TB_T + s] = pv;
        D = fmaf(pv, dPm[lane * TB_T + s], D);
      }
      for (int s = 0; s < T; ++s)
        dSm[lane * TB_T + s] = Pm[lane * TB_T + s] * (dPm[lane * TB_T + s] - D) * scale;
    }
    __builtin_amdgcn_wave_barrier();
    for (int t = 0; t < T; ++t) {
      float dq = 0.f, dk = 0.f, dv = 0.f;
      for (int s = 0; s < T; ++s) {
        dq = fmaf(dSm[t * TB_T + s], K[s * 64 + lane], dq);
        dk = fmaf(dSm[s * TB_T + t], Q[s * 64 + lane], dk);
        dv = fmaf(Pm[s * TB_T + t], G[s * 64 + lane], dv);
      }
      const int64_t row = row0 + (int64_t)t * HW;
      dqkv[row * lddq + head * 64 + lane] = dq;
      dqkv[row * lddq + C + head * 64 + lane] = dk;
      dqkv[row * lddq + 2 * C + head * 64 + lane] = dv;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

__global__ __launch_bounds__(256) void cast_scale_kernel(const float* __restrict__ x, int64_t ldx,
                                                         f16* __restrict__ y, int64_t ldy, int64_t M, int C,
                                                         float scale) {
  const int cv = C >> 2;
  const int64_t total = M * cv;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int64_t m = idx / cv;
    const int c = (int)(idx - m * cv) * 4;
    const f32x4 v = *(const f32x4*)(x + m * ldx + c);
    f16x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (f16)(v[e] * scale);
    *(f16x4*)(y + m * ldy + c) = o;
  }
}

// fp32 / fp16 -> bfloat16, round to nearest even (NaN kept quiet)
__device__ __forceinline__ unsigned short bf16_rne(float f) {
  unsigned u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
template <typename T>
__global__ __launch_bounds__(256) void cast_bf16_kernel(const T* __restrict__ x, int64_t ldx,
                                                        unsigned short* __restrict__ y, int64_t ldy, int64_t M,
                                                        int C) {
  const int cv = C >> 2;
  const int64_t total = M * cv;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int64_t m = idx / cv;
    const int c = (int)(idx - m * cv) * 4;
    typedef unsigned short us4 __attribute__((ext_vector_type(4)));
    us4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = bf16_rne((float)x[m * ldx + c + e]);
    *(us4*)(y + m * ldy + c) = o;
  }
}

// fp32 -> fp16 / bf16 AND the column sums of every block of `rows_per_block` rows in the same pass: the incoming
// gradient of a Linear / convolution is rounded for the dgrad / wgrad GEMMs and summed for the bias (one block) or
// the per-frame epilogue vector (one block per frame) — one read of dY instead of two.
// grid (ceil(C/64), nblk * chunks), block 256 = 8 column groups (8 channels) x 32 row lanes.
template <bool BF>
__global__ __launch_bounds__(256) void cast_colsum_kernel(const float* __restrict__ x, int64_t ldx,
                                                          unsigned short* __restrict__ y, int64_t ldy,
                                                          int64_t rows_per_block, int chunks, int rows_per_chunk, int C,
                                                          float* __restrict__ sums, float* __restrict__ total) {
  __shared__ float red[32][65];
  const int t = threadIdx.x, cg = t & 7, rl = t >> 3;
  const int c = blockIdx.x * 64 + cg * 8;
  const int blk = blockIdx.y / chunks, ch = blockIdx.y - blk * chunks;
  const int64_t r_begin = (int64_t)ch * rows_per_chunk;
  int64_t r_end = r_begin + rows_per_chunk;
  if (r_end > rows_per_block) r_end = rows_per_block;
  const int64_t base = (int64_t)blk * rows_per_block;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (c < C) {
    for (int64_t r = r_begin + rl; r < r_end; r += 32) {
      const float* src = x + (base + r) * ldx + c;
      const f32x4 a = *(const f32x4*)src, b = *(const f32x4*)(src + 4);
      typedef unsigned short us8 __attribute__((ext_vector_type(8)));
      us8 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc[e] += a[e];
        acc[e + 4] += b[e];
        if (BF) {
          o[e] = bf16_rne(a[e]);
          o[e + 4] = bf16_rne(b[e]);
        } else {
          o[e] = __builtin_bit_cast(unsigned short, (f16)a[e]);
          o[e + 4] = __builtin_bit_cast(unsigned short, (f16)b[e]);
        }
      }
      *(us8*)(y + (base + r) * ldy + c) = o;
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[rl][cg * 8 + e] = acc[e];
  __syncthreads();
  if (t < 64 && blockIdx.x * 64 + t < C) {
    float s0 = 0.f;
#pragma unroll
    for (int r = 0; r < 32; ++r) s0 += red[r][t];
    atomicAdd(sums + (int64_t)blk * C + blockIdx.x * 64 + t, s0);
    if (total) atomicAdd(total + blockIdx.x * 64 + t, s0);      // ABI v7: the sum over ALL row blocks too (the bias gradient)
  }
}

// torch.optim.Adam (no amsgrad; L2 weight decay added to the gradient), one fused pass
// (diffusion.py:412-431 instantiates the optimizer the config names — Adam, lr 2e-5 for GCD).
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                   float lr, float b1, float b2, float eps, float wd,
                                                   float bc1, float bc2_sqrt, float gscale) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    float gi = g[i] * gscale;
    if (wd != 0.f) gi = fmaf(wd, p[i], gi);
    const float mi = fmaf(b1, m[i], (1.0f - b1) * gi);
    const float vi = fmaf(b2, v[i], (1.0f - b2) * gi * gi);
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] -= (lr / bc1) * (mi / denom);
  }
}

// The same step over up to ADAM_MT tensors per launch (the UNet has ~1300 parameter tensors, most of them a few
// KB: one launch each costs more than their arithmetic).  Block b works on chunk b - first[t] of tensor t.
constexpr int ADAM_MT = 48;
constexpr int ADAM_CHUNK = 1 << 14;   // 16 float4 per thread: 4 x 64 KB in flight per block
struct AdamMulti {
  float* p[ADAM_MT];
  const float* g[ADAM_MT];
  float* m[ADAM_MT];
  float* v[ADAM_MT];
  int64_t n[ADAM_MT];
  int first[ADAM_MT + 1];   // prefix sum of the tensors' chunk counts
  int count;
};

__global__ __launch_bounds__(256) void adam_multi_kernel(const AdamMulti a, float lr, float b1, float b2, float eps,
                                                         float wd, float bc1, float bc2_sqrt, float gscale) {
  int t = 0;
  while (t + 1 < a.count && (int)blockIdx.x >= a.first[t + 1]) ++t;
  const int64_t i0 = (int64_t)((int)blockIdx.x - a.first[t]) * ADAM_CHUNK;
  const int64_t i1 = i0 + ADAM_CHUNK < a.n[t] ? i0 + ADAM_CHUNK : a.n[t];
  float* __restrict__ p = a.p[t];
  const float* __restrict__ g = a.g[t];
  float* __restrict__ m = a.m[t];
  float* __restrict__ v = a.v[t];
  const float step = lr / bc1, omb1 = 1.0f - b1, omb2 = 1.0f - b2;
  auto upd = [&](float& pi, float gi, float& mi, float& vi) {
    gi *= gscale;
    if (wd != 0.f) gi = fmaf(wd, pi, gi);
    mi = fmaf(b1, mi, omb1 * gi);
    vi = fmaf(b2, vi, omb2 * gi * gi);
    pi -= step * (mi / (sqrtf(vi) / bc2_sqrt + eps));
  };
  // 16-byte vectors where all four bases are 16-byte aligned (torch allocations are); chunks start at multiples
  // of 16 K elements, so only the tensor's last few elements take the scalar path
  const bool vec = (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0;
  int64_t i = i0;
  if (vec) {
    const int64_t nv = (i1 - i0) >> 2;
    for (int64_t k = threadIdx.x; k < nv; k += 256) {
      const int64_t j = i0 + 4 * k;
      f32x4 pv = *(f32x4*)(p + j), mv = *(f32x4*)(m + j), vv = *(f32x4*)(v + j);
      const f32x4 gv = *(const f32x4*)(g + j);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float pe = pv[e], me = mv[e], ve = vv[e];
        upd(pe, gv[e], me, ve);
        pv[e] = pe;
        mv[e] = me;
        vv[e] = ve;
      }
      *(f32x4*)(p + j) = pv;
      *(f32x4*)(m + j) = mv;
      *(f32x4*)(v + j) = vv;
    }
    i = i0 + 4 * nv;
  }
  for (i += threadIdx.x; i < i1; i += 256) upd(p[i], g[i], m[i], v[i]);
}

int grid_for(int64_t work_items) {
  int64_t b = (work_items + 255) / 256;
  if (b > 256 * 32) b = 256 * 32;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

// =================================================================================================
extern "C" int gcd_im2col3x3_f16(const void* x16, int64_t ldx, void* col16, int frames, int Cin, int Hi,
                                 int Wi, int Ho, int Wo, int stride, int upsample, int asym_pad,
                                 void* stream) {
  GCD_CHECK_ARG(x16 && col16, "gcd_im2col3x3_f16: null pointer");
  GCD_CHECK_ARG(frames > 0 && Cin > 0 && Cin % 8 == 0 && ldx % 8 == 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0,
                "gcd_im2col3x3_f16: bad geometry (Cin=%d must be a multiple of 8)", Cin);
  GCD_CHECK_ARG((stride == 1 || stride == 2) && (!upsample || stride == 1), "gcd_im2col3x3_f16: stride %d", stride);
  const ConvGeo g{Cin, Hi, Wi, Ho, Wo, stride, upsample ? 1 : 0, asym_pad ? 1 : 0};
  const int64_t M = (int64_t)frames * Ho * Wo;
  hipLaunchKernelGGL(im2col3x3_kernel, dim3(grid_for(M * 9 * (Cin / 8))), dim3(256), 0, (hipStream_t)stream,
                     (const f16*)x16, ldx, (f16*)col16, g, M);
  GCD_CHECK_LAUNCH();
  return 0;
}

extern "C" int gcd_col2im3x3_f32(const float* dcol, float* dx, int64_t lddx, int frames, int Cin, int Hi,
                                 int Wi, int Ho, int Wo, int stride, int upsample, int asym_pad,
                                 void* stream) {
  GCD_CHECK_ARG(dcol && dx, "gcd_col2im3x3_f32: null pointer");
  GCD_CHECK_ARG(frames > 0 && Cin > 0 && Cin % 4 == 0 && lddx % 4 == 0, "gcd_col2im3x3_f32: bad geometry");
  GCD_CHECK_ARG((stride == 1 || stride == 2) && (!upsample || stride == 1), "gcd_col2im3x3_f32: stride %d", stride);
  const ConvGeo g{Cin, Hi, Wi, Ho, Wo, stride, upsample ? 1 : 0, asym_pad ? 1 : 0};
  const int64_t Min = (int64_t)frames * Hi * Wi;
  hipLaunchKernelGGL(col2im3x3_kernel, dim3(grid_for(Min * (Cin / 4))), dim3(256), 0, (hipStream_t)stream, dcol,
                     dx, lddx, g, Min);
  GCD_CHECK_LAUNCH();
  return 0;
}

extern "C" int gcd_im2col_t3_f16(const void* x16, int64_t ldx, void* col16, int64_t M, int C, int T, int HW,
                                 void* stream) {
  GCD_CHECK_ARG(x16 && col16 && M > 0 && C > 0 && C % 8 == 0 && ldx % 8 == 0 && T > 0 && HW > 0 &&
                    M % ((int64_t)T * HW) == 0,
                "gcd_im2col_t3_f16: bad geometry (M=%lld C=%d T=%d HW=%d)", (long long)M, C, T, HW);
  hipLaunchKernelGGL(im2col_t3_kernel, dim3(grid_for(M * 3 * (C / 8))), dim3(256), 0, (hipStream_t)stream,
                     (const f16*)x16, ldx, (f16*)col16, C, T, HW, M);
  GCD_CHECK_LAUNCH();
  return 0;
}

extern "C" int gcd_col2im_t3_f32(const float* dcol, float* dx, int64_t lddx, int64_t M, int C, int T, int HW,
                                 void* stream) {
  GCD_CHECK_ARG(dcol && dx && M > 0 && C > 0 && C % 4 == 0 && lddx % 4 == 0 && T > 0 && HW > 0 &&
                    M % ((int64_t)T * HW) == 0,
                "gcd_col2im_t3_f32: bad geometry");
  hipLaunchKernelGGL(col2im_t3_kernel, dim3(grid_for(M * (C / 4))), dim3(256), 0, (hipStream_t)stream, dcol, dx,
                     lddx, C, T, HW, M);
  GCD_CHECK_LAUNCH();
  return 0;
}

extern "C" int gcd_rowblock_sum_f32(const float* x, int64_t ldx, int64_t M, int N, int64_t rows_per_block,
                                    float* out_zeroed, void* stream) {
  GCD_CHECK_ARG(x && out_zeroed && M > 0 && N > 0 && rows_per_block > 0 && M % rows_per_block == 0,
                "gcd_rowblock_sum_f32: M=%lld rows_per_block=%lld", (long long)M, (long long)rows_per_block);
  GCD_CHECK_ARG(N % 4 == 0 && ldx % 4 == 0 && ((uintptr_t)x & 15) == 0,
                "gcd_rowblock_sum_f32: N=%d / ldx=%lld must be multiples of 4, x 16-byte aligned", N, (long long)ldx);
  const int64_t nblk = M / rows_per_block;
  GCD_CHECK_ARG(nblk <= 65535, "gcd_rowblock_sum_f32: too many blocks");
  int64_t nsplit = rows_per_block / 256;
  if (nsplit < 1) nsplit = 1;
  if (nsplit > 64) nsplit = 64;
  hipLaunchKernelGGL(rowblock_sum_kernel, dim3((N + 255) / 256, (unsigned)nblk, (unsigned)nsplit), dim3(256), 0,
                     (hipStream_t)stream, x, ldx, rows_per_block, N, out_zeroed);
  GCD_CHECK_LAUNCH();
  return 0;
}

// Row chunks per instance of the GroupNorm backward's reduction pass, and with them the scratch it needs:
// >= 64 rows per chunk, ~700 workgroups over the launch.
static int gn_bwd_chunks(int64_t rows_per_inst, int ninst, int* rows_per_chunk) {
  int64_t want = (704 + ninst - 1) / ninst;
  int64_t nchunks = (rows_per_inst + 63) / 64;
  if (nchunks > want) nchunks = want;
  if (nchunks < 1) nchunks = 1;
  const int rpc = (int)((rows_per_inst + nchunks - 1) / nchunks);
  nchunks = (rows_per_inst + rpc - 1) / rpc;
  if (rows_per_chunk) *rows_per_chunk = rpc;
  return (int)nchunks;
}

extern "C" int64_t gcd_groupnorm_bwd_scratch_floats(int C, int64_t M, int64_t rows_per_inst) {
  if (C <= 0 || M <= 0 || rows_per_inst <= 0 || M % rows_per_inst != 0) return 0;
  const int ninst = (int)(M / rows_per_inst);
  return (int64_t)ninst * gn_bwd_chunks(rows_per_inst, ninst, nullptr) * 2 * C;
}

extern "C" int gcd_groupnorm_bwd(const float* x, int64_t ldx, const float* dy, int64_t lddy, int C, int64_t M,
                                 int64_t rows_per_inst, const float* stats, const float* gamma,
                                 const float* beta, int silu, double* AB, float* scratch, int64_t scratch_floats,
                                 float* dx, int64_t lddx, const float* dx_add, int64_t ld_add, void* stream) {
  GCD_CHECK_ARG(!dx_add || ld_add % 4 == 0, "gcd_groupnorm_bwd: ld_add");
  GCD_CHECK_ARG(x && dy && stats && gamma && beta && AB && scratch && dx, "gcd_groupnorm_bwd: null pointer");
  GCD_CHECK_ARG(C > 0 && C % 32 == 0 && C % 4 == 0 && ldx % 4 == 0 && lddy % 4 == 0 && lddx % 4 == 0,
                "gcd_groupnorm_bwd: C=%d", C);
  GCD_CHECK_ARG(rows_per_inst > 0 && M > 0 && M % rows_per_inst == 0, "gcd_groupnorm_bwd: M / rows_per_inst");
  GCD_CHECK_ARG(scratch_floats >= gcd_groupnorm_bwd_scratch_floats(C, M, rows_per_inst) && ((uintptr_t)scratch & 15) == 0,
                "gcd_groupnorm_bwd: scratch of %lld floats, need %lld (gcd_groupnorm_bwd_scratch_floats), 16-byte aligned",
                (long long)scratch_floats, (long long)gcd_groupnorm_bwd_scratch_floats(C, M, rows_per_inst));
  const int ninst = (int)(M / rows_per_inst);
  hipStream_t s = (hipStream_t)stream;
  const int cv4 = C / 4;
  int kpass = 1;
  while (cv4 % kpass != 0 || cv4 / kpass > 256) ++kpass;
  const int txw = cv4 / kpass;
  const int rpp = 256 / txw;
  int rpc = 0;
  const int nchunks = gn_bwd_chunks(rows_per_inst, ninst, &rpc);
  GCD_CHECK_ARG(ninst <= 65535 && C * 24 <= 160 * 1024 - 512, "gcd_groupnorm_bwd: %d instances / C=%d too large", ninst, C);
  hipLaunchKernelGGL(gn_bwd_reduce_kernel, dim3((unsigned)nchunks, ninst), dim3(256), rpp * txw * 8 * sizeof(float), s, x,
                     ldx, dy, lddy, C, rows_per_inst, rpc, txw, kpass, stats, gamma, beta, silu, scratch);
  GCD_CHECK_LAUNCH();
  hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3((C + 63) / 64, ninst), dim3(256), 0, s, scratch, nchunks, C, AB);
  GCD_CHECK_LAUNCH();
  int achunks = (int)((rows_per_inst + 63) / 64);
  if (achunks > 1024) achunks = 1024;
  const int arpc = (int)((rows_per_inst + achunks - 1) / achunks);
  static GcdPerDeviceOnce attr_once;
  if (C * 24 > 48 * 1024) GCD_CHECK_HIP(attr_once.opt_in((const void*)gn_bwd_apply_kernel, 160 * 1024 - 512));
  hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3(achunks, ninst), dim3(256), C * 24, s, x, ldx, dy, lddy, C,
                     rows_per_inst, arpc, stats, gamma, beta, silu, AB, dx, lddx, dx_add, ld_add);
  GCD_CHECK_LAUNCH();
  return 0;
}

extern "C" int gcd_layernorm_bwd(const float* x, int64_t ldx, const float* dy, int64_t lddy, int64_t M, int C,
                                 const float* gamma, float eps, float* dx, int64_t lddx, float* dgamma_zeroed,
                                 float* dbeta_zeroed, const float* dx_add, int64_t ld_add, void* stream) {
  GCD_CHECK_ARG(!dx_add || ld_add % 4 == 0, "gcd_layernorm_bwd: ld_add");
  GCD_CHECK_ARG(x && dy && gamma && dx && dgamma_zeroed && dbeta_zeroed, "gcd_layernorm_bwd: null pointer");
  GCD_CHECK_ARG(C > 0 && C % 4 == 0 && C <= 1280 && ldx % 4 == 0 && lddy % 4 == 0 && lddx % 4 == 0 && M > 0,
                "gcd_layernorm_bwd: C=%d (multiple of 4, <= 1280)", C);
  int64_t blocks = (M + 3) / 4;
  if (blocks > 768) blocks = 768;      // 3 workgroups per CU; every workgroup ends with 2 C atomics
  hipLaunchKernelGGL(ln_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, ldx, dy, lddy,
                     M, C, gamma, eps, dx, lddx, dgamma_zeroed, dbeta_zeroed, dx_add, ld_add);
  GCD_CHECK_LAUNCH();
  return 0;
}

extern "C" int gcd_geglu_fwd_f32(const float* h, int64_t ldh, float* out, int64_t ldo, int64_t M, int H,
                                 void* stream) {
  GCD_CHECK_ARG(h && out && M > 0 && H > 0 && H % 4 == 0 && ldh % 4 == 0 && ldo % 4 == 0, "gcd_geglu_fwd_f32: bad args");
  hipLaunchKernelGGL(geglu_fwd_kernel, dim3(grid_for(M * (H / 4))), dim3(256), 0, (hipStream_t)stream, h, ldh,
                     out, ldo, M, H);
  GCD_CHECK_LAUNCH();
  return 0;
}

extern "C" int gcd_geglu_fwd_f16(const float* h, int64_t ldh, void* out16, int64_t ldo, int64_t M, int H,
                                 void* stream) {
  GCD_CHECK_ARG(h && out16 && M > 0 && H > 0 && H % 8 == 0 && ldh % 4 == 0 && ldo % 8 == 0 &&
                    (((uintptr_t)h | (uintptr_t)out16) & 15) == 0,
                "gcd_geglu_fwd_f16: bad args (H=%d must be a multiple of 8)", H);
  hipLaunchKernelGGL(geglu_fwd_f16_kernel, dim3(grid_for(M * (H / 8))), dim3(256), 0, (hipStream_t)stream, h, ldh,
                     (f16*)out16, ldo, M, H, 0);
  GCD_CHECK_LAUNCH();
  return 0;
}

extern "C" int gcd_geglu_fwd_bf16(const float* h, int64_t ldh, void* out16, int64_t ldo, int64_t M, int H,
                                  void* stream) {
  GCD_CHECK_ARG(h && out16 && M > 0 && H > 0 && H % 8 == 0 && ldh % 4 == 0 && ldo % 8 == 0 &&
                    (((uintptr_t)h | (uintptr_t)out16) & 15) == 0,
                "gcd_geglu_fwd_bf16: bad args (H=%d must be a multiple of 8)", H);
  hipLaunchKernelGGL(geglu_fwd_f16_kernel, dim3(grid_for(M * (H / 8))), dim3(256), 0, (hipStream_t)stream, h, ldh,
                     (f16*)out16, ldo, M, H, 1);
  GCD_CHECK_LAUNCH();
  return 0;
}

extern "C" int gcd_geglu_bwd_f32(const float* h, int64_t ldh, const float* dout, int64_t lddo, float* dh,
                                 int64_t lddh, int64_t M, int H, void* stream) {
  GCD_CHECK_ARG(h && dout && dh && M > 0 && H > 0 && H % 4 == 0 && ldh % 4 == 0 && lddo % 4 == 0 && lddh % 4 == 0,
                "gcd_geglu_bwd_f32: bad args");
  hipLaunchKernelGGL(geglu_bwd_kernel, dim3(grid_for(M * (H / 4))), dim3(256), 0, (hipStream_t)stream, h, ldh,
                     dout, lddo, dh, lddh, M, H);
  GCD_CHECK_LAUNCH();
  return 0;
}

extern "C" int gcd_softmax_bwd_rows(const void* P16, int64_t ldp, const float* dP, int64_t lddp, void* dS16,
                                    int64_t ldds, int64_t R, int S, float scale, void* stream) {
  GCD_CHECK_ARG(P16 && dP && dS16 && R > 0 && R < (1ll << 31) && S > 0, "gcd_softmax_bwd_rows: bad args");
  hipLaunchKernelGGL(softmax_bwd_rows_kernel, dim3((unsigned)R), dim3(256), 0, (hipStream_t)stream,
                     (const f16*)P16, ldp, dP, lddp, (f16*)dS16, ldds, S, scale);
  GCD_CHECK_LAUNCH();
  return 0;
}

extern "C" int gcd_attn_temporal_bwd(const void* qkv16, int64_t ld, const float* dO, int64_t lddo, float* dqkv,
                                     int64_t lddq, int clips, int T, int HW, int heads, void* stream) {
  GCD_CHECK_ARG(qkv16 && dO && dqkv, "gcd_attn_temporal_bwd: null pointer");
  GCD_CHECK_ARG(clips > 0 && T >= 1 && T <= TB_T && HW > 0 && heads > 0,
                "gcd_attn_temporal_bwd: T=%d (supported 1..16)", T);
  const int64_t nprob = (int64_t)clips * HW * heads;
  int64_t blocks = (nprob + 1) / 2;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(attn_temporal_bwd_kernel, dim3((unsigned)blocks), dim3(128), 0, (hipStream_t)stream,
                     (const f16*)qkv16, ld, dO, lddo, dqkv, lddq, nprob, T, HW, heads, 0.125f);
  GCD_CHECK_LAUNCH();
  return 0;
}

extern "C" int gcd_cast_scale_f32_f16(const float* x, int64_t ldx, void* y16, int64_t ldy, int64_t M, int C,
                                      float scale, void* stream) {
  GCD_CHECK_ARG(x && y16 && M > 0 && C > 0 && C % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0, "gcd_cast_scale_f32_f16: bad args");
  hipLaunchKernelGGL(cast_scale_kernel, dim3(grid_for(M * (C / 4))), dim3(256), 0, (hipStream_t)stream, x, ldx,
                     (f16*)y16, ldy, M, C, scale);
  GCD_CHECK_LAUNCH();
  return 0;
}

extern "C" int gcd_cast_f32_bf16(const float* x, int64_t ldx, void* y16, int64_t ldy, int64_t M, int C,
                                 void* stream) {
  GCD_CHECK_ARG(x && y16 && M > 0 && C > 0 && C % 4 == 0 && ldy % 4 == 0, "gcd_cast_f32_bf16: bad args");
  hipLaunchKernelGGL(cast_bf16_kernel<float>, dim3(grid_for(M * (C / 4))), dim3(256), 0, (hipStream_t)stream, x,
                     ldx, (unsigned short*)y16, ldy, M, C);
  GCD_CHECK_LAUNCH();
  return 0;
}

extern "C" int gcd_cast_colsum_f32(const float* x, int64_t ldx, void* y16, int64_t ldy, int64_t M, int C,
                                   int64_t rows_per_block, float* sums_zeroed, int to_bf16, float* total_zeroed,
                                   void* stream) {
  GCD_CHECK_ARG(x && y16 && sums_zeroed && M > 0 && C > 0 && C % 8 == 0 && ldx % 4 == 0 && ldy % 8 == 0 &&
                    (((uintptr_t)x | (uintptr_t)y16) & 15) == 0,
                "gcd_cast_colsum_f32: bad args (C=%d must be a multiple of 8, 16-byte aligned rows)", C);
  GCD_CHECK_ARG(rows_per_block > 0 && M % rows_per_block == 0, "gcd_cast_colsum_f32: M=%lld rows_per_block=%lld",
                (long long)M, (long long)rows_per_block);
  const int64_t nblk = M / rows_per_block;
  const int colb = (C + 63) / 64;
  // ~2048 workgroups over the launch, at least 32 rows per chunk
  int64_t chunks = (2048 + nblk * colb - 1) / (nblk * colb);
  if (chunks < 1) chunks = 1;
  if (chunks > (rows_per_block + 31) / 32) chunks = (rows_per_block + 31) / 32;
  const int rpc = (int)((rows_per_block + chunks - 1) / chunks);
  chunks = (rows_per_block + rpc - 1) / rpc;
  GCD_CHECK_ARG(nblk * chunks <= 65535, "gcd_cast_colsum_f32: too many row blocks (%lld)", (long long)(nblk * chunks));
  const dim3 grid(colb, (unsigned)(nblk * chunks));
  if (to_bf16)
    hipLaunchKernelGGL(cast_colsum_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, x, ldx, (unsigned short*)y16,
                       ldy, rows_per_block, (int)chunks, rpc, C, sums_zeroed, total_zeroed);
  else
    hipLaunchKernelGGL(cast_colsum_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, x, ldx, (unsigned short*)y16,
                       ldy, rows_per_block, (int)chunks, rpc, C, sums_zeroed, total_zeroed);
  GCD_CHECK_LAUNCH();
  return 0;
}

extern "C" int gcd_cast_f16_bf16(const void* x16, int64_t ldx, void* y16, int64_t ldy, int64_t M, int C,
                                 void* stream) {
  GCD_CHECK_ARG(x16 && y16 && M > 0 && C > 0 && C % 4 == 0 && ldy % 4 == 0, "gcd_cast_f16_bf16: bad args");
  hipLaunchKernelGGL(cast_bf16_kernel<f16>, dim3(grid_for(M * (C / 4))), dim3(256), 0, (hipStream_t)stream,
                     (const f16*)x16, ldx, (unsigned short*)y16, ldy, M, C);
  GCD_CHECK_LAUNCH();
  return 0;
}

extern "C" int gcd_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                             float beta2, float eps, float weight_decay, int step, float grad_scale,
                             void* stream) {
  GCD_CHECK_ARG(p && g && m && v && n > 0 && step >= 1, "gcd_adam_step: bad args");
  const float bc1 = 1.0f - powf(beta1, (float)step);
  const float bc2s = sqrtf(1.0f - powf(beta2, (float)step));
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr,
                     beta1, beta2, eps, weight_decay, bc1, bc2s, grad_scale);
  GCD_CHECK_LAUNCH();
  return 0;
}

extern "C" int gcd_adam_step_multi(int count, float* const* p, const float* const* g, float* const* m,
                                   float* const* v, const int64_t* n, float lr, float beta1, float beta2,
                                   float eps, float weight_decay, int step, float grad_scale, void* stream) {
  GCD_CHECK_ARG(count >= 0 && (count == 0 || (p && g && m && v && n)) && step >= 1, "gcd_adam_step_multi: bad args");
  const float bc1 = 1.0f - powf(beta1, (float)step);
  const float bc2s = sqrtf(1.0f - powf(beta2, (float)step));
  int i = 0;
  while (i < count) {
    AdamMulti a;
    a.count = 0;
    a.first[0] = 0;
    // a launch takes up to ADAM_MT tensors and at most ~64 K blocks
    while (i < count && a.count < ADAM_MT) {
      GCD_CHECK_ARG(p[i] && g[i] && m[i] && v[i] && n[i] >= 0, "gcd_adam_step_multi: tensor %d: null pointer", i);
      const int64_t chunks = (n[i] + ADAM_CHUNK - 1) / ADAM_CHUNK;
      if (a.count > 0 && a.first[a.count] + chunks > 65535) break;
      GCD_CHECK_ARG(chunks < (1ll << 30), "gcd_adam_step_multi: tensor %d too large", i);
      a.p[a.count] = p[i];
      a.g[a.count] = g[i];
      a.m[a.count] = m[i];
      a.v[a.count] = v[i];
      a.n[a.count] = n[i];
      a.first[a.count + 1] = a.first[a.count] + (int)chunks;
      ++a.count;
      ++i;
    }
    if (a.first[a.count] > 0) {
      hipLaunchKernelGGL(adam_multi_kernel, dim3((unsigned)a.first[a.count]), dim3(256), 0, (hipStream_t)stream, a,
                         lr, beta1, beta2, eps, weight_decay, bc1, bc2s, grad_scale);
      GCD_CHECK_LAUNCH();
    }
  }
  return 0;
}
