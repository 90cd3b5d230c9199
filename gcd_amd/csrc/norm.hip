// norm.hip — GroupNorm(32) and LayerNorm over the fp32 token-major residual stream (gfx950).
//
// These are the HBM-bound kernels of the VideoUNet forward (4.36 G GroupNorm
// elements + 5.09 G LayerNorm elements per step).  Design rules used here:
//   * every global access is a 16-byte vector (float4 in, 4 x fp16 = 8 B or 8 x fp16 = 16 B out);
//   * statistics are accumulated in fp64 per lane and reduced with wavefront shuffles + LDS, so
//     E[x^2] - mean^2 is safe for 1.3 M-element groups (time_stack GroupNorm over T*H*W);
//   * GroupNorm reads a *virtual concat* [x1 | x2] so the decoder's torch.cat (video_model.py:524)
//     never materialises in fp32; the fp16 operand it feeds to the next conv is written once,
//     normalised + SiLU'd, together with an optional raw fp16 copy for the 1x1 skip conv.
#include "common.h"

int gcd_tune_get(int knob);   // runtime.hip
#define GCD_STREAM_DEFAULT 3   // set from profiles/r04q_* / r04r_* (same-box sweeps)

// Walk orders of the two big streaming kernels (GroupNorm apply, LayerNorm): position v of the dispatch order -> block
// of rows.  Pure scheduling (see gcd_gemm_desc.sched in gcd_amd.h): a kernel that reads what the previous launch wrote
// finds in the 256 MB Infinity Cache what that launch wrote LAST.
//   0 front to back      1 back to front
//   2 / 3  the tensor as eight contiguous regions walked concurrently, position v -> region v % 8 — the order in which
//          a persistent GEMM's eight XCD shares are written — every region back to front (2) / front to back (3)
__device__ __forceinline__ int64_t gcd_walk(int64_t v, int64_t nb, int order) {
  if (order == 0) return v;
  if (order == 1) return nb - 1 - v;
  const int64_t q = nb >> 3;
  const int rem = (int)(nb & 7), r = (int)(v & 7);
  const int64_t pos = v >> 3;
  const int64_t start = r < rem ? r * (q + 1) : rem * (q + 1) + (r - rem) * q;
  const int64_t len = q + (r < rem ? 1 : 0);
  return order == 2 ? start + len - 1 - pos : start + pos;
}

// Streaming loads / stores of the two big streamers, optionally non-temporal (GCD_TUNE_STREAM bit 0 loads, bit 1
// stores; A/B knob): NT = 2 * stores + loads.
template <int NT>
__device__ __forceinline__ f32x4 gcd_ld16(const float* p) {
  if constexpr ((NT & 1) != 0) return __builtin_nontemporal_load((const f32x4*)p);
  else return *(const f32x4*)p;
}
template <int NT, typename V, typename T>
__device__ __forceinline__ void gcd_st(T* p, const V v) {
  if constexpr ((NT & 2) != 0) __builtin_nontemporal_store(v, (V*)p);
  else *(V*)p = v;
}
// Host: the NT mode of a launch that streams `bytes` (fp32 in + fp16 out).  GCD_TUNE_STREAM: bits 0-1 the mode
// (0 = the default, see below; 4 = plain loads and stores everywhere), bits 4-7 a size threshold in units of 100 MB
// below which a launch keeps plain accesses (its tensors fit the caches and the next launch reads them from there).
static int gcd_stream_nt(int64_t bytes) {
  int knob = gcd_tune_get(GCD_TUNE_STREAM);
  if (knob == 0) knob = GCD_STREAM_DEFAULT;
  if (knob & 4) return 0;
  const int64_t thresh = (int64_t)((knob >> 4) & 15) * 100 * 1000 * 1000;
  return bytes >= thresh ? (knob & 3) : 0;
}

// ------------------------------------------------------------------------------------------------
// GroupNorm statistics
// grid = (nchunks, ninst).  A thread keeps ONE float4 column for the whole chunk (block = txw
// columns x rpp rows, txw = C/4 or C/8, so no lane idles on a ragged column tail): its fp64
// sum / sum-of-squares stay in registers, the block then merges them per group with LDS atomics.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void gn_stats_partial_kernel(
    const float* __restrict__ x1, int64_t ld1, int C1, const float* __restrict__ x2, int64_t ld2,
    int C2, int64_t rows_per_inst, int rows_per_chunk, int txw, int kpass,
    double* __restrict__ partial) {
  __shared__ double acc[32][2];
  const int t = threadIdx.x;
  if (t < 64) acc[t >> 1][t & 1] = 0.0;
  __syncthreads();
  const int C = C1 + C2;
  const int cg = C / 32;
  const int inst = blockIdx.y, chunk = blockIdx.x;
  const int64_t r0 = (int64_t)chunk * rows_per_chunk;
  int64_t r1 = r0 + rows_per_chunk;
  if (r1 > rows_per_inst) r1 = rows_per_inst;
  const int64_t base = (int64_t)inst * rows_per_inst;
  const int tx = t % txw, ty = t / txw;
  const int rpp = blockDim.x / txw;
  for (int kp = 0; kp < kpass; ++kp) {
    const int c = (tx + kp * txw) * 4;
    const float* src;
    int64_t ld;
    if (c < C1) {
      src = x1 + c;
      ld = ld1;
    } else {
      src = x2 + (c - C1);
      ld = ld2;
    }
    double s[4] = {0, 0, 0, 0}, ss[4] = {0, 0, 0, 0};
    // four rows in flight per thread
    int64_t r = r0 + ty;
    for (; r + 3 * rpp < r1; r += 4 * rpp) {
      const f32x4 v = *(const f32x4*)(src + (base + r) * ld);
      const f32x4 w = *(const f32x4*)(src + (base + r + rpp) * ld);
      const f32x4 y = *(const f32x4*)(src + (base + r + 2 * rpp) * ld);
      const f32x4 z = *(const f32x4*)(src + (base + r + 3 * rpp) * ld);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const double d = (double)v[e], f = (double)w[e], g = (double)y[e], h = (double)z[e];
        s[e] += (d + f) + (g + h);
        ss[e] += (d * d + f * f) + (g * g + h * h);
      }
    }
    for (; r < r1; r += rpp) {
      const f32x4 v = *(const f32x4*)(src + (base + r) * ld);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const double d = (double)v[e];
        s[e] += d;
        ss[e] += d * d;
      }
    }
    // merge the (at most two) groups this float4 touches, then LDS atomics
    const int g0 = c / cg, g3 = (c + 3) / cg;
    if (g0 == g3) {
      atomicAdd(&acc[g0][0], s[0] + s[1] + s[2] + s[3]);
      atomicAdd(&acc[g0][1], ss[0] + ss[1] + ss[2] + ss[3]);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int g = (c + e) / cg;
        atomicAdd(&acc[g][0], s[e]);
        atomicAdd(&acc[g][1], ss[e]);
      }
    }
  }
  __syncthreads();
  if (t < 64) {
    double* dst = partial + ((int64_t)inst * gridDim.x + chunk) * 64;
    dst[t] = acc[t >> 1][t & 1];
  }
}

// grid = ninst, block = 256: thread (part = t >> 6, comp = t & 63) sums every 4th chunk; lanes
// (2g, 2g+1) of comp hold (sum, sumsq) of group g.
__global__ __launch_bounds__(256) void gn_stats_final_kernel(const double* __restrict__ partial,
                                                             int nchunks, double count, float eps,
                                                             float* __restrict__ stats) {
  __shared__ double red[4][64];
  const int inst = blockIdx.x, t = threadIdx.x & 63, part = threadIdx.x >> 6;
  double a = 0.0;
  const double* src = partial + (int64_t)inst * nchunks * 64 + t;
  for (int i = part; i < nchunks; i += 4) a += src[(int64_t)i * 64];
  red[part][t] = a;
  __syncthreads();
  if (part != 0) return;
  a = red[0][t] + red[1][t] + red[2][t] + red[3][t];
  const double other = __shfl_xor(a, 1);
  const double sum = (t & 1) ? other : a;
  const double sq = (t & 1) ? a : other;
  const double mean = sum / count;
  double var = sq / count - mean * mean;
  if (var < 0.0) var = 0.0;
  const float out = (t & 1) ? (float)(1.0 / sqrt(var + (double)eps)) : (float)mean;
  stats[inst * 64 + t] = out;
}

extern "C" int gcd_groupnorm_stats(const float* x1, int64_t ld1, int C1, const float* x2,
                                   int64_t ld2, int C2, int64_t M, int64_t rows_per_inst, float eps,
                                   double* partial, int nchunks, float* stats, void* stream) {
  const int C = C1 + C2;
  GCD_CHECK_ARG(x1 && partial && stats, "gcd_groupnorm_stats: null pointer");
  GCD_CHECK_ARG(C1 > 0 && C1 % 4 == 0 && C2 >= 0 && C2 % 4 == 0 && C % 32 == 0,
                "gcd_groupnorm_stats: channels C1=%d C2=%d (need %%4 and C%%32)", C1, C2);
  GCD_CHECK_ARG(C2 == 0 || x2, "gcd_groupnorm_stats: x2 null with C2=%d", C2);
  GCD_CHECK_ARG(rows_per_inst > 0 && M > 0 && M % rows_per_inst == 0,
                "gcd_groupnorm_stats: M=%lld not a multiple of rows_per_inst=%lld", (long long)M,
                (long long)rows_per_inst);
  GCD_CHECK_ARG(nchunks > 0 && nchunks <= 65535, "gcd_groupnorm_stats: nchunks=%d", nchunks);
  GCD_CHECK_ARG(ld1 % 4 == 0 && (C2 == 0 || ld2 % 4 == 0), "gcd_groupnorm_stats: ld %% 4");
  const int ninst = (int)(M / rows_per_inst);
  const int rpc = (int)((rows_per_inst + nchunks - 1) / nchunks);
  hipStream_t s = (hipStream_t)stream;
  // thread <-> float4-column map: all C/4 columns at once when they fit a workgroup, else 2..4 passes
  const int cv4 = C / 4;
  int kpass = 1;
  while (cv4 % kpass != 0 || cv4 / kpass > 512) ++kpass;
  const int txw = cv4 / kpass;
  int rpp = 320 / txw;
  if (rpp < 1) rpp = 1;
  hipLaunchKernelGGL(gn_stats_partial_kernel, dim3(nchunks, ninst), dim3(txw * rpp), 0, s, x1, ld1, C1,
                     x2, ld2, C2, rows_per_inst, rpc, txw, kpass, partial);
  GCD_CHECK_LAUNCH();
  const double count = (double)rows_per_inst * (double)(C / 32);
  hipLaunchKernelGGL(gn_stats_final_kernel, dim3(ninst), dim3(256), 0, s, partial, nchunks, count,
                     eps, stats);
  GCD_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// GroupNorm statistics from the column sums a producing GEMM epilogue left behind (gemm_common.h,
// STATS): cs[(2 b + s) * C + c] = sum over the 64 rows of block b of x[., c]^(s + 1).  No pass over x.
// grid = (32 groups, ninst), block 256 / 1024: a block folds its group's cg channels over the instance's
// rows_per_inst / 64 blocks in fp64 (virtual concat: channels >= C1 come from cs2).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void gn_stats_colsums_kernel(const float* __restrict__ cs1, int C1,
                                                               const float* __restrict__ cs2, int C2,
                                                               int blocks_per_inst, double count,
                                                               float eps, float* __restrict__ stats) {
  __shared__ double red[2][16];
  const int g = blockIdx.x, inst = blockIdx.y, t = threadIdx.x;
  const int nthr = blockDim.x, nwave = nthr >> 6;
  const int cg = (C1 + C2) / 32;
  const int64_t b0 = (int64_t)inst * blocks_per_inst;
  const int total = blocks_per_inst * cg;
  double s = 0.0, q = 0.0;
  for (int idx = t; idx < total; idx += nthr) {
    const int b = idx / cg;
    const int c = g * cg + (idx - b * cg);
    const float* src;
    int ld;
    if (c < C1) {
      src = cs1 + c;
      ld = C1;
    } else {
      src = cs2 + (c - C1);
      ld = C2;
    }
    const int64_t row = (b0 + b) * 2;
    s += (double)src[row * ld];
    q += (double)src[(row + 1) * ld];
  }
  s = wave_sum_d(s);
  q = wave_sum_d(q);
  if ((t & 63) == 0) {
    red[0][t >> 6] = s;
    red[1][t >> 6] = q;
  }
  __syncthreads();
  if (t == 0) {
    double sum = 0.0, sq = 0.0;
    for (int w = 0; w < nwave; ++w) {
      sum += red[0][w];
      sq += red[1][w];
    }
    const double mean = sum / count;
    double var = sq / count - mean * mean;
    if (var < 0.0) var = 0.0;
    stats[inst * 64 + 2 * g] = (float)mean;
    stats[inst * 64 + 2 * g + 1] = (float)(1.0 / sqrt(var + (double)eps));
  }
}

extern "C" int gcd_groupnorm_stats_from_colsums(const float* cs1, int C1, const float* cs2, int C2,
                                                int64_t M, int64_t rows_per_inst, float eps,
                                                float* stats, void* stream) {
  const int C = C1 + C2;
  GCD_CHECK_ARG(cs1 && stats, "gcd_groupnorm_stats_from_colsums: null pointer");
  GCD_CHECK_ARG(C1 > 0 && C2 >= 0 && C % 32 == 0 && (C2 == 0 || cs2),
                "gcd_groupnorm_stats_from_colsums: channels C1=%d C2=%d", C1, C2);
  GCD_CHECK_ARG(rows_per_inst > 0 && rows_per_inst % 64 == 0 && M > 0 && M % rows_per_inst == 0,
                "gcd_groupnorm_stats_from_colsums: M=%lld rows_per_inst=%lld (need 64-row blocks that "
                "do not straddle instances)", (long long)M, (long long)rows_per_inst);
  const int64_t bpi = rows_per_inst / 64;
  GCD_CHECK_ARG(bpi * (C / 32) < (1ll << 31), "gcd_groupnorm_stats_from_colsums: instance too large");
  const int ninst = (int)(M / rows_per_inst);
  const double count = (double)rows_per_inst * (double)(C / 32);
  // few instances with many blocks each (the time_stack GroupNorm: 2 clips x 2016 blocks at 72x128):
  // 1024 threads per (group, instance) keep the per-thread chains short
  const int nthr = bpi * (C / 32) > 4096 ? 1024 : 256;
  hipLaunchKernelGGL(gn_stats_colsums_kernel, dim3(32, ninst), dim3(nthr), 0, (hipStream_t)stream, cs1,
                     C1, cs2, C2, (int)bpi, count, eps, stats);
  GCD_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// GroupNorm apply (+SiLU) -> fp16, optional raw fp16 copy.
// grid = (row chunks, ninst); block 256.  Per-channel scale/shift are built once per block in LDS.
// ------------------------------------------------------------------------------------------------
template <int NT>
__global__ __launch_bounds__(256) void gn_apply_kernel(
    const float* __restrict__ x1, int64_t ld1, int C1, const float* __restrict__ x2, int64_t ld2,
    int C2, int64_t rows_per_inst, int rows_per_chunk, const float* __restrict__ stats,
    const float* __restrict__ gamma, const float* __restrict__ beta, int silu, f16* __restrict__ y,
    int64_t ldy, f16* __restrict__ raw, int64_t ldraw) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* sc = (float*)smem_raw;  // [C] scale
  const int C = C1 + C2;
  float* sh = sc + C;            // [C] shift
  const int cg = C / 32;
  const int order = (silu >> 1) & 3;    // walk order of the (instance, chunk) blocks (gcd_walk; see gcd_amd.h)
  const bool obf = (silu >> 3) & 1;     // ABI v7: bfloat16 outputs (the fine-tune step's bf16 operands, written directly)
  silu &= 1;
  const int64_t blk = gcd_walk((int64_t)blockIdx.y * gridDim.x + blockIdx.x, (int64_t)gridDim.x * gridDim.y, order);
  const int inst = (int)(blk / gridDim.x);
  const int chunk = (int)(blk - (int64_t)inst * gridDim.x);
  const int t = threadIdx.x;
  for (int c = t; c < C; c += 256) {
    const int g = c / cg;
    const float mean = stats[inst * 64 + 2 * g], rstd = stats[inst * 64 + 2 * g + 1];
    const float a = rstd * gamma[c];
    sc[c] = a;
    sh[c] = beta[c] - mean * a;
  }
  __syncthreads();
  // One float4 (4 channels) per lane and access: a wave's load covers 1 KB of one row and its fp16
  // store 512 B — whole 128-byte lines per instruction (8 channels per lane left every load
  // instruction with half-used lines).  Two vectors in flight per thread.
  const int cv4 = C >> 2;
  const int64_t r0 = (int64_t)chunk * rows_per_chunk;
  int64_t nrows = rows_per_inst - r0;
  if (nrows > rows_per_chunk) nrows = rows_per_chunk;
  const int64_t base = (int64_t)inst * rows_per_inst + r0;
  const int total = (int)(nrows * cv4);        // a chunk is ~16 K elements: 32-bit indices
  // Flat index idx = r * cv4 + c4 walks the chunk with stride 256: (r, c4) advance by (256 / cv4, 256 % cv4) with one
  // carry — no division in the loop (round 4: the 64-bit idx / cv4 per vector cost more VALU than the normalisation).
  int r = t / cv4, c4 = t - r * cv4;
  const int qs = 256 / cv4, rs = 256 - qs * cv4;
  auto src_of = [&](int rr, int c) -> const float* {
    return (c < C1) ? x1 + (base + rr) * ld1 + c : x2 + (base + rr) * ld2 + (c - C1);
  };
  auto emit = [&](const f32x4 v, int rr, int c) {
    const f32x4 a = *(const f32x4*)(sc + c), b = *(const f32x4*)(sh + c);
    f16x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float u = v[e] * a[e] + b[e];
      if (silu) u = silu_f(u);
      o[e] = gcd_cvt16(u, obf);
    }
    gcd_st<NT, f16x4>(y + (base + rr) * ldy + c, o);
    if (raw) {
      f16x4 q;
#pragma unroll
      for (int e = 0; e < 4; ++e) q[e] = gcd_cvt16(v[e], obf);
      gcd_st<NT, f16x4>(raw + (base + rr) * ldraw + c, q);
    }
  };
  // four vectors (64 B) in flight per thread
  int idx = t;
  for (; idx + 768 < total; idx += 1024) {
    int rr[4], cc[4];
    f32x4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      rr[k] = r;
      cc[k] = c4 * 4;
      v[k] = gcd_ld16<NT>(src_of(r, c4 * 4));
      r += qs;
      c4 += rs;
      if (c4 >= cv4) {
        c4 -= cv4;
        ++r;
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) emit(v[k], rr[k], cc[k]);
  }
  for (; idx < total; idx += 256) {
    const f32x4 v = gcd_ld16<NT>(src_of(r, c4 * 4));
    emit(v, r, c4 * 4);
    r += qs;
    c4 += rs;
    if (c4 >= cv4) {
      c4 -= cv4;
      ++r;
    }
  }
}

extern "C" int gcd_groupnorm_apply(const float* x1, int64_t ld1, int C1, const float* x2,
                                   int64_t ld2, int C2, int64_t M, int64_t rows_per_inst,
                                   const float* stats, const float* gamma, const float* beta,
                                   int silu, void* y16, int64_t ldy, void* raw16, int64_t ldraw,
                                   void* stream) {
  const int C = C1 + C2;
  GCD_CHECK_ARG(x1 && stats && gamma && beta && y16, "gcd_groupnorm_apply: null pointer");
  GCD_CHECK_ARG(C1 > 0 && C1 % 8 == 0 && C2 >= 0 && C2 % 8 == 0 && C % 32 == 0,
                "gcd_groupnorm_apply: channels C1=%d C2=%d (need %%8 and C%%32)", C1, C2);
  GCD_CHECK_ARG(C2 == 0 || x2, "gcd_groupnorm_apply: x2 null with C2=%d", C2);
  GCD_CHECK_ARG(rows_per_inst > 0 && M > 0 && M % rows_per_inst == 0,
                "gcd_groupnorm_apply: M=%lld not a multiple of rows_per_inst=%lld", (long long)M,
                (long long)rows_per_inst);
  GCD_CHECK_ARG(ld1 % 4 == 0 && (C2 == 0 || ld2 % 4 == 0) && ldy % 8 == 0 &&
                    (!raw16 || ldraw % 8 == 0),
                "gcd_groupnorm_apply: leading dimensions must keep 16-byte alignment");
  GCD_CHECK_ARG(C * 8 <= 64 * 1024, "gcd_groupnorm_apply: C=%d too large for the LDS table", C);
  const int ninst = (int)(M / rows_per_inst);
  // ~16 K elements per block keeps >= 1000 blocks in flight at the UNet's sizes
  int rpc = (int)((16384 + C - 1) / C);
  if (rpc < 1) rpc = 1;
  int64_t nchunks = (rows_per_inst + rpc - 1) / rpc;
  if (nchunks > 65535) {
    nchunks = 65535;
    rpc = (int)((rows_per_inst + nchunks - 1) / nchunks);
    nchunks = (rows_per_inst + rpc - 1) / rpc;
  }
#define GCD_GN_APPLY(NT)                                                                                  \
  hipLaunchKernelGGL(gn_apply_kernel<NT>, dim3((unsigned)nchunks, ninst), dim3(256), C * 8, (hipStream_t)stream, \
                     x1, ld1, C1, x2, ld2, C2, rows_per_inst, rpc, stats, gamma, beta, silu, (f16*)y16, ldy,     \
                     (f16*)raw16, ldraw)
  switch (gcd_stream_nt((int64_t)M * C * 6)) {
    case 1: GCD_GN_APPLY(1); break;
    case 2: GCD_GN_APPLY(2); break;
    case 3: GCD_GN_APPLY(3); break;
    default: GCD_GN_APPLY(0); break;
  }
#undef GCD_GN_APPLY
  GCD_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// LayerNorm: one wavefront per token row, row held in registers (C <= 2048).
// ------------------------------------------------------------------------------------------------
template <int NV>  // float4 vectors per lane
__global__ __launch_bounds__(256) void layernorm_kernel(
    const float* __restrict__ x, int64_t ldx, int64_t M, int C, const float* __restrict__ gamma,
    const float* __restrict__ beta, float eps, const float* __restrict__ addvec, int64_t ld_addvec,
    int rows_per_vec, float* __restrict__ sum_out, int64_t ld_sum, f16* __restrict__ y,
    int64_t ldy, int order) {
  const int lane = threadIdx.x & 63;
  const int cv4 = C >> 2;
  const int64_t nrb = (M + 3) >> 2;   // blocks of 4 rows (one per wave), walked in `order`
  for (int64_t pos = blockIdx.x; pos < nrb; pos += gridDim.x) {
    const int64_t m = gcd_walk(pos, nrb, order & 3) * 4 + (threadIdx.x >> 6);
    if (m >= M) continue;
    const float* row = x + m * ldx;
    const float* av = addvec ? addvec + (m / rows_per_vec) * ld_addvec : nullptr;
    f32x4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int cv = lane + 64 * i;
      if (cv < cv4) {
        v[i] = *(const f32x4*)(row + cv * 4);
        if (av) v[i] += *(const f32x4*)(av + cv * 4);
        s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
      } else {
        v[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int cv = lane + 64 * i;
      if (cv < cv4) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float d = v[i][e] - mean;
          q += d * d;
        }
      }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int cv = lane + 64 * i;
      if (cv < cv4) {
        const f32x4 g = *(const f32x4*)(gamma + cv * 4), b = *(const f32x4*)(beta + cv * 4);
        f16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = gcd_cvt16((v[i][e] - mean) * rstd * g[e] + b[e], (order & 4) != 0);
        *(f16x4*)(y + m * ldy + cv * 4) = o;
        if (sum_out) *(f32x4*)(sum_out + m * ld_sum + cv * 4) = v[i];
      }
    }
  }
}

// Four (two) rows per wavefront, LPR = 16 (32) lanes per row, for C = 320 (640).  The kernel above
// keeps ONE row per wave in flight behind two 6-step cross-lane reductions and leaves 3/8 of its lanes
// idle at C = 320 (measured 3.6 TB/s); here a 16-lane group reads 256 contiguous bytes of its row per
// instruction, the reductions are four DPP adds inside the 16-lane row (no LDS traffic, one extra
// shuffle for 32 lanes), and the independent rows of a wave overlap each other's latency.
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));  // row_mirror
  return v;
}

template <int NV, int LPR, int NT>  // float4 vectors per lane, lanes per row: C = 4 * NV * LPR; NT: gcd_ld16 / gcd_st
__global__ __launch_bounds__(256) void layernorm16_kernel(
    const float* __restrict__ x, int64_t ldx, int64_t M, const float* __restrict__ gamma,
    const float* __restrict__ beta, float eps, const float* __restrict__ addvec, int64_t ld_addvec,
    int rows_per_vec, float* __restrict__ sum_out, int64_t ld_sum, f16* __restrict__ y,
    int64_t ldy, int order) {
  constexpr int C = NV * LPR * 4, RS = LPR * 4;   // RS: floats between a lane's consecutive vectors
  constexpr int RPB = 256 / LPR;                  // rows per workgroup
  const int l16 = threadIdx.x & (LPR - 1);
  const int64_t nrb = (M + RPB - 1) / RPB;        // row blocks, walked in `order`
  auto row_sum = [](float v) {
    v = row16_sum(v);
    if (LPR == 32) v += __shfl_xor(v, 16);
    return v;
  };
  for (int64_t pos = blockIdx.x; pos < nrb; pos += gridDim.x) {
    const int64_t m = gcd_walk(pos, nrb, order & 3) * RPB + threadIdx.x / LPR;
    if (m >= M) continue;   // (whole 16- / 32-lane row groups leave together: the DPP sums stay inside a group)
    const float* row = x + m * ldx + l16 * 4;
    f32x4 v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = gcd_ld16<NT>(row + RS * i);
    if (addvec) {
      const float* av = addvec + (m / rows_per_vec) * ld_addvec + l16 * 4;
#pragma unroll
      for (int i = 0; i < NV; ++i) v[i] += *(const f32x4*)(av + RS * i);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    const float mean = row_sum(s) * (1.0f / (float)C);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d = v[i][e] - mean;
        q += d * d;
      }
    const float rstd = rsqrtf(row_sum(q) * (1.0f / (float)C) + eps);
    f16* yr = y + m * ldy + l16 * 4;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const f32x4 g = *(const f32x4*)(gamma + l16 * 4 + RS * i), b = *(const f32x4*)(beta + l16 * 4 + RS * i);
      f16x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = gcd_cvt16((v[i][e] - mean) * rstd * g[e] + b[e], (order & 4) != 0);
      gcd_st<NT, f16x4>(yr + RS * i, o);
      if (sum_out) gcd_st<NT, f32x4>(sum_out + m * ld_sum + l16 * 4 + RS * i, v[i]);
    }
  }
}

extern "C" int gcd_layernorm_f16(const float* x, int64_t ldx, int64_t M, int C, const float* gamma,
                                 const float* beta, float eps, const float* addvec,
                                 int64_t ld_addvec, int rows_per_vec, float* sum_out,
                                 int64_t ld_sum, void* y16, int64_t ldy, int order, void* stream) {
  GCD_CHECK_ARG(x && gamma && beta && y16, "gcd_layernorm_f16: null pointer");
  GCD_CHECK_ARG(M > 0 && C > 0 && C % 4 == 0 && C <= 2048, "gcd_layernorm_f16: M=%lld C=%d",
                (long long)M, C);
  GCD_CHECK_ARG(ldx % 4 == 0 && ldy % 4 == 0, "gcd_layernorm_f16: ld alignment");
  if (addvec)
    GCD_CHECK_ARG(rows_per_vec > 0 && ld_addvec % 4 == 0, "gcd_layernorm_f16: addvec geometry");
  if (sum_out) GCD_CHECK_ARG(ld_sum % 4 == 0, "gcd_layernorm_f16: ld_sum alignment");
  GCD_CHECK_ARG(order >= 0 && order <= 7, "gcd_layernorm_f16: order=%d (0..3, + 4 for bfloat16 output)", order);
  hipStream_t s = (hipStream_t)stream;
  if (C == 320 || C == 640) {   // 16 / 32 lanes per row, 5 vectors per lane
    const int rpb = C == 320 ? 16 : 8;   // rows per 256-thread block
    int64_t blocks16 = (M + rpb - 1) / rpb;
    if (blocks16 > 8192) blocks16 = 8192;
#define GCD_LN16(LPR, NT)                                                                                    \
  hipLaunchKernelGGL((layernorm16_kernel<5, LPR, NT>), dim3((unsigned)blocks16), dim3(256), 0, s, x, ldx, M, gamma, \
                     beta, eps, addvec, ld_addvec, rows_per_vec, sum_out, ld_sum, (f16*)y16, ldy, order)
    const int nt = gcd_stream_nt(M * C * 6);
    if (C == 320) {
      switch (nt) {
        case 1: GCD_LN16(16, 1); break;
        case 2: GCD_LN16(16, 2); break;
        case 3: GCD_LN16(16, 3); break;
        default: GCD_LN16(16, 0); break;
      }
    } else {
      switch (nt) {
        case 1: GCD_LN16(32, 1); break;
        case 2: GCD_LN16(32, 2); break;
        case 3: GCD_LN16(32, 3); break;
        default: GCD_LN16(32, 0); break;
      }
    }
#undef GCD_LN16
    GCD_CHECK_LAUNCH();
    return 0;
  }
  int64_t blocks = (M + 3) / 4;
  if (blocks > 16384) blocks = 16384;  // grid-stride beyond 8 workgroups per CU
  const int nv = (C / 4 + 63) / 64;
#define GCD_LN_LAUNCH(NV)                                                                        \
  hipLaunchKernelGGL(layernorm_kernel<NV>, dim3((unsigned)blocks), dim3(256), 0, s, x, ldx, M, C, \
                     gamma, beta, eps, addvec, ld_addvec, rows_per_vec, sum_out, ld_sum,         \
                     (f16*)y16, ldy, order)
  switch (nv) {
    case 1: GCD_LN_LAUNCH(1); break;
    case 2: GCD_LN_LAUNCH(2); break;
    case 3: GCD_LN_LAUNCH(3); break;
    case 4: GCD_LN_LAUNCH(4); break;
    case 5: GCD_LN_LAUNCH(5); break;
    case 6: GCD_LN_LAUNCH(6); break;
    case 7: GCD_LN_LAUNCH(7); break;
    default: GCD_LN_LAUNCH(8); break;
  }
#undef GCD_LN_LAUNCH
  GCD_CHECK_LAUNCH();
  return 0;
}
