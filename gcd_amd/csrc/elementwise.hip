// elementwise.hip — the small kernels around the contractions: tiny per-frame MLPs (fp32), input
// packing / output unpacking between the reference's NCHW fp32 boundary and the token-major device
// layout, casts, and the fused CFG + denoiser-affine + Euler update of the sampler.
#include "common.h"

// ------------------------------------------------------------------------------------------------
// y[M,N] = [y +] act_out( act_in(x[M,K]) @ W[N,K]^T + b )   fp32, any M (32 rows per launch).
// A skinny GEMM that only has to stream W (up to 190 MB for the 44 stacked emb_layers) at HBM rate:
// workgroup = 4 waves x 4 output columns; x is staged (activation applied) through LDS in K chunks
// of 256; a lane owns (column lane >> 4, k-slot lane & 15): the 16 lanes of a column read 256
// contiguous bytes of its W row per instruction and keep MMAX row accumulators, the x operands come
// from LDS as 16-byte reads that are broadcast over the 4 columns.  One 16-lane reduction at the end.
// ------------------------------------------------------------------------------------------------
#define SM_KC 256
#define SM_LDX (SM_KC + 4)

template <int MMAX>
__global__ __launch_bounds__(256, 2) void linear_smallm_kernel(const float* __restrict__ x, int64_t ldx,
                                                            const float* __restrict__ W,
                                                            const float* __restrict__ b,
                                                            float* __restrict__ y, int64_t ldy, int M,
                                                            int N, int K, int flags) {
  __shared__ __attribute__((aligned(16))) float xs[MMAX * SM_LDX];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int kp = lane & 15;
  const int n = blockIdx.x * 16 + wave * 4 + (lane >> 4);
  const float* wrow = W + (int64_t)(n < N ? n : N - 1) * K;
  float acc[MMAX];
#pragma unroll
  for (int m = 0; m < MMAX; ++m) acc[m] = 0.f;
  for (int kc = 0; kc < K; kc += SM_KC) {
    // stage x[:, kc:kc+256] (zero padded) with the input activation applied
    for (int idx = t; idx < MMAX * 64; idx += 256) {
      const int mm = idx >> 6, kv = (idx & 63) * 4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (mm < M && kc + kv < K) {
        v = *(const f32x4*)(x + (int64_t)mm * ldx + kc + kv);
        if (flags & 1) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = silu_f(v[e]);
        }
      }
      *(f32x4*)(xs + mm * SM_LDX + kv) = v;
    }
    __syncthreads();
    f32x4 wv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = kc + 64 * i + kp * 4;
      wv[i] = k < K ? *(const f32x4*)(wrow + k) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float* xr = xs + 64 * i + kp * 4;
#pragma unroll
      for (int m = 0; m < MMAX; ++m) {
        const f32x4 xv = *(const f32x4*)(xr + m * SM_LDX);
        float a = acc[m];
        a = fmaf(xv[0], wv[i][0], a);
        a = fmaf(xv[1], wv[i][1], a);
        a = fmaf(xv[2], wv[i][2], a);
        a = fmaf(xv[3], wv[i][3], a);
        acc[m] = a;
      }
    }
    __syncthreads();
  }
  // reduce over the 16 k-slots of a column; afterwards every lane of the group holds the totals
#pragma unroll
  for (int m = 0; m < MMAX; ++m) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) acc[m] += __shfl_xor(acc[m], o);
  }
  if (n < N) {
    const float bn = b ? b[n] : 0.f;
#pragma unroll
    for (int m = 0; m < MMAX; ++m) {
      if ((m & 15) == kp && m < M) {   // lane kp of the group writes rows kp and kp + 16
        float v = acc[m] + bn;
        if (flags & 2) v = silu_f(v);
        float* dst = y + (int64_t)m * ldy + n;
        if (flags & 4) v += *dst;
        *dst = v;
      }
    }
  }
}

// The same product for WIDE weights (N >= 4096: the 44 stacked emb_layers of a step are one [~40 000, 1280] fp32 matrix,
// 205 MB) on the fp32 matrix pipe.  The VALU kernel above re-reads its x operand from LDS once per weight row (128
// ds_read_b128 per 4 KB of W and wave): 243 us = 0.85 TB/s for that launch.  Here a wave owns 16 weight rows; lane
// (r, kg) loads W[n0 + r][k + 4 kg .. + 3] (64 contiguous bytes per row and instruction, 128 k in flight per wave
// under the MFMAs of the previous 128), x comes from LDS once per 16 rows of W, and
// v_mfma_f32_16x16x4_f32 (fp32 products, fp32 accumulation — the arithmetic of the VALU kernel up to summation order) does
// 16 x 16 x 4 per instruction: 8 B/clk/SIMD of W at two row blocks of x, above what HBM delivers per CU.
template <int MB>      // 16-row blocks of x (M <= 16 MB)
__global__ __launch_bounds__(256, 4) void linear_smallm_mfma_kernel(const float* __restrict__ x, int64_t ldx,
                                                                    const float* __restrict__ W, const float* __restrict__ b,
                                                                    float* __restrict__ y, int64_t ldy, int M, int N, int K,
                                                                    int flags) {
  __shared__ __attribute__((aligned(16))) float xs[MB * 16 * SM_LDX];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int r = lane & 15, kg = lane >> 4;
  const int n0 = (blockIdx.x * 4 + wave) * 16;
  const int n = n0 + r;
  const float* wrow = W + (int64_t)(n < N ? n : N - 1) * K + 4 * kg;
  f32x4 acc[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) acc[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
  // W in units of 128 k (8 steps of 16), two register sets: the next unit's loads are issued before this unit's MFMAs
  // (64 + ~40 registers: four workgroups per CU, so that the ~630 workgroups of the wide launch are all resident at once)
  f32x4 wa[8], wb[8];
  auto load_w = [&](int k0, f32x4 (&w)[8]) {
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const int k = k0 + 16 * ks + 4 * kg;
      w[ks] = k < K ? __builtin_nontemporal_load((const f32x4*)(wrow + k0 + 16 * ks)) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto stage_x = [&](int kc) {
    for (int idx = t; idx < MB * 16 * 64; idx += 256) {
      const int mm = idx >> 6, kv = (idx & 63) * 4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (mm < M && kc + kv < K) {
        v = *(const f32x4*)(x + (int64_t)mm * ldx + kc + kv);
        if (flags & 1) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = silu_f(v[e]);
        }
      }
      *(f32x4*)(xs + mm * SM_LDX + kv) = v;
    }
  };
  auto compute = [&](int half, const f32x4 (&w)[8]) {
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        const f32x4 xv = *(const f32x4*)(xs + (mb * 16 + r) * SM_LDX + 128 * half + 16 * ks + 4 * kg);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[j], w[ks][j], acc[mb], 0, 0, 0);
      }
    }
  };
  load_w(0, wa);
  for (int kc = 0; kc < K; kc += SM_KC) {
    stage_x(kc);
    __syncthreads();
    load_w(kc + 128, wb);      // (beyond K: zeros, no loads)
    __builtin_amdgcn_sched_barrier(0);
    compute(0, wa);
    __builtin_amdgcn_sched_barrier(0);
    load_w(kc + SM_KC, wa);
    __builtin_amdgcn_sched_barrier(0);
    compute(1, wb);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
  }
  // accumulator layout: lane (r, kg) holds y[16 mb + 4 kg + e][n0 + r]
  if (n < N) {
    const float bn = b ? b[n] : 0.f;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int m = 16 * mb + 4 * kg + e;
        if (m < M) {
          float v = acc[mb][e] + bn;
          if (flags & 2) v = silu_f(v);
          float* dst = y + (int64_t)m * ldy + n;
          if (flags & 4) v += *dst;
          *dst = v;
        }
      }
    }
  }
}

extern "C" int gcd_linear_smallm_f32(const float* x, int64_t ldx, const float* W, const float* b,
                                     float* y, int64_t ldy, int M, int N, int K, int act_flags,
                                     void* stream) {
  GCD_CHECK_ARG(x && W && y, "gcd_linear_smallm_f32: null pointer");
  GCD_CHECK_ARG(M >= 1, "gcd_linear_smallm_f32: M=%d (must be >= 1)", M);
  GCD_CHECK_ARG(N >= 1 && K >= 4 && K % 4 == 0 && ldx % 4 == 0,
                "gcd_linear_smallm_f32: N=%d K=%d ldx=%lld (K, ldx must be multiples of 4)", N, K,
                (long long)ldx);
  const dim3 grid((N + 15) / 16), block(256);
  hipStream_t s = (hipStream_t)stream;
  // any M: rows go through the kernel 32 at a time (two clips under CFG are 56 frames; the weights
  // of one launch are a few MB and stay in L2 / Infinity Cache for the next chunk)
  for (int m0 = 0; m0 < M; m0 += 32) {
    const int mc = M - m0 < 32 ? M - m0 : 32;
    const float* xc = x + (int64_t)m0 * ldx;
    float* yc = y + (int64_t)m0 * ldy;
    if (N >= 4096 && mc > 4) {      // wide weights: the fp32 matrix-pipe kernel streams them at the memory rate
      const dim3 gridw((N + 63) / 64);
      if (mc <= 16)
        hipLaunchKernelGGL(linear_smallm_mfma_kernel<1>, gridw, block, 0, s, xc, ldx, W, b, yc, ldy, mc, N, K, act_flags);
      else
        hipLaunchKernelGGL(linear_smallm_mfma_kernel<2>, gridw, block, 0, s, xc, ldx, W, b, yc, ldy, mc, N, K, act_flags);
    } else if (mc <= 4)
      hipLaunchKernelGGL(linear_smallm_kernel<4>, grid, block, 0, s, xc, ldx, W, b, yc, ldy, mc, N, K, act_flags);
    else if (mc <= 16)
      hipLaunchKernelGGL(linear_smallm_kernel<16>, grid, block, 0, s, xc, ldx, W, b, yc, ldy, mc, N, K, act_flags);
    else
      hipLaunchKernelGGL(linear_smallm_kernel<32>, grid, block, 0, s, xc, ldx, W, b, yc, ldy, mc, N, K, act_flags);
  }
  GCD_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// NCHW fp32 -> token-major fp16 with per-frame c_in scaling, CFG duplication and channel concat.
// out16[(n*HW + p)*Cpad + c]: c < Cx: x[n % nx][c][p] * c_in[n]; Cx <= c < Cx+Cc: concat[n][c-Cx][p];
// rest 0 (padding up to the GEMM's 64-channel K granule).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_input_kernel(const float* __restrict__ x, int nx, int Cx,
                                                         const float* __restrict__ cc, int Cc,
                                                         const float* __restrict__ c_in, int N,
                                                         int HW, f16* __restrict__ out, int Cpad) {
  const int64_t total = (int64_t)N * HW;
  const int nvec = Cpad >> 3;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * 256) {
    const int n = (int)(idx / HW);
    const int p = (int)(idx - (int64_t)n * HW);
    const float ci = c_in ? c_in[n] : 1.0f;
    const float* xs = x + (int64_t)(n % nx) * Cx * HW + p;
    const float* cs = cc ? cc + (int64_t)n * Cc * HW + p : nullptr;
    f16* dst = out + idx * Cpad;
    for (int v = 0; v < nvec; ++v) {
      f16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c = v * 8 + e;
        float val = 0.f;
        if (c < Cx) val = xs[(int64_t)c * HW] * ci;
        else if (c < Cx + Cc) val = cs[(int64_t)(c - Cx) * HW];
        o[e] = (f16)val;
      }
      *(f16x8*)(dst + v * 8) = o;
    }
  }
}

extern "C" int gcd_pack_input(const float* x, int nx, int Cx, const float* concat, int Cc,
                              const float* c_in, int N, int HW, void* out16, int Cpad,
                              void* stream) {
  GCD_CHECK_ARG(x && out16, "gcd_pack_input: null pointer");
  GCD_CHECK_ARG(nx > 0 && N > 0 && N % nx == 0 && HW > 0, "gcd_pack_input: N=%d nx=%d HW=%d", N, nx,
                HW);
  GCD_CHECK_ARG(Cc == 0 || concat, "gcd_pack_input: concat null with Cc=%d", Cc);
  GCD_CHECK_ARG(Cpad % 8 == 0 && Cpad >= Cx + Cc, "gcd_pack_input: Cpad=%d < %d", Cpad, Cx + Cc);
  int64_t blocks = ((int64_t)N * HW + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(pack_input_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x,
                     nx, Cx, concat, Cc, c_in, N, HW, (f16*)out16, Cpad);
  GCD_CHECK_LAUNCH();
  return 0;
}

// token-major fp32 [N*HW, ld] (first Cout channels) -> NCHW fp32 [N, Cout, HW]
__global__ __launch_bounds__(256) void unpack_output_kernel(const float* __restrict__ in, int64_t ld,
                                                            float* __restrict__ out, int Cout, int N,
                                                            int HW) {
  const int64_t total = (int64_t)N * HW;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * 256) {
    const int n = (int)(idx / HW);
    const int p = (int)(idx - (int64_t)n * HW);
    for (int c = 0; c < Cout; ++c) out[((int64_t)n * Cout + c) * HW + p] = in[idx * ld + c];
  }
}

extern "C" int gcd_unpack_output(const float* in, int64_t ld, float* out, int Cout, int N, int HW,
                                 void* stream) {
  GCD_CHECK_ARG(in && out, "gcd_unpack_output: null pointer");
  GCD_CHECK_ARG(Cout > 0 && N > 0 && HW > 0 && ld >= Cout, "gcd_unpack_output: bad geometry");
  int64_t blocks = ((int64_t)N * HW + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(unpack_output_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     in, ld, out, Cout, N, HW);
  GCD_CHECK_LAUNCH();
  return 0;
}

// AE3DConv's time_mix_conv (Conv3d C -> C, kernel (3,1,1), zero padded in time) fused with the
// token-major -> NCHW layout change:
//   out[n][co][p] = b[co] + sum_{dt, ci} w[co][ci][dt] * in[((n + dt - 1) * HW + p) * ld + ci]
// for frames n + dt - 1 inside the clip of T frames that holds n.  C <= 4 (RGB): HBM-bound.
template <int C>
__global__ __launch_bounds__(256) void time_mix_unpack_kernel(const float* __restrict__ in, int64_t ld,
                                                              const float* __restrict__ w,
                                                              const float* __restrict__ b,
                                                              float* __restrict__ out, int N, int T,
                                                              int HW) {
  float wr[C][C][3], br[C];
#pragma unroll
  for (int co = 0; co < C; ++co) {
    br[co] = b[co];
#pragma unroll
    for (int ci = 0; ci < C; ++ci)
#pragma unroll
      for (int dt = 0; dt < 3; ++dt) wr[co][ci][dt] = w[(co * C + ci) * 3 + dt];
  }
  const int64_t total = (int64_t)N * HW;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * 256) {
    const int n = (int)(idx / HW);
    const int p = (int)(idx - (int64_t)n * HW);
    const int t = n % T;
    float acc[C];
#pragma unroll
    for (int co = 0; co < C; ++co) acc[co] = br[co];
#pragma unroll
    for (int dt = 0; dt < 3; ++dt) {
      const int tt = t + dt - 1;
      if (tt < 0 || tt >= T) continue;
      const float* src = in + (idx + (int64_t)(dt - 1) * HW) * ld;
      float v[C];
      if (C == 4) {
        const f32x4 q = *(const f32x4*)src;
#pragma unroll
        for (int ci = 0; ci < C; ++ci) v[ci] = q[ci];
      } else {
#pragma unroll
        for (int ci = 0; ci < C; ++ci) v[ci] = src[ci];
      }
#pragma unroll
      for (int co = 0; co < C; ++co)
#pragma unroll
        for (int ci = 0; ci < C; ++ci) acc[co] = fmaf(wr[co][ci][dt], v[ci], acc[co]);
    }
#pragma unroll
    for (int co = 0; co < C; ++co) out[((int64_t)n * C + co) * HW + p] = acc[co];
  }
}

extern "C" int gcd_time_mix_unpack(const float* in, int64_t ld, const float* w, const float* b,
                                   float* out, int C, int N, int T, int HW, void* stream) {
  GCD_CHECK_ARG(in && w && b && out, "gcd_time_mix_unpack: null pointer");
  GCD_CHECK_ARG(C >= 1 && C <= 4 && N > 0 && T > 0 && N % T == 0 && HW > 0 && ld >= C && ld % 4 == 0,
                "gcd_time_mix_unpack: C=%d (1..4) N=%d T=%d HW=%d ld=%lld", C, N, T, HW, (long long)ld);
  int64_t blocks = ((int64_t)N * HW + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  const dim3 g((unsigned)blocks), blk(256);
  hipStream_t s = (hipStream_t)stream;
  switch (C) {
    case 1: hipLaunchKernelGGL(time_mix_unpack_kernel<1>, g, blk, 0, s, in, ld, w, b, out, N, T, HW); break;
    case 2: hipLaunchKernelGGL(time_mix_unpack_kernel<2>, g, blk, 0, s, in, ld, w, b, out, N, T, HW); break;
    case 3: hipLaunchKernelGGL(time_mix_unpack_kernel<3>, g, blk, 0, s, in, ld, w, b, out, N, T, HW); break;
    default: hipLaunchKernelGGL(time_mix_unpack_kernel<4>, g, blk, 0, s, in, ld, w, b, out, N, T, HW); break;
  }
  GCD_CHECK_LAUNCH();
  return 0;
}

// fp32 [M, C] (ld) -> fp16 [M, C] (ld), C % 8 == 0
__global__ __launch_bounds__(256) void cast_f32_f16_kernel(const float* __restrict__ x, int64_t ldx,
                                                           f16* __restrict__ y, int64_t ldy,
                                                           int64_t M, int C) {
  const int cv8 = C >> 3;
  const int64_t total = M * cv8;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * 256) {
    const int64_t r = idx / cv8;
    const int c = (int)(idx - r * cv8) * 8;
    const f32x4 a = *(const f32x4*)(x + r * ldx + c), b = *(const f32x4*)(x + r * ldx + c + 4);
    f16x8 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      o[e] = (f16)a[e];
      o[e + 4] = (f16)b[e];
    }
    *(f16x8*)(y + r * ldy + c) = o;
  }
}

extern "C" int gcd_cast_f32_f16(const float* x, int64_t ldx, void* y16, int64_t ldy, int64_t M, int C,
                                void* stream) {
  GCD_CHECK_ARG(x && y16, "gcd_cast_f32_f16: null pointer");
  GCD_CHECK_ARG(M > 0 && C > 0 && C % 8 == 0 && ldx % 4 == 0 && ldy % 8 == 0,
                "gcd_cast_f32_f16: M=%lld C=%d ldx=%lld ldy=%lld", (long long)M, C, (long long)ldx,
                (long long)ldy);
  int64_t blocks = (M * (C / 8) + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(cast_f32_f16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     x, ldx, (f16*)y16, ldy, M, C);
  GCD_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Pieces of the first-stage VideoDecoder's single-head mid attention (head dim = 512 channels, so it
// runs as two GEMMs with these two kernels in between; diffusionmodules/model.py:164-202).
//   softmax over the rows of an fp32 score matrix -> fp16 probabilities; one workgroup per row, the
//   row lives in registers (C <= 16384), max / sum by wavefront shuffles + LDS;
//   fp16 [R, C] -> [C, R] transpose through a padded 64 x 64 LDS tile (V -> V^T, the "W" operand of P V).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ x, int64_t ldx,
                                                           f16* __restrict__ y, int64_t ldy, int C) {
  __shared__ float red[8];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const float* row = x + (int64_t)blockIdx.x * ldx;
  f32x4 v[16];
  const int cv4 = C >> 2;
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = t + 256 * i;
    if (c < cv4) {
      v[i] = *(const f32x4*)(row + c * 4);
      mx = fmaxf(fmaxf(mx, fmaxf(v[i][0], v[i][1])), fmaxf(v[i][2], v[i][3]));
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = t + 256 * i;
    if (c < cv4) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[i][e] = __expf(v[i][e] - mx);
        sum += v[i][e];
      }
    }
  }
  sum = wave_sum(sum);
  if (lane == 0) red[4 + wave] = sum;
  __syncthreads();
  const float inv = 1.0f / (red[4] + red[5] + red[6] + red[7]);
  f16* out = y + (int64_t)blockIdx.x * ldy;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = t + 256 * i;
    if (c < cv4) {
      f16x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (f16)(v[i][e] * inv);
      *(f16x4*)(out + c * 4) = o;
    }
  }
}

extern "C" int gcd_softmax_rows_f16(const float* x, int64_t ldx, void* y16, int64_t ldy, int64_t R,
                                    int C, void* stream) {
  GCD_CHECK_ARG(x && y16, "gcd_softmax_rows_f16: null pointer");
  GCD_CHECK_ARG(R > 0 && R < (1ll << 31) && C > 0 && C % 4 == 0 && C <= 16384 && ldx % 4 == 0 &&
                    ldy % 4 == 0,
                "gcd_softmax_rows_f16: R=%lld C=%d (C %% 4 == 0, <= 16384) ldx=%lld ldy=%lld",
                (long long)R, C, (long long)ldx, (long long)ldy);
  hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)R), dim3(256), 0, (hipStream_t)stream, x, ldx,
                     (f16*)y16, ldy, C);
  GCD_CHECK_LAUNCH();
  return 0;
}

__global__ __launch_bounds__(256) void transpose_f16_kernel(const f16* __restrict__ x, int64_t ldx,
                                                            f16* __restrict__ y, int64_t ldy, int R,
                                                            int C) {
  __shared__ f16 tile[64][66];
  const int t = threadIdx.x;
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int r = (t >> 6) + 4 * i, c = t & 63;
    tile[r][c] = (r0 + r < R && c0 + c < C) ? x[(int64_t)(r0 + r) * ldx + c0 + c] : (f16)0;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = (t >> 6) + 4 * i, r = t & 63;
    if (c0 + c < C && r0 + r < R) y[(int64_t)(c0 + c) * ldy + r0 + r] = tile[r][c];
  }
}

// The same transpose on 16-byte vectors (the operands of the fine-tune step's weight gradients are whole
// activation tensors transposed so that the token axis becomes the GEMM's contraction axis): a 64 x 64 tile is
// loaded as f16x8 rows, read back column-wise from LDS and stored as two f16x8 per lane — 32 contiguous bytes of
// an output row per lane instead of 2.  Rows r >= R of the tile enter as zeros and are written too (up to Rp
// output columns): the zero padding of the contraction axis costs no separate fill.
__global__ __launch_bounds__(256) void transpose_f16_vec_kernel(const f16* __restrict__ x, int64_t ldx,
                                                                f16* __restrict__ y, int64_t ldy, int R,
                                                                int C, int Rp) {
  __shared__ __attribute__((aligned(16))) f16 tile[64][72];
  const int t = threadIdx.x;
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int p = i * 256 + t;
    const int r = p >> 3, ch = p & 7;
    f16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (r0 + r < R && c0 + ch * 8 < C) v = *(const f16x8*)(x + (int64_t)(r0 + r) * ldx + c0 + ch * 8);
    *(f16x8*)(&tile[r][ch * 8]) = v;
  }
  __syncthreads();
  const int c = t >> 2, rg = (t & 3) * 16;
  if (c0 + c < C && r0 + rg < Rp) {
    f16x8 o0, o1;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      o0[e] = tile[rg + e][c];
      o1[e] = tile[rg + 8 + e][c];
    }
    f16* dst = y + (int64_t)(c0 + c) * ldy + r0 + rg;
    *(f16x8*)dst = o0;
    *(f16x8*)(dst + 8) = o1;
  }
}

extern "C" int gcd_transpose_f16(const void* x, int64_t ldx, void* y, int64_t ldy, int R, int C,
                                 void* stream) {
  GCD_CHECK_ARG(x && y, "gcd_transpose_f16: null pointer");
  GCD_CHECK_ARG(R > 0 && C > 0 && ldx >= C && ldy >= R, "gcd_transpose_f16: R=%d C=%d ldx=%lld ldy=%lld",
                R, C, (long long)ldx, (long long)ldy);
  if (R % 16 == 0 && C % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0) {
    hipLaunchKernelGGL(transpose_f16_vec_kernel, dim3((C + 63) / 64, (R + 63) / 64), dim3(256), 0,
                       (hipStream_t)stream, (const f16*)x, ldx, (f16*)y, ldy, R, C, R);
    GCD_CHECK_LAUNCH();
    return 0;
  }
  hipLaunchKernelGGL(transpose_f16_kernel, dim3((C + 63) / 64, (R + 63) / 64), dim3(256), 0,
                     (hipStream_t)stream, (const f16*)x, ldx, (f16*)y, ldy, R, C);
  GCD_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Sampler kernels
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cfg_euler_kernel(const float* __restrict__ x,
                                                        const float* __restrict__ net,
                                                        const float* __restrict__ scale,
                                                        const float* __restrict__ sig,
                                                        float* __restrict__ xo, int nx, int T,
                                                        int64_t chw) {
  const float sigma = sig[0], sigma_next = sig[1];
  const float s2 = sigma * sigma + 1.0f;
  const float c_skip = 1.0f / s2;
  const float c_out = -sigma / sqrtf(s2);
  const float dt = sigma_next - sigma;
  const int64_t total = (int64_t)nx * chw;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * 256) {
    const int n = (int)(idx / chw);
    const float xv = x[idx];
    const float du = net[idx] * c_out + xv * c_skip;
    const float dc = net[idx + total] * c_out + xv * c_skip;
    const float den = du + scale[n % T] * (dc - du);
    const float d = (xv - den) / sigma;
    xo[idx] = xv + dt * d;
  }
}

extern "C" int gcd_cfg_euler_step(const float* x, const float* net, const float* scale,
                                  const float* sig, float* x_out, int nx, int T, int64_t chw,
                                  void* stream) {
  GCD_CHECK_ARG(x && net && scale && sig && x_out, "gcd_cfg_euler_step: null pointer");
  GCD_CHECK_ARG(nx > 0 && T > 0 && chw > 0, "gcd_cfg_euler_step: empty problem");
  int64_t blocks = ((int64_t)nx * chw + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(cfg_euler_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x,
                     net, scale, sig, x_out, nx, T, chw);
  GCD_CHECK_LAUNCH();
  return 0;
}

__global__ void edm_scalings_kernel(const float* __restrict__ sig, float* __restrict__ c_in,
                                    float* __restrict__ c_noise, int N) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const float s = sig[0];
  c_in[n] = 1.0f / sqrtf(s * s + 1.0f);
  c_noise[n] = 0.25f * logf(s);
}

extern "C" int gcd_edm_scalings(const float* sig, float* c_in, float* c_noise, int N, void* stream) {
  GCD_CHECK_ARG(sig && c_in && c_noise && N > 0, "gcd_edm_scalings: bad arguments");
  hipLaunchKernelGGL(edm_scalings_kernel, dim3((N + 63) / 64), dim3(64), 0, (hipStream_t)stream, sig,
                     c_in, c_noise, N);
  GCD_CHECK_LAUNCH();
  return 0;
}

__global__ void timestep_embedding_kernel(const float* __restrict__ t, float* __restrict__ emb, int N,
                                          int dim, float max_period) {
  const int half = dim / 2;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= N * half) return;
  const int n = idx / half, k = idx - n * half;
  const float f = expf(-logf(max_period) * (float)k / (float)half);
  const float a = t[n] * f;
  emb[(int64_t)n * dim + k] = cosf(a);
  emb[(int64_t)n * dim + half + k] = sinf(a);
  if ((dim & 1) && k == 0) emb[(int64_t)n * dim + dim - 1] = 0.f;
}

extern "C" int gcd_timestep_embedding(const float* t, float* emb, int N, int dim, float max_period,
                                      void* stream) {
  GCD_CHECK_ARG(t && emb && N > 0 && dim >= 2, "gcd_timestep_embedding: bad arguments");
  const int total = N * (dim / 2);
  hipLaunchKernelGGL(timestep_embedding_kernel, dim3((total + 255) / 256), dim3(256), 0,
                     (hipStream_t)stream, t, emb, N, dim, max_period);
  GCD_CHECK_LAUNCH();
  return 0;
}
