// train_wgrad.hip — libgcd_amd_train.so: kernels that only the fine-tune step uses (BASELINE.json cfg4), built beside
// libgcd_amd.so from the same tree.  A library of its own so that the sampler's library — whose source digest stamps the
// PMC traffic profile `bench.py` quotes — does not change when a training kernel does.  C ABI: include/gcd_amd_train.h.
//
// gcd_wgrad_tr_f16: the weight gradient of a Linear (loss.py / Lightning's backward run it as torch.autograd of
// nn.Linear, attention.py:87-113, 272-303):
//
//     dW[N, K] (fp32) = dY^T X,     dY [M, N], X [M, K] 16-bit, ROW-MAJOR as the forward pass left them,  M = tokens
//
// The MFMA wants 8 consecutive contraction indices per lane — 8 ROWS of a row-major [token][channel] tile.  gfx950's
// ds_read_b64_tr_b16 hands a 16-lane group the transpose of a 4-row x 16-column block of 16-bit elements (lane i supplies
// row i / 4, columns 4 (i % 4) .. + 3, at its own address; lane j receives column j of the four rows;
// profiles/r04_probe_ds_read_tr_b16.txt), so two of them per 16 x 16 operand block build both MFMA operands straight from
// row-major LDS tiles: no transposed copies of dY and X in HBM (what autograd_ops._grad_contractions makes for
// gcd_gemm_f16: 13 ms of transposes per step at cfg4's shape).  Validated and timed stand-alone first
// (tools/gemm_tr_probe.cpp, profiles/r04v_gemm_tr_probe.txt: 185-600 TF/s on the step's shapes, untuned).
//
// Kernel: 128 (n) x 128 (k) output tile per workgroup, 4 waves of 64 x 64 (16 accumulators of v_mfma_f32_16x16x32),
// 32 tokens per step, operands through registers into a double-buffered LDS tile (272-byte rows), the token axis split
// over S workgroups whose fp32 partial outputs a second launch folds (no atomics).  Untuned: no LDS-DMA, no swizzle,
// one barrier per 16 MFMAs per wave.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/gcd_amd_train.h"

typedef _Float16 f16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef short v4s __attribute__((__vector_size__(4 * sizeof(short))));
#define GCD_AS3 __attribute__((address_space(3)))

static thread_local char g_err[512] = "";
static void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* gcd_train_last_error(void) { return g_err; }
extern "C" int gcd_train_abi_version(void) { return GCD_AMD_TRAIN_ABI_VERSION; }

#define CHECK_ARG(cond, ...)  \
  do {                        \
    if (!(cond)) {            \
      set_error(__VA_ARGS__); \
      return 2;               \
    }                         \
  } while (0)

namespace {

constexpr int TN = 128, TK = 128, TM = 32;     // output tile, tokens per step
constexpr int PITCH = TN * 2 + 16;             // bytes per LDS row: 128 16-bit elements + 16 B pad (TN == TK)
constexpr int TILE_BYTES = TM * PITCH;

// 8 consecutive tokens (rows 8 g .. 8 g + 7 of the step's tile, g = lane >> 4) of column c0 + (lane & 15)
__device__ __forceinline__ f16x8 frag_tr(const char* tile, int c0, int lane) {
  const int g = lane >> 4, i = lane & 15;
  const char* p = tile + (8 * g + (i >> 2)) * PITCH + (c0 + 4 * (i & 3)) * 2;
  const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((GCD_AS3 v4s*)(p));
  const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((GCD_AS3 v4s*)(p + 4 * PITCH));
  const f16x4 l4 = __builtin_bit_cast(f16x4, lo), h4 = __builtin_bit_cast(f16x4, hi);
  return (f16x8){l4[0], l4[1], l4[2], l4[3], h4[0], h4[1], h4[2], h4[3]};
}

// grid (ceil(K / TK), ceil(N / TN), S): workgroup (kt, nt, s) accumulates tokens [s * mper, (s + 1) * mper) into
// part[s][N][K].  The 16-bit payload travels as f16x8 bit patterns; BF16 only selects the MFMA.
template <bool BF16>
__global__ __launch_bounds__(256) void wgrad_tr_kernel(const f16* __restrict__ dY, int64_t lddy,
                                                       const f16* __restrict__ X, int64_t ldx,
                                                       float* __restrict__ part, int64_t M, int N, int K, int64_t mper) {
  __shared__ __attribute__((aligned(16))) char smem[2 * 2 * TILE_BYTES];   // [buffer][operand]
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wn = wave >> 1, wk = wave & 1;
  const int n0 = blockIdx.y * TN, k0 = blockIdx.x * TK;
  const int64_t m_begin = (int64_t)blockIdx.z * mper;
  int64_t m_end = m_begin + mper;
  if (m_end > M) m_end = M;
  const int nsteps = m_end > m_begin ? (int)((m_end - m_begin + TM - 1) / TM) : 0;
  const int srow = t >> 4, schunk = t & 15;      // staging: row t / 16 (+ 16), 16-byte chunk t % 16 of a tile row
  f16x8 ra[2], rb[2];
  auto gload = [&](int step) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int64_t m = m_begin + (int64_t)step * TM + srow + 16 * h;
      const int n = n0 + 8 * schunk, k = k0 + 8 * schunk;
      const f16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
      ra[h] = (m < m_end && n < N) ? *(const f16x8*)(dY + m * lddy + n) : z;
      rb[h] = (m < m_end && k < K) ? *(const f16x8*)(X + m * ldx + k) : z;
    }
  };
  auto lstore = [&](int buf) {
    char* a = smem + buf * 2 * TILE_BYTES;
    char* b = a + TILE_BYTES;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      *(f16x8*)(a + (srow + 16 * h) * PITCH + schunk * 16) = ra[h];
      *(f16x8*)(b + (srow + 16 * h) * PITCH + schunk * 16) = rb[h];
    }
  };
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (nsteps > 0) {
    gload(0);
    lstore(0);
  }
  __syncthreads();
  for (int s = 0; s < nsteps; ++s) {
    if (s + 1 < nsteps) gload(s + 1);                 // in flight under this step's MFMAs
    const char* a = smem + (s & 1) * 2 * TILE_BYTES;
    const char* b = a + TILE_BYTES;
    f16x8 fa[4], fb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      fa[i] = frag_tr(a, 64 * wn + 16 * i, lane);
      fb[i] = frag_tr(b, 64 * wk + 16 * i, lane);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if constexpr (BF16)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[i]),
                                                              __builtin_bit_cast(bf16x8, fb[j]), acc[i][j], 0, 0, 0);
        else
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i], fb[j], acc[i][j], 0, 0, 0);
      }
    if (s + 1 < nsteps) lstore((s + 1) & 1);          // the other buffer: its last readers passed the barrier below
    __syncthreads();
  }
  // accumulator block (i, j): C[n = 4 (lane >> 4) + e][k = lane & 15]
  float* out = part + (int64_t)blockIdx.z * N * K;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int n = n0 + 64 * wn + 16 * i + 4 * (lane >> 4) + e;
        const int k = k0 + 64 * wk + 16 * j + (lane & 15);
        if (n < N && k < K) out[(int64_t)n * K + k] = acc[i][j][e];
      }
}

// dW[n][k] = sum over the S slices; K % 4 == 0
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dW,
                                                           int64_t lddw, int N, int K, int S) {
  const int64_t idx = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  const int64_t NK = (int64_t)N * K;
  if (idx >= NK) return;
  f32x4 a = {0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < S; ++s) a += *(const f32x4*)(part + (int64_t)s * NK + idx);
  const int64_t n = idx / K;
  *(f32x4*)(dW + n * lddw + (idx - n * K)) = a;
}

// token slices: ~1024 workgroups over the launch, at least 256 tokens per slice, at most 64 slices
int wgrad_slices(int64_t M, int N, int K) {
  const int64_t tiles = (int64_t)((N + TN - 1) / TN) * ((K + TK - 1) / TK);
  int64_t S = 1024 / (tiles > 0 ? tiles : 1);
  if (S > 64) S = 64;
  if (S > M / 256) S = M / 256;
  if (S < 1) S = 1;
  return (int)S;
}

}  // namespace

extern "C" int64_t gcd_wgrad_tr_scratch_floats(int64_t M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  return (int64_t)wgrad_slices(M, N, K) * N * K;
}

extern "C" int gcd_wgrad_tr_f16(const void* dy16, int64_t lddy, const void* x16, int64_t ldx, int64_t M, int N, int K,
                                int bf16, float* dW, int64_t lddw, float* scratch, int64_t scratch_floats,
                                void* stream) {
  CHECK_ARG(dy16 && x16 && dW && scratch, "gcd_wgrad_tr_f16: null pointer");
  CHECK_ARG(M > 0 && N > 0 && K > 0 && N % 8 == 0 && K % 8 == 0,
            "gcd_wgrad_tr_f16: M=%lld N=%d K=%d (N and K must be multiples of 8)", (long long)M, N, K);
  CHECK_ARG(lddy % 8 == 0 && lddy >= N && ldx % 8 == 0 && ldx >= K && lddw % 4 == 0 && lddw >= K,
            "gcd_wgrad_tr_f16: leading dimensions lddy=%lld ldx=%lld lddw=%lld", (long long)lddy, (long long)ldx,
            (long long)lddw);
  CHECK_ARG((((uintptr_t)dy16 | (uintptr_t)x16 | (uintptr_t)dW | (uintptr_t)scratch) & 15) == 0,
            "gcd_wgrad_tr_f16: operands must be 16-byte aligned");
  const int S = wgrad_slices(M, N, K);
  CHECK_ARG(scratch_floats >= (int64_t)S * N * K, "gcd_wgrad_tr_f16: scratch of %lld floats, need %lld "
            "(gcd_wgrad_tr_scratch_floats)", (long long)scratch_floats, (long long)S * N * K);
  const int64_t mper = ((M + S - 1) / S + TM - 1) / TM * TM;
  const dim3 grid((K + TK - 1) / TK, (N + TN - 1) / TN, S);
  hipStream_t s = (hipStream_t)stream;
  if (bf16)
    hipLaunchKernelGGL(wgrad_tr_kernel<true>, grid, dim3(256), 0, s, (const f16*)dy16, lddy, (const f16*)x16, ldx,
                       scratch, M, N, K, mper);
  else
    hipLaunchKernelGGL(wgrad_tr_kernel<false>, grid, dim3(256), 0, s, (const f16*)dy16, lddy, (const f16*)x16, ldx,
                       scratch, M, N, K, mper);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) {
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)(((int64_t)N * K / 4 + 255) / 256)), dim3(256), 0, s, scratch,
                       dW, lddw, N, K, S);
    e = hipGetLastError();
  }
  if (e != hipSuccess) {
    set_error("gcd_wgrad_tr_f16: launch failed: %s", hipGetErrorString(e));
    return 1;
  }
  return 0;
}
