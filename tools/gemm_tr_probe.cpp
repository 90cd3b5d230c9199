// tools/gemm_tr_probe.cpp — a token-contracting GEMM that reads BOTH operands row-major and transposes them on the LDS
// read (ds_read_b64_tr_b16), as a standalone probe (no torch, no libgcd_amd):
//
//     dW[N, K] (fp32) = dY^T X,     dY [M, N] fp16 row-major,  X [M, K] fp16 row-major,  contraction over the M tokens
//
// This is the shape of every weight gradient of the fine-tune step (dW = dY^T X with M = 43 008 tokens at
// cfg4's shape), which libgcd_amd computes today as  transpose(dY), transpose(X)  (13 ms of transposes per step) + a split-K
// gcd_gemm_f16.  The MFMA wants 8 consecutive contraction indices per lane; in a row-major [m][n] tile those are 8 ROWS.
// ds_read_b64_tr_b16 (lane map: profiles/r04_probe_ds_read_tr_b16.txt) hands a 16-lane group the transpose of a 4-row x
// 16-column block — lane j gets column j of the four rows — so two of them per 16 x 16 block build the operand straight
// from the row-major tile.  The probe checks that construction against an fp64 host reference and times an UNTUNED kernel
// (128 x 128 output tile, 4 waves of 64 x 64, 32 tokens per step, register-staged double buffer, split over the tokens
// with fp32 partial outputs + a reduce pass) on the step's wgrad shapes; it is the starting point for the product kernel,
// not the product.
//
//   hipcc -O3 --offload-arch=gfx950 tools/gemm_tr_probe.cpp -o tools/gemm_tr_probe && tools/gemm_tr_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

typedef _Float16 f16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef short v4s __attribute__((__vector_size__(4 * sizeof(short))));
#define AS3 __attribute__((address_space(3)))

#define CK(x)                                                                              \
  do {                                                                                     \
    hipError_t e_ = (x);                                                                   \
    if (e_ != hipSuccess) {                                                                \
      fprintf(stderr, "%s failed: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__);   \
      exit(1);                                                                             \
    }                                                                                      \
  } while (0)

constexpr int TN = 128, TK = 128, TM = 32;     // output tile, tokens per step
constexpr int PITCH = TN * 2 + 16;             // bytes per LDS row: 128 fp16 + 16 B pad (TN == TK)
constexpr int TILE_BYTES = TM * PITCH;         // one operand tile of one step

// 8 consecutive tokens (rows 8 g .. 8 g + 7 of the step's tile, g = lane >> 4) of column c0 + (lane & 15): two
// transposing reads of 4 rows each.  Lane i of a 16-lane group supplies the address of row i / 4, columns 4 (i % 4) .. + 3
// of the 4 x 16 block; it receives column i.
__device__ __forceinline__ f16x8 frag_tr(const char* tile, int c0, int lane) {
  const int g = lane >> 4, i = lane & 15;
  const char* p = tile + (8 * g + (i >> 2)) * PITCH + (c0 + 4 * (i & 3)) * 2;
  const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((AS3 v4s*)(p));
  const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((AS3 v4s*)(p + 4 * PITCH));
  const f16x4 l4 = __builtin_bit_cast(f16x4, lo), h4 = __builtin_bit_cast(f16x4, hi);
  return (f16x8){l4[0], l4[1], l4[2], l4[3], h4[0], h4[1], h4[2], h4[3]};
}

// grid (ceil(K / TK), ceil(N / TN), S): block (kt, nt, s) accumulates tokens [s * mper, (s + 1) * mper) into
// part[s][N][K] (plain stores; reduce_kernel folds the S slices)
__global__ __launch_bounds__(256) void wgrad_tr_kernel(const f16* __restrict__ dY, const f16* __restrict__ X,
                                                       float* __restrict__ part, int M, int N, int K, int mper) {
  __shared__ __attribute__((aligned(16))) char smem[2 * 2 * TILE_BYTES];   // [buffer][operand]
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wn = wave >> 1, wk = wave & 1;
  const int n0 = blockIdx.y * TN, k0 = blockIdx.x * TK;
  const int m_begin = blockIdx.z * mper;
  int m_end = m_begin + mper;
  if (m_end > M) m_end = M;
  const int nsteps = (m_end - m_begin + TM - 1) / TM;
  // staging map: thread -> row t / 16 (+ 16), 16-byte chunk t % 16 of the 256-byte tile row
  const int srow = t >> 4, schunk = t & 15;
  f16x8 ra[2], rb[2];
  auto gload = [&](int step) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int m = m_begin + step * TM + srow + 16 * h;
      const int n = n0 + 8 * schunk, k = k0 + 8 * schunk;
      const f16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
      ra[h] = (m < m_end && n < N) ? *(const f16x8*)(dY + (int64_t)m * N + n) : z;    // (N, K multiples of 8)
      rb[h] = (m < m_end && k < K) ? *(const f16x8*)(X + (int64_t)m * K + k) : z;
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
      for (int j = 0; j < 4; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    if (s + 1 < nsteps) lstore((s + 1) & 1);          // the other buffer: its last readers passed the barrier below
    __syncthreads();
  }
  // C[n = 4 (lane >> 4) + e][k = lane & 15] per 16 x 16 block
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

__global__ void reduce_kernel(const float* __restrict__ part, float* __restrict__ dW, int64_t NK, int S) {
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= NK) return;
  f32x4 a = {0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < S; ++s) a += *(const f32x4*)(part + (int64_t)s * NK + i);
  *(f32x4*)(dW + i) = a;
}

static float frand(uint32_t& st) {
  st = st * 1664525u + 1013904223u;
  return ((st >> 8) & 0xffff) / 65536.0f - 0.5f;
}

static int run(int M, int N, int K, int S, int iters, bool check) {
  std::vector<f16> hY((size_t)M * N), hX((size_t)M * K);
  uint32_t st = 12345u + M + 7 * N + 13 * K;
  for (auto& v : hY) v = (f16)frand(st);
  for (auto& v : hX) v = (f16)frand(st);
  f16 *dY, *dX;
  float *part, *dW;
  CK(hipMalloc(&dY, hY.size() * 2));
  CK(hipMalloc(&dX, hX.size() * 2));
  CK(hipMalloc(&part, (size_t)S * N * K * 4));
  CK(hipMalloc(&dW, (size_t)N * K * 4));
  CK(hipMemcpy(dY, hY.data(), hY.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dX, hX.data(), hX.size() * 2, hipMemcpyHostToDevice));
  const int mper = ((M + S - 1) / S + TM - 1) / TM * TM;
  const dim3 grid((K + TK - 1) / TK, (N + TN - 1) / TN, S);
  auto launch = [&]() {
    hipLaunchKernelGGL(wgrad_tr_kernel, grid, dim3(256), 0, 0, dY, dX, part, M, N, K, mper);
    hipLaunchKernelGGL(reduce_kernel, dim3((unsigned)(((int64_t)N * K / 4 + 255) / 256)), dim3(256), 0, 0, part, dW,
                       (int64_t)N * K, S);
  };
  launch();
  CK(hipDeviceSynchronize());
  int bad = 0;
  double worst = 0.0;
  if (check) {
    std::vector<float> hW((size_t)N * K);
    CK(hipMemcpy(hW.data(), dW, hW.size() * 4, hipMemcpyDeviceToHost));
    uint32_t ps = 99u;
    const int nsamp = 4000;
    for (int q = 0; q < nsamp; ++q) {
      ps = ps * 1664525u + 1013904223u;
      const int n = (ps >> 8) % N;
      ps = ps * 1664525u + 1013904223u;
      const int k = (ps >> 8) % K;
      double ref = 0.0;
      for (int m = 0; m < M; ++m) ref += (double)hY[(size_t)m * N + n] * (double)hX[(size_t)m * K + k];
      const double err = fabs(ref - hW[(size_t)n * K + k]);
      const double tol = 2e-3 * sqrt((double)M) * 0.083 + 1e-4;   // fp32 accumulation of M products of variance 1/144
      if (err > worst) worst = err;
      if (err > tol) ++bad;
    }
  }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) launch();
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / iters, tf = 2.0 * M * N * K / us * 1e-6;
  printf("M %6d  N %5d  K %5d  S %3d | %8.1f us  %7.1f TF/s | %s (max |err| %.3e over 4000 samples)\n", M, N, K, S, us, tf,
         check ? (bad ? "MISMATCH" : "ok") : "-", worst);
  hipFree(dY);
  hipFree(dX);
  hipFree(part);
  hipFree(dW);
  return bad;
}

int main() {
  int bad = 0;
  // correctness: ragged N and K (not multiples of the tile), a token count that S does not divide evenly
  bad += run(2080, 200, 328, 3, 1, true);
  bad += run(4096, 320, 640, 8, 1, true);
  // the fine-tune step's wgrad shapes at cfg4 (M = frames x pixels of a level)
  const int shapes[][3] = {{43008, 320, 320}, {43008, 320, 1280}, {43008, 2560, 320}, {43008, 320, 2880},
                           {10752, 640, 640}, {10752, 640, 5760}, {2688, 1280, 1280}, {2688, 1280, 11520}};
  for (auto& s : shapes) {
    const int tiles = ((s[1] + TN - 1) / TN) * ((s[2] + TK - 1) / TK);
    int S = 1024 / tiles;
    if (S < 1) S = 1;
    if (S > 64) S = 64;
    bad += run(s[0], s[1], s[2], S, 5, false);
  }
  printf(bad ? "RESULT: MISMATCH\n" : "RESULT: the transposing-read operand construction is correct\n");
  return bad ? 1 : 0;
}
