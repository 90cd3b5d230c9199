// tools/hbm_epi.cpp — the fp32 residual read-modify-write of a GEMM epilogue WITHOUT the GEMM: 256
// persistent workgroups of 8 waves walk 256 x 320 tiles of an fp32 [M, 320] matrix with exactly the
// accumulator-layout access pattern of gcd_epi_f32_full (lane -> row l31 + 32 j, 16 B at column
// 32 i + 8 g + 4 hh), D batches of 4 vectors in flight per wave.  Tells whether the pattern and the
// per-CU concurrency — not the K loop — bound the residual GEMMs.
//   hipcc -O2 --offload-arch=gfx950 tools/hbm_epi.cpp -o tools/hbm_epi && tools/hbm_epi
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

// PAT 0: accumulator layout (32 rows x 32 B per instruction); PAT 1: row-contiguous (8 rows x 128 B)
template <int D, int PAT, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void epi(float* x, int M, int tiles, float a) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    // WAVES == 8: 4 x 2 waves of 64 x 160; WAVES == 4: 4 x 1 waves of 64 x 320 done as two halves
    for (int half = 0; half < (WAVES == 8 ? 1 : 2); ++half) {
      const int wm = wave & 3, wn = WAVES == 8 ? wave >> 2 : half;
      const int m_base = tile * 256 + 64 * wm, n_base = 160 * wn;
      f32x4 q[D][4];
      auto addr = [&](int b, int g) -> f32x4* {
        const int j = b / 5, i = b - 5 * j;
        if (PAT == 0) return (f32x4*)(x + (size_t)(m_base + 32 * j + l31) * 320 + n_base + 32 * i + 8 * g + 4 * hh);
        // 8 rows x 128 B per instruction: lane -> row 8 g + (lane >> 3), 16 B chunk lane & 7
        return (f32x4*)(x + (size_t)(m_base + 32 * j + 8 * g + (lane >> 3)) * 320 + n_base + 32 * i + 4 * (lane & 7));
      };
#pragma unroll
      for (int b = 0; b < D; ++b)
#pragma unroll
        for (int g = 0; g < 4; ++g) q[b][g] = *addr(b, g);
#pragma unroll
      for (int b = 0; b < 10; ++b) {
#pragma unroll
        for (int g = 0; g < 4; ++g) *addr(b, g) = q[b % D][g] * a + a;
        if (b + D < 10) {
#pragma unroll
          for (int g = 0; g < 4; ++g) q[b % D][g] = *addr(b + D, g);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
}

template <int D, int PAT, int WAVES>
static void run(float* x, int M, int blocks) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<float> t;
  for (int it = 0; it < 6; ++it) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((epi<D, PAT, WAVES>), dim3(blocks), dim3(WAVES * 64), 0, 0, x, M, M / 256, 1.0001f);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (it) t.push_back(ms);
  }
  std::sort(t.begin(), t.end());
  const double us = t[t.size() / 2] * 1e3;
  printf("  D=%d pattern=%s waves/WG=%d WGs=%4d : %7.1f us  %5.2f TB/s (read + write)\n", D,
         PAT ? "8x128B" : "32x32B", WAVES, blocks, us, (double)M * 320 * 8 / us * 1e-6);
}

int main() {
  const int M = 258048;
  float* x;
  CK(hipMalloc(&x, (size_t)M * 320 * 4));
  CK(hipMemset(x, 0, (size_t)M * 320 * 4));
  printf("in-place RMW of fp32 [%d, 320] (330 MB), epilogue-shaped:\n", M);
  run<1, 0, 8>(x, M, 256);
  run<2, 0, 8>(x, M, 256);
  run<3, 0, 8>(x, M, 256);
  run<5, 0, 8>(x, M, 256);
  run<10, 0, 8>(x, M, 256);
  run<2, 1, 8>(x, M, 256);
  run<5, 1, 8>(x, M, 256);
  run<10, 1, 8>(x, M, 256);
  run<2, 0, 8>(x, M, 512);
  run<5, 0, 8>(x, M, 512);
  run<2, 0, 8>(x, M, 1008);
  run<5, 0, 8>(x, M, 1008);
  run<10, 0, 8>(x, M, 1008);
  run<2, 0, 4>(x, M, 1008);
  run<5, 0, 4>(x, M, 1008);
  return 0;
}
