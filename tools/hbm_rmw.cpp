// tools/hbm_rmw.cpp — what HBM3E gives the epilogue-shaped access mixes of the fp32 residual stream
// on one MI355X: read-only, write-only, copy, in-place read-modify-write, and RMW + an fp16 read
// (the traffic of a residual GEMM epilogue), float4 per lane, UNROLL independent loads in flight.
//   hipcc -O2 --offload-arch=gfx950 tools/hbm_rmw.cpp -o tools/hbm_rmw && tools/hbm_rmw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int U>
__global__ __launch_bounds__(256) void k(const f32x4* __restrict__ src, f32x4* dst, size_t n, float a) {
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  f32x4 acc = {0, 0, 0, 0};
  for (; i + (U - 1) * stride < n; i += U * stride) {
    f32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (MODE == 0 || MODE == 2) v[u] = src[i + u * stride];          // read / copy source
      if (MODE == 3) v[u] = dst[i + u * stride];                       // in place
      if (MODE == 1) v[u] = f32x4{a, a, a, a};
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (MODE == 0) acc += v[u];
      else dst[i + u * stride] = v[u] * a + a;
    }
  }
  if (MODE == 0 && acc[0] == 123.456f) dst[0] = acc;
}

template <int MODE, int U>
static void run(const char* name, const f32x4* s, f32x4* d, size_t n, double bytes_per_elem, int blocks) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<float> t;
  for (int it = 0; it < 6; ++it) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<MODE, U>), dim3(blocks), dim3(256), 0, 0, s, d, n, 1.0001f);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (it) t.push_back(ms);
  }
  std::sort(t.begin(), t.end());
  const double us = t[t.size() / 2] * 1e3;
  printf("  %-28s U=%d blocks=%5d  %8.1f us  %6.2f TB/s\n", name, U, blocks, us, n * 16.0 * bytes_per_elem / us * 1e-6);
}

int main() {
  for (size_t mb : {330, 1320}) {
    const size_t n = mb * 1000000 / 16;
    f32x4 *a, *b;
    CK(hipMalloc(&a, n * 16)); CK(hipMalloc(&b, n * 16));
    CK(hipMemset(a, 0, n * 16)); CK(hipMemset(b, 0, n * 16));
    printf("%zu MB per stream:\n", mb);
    for (int blocks : {2048, 8192}) {
      run<0, 4>("read only", a, b, n, 1, blocks);
      run<1, 4>("write only", a, b, n, 1, blocks);
      run<2, 4>("copy (read + write)", a, b, n, 2, blocks);
      run<3, 4>("in-place RMW", a, b, n, 2, blocks);
      run<3, 8>("in-place RMW", a, b, n, 2, blocks);
      run<3, 2>("in-place RMW", a, b, n, 2, blocks);
    }
    CK(hipFree(a)); CK(hipFree(b));
  }
  return 0;
}
