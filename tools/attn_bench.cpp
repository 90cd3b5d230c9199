// tools/attn_bench.cpp — standalone A/B harness for the spatial-attention kernels of libgcd_amd.so
// (the shapes of one VideoUNet step at 14x72x128 latents).  Variants are selected with
// gcd_tune_set(GCD_TUNE_ATTN_IMPL, v); every variant is compared with variant 1 on the full output.
//
//   hipcc -O2 --offload-arch=gfx950 tools/attn_bench.cpp -Iinclude -Lgcd_amd -lgcd_amd \
//         -Wl,-rpath,'$ORIGIN/../gcd_amd' -o tools/attn_bench
//   tools/attn_bench [iters] [impl ...]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "gcd_amd.h"

#define CK(x)                                                                            \
  do {                                                                                   \
    hipError_t e_ = (x);                                                                 \
    if (e_ != hipSuccess) {                                                              \
      fprintf(stderr, "%s failed: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); \
      exit(1);                                                                           \
    }                                                                                    \
  } while (0)

typedef _Float16 f16;

__global__ void fill_f16(f16* p, size_t n, uint32_t seed, float scale) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    uint32_t x = (uint32_t)i * 2654435761u + seed;
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    p[i] = (f16)(((float)(x & 0xffff) / 32768.0f - 1.0f) * scale);
  }
}
__global__ void cmp_kernel(const f16* a, const f16* b, size_t n, unsigned* res) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  float md = 0.f, mb = 0.f;
  for (; i < n; i += stride) {
    const float x = (float)a[i], y = (float)b[i];
    float d = fabsf(x - y);
    if (!(d == d)) d = INFINITY;
    md = fmaxf(md, d);
    mb = fmaxf(mb, fabsf(y));
  }
  atomicMax(&res[0], __float_as_uint(md));
  atomicMax(&res[1], __float_as_uint(mb));
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 3;
  std::vector<int> impls = {1};
  for (int i = 2; i < argc; ++i) impls.push_back(atoi(argv[i]));
  struct Sh { int frames, S, heads, count; const char* what; };
  const Sh shapes[] = {{28, 9216, 5, 5, "L0 72x128"}, {28, 2304, 10, 5, "L1 36x64"},
                       {28, 576, 20, 5, "L2 18x32"}, {28, 144, 20, 1, "mid 9x16"}};
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  unsigned* res;
  CK(hipMalloc(&res, 8));
  std::vector<double> tot(impls.size(), 0.0);
  double tot_fl = 0;
  for (const Sh& s : shapes) {
    const int C = s.heads * 64, S_pad = (s.S + 63) / 64 * 64;
    const size_t M = (size_t)s.frames * s.S;
    f16 *qkv, *vt, *o_ref, *o;
    CK(hipMalloc(&qkv, M * 3 * C * 2));
    CK(hipMalloc(&vt, (size_t)s.frames * s.heads * 64 * S_pad * 2));
    CK(hipMalloc(&o_ref, M * C * 2));
    CK(hipMalloc(&o, M * C * 2));
    fill_f16<<<2048, 256, 0, st>>>(qkv, M * 3 * C, 11u, 2.0f);
    if (gcd_attn_transpose_v(qkv, 3 * C, s.frames, s.S, s.heads, vt, S_pad, st)) { fprintf(stderr, "%s\n", gcd_last_error()); return 1; }
    const double fl = 4.0 * s.frames * s.heads * (double)s.S * s.S * 64;
    printf("%-10s frames %d S %5d heads %2d:", s.what, s.frames, s.S, s.heads);
    for (size_t v = 0; v < impls.size(); ++v) {
      gcd_tune_set(GCD_TUNE_ATTN_IMPL, impls[v]);
      f16* dst = v == 0 ? o_ref : o;
      std::vector<float> t;
      for (int it = 0; it < iters + 1; ++it) {
        CK(hipEventRecord(e0, st));
        // timed: the product form (q pre-scaled by log2(e)/8 when W_q was packed); same work per score
        if (gcd_attn_spatial_f16(qkv, 3 * C, vt, S_pad, dst, C, s.frames, s.S, s.heads, 1, st)) { fprintf(stderr, "%s\n", gcd_last_error()); return 1; }
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (it >= 1) t.push_back(ms * 1e3f);
      }
      std::sort(t.begin(), t.end());
      const float us = t[t.size() / 2];
      float md = 0.f;
      if (v > 0) {
        CK(hipMemsetAsync(res, 0, 8, st));
        cmp_kernel<<<1024, 256, 0, st>>>(o, o_ref, M * C, res);
        unsigned hr[2];
        CK(hipMemcpyAsync(hr, res, 8, hipMemcpyDeviceToHost, st));
        CK(hipStreamSynchronize(st));
        memcpy(&md, &hr[0], 4);
      }
      printf("  [impl %d] %8.1f us %7.1f TF/s d=%.1e", impls[v], us, fl / us * 1e-6, md);
      tot[v] += us * 1e-3 * s.count;
    }
    printf("\n");
    tot_fl += fl * s.count;
    hipFree(qkv); hipFree(vt); hipFree(o_ref); hipFree(o);
  }
  for (size_t v = 0; v < impls.size(); ++v)
    printf("impl %d: %.2f ms per step (%.0f TF/s over %.2f TFLOP)\n", impls[v], tot[v], tot_fl / tot[v] * 1e-9, tot_fl * 1e-12);
  return 0;
}
