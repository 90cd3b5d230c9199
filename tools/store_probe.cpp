// tools/store_probe.cpp — what does the SHAPE of a tile's output stores cost?  (round-2 experiment)
//
// The GEGLU epilogue of gemm_pp writes, per 256 x 320 tile, 256 rows x 320 B of fp16 (160 hidden
// columns) into a [258048, 1280] fp16 matrix: row segments that are 32-byte but not 128-byte aligned,
// whose neighbours in the same 128-B lines are written by OTHER workgroups at other times.  This probe
// writes the same 660 MB with 256 persistent 8-wave workgroups that walk tiles in the GEMM's order
// (4 M-tiles x all N-tiles per group) and differ only in the segment width SEG (bytes per row per tile):
//   320 (the product), 256, 512, 640, 2560 (whole rows).  Each wave stores 16 B per lane, row-contiguous.
//   hipcc -O2 --offload-arch=gfx950 tools/store_probe.cpp -o tools/store_probe && tools/store_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

// rows of 2560 B; a tile = 256 rows x SEG bytes; waves split the tile's rows (32 each).
template <int SEG, bool XCD>
__global__ __launch_bounds__(512) void wr(char* out, int tiles_m, float v) {
  constexpr int TN = 2560 / SEG;               // N-tiles per row
  constexpr int CH = SEG / 16;                 // 16-B chunks per row segment
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int ntiles = tiles_m * TN;
  const f32x4 val = {v, v, v, v};
  // XCD: the product's mapping — workgroup b runs on XCD b % 8, each XCD owns a contiguous range of the
  // tile order (so the workgroups that write the two halves of a 128-B line share one L2)
  int L0 = blockIdx.x, L1 = ntiles, Ls = gridDim.x;
  if (XCD) {
    const int q = ntiles >> 3, r = ntiles & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    L0 = start + idx;
    L1 = start + q + (xcd < r ? 1 : 0);
    Ls = gridDim.x >> 3;
  }
  for (int L = L0; L < L1; L += Ls) {
    const int per_group = 4 * TN;
    const int gi = L / per_group, rem = L - gi * per_group;
    const int gm = min(4, tiles_m - gi * 4);
    const int tn = rem / gm, tm = gi * 4 + rem - tn * gm;
    char* base = out + (size_t)(tm * 256 + wave * 32) * 2560 + (size_t)tn * SEG;
    for (int idx = lane; idx < 32 * CH; idx += 64) {
      const int row = idx / CH, ch = idx - row * CH;
      *(f32x4*)(base + (size_t)row * 2560 + ch * 16) = val;
    }
  }
}

template <int SEG, bool XCD>
static void run(char* x, int tiles_m) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<float> t;
  for (int it = 0; it < 6; ++it) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((wr<SEG, XCD>), dim3(256), dim3(512), 0, 0, x, tiles_m, 1.0f + it);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (it) t.push_back(ms);
  }
  std::sort(t.begin(), t.end());
  const double bytes = (double)tiles_m * 256 * 2560;
  printf("%s segment %4d B per row per tile: %7.1f us  %6.2f TB/s\n", XCD ? "xcd-aware " : "round-robin", SEG, t[t.size() / 2] * 1e3, bytes / (t[t.size() / 2] * 1e-3) / 1e12);
}

int main() {
  const int tiles_m = 1008;   // 258048 rows
  char* x;
  CK(hipMalloc(&x, (size_t)tiles_m * 256 * 2560));
  CK(hipMemset(x, 0, (size_t)tiles_m * 256 * 2560));
  run<320, false>(x, tiles_m);
  run<256, false>(x, tiles_m);
  run<640, false>(x, tiles_m);
  run<2560, false>(x, tiles_m);
  run<320, true>(x, tiles_m);
  run<256, true>(x, tiles_m);
  run<640, true>(x, tiles_m);
  run<2560, true>(x, tiles_m);
  run<320, true>(x, tiles_m);
  return 0;
}
