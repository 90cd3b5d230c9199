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
// (128 x 128 output tile, 4 waves of 64 x 64, 32 or 64 tokens per step, register-staged double buffer, split over the
// tokens with fp32 partial outputs + a reduce pass) on the step's wgrad shapes.  Since the end of round 4 the kernels are
// the product's own (gcd_amd/csrc/train_wgrad_kernel.h, launched by gcd_wgrad_tr_f16): this file compiles that header.
//
//   hipcc -O3 --offload-arch=gfx950 tools/gemm_tr_probe.cpp -o tools/gemm_tr_probe && tools/gemm_tr_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../gcd_amd/csrc/train_wgrad_kernel.h"   // the kernels libgcd_amd_train.so launches, compiled here as they are

#define CK(x)                                                                              \
  do {                                                                                     \
    hipError_t e_ = (x);                                                                   \
    if (e_ != hipSuccess) {                                                                \
      fprintf(stderr, "%s failed: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__);   \
      exit(1);                                                                             \
    }                                                                                      \
  } while (0)

static float frand(uint32_t& st) {
  st = st * 1664525u + 1013904223u;
  return ((st >> 8) & 0xffff) / 65536.0f - 0.5f;
}
static uint16_t to_bits(float v, bool bf16) {
  if (bf16) {
    uint32_t u;
    memcpy(&u, &v, 4);
    return (uint16_t)(u >> 16);          // truncation: the reference below uses the truncated value
  }
  const _Float16 h = (_Float16)v;
  uint16_t b;
  memcpy(&b, &h, 2);
  return b;
}
static double from_bits(uint16_t b, bool bf16) {
  if (bf16) {
    const uint32_t u = (uint32_t)b << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
  }
  _Float16 h;
  memcpy(&h, &b, 2);
  return (double)h;
}

template <bool BF16, int TM>
static int run(int M, int N, int K, int iters, bool check) {
  std::vector<uint16_t> hY((size_t)M * N), hX((size_t)M * K);
  uint32_t st = 12345u + M + 7 * N + 13 * K;
  for (auto& v : hY) v = to_bits(frand(st), BF16);
  for (auto& v : hX) v = to_bits(frand(st), BF16);
  const int S = gcd_wgrad::slices(M, N, K);
  void *dY, *dX;
  float *part, *dW;
  CK(hipMalloc(&dY, hY.size() * 2));
  CK(hipMalloc(&dX, hX.size() * 2));
  CK(hipMalloc(&part, (size_t)S * N * K * 4));
  CK(hipMalloc(&dW, (size_t)N * K * 4));
  CK(hipMemcpy(dY, hY.data(), hY.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dX, hX.data(), hX.size() * 2, hipMemcpyHostToDevice));
  auto launch = [&]() { CK((gcd_wgrad::launch<BF16, TM>(dY, N, dX, K, M, N, K, dW, K, gcd_wgrad::Layout{1, N, K, 0}, part, 0))); };
  launch();
  CK(hipDeviceSynchronize());
  int bad = 0;
  double worst = 0.0;
  if (check) {
    std::vector<float> hW((size_t)N * K);
    CK(hipMemcpy(hW.data(), dW, hW.size() * 4, hipMemcpyDeviceToHost));
    uint32_t ps = 99u;
    for (int q = 0; q < 3000; ++q) {
      ps = ps * 1664525u + 1013904223u;
      const int n = (ps >> 8) % N;
      ps = ps * 1664525u + 1013904223u;
      const int k = (ps >> 8) % K;
      double ref = 0.0;
      for (int m = 0; m < M; ++m) ref += from_bits(hY[(size_t)m * N + n], BF16) * from_bits(hX[(size_t)m * K + k], BF16);
      const double err = fabs(ref - hW[(size_t)n * K + k]);
      if (err > worst) worst = err;
      if (err > 1e-3) ++bad;      // exact products, fp32 accumulation: ~1e-5; a wrong lane map is off by O(1)
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
  printf("%s TM %2d  M %6d  N %5d  K %5d  S %3d | %8.1f us  %7.1f TF/s | %s (max |err| %.3e)\n", BF16 ? "bf16" : "fp16", TM, M, N,
         K, S, us, tf, check ? (bad ? "MISMATCH" : "ok") : "-", worst);
  hipFree(dY);
  hipFree(dX);
  hipFree(part);
  hipFree(dW);
  return bad;
}

int main() {
  int bad = 0;
  // correctness: ragged N and K (not multiples of the tile), token counts the slices / steps do not divide
  bad += run<false, 32>(2080, 200, 328, 1, true);
  bad += run<false, 64>(2080, 200, 328, 1, true);
  bad += run<true, 32>(4100, 320, 640, 1, true);
  bad += run<true, 64>(4100, 320, 640, 1, true);
  // (round 6) the 160 x 160 tile (gcd_wgrad::tile_of: N and K multiples of 160, one of them not of 128; GCD_WGRAD_TILE=128
  // forces the old tile for the A/B): ragged token counts, both step sizes, 4 x 6 and 2 x 2 tiles
  bad += run<false, 32>(4100, 640, 960, 1, true);
  bad += run<false, 64>(3001, 320, 320, 1, true);
  bad += run<true, 32>(2999, 960, 320, 1, true);
  // the fine-tune step's wgrad shapes at cfg4 (M = frames x pixels of a level), 32 vs 64 tokens per step
  const int shapes[][3] = {{43008, 320, 320}, {43008, 320, 1280}, {43008, 2560, 320}, {43008, 320, 2880},
                           {10752, 640, 640}, {10752, 640, 5760}, {2688, 1280, 1280}, {2688, 1280, 11520}};
  for (auto& s : shapes) {
    bad += run<false, 32>(s[0], s[1], s[2], 5, false);
    bad += run<false, 64>(s[0], s[1], s[2], 5, false);
  }
  printf(bad ? "RESULT: MISMATCH\n" : "RESULT: the transposing-read operand construction is correct\n");
  return bad ? 1 : 0;
}
