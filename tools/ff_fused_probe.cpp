// tools/ff_fused_probe.cpp — the fused FeedForward kernel (gcd_amd/csrc/ff_fused_kernel.h) against the two-GEMM path of
// libgcd_amd.so (GEGLU GEMM -> 660 MB fp16 hidden tensor -> FF-out GEMM with the fp32 residual) on the L0 shape of one
// VideoUNet step at 14 x 72 x 128 latents: M = 258 048 tokens, C = 320, hidden 1280.  Same operands for both; full-output
// comparison, an fp64 host check of sampled entries, HIP-event timing (median).
//
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 tools/ff_fused_probe.cpp -Iinclude -Igcd_amd/csrc -Lgcd_amd -lgcd_amd \
//         -Wl,-rpath,'$ORIGIN/../gcd_amd' -o tools/ff_fused_probe
//   tools/ff_fused_probe [M] [iters]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#define FF_KERNEL ff_fused_kernel_probe      // not the library's instantiations (same template, same arguments)
#include "ff_fused_kernel.h"

#define CK(x)                                                                            \
  do {                                                                                   \
    hipError_t e_ = (x);                                                                 \
    if (e_ != hipSuccess) {                                                              \
      fprintf(stderr, "%s failed: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); \
      exit(1);                                                                           \
    }                                                                                    \
  } while (0)

void gcd_set_error(const char*, ...) {}

__global__ void fill_f16(f16* p, size_t n, uint32_t seed, float scale) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    uint32_t x = (uint32_t)i * 2654435761u + seed;
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    p[i] = (f16)(((float)(x & 0xffff) / 32768.0f - 1.0f) * scale);
  }
}
__global__ void fill_f32(float* p, size_t n, uint32_t seed, float scale) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    uint32_t x = (uint32_t)i * 2246822519u + seed;
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    p[i] = ((float)(x & 0xffff) / 32768.0f - 1.0f) * scale;
  }
}
// res[0] = max |a-b|, res[1] = max |b|, res[2..3] = sum (a-b)^2, sum b^2 (double bits via atomicAdd on doubles)
__global__ void cmp_kernel(const float* a, const float* b, size_t n, unsigned* res, double* sums) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  float md = 0.f, mb = 0.f;
  double sd = 0, sb = 0;
  for (; i < n; i += stride) {
    const float x = a[i], y = b[i];
    float d = fabsf(x - y);
    if (!(d == d)) d = INFINITY;
    md = fmaxf(md, d);
    mb = fmaxf(mb, fabsf(y));
    sd += (double)d * d;
    sb += (double)y * y;
  }
  atomicMax(&res[0], __float_as_uint(md));
  atomicMax(&res[1], __float_as_uint(mb));
  atomicAdd(&sums[0], sd);
  atomicAdd(&sums[1], sb);
}

static int g_grid = 256;
template <int T, int DEG>
static float run_fused(const FfK& k, hipStream_t st, int iters, hipEvent_t e0, hipEvent_t e1) {
  static bool opted = false;
  if (!opted) {
    CK(hipFuncSetAttribute((const void*)ff_fused_kernel_probe<T, DEG, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, FF_SMEM));
    opted = true;
  }
  std::vector<float> t;
  for (int it = 0; it < iters + 2; ++it) {
    CK(hipEventRecord(e0, st));
    ff_fused_kernel_probe<T, DEG, 0><<<g_grid, 256, FF_SMEM, st>>>(k);
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (it >= 2) t.push_back(ms * 1e3f);
  }
  std::sort(t.begin(), t.end());
  return t[t.size() / 2];
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 28 * 72 * 128;
  const int iters = argc > 2 ? atoi(argv[2]) : 9;
  g_grid = argc > 3 ? atoi(argv[3]) : 256;      // workgroups of the fused kernel (fewer: is a tile boundary slower when every CU is at one?)
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  f16 *X, *w1, *w2, *Wp, *hid, *zero;
  float *b1, *b2, *R, *o_ref, *o_f;
  unsigned* res;
  double* sums;
  CK(hipMalloc(&X, (size_t)M * 320 * 2));
  CK(hipMalloc(&w1, 2560 * 320 * 2));
  CK(hipMalloc(&w2, 320 * 1280 * 2));
  CK(hipMalloc(&Wp, (size_t)(FF_NCH + 1) * FF_CHUNK_BYTES));
  CK(hipMalloc(&hid, (size_t)M * 1280 * 2));
  CK(hipMalloc(&zero, 16384));
  CK(hipMemset(zero, 0, 16384));
  CK(hipMalloc(&b1, 2560 * 4));
  CK(hipMalloc(&b2, 320 * 4));
  CK(hipMalloc(&R, (size_t)M * 320 * 4));
  CK(hipMalloc(&o_ref, (size_t)M * 320 * 4));
  CK(hipMalloc(&o_f, (size_t)M * 320 * 4));
  CK(hipMalloc(&res, 8));
  CK(hipMalloc(&sums, 16));
  fill_f16<<<2048, 256, 0, st>>>(X, (size_t)M * 320, 1u, 2.0f);
  fill_f16<<<256, 256, 0, st>>>(w1, 2560 * 320, 2u, 1.7f / sqrtf(320.f));
  fill_f16<<<256, 256, 0, st>>>(w2, 320 * 1280, 3u, 1.7f / sqrtf(1280.f));
  fill_f32<<<16, 256, 0, st>>>(b1, 2560, 4u, 0.5f);
  fill_f32<<<4, 256, 0, st>>>(b2, 320, 5u, 0.5f);
  fill_f32<<<2048, 256, 0, st>>>(R, (size_t)M * 320, 6u, 1.0f);
  ff_pack_kernel<<<((FF_NCH + 1) * 60 * 64 + 255) / 256, 256, 0, st>>>(w1, w2, Wp, 0);
  CK(hipStreamSynchronize(st));

  // ---- the two-GEMM path of the library ----
  float* skws;
  CK(hipMalloc(&skws, 65536));
  gcd_gemm_desc d1, d2;
  memset(&d1, 0, sizeof(d1));
  d1.A = X; d1.W = w1; d1.lda = 320; d1.ldo = 1280; d1.M = M; d1.N = 2560; d1.K = 320; d1.mode = GCD_GEMM_PLAIN;
  d1.stride = 1; d1.bias = b1; d1.s_acc = d1.s_r1 = d1.s_r2 = 1.0f; d1.out_kind = GCD_OUT_GEGLU; d1.zero_page = zero;
  d1.out = hid; d1.workspace = skws; d1.workspace_bytes = 65536;
  d2 = d1;
  d2.A = hid; d2.W = w2; d2.lda = 1280; d2.ldo = 320; d2.N = 320; d2.K = 1280; d2.bias = b2; d2.out_kind = GCD_OUT_F32;
  d2.out = o_ref; d2.R1 = R; d2.ldr1 = 320; d2.sched = 1;
  float us_pair[3] = {0, 0, 0};
  {
    std::vector<float> ta, tb, tp;
    for (int it = 0; it < iters + 2; ++it) {
      float ms1, ms2;
      CK(hipEventRecord(e0, st));
      if (gcd_gemm_f16(&d1, st)) { fprintf(stderr, "gemm failed: %s\n", gcd_last_error()); return 1; }
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms1, e0, e1));
      CK(hipEventRecord(e0, st));
      if (gcd_gemm_f16(&d2, st)) { fprintf(stderr, "gemm failed: %s\n", gcd_last_error()); return 1; }
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms2, e0, e1));
      // back to back, one pair of events (what the step sees: FF-out finds part of the hidden tensor in the caches)
      CK(hipEventRecord(e0, st));
      gcd_gemm_f16(&d1, st);
      gcd_gemm_f16(&d2, st);
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      float msp;
      CK(hipEventElapsedTime(&msp, e0, e1));
      if (it >= 2) { ta.push_back(ms1 * 1e3f); tb.push_back(ms2 * 1e3f); tp.push_back(msp * 1e3f); }
    }
    std::sort(ta.begin(), ta.end()); std::sort(tb.begin(), tb.end()); std::sort(tp.begin(), tp.end());
    us_pair[0] = ta[ta.size() / 2]; us_pair[1] = tb[tb.size() / 2]; us_pair[2] = tp[tp.size() / 2];
  }
  const double fl = 2.0 * M * 320.0 * 3840.0;
  printf("M = %d   two-GEMM path: GEGLU %.1f us + FF-out %.1f us = %.1f us; back to back %.1f us = %.0f TF/s\n", M,
         us_pair[0], us_pair[1], us_pair[0] + us_pair[1], us_pair[2], fl / us_pair[2] * 1e-6);

  FfK k;
  memset(&k, 0, sizeof(k));
  k.X = X; k.ldx = 320; k.Wp = Wp; k.b1 = b1; k.b2 = b2; k.R1 = R; k.ldr1 = 320; k.out = o_f; k.ldo = 320;
  k.s_acc = k.s_r2 = 1.0f; k.M = M;
  unsigned long long* dbg = nullptr;
#if defined(FF_TIMING) || (defined(FF_STAMP_MODE) && FF_STAMP_MODE >= 10)
  CK(hipMalloc(&dbg, 64 * 64 * 8));
  CK(hipMemset(dbg, 0, 64 * 64 * 8));
  k.dbg = dbg;
#endif
  int bad = 0;
  for (int var = 0; var < 4; ++var) {
    const int T = 2 + (var >> 1), DEG = (var & 1) ? 15 : 19;
    CK(hipMemsetAsync(o_f, 0xff, (size_t)M * 320 * 4, st));
    const float us = var == 0 ? run_fused<2, 19>(k, st, iters, e0, e1) : var == 1 ? run_fused<2, 15>(k, st, iters, e0, e1)
                     : var == 2 ? run_fused<3, 19>(k, st, iters, e0, e1) : run_fused<3, 15>(k, st, iters, e0, e1);
    CK(hipMemsetAsync(res, 0, 8, st));
    CK(hipMemsetAsync(sums, 0, 16, st));
    cmp_kernel<<<1024, 256, 0, st>>>(o_f, o_ref, (size_t)M * 320, res, sums);
    unsigned hres[2];
    double hs[2];
    CK(hipMemcpyAsync(hres, res, 8, hipMemcpyDeviceToHost, st));
    CK(hipMemcpyAsync(hs, sums, 16, hipMemcpyDeviceToHost, st));
    CK(hipStreamSynchronize(st));
    float md, mb;
    memcpy(&md, &hres[0], 4);
    memcpy(&mb, &hres[1], 4);
    // fp64 host check of sampled outputs (hidden rounded to fp16 as both device paths do)
    double hostmax = 0;
    {
      std::vector<f16> xr(320), hw1(2560 * 320), hw2(320 * 1280);
      std::vector<float> hb1(2560), hb2(320);
      CK(hipMemcpy(hw1.data(), w1, hw1.size() * 2, hipMemcpyDeviceToHost));
      CK(hipMemcpy(hw2.data(), w2, hw2.size() * 2, hipMemcpyDeviceToHost));
      CK(hipMemcpy(hb1.data(), b1, 2560 * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(hb2.data(), b2, 320 * 4, hipMemcpyDeviceToHost));
      for (int q = 0; q < 6; ++q) {
        const int m = q == 0 ? 0 : q == 1 ? M - 1 : (int)(((uint64_t)q * 2654435761u + 12345) % M);
        CK(hipMemcpy(xr.data(), X + (size_t)m * 320, 640, hipMemcpyDeviceToHost));
        std::vector<double> h(1280);
        for (int hh = 0; hh < 1280; ++hh) {
          const int rv = (hh / 16) * 32 + hh % 16, rg = rv + 16;
          double v = hb1[rv], gt = hb1[rg];
          for (int c = 0; c < 320; ++c) {
            v += (double)xr[c] * (double)hw1[(size_t)rv * 320 + c];
            gt += (double)xr[c] * (double)hw1[(size_t)rg * 320 + c];
          }
          const double ge = 0.5 * gt * (1.0 + erf(gt * 0.70710678118654752440));
          h[hh] = (double)(f16)(float)(v * ge);
        }
        std::vector<float> got(320), rr(320);
        CK(hipMemcpy(got.data(), o_f + (size_t)m * 320, 1280, hipMemcpyDeviceToHost));
        CK(hipMemcpy(rr.data(), R + (size_t)m * 320, 1280, hipMemcpyDeviceToHost));
        for (int c = 0; c < 320; ++c) {
          double acc = hb2[c] + rr[c];
          for (int hh = 0; hh < 1280; ++hh) acc += h[hh] * (double)hw2[(size_t)c * 1280 + hh];
          hostmax = fmax(hostmax, fabs(acc - got[c]));
        }
      }
    }
    if (!(md < 1e30f)) {      // where are the wrong values?
      std::vector<float> ho((size_t)std::min(M, 4096) * 320), hr((size_t)std::min(M, 4096) * 320);
      CK(hipMemcpy(ho.data(), o_f, ho.size() * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(hr.data(), o_ref, hr.size() * 4, hipMemcpyDeviceToHost));
      int shown = 0;
      for (size_t i = 0; i < ho.size() && shown < 24; ++i)
        if (!(fabsf(ho[i] - hr[i]) < 1e-2f)) { printf("   bad [row %zu col %zu] got %g want %g\n", i / 320, i % 320, ho[i], hr[i]); ++shown; }
      size_t nbad = 0;
      std::vector<int> colbad(320, 0), rowbad(64, 0);
      for (size_t i = 0; i < ho.size(); ++i)
        if (!(fabsf(ho[i] - hr[i]) < 1e-2f)) { ++nbad; ++colbad[i % 320]; ++rowbad[(i / 320) % 64]; }
      printf("   %zu bad of %zu; by column block of 16:", nbad, ho.size());
      for (int c = 0; c < 20; ++c) { int q = 0; for (int j = 0; j < 16; ++j) q += colbad[16 * c + j]; printf(" %d", q); }
      printf("\n   by row mod 64:");
      for (int c = 0; c < 64; ++c) printf(" %d", rowbad[c]);
      printf("\n");
    }
    const double rel = sqrt(hs[0] / fmax(hs[1], 1e-30));
    const bool ok = rel < 2e-4 && hostmax < 2e-2 * fmax(mb, 1.f);
    if (!ok) ++bad;
    printf("fused T=%d deg %d (%3d-token tiles, %d tiles): %8.1f us = %6.0f TF/s | vs two-GEMM: %.2fx | max|diff| %.3e (max|ref| %.2f) "
           "rel-L2 %.2e | fp64 host check max %.3e %s\n",
           T, DEG, 64 * T, (M + 64 * T - 1) / (64 * T), us, fl / us * 1e-6, us_pair[2] / us, md, mb, rel, hostmax,
           ok ? "" : "MISMATCH");
    fflush(stdout);
#ifdef FF_TIMING
    {
      std::vector<unsigned long long> h(64 * 64);
      CK(hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost));
      const int nt = ((M + 64 * T - 1) / (64 * T) + 255) / 256;
      printf("  s_memtime ticks of workgroup 0 / wave 0 (100 MHz constant clock? compare with the wall time): per tile\n");
      for (int tl = 0; tl < nt && tl < 64; ++tl) {
        const unsigned long long* q = &h[tl * 64];
        if (!q[5]) continue;
        printf("   tile %d: setup %llu | first iteration %llu | 39 iterations %llu (%llu each) | last iteration + stores %llu | "
               "boundary wait + barrier %llu | total %llu\n",
               tl, q[1] - q[0], q[2] - q[1], q[3] - q[2], (q[3] - q[2]) / 39, q[4] - q[3], q[5] - q[4], q[5] - q[0]);
        if (q[14])
          printf("           first iteration: slots 0-19 (+ residual loads) %llu | 20-39 %llu | 40-53 %llu | vmcnt(0) %llu | barrier %llu | rest %llu"
                 "   last iteration: loads + GELU tail %llu | slots 40-49 %llu | 50-59 %llu\n",
                 q[10] - q[1], q[11] - q[10], q[12] - q[11], q[13] - q[12], q[14] - q[13], q[2] - q[14], q[21] - q[3], q[22] - q[21],
                 q[4] - q[22]);
      }
    }
#endif
  }
  // ---- the LayerNorm form: x = ff(norm(x + pos)) + (x + pos) in one launch, against LayerNorm kernel + two GEMMs ----
  {
    f16 *WpL, *X2;
    float *gam, *bet, *pos, *zsum, *o_ref2;
    const int rpv = 9216;
    const int nvec = (M + rpv - 1) / rpv;
    CK(hipMalloc(&WpL, (size_t)(FF_NCH + 1) * FF_CHUNK_BYTES));
    CK(hipMalloc(&X2, (size_t)M * 320 * 2));
    CK(hipMalloc(&gam, 320 * 4));
    CK(hipMalloc(&bet, 320 * 4));
    CK(hipMalloc(&pos, (size_t)nvec * 320 * 4));
    CK(hipMalloc(&zsum, (size_t)M * 320 * 4));
    CK(hipMalloc(&o_ref2, (size_t)M * 320 * 4));
    fill_f32<<<4, 256, 0, st>>>(gam, 320, 7u, 1.0f);
    fill_f32<<<4, 256, 0, st>>>(bet, 320, 8u, 0.5f);
    fill_f32<<<64, 256, 0, st>>>(pos, (size_t)nvec * 320, 9u, 0.7f);
    ff_pack_kernel<<<((FF_NCH + 1) * 60 * 64 + 255) / 256, 256, 0, st>>>(w1, w2, WpL, 1);
    CK(hipStreamSynchronize(st));
    gcd_gemm_desc e1d = d1, e2d = d2;
    e1d.A = X2;
    e2d.R1 = zsum; e2d.out = o_ref2;
    std::vector<float> tl, tp;
    for (int it = 0; it < iters + 2; ++it) {
      float ms1, ms2;
      CK(hipEventRecord(e0, st));
      if (gcd_layernorm_f16(R, 320, M, 320, gam, bet, 1e-5f, pos, 320, rpv, zsum, 320, X2, 320, 0, st)) {
        fprintf(stderr, "layernorm failed: %s\n", gcd_last_error());
        return 1;
      }
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms1, e0, e1));
      CK(hipEventRecord(e0, st));
      gcd_gemm_f16(&e1d, st);
      gcd_gemm_f16(&e2d, st);
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms2, e0, e1));
      if (it >= 2) { tl.push_back(ms1 * 1e3f); tp.push_back(ms2 * 1e3f); }
    }
    std::sort(tl.begin(), tl.end()); std::sort(tp.begin(), tp.end());
    const float us_ln = tl[tl.size() / 2], us_2 = tp[tp.size() / 2];
    FfK kl = k;
    kl.X = nullptr; kl.R1 = nullptr; kl.Wp = WpL; kl.x32 = R; kl.ldx32 = 320; kl.ln_gamma = gam; kl.ln_beta = bet; kl.ln_eps = 1e-5f;
    kl.addvec = pos; kl.ld_addvec = 320; kl.rows_per_vec = rpv;
    CK(hipFuncSetAttribute((const void*)ff_fused_kernel_probe<2, 19, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, FF_SMEM));
    std::vector<float> tf;
    for (int it = 0; it < iters + 2; ++it) {
      CK(hipEventRecord(e0, st));
      ff_fused_kernel_probe<2, 19, 0, true><<<g_grid, 256, FF_SMEM, st>>>(kl);
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      CK(hipGetLastError());
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (it >= 2) tf.push_back(ms * 1e3f);
    }
    std::sort(tf.begin(), tf.end());
#if defined(FF_DEBUGX) && (FF_DEBUGX == 4 || FF_DEBUGX == 5)
    {
      const int hbase = FF_DEBUGX == 4 ? 0 : 38 * 32;
      f16* dx;
      const size_t nd = (size_t)256 * 256 * 2 * 8;
      CK(hipMalloc(&dx, nd * 2));
      CK(hipMemset(dx, 0, nd * 2));
      kl.dbg = (unsigned long long*)dx;
      ff_fused_kernel_probe<2, 19, 0, true><<<g_grid, 256, FF_SMEM, st>>>(kl);
      CK(hipStreamSynchronize(st));
      kl.dbg = nullptr;
      gcd_gemm_f16(&e1d, st);      // the reference hidden tensor (LayerNorm kernel's output through the GEGLU GEMM)
      CK(hipStreamSynchronize(st));
      std::vector<f16> hx(nd), hh((size_t)std::min(M, 256 * 128) * 1280);
      CK(hipMemcpy(hx.data(), dx, nd * 2, hipMemcpyDeviceToHost));
      CK(hipMemcpy(hh.data(), hid, hh.size() * 2, hipMemcpyDeviceToHost));
      double worst[2] = {0, 0};
      int shown = 0;
      const int ntile = (M + 127) / 128;
      for (int b = 0; b < std::min(ntile, g_grid) && b < 256; ++b)
        for (int t = 0; t < 256; ++t)
          for (int tb = 0; tb < 2; ++tb)
            for (int q = 0; q < 8; ++q) {
              const int wave = t >> 6, lane = t & 63, g = lane >> 4, r = lane & 15;
              const int row = 128 * b + 32 * wave + 16 * tb + r;
              if (row >= M || row >= 256 * 128) continue;
              const int hcol = q < 4 ? 4 * g + q : 12 + 4 * g + q;
              const float got = (float)hx[(((size_t)b * 256 + t) * 2 + tb) * 8 + q], want = (float)hh[(size_t)row * 1280 + hbase + hcol];
              const double er = fabs(got - want);
              if (er > worst[tb]) worst[tb] = er;
              if (er > 0.02 && shown++ < 8) printf("   H mismatch wg %d t %d tb %d q %d (row %d hidden %d): got %g want %g\n", b, t, tb, q, row, hcol, got, want);
            }
      printf("   H(0) as GEMM2 received it vs the GEGLU GEMM's hidden tensor: max |diff| tb0 %.4g, tb1 %.4g\n", worst[0], worst[1]);
    }
#elif defined(FF_DEBUGX) && FF_DEBUGX == 3
    {
      float* dx;
      const size_t nd = (size_t)256 * 256 * 40 * 4;
      CK(hipMalloc(&dx, nd * 4));
      CK(hipMemset(dx, 0, nd * 4));
      kl.dbg = (unsigned long long*)dx;
      ff_fused_kernel_probe<2, 19, 0, true><<<g_grid, 256, FF_SMEM, st>>>(kl);
      CK(hipStreamSynchronize(st));
      kl.dbg = nullptr;
      std::vector<float> hx(nd), hz((size_t)std::min(M, 256 * 128) * 320);
      CK(hipMemcpy(hx.data(), dx, nd * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(hz.data(), zsum, hz.size() * 4, hipMemcpyDeviceToHost));
      double worst[2] = {0, 0};
      int shown = 0;
      const int ntile = (M + 127) / 128;
      for (int b = 0; b < std::min(ntile, g_grid) && b < 256; ++b)
        for (int t = 0; t < 256; ++t)
          for (int tb = 0; tb < 2; ++tb)
            for (int cb = 0; cb < 20; ++cb)
              for (int e = 0; e < 4; ++e) {
                const int wave = t >> 6, lane = t & 63, g = lane >> 4, r = lane & 15;
                const int row = 128 * b + 32 * wave + 16 * tb + r;
                if (row >= M || row >= 256 * 128) continue;
                const int ch = 16 * cb + 4 * g + e;
                const float got = hx[(((size_t)b * 256 + t) * 40 + tb * 20 + cb) * 4 + e], want = hz[(size_t)row * 320 + ch];
                const double er = fabs(got - want);
                if (er > worst[tb]) worst[tb] = er;
                if (er > 1e-3 && shown++ < 8) printf("   acc mismatch wg %d t %d tb %d cb %d e %d (row %d ch %d): got %g want %g\n", b, t, tb, cb, e, row, ch, got, want);
              }
      printf("   accumulators after the first iteration vs x + pos: max |diff| tb0 %.4g, tb1 %.4g\n", worst[0], worst[1]);
    }
#elif defined(FF_DEBUGX)
    {
      f16* dx;
      const size_t nd = (size_t)256 * 256 * 20 * 8;
      CK(hipMalloc(&dx, nd * 2));
      CK(hipMemset(dx, 0, nd * 2));
      kl.dbg = (unsigned long long*)dx;
      ff_fused_kernel_probe<2, 19, 0, true><<<g_grid, 256, FF_SMEM, st>>>(kl);
      CK(hipStreamSynchronize(st));
      kl.dbg = nullptr;
      std::vector<f16> hx(nd), hl((size_t)std::min(M, 256 * 128) * 320);
      CK(hipMemcpy(hx.data(), dx, nd * 2, hipMemcpyDeviceToHost));
      CK(hipMemcpy(hl.data(), X2, hl.size() * 2, hipMemcpyDeviceToHost));
      // workgroup b's first tile = tile b: rows 128 b + 32 wave + 16 tb + r; fragment ks element q <-> channel 32 ks + 16 (q >> 2) + 4 g + (q & 3)
      double worst[2] = {0, 0};
      int shown = 0;
      const int ntile = (M + 127) / 128;
      for (int b = 0; b < std::min(ntile, g_grid) && b < 256; ++b)
        for (int t = 0; t < 256; ++t)
          for (int tb = 0; tb < 2; ++tb)
            for (int ks = 0; ks < 10; ++ks)
              for (int q = 0; q < 8; ++q) {
                const int wave = t >> 6, lane = t & 63, g = lane >> 4, r = lane & 15;
                const int row = 128 * b + 32 * wave + 16 * tb + r;
                if (row >= M || row >= 256 * 128) continue;
                const int ch = 32 * ks + 16 * (q >> 2) + 4 * g + (q & 3);
                const float got = (float)hx[(((size_t)b * 256 + t) * 20 + tb * 10 + ks) * 8 + q], want = (float)hl[(size_t)row * 320 + ch];
                const double e = fabs(got - want);
                if (e > worst[tb]) worst[tb] = e;
                if (e > 0.02 && shown++ < 8) printf("   X mismatch wg %d t %d tb %d ks %d q %d (row %d ch %d): got %g want %g\n", b, t, tb, ks, q, row, ch, got, want);
              }
      printf("   LN operand fragments vs the LayerNorm kernel's output: max |diff| tb0 %.4g, tb1 %.4g\n", worst[0], worst[1]);
    }
#endif
#if (FF_ABL & 2) != 0
    {   // no GELU operations: H = 0, the result must be z + b2 exactly
      const size_t n = (size_t)std::min(M, 4096) * 320;
      std::vector<float> ho(n), hz(n), hb(320);
      CK(hipMemcpy(ho.data(), o_f, n * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(hz.data(), zsum, n * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(hb.data(), b2, 320 * 4, hipMemcpyDeviceToHost));
      std::vector<int> rowbad(64, 0);
      size_t nb = 0;
      for (size_t i = 0; i < n; ++i)
        if (fabsf(ho[i] - (hz[i] + hb[i % 320])) > 1e-5f) { ++nb; ++rowbad[(i / 320) % 64]; }
      printf("   H = 0 build: %zu of %zu outputs differ from z + b2; by row mod 64:", nb, n);
      for (int c = 0; c < 64; ++c) printf(" %d", rowbad[c]);
      printf("\n");
    }
#endif
    CK(hipMemsetAsync(res, 0, 8, st));
    CK(hipMemsetAsync(sums, 0, 16, st));
    cmp_kernel<<<1024, 256, 0, st>>>(o_f, o_ref2, (size_t)M * 320, res, sums);
    unsigned hres[2];
    double hs[2];
    CK(hipMemcpyAsync(hres, res, 8, hipMemcpyDeviceToHost, st));
    CK(hipMemcpyAsync(hs, sums, 16, hipMemcpyDeviceToHost, st));
    CK(hipStreamSynchronize(st));
    float md, mb;
    memcpy(&md, &hres[0], 4);
    memcpy(&mb, &hres[1], 4);
    if (sqrt(hs[0] / fmax(hs[1], 1e-30)) > 1e-3) {
      std::vector<float> ho((size_t)std::min(M, 4096) * 320), hr((size_t)std::min(M, 4096) * 320);
      CK(hipMemcpy(ho.data(), o_f, ho.size() * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(hr.data(), o_ref2, hr.size() * 4, hipMemcpyDeviceToHost));
      size_t nbad = 0;
      std::vector<int> colbad(320, 0), rowbad(128, 0);
      int shown = 0;
      for (size_t i = 0; i < ho.size(); ++i)
        if (!(fabsf(ho[i] - hr[i]) < 1e-2f)) {
          ++nbad; ++colbad[i % 320]; ++rowbad[(i / 320) % 128];
          if (shown++ < 6) printf("   bad [row %zu col %zu] got %g want %g\n", i / 320, i % 320, ho[i], hr[i]);
        }
      {   // is the FeedForward part of a wrong row the FeedForward part of its partner row in the other token block?
        std::vector<float> hz(ho.size());
        CK(hipMemcpy(hz.data(), zsum, hz.size() * 4, hipMemcpyDeviceToHost));
        double d_self = 0, d_partner = 0, nrm = 0;
        for (int row = 16; row < 32; ++row)
          for (int c = 0; c < 320; ++c) {
            const double ff_got = ho[row * 320 + c] - hz[row * 320 + c];
            const double ff_self = hr[row * 320 + c] - hz[row * 320 + c];
            const double ff_part = hr[(row - 16) * 320 + c] - hz[(row - 16) * 320 + c];
            d_self += (ff_got - ff_self) * (ff_got - ff_self);
            d_partner += (ff_got - ff_part) * (ff_got - ff_part);
            nrm += ff_self * ff_self;
          }
        printf("   rows 16-31: FeedForward part vs its own reference %.3e, vs the partner row's (row - 16) %.3e (relative)\n",
               sqrt(d_self / nrm), sqrt(d_partner / nrm));
      }
      printf("   LN form: %zu bad of %zu; by column block of 16:", nbad, ho.size());
      for (int c = 0; c < 20; ++c) { int q = 0; for (int j = 0; j < 16; ++j) q += colbad[16 * c + j]; printf(" %d", q); }
      printf("\n   by row mod 128:");
      for (int c = 0; c < 128; ++c) printf(" %d", rowbad[c]);
      printf("\n");
    }
    const double rel = sqrt(hs[0] / fmax(hs[1], 1e-30));
    const bool ok = rel < 3e-4;      // (the LayerNorm output is rounded to fp16 in both paths: a last-bit difference of the fp32
                                     //  statistics flips a rounding here and there)
    if (!ok) ++bad;
    printf("LayerNorm form, T=2: LayerNorm kernel %.1f us + two GEMMs %.1f us = %.1f us | one launch %.1f us (%.2fx) | max|diff| %.3e "
           "(max|ref| %.2f) rel-L2 %.2e %s\n", us_ln, us_2, us_ln + us_2, tf[tf.size() / 2], (us_ln + us_2) / tf[tf.size() / 2], md, mb,
           rel, ok ? "" : "MISMATCH");
  }
  printf("%s\n", bad ? "RESULT: MISMATCH" : "RESULT: agree");
  return bad ? 1 : 0;
}
