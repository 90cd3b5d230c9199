// tools/gemm_bench.cpp — standalone A/B harness for the GEMM kernels of libgcd_amd.so (no Python, no
// torch: it starts in a second on a fresh GPU box).  For every GEMM shape of one VideoUNet step at
// 14x72x128 latents it runs the general kernel (GCD_TUNE_GEMM_IMPL=1) and the ping-pong kernel (=2)
// through the public C ABI, checks them against each other on the FULL output and against an fp64
// host reference on sampled entries (PLAIN mode), and times both with HIP events.
//
//   hipcc -O2 --offload-arch=gfx950 tools/gemm_bench.cpp -Iinclude -Lgcd_amd -lgcd_amd \
//         -Wl,-rpath,'$ORIGIN/../gcd_amd' -o tools/gemm_bench
//   tools/gemm_bench [quick|full] [iters]
// The ablation variants (impl >= 32) exist only in tools/libgcd_amd_ablate.so
// (`python -m gcd_amd.csrc.build --ablation`, then link with -Ltools -lgcd_amd_ablate instead); against
// the product library those impl numbers fall through to the product kernel.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "gcd_amd.h"

#define CK(x)                                                                         \
  do {                                                                                \
    hipError_t e_ = (x);                                                              \
    if (e_ != hipSuccess) {                                                           \
      fprintf(stderr, "%s failed: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); \
      exit(1);                                                                        \
    }                                                                                 \
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
__global__ void fill_f32(float* p, size_t n, uint32_t seed, float scale) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    uint32_t x = (uint32_t)i * 2246822519u + seed;
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    p[i] = ((float)(x & 0xffff) / 32768.0f - 1.0f) * scale;
  }
}
// res[0] = max |a-b|, res[1] = max |b|  (as ordered uint bit patterns of non-negative floats)
template <typename T>
__global__ void cmp_kernel(const T* a, const T* b, size_t n, unsigned* res) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  float md = 0.f, mb = 0.f;
  for (; i < n; i += stride) {
    const float x = (float)a[i], y = (float)b[i];
    float d = fabsf(x - y);
    if (!(d == d)) d = INFINITY;   // NaN counts as infinitely wrong
    md = fmaxf(md, d);
    mb = fmaxf(mb, fabsf(y));
  }
  atomicMax(&res[0], __float_as_uint(md));
  atomicMax(&res[1], __float_as_uint(mb));
}

struct Shape {
  int mode, M, N, K;
  int Cin, Hi, Wi, Ho, Wo, stride, up, T, HW;
  int epi;      // 0: f32 + bias; 1: f32 + bias + R1 in place; 2: GEGLU; 3: f16 out; 4: rowvec+alpha+R1+R2
  int count;    // launches per step (weight of this shape)
  const char* what;
};

static std::vector<Shape> unet_shapes(bool quick) {
  const int F = 28, T = 14;
  std::vector<Shape> v;
  auto lin = [&](int M, int N, int K, int epi, int cnt, const char* w) {
    v.push_back({GCD_GEMM_PLAIN, M, N, K, 0, 0, 0, 0, 0, 1, 0, 0, 0, epi, cnt, w});
  };
  auto conv = [&](int H, int W, int Cin, int N, int stride, int up, int epi, int cnt, const char* w) {
    int Ho = up ? 2 * H : (H - 1) / stride + 1, Wo = up ? 2 * W : (W - 1) / stride + 1;
    v.push_back({GCD_GEMM_CONV3X3, F * Ho * Wo, N, 9 * Cin, Cin, H, W, Ho, Wo, stride, up, 0, 0, epi, cnt, w});
  };
  auto tconv = [&](int HW, int C, int epi, int cnt, const char* w) {
    v.push_back({GCD_GEMM_TEMPORAL3, F * HW, C, 3 * C, C, 0, 0, 0, 0, 1, 0, T, HW, epi, cnt, w});
  };
  const int M0 = F * 72 * 128, M1 = F * 36 * 64, M2 = F * 18 * 32, M3 = F * 9 * 16;
  lin(M0, 2560, 320, 2, 15, "L0 GEGLU");
  lin(M0, 320, 1280, 1, 15, "L0 FF out");
  lin(M0, 320, 320, 1, 20, "L0 proj/o");
  lin(M0, 960, 320, 3, 10, "L0 qkv");
  conv(72, 128, 320, 320, 1, 0, 4, 7, "L0 conv3x3");
  tconv(72 * 128, 320, 4, 10, "L0 convT");
  lin(M1, 5120, 640, 2, 15, "L1 GEGLU");
  lin(M1, 640, 2560, 1, 15, "L1 FF out");
  lin(M1, 640, 640, 1, 20, "L1 proj/o");
  conv(36, 64, 640, 640, 1, 0, 1, 6, "L1 conv3x3");
  lin(M2, 10240, 1280, 2, 15, "L2 GEGLU");
  lin(M2, 1280, 5120, 1, 15, "L2 FF out");
  conv(18, 32, 1280, 1280, 1, 0, 1, 7, "L2 conv3x3");
  lin(M2, 1280, 1280, 1, 20, "L2 proj/o");
  if (!quick) {
    lin(M1, 1920, 640, 3, 10, "L1 qkv");
    lin(M2, 3840, 1280, 3, 10, "L2 qkv");
    tconv(36 * 64, 640, 4, 10, "L1 convT");
    tconv(18 * 32, 1280, 4, 10, "L2 convT");
    conv(72, 128, 640, 320, 1, 0, 0, 2, "L0 conv 640->320");
    conv(72, 128, 960, 320, 1, 0, 0, 1, "L0 conv 960->320");
    conv(18, 32, 2560, 1280, 1, 0, 0, 2, "L2 conv 2560->1280");
    conv(36, 64, 1280, 1280, 1, 1, 0, 1, "L1 up conv (x2 fused)");
    conv(72, 128, 320, 320, 2, 0, 0, 1, "L0 down conv s2");
    conv(9, 16, 1280, 1280, 1, 0, 1, 12, "L3 conv3x3 (M=4032)");
    tconv(9 * 16, 1280, 4, 14, "L3 convT");
    lin(M3, 10240, 1280, 2, 3, "mid GEGLU");
    lin(M3 - 100, 640, 384, 4, 0, "ragged M");
    lin(1000, 48, 128, 0, 0, "tiny");
  }
  return v;
}

int main(int argc, char** argv) {
  bool quick = argc > 1 && !strcmp(argv[1], "quick");
  int iters = argc > 2 ? atoi(argv[2]) : 5;
  const char* only = argc > 3 ? argv[3] : nullptr;        // substring filter on the shape name
  std::vector<int> extra;                                 // extra GCD_TUNE_GEMM_IMPL values to time
  for (int i = 4; i < argc; ++i) extra.push_back(atoi(argv[i]));
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  f16* zero;
  CK(hipMalloc(&zero, 16384));
  CK(hipMemset(zero, 0, 16384));
  unsigned* res;
  CK(hipMalloc(&res, 8));
  double tot_ms[2] = {0, 0}, tot_fl = 0;
  int bad = 0;
  printf("%-24s %8s %6s %6s epi | %9s %8s | %9s %8s | %6s | %9s %9s\n", "shape", "M", "N", "K",
         "legacy us", "TF/s", "pp us", "TF/s", "speed", "maxdiff", "hostchk");
  for (const Shape& s : unet_shapes(quick)) {
    if (only && strcmp(only, "-") && !strstr(s.what, only)) continue;
    size_t a_rows = s.M;
    int lda = s.K;
    if (s.mode == GCD_GEMM_CONV3X3) { a_rows = (size_t)(s.M / (s.Ho * s.Wo)) * s.Hi * s.Wi; lda = s.Cin; }
    if (s.mode == GCD_GEMM_TEMPORAL3) lda = s.Cin;
    const size_t a_n = a_rows * lda, w_n = (size_t)s.N * s.K;
    const bool geglu = s.epi == 2, f16out = s.epi == 3 || geglu;
    const int ncols = geglu ? s.N / 2 : s.N;
    const size_t o_n = (size_t)s.M * ncols;
    f16 *A, *W;
    void *o[2];
    float *bias, *r1 = nullptr, *r2 = nullptr, *rv = nullptr, *alpha = nullptr;
    CK(hipMalloc(&A, a_n * 2));
    CK(hipMalloc(&W, w_n * 2));
    CK(hipMalloc(&bias, s.N * 4));
    for (int i = 0; i < 2; ++i) CK(hipMalloc(&o[i], o_n * (f16out ? 2 : 4)));
    fill_f16<<<2048, 256, 0, st>>>(A, a_n, 1u, 1.0f);
    fill_f16<<<2048, 256, 0, st>>>(W, w_n, 2u, 1.0f / sqrtf((float)s.K));
    fill_f32<<<64, 256, 0, st>>>(bias, s.N, 3u, 1.0f);
    const int rows_per_frame = s.mode == GCD_GEMM_CONV3X3 ? s.Ho * s.Wo : (s.HW ? s.HW : 4096);
    const int nfr = (s.M + rows_per_frame - 1) / rows_per_frame;
    if (s.epi == 1 || s.epi == 4) {
      CK(hipMalloc(&r1, o_n * 4));
      fill_f32<<<2048, 256, 0, st>>>(r1, o_n, 4u, 1.0f);
    }
    if (s.epi == 4) {
      CK(hipMalloc(&r2, o_n * 4));
      fill_f32<<<2048, 256, 0, st>>>(r2, o_n, 5u, 1.0f);
      CK(hipMalloc(&rv, (size_t)nfr * s.N * 4));
      fill_f32<<<64, 256, 0, st>>>(rv, (size_t)nfr * s.N, 6u, 1.0f);
      CK(hipMalloc(&alpha, nfr * 4));
      fill_f32<<<1, 64, 0, st>>>(alpha, nfr, 7u, 0.5f);
    }
    gcd_gemm_desc d;
    memset(&d, 0, sizeof(d));
    d.A = A; d.W = W; d.lda = lda; d.ldo = ncols; d.M = s.M; d.N = s.N; d.K = s.K; d.mode = s.mode;
    d.Cin = s.Cin; d.Hi = s.Hi; d.Wi = s.Wi; d.Ho = s.Ho; d.Wo = s.Wo; d.stride = s.stride;
    d.upsample = s.up; d.T = s.T; d.HW = s.HW; d.bias = bias; d.s_acc = d.s_r1 = d.s_r2 = 1.0f;
    d.out_kind = geglu ? GCD_OUT_GEGLU : f16out ? GCD_OUT_F16 : GCD_OUT_F32;
    d.zero_page = zero;
    float* skws = nullptr;   // scratch: split-K partials for the few-tile shapes, per-CU counters else
    {
      size_t wsb = 65536;
      if ((int64_t)((s.M + 255) / 256) * ((s.N + 319) / 320) <= 96) wsb = (size_t)4 * s.M * s.N * 4;
      CK(hipMalloc(&skws, wsb));
      d.workspace = skws;
      d.workspace_bytes = (int64_t)wsb;
    }
    if (r1) { d.R1 = r1; d.ldr1 = ncols; }
    if (s.epi == 4) {
      d.R2 = r2; d.ldr2 = ncols; d.rowvec = rv; d.ld_rowvec = s.N; d.rows_per_vec = rows_per_frame;
      d.frame_alpha = alpha; d.rows_per_alpha = rows_per_frame; d.r1_blend = 1;
    }
    float us[2] = {0, 0};
    for (int impl = 0; impl < 2; ++impl) {
      gcd_tune_set(GCD_TUNE_GEMM_IMPL, impl + 1);
      d.out = o[impl];
      std::vector<float> t;
      for (int it = 0; it < iters + 2; ++it) {
        CK(hipEventRecord(e0, st));
        if (gcd_gemm_f16(&d, st)) { fprintf(stderr, "gemm failed: %s\n", gcd_last_error()); exit(1); }
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (it >= 2) t.push_back(ms * 1e3f);
      }
      std::sort(t.begin(), t.end());
      us[impl] = t[t.size() / 2];
    }
    CK(hipMemsetAsync(res, 0, 8, st));
    if (f16out) cmp_kernel<f16><<<1024, 256, 0, st>>>((const f16*)o[1], (const f16*)o[0], o_n, res);
    else cmp_kernel<float><<<1024, 256, 0, st>>>((const float*)o[1], (const float*)o[0], o_n, res);
    unsigned hres[2];
    CK(hipMemcpyAsync(hres, res, 8, hipMemcpyDeviceToHost, st));
    CK(hipStreamSynchronize(st));
    float md, mb;
    memcpy(&md, &hres[0], 4);
    memcpy(&mb, &hres[1], 4);
    // host fp64 check of sampled entries (PLAIN, simple epilogues only)
    double hostchk = -1;
    if (s.mode == GCD_GEMM_PLAIN && (s.epi == 0 || s.epi == 3)) {
      hostchk = 0;
      std::vector<f16> arow(s.K), wrow(s.K);
      for (int q = 0; q < 24; ++q) {
        const int m = (int)(((uint64_t)q * 2654435761u + 12345) % s.M);
        const int n = (int)(((uint64_t)q * 40503u + 77) % s.N);
        const int mm = q < 2 ? (q ? s.M - 1 : 0) : m, nn = q < 2 ? (q ? s.N - 1 : 0) : n;
        CK(hipMemcpy(arow.data(), A + (size_t)mm * lda, s.K * 2, hipMemcpyDeviceToHost));
        CK(hipMemcpy(wrow.data(), W + (size_t)nn * s.K, s.K * 2, hipMemcpyDeviceToHost));
        float b;
        CK(hipMemcpy(&b, bias + nn, 4, hipMemcpyDeviceToHost));
        double acc = b;
        for (int k = 0; k < s.K; ++k) acc += (double)arow[k] * (double)wrow[k];
        float got;
        if (f16out) { f16 g; CK(hipMemcpy(&g, (f16*)o[1] + (size_t)mm * ncols + nn, 2, hipMemcpyDeviceToHost)); got = (float)g; }
        else CK(hipMemcpy(&got, (float*)o[1] + (size_t)mm * ncols + nn, 4, hipMemcpyDeviceToHost));
        hostchk = fmax(hostchk, fabs(got - acc));
      }
    }
    const double fl = 2.0 * s.M * s.N * s.K;
    for (int impl : extra) {   // ablation / variant timings (results not checked)
      if (impl >= 32 && s.mode != GCD_GEMM_PLAIN) continue;
      gcd_tune_set(GCD_TUNE_GEMM_IMPL, impl);
      d.out = o[1];
      std::vector<float> t;
      for (int it = 0; it < iters + 2; ++it) {
        CK(hipEventRecord(e0, st));
        if (gcd_gemm_f16(&d, st)) { fprintf(stderr, "gemm failed: %s\n", gcd_last_error()); exit(1); }
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (it >= 2) t.push_back(ms * 1e3f);
      }
      std::sort(t.begin(), t.end());
      float xd = -1.f;
      if (impl < 32) {   // a product kernel: check it against the general kernel's output
        CK(hipMemsetAsync(res, 0, 8, st));
        if (f16out) cmp_kernel<f16><<<1024, 256, 0, st>>>((const f16*)o[1], (const f16*)o[0], o_n, res);
        else cmp_kernel<float><<<1024, 256, 0, st>>>((const float*)o[1], (const float*)o[0], o_n, res);
        unsigned hr[2];
        CK(hipMemcpyAsync(hr, res, 8, hipMemcpyDeviceToHost, st));
        CK(hipStreamSynchronize(st));
        memcpy(&xd, &hr[0], 4);
        if (xd > (f16out ? 4e-3 : 2e-4) * fmax(mb, 1.0f)) ++bad;
      }
      printf("    impl %3d: %9.1f us %8.1f TF/s  (%.2fx legacy)  maxdiff %9.2e\n", impl, t[t.size() / 2],
             fl / t[t.size() / 2] * 1e-6, us[0] / t[t.size() / 2], xd);
    }
    const double tol = (f16out ? 4e-3 : 2e-4) * fmax(mb, 1.0f);
    const bool ok = md <= tol && (hostchk < 0 || hostchk <= (f16out ? 8e-3 : 1e-3) * fmax(mb, 1.0f));
    if (!ok) ++bad;
    printf("%-24s %8d %6d %6d  %d  | %9.1f %8.1f | %9.1f %8.1f | %5.2fx | %9.2e %9.2e %s\n", s.what, s.M,
           s.N, s.K, s.epi, us[0], fl / us[0] * 1e-6, us[1], fl / us[1] * 1e-6, us[0] / us[1], md,
           hostchk, ok ? "" : "MISMATCH");
    fflush(stdout);
    tot_ms[0] += us[0] * 1e-3 * s.count;
    tot_ms[1] += us[1] * 1e-3 * s.count;
    tot_fl += fl * s.count;
    hipFree(A); hipFree(W); hipFree(bias); hipFree(o[0]); hipFree(o[1]);
    if (r1) hipFree(r1);
    if (r2) hipFree(r2);
    if (rv) hipFree(rv);
    if (alpha) hipFree(alpha);
    if (skws) hipFree(skws);
  }
  printf("weighted by launches/step: legacy %.2f ms (%.0f TF/s)  pp %.2f ms (%.0f TF/s)  over %.1f TFLOP\n",
         tot_ms[0], tot_fl / tot_ms[0] * 1e-9, tot_ms[1], tot_fl / tot_ms[1] * 1e-9, tot_fl * 1e-12);
  printf("%s\n", bad ? "RESULT: MISMATCHES" : "RESULT: all shapes agree");
  return bad ? 1 : 0;
}
