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
// Kernel (train_wgrad_kernel.h): 128 (n) x 128 (k) output tile per workgroup, 4 waves of 64 x 64 (16 accumulators of
// v_mfma_f32_16x16x32) — or, round 6, 160 x 160 with 4 waves of 80 x 80 where that removes padded work (every width of the
// UNet is a multiple of 320 = 2.5 x 128; gcd_wgrad::tile_of) — 32 or 64 tokens per step, operands through registers into a double-buffered LDS tile (272-byte
// rows), the token axis split over S workgroups whose fp32 partial outputs a second launch folds (no atomics).  Untuned: no
// LDS-DMA, no swizzle, one barrier per 16 / 32 MFMAs per wave.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/gcd_amd_train.h"
#include "train_wgrad_kernel.h"

static thread_local char g_err[512] = "";
void gcd_train_set_error(const char* fmt, ...) {     // shared with train_ops.hip
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
#define set_error gcd_train_set_error
extern "C" const char* gcd_train_last_error(void) { return g_err; }
extern "C" int gcd_train_abi_version(void) { return GCD_AMD_TRAIN_ABI_VERSION; }

#define CHECK_ARG(cond, ...)  \
  do {                        \
    if (!(cond)) {            \
      set_error(__VA_ARGS__); \
      return 2;               \
    }                         \
  } while (0)

// Tokens per step of the kernel.  32 is the form A/B'd inside the whole fine-tune step (profiles/r04w_wgrad_ab.txt); the
// 64-token form (one barrier per 32 MFMAs of a wave, 70 KB of LDS) is 9-25 % faster on the large weight gradients of the
// 43 008-token level and 3-19 % slower on the small ones (tools/gemm_tr_probe.cpp, which compiles the same header and
// fp64-checks both: profiles/r04x_gemm_tr_probe_product_header_tm32_vs_tm64.txt), hence the rule below.
// GCD_WGRAD_TM=32 / 64 forces one form.
static int wgrad_tm(int64_t M, int N, int K) {
  static const int forced = [] {
    const char* e = getenv("GCD_WGRAD_TM");
    const int v = e ? atoi(e) : 0;
    return (v == 32 || v == 64) ? v : 0;
  }();
  if (forced) return forced;
  // the 160 x 160 tile (round 6) takes 86 KB of LDS at 64 tokens per step — one workgroup per CU; at 32 it is 43 KB
  if (gcd_wgrad::tile_of(N, K) == 160) return 32;
  return (M >= 32768 && (int64_t)N * K >= 400000) ? 64 : GCD_WGRAD_TM_DEFAULT;
}

extern "C" int64_t gcd_wgrad_tr_scratch_floats(int64_t M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  return (int64_t)gcd_wgrad::slices(M, N, K) * N * K;
}

extern "C" int gcd_wgrad_tr_f16(const void* dy16, int64_t lddy, const void* x16, int64_t ldx, int64_t M, int N, int K,
                                int bf16, float* dW, int64_t lddw, float* scratch, int64_t scratch_floats,
                                void* stream) {
  return gcd_wgrad_tr_f16_ex(dy16, lddy, x16, ldx, M, N, K, bf16, dW, lddw, 1, N, K, 0, scratch, scratch_floats, stream);
}

extern "C" int gcd_wgrad_tr_f16_ex(const void* dy16, int64_t lddy, const void* x16, int64_t ldx, int64_t M, int N, int K,
                                   int bf16, float* dW, int64_t lddw, int taps, int N_real, int C_real, int accumulate,
                                   float* scratch, int64_t scratch_floats, void* stream) {
  CHECK_ARG(taps >= 1 && taps <= 9 && K % taps == 0 && N_real >= 1 && N_real <= N && C_real >= 1 && C_real <= K / taps,
            "gcd_wgrad_tr_f16_ex: taps=%d N_real=%d C_real=%d for N=%d K=%d", taps, N_real, C_real, N, K);
  CHECK_ARG(taps > 1 || (C_real % 4 == 0), "gcd_wgrad_tr_f16_ex: C_real=%d must be a multiple of 4 for taps = 1", C_real);
  const gcd_wgrad::Layout lay = {taps, N_real, C_real, accumulate};
  CHECK_ARG(dy16 && x16 && dW && scratch, "gcd_wgrad_tr_f16: null pointer");
  CHECK_ARG(M > 0 && N > 0 && K > 0 && N % 8 == 0 && K % 8 == 0,
            "gcd_wgrad_tr_f16: M=%lld N=%d K=%d (N and K must be multiples of 8)", (long long)M, N, K);
  CHECK_ARG(lddy % 8 == 0 && lddy >= N && ldx % 8 == 0 && ldx >= K && (taps > 1 || (lddw % 4 == 0 && lddw >= C_real)),
            "gcd_wgrad_tr_f16: leading dimensions lddy=%lld ldx=%lld lddw=%lld", (long long)lddy, (long long)ldx,
            (long long)lddw);
  CHECK_ARG((((uintptr_t)dy16 | (uintptr_t)x16 | (uintptr_t)dW | (uintptr_t)scratch) & 15) == 0,
            "gcd_wgrad_tr_f16: operands must be 16-byte aligned");
  const int64_t need = gcd_wgrad_tr_scratch_floats(M, N, K);
  CHECK_ARG(scratch_floats >= need, "gcd_wgrad_tr_f16: scratch of %lld floats, need %lld (gcd_wgrad_tr_scratch_floats)",
            (long long)scratch_floats, (long long)need);
  hipStream_t s = (hipStream_t)stream;
  hipError_t e;
  if (wgrad_tm(M, N, K) == 64)
    e = bf16 ? gcd_wgrad::launch<true, 64>(dy16, lddy, x16, ldx, M, N, K, dW, lddw, lay, scratch, s)
             : gcd_wgrad::launch<false, 64>(dy16, lddy, x16, ldx, M, N, K, dW, lddw, lay, scratch, s);
  else
    e = bf16 ? gcd_wgrad::launch<true, 32>(dy16, lddy, x16, ldx, M, N, K, dW, lddw, lay, scratch, s)
             : gcd_wgrad::launch<false, 32>(dy16, lddy, x16, ldx, M, N, K, dW, lddw, lay, scratch, s);
  if (e != hipSuccess) {
    set_error("gcd_wgrad_tr_f16: launch failed: %s", hipGetErrorString(e));
    return 1;
  }
  return 0;
}

// The weight gradient of a stride-1 3 x 3 convolution (conv = 1) or of the (3,1,1) temporal convolution (conv = 2) with the
// X operand gathered IMPLICITLY per tap (train_wgrad_kernel.h, CONV): x16 is the convolution's INPUT activation [M, Cp]
// (token-major, the operand the forward pass left), dy16 [M, N] its output gradient; K = taps * Cp.  The result lands in the
// parameter's layout [N_real][C_real][taps] as gcd_wgrad_tr_f16_ex's.  No im2col tensor.
extern "C" int gcd_wgrad_conv_tr_f16(const void* dy16, int64_t lddy, const void* x16, int64_t ldx, int64_t M, int N, int Cp,
                                     int conv, int Ho, int Wo, int T, int HW, int bf16, float* dW, int N_real, int C_real,
                                     int accumulate, float* scratch, int64_t scratch_floats, void* stream) {
  CHECK_ARG(conv == 1 || conv == 2, "gcd_wgrad_conv_tr_f16: conv=%d (1: 3x3 stride 1, 2: (3,1,1))", conv);
  const int taps = conv == 1 ? 9 : 3;
  const int K = taps * Cp;
  CHECK_ARG(dy16 && x16 && dW && scratch, "gcd_wgrad_conv_tr_f16: null pointer");
  CHECK_ARG(M > 0 && N > 0 && Cp > 0 && N % 8 == 0 && Cp % 8 == 0 && lddy % 8 == 0 && lddy >= N && ldx % 8 == 0 && ldx >= Cp,
            "gcd_wgrad_conv_tr_f16: M=%lld N=%d Cp=%d lddy=%lld ldx=%lld", (long long)M, N, Cp, (long long)lddy, (long long)ldx);
  CHECK_ARG(N_real >= 1 && N_real <= N && C_real >= 1 && C_real <= Cp, "gcd_wgrad_conv_tr_f16: N_real=%d C_real=%d", N_real,
            C_real);
  if (conv == 1) CHECK_ARG(Ho > 0 && Wo > 0 && M % ((int64_t)Ho * Wo) == 0, "gcd_wgrad_conv_tr_f16: M=%lld is not frames x %d x %d",
                           (long long)M, Ho, Wo);
  else CHECK_ARG(T > 0 && HW > 0 && M % ((int64_t)T * HW) == 0, "gcd_wgrad_conv_tr_f16: M=%lld is not clips x %d x %d",
                 (long long)M, T, HW);
  CHECK_ARG((((uintptr_t)dy16 | (uintptr_t)x16 | (uintptr_t)dW | (uintptr_t)scratch) & 15) == 0,
            "gcd_wgrad_conv_tr_f16: operands must be 16-byte aligned");
  const int64_t need = gcd_wgrad_tr_scratch_floats(M, N, K);
  CHECK_ARG(scratch_floats >= need, "gcd_wgrad_conv_tr_f16: scratch of %lld floats, need %lld", (long long)scratch_floats,
            (long long)need);
  const gcd_wgrad::Layout lay = {taps, N_real, C_real, accumulate};
  const gcd_wgrad::ConvGeo geo = {conv, Cp, Ho, Wo, T, HW};
  hipStream_t s = (hipStream_t)stream;
  hipError_t e;
  const bool big = wgrad_tm(M, N, K) == 64;
#define GCD_WC(BF, TMV, CV) gcd_wgrad::launch<BF, TMV, CV>(dy16, lddy, x16, ldx, M, N, K, dW, C_real, lay, scratch, s, geo)
  if (conv == 1) e = big ? (bf16 ? GCD_WC(true, 64, 1) : GCD_WC(false, 64, 1)) : (bf16 ? GCD_WC(true, 32, 1) : GCD_WC(false, 32, 1));
  else e = big ? (bf16 ? GCD_WC(true, 64, 2) : GCD_WC(false, 64, 2)) : (bf16 ? GCD_WC(true, 32, 2) : GCD_WC(false, 32, 2));
#undef GCD_WC
  if (e != hipSuccess) {
    set_error("gcd_wgrad_conv_tr_f16: launch failed: %s", hipGetErrorString(e));
    return 1;
  }
  return 0;
}
