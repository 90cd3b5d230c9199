// Shared device/host helpers for libgcd_amd (gfx950 only: wave = 64 lanes, MFMA f16, LDS-DMA).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include <atomic>
#include <mutex>

#include "../../include/gcd_amd.h"

typedef _Float16 f16;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define GCD_AS1 __attribute__((address_space(1)))
#define GCD_AS3 __attribute__((address_space(3)))

// ---- error plumbing (host) ---------------------------------------------------------------------
void gcd_set_error(const char* fmt, ...);

#define GCD_CHECK_ARG(cond, ...)      \
  do {                                \
    if (!(cond)) {                    \
      gcd_set_error(__VA_ARGS__);     \
      return 2;                       \
    }                                 \
  } while (0)

#define GCD_CHECK_HIP(expr)                                                              \
  do {                                                                                   \
    hipError_t _e = (expr);                                                              \
    if (_e != hipSuccess) {                                                              \
      gcd_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__,     \
                    __LINE__);                                                           \
      return 1;                                                                          \
    }                                                                                    \
  } while (0)

#define GCD_CHECK_LAUNCH() GCD_CHECK_HIP(hipGetLastError())

// hipFuncSetAttribute (the > 64 KB dynamic-LDS opt-in) is a per-DEVICE property of a kernel: a
// launcher keeps one of these (static) and sets the attribute the first time it runs on each device
// ordinal, so a process that drives several GPUs does not launch un-opted kernels on the second one.
struct GcdPerDeviceOnce {
  std::atomic<unsigned long long> done[4];
  std::mutex mu;
  GcdPerDeviceOnce() {
    for (auto& w : done) w.store(0, std::memory_order_relaxed);
  }
  // hipFuncSetAttribute(fn, MaxDynamicSharedMemorySize, bytes) once per device ordinal.  Safe for several
  // host threads (one per GPU is the normal multi-GPU setup): the bit of a device is published only AFTER
  // its attribute call has returned, and late arrivals wait on the mutex instead of launching early.
  hipError_t opt_in(const void* fn, int bytes) {
    int d = 0;
    hipError_t e = hipGetDevice(&d);
    if (e != hipSuccess) return e;
    if (d < 0 || d >= 256)   // beyond the table: no memo, set it every time
      return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    const unsigned long long bit = 1ull << (d & 63);
    std::atomic<unsigned long long>& w = done[d >> 6];
    if (w.load(std::memory_order_acquire) & bit) return hipSuccess;
    std::lock_guard<std::mutex> g(mu);
    if (w.load(std::memory_order_relaxed) & bit) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) w.fetch_or(bit, std::memory_order_release);
    return e;
  }
};

// ---- device helpers ----------------------------------------------------------------------------
// Asynchronous 16-byte global -> LDS copy (global_load_lds_dwordx4).  The LDS destination is
// wave-uniform base + lane*16, the global source address is per lane.
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const GCD_AS1 void*)gsrc, (GCD_AS3 void*)lds_wave_base, 16, 0,
                                   0);
}

// LDS tile of ROWS x 64 fp16 (128 B rows, eight 16-B chunks per row).  Logical chunk c of row r
// lives at physical chunk c ^ ((r >> 1) & 7): a 16-lane ds_read_b128 group that reads one chunk
// column of 16 consecutive rows then touches 16 distinct 16-B slots of the 256-B bank row.
__device__ __forceinline__ int lds_tile_off(int row, int chunk) {
  return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
}

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }

// exact (erf) GELU, as torch.nn.functional.gelu default
__device__ __forceinline__ float gelu_f(float x) {
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}

// GELU (erf form): g * Phi(g) with the normal CDF as an odd minimax polynomial on the clamped
// argument, Phi(x) = 1/2 + xc * P(u), xc = clamp(x, -4.5, 4.5), u = 2 xc^2 / 4.5^2 - 1, degree-9 P
// (Lawson fit, |Phi error| <= 7.1e-6 including the clamp, see tools/fit_gelu.py).  14 full-rate VALU
// instructions and no transcendental: the GEGLU epilogues are instruction-issue bound, libm's erff
// (~40 instructions) and even a 2-transcendental Abramowitz-Stegun form cost 2-3x as much.
__device__ __forceinline__ float gelu_fast(float g) {
  const float xc = __builtin_amdgcn_fmed3f(g, -4.5f, 4.5f);
  const float u = fmaf(xc * xc, 2.0f / 20.25f, -1.0f);
  float p = -1.928577467e-03f;
  p = fmaf(p, u, 4.953202455e-03f);
  p = fmaf(p, u, -6.067269522e-03f);
  p = fmaf(p, u, 9.667675117e-03f);
  p = fmaf(p, u, -1.849089463e-02f);
  p = fmaf(p, u, 2.885118603e-02f);
  p = fmaf(p, u, -4.023398011e-02f);
  p = fmaf(p, u, 5.463833202e-02f);
  p = fmaf(p, u, -7.718616753e-02f);
  p = fmaf(p, u, 1.569060299e-01f);
  return g * fmaf(xc, p, 0.5f);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// fp32 -> 16-bit operand, as the bit pattern of an f16 slot: fp16 (round to nearest even, the hardware conversion) or, when
// `bf16`, bfloat16 (round to nearest even; NaN stays NaN).  ABI v7: the normalisation / GEGLU kernels write the fine-tune
// step's bfloat16 GEMM operands DIRECTLY (one rounding from fp32, no fp16 hop and none of fp16's range).
#ifdef __HIPCC__
__device__ __forceinline__ _Float16 gcd_cvt16(float v, bool bf16) {
  if (bf16) {
    unsigned u = __float_as_uint(v);
    unsigned short r;
    if ((u & 0x7fffffffu) > 0x7f800000u) r = (unsigned short)((u >> 16) | 0x40);
    else r = (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
    return __builtin_bit_cast(_Float16, r);
  }
  return (_Float16)v;
}
#endif
