// Shared device/host helpers for libgcd_amd (gfx950 only: wave = 64 lanes, MFMA f16, LDS-DMA).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

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

// GELU (erf form) with erfc from Abramowitz-Stegun 7.1.26 (|erf error| <= 1.5e-7, far below the
// fp16 rounding of every consumer): 2 transcendentals + ~12 VALU instead of libm's erff.
//   gelu(g) = g * Phi(g),  Phi(g) = 1 - erfc(|g|/sqrt2)/2 for g >= 0, erfc(|g|/sqrt2)/2 otherwise
__device__ __forceinline__ float gelu_fast(float g) {
  const float x = fabsf(g) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, x, 1.0f));
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  poly *= t;
  const float h = 0.5f * poly * __builtin_amdgcn_exp2f(-x * x * 1.4426950408889634f);
  return g * (g >= 0.f ? 1.0f - h : h);
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
