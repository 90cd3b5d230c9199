// gemm_common.h — kernel-side view of gcd_gemm_desc shared by the GEMM kernels of libgcd_amd
// (gemm.hip: general 128-row tiles; gemm_pp.hip: 256x320 ping-pong tiles).
#pragma once
#include "common.h"

struct GemmK {
  const f16* A;
  const f16* W;
  void* out;
  int64_t lda, ldo;
  int M, N, K;
  int Cin, Hi, Wi, Ho, Wo, stride, up, T, HW;
  const float* bias;
  const float* rowvec;
  int64_t ld_rowvec;
  int rows_per_vec;
  const float* R1;
  int64_t ldr1;
  const float* R2;
  int64_t ldr2;
  float s_acc, s_r1, s_r2;
  const float* frame_alpha;
  int rows_per_alpha;
  int r1_blend;
  int out_kind;
  const f16* zero;
  int tiles_m, tiles_n;
};

// gemm_pp.hip: the 256 x 320 ping-pong kernel.
bool gcd_gemm_pp_supported(const GemmK& k, int mode);
int gcd_gemm_pp_launch(const GemmK& k, int mode, hipStream_t s);

// runtime.hip: tuning knobs (gcd_tune_set / environment), see include/gcd_amd.h
int gcd_tune_get(int knob);
