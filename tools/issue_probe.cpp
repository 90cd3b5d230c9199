// tools/issue_probe.cpp — which instruction streams of the two waves that share a SIMD overlap on gfx950?
// One workgroup of 8 waves per CU (2 per SIMD: waves w and w+4).  Waves 0-3 run stream X, waves 4-7 run
// stream Y (or exit at once), each timing itself with s_memtime; also single-wave interleavings.
//   hipcc -O2 --offload-arch=gfx950 tools/issue_probe.cpp -o tools/issue_probe && tools/issue_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

enum { S_NONE = 0, S_MFMA, S_EXP, S_FMA, S_PKFMA, S_MFMA_EXP2, S_MFMA_FMA7, S_MFMA_EXP1, S_CVT, S_MFMA_DEP, S_V1, S_V2, S_V3, S_V4, S_V5, S_V6, S_V7, S_V8, S_V9, S_W0, S_W1, S_W2, S_W3, S_W4, S_W5, S_BR0, S_BR1, S_BR2 };

template <int KIND>
__device__ __forceinline__ void run_stream(int iters, float seed, float* sink) {
  f16x8 a, b;
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(seed + i); b[i] = (_Float16)(seed * 0.5f + i); }
  f32x16 acc[4];
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[k][i] = 0.f;
  float v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = seed * (i + 1) * 1e-3f;
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  f32x2 pv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) pv[i] = f32x2{seed * i, seed + i};
  for (int it = 0; it < iters; ++it) {
    if (KIND == S_MFMA) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[k], 0, 0, 0);
    } else if (KIND == S_MFMA_DEP) {   // 16 back-to-back dependent MFMAs
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[0], 0, 0, 0);
    } else if (KIND == S_EXP) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = __builtin_amdgcn_exp2f(v[i]);
    } else if (KIND == S_FMA) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(seed));
    } else if (KIND == S_PKFMA) {
#pragma unroll
      for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(pv[i]));
    } else if (KIND == S_CVT) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 8; i += 2) {
          unsigned o;
          asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(o) : "v"(v[i]), "v"(v[i + 1]));
          asm volatile("v_dot2c_f32_f16 %0, %1, %1" : "+v"(v[i]) : "v"(o));
        }
    } else if (KIND == S_MFMA_EXP2) {   // 16 x { MFMA, exp, exp }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc[r & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[r & 3], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        v[(2 * r) & 7] = __builtin_amdgcn_exp2f(v[(2 * r) & 7]);
        v[(2 * r + 1) & 7] = __builtin_amdgcn_exp2f(v[(2 * r + 1) & 7]);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else if (KIND == S_MFMA_EXP1) {   // 16 x { MFMA, exp }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc[r & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[r & 3], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        v[r & 7] = __builtin_amdgcn_exp2f(v[r & 7]);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else if (KIND == S_V1 || KIND == S_V2 || KIND == S_V3) {
      // attention-like group: MFMA, 2 exp, cvt_pk of the previous two exps, dot2c
      // V2: the exponentials read an accumulator block that no MFMA in flight writes (acc[2] / acc[3])
      // V3: additionally the packed result becomes (part of) the next MFMA's B operand
      float e0 = v[0], e1 = v[1];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc[r & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[r & 1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        unsigned o;
        asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(o) : "v"(e0), "v"(e1));
        asm volatile("v_dot2c_f32_f16 %0, %1, %1" : "+v"(v[7]) : "v"(o));
        if (KIND == S_V1) {
          e0 = __builtin_amdgcn_exp2f(v[(2 * r) & 3]);
          e1 = __builtin_amdgcn_exp2f(v[(2 * r + 1) & 3]);
        } else {
          e0 = __builtin_amdgcn_exp2f(acc[2][r]);
          e1 = __builtin_amdgcn_exp2f(acc[3][r]);
        }
        if (KIND == S_V3) {
          typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
          u32x4 bb = __builtin_bit_cast(u32x4, b);
          bb[r & 3] = o;
          b = __builtin_bit_cast(f16x8, bb);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      v[0] = e0; v[1] = e1;
    } else if (KIND == S_V4 || KIND == S_V5 || KIND == S_V6) {
      // V4: like V1 with every dependency two groups apart: cvt(r) packs the exps of group r-2, dot2(r)
      //     sums the pack of group r-1.  V5: V4 without the dot2.  V6: V4 with the order M, e, e, cvt, dot2
      float ea0 = v[0], ea1 = v[1], eb0 = v[2], eb1 = v[3];
      unsigned op = 0;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc[r & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[r & 1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        unsigned o;
        float n0, n1;
        if (KIND == S_V6) {
          n0 = __builtin_amdgcn_exp2f(acc[2][r]);
          n1 = __builtin_amdgcn_exp2f(acc[3][r]);
        }
        if (KIND != S_V5) asm volatile("v_dot2c_f32_f16 %0, %1, %1" : "+v"(v[7]) : "v"(op));
        asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(o) : "v"(eb0), "v"(eb1));
        if (KIND != S_V6) {
          n0 = __builtin_amdgcn_exp2f(acc[2][r]);
          n1 = __builtin_amdgcn_exp2f(acc[3][r]);
        }
        op = o;
        eb0 = ea0; eb1 = ea1; ea0 = n0; ea1 = n1;
        __builtin_amdgcn_sched_barrier(0);
      }
      v[0] = ea0; v[1] = ea1; v[2] = eb0; v[3] = eb1; v[4] += __uint_as_float(op);
    } else if (KIND == S_V7 || KIND == S_V8 || KIND == S_V9) {
      // {MFMA, cvt_pk(prev exps), SUM, exp, exp}: SUM = v_pk_add_f32 (V7), 2 x v_add_f32 (V8), v_dot2_f32_f16 (V9)
      float e0 = v[0], e1 = v[1];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc[r & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[r & 1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        unsigned o;
        asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(o) : "v"(e0), "v"(e1));
        if (KIND == S_V7) {
          f32x2 ee = {e0, e1};
          asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(pv[0]) : "v"(ee));
        } else if (KIND == S_V8) {
          asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[6]) : "v"(e0));
          asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[7]) : "v"(e1));
        } else {
          asm volatile("v_dot2_f32_f16 %0, %1, %1, %0" : "+v"(v[7]) : "v"(o));
        }
        v[5] += __uint_as_float(o) * 0.f;
        e0 = __builtin_amdgcn_exp2f(acc[2][r]);
        e1 = __builtin_amdgcn_exp2f(acc[3][r]);
        __builtin_amdgcn_sched_barrier(0);
      }
      v[0] = e0; v[1] = e1;
    } else if (KIND == S_W0 || KIND == S_W1 || KIND == S_W2) {
      // {MFMA(A = af[r & 3]), exp, exp, add, add}: W0 VALU results go to registers no MFMA reads;
      // W1: the exponentials overwrite the A operand of the MFMA issued in THIS group (dead after it);
      // W2: they overwrite the A operand of the PREVIOUS group's MFMA
      typedef float f32x4 __attribute__((ext_vector_type(4)));
      f32x4 af[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = __builtin_bit_cast(f32x4, a);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[r & 1]) : "v"(af[r & 3]), "v"(b));
        float x0, x1;
        asm volatile("v_exp_f32 %0, %1" : "=v"(x0) : "v"(v[(2 * r) & 3]));
        asm volatile("v_exp_f32 %0, %1" : "=v"(x1) : "v"(v[(2 * r + 1) & 3]));
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[6]) : "v"(x0));
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[7]) : "v"(x1));
        if (KIND == S_W1) {
          asm volatile("v_mul_f32 %0, %1, %1" : "=v"(af[r & 3][0]) : "v"(x0));
          asm volatile("v_mul_f32 %0, %1, %1" : "=v"(af[r & 3][1]) : "v"(x1));
        } else if (KIND == S_W2) {
          asm volatile("v_mul_f32 %0, %1, %1" : "=v"(af[(r + 3) & 3][0]) : "v"(x0));
          asm volatile("v_mul_f32 %0, %1, %1" : "=v"(af[(r + 3) & 3][1]) : "v"(x1));
        } else {
          asm volatile("v_mul_f32 %0, %1, %1" : "=v"(v[4]) : "v"(x0));
          asm volatile("v_mul_f32 %0, %1, %1" : "=v"(v[5]) : "v"(x1));
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] += af[i][0] + af[i][1];
    } else if (KIND == S_W3 || KIND == S_W4 || KIND == S_W5) {
      // the attention kernel's group as emitted: {s_nop, MFMA, exp, exp, add(prev exp), add(prev exp)}
      // W4: exps read an accumulator block written by MFMAs of the PREVIOUS iteration (acc[2], acc[3] get
      //     one MFMA each per iteration, like S^T of the other query block).  W5: W3 + 16 cvt_pk tail.
      float p0 = v[0], p1 = v[1];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        asm volatile("s_nop 0");
        if (KIND == S_W4 && r >= 14)
          asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[r - 12]) : "v"(a), "v"(b));
        else
          asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[r & 1]) : "v"(a), "v"(b));
        float x0, x1;
        if (KIND == S_W4) {
          asm volatile("v_exp_f32 %0, %1" : "=v"(x0) : "v"(acc[2][r]));
          asm volatile("v_exp_f32 %0, %1" : "=v"(x1) : "v"(acc[3][r]));
        } else {
          asm volatile("v_exp_f32 %0, %1" : "=v"(x0) : "v"(v[(2 * r) & 3]));
          asm volatile("v_exp_f32 %0, %1" : "=v"(x1) : "v"(v[(2 * r + 1) & 3]));
        }
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[6]) : "v"(p0));
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[7]) : "v"(p1));
        p0 = x0; p1 = x1;
      }
      if (KIND == S_W5) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          unsigned o;
          asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(o) : "v"(p0), "v"(p1));
          asm volatile("" :: "v"(o));
        }
      }
      v[0] = p0; v[1] = p1;
    } else if (KIND == S_BR0 || KIND == S_BR1 || KIND == S_BR2) {
      // 4 x { 8 fma, a uniform branch around a block of 128 (BR1) / 16 (BR2) fma that is never executed };
      // BR0: the same 32 fma without the branches.  Prices a TAKEN forward branch.
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(seed));
        if (KIND != S_BR0) {
          const int z = __builtin_amdgcn_readfirstlane(iters > (1 << 30) ? 1 : 0);   // 0
          if (KIND == S_BR1)
            asm volatile("s_cmp_eq_u32 %1, 0\n\ts_cbranch_scc1 1f\n\t.rept 128\n\tv_fma_f32 %0, %0, %0, %0\n\t.endr\n1:"
                         : "+v"(v[r]) : "s"(z) : "scc");
          else
            asm volatile("s_cmp_eq_u32 %1, 0\n\ts_cbranch_scc1 1f\n\t.rept 16\n\tv_fma_f32 %0, %0, %0, %0\n\t.endr\n1:"
                         : "+v"(v[r]) : "s"(z) : "scc");
          asm volatile("" ::: "memory");
        }
      }
    } else if (KIND == S_MFMA_FMA7) {   // 16 x { MFMA, 7 fma }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc[r & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[r & 3], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 7; ++i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(seed));
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[k][i];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += v[i];
#pragma unroll
  for (int i = 0; i < 4; ++i) s += pv[i][0] + pv[i][1];
  if (s == 12345.678f) *sink = s;
}

template <int X, int Y>
__global__ __launch_bounds__(512) void probe(int iters, float seed, float* sink, long long* cyc) {
  const int wave = threadIdx.x >> 6;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  if (wave < 4) {
    if (X != S_NONE) run_stream<X>(iters, seed, sink);
  } else {
    if (Y != S_NONE) run_stream<Y>(iters, seed, sink);
  }
  const long long t1 = __builtin_readcyclecounter();
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int X, int Y>
void run(const char* name, int nx, int ny) {   // nx / ny: instructions of interest per iteration
  const int iters = 2000, blocks = 256;
  float* sink; long long* cyc;
  CK(hipMalloc(&sink, 4)); CK(hipMalloc(&cyc, blocks * 8 * sizeof(long long)));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((probe<X, Y>), dim3(blocks), dim3(512), 0, 0, 10, 1.0f, sink, cyc);
  hipEventRecord(e0);
  hipLaunchKernelGGL((probe<X, Y>), dim3(blocks), dim3(512), 0, 0, iters, 1.0f, sink, cyc);
  hipEventRecord(e1);
  CK(hipDeviceSynchronize());
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(blocks * 8);
  CK(hipMemcpy(h.data(), cyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
  double cx = 0, cy = 0;
  for (int b = 0; b < blocks; ++b) { for (int w = 0; w < 4; ++w) cx += h[b * 8 + w]; for (int w = 4; w < 8; ++w) cy += h[b * 8 + w]; }
  cx /= blocks * 4.0 * iters; cy /= blocks * 4.0 * iters;
  // s_memtime counts at 100 MHz on gfx9; report wall time per iteration in ns as well
  printf("%-44s  %8.1f us total | X: %7.2f ticks/iter (%d instr)  Y: %7.2f ticks/iter (%d instr) | %.1f ns/iter\n",
         name, ms * 1e3, cx, nx, cy, ny, ms * 1e6 / iters);
  hipFree(sink); hipFree(cyc);
}

int main() {
  run<S_MFMA, S_NONE>("X=16 MFMA (indep x4)            Y=-", 16, 0);
  run<S_MFMA_DEP, S_NONE>("X=16 MFMA (dependent chain)     Y=-", 16, 0);
  run<S_EXP, S_NONE>("X=32 exp                        Y=-", 32, 0);
  run<S_FMA, S_NONE>("X=32 fma                        Y=-", 32, 0);
  run<S_PKFMA, S_NONE>("X=32 pk_fma                     Y=-", 32, 0);
  run<S_CVT, S_NONE>("X=16 cvt_pk + 16 dot2c          Y=-", 32, 0);
  run<S_MFMA, S_MFMA>("X=16 MFMA                       Y=16 MFMA", 16, 16);
  run<S_EXP, S_EXP>("X=32 exp                        Y=32 exp", 32, 32);
  run<S_FMA, S_FMA>("X=32 fma                        Y=32 fma", 32, 32);
  run<S_MFMA, S_EXP>("X=16 MFMA                       Y=32 exp", 16, 32);
  run<S_MFMA, S_FMA>("X=16 MFMA                       Y=32 fma", 16, 32);
  run<S_MFMA, S_PKFMA>("X=16 MFMA                       Y=32 pk_fma", 16, 32);
  run<S_MFMA, S_CVT>("X=16 MFMA                       Y=cvt+dot2c", 16, 32);
  run<S_MFMA_DEP, S_EXP>("X=16 MFMA dep                   Y=32 exp", 16, 32);
  run<S_MFMA_EXP2, S_NONE>("X=16 x {MFMA, 2 exp}            Y=-", 48, 0);
  run<S_MFMA_EXP1, S_NONE>("X=16 x {MFMA, 1 exp}            Y=-", 32, 0);
  run<S_MFMA_FMA7, S_NONE>("X=16 x {MFMA, 7 fma}            Y=-", 128, 0);
  run<S_MFMA_EXP2, S_MFMA_EXP2>("X=16 x {MFMA, 2 exp}            Y=same", 48, 48);
  run<S_MFMA_EXP1, S_MFMA_EXP1>("X=16 x {MFMA, 1 exp}            Y=same", 32, 32);
  run<S_MFMA_FMA7, S_MFMA_FMA7>("X=16 x {MFMA, 7 fma}            Y=same", 128, 128);
  run<S_V1, S_NONE>("X=16 x {MFMA, cvt, dot2, 2 exp}      Y=-", 80, 0);
  run<S_V1, S_V1>("X=16 x {MFMA, cvt, dot2, 2 exp}      Y=same", 80, 80);
  run<S_V2, S_NONE>("X=V1 with exp(accumulator)           Y=-", 80, 0);
  run<S_V2, S_V2>("X=V1 with exp(accumulator)           Y=same", 80, 80);
  run<S_V3, S_NONE>("X=V2 + packed P feeds next MFMA B    Y=-", 80, 0);
  run<S_V3, S_V3>("X=V2 + packed P feeds next MFMA B    Y=same", 80, 80);
  run<S_V4, S_NONE>("X=V4 deps two groups apart           Y=-", 80, 0);
  run<S_V4, S_V4>("X=V4 deps two groups apart           Y=same", 80, 80);
  run<S_V5, S_NONE>("X=V5 = V4 without dot2               Y=-", 64, 0);
  run<S_V6, S_NONE>("X=V6 = V4, order M e e dot2 cvt      Y=-", 80, 0);
  run<S_V7, S_NONE>("X={M, cvt, pk_add_f32, fma, e, e}    Y=-", 96, 0);
  run<S_V8, S_NONE>("X={M, cvt, 2 add_f32, fma, e, e}     Y=-", 112, 0);
  run<S_V9, S_NONE>("X={M, cvt, dot2_f32_f16, fma, e, e}  Y=-", 96, 0);
  run<S_V7, S_V7>("X={M, cvt, pk_add_f32, fma, e, e}    Y=same", 96, 96);
  run<S_W0, S_NONE>("X={M, e, e, add, add, mul, mul} dst free         Y=-", 112, 0);
  run<S_W1, S_NONE>("X=... dst = A operand of this group's MFMA      Y=-", 112, 0);
  run<S_W2, S_NONE>("X=... dst = A operand of previous group's MFMA  Y=-", 112, 0);
  run<S_W3, S_NONE>("X={nop, M, e, e, add(prev), add(prev)}         Y=-", 96, 0);
  run<S_W4, S_NONE>("X=W3, exps read last iteration's MFMA results  Y=-", 96, 0);
  run<S_W5, S_NONE>("X=W3 + 16 cvt_pk tail                          Y=-", 112, 0);
  run<S_W3, S_W3>("X={nop, M, e, e, add(prev), add(prev)}         Y=same", 96, 96);
  run<S_BR0, S_NONE>("X=4 x 8 fma, no branches                              Y=-", 32, 0);
  run<S_BR1, S_NONE>("X=4 x {8 fma, branch over 128 dead instructions}      Y=-", 32, 0);
  run<S_BR2, S_NONE>("X=4 x {8 fma, branch over 16 dead instructions}       Y=-", 32, 0);
  run<S_BR1, S_BR1>("X=4 x {8 fma, branch over 128 dead instructions}      Y=same", 32, 32);
  return 0;
}
