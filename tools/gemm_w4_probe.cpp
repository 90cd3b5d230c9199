// tools/gemm_w4_probe.cpp — round-2 experiment: is a ONE-WAVE-PER-SIMD K loop faster than gemm_pp's?
//
// gemm_pp (gcd_amd/csrc/gemm_pp.hip) runs 8 waves of 64 x 160 per 256 x 320 tile; its K loop is bound by
// the LDS port (148 KB per 32-deep step against the 164 KB the port moves in the step's 1280 MFMA
// cycles).  This probe keeps the tile, the 4-slot LDS-DMA ring and the bank swizzle but
// gives the tile to FOUR waves of 128 x 160 (2 x 2), one per SIMD, 320 accumulator registers each:
// 72 KB of fragment reads + 36 KB of DMA per step (-27 % LDS bytes per FLOP), one barrier per 40
// MFMAs instead of two per 10, fragments of the next half-step prefetched under the current one's MFMAs.
// Plain GEMM, fp32 output straight from the accumulator layout (no epilogue work): K-loop rate only.
//   hipcc -O3 --offload-arch=gfx950 tools/gemm_w4_probe.cpp -o tools/gemm_w4_probe && tools/gemm_w4_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <type_traits>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef _Float16 f16;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define AS1 __attribute__((address_space(1)))
#define AS3 __attribute__((address_space(3)))

__device__ __forceinline__ void glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const AS1 void*)g, (AS3 void*)l, 16, 0, 0);
}
#define VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define BAR() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)

constexpr int BN = 320;
constexpr int W_BYTES = BN * 64;

struct P {
  const f16* A;
  const f16* W;
  float* C;
  int M, N, K, tiles_m, tiles_n;
};

// VARIANT 0: fragments of a whole 32-deep step read up front (no prefetch across MFMAs)
// VARIANT 1: half-step ping-pong: the reads of the next half-step are issued before this one's MFMAs
// TM: 32-row blocks per wave (4: 128 x 160 wave tiles, 256-row workgroup tile, 320 accumulator registers —
// more than the 256 AGPRs: hipcc then shuttles accumulators between the files around every MFMA;
// 3: 96 x 160, 192-row tile, 240 accumulator registers)
template <int VARIANT, int TM>
__global__ __launch_bounds__(256, 1) void gemm_w4(const P p) {
  constexpr int BM = 64 * TM, A_BYTES = BM * 64, SLOT = A_BYTES + W_BYTES;
#define VM_TWO() do { if constexpr (TM == 4) { VMCNT(18); } else { VMCNT(16); } } while (0)
#define VM_ONE() do { if constexpr (TM == 4) { VMCNT(9); } else { VMCNT(8); } } while (0)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, hh = lane >> 5;
  int L;
  {
    const int nblk = gridDim.x, bid = blockIdx.x;
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  int tile_m, tile_n;
  {
    const int per_group = 4 * p.tiles_n;
    const int gi = L / per_group, rem = L - gi * per_group;
    const int gm = min(4, p.tiles_m - gi * 4);
    tile_n = rem / gm;
    tile_m = gi * 4 + rem - tile_n * gm;
  }
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int S = p.K >> 5;
  // ---- DMA: a piece = 16 rows x 64 B; wave w stages A rows 64w..64w+63 (4 pieces), W rows 80w..80w+79 (5)
  const int lrow = lane >> 2;
  const int lc16 = ((lane & 3) ^ ((lane >> 4) & 3)) << 4;
  const char* a_src[TM];
  const char* w_src[5];
#pragma unroll
  for (int q = 0; q < TM; ++q) {
    int m = m0 + 16 * TM * wave + 16 * q + lrow;
    m = m < p.M ? m : p.M - 1;
    a_src[q] = (const char*)p.A + (int64_t)m * p.K * 2 + lc16;
  }
#pragma unroll
  for (int q = 0; q < 5; ++q) {
    int n = n0 + 80 * wave + 16 * q + lrow;
    n = n < p.N ? n : p.N - 1;
    w_src[q] = (const char*)p.W + (int64_t)n * p.K * 2 + lc16;
  }
  auto issue = [&](int s) {
    if (s < S) {
      char* dst = smem + (s & 3) * SLOT;
#pragma unroll
      for (int q = 0; q < TM; ++q) glds16(a_src[q] + s * 64, dst + (16 * TM * wave + 16 * q) * 64);
#pragma unroll
      for (int q = 0; q < 5; ++q) glds16(w_src[q] + s * 64, dst + A_BYTES + (80 * wave + 16 * q) * 64);
    }
  };
  // ---- fragment addresses
  const int swz = (l31 >> 2) & 3;
  int rdA[2], rdW[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const int ch = ((2 * ks + hh) ^ swz) << 4;
    rdA[ks] = (32 * TM * wm + l31) * 64 + ch;
    rdW[ks] = A_BYTES + (160 * wn + l31) * 64 + ch;
  }
  f32x16 acc[5][TM];
#pragma unroll
  for (int i = 0; i < 5; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  f16x8 af[2][TM], wf[2][5];
  auto load = [&](int set, int slot, int ks) {
    const char* base = smem + slot * SLOT;
#pragma unroll
    for (int j = 0; j < TM; ++j) af[set][j] = *(const f16x8*)(base + rdA[ks] + j * 2048);
#pragma unroll
    for (int i = 0; i < 5; ++i) wf[set][i] = *(const f16x8*)(base + rdW[ks] + i * 2048);
  };
  auto mma = [&](int set) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
      for (int j = 0; j < TM; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[set][i], af[set][j], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };

  issue(0);
  issue(1);
  issue(2);
  if (S > 2) {
    VM_TWO();
  } else if (S > 1) {
    VM_ONE();
  } else {
    VMCNT(0);
  }
  BAR();
  if (VARIANT == 0) {
    for (int s = 0; s < S; ++s) {
      const int slot = s & 3;
      issue(s + 3);
      load(0, slot, 0);
      load(1, slot, 1);
      __builtin_amdgcn_sched_barrier(0);
      mma(0);
      mma(1);
      __builtin_amdgcn_sched_barrier(0);
      if (s + 3 < S) {
        VM_TWO();
      } else if (s + 2 < S) {
        VM_ONE();
      } else {
        VMCNT(0);
      }
      LGKM0();
      BAR();
    }
  } else {
    // hipcc's waitcnt pass cannot see through the inline-asm waits: it puts `s_waitcnt lgkmcnt(0)` in front
    // of the first MFMA that uses fragments read in the previous phase — and that wait also covers any
    // ds_read issued since.  So every phase is: first MFMA (its wait finds nothing outstanding), THEN the
    // reads of the next half-step, then the other MFMAs.
    // One instruction stream per SIMD: nothing hides an instruction's issue time but the MFMA that is
    // executing.  An LDS-DMA piece costs ~60-100 cycles of issue, a ds_read_b128 ~16: they are dealt out
    // between the MFMAs (one read after each of the first 8, one DMA piece after every other MFMA) and
    // pinned there with sched_barrier, instead of the compiler's "all loads first".
    auto frag_read = [&](int set, int slot, int ks, int idx) {   // idx 0..TM-1: A block, TM..TM+4: W block
      const char* base = smem + slot * SLOT;
      if (idx < TM) af[set][idx] = *(const f16x8*)(base + rdA[ks] + idx * 2048);
      else wf[set][idx - TM] = *(const f16x8*)(base + rdW[ks] + (idx - TM) * 2048);
    };
    auto piece = [&](int s, int q) {   // q 0..TM-1: A pieces, TM..TM+4: W pieces of this wave (s < S)
      char* dst = smem + (s & 3) * SLOT;
      if (q < TM) glds16(a_src[q] + s * 64, dst + (16 * TM * wave + 16 * q) * 64);
      else glds16(w_src[q - TM] + s * 64, dst + A_BYTES + (80 * wave + 16 * (q - TM)) * 64);
    };
    constexpr int NF = TM + 5, NMMA = 5 * TM;
    // FULL: steady state (slot s+3 exists: DMA it, two slots stay in flight); tail otherwise
    auto body = [&](auto full, int s) {
      constexpr bool FULL = decltype(full)::value;
      const int slot = s & 3;
      // ---- half-step 0: MFMAs on set 0; reads of (slot, ks 1) into set 1; DMA of slot s+3 ----
#pragma unroll
      for (int k = 0; k < NMMA; ++k) {
        const int i = k / TM, j = k - i * TM;
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[0][i], af[0][j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (k < NF) frag_read(1, slot, 1, k);
        if (FULL && k >= 1 && k - 1 < NF) piece(s + 3, k - 1);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (FULL) {
        VM_TWO();
      } else if (s + 2 < S) {
        VM_ONE();
      } else {
        VMCNT(0);
      }
      LGKM0();
      BAR();
      // ---- half-step 1: MFMAs on set 1; reads of (slot s+1, ks 0) into set 0 ----
#pragma unroll
      for (int k = 0; k < NMMA; ++k) {
        const int i = k / TM, j = k - i * TM;
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[1][i], af[1][j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (k < NF && (FULL || s + 1 < S)) frag_read(0, (s + 1) & 3, 0, k);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    load(0, 0, 0);
    int s = 0;
    for (; s + 3 < S; ++s) body(std::true_type{}, s);
    for (; s < S; ++s) body(std::false_type{}, s);
  }
  // ---- plain fp32 stores from the accumulator layout ----
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    const int m = m0 + 32 * TM * wm + 32 * j + l31;
    if (m >= p.M) continue;
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = n0 + 160 * wn + 32 * i + 8 * g + 4 * hh;
        if (n >= p.N) continue;
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g + e];
        *(f32x4*)(p.C + (int64_t)m * p.N + n) = v;
      }
  }
}

__global__ void fill(f16* p, size_t n, uint32_t seed, float scale) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t x = (uint32_t)i * 2654435761u + seed;
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    p[i] = (f16)(((float)(x & 0xffff) / 32768.0f - 1.0f) * scale);
  }
}

template <int VARIANT, int TM>
static double run(P p, int iters) {
  constexpr int BM = 64 * TM, SMEM = 4 * (BM * 64 + W_BYTES);
  p.tiles_m = (p.M + BM - 1) / BM;
  const int nblk = p.tiles_m * p.tiles_n;
  static bool set = false;
  if (!set) {
    CK(hipFuncSetAttribute((const void*)gemm_w4<VARIANT, TM>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
    set = true;
  }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<float> t;
  for (int it = 0; it < iters + 1; ++it) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((gemm_w4<VARIANT, TM>), dim3(nblk), dim3(256), SMEM, 0, p);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (it) t.push_back(ms);
  }
  CK(hipGetLastError());
  std::sort(t.begin(), t.end());
  return t[t.size() / 2] * 1e-3;
}

int main() {
  struct Sh { const char* name; int M, N, K; };
  const Sh shapes[] = {{"L2 FF out", 16128, 1280, 5120}, {"L2 conv-like K=11520", 16128, 1280, 11520},
                       {"L2 GEGLU-like", 16128, 10240, 1280}, {"L1 K=5760", 64512, 640, 5760},
                       {"L0 K=320 N=2560", 258048, 2560, 320}, {"L0 K=1280", 258048, 320, 1280},
                       {"check small", 768, 640, 192}};
  for (const Sh& s : shapes) {
    f16 *A, *W;
    float* C;
    CK(hipMalloc(&A, (size_t)s.M * s.K * 2));
    CK(hipMalloc(&W, (size_t)s.N * s.K * 2));
    CK(hipMalloc(&C, (size_t)s.M * s.N * 4));
    fill<<<1024, 256>>>(A, (size_t)s.M * s.K, 1u, 1.0f);
    fill<<<1024, 256>>>(W, (size_t)s.N * s.K, 2u, 1.0f / sqrtf((float)s.K));
    P p{A, W, C, s.M, s.N, s.K, 0, (s.N + BN - 1) / BN};
    const double fl = 2.0 * s.M * s.N * s.K;
    const double t0 = run<1, 4>(p, 5), t1 = run<1, 3>(p, 5);
    // host check of sampled entries (variant 1 output is in C)
    std::vector<f16> hA((size_t)s.M * s.K), hW((size_t)s.N * s.K);
    double maxerr = 0;
    if ((size_t)s.M * s.K < (size_t)400e6) {
      CK(hipMemcpy(hA.data(), A, hA.size() * 2, hipMemcpyDeviceToHost));
      CK(hipMemcpy(hW.data(), W, hW.size() * 2, hipMemcpyDeviceToHost));
      for (int k = 0; k < 64; ++k) {
        const int m = (int)(((uint64_t)k * 2654435761u) % s.M), n = (int)(((uint64_t)k * 40503u + 17) % s.N);
        double ref = 0;
        for (int kk = 0; kk < s.K; ++kk) ref += (double)hA[(size_t)m * s.K + kk] * (double)hW[(size_t)n * s.K + kk];
        float got;
        CK(hipMemcpy(&got, C + (size_t)m * s.N + n, 4, hipMemcpyDeviceToHost));
        maxerr = std::max(maxerr, fabs(ref - got));
      }
    } else {
      maxerr = -1;
    }
    printf("%-22s M=%6d N=%5d K=%5d | 128x160 waves %8.1f us %7.1f TF/s | 96x160 waves %8.1f us %7.1f TF/s | max err %.2e\n",
           s.name, s.M, s.N, s.K, t0 * 1e6, fl / t0 / 1e12, t1 * 1e6, fl / t1 / 1e12, maxerr);
    CK(hipFree(A)); CK(hipFree(W)); CK(hipFree(C));
  }
  return 0;
}
