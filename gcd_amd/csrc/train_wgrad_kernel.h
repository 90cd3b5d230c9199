// train_wgrad_kernel.h — the device code of gcd_wgrad_tr_f16 (train_wgrad.hip), in a header of its own so that
// tools/gemm_tr_probe.cpp compiles and fp64-checks EXACTLY the kernels the library launches.
#pragma once
#include <atomic>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

typedef _Float16 f16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef short v4s __attribute__((__vector_size__(4 * sizeof(short))));
#define GCD_AS3 __attribute__((address_space(3)))
#ifndef GCD_WGRAD_TM_DEFAULT
#define GCD_WGRAD_TM_DEFAULT 32   // tokens per step the library launches with (train_wgrad.hip)
#endif

namespace gcd_wgrad {

constexpr int TN = 128, TK = 128;              // output tile
// LDS rows of 128 16-bit elements (TN == TK).  Round 5: 256-byte rows, the eight 32-byte segments of a row XOR-swizzled by
// the row, instead of 272-byte padded rows.  ds_read_b64_tr_b16 is serviced in groups of 32 lanes over 64 banks of 4 B
// (MI355X_MICROARCH.md, LDS): a group is two 4-row x 32-byte blocks (rows r .. r + 3 and r + 8 .. r + 11 of one 16-column
// block).  With 272-byte rows consecutive rows sit 16 bytes apart in bank space, so every block overlaps itself 2-way —
// every transposing read took 2x its LDS cycles in a kernel that is LDS-bound (0.5 KB of fragments per MFMA).  No uniform
// row pitch separates all eight row segments (8 rows further is always +0 or +128 bytes); the swizzle
//     segment' = segment ^ ((row & 3) | ((row >> 3) & 1) << 2)
// gives the eight rows of a group eight different segments: conflict-free.  GCD_WGRAD_PAD=1 keeps the padded layout (A/B).
#ifndef GCD_WGRAD_PAD
#define GCD_WGRAD_PAD 0
#endif
constexpr int PITCH = GCD_WGRAD_PAD ? TN * 2 + 16 : TN * 2;
// Round 6: a second tile shape, 160 x 160 (4 waves of 80 x 80 = 5 x 5 accumulator blocks).  Every width of the UNet is a
// multiple of 320 = 2.5 x 128: on 128-wide tiles a 320 x 320 weight gradient computes 384 x 384 (1.44x), 2560 x 320 computes
// 2560 x 384 (1.2x) — and those are the gradients of the 43 008-token level, the expensive ones; 160 divides every width.
// A wave's 80 x 80 tile also reads 10 fragments per 25 MFMAs (0.4) where 64 x 64 reads 8 per 16 (0.5).  Its rows are 320
// bytes + 16 of padding (ten 32-byte segments do not take the XOR swizzle).
template <int TILE>
struct Geo {
  static constexpr int pitch = TILE == 128 ? PITCH : TILE * 2 + 16;
  static constexpr bool swz = TILE == 128 && !GCD_WGRAD_PAD;
};
__device__ __forceinline__ int swz_seg(int row) { return GCD_WGRAD_PAD ? 0 : ((row & 3) | (((row >> 3) & 1) << 2)); }
// byte offset of the 16-byte chunk `chunk` of tile row `row`
template <int TILE = 128>
__device__ __forceinline__ int lds_chunk_off(int row, int chunk) {
  if constexpr (Geo<TILE>::swz) return row * Geo<TILE>::pitch + ((((chunk >> 1) ^ swz_seg(row)) << 5) | ((chunk & 1) << 4));
  else return row * Geo<TILE>::pitch + chunk * 16;
}
// LDS of a launch with TM tokens per step: [buffer][operand][TM rows]
template <int TILE = 128>
constexpr int smem_bytes(int TM) { return 2 * 2 * TM * Geo<TILE>::pitch; }

// 8 consecutive tokens (rows r0 + 8 g .. + 7 of the step's tile, g = lane >> 4) of column c0 + (lane & 15): two
// transposing reads of 4 rows each.  Lane i of a 16-lane group supplies the address of row i / 4, columns 4 (i % 4) .. + 3
// of the 4 x 16 block; it receives column i (profiles/r04_probe_ds_read_tr_b16.txt).
template <int TILE = 128>
__device__ __forceinline__ f16x8 frag_tr(const char* tile, int r0, int c0, int lane) {
  constexpr int P = Geo<TILE>::pitch;
  const int g = lane >> 4, i = lane & 15;
  const int row = r0 + 8 * g + (i >> 2);          // (row + 4 has the same swizzle: bits 0-1 and bit 3 are unchanged)
  const char* p = tile + row * P + (Geo<TILE>::swz ? ((((c0 >> 4) ^ swz_seg(row)) << 5) | ((i & 3) << 3)) : (c0 * 2 + ((i & 3) << 3)));
  const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((GCD_AS3 v4s*)(p));
  const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((GCD_AS3 v4s*)(p + 4 * P));
  const f16x4 l4 = __builtin_bit_cast(f16x4, lo), h4 = __builtin_bit_cast(f16x4, hi);
  return (f16x8){l4[0], l4[1], l4[2], l4[3], h4[0], h4[1], h4[2], h4[3]};
}

// grid (ceil(K / TK), ceil(N / TN), S): workgroup (kt, nt, s) accumulates tokens [s * mper, (s + 1) * mper) into
// part[s][N][K].  128 x 128 output tile, 4 waves of 64 x 64 (16 accumulators of v_mfma_f32_16x16x32), TM tokens per
// step (32 or 64: one barrier per 16 / 32 MFMAs of a wave), operands through registers into a double-buffered LDS tile.
// The 16-bit payload travels as f16x8 bit patterns; BF16 only selects the MFMA.  Dynamic LDS: smem_bytes(TM).
// CONV (round 5): the X operand of a convolution's weight gradient read IMPLICITLY — column k = tap * Cp + c of the
// contraction is channel c of the input token that tap `tap` pairs with output token m (zero outside the image / clip) —
// instead of from an im2col'd copy (43 008 x 2880 fp16 = 248 MB written and re-read per L0 convolution).
//   1: Conv2d 3x3, stride 1, same size (rows are (frame, y, x); tap = 3 kh + kw pairs m with m + (kh - 1) Wo + (kw - 1))
//   2: Conv3d (3,1,1) over the T frames of a clip (rows are (clip, t, hw); tap kt pairs m with m + (kt - 1) HW)
struct ConvGeo {
  int conv, Cp, Ho, Wo, T, HW;
};

template <bool BF16, int TM, int CONV = 0, int TILE = 128>
__global__ __launch_bounds__(256) void wgrad_tr_kernel(const f16* __restrict__ dY, int64_t lddy,
                                                       const f16* __restrict__ X, int64_t ldx,
                                                       float* __restrict__ part, int64_t M, int N, int K, int64_t mper,
                                                       ConvGeo geo) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  static_assert(TILE == 128 || TILE == 160, "output tile");
  constexpr int P = Geo<TILE>::pitch, WT = TILE / 2, NB = WT / 16;     // wave tile WT x WT = NB x NB accumulator blocks
  constexpr int TILE_BYTES = TM * P, NP = TM / 16;     // NP: 16-row staging passes per operand (chunks 0 .. 15 of a row)
  constexpr bool XC = TILE == 160;                     // chunks 16 .. 19 of a row: thread t stages row t / 4, chunk 16 + t % 4
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wn = wave >> 1, wk = wave & 1;
  const int n0 = blockIdx.y * TILE, k0 = blockIdx.x * TILE;
  const int64_t m_begin = (int64_t)blockIdx.z * mper;
  int64_t m_end = m_begin + mper;
  if (m_end > M) m_end = M;
  const int nsteps = m_end > m_begin ? (int)((m_end - m_begin + TM - 1) / TM) : 0;
  const int srow = t >> 4, schunk = t & 15;      // staging: row t / 16 (+ 16 per pass), 16-byte chunk t % 16 of a tile row
  const int xrow = t >> 2, xchunk = 16 + (t & 3);
  const bool xact = XC && xrow < TM;
  f16x8 ra[NP], rb[NP], rax = {0, 0, 0, 0, 0, 0, 0, 0}, rbx = rax;
  // CONV: a thread's 16-byte chunk of the K-tile belongs to ONE tap (Cp % 8 == 0): its row shift and channel offset are
  // fixed; the image / clip position of each of its rows is a running counter advanced by TM per step (gload is called
  // for steps 0, 1, 2, ... in order), so no division sits in the loop.
  struct Tap {
    int tap, ch;
    int64_t shift;
  };
  auto tap_of = [&](int chunk) {
    Tap q = {0, 0, 0};
    if constexpr (CONV != 0) {
      const int k = k0 + 8 * chunk;
      q.tap = k / geo.Cp;
      q.ch = k - q.tap * geo.Cp;
      if (CONV == 1) q.shift = (int64_t)(q.tap / 3 - 1) * geo.Wo + (q.tap % 3 - 1);
      else q.shift = (int64_t)(q.tap - 1) * geo.HW;
    }
    return q;
  };
  auto pos_of = [&](int64_t m, int& pa, int& pb) {
    if (CONV == 1) {       // pa = y, pb = x of the output token
      const int64_t r = m % ((int64_t)geo.Ho * geo.Wo);
      pa = (int)(r / geo.Wo);
      pb = (int)(r - (int64_t)pa * geo.Wo);
    } else {               // pa = frame within the clip, pb = pixel
      const int64_t f = m / geo.HW;
      pa = (int)(f % geo.T);
      pb = (int)(m - f * geo.HW);
    }
  };
  const Tap tq = tap_of(schunk), tx = tap_of(xchunk);
  int p_a[NP], p_b[NP], p_ax = 0, p_bx = 0;
  if constexpr (CONV != 0) {
#pragma unroll
    for (int h = 0; h < NP; ++h) pos_of(m_begin + srow + 16 * h, p_a[h], p_b[h]);
    if constexpr (XC) pos_of(m_begin + xrow, p_ax, p_bx);
  }
  // one 16-byte chunk of each operand: row m of the step, tile chunk `chunk` (8 columns of dY / of the contraction)
  auto load_pair = [&](int64_t m, int chunk, const Tap& q, int& pa, int& pb, f16x8& va, f16x8& vb) {
    const int n = n0 + 8 * chunk, k = k0 + 8 * chunk;
    const f16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    va = (m < m_end && n < N) ? *(const f16x8*)(dY + m * lddy + n) : z;
    if constexpr (CONV == 0) {
      vb = (m < m_end && k < K) ? *(const f16x8*)(X + m * ldx + k) : z;
    } else {
      bool ok = m < m_end && k < K;
      if (CONV == 1) {
        const int yy = pa + q.tap / 3 - 1, xx = pb + q.tap % 3 - 1;
        ok = ok && (unsigned)yy < (unsigned)geo.Ho && (unsigned)xx < (unsigned)geo.Wo;
        pb += TM;                                   // this row's position at the next step
        while (pb >= geo.Wo) {
          pb -= geo.Wo;
          if (++pa == geo.Ho) pa = 0;
        }
      } else {
        const int tt = pa + q.tap - 1;
        ok = ok && (unsigned)tt < (unsigned)geo.T;
        pb += TM;
        while (pb >= geo.HW) {
          pb -= geo.HW;
          if (++pa == geo.T) pa = 0;
        }
      }
      vb = ok ? *(const f16x8*)(X + (m + q.shift) * ldx + q.ch) : z;
    }
  };
  auto gload = [&](int step) {
#pragma unroll
    for (int h = 0; h < NP; ++h)
      load_pair(m_begin + (int64_t)step * TM + srow + 16 * h, schunk, tq, p_a[h], p_b[h], ra[h], rb[h]);
    if constexpr (XC) {
      if (xact) load_pair(m_begin + (int64_t)step * TM + xrow, xchunk, tx, p_ax, p_bx, rax, rbx);
    }
  };
  auto lstore = [&](int buf) {
    char* a = smem + buf * 2 * TILE_BYTES;
    char* b = a + TILE_BYTES;
#pragma unroll
    for (int h = 0; h < NP; ++h) {
      *(f16x8*)(a + lds_chunk_off<TILE>(srow + 16 * h, schunk)) = ra[h];
      *(f16x8*)(b + lds_chunk_off<TILE>(srow + 16 * h, schunk)) = rb[h];
    }
    if constexpr (XC) {
      if (xact) {
        *(f16x8*)(a + lds_chunk_off<TILE>(xrow, xchunk)) = rax;
        *(f16x8*)(b + lds_chunk_off<TILE>(xrow, xchunk)) = rbx;
      }
    }
  };
  f32x4 acc[NB][NB];
#pragma unroll
  for (int i = 0; i < NB; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (nsteps > 0) {
    gload(0);
    lstore(0);
  }
  __syncthreads();
  for (int s = 0; s < nsteps; ++s) {
    if (s + 1 < nsteps) gload(s + 1);                 // in flight under this step's MFMAs
    const char* a = smem + (s & 1) * 2 * TILE_BYTES;
    const char* b = a + TILE_BYTES;
#pragma unroll
    for (int ks = 0; ks < TM / 32; ++ks) {
      f16x8 fa[NB], fb[NB];
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        fa[i] = frag_tr<TILE>(a, 32 * ks, WT * wn + 16 * i, lane);
        fb[i] = frag_tr<TILE>(b, 32 * ks, WT * wk + 16 * i, lane);
      }
#pragma unroll
      for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          if constexpr (BF16)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[i]),
                                                                __builtin_bit_cast(bf16x8, fb[j]), acc[i][j], 0, 0, 0);
          else
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
    }
    if (s + 1 < nsteps) lstore((s + 1) & 1);          // the other buffer: its last readers passed the barrier below
    __syncthreads();
  }
  // accumulator block (i, j): C[n = 4 (lane >> 4) + e][k = lane & 15]
  float* out = part + (int64_t)blockIdx.z * N * K;
#pragma unroll
  for (int i = 0; i < NB; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int n = n0 + WT * wn + 16 * i + 4 * (lane >> 4) + e;
        const int k = k0 + WT * wk + 16 * j + (lane & 15);
        if (n < N && k < K) out[(int64_t)n * K + k] = acc[i][j][e];
      }
}

// Destination layout of the folded gradient (gcd_wgrad_tr_f16_ex): the parameter's own shape [N_real][C_real][taps].
struct Layout {
  int taps, N_real, C_real, accumulate;
};

// dW[n][k] = sum over the S slices; K % 4 == 0, taps == 1 (rows of lddw floats; cropped to N_real x C_real, C_real % 4 == 0)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dW,
                                                           int64_t lddw, int N, int K, int S, Layout lay) {
  const int64_t idx = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  const int64_t NK = (int64_t)N * K;
  if (idx >= NK) return;
  const int64_t n = idx / K;
  const int k = (int)(idx - n * K);
  if (n >= lay.N_real || k >= lay.C_real) return;
  f32x4 a = {0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < S; ++s) a += *(const f32x4*)(part + (int64_t)s * NK + idx);
  f32x4* dst = (f32x4*)(dW + n * lddw + k);
  if (lay.accumulate) a += *dst;
  *dst = a;
}

// taps > 1: K = taps * Kc; thread (n, c) folds the slices of its `taps` columns tap * Kc + c (reads coalesced along c per
// tap) and writes element [n][c][0 .. taps) of the parameter: `taps` consecutive floats, contiguous across the workgroup.
__global__ __launch_bounds__(256) void wgrad_reduce_taps_kernel(const float* __restrict__ part, float* __restrict__ dW, int N,
                                                                int K, int S, Layout lay) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = (int64_t)lay.N_real * lay.C_real;
  if (i >= total) return;
  const int n = (int)(i / lay.C_real), c = (int)(i - (int64_t)n * lay.C_real);
  const int Kc = K / lay.taps;
  const int64_t NK = (int64_t)N * K;
  float* dst = dW + i * lay.taps;
  for (int tap = 0; tap < lay.taps; ++tap) {
    float a = 0.f;
    const float* src = part + (int64_t)n * K + (int64_t)tap * Kc + c;
    for (int s = 0; s < S; ++s) a += src[(int64_t)s * NK];
    dst[tap] = lay.accumulate ? dst[tap] + a : a;
  }
}

// token slices: ~1024 workgroups over the launch, at least 256 tokens per slice, at most 64 slices
// Which output tile a launch takes: 160 x 160 where it removes padded work (N and K multiples of 160 and at least one of
// them not a multiple of 128 — the 320-, 960-, 2880-wide gradients), 128 x 128 otherwise.  GCD_WGRAD_TILE=128 / 160 forces one.
inline int tile_of(int N, int K) {
  static const int forced = [] {
    const char* e = getenv("GCD_WGRAD_TILE");
    const int v = e ? atoi(e) : 0;
    return (v == 128 || v == 160) ? v : 0;
  }();
  if (forced == 128) return 128;
  const bool fits = N % 160 == 0 && K % 160 == 0;
  if (forced == 160) return fits ? 160 : 128;
  return fits && (N % 128 != 0 || K % 128 != 0) ? 160 : 128;
}

inline int slices(int64_t M, int N, int K) {
  const int T_ = tile_of(N, K);
  const int64_t tiles = (int64_t)((N + T_ - 1) / T_) * ((K + T_ - 1) / T_);
  int64_t S = 1024 / (tiles > 0 ? tiles : 1);
  if (S > 64) S = 64;
  if (S > M / 256) S = M / 256;
  if (S < 1) S = 1;
  return (int)S;
}

// Launch both passes on `s` with TM tokens per step (32 or 64; 64 needs the > 64 KB dynamic-LDS opt-in, done here once per
// process and device by the caller's flag).  Returns the HIP error of the launches.
template <bool BF16, int TM, int CONV, int TILE>
inline hipError_t launch_tile(const void* dy16, int64_t lddy, const void* x16, int64_t ldx, int64_t M, int N, int K,
                              float* dW, int64_t lddw, Layout lay, float* scratch, hipStream_t s, ConvGeo geo) {
  const int S = slices(M, N, K);
  const int64_t mper = ((M + S - 1) / S + TM - 1) / TM * TM;
  const dim3 grid((K + TILE - 1) / TILE, (N + TILE - 1) / TILE, S);
  auto fn = wgrad_tr_kernel<BF16, TM, CONV, TILE>;
  if (smem_bytes<TILE>(TM) > 64 * 1024) {
    // the > 64 KB dynamic-LDS opt-in: ONE driver call per process, device and instantiation (not one per weight
    // gradient; also keeps the launch path free of driver calls under stream capture after the first step)
    static std::atomic<bool> opted[64];
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64 || !opted[dev].load(std::memory_order_acquire)) {
      e = hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, smem_bytes<TILE>(TM));
      if (e != hipSuccess) return e;
      if (dev >= 0 && dev < 64) opted[dev].store(true, std::memory_order_release);
    }
  }
  hipLaunchKernelGGL(fn, grid, dim3(256), smem_bytes<TILE>(TM), s, (const f16*)dy16, lddy, (const f16*)x16, ldx, scratch, M, N,
                     K, mper, geo);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  if (lay.taps > 1)
    hipLaunchKernelGGL(wgrad_reduce_taps_kernel, dim3((unsigned)(((int64_t)lay.N_real * lay.C_real + 255) / 256)), dim3(256),
                       0, s, scratch, dW, N, K, S, lay);
  else
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)(((int64_t)N * K / 4 + 255) / 256)), dim3(256), 0, s, scratch, dW,
                       lddw, N, K, S, lay);
  return hipGetLastError();
}

template <bool BF16, int TM, int CONV = 0>
inline hipError_t launch(const void* dy16, int64_t lddy, const void* x16, int64_t ldx, int64_t M, int N, int K,
                         float* dW, int64_t lddw, Layout lay, float* scratch, hipStream_t s, ConvGeo geo = ConvGeo{0, 0, 0, 0, 0, 0}) {
  if (tile_of(N, K) == 160) return launch_tile<BF16, TM, CONV, 160>(dy16, lddy, x16, ldx, M, N, K, dW, lddw, lay, scratch, s, geo);
  return launch_tile<BF16, TM, CONV, 128>(dy16, lddy, x16, ldx, M, N, K, dW, lddw, lay, scratch, s, geo);
}

}  // namespace gcd_wgrad
