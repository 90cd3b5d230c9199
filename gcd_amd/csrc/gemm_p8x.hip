// gemm_p8x.hip — the tile epilogue under the NEXT tile's K loop (round 4).
//
// gemm_p8.hip runs K loop -> epilogue -> K loop in series: its 160 accumulator registers leave no room for a second
// tile.  Measured split of its fp16-output shapes (profiles/r04e_p8_ablations.txt, us): L0 GEGLU 597 = K loop 349 +
// epilogue 129 (+ 119 of interaction), L0 q|k|v 218 = 132 + 111, L1 GEGLU 410 = 322 + 68: while a CU stores and
// evaluates GELUs its matrix pipe idles.  Here a workgroup owns 256 x 160 tiles and every wave TWO accumulator sets of
// 80 registers (its 64 x 80 wave tile as 5 x 4 blocks of v_mfma_f32_16x16x32_f16): `acc` takes tile i+1's K loop while
// `prev` — tile i — is finished in ten slices issued INSIDE the MFMA clusters of tile i+1's first ten phases (a wave's
// own VALU / LDS / store instructions issue in the shadow of its own MFMAs: the issue rule of DESIGN.md §3.3).
// The persistent workgroup sees ONE stream of 32-deep sub-tiles across all its tiles: no per-tile prologue, no drain.
//
// Scope: GCD_GEMM_PLAIN, more tiles than CUs (persistent), M %% 256 == 0, N %% 160 == 0, K / 32 == 10 or >= 12,
// GCD_OUT_GEGLU (bias only) or GCD_OUT_F16 without residuals / per-frame vectors — the GEGLU and q|k|v projections of
// the two high-resolution levels.
//
// RESULT (MI355X, profiles/r04g_p8x_ab.txt): bit-identical outputs, and SLOWER than gemm_p8.hip on every shape it takes —
// L0 GEGLU 692 vs 573 us, L0 q|k|v 247 vs 215, L1 GEGLU 540 vs 407, L2 GEGLU 471 vs 351 — because the K loop pays for
// the second accumulator set: a 256 x 160 tile stages 26 KB per 20 MFMAs and wave = 41.6 B per MFMA clock against 28 for
// the 256 x 320 tile (the vector-memory -> LDS path peaks at 64 B / clk, and 16-row x 64-byte pieces use half of every
// 128-byte line), so its K loop runs at ~0.95 PF/s where gemm_p8's runs at 1.2-1.4.  What the overlap hides (the
// 70-130 us epilogues) is less than what the narrower tile loses.  Kept behind GCD_TUNE_GEMM_IMPL = 10 as the record of
// the experiment; the automatic choice never takes it.
//
// K loop: ring of FOUR slots of one 32-deep sub-tile (A 256 rows x 64 B + W 160 rows x 64 B = 26 KB; 104 KB), filled
// three sub-tiles ahead by buffer_load ... lds (out-of-range rows read zeros).  64-byte LDS rows, slot s of row r holds
// 16-byte chunk s ^ ((r >> 1) & 3) (conflict-free for the 16-lane read groups of ds_read_b128 with row = lane & 15,
// chunk = lane >> 4; swizzle on the per-lane SOURCE address).  One phase per sub-tile:
//     { stage sub-tile u+3 (3-4 pieces) | 9 ds_read_b128 | s_waitcnt vmcnt(N) | lgkmcnt(0) | s_barrier |
//       20 MFMA + epilogue slice of the previous tile | s_barrier },
// the two wave groups shifted by one barrier as in gemm_p8.hip.  N = (pieces of the two newest stages) + (stores the
// two previous phases issued): sub-tile u+1 has landed, nothing younger is waited for.
// GEGLU pairing: weight rows are interleaved 16 value / 16 gate (packing.pack_geglu); the DMA maps them so that every
// 16-row MFMA block holds 8 value rows + their 8 gate rows — value in lanes 0-31, gate in lanes 32-63 of the same
// token, exchanged with v_permlane32_swap across the two token blocks of a slice.
#include <type_traits>

#include "gemm_common.h"

namespace {

constexpr int X_BM = 256, X_BN = 160;
constexpr int X_A_BYTES = X_BM * 64;                   // 16384
constexpr int X_W_BYTES = X_BN * 64;                   // 10240
constexpr int X_SLOT = X_A_BYTES + X_W_BYTES;          // 26624
constexpr int X_RING = 4 * X_SLOT;                     // 106496
constexpr int X_STAGE_BYTES = 5632;                    // per wave: [32][80] fp16 rows of 176 B (GEGLU: [32][40], 96 B)
constexpr int X_STAGE0 = X_RING;
constexpr int X_BIAS0 = X_STAGE0 + 8 * X_STAGE_BYTES;  // 151552
constexpr int X_SMEM = X_BIAS0 + 3 * 160 * 4;          // 153472: three addend buffers (tile in `acc`, in `prev`, spare)
constexpr int X_GROUP_M = 4;

#define X_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define X_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define X_BAR()                          \
  do {                                   \
    __builtin_amdgcn_sched_barrier(0);   \
    __builtin_amdgcn_s_barrier();        \
    __builtin_amdgcn_sched_barrier(0);   \
  } while (0)

typedef f32x4 XAcc[5][4];   // [channel block][token block]: C[16 t + (lane & 15)][16 c + 4 (lane >> 4) + e]

// EPI: 1 GEGLU, 2 fp16 rows.  NS10: K == 320 (a tile is exactly its ten slice phases).
template <int EPI, bool NS10>
__global__ __launch_bounds__(512, 2) void gemm_p8x_kernel(const GemmK p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int grp = wave >> 2;
  const int wm = wave & 3, wn = grp;
  constexpr int NSTORE = EPI == 1 ? 3 : 5;     // store instructions of one 32-token half per wave

  int L0, L_end, L_step;
  {
    const int nblk = p.tiles_m * p.tiles_n, bid = blockIdx.x;
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    L0 = start + idx;
    L_end = start + q + (xcd < r ? 1 : 0);
    L_step = (int)(gridDim.x >> 3);
  }
  const int nS = p.K >> 5;
  auto tile_coords = [&](int Lx, int& m0, int& n0) {
    const int per_group = X_GROUP_M * p.tiles_n;
    const int gi = Lx / per_group;
    const int rem = Lx - gi * per_group;
    const int m_first = gi * X_GROUP_M;
    const int gm = min(X_GROUP_M, p.tiles_m - m_first);
    const int tn = rem / gm;
    m0 = (m_first + rem - tn * gm) * X_BM;
    n0 = tn * X_BN;
  };

  // ---- staging: one cursor, three sub-tiles ahead of the compute position ----
  const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrcW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, (int)p.w_bytes, 0x00020000);
  const unsigned lrow = lane >> 2;                                        // row of a 16-row piece
  const unsigned srcchunk = ((lane & 3) ^ ((lrow >> 1) & 3)) << 4;
  const unsigned lda2 = (unsigned)(p.lda * 2), k2 = (unsigned)p.K * 2u;
  // W piece g (16 LDS rows = one MFMA block): GEGLU rows 0-7 <- value rows, 8-15 <- the gate rows 16 further on
  const unsigned w_lane = (EPI == 1 ? lrow + (lrow >= 8 ? 8u : 0u) : lrow) * k2 + srcchunk;
  auto w_piece_row = [&](int g, int n0) { return EPI == 1 ? n0 + 32 * (g >> 1) + 8 * (g & 1) : n0 + 16 * g; };
  int Lc = L0, sc = 0, spos = 0;
  bool c_live = Lc < L_end;
  unsigned a_off_c = 0, w_roff_c0 = 0, w_roff_c1 = 0;
  auto cursor_set_tile = [&]() {
    int m0c, n0c;
    tile_coords(Lc, m0c, n0c);
    a_off_c = (unsigned)(m0c + 32 * wave + (int)lrow) * lda2 + srcchunk;
    w_roff_c0 = (unsigned)w_piece_row(wave, n0c) * k2;
    w_roff_c1 = (unsigned)w_piece_row(8 + (wave & 1), n0c) * k2;
  };
  auto bload = [&](__amdgpu_buffer_rsrc_t rs, unsigned voff, int soff, char* lds_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (GCD_AS3 void*)lds_wave_base, 16, (int)voff, soff, 0, 0);
  };
  // pieces per stage call: waves 0, 1: 4 (A 2 + W 2), waves 2-7: 3
  auto stage_next = [&]() {
    if (c_live) {
      char* dst = smem + (spos & 3) * X_SLOT;
      const int ko = sc * 64;
      bload(rsrcA, a_off_c, ko, dst + wave * 2048);
      bload(rsrcA, a_off_c + 16u * lda2, ko, dst + wave * 2048 + 1024);
      bload(rsrcW, w_lane + w_roff_c0, ko, dst + X_A_BYTES + wave * 1024);
      if (wave < 2) bload(rsrcW, w_lane + w_roff_c1, ko, dst + X_A_BYTES + (8 + wave) * 1024);
      if (++sc == nS) {
        sc = 0;
        Lc += L_step;
        c_live = Lc < L_end;
        if (c_live) cursor_set_tile();
      }
    }
    ++spos;
  };

  // ---- fragment read addresses ----
  const int r15 = lane & 15, q4 = lane >> 4;
  const int rdch = (q4 ^ ((r15 >> 1) & 3)) << 4;
  const int rdA = (64 * wm + r15) * 64 + rdch;
  const int rdW = X_A_BYTES + (80 * wn + r15) * 64 + rdch;

  XAcc acc, prev;
#pragma unroll
  for (int c = 0; c < 5; ++c)
#pragma unroll
    for (int tb = 0; tb < 4; ++tb) {
      acc[c][tb] = (f32x4){0.f, 0.f, 0.f, 0.f};
      prev[c][tb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  f16x8 af[4], wf[5];
  int upos = 0;                     // compute position in the stream (slot = upos & 3)
  bool have_prev = false;           // `prev` holds a real tile (its stores are issued)
  int pm0 = 0, pn0 = 0;             // the tile in `prev`
  int bias_idx = 0;                 // LDS addend buffer (of three) of the tile in `acc`; `prev` uses the one before
  bool stored_last = false;         // the previous tile's phase 9 issued its stores (vmcnt accounting, NS10)
  char* const stage = smem + X_STAGE0 + wave * X_STAGE_BYTES;

  // ---- the epilogue of `prev`, in ten slices (token half s / 5, channel block s % 5) + two store groups ----
  auto slice = [&](auto S) {
    constexpr int s = decltype(S)::value;
    constexpr int half = s / 5, cb = s % 5;
    const float* pb = (const float*)(smem + X_BIAS0) + (bias_idx == 0 ? 2 : bias_idx - 1) * 160 + 80 * wn + 16 * cb;
    if constexpr (EPI == 1) {
      const int q2 = q4 & 1;
      const f32x4 ba = *(const f32x4*)(pb + 4 * q2), bg = *(const f32x4*)(pb + 8 + 4 * q2);
      f16x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(prev[cb][2 * half][e]),
                                                         __float_as_uint(prev[cb][2 * half + 1][e]), false, false);
        const float v = __uint_as_float(sw[0]), g = __uint_as_float(sw[1]);   // value | gate of this lane's token
        o[e] = (f16)((v + ba[e]) * gelu_fast(g + bg[e]));
      }
      // lanes 0-31: token 16 (2 half) + r15, lanes 32-63: token 16 (2 half + 1) + r15; outputs 8 cb + 4 q2 + e
      *(f16x4*)(stage + (((lane >> 5) << 4) + r15) * 96 + (8 * cb + 4 * q2) * 2) = o;
    } else {
      const f32x4 bv = *(const f32x4*)(pb + 4 * q4);
#pragma unroll
      for (int th = 0; th < 2; ++th) {
        f16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (f16)((prev[cb][2 * half + th][e] + bv[e]) * p.s_acc);
        *(f16x4*)(stage + (16 * th + r15) * 176 + (16 * cb + 4 * q4) * 2) = o;
      }
    }
  };
  auto store_half = [&](int half) {   // the 32 staged rows of token half `half` -> global, 16 bytes per lane and store
    if (have_prev) {
      constexpr int PPR = EPI == 1 ? 5 : 10;             // 16-byte pieces per row
      constexpr int ROWB = EPI == 1 ? 96 : 176;
      f16* outp = (f16*)p.out + (int64_t)(pm0 + 64 * wm + 32 * half) * p.ldo + (EPI == 1 ? (pn0 >> 1) + 40 * wn : pn0 + 80 * wn);
#pragma unroll
      for (int it = 0; it < NSTORE; ++it) {
        const int idx = it * 64 + lane;
        const int row = idx / PPR, ch = idx - row * PPR;
        if (idx < 32 * PPR) {
          const f16x8 v = *(const f16x8*)(stage + row * ROWB + ch * 16);
          *(f16x8*)(outp + (int64_t)row * p.ldo + ch * 8) = v;
        }
      }
    }
  };

  auto read_frags = [&]() {
    const char* base = smem + (upos & 3) * X_SLOT;
#pragma unroll
    for (int tb = 0; tb < 4; ++tb) af[tb] = *(const f16x8*)(base + rdA + tb * 1024);
#pragma unroll
    for (int c = 0; c < 5; ++c) wf[c] = *(const f16x8*)(base + rdW + c * 1024);
  };
  auto mma = [&]() {
#pragma unroll
    for (int c = 0; c < 5; ++c)
#pragma unroll
      for (int tb = 0; tb < 4; ++tb)
        acc[c][tb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[c], af[tb], acc[c][tb], 0, 0, 0);
  };
  // counted wait of a phase: everything but the two newest stage calls (and the stores issued between them) has
  // landed.  s2 / s1: store instructions issued in the MFMA blocks two / one phases ago (only when have_prev).
  auto wait_landed = [&](auto EXTRA, bool stores_issued) {
    constexpr int extra = decltype(EXTRA)::value;
    if (!c_live) {   // the cursor is exhausted: fewer pieces than counted may be in flight
      X_VMCNT(0);
    } else if (extra == 0 || !stores_issued) {
      if (wave < 2) X_VMCNT(8);
      else X_VMCNT(6);
    } else if (extra == 3) {
      if (wave < 2) X_VMCNT(11);
      else X_VMCNT(9);
    } else {   // 5
      if (wave < 2) X_VMCNT(13);
      else X_VMCNT(11);
    }
  };
  using Z0 = std::integral_constant<int, 0>;
  using ZS = std::integral_constant<int, NSTORE>;

  auto run = [&](auto GRP) {
    constexpr int G = decltype(GRP)::value;
    // one phase; I = index within the tile (0..11 unrolled, -1 rolled)
    auto phase = [&](auto IDX) {
      constexpr int I = decltype(IDX)::value;
      stage_next();
      __builtin_amdgcn_sched_barrier(0);
      read_frags();
      // stores are issued in phases 5 and 9 of a tile; with NS10 the phases before 0 are the previous tile's 8 and 9
      if constexpr (I == 6 || I == 7 || I == 10 || I == 11) wait_landed(ZS{}, have_prev);
      else if constexpr (NS10 && (I == 0 || I == 1)) wait_landed(ZS{}, stored_last);
      else wait_landed(Z0{}, false);
      X_LGKM0();
      X_BAR();
      __builtin_amdgcn_s_setprio(1);
      mma();
      if constexpr (I >= 0 && I < 10) {
        if constexpr (I == 5) store_half(0);
        slice(std::integral_constant<int, (I >= 0 && I < 10) ? I : 0>{});
        if constexpr (I == 9) store_half(1);
      }
      __builtin_amdgcn_s_setprio(0);
      X_BAR();
      ++upos;
    };
    // prologue: three sub-tiles in flight, the first one landed
    if (c_live) cursor_set_tile();
    stage_next();
    stage_next();
    stage_next();
    if (wave < 2) X_VMCNT(8);
    else X_VMCNT(6);
    X_BAR();
    if (G == 1) X_BAR();
    for (int L = L0; L < L_end; L += L_step) {
      // tile switch: the finished tile moves to `prev`, its epilogue rides in this tile's first ten phases
#pragma unroll
      for (int c = 0; c < 5; ++c)
#pragma unroll
        for (int tb = 0; tb < 4; ++tb) {
          prev[c][tb] = acc[c][tb];
          acc[c][tb] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
      int m0, n0;
      tile_coords(L, m0, n0);
      if (t < 40) {   // this tile's 160 per-channel addends in LDS-row order
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (p.bias) {
          const int row = 4 * t;   // LDS row of the tile; GEGLU block row r: value r < 8, gate r >= 8 (+8 rows on)
          const int g = row >> 4, r = row & 15;
          const int n = EPI == 1 ? n0 + 32 * (g >> 1) + 8 * (g & 1) + r + (r >= 8 ? 8 : 0) : n0 + row;
          v = *(const f32x4*)(p.bias + n);
        }
        *(f32x4*)((float*)(smem + X_BIAS0) + bias_idx * 160 + 4 * t) = v;
      }
      phase(std::integral_constant<int, 0>{});
      phase(std::integral_constant<int, 1>{});
      phase(std::integral_constant<int, 2>{});
      phase(std::integral_constant<int, 3>{});
      phase(std::integral_constant<int, 4>{});
      phase(std::integral_constant<int, 5>{});
      phase(std::integral_constant<int, 6>{});
      phase(std::integral_constant<int, 7>{});
      phase(std::integral_constant<int, 8>{});
      phase(std::integral_constant<int, 9>{});
      if constexpr (!NS10) {
        phase(std::integral_constant<int, 10>{});
        phase(std::integral_constant<int, 11>{});
        for (int s = 12; s < nS; ++s) phase(std::integral_constant<int, -1>{});
      }
      stored_last = have_prev;
      have_prev = true;
      pm0 = m0;
      pn0 = n0;
      bias_idx = bias_idx == 2 ? 0 : bias_idx + 1;   // (a buffer is rewritten three tiles later: the lagging wave
                                                      //  group may still read the older one in its phase 9)
    }
    if (G == 0) X_BAR();
    // drain: the last tile's epilogue, nothing to hide it under
#pragma unroll
    for (int c = 0; c < 5; ++c)
#pragma unroll
      for (int tb = 0; tb < 4; ++tb) prev[c][tb] = acc[c][tb];
    slice(std::integral_constant<int, 0>{});
    slice(std::integral_constant<int, 1>{});
    slice(std::integral_constant<int, 2>{});
    slice(std::integral_constant<int, 3>{});
    slice(std::integral_constant<int, 4>{});
    store_half(0);
    slice(std::integral_constant<int, 5>{});
    slice(std::integral_constant<int, 6>{});
    slice(std::integral_constant<int, 7>{});
    slice(std::integral_constant<int, 8>{});
    slice(std::integral_constant<int, 9>{});
    store_half(1);
  };
  if (grp == 0) run(std::integral_constant<int, 0>{});
  else run(std::integral_constant<int, 1>{});
}

template <int EPI, bool NS10>
int launch_p8x(const GemmK& k, hipStream_t s) {
  static GcdPerDeviceOnce attr_once;
  auto fn = gemm_p8x_kernel<EPI, NS10>;
  GCD_CHECK_HIP(attr_once.opt_in((const void*)fn, X_SMEM));
  GemmK kk = k;
  kk.a_frames = 0;
  kk.a_bytes = (uint32_t)((((int64_t)k.M - 1) * k.lda + k.K) * 2);
  kk.w_bytes = (uint32_t)((int64_t)k.N * k.K * 2);
  kk.tiles_m = k.M / X_BM;
  kk.tiles_n = k.N / X_BN;
  hipLaunchKernelGGL(fn, dim3(256), dim3(512), X_SMEM, s, kk);
  GCD_CHECK_LAUNCH();
  return 0;
}

}  // namespace

bool gcd_gemm_p8x_supported(const GemmK& k, int mode) {
  if (mode != GCD_GEMM_PLAIN || k.M % X_BM != 0 || k.N % X_BN != 0 || k.K % 32 != 0) return false;
  const int nS = k.K / 32;
  if (nS != 10 && nS < 12) return false;
  if ((int64_t)(k.M / X_BM) * (k.N / X_BN) <= 256) return false;
  if (k.operand_bf16 || k.ln_out || k.a_blocked || k.out_blocked || k.colstats) return false;
  if (k.R1 || k.R2 || k.rowvec || k.frame_alpha) return false;
  if (k.out_kind == GCD_OUT_GEGLU) {
    if (k.N % 32 != 0 || (k.ldo & 7) != 0) return false;
  } else if (k.out_kind == GCD_OUT_F16) {
    if ((k.ldo & 7) != 0) return false;
  } else {
    return false;
  }
  if (((int64_t)k.M + 256) * k.lda * 2 + (int64_t)k.K * 2 >= 0xFFFFFF00ll) return false;
  if (((int64_t)k.N + 320) * k.K * 2 >= 0xFFFFFF00ll) return false;
  return true;
}

int gcd_gemm_p8x_launch(const GemmK& k, hipStream_t s) {
  const bool ns10 = k.K == 320;
  if (k.out_kind == GCD_OUT_GEGLU) return ns10 ? launch_p8x<1, true>(k, s) : launch_p8x<1, false>(k, s);
  return ns10 ? launch_p8x<2, true>(k, s) : launch_p8x<2, false>(k, s);
}
