// gemm_pp.hip — 256 x 320 "ping-pong" MFMA GEMM for gfx950 (MI355X): the fast path of the GEMM
// family (nn.Linear, Conv2d 3x3 / 1x1, Conv3d (3,1,1) as implicit GEMMs) on the large-M shapes of
// the SVD VideoUNet.  Same contract and epilogue as gemm.hip (see gcd_gemm_desc); gemm.hip remains
// the general kernel for small / ragged problems.
//
// Why this shape.  Every output width of the UNet is a multiple of 320 (320, 640, 960, 1280, 1920,
// 2560, 3840, 5120, 10240) and every token count but the bottleneck's is a multiple of 256, so a
// 256 x 320 block tile has no padding waste.  Per 32-deep K slice a CU stages (256 + 320) x 64 B =
// 36 KB for 2*256*320*32 = 5.2 MFLOP, i.e. 28.8 B per MFMA clock — well inside the 64 B/clk of the
// vector-memory -> LDS path that caps 128 x 128 tiles at ~900 TFLOP/s.
//
// Structure (one workgroup = 8 waves = 2 per SIMD, one workgroup per CU):
//   * waves form a 4 (M) x 2 (N) grid; a wave owns 64 tokens x 160 channels = 2 x 5 tiles of
//     v_mfma_f32_32x32x16_f16 (operands swapped: W rows are the MFMA "A" operand, so a lane ends up
//     with 4 consecutive output channels of one token and the epilogue moves 16-byte vectors);
//   * K is consumed in 32-deep sub-tiles that live in a ring of FOUR 36 KB LDS slots (144 of the
//     160 KB); global_load_lds_dwordx4 (LDS-DMA) fills a slot three sub-tiles ahead of its use.
//     The LDS image of a DMA is lane-linear, so the bank swizzle (chunk ^= (row >> 2) & 3) is
//     applied to the per-lane SOURCE address and undone by the ds_read_b128 address;
//   * the two wave groups (waves 0-3 / 4-7, one of each per SIMD) run the same phase sequence
//         { ds_read 7 fragments + issue <= 3 DMA pieces | s_barrier | 10 MFMA | s_barrier }
//     but group 1 is shifted by ONE barrier: while one wave of a SIMD feeds the matrix pipe its
//     partner reads LDS and issues DMA, then they swap (2 phases per sub-tile, 16-deep each);
//   * DMA completion is tracked with COUNTED s_waitcnt vmcnt(8) (never 0 in steady state): a wave
//     only waits for the pieces of the sub-tile that is read two barriers later; 8 newer pieces
//     stay in flight across the barriers.
//   * hazards (sigma = sub-tile index, slot = sigma & 3):
//       RAW  a reader passes a barrier that every wave reached after its own counted wait for sigma;
//       WAR  a slot is re-filled at the earliest two barriers after the last ds_read of its previous
//            content was retired by the reading wave's lgkmcnt wait.
//   * blockIdx -> tile map: each XCD owns a contiguous range of tiles, walked in groups of 4 M-tiles
//     x all N-tiles with M fastest, so the 32 workgroups an XCD runs concurrently share A row panels
//     and W column panels through that XCD's L2.
#include "gemm_common.h"

namespace {

constexpr int PP_BM = 256, PP_BN = 320;
constexpr int PP_A_BYTES = PP_BM * 64;                  // A sub-tile: 256 rows x 32 fp16
constexpr int PP_W_BYTES = PP_BN * 64;                  // W sub-tile: 320 rows x 32 fp16
constexpr int PP_SLOT = PP_A_BYTES + PP_W_BYTES;        // 36864
constexpr int PP_SMEM = 4 * PP_SLOT;                    // 147456: the ring
// Epilogue staging: 8 wave-private regions of GCD_EPI_STAGE_BYTES that start at slot 2 and run 12 KB past
// the ring, so that slots 0 and 1 stay free for the NEXT tile's first two sub-tiles while a persistent
// workgroup is in its epilogue; then two buffers of 320 per-column addends (this tile's / the next's).
constexpr int PP_STAGE0 = 2 * PP_SLOT;
constexpr int PP_BIAS0 = PP_STAGE0 + 8 * GCD_EPI_STAGE_BYTES;   // 159744
static_assert(PP_BIAS0 >= PP_SMEM, "staging must cover the ring's tail");
constexpr int PP_SMEM_LAUNCH = PP_BIAS0 + 2 * 320 * 4;          // 162304 of the 163840 B of LDS
constexpr int PP_GROUP_M = 4;

#define PP_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

typedef __bf16 pp_bf16x8 __attribute__((ext_vector_type(8)));

// VAR: ablation switches for tools/gemm_bench (0 = the product kernel; any other value computes
// WRONG results and exists only to price the parts of the loop):
//   1 no DMA in the main loop   2 no ds_read in the loop   4 no one-barrier stagger
//   8 no s_setprio             16 no barriers in the loop
// VAR bit 65536: the operands are bfloat16 (gcd_gemm_desc.operand_bf16, fp32 output): the same 16-bit staging,
// swizzle and fragment reads, v_mfma_f32_32x32x16_bf16 instead of ..._f16 — same shape and rate on gfx950.
template <int MODE, int VAR>
__global__ __launch_bounds__(512, 2) void gemm_pp_kernel(const GemmK p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int grp = wave >> 2;            // 0 leads, 1 runs one barrier behind
  const int wm = wave & 3, wn = grp;
  const int l31 = lane & 31, hh = lane >> 5;

  // ---- XCD-aware, panel-sharing tile assignment (bijective for any grid) ----
  // Each XCD owns a contiguous range of the linear tile order.  One tile per workgroup, or (VAR bit
  // 2048, persistent) 32 workgroups per XCD that walk their XCD's range with stride 32.
  constexpr bool PERSIST = (VAR & 2048) != 0;
  // A operand tile-blocked in 160-column blocks ([M/256][K/160][256][160], the GEGLU hidden tensor as
  // out_blocked wrote it): PLAIN mode only, own instantiation so the plain K walk keeps its address math
  constexpr bool ABLK = MODE == GCD_GEMM_PLAIN && (VAR & 32) != 0 && !(VAR & 31);
  int L, L_end, L_step;
  {
    const int nblk = PERSIST ? p.tiles_m * p.tiles_n : (int)gridDim.x, bid = blockIdx.x;
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    L = start + idx;
    L_end = PERSIST ? start + q + (xcd < r ? 1 : 0) : L + 1;
    L_step = PERSIST ? (int)(gridDim.x >> 3) : 1;
  }
  // ---- per-tile state (set_tile) ----
  int m0 = 0, n0 = 0, kz = 0, s_begin = 0, S = p.K >> 5;
  const int lrow = lane >> 2;
  const int lc16 = ((lane & 3) ^ ((lane >> 4) & 3)) << 4;   // logical 16-B chunk this lane fetches
  // DMA bookkeeping: a piece = 16 rows x 64 B = one global_load_lds_dwordx4 of a wave.
  // A: wave w loads pieces 2w, 2w+1 (tile rows 32w .. 32w+31)
  const char* a_base[2] = {nullptr, nullptr};
  int a_inc[2] = {2, 2};     // bytes per channel step: 2, or 0 when the tap reads the zero page
  int a_y[2] = {0, 0}, a_x[2] = {0, 0};
  int64_t a_fb[2] = {0, 0};
  int a_tap = 0, a_c0 = 0;   // K position of the next A issue (conv modes), block-uniform
  // W: group 0 wave w loads pieces 3w .. 3w+2, group 1 wave w' loads 12+2w', 13+2w'
  const int w_first = grp == 0 ? 3 * wave : 12 + 2 * (wave - 4);
  const char* w_base[3] = {nullptr, nullptr, nullptr};
  float* lds_bias = (float*)(smem + PP_BIAS0);   // 320 staged per-column addends (of two buffers)
  bool lds_bias_ok = false;   // staged addends cover bias (+ rowvec) of the whole tile
  bool alpha_uni = true;      // one frame_alpha entry serves the whole tile

  auto set_tap = [&](int tap) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (MODE == GCD_GEMM_CONV3X3) {
        const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
        int iy, ix;
        bool ok;
        if (p.up) {
          const int uy = a_y[i] + dy, ux = a_x[i] + dx;
          ok = uy >= 0 && uy < p.Ho && ux >= 0 && ux < p.Wo;
          iy = uy >> 1;
          ix = ux >> 1;
        } else {
          iy = a_y[i] * p.stride + dy + p.asym;
          ix = a_x[i] * p.stride + dx + p.asym;
          ok = iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi;
        }
        a_base[i] = ok ? (const char*)p.A + (a_fb[i] + (int64_t)iy * p.Wi + ix) * p.lda * 2 + lc16
                       : (const char*)p.zero + lc16;
        a_inc[i] = ok ? 2 : 0;
      } else if (MODE == GCD_GEMM_TEMPORAL3) {
        const int dt = tap - 1;
        const int tt = a_y[i] + dt;
        const bool ok = tt >= 0 && tt < p.T;
        a_base[i] = ok ? (const char*)p.A + (a_fb[i] + (int64_t)dt * p.HW) * p.lda * 2 + lc16
                       : (const char*)p.zero + lc16;
        a_inc[i] = ok ? 2 : 0;
      }
    }
  };
  // Tile Lx of the linear order -> (m0, n0), DMA source addresses, staged epilogue addends.
  auto set_tile = [&](int Lx) {
    int tile_m, tile_n;
    kz = (VAR & 16384) ? Lx % p.splitk : 0;          // split-K: K slice of this workgroup
    {
      const int Lt = (VAR & 16384) ? Lx / p.splitk : Lx;
      const int per_group = PP_GROUP_M * p.tiles_n;
      const int gi = Lt / per_group;
      const int rem = Lt - gi * per_group;
      const int m_first = gi * PP_GROUP_M;
      const int gm = min(PP_GROUP_M, p.tiles_m - m_first);
      tile_n = rem / gm;
      tile_m = m_first + rem - tile_n * gm;
    }
    m0 = tile_m * PP_BM;
    n0 = tile_n * PP_BN;
    // 32-deep sub-tiles of this workgroup: all of K, or slice kz of it
    s_begin = 0;
    S = p.K >> 5;
    if (VAR & 16384) {
      const int per = (S + p.splitk - 1) / p.splitk;
      s_begin = kz * per;
      S = min(per, S - s_begin);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int m = m0 + 32 * wave + 16 * i + lrow;
      m = m < p.M ? m : p.M - 1;
      a_y[i] = a_x[i] = 0;
      a_fb[i] = 0;
      a_base[i] = nullptr;
      a_inc[i] = 2;
      if (MODE == GCD_GEMM_PLAIN && ABLK) {
        a_base[i] = (const char*)p.A + ((int64_t)(m >> 8) * (p.K / 160) * 256 + (m & 255)) * 320 + lc16;
      } else if (MODE == GCD_GEMM_PLAIN) {
        a_base[i] = (const char*)p.A + (int64_t)m * p.lda * 2 + lc16;
      } else if (MODE == GCD_GEMM_CONV3X3) {
        const int hw = p.Ho * p.Wo;
        const int n = m / hw;
        const int rem = m - n * hw;
        a_y[i] = rem / p.Wo;
        a_x[i] = rem - a_y[i] * p.Wo;
        a_fb[i] = (int64_t)n * p.Hi * p.Wi;
      } else {
        a_y[i] = (m / p.HW) % p.T;
        a_fb[i] = m;
      }
    }
    a_tap = (MODE != GCD_GEMM_PLAIN) ? (s_begin * 32) / p.Cin : 0;
    a_c0 = (MODE != GCD_GEMM_PLAIN) ? s_begin * 32 - a_tap * p.Cin : 0;
    if (MODE != GCD_GEMM_PLAIN) set_tap(a_tap);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      int n = n0 + 16 * (w_first + j) + lrow;
      n = n < p.N ? n : p.N - 1;
      w_base[j] = (const char*)p.W + (int64_t)n * p.K * 2 + lc16;
    }
    // epilogue addends of this tile's 320 columns: bias, plus the rowvec row when one row serves
    // the whole tile (always at the 72x128 / 36x64 levels: 9216 and 2304 rows per frame)
    if (!(VAR & (16384 | 8192))) {   // (the colstats build, VAR & 4096, stages them too)
      const int m_last = min(m0 + PP_BM, p.M) - 1;
      const bool rv_uni = !p.rowvec || (m0 / p.rows_per_vec == m_last / p.rows_per_vec);
      lds_bias_ok = rv_uni;
      alpha_uni = !p.frame_alpha || (m0 / p.rows_per_alpha == m_last / p.rows_per_alpha);
      if (t < 80) {
        const int n = n0 + 4 * t;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (n < p.N) {
          if (p.bias) v = *(const f32x4*)(p.bias + n);
          if (p.rowvec && rv_uni)
            v += *(const f32x4*)(p.rowvec + (int64_t)(m0 / p.rows_per_vec) * p.ld_rowvec + n);
        }
        *(f32x4*)(lds_bias + 4 * t) = v;
      }
    }
  };

  bool dma_on = true;
  auto issue_A = [&](int sigma) {
    if (sigma < S && dma_on) {
      char* dst = smem + (sigma & 3) * PP_SLOT + wave * 2048;
      if (MODE == GCD_GEMM_PLAIN && ABLK) {
        const int sg = s_begin + sigma, kb = sg / 5;          // 5 sub-tiles of 32 columns per 160-column block
        const int off = kb * (256 * 320) + (sg - 5 * kb) * 64;
        glds16(a_base[0] + off, dst);
        glds16(a_base[1] + off, dst + 1024);
      } else if (MODE == GCD_GEMM_PLAIN) {
        glds16(a_base[0] + (s_begin + sigma) * 64, dst);
        glds16(a_base[1] + (s_begin + sigma) * 64, dst + 1024);
      } else {
        glds16(a_base[0] + a_c0 * a_inc[0], dst);
        glds16(a_base[1] + a_c0 * a_inc[1], dst + 1024);
      }
      if (MODE != GCD_GEMM_PLAIN) {
        a_c0 += 32;
        if (a_c0 == p.Cin) {
          a_c0 = 0;
          ++a_tap;
          set_tap(a_tap);
        }
      }
    }
  };
  auto issue_W = [&](int j, int sigma) {
    if (sigma < S && dma_on)
      glds16(w_base[j] + (s_begin + sigma) * 64,
             smem + (sigma & 3) * PP_SLOT + PP_A_BYTES + (w_first + j) * 1024);
  };
  // prologue in steady-state order: sub-tiles 0, 1 (part a) and (group 0: the first part of) 2 (part b)
  auto issue_prologue_a = [&]() {
#pragma unroll
    for (int sg = 0; sg < 2; ++sg) {
      issue_A(sg);
      issue_W(0, sg);
      issue_W(1, sg);
      if (grp == 0) issue_W(2, sg);
    }
  };
  auto issue_prologue_b = [&]() {
    issue_A(2);
    issue_W(0, 2);
    if (grp != 0) issue_W(1, 2);
  };
  // Cross-tile prefetch (persistent kernels): part a of the NEXT tile is issued before this tile's
  // epilogue (slots 0 / 1 are not staging space), so its HBM / L2 latency and set_tile's address
  // arithmetic run under the epilogue.  On gfx9 stores share vmcnt with the DMA loads and retire in
  // order, so the first two counted waits of the next K loop admit the epilogue's `nst` stores (a
  // LOWER bound of what the epilogue path issued after part a: more only makes the wait stricter).
  // PLAIN mode only: the conv modes carry tap state through the epilogue, which costs them spills in
  // the K loop (measured: -20 % on the 3x3 convs), and their long K loops hide little of it anyway.
  constexpr bool XPF = PERSIST && MODE == GCD_GEMM_PLAIN && !(VAR & (16384 | 8192));
  int nst = 0;
  auto wait_sub0 = [&]() {   // "all but the newest 8 + nst pieces have landed"
    if (nst == 40) PP_VMCNT(48);
    else if (nst == 20) PP_VMCNT(28);
    else if (nst == 10) PP_VMCNT(18);
    else PP_VMCNT(8);
  };

  // ---- fragment read addresses (per lane, within a slot) ----
  const int swz = (l31 >> 2) & 3;
  int rdA[2], rdW[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const int ch = ((2 * ks + hh) ^ swz) << 4;
    rdA[ks] = (64 * wm + l31) * 64 + ch;
    rdW[ks] = PP_A_BYTES + (160 * wn + l31) * 64 + ch;
  }

  if (XPF && L < L_end) {      // (a surplus workgroup of a rounded-up grid has no tile)
    set_tile(L);
    issue_prologue_a();
  }
  for (; L < L_end; L += L_step) {
  if (!XPF) {
    set_tile(L);
    issue_prologue_a();
  }
  issue_prologue_b();

  f32x16 acc[5][2];
#pragma unroll
  for (int i = 0; i < 5; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  f16x8 af[2], wf[5];
  bool reads_on = true;
  auto load_frags = [&](int slot, int ks) {
    if ((VAR & 2) && !reads_on) {
#pragma unroll
      for (int j = 0; j < 2; ++j) asm volatile("" : "+v"(af[j]));
#pragma unroll
      for (int i = 0; i < 5; ++i) asm volatile("" : "+v"(wf[i]));
      return;
    }
    const char* base = smem + slot * PP_SLOT;
#pragma unroll
    for (int j = 0; j < 2; ++j) af[j] = *(const f16x8*)(base + rdA[ks] + j * 2048);
#pragma unroll
    for (int i = 0; i < 5; ++i) wf[i] = *(const f16x8*)(base + rdW[ks] + i * 2048);
  };
  auto mma = [&]() {
    if (!(VAR & 8)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if constexpr ((VAR & 65536) != 0)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(pp_bf16x8, wf[i]),
                                                              __builtin_bit_cast(pp_bf16x8, af[j]), acc[i][j], 0, 0, 0);
        else
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[i], af[j], acc[i][j], 0, 0, 0);
      }
    if (!(VAR & 8)) __builtin_amdgcn_s_setprio(0);
  };
  bool bars_on = true;
#define PP_BAR()                                                   \
  do {                                                             \
    __builtin_amdgcn_sched_barrier(0);                             \
    if (!(VAR & 16) || bars_on) __builtin_amdgcn_s_barrier();      \
    __builtin_amdgcn_sched_barrier(0);                             \
  } while (0)

  if (S > 2) {
    wait_sub0();   // everything of sub-tile 0 has landed (8 newer pieces [+ nst stores] may still fly)
  } else {
    PP_VMCNT(0);
  }
  PP_BAR();
  if (VAR & 1) dma_on = false;
  if (VAR & 2) {
    load_frags(0, 0);
    reads_on = false;
  }
  if (VAR & 16) bars_on = false;

  if (grp == 0) {
    for (int s = 0; s < S; ++s) {
      const int slot = s & 3;
      load_frags(slot, 0);
      issue_W(1, s + 2);
      issue_W(2, s + 2);
      PP_BAR();
      mma();
      PP_BAR();
      load_frags(slot, 1);
      issue_A(s + 3);
      issue_W(0, s + 3);
      if (s + 3 < S) {
        if (XPF && s == 0) wait_sub0();   // the epilogue's stores sit between sub-tiles 1 and 2
        else PP_VMCNT(8);   // sub-tile s+1 complete: newer = 5 (s+2) + 3 (first part of s+3)
      } else {
        PP_VMCNT(0);
      }
      PP_BAR();
      mma();
      PP_BAR();
    }
    if (!(VAR & 4)) PP_BAR();   // pairs with group 1's last barrier
  } else {
    if (!(VAR & 4)) PP_BAR();   // the one-barrier stagger
    for (int s = 0; s < S; ++s) {
      const int slot = s & 3;
      load_frags(slot, 0);
      issue_A(s + 3);
      PP_BAR();
      mma();
      PP_BAR();
      load_frags(slot, 1);
      issue_W(0, s + 3);
      issue_W(1, s + 3);
      if (s + 3 < S) {
        if (XPF && s == 0) wait_sub0();
        else PP_VMCNT(8);   // sub-tile s+1 complete: newer = 4 (s+2) + 4 (s+3)
      } else {
        PP_VMCNT(0);
      }
      PP_BAR();
      mma();
      PP_BAR();
    }
  }

  // ---- epilogue (gemm_common.h) ----
  // (opaque copies: keeps the compiler from hoisting the epilogue's address arithmetic above the K
  //  loop, where it would cost registers the loop does not have)
  int wm_base = m0 + 64 * wm, wn_base = n0 + 160 * wn, elane = lane;
  const int e_m0 = m0;
  const bool e_bias_ok = lds_bias_ok, e_alpha_uni = alpha_uni;
  const float* const e_bias = lds_bias;
  char* const e_stage = smem + PP_STAGE0 + wave * GCD_EPI_STAGE_BYTES;
  if (XPF && L + L_step < L_end) {   // the next tile's addresses, addends (other buffer) and first DMA
    lds_bias = (float*)(smem + PP_BIAS0) + (e_bias == (const float*)(smem + PP_BIAS0) ? 320 : 0);
    set_tile(L + L_step);
    issue_prologue_a();
  }
  nst = 0;
  asm volatile("" : "+s"(wm_base), "+s"(wn_base), "+v"(elane));
  if constexpr ((VAR & 16384) != 0) {   // split-K: raw fp32 partial sums of this K slice
    GemmK q = p;
    q.out = (float*)p.out + (int64_t)kz * p.split_stride;
    gcd_epilogue_64x160<8>(q, acc, wm_base, wn_base, elane, smem);
  } else if constexpr ((VAR & 8192) != 0) {   // own instantiation: fused LayerNorm (N == 320)
    gcd_epilogue_64x160_ln(p, acc, wm_base, wn_base, elane, wm, wn, (float*)smem);
  } else if constexpr ((VAR & 4096) != 0) {
    // own instantiation: fp32 rows + per-64-row column statistics for the next GroupNorm
    // (gcd_gemm_desc.colstats).  gcd_gemm_f16 validated that EVERY tile of the launch is full, has
    // tile-uniform per-frame vectors / blend factors and at most one residual.
    float sa = p.s_acc, sr1 = p.s_r1;
    if (p.frame_alpha) {
      const float al = p.frame_alpha[e_m0 / p.rows_per_alpha];
      sa = 1.0f - al;
      if (p.r1_blend) sr1 *= 1.0f - al;
    }
    const float* lb = e_bias + 160 * wn;
    char* stage = e_stage;
    p.R1 ? gcd_epi_f32_rows_full<true, false, false, true>(p, acc, wm_base, wn_base, elane, lb, stage, sa, sr1, 0.f)
         : gcd_epi_f32_rows_full<false, false, false, true>(p, acc, wm_base, wn_base, elane, lb, stage, sa, sr1, 0.f);
    nst = 40;   // 10 column-block x token-half tiles x 4 row-group stores (the statistics stores come on top)
  } else {
    constexpr int EV = ((VAR >> 6) & 31) | ((VAR & 32768) ? 32 : 0);
    const bool full = wm_base + 64 <= p.M && wn_base + 160 <= p.N && e_bias_ok;
    const float* lb = e_bias + 160 * wn;
#ifdef GCD_ABLATION_BUILD
    if (EV == 8 && full && p.out_kind == GCD_OUT_GEGLU) {
      // ablation: the product's staged GEGLU epilogue, but every tile of a wave lands on the same 10 KB
      // (L2-resident): prices the memory side of the stores against their issue / staging side
      GemmK q = p;
      q.out = (f16*)p.out + (int64_t)((blockIdx.x & 255) * 8 + wave) * 64 * 80;
      q.ldo = 80;
      q.out_blocked = 0;
      gcd_epi_geglu_rows_full(q, acc, 0, 0, elane, lb, e_stage);
    } else
#endif
    if (EV == 0 && full && p.out_kind == GCD_OUT_GEGLU && (p.ldo & 7) == 0) {
      gcd_epi_geglu_rows_full(p, acc, wm_base, wn_base, elane, lb, e_stage);
      nst = 10;
    } else if (EV == 0 && full && p.out_kind == GCD_OUT_F16 && !p.R1 && !p.R2 && !p.frame_alpha &&
               (p.ldo & 7) == 0) {
      gcd_epi_f16_rows_full(p, acc, wm_base, wn_base, elane, lb, e_stage);
      nst = 20;
    } else if (EV == 0 && full && e_alpha_uni && p.out_kind == GCD_OUT_F16 && p.R1 && p.R2 && (p.ldo & 3) == 0) {
      float sa = p.s_acc, sr1 = p.s_r1, sr2 = p.s_r2;
      if (p.frame_alpha) {
        const float al = p.frame_alpha[e_m0 / p.rows_per_alpha];
        sa = 1.0f - al;
        sr2 = al;
        if (p.r1_blend) sr1 *= 1.0f - al;
      }
      gcd_epi_f32_rows_full<true, true, true>(p, acc, wm_base, wn_base, elane, lb,
                                              e_stage, sa, sr1, sr2);
      nst = 40;
    } else if (EV == 0 && full && e_alpha_uni && p.out_kind == GCD_OUT_F32) {
      float sa = p.s_acc, sr1 = p.s_r1, sr2 = p.s_r2;
      if (p.frame_alpha) {
        const float al = p.frame_alpha[e_m0 / p.rows_per_alpha];
        sa = 1.0f - al;
        sr2 = al;
        if (p.r1_blend) sr1 *= 1.0f - al;
      }
      char* stage = e_stage;
      if (p.R2) {
        p.R1 ? gcd_epi_f32_rows_full<true, true>(p, acc, wm_base, wn_base, elane, lb, stage, sa, sr1, sr2)
                   : gcd_epi_f32_rows_full<false, true>(p, acc, wm_base, wn_base, elane, lb, stage, sa, sr1, sr2);
      } else {
        p.R1 ? gcd_epi_f32_rows_full<true, false>(p, acc, wm_base, wn_base, elane, lb, stage, sa, sr1, sr2)
                   : gcd_epi_f32_rows_full<false, false>(p, acc, wm_base, wn_base, elane, lb, stage, sa, sr1, sr2);
      }
      nst = 40;
    } else {
      // ragged edge tiles, fp16 outputs, a rowvec / alpha that changes inside the tile
      gcd_epilogue_64x160<EV>(p, acc, wm_base, wn_base, elane, e_stage);
    }
  }
  if (!XPF) nst = 0;   // without the prefetch every DMA piece of the next tile is younger than the stores
  if (PERSIST) __syncthreads();   // epilogue LDS use vs the next tile's prologue DMA
  }   // tile loop
}

template <int MODE, int VAR = 0>
int launch_pp(const GemmK& k, hipStream_t s) {
  static GcdPerDeviceOnce attr_once;
  auto fn = gemm_pp_kernel<MODE, VAR>;
  GCD_CHECK_HIP(attr_once.opt_in((const void*)fn, PP_SMEM_LAUNCH));
  GemmK kk = k;
  kk.tiles_m = (k.M + PP_BM - 1) / PP_BM;
  kk.tiles_n = (k.N + PP_BN - 1) / PP_BN;
  int64_t nblk = (int64_t)kk.tiles_m * kk.tiles_n;
  GCD_CHECK_ARG(nblk > 0 && nblk < (1ll << 31), "gcd_gemm_f16 (pp): bad grid %lld", (long long)nblk);
  // Round 5: the product library instantiates PERSISTENT kernels only; <= 256 tiles run on a grid of the tile count
  // rounded up to a multiple of 8 (same tile -> workgroup map as one workgroup per tile; surplus workgroups find no tile).
  if (VAR & 2048) nblk = nblk > 256 ? 256 : (nblk + 7) / 8 * 8;
  hipLaunchKernelGGL(fn, dim3((unsigned)nblk), dim3(512), PP_SMEM_LAUNCH, s, kk);
  GCD_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// split-K: sum the K-slice partials and run the epilogue (bias, per-frame vectors, residuals with
// AlphaBlender scales, fp32 / fp16 out) on 16-byte vectors.  grid-stride over M * N / 4.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const GemmK p, const float* __restrict__ ws,
                                                            int splitk) {
  const int nv = p.N >> 2;
  const int64_t total = (int64_t)p.M * nv;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * 256) {
    const int m = (int)(idx / nv);
    const int n = (int)(idx - (int64_t)m * nv) * 4;
    f32x4 v = *(const f32x4*)(ws + (int64_t)m * p.N + n);
    for (int z = 1; z < splitk; ++z) v += *(const f32x4*)(ws + z * p.split_stride + (int64_t)m * p.N + n);
    float sa = p.s_acc, sr1 = p.s_r1, sr2 = p.s_r2;
    if (p.frame_alpha) {
      const float al = p.frame_alpha[m / p.rows_per_alpha];
      sa = 1.0f - al;
      sr2 = al;
      if (p.r1_blend) sr1 *= 1.0f - al;
    }
    if (p.bias) v += *(const f32x4*)(p.bias + n);
    if (p.rowvec) v += *(const f32x4*)(p.rowvec + (int64_t)(m / p.rows_per_vec) * p.ld_rowvec + n);
    v *= sa;
    if (p.R1) v += sr1 * *(const f32x4*)(p.R1 + (int64_t)m * p.ldr1 + n);
    if (p.R2) v += sr2 * *(const f32x4*)(p.R2 + (int64_t)m * p.ldr2 + n);
    if (p.out_kind == GCD_OUT_F32) {
      *(f32x4*)((float*)p.out + (int64_t)m * p.ldo + n) = v;
    } else {
      f16x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (f16)v[e];
      *(f16x4*)((f16*)p.out + (int64_t)m * p.ldo + n) = o;
    }
  }
}

static int launch_splitk_reduce(const GemmK& k, int splitk, const float* ws, hipStream_t s) {
  GemmK kr = k;
  kr.split_stride = (int64_t)k.M * k.N;
  const int64_t nvec = (int64_t)k.M * (k.N >> 2);
  int64_t rb = (nvec + 255) / 256;
  if (rb > 4096) rb = 4096;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)rb), dim3(256), 0, s, kr, ws, splitk);
  GCD_CHECK_LAUNCH();
  return 0;
}

template <int MODE, int BF = 0>
int launch_pp_splitk(const GemmK& k, int splitk, float* ws, hipStream_t s) {
  static GcdPerDeviceOnce attr_once;
  auto fn = gemm_pp_kernel<MODE, 16384 + BF>;
  GCD_CHECK_HIP(attr_once.opt_in((const void*)fn, PP_SMEM_LAUNCH));
  GemmK kk = k;
  kk.tiles_m = (k.M + PP_BM - 1) / PP_BM;
  kk.tiles_n = (k.N + PP_BN - 1) / PP_BN;
  kk.splitk = splitk;
  kk.split_stride = (int64_t)k.M * k.N;
  // the GEMM pass writes plain partial sums [splitk][M][N]; the epilogue inputs go to the reduce pass
  kk.out = ws;
  kk.ldo = k.N;
  kk.out_kind = GCD_OUT_F32;
  kk.bias = kk.rowvec = kk.R1 = kk.R2 = kk.frame_alpha = nullptr;
  kk.s_acc = 1.0f;
  kk.ln_out = nullptr;
  const int64_t nblk = (int64_t)kk.tiles_m * kk.tiles_n * splitk;
  hipLaunchKernelGGL(fn, dim3((unsigned)nblk), dim3(512), PP_SMEM_LAUNCH, s, kk);
  GCD_CHECK_LAUNCH();
  return launch_splitk_reduce(k, splitk, ws, s);
}

}  // namespace

// gemm_p8.hip: the 8-phase 16x16x32 K loop on the same tile (GCD_TUNE_GEMM_IMPL = 3 keeps this file's ring kernel)
bool gcd_gemm_p8_supported(const GemmK& k, int mode);
int gcd_gemm_p8_launch(const GemmK& k, int mode, bool persist, hipStream_t s);
int gcd_gemm_p8_launch_splitk(const GemmK& k, int mode, int splitk, float* ws, hipStream_t s);
// gemm_p8x.hip: 256 x 160 tiles, the fp16-output epilogues interleaved into the next tile's K loop
bool gcd_gemm_p8x_supported(const GemmK& k, int mode);
int gcd_gemm_p8x_launch(const GemmK& k, hipStream_t s);

int gcd_gemm_pp_launch_splitk(const GemmK& k, int mode, int splitk, float* ws, hipStream_t s) {
  if (gcd_tune_get(GCD_TUNE_GEMM_IMPL) != 3 && gcd_gemm_p8_supported(k, mode)) {
    if (const int rc = gcd_gemm_p8_launch_splitk(k, mode, splitk, ws, s)) return rc;
    return launch_splitk_reduce(k, splitk, ws, s);
  }
  if (k.operand_bf16) {
    switch (mode) {
      case GCD_GEMM_PLAIN:
        return launch_pp_splitk<GCD_GEMM_PLAIN, 65536>(k, splitk, ws, s);
      case GCD_GEMM_CONV3X3:
        return launch_pp_splitk<GCD_GEMM_CONV3X3, 65536>(k, splitk, ws, s);
      default:
        return launch_pp_splitk<GCD_GEMM_TEMPORAL3, 65536>(k, splitk, ws, s);
    }
  }
  switch (mode) {
    case GCD_GEMM_PLAIN:
      return launch_pp_splitk<GCD_GEMM_PLAIN>(k, splitk, ws, s);
    case GCD_GEMM_CONV3X3:
      return launch_pp_splitk<GCD_GEMM_CONV3X3>(k, splitk, ws, s);
    default:
      return launch_pp_splitk<GCD_GEMM_TEMPORAL3>(k, splitk, ws, s);
  }
}

// Shapes the ping-pong kernel accepts (the caller has validated the descriptor already).
bool gcd_gemm_pp_supported(const GemmK& k, int mode) {
  if (k.K % 32 != 0 || k.N % 4 != 0) return false;
  if (mode != GCD_GEMM_PLAIN && k.Cin % 32 != 0) return false;
  if (k.out_kind == GCD_OUT_GEGLU && k.N % 32 != 0) return false;
  return true;
}

int gcd_gemm_pp_launch(const GemmK& k, int mode, hipStream_t s) {
#ifdef GCD_ABLATION_BUILD
  // Wrong-by-design ablation instantiations (no DMA / no LDS reads / no stores ...): compiled only
  // into tools/libgcd_amd_ablate.so (`python -m gcd_amd.csrc.build --ablation`) for tools/gemm_bench,
  // never into the product library gcd_amd/libgcd_amd.so.
  const int var = gcd_tune_get(GCD_TUNE_GEMM_IMPL) - 32;   // 32 + VAR (PLAIN only)
  if (var > 0 && mode == GCD_GEMM_PLAIN) {
    switch (var) {
      case 1: return launch_pp<GCD_GEMM_PLAIN, 1>(k, s);
      case 2: return launch_pp<GCD_GEMM_PLAIN, 2>(k, s);
      case 3: return launch_pp<GCD_GEMM_PLAIN, 3>(k, s);
      case 4: return launch_pp<GCD_GEMM_PLAIN, 4>(k, s);
      case 8: return launch_pp<GCD_GEMM_PLAIN, 8>(k, s);
      case 19: return launch_pp<GCD_GEMM_PLAIN, 19>(k, s);
      // pure-MFMA loop (19) + epilogue experiments (EV << 6)
      case 19 + 64: return launch_pp<GCD_GEMM_PLAIN, 19 + 64>(k, s);      // no residual loads
      case 19 + 128: return launch_pp<GCD_GEMM_PLAIN, 19 + 128>(k, s);    // no stores
      case 19 + 192: return launch_pp<GCD_GEMM_PLAIN, 19 + 192>(k, s);    // neither
      case 1024: return launch_pp<GCD_GEMM_PLAIN, 1024>(k, s);            // transposed epilogue forced
      case 2048 + 32768: return launch_pp<GCD_GEMM_PLAIN, 2048 + 32768>(k, s);   // persistent, no epilogue
      case 2048 + 128: return launch_pp<GCD_GEMM_PLAIN, 2048 + 128>(k, s);       // persistent, no stores
      case 2048 + 256: return launch_pp<GCD_GEMM_PLAIN, 2048 + 256>(k, s);       // persistent, GEGLU without GELU
      case 2048 + 384: return launch_pp<GCD_GEMM_PLAIN, 2048 + 384>(k, s);       // neither
      case 2048 + 512: return launch_pp<GCD_GEMM_PLAIN, 2048 + 512>(k, s);       // staged GEGLU epilogue, stores to an L2-resident scratch
      default: break;
    }
  }
#endif
  // the 8-phase K loop (gemm_p8.hip) wherever it applies; knob 3 keeps this file's 32-deep ring kernel.
  const int impl_knob = gcd_tune_get(GCD_TUNE_GEMM_IMPL);
#ifdef GCD_ABLATION_BUILD
  // Knob 10 (ablation build only): the GEGLU / q|k|v projections of the large grids on gemm_p8x.hip, the epilogue-under-
  // the-next-K-loop kernel — correct, and 15-35 % SLOWER than gemm_p8 (profiles/r04g_p8x_ab.txt).
  if (impl_knob == 10 && gcd_gemm_p8x_supported(k, mode)) return gcd_gemm_p8x_launch(k, s);
#endif
  if (impl_knob != 3 && gcd_gemm_p8_supported(k, mode)) return gcd_gemm_p8_launch(k, mode, true, s);
  if (k.ln_out || k.a_blocked) {
#ifdef GCD_ABLATION_BUILD
    const int64_t tiles = (int64_t)((k.M + PP_BM - 1) / PP_BM) * ((k.N + PP_BN - 1) / PP_BN);
    const bool persist = tiles > 256;
    if (k.ln_out) {   // validated by gcd_gemm_f16: PLAIN mode, N == 320
      if (mode != GCD_GEMM_PLAIN) {
        gcd_set_error("gcd_gemm_f16: fused LayerNorm is implemented for GCD_GEMM_PLAIN only");
        return 2;
      }
      return persist ? launch_pp<GCD_GEMM_PLAIN, 2048 + 8192>(k, s) : launch_pp<GCD_GEMM_PLAIN, 8192>(k, s);
    }
    return persist ? launch_pp<GCD_GEMM_PLAIN, 2048 + 32>(k, s) : launch_pp<GCD_GEMM_PLAIN, 32>(k, s);
#else
    // (gcd_gemm_f16 refuses these descriptors in the product build; records of experiments that did not pay)
    gcd_set_error("gcd_gemm_f16: fused LayerNorm / a_blocked exist in the ablation build only");
    return 2;
#endif
  }
  if (k.operand_bf16) {   // validated by gcd_gemm_f16: fp32 output, no colstats / LayerNorm / blocked layouts
    switch (mode) {
      case GCD_GEMM_PLAIN:
        return launch_pp<GCD_GEMM_PLAIN, 2048 + 65536>(k, s);
      case GCD_GEMM_CONV3X3:
        return launch_pp<GCD_GEMM_CONV3X3, 2048 + 65536>(k, s);
      default:
        return launch_pp<GCD_GEMM_TEMPORAL3, 2048 + 65536>(k, s);
    }
  }
  if (k.colstats) {   // validated by gcd_gemm_f16 (colstats_shape_ok)
    switch (mode) {
      case GCD_GEMM_PLAIN:
        return launch_pp<GCD_GEMM_PLAIN, 2048 + 4096>(k, s);
      case GCD_GEMM_CONV3X3:
        return launch_pp<GCD_GEMM_CONV3X3, 2048 + 4096>(k, s);
      default:
        return launch_pp<GCD_GEMM_TEMPORAL3, 2048 + 4096>(k, s);
    }
  }
  switch (mode) {
    case GCD_GEMM_PLAIN:
      return launch_pp<GCD_GEMM_PLAIN, 2048>(k, s);
    case GCD_GEMM_CONV3X3:
      return launch_pp<GCD_GEMM_CONV3X3, 2048>(k, s);
    default:
      return launch_pp<GCD_GEMM_TEMPORAL3, 2048>(k, s);
  }
}
