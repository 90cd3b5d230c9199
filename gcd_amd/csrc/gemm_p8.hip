// gemm_p8.hip — the 256 x 320 GEMM tile with an 8-phase K loop on v_mfma_f32_16x16x32_f16 (round 4): the K-loop
// structure of the guide's "256^2 8-phase template", measured on this pool against gemm_pp.hip's 32-deep ring of
// v_mfma_f32_32x32x16 (tools/gemm_8phase.cpp, profiles/r04a_gemm_8phase_calibration.txt, uniform random operands):
//   plain GEMMs K >= 5760: 1.04-1.37 PF/s vs 0.90-1.19;  bare MFMA streams of 2 waves / SIMD: the 16x16x32 shape sustains
//   1.97 PF/s where 32x32x16 sustains 1.64 (both 2.46-2.49 on zeros): the matrix pipe is POWER-limited on real data
//   and the 4-pass shape costs less energy per FLOP, on top of the loop structure (zeros: 1.84-1.94 vs 1.45-1.47 PF/s).
// Same contract, same tile, same wave grid (4 (M) x 2 (N) waves of 64 tokens x 160 channels), same LDS budget, same
// XCD-aware persistent tile walk and the same row-major epilogues (gemm_common.h, accumulator layout GcdAcc16) as
// gemm_pp.hip, which stays for K / Cin that are multiples of 32 but not 64, bf16 operands, the fused LayerNorm and the
// blocked hidden layout.
//
// Operand fetch: buffer_load_dwordx4 ... lds through one buffer descriptor per operand (raw, stride 0, num_records =
// the operand's extent).  The per-lane byte offset is range-checked by the hardware and an out-of-range lane gets
// ZEROS: rows past M / N, out-of-image conv taps and out-of-clip temporal taps simply carry an offset past the extent —
// no clamping, no zero page, no per-row validity state in the K loop.  (The K offset rides in the scalar offset, which
// the range check ignores; a valid row stays valid, an invalid one is marked with offset 0xFFFFFF00.)
//
// K loop.  BK = 64: one K-tile = A 256 rows x 128 B + W 320 rows x 128 B = 72 KB; two buffers (144 KB).  LDS rows are
// 128 B = eight 16-byte chunks; slot s of row r holds chunk s ^ (r & 7): the swizzle is applied to the per-lane SOURCE
// address of the LDS-DMA (global_load_lds_dwordx4 writes lane-linear: a piece = 8 rows x 128 B, every row one full line)
// and undone by the ds_read_b128 address — a 16-lane read group then touches 16 distinct 16-byte slots.
// A wave's 64 x 160 tile = 4 token blocks x 10 channel blocks of 16 x 16 (160 accumulator registers); per K-tile four
// phases, one C-quadrant each, ordered so that one operand stays in registers from phase to phase:
//     P1 (th0, cp0): read A th0 (4) + W cp0 (10) | stage RA0 of K-tile t+1 (2 pieces) | 20 MFMA
//     P2 (th1, cp0): read A th1 (4)              | stage RW0 of K-tile t+2 (3 | 2)     | 20 MFMA
//     P3 (th1, cp1): read W cp1 (10)             | stage RA1 of K-tile t+2 (2)         | 20 MFMA
//     P4 (th0, cp1): read A th0 (4)              | stage RW1 of K-tile t+2 (2 | 3), s_waitcnt vmcnt(7) | 20 MFMA
// each phase = { ds_reads, LDS-DMA, [vmcnt], lgkmcnt(0), s_barrier, s_setprio 1, MFMAs, s_setprio 0, s_barrier }, the two
// wave groups (waves 0-3 / 4-7 = one wave of every SIMD each) shifted by ONE barrier so that one wave of a SIMD feeds the
// matrix pipe while its partner loads.  Regions of a buffer: RA0 / RA1 = token halves th0 / th1 of every wave row,
// RW0 / RW1 = channel halves cp0 / cp1 of both wave columns.
//   WAR: a wave retires its ds_reads (lgkmcnt(0)) BEFORE the first barrier of the phase, so a region is re-staged one
//        phase after its last read by either group (RA0: read P1 + P4 -> staged P1; RW0: P1 -> P2; RA1: P2 -> P3;
//        RW1: P3 -> P4).
//   RAW: ONE counted wait per K-tile (P4): everything but the 7 newest pieces (RW0, RA1, RW1 of K-tile t+2) has landed,
//        i.e. K-tile t+1 is complete; the wait precedes a barrier that the other group passes before its P1 reads.
//        vmcnt never reaches 0 in steady state; 7-9 pieces per wave stay in flight across the barriers.
#include <type_traits>

#include "gemm_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int P8_BM = 256, P8_BN = 320;
constexpr int P8_A_BYTES = P8_BM * 128;                  // 32768
constexpr int P8_W_BYTES = P8_BN * 128;                  // 40960
constexpr int P8_BUF = P8_A_BYTES + P8_W_BYTES;          // 73728
constexpr int P8_SMEM = 2 * P8_BUF;                      // 147456
// Epilogue staging: 8 wave-private regions from the start of buffer 1 to 12 KB past the buffers, so that buffer 0
// stays free for the NEXT tile's first K-tile while a persistent workgroup is in its epilogue; then two buffers of
// 320 per-column addends (this tile's / the next's).
constexpr int P8_STAGE0 = P8_BUF;
constexpr int P8_BIAS0 = P8_STAGE0 + 8 * GCD_EPI_STAGE_BYTES;    // 159744
static_assert(P8_BIAS0 >= P8_SMEM, "staging must cover the tail of buffer 1");
constexpr int P8_SMEM_LAUNCH = P8_BIAS0 + 2 * 320 * 4;           // 162304 of the 163840 B of LDS
constexpr int P8_GROUP_M = 4;
// MODE 3 (internal, chosen by gcd_gemm_p8_launch for stride-1 3x3 convolutions whose rows are 64-token aligned):
// CONV3X3 with a HALO A panel.  K order (kh, cin-chunk, kw): one staged panel — the tile's image-row segments, each with
// one halo pixel on either side, padded to a multiple of 8 rows — serves the three kw K-tiles, whose fragments are read at
// row offsets 0 / 1 / 2: the A side of the LDS-DMA (and of the L2 -> LDS traffic) drops by 3.
constexpr int P8_CONV_HALO = 3;
constexpr int P8_PANEL = 288 * 128;                      // 36864: largest panel (four 64-pixel segments of 72 rows)
constexpr int P8_HALO_W0 = 2 * P8_PANEL;                 // 73728: the two W K-tile buffers follow the two panels
static_assert(P8_HALO_W0 + 2 * P8_W_BYTES <= P8_BIAS0, "halo layout must fit below the addend buffers");

#define P8_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define P8_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define P8_BAR()                         \
  do {                                   \
    __builtin_amdgcn_sched_barrier(0);   \
    __builtin_amdgcn_s_barrier();        \
    __builtin_amdgcn_sched_barrier(0);   \
  } while (0)

// VAR bits: 2048 persistent (256 workgroups walk the tiles), 4096 per-64-row column statistics (gcd_gemm_desc.colstats),
// 8192 bfloat16 operands (round 5: the fine-tune step's GEMMs in the type cfg4 names; fp32 outputs only),
// 16384 split-K (raw fp32 partial sums of K slice L % splitk).
// Round 5: the product library carries PERSISTENT instantiations only.  A launch of <= 256 tiles runs the same kernel on
// a grid of its tile count rounded up to a multiple of 8 (every XCD then holds at least its share of workgroups; the
// surplus ones find no tile and leave) — same tile -> workgroup map as the old one-workgroup-per-tile grid, half the code.
// Ablation bits (GCD_ABLATION_BUILD only, tools/gemm_bench_ablate; wrong results by design except 4 and 8):
//   1 no epilogue   2 no K loop (epilogue of zeros)   4 __syncthreads() (vmcnt(0) drain) at the end of a tile instead of
//   the LDS-only barrier   8 no cross-tile prefetch   16 / 32 / 64 flip the mode's default of: ds_reads before the
//   LDS-DMA of a phase (PLAIN default; the conv modes stage first, for their tap arithmetic) / no s_setprio around the
//   MFMA clusters (PLAIN default) / tile walk in groups of 8 instead of 4 M-tiles (PLAIN default).
//   128 / 256 (round 5): GEGLU epilogue without the GELU polynomial / without any arithmetic (profiles/r05_geglu_epilogue_valu.txt)
//   Measured (profiles/r04j_p8_variants.txt, two runs of 15): with all three L0 GEGLU 612-617 -> 500-509 us, L1 GEGLU
//   400-403 -> 376-381, L2 GEGLU 337-346 -> 321-327; FF-out / proj / q|k|v within +-1 %.
// EPI: which full-tile fast path of the epilogue this instantiation carries (the generic direct path is always there,
// for ragged tiles): 0 all of them (run-time choice per tile), 1 GEGLU, 2 fp16 rows without residuals, 3 fp32 rows
// with at most the first residual, 4 fp32 rows with both residuals (AlphaBlender), 5 fp16 rows with both residuals.  One path per kernel keeps the tile boundary free of register spills (whose reloads
// wait vmcnt(0) and thereby drain the epilogue's stores and the prefetched K-tile).
template <int MODE, int VAR, int EPI = 0>
__global__ __launch_bounds__(512, 2) void gemm_p8_kernel(const GemmK p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int grp = wave >> 2;            // 0 leads, 1 runs one barrier behind
  const int wm = wave & 3, wn = grp;
  constexpr bool PERSIST = (VAR & 2048) != 0;
  constexpr bool SPLITK = (VAR & 16384) != 0;
  constexpr bool STATS = (VAR & 4096) != 0;
  constexpr bool BF16 = (VAR & 8192) != 0;      // bfloat16 operands (v_mfma_f32_16x16x32_bf16): the fine-tune step, cfg4
  constexpr bool READS_FIRST = (MODE == GCD_GEMM_PLAIN) != ((VAR & 16) != 0);
  constexpr bool SETPRIO = (MODE != GCD_GEMM_PLAIN) != ((VAR & 32) != 0);
  constexpr int GM = ((MODE == GCD_GEMM_PLAIN) != ((VAR & 64) != 0)) ? 8 : P8_GROUP_M;

  // ---- XCD-aware, panel-sharing tile assignment (bijective for any grid; as gemm_pp.hip) ----
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

  // ---- per-tile state ----
  int m0 = 0, n0 = 0, kz = 0, s_begin = 0, nK = p.K >> 6;
  const int lrow = lane >> 3;
  const unsigned srcchunk = ((lane & 7) ^ (lane >> 3)) << 4;     // the 16-byte chunk of its row this lane fetches
  const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrcW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, (int)p.w_bytes, 0x00020000);
  constexpr unsigned OOB = 0xFFFFFF00u;
  // A pieces of this wave: region reg (0: RA0, 1: RA1), j = 0, 1: 8-row group g = 2 wave + j of the region's 16 groups,
  // tile rows 64 (g >> 2) + 32 reg + 8 (g & 3) + lrow.  Index r = 2 reg + j.
  // PLAIN: the four rows differ by wave-uniform multiples of lda: one per-lane offset (r = 0) + scalar deltas.
  // conv modes: a per-row offset of the current tap (OOB when the tap leaves the image / clip), re-derived from the
  // packed output coordinates when the K walk crosses a tap.
  constexpr bool HALO = MODE == P8_CONV_HALO;
  unsigned a_off[4] = {0, 0, 0, 0};   // (HALO: panel pieces 0-3 of this wave, for the kh of the panel being staged)
  unsigned a_pk[4] = {0, 0, 0, 0};    // TEMPORAL3: frame-in-clip << 26 | token
  int a_tap[2] = {0, 0}, a_c0[2] = {0, 0};     // K position of the next stage of each region (conv modes), block-uniform
  const unsigned a_d8 = (unsigned)(8 * p.lda * 2), a_d32 = (unsigned)(32 * p.lda * 2);
  // W pieces: group 0 waves stage 3 pieces of RW0 and 2 of RW1, group 1 waves 2 and 3 (20 8-row groups per region);
  // group g of a region = W tile rows 160 (g / 10) + 80 region + 8 (g % 10) .. + 7.
  const unsigned w_lane = (unsigned)lrow * (unsigned)p.K * 2u + srcchunk;
  int w_row[2][3];          // tile row of piece j of region reg
  unsigned w_roff[2][3];    // byte offset of row n0 + w_row in W (rows past N are out of range: zeros)
  float* lds_bias = (float*)(smem + P8_BIAS0);
  bool lds_bias_ok = false;
  bool alpha_uni = true;

  // CONV3X3 (Wo a multiple of 8, checked by gcd_gemm_p8_supported): the 8 rows of a piece are 8 consecutive x of ONE
  // image row, so (frame, y, x0) of a piece is wave-uniform — one packed SGPR per piece, frame << 21 | y << 11 | x0,
  // derived once per tile; at a tap change the row part of the address is scalar arithmetic and a lane only adds its
  // own x (lane >> 3) and range-checks it.
  unsigned a_pc[5] = {0, 0, 0, 0, 0};
  // HALO geometry: segments of w = min(Wo, 256) pixels, ws = w + 8 panel rows each; piece q = wave + 8 j
  const int h_w = HALO ? min(p.Wo, 256) : 0, h_ws = h_w + 8;
  const int h_npieces = HALO ? (256 / max(h_w, 1)) * h_ws / 8 : 0;
  const int h_nck = HALO ? p.Cin >> 6 : 1;                 // 64-channel chunks per tap
  const bool h_has5 = wave + 32 < h_npieces;
  int h_kh = 0, h_ck = 0;                                  // (kh, chunk) of the NEXT panel to stage
  int h_wkh = 0, h_wck = 0, h_wkw = 0;                     // (kh, chunk, kw) of the next W K-tile to stage
  // (the lane id is made opaque wherever a loop-invariant per-lane value would otherwise be hoisted out of the K loop
  //  and then SPILLED: a reload waits vmcnt(0) and drains the DMA pipeline)
  auto lane_now = [&]() -> unsigned {
    unsigned ln = __lane_id();
    asm volatile("" : "+v"(ln));
    return ln;
  };
  auto halo_piece_off = [&](int j, int kh) -> unsigned {   // per-lane offset of panel piece j for tap row kh
    const unsigned ln = lane_now();
    const unsigned sc = ((ln & 7) ^ (ln >> 3)) << 4;
    const unsigned lda2 = (unsigned)(p.lda * 2);
    const unsigned pc = a_pc[j];
    const int x0p1 = pc & 2047, y = (pc >> 11) & 1023, fr = (int)(pc >> 21);
    const int iy = y + kh - 1, ix = x0p1 - 1 + (int)(ln >> 3);
    const bool row_ok = iy >= 0 && iy < p.Hi && fr < p.a_frames;
    const unsigned rowbase = (unsigned)((fr * p.Hi + iy) * p.Wi) * lda2;
    return (row_ok && (unsigned)ix < (unsigned)p.Wi) ? rowbase + (unsigned)ix * lda2 + sc : OOB;
  };
  auto halo_set_kh = [&](int kh) {   // pieces 0-3 keep their offsets in registers; the 5th (1-4 waves) is derived when staged
#pragma unroll
    for (int j = 0; j < 4; ++j) a_off[j] = halo_piece_off(j, kh);
  };
  auto unpack_set = [&](int r, int tap) {
    if (MODE == GCD_GEMM_CONV3X3) {
      const unsigned pc = a_pc[r];
      const int x0 = pc & 2047, y = (pc >> 11) & 1023, fr = (int)(pc >> 21);
      const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
      const unsigned ln = __lane_id();      // (re-derived here: nothing of this block stays live through the K loop)
      const int xv = x0 + (int)(ln >> 3);
      const unsigned sc = ((ln & 7) ^ (ln >> 3)) << 4;
      int iy, ix;
      bool row_ok, ok;
      if (p.up) {
        const int uy = y + dy, ux = xv + dx;
        row_ok = uy >= 0 && uy < p.Ho;
        ok = (unsigned)ux < (unsigned)p.Wo;
        iy = uy >> 1;
        ix = ux >> 1;
      } else {
        iy = y * p.stride + dy + p.asym;
        ix = xv * p.stride + dx + p.asym;
        row_ok = iy >= 0 && iy < p.Hi;
        ok = (unsigned)ix < (unsigned)p.Wi;
      }
      row_ok = row_ok && fr < p.a_frames;
      const unsigned lda2 = (unsigned)(p.lda * 2);
      const unsigned rowbase = (unsigned)((fr * p.Hi + iy) * p.Wi) * lda2;     // wave-uniform
      a_off[r] = (ok && row_ok) ? rowbase + (unsigned)ix * lda2 + sc : OOB;
    } else if (MODE == GCD_GEMM_TEMPORAL3) {
      const unsigned pk = a_pk[r];
      const int m = pk & 0x3ffffff, tt = (int)(pk >> 26) + tap - 1;
      const bool ok = tt >= 0 && tt < p.T && m < p.M;
      a_off[r] = ok ? (unsigned)(m + (tap - 1) * p.HW) * (unsigned)(p.lda * 2) + srcchunk : OOB;
    }
  };
  auto set_tap = [&](int reg, int tap) {
    unpack_set(2 * reg, tap);
    unpack_set(2 * reg + 1, tap);
  };

  // Tile Lx of the linear order -> (m0, n0), DMA source offsets, staged epilogue addends.
  auto set_tile = [&](int Lx) {
    int tile_m, tile_n;
    if (!SPLITK && (p.sched & 1)) {
      // gcd_gemm_desc.sched bit 0: the XCD's share [start, start + cnt) of the tile order is walked from its END —
      // position Lx of the walk is tile 2 start + cnt - 1 - Lx (pure scheduling; the share is re-derived here, once
      // per tile, so that nothing of it stays live through the K loop)
      const int nblk = PERSIST ? p.tiles_m * p.tiles_n : (int)gridDim.x;
      const int q = nblk >> 3, r = nblk & 7, xcd = blockIdx.x & 7;
      const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
      Lx = 2 * start + q + (xcd < r ? 1 : 0) - 1 - Lx;
    }
    kz = SPLITK ? Lx % p.splitk : 0;
    {
      const int Lt = SPLITK ? Lx / p.splitk : Lx;
      const int per_group = GM * p.tiles_n;
      const int gi = Lt / per_group;
      const int rem = Lt - gi * per_group;
      const int m_first = gi * GM;
      const int gm = min(GM, p.tiles_m - m_first);
      tile_n = rem / gm;
      tile_m = m_first + rem - tile_n * gm;
    }
    m0 = tile_m * P8_BM;
    n0 = tile_n * P8_BN;
    s_begin = 0;
    nK = p.K >> 6;
    if (SPLITK) {
      const int per = (nK + p.splitk - 1) / p.splitk;
      s_begin = kz * per;
      nK = max(0, min(per, nK - s_begin));
    }
    if (VAR & 2) nK = 0;
    if (MODE == GCD_GEMM_PLAIN) {
      const int g = 2 * wave;
      const int m = m0 + 64 * (g >> 2) + 8 * (g & 3) + lrow;
      a_off[0] = (unsigned)m * (unsigned)(p.lda * 2) + srcchunk;    // rows >= M land past the extent: zeros
    } else if (HALO) {
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const int L0r = 8 * (wave + 8 * j);                 // first panel row of the piece
        const int seg = L0r / h_ws, i0 = L0r - seg * h_ws;  // segment, row within it (row i <-> pixel x = i - 1)
        const int ms = m0 + seg * h_w;                      // first output token of the segment
        const int hw = p.Ho * p.Wo;
        const int fr = ms / hw;
        const int rem = ms - fr * hw;
        const int y = rem / p.Wo;
        a_pc[j] = __builtin_amdgcn_readfirstlane(
            (int)(((unsigned)fr << 21) | ((unsigned)y << 11) | (unsigned)(rem - y * p.Wo + i0)));   // x0 + 1
      }
      h_kh = h_ck = 0;
      h_wkh = h_wck = h_wkw = 0;
      halo_set_kh(0);
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int g = 2 * wave + (r & 1);
        const int m = m0 + 64 * (g >> 2) + 32 * (r >> 1) + 8 * (g & 3) + lrow;
        if (MODE == GCD_GEMM_TEMPORAL3) a_pk[r] = ((unsigned)((m / p.HW) % p.T) << 26) | (unsigned)m;
        if (MODE == GCD_GEMM_CONV3X3) {
          const int mp = m0 + 64 * (g >> 2) + 32 * (r >> 1) + 8 * (g & 3);     // first row of the piece: wave-uniform
          const int hw = p.Ho * p.Wo;
          const int fr = mp / hw;          // fr >= a_frames for pieces past M: unpack_set marks them out of range
          const int rem = mp - fr * hw;
          const int y = rem / p.Wo;
          a_pc[r] = ((unsigned)fr << 21) | ((unsigned)y << 11) | (unsigned)(rem - y * p.Wo);
        }
      }
      const int k0 = s_begin * 64;
      a_tap[0] = a_tap[1] = k0 / p.Cin;
      a_c0[0] = a_c0[1] = k0 - a_tap[0] * p.Cin;
      set_tap(0, a_tap[0]);
      set_tap(1, a_tap[1]);
    }
#pragma unroll
    for (int reg = 0; reg < 2; ++reg)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        // RW0: group 0 waves own groups 3w .. 3w+2, group 1 waves 12 + 2w' ..; RW1: group 0 2w .. 2w+1, group 1 8 + 3w' ..
        int g;
        if (reg == 0) g = grp == 0 ? 3 * wave + j : 12 + 2 * (wave - 4) + j;
        else g = grp == 0 ? 2 * wave + j : 8 + 3 * (wave - 4) + j;
        g = g < 20 ? g : 19;        // (the unused third slot of a 2-piece wave)
        const int row = 160 * (g / 10) + 80 * reg + 8 * (g % 10);
        w_row[reg][j] = row;
        w_roff[reg][j] = (unsigned)(n0 + row) * (unsigned)p.K * 2u;
      }
    if (!SPLITK) {
      const int m_last = min(m0 + P8_BM, p.M) - 1;
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

  // ---- staging ----
  auto bload = [&](__amdgpu_buffer_rsrc_t rs, unsigned voff, int soff, char* lds_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (GCD_AS3 void*)lds_wave_base, 16, (int)voff, soff, 0, 0);
  };
  auto stage_A = [&](int kt, int reg) {
    if (kt < nK) {
      const int g = 2 * wave;
      char* dst = smem + (kt & 1) * P8_BUF + (64 * (g >> 2) + 32 * reg + 8 * (g & 3)) * 128;
      if (MODE == GCD_GEMM_PLAIN) {
        const int ko = (s_begin + kt) * 128;
        const unsigned o = a_off[0] + (reg ? a_d32 : 0u);
        bload(rsrcA, o, ko, dst);
        bload(rsrcA, o + a_d8, ko, dst + 1024);
      } else {
        const int c2 = a_c0[reg] * 2;
        bload(rsrcA, a_off[2 * reg], c2, dst);
        bload(rsrcA, a_off[2 * reg + 1], c2, dst + 1024);
        a_c0[reg] += 64;
        if (a_c0[reg] == p.Cin) {
          a_c0[reg] = 0;
          ++a_tap[reg];
          set_tap(reg, a_tap[reg]);
        }
      }
    }
  };
  auto stage_W = [&](int kt, int reg, auto CNT) {
    constexpr int cnt = decltype(CNT)::value;
    if (kt < nK) {
      const int ko = (s_begin + kt) * 128;
      char* dst = smem + (kt & 1) * P8_BUF + P8_A_BYTES;
#pragma unroll
      for (int j = 0; j < cnt; ++j) bload(rsrcW, w_lane + w_roff[reg][j], ko, dst + w_row[reg][j] * 128);
    }
  };

  // HALO: one piece (j) of the next panel (index pn, buffer pn & 1); the panel counters advance with piece 3
  auto halo_stage_piece = [&](int pn, int npanels, int j) {
    if (pn < npanels && wave + 8 * j < h_npieces)
      bload(rsrcA, j < 4 ? a_off[j < 4 ? j : 0] : halo_piece_off(4, h_kh), h_ck * 128,
            smem + (pn & 1) * P8_PANEL + (wave + 8 * j) * 1024);
  };
  auto halo_next_panel = [&](int pn, int npanels) {   // before the first piece of panel pn: its kh's offsets
    if (pn < npanels && pn > 0) {
      if (++h_ck == h_nck) {
        h_ck = 0;
        ++h_kh;
        halo_set_kh(h_kh);
      }
    }
  };
  auto halo_stage_W = [&](int kt, int nkt, int reg, auto CNT) {
    constexpr int cnt = decltype(CNT)::value;
    if (kt < nkt) {
      const int ko = ((h_wkh * 3 + h_wkw) * p.Cin + h_wck * 64) * 2;
      char* dst = smem + P8_HALO_W0 + (kt & 1) * P8_W_BYTES;
#pragma unroll
      for (int j = 0; j < cnt; ++j) bload(rsrcW, w_lane + w_roff[reg][j], ko, dst + w_row[reg][j] * 128);
      if (reg == 1) {            // both regions of this K-tile are out: the W cursor moves on
        if (++h_wkw == 3) {
          h_wkw = 0;
          if (++h_wck == h_nck) {
            h_wck = 0;
            ++h_wkh;
          }
        }
      }
    }
  };

  // ---- fragment read addresses (per lane, within a buffer): row (lane & 15) of a 16-row block, logical chunk
  //      ks * 4 + (lane >> 4), physical slot = chunk ^ (row & 7) ----
  //      (k-step 1 = the same address with bit 6 flipped: chunk 4 + q = chunk q ^ 4)
  const int rdA0 = (64 * wm + (lane & 15)) * 128 + (((lane >> 4) ^ (lane & 7)) << 4);
  const int rdW0 = (HALO ? 0 : P8_A_BYTES) + (160 * wn + (lane & 15)) * 128 + (((lane >> 4) ^ (lane & 7)) << 4);
  // HALO: token (seg, x) of tap kw sits in panel row seg * ws + x + kw; the wave's 64 tokens are 64 consecutive x.
  // One base register; the kw-dependent row / slot part is re-derived per read (5 VALU, kw is a compile-time constant)
  int rdAh0 = 0;
  if (HALO) {
    const int seg_w = (64 * wm) / h_w;
    rdAh0 = (seg_w * h_ws + (64 * wm - seg_w * h_w) + (lane & 15)) * 128;
  }

  // Cross-tile prefetch (persistent PLAIN kernels, as gemm_pp.hip): the NEXT tile's set_tile + K-tile 0 are issued
  // BEFORE the epilogue (buffer 0 is not staging space); the first counted wait of the next K loop admits the
  // epilogue's `nst` stores (a LOWER bound of what the epilogue issued: stores share the in-order vmcnt on gfx9).
  constexpr bool XPF = PERSIST && MODE == GCD_GEMM_PLAIN && !SPLITK && !(VAR & 8);
  int nst = 0;

  auto run_tiles = [&](auto GRP) {
    constexpr int G = decltype(GRP)::value;
    using C0 = std::integral_constant<int, G == 0 ? 3 : 2>;   // pieces of RW0 / RW1 this wave stages
    using C1 = std::integral_constant<int, G == 0 ? 2 : 3>;
    auto prologue_a = [&]() {      // K-tile 0, complete (9 pieces)
      stage_A(0, 0);
      stage_W(0, 0, C0{});
      stage_A(0, 1);
      stage_W(0, 1, C1{});
    };
    auto prologue_b = [&]() {      // what the steady state stages in P2 .. P4 of "K-tile -1" (7 pieces)
      stage_W(1, 0, C0{});
      stage_A(1, 1);
      stage_W(1, 1, C1{});
    };
    if (XPF && L < L_end) {      // (a surplus workgroup of a rounded-up grid has no tile)
      set_tile(L);
      prologue_a();
    }
    for (; L < L_end; L += L_step) {
      const int npanels = HALO ? nK / 3 : 0;     // (set below by set_tile for HALO: nK = 9 Cin / 64)
      if constexpr (HALO) {
        set_tile(L);
#pragma unroll
        for (int j = 0; j < 5; ++j) halo_stage_piece(0, nK / 3, j);    // panel 0, complete
        halo_stage_W(0, nK, 0, C0{});
        halo_stage_W(0, nK, 1, C1{});
        halo_stage_W(1, nK, 0, C0{});
        halo_stage_W(1, nK, 1, C1{});
      } else {
        if (!XPF) {
          set_tile(L);
          prologue_a();
        }
        prologue_b();
      }
      (void)npanels;

      GcdAcc16 acc;
#pragma unroll
      for (int c = 0; c < 10; ++c)
#pragma unroll
        for (int tb = 0; tb < 4; ++tb) acc[c][tb] = (f32x4){0.f, 0.f, 0.f, 0.f};
      f16x8 af[2][2], wf[5][2];

      auto read_A = [&](const char* buf, int th) {
        const char* b0 = buf + rdA0 + th * 4096;
        const char* b1 = buf + (rdA0 ^ 64) + th * 4096;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          af[i][0] = *(const f16x8*)(b0 + i * 2048);
          af[i][1] = *(const f16x8*)(b1 + i * 2048);
        }
      };
      auto read_W = [&](const char* buf, int cp) {
        const char* b0 = buf + rdW0 + cp * 10240;
        const char* b1 = buf + (rdW0 ^ 64) + cp * 10240;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
          wf[i][0] = *(const f16x8*)(b0 + i * 2048);
          wf[i][1] = *(const f16x8*)(b1 + i * 2048);
        }
      };
      auto mma = [&](auto TH, auto CP) {
        constexpr int th = decltype(TH)::value, cp = decltype(CP)::value;
        if (SETPRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int i = 0; i < 5; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              if constexpr (BF16)
                acc[5 * cp + i][2 * th + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                    __builtin_bit_cast(bf16x8, wf[i][ks]), __builtin_bit_cast(bf16x8, af[j][ks]), acc[5 * cp + i][2 * th + j], 0,
                    0, 0);
              else
                acc[5 * cp + i][2 * th + j] =
                    __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[i][ks], af[j][ks], acc[5 * cp + i][2 * th + j], 0, 0, 0);
        if (SETPRIO) __builtin_amdgcn_s_setprio(0);
      };
      using I0 = std::integral_constant<int, 0>;
      using I1 = std::integral_constant<int, 1>;

      if constexpr (HALO) {
        const int np = nK / 3;
        auto read_Ah = [&](const char* panel, int kw, int th) {
          const unsigned ln = lane_now();
          const int a0 = rdAh0 + kw * 128 + (int)(((ln >> 4) ^ ((ln + kw) & 7)) << 4);
          const char* b0 = panel + a0 + th * 4096;
          const char* b1 = panel + (a0 ^ 64) + th * 4096;
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            af[i][0] = *(const f16x8*)(b0 + i * 2048);
            af[i][1] = *(const f16x8*)(b1 + i * 2048);
          }
        };
        // one K-tile (panel P, tap column KW): the four phases of the plain loop, A from the panel at row offset KW;
        // W regions re-staged as there (RW0 in P2, RW1 in P4, K-tile t+2); in the KW = 0 K-tile the NEXT panel is staged,
        // one piece per phase (piece 4 rides with piece 0): 8+ phases before its first read
        auto ktile = [&](int P, auto KWT) {
          constexpr int KW = decltype(KWT)::value;
          const int kt = 3 * P + KW;
          const char* panel = smem + (P & 1) * P8_PANEL;
          const char* wbuf = smem + P8_HALO_W0 + (kt & 1) * P8_W_BYTES;
          // P1 (th0, cp0)
          if (KW == 0) {
            halo_next_panel(P + 1, np);
            halo_stage_piece(P + 1, np, 0);
            halo_stage_piece(P + 1, np, 4);
          }
          __builtin_amdgcn_sched_barrier(0);
          read_Ah(panel, KW, 0);
          read_W(wbuf, 0);
          P8_LGKM0();
          P8_BAR();
          mma(I0{}, I0{});
          P8_BAR();
          // P2 (th1, cp0)
          if (KW == 0) halo_stage_piece(P + 1, np, 1);
          halo_stage_W(kt + 2, nK, 0, C0{});
          __builtin_amdgcn_sched_barrier(0);
          read_Ah(panel, KW, 1);
          P8_LGKM0();
          P8_BAR();
          mma(I1{}, I0{});
          P8_BAR();
          // P3 (th1, cp1)
          if (KW == 0) halo_stage_piece(P + 1, np, 2);
          __builtin_amdgcn_sched_barrier(0);
          read_W(wbuf, 1);
          P8_LGKM0();
          P8_BAR();
          mma(I1{}, I1{});
          P8_BAR();
          // P4 (th0, cp1)
          if (KW == 0) halo_stage_piece(P + 1, np, 3);
          halo_stage_W(kt + 2, nK, 1, C1{});
          __builtin_amdgcn_sched_barrier(0);
          read_Ah(panel, KW, 0);
          // W K-tile kt+1 (and with it everything older, the next panel included) has landed; in flight stay the 5 W
          // pieces of kt+2 and, in a KW = 0 K-tile, the 4-5 panel pieces just issued
          if (kt + 2 >= nK) {
            P8_VMCNT(0);
          } else if (KW == 0 && P + 1 < np) {
            if (h_has5) P8_VMCNT(10);
            else P8_VMCNT(9);
          } else {
            P8_VMCNT(5);
          }
          P8_LGKM0();
          P8_BAR();
          mma(I0{}, I1{});
          P8_BAR();
        };
        P8_VMCNT(5);   // panel 0 and W K-tile 0 have landed (the 5 pieces of W K-tile 1 may still fly)
        P8_BAR();
        if (G == 1) P8_BAR();
        for (int P = 0; P < np; ++P) {
          ktile(P, std::integral_constant<int, 0>{});
          ktile(P, std::integral_constant<int, 1>{});
          ktile(P, std::integral_constant<int, 2>{});
        }
      } else {
      // K-tile 0 has landed (the 7 pieces of prologue_b [+ nst epilogue stores] may still fly)
      if (nK > 1) {
        if (nst == 40) P8_VMCNT(47);
        else if (nst == 20) P8_VMCNT(27);
        else if (nst == 10) P8_VMCNT(17);
        else P8_VMCNT(7);
      } else {
        P8_VMCNT(0);
      }
      P8_BAR();
      if (G == 1) P8_BAR();   // the one-barrier stagger

      for (int kt = 0; kt < nK; ++kt) {
        const char* buf = smem + (kt & 1) * P8_BUF;
        // (each phase stages FIRST: the tap arithmetic of the conv modes then runs while no fragment is live)
        // P1 (th0, cp0)
        if (!READS_FIRST) stage_A(kt + 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        read_A(buf, 0);
        read_W(buf, 0);
        if (READS_FIRST) {
          __builtin_amdgcn_sched_barrier(0);
          stage_A(kt + 1, 0);
        }
        P8_LGKM0();
        P8_BAR();
        mma(I0{}, I0{});
        P8_BAR();
        // P2 (th1, cp0)
        if (!READS_FIRST) stage_W(kt + 2, 0, C0{});
        __builtin_amdgcn_sched_barrier(0);
        read_A(buf, 1);
        if (READS_FIRST) {
          __builtin_amdgcn_sched_barrier(0);
          stage_W(kt + 2, 0, C0{});
        }
        P8_LGKM0();
        P8_BAR();
        mma(I1{}, I0{});
        P8_BAR();
        // P3 (th1, cp1)
        if (!READS_FIRST) stage_A(kt + 2, 1);
        __builtin_amdgcn_sched_barrier(0);
        read_W(buf, 1);
        if (READS_FIRST) {
          __builtin_amdgcn_sched_barrier(0);
          stage_A(kt + 2, 1);
        }
        P8_LGKM0();
        P8_BAR();
        mma(I1{}, I1{});
        P8_BAR();
        // P4 (th0, cp1)
        if (!READS_FIRST) stage_W(kt + 2, 1, C1{});
        __builtin_amdgcn_sched_barrier(0);
        read_A(buf, 0);
        if (READS_FIRST) {
          __builtin_amdgcn_sched_barrier(0);
          stage_W(kt + 2, 1, C1{});
        }
        if (kt + 2 < nK) P8_VMCNT(7);   // K-tile kt+1 complete; RW0 / RA1 / RW1 of kt+2 (7 pieces) stay in flight
        else P8_VMCNT(0);
        P8_LGKM0();
        P8_BAR();
        mma(I0{}, I1{});
        P8_BAR();
      }
      }   // !HALO
      if (G == 0) P8_BAR();   // pairs with the stagger: every wave's reads of this tile are retired behind it

      // ---- epilogue (gemm_common.h) ----
      int wm_base = m0 + 64 * wm, wn_base = n0 + 160 * wn, elane = lane;
      const int e_m0 = m0;
      const bool e_bias_ok = lds_bias_ok, e_alpha_uni = alpha_uni;
      const float* const e_bias = lds_bias;
      char* const e_stage = smem + P8_STAGE0 + wave * GCD_EPI_STAGE_BYTES;
      const int e_kz = kz;
      if (XPF && L + L_step < L_end) {   // the next tile's addresses, addends (other buffer) and first K-tile
        lds_bias = (float*)(smem + P8_BIAS0) + (e_bias == (const float*)(smem + P8_BIAS0) ? 320 : 0);
        set_tile(L + L_step);
        prologue_a();
      }
      nst = 0;
      asm volatile("" : "+s"(wm_base), "+s"(wn_base), "+v"(elane));
      if constexpr ((VAR & 1) != 0) {   // ablation: K loop only
#pragma unroll
        for (int c = 0; c < 10; ++c)
#pragma unroll
          for (int tb = 0; tb < 4; ++tb) asm volatile("" ::"v"(acc[c][tb]));
      } else if constexpr (SPLITK) {   // raw fp32 partial sums of this K slice
        GemmK q = p;
        q.out = (float*)p.out + (int64_t)e_kz * p.split_stride;
        gcd_epilogue_64x160(q, acc, wm_base, wn_base, elane);
      } else if constexpr (STATS) {
        // fp32 rows + per-64-row column statistics for the next GroupNorm (gcd_gemm_desc.colstats): gcd_gemm_f16
        // validated that EVERY tile of the launch is full, has tile-uniform per-frame vectors / blend factors and at
        // most one residual
        float sa = p.s_acc, sr1 = p.s_r1;
        if (p.frame_alpha) {
          const float al = p.frame_alpha[e_m0 / p.rows_per_alpha];
          sa = 1.0f - al;
          if (p.r1_blend) sr1 *= 1.0f - al;
        }
        const float* lb = e_bias + 160 * wn;
        p.R1 ? gcd_epi_f32_rows_full<true, false, false, true>(p, acc, wm_base, wn_base, elane, lb, e_stage, sa, sr1, 0.f)
             : gcd_epi_f32_rows_full<false, false, false, true>(p, acc, wm_base, wn_base, elane, lb, e_stage, sa, sr1, 0.f);
        nst = 40;
      } else {
        const bool full = wm_base + 64 <= p.M && wn_base + 160 <= p.N && e_bias_ok;
        const float* lb = e_bias + 160 * wn;
        constexpr bool E_GEGLU = EPI == 0 || EPI == 1, E_F16 = EPI == 0 || EPI == 2, E_F32 = EPI == 0 || EPI == 3,
                       E_R2 = EPI == 0 || EPI == 4, E_R2H = EPI == 0 || EPI == 5;
        if (E_GEGLU && full && p.out_kind == GCD_OUT_GEGLU && (p.ldo & 7) == 0 && !p.out_blocked) {
#ifdef GCD_ABLATION_BUILD
          if constexpr ((VAR & 128) != 0) gcd_epi_geglu_rows_full<GcdAcc16, 1>(p, acc, wm_base, wn_base, elane, lb, e_stage);
          else if constexpr ((VAR & 256) != 0) gcd_epi_geglu_rows_full<GcdAcc16, 2>(p, acc, wm_base, wn_base, elane, lb, e_stage);
          else
#endif
            gcd_epi_geglu_rows_full(p, acc, wm_base, wn_base, elane, lb, e_stage);
          nst = 10;
        } else if (E_F16 && full && p.out_kind == GCD_OUT_F16 && !p.R1 && !p.R2 && !p.frame_alpha && (p.ldo & 7) == 0) {
          gcd_epi_f16_rows_full(p, acc, wm_base, wn_base, elane, lb, e_stage);
          nst = 20;
        } else if (E_R2H && full && e_alpha_uni && p.out_kind == GCD_OUT_F16 && p.R1 && p.R2 && (p.ldo & 3) == 0) {
          float sa = p.s_acc, sr1 = p.s_r1, sr2 = p.s_r2;
          if (p.frame_alpha) {
            const float al = p.frame_alpha[e_m0 / p.rows_per_alpha];
            sa = 1.0f - al;
            sr2 = al;
            if (p.r1_blend) sr1 *= 1.0f - al;
          }
          gcd_epi_f32_rows_full<true, true, true>(p, acc, wm_base, wn_base, elane, lb, e_stage, sa, sr1, sr2);
          nst = 40;
        } else if ((E_F32 || E_R2) && full && e_alpha_uni && p.out_kind == GCD_OUT_F32 &&
                   (p.R2 ? (E_R2 && (EPI == 0 || p.R1)) : E_F32)) {
          float sa = p.s_acc, sr1 = p.s_r1, sr2 = p.s_r2;
          if (p.frame_alpha) {
            const float al = p.frame_alpha[e_m0 / p.rows_per_alpha];
            sa = 1.0f - al;
            sr2 = al;
            if (p.r1_blend) sr1 *= 1.0f - al;
          }
          if constexpr (E_R2) {
            if (p.R2) {
              if (EPI != 0 || p.R1) gcd_epi_f32_rows_full<true, true>(p, acc, wm_base, wn_base, elane, lb, e_stage, sa, sr1, sr2);
              else gcd_epi_f32_rows_full<false, true>(p, acc, wm_base, wn_base, elane, lb, e_stage, sa, sr1, sr2);
            }
          }
          if constexpr (E_F32) {
            if (!p.R2) {
              p.R1 ? gcd_epi_f32_rows_full<true, false>(p, acc, wm_base, wn_base, elane, lb, e_stage, sa, sr1, sr2)
                   : gcd_epi_f32_rows_full<false, false>(p, acc, wm_base, wn_base, elane, lb, e_stage, sa, sr1, sr2);
            }
          }
          nst = 40;
        } else {
          gcd_epilogue_64x160(p, acc, wm_base, wn_base, elane);
        }
      }
      if (!XPF) nst = 0;   // without the prefetch every DMA piece of the next tile is younger than the stores
      if (PERSIST) {
        // The epilogue's LDS staging reads are retired (lgkmcnt) before the next tile's K-tile 1 pieces land in buffer 1;
        // its global STORES and the prefetched K-tile stay in flight across this barrier — __syncthreads() would wait
        // vmcnt(0) here (an LDS-DMA is pending) and drain both (measured: L0 GEGLU 665 -> 597 us, L1 431 -> 410,
        // profiles/r04e_p8_ablations.txt).  VAR bit 4 keeps the draining form for the A/B.
        if constexpr ((VAR & 4) != 0) {
          __syncthreads();
        } else {
          P8_LGKM0();
          P8_BAR();
        }
      }
    }   // tile loop
  };
  if (grp == 0) run_tiles(std::integral_constant<int, 0>{});
  else run_tiles(std::integral_constant<int, 1>{});
}

// Extents of the two operands in bytes (the buffer descriptors' num_records): the last row ends after its K (or Cin)
// channels, whatever the row stride.
template <int MODE>
void p8_extents(GemmK& kk) {
  int64_t rows = kk.M, width = kk.K;
  kk.a_frames = 0;
  if (MODE == GCD_GEMM_CONV3X3 || MODE == P8_CONV_HALO) {
    kk.a_frames = kk.M / (kk.Ho * kk.Wo);
    rows = (int64_t)kk.a_frames * kk.Hi * kk.Wi;
    width = kk.Cin;
  } else if (MODE == GCD_GEMM_TEMPORAL3) {
    width = kk.Cin;
  }
  kk.a_bytes = (uint32_t)(((rows - 1) * kk.lda + width) * 2);
  kk.w_bytes = (uint32_t)((int64_t)kk.N * kk.K * 2);
}

template <int MODE, int VAR = 0, int EPI = 0>
int launch_p8(const GemmK& k, hipStream_t s) {
  static GcdPerDeviceOnce attr_once;
  auto fn = gemm_p8_kernel<MODE, VAR, EPI>;
  GCD_CHECK_HIP(attr_once.opt_in((const void*)fn, P8_SMEM_LAUNCH));
  GemmK kk = k;
  p8_extents<MODE>(kk);
  kk.tiles_m = (k.M + P8_BM - 1) / P8_BM;
  kk.tiles_n = (k.N + P8_BN - 1) / P8_BN;
  int64_t nblk = (int64_t)kk.tiles_m * kk.tiles_n;
  GCD_CHECK_ARG(nblk > 0 && nblk < (1ll << 31), "gcd_gemm_f16 (p8): bad grid %lld", (long long)nblk);
  static_assert((VAR & 2048) != 0, "the product library instantiates persistent kernels only");
  nblk = nblk > 256 ? 256 : (nblk + 7) / 8 * 8;   // one workgroup per CU, 32 per XCD; fewer tiles: a multiple of 8
  hipLaunchKernelGGL(fn, dim3((unsigned)nblk), dim3(512), P8_SMEM_LAUNCH, s, kk);
  GCD_CHECK_LAUNCH();
  return 0;
}

template <int MODE, int BF = 0>
int launch_p8_splitk(const GemmK& k, int splitk, float* ws, hipStream_t s) {
  static GcdPerDeviceOnce attr_once;
  auto fn = gemm_p8_kernel<MODE, 16384 + BF>;
  GCD_CHECK_HIP(attr_once.opt_in((const void*)fn, P8_SMEM_LAUNCH));
  GemmK kk = k;
  p8_extents<MODE>(kk);
  kk.tiles_m = (k.M + P8_BM - 1) / P8_BM;
  kk.tiles_n = (k.N + P8_BN - 1) / P8_BN;
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
  hipLaunchKernelGGL(fn, dim3((unsigned)nblk), dim3(512), P8_SMEM_LAUNCH, s, kk);
  GCD_CHECK_LAUNCH();
  return 0;
}

}  // namespace

// Shapes the 8-phase kernel takes over from gemm_pp.hip (the caller has validated the descriptor already).
bool gcd_gemm_p8_supported(const GemmK& k, int mode) {
  if (k.K % 64 != 0 || k.N % 16 != 0 || k.N < 16) return false;
  if (mode != GCD_GEMM_PLAIN && (k.Cin % 64 != 0 || k.Cin <= 0)) return false;
  if (k.ln_out || k.a_blocked || k.out_blocked) return false;
  if (k.operand_bf16 && (k.out_kind != GCD_OUT_F32 || k.colstats || k.R2)) return false;   // (what the fine-tune step launches)
  // 32-bit buffer offsets: every offset the kernel forms (rows up to 255 past M / 319 past N included) must stay
  // below the out-of-range marker
  const int64_t a_rows = mode == GCD_GEMM_CONV3X3 ? (int64_t)(k.M / ((int64_t)k.Ho * k.Wo)) * k.Hi * k.Wi : (int64_t)k.M;
  if ((a_rows + 256 + (mode == GCD_GEMM_TEMPORAL3 ? k.HW : 0)) * k.lda * 2 + (int64_t)k.K * 2 >= 0xFFFFFF00ll) return false;
  if (((int64_t)k.N + 320) * k.K * 2 >= 0xFFFFFF00ll) return false;
  if (k.out_kind == GCD_OUT_GEGLU && k.N % 32 != 0) return false;
  if (mode == GCD_GEMM_CONV3X3) {   // a piece = 8 consecutive x of one image row; packed (frame, y, x0): 11 + 10 + 11 bits
    const int64_t frames = (int64_t)k.M / ((int64_t)k.Ho * k.Wo);
    if (k.Wo % 8 != 0 || k.Ho > 1024 || k.Wo > 2048 || frames > 2046) return false;
  }
  if (mode == GCD_GEMM_TEMPORAL3 && (k.M + 256 >= (1 << 26) || k.T > 31)) return false;
  return true;
}

// Stride-1 3 x 3 convolutions that take the halo-panel K loop (MODE 3): whole 64-token wave spans inside one image row,
// full tiles, fp32 output (the UNet's ResBlock / up-path convolutions at 72 x 128 and 36 x 64).  Knob 11: never.
static bool p8_halo_ok(const GemmK& k, int mode) {
  if (mode != GCD_GEMM_CONV3X3 || k.stride != 1 || k.up || k.asym || k.out_kind != GCD_OUT_F32) return false;
  if (k.M % P8_BM != 0 || k.Wo % 64 != 0 || k.Wo > 2040) return false;
  if (!(256 % k.Wo == 0 || k.Wo % 256 == 0)) return false;
  if (k.Hi != k.Ho || k.Wi != k.Wo || k.K != 9 * k.Cin || k.operand_bf16) return false;
  return gcd_tune_get(GCD_TUNE_GEMM_IMPL) != 11;
}

int gcd_gemm_p8_launch(const GemmK& k, int mode, bool persist, hipStream_t s) {
  (void)persist;      // (round 5: every launch runs a persistent instantiation, see launch_p8)
#ifdef GCD_ABLATION_BUILD
  if (gcd_tune_get(GCD_TUNE_GEMM_IMPL) == 12 && mode != GCD_GEMM_PLAIN && k.out_kind == GCD_OUT_F32 && !k.colstats && !k.operand_bf16) {
    // A/B: the conv modes without s_setprio and with groups of 8 M-tiles (the PLAIN defaults)
    if (p8_halo_ok(k, mode))
      return k.R2 ? launch_p8<P8_CONV_HALO, 2048 + 96, 4>(k, s) : launch_p8<P8_CONV_HALO, 2048 + 96, 3>(k, s);
    if (mode == GCD_GEMM_CONV3X3)
      return k.R2 ? launch_p8<GCD_GEMM_CONV3X3, 2048 + 96, 4>(k, s) : launch_p8<GCD_GEMM_CONV3X3, 2048 + 96, 3>(k, s);
    return k.R2 ? launch_p8<GCD_GEMM_TEMPORAL3, 2048 + 96, 4>(k, s) : launch_p8<GCD_GEMM_TEMPORAL3, 2048 + 96, 3>(k, s);
  }
  {   // GCD_TUNE_GEMM_IMPL = 64 + ablation bits (PLAIN)
    const int var = gcd_tune_get(GCD_TUNE_GEMM_IMPL) - 64;
    if (var > 0 && var <= 256 + 2 && mode == GCD_GEMM_PLAIN && !k.colstats && !k.operand_bf16) {
      const int epi = k.out_kind == GCD_OUT_GEGLU ? 1 : (k.out_kind == GCD_OUT_F16 && !k.R1 && !k.R2 && !k.frame_alpha) ? 2 : 3;
      if (k.R2) return -1 + 0 * gcd_tune_get(0);   // (no ablation instantiations of the two-residual path)
#define P8_ABL(V)                                                             \
  case V:                                                                     \
    return epi == 1   ? launch_p8<GCD_GEMM_PLAIN, 2048 + V, 1>(k, s)          \
           : epi == 2 ? launch_p8<GCD_GEMM_PLAIN, 2048 + V, 2>(k, s)          \
                      : launch_p8<GCD_GEMM_PLAIN, 2048 + V, 3>(k, s);
      switch (var) {
        P8_ABL(1)
        P8_ABL(2)
        P8_ABL(4)
        P8_ABL(8)
        P8_ABL(16)
        P8_ABL(32)
        P8_ABL(64)
        P8_ABL(48)
        P8_ABL(80)
        P8_ABL(96)
        P8_ABL(112)
        P8_ABL(128)        // GEGLU epilogue without the GELU polynomial
        P8_ABL(256)        // GEGLU epilogue without any arithmetic
        P8_ABL(128 + 2)    // ... and without the K loop: the epilogues alone
        P8_ABL(256 + 2)
        default: break;
      }
#undef P8_ABL
    }
  }
#endif
  if (k.operand_bf16) {   // gcd_gemm_p8_supported: fp32 rows, at most the first residual, no colstats
    switch (mode) {
      case GCD_GEMM_PLAIN:
        return launch_p8<GCD_GEMM_PLAIN, 2048 + 8192, 3>(k, s);
      case GCD_GEMM_CONV3X3:
        return launch_p8<GCD_GEMM_CONV3X3, 2048 + 8192, 3>(k, s);
      default:
        return launch_p8<GCD_GEMM_TEMPORAL3, 2048 + 8192, 3>(k, s);
    }
  }
  if (p8_halo_ok(k, mode)) {
    if (k.colstats) return launch_p8<P8_CONV_HALO, 2048 + 4096>(k, s);
    if (k.R2) return launch_p8<P8_CONV_HALO, 2048, 4>(k, s);
    return launch_p8<P8_CONV_HALO, 2048, 3>(k, s);
  }
  if (k.colstats) {
    switch (mode) {
      case GCD_GEMM_PLAIN:
        return launch_p8<GCD_GEMM_PLAIN, 2048 + 4096>(k, s);
      case GCD_GEMM_CONV3X3:
        return launch_p8<GCD_GEMM_CONV3X3, 2048 + 4096>(k, s);
      default:
        return launch_p8<GCD_GEMM_TEMPORAL3, 2048 + 4096>(k, s);
    }
  }
  switch (mode) {
    case GCD_GEMM_PLAIN:
      // one epilogue fast path per PLAIN instantiation (EPI), chosen from the descriptor
      if (k.out_kind == GCD_OUT_GEGLU) return launch_p8<GCD_GEMM_PLAIN, 2048, 1>(k, s);
      if (k.out_kind == GCD_OUT_F16 && !k.R1 && !k.R2 && !k.frame_alpha) return launch_p8<GCD_GEMM_PLAIN, 2048, 2>(k, s);
      if (k.out_kind == GCD_OUT_F16) return launch_p8<GCD_GEMM_PLAIN, 2048, 5>(k, s);
      if (k.R2) return launch_p8<GCD_GEMM_PLAIN, 2048, 4>(k, s);
      return launch_p8<GCD_GEMM_PLAIN, 2048, 3>(k, s);
    case GCD_GEMM_CONV3X3:
      if (k.out_kind != GCD_OUT_F32)   // fp16 / GEGLU outputs of a convolution: the all-paths instantiation
        return launch_p8<GCD_GEMM_CONV3X3, 2048>(k, s);
      if (k.R2) return launch_p8<GCD_GEMM_CONV3X3, 2048, 4>(k, s);
      return launch_p8<GCD_GEMM_CONV3X3, 2048, 3>(k, s);
    default:
      if (k.out_kind != GCD_OUT_F32) return launch_p8<GCD_GEMM_TEMPORAL3, 2048>(k, s);
      if (k.R2) return launch_p8<GCD_GEMM_TEMPORAL3, 2048, 4>(k, s);
      return launch_p8<GCD_GEMM_TEMPORAL3, 2048, 3>(k, s);
  }
}

int gcd_gemm_p8_launch_splitk(const GemmK& k, int mode, int splitk, float* ws, hipStream_t s) {
  if (k.operand_bf16) {
    switch (mode) {
      case GCD_GEMM_PLAIN:
        return launch_p8_splitk<GCD_GEMM_PLAIN, 8192>(k, splitk, ws, s);
      case GCD_GEMM_CONV3X3:
        return launch_p8_splitk<GCD_GEMM_CONV3X3, 8192>(k, splitk, ws, s);
      default:
        return launch_p8_splitk<GCD_GEMM_TEMPORAL3, 8192>(k, splitk, ws, s);
    }
  }
  switch (mode) {
    case GCD_GEMM_PLAIN:
      return launch_p8_splitk<GCD_GEMM_PLAIN>(k, splitk, ws, s);
    case GCD_GEMM_CONV3X3:
      return launch_p8_splitk<GCD_GEMM_CONV3X3>(k, splitk, ws, s);
    default:
      return launch_p8_splitk<GCD_GEMM_TEMPORAL3>(k, splitk, ws, s);
  }
}
