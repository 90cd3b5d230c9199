/*
 * gcd_amd.h — C ABI of libgcd_amd.so: the MI355X (gfx950 / CDNA4) kernels behind GCD's denoising
 * hot path (SVD VideoUNet forward + EulerEDM sampling step).
 *
 * The reference (basilevh/gcd) has no FFI of its own: its hot path bottoms out in torch.nn ops
 * (SURVEY.md §8b).  Every entry point below therefore cites the reference *operator* it replaces
 * (path:line under /root/reference/gcd-model/), which is what a maintainer would bind with a
 * ctypes stub from the sgm modules.
 *
 * Conventions
 *   - Plain pointers + sizes only.  All pointers are DEVICE pointers unless stated otherwise.
 *   - Activations are token-major ("NHWC"): a tensor of N frames x H x W pixels x C channels is a
 *     row-major matrix [M = N*H*W, C] with an explicit leading dimension (in elements).
 *   - The residual stream is fp32; MFMA operands (normalised activations, q/k/v, FF hidden) are
 *     fp16; every contraction accumulates in fp32; norm statistics are fp32/fp64.
 *   - Every launch goes to the caller's hipStream_t (passed as void*); nothing here synchronises,
 *     allocates device memory or touches another stream, so a sequence of calls can be captured
 *     into a hipGraph by the caller.
 *   - Return value: 0 on success, non-zero on error; gcd_last_error() returns a thread-local
 *     message for the last failing call.
 */
#ifndef GCD_AMD_H
#define GCD_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GCD_AMD_ABI_VERSION 9

/* ---- library ------------------------------------------------------------------------------ */
int gcd_abi_version(void);
const char* gcd_last_error(void);
/* Fills name (<= cap bytes) with the device's gcnArchName; fails if no HIP device is usable. */
int gcd_device_info(int device, char* name, int cap, int* num_cus, size_t* hbm_bytes);

/* Kernel-selection knobs (diagnostics / A-B benchmarking; results are identical up to fp32
 * accumulation order whichever kernel runs).  Initial values come from the environment variables
 * GCD_GEMM_IMPL / GCD_ATTN_IMPL.
 *   GCD_TUNE_GEMM_IMPL: 0 = automatic, 1 = general 128-row kernel, 2 = 256x320 tile kernels (the 8-phase
 *                       16x16x32 K loop where it applies, else the 32-deep ring kernel), 3 = 256x320 tile, ring
 *                       kernel only, 4 = 256x320 tile, never persistent; 10 = like 2, with the overlapped-epilogue
 *                       256x160 kernel (gemm_p8x, an experiment that lost its A/B) on the large fp16-output grids;
 *                       11 = like 0 without the halo-panel K loop of the stride-1 3x3 convolutions; 5 / 6 = general kernel, never / always
 *                       64-row tiles; 7 = like 0 (split-K allowed, used by tests); >= 32: ablation builds
 *   GCD_TUNE_ATTN_IMPL: 0 = automatic, 1..3 select spatial-attention kernel variants, 16 = temporal attention on the
 *                       16-lanes-per-problem VALU kernel instead of the one-problem-per-wave MFMA kernel
 *   GCD_TUNE_PP_MIN_TILES: automatic GEMM choice takes the ping-pong kernel from this many 256x320 tiles
 *                       (0 = the default, 192; environment GCD_PP_MIN_TILES)                   */
#define GCD_TUNE_GEMM_IMPL 0
#define GCD_TUNE_ATTN_IMPL 1
#define GCD_TUNE_PP_MIN_TILES 2
#define GCD_TUNE_STREAM 3       /* non-temporal accesses of the two big streamers, LayerNorm (C = 320 / 640) and GroupNorm
                                   apply (environment GCD_STREAM): 0 = the library's default; else bit 0 non-temporal
                                   loads, bit 1 non-temporal stores, bit 2 (value 4) plain accesses everywhere, bits 4-7 =
                                   only for launches that stream at least that many hundred MB */
#define GCD_TUNE_COUNT 4
int gcd_tune_set(int knob, int value);

/* ---- GEMM family (Linear / Conv2d 3x3 / Conv2d 1x1 / Conv3d (3,1,1) as implicit GEMM) ------ */
/* A-operand addressing modes */
#define GCD_GEMM_PLAIN 0     /* A is [M, K] fp16, row stride lda                          */
#define GCD_GEMM_CONV3X3 1   /* A is NHWC fp16 [frames, Hi, Wi, Cin]; K = 9*Cin (kh,kw,cin) */
#define GCD_GEMM_TEMPORAL3 2 /* A is [(b t) HW, Cin]; K = 3*Cin (kt,cin), zero pad in time  */

/* epilogue output kinds */
#define GCD_OUT_F32 0   /* out fp32 [M, N]                                    */
#define GCD_OUT_F16 1   /* out fp16 [M, N]                                    */
#define GCD_OUT_GEGLU 2 /* out fp16 [M, N/2] = a * gelu(g); weight rows interleaved per 16
                           (rows 32j..32j+15 = value rows, 32j+16..32j+31 = gate rows)        */

typedef struct gcd_gemm_desc {
  /* operands */
  const void* A;     /* fp16 */
  const void* W;     /* fp16 [N, K] row-major (torch Linear layout; conv weights pre-permuted) */
  void* out;         /* fp32 or fp16, see out_kind */
  int64_t lda;       /* elements */
  int64_t ldo;       /* elements */
  int32_t M, N, K;
  int32_t mode;      /* GCD_GEMM_* */
  /* conv geometry (mode != PLAIN) */
  int32_t Cin;       /* channels per tap */
  int32_t Hi, Wi;    /* input spatial size (pre-upsample)        */
  int32_t Ho, Wo;    /* output spatial size                       */
  int32_t stride;    /* 1 or 2 (CONV3X3)                          */
  int32_t upsample;  /* 1: nearest x2 upsample fused before conv  */
  int32_t T;         /* frames per clip (TEMPORAL3)               */
  int32_t HW;        /* pixels per frame (TEMPORAL3)              */
  /* epilogue: out = s_acc[frame] * (acc + bias[n] + rowvec[m / rows_per_vec][n])
                     + s_r1 * R1[m][n] + s_r2[frame] * R2[m][n]                               */
  const float* bias;        /* [N] or NULL                                                    */
  const float* rowvec;      /* [ceil(M/rows_per_vec), ld_rowvec] or NULL                      */
  int64_t ld_rowvec;
  int32_t rows_per_vec;
  const float* R1;          /* fp32 residual or NULL (may alias out)                          */
  int64_t ldr1;
  const float* R2;          /* second fp32 residual (AlphaBlender partner) or NULL            */
  int64_t ldr2;
  float s_acc, s_r1, s_r2;  /* scalar scales used when frame_alpha == NULL                    */
  const float* frame_alpha; /* optional [ceil(M/rows_per_alpha)]: if set, s_acc := (1-alpha),
                               s_r2 := alpha (per frame), s_r1 := s_r1*(1-alpha) if r1_blend  */
  int32_t rows_per_alpha;
  int32_t r1_blend;
  int32_t out_kind;         /* GCD_OUT_* */
  const void* zero_page;    /* >= 256 B of zeros (device), required for conv modes            */
  /* optional fused LayerNorm of the OUTPUT rows (fp32 out, N == 320 only: one 256x320 tile holds a
     whole row; see gcd_gemm_ln_fusable):  z = out_row [+ ln_addvec[m / ln_rows_per_vec]],
     ln_sum_out (optional fp32 [M, N]) = z,  ln_out16 (fp16 [M, N]) = LN(z) * ln_gamma + ln_beta.
     Replaces the nn.LayerNorm that reads the residual stream right after this GEMM wrote it
     (attention.py:519-521, video_attention.py:50,90-93, incl. the x + time_pos_embed form of
     video_attention.py:283-284).                                                                  */
  void* ln_out16;
  int64_t ld_ln_out;
  const float* ln_gamma;
  const float* ln_beta;
  float ln_eps;
  int32_t ln_rows_per_vec;
  const float* ln_addvec;
  int64_t ld_ln_addvec;
  float* ln_sum_out;
  int64_t ld_ln_sum;
  /* optional scratch (device, 16-byte aligned): lets gcd_gemm_f16 split K over several workgroups
     when the output has too few 256x320 tiles to fill the chip (the 9x16 bottleneck level: 64 tiles
     on 256 CUs) — fp32 partial sums [splits][M][N] + one reduce-and-epilogue launch.  Needs
     4 * M * N * 4 bytes to be used; NULL or too small = never split.                               */
  void* workspace;
  int64_t workspace_bytes;
  /* CONV3X3, stride 2 only: 1 = the asymmetric (0,1,0,1) zero padding of the autoencoder's Downsample
     (F.pad + Conv2d(stride 2, padding 0), diffusionmodules/model.py:76-91): input pixel
     (2y + ky, 2x + kx), ky, kx in 0..2, Ho = Hi / 2; 0 = the symmetric padding 1 of every other conv. */
  int32_t asym_pad;
  /* optional GroupNorm pre-reduction of the fp32 OUTPUT (ABI v4): per block of 64 consecutive output
     rows b = m / 64 and per column n,
         colstats[(2 b + 0) * N + n] = sum  over the block's rows of out[m][n]
         colstats[(2 b + 1) * N + n] = sum  over the block's rows of out[m][n]^2
     of the values exactly as stored (after bias / rowvec / residual / blend), accumulated in fp32 by
     the tile epilogue that holds them in registers anyway — the GroupNorm that reads this tensor next
     (gcd_groupnorm_stats_from_colsums) then needs no pass over it.  Honoured only where every tile
     takes the row-major fast epilogue (gcd_gemm_colstats_supported); otherwise gcd_gemm_f16 fails.
     [2 * M / 64, N] floats.                                                                        */
  float* colstats;
  /* Tile-blocked GEGLU hidden tensor (ABI v4).  A 256 x 320 GEGLU tile produces 256 rows x 160 hidden
     columns = 320-byte row segments that are not 128-byte aligned in a row-major [M, N/2] matrix, and
     their neighbours in the same cache lines are written by other workgroups: measured 25 % slower than
     aligned segments (tools/store_probe.cpp).  With out_blocked (GCD_OUT_GEGLU only) tile (tm, tn) is
     stored as one contiguous 80 KB block:  out[((tm * N/320 + tn) * 256 + r) * 160 + c];  a_blocked
     (GCD_GEMM_PLAIN only) reads an A operand laid out that way ([M/256][K/160][256][160]), i.e. the
     FeedForward's second Linear consumes the hidden columns in the order the first one wrote them.
     Needs M % 256 == 0, N % 320 == 0 resp. K % 160 == 0 and the ping-pong kernel
     (gcd_gemm_hidden_blocked_supported).                                                           */
  int32_t out_blocked;
  int32_t a_blocked;
  /* 1: A and W hold bfloat16 instead of fp16 (GCD_GEMM_PLAIN, fp32 output, K % 64 == 0; runs on the
     general 128-row kernel with v_mfma_f32_16x16x32_bf16).  Same MFMA rate as fp16 on gfx950: the point
     is exponent range — the fine-tune step's gradient GEMMs without loss scaling — at 8 instead of 11
     significant bits (7x over the 1e-3 inference tolerance, so inference stays fp16). */
  int32_t operand_bf16;
  /* Scheduling hints (ABI v6; results are bit-identical whatever they say).  bit 0: the tiles are walked from the
     END of the output — every XCD takes its contiguous share of the tile order back to front — instead of from the
     start.  A launch that reads what the previous launch wrote finds the END of that tensor in the 256 MB Infinity
     Cache (the start has been evicted by the rest): walking in the opposite direction of the producer turns part of
     the consumer's HBM reads into cache hits.  Honoured by the 8-phase 256 x 320 kernel without split-K, ignored
     elsewhere.  bit 1: the automatic kernel choice takes the 256 x 320 tile kernels from 128 tiles instead of 192
     (GCD_TUNE_PP_MIN_TILES unset): the fine-tune step's GEMMs at 32 x 48 latents, where the 84-190-tile shapes of
     the middle levels run 3 % of the step faster there (profiles/r04n_train_ab.txt); the sampler's shapes at
     72 x 128 were tuned at 192.                                                                                   */
  int32_t sched;
} gcd_gemm_desc;

/* Replaces torch.nn.Linear / Conv2d / Conv3d forward on the hot path:
 *   attention.py:87-113 (GEGLU/FeedForward), attention.py:272-278,300-303,344 (q/k/v/out
 *   projections), attention.py:663,693-699 (proj_in/proj_out), openaimodel.py:270-274,293-307,
 *   311-318 (ResBlock convs, dims=2 and dims=3), openaimodel.py:139-142,199-206 (Up/Downsample),
 *   diffusionmodules/util.py:358-369 (AlphaBlender, folded into the epilogue).                 */
int gcd_gemm_f16(const gcd_gemm_desc* desc, void* stream);
/* 1 if gcd_gemm_f16 would accept the descriptor's fused-LayerNorm request for this shape (the
 * automatic kernel choice lands on the 256x320 ping-pong kernel and N == 320), else 0.          */
int gcd_gemm_ln_fusable(int M, int N, int K, int mode);
/* 1 if gcd_gemm_f16 would honour desc->colstats for this descriptor (fp32 out, no second residual,
 * M % 256 == 0, N % 320 == 0, rowvec / frame_alpha constant over each 256-row tile, the automatic
 * kernel choice lands on the 256x320 ping-pong kernel without split-K), else 0.                  */
int gcd_gemm_colstats_supported(const gcd_gemm_desc* desc);
/* 1 if a FeedForward of M tokens whose GEGLU projection has N_geglu output columns (hidden = N_geglu / 2)
 * and whose second Linear has N_out columns can keep its hidden tensor tile-blocked (out_blocked on the
 * first GEMM, a_blocked on the second): both launches must land on the ping-pong kernel.          */
int gcd_gemm_hidden_blocked_supported(int M, int N_geglu, int N_out);

/* ---- FeedForward(GEGLU) as ONE kernel (ABI v8; the C = 320 / hidden = 1280 level) ------------------------------
 * out[M, 320] = s_acc * ( W2 . (value * gelu(gate)) + b2 + R1 ) + s_r2 * R2,   value | gate = W1 . x + b1,
 * the 1280-wide hidden tensor rounded to fp16 once (the rounding point of the two-GEMM path) and never written to
 * memory: per 128-token tile the workgroup streams W1 / W2 through LDS in 32-hidden-unit chunks, keeps the tile's x
 * fragments in registers and the output accumulators — initialised with R1 — in the accumulator file
 * (gcd_amd/csrc/ff_fused_kernel.h).  Replaces, for that level, FeedForward.forward (attention.py:87-121) together with
 * the residual adds around it (attention.py:566-572, video_attention.py:109-140) and the AlphaBlender
 * (util.py:364-368; frame_alpha: s_acc := 1 - alpha, s_r2 := alpha per `rows_per_alpha` rows, with R1 inside the
 * blended term — gcd_gemm_desc's r1_blend form).  Same arithmetic as gcd_gemm_f16(GCD_OUT_GEGLU) + gcd_gemm_f16 up to
 * fp32 summation order (R1 enters the accumulation first).                                                          */
typedef struct gcd_ff_desc {
  const void* X;            /* fp16 [M, 320] LayerNorm output, leading dimension ldx (elements); NULL in the LN form */
  int64_t ldx;
  const void* wp;           /* gcd_ff_pack_f16 output (gcd_ff_packed_bytes() bytes)                        */
  const float* b1;          /* [2560] in the GEGLU row order of GCD_OUT_GEGLU (16 value / 16 gate)         */
  const float* b2;          /* [320]                                                                        */
  const float* R1;          /* fp32 [M, 320] residual, REQUIRED; out may alias it                           */
  int64_t ldr1;
  const float* R2;          /* fp32 [M, 320] or NULL; out may alias it                                      */
  int64_t ldr2;
  void* out;                /* fp32 [M, 320]; fp16 allowed (GCD_OUT_F16) when R2 is given                   */
  int64_t ldo;
  int32_t out_kind;         /* GCD_OUT_F32 / GCD_OUT_F16                                                    */
  const float* frame_alpha; /* optional [ceil(M / rows_per_alpha)]                                          */
  int32_t rows_per_alpha;   /* multiple of 32                                                               */
  float s_acc, s_r2;        /* used when frame_alpha == NULL                                                */
  int32_t M, C, hidden;     /* C = 320, hidden = 1280                                                       */
  int32_t sched;            /* bit 0: walk the tiles from the end (as gcd_gemm_desc.sched)                  */
  /* The LayerNorm form (ln_gamma != NULL; wp packed with for_ln = 1): the nn.LayerNorm in front of the FeedForward
     (attention.py:519-521, 566-572; video_attention.py:50, 90-93, 109-140) is computed by the same kernel.
        z = x32[m] + addvec[m / rows_per_vec]      (addvec: the x + time_pos_embed of video_attention.py:283-284, or NULL)
        out = s_acc * ( FF( LN(z) * ln_gamma + ln_beta ) + z ) + s_r2 * R2
     z is read once (fp32, 16-byte aligned rows) and is the residual: X and R1 must be NULL.  out may alias x32 or R2.  */
  const float* x32;
  int64_t ldx32;
  const float* ln_gamma;
  const float* ln_beta;
  float ln_eps;
  const float* addvec;
  int64_t ld_addvec;
  int32_t rows_per_vec;     /* multiple of 32                                                               */
} gcd_ff_desc;
int64_t gcd_ff_packed_bytes(void);
/* w1: fp16 [2560, 320] in GCD_OUT_GEGLU row order, w2: fp16 [320, 1280] -> wp, the weights in MFMA-fragment order
 * (per 32-hidden-unit chunk 40 W1 fragments + 20 W2 fragments of 1 KB).  Once per parameter version.  for_ln = 1:
 * for the LayerNorm form (W1's K order follows the channel strips of the accumulator layout x32 is read in).        */
int gcd_ff_pack_f16(const void* w1, const void* w2, void* wp, int for_ln, void* stream);
int gcd_ff_fused_supported(int M, int C, int hidden);
int gcd_ff_fused_f16(const gcd_ff_desc* desc, void* stream);

/* ---- LayerNorm + q | k | v projection as one launch where the model width is 320 (ABI v9) ------------------------------
 * out16[M, N] (fp16) = LN(x32[M, 320]; gamma, beta, eps) @ W[N, 320]^T: `self.attn1(self.norm1(x), ...)`'s to_q | to_k | to_v
 * of attention.py:519-521 / 300-316 and video_attention.py:90-93 (no bias; the softmax scale is folded into the q rows of
 * W by the caller, as for gcd_gemm_f16).  Replaces gcd_layernorm_f16 + gcd_gemm_f16(GCD_OUT_F16) with the same rounding
 * points (LayerNorm output to fp16, fp32 accumulation, one fp16 rounding of the result); the fp32 rows are read once and
 * the normalised operand never reaches memory.  wp = gcd_lnqkv_pack_f16(W) (gcd_lnqkv_packed_bytes(N) bytes), once per
 * parameter version: W in MFMA-fragment order per chunk of 64 output features.  N % 64 == 0.  sched bit 0: walk the
 * 256-token tiles from the end (pure scheduling, as gcd_gemm_desc.sched).                                                  */
int64_t gcd_lnqkv_packed_bytes(int N);
int gcd_lnqkv_supported(int C, int N);
int gcd_lnqkv_pack_f16(const void* W, int N, void* wp, void* stream);
int gcd_lnqkv_f16(const float* x32, int64_t ldx32, const float* gamma, const float* beta, float eps, const void* wp,
                  void* out16, int64_t ldo, int M, int C, int N, int sched, void* stream);

/* y[M,N] (fp32) = [y +] act_out( act_in(x[M,K]) @ W[N,K]^T + b ), fp32 weights, any M >= 1 (rows are processed 32 at a time).
 * act flags: bit0 = SiLU on input, bit1 = SiLU on output, bit2 = accumulate into y.
 * Replaces the tiny per-frame MLPs: video_model.py:155-200,485-497 (time/label/aux embeds),
 * openaimodel.py:287-293 (emb_layers), video_attention.py:216-222 (time_pos_embed),
 * and the 1-key cross-attention collapse to_out(to_v(ctx)) (attention.py:300-344).            */
int gcd_linear_smallm_f32(const float* x, int64_t ldx, const float* W, const float* b, float* y,
                          int64_t ldy, int M, int N, int K, int act_flags, void* stream);

/* ---- normalisation -------------------------------------------------------------------------- */
/* GroupNorm(32 groups) statistics over `rows_per_inst` consecutive rows (HW for a frame,
 * T*HW for the time_stack variant) of x = [x1 (C1 ch) | x2 (C2 ch)] fp32.
 * partial: workspace of ninst*nchunks*32*2 doubles; stats: ninst*32*2 floats (mean, rstd).
 * Replaces diffusionmodules/util.py:259-276 (GroupNorm32) and attention.py:125-128 (Normalize). */
int gcd_groupnorm_stats(const float* x1, int64_t ld1, int C1, const float* x2, int64_t ld2, int C2,
                        int64_t M, int64_t rows_per_inst, float eps, double* partial, int nchunks,
                        float* stats, void* stream);
/* The same statistics from the per-64-row column sums a producing gcd_gemm_f16 left behind
 * (desc->colstats) instead of a pass over the tensor: x = [x1 | x2] virtual concat with column sums
 * cs1 [2*M/64, C1] and cs2 [2*M/64, C2] (cs2 NULL when C2 == 0); rows_per_inst % 64 == 0.
 * fp64 accumulation over the blocks.  stats as gcd_groupnorm_stats.                               */
int gcd_groupnorm_stats_from_colsums(const float* cs1, int C1, const float* cs2, int C2, int64_t M,
                                     int64_t rows_per_inst, float eps, float* stats, void* stream);
/* y16 = [silu]((x - mean) * rstd * gamma + beta) as fp16 [M, C1+C2]; raw16 (optional) = fp16(x).
 * silu: bit 0 = apply SiLU; bits 1-2 = walk order of the row blocks (pure scheduling, cf. gcd_gemm_desc.sched: what
 * the previous launch wrote LAST is what the 256 MB Infinity Cache still holds): 0 front to back, 1 back to front,
 * 2 / 3 the tensor as eight contiguous regions walked concurrently — the order in which a persistent GEMM's eight XCD
 * shares are written — every region back to front (2) / front to back (3).  Bit 3 (ABI v7): y16 / raw16 are BFLOAT16,
 * rounded once from fp32 (the bf16 GEMM operands of the fine-tune step, cfg4).                      */
int gcd_groupnorm_apply(const float* x1, int64_t ld1, int C1, const float* x2, int64_t ld2, int C2,
                        int64_t M, int64_t rows_per_inst, const float* stats, const float* gamma,
                        const float* beta, int silu, void* y16, int64_t ldy, void* raw16,
                        int64_t ldraw, void* stream);
/* LayerNorm over C of (x + addvec[m / rows_per_vec]) -> fp16; optionally writes the fp32 sum back
 * (x_mix = x + time_pos_emb, video_attention.py:283-284).  Replaces nn.LayerNorm at
 * attention.py:519-521 and video_attention.py:50,90-93.  order: walk order of the row blocks, 0..3 as
 * gcd_groupnorm_apply's (ABI v6); + 4 (ABI v7): y16 is bfloat16, rounded once from fp32.           */
int gcd_layernorm_f16(const float* x, int64_t ldx, int64_t M, int C, const float* gamma,
                      const float* beta, float eps, const float* addvec, int64_t ld_addvec,
                      int rows_per_vec, float* sum_out, int64_t ld_sum, void* y16, int64_t ldy,
                      int order, void* stream);

/* ---- attention ------------------------------------------------------------------------------ */
/* vt[((f*heads+h)*64 + d) * S_pad + p(s)] = qkv[(f*S+s)*ld + 2C + h*64 + d], zero padded to S_pad;
 * p swaps the two middle quads of every group of 16 keys (0-3, 8-11, 4-7, 12-15), the order in
 * which gcd_attn_spatial_f16's P V MFMA consumes them.  vt is private to that kernel.            */
int gcd_attn_transpose_v(const void* qkv, int64_t ld, int frames, int S, int heads, void* vt,
                         int S_pad, void* stream);
/* Spatial self-attention, head dim 64, softmax scale 1/8 (F.scaled_dot_product_attention at
 * attention.py:331-335).  qkv: fp16 [frames*S, 3C] (q | k | v, head-major columns), vt from
 * gcd_attn_transpose_v, out: fp16 [frames*S, C].
 * q_prescaled != 0: the q columns already carry the factor log2(e)/8 (folded into W_q in fp32 when
 * the projection weights are packed — one fp16 rounding, same as without it), so scores are exp2
 * arguments and the kernel spends no instruction on scaling.  0: plain q, the kernel scales.      */
int gcd_attn_spatial_f16(const void* qkv, int64_t ld, const void* vt, int S_pad, void* out,
                         int64_t ldo, int frames, int S, int heads, int q_prescaled, void* stream);
/* Temporal self-attention over T <= 16 frames per (clip, pixel, head)
 * (video_attention.py:114,126-129 -> attention.py:331-335).  Rows are ((b*T + t)*HW + s).       */
int gcd_attn_temporal_f16(const void* qkv, int64_t ld, void* out, int64_t ldo, int clips, int T,
                          int HW, int heads, void* stream);

/* ---- first-stage decoder (VideoDecoder) helpers ------------------------------------------------ */
/* y16[r, :] = softmax(x[r, :]) for an fp32 score matrix [R, C] (C % 4 == 0, C <= 16384): the softmax
 * of the VAE decoder's single-head mid attention (diffusionmodules/model.py:180-198, head dim = the
 * 512 channels, so Q K^T and P V run on gcd_gemm_f16 with the 1/sqrt(C) scale in its epilogue).   */
int gcd_softmax_rows_f16(const float* x, int64_t ldx, void* y16, int64_t ldy, int64_t R, int C,
                         void* stream);
/* fp16 [R, C] -> [C, R] (V -> V^T, the [N, K] operand of the P V GEMM).                           */
int gcd_transpose_f16(const void* x, int64_t ldx, void* y, int64_t ldy, int R, int C, void* stream);
/* AE3DConv.time_mix_conv (temporal_ae.py:84-107: Conv3d C -> C, kernel (3,1,1), zero padding in
 * time, applied to the output of the last 3x3 conv) fused with the token-major -> NCHW change:
 *   out[n][co][p] = b[co] + sum_{dt,ci} w[co][ci][dt] * in[((n + dt - 1)*HW + p)*ld + ci]
 * over the frames of n's clip of T frames.  in: fp32 [N*HW, ld]; w: fp32 [C, C, 3] (the Conv3d weight
 * with its two unit axes dropped); out: fp32 [N, C, HW]; C <= 4.                                   */
int gcd_time_mix_unpack(const float* in, int64_t ld, const float* w, const float* b, float* out,
                        int C, int N, int T, int HW, void* stream);

/* ---- UNet ends, casts ----------------------------------------------------------------------- */
/* NCHW fp32 -> token-major fp16, fusing the sampler-side glue in front of the first conv:
 *   out16[(n*HW + p)*Cpad + c] = x[n % nx][c][p] * c_in[n]      c < Cx       (denoiser.py:46-48,
 *                              = concat[n][c - Cx][p]            c < Cx + Cc   guiders.py:89-100,
 *                              = 0                               otherwise     wrappers.py:26)
 * x: [nx, Cx, HW]; concat: [N, Cc, HW] or NULL; c_in: [N] or NULL (= 1); Cpad % 8 == 0.  The first
 * conv (video_model.py:205-211, 8 -> 320 channels) then runs on the implicit-GEMM kernel with its
 * input channels zero-padded to the 64-channel K granule.                                        */
int gcd_pack_input(const float* x, int nx, int Cx, const float* concat, int Cc, const float* c_in,
                   int N, int HW, void* out16, int Cpad, void* stream);
/* token-major fp32 [N*HW, ld] (first Cout channels) -> NCHW fp32 [N, Cout, HW]: the layout change
 * behind the final conv (video_model.py:455-459,537), whose 4 output channels are computed by the
 * implicit-GEMM kernel with N padded to 16.                                                      */
int gcd_unpack_output(const float* in, int64_t ld, float* out, int Cout, int N, int HW,
                      void* stream);
int gcd_cast_f32_f16(const float* x, int64_t ldx, void* y16, int64_t ldy, int64_t M, int C,
                     void* stream);

/* ---- sampler -------------------------------------------------------------------------------- */
/* One fused EulerEDM update with LinearPredictionGuider CFG and the Denoiser affine:
 *   D_u = net[n]*c_out + x*c_skip, D_c = net[n + nx]*c_out + x*c_skip      (denoiser.py:46-49)
 *   D   = D_u + scale[n % T] * (D_c - D_u)                                  (guiders.py:79-87)
 *   d   = (x - D) / sigma ; x_out = x + (sigma_next - sigma) * d            (sampling_utils.py:34-35,
 *                                                                            sampling.py:86-87,114-117)
 * x, x_out: [nx, chw] fp32 (may alias); net: [2*nx, chw] fp32; scale: [T] fp32 (device);
 * sig: device pointer to {sigma, sigma_next} so that the same captured graph serves every step. */
int gcd_cfg_euler_step(const float* x, const float* net, const float* scale, const float* sig,
                       float* x_out, int nx, int T, int64_t chw, void* stream);
/* c_in[n] = 1/sqrt(s^2+1), c_noise[n] = 0.25*ln(s) for every frame, from the device sigma
 * (denoiser_scaling.py:53-61).                                                                 */
int gcd_edm_scalings(const float* sig, float* c_in, float* c_noise, int N, void* stream);
/* emb[n, :] = [cos(t_n f_k), sin(t_n f_k)], f_k = exp(-ln(max_period) k / half)
 * (diffusionmodules/util.py:207-231).                                                          */
int gcd_timestep_embedding(const float* t, float* emb, int N, int dim, float max_period,
                           void* stream);

/* ---- fine-tune step: backward pass (BASELINE.json cfg4) ----------------------------------------- */
/* The reference trains the same VideoUNet through torch.autograd (loss.py:115-273 forward,
 * Lightning backward, diffusion.py:412-431 optimizer).  The contractions of the backward pass run on
 * gcd_gemm_f16 with transposed operands; these are the kernels around them (gcd_amd/csrc/backward.hip).
 * fp32 gradients, row-major with explicit leading dimensions, like the forward.                       */
/* 3x3 taps of x16 [frames*Hi*Wi, Cin] laid out as the implicit GEMM's A operand: col16 [frames*Ho*Wo,
 * 9*Cin], K order (kh, kw, cin) — the wgrad operand of Conv2d 3x3 (openaimodel.py:270-274,299-307),
 * stride 1 / 2, fused x2 upsample, asymmetric padding as gcd_gemm_f16's CONV3X3 mode.                */
int gcd_im2col3x3_f16(const void* x16, int64_t ldx, void* col16, int frames, int Cin, int Hi, int Wi,
                      int Ho, int Wo, int stride, int upsample, int asym_pad, void* stream);
/* dx [frames*Hi*Wi, Cin] = transpose of the above applied to dcol fp32 [frames*Ho*Wo, 9*Cin] (the dgrad
 * of the convolution once dcol = dY @ W has been formed by a plain GEMM); a gather, deterministic.   */
int gcd_col2im3x3_f32(const float* dcol, float* dx, int64_t lddx, int frames, int Cin, int Hi, int Wi,
                      int Ho, int Wo, int stride, int upsample, int asym_pad, void* stream);
/* The same pair for the (3,1,1) temporal convolution of time_stack (video_model.py:42-60): rows are
 * (clip, t, hw), col [M, 3*C] in K order (kt, cin).                                                  */
int gcd_im2col_t3_f16(const void* x16, int64_t ldx, void* col16, int64_t M, int C, int T, int HW,
                      void* stream);
int gcd_col2im_t3_f32(const float* dcol, float* dx, int64_t lddx, int64_t M, int C, int T, int HW,
                      void* stream);
/* out[b][n] += sum of x over rows b*rows_per_block .. +rows_per_block-1 (out zeroed by the caller):
 * bias gradients (one block) and the gradients of per-frame epilogue vectors (emb_layers output).    */
int gcd_rowblock_sum_f32(const float* x, int64_t ldx, int64_t M, int N, int64_t rows_per_block,
                         float* out_zeroed, void* stream);
/* GroupNorm(32) [+SiLU] backward: x, dy, dx fp32 [M, C]; stats from gcd_groupnorm_stats; AB: ninst*C*2
 * doubles that RECEIVE per (instance, channel) sum(dz), sum(dz*xhat) — dbeta / dgamma are their sums over
 * instances (no pre-zeroing: ABI v6 replaced the atomics of the reduction pass by per-chunk partial sums in
 * `scratch`, >= gcd_groupnorm_bwd_scratch_floats(C, M, rows_per_inst) floats, 16-byte aligned, folded in
 * fp64 by a second launch).  Replaces autograd of util.py:259-276 + SiLU.                            */
int64_t gcd_groupnorm_bwd_scratch_floats(int C, int64_t M, int64_t rows_per_inst);
int gcd_groupnorm_bwd(const float* x, int64_t ldx, const float* dy, int64_t lddy, int C, int64_t M,
                      int64_t rows_per_inst, const float* stats, const float* gamma, const float* beta,
                      int silu, double* AB, float* scratch, int64_t scratch_floats, float* dx, int64_t lddx,
                      const float* dx_add, int64_t ld_add, void* stream);
/* LayerNorm backward (rows of C <= 1280): dx, and dgamma / dbeta accumulated into zeroed [C] buffers.
 * dx_add (ABI v7, optional, both norm backwards): a second gradient of the same tensor (the residual branch's) added to dx
 * on the way out — saves an elementwise add pass per norm in the planned fine-tune engine. */
int gcd_layernorm_bwd(const float* x, int64_t ldx, const float* dy, int64_t lddy, int64_t M, int C,
                      const float* gamma, float eps, float* dx, int64_t lddx, float* dgamma_zeroed,
                      float* dbeta_zeroed, const float* dx_add, int64_t ld_add, void* stream);
/* GEGLU on the fp32 projection h = [value | gate] [M, 2H] (attention.py:87-97): out = value*gelu(gate)
 * (exact erf) and its backward.                                                                      */
int gcd_geglu_fwd_f32(const float* h, int64_t ldh, float* out, int64_t ldo, int64_t M, int H, void* stream);
/* the forward with the result rounded to fp16 (the operand of FeedForward's second Linear, written once)  */
int gcd_geglu_fwd_f16(const float* h, int64_t ldh, void* out16, int64_t ldo, int64_t M, int H, void* stream);
/* ABI v7: the same rounded ONCE to bfloat16 (the fine-tune step with bf16 operands, cfg4: no fp16 hop, none of fp16's
 * range on the hidden tensor). */
int gcd_geglu_fwd_bf16(const float* h, int64_t ldh, void* out16, int64_t ldo, int64_t M, int H, void* stream);
int gcd_geglu_bwd_f32(const float* h, int64_t ldh, const float* dout, int64_t lddo, float* dh, int64_t lddh,
                      int64_t M, int H, void* stream);
/* Softmax backward over R rows of S scores: dS16 = P16 * (dP - rowsum(P16*dP)) * scale.              */
int gcd_softmax_bwd_rows(const void* P16, int64_t ldp, const float* dP, int64_t lddp, void* dS16,
                         int64_t ldds, int64_t R, int S, float scale, void* stream);
/* Backward of gcd_attn_spatial_f16 (flash style: no S x S tensor in memory; reference: torch.autograd through
 * F.scaled_dot_product_attention, attention.py:331-335).  qkv16 / out16: what the forward consumed and produced
 * (fp16 [frames*S, 3C] / [frames*S, C]), dout16: the incoming gradient rounded to fp16 [frames*S, C], dqkv32:
 * fp32 [frames*S, 3C] (every element written), scale: the softmax scale (1/8).  ws: scratch of
 * gcd_attn_spatial_bwd_ws_bytes(frames, S, heads) bytes (transposed operands, log-sum-exp and delta rows). */
int64_t gcd_attn_spatial_bwd_ws_bytes(int frames, int S, int heads);
int gcd_attn_spatial_bwd(const void* qkv16, int64_t ld, const void* out16, int64_t ldo, const void* dout16,
                         int64_t lddo, void* dqkv32, int64_t ldg, void* ws, int64_t ws_bytes, int frames, int S,
                         int heads, float scale, void* stream);
/* Backward of gcd_attn_temporal_f16 (T <= 16 tokens, d = 64): dqkv fp32 [M, 3C] from dO fp32 [M, C].  */
int gcd_attn_temporal_bwd(const void* qkv16, int64_t ld, const float* dO, int64_t lddo, float* dqkv,
                          int64_t lddq, int clips, int T, int HW, int heads, void* stream);
/* y16 = fp16(x * scale): gradients enter the fp16 GEMMs pre-scaled (loss scaling), the GEMM's s_acc
 * removes the factor in fp32.                                                                        */
/* y = bfloat16(x) (round to nearest even), for operand_bf16 GEMMs; x fp32 or (the _f16 form) fp16.      */
int gcd_cast_f32_bf16(const float* x, int64_t ldx, void* y16, int64_t ldy, int64_t M, int C, void* stream);
int gcd_cast_f16_bf16(const void* x16, int64_t ldx, void* y16, int64_t ldy, int64_t M, int C, void* stream);
/* y16 = fp16 (or bfloat16 when to_bf16) of x AND sums[b][c] += sum of x over rows b*rows_per_block .. (sums zeroed by the
 * caller, [M / rows_per_block, C]): the incoming gradient of a Linear / convolution rounded for its GEMMs and summed
 * for its bias / per-frame-vector gradient in ONE pass over it.  total_zeroed (ABI v7, optional): [C], receives the sums
 * over ALL rows as well — the bias gradient of a node that also has a per-frame vector, with no second reduction.       */
int gcd_cast_colsum_f32(const float* x, int64_t ldx, void* y16, int64_t ldy, int64_t M, int C, int64_t rows_per_block,
                        float* sums_zeroed, int to_bf16, float* total_zeroed, void* stream);
int gcd_cast_scale_f32_f16(const float* x, int64_t ldx, void* y16, int64_t ldy, int64_t M, int C,
                           float scale, void* stream);
/* torch.optim.Adam step (no amsgrad; weight_decay added to the gradient), in place; grad_scale is
 * multiplied into g first (1 / loss_scale, 1 / world_size ...).  diffusion.py:412-431.                */
int gcd_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                  float beta2, float eps, float weight_decay, int step, float grad_scale, void* stream);
/* The same step for `count` tensors (HOST arrays of device pointers / element counts) in as few launches as
 * 48 tensors / 64 K chunks of 16 K elements each allow: the ~1300 parameter tensors of the UNet in ~30 launches. */
int gcd_adam_step_multi(int count, float* const* p, const float* const* g, float* const* m, float* const* v,
                        const int64_t* n, float lr, float beta1, float beta2, float eps, float weight_decay,
                        int step, float grad_scale, void* stream);

/* ---- stream / graph plumbing ---------------------------------------------------------------- */
int gcd_graph_begin_capture(void* stream);
int gcd_graph_end_capture(void* stream, void** graph_exec_out);
int gcd_graph_launch(void* graph_exec, void* stream);
int gcd_graph_destroy(void* graph_exec);
/* HIP events on an explicit stream (bench.py times kernels on the launch stream). */
int gcd_event_create(void** ev);
int gcd_event_record(void* ev, void* stream);
int gcd_event_sync(void* ev);
int gcd_event_elapsed_ms(void* start, void* stop, float* ms);
int gcd_event_destroy(void* ev);
int gcd_stream_sync(void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GCD_AMD_H */
