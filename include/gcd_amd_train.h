/* gcd_amd_train.h — C ABI of gcd_amd/libgcd_amd_train.so: kernels that only the fine-tune step (BASELINE.json cfg4) uses.
 * Same rules as gcd_amd.h: raw device pointers + explicit leading dimensions (in elements), the caller's hipStream_t, no
 * internal allocation or synchronisation, `int` status (0 = ok) + a thread-local message (gcd_train_last_error).
 * A library of its own so that the sampler's library, whose source digest stamps the PMC traffic profile `bench.py` quotes,
 * does not change when a training kernel does. */
#ifndef GCD_AMD_TRAIN_H
#define GCD_AMD_TRAIN_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define GCD_AMD_TRAIN_ABI_VERSION 2

int gcd_train_abi_version(void);
const char* gcd_train_last_error(void);

/* Weight gradient of a Linear / 1x1 convolution / im2col'd convolution:  dW[N, K] (fp32, row stride lddw) = dY^T X  with
 * dY [M, N] and X [M, K] 16-bit (fp16, or bfloat16 when bf16 != 0), ROW-MAJOR as the forward pass left them; the
 * contraction runs over the M tokens.  Both operands are transposed on the LDS read (ds_read_b64_tr_b16): no transposed
 * copies in HBM.  N and K multiples of 8, lddy / ldx multiples of 8, 16-byte aligned pointers; `scratch` receives the fp32
 * partial outputs of the token slices (>= gcd_wgrad_tr_scratch_floats(M, N, K) floats) that a second launch folds.
 * Replaces torch.autograd's weight gradient of nn.Linear / nn.Conv2d on the fine-tune path (attention.py:87-113,272-303;
 * openaimodel.py:270-318; loss.py:115-273 drives the backward pass). */
int64_t gcd_wgrad_tr_scratch_floats(int64_t M, int N, int K);
int gcd_wgrad_tr_f16(const void* dy16, int64_t lddy, const void* x16, int64_t ldx, int64_t M, int N, int K, int bf16,
                     float* dW, int64_t lddw, float* scratch, int64_t scratch_floats, void* stream);

/* The same contraction written in the PARAMETER's own layout, cropped (round 5: the planned fine-tune engine writes every
 * weight gradient straight into the parameter's .grad — no padded temporary, no permuted copy).  K = taps * Kc: column
 * k = tap * Kc + c of the contraction is element [n][c][tap] of a parameter of shape [N_real][C_real][taps] (Conv2d 3x3 in
 * the (kh, kw, cin) K order of the im2col'd operand: taps = 9; Conv3d (3,1,1): taps = 3; Linear: taps = 1, where lddw is
 * honoured as the row stride).  Rows n >= N_real and channels c >= C_real (the zero padding of the first / last
 * convolution) are dropped.  accumulate != 0 adds onto dW instead of overwriting (gradient accumulation). */
int gcd_wgrad_tr_f16_ex(const void* dy16, int64_t lddy, const void* x16, int64_t ldx, int64_t M, int N, int K, int bf16,
                        float* dW, int64_t lddw, int taps, int N_real, int C_real, int accumulate, float* scratch,
                        int64_t scratch_floats, void* stream);

/* The weight gradient of a convolution WITHOUT an im2col'd operand (round 5): conv = 1: Conv2d 3x3, stride 1, same size
 * (openaimodel.py:270-318; rows of x16 / dy16 are (frame, y, x), M = frames * Ho * Wo); conv = 2: Conv3d (3,1,1) over the T
 * frames of a clip (video_model.py:42-60; rows (clip, t, hw), M = clips * T * HW).  x16 [M, Cp] is the convolution's INPUT
 * activation as the forward pass left it; column tap * Cp + c of the contraction is gathered per tap inside the kernel
 * (zero outside the image / clip).  dW receives the parameter's layout [N_real][C_real][taps] (taps = 9 / 3), cropped. */
int gcd_wgrad_conv_tr_f16(const void* dy16, int64_t lddy, const void* x16, int64_t ldx, int64_t M, int N, int Cp, int conv,
                          int Ho, int Wo, int T, int HW, int bf16, float* dW, int N_real, int C_real, int accumulate,
                          float* scratch, int64_t scratch_floats, void* stream);

/* ---- multi-tensor weight pack (train_ops.hip) -------------------------------------------------------------------------
 * One launch turns fp32 parameters into the 16-bit operand forms of the forward and backward GEMMs.  Entry: the parameter
 * viewed as [N][C][taps] contiguous (Linear / 1x1 conv: taps = 1; Conv2d 3x3: 9; Conv3d (3,1,1): 3), written, rounded once
 * (fp16, or bfloat16 when the launch says so), as
 *   dst_f[n * f_ns + tap * f_ts + c]                      (forward operand  W16[N][K], K order (tap, c)),            and / or
 *   dst_t[c * t_cs + tapx * t_ts + n],  tapx = mirror ? taps - 1 - tap : tap
 *                                                         (dgrad operand: W^T of a Linear; the tap-mirrored, role-swapped
 *                                                          weight of a stride-1 convolution's dgrad-as-convolution).
 * Padding (first / last convolution) is the caller's: destinations are allocated zeroed, only real elements are written.
 * tile0 = first workgroup of the entry (32 x 32 x taps tiles, tiles_c = ceil(C / 32)); entries sorted by tile0.
 * Replaces, per optimizer step, the casts / transposes / flips / permuted copies of every parameter
 * (diffusion.py:412-431 leaves this to autocast's weight casts). */
typedef struct gcd_pack_entry {
  const float* src;
  void* dst_f;
  void* dst_t;
  int64_t f_ns, f_ts, t_cs, t_ts;
  int32_t N, C, taps, mirror;
  int32_t tile0, tiles_c;
} gcd_pack_entry;
int gcd_train_pack_weights(const gcd_pack_entry* table_dev, int n_entries, int total_tiles, int bf16, void* stream);

/* ---- AlphaBlender (util.py:358-369) and its backward --------------------------------------------------------------------
 * y = a xs + (1 - a) xt with a per frame of rows_per_frame token rows (fp32, C % 4 == 0, leading dimensions % 4 == 0).
 * Backward: d_xs (+)= a dy (accumulate_xs adds onto d_xs; d_xs may be NULL), d_xt = (1 - a) dy, and — when d_alpha_zeroed is given —
 * d_alpha[frame] += sum dy (xs - xt) (atomic partial sums onto a buffer the caller zeroed). */
int gcd_blend_fwd_f32(const float* xs, int64_t ld_s, const float* xt, int64_t ld_t, const float* alpha, int64_t M, int C,
                      int64_t rows_per_frame, float* y, int64_t ld_y, void* stream);
int gcd_blend_bwd_f32(const float* dy, int64_t ld_dy, const float* xs, int64_t ld_s, const float* xt, int64_t ld_t,
                      const float* alpha, int64_t M, int C, int64_t rows_per_frame, float* d_xs, int64_t ld_dxs,
                      int accumulate_xs, float* d_xt, int64_t ld_dxt, float* d_alpha_zeroed, void* stream);

/* GroupNorm affine gradients from the AB[inst][C][2] doubles gcd_groupnorm_bwd (gcd_amd.h) leaves behind:
 * dbeta[c] (+)= sum over instances of AB[.][c][0], dgamma[c] (+)= ... AB[.][c][1] — one launch instead of a torch reduction,
 * two slices and two copies per GroupNorm. */
int gcd_gn_affine_grads(const double* AB, int ninst, int C, float* dgamma, float* dbeta, int accumulate, void* stream);

/* ---- grouped few-row Linears, fp32 ----------------------------------------------------------------------------------------
 * The network's M <= 32-row Linears (emb_layers of the 44 ResBlocks, openaimodel.py:287-293, 343-347; the one-key
 * cross-attention chains to_out(to_v(ctx)), attention.py:272-303; time_pos_embed; the embedding MLPs) as TABLES of
 * independent problems: one launch each for forward, dgrad and wgrad.  K % 4 == 0, W [N][K] contiguous fp32.
 *   fwd    y[m][n]  = sum_k act(x[m][k]) W[n][k] + b[n]        flags: 1 = act is SiLU (else identity), 4 = y += instead of =
 *                     blocks of a problem: ceil(N / 16)
 *   dgrad  dx[m][k] = dact(x[m][k]) sum_n y[m][n] W[n][k]      (y = the incoming gradient) flags: 1 = multiply by silu'(x),
 *                     4 = atomicAdd into dx (shared by several n slices / problems; zeroed by the caller) — a plain store is
 *                     legal only with N <= 64.  blocks: ceil(K / 256) * ceil(N / 64)
 *   wgrad  dW[n][k] = sum_m y[m][n] act(x[m][k]), db[n] = sum_m y[m][n]    flags: 1 = act is SiLU, 8 = accumulate onto dW / db
 *                     blocks: ceil(K / 256) * ceil(N / 64).  M may exceed 32 here (the rows are walked 32 at a time inside
 *                     the kernel); fwd and dgrad take larger M as several problems of <= 32 rows each
 * block0 = first workgroup of the problem in the launch; entries sorted by block0. */
typedef struct gcd_smallm_problem {
  const float* x;
  const float* W;
  const float* b;
  float* y;        /* fwd: output; dgrad / wgrad: the incoming gradient dy */
  float* dx;
  float* dW;
  float* db;
  int64_t ldx, ldy, lddx;
  int32_t M, N, K, flags;
  int32_t block0, reserved;
} gcd_smallm_problem;
int gcd_smallm_fwd(const gcd_smallm_problem* table_dev, int n_prob, int total_blocks, void* stream);
int gcd_smallm_dgrad(const gcd_smallm_problem* table_dev, int n_prob, int total_blocks, void* stream);
int gcd_smallm_wgrad(const gcd_smallm_problem* table_dev, int n_prob, int total_blocks, void* stream);

#ifdef __cplusplus
}
#endif
#endif
