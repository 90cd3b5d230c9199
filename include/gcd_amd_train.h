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

#define GCD_AMD_TRAIN_ABI_VERSION 1

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

#ifdef __cplusplus
}
#endif
#endif
