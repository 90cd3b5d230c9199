// ff_fused.hip — C ABI of the one-kernel FeedForward of the C = 320 level (ff_fused_kernel.h; round 6).
//   gcd_ff_packed_bytes / gcd_ff_pack_f16: the fragment-order weight stream, once per parameter version
//   gcd_ff_fused_f16: out = sa (W2 (value * gelu(gate)) + b2 + R1) + sr2 R2 from the LayerNorm'd tokens, or — the form the
//   engine uses — from the fp32 residual stream itself, LayerNorm included (x = ff(norm(x)) + x in one launch)
// Compiled with -fno-slp-vectorize (gcd_amd/csrc/build.py): a packed fp32 VALU instruction does not issue in the shadow
// of an MFMA (tools/issue_probe), and hipcc's SLP pass pairs the GELU polynomial's scalar FMAs into v_pk_fma_f32.
#include "ff_fused_kernel.h"

extern "C" int64_t gcd_ff_packed_bytes(void) { return (int64_t)(FF_NCH + 1) * FF_CHUNK_BYTES; }

extern "C" int gcd_ff_pack_f16(const void* w1, const void* w2, void* wp, int for_ln, void* stream) {
  GCD_CHECK_ARG(w1 && w2 && wp, "gcd_ff_pack_f16: null pointer");
  GCD_CHECK_ARG(((uintptr_t)w1 & 15) == 0 && ((uintptr_t)wp & 15) == 0, "gcd_ff_pack_f16: w1 / wp must be 16-byte aligned");
  const int n = (FF_NCH + 1) * 60 * 64;
  ff_pack_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>((const f16*)w1, (const f16*)w2, (f16*)wp, for_ln ? 1 : 0);
  GCD_CHECK_LAUNCH();
  return 0;
}

extern "C" int gcd_ff_fused_supported(int M, int C, int hidden) {
  return M >= 1 && C == FF_C && hidden == FF_HID;
}

namespace {
template <int EPI, bool LN>
int launch(const FfK& k, hipStream_t s) {
  static GcdPerDeviceOnce once;
  GCD_CHECK_HIP(once.opt_in((const void*)ff_fused_kernel<2, 19, EPI, LN>, FF_SMEM));
  static std::atomic<int> cus{0};
  int n = cus.load(std::memory_order_relaxed);
  if (n == 0) {
    int d = 0;
    GCD_CHECK_HIP(hipGetDevice(&d));
    GCD_CHECK_HIP(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d));
    cus.store(n, std::memory_order_relaxed);
  }
  const int ntiles = (k.M + 127) / 128;
  ff_fused_kernel<2, 19, EPI, LN><<<ntiles < n ? ntiles : n, 256, FF_SMEM, s>>>(k);
  GCD_CHECK_LAUNCH();
  return 0;
}
}  // namespace

extern "C" int gcd_ff_fused_f16(const gcd_ff_desc* d, void* stream) {
  GCD_CHECK_ARG(d, "gcd_ff_fused_f16: null descriptor");
  GCD_CHECK_ARG(d->M >= 1, "gcd_ff_fused_f16: M = %d", d->M);
  GCD_CHECK_ARG(d->C == FF_C && d->hidden == FF_HID, "gcd_ff_fused_f16: C = %d, hidden = %d (this kernel is the 320 / 1280 level)",
                d->C, d->hidden);
  const bool ln = d->ln_gamma != nullptr;
  GCD_CHECK_ARG(d->wp && d->b1 && d->b2 && d->out, "gcd_ff_fused_f16: null operand");
  if (ln) {
    GCD_CHECK_ARG(d->x32 && d->ln_beta, "gcd_ff_fused_f16: the LayerNorm form needs x32, ln_gamma and ln_beta");
    GCD_CHECK_ARG(!d->X && !d->R1, "gcd_ff_fused_f16: the LayerNorm form reads x32 only (it IS the residual): X and R1 must be NULL");
    GCD_CHECK_ARG(d->ldx32 >= FF_C && d->ldx32 % 4 == 0 && ((uintptr_t)d->x32 & 15) == 0,
                  "gcd_ff_fused_f16: x32 rows must be 16-byte aligned");
    GCD_CHECK_ARG(((uintptr_t)d->ln_gamma & 15) == 0 && ((uintptr_t)d->ln_beta & 15) == 0,
                  "gcd_ff_fused_f16: ln_gamma / ln_beta must be 16-byte aligned");
    GCD_CHECK_ARG(!d->addvec || (d->rows_per_vec > 0 && d->rows_per_vec % 32 == 0 && d->ld_addvec % 4 == 0 &&
                                 ((uintptr_t)d->addvec & 15) == 0),
                  "gcd_ff_fused_f16: addvec needs rows_per_vec %% 32 == 0 (one vector per wave tile) and 16-byte aligned rows");
  } else {
    GCD_CHECK_ARG(d->X, "gcd_ff_fused_f16: null operand X");
    GCD_CHECK_ARG(d->R1, "gcd_ff_fused_f16: the residual R1 is required (it initialises the accumulators)");
    GCD_CHECK_ARG(d->ldx >= FF_C && d->ldx % 8 == 0 && ((uintptr_t)d->X & 15) == 0, "gcd_ff_fused_f16: X rows must be 16-byte aligned");
    GCD_CHECK_ARG(d->ldr1 >= FF_C && d->ldr1 % 4 == 0 && ((uintptr_t)d->R1 & 15) == 0, "gcd_ff_fused_f16: R1 rows must be 16-byte aligned");
    GCD_CHECK_ARG(!d->addvec, "gcd_ff_fused_f16: addvec belongs to the LayerNorm form");
  }
  GCD_CHECK_ARG(!d->R2 || (d->ldr2 >= FF_C && d->ldr2 % 4 == 0 && ((uintptr_t)d->R2 & 15) == 0),
                "gcd_ff_fused_f16: R2 rows must be 16-byte aligned");
  GCD_CHECK_ARG(d->out_kind == GCD_OUT_F32 || d->out_kind == GCD_OUT_F16, "gcd_ff_fused_f16: out_kind %d", d->out_kind);
  GCD_CHECK_ARG(d->ldo >= FF_C && d->ldo % 4 == 0 && ((uintptr_t)d->out & 15) == 0, "gcd_ff_fused_f16: out rows must be 16-byte aligned");
  GCD_CHECK_ARG(((uintptr_t)d->wp & 15) == 0 && ((uintptr_t)d->b1 & 15) == 0 && ((uintptr_t)d->b2 & 15) == 0,
                "gcd_ff_fused_f16: wp / b1 / b2 must be 16-byte aligned");
  GCD_CHECK_ARG(!d->frame_alpha || (d->rows_per_alpha > 0 && d->rows_per_alpha % 32 == 0),
                "gcd_ff_fused_f16: rows_per_alpha = %d must be a multiple of 32 (one alpha per wave tile)", d->rows_per_alpha);
  GCD_CHECK_ARG(d->out_kind == GCD_OUT_F32 || d->R2, "gcd_ff_fused_f16: the fp16 result exists for the blended form only");
  FfK k;
  k.X = (const f16*)d->X;
  k.ldx = d->ldx;
  k.x32 = d->x32;
  k.ldx32 = d->ldx32;
  k.ln_gamma = d->ln_gamma;
  k.ln_beta = d->ln_beta;
  k.ln_eps = d->ln_eps;
  k.addvec = d->addvec;
  k.ld_addvec = d->ld_addvec;
  k.rows_per_vec = d->rows_per_vec;
  k.Wp = (const f16*)d->wp;
  k.b1 = d->b1;
  k.b2 = d->b2;
  k.R1 = d->R1;
  k.ldr1 = d->ldr1;
  k.R2 = d->R2;
  k.ldr2 = d->ldr2;
  k.out = d->out;
  k.ldo = d->ldo;
  k.out_f16 = d->out_kind == GCD_OUT_F16;
  k.frame_alpha = d->frame_alpha;
  k.rows_per_alpha = d->rows_per_alpha;
  k.s_acc = d->s_acc;
  k.s_r2 = d->s_r2;
  k.M = d->M;
  k.sched = d->sched;
  k.dbg = nullptr;
  hipStream_t s = (hipStream_t)stream;
  if (ln) {
    if (!d->R2) return launch<0, true>(k, s);
    return d->out_kind == GCD_OUT_F16 ? launch<2, true>(k, s) : launch<1, true>(k, s);
  }
  if (!d->R2) return launch<0, false>(k, s);
  return d->out_kind == GCD_OUT_F16 ? launch<2, false>(k, s) : launch<1, false>(k, s);
}
