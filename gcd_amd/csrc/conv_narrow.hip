// conv_narrow.hip — the 3x3 convolution with at most 16 output channels: the UNet's output head
// (GroupNorm -> SiLU -> Conv2d(320, 4, 3, padding=1), openaimodel.py / video_model.py:455-459: N = 4, padded to 16).
//
// On the 256 x 320 / 128 x 160 GEMM tiles this launch spends 10-20x its useful matrix work on padded columns and
// re-stages the A panel per tile: 274-380 us at 72 x 128 for 5.9 GFLOP.  What it has to do is read the fp16 input
// (165 MB at 72 x 128) once from HBM; everything else is small.  So:
//   * the whole weight tensor [16, 9 Cin] fp16 (92 KB at Cin = 320) sits in LDS in MFMA fragment order, filled once per
//     workgroup;
//   * the contraction runs transposed, out^T[n][pixel] = W[n][k] patches^T[k][pixel], with v_mfma_f32_16x16x32_f16: the
//     weight fragment is the A operand (ds_read_b128), the patch fragment is the B operand and comes STRAIGHT from global
//     memory — lane (pixel c, k group g) reads the 16 bytes of channels c0 + 8g .. 8g + 7 of its (shifted) input pixel, so
//     the four lanes of a pixel cover 64 contiguous bytes and the ten steps of a tap walk the pixel's 640-byte row: every
//     fetched line is used whole; zero padding = a range-checked buffer load with an out-of-range offset;
//   * a wave owns a strip of 16 pixels x CN_R output rows and marches DOWN it input row by input row: one input row feeds
//     the three output rows it belongs to (three accumulators, rotated), and its two horizontally shifted operands are made
//     in registers (DPP row shifts + a two-lane halo load), so a strip reads every input line ONCE (+ 2 / CN_R for the rows
//     above and below the segment).  Measured on the way here, 72 x 128, cold caches: output-stationary, nine taps per 32
//     pixels 176-193 us; row marching with three shifted reads per row 100-102 us at one AND at two units of prefetch —
//     both at 10-13 bytes per clock and CU of vector-L1 misses, the rate every kernel of this library sees for loads that
//     miss the L1 (wherever they then hit): the lever is fewer L1 misses, not more of them in flight;
//   * the loads of the next input row are in flight under the MFMAs of the current one (two register sets);
//   * the accumulator layout holds 4 consecutive n of one pixel per lane: one 16-byte store, 1 KB contiguous per wave.
// Same arithmetic as the general kernel (fp16 products, fp32 accumulation) up to summation order.
#include "common.h"
#include "gemm_common.h"

namespace {

// NCH = Cin / 32: 32-channel chunks per tap; a workgroup = CN_WAVES strips of 16 pixels x CN_R output rows
template <int NCH, int CN_WAVES, int CN_R>
__global__ __launch_bounds__(CN_WAVES * 64) void conv3x3_narrow_kernel(const GemmK p, const int nxb, const int nseg,
                                                                       const int nstrips) {
  static_assert(CN_R % 2 == 0, "the row loop is unrolled by two");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int c = lane & 15, g = lane >> 4;
  const int K = 9 * NCH * 32;
  // weights -> LDS, fragment order: step j (tap, chunk), lane l holds W[l % 16][32 j + 8 (l / 16) .. + 7]
  for (int idx = t; idx < 9 * NCH * 64; idx += CN_WAVES * 64) {
    const int j = idx >> 6, l = idx & 63;
    *(f16x8*)(smem + idx * 16) = *(const f16x8*)(p.W + (int64_t)(l & 15) * K + j * 32 + 8 * (l >> 4));
  }
  __syncthreads();
  const int strip = blockIdx.x * CN_WAVES + wave;      // (frame, row segment, 16-pixel column block), column block fastest
  if (strip >= nstrips) return;
  const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)p.a_bytes, 0x00020000);
  const f32x4 bv = p.bias ? *(const f32x4*)(p.bias + 4 * g) : f32x4{0.f, 0.f, 0.f, 0.f};
  const int xb = strip % nxb, fs = strip / nxb, seg = fs % nseg, f = fs / nseg;
  const int x = 16 * xb + c, r0 = seg * CN_R;
  const int pitch = (int)p.lda * 2;      // bytes per pixel
  const unsigned char* wl = smem + lane * 16;

  // one input row = NCH centre loads (the strip's 16 pixels) + NCH halo loads (lane c = 0: pixel x0 - 1, c = 15: x0 + 16, the
  // other lanes out of range: no traffic); the two horizontally shifted operands are made from them by DPP row shifts
  f16x8 ctrA[NCH], halA[NCH], ctrB[NCH], halB[NCH];
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0, acc2 = acc0;      // output rows iy - 1, iy, iy + 1 of the input row iy in work
  const int xh = c == 0 ? x - 1 : x + 1;
  const bool halo_lane = c == 0 || c == 15;
  auto issue = [&](int iy, f16x8 (&ctr)[NCH], f16x8 (&hal)[NCH]) {
    const bool okr = (unsigned)iy < (unsigned)p.Hi && iy <= r0 + CN_R;
    const int rowo = (f * p.Hi + iy) * p.Wi;
    const int oc = okr && x < p.Wi ? (rowo + x) * pitch + 16 * g : (int)0x7fffffff;      // out of range -> zeros
    const int oh = okr && halo_lane && (unsigned)xh < (unsigned)p.Wi ? (rowo + xh) * pitch + 16 * g : (int)0x7fffffff;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      ctr[ch] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rsrcA, oc, ch * 64, 0));
      hal[ch] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rsrcA, oh, ch * 64, 0));
    }
  };
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  auto mma = [&](const f16x8 (&ctr)[NCH], const f16x8 (&hal)[NCH]) {
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      const i32x4 cv = __builtin_bit_cast(i32x4, ctr[ch]), hv = __builtin_bit_cast(i32x4, hal[ch]);
      i32x4 lv, rv;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        lv[e] = __builtin_amdgcn_update_dpp(hv[e], cv[e], 0x111, 0xf, 0xf, false);      // row_shr:1: lane c <- c - 1; c = 0 keeps the halo
        rv[e] = __builtin_amdgcn_update_dpp(hv[e], cv[e], 0x101, 0xf, 0xf, false);      // row_shl:1: lane c <- c + 1; c = 15 keeps the halo
      }
      const f16x8 frag[3] = {__builtin_bit_cast(f16x8, lv), ctr[ch], __builtin_bit_cast(f16x8, rv)};
#pragma unroll
      for (int kwi = 0; kwi < 3; ++kwi) {
        // tap (kh, kw) of output row oy reads input row oy + kh - 1: input row iy is kh = 2 for oy = iy - 1, 1 for iy, 0 for iy + 1
        const f16x8 w2 = *(const f16x8*)(wl + ((6 + kwi) * NCH + ch) * 1024);
        const f16x8 w1 = *(const f16x8*)(wl + ((3 + kwi) * NCH + ch) * 1024);
        const f16x8 w0 = *(const f16x8*)(wl + ((0 + kwi) * NCH + ch) * 1024);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2, frag[kwi], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1, frag[kwi], acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0, frag[kwi], acc2, 0, 0, 0);
      }
    }
  };
  auto finish_row = [&](int iy) {      // input row iy done: output row iy - 1 is complete
    const int oy = iy - 1;
    if (oy >= r0 && oy < r0 + CN_R && oy < p.Ho && x < p.Wo)
      *(f32x4*)((float*)p.out + ((int64_t)(f * p.Ho + oy) * p.Wo + x) * p.ldo + 4 * g) = (acc0 + bv) * p.s_acc;
    acc0 = acc1;
    acc1 = acc2;
    acc2 = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  // two register sets, one input row ahead (2 NCH loads in flight per wave under 9 NCH MFMAs); the scheduling barriers keep
  // hipcc from hoisting a third row's loads over a compute phase
  issue(r0 - 1, ctrA, halA);
#pragma unroll 1
  for (int iy = r0 - 1; iy <= r0 + CN_R; iy += 2) {
    issue(iy + 1, ctrB, halB);
    __builtin_amdgcn_sched_barrier(0);
    mma(ctrA, halA);
    __builtin_amdgcn_sched_barrier(0);
    finish_row(iy);
    issue(iy + 2, ctrA, halA);
    __builtin_amdgcn_sched_barrier(0);
    mma(ctrB, halB);
    __builtin_amdgcn_sched_barrier(0);
    finish_row(iy + 1);
  }
}

}      // namespace

// The shapes the narrow kernel takes: N == 16, stride 1, no upsample, fp32 output with bias and scalar scale only.
bool gcd_conv3x3_narrow_supported(const GemmK& k, int mode) {
  if (mode != GCD_GEMM_CONV3X3 || k.N != 16 || k.stride != 1 || k.up || k.asym) return false;
  if (k.Cin != 320 && k.Cin != 64) return false;
  if (k.out_kind != GCD_OUT_F32 || k.R1 || k.R2 || k.rowvec || k.frame_alpha || k.colstats || k.ln_out || k.operand_bf16 ||
      k.out_blocked || k.a_blocked || k.splitk != 1)
    return false;
  if (k.Ho != k.Hi || k.Wo != k.Wi || k.lda % 8 != 0 || k.ldo % 4 != 0) return false;
  if ((int64_t)k.M * k.lda * 2 >= (int64_t)0x7fffffff) return false;      // 32-bit buffer offsets
  return true;
}

namespace {
template <int NCH, int WAVES, int R>
int launch_narrow(const GemmK& k, hipStream_t s) {
  static GcdPerDeviceOnce once;
  const int frames = k.M / (k.Ho * k.Wo);
  const int nxb = (k.Wo + 15) / 16, nseg = (k.Ho + R - 1) / R, nstrips = frames * nseg * nxb;
  const int lds = 9 * k.Cin * 16 * 2;
  GCD_CHECK_HIP(once.opt_in((const void*)conv3x3_narrow_kernel<NCH, WAVES, R>, lds));
  hipLaunchKernelGGL((conv3x3_narrow_kernel<NCH, WAVES, R>), dim3((nstrips + WAVES - 1) / WAVES), dim3(WAVES * 64), lds, s, k,
                     nxb, nseg, nstrips);
  GCD_CHECK_LAUNCH();
  return 0;
}
}      // namespace

int gcd_conv3x3_narrow_launch(const GemmK& k0, hipStream_t s) {
  GemmK k = k0;
  k.a_bytes = (uint32_t)((int64_t)k.M * k.lda * 2);
  static const int variant = getenv("GCD_CONV_NARROW") ? atoi(getenv("GCD_CONV_NARROW")) : 0;      // (development: strip shape A/B)
  if (k.Cin == 320)
    return variant == 1 ? launch_narrow<10, 8, 4>(k, s) : variant == 2 ? launch_narrow<10, 8, 2>(k, s)
         : variant == 3 ? launch_narrow<10, 4, 8>(k, s) : launch_narrow<10, 8, 8>(k, s);
  return launch_narrow<2, 8, 8>(k, s);
}
