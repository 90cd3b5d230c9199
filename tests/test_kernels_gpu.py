"""Per-kernel parity of libgcd_amd against plain PyTorch fp32 references (CPU).

Operand precision is fp16 with fp32 accumulation, so inputs are rounded to fp16 *before* both
paths; the remaining error is accumulation order + the fp16 rounding of fp16 outputs.
Tolerances (rel-L2): fp32 outputs 2e-5 (norms) / 1e-4 (contractions), fp16 outputs 6e-4.
"""
import math

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_l2

pytestmark = pytest.mark.gpu

TOL_F32 = 1e-4
TOL_F16 = 6e-4


def _h(t):  # round to fp16 and back: the operand both paths see
    return t.to(torch.float16).float()


def _gen(seed):
    return torch.Generator().manual_seed(seed)


_IMPL_IDS = {0: "auto", 2: "tile256x320", 3: "ring32", 6: "tile64"}


def pytest_generate_tests(metafunc):
    """Every GEMM test of this module runs with the automatic GEMM kernel choice, with the 256x320 tile kernels forced
    (2: the 8-phase 16x16x32 K loop of gemm_p8.hip wherever it applies; 3: the 32-deep ring kernel of gemm_pp.hip
    only) and with the general kernel's 64-row tiles forced; tests that launch no GEMM run once."""
    if "gemm_impl" in metafunc.fixturenames:
        gemmish = any(k in metafunc.function.__name__ for k in ("gemm", "conv", "feedforward", "graph_capture"))
        vals = [0, 2, 3, 6] if gemmish else [0]
        metafunc.parametrize("gemm_impl", vals, ids=[_IMPL_IDS[v] for v in vals], indirect=True)


@pytest.fixture(autouse=True)
def gemm_impl(request):
    from gcd_amd import ops
    ops.tune_set(ops.TUNE_GEMM_IMPL, request.param)
    yield request.param
    ops.tune_set(ops.TUNE_GEMM_IMPL, 0)


# ------------------------------------------------------------------------------------------ GEMM
def test_gemm_pingpong_k32_and_persistent(gpu, gemm_impl):
    """Shapes only the ping-pong kernel takes: K = 96 (three 32-deep sub-tiles, not a multiple of the
    general kernel's 64) and a 40 x 8 tile grid (> 256 tiles: the persistent tile loop), ragged in
    M and N, with every epilogue input."""
    from gcd_amd import ops
    if gemm_impl == 6:
        pytest.skip("general kernel needs K % 64 == 0")
    g = _gen(21)
    for (M, N, K) in [(700, 352, 96), (256 * 39 + 77, 320 * 7 + 64, 64)]:
        a = _h(torch.randn(M, K, generator=g))
        w = _h(torch.randn(N, K, generator=g) / math.sqrt(K))
        bias = torch.randn(N, generator=g)
        r1 = torch.randn(M, N, generator=g)
        rows = 333
        rv = torch.randn((M + rows - 1) // rows, N, generator=g)
        ref = a @ w.t() + bias + rv.repeat_interleave(rows, 0)[:M] + r1
        out = torch.empty(M, N, device=gpu)
        ops.gemm(a.half().to(gpu), w.half().to(gpu), out, M=M, bias=bias.to(gpu), rowvec=rv.to(gpu),
                 rows_per_vec=rows, r1=r1.to(gpu))
        torch.cuda.synchronize()
        e = rel_l2(out, ref)
        assert e < TOL_F32, f"M={M} N={N} K={K}: rel-L2 {e:.3e}"
        out16 = torch.empty(M, N, device=gpu, dtype=torch.float16)
        ops.gemm(a.half().to(gpu), w.half().to(gpu), out16, M=M, out_kind=ops.OUT_F16)
        torch.cuda.synchronize()
        e = rel_l2(out16.float(), a @ w.t())
        assert e < TOL_F16, f"f16 out M={M} N={N} K={K}: rel-L2 {e:.3e}"


def test_gemm_split_k(gpu, gemm_impl):
    """Few output tiles and a long K: the automatic choice splits K over 4 workgroups per tile (fp32
    partial sums in the scratch + a reduce kernel that runs the whole epilogue).  Linear and 3x3 conv."""
    from gcd_amd import ops, packing
    g = _gen(24)
    M, N, K = 1000, 640, 7680
    a = _h(torch.randn(M, K, generator=g))
    w = _h(torch.randn(N, K, generator=g) / math.sqrt(K))
    bias = torch.randn(N, generator=g)
    r1, r2 = torch.randn(M, N, generator=g), torch.randn(M, N, generator=g)
    rows = 125
    alpha = torch.rand(M // rows, generator=g)
    al = alpha.repeat_interleave(rows)[:, None]
    ref = al * r2 + (1 - al) * (a @ w.t() + bias + r1)
    out = torch.empty(M, N, device=gpu)
    ops.gemm(a.half().to(gpu), w.half().to(gpu), out, M=M, bias=bias.to(gpu), r1=r1.to(gpu),
             r2=r2.to(gpu), frame_alpha=alpha.to(gpu), rows_per_alpha=rows, r1_blend=True)
    torch.cuda.synchronize()
    assert rel_l2(out, ref) < TOL_F32
    out16 = torch.empty(M, N, device=gpu, dtype=torch.float16)
    ops.gemm(a.half().to(gpu), w.half().to(gpu), out16, M=M, out_kind=ops.OUT_F16)
    torch.cuda.synchronize()
    assert rel_l2(out16.float(), a @ w.t()) < TOL_F16
    frames, Cin, Cout, H, W = 4, 896, 320, 9, 16           # K = 8064: slices start mid-tap
    x = _h(torch.randn(frames, Cin, H, W, generator=g))
    wc = _h(torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin))
    b = torch.randn(Cout, generator=g)
    refc = F.conv2d(x, wc, b, padding=1).permute(0, 2, 3, 1).reshape(frames * H * W, Cout)
    ac = x.permute(0, 2, 3, 1).reshape(frames * H * W, Cin).half().to(gpu)
    outc = torch.empty(frames * H * W, Cout, device=gpu)
    ops.gemm(ac, packing.pack_conv3x3(wc).to(gpu), outc, M=frames * H * W, mode=ops.GEMM_CONV3X3,
             bias=b.to(gpu), conv=dict(Cin=Cin, Hi=H, Wi=W, Ho=H, Wo=W, stride=1, upsample=0))
    torch.cuda.synchronize()
    assert rel_l2(outc, refc) < TOL_F32


def test_conv3x3_pingpong_full_tiles(gpu, gemm_impl):
    """A conv whose grid fills whole 256 x 320 tiles (what the automatic choice sends to the ping-pong
    kernel): 320 -> 320 channels on 28 frames of 24 x 32, stride 1 / stride 2 / fused x2 upsample."""
    from gcd_amd import ops, packing
    g = _gen(22)
    frames, Cin, Cout = 28, 320, 320
    for (H, W, stride, up) in [(24, 32, 1, 0), (24, 32, 2, 0), (12, 16, 1, 1)]:
        x = _h(torch.randn(frames, Cin, H, W, generator=g))
        w = _h(torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin))
        b = torch.randn(Cout, generator=g)
        xin = F.interpolate(x, scale_factor=2, mode="nearest") if up else x
        ref = F.conv2d(xin, w, b, stride=stride, padding=1)
        Ho, Wo = ref.shape[-2:]
        ref_tok = ref.permute(0, 2, 3, 1).reshape(frames * Ho * Wo, Cout)
        a = x.permute(0, 2, 3, 1).reshape(frames * H * W, Cin).half().to(gpu)
        out = torch.empty(frames * Ho * Wo, Cout, device=gpu)
        ops.gemm(a, packing.pack_conv3x3(w).to(gpu), out, M=frames * Ho * Wo, mode=ops.GEMM_CONV3X3,
                 bias=b.to(gpu), conv=dict(Cin=Cin, Hi=H, Wi=W, Ho=Ho, Wo=Wo, stride=stride, upsample=up))
        torch.cuda.synchronize()
        e = rel_l2(out, ref_tok)
        assert e < TOL_F32, f"conv3x3 {H}x{W} s{stride} up{up}: rel-L2 {e:.3e}"


@pytest.mark.parametrize("frames,Cin,H,W", [(3, 320, 9, 16), (2, 320, 36, 64), (5, 64, 7, 13), (2, 320, 18, 40)])
def test_conv3x3_narrow_output_head(gpu, frames, Cin, H, W):
    """The UNet's output head, Conv2d(C, 4, 3, padding=1) with the four output channels padded to 16 (openaimodel.py
    `self.out`, video_model.py:455-459): the N == 16 kernel of conv_narrow.hip (weights resident in LDS, patches straight
    from global memory with range-checked zero padding) against torch's fp32 conv on the same fp16-rounded operands and
    against the general kernel on the same descriptor; ragged pixel counts (M not a multiple of the 256-pixel task), image
    borders in every lane position, a scalar output scale."""
    from gcd_amd import ops, packing
    g = _gen(31)
    x = _h(torch.randn(frames, Cin, H, W, generator=g))
    w = _h(torch.randn(4, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin))
    b = torch.randn(4, generator=g)
    ref = F.conv2d(x, w, b, padding=1).permute(0, 2, 3, 1).reshape(frames * H * W, 4)
    a = x.permute(0, 2, 3, 1).reshape(frames * H * W, Cin).half().to(gpu)
    wp = packing.pack_conv3x3(w, cout_pad=16).to(gpu)
    bp = torch.zeros(16)
    bp[:4] = b
    M = frames * H * W
    conv = dict(Cin=Cin, Hi=H, Wi=W, Ho=H, Wo=W, stride=1, upsample=0)
    out = torch.full((M, 16), float("nan"), device=gpu)
    ops.gemm(a, wp, out, M=M, mode=ops.GEMM_CONV3X3, bias=bp.to(gpu), conv=conv)
    torch.cuda.synchronize()
    e = rel_l2(out[:, :4], ref)
    assert e < TOL_F32, f"narrow conv {frames}x{H}x{W}, Cin {Cin}: rel-L2 {e:.3e}"
    assert torch.equal(out[:, 4:], torch.zeros_like(out[:, 4:]))          # padded columns: zero weights, zero bias
    gen = torch.empty(M, 16, device=gpu)
    ops.tune_set(ops.TUNE_GEMM_IMPL, 1)
    try:
        ops.gemm(a, wp, gen, M=M, mode=ops.GEMM_CONV3X3, bias=bp.to(gpu), conv=conv)
    finally:
        ops.tune_set(ops.TUNE_GEMM_IMPL, 0)
    e2 = rel_l2(out, gen)
    print(f"narrow conv Cin {Cin} {frames}x{H}x{W}: vs fp32 torch {e:.2e}, vs the general kernel {e2:.2e}")
    assert e2 < 2e-6
    ops.gemm(a, wp, gen, M=M, mode=ops.GEMM_CONV3X3, bias=bp.to(gpu), conv=conv, s_acc=0.25)
    torch.cuda.synchronize()
    assert rel_l2(gen, 0.25 * out) < 1e-6


@pytest.mark.parametrize("M,N,K", [(300, 320, 320), (128, 256, 64), (1000, 64, 640),
                                   (257, 960, 128), (4032, 1280, 1280), (112, 48, 192)])
def test_gemm_plain_bias_residual(gpu, M, N, K):
    from gcd_amd import ops
    g = _gen(1)
    a = _h(torch.randn(M, K, generator=g))
    w = _h(torch.randn(N, K, generator=g) / math.sqrt(K))
    bias = torch.randn(N, generator=g)
    rows_per_vec = 50
    rv = torch.randn((M + rows_per_vec - 1) // rows_per_vec, N, generator=g)
    r1 = torch.randn(M, N, generator=g)
    r2 = torch.randn(M, N, generator=g)
    ref = 0.7 * (a @ w.t() + bias + rv.repeat_interleave(rows_per_vec, 0)[:M]) + 0.5 * r1 - 1.25 * r2
    out = torch.empty(M, N, device=gpu)
    ops.gemm(a.half().to(gpu), w.half().to(gpu), out, M=M, bias=bias.to(gpu), rowvec=rv.to(gpu),
             rows_per_vec=rows_per_vec, r1=r1.to(gpu), r2=r2.to(gpu), s_acc=0.7, s_r1=0.5, s_r2=-1.25)
    torch.cuda.synchronize()
    e = rel_l2(out, ref)
    assert e < TOL_F32, f"gemm plain M={M} N={N} K={K}: rel-L2 {e:.3e}"
    # fp16 output, no epilogue extras, transpose-detecting (asymmetric W)
    out16 = torch.empty(M, N, device=gpu, dtype=torch.float16)
    ops.gemm(a.half().to(gpu), w.half().to(gpu), out16, M=M, out_kind=ops.OUT_F16)
    torch.cuda.synchronize()
    e = rel_l2(out16.float(), a @ w.t())
    assert e < TOL_F16, f"gemm f16 out M={M} N={N} K={K}: rel-L2 {e:.3e}"


@pytest.mark.parametrize("M,N,K", [(768, 320, 64), (1024, 640, 320), (66 * 1024, 320, 96)])
def test_gemm_full_tile_fast_epilogues(gpu, M, N, K):
    """Whole 256 x 320 tiles take the row-major pipelined epilogues of the ping-pong kernel
    (gemm_common.h: gcd_epi_f32_rows_full / gcd_epi_geglu_rows_full / gcd_epi_f16_rows_full); the last
    shape has more tiles than CUs (persistent walk).  Every combination the UNet uses, with the
    tile-uniform and the non-uniform (generic fallback) forms of rowvec / frame_alpha."""
    from gcd_amd import ops, packing
    ops.tune_set(ops.TUNE_GEMM_IMPL, 2)
    try:
        g = _gen(31)
        a = _h(torch.randn(M, K, generator=g))
        w = _h(torch.randn(N, K, generator=g) / math.sqrt(K))
        bias = torch.randn(N, generator=g)
        r1 = torch.randn(M, N, generator=g)
        r2 = torch.randn(M, N, generator=g)
        ag, wg = a.half().to(gpu), w.half().to(gpu)
        acc = a @ w.t()
        # residual in place + bias (attention out / FF out / proj_out)
        x = r1.to(gpu).clone()
        ops.gemm(ag, wg, x, M=M, bias=bias.to(gpu), r1=x)
        assert rel_l2(x, acc + bias + r1) < TOL_F32
        # bias only, scaled
        out = torch.empty(M, N, device=gpu)
        ops.gemm(ag, wg, out, M=M, bias=bias.to(gpu), s_acc=0.37)
        assert rel_l2(out, 0.37 * (acc + bias)) < TOL_F32
        for rows in (256, 512, 100):      # per-frame vector / alpha: uniform per tile (256, 512) or not (100)
            rv = torch.randn((M + rows - 1) // rows, N, generator=g)
            alpha = torch.rand((M + rows - 1) // rows, generator=g)
            al = alpha.repeat_interleave(rows)[:M, None]
            rvx = rv.repeat_interleave(rows, 0)[:M]
            # conv1 / attention-out form: bias + rowvec (+ residual)
            ops.gemm(ag, wg, out, M=M, bias=bias.to(gpu), rowvec=rv.to(gpu), rows_per_vec=rows, r1=r1.to(gpu))
            assert rel_l2(out, acc + bias + rvx + r1) < TOL_F32, rows
            # transformer blend: alpha*x + (1-alpha)*(acc + bias + x_mix), fp32 and fp16 results
            ref = al * r2 + (1 - al) * (acc + bias + r1)
            ops.gemm(ag, wg, out, M=M, bias=bias.to(gpu), r1=r1.to(gpu), r2=r2.to(gpu),
                     frame_alpha=alpha.to(gpu), rows_per_alpha=rows, r1_blend=True)
            assert rel_l2(out, ref) < TOL_F32, rows
            out16 = torch.empty(M, N, device=gpu, dtype=torch.float16)
            ops.gemm(ag, wg, out16, M=M, bias=bias.to(gpu), r1=r1.to(gpu), r2=r2.to(gpu),
                     frame_alpha=alpha.to(gpu), rows_per_alpha=rows, r1_blend=True, out_kind=ops.OUT_F16)
            assert rel_l2(out16.float(), ref) < TOL_F16, rows
            # resblock blend: x_s + (1-alpha)*(acc + bias)
            ops.gemm(ag, wg, out, M=M, bias=bias.to(gpu), r1=r1.to(gpu), frame_alpha=alpha.to(gpu),
                     rows_per_alpha=rows, r1_blend=False)
            assert rel_l2(out, r1 + (1 - al) * (acc + bias)) < TOL_F32, rows
        # fp16 result without residuals (q|k|v), with and without bias
        out16 = torch.empty(M, N, device=gpu, dtype=torch.float16)
        ops.gemm(ag, wg, out16, M=M, out_kind=ops.OUT_F16)
        assert rel_l2(out16.float(), acc) < TOL_F16
        ops.gemm(ag, wg, out16, M=M, bias=bias.to(gpu), out_kind=ops.OUT_F16)
        assert rel_l2(out16.float(), acc + bias) < TOL_F16
        # GEGLU
        wp, bp = packing.pack_geglu(w, bias)
        hid = torch.empty(M, N // 2, device=gpu, dtype=torch.float16)
        ops.gemm(ag, wp.to(gpu), hid, M=M, bias=bp.to(gpu), out_kind=ops.OUT_GEGLU)
        lin = acc + bias
        ref = lin[:, :N // 2] * F.gelu(lin[:, N // 2:])
        assert rel_l2(hid.float(), ref) < TOL_F16
        torch.cuda.synchronize()
    finally:
        ops.tune_set(ops.TUNE_GEMM_IMPL, 0)


def test_gemm_inplace_residual_and_ld(gpu):
    """R1 aliases out (residual stream update in place) and operands are strided views."""
    from gcd_amd import ops
    g = _gen(2)
    M, N, K = 500, 320, 192
    abuf = _h(torch.randn(M, K + 64, generator=g))
    a = abuf[:, 8:8 + K]
    w = _h(torch.randn(N, K, generator=g) / math.sqrt(K))
    xbuf = torch.randn(M, N + 32, generator=g)
    ref = xbuf[:, 4:4 + N] + a @ w.t()
    xg = xbuf.to(gpu)
    xv = xg[:, 4:4 + N]
    ag = abuf.half().to(gpu)[:, 8:8 + K]
    ops.gemm(ag, w.half().to(gpu), xv, M=M, r1=xv)
    torch.cuda.synchronize()
    assert rel_l2(xv, ref) < TOL_F32
    # columns outside the view are untouched
    assert torch.equal(xg[:, :4].cpu(), xbuf[:, :4]) and torch.equal(xg[:, 4 + N:].cpu(), xbuf[:, 4 + N:])


def test_gemm_frame_alpha_blend(gpu):
    from gcd_amd import ops
    g = _gen(3)
    rows_per_frame, frames, N, K = 48, 6, 160, 128
    M = rows_per_frame * frames
    a = _h(torch.randn(M, K, generator=g))
    w = _h(torch.randn(N, K, generator=g) / math.sqrt(K))
    bias = torch.randn(N, generator=g)
    r1 = torch.randn(M, N, generator=g)
    r2 = torch.randn(M, N, generator=g)
    alpha = torch.tensor([0.3, 1.0, 0.62, 0.0, 1.0, 0.5])
    al = alpha.repeat_interleave(rows_per_frame)[:, None]
    # transformer form: alpha*x + (1-alpha)*(acc + bias + x3)
    ref = al * r2 + (1 - al) * (a @ w.t() + bias + r1)
    out = torch.empty(M, N, device=gpu)
    ops.gemm(a.half().to(gpu), w.half().to(gpu), out, M=M, bias=bias.to(gpu), r1=r1.to(gpu),
             r2=r2.to(gpu), frame_alpha=alpha.to(gpu), rows_per_alpha=rows_per_frame, r1_blend=True)
    torch.cuda.synchronize()
    assert rel_l2(out, ref) < TOL_F32
    # resblock form: x_s + (1-alpha)*(acc + bias)
    ref = r1 + (1 - al) * (a @ w.t() + bias)
    ops.gemm(a.half().to(gpu), w.half().to(gpu), out, M=M, bias=bias.to(gpu), r1=r1.to(gpu),
             frame_alpha=alpha.to(gpu), rows_per_alpha=rows_per_frame, r1_blend=False)
    torch.cuda.synchronize()
    assert rel_l2(out, ref) < TOL_F32


@pytest.mark.parametrize("M,C", [(200, 64), (1024, 320), (130, 640)])
def test_gemm_geglu(gpu, M, C):
    from gcd_amd import ops, packing
    g = _gen(4)
    inner = 4 * C
    a = _h(torch.randn(M, C, generator=g))
    w = _h(torch.randn(2 * inner, C, generator=g) / math.sqrt(C))
    b = torch.randn(2 * inner, generator=g)
    h = a @ w.t() + b
    ref = h[:, :inner] * F.gelu(h[:, inner:])
    wp, bp = packing.pack_geglu(w, b)
    out = torch.empty(M, inner, device=gpu, dtype=torch.float16)
    ops.gemm(a.half().to(gpu), wp.to(gpu), out, M=M, bias=bp.to(gpu), out_kind=ops.OUT_GEGLU)
    torch.cuda.synchronize()
    e = rel_l2(out.float(), ref)
    assert e < TOL_F16, f"geglu rel-L2 {e:.3e}"


@pytest.mark.parametrize("M,form", [(128 * 256 + 77, "plain"), (4096, "blend32"), (9216 * 2, "blend16"), (1, "plain")])
def test_ff_fused_one_kernel(gpu, M, form):
    """FeedForward(GEGLU) of the C = 320 level as ONE kernel (gcd_ff_fused_f16, ff_fused_kernel.h): LayerNorm'd tokens in,
    residual stream out, the hidden tensor never in memory — against the fp32 torch FeedForward (attention.py:87-121) with
    the residual / AlphaBlender forms of video_attention.py:109-140, util.py:364-368, and against the library's two-GEMM
    path (same fp16 rounding point of the hidden tensor: agreement to fp32 summation order).  Sizes: more than one round of
    128-token tiles with a ragged last tile, two frames of alphas, one token."""
    from gcd_amd import ops, packing
    g = _gen(611)
    C, H = 320, 1280
    x = _h(torch.randn(M, C, generator=g))
    w1 = _h(torch.randn(2 * H, C, generator=g) / math.sqrt(C))
    b1 = torch.randn(2 * H, generator=g) * 0.5
    w2 = _h(torch.randn(C, H, generator=g) / math.sqrt(H))
    b2 = torch.randn(C, generator=g)
    r1 = torch.randn(M, C, generator=g)
    r2 = torch.randn(M, C, generator=g)
    rows_per_alpha = 9216 if form == "blend16" else 2048
    alpha = torch.rand((M + rows_per_alpha - 1) // rows_per_alpha, generator=g)
    h = x @ w1.t() + b1
    hid = _h(h[:, :H] * F.gelu(h[:, H:]))          # the hidden tensor is an fp16 MFMA operand in both HIP paths
    ff = hid @ w2.t() + b2
    if form == "plain":
        ref = ff + r1
    else:
        a = alpha.repeat_interleave(rows_per_alpha)[:M, None]
        ref = (1 - a) * (ff + r1) + a * r2
    w1p, b1p = packing.pack_geglu(w1.to(gpu), b1.to(gpu))
    w2p = w2.half().to(gpu)
    wp = ops.ff_pack(w1p, w2p)
    xg, r1g, r2g, b2g, ag = x.half().to(gpu), r1.to(gpu), r2.to(gpu), b2.to(gpu), alpha.to(gpu)
    out = torch.empty(M, C, device=gpu, dtype=torch.float16 if form == "blend16" else torch.float32)
    two = torch.empty_like(out)
    hid16 = torch.empty(M, H, device=gpu, dtype=torch.float16)
    ops.gemm(xg, w1p, hid16, M=M, bias=b1p, out_kind=ops.OUT_GEGLU)
    if form == "plain":
        ops.ff_fused(xg, wp, b1p, b2g, out, M=M, r1=r1g)
        ops.gemm(hid16, w2p, two, M=M, bias=b2g, r1=r1g)
    else:
        kind = ops.OUT_F16 if form == "blend16" else ops.OUT_F32
        ops.ff_fused(xg, wp, b1p, b2g, out, M=M, r1=r1g, r2=r2g, out_kind=kind, frame_alpha=ag, rows_per_alpha=rows_per_alpha)
        ops.gemm(hid16, w2p, two, M=M, bias=b2g, r1=r1g, r2=r2g, out_kind=kind, frame_alpha=ag,
                 rows_per_alpha=rows_per_alpha, r1_blend=True)
    torch.cuda.synchronize()
    e = rel_l2(out.float(), ref)
    e2 = rel_l2(out.float(), two.float())
    print(f"fused FeedForward M = {M} {form}: vs fp32 torch {e:.2e}, vs the two-GEMM path {e2:.2e}")
    assert e < (TOL_F16 if form == "blend16" else 3e-4), f"rel-L2 {e:.3e}"
    assert e2 < (6e-4 if form == "blend16" else 5e-5), f"vs two-GEMM path {e2:.3e}"
    # in place (out aliases the residual it initialises the accumulators with), as the engine calls it
    if form == "plain":
        inpl = r1g.clone()
        ops.ff_fused(xg, wp, b1p, b2g, inpl, M=M, r1=inpl)
        torch.cuda.synchronize()
        assert torch.equal(inpl, out)


@pytest.mark.parametrize("M,ld", [(256 * 70 + 77, 320), (9216 * 3, 320), (5, 320), (1000, 384)])
def test_layernorm_qkv_one_kernel(gpu, M, ld):
    """q | k | v = to_qkv(norm1(x)) (attention.py:519-521 with attention.py:300-316; video_attention.py:90-93) as ONE launch
    from the fp32 residual stream at model width 320 (lnqkv.hip): against fp32 torch with the LayerNorm output rounded to
    fp16 where the HIP paths round it, and against the two launches it replaces (gcd_layernorm_f16 + gcd_gemm_f16 with an
    fp16 output) — same rounding points, so the two HIP results differ by fp32 summation order and a handful of fp16
    roundings that fall the other way; ragged token counts, a padded row stride, both walk directions bit-identical."""
    from gcd_amd import ops
    g = _gen(77)
    C, N = 320, 960
    x = torch.randn(M, C, generator=g) * 1.3 + 0.4
    w = _h(torch.randn(N, C, generator=g) / math.sqrt(C))
    gamma, beta = torch.randn(C, generator=g), torch.randn(C, generator=g) * 0.5
    ref = _h(F.layer_norm(x, (C,), gamma, beta, 1e-5)) @ w.t()
    xg = torch.zeros(M, ld, device=gpu)
    xg[:, :C] = x.to(gpu)
    xv = xg[:, :C]
    w16 = w.half().to(gpu)
    wp = ops.lnqkv_pack(w16)
    gg, bg = gamma.to(gpu), beta.to(gpu)
    out = torch.full((M, N), float("nan"), device=gpu, dtype=torch.float16)
    ops.lnqkv(xv, gg, bg, wp, out, M=M, N=N)
    torch.cuda.synchronize()
    e = rel_l2(out.float(), ref)
    assert e < TOL_F16, f"LayerNorm + q|k|v in one launch, M = {M}: rel-L2 {e:.3e}"
    x16 = torch.empty(M, C, device=gpu, dtype=torch.float16)
    ops.layernorm(xv, gg, bg, x16)
    two = torch.empty(M, N, device=gpu, dtype=torch.float16)
    ops.gemm(x16, w16, two, M=M, out_kind=ops.OUT_F16)
    torch.cuda.synchronize()
    e2 = rel_l2(out.float(), two.float())
    print(f"LayerNorm + q|k|v in one launch, M = {M}: vs fp32 torch {e:.2e}, vs LayerNorm kernel + GEMM {e2:.2e}")
    assert e2 < 3e-4
    rev = torch.empty_like(out)
    ops.lnqkv(xv, gg, bg, wp, rev, M=M, N=N, sched=1)
    torch.cuda.synchronize()
    assert torch.equal(rev, out)


@pytest.mark.parametrize("M,form", [(128 * 256 + 77, "plain"), (9216 * 3, "pos"), (9216 * 2, "blend16"), (5, "plain")])
def test_ff_fused_with_its_layernorm(gpu, M, form):
    """The form the engine uses: x = ff(norm(x + pos)) + (x + pos) [AlphaBlender with a second stream] as ONE launch from the
    fp32 residual stream (gcd_ff_desc.ln_gamma != NULL) — the nn.LayerNorm in front of the FeedForward (attention.py:519-521,
    566-572; video_attention.py:90-93, 109-140, 283-284) is computed by the kernel, whose accumulators start from the very
    values it normalises.  Against fp32 torch (LayerNorm output and hidden tensor rounded to fp16 where the HIP paths round
    them) and against LayerNorm kernel + two GEMMs; in place, as the engine calls it."""
    from gcd_amd import ops, packing
    g = _gen(612)
    C, H = 320, 1280
    x = torch.randn(M, C, generator=g) * 1.5 + 0.3
    w1 = _h(torch.randn(2 * H, C, generator=g) / math.sqrt(C))
    b1 = torch.randn(2 * H, generator=g) * 0.5
    w2 = _h(torch.randn(C, H, generator=g) / math.sqrt(H))
    b2 = torch.randn(C, generator=g)
    gamma, beta = torch.randn(C, generator=g), torch.randn(C, generator=g) * 0.5
    r2 = torch.randn(M, C, generator=g)
    rpv = 9216
    nvec = (M + rpv - 1) // rpv
    pos = torch.randn(nvec, C, generator=g) * 0.7
    alpha = torch.rand(nvec, generator=g)
    z = x + pos.repeat_interleave(rpv, 0)[:M] if form != "plain" else x
    xn = _h(F.layer_norm(z, (C,), gamma, beta, 1e-5))
    h = xn @ w1.t() + b1
    hid = _h(h[:, :H] * F.gelu(h[:, H:]))
    ff = hid @ w2.t() + b2
    if form == "blend16":
        a = alpha.repeat_interleave(rpv)[:M, None]
        ref = (1 - a) * (ff + z) + a * r2
    else:
        ref = ff + z
    w1p, b1p = packing.pack_geglu(w1.to(gpu), b1.to(gpu))
    w2p = w2.half().to(gpu)
    wp = ops.ff_pack(w1p, w2p, for_ln=True)
    xg, r2g, b2g, gg, bg = x.to(gpu), r2.to(gpu), b2.to(gpu), gamma.to(gpu), beta.to(gpu)
    ln = dict(gamma=gg, beta=bg)
    if form != "plain":
        ln.update(addvec=pos.to(gpu), rows_per_vec=rpv)
    kind = ops.OUT_F16 if form == "blend16" else ops.OUT_F32
    out = torch.empty(M, C, device=gpu, dtype=torch.float16 if form == "blend16" else torch.float32)
    kw = dict(r2=r2g, out_kind=kind, frame_alpha=alpha.to(gpu), rows_per_alpha=rpv) if form == "blend16" else {}
    ops.ff_fused(xg, wp, b1p, b2g, out, M=M, ln=ln, **kw)
    # LayerNorm kernel + two GEMMs
    x16 = torch.empty(M, C, device=gpu, dtype=torch.float16)
    zg = torch.empty(M, C, device=gpu) if form != "plain" else xg
    ops.layernorm(xg, gg, bg, x16, addvec=ln.get("addvec"), rows_per_vec=rpv, sum_out=zg if form != "plain" else None)
    hid16 = torch.empty(M, H, device=gpu, dtype=torch.float16)
    two = torch.empty_like(out)
    ops.gemm(x16, w1p, hid16, M=M, bias=b1p, out_kind=ops.OUT_GEGLU)
    if form == "blend16":
        ops.gemm(hid16, w2p, two, M=M, bias=b2g, r1=zg, r2=r2g, out_kind=kind, frame_alpha=alpha.to(gpu), rows_per_alpha=rpv,
                 r1_blend=True)
    else:
        ops.gemm(hid16, w2p, two, M=M, bias=b2g, r1=zg)
    torch.cuda.synchronize()
    e, e2 = rel_l2(out.float(), ref), rel_l2(out.float(), two.float())
    print(f"LayerNorm + FeedForward in one launch, M = {M} {form}: vs fp32 torch {e:.2e}, vs LayerNorm kernel + two GEMMs {e2:.2e}")
    # (a last-bit difference of the fp32 statistics flips an fp16 rounding of the normalised operand here and there: 1e-4)
    assert e < (TOL_F16 if form == "blend16" else 3e-4), f"rel-L2 {e:.3e}"
    assert e2 < (6e-4 if form == "blend16" else 2e-4), f"vs the three-launch path {e2:.3e}"
    if form == "plain":      # in place: out aliases the stream it reads
        inpl = xg.clone()
        ops.ff_fused(inpl, wp, b1p, b2g, inpl, M=M, ln=ln)
        torch.cuda.synchronize()
        assert torch.equal(inpl, out)
    with pytest.raises(AssertionError):                    # the LayerNorm form has no separate residual operand
        ops.ff_fused(xg, wp, b1p, b2g, out, M=M, ln=ln, r1=xg, **kw)


def test_ff_fused_refuses_what_it_cannot_do(gpu):
    from gcd_amd import ops, _lib
    x = torch.zeros(256, 320, device=gpu, dtype=torch.float16)
    wp = torch.zeros(int(_lib.load().gcd_ff_packed_bytes()) // 2, device=gpu, dtype=torch.float16)
    b1, b2 = torch.zeros(2560, device=gpu), torch.zeros(320, device=gpu)
    out = torch.zeros(256, 320, device=gpu)
    with pytest.raises(Exception, match="rows_per_alpha"):
        ops.ff_fused(x, wp, b1, b2, out, M=256, r1=out, r2=out, frame_alpha=torch.zeros(26, device=gpu), rows_per_alpha=10)
    assert not ops.ff_fused_ok(10 ** 6, 640, 2560, enabled=True) and ops.ff_fused_ok(10 ** 6, 320, 1280, enabled=True)
    assert not ops.ff_fused_ok(1000, 320, 1280, enabled=True)       # too few tiles to pay


def test_feedforward_tile_blocked_hidden(gpu, gemm_impl):
    """GEGLU output written tile-blocked ([M/256][N/320][256][160]) and consumed that way by the second
    Linear: bit-identical to the row-major pair (same products, same accumulation order), the block
    layout itself checked against the row-major hidden tensor; unsupported shapes are refused."""
    from gcd_amd import ops, packing
    g = _gen(41)
    M, C = 256 * 50, 320                         # 50 x 8 GEGLU tiles, 50 FF-out tiles
    H = 4 * C
    x = _h(torch.randn(M, C, generator=g)).half().to(gpu)
    w1, b1 = packing.pack_geglu((torch.randn(2 * H, C, generator=g) / math.sqrt(C)).to(gpu),
                                torch.randn(2 * H, generator=g).to(gpu))
    w2 = (torch.randn(C, H, generator=g) / math.sqrt(H)).half().to(gpu)
    b2 = torch.randn(C, generator=g).to(gpu)
    r1 = torch.randn(M, C, generator=g).to(gpu)
    ok = ops.gemm_hidden_blocked_ok(M, 2 * H, C, enabled=True)
    # Round 5: the blocked hidden layout (measured neutral) is compiled into the ABLATION build only
    # (tools/libgcd_amd_ablate.so through GCD_AMD_LIB): the product library answers "no" and refuses the descriptors.
    import os
    ablation = "ablate" in os.environ.get("GCD_AMD_LIB", "")
    assert ok == ({0: False, 2: True, 3: True, 6: False}[gemm_impl] if ablation else False)   # 50 FF-out tiles < 192: automatic says no
    hid = torch.empty(M, H, dtype=torch.float16, device=gpu)
    out = torch.empty(M, C, device=gpu)
    ops.gemm(x, w1, hid, M=M, bias=b1, out_kind=ops.OUT_GEGLU)
    ops.gemm(hid, w2, out, M=M, bias=b2, r1=r1)
    if not ok:
        with pytest.raises(Exception, match="out_blocked|a_blocked"):
            if gemm_impl == 6 or not ablation:
                ops.gemm(x, w1, hid, M=M, bias=b1, out_kind=ops.OUT_GEGLU, out_blocked=True)
            else:                                 # automatic: the GEGLU launch is large enough, FF-out is not
                ops.gemm(hid, w2, out, M=M, bias=b2, r1=r1, a_blocked=True)
        return
    hid_b = torch.empty(M * H, dtype=torch.float16, device=gpu)
    out_b = torch.empty(M, C, device=gpu)
    ops.gemm(x, w1, hid_b.view(M, H), M=M, bias=b1, out_kind=ops.OUT_GEGLU, out_blocked=True)
    ops.gemm(hid_b.view(M, H), w2, out_b, M=M, bias=b2, r1=r1, a_blocked=True)
    torch.cuda.synchronize()
    want = hid.reshape(M // 256, 256, H // 160, 160).permute(0, 2, 1, 3).reshape(-1)
    assert torch.equal(hid_b, want), "blocked hidden layout differs from [M/256][H/160][256][160]"
    assert torch.equal(out_b, out), "tile-blocked FeedForward differs from the row-major one"
    with pytest.raises(Exception, match="out_blocked"):
        ops.gemm(x[:300], w1, hid[:300], M=300, bias=b1, out_kind=ops.OUT_GEGLU, out_blocked=True)


@pytest.mark.parametrize("frames,H,W,Cin,Cout,stride,up", [
    (3, 10, 12, 64, 64, 1, 0), (2, 9, 7, 128, 320, 1, 0), (2, 12, 16, 64, 128, 2, 0),
    (2, 9, 7, 64, 64, 2, 0), (2, 5, 6, 128, 64, 1, 1), (28, 16, 16, 64, 64, 1, 0)])
def test_conv3x3(gpu, frames, H, W, Cin, Cout, stride, up):
    from gcd_amd import ops, packing
    g = _gen(5)
    x = _h(torch.randn(frames, Cin, H, W, generator=g))
    w = _h(torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin))
    b = torch.randn(Cout, generator=g)
    emb = torch.randn(frames, Cout, generator=g)
    xin = F.interpolate(x, scale_factor=2, mode="nearest") if up else x
    ref = F.conv2d(xin, w, b, stride=stride, padding=1) + emb[:, :, None, None]
    Ho, Wo = ref.shape[-2:]
    ref_tok = ref.permute(0, 2, 3, 1).reshape(frames * Ho * Wo, Cout)
    a = x.permute(0, 2, 3, 1).reshape(frames * H * W, Cin).half().to(gpu)
    out = torch.empty(frames * Ho * Wo, Cout, device=gpu)
    ops.gemm(a, packing.pack_conv3x3(w).to(gpu), out, M=frames * Ho * Wo, mode=ops.GEMM_CONV3X3,
             bias=b.to(gpu), rowvec=emb.to(gpu), rows_per_vec=Ho * Wo,
             conv=dict(Cin=Cin, Hi=H, Wi=W, Ho=Ho, Wo=Wo, stride=stride, upsample=up))
    torch.cuda.synchronize()
    e = rel_l2(out, ref_tok)
    assert e < TOL_F32, f"conv3x3 rel-L2 {e:.3e}"


@pytest.mark.parametrize("frames,H,W,Cin,Cout,res", [
    (2, 4, 64, 64, 320, 0),        # four 64-pixel row segments per 256-token tile
    (3, 2, 128, 128, 320, 2),      # two 128-pixel segments; tiles start on odd / even image rows; both residuals
    (2, 3, 128, 192, 640, 1),      # three chunks per tap; 384 tokens per frame: a frame boundary falls INSIDE a tile
    (1, 2, 256, 64, 320, 1),       # one 256-pixel segment per tile
    (1, 1, 512, 64, 320, 0)])      # segments of a longer image row: the right halo is a real neighbour pixel
def test_conv3x3_halo_panel(gpu, gemm_impl, frames, H, W, Cin, Cout, res):
    """The halo-panel K loop of gemm_p8.hip (stride-1 3x3 convolutions whose image rows are 64-token aligned: K order
    (kh, cin-chunk, kw), one staged A panel with halo pixels serving the three kw taps): borders (zeros from the
    out-of-range buffer offsets), frame boundaries inside a tile, residual epilogues, and the per-64-row column sums."""
    from gcd_amd import ops, packing
    g = _gen(55 + W)
    x = _h(torch.randn(frames, Cin, H, W, generator=g))
    w = _h(torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin))
    b = torch.randn(Cout, generator=g)
    M = frames * H * W
    assert M % 256 == 0
    ref = F.conv2d(x, w, b, padding=1).permute(0, 2, 3, 1).reshape(M, Cout)
    r1 = torch.randn(M, Cout, generator=g) if res >= 1 else None
    r2 = torch.randn(M, Cout, generator=g) if res >= 2 else None
    if r1 is not None:
        ref = ref + r1
    if r2 is not None:
        ref = ref + 0.5 * r2
    a = x.permute(0, 2, 3, 1).reshape(M, Cin).contiguous().half().to(gpu)
    out = torch.empty(M, Cout, device=gpu)
    kw = dict(M=M, mode=ops.GEMM_CONV3X3, bias=b.to(gpu),
              conv=dict(Cin=Cin, Hi=H, Wi=W, Ho=H, Wo=W, stride=1, upsample=0))
    if r1 is not None:
        kw.update(r1=r1.to(gpu))
    if r2 is not None:
        kw.update(r2=r2.to(gpu), s_r2=0.5)
    ops.gemm(a, packing.pack_conv3x3(w).to(gpu), out, **kw)
    torch.cuda.synchronize()
    e = rel_l2(out, ref)
    assert e < TOL_F32, f"halo conv3x3 {frames}x{H}x{W} Cin {Cin}: rel-L2 {e:.3e}"
    # the same launch with GroupNorm column statistics (own kernel instantiation), where the shape allows them:
    # rows 2k / 2k+1 of the buffer = sums / sums of squares over the 64 output rows of block k (_colsum_ref)
    wp = packing.pack_conv3x3(w).to(gpu)
    out2 = torch.empty(M, Cout, device=gpu)
    if res <= 1 and ops.gemm(a, wp, out2, probe_colstats=True, **kw):
        cs = torch.full((2 * (M // 64), Cout), float("nan"), device=gpu)
        ops.gemm(a, wp, out2, colstats=cs, **kw)
        torch.cuda.synchronize()
        assert rel_l2(out2, ref) < TOL_F32
        want = _colsum_ref(out2.cpu())
        assert not torch.isnan(cs).any()
        assert rel_l2(cs[0::2], want[0::2]) < 2e-6 and rel_l2(cs[1::2], want[1::2]) < 2e-6


def test_conv3x3_padded_channels(gpu):
    """First conv (8 -> C, input channels zero-padded to 64) and last conv (C -> 4, N padded to 16)."""
    from gcd_amd import ops, packing
    g = _gen(6)
    frames, H, W = 2, 8, 8
    x = _h(torch.randn(frames, 8, H, W, generator=g))
    w = _h(torch.randn(64, 8, 3, 3, generator=g) / math.sqrt(72))
    ref = F.conv2d(x, w, None, padding=1).permute(0, 2, 3, 1).reshape(-1, 64)
    a = torch.zeros(frames * H * W, 64)
    a[:, :8] = x.permute(0, 2, 3, 1).reshape(-1, 8)
    out = torch.empty(frames * H * W, 64, device=gpu)
    ops.gemm(a.half().to(gpu), packing.pack_conv3x3(w, cin_pad=64).to(gpu), out, M=frames * H * W,
             mode=ops.GEMM_CONV3X3, conv=dict(Cin=64, Hi=H, Wi=W, Ho=H, Wo=W, stride=1, upsample=0))
    torch.cuda.synchronize()
    assert rel_l2(out, ref) < TOL_F32
    x2 = _h(torch.randn(frames, 64, H, W, generator=g))
    w2 = _h(torch.randn(4, 64, 3, 3, generator=g) / math.sqrt(576))
    b2 = torch.randn(4, generator=g)
    ref2 = F.conv2d(x2, w2, b2, padding=1)
    bp = torch.zeros(16)
    bp[:4] = b2
    out2 = torch.empty(frames * H * W, 16, device=gpu)
    ops.gemm(x2.permute(0, 2, 3, 1).reshape(-1, 64).half().to(gpu),
             packing.pack_conv3x3(w2, cout_pad=16).to(gpu), out2, M=frames * H * W,
             mode=ops.GEMM_CONV3X3, bias=bp.to(gpu),
             conv=dict(Cin=64, Hi=H, Wi=W, Ho=H, Wo=W, stride=1, upsample=0))
    nchw = torch.empty(frames, 4, H, W, device=gpu)
    ops.unpack_output(out2, nchw, 4, frames, H * W)
    torch.cuda.synchronize()
    assert rel_l2(nchw, ref2) < TOL_F32


@pytest.mark.parametrize("clips,T,HW,C", [(2, 5, 12, 64), (2, 14, 64, 128), (1, 3, 200, 64)])
def test_conv_temporal3(gpu, clips, T, HW, C):
    from gcd_amd import ops, packing
    g = _gen(7)
    x = _h(torch.randn(clips, C, T, HW, 1, generator=g))          # b c t h w
    w = _h(torch.randn(C, C, 3, 1, 1, generator=g) / math.sqrt(3 * C))
    b = torch.randn(C, generator=g)
    ref = F.conv3d(x, w, b, padding=(1, 0, 0))                     # b c t hw 1
    ref_tok = ref[..., 0].permute(0, 2, 3, 1).reshape(clips * T * HW, C)
    a = x[..., 0].permute(0, 2, 3, 1).reshape(clips * T * HW, C).contiguous().half().to(gpu)
    out = torch.empty(clips * T * HW, C, device=gpu)
    ops.gemm(a, packing.pack_conv_t3(w).to(gpu), out, M=clips * T * HW, mode=ops.GEMM_TEMPORAL3,
             bias=b.to(gpu), conv=dict(Cin=C, T=T, HW=HW))
    torch.cuda.synchronize()
    e = rel_l2(out, ref_tok)
    assert e < TOL_F32, f"temporal conv rel-L2 {e:.3e}"


# ----------------------------------------------------------------------------------------- norms
@pytest.mark.parametrize("frames,HW,C1,C2,per_clip_T", [
    (4, 100, 320, 0, 0), (4, 37, 64, 0, 0), (3, 64, 640, 320, 0), (4, 50, 128, 0, 2),
    (28, 256, 320, 0, 14), (2, 300, 1280, 640, 0)])
def test_groupnorm(gpu, frames, HW, C1, C2, per_clip_T):
    from gcd_amd import ops
    g = _gen(8)
    C = C1 + C2
    x = torch.randn(frames, C, HW, generator=g) * 3 + 1.5          # n c hw, non-zero mean
    gamma = torch.randn(C, generator=g)
    beta = torch.randn(C, generator=g)
    eps = 1e-5
    if per_clip_T:  # time_stack GroupNorm: stats over (T, HW) per clip
        xr = x.reshape(frames // per_clip_T, per_clip_T, C, HW).permute(0, 2, 1, 3)
        ref = F.group_norm(xr, 32, gamma, beta, eps).permute(0, 2, 1, 3).reshape(frames, C, HW)
        rows = per_clip_T * HW
    else:
        ref = F.group_norm(x, 32, gamma, beta, eps)
        rows = HW
    ref = F.silu(ref).permute(0, 2, 1).reshape(frames * HW, C)
    tok = x.permute(0, 2, 1).reshape(frames * HW, C).contiguous()
    x1 = tok[:, :C1].contiguous().to(gpu)
    x2 = tok[:, C1:].contiguous().to(gpu) if C2 else None
    nch = ops.gn_nchunks(rows)
    ninst = frames * HW // rows
    partial = torch.empty(ninst * nch * 64, dtype=torch.float64, device=gpu)
    stats = torch.empty(ninst * 64, device=gpu)
    ops.groupnorm_stats(x1, x2, rows, eps, partial, stats, nch)
    y = torch.empty(frames * HW, C, dtype=torch.float16, device=gpu)
    raw = torch.empty(frames * HW, C, dtype=torch.float16, device=gpu)
    ops.groupnorm_apply(x1, x2, rows, stats, gamma.to(gpu), beta.to(gpu), True, y, raw)
    torch.cuda.synchronize()
    e = rel_l2(y.float(), ref)
    assert e < TOL_F16, f"groupnorm rel-L2 {e:.3e}"
    assert rel_l2(raw.float(), tok) < TOL_F16


@pytest.mark.parametrize("M,N,K", [(300, 320, 320), (1000, 64, 640), (4032, 1280, 1280), (77, 48, 64)])
def test_gemm_bf16_operands(gpu, M, N, K):
    """gcd_gemm_desc.operand_bf16: bfloat16 A and W on the general kernel (v_mfma_f32_16x16x32_bf16), fp32
    output with bias and residual — against fp32 math on the same bf16-rounded operands; the cast kernels
    against torch's round-to-nearest-even; unsupported combinations are refused."""
    from gcd_amd import ops
    g = _gen(51)
    a32, w32 = torch.randn(M, K, generator=g) * 3, torch.randn(N, K, generator=g) / math.sqrt(K)
    bias, r1 = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    a = torch.empty(M, K, dtype=torch.bfloat16, device=gpu)
    ops.cast_bf16(a32.to(gpu), a)
    assert torch.equal(a.cpu(), a32.to(torch.bfloat16)), "gcd_cast_f32_bf16 is not round-to-nearest-even"
    w = torch.empty(N, K, dtype=torch.bfloat16, device=gpu)
    ops.cast_bf16(w32.half().to(gpu), w)                     # the fp16 -> bf16 form
    assert torch.equal(w.cpu(), w32.half().to(torch.bfloat16))
    ref = a.cpu().float() @ w.cpu().float().t() + bias + r1
    out = torch.empty(M, N, device=gpu)
    ops.gemm(a, w, out, M=M, bias=bias.to(gpu), r1=r1.to(gpu), operand_bf16=True)
    torch.cuda.synchronize()
    e = rel_l2(out, ref)
    assert e < TOL_F32, f"bf16 GEMM M={M} N={N} K={K}: rel-L2 {e:.3e}"
    with pytest.raises(Exception, match="bf16"):
        ops.gemm(a, w, torch.empty(M, N, device=gpu, dtype=torch.float16), M=M, out_kind=ops.OUT_F16,
                 operand_bf16=True)
    with pytest.raises(AssertionError):
        ops.gemm(a.to(torch.float16), w, out, M=M, operand_bf16=True)


def _colsum_ref(out, rows=64):
    b = out.double().reshape(out.shape[0] // rows, rows, out.shape[1])
    return torch.stack([b.sum(1), (b * b).sum(1)], 1).reshape(-1, out.shape[1])   # [2 * blocks, N]


@pytest.mark.parametrize("case", ["plain_r1_persistent", "plain_small", "conv_rowvec", "temporal_alpha"])
def test_gemm_column_sums_for_groupnorm(gpu, gemm_impl, case):
    """gcd_gemm_desc.colstats: per-64-row column sums / sums of squares of the fp32 output written by
    the epilogue that produces it (ping-pong kernel, full tiles), for the three GEMM modes and the
    epilogue inputs the UNet's GroupNorm producers use; and the statistics derived from them."""
    from gcd_amd import ops, packing
    g = _gen(31)
    kw, ref = {}, None
    if case in ("plain_r1_persistent", "plain_small"):
        M, N, K = (256 * 70, 1280, 128) if case == "plain_r1_persistent" else (256 * 3, 320, 64)
        a = _h(torch.randn(M, K, generator=g))
        w = _h(torch.randn(N, K, generator=g) / math.sqrt(K))
        bias = torch.randn(N, generator=g)
        r1 = torch.randn(M, N, generator=g) + 0.5
        ref = a @ w.t() + bias + r1
        A, Wp = a.half().to(gpu), w.half().to(gpu)
        kw = dict(M=M, bias=bias.to(gpu), r1=r1.to(gpu))
    elif case == "conv_rowvec":
        frames, H, W, Cin, N = 3, 16, 32, 64, 320            # HW = 512: rowvec constant over every tile
        M = frames * H * W
        x = _h(torch.randn(frames, Cin, H, W, generator=g))
        wt = _h(torch.randn(N, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin))
        bias = torch.randn(N, generator=g)
        rv = torch.randn(frames, N, generator=g)
        ref = (F.conv2d(x, wt, bias, padding=1) + rv[:, :, None, None]).permute(0, 2, 3, 1).reshape(M, N)
        A = x.permute(0, 2, 3, 1).reshape(M, Cin).contiguous().half().to(gpu)
        Wp = packing.pack_conv3x3(wt.to(gpu))
        kw = dict(M=M, mode=ops.GEMM_CONV3X3, bias=bias.to(gpu), rowvec=rv.to(gpu), rows_per_vec=H * W,
                  conv=dict(Cin=Cin, Hi=H, Wi=W, Ho=H, Wo=W, stride=1, upsample=0))
    else:
        clips, T, HW, C = 1, 3, 256, 320
        M = clips * T * HW
        x = _h(torch.randn(clips, C, T, HW, generator=g))
        wt = _h(torch.randn(C, C, 3, generator=g) / math.sqrt(3 * C))
        bias = torch.randn(C, generator=g)
        r1 = torch.randn(M, C, generator=g)
        alpha = torch.rand(clips * T, generator=g)
        conv = F.conv1d(x.permute(0, 3, 1, 2).reshape(clips * HW, C, T), wt, bias, padding=1)
        conv = conv.reshape(clips, HW, C, T).permute(0, 3, 1, 2).reshape(M, C)
        al = alpha.repeat_interleave(HW)[:, None]
        ref = (1 - al) * conv + r1                            # r1_blend False: x_s + (1 - alpha) * (conv + b)
        A = x.permute(0, 2, 3, 1).reshape(M, C).contiguous().half().to(gpu)
        Wp = packing.pack_conv_t3(wt.reshape(C, C, 3, 1, 1).to(gpu))
        N = C
        kw = dict(M=M, mode=ops.GEMM_TEMPORAL3, bias=bias.to(gpu), r1=r1.to(gpu),
                  frame_alpha=alpha.to(gpu), rows_per_alpha=HW, r1_blend=False, conv=dict(Cin=C, T=T, HW=HW))
    out = torch.empty(M, N, device=gpu)
    ok = ops.gemm(A, Wp, out, probe_colstats=True, **kw)
    # automatic choice: only grids of >= 192 tiles go to the ping-pong kernel; tile64 forces the other one
    expect = {0: case == "plain_r1_persistent", 2: True, 3: True, 6: False}[gemm_impl]
    assert ok == expect, f"colstats support: got {ok}, expected {expect}"
    cs = torch.full((2 * (M // 64), N), float("nan"), device=gpu)
    if not ok:
        with pytest.raises(Exception, match="colstats"):
            ops.gemm(A, Wp, out, colstats=cs, **kw)
        return
    ops.gemm(A, Wp, out, colstats=cs, **kw)
    torch.cuda.synchronize()
    assert rel_l2(out, ref) < TOL_F32
    want = _colsum_ref(out.cpu())                              # sums of what was actually stored
    assert not torch.isnan(cs).any()
    assert rel_l2(cs[0::2], want[0::2]) < 2e-6 and rel_l2(cs[1::2], want[1::2]) < 2e-6
    # statistics from the sums == statistics from a pass over the tensor (32 groups; one instance per
    # 256 rows, and the whole tensor as one instance)
    for rows in (256, M):
        ninst = M // rows
        st_a = torch.empty(ninst * 64, device=gpu)
        st_b = torch.empty(ninst * 64, device=gpu)
        ops.groupnorm_stats_from_colsums(cs, N, None, 0, M, rows, 1e-5, st_a)
        nch = ops.gn_nchunks(rows, ninst)
        partial = torch.empty(ninst * nch * 64, dtype=torch.float64, device=gpu)
        ops.groupnorm_stats(out, None, rows, 1e-5, partial, st_b, nch)
        torch.cuda.synchronize()
        assert torch.allclose(st_a, st_b, rtol=2e-5, atol=2e-6), f"rows={rows}: {(st_a - st_b).abs().max()}"


def test_groupnorm_stats_from_colsums_virtual_concat(gpu):
    """[x1 | x2] with separate column sums (decoder skip concat), group boundaries inside and across the
    two sources, vs torch GroupNorm statistics; bad geometry is refused."""
    from gcd_amd import ops
    g = _gen(32)
    frames, HW, C1, C2 = 3, 128, 640, 320
    M = frames * HW
    x = torch.randn(M, C1 + C2, generator=g) * 2 + torch.randn(C1 + C2, generator=g)
    cs1 = _colsum_ref(x[:, :C1]).float().to(gpu)
    cs2 = _colsum_ref(x[:, C1:]).float().to(gpu)
    stats = torch.empty(frames * 64, device=gpu)
    ops.groupnorm_stats_from_colsums(cs1, C1, cs2, C2, M, HW, 1e-6, stats)
    torch.cuda.synchronize()
    xg = x.double().reshape(frames, HW, 32, (C1 + C2) // 32)
    mean = xg.mean((1, 3))
    rstd = 1.0 / torch.sqrt(xg.var((1, 3), unbiased=False) + 1e-6)
    got = stats.cpu().reshape(frames, 32, 2).double()
    assert torch.allclose(got[..., 0], mean, rtol=1e-5, atol=1e-6)
    assert torch.allclose(got[..., 1], rstd, rtol=1e-5, atol=1e-6)
    with pytest.raises(Exception, match="64-row"):
        ops.groupnorm_stats_from_colsums(cs1, C1, cs2, C2, M, 96, 1e-6, torch.empty(4 * 64, device=gpu))


@pytest.mark.parametrize("M,C", [(100, 64), (1000, 320), (77, 640), (64, 1280)])
def test_layernorm(gpu, M, C):
    from gcd_amd import ops
    g = _gen(9)
    x = torch.randn(M, C, generator=g) * 2 + 0.5
    gamma, beta = torch.randn(C, generator=g), torch.randn(C, generator=g)
    rows_per_vec = 10
    add = torch.randn((M + 9) // 10, C, generator=g)
    xs = x + add.repeat_interleave(rows_per_vec, 0)[:M]
    ref = F.layer_norm(xs, (C,), gamma, beta, 1e-5)
    y = torch.empty(M, C, dtype=torch.float16, device=gpu)
    s = torch.empty(M, C, device=gpu)
    ops.layernorm(x.to(gpu), gamma.to(gpu), beta.to(gpu), y, addvec=add.to(gpu),
                  rows_per_vec=rows_per_vec, sum_out=s)
    torch.cuda.synchronize()
    assert rel_l2(y.float(), ref) < TOL_F16
    assert rel_l2(s, xs) < 1e-6
    ops.layernorm(x.to(gpu), gamma.to(gpu), beta.to(gpu), y)
    torch.cuda.synchronize()
    assert rel_l2(y.float(), F.layer_norm(x, (C,), gamma, beta, 1e-5)) < TOL_F16


# ------------------------------------------------------------------------- walk orders (scheduling only)
@pytest.mark.parametrize("M,C", [(1000, 320), (77, 640), (70, 1280), (16 * 8192 + 37, 320)])
def test_layernorm_walk_orders_are_bit_identical(gpu, M, C):
    """gcd_layernorm_f16's `order` (back to front / the GEMM's eight-region orders) is scheduling only: every order
    writes exactly what order 0 writes — ragged row counts, fewer row blocks than regions, the grid-stride tail."""
    from gcd_amd import ops
    g = _gen(91)
    x = (torch.randn(M, C, generator=g) * 2 + 0.5).to(gpu)
    gamma, beta = torch.randn(C, generator=g).to(gpu), torch.randn(C, generator=g).to(gpu)
    add = torch.randn((M + 9) // 10, C, generator=g).to(gpu)
    outs = []
    for order in range(5):      # 4: order 0 again with non-temporal loads and stores
        y = torch.full((M, C), float("nan"), dtype=torch.float16, device=gpu)
        s = torch.full((M, C), float("nan"), device=gpu)
        ops.tune_set(3, 3 if order == 4 else 4)     # GCD_TUNE_STREAM: 4 = plain accesses, 3 = non-temporal loads + stores
        try:
            ops.layernorm(x, gamma, beta, y, addvec=add, rows_per_vec=10, sum_out=s, order=order % 4)
            torch.cuda.synchronize()
        finally:
            ops.tune_set(3, 0)
        outs.append((y.cpu(), s.cpu()))
    assert not torch.isnan(outs[0][0].float()).any() and not torch.isnan(outs[0][1]).any()
    for order in range(1, 5):
        assert torch.equal(outs[order][0], outs[0][0]) and torch.equal(outs[order][1], outs[0][1]), order


@pytest.mark.parametrize("frames,HW,C1,C2", [(4, 100, 320, 0), (3, 64, 640, 320), (28, 256, 320, 0), (1, 40, 64, 0)])
def test_groupnorm_apply_walk_orders_are_bit_identical(gpu, frames, HW, C1, C2):
    from gcd_amd import ops
    g = _gen(92)
    C = C1 + C2
    tok = torch.randn(frames * HW, C, generator=g) * 3 + 1.5
    x1 = tok[:, :C1].contiguous().to(gpu)
    x2 = tok[:, C1:].contiguous().to(gpu) if C2 else None
    gamma, beta = torch.randn(C, generator=g).to(gpu), torch.randn(C, generator=g).to(gpu)
    nch = ops.gn_nchunks(HW)
    partial = torch.empty(frames * nch * 64, dtype=torch.float64, device=gpu)
    stats = torch.empty(frames * 64, device=gpu)
    ops.groupnorm_stats(x1, x2, HW, 1e-5, partial, stats, nch)
    outs = []
    for order in range(5):      # 4: order 0 again with non-temporal loads and stores
        y = torch.full((frames * HW, C), float("nan"), dtype=torch.float16, device=gpu)
        raw = torch.full((frames * HW, C), float("nan"), dtype=torch.float16, device=gpu)
        ops.tune_set(3, 3 if order == 4 else 4)     # GCD_TUNE_STREAM: 4 = plain accesses, 3 = non-temporal loads + stores
        try:
            ops.groupnorm_apply(x1, x2, HW, stats, gamma, beta, True, y, raw, order=order % 4)
            torch.cuda.synchronize()
        finally:
            ops.tune_set(3, 0)
        outs.append((y.cpu(), raw.cpu()))
    assert not torch.isnan(outs[0][0].float()).any() and not torch.isnan(outs[0][1].float()).any()
    for order in range(1, 5):
        assert torch.equal(outs[order][0], outs[0][0]) and torch.equal(outs[order][1], outs[0][1]), order


@pytest.mark.parametrize("case", ["plain_persistent", "plain_few_tiles", "geglu", "conv_halo", "conv3x3", "temporal"])
def test_gemm_reverse_tile_walk_is_bit_identical(gpu, case):
    """gcd_gemm_desc.sched bit 0 (every XCD walks its share of the tiles from the end) changes the order in which the
    tiles are computed, nothing else: persistent and one-tile-per-workgroup grids, ragged edges, tile counts that do
    not divide by 8, the halo-panel / tap-walking / temporal A paths and the in-place residual."""
    from gcd_amd import ops, packing
    ops.tune_set(ops.TUNE_GEMM_IMPL, 2)
    try:
        g = _gen(93)

        def run(sched):
            gg = _gen(94)
            if case in ("plain_persistent", "plain_few_tiles", "geglu"):
                M, N, K = {"plain_persistent": (256 * 67 + 100, 320 * 5 + 64, 128), "plain_few_tiles": (256 * 5, 640, 320),
                           "geglu": (256 * 70, 1280, 64)}[case]
                a = torch.randn(M, K, generator=gg).half().to(gpu)
                w = (torch.randn(N, K, generator=gg) / math.sqrt(K)).half()
                bias = torch.randn(N, generator=gg)
                if case == "geglu":
                    wp, bp = packing.pack_geglu(w.float(), bias)
                    out = torch.full((M, N // 2), float("nan"), dtype=torch.float16, device=gpu)
                    ops.gemm(a, wp.to(gpu), out, M=M, bias=bp.to(gpu), out_kind=ops.OUT_GEGLU, sched=sched)
                else:
                    out = torch.randn(M, N, generator=gg).to(gpu)       # residual, updated in place
                    ops.gemm(a, w.to(gpu), out, M=M, bias=bias.to(gpu), r1=out, sched=sched)
                return out
            if case in ("conv_halo", "conv3x3"):
                frames, H, W, Cin, Cout = (9, 4, 64, 64, 320) if case == "conv_halo" else (28, 24, 32, 128, 320)
                x = torch.randn(frames, Cin, H, W, generator=gg)
                w = torch.randn(Cout, Cin, 3, 3, generator=gg) / math.sqrt(9 * Cin)
                M = frames * H * W
                a = x.permute(0, 2, 3, 1).reshape(M, Cin).contiguous().half().to(gpu)
                out = torch.full((M, Cout), float("nan"), device=gpu)
                ops.gemm(a, packing.pack_conv3x3(w).to(gpu), out, M=M, mode=ops.GEMM_CONV3X3, sched=sched,
                         bias=torch.randn(Cout, generator=gg).to(gpu),
                         conv=dict(Cin=Cin, Hi=H, Wi=W, Ho=H, Wo=W, stride=1, upsample=0))
                return out
            clips, T, HW, Cc = 2, 14, 700, 128
            M = clips * T * HW
            a = torch.randn(M, Cc, generator=gg).half().to(gpu)
            w = torch.randn(320, Cc, 3, 1, 1, generator=gg) / math.sqrt(3 * Cc)
            out = torch.full((M, 320), float("nan"), device=gpu)
            ops.gemm(a, packing.pack_conv_t3(w).to(gpu), out, M=M, mode=ops.GEMM_TEMPORAL3, sched=sched,
                     conv=dict(Cin=Cc, T=T, HW=HW))
            return out

        fwd, rev = run(0), run(1)
        torch.cuda.synchronize()
        assert not torch.isnan(fwd.float()).any()
        assert torch.equal(fwd, rev)
    finally:
        ops.tune_set(ops.TUNE_GEMM_IMPL, 0)


# ------------------------------------------------------------------------------------- attention
@pytest.fixture(params=[0, 1, 2, 3], ids=["attn-auto", "attn-q32", "attn-q64", "attn-q64p"])
def attn_impl(request, gpu):
    """The three spatial-attention kernels (32 queries per wave; 64 per wave; 64 per wave software-pipelined
    = what "auto" picks for S >= 1024) over the same cases — forced, so the pipelined kernel also sees 1-, 2-
    and 3-tile sequences and the reference-shift case."""
    from gcd_amd import ops
    ops.tune_set(ops.TUNE_ATTN_IMPL, request.param)
    yield request.param
    ops.tune_set(ops.TUNE_ATTN_IMPL, 0)


@pytest.mark.parametrize("frames,S,heads", [(2, 4, 1), (2, 16, 2), (1, 64, 2), (2, 100, 3),
                                            (2, 144, 2), (1, 200, 5), (2, 576, 2), (1, 1300, 1), (2, 2304, 1)])
def test_attention_spatial(gpu, attn_impl, frames, S, heads):
    from gcd_amd import ops
    g = _gen(10)
    C = heads * 64
    qkv = _h(torch.randn(frames * S, 3 * C, generator=g) * 1.5)
    # a spiked key forces a large running-max jump mid-sequence (online-softmax rescale path)
    if S >= 100:
        qkv[70, C:C + 64] *= 6.0
    q, k, v = [t.reshape(frames, S, heads, 64).permute(0, 2, 1, 3) for t in qkv.split(C, dim=1)]
    ref = F.scaled_dot_product_attention(q.double(), k.double(), v.double())
    ref = ref.permute(0, 2, 1, 3).reshape(frames * S, C).float()
    S_pad = (S + 63) // 64 * 64
    qg = qkv.half().to(gpu)
    vt = torch.empty(frames * heads * 64 * S_pad, dtype=torch.float16, device=gpu)
    ops.attn_transpose_v(qg, frames, S, heads, vt, S_pad)
    out = torch.empty(frames * S, C, dtype=torch.float16, device=gpu)
    ops.attn_spatial(qg, vt, S_pad, out, frames, S, heads)
    torch.cuda.synchronize()
    vt_ref = torch.zeros(frames, heads, 64, S_pad)
    vt_ref[..., :S] = v.permute(0, 1, 3, 2)
    # gcd_attn_transpose_v stores every 16-key group as keys 0-3, 8-11, 4-7, 12-15
    perm = torch.arange(S_pad).reshape(-1, 4, 4)[:, [0, 2, 1, 3]].reshape(-1)
    assert torch.equal(vt.cpu().float().reshape(frames, heads, 64, S_pad), vt_ref[..., perm]), "V^T mismatch"
    e = rel_l2(out.float(), ref)
    assert e < 1.5e-3, f"spatial attention S={S}: rel-L2 {e:.3e}"
    # q_prescaled = 1: q carries log2(e)/8 (what packing.pack_qkv folds into W_q); reference on the
    # fp16-rounded scaled q so that both paths see the same operand
    qs = _h(qkv[:, :C] * ops.ATTN_Q_SCALE_LOG2)
    q2 = (qs / ops.ATTN_Q_SCALE_LOG2).reshape(frames, S, heads, 64).permute(0, 2, 1, 3)
    ref2 = F.scaled_dot_product_attention(q2.double(), k.double(), v.double())
    ref2 = ref2.permute(0, 2, 1, 3).reshape(frames * S, C).float()
    qg2 = torch.cat([qs, qkv[:, C:]], 1).half().to(gpu)
    out2 = torch.empty_like(out)
    ops.attn_spatial(qg2, vt, S_pad, out2, frames, S, heads, q_prescaled=True)
    torch.cuda.synchronize()
    e = rel_l2(out2.float(), ref2)
    assert e < 1.5e-3, f"spatial attention (prescaled q) S={S}: rel-L2 {e:.3e}"


def test_attention_spatial_reference_shift(gpu, attn_impl):
    """Scores that drift up and down by far more than the lazy-rescale threshold, a first tile whose
    scores are hugely negative, and a late outlier key: every branch of the reference-max logic."""
    from gcd_amd import ops
    g = _gen(12)
    frames, S, heads = 1, 640, 1
    C = 64
    q = torch.randn(S, 64, generator=g)
    k = torch.randn(S, 64, generator=g)
    v = torch.randn(S, 64, generator=g)
    u = torch.randn(64, generator=g)
    u /= u.norm()
    q = q + 30.0 * u                               # every query has a big component along u
    ramp = torch.linspace(-4.0, 4.0, S)            # keys drift along u: scores sweep ~[-120, 120]/8
    k = k + ramp[:, None] * u
    k[500] += 3.0 * u                               # late outlier
    qkv = _h(torch.cat([q, k, v], 1))
    qq, kk, vv = [t.reshape(1, S, 1, 64).permute(0, 2, 1, 3) for t in qkv.split(C, dim=1)]
    ref = F.scaled_dot_product_attention(qq.double(), kk.double(), vv.double())
    ref = ref.permute(0, 2, 1, 3).reshape(S, C).float()
    S_pad = S
    qg = qkv.half().to(gpu)
    vt = torch.empty(64 * S_pad, dtype=torch.float16, device=gpu)
    ops.attn_transpose_v(qg, frames, S, heads, vt, S_pad)
    out = torch.empty(S, C, dtype=torch.float16, device=gpu)
    ops.attn_spatial(qg, vt, S_pad, out, frames, S, heads)
    torch.cuda.synchronize()
    assert torch.isfinite(out).all()
    e = rel_l2(out.float(), ref)
    assert e < 2e-3, f"reference-shift attention: rel-L2 {e:.3e}"


@pytest.mark.parametrize("kernel", ["mfma", "valu"])
@pytest.mark.parametrize("clips,T,HW,heads", [(2, 14, 10, 3), (1, 4, 33, 1), (2, 16, 7, 5),
                                              (2, 14, 64, 20), (1, 1, 5, 2), (2, 14, 1500, 5)])
def test_attention_temporal(gpu, clips, T, HW, heads, kernel):
    """Both kernels of gcd_attn_temporal_f16: one problem per wave on the matrix pipe (the default) and the
    16-lanes-per-problem VALU kernel (GCD_TUNE_ATTN_IMPL = 16); the last shape has more problems than resident waves
    (the mixed-radix problem walk wraps clips, pixels and heads)."""
    from gcd_amd import ops
    ops.tune_set(ops.TUNE_ATTN_IMPL, 16 if kernel == "valu" else 0)
    g = _gen(11)
    C = heads * 64
    M = clips * T * HW
    qkv = _h(torch.randn(M, 3 * C, generator=g) * 1.5)
    q, k, v = [t.reshape(clips, T, HW, heads, 64).permute(0, 2, 3, 1, 4) for t in qkv.split(C, dim=1)]
    ref = F.scaled_dot_product_attention(q.double(), k.double(), v.double())   # b s h t d
    ref = ref.permute(0, 3, 1, 2, 4).reshape(M, C).float()
    out = torch.empty(M, C, dtype=torch.float16, device=gpu)
    try:
        ops.attn_temporal(qkv.half().to(gpu), out, clips, T, HW, heads)
        torch.cuda.synchronize()
    finally:
        ops.tune_set(ops.TUNE_ATTN_IMPL, 0)
    e = rel_l2(out.float(), ref)
    assert e < TOL_F16, f"temporal attention ({kernel}) rel-L2 {e:.3e}"


# ----------------------------------------------------------------------------------- small pieces
@pytest.mark.parametrize("M,N,K", [(28, 1280, 320), (28, 1280, 1280), (14, 320, 1280), (2, 64, 1024),
                                   (28, 100, 768), (28, 1280, 128), (5, 33, 20), (16, 48, 260),
                                   (56, 1280, 320), (33, 64, 128), (100, 48, 64),   # M > 32: two clips under CFG
                                   (28, 4160, 1280), (14, 4101, 260), (40, 8192, 320), (5, 4096, 516)])   # N >= 4096: the fp32 matrix-pipe kernel
def test_linear_smallm(gpu, M, N, K):
    from gcd_amd import ops
    g = _gen(12)
    x, w, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / math.sqrt(K), torch.randn(N, generator=g)
    y0 = torch.randn(M, N, generator=g)
    ref = y0 + F.silu(F.linear(F.silu(x), w, b))
    y = y0.clone().to(gpu)
    ops.linear_smallm(x.to(gpu), w.to(gpu), b.to(gpu), y, silu_in=True, silu_out=True, accumulate=True)
    torch.cuda.synchronize()
    assert rel_l2(y, ref) < 1e-5
    y = torch.empty(M, N, device=gpu)
    ops.linear_smallm(x.to(gpu), w.to(gpu), None, y)
    torch.cuda.synchronize()
    assert rel_l2(y, F.linear(x, w)) < 1e-5


def test_pack_unpack_cast(gpu):
    from gcd_amd import ops
    g = _gen(13)
    nx, N, H, W = 3, 6, 5, 7
    x = torch.randn(nx, 4, H, W, generator=g)
    cc = torch.randn(N, 4, H, W, generator=g)
    c_in = torch.rand(N, generator=g) + 0.5
    ref = torch.zeros(N, H * W, 64)
    ref[:, :, :4] = (x.repeat(2, 1, 1, 1) * c_in[:, None, None, None]).permute(0, 2, 3, 1).reshape(N, H * W, 4)
    ref[:, :, 4:8] = cc.permute(0, 2, 3, 1).reshape(N, H * W, 4)
    out = torch.full((N * H * W, 64), 7.0, dtype=torch.float16, device=gpu)
    ops.pack_input(x.to(gpu), cc.to(gpu), c_in.to(gpu), N, H * W, out, 64)
    torch.cuda.synchronize()
    assert torch.equal(out.cpu(), ref.reshape(-1, 64).half())
    tok = torch.randn(N * H * W, 16, generator=g)
    nchw = torch.empty(N, 4, H, W, device=gpu)
    ops.unpack_output(tok.to(gpu), nchw, 4, N, H * W)
    torch.cuda.synchronize()
    assert torch.equal(nchw.cpu(), tok[:, :4].reshape(N, H, W, 4).permute(0, 3, 1, 2))
    x32 = torch.randn(100, 64, generator=g)
    y16 = torch.empty(100, 64, dtype=torch.float16, device=gpu)
    ops.cast_f16(x32.to(gpu), y16)
    torch.cuda.synchronize()
    assert torch.equal(y16.cpu(), x32.half())


def test_sampler_kernels(gpu):
    from gcd_amd import ops
    g = _gen(14)
    nx, T = 4, 2
    x = torch.randn(nx, 4, 6, 6, generator=g) * 50
    net = torch.randn(2 * nx, 4, 6, 6, generator=g)
    scale = torch.tensor([1.0, 1.5])
    sigma, sigma_next = 37.5, 21.0
    c_skip, c_out = 1 / (sigma ** 2 + 1), -sigma / math.sqrt(sigma ** 2 + 1)
    den = net * c_out + torch.cat([x, x]) * c_skip
    du, dc = den.chunk(2)
    d_ = du + scale.repeat(nx // T)[:, None, None, None] * (dc - du)
    ref = x + (sigma_next - sigma) * ((x - d_) / sigma)
    out = torch.empty_like(x, device=gpu)
    sig = torch.tensor([sigma, sigma_next], device=gpu)
    ops.cfg_euler_step(x.to(gpu), net.to(gpu), scale.to(gpu), sig, out, T)
    c_in, c_noise = torch.empty(8, device=gpu), torch.empty(8, device=gpu)
    ops.edm_scalings(sig, c_in, c_noise)
    t = torch.tensor([0.0, 1.6378, -1.55, 13.0], device=gpu)
    emb = torch.empty(4, 320, device=gpu)
    ops.timestep_embedding(t, emb)
    torch.cuda.synchronize()
    assert rel_l2(out, ref) < 1e-5
    assert abs(float(c_in[0]) - 1 / math.sqrt(sigma ** 2 + 1)) < 1e-7
    assert abs(float(c_noise[3]) - 0.25 * math.log(sigma)) < 1e-6
    half = 160
    freqs = torch.exp(-math.log(10000) * torch.arange(half, dtype=torch.float32) / half)
    args = t.cpu()[:, None] * freqs[None]
    ref_emb = torch.cat([torch.cos(args), torch.sin(args)], -1)
    assert (emb.cpu() - ref_emb).abs().max() < 2e-6


def test_graph_capture_replay(gpu):
    """A captured launch sequence replays with new data in the same buffers."""
    import ctypes as C
    from gcd_amd import _lib, ops
    lib = _lib.load()
    g = _gen(15)
    M, N, K = 256, 160, 128
    a = torch.randn(M, K, generator=g).half().to(gpu)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).half().to(gpu)
    out = torch.zeros(M, N, device=gpu)
    y16 = torch.zeros(M, N, dtype=torch.float16, device=gpu)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        ops.gemm(a, w, out, M=M)          # warm up (module load, attribute set) outside capture
        st.synchronize()
        _lib.check(lib.gcd_graph_begin_capture(st.cuda_stream))
        ops.gemm(a, w, out, M=M)
        ops.cast_f16(out, y16)
        exe = C.c_void_p()
        _lib.check(lib.gcd_graph_end_capture(st.cuda_stream, C.byref(exe)))
        a.copy_(torch.randn(M, K, generator=g).half())
        st.synchronize()
        _lib.check(lib.gcd_graph_launch(exe, st.cuda_stream))
        _lib.check(lib.gcd_stream_sync(st.cuda_stream))
    ref = a.float().cpu() @ w.float().cpu().t()
    assert rel_l2(y16.float(), ref) < TOL_F16
    _lib.check(lib.gcd_graph_destroy(exe))
