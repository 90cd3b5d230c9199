"""GPU: gradient parity of the fine-tune step's HIP operators (gcd_amd.autograd_ops, through the C ABI)
with torch.autograd — per operator against plain fp32 torch on the CPU, per block and for a whole
TINY-width VideoUNet training step against torch.autograd over the CPU oracle (oracle/svd_unet_ref.py),
then the optimizer step (BASELINE.json cfg4).

Operands are fp16 with fp32 accumulation in both directions (activations and incoming gradients are
rounded once per contraction), so per-operator gradients carry ~2^-11 relative noise per operand:
tolerance 3e-3 rel-L2 per operator, 5e-3 on the gradients of whole blocks / the whole network, as
measured values are printed."""
import math

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_l2
from oracle import svd_unet_ref as O, weights

pytestmark = pytest.mark.gpu
TOL_OP = 3e-3
TOL_NET = 5e-3


def _gen(seed):
    return torch.Generator().manual_seed(seed)


def _leaf(t, gpu=None):
    t = t.clone().to(gpu) if gpu is not None else t.clone()
    return t.requires_grad_(True)


def _tok(x):            # [n, c, h, w] -> [n*h*w, c]
    n, c, h, w = x.shape
    return x.permute(0, 2, 3, 1).reshape(n * h * w, c).contiguous()


def _untok(t, n, h, w):
    return t.reshape(n, h, w, -1).permute(0, 3, 1, 2)


def _check(name, got, ref, tol=TOL_OP):
    e = rel_l2(got, ref)
    print(f"  {name}: rel-L2 {e:.2e}")
    assert e < tol, f"{name}: rel-L2 {e:.3e}"


# ------------------------------------------------------------------------------------------ operators
@pytest.mark.parametrize("M,K,N,bias", [(300, 64, 128, True), (28, 256, 64, True), (1000, 320, 48, False),
                                        (43008, 320, 960, False), (10752, 640, 5120, True)])
def test_linear_backward(gpu, M, K, N, bias):
    """(the last two: cfg4's token counts — dW is a 4- / 40-tile GEMM over 43008 / 10752 tokens, split-K)"""
    from gcd_amd import autograd_ops as A
    g = _gen(1)
    x, w, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / math.sqrt(K), torch.randn(N, generator=g)
    dy = torch.randn(M, N, generator=g)
    xr, wr, br = _leaf(x), _leaf(w), _leaf(b)
    F.linear(xr, wr, br if bias else None).backward(dy)
    xg, wg, bg = _leaf(x, gpu), _leaf(w, gpu), _leaf(b, gpu)
    y = A.linear(xg, wg, bg if bias else None)
    _check("y", y, F.linear(x, w, b if bias else None))
    y.backward(dy.to(gpu))
    _check("dx", xg.grad, xr.grad)
    _check("dw", wg.grad, wr.grad)
    if bias:
        _check("db", bg.grad, br.grad, 1e-5)


@pytest.mark.parametrize("frames,H,W,Cin,Cout,stride,up", [
    (2, 8, 8, 64, 64, 1, False), (3, 6, 10, 64, 128, 2, False), (2, 4, 6, 128, 64, 1, True),
    (2, 8, 8, 8, 64, 1, False), (2, 8, 8, 64, 4, 1, False)])
def test_conv3x3_backward(gpu, frames, H, W, Cin, Cout, stride, up):
    from gcd_amd import autograd_ops as A
    g = _gen(2)
    x = torch.randn(frames, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)
    b = torch.randn(Cout, generator=g)
    xr, wr, br = _leaf(x), _leaf(w), _leaf(b)
    inp = F.interpolate(xr, scale_factor=2, mode="nearest") if up else xr
    yr = F.conv2d(inp, wr, br, stride=stride, padding=1)
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy)
    xg, wg, bg = _leaf(_tok(x), gpu), _leaf(w, gpu), _leaf(b, gpu)
    y = A.conv3x3(xg, wg, bg, frames, H, W, stride=stride, upsample=up)
    _check("y", y, _tok(yr))
    y.backward(_tok(dy).to(gpu))
    _check("dx", xg.grad, _tok(xr.grad))
    _check("dw", wg.grad, wr.grad)
    _check("db", bg.grad, br.grad, 1e-5)


@pytest.mark.parametrize("clips,T,HW,C", [(2, 5, 12, 64), (1, 14, 16, 128)])
def test_conv_t3_backward(gpu, clips, T, HW, C):
    from gcd_amd import autograd_ops as A
    g = _gen(3)
    x = torch.randn(clips, C, T, HW, 1, generator=g)
    w = torch.randn(C, C, 3, 1, 1, generator=g) / math.sqrt(3 * C)
    b = torch.randn(C, generator=g)
    xr, wr, br = _leaf(x), _leaf(w), _leaf(b)
    yr = F.conv3d(xr, wr, br, padding=(1, 0, 0))
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy)
    tok = lambda t: t[..., 0].permute(0, 2, 3, 1).reshape(clips * T * HW, C).contiguous()   # noqa: E731
    xg, wg, bg = _leaf(tok(x), gpu), _leaf(w, gpu), _leaf(b, gpu)
    y = A.conv_t3(xg, wg, bg, T, HW)
    _check("y", y, tok(yr))
    y.backward(tok(dy).to(gpu))
    _check("dx", xg.grad, tok(xr.grad))
    _check("dw", wg.grad, wr.grad)
    _check("db", bg.grad, br.grad, 1e-5)


@pytest.mark.parametrize("frames,HW,C,per_clip_T,silu", [(4, 64, 64, 0, True), (4, 50, 320, 2, True), (3, 33, 128, 0, False)])
def test_groupnorm_backward(gpu, frames, HW, C, per_clip_T, silu):
    from gcd_amd import autograd_ops as A
    g = _gen(4)
    x = torch.randn(frames * HW, C, generator=g) * 2 + 0.7
    gamma, beta = torch.randn(C, generator=g), torch.randn(C, generator=g)
    dy = torch.randn(frames * HW, C, generator=g)
    rows = (per_clip_T or 1) * HW
    xr, gr, br = _leaf(x), _leaf(gamma), _leaf(beta)
    xn = xr.reshape(frames * HW // rows, rows, C).permute(0, 2, 1)
    yr = F.group_norm(xn, 32, gr, br, 1e-5)
    yr = (F.silu(yr) if silu else yr).permute(0, 2, 1).reshape(frames * HW, C)
    yr.backward(dy)
    xg, gg, bg = _leaf(x, gpu), _leaf(gamma, gpu), _leaf(beta, gpu)
    y = A.group_norm(xg, gg, bg, rows, 1e-5, silu)
    _check("y", y, yr, 6e-4)
    y.backward(dy.to(gpu))
    _check("dx", xg.grad, xr.grad, 1e-4)
    _check("dgamma", gg.grad, gr.grad, 1e-4)
    _check("dbeta", bg.grad, br.grad, 1e-4)


@pytest.mark.parametrize("M,C", [(100, 64), (500, 320), (33, 1280)])
def test_layernorm_backward(gpu, M, C):
    from gcd_amd import autograd_ops as A
    g = _gen(5)
    x = torch.randn(M, C, generator=g) * 1.5 + 0.3
    gamma, beta = torch.randn(C, generator=g), torch.randn(C, generator=g)
    dy = torch.randn(M, C, generator=g)
    xr, gr, br = _leaf(x), _leaf(gamma), _leaf(beta)
    F.layer_norm(xr, (C,), gr, br, 1e-5).backward(dy)
    xg, gg, bg = _leaf(x, gpu), _leaf(gamma, gpu), _leaf(beta, gpu)
    A.layer_norm(xg, gg, bg).backward(dy.to(gpu))
    _check("dx", xg.grad, xr.grad, 1e-4)
    _check("dgamma", gg.grad, gr.grad, 1e-4)
    _check("dbeta", bg.grad, br.grad, 1e-4)


def test_geglu_backward(gpu):
    from gcd_amd import autograd_ops as A
    g = _gen(6)
    h = torch.randn(200, 512, generator=g) * 1.5
    dout = torch.randn(200, 256, generator=g)
    hr = _leaf(h)
    a, gate = hr.chunk(2, dim=-1)
    yr = a * F.gelu(gate)
    yr.backward(dout)
    hg = _leaf(h, gpu)
    y = A.geglu(hg)
    _check("y", y, yr, 1e-5)
    y.backward(dout.to(gpu))
    _check("dh", hg.grad, hr.grad, 1e-5)


@pytest.mark.parametrize("frames,S,heads", [(2, 64, 2), (1, 128, 1), (2, 16, 1), (1, 100, 2), (3, 24, 4), (1, 33, 1),
                                            (2, 201, 1), (2, 1536, 2)])
def test_spatial_attention_backward(gpu, frames, S, heads):
    from gcd_amd import autograd_ops as A
    g = _gen(7)
    C = heads * 64
    qkv = torch.randn(frames * S, 3 * C, generator=g)
    dO = torch.randn(frames * S, C, generator=g)
    qr = _leaf(qkv)
    q, k, v = (t.reshape(frames, S, heads, 64).transpose(1, 2) for t in qr.chunk(3, dim=-1))
    yr = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(frames * S, C)
    yr.backward(dO)
    qg = _leaf(qkv, gpu)
    y = A.spatial_attention(qg, frames, S, heads)
    _check("out", y, yr, 1.5e-3)
    y.backward(dO.to(gpu))
    _check("dqkv", qg.grad, qr.grad)


def test_spatial_attention_backward_small_gradients_and_scale(gpu):
    """The flash backward against fp64 math on the SAME fp16-rounded q | k | v, O and dO it consumes: isolates
    the kernels' own error (P and dS rounded to fp16 for the MFMAs) from the rounding of the inputs; with
    row-dependent magnitudes so that a wrong lse / delta row mapping cannot hide."""
    from gcd_amd import ops
    g = _gen(71)
    frames, S, heads = 2, 300, 3
    C = heads * 64
    qkv = (torch.randn(frames * S, 3 * C, generator=g) * torch.linspace(0.3, 2.0, frames * S)[:, None]).half()
    dO = (torch.randn(frames * S, C, generator=g) * torch.linspace(2.0, 0.1, frames * S)[:, None]).half()
    q, k, v = (t.double().reshape(frames, S, heads, 64).transpose(1, 2).requires_grad_(True) for t in qkv.chunk(3, dim=-1))
    o = F.scaled_dot_product_attention(q, k, v)
    o16 = o.detach().transpose(1, 2).reshape(frames * S, C).half()
    # delta uses the fp16 O the forward produced: feed the same here
    o.backward(dO.double().reshape(frames, S, heads, 64).transpose(1, 2))
    ref = torch.cat([t.grad.transpose(1, 2).reshape(frames * S, C) for t in (q, k, v)], 1)
    dqkv = torch.empty(frames * S, 3 * C, device=gpu)
    ws = torch.empty(ops.attn_spatial_bwd_ws_bytes(frames, S, heads), dtype=torch.uint8, device=gpu)
    ops.attn_spatial_bwd(qkv.to(gpu), o16.to(gpu), dO.to(gpu), dqkv, frames, S, heads, ws)
    for i, name in enumerate("qkv"):
        _check("d" + name, dqkv[:, i * C:(i + 1) * C], ref[:, i * C:(i + 1) * C], 1e-3)


@pytest.mark.parametrize("clips,T,HW,heads", [(2, 14, 6, 2), (1, 4, 33, 1), (1, 16, 5, 3)])
def test_temporal_attention_backward(gpu, clips, T, HW, heads):
    from gcd_amd import autograd_ops as A
    g = _gen(8)
    C = heads * 64
    M = clips * T * HW
    qkv = torch.randn(M, 3 * C, generator=g)
    dO = torch.randn(M, C, generator=g)
    qr = _leaf(qkv)
    # rows (clip, t, hw) -> batch (clip, hw), tokens t
    q, k, v = (t.reshape(clips, T, HW, heads, 64).permute(0, 2, 3, 1, 4) for t in qr.chunk(3, dim=-1))
    yr = F.scaled_dot_product_attention(q, k, v).permute(0, 3, 1, 2, 4).reshape(M, C)
    yr.backward(dO)
    qg = _leaf(qkv, gpu)
    y = A.temporal_attention(qg, clips, T, HW, heads)
    _check("out", y, yr, 1.5e-3)
    y.backward(dO.to(gpu))
    _check("dqkv", qg.grad, qr.grad, 1e-3)      # fp32 backward on fp16-rounded q, k, v


@pytest.mark.parametrize("dtype", ["fp16", "bf16"])
@pytest.mark.parametrize("M,N,K", [(4096, 320, 1280), (1000, 208, 336), (2688, 1280, 640), (100, 8, 24),
                                   # M >= 32768 and N K >= 400 000: the library picks the 64-TOKEN form (69 632 B of dynamic
                                   # LDS behind the opt-in) — what cfg4's large weight gradients run; ragged token count
                                   (32776, 1024, 400)])
def test_weight_gradient_transposing_read_kernel(gpu, M, N, K, dtype):
    """gcd_wgrad_tr_f16 (libgcd_amd_train.so): dW = dY^T X with both operands row-major, transposed on the LDS read
    (ds_read_b64_tr_b16) — against fp32 torch and against the round-3 path (transposed copies + split-K gcd_gemm_f16),
    ragged tiles, a token count that is not a multiple of the 32-token step, strided operand views."""
    from gcd_amd import autograd_ops as A
    dt = torch.float16 if dtype == "fp16" else torch.bfloat16
    g = torch.Generator().manual_seed(5)
    dyb = (torch.randn(M, N + 8, generator=g) * 0.5).to(dt).to(gpu)
    xb = torch.randn(M, K + 16, generator=g).to(dt).to(gpu)
    dy, x = dyb[:, :N], xb[:, 8:8 + K]                       # row strides N + 8 / K + 16, 16-byte aligned starts
    ref = dy.float().cpu().t() @ x.float().cpu()
    old = A.WGRAD_IMPL
    try:
        A.set_wgrad_impl("tr")
        dw_tr = A._wgrad(dy, x)
        A.set_wgrad_impl("gemm")
        dw_gemm = A._wgrad(dy.contiguous(), x.contiguous()) if K % 16 == 0 and N % 16 == 0 else None
        torch.cuda.synchronize()
    finally:
        A.set_wgrad_impl(old)
    tol = 1e-4                                               # exact products, fp32 accumulation in both paths
    assert rel_l2(dw_tr, ref) < tol, rel_l2(dw_tr, ref)
    if dw_gemm is not None:
        assert rel_l2(dw_tr, dw_gemm.cpu()) < 2 * tol


def test_adam_step_vs_torch(gpu):
    from gcd_amd.training import AdamHIP
    g = _gen(9)
    shapes = [(300, 7), (5,), (64, 64)]
    ps = [torch.randn(s, generator=g) for s in shapes]
    ref = [torch.nn.Parameter(p.clone()) for p in ps]
    mine = [torch.nn.Parameter(p.clone().to(gpu)) for p in ps]
    opt_r = torch.optim.Adam(ref, lr=2e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    opt_m = AdamHIP(mine, lr=2e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    for step in range(4):
        grads = [torch.randn(s, generator=g) for s in shapes]
        for p, q, gr in zip(ref, mine, grads):
            p.grad = gr.clone()
            q.grad = (gr * 64.0).to(gpu)               # a loss scale that the step removes
        opt_r.step()
        opt_m.step(grad_scale=1.0 / 64.0)
    for p, q in zip(ref, mine):
        assert rel_l2(q, p) < 1e-6


def test_qkv_linear_backward(gpu):
    from gcd_amd import autograd_ops as A
    g = _gen(41)
    M, C = 700, 128
    x = torch.randn(M, C, generator=g)
    ws = [torch.randn(C, C, generator=g) / math.sqrt(C) for _ in range(3)]
    dy = torch.randn(M, 3 * C, generator=g)
    xr, wr = _leaf(x), [_leaf(w) for w in ws]
    F.linear(xr, torch.cat(wr, 0)).backward(dy)
    xg, wg = _leaf(x, gpu), [_leaf(w, gpu) for w in ws]
    y = A.qkv_linear(xg, *wg)
    _check("y", y, F.linear(x, torch.cat(ws, 0)))
    y.backward(dy.to(gpu))
    _check("dx", xg.grad, xr.grad)
    for a, b, n in zip(wg, wr, "qkv"):
        _check("dw" + n, a.grad, b.grad)


def test_fused_node_norm_contraction_vector_residual(gpu):
    """autograd_ops.Fused: [LayerNorm | GroupNorm+SiLU] -> [Linear | Conv3x3 | Conv (3,1,1)] + per-frame vector +
    residual as ONE graph node, against the same chain of torch ops: output and every gradient (input, norm
    affine, weight, bias, vector, residual)."""
    from gcd_amd import autograd_ops as A
    g = _gen(43)
    frames, H, W, C, Co = 4, 8, 8, 64, 128
    HW, M = H * W, 4 * 64
    # ---- LayerNorm -> Linear + per-frame vector + residual ----
    x = torch.randn(M, C, generator=g)
    ln = torch.nn.LayerNorm(C)
    with torch.no_grad():
        ln.weight.copy_(1 + 0.2 * torch.randn(C, generator=g))
        ln.bias.copy_(0.1 * torch.randn(C, generator=g))
    w, b = torch.randn(Co, C, generator=g) / math.sqrt(C), torch.randn(Co, generator=g)
    vec, res, dy = torch.randn(frames, Co, generator=g), torch.randn(M, Co, generator=g), torch.randn(M, Co, generator=g)
    leaves = [_leaf(t) for t in (x, w, b, vec, res)]
    xr, wr, br, vr, rr = leaves
    yr = F.linear(ln(xr), wr, br) + vr.repeat_interleave(HW, 0) + rr
    yr.backward(dy)
    lg = torch.nn.LayerNorm(C).to(gpu)
    lg.load_state_dict(ln.state_dict())
    xg, wg, bg, vg, rg = (_leaf(t, gpu) for t in (x, w, b, vec, res))
    y = A.linear(xg, wg, bg, norm=("ln", lg, 1e-5), residual=rg, rowvec=(vg, HW))
    print("LayerNorm -> Linear + vector + residual:")
    _check("y", y, yr, 1e-3)
    y.backward(dy.to(gpu))
    for n, a_, b_ in [("dx", xg, xr), ("dw", wg, wr), ("db", bg, br), ("dvec", vg, vr), ("dres", rg, rr),
                      ("dgamma", lg.weight, ln.weight), ("dbeta", lg.bias, ln.bias)]:
        _check(n, a_.grad, b_.grad)
    # ---- GroupNorm + SiLU -> Conv3x3 + per-frame vector + residual ----
    x = torch.randn(frames, C, H, W, generator=g)
    gn = torch.nn.GroupNorm(32, C)
    with torch.no_grad():
        gn.weight.copy_(1 + 0.2 * torch.randn(C, generator=g))
        gn.bias.copy_(0.1 * torch.randn(C, generator=g))
    w, b = torch.randn(Co, C, 3, 3, generator=g) / math.sqrt(9 * C), torch.randn(Co, generator=g)
    res = torch.randn(frames, Co, H, W, generator=g)
    dy = torch.randn(frames, Co, H, W, generator=g)
    xr, wr, br, vr, rr = (_leaf(t) for t in (x, w, b, vec, res))
    yr = F.conv2d(F.silu(gn(xr)), wr, br, padding=1) + vr[:, :, None, None] + rr
    yr.backward(dy)
    gg = torch.nn.GroupNorm(32, C).to(gpu)
    gg.load_state_dict(gn.state_dict())
    xg, wg, bg, vg, rg = _leaf(_tok(x), gpu), _leaf(w, gpu), _leaf(b, gpu), _leaf(vec, gpu), _leaf(_tok(res), gpu)
    y = A.conv3x3(xg, wg, bg, frames, H, W, norm=("gn", gg, HW, 1e-5, True), residual=rg, rowvec=(vg, HW))
    print("GroupNorm+SiLU -> Conv3x3 + vector + residual:")
    _check("y", y, _tok(yr), 1e-3)
    y.backward(_tok(dy).to(gpu))
    for n, a_, b_ in [("dx", xg.grad, _tok(xr.grad)), ("dw", wg.grad, wr.grad), ("db", bg.grad, br.grad),
                      ("dvec", vg.grad, vr.grad), ("dres", rg.grad, _tok(rr.grad)),
                      ("dgamma", gg.weight.grad, gn.weight.grad), ("dbeta", gg.bias.grad, gn.bias.grad)]:
        _check(n, a_, b_)
    # ---- GroupNorm over T*H*W + SiLU -> Conv (3,1,1) + residual ----
    clips, T = 2, 2
    gn3 = torch.nn.GroupNorm(32, C)
    w3 = torch.randn(C, C, 3, 1, 1, generator=g) / math.sqrt(3 * C)
    x5 = x.reshape(clips, T, C, H, W).permute(0, 2, 1, 3, 4).contiguous()            # b c t h w
    xr, wr = _leaf(x5), _leaf(w3)
    yr = F.conv3d(F.silu(gn3(xr)), wr, None, padding=(1, 0, 0)) + xr
    dy5 = torch.randn(yr.shape, generator=g)
    yr.backward(dy5)
    tok5 = lambda t: t.permute(0, 2, 3, 4, 1).reshape(clips * T * HW, C).contiguous()    # noqa: E731
    g3 = torch.nn.GroupNorm(32, C).to(gpu)
    xg, wg = _leaf(tok5(x5), gpu), _leaf(w3, gpu)
    y = A.conv_t3(xg, wg, None, T, HW, norm=("gn", g3, T * HW, 1e-5, True), residual=xg)
    print("GroupNorm(T*H*W)+SiLU -> Conv (3,1,1) + residual (the block input itself):")
    _check("y", y, tok5(yr), 1e-3)
    y.backward(tok5(dy5).to(gpu))
    _check("dx", xg.grad, tok5(xr.grad))
    _check("dw", wg.grad, wr.grad)


# -------------------------------------------------------------------------------------- blocks, network
def _tiny_unet(gpu, salt=0):
    from gcd_amd.video_model import VideoUNet
    with torch.device("meta"):
        net = VideoUNet(**O.TINY.as_reference_kwargs())
    sd = weights.synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, salt)
    net = net.to_empty(device=gpu)
    net.load_state_dict(sd)
    return net.train(), sd


def _grad_dict(named, prefix=""):
    return {prefix + k: v.grad for k, v in named if v.grad is not None}


def test_video_resblock_and_transformer_gradients_vs_oracle(gpu):
    """One VideoResBlock and one SpatialVideoTransformer of the TINY UNet (input_blocks.1): output,
    input gradient and every parameter gradient vs torch.autograd over the oracle."""
    from gcd_amd import training as TR
    net, sd = _tiny_unet(gpu)
    T, H, W = 4, 8, 8
    frames = 2 * T
    g = _gen(11)
    C = 64
    x = torch.randn(frames, C, H, W, generator=g)
    emb = torch.randn(frames, 4 * C, generator=g)
    ctx = torch.randn(frames, 1, 64, generator=g)
    ioi = torch.zeros(2, T)
    ioi[1, 2] = 1.0
    # ---- VideoResBlock ----
    p = "input_blocks.1.0"
    sdr = {k: _leaf(v) for k, v in sd.items() if k.startswith(p)}
    xr, er = _leaf(x), _leaf(emb)
    yr = O._video_resblock(sdr, p, xr, er, T, ioi)
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy)
    rb = net.input_blocks[1][0]
    xg, eg = _leaf(_tok(x), gpu), _leaf(emb, gpu)
    y = TR._video_resblock(rb, xg, eg, frames, T, H, W, ioi.to(gpu))
    print("VideoResBlock:")
    _check("out", _untok(y, frames, H, W), yr, 2e-3)
    y.backward(_tok(dy).to(gpu))
    _check("dx", _untok(xg.grad, frames, H, W), xr.grad, TOL_NET)
    _check("demb", eg.grad, er.grad, TOL_NET)
    for name, prm in rb.named_parameters():
        _check("d " + name, prm.grad, sdr[f"{p}.{name}"].grad, TOL_NET)
    # ---- SpatialVideoTransformer ----
    p = "input_blocks.1.1"
    sdr = {k: _leaf(v) for k, v in sd.items() if k.startswith(p)}
    xr, cr = _leaf(x), _leaf(ctx)
    yr = O._spatial_video_transformer(sdr, p, xr, cr, T, ioi, O.TINY)
    yr.backward(dy)
    tr = net.input_blocks[1][1]
    for prm in tr.parameters():
        prm.grad = None
    xg = _leaf(_tok(x), gpu)
    y = TR._transformer(tr, xg, ctx.reshape(frames, -1).to(gpu), frames, T, H, W, ioi.to(gpu))
    print("SpatialVideoTransformer:")
    _check("out", _untok(y, frames, H, W), yr, 2e-3)
    y.backward(_tok(dy).to(gpu))
    _check("dx", _untok(xg.grad, frames, H, W), xr.grad, TOL_NET)
    dead = 0
    for name, prm in tr.named_parameters():
        ref = sdr[f"{p}.{name}"].grad
        if ref is None or float(ref.abs().max()) == 0.0:      # to_q / to_k / norm2 of the 1-key cross-attention
            assert prm.grad is None or float(prm.grad.abs().max()) == 0.0, name
            dead += 1
            continue
        _check("d " + name, prm.grad, ref, TOL_NET)
    assert dead >= 8, "the one-key cross-attention's q / k / norm2 must receive exactly zero gradient"


# The autograd engine of rounds 2-4 stays in the tree and is compared with the planned engine on the same kernels in
# tests/test_train_plan_gpu.py; its own runs through the end-to-end goldens below are duplicates of what rounds 2-4
# recorded (and cost 70 s of GPU time): they run with GCD_TEST_FULL=1 (profiles/r05_gpu_tests_tail.log: all of them green).
_FULL = __import__("os").environ.get("GCD_TEST_FULL", "0") == "1"


@pytest.fixture
def train_engine(request):
    """Run a test on one of the two engines of the fine-tune step (training.TRAIN_ENGINE), restoring the default."""
    from gcd_amd import training as TR
    old = TR.TRAIN_ENGINE
    TR.set_train_engine(request.param)
    yield request.param
    TR.set_train_engine(old)


@pytest.mark.parametrize("train_engine", ["planned", "autograd"] if _FULL else ["planned"], indirect=True)
@pytest.mark.parametrize("step", [0, 2500])
def test_unet_training_step_vs_oracle(gpu, step, train_engine):
    """BASELINE.json cfg4 at TINY width: denoiser + loss forward, backward through the whole VideoUNet on
    HIP kernels, Adam step — loss, every parameter gradient and the updated parameters vs
    torch.autograd / torch.optim.Adam over the CPU oracle.  step 0: plain mean loss (the focal top-k is
    not active yet); step 2500: top 55 % of the per-pixel losses — which pixels are "top" is a discrete
    choice, so a 1e-3 difference in the network output flips the membership of pixels at the threshold
    and the gradients differ by more than operand rounding alone."""
    from gcd_amd import training as TR
    from oracle import loss_ref as LR
    net, sd = _tiny_unet(gpu, salt=3)
    T, H, W, B = 4, 16, 16, 2          # 2 x 2 tokens at the bottleneck
    BT = B * T
    cfg = O.TINY
    g = _gen(12)
    x0 = torch.randn(BT, 4, H, W, generator=g)
    noise = torch.randn(BT, 4, H, W, generator=g)
    cond = {"crossattn": torch.randn(BT, 1, cfg.context_dim, generator=g),
            "concat": torch.randn(BT, 4, H, W, generator=g) * 0.8,
            "vector": torch.randn(BT, cfg.adm_in_channels + cfg.aux_emb_dim, generator=g).clamp(-1, 1)}
    sig = LR.harmonize(LR.edm_sigmas(torch.randn(BT, generator=g), 1.0, 1.6), T)
    ioi = torch.zeros(B, T)
    loss_scale = 256.0
    # ---- oracle: fp32 CPU autograd ----
    sdr = {k: _leaf(v) for k, v in sd.items()}
    noised = x0 + noise * sig[:, None, None, None]
    out_r = O.denoise(sdr, cfg, noised, sig, cond, T, ioi)
    loss_r = LR.get_loss(out_r, x0, LR.edm_weighting(sig, 1.0)[:, None, None, None], step, "l2", 0.1, 5000).mean()
    loss_r.backward()
    # ---- product ----
    den = TR.TrainDenoiser({"target": "gcd_amd.denoiser_scaling.VScalingWithEDMcNoise"})
    loss_fn = TR.StandardDiffusionLoss(
        sigma_sampler_config={"target": "gcd_amd.training.EDMSampling", "params": {"p_mean": 1.0, "p_std": 1.6}},
        loss_weighting_config={"target": "gcd_amd.training.EDMWeighting", "params": {"sigma_data": 1.0}},
        focus_top=0.1, focus_steps=5000, batch2model_keys=["image_only_indicator", "num_video_frames"])
    cg = {k: v.to(gpu) for k, v in cond.items()}
    sg = sig.to(gpu)
    out = den(net, noised.to(gpu), sg, cg, num_video_frames=T, image_only_indicator=ioi.to(gpu))
    w = loss_fn.loss_weighting(sg)[:, None, None, None]
    loss = loss_fn.get_loss(out, x0.to(gpu), w, {"global_step": step}).mean()
    (loss * loss_scale).backward()
    torch.cuda.synchronize()
    print(f"loss {float(loss):.6f} vs oracle {float(loss_r):.6f}")
    assert abs(float(loss) / float(loss_r) - 1.0) < 2e-3
    _check("denoiser output", out, out_r, 2e-3)
    num = den_ = 0.0
    worst = ("", 0.0)
    nz = 0
    errs = []
    for name, prm in net.named_parameters():
        ref = sdr[name].grad
        if ref is None or float(ref.abs().max()) == 0.0:
            assert prm.grad is None or float(prm.grad.abs().max()) == 0.0, name
            continue
        got = prm.grad.double().cpu() / loss_scale
        num += float((got - ref.double()).pow(2).sum())
        den_ += float(ref.double().pow(2).sum())
        e = rel_l2(got, ref)
        nz += 1
        if ref.numel() < 64:
            # a scalar blend logit's gradient is one long cancelling sum (sum dy (x_s - x_t)): its relative
            # error is not bounded by operand rounding; it is covered by the global norm below
            continue
        errs.append((e, name))
        if e > worst[1]:
            worst = (name, e)
    total = math.sqrt(num / den_)
    errs.sort(reverse=True)
    print(f"step {step}: parameter gradients: {nz} tensors, global rel-L2 {total:.3e}; worst: "
          + ", ".join(f"{n} {e:.2e}" for e, n in errs[:6]))
    tol = TOL_NET if step == 0 else 3 * TOL_NET
    assert total < tol and worst[1] < 4 * tol
    # ---- optimizer step (diffusion.py:412-431: Adam, lr from the config) ----
    if step != 0:
        return          # the update of a first Adam step is lr * sign(g): only meaningful where g is well resolved
    lr = 1e-3
    ref_params = [torch.nn.Parameter(sd[n].clone()) for n, _ in net.named_parameters()]
    for rp, (n, _) in zip(ref_params, net.named_parameters()):
        rp.grad = sdr[n].grad.clone() if sdr[n].grad is not None else torch.zeros_like(rp)
    torch.optim.Adam(ref_params, lr=lr).step()
    opt = TR.AdamHIP(net.parameters(), lr=lr)
    opt.step(grad_scale=1.0 / loss_scale)
    torch.cuda.synchronize()
    moved = 0.0
    for rp, (n, prm) in zip(ref_params, net.named_parameters()):
        if prm.grad is None:
            continue
        # the first Adam step moves every weight by ~lr * sign(grad): compare the updates themselves
        du, dr = prm.detach().cpu() - sd[n], rp.detach() - sd[n]
        big = sdr[n].grad.abs() > 0.05 * sdr[n].grad.abs().max()      # sign(g) is noise where g is ~0
        if big.any():
            assert rel_l2(du[big], dr[big]) < 2e-2, n
            moved += float(dr[big].abs().sum())
    assert moved > 0


# (the planned engine — the default — on all three fixtures; the autograd engine of rounds 2-4 on the first: 40 s each)
@pytest.mark.parametrize("train_engine,dtype,fixture", [
    ("planned", "fp16", "train_kubric_32x48.pt"), ("planned", "bf16", "train_kubric_32x48.pt"),
    ("planned", "fp16", "train_kubric_32x48_focal.pt")] + ([("autograd", "fp16", "train_kubric_32x48.pt")] if _FULL else []),
    indirect=["train_engine"])
def test_training_step_full_width_cfg4_vs_reference_golden(gpu, dtype, fixture, train_engine):
    """BASELINE.json cfg4 at its own shape: ONE fine-tune step of the full-width 1.53 B-parameter Kubric VideoUNet on
    2 clips x 14 frames of 32 x 48 latents (N = 28, activation checkpointing as in every GCD config) against the
    UNMODIFIED reference classes' fp32 torch.autograd run on the CPU (oracle/make_golden_cfg4.py): loss, denoiser
    output, and for each of the ~1400 parameter gradients its norm and 128 strided samples.  Run with fp16 GEMM
    operands (the inference engine's arithmetic) and with bf16 (what cfg4 names); the tolerance met is printed.
    The `_focal` fixture is the same step at global_step 2500: the annealed top-fraction loss keeps the 55 % largest
    per-pixel losses of every frame — a discrete choice that a 1e-3 difference of the output flips for the pixels at the
    threshold, so its gradients differ by more than operand rounding (bars x3)."""
    from pathlib import Path
    gold = Path(__file__).resolve().parent / "golden" / fixture
    if not gold.exists():
        pytest.skip(f"tests/golden/{fixture} has not been generated (python -m oracle.make_golden_cfg4 [focal])")
    from gcd_amd import autograd_ops as A
    from gcd_amd import training as TR
    from gcd_amd.video_model import VideoUNet
    from oracle.make_golden_cfg4 import inputs
    from oracle.make_golden_fullres import sample
    G = torch.load(gold)
    cfg = O.KUBRIC
    with torch.device("meta"):
        net = VideoUNet(**cfg.as_reference_kwargs())
    sd = weights.synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, G["salt"])
    net = net.to_empty(device=gpu)
    net.load_state_dict(sd)
    del sd
    net.train()
    x0, noise, cond, sig = inputs()
    B, T = G["B"], G["T"]
    loss_scale = 1024.0
    A.set_train_dtype(dtype)
    A.PACK.clear()
    try:
        den = TR.TrainDenoiser({"target": "gcd_amd.denoiser_scaling.VScalingWithEDMcNoise"}, use_checkpoint=True)
        loss_fn = TR.StandardDiffusionLoss(
            sigma_sampler_config={"target": "gcd_amd.training.EDMSampling", "params": {"p_mean": 1.0, "p_std": 1.6}},
            loss_weighting_config={"target": "gcd_amd.training.EDMWeighting", "params": {"sigma_data": 1.0}},
            focus_top=0.1, focus_steps=5000, batch2model_keys=["image_only_indicator", "num_video_frames"])
        noised = (x0 + noise * sig[:, None, None, None]).to(gpu)
        sg = sig.to(gpu)
        out = den(net, noised, sg, {k: v.to(gpu) for k, v in cond.items()}, num_video_frames=T,
                  image_only_indicator=torch.zeros(B, T, device=gpu))
        w = loss_fn.loss_weighting(sg)[:, None, None, None]
        loss = loss_fn.get_loss(out, x0.to(gpu), w, {"global_step": G["step"]}).mean()
        (loss * loss_scale).backward()
        torch.cuda.synchronize()
    finally:
        A.set_train_dtype("fp16")
        A.PACK.clear()
    # Bars.  fp16: 2e-3 / 5e-3 (measured 5.8e-4 gradients).  bf16, round 6: pinned to the reference instead of a 6x cushion —
    # the UNMODIFIED reference under its own `bf16-mixed` arithmetic (torch.autocast(bfloat16), what cfg4's config runs;
    # oracle/make_autocast_bars.py on this fixture's inputs) is 7.7e-3 (output) / 9.1e-2 (gradients) away from its own fp32
    # run: tests/golden/autocast_bars.json.  This path (bf16 GEMM operands, fp32 everything else) measures 4.8e-3 on the
    # gradients and 4.4e-3 on the output; its bars are 1.5x those measurements and, asserted below, inside the reference's own
    # bf16 distance.
    tol_out, tol_g = (2e-3, 5e-3) if dtype == "fp16" else (7e-3, 7.5e-3)
    if dtype == "bf16":
        import json
        ref = json.loads((Path(__file__).resolve().parent / "golden" / "autocast_bars.json").read_text())[
            "cfg4_reference_bf16_autocast_vs_own_fp32"]
        assert tol_g < 0.25 * ref["gradient_global_rel_l2"] and tol_out < ref["output_rel_l2"]
    if G["step"] > 0:
        tol_g *= 3.0
    e_out = rel_l2(sample(out.detach().cpu(), 65536), G["out_samples"])
    print(f"[{dtype}, step {G['step']}] loss {float(loss):.6f} vs reference {G['loss']:.6f}; denoiser output rel-L2 {e_out:.2e}")
    assert abs(float(loss) / G["loss"] - 1.0) < (2e-3 if dtype == "fp16" else 1e-2)
    assert e_out < tol_out
    num = den_ = 0.0
    worst_norm, worst = ("", 0.0), ("", 0.0)
    seen = 0
    total_ref = sum(v * v for v in G["grad_norms"].values()) ** 0.5
    n_dead = 0
    for name, prm in net.named_parameters():
        if name in G["dead"] or G["grad_norms"][name] < 1e-7 * total_ref:
            # to_q / to_k / norm2 of the one-key cross-attentions: exactly zero in exact arithmetic (and here: the
            # graph never reaches them); the reference's CPU flash kernel leaves rounding noise of ~1e-10 relative
            assert prm.grad is None or float(prm.grad.double().norm()) / loss_scale < 1e-7 * total_ref, name
            n_dead += 1
            continue
        ref_s = G["grad_samples"][name].double()
        got = prm.grad.detach().float().cpu() / loss_scale
        got_s = sample(got, 32 if G["step"] > 0 else 128).double()      # the fixture's grid (make_golden_cfg4.py)
        num += float((got_s - ref_s).pow(2).sum())
        den_ += float(ref_s.pow(2).sum())
        seen += 1
        rn = abs(float(got.double().norm()) / G["grad_norms"][name] - 1.0)
        if got.numel() >= 64:
            e = float((got_s - ref_s).norm() / ref_s.norm().clamp_min(1e-30))
            if e > worst[1]:
                worst = (name, e)
            if rn > worst_norm[1]:
                worst_norm = (name, rn)
    g_err = (num / den_) ** 0.5
    print(f"[{dtype}] {seen} parameter gradients: sampled global rel-L2 {g_err:.2e}; worst tensor {worst[0]} {worst[1]:.2e}; "
          f"worst norm ratio off by {worst_norm[1]:.2e} ({worst_norm[0]})")
    assert seen + n_dead == len(G["grad_norms"]) + len(G["dead"]) and n_dead >= 128
    assert g_err < tol_g, f"global gradient rel-L2 {g_err:.3e}"
    assert worst_norm[1] < 4 * tol_g, f"gradient norm of {worst_norm[0]} off by {worst_norm[1]:.3e}"
    del net
    torch.cuda.empty_cache()


def test_activation_checkpointing_matches(gpu):
    """`use_checkpoint` (True in every GCD config): ResBlocks and transformers re-run on HIP kernels during
    the backward pass — same gradients, less memory held between forward and backward."""
    from gcd_amd import training as TR
    net, sd = _tiny_unet(gpu, salt=5)
    T, H, W = 4, 16, 16
    cfg = O.TINY
    g = _gen(21)
    x = torch.randn(2 * T, 8, H, W, generator=g).to(gpu)
    ts = torch.linspace(-1.0, 1.5, 2 * T).to(gpu)
    ctx = torch.randn(2 * T, 1, cfg.context_dim, generator=g).to(gpu)
    y = torch.randn(2 * T, cfg.adm_in_channels + cfg.aux_emb_dim, generator=g).clamp(-1, 1).to(gpu)
    ioi = torch.zeros(2, T, device=gpu)
    tgt = torch.randn(2 * T, 4, H, W, generator=g).to(gpu)
    grads, held = [], []
    for ck in (False, True):
        for p in net.parameters():
            p.grad = None
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        base = torch.cuda.memory_allocated()
        out = TR.unet_forward_train(net, x, ts, ctx, y, T, ioi, use_checkpoint=ck)
        torch.cuda.synchronize()
        held.append(torch.cuda.memory_allocated() - base)
        ((out - tgt) ** 2).mean().backward()
        torch.cuda.synchronize()
        grads.append({n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None})
    assert grads[0].keys() == grads[1].keys() and len(grads[0]) > 1000
    for n in grads[0]:
        # Not bit-equal between two runs: bias / per-frame-vector / norm-parameter gradients are fp32 sums of atomics
        # whose ORDER differs from run to run (1e-7 .. 1e-6), and wherever such a sum feeds a later contraction (the
        # time embedding's gradient is the sum over all 44 emb_layers) its fp16 rounding turns a last-bit difference
        # into 2^-11 on that element: ~1e-4 on a whole tensor.  A wrong recomputation would be O(1).
        assert rel_l2(grads[1][n], grads[0][n]) < 1e-3, n
    print(f"activations held after forward: {held[0] / 2**20:.0f} MiB plain, {held[1] / 2**20:.0f} MiB checkpointed")
    assert held[1] < 0.5 * held[0]


@pytest.mark.parametrize("mode", ["no_grad", "frozen"])
def test_pack_cache_never_serves_a_recycled_temporary(gpu, mode):
    """the packed-operand cache was keyed on (address, shape, version) of whatever tensor it was
    handed; the q | k | v weight used to be a torch.cat temporary, which under no_grad / with frozen weights is
    a leaf whose address the allocator recycles for the NEXT block's temporary — which then ran with the
    previous block's weights.  Two consecutive self-attentions with different weights, evaluated repeatedly."""
    from gcd_amd import autograd_ops as A
    from gcd_amd import training as TR
    net, sd = _tiny_unet(gpu)
    A.PACK.clear()
    A.PACK.attach(net)
    atts = [net.input_blocks[1][1].transformer_blocks[0].attn1, net.input_blocks[2][1].transformer_blocks[0].attn1]
    frames, S, heads = 2, 64, atts[0].heads
    C = heads * 64
    g = _gen(51)
    x = torch.randn(frames * S, C, generator=g)

    def ref(att):
        w = {n: p.detach().cpu() for n, p in att.named_parameters()}
        q, k, v = (F.linear(x, w[f"to_{n}.weight"]).reshape(frames, S, heads, 64).transpose(1, 2) for n in "qkv")
        o = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(frames * S, C)
        return F.linear(o, w["to_out.0.weight"], w["to_out.0.bias"])

    if mode == "frozen":
        for a in atts:
            for p in a.parameters():
                p.requires_grad_(False)
    xg = x.to(gpu)
    for rep in range(3):
        for i, a in enumerate(atts):
            if mode == "no_grad":
                with torch.no_grad():
                    y = TR._self_attention(a, xg, "spatial", (frames, S, heads))
            else:
                y = TR._self_attention(a, xg, "spatial", (frames, S, heads))
            _check(f"rep {rep} block {i}", y, ref(a), 2e-3)
    # and a temporary handed to the cache is packed, never kept
    t = torch.randn(64, 64, device=gpu)
    n0 = len(A.PACK._d)
    A.PACK.get(t, "lin_fp16", lambda w: w.half())
    assert len(A.PACK._d) == n0


def test_bf16_operands_forward_and_backward(gpu):
    """`set_train_dtype("bf16")` (BASELINE.json cfg4 names bf16): every GEMM-family contraction, forward and
    backward, on bfloat16 operands — v_mfma_f32_32x32x16_bf16 on the ping-pong kernel for all three modes —
    against fp32 torch; tolerance is bf16's (8 significant bits per operand)."""
    from gcd_amd import autograd_ops as A
    A.set_train_dtype("bf16")
    try:
        g = _gen(61)
        for (M, K, N) in [(2048, 320, 640), (43008, 320, 320)]:
            x, w, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / math.sqrt(K), torch.randn(N, generator=g)
            dy = torch.randn(M, N, generator=g)
            xr, wr, br = _leaf(x), _leaf(w), _leaf(b)
            yr = F.linear(xr, wr, br)
            yr.backward(dy)
            xg, wg, bg = _leaf(x, gpu), _leaf(w, gpu), _leaf(b, gpu)
            y = A.linear(xg, wg, bg)
            print(f"Linear {M}x{K}x{N}, bf16 operands:")
            _check("y", y, yr, 8e-3)
            y.backward(dy.to(gpu))
            _check("dx", xg.grad, xr.grad, 8e-3)
            _check("dw", wg.grad, wr.grad, 8e-3)
        frames, H, W, C = 4, 16, 16, 320          # 1024 tokens x 320: full ping-pong tiles
        x = torch.randn(frames, C, H, W, generator=g)
        w = torch.randn(C, C, 3, 3, generator=g) / math.sqrt(9 * C)
        xr, wr = _leaf(x), _leaf(w)
        yr = F.conv2d(xr, wr, None, padding=1)
        dy = torch.randn(yr.shape, generator=g)
        yr.backward(dy)
        xg, wg = _leaf(_tok(x), gpu), _leaf(w, gpu)
        y = A.conv3x3(xg, wg, None, frames, H, W)
        print("Conv3x3, bf16 operands:")
        _check("y", y, _tok(yr), 8e-3)
        y.backward(_tok(dy).to(gpu))
        _check("dx", xg.grad, _tok(xr.grad), 8e-3)
        _check("dw", wg.grad, wr.grad, 8e-3)
        clips, T, HW = 2, 14, 64
        x = torch.randn(clips, C, T, HW, 1, generator=g)
        w = torch.randn(C, C, 3, 1, 1, generator=g) / math.sqrt(3 * C)
        xr, wr = _leaf(x), _leaf(w)
        yr = F.conv3d(xr, wr, None, padding=(1, 0, 0))
        dy = torch.randn(yr.shape, generator=g)
        yr.backward(dy)
        tok = lambda t: t[..., 0].permute(0, 2, 3, 1).reshape(clips * T * HW, C).contiguous()   # noqa: E731
        xg, wg = _leaf(tok(x), gpu), _leaf(w, gpu)
        y = A.conv_t3(xg, wg, None, T, HW)
        print("Conv (3,1,1), bf16 operands:")
        _check("y", y, tok(yr), 8e-3)
        y.backward(tok(dy).to(gpu))
        _check("dx", xg.grad, tok(xr.grad), 8e-3)
        _check("dw", wg.grad, wr.grad, 8e-3)
    finally:
        A.set_train_dtype("fp16")


def test_bf16_gradient_contractions(gpu):
    """`set_grad_dtype("bf16")`: the backward contractions on bfloat16 operands (fp32's exponent range: no
    loss scale; 8 significant bits) — operator gradients and one VideoResBlock against torch.autograd.
    Unscaled gradients of 1e-6 magnitude, which fp16 would flush, must come through."""
    from gcd_amd import autograd_ops as A
    from gcd_amd import training as TR
    A.set_grad_dtype("bf16")
    try:
        g = _gen(31)
        M, K, N = 500, 320, 128
        x, w, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / math.sqrt(K), torch.randn(N, generator=g)
        dy = torch.randn(M, N, generator=g) * 1e-6                    # far below fp16's normal range
        xr, wr, br = _leaf(x), _leaf(w), _leaf(b)
        F.linear(xr, wr, br).backward(dy)
        xg, wg, bg = _leaf(x, gpu), _leaf(w, gpu), _leaf(b, gpu)
        A.linear(xg, wg, bg).backward(dy.to(gpu))
        print("Linear, bf16 gradient operands, |dy| ~ 1e-6:")
        _check("dx", xg.grad, xr.grad, 1e-2)
        _check("dw", wg.grad, wr.grad, 1e-2)
        frames, H, W, C = 2, 8, 8, 64
        x = torch.randn(frames, C, H, W, generator=g)
        w = torch.randn(C, C, 3, 3, generator=g) / math.sqrt(9 * C)
        xr, wr = _leaf(x), _leaf(w)
        yr = F.conv2d(xr, wr, None, padding=1)
        dy = torch.randn(yr.shape, generator=g)
        yr.backward(dy)
        xg, wg = _leaf(_tok(x), gpu), _leaf(w, gpu)
        A.conv3x3(xg, wg, None, frames, H, W).backward(_tok(dy).to(gpu))
        print("Conv3x3, bf16 gradient operands:")
        _check("dx", xg.grad, _tok(xr.grad), 1e-2)
        _check("dw", wg.grad, wr.grad, 1e-2)
        # one VideoResBlock of the TINY UNet
        net, sd = _tiny_unet(gpu)
        T = 4
        x = torch.randn(2 * T, 64, 8, 8, generator=g)
        emb = torch.randn(2 * T, 256, generator=g)
        ioi = torch.zeros(2, T)
        p = "input_blocks.1.0"
        sdr = {k: _leaf(v) for k, v in sd.items() if k.startswith(p)}
        xr, er = _leaf(x), _leaf(emb)
        yr = O._video_resblock(sdr, p, xr, er, T, ioi)
        dy = torch.randn(yr.shape, generator=g)
        yr.backward(dy)
        rb = net.input_blocks[1][0]
        xg, eg = _leaf(_tok(x), gpu), _leaf(emb, gpu)
        TR._video_resblock(rb, xg, eg, 2 * T, T, 8, 8, ioi.to(gpu)).backward(_tok(dy).to(gpu))
        print("VideoResBlock, bf16 gradient operands:")
        _check("dx", _untok(xg.grad, 2 * T, 8, 8), xr.grad, 1.5e-2)
        worst = max(rel_l2(prm.grad, sdr[f"{p}.{n}"].grad) for n, prm in rb.named_parameters() if prm.numel() > 1)
        print(f"  worst parameter gradient rel-L2 {worst:.2e}")
        assert worst < 1.5e-2
    finally:
        A.set_grad_dtype("fp16")
