"""GPU: the planned fine-tune engine (gcd_amd/train_plan.py, round 5) and the kernels it adds (libgcd_amd_train.so:
gcd_train_pack_weights, gcd_wgrad_tr_f16_ex, gcd_blend_*, gcd_smallm_*), each against plain fp32 / fp64 torch, then the
whole TINY network planned vs the autograd engine on the same kernels.  The end-to-end goldens (cfg4 at its own shape vs
the unmodified reference; TINY vs the CPU oracle) run on BOTH engines in tests/test_backward_gpu.py."""
import ctypes as C

import pytest
import torch

from conftest import rel_l2
from oracle import svd_unet_ref as O, weights

pytestmark = pytest.mark.gpu


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _table(entries, cls, gpu):
    arr = (cls * len(entries))(*entries)
    d = torch.empty(C.sizeof(arr), dtype=torch.uint8, device=gpu)
    d.copy_(torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8))
    return d


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_pack_weights_all_forms(gpu, dtype):
    """One launch, six parameters: Linear (ragged), stacked q|k|v, 3x3 convolution with padded channels (stride-1 dgrad form
    and the stride-2 W^T form), (3,1,1) convolution — against the packing lambdas of autograd_ops / packing.py."""
    from gcd_amd import _lib, autograd_ops as A, packing
    g = torch.Generator().manual_seed(3)
    lin = torch.randn(100, 72, generator=g).to(gpu)
    q, k, v = (torch.randn(64, 64, generator=g).to(gpu) for _ in range(3))
    c3 = torch.randn(40, 8, 3, 3, generator=g).to(gpu)          # Cin 8 -> 64, Cout 40 -> 64
    c3b = torch.randn(64, 96, 3, 3, generator=g).to(gpu)
    t3 = torch.randn(64, 32, 3, 1, 1, generator=g).to(gpu)
    entries, outs = [], []

    def entry(src, N, Cc, taps, dst_f=None, f=(0, 0), dst_t=None, t=(0, 0), mirror=0):
        e = _lib.PackEntry()
        e.src, e.N, e.C, e.taps, e.mirror = src.data_ptr(), N, Cc, taps, mirror
        e.dst_f = 0 if dst_f is None else dst_f.data_ptr()
        e.dst_t = 0 if dst_t is None else dst_t.data_ptr()
        e.f_ns, e.f_ts = f
        e.t_cs, e.t_ts = t
        e.tiles_c = (Cc + 31) // 32
        entries.append(e)
    z = lambda *s: torch.zeros(*s, dtype=dtype, device=gpu)       # noqa: E731
    lf, lt = z(100, 72), z(72, 100)
    entry(lin, 100, 72, 1, lf, (72, 0), lt, (100, 0))
    qf, qt = z(192, 64), z(64, 192)
    for i, w in enumerate((q, k, v)):
        entry(w, 64, 64, 1, qf[64 * i:], (64, 0), qt[:, 64 * i:], (192, 0))
    cf, cd = z(64, 9 * 64), z(64, 9 * 64)
    entry(c3, 40, 8, 9, cf, (9 * 64, 64), cd, (9 * 64, 64), 1)
    cbf, cbt = z(64, 9 * 96), z(9 * 96, 64)
    entry(c3b, 64, 96, 9, cbf, (9 * 96, 96), cbt, (64, 96 * 64), 0)
    tf, td = z(64, 96), z(32, 192)
    entry(t3, 64, 32, 3, tf, (96, 32), td, (192, 64), 1)
    t0 = 0
    for e in entries:
        e.tile0 = t0
        t0 += ((e.N + 31) // 32) * e.tiles_c
    tab = _table(entries, _lib.PackEntry, gpu)
    _lib.check_train(_lib.load_train().gcd_train_pack_weights(tab.data_ptr(), len(entries), t0, int(dtype == torch.bfloat16),
                                                              _stream()), "pack")
    torch.cuda.synchronize()
    assert torch.equal(lf, lin.to(dtype)) and torch.equal(lt, lin.to(dtype).t())
    assert torch.equal(qf, torch.cat([q, k, v]).to(dtype)) and torch.equal(qt, torch.cat([q, k, v]).to(dtype).t())
    assert torch.equal(cf, packing.pack_conv3x3(c3, 64, 64, dtype))
    assert torch.equal(cd, A._pack_c3_dgrad(dtype, 64, 64)(c3))
    assert torch.equal(cbf, packing.pack_conv3x3(c3b, 96, 64, dtype))
    assert torch.equal(cbt, packing.pack_conv3x3(c3b, 96, 64, dtype).t())
    assert torch.equal(tf, packing.pack_conv_t3(t3, dtype)) and torch.equal(td, A._pack_t3_dgrad(dtype)(t3))


@pytest.mark.parametrize("taps,N,Nr,Cp,Cr", [(9, 64, 40, 64, 8), (9, 128, 128, 64, 64), (3, 64, 64, 96, 96), (1, 96, 96, 200, 200)])
def test_weight_gradient_in_parameter_layout(gpu, taps, N, Nr, Cp, Cr):
    """gcd_wgrad_tr_f16_ex: dW = dY^T X written as [N_real][C_real][taps], cropped — against the fp32 contraction
    permuted by torch; accumulate adds onto the destination."""
    from gcd_amd import _lib
    g = torch.Generator().manual_seed(7)
    M, K = 1000, taps * Cp
    dy = (torch.randn(M, N, generator=g) * 0.5).half().to(gpu)
    x = torch.randn(M, K, generator=g).half().to(gpu)
    ref = (dy.float().cpu().t() @ x.float().cpu()).reshape(N, taps, Cp).permute(0, 2, 1)[:Nr, :Cr].contiguous()
    lib = _lib.load_train()
    dst = torch.full((Nr, Cr, taps), 7.0, device=gpu)
    scratch = torch.empty(int(lib.gcd_wgrad_tr_scratch_floats(M, N, K)), device=gpu)
    for acc in (0, 1):
        _lib.check_train(lib.gcd_wgrad_tr_f16_ex(dy.data_ptr(), N, x.data_ptr(), K, M, N, K, 0, dst.data_ptr(), Cr, taps, Nr, Cr,
                                                 acc, scratch.data_ptr(), scratch.numel(), _stream()), "wgrad_ex")
        torch.cuda.synchronize()
        assert rel_l2(dst, ref * (1 + acc)) < 1e-4


def test_blend_forward_and_backward(gpu):
    from gcd_amd import _lib
    g = torch.Generator().manual_seed(9)
    frames, rows, Cc = 6, 37, 64
    M = frames * rows
    xs, xt, dy = (torch.randn(M, Cc, generator=g) for _ in range(3))
    a = torch.rand(frames, generator=g)
    a[2] = 1.0
    ar = a.repeat_interleave(rows)[:, None]
    lib = _lib.load_train()
    G = lambda t: t.to(gpu).contiguous()      # noqa: E731
    xsg, xtg, dyg, ag = G(xs), G(xt), G(dy), G(a)
    y = torch.empty(M, Cc, device=gpu)
    _lib.check_train(lib.gcd_blend_fwd_f32(xsg.data_ptr(), Cc, xtg.data_ptr(), Cc, ag.data_ptr(), M, Cc, rows, y.data_ptr(), Cc,
                                           _stream()), "blend_fwd")
    assert rel_l2(y, ar * xs + (1 - ar) * xt) < 1e-6
    dxs, dxt, dal = torch.empty(M, Cc, device=gpu), torch.empty(M, Cc, device=gpu), torch.zeros(frames, device=gpu)
    _lib.check_train(lib.gcd_blend_bwd_f32(dyg.data_ptr(), Cc, xsg.data_ptr(), Cc, xtg.data_ptr(), Cc, ag.data_ptr(), M, Cc, rows,
                                           dxs.data_ptr(), Cc, 0, dxt.data_ptr(), Cc, dal.data_ptr(), _stream()), "blend_bwd")
    assert rel_l2(dxs, ar * dy) < 1e-6 and rel_l2(dxt, (1 - ar) * dy) < 1e-6
    assert rel_l2(dal, (dy * (xs - xt)).reshape(frames, -1).double().sum(1)) < 1e-5


def test_grouped_few_row_linears(gpu):
    """gcd_smallm_fwd / _dgrad / _wgrad on a table of four problems (different N, K, M; SiLU on the input or not; one
    column-slice input) against fp64 torch.autograd."""
    from gcd_amd import _lib
    g = torch.Generator().manual_seed(11)
    shapes = [(28, 1280, 320, True), (28, 100, 72, False), (2, 64, 1280, False), (7, 36, 260, True)]   # (M, N, K, silu)
    # (more than 32 rows — 8 clips of 14 frames — go through train_plan._SmallGroup, which cuts forward / dgrad into 32-row
    #  problems; the wgrad kernel walks the rows itself: test_planned_engine_many_frames)
    lib = _lib.load_train()
    wide = torch.randn(28, 400, generator=g)
    items = []
    for i, (M, N, K, silu) in enumerate(shapes):
        x = wide[:M, 16:16 + K] if i == 1 else torch.randn(M, K, generator=g)
        items.append(dict(x=x, W=torch.randn(N, K, generator=g) / K ** 0.5, b=torch.randn(N, generator=g), silu=silu,
                          dy=torch.randn(M, N, generator=g)))
    wide_g = wide.to(gpu)
    for i, it in enumerate(items):
        it["xg"] = wide_g[:it["x"].shape[0], 16:16 + it["x"].shape[1]] if i == 1 else it["x"].to(gpu)
        for k2 in ("W", "b", "dy"):
            it[k2 + "g"] = it[k2].to(gpu)
        M, K = it["x"].shape
        N = it["W"].shape[0]
        it["yg"] = torch.empty(M, N, device=gpu)
        it["dxg"] = torch.zeros(M, K, device=gpu)
        it["dWg"] = torch.empty(N, K, device=gpu)
        it["dbg"] = torch.empty(N, device=gpu)

    def table(mode):
        probs, b0 = [], 0
        for it in items:
            M, K = it["x"].shape
            N = it["W"].shape[0]
            p = _lib.SmallmProblem()
            p.x, p.ldx, p.W, p.b = it["xg"].data_ptr(), it["xg"].stride(0), it["Wg"].data_ptr(), it["bg"].data_ptr()
            p.M, p.N, p.K, p.block0 = M, N, K, b0
            if mode == "fwd":
                p.y, p.ldy, p.flags = it["yg"].data_ptr(), N, int(it["silu"])
                b0 += (N + 15) // 16
            else:
                p.y, p.ldy = it["dyg"].data_ptr(), N
                p.dx, p.lddx, p.dW, p.db = it["dxg"].data_ptr(), K, it["dWg"].data_ptr(), it["dbg"].data_ptr()
                p.flags = int(it["silu"]) | (4 if mode == "dgrad" else 0)
                b0 += ((K + 255) // 256) * ((N + 63) // 64)
            probs.append(p)
        return _table(probs, _lib.SmallmProblem, gpu), len(probs), b0
    for mode, fn in (("fwd", lib.gcd_smallm_fwd), ("dgrad", lib.gcd_smallm_dgrad), ("wgrad", lib.gcd_smallm_wgrad)):
        tab, n, blocks = table(mode)
        _lib.check_train(fn(tab.data_ptr(), n, blocks, _stream()), mode)
    torch.cuda.synchronize()
    for it in items:
        x = it["x"].double().requires_grad_(True)
        W, b = it["W"].double().requires_grad_(True), it["b"].double().requires_grad_(True)
        a = torch.nn.functional.silu(x) if it["silu"] else x
        y = a @ W.t() + b
        y.backward(it["dy"].double())
        assert rel_l2(it["yg"], y.detach()) < 1e-5
        assert rel_l2(it["dxg"], x.grad) < 1e-5
        assert rel_l2(it["dWg"], W.grad) < 1e-5 and rel_l2(it["dbg"], b.grad) < 1e-5


def _tiny(gpu, salt=0):
    from gcd_amd.video_model import VideoUNet
    with torch.device("meta"):
        net = VideoUNet(**O.TINY.as_reference_kwargs())
    sd = weights.synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, salt)
    net = net.to_empty(device=gpu)
    net.load_state_dict(sd)
    return net.train()


@pytest.mark.parametrize("ckpt", [True, False])
def test_planned_engine_matches_autograd_engine(gpu, ckpt):
    """The TINY VideoUNet, forward + backward, planned vs autograd engine (same HIP operators; the planned engine runs the
    few-row Linears in fp32 where the autograd engine rounds them to fp16, hence the 2e-3 bar on their neighbourhood):
    output, every parameter gradient, the set of parameters reached; with an image-only frame so that the blenders' masks
    are exercised; with and without activation checkpointing."""
    from gcd_amd import autograd_ops as A, training as TR
    from gcd_amd.train_plan import unet_forward_planned
    net = _tiny(gpu, salt=5)
    cfg = O.TINY
    g = torch.Generator().manual_seed(21)
    T, H, W = 4, 16, 16
    x = torch.randn(2 * T, 8, H, W, generator=g).to(gpu)
    ts = torch.linspace(-1.0, 1.5, 2 * T).to(gpu)
    ctx = torch.randn(2 * T, 1, cfg.context_dim, generator=g).to(gpu)
    y = torch.randn(2 * T, cfg.adm_in_channels + cfg.aux_emb_dim, generator=g).clamp(-1, 1).to(gpu)
    ioi = torch.zeros(2, T, device=gpu)
    ioi[1, 2] = 1.0
    tgt = torch.randn(2 * T, 4, H, W, generator=g).to(gpu)
    res = {}
    for name, fn in (("autograd", TR.unet_forward_train), ("planned", unet_forward_planned)):
        A.PACK.clear()
        for p in net.parameters():
            p.grad = None
        out = fn(net, x, ts, ctx, y, T, ioi, use_checkpoint=ckpt)
        ((out - tgt) ** 2).mean().mul(64.0).backward()
        torch.cuda.synchronize()
        res[name] = (out.detach().clone(), {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None})
    (oa, ga), (op, gp) = res["autograd"], res["planned"]
    e = rel_l2(op, oa)
    print(f"planned vs autograd: output {e:.2e}")
    assert e < 3e-3       # (measured 1.4e-3: the embedding / emb_layers / cross-attention Linears are fp32 here, fp16 there)
    dead_a = {n for n, _ in net.named_parameters() if n not in ga}
    for n, _ in net.named_parameters():
        if n in dead_a:
            assert n not in gp or float(gp[n].abs().max()) == 0.0, n
    worst = ("", 0.0)
    num = den = 0.0
    for n, ref in ga.items():
        assert n in gp, n
        num += float((gp[n].double() - ref.double()).pow(2).sum())
        den += float(ref.double().pow(2).sum())
        if ref.numel() >= 64:
            en = rel_l2(gp[n], ref)
            if en > worst[1]:
                worst = (n, en)
    tot = (num / den) ** 0.5
    print(f"planned vs autograd: {len(ga)} gradients, global {tot:.2e}, worst {worst[0]} {worst[1]:.2e}")
    assert tot < 5e-3 and worst[1] < 2e-2


def test_planned_engine_under_hipgraphs_matches_eager(gpu):
    """GraphedPlan: two eager steps, then the forward and backward passes are captured as hipGraphs and replayed.  Five
    steps with fresh inputs and a parameter update in between (written through raw storage like gcd_adam_step_multi, so
    that only the in-graph weight pack can see it; a fixed perturbation rather than Adam, whose first steps are
    lr * sign(g) and turn a last-bit gradient difference into a different trajectory): the replayed steps' outputs and
    gradients equal an eager run's from the same parameters."""
    from gcd_amd import autograd_ops as A, train_plan as TP, training as TR
    cfg = O.TINY
    T, H, W = 4, 16, 16
    g = torch.Generator().manual_seed(33)
    steps = []
    for _ in range(5):
        steps.append(dict(x=torch.randn(2 * T, 8, H, W, generator=g).to(gpu), ts=torch.rand(2 * T, generator=g).to(gpu) * 2 - 1,
                          ctx=torch.randn(2 * T, 1, cfg.context_dim, generator=g).to(gpu),
                          y=torch.randn(2 * T, cfg.adm_in_channels + cfg.aux_emb_dim, generator=g).clamp(-1, 1).to(gpu),
                          tgt=torch.randn(2 * T, 4, H, W, generator=g).to(gpu)))
    ioi = torch.zeros(2, T, device=gpu)
    runs = {}
    for graph in (False, True):
        TP.set_use_graph(graph)
        try:
            net = _tiny(gpu, salt=7)
            A.PACK.clear()
            outs = []
            gp = torch.Generator(device=gpu).manual_seed(5)
            for st in steps:
                for p in net.parameters():
                    p.grad = None
                out = TP.unet_forward_planned(net, st["x"], st["ts"], st["ctx"], st["y"], T, ioi, use_checkpoint=True)
                ((out - st["tgt"]) ** 2).mean().mul(64.0).backward()
                outs.append((out.detach().clone(), {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}))
                with torch.no_grad():
                    for p in net.parameters():
                        p.data.view(-1).add_(torch.randn(p.numel(), generator=gp, device=gpu) * 0.02 * float(p.data.std() if p.numel() > 1 else 1.0))
                A.PACK.clear()         # what AdamHIP.step does after writing the parameters through raw pointers
            torch.cuda.synchronize()
            if graph:
                assert TP.plan_for(net).graphed.mode == "graph", "the steps after the warm-up must have been replays"
            runs[graph] = outs
        finally:
            TP.set_use_graph(False)
    for i, ((oe, ge), (og, gg)) in enumerate(zip(runs[False], runs[True])):
        assert rel_l2(og, oe) < 1e-4, f"step {i}: output {rel_l2(og, oe):.2e}"
        assert ge.keys() == gg.keys()
        worst = max(rel_l2(gg[n], ge[n]) for n in ge if ge[n].numel() >= 64)
        print(f"step {i}: graph vs eager output {rel_l2(og, oe):.1e}, worst gradient {worst:.1e}")
        assert worst < 2e-3      # (atomic summation order differs run to run; a stale pack or input would be O(1))


def test_planned_engine_many_frames(gpu):
    """5 clips x 8 frames = 40 frames (> the 32 rows one few-row problem holds): planned vs autograd engine."""
    from gcd_amd import autograd_ops as A, training as TR
    from gcd_amd.train_plan import unet_forward_planned
    net = _tiny(gpu, salt=9)
    cfg = O.TINY
    g = torch.Generator().manual_seed(41)
    T, clips, H, W = 8, 5, 8, 8
    n = T * clips
    x = torch.randn(n, 8, H, W, generator=g).to(gpu)
    ts = torch.linspace(-1.0, 1.5, n).to(gpu)
    ctx = torch.randn(n, 1, cfg.context_dim, generator=g).to(gpu)
    y = torch.randn(n, cfg.adm_in_channels + cfg.aux_emb_dim, generator=g).clamp(-1, 1).to(gpu)
    ioi = torch.zeros(clips, T, device=gpu)
    tgt = torch.randn(n, 4, H, W, generator=g).to(gpu)
    res = {}
    for name, fn in (("autograd", TR.unet_forward_train), ("planned", unet_forward_planned)):
        A.PACK.clear()
        for p in net.parameters():
            p.grad = None
        out = fn(net, x, ts, ctx, y, T, ioi, use_checkpoint=False)
        ((out - tgt) ** 2).mean().mul(64.0).backward()
        torch.cuda.synchronize()
        res[name] = (out.detach().clone(), {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None})
    (oa, ga), (op, gp) = res["autograd"], res["planned"]
    assert rel_l2(op, oa) < 3e-3
    num = sum(float((gp[k].double() - v.double()).pow(2).sum()) for k, v in ga.items())
    den = sum(float(v.double().pow(2).sum()) for v in ga.values())
    print(f"40 frames: output {rel_l2(op, oa):.2e}, gradients {(num / den) ** 0.5:.2e}")
    assert (num / den) ** 0.5 < 5e-3


def test_planned_engine_gradient_accumulation(gpu):
    """Two micro-batches without zero_grad() in between (accumulate_grad_batches, main.py:950): the second backward finds
    the plan's own views in .grad and ADDS — equal to the sum of the two batches' separate gradients; zero_grad() (set to
    None) starts over."""
    from gcd_amd import autograd_ops as A
    from gcd_amd.train_plan import unet_forward_planned
    net = _tiny(gpu, salt=13)
    cfg = O.TINY
    g = torch.Generator().manual_seed(51)
    T, H, W = 4, 8, 8

    def batch():
        return (torch.randn(2 * T, 8, H, W, generator=g).to(gpu), torch.linspace(-1.0, 1.0, 2 * T).to(gpu),
                torch.randn(2 * T, 1, cfg.context_dim, generator=g).to(gpu),
                torch.randn(2 * T, cfg.adm_in_channels + cfg.aux_emb_dim, generator=g).clamp(-1, 1).to(gpu),
                torch.randn(2 * T, 4, H, W, generator=g).to(gpu))
    ioi = torch.zeros(2, T, device=gpu)
    b1, b2 = batch(), batch()
    A.PACK.clear()

    def run(b):
        out = unet_forward_planned(net, b[0], b[1], b[2], b[3], T, ioi, use_checkpoint=True)
        ((out - b[4]) ** 2).mean().mul(64.0).backward()
    sep = []
    for b in (b1, b2):
        for p in net.parameters():
            p.grad = None
        run(b)
        sep.append({n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None})
    for p in net.parameters():
        p.grad = None
    run(b1)
    run(b2)          # no zero_grad(): accumulates
    torch.cuda.synchronize()
    worst = 0.0
    for n, p in net.named_parameters():
        if n in sep[0]:
            want = sep[0][n] + sep[1][n]
            if want.numel() >= 64 and float(want.abs().max()) > 0:
                worst = max(worst, rel_l2(p.grad, want))
    print(f"accumulated vs summed gradients: worst {worst:.1e}")
    assert worst < 1e-3


@pytest.mark.parametrize("bf16", [False, True])
def test_conv_weight_gradients_without_im2col(gpu, bf16):
    """gcd_wgrad_conv_tr_f16: the weight gradient of a stride-1 3x3 convolution and of the (3,1,1) temporal convolution with
    the X operand gathered per tap inside the kernel (no im2col tensor), written in the parameter's layout, cropped —
    against torch.nn.grad.conv{2,3}d_weight in fp32 on the 16-bit-rounded operands; ragged token counts, padded channels,
    more than one token slice, accumulate."""
    from gcd_amd import _lib
    dt = torch.bfloat16 if bf16 else torch.float16
    lib = _lib.load_train()
    g = torch.Generator().manual_seed(61)

    def run(dy_tok, x_tok, N, Cp, conv, Nr, Cr, **geo):
        M = dy_tok.shape[0]
        taps = 9 if conv == 1 else 3
        dst = torch.full((Nr, Cr, taps), 3.0, device=gpu)
        scratch = torch.empty(int(lib.gcd_wgrad_tr_scratch_floats(M, N, taps * Cp)), device=gpu)
        outs = []
        for acc in (0, 1):
            _lib.check_train(lib.gcd_wgrad_conv_tr_f16(
                dy_tok.data_ptr(), dy_tok.stride(0), x_tok.data_ptr(), x_tok.stride(0), M, N, Cp, conv, geo.get("Ho", 0),
                geo.get("Wo", 0), geo.get("T", 0), geo.get("HW", 0), int(bf16), dst.data_ptr(), Nr, Cr, acc, scratch.data_ptr(),
                scratch.numel(), _stream()), "wgrad_conv")
            torch.cuda.synchronize()
            outs.append(dst.clone())
        return outs
    # ---- 3x3: frames x H x W tokens, Cin 8 real of Cp 64 (the UNet's first convolution), Cout 40 ----
    for frames, H, W, Cin, Cp, Cout, N in [(3, 7, 9, 8, 64, 40, 40), (5, 16, 24, 72, 72, 64, 64), (2, 33, 48, 64, 64, 4, 32)]:
        x = (torch.randn(frames, Cin, H, W, generator=g)).to(dt).float()
        dy = (torch.randn(frames, Cout, H, W, generator=g) * 0.5).to(dt).float()
        ref = torch.nn.grad.conv2d_weight(x, (Cout, Cin, 3, 3), dy, padding=1).reshape(Cout, Cin, 9)
        xt = torch.zeros(frames * H * W, Cp)
        xt[:, :Cin] = x.permute(0, 2, 3, 1).reshape(-1, Cin)
        dyt = torch.zeros(frames * H * W, N)
        dyt[:, :Cout] = dy.permute(0, 2, 3, 1).reshape(-1, Cout)
        o0, o1 = run(dyt.to(dt).to(gpu), xt.to(dt).to(gpu), N, Cp, 1, Cout, Cin, Ho=H, Wo=W)
        assert rel_l2(o0, ref) < 2e-4, (frames, H, W, rel_l2(o0, ref))
        assert rel_l2(o1, 2 * ref) < 2e-4
    # ---- (3,1,1): clips x T x HW tokens ----
    for clips, T, HW, Cc, Cout in [(2, 5, 12, 64, 64), (3, 14, 35, 72, 40)]:
        x = torch.randn(clips, Cc, T, HW, 1, generator=g).to(dt).float()
        dy = (torch.randn(clips, Cout, T, HW, 1, generator=g) * 0.5).to(dt).float()
        ref = torch.nn.grad.conv3d_weight(x, (Cout, Cc, 3, 1, 1), dy, padding=(1, 0, 0)).reshape(Cout, Cc, 3)
        xt = x.permute(0, 2, 3, 4, 1).reshape(-1, Cc)
        dyt = dy.permute(0, 2, 3, 4, 1).reshape(-1, Cout)
        o0, o1 = run(dyt.to(dt).to(gpu).contiguous(), xt.to(dt).to(gpu).contiguous(), Cout, Cc, 2, Cout, Cc, T=T, HW=HW)
        assert rel_l2(o0, ref) < 2e-4, (clips, T, HW, rel_l2(o0, ref))
        assert rel_l2(o1, 2 * ref) < 2e-4


def _tiny_step_inputs(gpu, seed, T=4, H=16, W=16):
    cfg = O.TINY
    g = torch.Generator().manual_seed(seed)
    return dict(x=torch.randn(2 * T, 8, H, W, generator=g).to(gpu), ts=torch.linspace(-1.0, 1.5, 2 * T).to(gpu),
                ctx=torch.randn(2 * T, 1, cfg.context_dim, generator=g).to(gpu),
                y=torch.randn(2 * T, cfg.adm_in_channels + cfg.aux_emb_dim, generator=g).clamp(-1, 1).to(gpu),
                tgt=torch.randn(2 * T, 4, H, W, generator=g).to(gpu), ioi=torch.zeros(2, T, device=gpu), T=T)


def test_two_plans_alternating_in_one_process_do_not_interfere(gpu):
    """Round 6 (no process-global state in the training engines): two networks, each with its own TrainPlan, stepped
    ALTERNATELY in one process — forward A, forward B, backward A, backward B, with a gradient listener on A only — give,
    bit for bit, the gradients each gives when it runs alone; the listener hears A's parameters and none of B's."""
    from gcd_amd import autograd_ops as A, train_plan as TP
    from gcd_amd.train_plan import unet_forward_planned
    na, nb = _tiny(gpu, salt=5), _tiny(gpu, salt=6)
    ia, ib = _tiny_step_inputs(gpu, 41), _tiny_step_inputs(gpu, 42)

    def alone(net, s):
        for p in net.parameters():
            p.grad = None
        out = unet_forward_planned(net, s["x"], s["ts"], s["ctx"], s["y"], s["T"], s["ioi"], use_checkpoint=False)
        ((out - s["tgt"]) ** 2).mean().mul(64.0).backward()
        torch.cuda.synchronize()
        return out.detach().clone(), {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}
    oa, ga = alone(na, ia)
    ob, gb = alone(nb, ib)
    # run-to-run spread of a network stepped ALONE (the few reductions that use atomics are order-dependent): the bar for
    # "the same gradients" below is bit equality where a lone repeat is bit-equal, else that spread
    _, ga2 = alone(na, ia)
    spread = max((rel_l2(ga2[n], ga[n]) for n in ga if ga[n].numel() >= 64), default=0.0)
    lone_bit_equal = all(torch.equal(ga2[n], ga[n]) for n in ga)
    print(f"a lone repeat: bit-equal {lone_bit_equal}, worst tensor rel-L2 {spread:.2e}")
    heard = []
    TP.add_grad_listener(list(na.parameters()), heard.append)
    try:
        for net in (na, nb):
            for p in net.parameters():
                p.grad = None
        out_a = unet_forward_planned(na, ia["x"], ia["ts"], ia["ctx"], ia["y"], ia["T"], ia["ioi"], use_checkpoint=False)
        out_b = unet_forward_planned(nb, ib["x"], ib["ts"], ib["ctx"], ib["y"], ib["T"], ib["ioi"], use_checkpoint=False)
        ((out_a - ia["tgt"]) ** 2).mean().mul(64.0).backward()
        ((out_b - ib["tgt"]) ** 2).mean().mul(64.0).backward()
        torch.cuda.synchronize()
    finally:
        TP.remove_grad_listener(list(na.parameters()), heard.append)
    assert torch.equal(out_a, oa) and torch.equal(out_b, ob)
    worst = ("", 0.0)
    for net, ref in ((na, ga), (nb, gb)):
        for n, p in net.named_parameters():
            if n in ref:
                if lone_bit_equal:
                    assert torch.equal(p.grad, ref[n]), n
                elif ref[n].numel() >= 64:
                    e = rel_l2(p.grad, ref[n])
                    worst = max(worst, (n, e), key=lambda t: t[1])
    print(f"alternating vs alone: worst tensor {worst[0]} rel-L2 {worst[1]:.2e}")
    assert worst[1] <= max(10 * spread, 1e-6)
    ids_a, ids_b = {id(p) for p in na.parameters()}, {id(p) for p in nb.parameters()}
    assert heard and all(id(p) in ids_a for p in heard) and not any(id(p) in ids_b for p in heard)
    assert A._sink() is None


def test_planned_engine_inputs_that_need_gradients_and_no_grad_forwards(gpu):
    """Round-5 review: (1) the planned pass computes parameter gradients only — an input that requires grad (a conditioner
    trained through `vector`, as the kubric configs train SphericalEmbedder.proj) must not silently get none: the call runs
    on the operator-level engine and y.grad equals that engine's; (2) a forward under torch.no_grad() keeps nothing for a
    backward pass; (3) a backward of a forward that is no longer the latest raises instead of using the wrong activations."""
    import warnings
    from gcd_amd import training as TR
    from gcd_amd.train_plan import plan_for, unet_forward_planned
    net = _tiny(gpu, salt=5)
    s = _tiny_step_inputs(gpu, 43)
    grads = {}
    for name, fn in (("planned-entry", unet_forward_planned), ("autograd", TR.unet_forward_train)):
        y = s["y"].clone().requires_grad_()
        for p in net.parameters():
            p.grad = None
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            out = fn(net, s["x"], s["ts"], s["ctx"], y, s["T"], s["ioi"], use_checkpoint=True)
        ((out - s["tgt"]) ** 2).mean().mul(64.0).backward()
        torch.cuda.synchronize()
        assert y.grad is not None and float(y.grad.abs().max()) > 0
        grads[name] = y.grad.clone()
    assert torch.equal(grads["planned-entry"], grads["autograd"])
    plan = plan_for(net, True)
    with torch.no_grad():
        o1 = unet_forward_planned(net, s["x"], s["ts"], s["ctx"], s["y"], s["T"], s["ioi"], use_checkpoint=True)
    assert plan._trace is None and not o1.requires_grad
    out = unet_forward_planned(net, s["x"], s["ts"], s["ctx"], s["y"], s["T"], s["ioi"], use_checkpoint=True)
    with torch.no_grad():
        unet_forward_planned(net, s["x"], s["ts"], s["ctx"], s["y"], s["T"], s["ioi"], use_checkpoint=True)
    with pytest.raises(RuntimeError, match="no longer the network's latest"):
        ((out - s["tgt"]) ** 2).mean().backward()
