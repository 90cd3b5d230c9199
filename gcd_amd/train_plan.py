"""The fine-tune step of GCD as a PLANNED pass (round 5) — BASELINE.json cfg4; SURVEY.md §8a a23, §8(f)-2.

`gcd_amd.training.unet_forward_train` (rounds 2-4) runs VideoUNet.forward as a torch.autograd graph whose nodes are the
HIP operators of `autograd_ops`: correct, and 13 750 launches per step of which thousands are torch's own fills, casts,
copies, transposes, flips, adds and lerps around the tape (profiles/r04_train_kernel_stats.txt).  This module runs the same
operators WITHOUT a tape:

  * `TrainPlan.forward` walks the network unit by unit (VideoResBlock, SpatialVideoTransformer, the plain convolutions:
    video_model.py:461-540) and keeps, under activation checkpointing, only each unit's input — the reference's own
    checkpoint sites (openaimodel.py:326-329, attention.py:544-546, video_attention.py:104-105);
  * `TrainPlan.backward` walks the units in reverse: re-run the unit keeping its operator contexts, then the unit's
    hand-written backward — every gradient routed explicitly, weight gradients written by the transposing-read kernel
    STRAIGHT into the parameter's `.grad` in the parameter's own layout (`gcd_wgrad_tr_f16_ex`: no padded temporary, no
    permuted copy), residual sums folded where a kernel can take them;
  * every fp32 parameter becomes its 16-bit GEMM operand forms (forward, transposed, tap-mirrored dgrad) in ONE launch per
    optimizer step (`gcd_train_pack_weights`) — the forms land in `autograd_ops.PACK`, so the operator code is shared with
    the autograd path and cannot drift from it;
  * the few-row fp32 Linears (emb_layers of the 44 ResBlocks, the one-key cross-attention chains, time_pos_embed, the
    embedding MLPs: ~700 launches per step before) run as grouped launches (`gcd_smallm_fwd / _dgrad / _wgrad`);
  * AlphaBlender and its backward are one kernel each (`gcd_blend_fwd_f32` / `gcd_blend_bwd_f32`).

`unet_forward_planned` puts the whole network behind ONE autograd node, so `TrainDenoiser`, `StandardDiffusionLoss` and
`loss.backward()` (loss.py:115-273) stay what they are.  Parity: tests/test_train_plan_gpu.py (every gradient against the
autograd path on the same kernels, and the cfg4 goldens made by the unmodified reference).
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, Dict, List, Optional

import torch
import torch.nn as nn

from . import _lib
from . import autograd_ops as A
from . import ops
from .video_model import Downsample, SpatialVideoTransformer, Upsample, VideoResBlock, VideoUNet

_f32 = torch.float32
# Gradient listeners: callables fn(param) told the moment a parameter's gradient is final (training.GradBucketer registers
# its _on_grad: the planned engine assigns .grad itself, so torch's post-accumulate hooks never fire).  Round 6: they hang
# on the PARAMETER (`param._gcd_grad_listeners`), not in a process-wide list — a second network, or a second bucketer on
# other parameters, is never called for gradients that are not its own, and a listener dies with its parameters.
def add_grad_listener(params, fn: Callable) -> None:
    for p in params:
        ls = p.__dict__.setdefault("_gcd_grad_listeners", [])
        if fn not in ls:
            ls.append(fn)


def remove_grad_listener(params, fn: Callable) -> None:
    for p in params:
        ls = p.__dict__.get("_gcd_grad_listeners")
        if ls and fn in ls:
            ls.remove(fn)
            if not ls:
                del p.__dict__["_gcd_grad_listeners"]


def _notify(plan, p) -> None:
    if plan.on_param_grad is not None:
        plan.on_param_grad(p)
    if plan.listeners_enabled:
        for fn in tuple(p.__dict__.get("_gcd_grad_listeners", ())):
            fn(p)


def _stream():
    return torch.cuda.current_stream().cuda_stream


class _Ctx:
    """What `autograd_ops.Fused.forward / backward` (and the attention Functions) need of a torch ctx — their static methods
    are called directly, there is no tape."""
    needs_input_grad = ()

    def save_for_backward(self, *t):
        self.saved_tensors = t


# ------------------------------------------------------------------------------------------------------------------------
# operator wrappers: forward returns (y, ctx); backward takes (ctx, dy)
# ------------------------------------------------------------------------------------------------------------------------
def _fused_fwd(kind, x, params, geo=None, norm=None, residual=None, rowvec=None):
    """norm: None | ("ln", module, eps) | ("gn", module, rows_per_inst, eps, silu) | ("geglu",); rowvec: (vec, rows)."""
    spec = dict(kind=kind, geo=geo, norm=None, rows_per_vec=None if rowvec is None else rowvec[1])
    gamma = beta = None
    if norm is not None and norm[0] == "geglu":
        spec["norm"] = ("geglu",)
    elif norm is not None:
        gamma, beta = norm[1].weight, norm[1].bias
        spec["norm"] = (norm[0],) + tuple(norm[2:])
    ctx = _Ctx()
    ctx.params = params
    ctx.norm_mod = None if gamma is None else norm[1]
    y = A.Fused.forward(ctx, spec, x, residual, None if rowvec is None else rowvec[0], gamma, beta, *params)
    return y, ctx


def _fused_fwd_from16(kind, x16, params, residual=None, rowvec=None):
    """A Linear whose input already exists as the 16-bit operand (an attention core's fp16 output): no fp32 image of it, no
    cast back (autograd_ops' fp16 pass-through, which the autograd engine measured slower because it holds the tensors)."""
    x16._gcd_f16 = (x16, x16._version)
    try:
        with A.f16_passthrough(True):
            return _fused_fwd(kind, x16, params, residual=residual, rowvec=rowvec)
    finally:
        del x16._gcd_f16          # (the tag refers to the tensor itself: no reference cycle left behind)


def _fused_bwd(plan: "TrainPlan", ctx: _Ctx, dy: torch.Tensor, need_x: bool = True, need_vec: bool = True, dx_add=None):
    """-> (dx | None, d_vec | None).  Parameter gradients go to `plan.sink`.  dx_add: another gradient of the same input
    (the residual branch's), added to dx — inside the LayerNorm / GroupNorm backward kernel when the node has one."""
    has_res, has_vec, has_bias = ctx.has
    norm = ctx.spec.get("norm")
    fused_add = dx_add is not None and norm is not None and norm[0] in ("ln", "gn")
    ctx.dx_add = dx_add.contiguous() if fused_add else None
    params = ctx.params
    want = [p is not None and p.requires_grad for p in params]
    gm = ctx.norm_mod
    want_norm = gm is not None and gm.weight.requires_grad
    ctx.needs_input_grad = (False, need_x, has_res, has_vec and need_vec, want_norm, want_norm, *want)
    ctx.norm_dest = (plan.grad_dest(gm.weight), plan.grad_dest(gm.bias)) if want_norm else None
    bias = params[1] if len(params) == 2 else None
    ctx.bias_dest = plan.grad_dest(bias) if bias is not None and bias.requires_grad else None
    with A.grad_sink(plan):                # weight gradients are written in place (autograd_ops._sink_dest)
        out = A.Fused.backward(ctx, dy)
    dx, d_vec, dgamma, dbeta = out[1], out[3], out[4], out[5]
    if dx_add is not None and not fused_add and dx is not None:
        dx = dx.add_(dx_add)
    if want_norm:
        plan.sink(gm.weight, dgamma)
        plan.sink(gm.bias, dbeta)
    for p, g in zip(params, out[6:]):
        if p is not None and g is not None:
            plan.sink(p, g)
    return dx, d_vec


def _c3_geo(frames, Hi, Wi, stride=1, upsample=False):
    if upsample:
        Ho, Wo = 2 * Hi, 2 * Wi
    else:
        Ho, Wo = (Hi - 1) // stride + 1, (Wi - 1) // stride + 1
    return dict(frames=frames, Hi=Hi, Wi=Wi, Ho=Ho, Wo=Wo, stride=stride, upsample=int(upsample))


def _ln(m):
    return ("ln", m, 1e-5)


_QKV16 = dict(out16=True)      # q | k | v straight out of the GEMM in fp16 (autograd_ops._contract_fwd)


def zeros_or_new(plan, m, n):
    z = plan.zeros(m, n)
    return z if z is not None else torch.zeros(m, n, dtype=_f32, device=plan.device)


def _add(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """a + b in place on `a` when a is ours to overwrite (a fresh gradient tensor)."""
    return a.add_(b)


# ------------------------------------------------------------------------------------------------------------------------
# grouped few-row Linears
# ------------------------------------------------------------------------------------------------------------------------
class _SmallGroup:
    """A table of independent y = act(x) W^T + b problems of <= 32 rows (gcd_smallm_problem) and the launches over it."""

    def __init__(self, device, name=""):
        self.device = device
        self.name = name
        self.items: List[dict] = []
        self._tabs: Dict[str, tuple] = {}

    def add(self, x, lin: nn.Linear, y, silu_in=False, accumulate=False, dx=None, dx_silu=False):
        """y = act(x) lin.weight^T + lin.bias: x [M, K] fp32 (row-major view), y [M, N] fp32 output (accumulate: y += ...;
        never two problems of one launch onto the same y — they run concurrently).  dx: where the input gradient
        accumulates (zeroed by the caller; several problems may share it), or None when x needs no gradient."""
        w, b = lin.weight, lin.bias
        assert x.dtype == _f32 and y.dtype == _f32 and x.stride(1) == 1 and y.stride(1) == 1
        assert x.shape[1] % 4 == 0 and w.is_contiguous()
        self.items.append(dict(x=x, w=w, b=b, y=y, silu=silu_in, acc=accumulate, dx=dx, dx_silu=dx_silu))
        self._tabs.clear()
        return y

    def _table(self, mode: str, plan: "TrainPlan"):
        key = mode
        if key in self._tabs:
            return self._tabs[key]
        # the device table of this (stage, mode) from the previous step, when every pointer in it is unchanged: the few-row
        # tensors live in the plan's persistent buffers and arenas, whose bump allocation is deterministic
        # (everything a table entry is made of: operand / weight / bias / gradient-destination pointers, shapes, strides and
        #  flags — a table that is served from the cache must be the table that would be built)
        def _ptr(t):
            return 0 if t is None else t.data_ptr()
        wg = mode == "wgrad"
        sig = (mode, plan.accumulate if wg else False) + tuple(
            (_ptr(it["x"]), _ptr(it["y"]), _ptr(it.get("dy")), _ptr(it["dx"]), _ptr(it["w"]), _ptr(it["b"]),
             tuple(it["x"].shape), it["x"].stride(0), it["y"].stride(0), it["w"].shape[0],
             0 if it.get("dy") is None else it["dy"].stride(0), 0 if it["dx"] is None else it["dx"].stride(0),
             it["silu"], it["acc"], it["dx_silu"], it["w"].requires_grad, it["b"] is not None and it["b"].requires_grad,
             _ptr(plan.grad_dest(it["w"])) if wg and it["w"].requires_grad else 0,
             _ptr(plan.grad_dest(it["b"])) if wg and it["b"] is not None and it["b"].requires_grad else 0)
            for it in self.items)
        hit = plan._table_cache.get((self.name, mode))
        if hit is not None and hit[0] == sig:
            self._tabs[key] = hit[1]
            return hit[1]
        # a rebuild copies a host table to the device: not something a stream capture can hold (the host buffer is gone at
        # replay).  GraphedPlan captures only after warm-up steps with the same signature, which leave every table cached.
        if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
            raise RuntimeError(f"train_plan: the few-row problem table {self.name}/{mode} would be rebuilt inside a stream "
                               "capture (its pointers, shapes or flags differ from the warm-up steps)")
        probs, block0 = [], 0
        for it in self.items:
            x, w, y = it["x"], it["w"], it["y"]
            Mtot, K = x.shape
            N = w.shape[0]
            # forward and dgrad: rows are independent -> one problem per 32 rows; wgrad: one problem, the kernel walks the rows
            chunks = [(0, Mtot)] if mode == "wgrad" else [(m0, min(32, Mtot - m0)) for m0 in range(0, Mtot, 32)]
            for m0, M in chunks:
                p = _lib.SmallmProblem()
                p.x, p.ldx, p.W = x[m0:].data_ptr(), x.stride(0), w.data_ptr()
                p.M, p.N, p.K = M, N, K
                p.block0 = block0
                if mode == "fwd":
                    p.b = 0 if it["b"] is None else it["b"].data_ptr()
                    p.y, p.ldy = y[m0:].data_ptr(), y.stride(0)
                    p.flags = int(it["silu"]) | (4 if it["acc"] else 0)
                    block0 += (N + 15) // 16
                else:
                    dy = it["dy"]
                    p.y, p.ldy = dy[m0:].data_ptr(), dy.stride(0)
                    if mode == "dgrad":
                        if it["dx"] is None:
                            continue
                        p.dx, p.lddx = it["dx"][m0:].data_ptr(), it["dx"].stride(0)
                        p.flags = (1 if it["dx_silu"] else 0) | 4
                    else:
                        if not w.requires_grad:
                            continue
                        p.dW = plan.grad_dest(w).data_ptr()
                        bb = it["b"]
                        p.db = 0 if bb is None or not bb.requires_grad else plan.grad_dest(bb).data_ptr()
                        p.flags = int(it["silu"]) | (8 if plan.accumulate else 0)
                    block0 += ((K + 255) // 256) * ((N + 63) // 64)
                probs.append(p)
        if not probs:
            self._tabs[key] = (None, 0, 0)
        else:
            arr = (_lib.SmallmProblem * len(probs))(*probs)
            dev = torch.empty(C.sizeof(arr), dtype=torch.uint8, device=self.device)
            host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
            dev.copy_(host)
            self._tabs[key] = (dev, len(probs), block0)
        plan._table_cache[(self.name, mode)] = (sig, self._tabs[key])
        return self._tabs[key]

    def forward(self, plan):
        dev, n, blocks = self._table("fwd", plan)
        if n:
            _lib.check_train(_lib.load_train().gcd_smallm_fwd(dev.data_ptr(), n, blocks, _stream()), "gcd_smallm_fwd")

    def backward(self, plan, dys: List[torch.Tensor]):
        """dys[i]: gradient of item i's output (fp32 [M, N]); parameter gradients go to the flat buffer, input gradients
        accumulate into the items' dx."""
        for it, dy in zip(self.items, dys):
            assert dy.dtype == _f32 and dy.stride(1) == 1
            it["dy"] = dy
        self._tabs.pop("dgrad", None)
        self._tabs.pop("wgrad", None)
        lib = _lib.load_train()
        dev, n, blocks = self._table("dgrad", plan)
        if n:
            _lib.check_train(lib.gcd_smallm_dgrad(dev.data_ptr(), n, blocks, _stream()), "gcd_smallm_dgrad")
        dev, n, blocks = self._table("wgrad", plan)
        if n:
            _lib.check_train(lib.gcd_smallm_wgrad(dev.data_ptr(), n, blocks, _stream()), "gcd_smallm_wgrad")
        for it in self.items:
            for p in (it["w"], it["b"]):
                if p is not None and p.requires_grad:
                    plan.touched(p)


# ------------------------------------------------------------------------------------------------------------------------
# units
# ------------------------------------------------------------------------------------------------------------------------
class _ResUnit:
    """VideoResBlock.forward (video_model.py:62-81): ResBlock 2-D, the (3,1,1) time_stack ResBlock, AlphaBlender.  The two
    emb_layers vectors come in from the grouped few-row launch (`e2d`, `et`: [frames, Cout])."""

    def __init__(self, plan, rb: VideoResBlock, idx: int):
        self.plan, self.rb, self.idx = plan, rb, idx

    def fwd(self, x, H, W, save: bool):
        plan, rb = self.plan, self.rb
        N, T = plan.N, plan.T
        HW = H * W
        e2d, et = plan.emb_vecs[self.idx]
        geo = _c3_geo(N, H, W)
        h, c1 = _fused_fwd("c3", x, (rb.in_layers[2].weight, rb.in_layers[2].bias), geo=geo,
                           norm=("gn", rb.in_layers[0], HW, 1e-5, True), rowvec=(e2d, HW))
        cs = None
        if isinstance(rb.skip_connection, nn.Identity):
            skip = x
        else:
            skip, cs = _fused_fwd("lin", x, (rb.skip_connection.weight, rb.skip_connection.bias))
        xs, c2 = _fused_fwd("c3", h, (rb.out_layers[3].weight, rb.out_layers[3].bias), geo=geo,
                            norm=("gn", rb.out_layers[0], HW, 1e-5, True), residual=skip)
        del h, skip
        ts = rb.time_stack
        rows = T * HW
        tgeo = dict(T=T, HW=HW)
        h, c3 = _fused_fwd("t3", xs, (ts.in_layers[2].weight, ts.in_layers[2].bias), geo=tgeo,
                           norm=("gn", ts.in_layers[0], rows, 1e-5, True), rowvec=(et, HW))
        xt, c4 = _fused_fwd("t3", h, (ts.out_layers[3].weight, ts.out_layers[3].bias), geo=tgeo,
                            norm=("gn", ts.out_layers[0], rows, 1e-5, True), residual=xs)
        del h
        y = plan.blend_fwd(rb.time_mixer, xs, xt, HW)
        if not save:
            return y, None
        return y, (c1, cs, c2, c3, c4, xs, xt, HW)

    def bwd(self, saved, dy):
        plan, rb = self.plan, self.rb
        c1, cs, c2, c3, c4, xs, xt, HW = saved
        dy = dy.contiguous()
        # y = a xs + (1 - a) xt and xt = t3(...) + xs: the gradient that reaches xs directly is a dy + (1 - a) dy = dy
        d_xt = plan.blend_bwd(rb.time_mixer, dy, xs, xt, HW, want_xs=False)[1]
        del xs, xt
        # time_stack: xt = t3(gn(h)) + xs ; h = t3(gn(xs)) + et
        dh, _ = _fused_bwd(plan, c4, d_xt)
        del d_xt
        d_xs, d_et = _fused_bwd(plan, c3, dh, dx_add=dy)
        del dh
        # 2-D: xs = c3(gn(h)) + skip ; h = c3(gn(x)) + e2d
        dh, _ = _fused_bwd(plan, c2, d_xs)
        if cs is None:
            dx, d_e2d = _fused_bwd(plan, c1, dh, dx_add=d_xs)
        else:
            dx, d_e2d = _fused_bwd(plan, c1, dh)
            dsk, _ = _fused_bwd(plan, cs, d_xs)
            dx = _add(dx, dsk)
        plan.emb_grads[self.idx] = (d_e2d, d_et)
        return dx


class _AttnUnit:
    """SpatialVideoTransformer.forward (video_attention.py:230-301) on token-major rows: every LayerNorm is the prologue of
    the GEMM it feeds, every `+ x` the epilogue of the GEMM that produces the other addend; the one-key cross-attention
    vectors (`ca_s` [frames, C], `ca_t` [clips, C]) and the frame-position embedding (`pos` [frames, C]) come in from the
    grouped few-row launches."""

    def __init__(self, plan, tr: SpatialVideoTransformer, idx: int):
        self.plan, self.tr, self.idx = plan, tr, idx
        assert len(tr.transformer_blocks) == len(tr.time_stack)

    def _ff(self, ff, x, norm_mod, save_list):
        h, ca = _fused_fwd("lin", x, (ff.net[0].proj.weight, ff.net[0].proj.bias), norm=_ln(norm_mod))
        y, cb = _fused_fwd("lin", h, (ff.net[2].weight, ff.net[2].bias), norm=("geglu",), residual=x)
        save_list += [ca, cb]
        return y

    def _ff_bwd(self, ca, cb, d):
        """d = gradient of y = ff(ln(x)) + x -> gradient of x (a fresh tensor when d must survive: it does not here)."""
        plan = self.plan
        du, _ = _fused_bwd(plan, cb, d)
        dx, _ = _fused_bwd(plan, ca, du, dx_add=d)
        return dx

    def fwd(self, x, H, W, save: bool):
        plan, tr = self.plan, self.tr
        N, T = plan.N, plan.T
        HW, clips, heads = H * W, plan.N // plan.T, tr.heads
        vecs = plan.attn_vecs[self.idx]
        ctxs: List = []
        h, c_in = _fused_fwd("lin", x, (tr.proj_in.weight, tr.proj_in.bias), norm=("gn", tr.norm, HW, 1e-6, False))
        ctxs.append(c_in)
        blends = []
        for d, (sb, tb) in enumerate(zip(tr.transformer_blocks, tr.time_stack)):
            ca_s, ca_t, pos = vecs[d]
            # spatial BasicTransformerBlock (attention.py:551-572)
            qkv, cq = _fused_fwd("qkv", h, (sb.attn1.to_q.weight, sb.attn1.to_k.weight, sb.attn1.to_v.weight), norm=_ln(sb.norm1),
                                 geo=_QKV16)
            sa = _Ctx()
            sa.want16 = True
            o = A.SpatialAttention.forward(sa, qkv, N, HW, heads)
            del qkv
            h1, co = _fused_fwd_from16("lin", o, (sb.attn1.to_out[0].weight, sb.attn1.to_out[0].bias), residual=h,
                                       rowvec=(ca_s, HW))
            del o, h
            cl: List = [cq, sa, co]
            h2 = self._ff(sb.ff, h1, sb.norm3, cl)
            del h1
            # temporal VideoTransformerBlock (video_attention.py:109-140) on x + frame position embedding
            xm = plan.add_rowvec(h2, pos, HW)
            xm = self._ff(tb.ff_in, xm, tb.norm_in, cl)
            qkv, cq = _fused_fwd("qkv", xm, (tb.attn1.to_q.weight, tb.attn1.to_k.weight, tb.attn1.to_v.weight), norm=_ln(tb.norm1),
                                 geo=_QKV16)
            ta = _Ctx()
            ta.want16 = True
            o = A.TemporalAttention.forward(ta, qkv, clips, T, HW, heads)
            del qkv
            xm2, co = _fused_fwd_from16("lin", o, (tb.attn1.to_out[0].weight, tb.attn1.to_out[0].bias), residual=xm,
                                        rowvec=(ca_t, T * HW))
            del o, xm
            cl += [cq, ta, co]
            xm3 = self._ff(tb.ff, xm2, tb.norm3, cl)
            del xm2
            h = plan.blend_fwd(tr.time_mixer, h2, xm3, HW)
            blends.append((h2, xm3))
            ctxs.append(cl)
        y, c_out = _fused_fwd("lin", h, (tr.proj_out.weight, tr.proj_out.bias), residual=x)
        if not save:
            return y, None
        return y, (ctxs, blends, c_out, HW)

    def bwd(self, saved, dy):
        plan, tr = self.plan, self.tr
        ctxs, blends, c_out, HW = saved
        dh, _ = _fused_bwd(plan, c_out, dy)
        grads = []
        for d in reversed(range(len(tr.transformer_blocks))):
            cq1, sa, co1, cf1a, cf1b, cfia, cfib, cq2, ta, co2, cf2a, cf2b = ctxs[1 + d]
            h2, xm3 = blends[d]
            d_h2, d_xm3 = plan.blend_bwd(tr.time_mixer, dh, h2, xm3, HW)
            del dh
            d_xm2 = self._ff_bwd(cf2a, cf2b, d_xm3)
            del d_xm3
            # xm2 = to_out(attn(qkv(ln(xm1)))) + ca_t + xm1
            do, d_ca_t = _fused_bwd(plan, co2, d_xm2)
            dqkv = A.TemporalAttention.backward(ta, do)[0]
            del do
            d_xm1, _ = _fused_bwd(plan, cq2, dqkv, dx_add=d_xm2)
            del dqkv, d_xm2
            d_xm0 = self._ff_bwd(cfia, cfib, d_xm1)
            del d_xm1
            d_pos = plan.rowblock_sum(d_xm0, HW)           # xm0 = h2 + pos per frame
            _add(d_h2, d_xm0)
            del d_xm0
            d_h1 = self._ff_bwd(cf1a, cf1b, d_h2)
            del d_h2
            do, d_ca_s = _fused_bwd(plan, co1, d_h1)
            dqkv = A.SpatialAttention.backward(sa, do)[0]
            del do
            dh, _ = _fused_bwd(plan, cq1, dqkv, dx_add=d_h1)
            del dqkv, d_h1
            grads.append((d_ca_s, d_ca_t, d_pos))
        dx, _ = _fused_bwd(plan, ctxs[0], dh, dx_add=dy)
        plan.attn_grads[self.idx] = grads[::-1]
        return dx


class _ConvUnit:
    """A plain 3 x 3 convolution of the network: the input convolution, Downsample (stride 2), Upsample (fused x2), and the
    output head GroupNorm + SiLU + convolution (video_model.py:455-459)."""

    def __init__(self, plan, conv: nn.Conv2d, stride=1, upsample=False, norm=None, first=False):
        self.plan, self.conv, self.stride, self.upsample, self.norm, self.first = plan, conv, stride, upsample, norm, first

    def out_hw(self, H, W):
        g = _c3_geo(1, H, W, self.stride, self.upsample)
        return g["Ho"], g["Wo"]

    def fwd(self, x, H, W, save: bool):
        norm = None if self.norm is None else ("gn", self.norm, H * W, 1e-5, True)
        y, c = _fused_fwd("c3", x, (self.conv.weight, self.conv.bias), geo=_c3_geo(self.plan.N, H, W, self.stride, self.upsample),
                          norm=norm)
        return y, (c if save else None)

    def bwd(self, saved, dy):
        dx, _ = _fused_bwd(self.plan, saved, dy, need_x=(not self.first) or self.plan.need_dx)
        return dx


# ------------------------------------------------------------------------------------------------------------------------
# the plan
# ------------------------------------------------------------------------------------------------------------------------
class TrainPlan:
    """One fine-tune forward + backward of a VideoUNet on `clips` x T frames of H x W latents, without a tape.

        plan = TrainPlan(unet)                                    # once per network
        out = plan.forward(x, timesteps, context, y, T, ioi)      # (N, 4, H, W); keeps unit inputs
        dx = plan.backward(d_out)                                 # parameter .grad filled; returns d x | None
        optimizer.step(); plan.repack()                           # after the parameters changed

    Parameter gradients live in ONE flat fp32 buffer (`p.grad` are views of it): weight gradients are written in place by
    the kernels, the 1-D parameters' region is zeroed by one memset per step.  `accumulate=True` (gradient accumulation,
    main.py:950) adds onto what is there instead.  `on_param_grad(p)` is called once per parameter as soon as its gradient
    is final (training.GradBucketer._on_grad: the overlapped data-parallel exchange)."""

    def __init__(self, unet: VideoUNet, use_checkpoint: Optional[bool] = None,
                 on_param_grad: Optional[Callable[[nn.Parameter], None]] = None):
        self.unet = unet
        self.set_checkpoint(use_checkpoint)
        self.on_param_grad = on_param_grad
        self.listeners_enabled = True
        self.accumulate = False
        self.need_dx = False
        dev = next(unet.parameters()).device
        self.device = dev
        if dev.type != "cuda":
            raise _lib.GcdError("TrainPlan needs the network on a GPU: gcd_amd has no CPU path")
        A.PACK.attach(unet)
        # ---- flat gradient buffer: [>= 2-D parameters | 1-D parameters] ----
        ps = [p for p in unet.parameters() if p.requires_grad]
        big = [p for p in ps if p.dim() >= 2]
        small = [p for p in ps if p.dim() < 2]
        n_big = sum((p.numel() + 3) // 4 * 4 for p in big)
        n_small = sum((p.numel() + 3) // 4 * 4 for p in small)
        self.flat = torch.zeros(n_big + n_small, dtype=_f32, device=dev)
        self.flat_small = self.flat[n_big:]
        self._gview: Dict[int, torch.Tensor] = {}
        off = 0
        for p in big + small:
            self._gview[id(p)] = self.flat[off:off + p.numel()].view(p.shape)
            off += (p.numel() + 3) // 4 * 4
        self._reached = set()
        self.graphed = None
        # per-step arena of zeroed fp32 accumulators (atomic sums: bias / per-frame-vector gradients, few-row dgrads, blend
        # partials): ONE memset per step instead of a fill launch per accumulator
        self._arena = torch.zeros(8 << 20, dtype=_f32, device=dev)
        self._arena_off = 0
        # persistent (un-zeroed) storage of the few-row tensors, bump-allocated in the same order every step, so that their
        # addresses — and with them the grouped launches' device tables — do not change from step to step
        self._sbuf = torch.empty(16 << 20, dtype=_f32, device=dev)
        self._sbuf_off = 0
        self._table_cache: Dict[tuple, tuple] = {}
        self._pos_in: Dict[tuple, torch.Tensor] = {}
        self._units = self._build_units()
        self._blenders = [rb.time_mixer for rb in self.res_blocks] + [tr.time_mixer for tr in self.transformers]
        self._blender_slot = {id(b): i for i, b in enumerate(self._blenders)}
        self._blender_fixed = torch.tensor([b.merge_strategy == "fixed" for b in self._blenders], device=dev)
        self._blender_with_images = torch.tensor([b.merge_strategy == "learned_with_images" for b in self._blenders],
                                                 device=dev)
        # positions of the (scalar) mix factors in the flat gradient buffer: their gradients are scattered in one launch
        base = self.flat.data_ptr()
        self._mix_params = [b.mix_factor for b in self._blenders
                            if b.merge_strategy != "fixed" and b.mix_factor.requires_grad and id(b.mix_factor) in self._gview]
        self._mix_slots = torch.tensor([self._blender_slot[id(b)] for b in self._blenders
                                        if b.merge_strategy != "fixed" and b.mix_factor.requires_grad
                                        and id(b.mix_factor) in self._gview], dtype=torch.long, device=dev)
        self._mix_idx = torch.tensor([(self._gview[id(p)].data_ptr() - base) // 4 for p in self._mix_params],
                                     dtype=torch.long, device=dev)
        self._pack_tables = None
        self._pack_dtypes = None

    # ---- activation checkpointing: a memory decision, made against 288 GB ----
    # bytes of unit contexts kept per (L0 token x model channel) when nothing is recomputed: measured on the full-width
    # network, 2 clips 54.3 GB vs 32.9 GB peak and 8 clips 132 GB vs 47 GB (profiles/r05_train_step_fp16*.json)
    CTX_BYTES_PER_TOKEN_CHANNEL = 1600

    def set_checkpoint(self, use_checkpoint: Optional[bool]) -> None:
        """True / False: as told.  None (the caller follows the network's own `use_checkpoint`, True in every GCD config
        because the reference trains on 80 GB parts): "auto" — recompute only when the contexts would not fit in the free
        HBM of this device; GCD_TRAIN_CHECKPOINT = on | off | auto overrides."""
        import os
        env = os.environ.get("GCD_TRAIN_CHECKPOINT", "")
        if use_checkpoint is None:
            pol = env if env in ("on", "off", "auto") else ("auto" if self.unet.use_checkpoint else "off")
        else:
            pol = "on" if use_checkpoint else "off"
        self.checkpoint_policy = pol
        self.use_checkpoint = pol != "off"

    @classmethod
    def contexts_fit(cls, tokens: int, model_channels: int, free_bytes: int) -> bool:
        """True when the unit contexts of one step (CTX_BYTES_PER_TOKEN_CHANNEL per L0 token and model channel, + 25 %)
        fit in `free_bytes` with 8 GB to spare — i.e. when nothing needs to be recomputed."""
        return cls.CTX_BYTES_PER_TOKEN_CHANNEL * tokens * model_channels * 1.25 <= free_bytes - (8 << 30)

    def _decide_checkpoint(self, tokens: int) -> None:
        if self.checkpoint_policy != "auto":
            self.use_checkpoint = self.checkpoint_policy == "on"
            return
        free, _ = torch.cuda.mem_get_info(self.device)
        free += torch.cuda.memory_reserved(self.device) - torch.cuda.memory_allocated(self.device)
        self.use_checkpoint = not self.contexts_fit(tokens, self.unet.model_channels, free)

    # a plan is scratch state of ONE network object on one device: copies and pickles of the network do not carry it
    # (plan_for builds a fresh one on first use)
    def __deepcopy__(self, memo):
        return None

    def __reduce__(self):
        return (type(None), ())

    def zeros(self, m: int, n: int) -> Optional[torch.Tensor]:
        """A zeroed [m, n] view of the step's arena (None when it is exhausted: the caller then allocates)."""
        need = (m * n + 3) // 4 * 4
        if self._arena_off + need > self._arena.numel():
            return None
        v = self._arena[self._arena_off:self._arena_off + m * n].view(m, n)
        self._arena_off += need
        return v

    def snew(self, m: int, n: int) -> torch.Tensor:
        need = (m * n + 3) // 4 * 4
        if self._sbuf_off + need > self._sbuf.numel():
            return torch.empty(m, n, dtype=_f32, device=self.device)
        v = self._sbuf[self._sbuf_off:self._sbuf_off + m * n].view(m, n)
        self._sbuf_off += need
        return v

    # ---- gradients ----
    def grad_dest(self, p: torch.Tensor) -> Optional[torch.Tensor]:
        """Where the gradient of parameter tensor `p` is written (a view of the flat buffer in p's shape), or None when p
        is not one of this plan's parameters (autograd_ops then returns a fresh tensor as before)."""
        return self._gview.get(id(p))

    def touched(self, p) -> None:
        """The gradient of parameter p is final for this backward pass: publish it as p.grad and tell the listeners
        (training.GradBucketer launches a bucket's all-reduce the moment its last gradient is final)."""
        if id(p) in self._reached:
            return
        self._reached.add(id(p))
        p.grad = self._gview[id(p)]
        _notify(self, p)

    def grads_are_live(self) -> bool:
        """True when the parameters' .grad are this plan's views, i.e. a previous backward of the current accumulation
        window wrote them (torch semantics: a backward pass ADDS to an existing .grad; zero_grad() sets them to None)."""
        live = foreign = False
        for p in self.unet.parameters():
            if p.grad is None:
                continue
            v = self._gview.get(id(p))
            if v is not None and p.grad.data_ptr() == v.data_ptr():
                live = True
            else:
                foreign = True
        if foreign:
            raise NotImplementedError("TrainPlan: a parameter holds a .grad that is not this plan's buffer; call "
                                      "optimizer.zero_grad() (set to None) before the first backward of a step")
        return live

    def sink(self, p: nn.Parameter, g: Optional[torch.Tensor]) -> None:
        """Gradient `g` of parameter p is final: it either already sits in p's slot of the flat buffer (written in place
        by a kernel) or is copied / added there."""
        if g is None:
            return
        dst = self._gview[id(p)]
        if g.data_ptr() != dst.data_ptr():
            g = g.reshape(dst.shape)
            if self.accumulate:
                dst.add_(g)
            else:
                dst.copy_(g)
        self.touched(p)

    # ---- structure ----
    def _build_units(self):
        unet = self.unet
        self.res_blocks: List[VideoResBlock] = []
        self.transformers: List[SpatialVideoTransformer] = []

        def units_of(seq, first=False):
            out = []
            for m in seq:
                if isinstance(m, VideoResBlock):
                    out.append(_ResUnit(self, m, len(self.res_blocks)))
                    self.res_blocks.append(m)
                elif isinstance(m, SpatialVideoTransformer):
                    out.append(_AttnUnit(self, m, len(self.transformers)))
                    self.transformers.append(m)
                elif isinstance(m, Downsample):
                    out.append(_ConvUnit(self, m.op, stride=2))
                elif isinstance(m, Upsample):
                    out.append(_ConvUnit(self, m.conv, upsample=True))
                elif isinstance(m, nn.Conv2d):
                    out.append(_ConvUnit(self, m, first=first))
                else:
                    raise NotImplementedError(type(m).__name__)
            return out
        u = dict(inp=[units_of(b, first=(i == 0)) for i, b in enumerate(unet.input_blocks)],
                 mid=units_of(unet.middle_block),
                 out=[units_of(b) for b in unet.output_blocks],
                 head=_ConvUnit(self, unet.out[2], norm=unet.out[0]))
        return u

    # ---- weight operand forms: one launch per optimizer step ----
    def repack(self) -> None:
        """Rebuild every 16-bit operand form from the fp32 parameters (call after the optimizer step) and hand them to
        `autograd_ops.PACK`, where the operator code looks them up."""
        fdt, gdt = A._dt(A.FWD_DTYPE), A._dt(A.GRAD_DTYPE)
        if self._pack_tables is None or self._pack_dtypes != (fdt, gdt):
            self._build_pack_tables(fdt, gdt)
        lib = _lib.load_train()
        for dev_tab, n, tiles, bf16 in self._pack_tables:
            _lib.check_train(lib.gcd_train_pack_weights(dev_tab.data_ptr(), n, tiles, bf16, _stream()),
                             "gcd_train_pack_weights")
        A.PACK.clear()
        for owners, kind, tensor in self._pack_forms:
            key = (tuple(q.data_ptr() for q in owners), tuple(tuple(q.shape) for q in owners), kind)
            A.PACK._d[key] = (tuple(q._version for q in owners), tensor)
        # the parameter versions these forms were made from: `forward` repacks when ANY optimizer (not only the in-repo
        # fused Adam, which clears PACK) has stepped since — a stale form would otherwise be rebuilt per weight with torch ops
        self._packed_versions = tuple(q._version for owners, _, _ in self._pack_forms for q in owners)

    def _pack_is_stale(self) -> bool:
        if self._pack_tables is None or not A.PACK._d or getattr(self, "_packed_versions", None) is None:
            return True
        return self._packed_versions != tuple(q._version for owners, _, _ in self._pack_forms for q in owners)

    def _build_pack_tables(self, fdt, gdt) -> None:
        dev = self.device
        entries_f: List[_lib.PackEntry] = []
        entries_t: List[_lib.PackEntry] = []
        self._pack_forms = []

        def entry(src, N, Cc, taps):
            e = _lib.PackEntry()
            e.src, e.N, e.C, e.taps = src.data_ptr(), N, Cc, taps
            e.tiles_c = (Cc + 31) // 32
            return e

        def add_lin(ws, kind_f, kind_t):
            """One or several [N, K] parameters stacked along N (q | k | v)."""
            K = ws[0].reshape(ws[0].shape[0], -1).shape[1]
            Ntot = sum(w.shape[0] for w in ws)
            Wf = torch.empty(Ntot, K, dtype=fdt, device=dev)
            Wt = torch.empty(K, Ntot, dtype=gdt, device=dev)
            n0 = 0
            for w in ws:
                N = w.shape[0]
                ef = entry(w, N, K, 1)
                ef.dst_f, ef.f_ns, ef.f_ts = Wf[n0:].data_ptr(), K, 0
                entries_f.append(ef)
                et = entry(w, N, K, 1)
                et.dst_t, et.t_cs, et.t_ts, et.mirror = Wt[:, n0:].data_ptr(), Ntot, 0, 0
                entries_t.append(et)
                n0 += N
            self._pack_forms.append((tuple(ws), kind_f(fdt), Wf))
            self._pack_forms.append((tuple(ws), kind_t(gdt), Wt))

        def add_c3(w, stride):
            Cout, Cin = w.shape[0], w.shape[1]
            _, _, cin_p, cout_p = A._c3_dims(w, None)
            Wf = torch.zeros(cout_p, 9 * cin_p, dtype=fdt, device=dev)
            ef = entry(w, Cout, Cin, 9)
            ef.dst_f, ef.f_ns, ef.f_ts = Wf.data_ptr(), 9 * cin_p, cin_p
            entries_f.append(ef)
            self._pack_forms.append(((w,), f"c3_{cin_p}_{cout_p}_{fdt}", Wf))
            et = entry(w, Cout, Cin, 9)
            if stride == 1:       # dgrad as a convolution of dY: W'[cin][(2-kh, 2-kw)][cout]
                Wd = torch.zeros(cin_p, 9 * cout_p, dtype=gdt, device=dev)
                et.dst_t, et.t_cs, et.t_ts, et.mirror = Wd.data_ptr(), 9 * cout_p, cout_p, 1
                self._pack_forms.append(((w,), f"c3d_{cin_p}_{cout_p}_{gdt}", Wd))
            else:                 # stride 2: dcol = dY W by a plain GEMM -> W^T of the packed form
                Wd = torch.zeros(9 * cin_p, cout_p, dtype=gdt, device=dev)
                et.dst_t, et.t_cs, et.t_ts, et.mirror = Wd.data_ptr(), cout_p, cin_p * cout_p, 0
                self._pack_forms.append(((w,), f"c3t_{cin_p}_{cout_p}_{gdt}", Wd))
            entries_t.append(et)

        def add_t3(w):
            Cout, Cin = w.shape[0], w.shape[1]
            Wf = torch.empty(Cout, 3 * Cin, dtype=fdt, device=dev)
            ef = entry(w, Cout, Cin, 3)
            ef.dst_f, ef.f_ns, ef.f_ts = Wf.data_ptr(), 3 * Cin, Cin
            entries_f.append(ef)
            self._pack_forms.append(((w,), f"t3_{fdt}", Wf))
            Wd = torch.empty(Cin, 3 * Cout, dtype=gdt, device=dev)
            et = entry(w, Cout, Cin, 3)
            et.dst_t, et.t_cs, et.t_ts, et.mirror = Wd.data_ptr(), 3 * Cout, Cout, 1
            entries_t.append(et)
            self._pack_forms.append(((w,), f"t3d_{gdt}", Wd))

        lin_f = lambda d: f"lin_{d}"          # noqa: E731
        lin_t = lambda d: f"lin_t_{d}"        # noqa: E731
        qkv_f = lambda d: f"qkv_{d}"          # noqa: E731
        qkv_t = lambda d: f"qkv_t_{d}"        # noqa: E731
        unet = self.unet
        add_c3(unet.input_blocks[0][0].weight, 1)
        add_c3(unet.out[2].weight, 1)
        for m in unet.modules():
            if isinstance(m, Downsample):
                add_c3(m.op.weight, 2)
            elif isinstance(m, Upsample):
                add_c3(m.conv.weight, 1)
        for rb in self.res_blocks:
            add_c3(rb.in_layers[2].weight, 1)
            add_c3(rb.out_layers[3].weight, 1)
            if not isinstance(rb.skip_connection, nn.Identity):
                add_lin([rb.skip_connection.weight], lin_f, lin_t)
            add_t3(rb.time_stack.in_layers[2].weight)
            add_t3(rb.time_stack.out_layers[3].weight)
        for tr in self.transformers:
            add_lin([tr.proj_in.weight], lin_f, lin_t)
            add_lin([tr.proj_out.weight], lin_f, lin_t)
            for sb, tb in zip(tr.transformer_blocks, tr.time_stack):
                for blk in (sb, tb):
                    at = blk.attn1
                    add_lin([at.to_q.weight, at.to_k.weight, at.to_v.weight], qkv_f, qkv_t)
                    add_lin([at.to_out[0].weight], lin_f, lin_t)
                    add_lin([blk.ff.net[0].proj.weight], lin_f, lin_t)
                    add_lin([blk.ff.net[2].weight], lin_f, lin_t)
                add_lin([tb.ff_in.net[0].proj.weight], lin_f, lin_t)
                add_lin([tb.ff_in.net[2].weight], lin_f, lin_t)

        def table(entries, bf16):
            t0 = 0
            for e in entries:
                e.tile0 = t0
                t0 += ((e.N + 31) // 32) * e.tiles_c
            arr = (_lib.PackEntry * len(entries))(*entries)
            d = torch.empty(C.sizeof(arr), dtype=torch.uint8, device=dev)
            d.copy_(torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8))
            return (d, len(entries), t0, int(bf16))

        if fdt == gdt:        # one pass over the parameters writes both forms
            merged = []
            for ef, et in zip(entries_f, entries_t):
                ef.dst_t, ef.t_cs, ef.t_ts, ef.mirror = et.dst_t, et.t_cs, et.t_ts, et.mirror
                merged.append(ef)
            self._pack_tables = [table(merged, fdt == torch.bfloat16)]
        else:
            self._pack_tables = [table(entries_f, fdt == torch.bfloat16), table(entries_t, gdt == torch.bfloat16)]
        self._pack_dtypes = (fdt, gdt)

    # ---- small kernels ----
    def _make_alphas(self) -> None:
        """AlphaBlender.get_alpha per frame (util.py:342-356) for ALL blenders of the network at once -> a [n_blenders,
        frames] table (row = blender), and `live` = d alpha / d mix_factor per frame (0 on image-only frames and for the
        "fixed" strategy).  A handful of launches per step instead of ~7 per blender."""
        bl = self._blenders
        frames = self.N
        ioi = self.ioi.reshape(-1).bool()
        mix = torch.cat([b.mix_factor.detach().float().reshape(1) for b in bl])            # [n_bl]
        fixed = self._blender_fixed
        s = torch.sigmoid(mix)
        a = torch.where(fixed, mix, s)[:, None].expand(len(bl), frames)
        with_images = self._blender_with_images[:, None] & ioi[None, :]
        a = torch.where(with_images, torch.ones_like(a), a).contiguous()
        live = (s * (1.0 - s))[:, None].expand(len(bl), frames)
        live = torch.where(with_images | fixed[:, None], torch.zeros_like(live), live).contiguous()
        self._alpha_tab, self._live_tab = a, live

    def _alpha(self, blender):
        slot = self._blender_slot[id(blender)]
        return self._alpha_tab[slot], self._live_tab[slot], slot, blender

    def blend_fwd(self, blender, xs, xt, rows):
        a = self._alpha(blender)[0]
        y = torch.empty_like(xs)
        _lib.check_train(_lib.load_train().gcd_blend_fwd_f32(xs.data_ptr(), xs.stride(0), xt.data_ptr(), xt.stride(0),
                                                             a.data_ptr(), xs.shape[0], xs.shape[1], rows, y.data_ptr(),
                                                             y.stride(0), _stream()), "gcd_blend_fwd_f32")
        return y

    def blend_bwd(self, blender, dy, xs, xt, rows, want_xs: bool = True):
        """-> (d_xs | None, d_xt) fresh tensors; the mix factor's gradient partials accumulate in the step's arena."""
        a, live, slot, _ = self._alpha(blender)
        dy = dy.contiguous()
        d_xs, d_xt = (torch.empty_like(dy) if want_xs else None), torch.empty_like(dy)
        want = blender.merge_strategy != "fixed" and blender.mix_factor.requires_grad
        dal = self._dalpha[slot] if want else None
        _lib.check_train(_lib.load_train().gcd_blend_bwd_f32(
            dy.data_ptr(), dy.stride(0), xs.data_ptr(), xs.stride(0), xt.data_ptr(), xt.stride(0), a.data_ptr(),
            dy.shape[0], dy.shape[1], rows, 0 if d_xs is None else d_xs.data_ptr(), dy.stride(0), 0, d_xt.data_ptr(),
            d_xt.stride(0),
            0 if dal is None else dal.data_ptr(), _stream()), "gcd_blend_bwd_f32")
        return d_xs, d_xt

    def add_rowvec(self, x, vec, rows):
        """x [M, C] + vec[m // rows] (the frame position embedding)."""
        return x + vec.repeat_interleave(rows, dim=0)

    def rowblock_sum(self, x, rows):
        with A.grad_sink(self):
            return A._colsum(x.contiguous(), rows)

    # ---- forward ----
    def forward(self, x, timesteps, context, y, num_video_frames: int, image_only_indicator, record: bool = True):
        """record=False (validation / image logging under torch.no_grad()): nothing is kept for a backward pass — no unit
        contexts, no trace.  The plan holds ONE forward's state: a backward of an EARLIER forward raises (see `_serial`)."""
        unet = self.unet
        ops._need_gpu(x, timesteps, context, y)
        # every forward gets a serial number; `_PlannedUNet.backward` refuses to run for any but the LATEST forward (its
        # activations are the only ones the plan still holds) instead of silently differentiating the wrong ones
        self._serial = getattr(self, "_serial", 0) + 1
        T = num_video_frames
        N, _, H, W = x.shape
        assert N % T == 0 and context.dim() == 3
        if context.shape[1] != 1:
            raise NotImplementedError("gcd_amd implements the single-token (CLIP image) context of SVD / GCD")
        self.N, self.T, self.H, self.W = N, T, H, W
        if not torch.cuda.is_current_stream_capturing():
            self._decide_checkpoint(N * H * W)
        self.ioi = image_only_indicator.to(x.device)
        self._make_alphas()
        if self._pack_dtypes != (A._dt(A.FWD_DTYPE), A._dt(A.GRAD_DTYPE)) or self._pack_is_stale():
            self.repack()
        self._reached = set()
        self._arena.zero_()
        self._arena_off = 0
        self._small_forward(timesteps, context, y)
        n_bl = len(self.res_blocks) + len(self.transformers)
        self._dalpha = zeros_or_new(self, n_bl, N)
        save = record and not self.use_checkpoint
        h = x.float().permute(0, 2, 3, 1).reshape(N * H * W, -1).contiguous()
        trace = []          # (unit, input, H, W, saved | None)
        hs: List[torch.Tensor] = []
        st = [H, W]

        def run(units, h):
            for u in units:
                hin = h
                h, sv = u.fwd(hin, st[0], st[1], save)
                if record:
                    trace.append((u, hin if self.use_checkpoint else None, st[0], st[1], sv))
                if isinstance(u, _ConvUnit):
                    st[0], st[1] = u.out_hw(st[0], st[1])
            return h
        for units in self._units["inp"]:
            h = run(units, h)
            hs.append(h)
            trace.append(("push",))
        h = run(self._units["mid"], h)
        for units in self._units["out"]:
            skip = hs.pop()
            trace.append(("cat", h.shape[1], skip.shape[1]))
            h = run(units, torch.cat([h, skip], dim=1))
        h = run([self._units["head"]], h)
        self._trace = trace if record else None
        out = h.reshape(N, H, W, -1).permute(0, 3, 1, 2).contiguous()
        return out

    # ---- backward ----
    def backward(self, d_out: torch.Tensor, accumulate: bool = False):
        """d_out (N, 4, H, W) fp32: gradient of the (scaled) loss with respect to `forward`'s result."""
        self.accumulate = accumulate
        N, H, W = self.N, self.H, self.W
        if not accumulate:
            self.flat_small.zero_()
        self.emb_grads = [None] * len(self.res_blocks)
        self.attn_grads = [None] * len(self.transformers)
        d = d_out.float().permute(0, 2, 3, 1).reshape(N * H * W, -1).contiguous()
        skips: List[torch.Tensor] = []
        trace = self._trace
        self._trace = None
        while trace:
            item = trace.pop()
            if item[0] == "cat":
                c1 = item[1]
                skips.append(d[:, c1:])
                d = d[:, :c1]
                continue
            if item[0] == "push":
                d = _add(d.contiguous() if not d.is_contiguous() else d, skips.pop())
                continue
            u, hin, h_, w_, sv = item
            if sv is None:
                _, sv = u.fwd(hin, h_, w_, True)
            d = u.bwd(sv, d)
            del sv, hin
        self._small_backward()
        self.accumulate = False
        return None

    # ---- the few-row side: embeddings, emb_layers, cross-attention vectors, position embeddings ----
    def _small_forward(self, timesteps, context, y):
        unet, dev, N, T = self.unet, self.device, self.N, self.T
        clips = N // T
        adm = unet.adm_in_channels
        E = unet.time_embed[0].weight.shape[0]
        self._sbuf_off = 0
        new = self.snew
        zeros = lambda m, n: zeros_or_new(self, m, n)      # noqa: E731

        def temb(t, dim, period):
            e = new(t.numel(), dim)
            ops.timestep_embedding(t.detach().float().contiguous(), e, float(period))
            return e
        ctx2d = new(N, context.shape[-1])
        ctx2d.copy_(context.reshape(N, -1))
        yv = new(N, y.shape[1])
        yv.copy_(y)
        # stage 1: first Linear of every embedding MLP (inputs known up front)
        g1 = _SmallGroup(dev, "g1")
        t_in = temb(timesteps, unet.model_channels, getattr(unet, "max_ddpm_temb_period", 10000.0))
        mlps = [(unet.time_embed, t_in), (unet.label_emb[0], yv[:, :adm])]
        if unet.aux_emb_dim > 0:
            mlps.append((unet.aux_label_emb, yv[:, adm:]))
        hid = []
        for seq, xin in mlps:
            hid.append(g1.add(xin, seq[0], new(N, seq[0].weight.shape[0])))
        pos_hid = []
        for i, tr in enumerate(self.transformers):
            key = (i, N, T)
            pin = self._pos_in.get(key)        # the frame-position sinusoids depend on (T, clips) only: made once
            if pin is None:
                fidx = torch.arange(T, device=dev, dtype=_f32).repeat(clips)
                pin = torch.empty(N, tr.in_channels, dtype=_f32, device=dev)
                ops.timestep_embedding(fidx, pin, float(tr.max_time_embed_period))
                self._pos_in[key] = pin
            pos_hid.append(g1.add(pin, tr.time_pos_embed[0], new(N, tr.time_pos_embed[0].weight.shape[0])))
        # cross-attention value projections: to_v(ctx) per frame (spatial) / per clip (temporal: first frame's context)
        ctx_clip = new(clips, ctx2d.shape[1])
        ctx_clip.copy_(ctx2d[::T])
        cav = []
        for tr in self.transformers:
            row = []
            for sb, tb in zip(tr.transformer_blocks, tr.time_stack):
                vs = g1.add(ctx2d, sb.attn2.to_v, new(N, sb.attn2.to_v.weight.shape[0]))
                vt = g1.add(ctx_clip, tb.attn2.to_v, new(clips, tb.attn2.to_v.weight.shape[0]))
                row.append((vs, vt))
            cav.append(row)
        g1.forward(self)
        # stage 2: second Linears (SiLU on the hidden rows); emb = sum of the embedding MLPs
        g2 = _SmallGroup(dev, "g2")
        self._d_hid = [zeros(*h.shape) for h in hid]
        embs = [g2.add(h, seq[2], new(N, E), silu_in=True, dx=self._d_hid[i], dx_silu=True)
                for i, ((seq, _), h) in enumerate(zip(mlps, hid))]      # (own outputs: problems of a launch run concurrently)
        self._d_pos_hid = [zeros(*h.shape) for h in pos_hid]
        pos = []
        for tr, h, dh in zip(self.transformers, pos_hid, self._d_pos_hid):
            pos.append(g2.add(h, tr.time_pos_embed[2], new(N, tr.in_channels), silu_in=True, dx=dh, dx_silu=True))
        self._d_cav = []
        self.attn_vecs = []
        for tr, row, p in zip(self.transformers, cav, pos):
            vecs, drow = [], []
            for (sb, tb), (vs, vt) in zip(zip(tr.transformer_blocks, tr.time_stack), row):
                dvs, dvt = zeros(*vs.shape), zeros(*vt.shape)
                ca_s = g2.add(vs, sb.attn2.to_out[0], new(N, sb.attn2.to_out[0].weight.shape[0]), dx=dvs)
                ca_t = g2.add(vt, tb.attn2.to_out[0], new(clips, tb.attn2.to_out[0].weight.shape[0]), dx=dvt)
                vecs.append((ca_s, ca_t, p))
                drow.append((dvs, dvt))
            self.attn_vecs.append(vecs)
            self._d_cav.append(drow)
        g2.forward(self)
        emb = new(N, E)
        torch.add(embs[0], embs[1], out=emb)
        for e in embs[2:]:
            emb += e
        # stage 3: the 44 emb_layers on SiLU(emb)
        g3 = _SmallGroup(dev, "g3")
        self._d_emb = zeros(N, E)
        self.emb_vecs = []
        for rb in self.res_blocks:
            l2, lt = rb.emb_layers[1], rb.time_stack.emb_layers[1]
            e2d = g3.add(emb, l2, new(N, l2.weight.shape[0]), silu_in=True, dx=self._d_emb, dx_silu=True)
            et = g3.add(emb, lt, new(N, lt.weight.shape[0]), silu_in=True, dx=self._d_emb, dx_silu=True)
            self.emb_vecs.append((e2d, et))
        g3.forward(self)
        self._groups = (g1, g2, g3)
        self._emb = emb

    def _small_backward(self):
        g1, g2, g3 = self._groups
        N = self.N
        # stage 3: emb_layers
        dys = []
        for de in self.emb_grads:
            dys += [de[0].contiguous(), de[1].contiguous()]
        g3.backward(self, dys)
        # stage 2: d emb (the same for every embedding MLP's output), d pos, d ca
        dys = [self._d_emb] * len(self._d_hid)
        for gr in self.attn_grads:
            # one frame-position embedding per transformer: the sum over its depth
            dp = gr[0][2]
            for more in gr[1:]:
                dp = dp + more[2]
            dys.append(dp.contiguous())
        for gr in self.attn_grads:
            for (d_ca_s, d_ca_t, _) in gr:
                dys += [d_ca_s.contiguous(), d_ca_t.contiguous()]
        g2.backward(self, dys)
        # stage 1: first Linears (inputs are data: no dx)
        dys = list(self._d_hid) + list(self._d_pos_hid)
        for drow in self._d_cav:
            for dvs, dvt in drow:
                dys += [dvs, dvt]
        g1.backward(self, dys)
        # the blenders' mix factors: d mix = sum over live frames of d alpha * s (1 - s) — all of them in three launches
        if self._mix_params:
            d_mix = (self._dalpha * self._live_tab).sum(1)[self._mix_slots]
            self.flat.index_put_((self._mix_idx,), d_mix, accumulate=self.accumulate)
            for p in self._mix_params:
                self.touched(p)
        self._groups = None


_LATE_MARKS = (".emb_layers.", ".attn2.to_v.", ".attn2.to_out.", "time_pos_embed.", "time_embed.", "label_emb.",
               "mix_factor")


def late_parameters(net) -> list:
    """The parameters whose gradients `_small_backward` finalises after the whole unit walk (three grouped launches for all
    few-row Linears + one for the mix factors): `training.GradBucketer(..., late=late_parameters(net))` keeps them out of the
    buckets that can leave during the backward pass."""
    return [p for n, p in net.named_parameters() if p.requires_grad and any(m in "." + n for m in _LATE_MARKS)]


class GraphedPlan:
    """The planned step under two hipGraphs (torch.cuda.graph): after `WARMUP` eager calls with the same signature the
    forward pass is captured, and the backward pass at the first backward after it; later steps copy the inputs into the
    static buffers and replay — no Python, no per-launch host work between the kernels.  What stays eager: the loss between
    the two graphs, the optimizer step, gradient accumulation (a backward that must ADD falls back to the eager pass), any
    call whose signature differs.  The weight pack (`repack`) is part of the forward graph: it re-reads the parameters the
    optimizer just changed.  Listeners (GradBucketer) are told about every gradient after the backward graph has been
    enqueued, i.e. the data-parallel exchange is not overlapped in this mode."""
    WARMUP = 2

    def __init__(self, plan: TrainPlan):
        self.plan = plan
        self.sig = None
        self.calls = 0
        self.g_fwd = self.g_bwd = None
        self.pool = None
        self.mode = "eager"

    def _reset(self, sig):
        self.sig, self.calls = sig, 0
        self.g_fwd = self.g_bwd = None
        self.static = None
        self.mode = "eager"

    def forward(self, x, timesteps, context, y, T, ioi):
        plan = self.plan
        sig = (tuple(x.shape), tuple(timesteps.shape), tuple(context.shape), tuple(y.shape), tuple(ioi.shape), T,
               A.FWD_DTYPE, A.GRAD_DTYPE, plan.checkpoint_policy, x.dtype, context.dtype, y.dtype)
        if sig != self.sig:
            self._reset(sig)
        if self.g_fwd is None and self.calls < self.WARMUP:
            self.calls += 1
            self.mode = "eager"
            return plan.forward(x, timesteps, context, y, T, ioi)
        if self.g_fwd is None:
            st = dict(x=x.clone(), t=timesteps.clone(), c=context.clone(), y=y.clone(), ioi=ioi.clone())
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                plan.repack()
                st["out"] = plan.forward(st["x"], st["t"], st["c"], st["y"], T, st["ioi"])
            self.pool = g.pool()
            self.g_fwd, self.static = g, st
            self._T = T
            self.mode = "captured_fwd"
            # (the capture enqueued nothing: run it once for this call's result)
        st = self.static
        st["x"].copy_(x)
        st["t"].copy_(timesteps)
        st["c"].copy_(context)
        st["y"].copy_(y)
        st["ioi"].copy_(ioi)
        self.g_fwd.replay()
        if self.mode != "captured_fwd":
            self.mode = "graph"
        return st["out"]

    def backward(self, d_out):
        plan = self.plan
        if self.mode == "eager":
            return plan.backward(d_out, accumulate=plan.grads_are_live())
        if plan.grads_are_live():
            raise NotImplementedError("GraphedPlan: gradient accumulation onto live .grad needs the eager pass "
                                      "(GCD_TRAIN_GRAPH=0) — the captured backward overwrites")
        st = self.static
        if self.g_bwd is None:
            st["d_out"] = d_out.detach().float().contiguous().clone()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            cb, plan.on_param_grad = plan.on_param_grad, None       # host callbacks must not run inside a capture
            plan.listeners_enabled = False
            try:
                with torch.cuda.graph(g, pool=self.pool):
                    plan.backward(st["d_out"], accumulate=False)
            finally:
                plan.listeners_enabled = True
                plan.on_param_grad = cb
            self.g_bwd = g
            self.reached = [p for p in plan.unet.parameters() if id(p) in plan._reached]
        st["d_out"].copy_(d_out)
        self.g_bwd.replay()
        for p in self.reached:
            p.grad = plan._gview[id(p)]
            _notify(plan, p)
        self.mode = "graph"
        return None


class _PlannedUNet(torch.autograd.Function):
    """The whole VideoUNet as ONE autograd node: forward = TrainPlan.forward, backward = TrainPlan.backward (which fills the
    parameters' .grad itself).  `anchor` is a dummy input that requires grad, so that autograd calls backward."""

    @staticmethod
    def forward(ctx, plan, anchor, x, timesteps, context, y, T, ioi):
        ctx.plan = plan
        ctx.graphed = USE_GRAPH and plan.graphed is not None
        if ctx.graphed:
            out = plan.graphed.forward(x, timesteps, context, y, T, ioi)
        else:
            out = plan.forward(x, timesteps, context, y, T, ioi)
        ctx.serial = getattr(plan, "_serial", 0)
        return out

    @staticmethod
    def backward(ctx, d_out):
        plan = ctx.plan
        if ctx.serial != getattr(plan, "_serial", 0):
            raise RuntimeError("gcd_amd planned training pass: backward of a forward that is no longer the network's latest "
                               "(another forward — e.g. a validation pass — ran in between and replaced the activations "
                               "the plan keeps); run the extra forward after backward")
        if ctx.graphed:
            plan.graphed.backward(d_out)
        else:
            plan.backward(d_out, accumulate=plan.grads_are_live())
        return (None,) * 8


# GCD_TRAIN_GRAPH=1: capture the planned forward and backward passes as two hipGraphs after two eager steps (GraphedPlan)
import os as _os  # noqa: E402
USE_GRAPH = _os.environ.get("GCD_TRAIN_GRAPH", "0") == "1"


def set_use_graph(on: bool) -> None:
    global USE_GRAPH
    USE_GRAPH = bool(on)


def plan_for(unet: VideoUNet, use_checkpoint: Optional[bool] = None) -> TrainPlan:
    """The network's plan, created on first use and kept ON the network object (a plain attribute, not a registered
    sub-module): the flat gradient buffer (4 B per parameter) and the persistent operand forms (4 B per matrix parameter)
    live exactly as long as the network does — dropping the network drops them (the two reference each other: a cycle the
    garbage collector takes)."""
    p = unet.__dict__.get("_gcd_train_plan")
    if p is None or p.unet is not unet:
        p = TrainPlan(unet, use_checkpoint)
        p._anchor = torch.zeros(1, device=p.device, requires_grad=True)
        p.graphed = GraphedPlan(p)
        object.__setattr__(unet, "_gcd_train_plan", p)
    else:
        p.set_checkpoint(use_checkpoint)
    return p


def unet_forward_planned(unet: VideoUNet, x, timesteps, context, y, num_video_frames: int, image_only_indicator,
                         use_checkpoint: Optional[bool] = None) -> torch.Tensor:
    """Drop-in for `training.unet_forward_train`: same arguments, same result, one autograd node."""
    if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in (x, context, y)):
        # The planned pass computes PARAMETER gradients only (its few-row Linears treat their inputs as data).  A
        # conditioner trained through `vector` / `crossattn` (the kubric configs train SphericalEmbedder.proj,
        # encoders/modules.py) or a gradient with respect to x needs the operator-level autograd engine: use it, loudly once.
        from . import training as _TR
        if not getattr(unet_forward_planned, "_warned", False):
            import warnings
            warnings.warn("gcd_amd: an input of the UNet requires grad; this step runs on the operator-level autograd engine "
                          "(training.unet_forward_train), not the planned pass", RuntimeWarning)
            unet_forward_planned._warned = True
        return _TR.unet_forward_train(unet, x, timesteps, context, y, num_video_frames, image_only_indicator,
                                      use_checkpoint=use_checkpoint)
    plan = plan_for(unet, use_checkpoint)
    if not torch.is_grad_enabled():
        return plan.forward(x, timesteps, context, y, num_video_frames, image_only_indicator, record=False)
    return _PlannedUNet.apply(plan, plan._anchor, x, timesteps, context, y, num_video_frames, image_only_indicator)
