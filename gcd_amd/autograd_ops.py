"""torch.autograd.Function wrappers that run BOTH directions of the VideoUNet's operators on
libgcd_amd kernels — the building blocks of the fine-tune step (BASELINE.json cfg4; SURVEY.md §8a a23,
§8(f)-2).  The reference trains through torch.autograd over nn.Conv2d / nn.Linear / F.sdpa
(loss.py:115-273 + Lightning's backward); here torch only keeps the graph: every forward and every
gradient is a hand-written gfx950 kernel through the C ABI.

Conventions
  * graph edges are fp32 token-major tensors [rows, channels] (gradients stay in fp32 between
    operators); MFMA operands are rounded to fp16 inside each operator, exactly as in the inference
    engine, with fp32 accumulation;
  * contractions of the backward pass reuse `gcd_gemm_f16`:  dX = dY @ W (weights transposed once per
    parameter version), dW = dY^T @ X (both operands transposed by `gcd_transpose_f16`, the token axis
    zero-padded to the GEMM's 32-deep K granule); convolutions go through im2col / col2im
    (`gcd_im2col3x3_f16` ...), the S x S attention products through plain GEMMs per (frame, head);
  * static loss scaling as in AMP: the caller multiplies the loss by `loss_scale` so that the fp16
    casts of the gradients keep their small values; parameter gradients come out scaled and the
    optimizer step (`gcd_adam_step(grad_scale=1/loss_scale)`) removes the factor.
This path is written for correctness first (gradient parity with torch.autograd on the CPU oracle,
tests/test_backward_gpu.py); it is unfused and is not the measured inference path.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

from . import _lib, ops, packing
from ._lib import GEMM_CONV3X3, GEMM_PLAIN, GEMM_TEMPORAL3, OUT_F16, OUT_F32, check

_f32 = torch.float32
_f16 = torch.float16
_bf16 = torch.bfloat16

# Operand type of the BACKWARD contractions (dX = dY W, dW = dY^T X): "fp16" (default; incoming gradients
# are rounded to fp16, the caller applies a static loss scale so that small values survive) or "bf16"
# (gcd_gemm_desc.operand_bf16 on the general GEMM kernel: fp32's exponent range, no loss scale needed,
# 8 significant bits).  The forward pass is fp16 either way.  Set through `set_grad_dtype`.
GRAD_DTYPE = os.environ.get("GCD_TRAIN_GRAD_DTYPE", "fp16")


def set_grad_dtype(name: str) -> None:
    global GRAD_DTYPE
    if name not in ("fp16", "bf16"):
        raise ValueError("grad dtype must be 'fp16' or 'bf16'")
    GRAD_DTYPE = name


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _ld(t):
    assert t.dim() == 2 and t.stride(1) == 1
    return t.stride(0)


def _cast16(x32: torch.Tensor) -> torch.Tensor:
    """fp32 [M, C] (any row stride) -> contiguous fp16."""
    y = torch.empty(x32.shape, dtype=_f16, device=x32.device)
    ops.cast_f16(x32, y)
    return y


def _cast16_into(x32: torch.Tensor, y16: torch.Tensor) -> None:
    """fp32 -> fp16 into a (possibly wider) buffer's leading columns; the 4-channel ends of the UNet go
    through a plain copy (the cast kernel moves 8 channels per lane)."""
    if x32.shape[1] % 8 == 0:
        ops.cast_f16(x32, y16)
    else:
        y16.copy_(x32)


def _t16_padded(x16: torch.Tensor, granule: int = 32) -> torch.Tensor:
    """[R, C] fp16 / bf16 -> [C, Rp] with Rp = R rounded up to the GEMM's K granule and zero padding (a
    GEMM operand whose contraction axis is the token axis)."""
    R, Cc = x16.shape
    Rp = (R + granule - 1) // granule * granule
    out = torch.zeros(Cc, Rp, dtype=x16.dtype, device=x16.device) if Rp != R else \
        torch.empty(Cc, Rp, dtype=x16.dtype, device=x16.device)
    ops.transpose_f16(x16, out[:, :R])
    return out


def _to_bf16(x: torch.Tensor) -> torch.Tensor:
    y = torch.empty(x.shape, dtype=_bf16, device=x.device)
    ops.cast_bf16(x, y)
    return y


def _grad_contractions(dy32: torch.Tensor, x16: torch.Tensor, w_t16: torch.Tensor, need_dx: bool,
                       need_dw: bool):
    """The two contractions every Linear-shaped backward needs, on gcd_gemm_f16:
         dX [M, K] = dY [M, N] @ W [N, K]      (w_t16 = W^T, [K, N], fp16)
         dW [N, K] = dY^T [N, M] @ X [M, K]    (x16 fp16 [M, K])
    in fp16 (default) or bf16 operands (GRAD_DTYPE; bf16 needs 64-deep contraction axes, else fp16)."""
    M, N = dy32.shape
    K = w_t16.shape[0]
    dev = dy32.device
    bf = GRAD_DTYPE == "bf16" and N % 64 == 0
    dy16 = _to_bf16(dy32) if bf else _cast16(dy32)
    dx = dw = None
    if need_dx:
        dx = torch.empty(M, K, dtype=_f32, device=dev)
        wt = w_t16.to(_bf16) if bf else w_t16
        if N % 32 == 0:
            ops.gemm(dy16, wt, dx, M=M, operand_bf16=bf)
        else:   # N a multiple of 16 only (the padded last conv): widen the contraction axis with zeros
            Np = (N + 31) // 32 * 32
            dyp = torch.zeros(M, Np, dtype=_f16, device=dev)
            dyp[:, :N] = dy16
            wtp = torch.zeros(K, Np, dtype=_f16, device=dev)
            wtp[:, :N] = wt
            ops.gemm(dyp, wtp, dx, M=M)
    if need_dw:
        dw = torch.empty(N, K, dtype=_f32, device=dev)
        g = 64 if bf else 32
        xs = _to_bf16(x16) if bf else x16
        ops.gemm(_t16_padded(dy16, g), _t16_padded(xs, g), dw, M=N, operand_bf16=bf)
    return dx, dw


def _colsum(x32: torch.Tensor, rows_per_block: Optional[int] = None) -> torch.Tensor:
    M, N = x32.shape
    rows = M if rows_per_block is None else rows_per_block
    out = torch.zeros(M // rows, N, dtype=_f32, device=x32.device)
    check(_lib.load().gcd_rowblock_sum_f32(x32.data_ptr(), _ld(x32), M, N, rows, out.data_ptr(), _stream()),
          "gcd_rowblock_sum_f32")
    return out


class _PackCache:
    """fp16 operand forms of a parameter, rebuilt when the parameter changes (torch's version counter;
    `clear()` after an optimizer step that writes through raw pointers).  Keyed by storage address and
    shape, not by Python object: under activation checkpointing the backward pass sees the parameter
    through a different tensor object than the forward did."""

    def __init__(self):
        self._d = {}

    def get(self, p: torch.Tensor, kind: str, fn):
        if not p.is_leaf:          # a temporary (e.g. the concatenated q|k|v weight): pack, do not keep
            return fn(p.detach())
        key = (p.data_ptr(), tuple(p.shape), kind)
        hit = self._d.get(key)
        if hit is not None and hit[0] == p._version:
            return hit[1]
        v = fn(p.detach())
        self._d[key] = (p._version, v)
        return v

    def clear(self):
        self._d.clear()


PACK = _PackCache()


# ------------------------------------------------------------------------------------------------
# Linear:  y = x @ W^T + b            (attention.py q/k/v/out, FeedForward, proj_in/out, emb MLPs)
# ------------------------------------------------------------------------------------------------
class Linear16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        ops._need_gpu(x, weight)
        M, K = x.shape
        N = weight.shape[0]
        if K % 32 or N % 16:
            raise NotImplementedError(f"Linear16: K={K} must be a multiple of 32 and N={N} of 16")
        x16 = _cast16(x)
        w16 = PACK.get(weight, "lin", packing.pack_linear)
        y = torch.empty(M, N, dtype=_f32, device=x.device)
        ops.gemm(x16, w16, y, M=M, bias=None if bias is None else bias.detach().float())
        ctx.save_for_backward(x16, weight)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x16, weight = ctx.saved_tensors
        M, K = x16.shape
        N = weight.shape[0]
        dy = dy.contiguous()
        db = None
        wt16 = PACK.get(weight, "lin_t", lambda w: w.t().contiguous().to(_f16))     # [K, N]
        dx, dw = _grad_contractions(dy, x16, wt16, ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = _colsum(dy)[0]
        return dx, dw, db


def linear(x, weight, bias=None):
    return Linear16.apply(x, weight, bias)


# ------------------------------------------------------------------------------------------------
# Conv2d 3x3 as implicit GEMM over token-major activations (stride 1 / 2, fused x2 nearest upsample)
# ------------------------------------------------------------------------------------------------
class Conv3x3(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, frames, Hi, Wi, stride, upsample):
        ops._need_gpu(x, weight)
        Cout, Cin = weight.shape[0], weight.shape[1]
        cin_p = (Cin + 63) // 64 * 64 if Cin % 32 else Cin          # first conv: 8 -> 64 channels
        cout_p = (Cout + 31) // 32 * 32 if Cout % 32 else Cout      # last conv: 4 -> 32 channels
        if upsample:
            Ho, Wo = 2 * Hi, 2 * Wi
        else:
            Ho, Wo = (Hi - 1) // stride + 1, (Wi - 1) // stride + 1
        Min, Mout = frames * Hi * Wi, frames * Ho * Wo
        assert x.shape == (Min, Cin)
        x16 = torch.zeros(Min, cin_p, dtype=_f16, device=x.device) if cin_p != Cin else None
        if x16 is None:
            x16 = _cast16(x)
        else:
            _cast16_into(x, x16[:, :Cin])
        w16 = PACK.get(weight, f"c3_{cin_p}_{cout_p}",
                       lambda w: packing.pack_conv3x3(w, cin_pad=cin_p, cout_pad=cout_p))
        b = None
        if bias is not None:
            b = torch.zeros(cout_p, dtype=_f32, device=x.device)
            b[:Cout] = bias.detach().float()
        y = torch.empty(Mout, cout_p, dtype=_f32, device=x.device)
        geo = dict(Cin=cin_p, Hi=Hi, Wi=Wi, Ho=Ho, Wo=Wo, stride=stride, upsample=int(upsample))
        ops.gemm(x16, w16, y, M=Mout, mode=GEMM_CONV3X3, bias=b, conv=geo)
        ctx.save_for_backward(x16, weight)
        ctx.geo, ctx.frames, ctx.dims = geo, frames, (Cin, Cout, cin_p, cout_p)
        ctx.has_bias = bias is not None
        return y[:, :Cout] if cout_p != Cout else y

    @staticmethod
    def backward(ctx, dy):
        x16, weight = ctx.saved_tensors
        Cin, Cout, cin_p, cout_p = ctx.dims
        geo, frames = ctx.geo, ctx.frames
        Mout = frames * geo["Ho"] * geo["Wo"]
        Min = frames * geo["Hi"] * geo["Wi"]
        dev = dy.device
        dyp = dy.contiguous()
        if cout_p != Cout:
            dyp = torch.zeros(Mout, cout_p, dtype=_f32, device=dev)
            dyp[:, :Cout] = dy
        lib = _lib.load()
        w16 = PACK.get(weight, f"c3_{cin_p}_{cout_p}",
                       lambda w: packing.pack_conv3x3(w, cin_pad=cin_p, cout_pad=cout_p))     # [cout_p, 9*cin_p]
        wt16 = PACK.get(weight, f"c3t_{cin_p}_{cout_p}", lambda w: w16.t().contiguous())      # [9*cin_p, cout_p]
        col = None
        if ctx.needs_input_grad[1]:
            col = torch.empty(Mout, 9 * cin_p, dtype=_f16, device=dev)
            check(lib.gcd_im2col3x3_f16(x16.data_ptr(), _ld(x16), col.data_ptr(), frames, cin_p, geo["Hi"],
                                        geo["Wi"], geo["Ho"], geo["Wo"], geo["stride"], geo["upsample"], 0,
                                        _stream()), "gcd_im2col3x3_f16")
        # the convolution as a Linear over im2col rows: dcol = dY W, dW = dY^T col
        dcol, dwp = _grad_contractions(dyp, col if col is not None else x16[:, :0], wt16,
                                       ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        dx = dw = db = None
        if dcol is not None:
            dxp = torch.empty(Min, cin_p, dtype=_f32, device=dev)
            check(lib.gcd_col2im3x3_f32(dcol.data_ptr(), dxp.data_ptr(), cin_p, frames, cin_p, geo["Hi"],
                                        geo["Wi"], geo["Ho"], geo["Wo"], geo["stride"], geo["upsample"], 0,
                                        _stream()), "gcd_col2im3x3_f32")
            dx = dxp[:, :Cin] if cin_p != Cin else dxp
        if dwp is not None:
            dw = dwp.reshape(cout_p, 3, 3, cin_p).permute(0, 3, 1, 2)[:Cout, :Cin].contiguous()
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = _colsum(dy.contiguous())[0]
        return dx, dw, db, None, None, None, None, None


def conv3x3(x, weight, bias, frames, Hi, Wi, stride=1, upsample=False):
    return Conv3x3.apply(x, weight, bias, frames, Hi, Wi, stride, upsample)


# ------------------------------------------------------------------------------------------------
# Conv3d (3,1,1) of the time_stack ResBlocks: a 3-tap GEMM over the frame axis, rows (clip, t, hw)
# ------------------------------------------------------------------------------------------------
class ConvT3(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, T, HW):
        ops._need_gpu(x, weight)
        M, Cc = x.shape
        Cout = weight.shape[0]
        if Cc % 32 or Cout % 32:
            raise NotImplementedError("ConvT3: channels must be multiples of 32")
        x16 = _cast16(x)
        w16 = PACK.get(weight, "t3", packing.pack_conv_t3)
        y = torch.empty(M, Cout, dtype=_f32, device=x.device)
        ops.gemm(x16, w16, y, M=M, mode=GEMM_TEMPORAL3, bias=None if bias is None else bias.detach().float(),
                 conv=dict(Cin=Cc, T=T, HW=HW))
        ctx.save_for_backward(x16, weight)
        ctx.T, ctx.HW, ctx.has_bias = T, HW, bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x16, weight = ctx.saved_tensors
        M, Cc = x16.shape
        Cout = weight.shape[0]
        dev = dy.device
        dy = dy.contiguous()
        lib = _lib.load()
        w16 = PACK.get(weight, "t3", packing.pack_conv_t3)
        wt16 = PACK.get(weight, "t3t", lambda w: w16.t().contiguous())            # [3C, Cout]
        col = None
        if ctx.needs_input_grad[1]:
            col = torch.empty(M, 3 * Cc, dtype=_f16, device=dev)
            check(lib.gcd_im2col_t3_f16(x16.data_ptr(), _ld(x16), col.data_ptr(), M, Cc, ctx.T, ctx.HW, _stream()),
                  "gcd_im2col_t3_f16")
        dcol, dwp = _grad_contractions(dy, col if col is not None else x16[:, :0], wt16,
                                       ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        dx = dw = db = None
        if dcol is not None:
            dx = torch.empty(M, Cc, dtype=_f32, device=dev)
            check(lib.gcd_col2im_t3_f32(dcol.data_ptr(), dx.data_ptr(), Cc, M, Cc, ctx.T, ctx.HW, _stream()),
                  "gcd_col2im_t3_f32")
        if dwp is not None:
            dw = dwp.reshape(Cout, 3, Cc).permute(0, 2, 1).reshape(Cout, Cc, 3, 1, 1).contiguous()
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = _colsum(dy)[0]
        return dx, dw, db, None, None


def conv_t3(x, weight, bias, T, HW):
    return ConvT3.apply(x, weight, bias, T, HW)


# ------------------------------------------------------------------------------------------------
# GroupNorm(32) [+ SiLU] and LayerNorm
# ------------------------------------------------------------------------------------------------
class GroupNormSiLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, rows_per_inst, eps, silu):
        ops._need_gpu(x)
        x = x.contiguous()
        M, Cc = x.shape
        ninst = M // rows_per_inst
        nch = ops.gn_nchunks(rows_per_inst, ninst)
        partial = torch.empty(ninst * nch * 64, dtype=torch.float64, device=x.device)
        stats = torch.empty(ninst * 64, dtype=_f32, device=x.device)
        ops.groupnorm_stats(x, None, rows_per_inst, eps, partial, stats, nch)
        y16 = torch.empty(M, Cc, dtype=_f16, device=x.device)
        g32, b32 = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
        ops.groupnorm_apply(x, None, rows_per_inst, stats, g32, b32, silu, y16)
        ctx.save_for_backward(x, stats, g32, b32)
        ctx.rows, ctx.silu = rows_per_inst, bool(silu)
        return y16.float()

    @staticmethod
    def backward(ctx, dy):
        x, stats, g32, b32 = ctx.saved_tensors
        M, Cc = x.shape
        ninst = M // ctx.rows
        dy = dy.contiguous()
        AB = torch.zeros(ninst, Cc, 2, dtype=torch.float64, device=x.device)
        dx = torch.empty_like(x)
        check(_lib.load().gcd_groupnorm_bwd(x.data_ptr(), _ld(x), dy.data_ptr(), _ld(dy), Cc, M, ctx.rows,
                                            stats.data_ptr(), g32.data_ptr(), b32.data_ptr(), int(ctx.silu),
                                            AB.data_ptr(), dx.data_ptr(), _ld(dx), _stream()),
              "gcd_groupnorm_bwd")
        ab = AB.sum(0).float()
        return dx, ab[:, 1].contiguous(), ab[:, 0].contiguous(), None, None, None


def group_norm(x, gamma, beta, rows_per_inst, eps=1e-5, silu=False):
    return GroupNormSiLU.apply(x, gamma, beta, rows_per_inst, eps, silu)


class LayerNorm16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        ops._need_gpu(x)
        x = x.contiguous()
        y16 = torch.empty(x.shape, dtype=_f16, device=x.device)
        g32, b32 = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
        ops.layernorm(x, g32, b32, y16, eps=eps)
        ctx.save_for_backward(x, g32)
        ctx.eps = eps
        return y16.float()

    @staticmethod
    def backward(ctx, dy):
        x, g32 = ctx.saved_tensors
        M, Cc = x.shape
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        dg = torch.zeros(Cc, dtype=_f32, device=x.device)
        db = torch.zeros(Cc, dtype=_f32, device=x.device)
        check(_lib.load().gcd_layernorm_bwd(x.data_ptr(), _ld(x), dy.data_ptr(), _ld(dy), M, Cc, g32.data_ptr(),
                                            ctx.eps, dx.data_ptr(), _ld(dx), dg.data_ptr(), db.data_ptr(),
                                            _stream()), "gcd_layernorm_bwd")
        return dx, dg, db, None


def layer_norm(x, gamma, beta, eps=1e-5):
    return LayerNorm16.apply(x, gamma, beta, eps)


# ------------------------------------------------------------------------------------------------
# GEGLU on the fp32 projection [value | gate]
# ------------------------------------------------------------------------------------------------
class Geglu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h):
        ops._need_gpu(h)
        h = h.contiguous()
        M, H2 = h.shape
        out = torch.empty(M, H2 // 2, dtype=_f32, device=h.device)
        check(_lib.load().gcd_geglu_fwd_f32(h.data_ptr(), _ld(h), out.data_ptr(), _ld(out), M, H2 // 2, _stream()),
              "gcd_geglu_fwd_f32")
        ctx.save_for_backward(h)
        return out

    @staticmethod
    def backward(ctx, dout):
        (h,) = ctx.saved_tensors
        M, H2 = h.shape
        dout = dout.contiguous()
        dh = torch.empty_like(h)
        check(_lib.load().gcd_geglu_bwd_f32(h.data_ptr(), _ld(h), dout.data_ptr(), _ld(dout), dh.data_ptr(),
                                            _ld(dh), M, H2 // 2, _stream()), "gcd_geglu_bwd_f32")
        return dh


def geglu(h):
    return Geglu.apply(h)


# ------------------------------------------------------------------------------------------------
# Self-attention over the H*W tokens of a frame (d = 64 per head); qkv = [q | k | v] rows of 3C.
# forward: the flash kernel of the inference path; backward: recompute P per (frame, head) with plain
# GEMMs, dV = P^T dO, dP = dO V^T, dS = P (dP - rowsum(P dP)) / 8, dQ = dS K, dK = dS^T Q.
# ------------------------------------------------------------------------------------------------
class SpatialAttention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, frames, S, heads):
        ops._need_gpu(qkv)
        M, C3 = qkv.shape
        Cc = C3 // 3
        assert M == frames * S and Cc == heads * 64
        qkv16 = _cast16(qkv.contiguous())
        S_pad = (S + 63) // 64 * 64
        vt = torch.empty(frames * heads * 64 * S_pad, dtype=_f16, device=qkv.device)
        ops.attn_transpose_v(qkv16, frames, S, heads, vt, S_pad)
        out16 = torch.empty(M, Cc, dtype=_f16, device=qkv.device)
        ops.attn_spatial(qkv16, vt, S_pad, out16, frames, S, heads, q_prescaled=False)
        ctx.save_for_backward(qkv16)
        ctx.dims = (frames, S, heads)
        return out16.float()

    @staticmethod
    def backward(ctx, dO):
        (qkv16,) = ctx.saved_tensors
        frames, S, heads = ctx.dims
        Cc = heads * 64
        dev = dO.device
        dO = dO.contiguous()
        if S % 4:
            raise NotImplementedError(f"SpatialAttention backward: {S} tokens per frame (needs a multiple of 4)")
        dqkv = torch.empty(frames * S, 3 * Cc, dtype=_f32, device=dev)
        lib = _lib.load()
        # the token axis is a GEMM contraction axis here: zero-pad it to the 64-deep granule (padded
        # keys never enter a softmax: it runs over the S x S corner only)
        Sp = (S + 63) // 64 * 64
        z16 = lambda *sh: torch.zeros(*sh, dtype=_f16, device=dev)      # noqa: E731
        qp, kp, vp, dOp = z16(Sp, 64), z16(Sp, 64), z16(Sp, 64), z16(Sp, 64)
        P, dS = z16(Sp, Sp), z16(Sp, Sp)
        Pt, dSt = torch.empty(Sp, Sp, dtype=_f16, device=dev), torch.empty(Sp, Sp, dtype=_f16, device=dev)
        scores = torch.empty(Sp, Sp, dtype=_f32, device=dev)
        dP = torch.empty(Sp, Sp, dtype=_f32, device=dev)
        kt, qt, dOt = (torch.empty(64, Sp, dtype=_f16, device=dev) for _ in range(3))
        for f in range(frames):
            rows = slice(f * S, (f + 1) * S)
            for h in range(heads):
                qp[:S] = qkv16[rows, h * 64:(h + 1) * 64]
                kp[:S] = qkv16[rows, Cc + h * 64:Cc + (h + 1) * 64]
                vp[:S] = qkv16[rows, 2 * Cc + h * 64:2 * Cc + (h + 1) * 64]
                ops.cast_f16(dO[rows, h * 64:(h + 1) * 64], dOp[:S])
                ops.gemm(qp, kp, scores, M=Sp, s_acc=0.125)
                ops.softmax_rows(scores[:S, :S], P[:S, :S])
                ops.transpose_f16(P, Pt)
                ops.transpose_f16(dOp, dOt)
                ops.gemm(Pt[:S], dOt, dqkv[rows, 2 * Cc + h * 64:2 * Cc + (h + 1) * 64], M=S)      # dV = P^T dO
                ops.gemm(dOp, vp, dP, M=Sp)                                                           # dP = dO V^T
                check(lib.gcd_softmax_bwd_rows(P.data_ptr(), Sp, dP.data_ptr(), Sp, dS.data_ptr(), Sp, S, S,
                                               0.125, _stream()), "gcd_softmax_bwd_rows")
                ops.transpose_f16(kp, kt)
                ops.transpose_f16(qp, qt)
                ops.transpose_f16(dS, dSt)
                ops.gemm(dS[:S], kt, dqkv[rows, h * 64:(h + 1) * 64], M=S)                            # dQ = dS K
                ops.gemm(dSt[:S], qt, dqkv[rows, Cc + h * 64:Cc + (h + 1) * 64], M=S)                 # dK = dS^T Q
        return dqkv, None, None, None


def spatial_attention(qkv, frames, S, heads):
    return SpatialAttention.apply(qkv, frames, S, heads)


class TemporalAttention(torch.autograd.Function):
    """Self-attention over the T frames of every pixel (video_attention.py:114-139 after the
    (b t) s c -> (b s) t c rearrange): rows stay (clip, t, hw)."""

    @staticmethod
    def forward(ctx, qkv, clips, T, HW, heads):
        ops._need_gpu(qkv)
        M, C3 = qkv.shape
        Cc = C3 // 3
        qkv16 = _cast16(qkv.contiguous())
        out16 = torch.empty(M, Cc, dtype=_f16, device=qkv.device)
        ops.attn_temporal(qkv16, out16, clips, T, HW, heads)
        ctx.save_for_backward(qkv16)
        ctx.dims = (clips, T, HW, heads)
        return out16.float()

    @staticmethod
    def backward(ctx, dO):
        (qkv16,) = ctx.saved_tensors
        clips, T, HW, heads = ctx.dims
        dO = dO.contiguous()
        dqkv = torch.empty(qkv16.shape, dtype=_f32, device=dO.device)
        check(_lib.load().gcd_attn_temporal_bwd(qkv16.data_ptr(), _ld(qkv16), dO.data_ptr(), _ld(dO),
                                                dqkv.data_ptr(), _ld(dqkv), clips, T, HW, heads, _stream()),
              "gcd_attn_temporal_bwd")
        return dqkv, None, None, None, None


def temporal_attention(qkv, clips, T, HW, heads):
    return TemporalAttention.apply(qkv, clips, T, HW, heads)


# ------------------------------------------------------------------------------------------------
# optimizer step
# ------------------------------------------------------------------------------------------------
def adam_step(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, step: int, lr: float,
              betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0, grad_scale: float = 1.0):
    """torch.optim.Adam semantics on flat fp32 tensors, in place (gcd_adam_step)."""
    ops._need_gpu(p, g, m, v)
    assert p.is_contiguous() and g.is_contiguous() and m.is_contiguous() and v.is_contiguous()
    check(_lib.load().gcd_adam_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), lr,
                                    betas[0], betas[1], eps, weight_decay, step, grad_scale, _stream()),
          "gcd_adam_step")
