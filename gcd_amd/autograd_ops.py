"""torch.autograd.Function wrappers that run BOTH directions of the VideoUNet's operators on
libgcd_amd kernels — the building blocks of the fine-tune step (BASELINE.json cfg4; SURVEY.md §8a a23,
§8(f)-2).  The reference trains through torch.autograd over nn.Conv2d / nn.Linear / F.sdpa
(loss.py:115-273 + Lightning's backward); here torch only keeps the graph: every forward and every
gradient is a hand-written gfx950 kernel through the C ABI.

Conventions
  * graph edges are fp32 token-major tensors [rows, channels] (gradients stay in fp32 between
    operators); MFMA operands are rounded to fp16 (or bf16: `set_train_dtype`) inside each operator,
    exactly as in the inference engine, with fp32 accumulation;
  * `Fused` is ONE node for [LayerNorm | GroupNorm(+SiLU) | GEGLU ->] Linear / q|k|v / Conv3x3 / Conv (3,1,1)
    [+ per-frame vector] [+ residual]: the prologue writes the 16-bit GEMM operand directly, the epilogue terms ride
    in the GEMM as in the inference engine;
  * backward contractions reuse `gcd_gemm_f16`: dX of a Linear = dY @ W (W^T packed once per parameter version), dX of
    a stride-1 convolution = the forward implicit-GEMM convolution on dY with mirrored taps (no col2im), dW = dY^T @ X
    with both operands transposed by the vector transpose kernel and the GEMM split up to 32 ways along the token
    axis (im2col feeds the convolutions' dW); dY is rounded and summed (bias / per-frame-vector gradients) in one pass;
  * the spatial attention backward is flash-style (gcd_attn_spatial_bwd, attn_bwd.hip): nothing S x S in memory;
  * static loss scaling as in AMP: the caller multiplies the loss by `loss_scale` so that the 16-bit casts of the
    gradients keep their small values; parameter gradients come out scaled and the optimizer step
    (`gcd_adam_step_multi(grad_scale=1/loss_scale)`) removes the factor.
Gradient parity: tests/test_backward_gpu.py — per operator vs fp32 torch, per block and for the whole network vs
torch.autograd over the CPU oracle, and the full-width step at cfg4's shape vs the unmodified reference classes
(7.0e-4).  DESIGN.md §11 has the measurements.
"""
from __future__ import annotations

import ctypes as C
import os
import weakref
from typing import Optional

import torch

from . import _lib, ops, packing
from ._lib import GEMM_CONV3X3, GEMM_PLAIN, GEMM_TEMPORAL3, OUT_F16, OUT_F32, check

_f32 = torch.float32
_f16 = torch.float16
_bf16 = torch.bfloat16

# Operand type of the GEMM-family contractions (every Linear / Conv2d / Conv3d product):
#   FWD_DTYPE   the forward pass,      GRAD_DTYPE   the backward pass (dX = dY W, dW = dY^T X)
# "fp16" (default; the inference engine's arithmetic) or "bf16" (BASELINE.json cfg4 names bf16:
# gcd_gemm_desc.operand_bf16 — v_mfma_f32_32x32x16_bf16 on the ping-pong kernel, same rate as fp16 on
# gfx950; fp32's exponent range at 8 significant bits).  The attention cores (softmax(QK^T)V and its
# backward) keep fp16 operands with an fp32 softmax in both modes, so the static loss scale stays.
FWD_DTYPE = os.environ.get("GCD_TRAIN_FWD_DTYPE", "fp16")
GRAD_DTYPE = os.environ.get("GCD_TRAIN_GRAD_DTYPE", "fp16")


def _chk_dtype(name: str) -> str:
    if name not in ("fp16", "bf16"):
        raise ValueError("dtype must be 'fp16' or 'bf16'")
    return name


def set_grad_dtype(name: str) -> None:
    """Operand type of the backward contractions only."""
    global GRAD_DTYPE
    GRAD_DTYPE = _chk_dtype(name)


def set_train_dtype(name: str) -> None:
    """Operand type of every GEMM-family contraction of the fine-tune step, forward and backward."""
    global FWD_DTYPE, GRAD_DTYPE
    FWD_DTYPE = GRAD_DTYPE = _chk_dtype(name)


def _dt(name: str) -> torch.dtype:
    return _bf16 if name == "bf16" else _f16


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _ld(t):
    assert t.dim() == 2 and t.stride(1) == 1
    return t.stride(0)


def _cast16(x32: torch.Tensor, dtype: torch.dtype = _f16) -> torch.Tensor:
    """fp32 [M, C] (any row stride) -> contiguous fp16 / bf16."""
    y = torch.empty(x32.shape, dtype=dtype, device=x32.device)
    (ops.cast_bf16 if dtype == _bf16 else ops.cast_f16)(x32, y)
    return y


def _cast16_into(x32: torch.Tensor, y16: torch.Tensor) -> None:
    """fp32 -> fp16 / bf16 into a (possibly wider) buffer's leading columns; the 4-channel ends of the UNet go
    through a plain copy (the cast kernel moves 8 channels per lane)."""
    if x32.shape[1] % 8 == 0:
        (ops.cast_bf16 if y16.dtype == _bf16 else ops.cast_f16)(x32, y16)
    else:
        y16.copy_(x32)


def _as_dtype(x16: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """A saved 16-bit operand in the other 16-bit type (forward and backward operand types differ)."""
    if x16.dtype == dtype:
        return x16
    if dtype == _bf16 and x16.is_contiguous():
        y = torch.empty(x16.shape, dtype=_bf16, device=x16.device)
        ops.cast_bf16(x16, y)
        return y
    return x16.to(dtype)


def _t16_padded(x16: torch.Tensor, granule: int = 64) -> torch.Tensor:
    """[R, C] fp16 / bf16 -> [C, Rp] with Rp = R rounded up to the GEMM's K granule and zero padding (a
    GEMM operand whose contraction axis is the token axis)."""
    R, Cc = x16.shape
    Rp = (R + granule - 1) // granule * granule
    out = torch.empty(Cc, Rp, dtype=x16.dtype, device=x16.device)
    if Rp != R:
        out[:, R:].zero_()
    ops.transpose_f16(x16, out[:, :R])
    return out


_FUSE_DY_SUMS = os.environ.get("GCD_TRAIN_FUSE_DY_SUMS", "1") != "0"      # A/B switch
# Weight gradients dW = dY^T X:  "gemm" = transposed copies of dY and X + the split-K gcd_gemm_f16 (round 3);
# "tr" = gcd_wgrad_tr_f16 (libgcd_amd_train.so, round 4): both operands stay row-major, transposed on the LDS read.
# Default "tr": same box, interleaved, forward + backward of the full-width step 0.1806 -> 0.1782 s, gradients equal to
# 2.7e-6 (profiles/r04w_wgrad_ab.txt); shapes the kernel does not take (N or K not multiples of 8) fall back to "gemm".
WGRAD_IMPL = "tr"


def set_wgrad_impl(name: str) -> None:
    global WGRAD_IMPL
    if name not in ("gemm", "tr"):
        raise ValueError(f"wgrad implementation must be 'gemm' or 'tr', got {name!r}")
    WGRAD_IMPL = name


set_wgrad_impl(os.environ.get("GCD_TRAIN_WGRAD", "tr"))     # a typo in the variable raises at import, it does not pick a path


# The planned engine (gcd_amd/train_plan.py) sets this to an object with `grad_dest(param) -> fp32 tensor of the parameter's
# shape | None` and `accumulate`: weight gradients are then written by the kernel STRAIGHT into that tensor, in the
# parameter's own layout (gcd_wgrad_tr_f16_ex) — no padded temporary, no permuted copy.  None on the autograd path.
# Round 6: not a module global any more.  The sink (and the fp16 pass-through switch below) live in a per-THREAD scope that a
# plan opens around the operator calls of ITS OWN backward / forward (`grad_sink(plan)`, `f16_passthrough(True)`): two plans,
# two host threads (one per GPU) or a DDP listener that re-enters the operators cannot see each other's setting.
import contextlib as _contextlib
import threading as _threading

_SCOPE = _threading.local()


def _sink():
    return getattr(_SCOPE, "sink", None)


@_contextlib.contextmanager
def grad_sink(plan):
    old = getattr(_SCOPE, "sink", None)
    _SCOPE.sink = plan
    try:
        yield plan
    finally:
        _SCOPE.sink = old


def _sink_dest(*params):
    """The flat-buffer destination of one parameter, or of several that lie back to back in it (q | k | v)."""
    _GRAD_SINK = _sink()
    if _GRAD_SINK is None:
        return None
    ds = [_GRAD_SINK.grad_dest(p) for p in params]
    if any(d is None for d in ds):
        return None
    for a, b in zip(ds, ds[1:]):
        if b.data_ptr() != a.data_ptr() + a.numel() * 4:
            return None
    return ds


def _wgrad(dy16: torch.Tensor, x16: torch.Tensor, dest: Optional[torch.Tensor] = None, taps: int = 1,
           n_real: Optional[int] = None, c_real: Optional[int] = None) -> torch.Tensor:
    """dW [N, K] fp32 = dY^T X for dy16 [M, N], x16 [M, K] (same 16-bit type, rows = tokens).
    dest (planned engine): write into this tensor instead — element [n][c][tap] of a parameter [n_real][c_real][taps],
    column k = tap * (K / taps) + c of the contraction (cropped to the real rows / channels); returns dest."""
    M, N = dy16.shape
    K = x16.shape[1]
    tr_ok = N % 8 == 0 and K % 8 == 0 and dy16.stride(1) == 1 and x16.stride(1) == 1 and \
        dy16.stride(0) % 8 == 0 and x16.stride(0) % 8 == 0 and dy16.data_ptr() % 16 == 0 and x16.data_ptr() % 16 == 0
    if dest is not None and tr_ok and dest.is_contiguous() and dest.data_ptr() % 16 == 0:
        n_real = N if n_real is None else n_real
        c_real = K // taps if c_real is None else c_real
        if taps > 1 or c_real % 4 == 0:
            lib = _lib.load_train()
            scratch = torch.empty(int(lib.gcd_wgrad_tr_scratch_floats(M, N, K)), dtype=_f32, device=dy16.device)
            _lib.check_train(lib.gcd_wgrad_tr_f16_ex(
                dy16.data_ptr(), dy16.stride(0), x16.data_ptr(), x16.stride(0), M, N, K, int(dy16.dtype == _bf16),
                dest.data_ptr(), c_real, taps, n_real, c_real, int(bool(_sink() is not None and _sink().accumulate)),
                scratch.data_ptr(), scratch.numel(), _stream()), "gcd_wgrad_tr_f16_ex")
            return dest
    dw = torch.empty(N, K, dtype=_f32, device=dy16.device)
    if WGRAD_IMPL == "tr" and tr_ok:
        lib = _lib.load_train()
        scratch = torch.empty(int(lib.gcd_wgrad_tr_scratch_floats(M, N, K)), dtype=_f32, device=dy16.device)
        _lib.check_train(lib.gcd_wgrad_tr_f16(dy16.data_ptr(), dy16.stride(0), x16.data_ptr(), x16.stride(0), M, N, K,
                                              int(dy16.dtype == _bf16), dw.data_ptr(), K, scratch.data_ptr(),
                                              scratch.numel(), _stream()), "gcd_wgrad_tr_f16")
        return dw
    _gemm(_t16_padded(dy16), _t16_padded(x16), dw, M=N)       # split-K over the tokens
    return dw
# A/B switch: the planned engine's convolution weight gradients gather X per tap inside the kernel (no im2col tensor)
WGRAD_IMPLICIT = os.environ.get("GCD_TRAIN_WGRAD_IMPLICIT", "1") != "0"


def _wgrad_conv(dy16: torch.Tensor, x16: torch.Tensor, dest: torch.Tensor, conv: int, n_real: int, c_real: int,
                Ho: int = 0, Wo: int = 0, T: int = 0, HW: int = 0) -> torch.Tensor:
    """Weight gradient of a stride-1 3x3 convolution (conv = 1) or the (3,1,1) temporal convolution (conv = 2) straight into
    `dest` (the parameter's [Cout, Cin, taps...] slot): x16 [M, Cp] is the convolution's input operand, dy16 [M, N]."""
    M, N = dy16.shape
    Cp = x16.shape[1]
    taps = 9 if conv == 1 else 3
    lib = _lib.load_train()
    scratch = torch.empty(int(lib.gcd_wgrad_tr_scratch_floats(M, N, taps * Cp)), dtype=_f32, device=dy16.device)
    _lib.check_train(lib.gcd_wgrad_conv_tr_f16(
        dy16.data_ptr(), dy16.stride(0), x16.data_ptr(), x16.stride(0), M, N, Cp, conv, Ho, Wo, T, HW,
        int(dy16.dtype == _bf16), dest.data_ptr(), n_real, c_real,
        int(bool(_sink() is not None and _sink().accumulate)), scratch.data_ptr(), scratch.numel(), _stream()),
        "gcd_wgrad_conv_tr_f16")
    return dest


_WS = {}


def _train_ws(device) -> torch.Tensor:
    """Split-K scratch of the fine-tune step: the weight gradients are few-tile GEMMs whose contraction runs
    over all tokens (gcd_gemm_f16 splits K up to 32 ways when the scratch holds the partial sums), 384 MB.
    One per (device, stream), like ops._splitk_ws: the partial sums of a launch live there until its reduce kernel
    has run, so two streams must not share it."""
    key = (torch.device(device).index or 0, ops._stream())
    w = _WS.get(key)
    if w is None:
        w = torch.empty(96 << 20, dtype=_f32, device=device)
        _WS[key] = w
    return w


_ATTN_WS = {}


def _attn_ws(device, nbytes: int) -> torch.Tensor:
    key = torch.device(device).index or 0
    w = _ATTN_WS.get(key)
    if w is None or w.numel() < nbytes:
        w = torch.empty(nbytes, dtype=torch.uint8, device=device)
        _ATTN_WS[key] = w
    return w


def _gemm(a16, w16, out, **kw):
    # sched = 2: the tile kernels from 128 tiles (gcd_gemm_desc.sched bit 1; 0.217 -> 0.210 s per step at cfg4's shape)
    return ops.gemm(a16, w16, out, operand_bf16=a16.dtype == _bf16, workspace=_train_ws(a16.device), sched=2, **kw)


def _cast16_colsum(dy32: torch.Tensor, dtype: torch.dtype, rows_per_block: Optional[int], out_sums=None, total=None):
    """(dy16, sums [M / rows, N]): the incoming gradient rounded for the GEMMs and its row-block column sums (bias /
    per-frame-vector gradients) in one pass (gcd_cast_colsum_f32); plain cast when no sums are wanted or the width
    is not a multiple of 8."""
    M, N = dy32.shape
    if rows_per_block is None or N % 8:
        return _cast16(dy32, dtype), (None if rows_per_block is None else _colsum(dy32, rows_per_block))
    y = torch.empty(M, N, dtype=dtype, device=dy32.device)
    # out_sums (planned engine): the bias's own slot of the flat gradient buffer, zeroed once per step — the kernel's
    # atomics land there directly (no fill, no copy)
    sums = out_sums if out_sums is not None else _zeros(M // rows_per_block, N, dy32.device)
    # total (planned engine): the bias's slot, receiving the sum over all row blocks in the same pass
    check(_lib.load().gcd_cast_colsum_f32(dy32.data_ptr(), _ld(dy32), y.data_ptr(), _ld(y), M, N, rows_per_block,
                                          sums.data_ptr(), int(dtype == _bf16), 0 if total is None else total.data_ptr(),
                                          _stream()), "gcd_cast_colsum_f32")
    return y, sums


def _grad_contractions(dy32: torch.Tensor, x16: torch.Tensor, w_t16, need_dx: bool, need_dw: bool, dy16=None, dw_dest=None):
    """The two contractions every Linear-shaped backward needs, on gcd_gemm_f16:
         dX [M, K] = dY [M, N] @ W [N, K]      (w_t16() -> W^T, [K, N], in the backward operand type)
         dW [N, K] = dY^T [N, M] @ X [M, K]    (x16 [M, K]; split-K over the tokens)
    in GRAD_DTYPE operands."""
    M, N = dy32.shape
    dev = dy32.device
    dt = _dt(GRAD_DTYPE)
    if dy16 is None:
        dy16 = _cast16(dy32, dt)
    dx = dw = None
    if need_dx:
        wt = w_t16(dt)
        K = wt.shape[0]
        dx = torch.empty(M, K, dtype=_f32, device=dev)
        if N % 32 == 0:
            _gemm(dy16, wt, dx, M=M)
        else:   # N a multiple of 16 only (the padded last conv): widen the contraction axis with zeros
            Np = (N + 63) // 64 * 64
            dyp = torch.zeros(M, Np, dtype=dt, device=dev)
            dyp[:, :N] = dy16
            wtp = torch.zeros(K, Np, dtype=dt, device=dev)
            wtp[:, :N] = wt
            _gemm(dyp, wtp, dx, M=M)
    if need_dw:
        dw = _wgrad(dy16, _as_dtype(x16, dt), dest=dw_dest)
    return dx, dw


def _zeros(m: int, n: int, device) -> torch.Tensor:
    """A zeroed fp32 [m, n] accumulator: from the planned engine's per-step arena (one memset per step) when it runs,
    else a fresh torch.zeros (one fill launch each)."""
    if _sink() is not None:
        z = _sink().zeros(m, n)
        if z is not None:
            return z
    return torch.zeros(m, n, dtype=_f32, device=device)


def _colsum(x32: torch.Tensor, rows_per_block: Optional[int] = None) -> torch.Tensor:
    M, N = x32.shape
    rows = M if rows_per_block is None else rows_per_block
    out = _zeros(M // rows, N, x32.device)
    check(_lib.load().gcd_rowblock_sum_f32(x32.data_ptr(), _ld(x32), M, N, rows, out.data_ptr(), _stream()),
          "gcd_rowblock_sum_f32")
    return out


class _PackCache:
    """fp16 / bf16 operand forms of the parameters of ATTACHED modules, rebuilt when a parameter changes
    (torch's version counter; `clear()` after an optimizer step that writes through raw pointers).

    Under activation checkpointing the backward pass sees a parameter through a detached alias, not the
    nn.Parameter object, so entries are looked up by storage address — but ONLY through the registry of
    `attach`: an address is trusted when a still-alive registered parameter of the same shape lives there
    (then the tensor in hand aliases that parameter's storage).  Anything else — temporaries such as a
    concatenated weight, tensors of an un-attached or rebuilt model — is packed and never kept: a freed
    temporary's address is handed out again by the caching allocator, and an address / version key alone
    would return the previous owner's weights."""

    def __init__(self):
        self._d = {}
        self._reg = {}          # data_ptr -> weakref to the nn.Parameter that owns the storage

    def attach(self, module: torch.nn.Module) -> None:
        """Register the module's parameters (idempotent, cheap: one dict probe per parameter)."""
        for q in module.parameters():
            r = self._reg.get(q.data_ptr())
            if r is None or r() is not q:
                self._reg[q.data_ptr()] = weakref.ref(q)

    def _owner(self, p: torch.Tensor):
        r = self._reg.get(p.data_ptr())
        q = None if r is None else r()
        # (a reshaped view of the whole parameter, e.g. a 1x1 convolution's [Cout, Cin, 1, 1] as [Cout, Cin], counts)
        if q is None or q.data_ptr() != p.data_ptr() or q.numel() != p.numel():
            return None
        return q

    def get(self, p: torch.Tensor, kind: str, fn):
        return self.get_multi((p,), kind, lambda ps: fn(ps[0]))

    def get_multi(self, ps, kind: str, fn):
        """One packed form of several parameters (e.g. the concatenated q|k|v weight): cached when every one
        of them is a registered parameter, keyed on all their addresses and versions."""
        owners = [self._owner(p) for p in ps]
        if any(q is None for q in owners):
            return fn([p.detach() for p in ps])
        key = (tuple(q.data_ptr() for q in owners), tuple(tuple(q.shape) for q in owners), kind)
        ver = tuple(q._version for q in owners)
        hit = self._d.get(key)
        if hit is not None and hit[0] == ver:
            return hit[1]
        v = fn([p.detach() for p in ps])
        self._d[key] = (ver, v)
        return v

    def clear(self):
        self._d.clear()


PACK = _PackCache()


# ------------------------------------------------------------------------------------------------
# The contractions.  Each has a forward and a backward helper on an already-prepared 16-bit activation
# `a16`; `Fused` below turns  [norm ->] contraction [+ per-frame vector] [+ residual]  into ONE graph node.
#   lin   y = a @ W^T + b                 (attention.py q/k/v/out, FeedForward, proj_in/out, emb MLPs)
#   qkv   q|k|v = a @ [Wq; Wk; Wv]^T      (one GEMM; attention.py:281-296 runs three)
#   c3    Conv2d 3x3 as implicit GEMM over token-major activations (stride 1 / 2, fused x2 upsample)
#   t3    Conv3d (3,1,1) of the time_stack ResBlocks: a 3-tap GEMM over the frame axis
# ------------------------------------------------------------------------------------------------
def _pack_lin(dt):
    return lambda w: w.detach().reshape(w.shape[0], -1).to(dt).contiguous()


def _pack_lin_t(dt):
    """[K, N] = W^T in the operand type: ONE rounding from the fp32 parameter (a contiguous cast), then the 16-byte-vector
    transpose kernel (torch's strided copy of `w.t().to(dt)` runs at a fraction of that; every Linear weight is repacked
    after every optimizer step)."""
    def pack(w):
        w16 = w.detach().reshape(w.shape[0], -1).to(dt)
        out = torch.empty(w16.shape[1], w16.shape[0], dtype=dt, device=w.device)
        if w16.is_cuda:
            ops.transpose_f16(w16.contiguous(), out)
            return out
        return w16.t().contiguous()
    return pack


def _pack_c3(dt, cin_p, cout_p):
    return lambda w: packing.pack_conv3x3(w, cin_pad=cin_p, cout_pad=cout_p, dtype=dt)


def _pack_c3_dgrad(dt, cin_p, cout_p):
    """The dgrad of a stride-1 3x3 convolution is a 3x3 convolution of dY with the taps mirrored and the
    channel roles exchanged: W'[cin][kh][kw][cout] = W[cout][cin][2-kh][2-kw]."""
    return lambda w: packing.pack_conv3x3(w.detach().permute(1, 0, 2, 3).flip(2, 3), cin_pad=cout_p,
                                          cout_pad=cin_p, dtype=dt)


def _pack_t3(dt):
    return lambda w: packing.pack_conv_t3(w, dtype=dt)


def _pack_t3_dgrad(dt):
    """dgrad of the (3,1,1) convolution = the same 3-tap GEMM over dY with W'[cin][kt][cout] = W[cout][cin][2-kt]."""
    return lambda w: packing.pack_conv_t3(w.detach().permute(1, 0, 2, 3, 4).flip(2), dtype=dt)


def _epi(bias, rowvec, residual):
    kw = {}
    if bias is not None:
        kw["bias"] = bias.detach().float()
    if rowvec is not None:
        v, rows = rowvec
        kw["rowvec"], kw["rows_per_vec"] = v.detach().float().contiguous(), rows
    if residual is not None:
        kw["r1"] = residual.detach().contiguous()
    return kw


def _c3_dims(weight, geo):
    Cout, Cin = weight.shape[0], weight.shape[1]
    cin_p = (Cin + 63) // 64 * 64 if Cin % 32 else Cin          # first conv: 8 -> 64 channels
    cout_p = (Cout + 31) // 32 * 32 if Cout % 32 else Cout      # last conv: 4 -> 32 channels
    return Cin, Cout, cin_p, cout_p


def _contract_fwd(kind, a16, params, geo, bias_vec_res):
    """-> y fp32 [Mout, N]"""
    dev = a16.device
    if kind == "lin":
        weight, bias = params
        M, K = a16.shape
        N = weight.shape[0]
        if K % 32 or N % 16:
            raise NotImplementedError(f"linear: K={K} must be a multiple of 32 and N={N} of 16")
        w16 = PACK.get(weight, f"lin_{a16.dtype}", _pack_lin(a16.dtype))
        y = torch.empty(M, N, dtype=_f32, device=dev)
        _gemm(a16, w16, y, M=M, **bias_vec_res)
        return y
    if kind == "qkv":
        M = a16.shape[0]
        dt = a16.dtype
        w16 = PACK.get_multi(params, f"qkv_{dt}", lambda ws: torch.cat([w.to(dt) for w in ws], 0).contiguous())
        if geo is not None and geo.get("out16") and dt == _f16 and not bias_vec_res:
            # planned engine: q | k | v leave the GEMM as the fp16 tensor the attention core reads (what the inference
            # engine does) — no fp32 result, no cast pass
            y = torch.empty(M, w16.shape[0], dtype=_f16, device=dev)
            _gemm(a16, w16, y, M=M, out_kind=OUT_F16)
            return y
        y = torch.empty(M, w16.shape[0], dtype=_f32, device=dev)
        _gemm(a16, w16, y, M=M, **bias_vec_res)
        return y
    if kind == "c3":
        weight, bias = params
        Cin, Cout, cin_p, cout_p = _c3_dims(weight, geo)
        frames, Ho, Wo = geo["frames"], geo["Ho"], geo["Wo"]
        Mout = frames * Ho * Wo
        w16 = PACK.get(weight, f"c3_{cin_p}_{cout_p}_{a16.dtype}", _pack_c3(a16.dtype, cin_p, cout_p))
        kw = dict(bias_vec_res)
        if cout_p != Cout:
            assert "r1" not in kw and "rowvec" not in kw
            if bias is not None:
                b = torch.zeros(cout_p, dtype=_f32, device=dev)
                b[:Cout] = bias.detach().float()
                kw["bias"] = b
        y = torch.empty(Mout, cout_p, dtype=_f32, device=dev)
        _gemm(a16, w16, y, M=Mout, mode=GEMM_CONV3X3, conv=dict(
            Cin=cin_p, Hi=geo["Hi"], Wi=geo["Wi"], Ho=Ho, Wo=Wo, stride=geo["stride"], upsample=geo["upsample"]), **kw)
        return y[:, :Cout] if cout_p != Cout else y
    if kind == "t3":
        weight, bias = params
        M, Cc = a16.shape
        Cout = weight.shape[0]
        if Cc % 32 or Cout % 32:
            raise NotImplementedError("conv_t3: channels must be multiples of 32")
        w16 = PACK.get(weight, f"t3_{a16.dtype}", _pack_t3(a16.dtype))
        y = torch.empty(M, Cout, dtype=_f32, device=dev)
        _gemm(a16, w16, y, M=M, mode=GEMM_TEMPORAL3, conv=dict(Cin=Cc, T=geo["T"], HW=geo["HW"]), **bias_vec_res)
        return y
    raise ValueError(kind)


def _contract_bwd(kind, dy, a16, params, geo, need_da, need_dw, dy16=None, db_pre=None):
    """dy fp32 contiguous -> (da fp32 | None, [parameter gradients in the order of `params`, bias included]).
    dy16 / db_pre: the gradient already rounded to the backward operand type and its column sums (the bias gradient),
    when the caller produced them in one pass."""
    dev = dy.device
    dt = _dt(GRAD_DTYPE)
    lib = _lib.load()

    def bias_grad(bias, wanted):
        if bias is None or not wanted:
            return None
        return db_pre if db_pre is not None else _colsum(dy)[0]
    if kind == "lin":
        weight, bias = params
        dst = _sink_dest(weight) if need_dw[0] else None
        da, dw = _grad_contractions(dy, a16, lambda d: PACK.get(weight, f"lin_t_{d}", _pack_lin_t(d)),
                                    need_da, need_dw[0], dy16,
                                    dw_dest=None if dst is None else dst[0].reshape(weight.shape[0], -1))
        if dw is not None:
            dw = dw.reshape(weight.shape)
        return da, [dw, bias_grad(bias, need_dw[1])]
    if kind == "qkv":
        dst = _sink_dest(*params) if all(need_dw) else None
        n = params[0].shape[0]
        da, dw = _grad_contractions(
            dy, a16,
            lambda d: PACK.get_multi(params, f"qkv_t_{d}",
                                     lambda ws: torch.cat([w.t().to(d) for w in ws], 1).contiguous()),
            need_da, any(need_dw), dy16,
            dw_dest=None if dst is None else torch.as_strided(dst[0], (3 * n, params[0].shape[1]), (params[0].shape[1], 1)))
        if dw is None:
            return da, [None, None, None]
        return da, [dw[:n], dw[n:2 * n], dw[2 * n:]]
    if kind == "c3":
        weight, bias = params
        Cin, Cout, cin_p, cout_p = _c3_dims(weight, geo)
        frames, Hi, Wi, Ho, Wo = geo["frames"], geo["Hi"], geo["Wi"], geo["Ho"], geo["Wo"]
        Mout, Min = frames * Ho * Wo, frames * Hi * Wi
        if cout_p != Cout:
            dyp = torch.zeros(Mout, cout_p, dtype=_f32, device=dev)
            dyp[:, :Cout] = dy
            dy16 = _cast16(dyp, dt)
        elif dy16 is None:
            dy16 = _cast16(dy, dt)
        da = dw = None
        if need_da and geo["stride"] == 1:
            # dgrad as an implicit-GEMM convolution of dY on the forward kernel (no dcol tensor, no col2im),
            # at the output resolution; a fused x2 upsample then sums every 2 x 2 block back onto its source
            wd = PACK.get(weight, f"c3d_{cin_p}_{cout_p}_{dt}", _pack_c3_dgrad(dt, cin_p, cout_p))
            dxo = torch.empty(Mout, cin_p, dtype=_f32, device=dev)
            _gemm(dy16, wd, dxo, M=Mout, mode=GEMM_CONV3X3,
                  conv=dict(Cin=cout_p, Hi=Ho, Wi=Wo, Ho=Ho, Wo=Wo, stride=1, upsample=0))
            if geo["upsample"]:
                dxo = dxo.reshape(frames, Hi, 2, Wi, 2, cin_p).sum(dim=(2, 4)).reshape(Min, cin_p)
            da = dxo[:, :Cin] if cin_p != Cin else dxo
        col = None
        # planned engine, stride-1 same-size convolutions: the weight gradient gathers its X operand per tap inside the
        # kernel (gcd_wgrad_conv_tr_f16) — no im2col tensor
        dst = _sink_dest(weight) if need_dw[0] else None
        implicit = dst is not None and geo["stride"] == 1 and not geo["upsample"] and WGRAD_IMPLICIT
        if (need_dw[0] and not implicit) or (need_da and da is None):
            xg = _as_dtype(a16, dt)
            col = torch.empty(Mout, 9 * cin_p, dtype=dt, device=dev)
            check(lib.gcd_im2col3x3_f16(xg.data_ptr(), _ld(xg), col.data_ptr(), frames, cin_p, Hi, Wi, Ho, Wo,
                                        geo["stride"], geo["upsample"], 0, _stream()), "gcd_im2col3x3_f16")
        if need_da and da is None:
            # stride 2 (the three Downsample convs): dcol = dY W by a plain GEMM, then the col2im gather
            wt = PACK.get(weight, f"c3t_{cin_p}_{cout_p}_{dt}",
                          lambda w: _pack_c3(dt, cin_p, cout_p)(w).t().contiguous())      # [9*cin_p, cout_p]
            dcol = torch.empty(Mout, 9 * cin_p, dtype=_f32, device=dev)
            _gemm(dy16, wt, dcol, M=Mout)
            dxp = torch.empty(Min, cin_p, dtype=_f32, device=dev)
            check(lib.gcd_col2im3x3_f32(dcol.data_ptr(), dxp.data_ptr(), cin_p, frames, cin_p, Hi, Wi, Ho, Wo,
                                        geo["stride"], geo["upsample"], 0, _stream()), "gcd_col2im3x3_f32")
            da = dxp[:, :Cin] if cin_p != Cin else dxp
        if need_dw[0]:
            if implicit:
                dw = _wgrad_conv(dy16, _as_dtype(a16, dt), dst[0], 1, Cout, Cin, Ho=Ho, Wo=Wo)
            elif dst is not None:           # straight into the parameter's [Cout, Cin, 3, 3] slot, cropped
                dw = _wgrad(dy16, col, dest=dst[0], taps=9, n_real=Cout, c_real=Cin)
            else:
                dwp = _wgrad(dy16, col)       # dW = dY^T col [cout_p, 9 * cin_p], contraction over the tokens
                dw = dwp.reshape(cout_p, 3, 3, cin_p).permute(0, 3, 1, 2)[:Cout, :Cin].contiguous()
        return da, [dw, bias_grad(bias, need_dw[1])]
    if kind == "t3":
        weight, bias = params
        M, Cc = a16.shape
        Cout = weight.shape[0]
        if dy16 is None:
            dy16 = _cast16(dy, dt)
        da = dw = None
        if need_da:
            wd = PACK.get(weight, f"t3d_{dt}", _pack_t3_dgrad(dt))          # [Cin, 3*Cout]
            da = torch.empty(M, Cc, dtype=_f32, device=dev)
            _gemm(dy16, wd, da, M=M, mode=GEMM_TEMPORAL3, conv=dict(Cin=Cout, T=geo["T"], HW=geo["HW"]))
        dst = _sink_dest(weight) if need_dw[0] else None
        if need_dw[0] and dst is not None and WGRAD_IMPLICIT:
            dw = _wgrad_conv(dy16, _as_dtype(a16, dt), dst[0], 2, Cout, Cc, T=geo["T"], HW=geo["HW"])
        elif need_dw[0]:
            xg = _as_dtype(a16, dt)
            col = torch.empty(M, 3 * Cc, dtype=dt, device=dev)
            check(lib.gcd_im2col_t3_f16(xg.data_ptr(), _ld(xg), col.data_ptr(), M, Cc, geo["T"], geo["HW"],
                                        _stream()), "gcd_im2col_t3_f16")
            if dst is not None:
                dw = _wgrad(dy16, col, dest=dst[0], taps=3, n_real=Cout, c_real=Cc)
            else:
                dwp = _wgrad(dy16, col)
                dw = dwp.reshape(Cout, 3, Cc).permute(0, 2, 1).reshape(Cout, Cc, 3, 1, 1).contiguous()
        return da, [dw, bias_grad(bias, need_dw[1])]
    raise ValueError(kind)


# ---- norms on raw tensors (the standalone Functions further down and `Fused` share them) ----
def _gn_fwd(x, gamma, beta, rows_per_inst, eps, silu, dtype=_f16):
    """-> the normalised activation as the 16-bit GEMM operand `dtype` (bf16: written directly, one rounding, ABI v7)."""
    M, Cc = x.shape
    ninst = M // rows_per_inst
    nch = ops.gn_nchunks(rows_per_inst, ninst)
    partial = torch.empty(ninst * nch * 64, dtype=torch.float64, device=x.device)
    stats = torch.empty(ninst * 64, dtype=_f32, device=x.device)
    ops.groupnorm_stats(x, None, rows_per_inst, eps, partial, stats, nch)
    y16 = torch.empty(M, Cc, dtype=dtype, device=x.device)
    g32, b32 = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
    ops.groupnorm_apply(x, None, rows_per_inst, stats, g32, b32, silu, y16)
    return y16, stats, g32, b32


def _gn_bwd(x, dy, stats, g32, b32, rows_per_inst, silu, dest=None, dx_add=None):
    """dest (planned engine): (dgamma, dbeta) slots of the flat gradient buffer, written by one small kernel."""
    M, Cc = x.shape
    ninst = M // rows_per_inst
    lib = _lib.load()
    AB = torch.empty(ninst, Cc, 2, dtype=torch.float64, device=x.device)
    # per-chunk partial sums of the reduction pass (no atomics, nothing to zero): a few hundred KB
    scratch = torch.empty(int(lib.gcd_groupnorm_bwd_scratch_floats(Cc, M, rows_per_inst)), dtype=_f32, device=x.device)
    dx = torch.empty_like(x)
    check(lib.gcd_groupnorm_bwd(x.data_ptr(), _ld(x), dy.data_ptr(), _ld(dy), Cc, M, rows_per_inst,
                                stats.data_ptr(), g32.data_ptr(), b32.data_ptr(), int(silu),
                                AB.data_ptr(), scratch.data_ptr(), scratch.numel(), dx.data_ptr(), _ld(dx),
                                0 if dx_add is None else dx_add.data_ptr(), 0 if dx_add is None else _ld(dx_add),
                                _stream()),
          "gcd_groupnorm_bwd")
    if dest is not None and dest[0] is not None and dest[1] is not None:
        _lib.check_train(_lib.load_train().gcd_gn_affine_grads(
            AB.data_ptr(), ninst, Cc, dest[0].data_ptr(), dest[1].data_ptr(),
            int(bool(_sink() is not None and _sink().accumulate)), _stream()), "gcd_gn_affine_grads")
        return dx, dest[0], dest[1]
    ab = AB.sum(0).float()
    return dx, ab[:, 1].contiguous(), ab[:, 0].contiguous()


def _ln_fwd(x, gamma, beta, eps, dtype=_f16):
    y16 = torch.empty(x.shape, dtype=dtype, device=x.device)
    g32, b32 = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
    ops.layernorm(x, g32, b32, y16, eps=eps)
    return y16, g32


def _ln_bwd(x, dy, g32, eps, dest=None, dx_add=None):
    """dest (planned engine): (dgamma, dbeta) slots of the flat gradient buffer (zeroed once per step): the kernel's
    atomics accumulate there directly."""
    M, Cc = x.shape
    dx = torch.empty_like(x)
    if dest is not None and dest[0] is not None and dest[1] is not None:
        dg, db = dest
    else:
        dgb = torch.zeros(2, Cc, dtype=_f32, device=x.device)      # one fill for both accumulators
        dg, db = dgb[0], dgb[1]
    check(_lib.load().gcd_layernorm_bwd(x.data_ptr(), _ld(x), dy.data_ptr(), _ld(dy), M, Cc, g32.data_ptr(),
                                        eps, dx.data_ptr(), _ld(dx), dg.data_ptr(), db.data_ptr(),
                                        0 if dx_add is None else dx_add.data_ptr(), 0 if dx_add is None else _ld(dx_add),
                                        _stream()), "gcd_layernorm_bwd")
    return dx, dg, db


class Fused(torch.autograd.Function):
    """[norm ->] contraction [+ per-frame vector] [+ residual] as ONE graph node.

    apply(spec, x, residual, rowvec, gamma, beta, *params)
      spec: dict(kind = lin | qkv | c3 | t3, geo = {...}, norm = None | ("ln", eps) | ("gn", rows_per_inst,
            eps, silu), rows_per_vec)
    The normalised activation goes from the norm kernel to the GEMM as the 16-bit operand it is (no fp32
    round trip, no cast pass); bias, the per-frame vector (emb_layers output, openaimodel.py:343-347) and the
    residual ride in the GEMM epilogue as in the inference engine.  Backward: the contraction's dgrad / wgrad,
    the norm's backward, d residual = dy, d rowvec = per-frame row sums of dy."""

    @staticmethod
    def forward(ctx, spec, x, residual, rowvec, gamma, beta, *params):
        ops._need_gpu(x)
        kind, geo, norm = spec["kind"], spec.get("geo"), spec.get("norm")
        dt = _dt(FWD_DTYPE)
        stats = g32 = b32 = None
        if norm is not None and norm[0] == "geglu":
            # x is the fp32 projection [value | gate] of a FeedForward (attention.py:87-97): value * gelu(gate),
            # rounded once to the operand type, straight into the second Linear
            x = x.contiguous()
            M, H2 = x.shape
            # (bf16 operands: rounded ONCE from fp32 — no fp16 hop, none of fp16's range on the hidden tensor)
            a16 = torch.empty(M, H2 // 2, dtype=dt, device=x.device)
            fn = _lib.load().gcd_geglu_fwd_bf16 if dt == _bf16 else _lib.load().gcd_geglu_fwd_f16
            check(fn(x.data_ptr(), _ld(x), a16.data_ptr(), _ld(a16), M, H2 // 2, _stream()), "gcd_geglu_fwd_16")
        elif norm is not None:
            x = x.contiguous()
            if norm[0] == "ln":
                a16, g32 = _ln_fwd(x, gamma, beta, norm[1], dt)
            else:
                a16, stats, g32, b32 = _gn_fwd(x, gamma, beta, norm[1], norm[2], norm[3], dt)
        elif _f16_passthrough_on() and getattr(x, "_gcd_f16", None) is not None and x._gcd_f16[1] == x._version and \
                x._gcd_f16[0].shape == x.shape:
            # x is the fp32 image of an fp16 tensor an attention core produced: that tensor IS the operand
            a16 = _as_dtype(x._gcd_f16[0], dt)
        elif kind == "c3" and _c3_dims(params[0], geo)[2] != x.shape[1]:
            a16 = torch.zeros(x.shape[0], _c3_dims(params[0], geo)[2], dtype=dt, device=x.device)
            _cast16_into(x, a16[:, :x.shape[1]])
        else:
            a16 = _cast16(x, dt)
        bias = None if kind == "qkv" else params[1]
        y = _contract_fwd(kind, a16, params, geo,
                          _epi(bias, None if rowvec is None else (rowvec, spec["rows_per_vec"]), residual))
        ctx.spec = spec
        ctx.n_params = len(params)
        ctx.has = (residual is not None, rowvec is not None, bias is not None)
        tensors = [a16] + [p for p in params if p is not None]
        if norm is not None and norm[0] == "geglu":
            tensors += [x]
        elif norm is not None:
            tensors += [x, g32] + ([stats, b32] if norm[0] == "gn" else [])
        ctx.save_for_backward(*tensors)
        return y

    @staticmethod
    def backward(ctx, dy):
        spec = ctx.spec
        kind, geo, norm = spec["kind"], spec.get("geo"), spec.get("norm")
        has_res, has_vec, has_bias = ctx.has
        saved = list(ctx.saved_tensors)
        a16 = saved.pop(0)
        if kind == "qkv":
            params = tuple(saved[:3])
            saved = saved[3:]
        else:
            weight = saved.pop(0)
            bias = saved.pop(0) if has_bias else None
            params = (weight, bias)
        dy = dy.contiguous()
        nig = ctx.needs_input_grad          # (spec, x, residual, rowvec, gamma, beta, *params)
        need_x = nig[1]
        need_norm_params = norm is not None and (nig[4] or nig[5])
        need_da = need_x or need_norm_params
        need_dw = list(nig[6:6 + ctx.n_params])
        # ONE pass over dY: rounded to the backward operand type for the dgrad / wgrad GEMMs, and summed per row block
        # for the bias gradient (one block) / the per-frame vector's gradient (one block per `rows_per_vec` rows)
        want_db = has_bias and kind != "qkv" and need_dw[1]
        want_vec = has_vec and nig[3]
        dy16 = db_pre = d_vec = None
        bias_dest = getattr(ctx, "bias_dest", None)        # planned engine: slots of the flat gradient buffer
        norm_dest = getattr(ctx, "norm_dest", None)
        if (want_db or want_vec) and _FUSE_DY_SUMS and dy.shape[1] % 8 == 0:
            rows = spec["rows_per_vec"] if want_vec else dy.shape[0]
            both = want_vec and want_db and bias_dest is not None
            dy16, sums = _cast16_colsum(dy, _dt(GRAD_DTYPE), rows,
                                        bias_dest.view(1, -1) if bias_dest is not None and want_db and not want_vec else None,
                                        bias_dest if both else None)
            if want_vec:
                d_vec = sums
            if want_db:
                db_pre = bias_dest if both else (sums.sum(0) if want_vec else sums[0])
        da, dps = _contract_bwd(kind, dy, a16, params, geo, need_da, need_dw, dy16, db_pre)
        dgamma = dbeta = None
        if norm is None:
            dx = da
        elif da is None:
            dx = None
        elif norm[0] == "geglu":
            (h,) = saved
            da = da.contiguous()
            dx = torch.empty_like(h)
            check(_lib.load().gcd_geglu_bwd_f32(h.data_ptr(), _ld(h), da.data_ptr(), _ld(da), dx.data_ptr(),
                                                _ld(dx), h.shape[0], h.shape[1] // 2, _stream()), "gcd_geglu_bwd_f32")
        elif norm[0] == "ln":
            x, g32 = saved
            dx, dgamma, dbeta = _ln_bwd(x, da.contiguous(), g32, norm[1], norm_dest, getattr(ctx, "dx_add", None))
        else:
            x, g32, stats, b32 = saved
            dx, dgamma, dbeta = _gn_bwd(x, da.contiguous(), stats, g32, b32, norm[1], norm[3], norm_dest,
                                        getattr(ctx, "dx_add", None))
        d_res = dy if has_res and nig[2] else None
        if want_vec and d_vec is None:
            d_vec = _colsum(dy, spec["rows_per_vec"])
        return (None, dx if need_x else None, d_res, d_vec, dgamma, dbeta, *dps)


def _fused(kind, x, params, geo=None, norm=None, residual=None, rowvec=None):
    """norm: None | ("ln", module, eps) | ("gn", module, rows_per_inst, eps, silu) | ("geglu",): the prologue applied
    to x; rowvec: (vector [n, N], rows)."""
    spec = dict(kind=kind, geo=geo, norm=None, rows_per_vec=None if rowvec is None else rowvec[1])
    gamma = beta = None
    if norm is not None and norm[0] == "geglu":
        spec["norm"] = ("geglu",)
    elif norm is not None:
        gamma, beta = norm[1].weight, norm[1].bias
        spec["norm"] = (norm[0],) + tuple(norm[2:])
    return Fused.apply(spec, x, residual, None if rowvec is None else rowvec[0], gamma, beta, *params)


def linear(x, weight, bias=None, *, norm=None, residual=None, rowvec=None):
    """y = [norm](x) @ W^T + b [+ rowvec per `rows` rows] [+ residual]; weight may be a 1x1 convolution's
    [Cout, Cin, 1, 1]."""
    return _fused("lin", x, (weight, bias), norm=norm, residual=residual, rowvec=rowvec)


def qkv_linear(x, wq, wk, wv, *, norm=None):
    return _fused("qkv", x, (wq, wk, wv), norm=norm)


def conv3x3(x, weight, bias, frames, Hi, Wi, stride=1, upsample=False, *, norm=None, residual=None, rowvec=None):
    if upsample:
        Ho, Wo = 2 * Hi, 2 * Wi
    else:
        Ho, Wo = (Hi - 1) // stride + 1, (Wi - 1) // stride + 1
    assert x.shape[0] == frames * Hi * Wi and x.shape[1] == weight.shape[1]
    geo = dict(frames=frames, Hi=Hi, Wi=Wi, Ho=Ho, Wo=Wo, stride=stride, upsample=int(upsample))
    return _fused("c3", x, (weight, bias), geo=geo, norm=norm, residual=residual, rowvec=rowvec)


def conv_t3(x, weight, bias, T, HW, *, norm=None, residual=None, rowvec=None):
    return _fused("t3", x, (weight, bias), geo=dict(T=T, HW=HW), norm=norm, residual=residual, rowvec=rowvec)


# ------------------------------------------------------------------------------------------------
# GroupNorm(32) [+ SiLU] and LayerNorm
# ------------------------------------------------------------------------------------------------
class GroupNormSiLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, rows_per_inst, eps, silu):
        ops._need_gpu(x)
        x = x.contiguous()
        y16, stats, g32, b32 = _gn_fwd(x, gamma, beta, rows_per_inst, eps, silu)
        ctx.save_for_backward(x, stats, g32, b32)
        ctx.rows, ctx.silu = rows_per_inst, bool(silu)
        return y16.float()

    @staticmethod
    def backward(ctx, dy):
        x, stats, g32, b32 = ctx.saved_tensors
        dx, dg, db = _gn_bwd(x, dy.contiguous(), stats, g32, b32, ctx.rows, ctx.silu)
        return dx, dg, db, None, None, None


def group_norm(x, gamma, beta, rows_per_inst, eps=1e-5, silu=False):
    return GroupNormSiLU.apply(x, gamma, beta, rows_per_inst, eps, silu)


class LayerNorm16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        ops._need_gpu(x)
        x = x.contiguous()
        y16, g32 = _ln_fwd(x, gamma, beta, eps)
        ctx.save_for_backward(x, g32)
        ctx.eps = eps
        return y16.float()

    @staticmethod
    def backward(ctx, dy):
        x, g32 = ctx.saved_tensors
        dx, dg, db = _ln_bwd(x, dy.contiguous(), g32, ctx.eps)
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
# forward: the flash kernel of the inference path; backward: the flash-style kernels of attn_bwd.hip
# (dV = P^T dO, dP = dO V^T, dS = P (dP - rowsum(dO O)) / 8, dQ = dS K, dK = dS^T Q, P recomputed per tile).
# ------------------------------------------------------------------------------------------------
class SpatialAttention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, frames, S, heads):
        ops._need_gpu(qkv)
        M, C3 = qkv.shape
        Cc = C3 // 3
        assert M == frames * S and Cc == heads * 64
        qkv16 = qkv if qkv.dtype == _f16 and qkv.is_contiguous() else _cast16(qkv.contiguous())
        S_pad = (S + 63) // 64 * 64
        vt = torch.empty(frames * heads * 64 * S_pad, dtype=_f16, device=qkv.device)
        ops.attn_transpose_v(qkv16, frames, S, heads, vt, S_pad)
        out16 = torch.empty(M, Cc, dtype=_f16, device=qkv.device)
        ops.attn_spatial(qkv16, vt, S_pad, out16, frames, S, heads, q_prescaled=False)
        ctx.save_for_backward(qkv16, out16)
        ctx.dims = (frames, S, heads)
        if getattr(ctx, "want16", False):      # planned engine: the fp16 result IS the next Linear's operand (no fp32 image)
            return out16
        _LAST_F16[0] = out16
        return out16.float()

    @staticmethod
    def backward(ctx, dO):
        """Flash-style backward on gcd_attn_spatial_bwd (attn_bwd.hip): P is recomputed tile by tile from a
        per-query log-sum-exp, one launch sequence for all (frame, head) pairs."""
        qkv16, out16 = ctx.saved_tensors
        frames, S, heads = ctx.dims
        dO16 = _cast16(dO.contiguous())
        dqkv = torch.empty(frames * S, 3 * heads * 64, dtype=_f32, device=dO.device)
        ws = _attn_ws(dO.device, ops.attn_spatial_bwd_ws_bytes(frames, S, heads))
        ops.attn_spatial_bwd(qkv16, out16, dO16, dqkv, frames, S, heads, ws)
        return dqkv, None, None, None


_LAST_F16 = [None]
# Off by default: measured (same box, interleaved; profiles/r03b_train_ab.txt) the step is 5 ms SLOWER with it — the
# forward's 16 attention outputs go through one cast less, but holding the fp16 tensors through the fp32 edge's
# lifetime costs more than the casts save.
_F16_PASSTHROUGH_DEFAULT = os.environ.get("GCD_TRAIN_F16_PASSTHROUGH", "0") != "0"


def _f16_passthrough_on() -> bool:
    return getattr(_SCOPE, "f16_passthrough", _F16_PASSTHROUGH_DEFAULT)


@_contextlib.contextmanager
def f16_passthrough(on: bool = True):
    """Scope (per thread) in which a Linear takes an input that already exists as the 16-bit operand without a cast."""
    old = getattr(_SCOPE, "f16_passthrough", None)
    _SCOPE.f16_passthrough = on
    try:
        yield
    finally:
        if old is None:
            del _SCOPE.f16_passthrough
        else:
            _SCOPE.f16_passthrough = old


def _with_f16(y: torch.Tensor) -> torch.Tensor:
    """Tag the fp32 result of an attention core with the fp16 tensor it is the image of, so that the Linear that
    consumes it takes the fp16 tensor as its operand instead of casting the fp32 copy back."""
    y._gcd_f16 = (_LAST_F16[0], y._version)
    _LAST_F16[0] = None
    return y


def spatial_attention(qkv, frames, S, heads):
    return _with_f16(SpatialAttention.apply(qkv, frames, S, heads))


class TemporalAttention(torch.autograd.Function):
    """Self-attention over the T frames of every pixel (video_attention.py:114-139 after the
    (b t) s c -> (b s) t c rearrange): rows stay (clip, t, hw)."""

    @staticmethod
    def forward(ctx, qkv, clips, T, HW, heads):
        ops._need_gpu(qkv)
        M, C3 = qkv.shape
        Cc = C3 // 3
        qkv16 = qkv if qkv.dtype == _f16 and qkv.is_contiguous() else _cast16(qkv.contiguous())
        out16 = torch.empty(M, Cc, dtype=_f16, device=qkv.device)
        ops.attn_temporal(qkv16, out16, clips, T, HW, heads)
        ctx.save_for_backward(qkv16)
        ctx.dims = (clips, T, HW, heads)
        if getattr(ctx, "want16", False):
            return out16
        _LAST_F16[0] = out16
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
    return _with_f16(TemporalAttention.apply(qkv, clips, T, HW, heads))


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
