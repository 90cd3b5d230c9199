"""Tensor-level wrappers over the libgcd_amd C ABI.

torch is used here for device memory and the current HIP stream only; every function launches one
hand-written gfx950 kernel through ctypes and raises if the tensors are not on a HIP device.
Token-major convention: an activation of `frames x H x W x C` is a 2-D tensor [frames*H*W, C].
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

from . import _lib
from ._lib import (GEMM_CONV3X3, GEMM_PLAIN, GEMM_TEMPORAL3, OUT_F16, OUT_F32, OUT_GEGLU, FfDesc, GemmDesc,
                   check)

_zero_pages = {}

# Optional per-launch profiler (bench.py): a list that receives one record per GEMM / attention
# launch with HIP events recorded on the launch stream.  None = off (the default, zero overhead).
_prof = None


def start_profile() -> list:
    global _prof
    _prof = []
    return _prof


def stop_profile() -> None:
    global _prof
    _prof = None


class _Timed:
    """Brackets one kernel launch with HIP events on the current stream (profiling only)."""

    def __init__(self, kind: str, flops: float, **info):
        self.rec = None
        if _prof is not None:
            lib = _lib.load()
            e0, e1 = C.c_void_p(), C.c_void_p()
            check(lib.gcd_event_create(C.byref(e0)))
            check(lib.gcd_event_create(C.byref(e1)))
            self.rec = dict(kind=kind, flops=flops, start=e0, stop=e1, **info)

    def __enter__(self):
        if self.rec is not None:
            check(_lib.load().gcd_event_record(self.rec["start"], _stream()))
        return self

    def __exit__(self, *exc):
        if self.rec is not None:
            check(_lib.load().gcd_event_record(self.rec["stop"], _stream()))
            _prof.append(self.rec)
        return False


TUNE_GEMM_IMPL, TUNE_ATTN_IMPL = 0, 1


def tune_set(knob: int, value: int) -> None:
    """Kernel-selection knob of libgcd_amd (gcd_tune_set): used by the tests to force every GEMM
    kernel over the same cases, and by the A/B tools."""
    check(_lib.load().gcd_tune_set(knob, value), "gcd_tune_set")


def _need_gpu(*ts) -> None:
    """Every kernel launches on the CURRENT device's current stream (`_stream`): operands must be GPU tensors of
    that device.  A tensor of another GPU is an error here, not a silent launch on the wrong device — wrap the call
    in `torch.cuda.device(t.device)` (engine.run does that itself)."""
    cur = None
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise _lib.GcdError(
                "gcd_amd kernels run on an AMD GPU only (got a CPU tensor); there is no CPU fallback"
            )
        if cur is None:
            cur = torch.cuda.current_device()
        if t.device.index != cur:
            raise _lib.GcdError(
                f"operand on cuda:{t.device.index} but the current device is cuda:{cur}: gcd_amd launches on the "
                "current device's current stream; make the operand's device current (torch.cuda.device(...))")


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def zero_page(device) -> torch.Tensor:
    key = torch.device(device).index or 0
    z = _zero_pages.get(key)
    if z is None:
        z = torch.zeros(1024, dtype=torch.float16, device=device)
        _zero_pages[key] = z
    return z


_splitk = {}
_SPLITK_ON = os.environ.get("GCD_SPLITK", "1") != "0"   # A/B switch
_COLSTATS_ON = os.environ.get("GCD_COLSTATS", "1") != "0"   # A/B switch: GroupNorm statistics from GEMM epilogues


def _splitk_ws(device) -> torch.Tensor:
    """Persistent split-K scratch (gcd_gemm_desc.workspace): 4 partial outputs of the largest few-tile
    problem of the SVD UNet, 4032 tokens x 1280 channels fp32 = 83 MB.  One per (device, stream): the
    partial sums of a launch live there until its reduce kernel has run, so two streams must not
    share it.  Entries are never dropped (captured hipGraphs hold the pointer); torch hands out stream
    handles from a fixed pool, which bounds their number."""
    key = (torch.device(device).index or 0, _stream())
    w = _splitk.get(key)
    if w is None:
        w = torch.empty(4 * 4096 * 1280, dtype=torch.float32, device=device)
        _splitk[key] = w
    return w


def _ld(t: torch.Tensor) -> int:
    assert t.dim() == 2 and t.stride(1) == 1, "expected a row-major 2-D view"
    return t.stride(0)


def gemm(a16: torch.Tensor, w16: torch.Tensor, out: torch.Tensor, *, M: int, mode: int = GEMM_PLAIN,
         out_kind: int = OUT_F32, bias=None, rowvec=None, rows_per_vec: int = 1, r1=None, r2=None,
         s_acc: float = 1.0, s_r1: float = 1.0, s_r2: float = 1.0, frame_alpha=None,
         rows_per_alpha: int = 1, r1_blend: bool = False, conv=None,
         alg_flops_scale: float = 1.0, ln=None, colstats=None, probe_colstats: bool = False,
         out_blocked: bool = False, a_blocked: bool = False, operand_bf16: bool = False,
         workspace: Optional[torch.Tensor] = None, sched: int = 0):
    """out = epilogue(A @ W^T); see gcd_gemm_desc in include/gcd_amd.h.

    conv: dict(Cin, Hi, Wi, Ho, Wo, stride, upsample) for GEMM_CONV3X3 or dict(Cin, T, HW) for
    GEMM_TEMPORAL3.  a16 is the token-major fp16 activation [rows, Cin].
    colstats: optional fp32 [2 * M / 64, N] that receives the per-64-row column sums / sums of squares
    of the fp32 output (gcd_gemm_desc.colstats) for `groupnorm_stats_from_colsums`.
    probe_colstats: do not launch; return whether `colstats` would be honoured for this call.
    sched: gcd_gemm_desc.sched (bit 0: walk the tiles from the end of the output; results unchanged).
    """
    _need_gpu(a16, w16, out)
    want = torch.bfloat16 if operand_bf16 else torch.float16
    assert a16.dtype == want and w16.dtype == want, f"operands must be {want}"
    N, K = w16.shape
    d = GemmDesc()
    d.A, d.W, d.out = a16.data_ptr(), w16.data_ptr(), out.data_ptr()
    d.lda, d.ldo = _ld(a16), _ld(out)      # (ignored for a tile-blocked operand / output)
    d.M, d.N, d.K, d.mode = M, N, K, mode
    if mode != GEMM_PLAIN:
        d.Cin = conv["Cin"]
        d.Hi, d.Wi = conv.get("Hi", 0), conv.get("Wi", 0)
        d.Ho, d.Wo = conv.get("Ho", 0), conv.get("Wo", 0)
        d.stride, d.upsample = conv.get("stride", 1), int(conv.get("upsample", 0))
        d.asym_pad = int(conv.get("asym_pad", 0))
        d.T, d.HW = conv.get("T", 0), conv.get("HW", 0)
        d.zero_page = zero_page(a16.device).data_ptr()
    d.bias = _p(bias)
    if rowvec is not None:
        d.rowvec, d.ld_rowvec, d.rows_per_vec = rowvec.data_ptr(), _ld(rowvec), rows_per_vec
    if r1 is not None:
        d.R1, d.ldr1 = r1.data_ptr(), _ld(r1)
    if r2 is not None:
        d.R2, d.ldr2 = r2.data_ptr(), _ld(r2)
    d.s_acc, d.s_r1, d.s_r2 = s_acc, s_r1, s_r2
    if frame_alpha is not None:
        d.frame_alpha, d.rows_per_alpha, d.r1_blend = frame_alpha.data_ptr(), rows_per_alpha, int(r1_blend)
    d.out_kind = out_kind
    d.out_blocked, d.a_blocked, d.operand_bf16 = int(out_blocked), int(a_blocked), int(operand_bf16)
    d.sched = int(sched)
    if ln is not None:
        # fused LayerNorm of the output rows: dict(gamma, beta, out16[, eps, addvec, rows_per_vec, sum_out])
        _need_gpu(ln["gamma"], ln["beta"], ln["out16"], ln.get("addvec"), ln.get("sum_out"))
        d.ln_out16, d.ld_ln_out = ln["out16"].data_ptr(), _ld(ln["out16"])
        d.ln_gamma, d.ln_beta = ln["gamma"].data_ptr(), ln["beta"].data_ptr()
        d.ln_eps = ln.get("eps", 1e-5)
        if ln.get("addvec") is not None:
            d.ln_addvec, d.ld_ln_addvec = ln["addvec"].data_ptr(), _ld(ln["addvec"])
            d.ln_rows_per_vec = ln["rows_per_vec"]
            if ln.get("sum_out") is not None:
                d.ln_sum_out, d.ld_ln_sum = ln["sum_out"].data_ptr(), _ld(ln["sum_out"])
    if probe_colstats:
        return bool(_COLSTATS_ON and _lib.load().gcd_gemm_colstats_supported(C.byref(d)))
    if colstats is not None:
        _need_gpu(colstats)
        assert colstats.dtype == torch.float32 and colstats.is_contiguous() and \
            colstats.numel() >= 2 * (M // 64) * N
        d.colstats = colstats.data_ptr()
    if _SPLITK_ON:
        # split-K scratch: the caller's (the fine-tune step brings a larger one for its weight gradients) or the
        # persistent per-(device, stream) one of the inference engine
        sk = workspace if workspace is not None else _splitk_ws(a16.device)
        d.workspace, d.workspace_bytes = sk.data_ptr(), sk.numel() * sk.element_size()
    expect = torch.float32 if out_kind == OUT_F32 else torch.float16
    assert out.dtype == expect, f"out dtype {out.dtype} does not match out_kind {out_kind}"
    with _Timed("gemm", 2.0 * M * N * K * alg_flops_scale, M=M, N=N, K=K, mode=mode, out_kind=out_kind,
                nres=int(r1 is not None) + int(r2 is not None), cin=(conv or {}).get("Cin", K),
                stride=(conv or {}).get("stride", 1), up=int((conv or {}).get("upsample", 0))):
        check(_lib.load().gcd_gemm_f16(C.byref(d), _stream()), "gcd_gemm_f16")
    return out


# ---- FeedForward(GEGLU) as one kernel (gcd_ff_fused_f16; the C = 320 / hidden = 1280 level) ----------------------------
# GCD_FF_FUSED=0 keeps the two-GEMM path everywhere (A/B).  The one-kernel form processes 128-token tiles on a persistent
# grid of one workgroup per CU; below ~4 rounds of tiles the rounding-up of the last round costs more than the fusion
# saves (tools/ff_fused_probe: 71 vs 39 us at M = 4096), so it is used from FF_FUSED_MIN_TOKENS tokens.
_FF_FUSED_ON = os.environ.get("GCD_FF_FUSED", "1") != "0"
FF_FUSED_MIN_TOKENS = int(os.environ.get("GCD_FF_FUSED_MIN_TOKENS", str(4 * 256 * 128)))


def ff_fused_ok(M: int, C_: int, hidden: int, enabled: Optional[bool] = None) -> bool:
    on = _FF_FUSED_ON if enabled is None else enabled
    return bool(on and M >= FF_FUSED_MIN_TOKENS and _lib.load().gcd_ff_fused_supported(M, C_, hidden))


def ff_pack(w1_geglu16: torch.Tensor, w2_16: torch.Tensor, for_ln: bool = False) -> torch.Tensor:
    """The fragment-order weight stream of one FeedForward (gcd_ff_pack_f16): w1 = packing.pack_geglu's fp16
    [2560, 320], w2 = fp16 [320, 1280].  Once per parameter version.  for_ln: for `ff_fused(..., ln=...)`."""
    _need_gpu(w1_geglu16, w2_16)
    assert w1_geglu16.dtype == w2_16.dtype == torch.float16 and w1_geglu16.is_contiguous() and w2_16.is_contiguous()
    assert tuple(w1_geglu16.shape) == (2560, 320) and tuple(w2_16.shape) == (320, 1280)
    lib = _lib.load()
    wp = torch.empty(int(lib.gcd_ff_packed_bytes()) // 2, device=w1_geglu16.device, dtype=torch.float16)
    check(lib.gcd_ff_pack_f16(w1_geglu16.data_ptr(), w2_16.data_ptr(), wp.data_ptr(), int(for_ln), _stream()),
          "gcd_ff_pack_f16")
    return wp


def ff_fused(x: torch.Tensor, wp: torch.Tensor, b1: torch.Tensor, b2: torch.Tensor, out: torch.Tensor, *, M: int,
             r1: Optional[torch.Tensor] = None, r2: Optional[torch.Tensor] = None, out_kind: int = OUT_F32,
             s_acc: float = 1.0, s_r2: float = 1.0, frame_alpha=None, rows_per_alpha: int = 1, sched: int = 0, ln=None):
    """out = s_acc (FF(x16) + r1) + s_r2 r2 in one kernel (gcd_ff_desc); frame_alpha: s_acc = 1 - alpha, s_r2 = alpha.
    ln = dict(gamma, beta[, eps, addvec, rows_per_vec]): the LayerNorm form — x is the fp32 residual stream,
    z = x + addvec, out = s_acc (FF(LN(z)) + z) + s_r2 r2 (wp from ff_pack(..., for_ln=True); r1 must be None)."""
    _need_gpu(x, wp, b1, b2, out, r1, r2, frame_alpha)
    assert out.dtype == (torch.float32 if out_kind == OUT_F32 else torch.float16)
    d = FfDesc()
    d.wp, d.b1, d.b2 = wp.data_ptr(), b1.data_ptr(), b2.data_ptr()
    if ln is not None:
        assert x.dtype == torch.float32 and r1 is None
        _need_gpu(ln["gamma"], ln["beta"], ln.get("addvec"))
        d.x32, d.ldx32 = x.data_ptr(), _ld(x)
        d.ln_gamma, d.ln_beta, d.ln_eps = ln["gamma"].data_ptr(), ln["beta"].data_ptr(), ln.get("eps", 1e-5)
        if ln.get("addvec") is not None:
            d.addvec, d.ld_addvec, d.rows_per_vec = ln["addvec"].data_ptr(), _ld(ln["addvec"]), ln["rows_per_vec"]
    else:
        assert x.dtype == torch.float16 and r1 is not None
        d.X, d.ldx = x.data_ptr(), _ld(x)
        d.R1, d.ldr1 = r1.data_ptr(), _ld(r1)
    if r2 is not None:
        d.R2, d.ldr2 = r2.data_ptr(), _ld(r2)
    d.out, d.ldo, d.out_kind = out.data_ptr(), _ld(out), out_kind
    if frame_alpha is not None:
        d.frame_alpha, d.rows_per_alpha = frame_alpha.data_ptr(), rows_per_alpha
    d.s_acc, d.s_r2 = s_acc, s_r2
    d.M, d.C, d.hidden, d.sched = M, 320, 1280, int(sched)
    with _Timed("gemm", 2.0 * M * 320 * 3840, M=M, N=320, K=3840, mode=GEMM_PLAIN, out_kind=out_kind,
                nres=1 + int(r2 is not None), cin=3840, stride=1, up=0, fused_ff=1):
        check(_lib.load().gcd_ff_fused_f16(C.byref(d), _stream()), "gcd_ff_fused_f16")
    return out


# ---- LayerNorm + q | k | v as one kernel (gcd_lnqkv_f16; model width 320) ------------------------------------------------
# GCD_LNQKV=0 keeps LayerNorm + GEMM everywhere (A/B).  256-token tiles on a persistent grid: used from 2 rounds of tiles.
_LNQKV_ON = os.environ.get("GCD_LNQKV", "1") != "0"
LNQKV_MIN_TOKENS = int(os.environ.get("GCD_LNQKV_MIN_TOKENS", str(2 * 256 * 256)))


def lnqkv_ok(M: int, C_: int, N: int, enabled: Optional[bool] = None) -> bool:
    on = _LNQKV_ON if enabled is None else enabled
    return bool(on and M >= LNQKV_MIN_TOKENS and _lib.load().gcd_lnqkv_supported(C_, N))


def lnqkv_pack(w16: torch.Tensor) -> torch.Tensor:
    """The fragment-order image of a [N, 320] fp16 projection weight (gcd_lnqkv_pack_f16).  Once per parameter version."""
    _need_gpu(w16)
    assert w16.dtype == torch.float16 and w16.is_contiguous() and w16.shape[1] == 320 and w16.shape[0] % 64 == 0
    lib = _lib.load()
    wp = torch.empty(int(lib.gcd_lnqkv_packed_bytes(w16.shape[0])) // 2, device=w16.device, dtype=torch.float16)
    check(lib.gcd_lnqkv_pack_f16(w16.data_ptr(), w16.shape[0], wp.data_ptr(), _stream()), "gcd_lnqkv_pack_f16")
    return wp


def lnqkv(x32: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, wp: torch.Tensor, out16: torch.Tensor, *, M: int, N: int,
          eps: float = 1e-5, sched: int = 0):
    """out16[M, N] = LN(x32[M, 320]) @ W^T in one kernel (wp = lnqkv_pack(W))."""
    _need_gpu(x32, gamma, beta, wp, out16)
    assert x32.dtype == torch.float32 and out16.dtype == torch.float16
    with _Timed("gemm", 2.0 * M * 320 * N, M=M, N=N, K=320, mode=GEMM_PLAIN, out_kind=OUT_F16, nres=0, cin=320, stride=1,
                up=0, fused_ln=1):
        check(_lib.load().gcd_lnqkv_f16(x32.data_ptr(), _ld(x32), gamma.data_ptr(), beta.data_ptr(), eps, wp.data_ptr(),
                                        out16.data_ptr(), _ld(out16), M, 320, N, int(sched), _stream()), "gcd_lnqkv_f16")
    return out16


# Tile-blocked GEGLU hidden tensor: implemented and bit-identical, but measured neutral in the full step
# (112.4-112.6 ms off vs 112.5-112.8 ms on, interleaved runs; tools/store_probe.cpp's 25 % store penalty
# for 320-byte segments does not surface inside the GEGLU GEMM), so it is off unless GCD_HIDDEN_BLOCKED=1.
_HIDDEN_BLOCKED_ON = os.environ.get("GCD_HIDDEN_BLOCKED", "0") == "1"


def gemm_hidden_blocked_ok(M: int, n_geglu: int, n_out: int, enabled: Optional[bool] = None) -> bool:
    """True if a FeedForward of M tokens can (and, by the GCD_HIDDEN_BLOCKED switch, should) keep its GEGLU
    hidden tensor tile-blocked (`gemm(..., out_blocked=True)` then `gemm(..., a_blocked=True)`;
    gcd_gemm_hidden_blocked_supported)."""
    on = _HIDDEN_BLOCKED_ON if enabled is None else enabled
    return bool(on and _lib.load().gcd_gemm_hidden_blocked_supported(M, n_geglu, n_out))


def gemm_ln_fusable(M: int, N: int, K: int, mode: int = GEMM_PLAIN) -> bool:
    """True if `gemm(..., ln=...)` is accepted for this shape (N == 320 on the ping-pong kernel)."""
    return bool(_lib.load().gcd_gemm_ln_fusable(M, N, K, mode))


def linear_smallm(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor], y: torch.Tensor, *,
                  silu_in: bool = False, silu_out: bool = False, accumulate: bool = False):
    _need_gpu(x, w, y)
    assert x.dtype == w.dtype == y.dtype == torch.float32
    M, K = x.shape
    N = w.shape[0]
    assert w.shape[1] == K and w.is_contiguous() and y.shape == (M, N)
    flags = int(silu_in) | (int(silu_out) << 1) | (int(accumulate) << 2)
    check(_lib.load().gcd_linear_smallm_f32(x.data_ptr(), _ld(x), w.data_ptr(), _p(b), y.data_ptr(),
                                            _ld(y), M, N, K, flags, _stream()),
          "gcd_linear_smallm_f32")
    return y


def gn_nchunks(rows_per_inst: int, ninst: int = 1, cap: int = 256) -> int:
    """Row chunks per GroupNorm instance for the statistics pass: ~1500 workgroups over the whole
    launch (6 per CU), never less than 64 rows per chunk, at most `cap` chunks per instance."""
    want = max(1, -(-1536 // max(1, ninst)))
    return max(1, min(cap, want, (rows_per_inst + 63) // 64))


def groupnorm_stats(x1, x2, rows_per_inst: int, eps: float, partial: torch.Tensor,
                    stats: torch.Tensor, nchunks: int):
    _need_gpu(x1, x2, partial, stats)
    M, C1 = x1.shape
    C2 = 0 if x2 is None else x2.shape[1]
    assert partial.dtype == torch.float64 and stats.dtype == torch.float32
    ninst = M // rows_per_inst
    assert partial.numel() >= ninst * nchunks * 64 and stats.numel() >= ninst * 64
    check(_lib.load().gcd_groupnorm_stats(x1.data_ptr(), _ld(x1), C1, _p(x2),
                                          0 if x2 is None else _ld(x2), C2, M, rows_per_inst,
                                          eps, partial.data_ptr(), nchunks, stats.data_ptr(),
                                          _stream()), "gcd_groupnorm_stats")
    return stats


def groupnorm_stats_from_colsums(cs1, C1: int, cs2, C2: int, M: int, rows_per_inst: int, eps: float,
                                 stats: torch.Tensor):
    """GroupNorm statistics of x = [x1 | x2] from the column sums the producing GEMMs left behind
    (`gemm(..., colstats=...)`): no pass over x."""
    _need_gpu(cs1, cs2, stats)
    assert stats.dtype == torch.float32 and stats.numel() >= (M // rows_per_inst) * 64
    check(_lib.load().gcd_groupnorm_stats_from_colsums(cs1.data_ptr(), C1, _p(cs2), C2, M, rows_per_inst,
                                                       eps, stats.data_ptr(), _stream()),
          "gcd_groupnorm_stats_from_colsums")
    return stats


def groupnorm_apply(x1, x2, rows_per_inst: int, stats, gamma, beta, silu: bool, y16, raw16=None,
                    reverse: bool = False, order: Optional[int] = None):
    """order: walk order of the row blocks (0..3, see gcd_groupnorm_apply in gcd_amd.h); `reverse` = order 1.
    A bfloat16 y16 (and raw16) is written directly, rounded once from fp32 (ABI v7)."""
    _need_gpu(x1, x2, stats, gamma, beta, y16, raw16)
    if order is None:
        order = 1 if reverse else 0
    assert 0 <= order <= 3
    M, C1 = x1.shape
    C2 = 0 if x2 is None else x2.shape[1]
    check(_lib.load().gcd_groupnorm_apply(x1.data_ptr(), _ld(x1), C1, _p(x2),
                                          0 if x2 is None else _ld(x2), C2, M, rows_per_inst,
                                          stats.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                          int(silu) | (order << 1) | (8 if y16.dtype == torch.bfloat16 else 0),
                                          y16.data_ptr(), _ld(y16), _p(raw16),
                                          0 if raw16 is None else _ld(raw16), _stream()),
          "gcd_groupnorm_apply")
    return y16


def layernorm(x, gamma, beta, y16, *, eps: float = 1e-5, addvec=None, rows_per_vec: int = 1,
              sum_out=None, order: int = 0):
    _need_gpu(x, gamma, beta, y16, addvec, sum_out)
    M, Cc = x.shape
    check(_lib.load().gcd_layernorm_f16(x.data_ptr(), _ld(x), M, Cc, gamma.data_ptr(),
                                        beta.data_ptr(), eps, _p(addvec),
                                        0 if addvec is None else _ld(addvec), rows_per_vec,
                                        _p(sum_out), 0 if sum_out is None else _ld(sum_out),
                                        y16.data_ptr(), _ld(y16), order | (4 if y16.dtype == torch.bfloat16 else 0),
                                        _stream()), "gcd_layernorm_f16")
    return y16


def attn_transpose_v(qkv16, frames: int, S: int, heads: int, vt16, S_pad: int):
    _need_gpu(qkv16, vt16)
    check(_lib.load().gcd_attn_transpose_v(qkv16.data_ptr(), _ld(qkv16), frames, S, heads,
                                           vt16.data_ptr(), S_pad, _stream()),
          "gcd_attn_transpose_v")
    return vt16


# softmax scale of the 64-wide heads in exp2 units; packing.pack_qkv folds it into W_q
ATTN_Q_SCALE_LOG2 = 0.125 * 1.4426950408889634


def attn_spatial(qkv16, vt16, S_pad: int, out16, frames: int, S: int, heads: int,
                 q_prescaled: bool = False):
    _need_gpu(qkv16, vt16, out16)
    with _Timed("attn_spatial", 4.0 * frames * heads * S * S * 64, S=S, frames=frames, heads=heads):
        check(_lib.load().gcd_attn_spatial_f16(qkv16.data_ptr(), _ld(qkv16), vt16.data_ptr(), S_pad,
                                               out16.data_ptr(), _ld(out16), frames, S, heads,
                                               int(q_prescaled), _stream()), "gcd_attn_spatial_f16")
    return out16


def attn_spatial_bwd(qkv16, out16, dout16, dqkv32, frames: int, S: int, heads: int, ws: torch.Tensor,
                     scale: float = 0.125):
    """Backward of `attn_spatial` (gcd_attn_spatial_bwd): dqkv32 fp32 [frames*S, 3C]; ws: uint8 scratch of at
    least `attn_spatial_bwd_ws_bytes(frames, S, heads)` bytes."""
    _need_gpu(qkv16, out16, dout16, dqkv32, ws)
    assert qkv16.dtype == out16.dtype == dout16.dtype == torch.float16 and dqkv32.dtype == torch.float32
    with _Timed("attn_spatial_bwd", 14.0 * frames * heads * S * S * 64, S=S, frames=frames, heads=heads):
        check(_lib.load().gcd_attn_spatial_bwd(qkv16.data_ptr(), _ld(qkv16), out16.data_ptr(), _ld(out16),
                                               dout16.data_ptr(), _ld(dout16), dqkv32.data_ptr(), _ld(dqkv32),
                                               ws.data_ptr(), ws.numel() * ws.element_size(), frames, S, heads,
                                               scale, _stream()), "gcd_attn_spatial_bwd")
    return dqkv32


def attn_spatial_bwd_ws_bytes(frames: int, S: int, heads: int) -> int:
    return int(_lib.load().gcd_attn_spatial_bwd_ws_bytes(frames, S, heads))


def attn_temporal(qkv16, out16, clips: int, T: int, HW: int, heads: int):
    _need_gpu(qkv16, out16)
    check(_lib.load().gcd_attn_temporal_f16(qkv16.data_ptr(), _ld(qkv16), out16.data_ptr(),
                                            _ld(out16), clips, T, HW, heads, _stream()),
          "gcd_attn_temporal_f16")
    return out16


def softmax_rows(x32, y16):
    """y16[r] = softmax(x32[r]) (fp32 scores -> fp16 probabilities), rows of length C <= 16384."""
    _need_gpu(x32, y16)
    R, Cc = x32.shape
    check(_lib.load().gcd_softmax_rows_f16(x32.data_ptr(), _ld(x32), y16.data_ptr(), _ld(y16), R, Cc,
                                           _stream()), "gcd_softmax_rows_f16")
    return y16


def transpose_f16(x16, y16):
    """y16 [C, R] = x16 [R, C]^T (fp16)."""
    _need_gpu(x16, y16)
    R, Cc = x16.shape
    assert y16.shape == (Cc, R)
    check(_lib.load().gcd_transpose_f16(x16.data_ptr(), _ld(x16), y16.data_ptr(), _ld(y16), R, Cc,
                                        _stream()), "gcd_transpose_f16")
    return y16


def time_mix_unpack(tok32, w, b, out_nchw, C: int, N: int, T: int, HW: int):
    """AE3DConv.time_mix_conv + token-major -> NCHW (gcd_time_mix_unpack); w fp32 [C, C, 3]."""
    _need_gpu(tok32, w, b, out_nchw)
    assert out_nchw.is_contiguous() and out_nchw.dtype == torch.float32
    assert w.is_contiguous() and w.dtype == torch.float32 and w.numel() == C * C * 3
    check(_lib.load().gcd_time_mix_unpack(tok32.data_ptr(), _ld(tok32), w.data_ptr(), b.data_ptr(),
                                          out_nchw.data_ptr(), C, N, T, HW, _stream()),
          "gcd_time_mix_unpack")
    return out_nchw


def pack_input(x, concat, c_in, N: int, HW: int, out16, Cpad: int):
    """x: [nx, Cx, H, W] fp32 contiguous; concat: [N, Cc, H, W] fp32 or None; out16: [N*HW, Cpad]."""
    _need_gpu(x, concat, c_in, out16)
    assert x.is_contiguous() and x.dtype == torch.float32
    nx, Cx = x.shape[0], x.shape[1]
    Cc = 0
    if concat is not None:
        assert concat.is_contiguous() and concat.dtype == torch.float32 and concat.shape[0] == N
        Cc = concat.shape[1]
    check(_lib.load().gcd_pack_input(x.data_ptr(), nx, Cx, _p(concat), Cc, _p(c_in), N, HW,
                                     out16.data_ptr(), Cpad, _stream()), "gcd_pack_input")
    return out16


def unpack_output(tok32, out_nchw, Cout: int, N: int, HW: int):
    _need_gpu(tok32, out_nchw)
    assert out_nchw.is_contiguous() and out_nchw.dtype == torch.float32
    check(_lib.load().gcd_unpack_output(tok32.data_ptr(), _ld(tok32), out_nchw.data_ptr(), Cout, N,
                                        HW, _stream()), "gcd_unpack_output")
    return out_nchw


def cast_bf16(x, y16):
    """fp32 or fp16 [M, C] -> bfloat16 (round to nearest even), for `gemm(..., operand_bf16=True)`."""
    _need_gpu(x, y16)
    M, Cc = x.shape
    assert y16.dtype == torch.bfloat16
    fn = _lib.load().gcd_cast_f32_bf16 if x.dtype == torch.float32 else _lib.load().gcd_cast_f16_bf16
    assert x.dtype in (torch.float32, torch.float16)
    check(fn(x.data_ptr(), _ld(x), y16.data_ptr(), _ld(y16), M, Cc, _stream()), "gcd_cast_*_bf16")
    return y16


def cast_f16(x32, y16):
    _need_gpu(x32, y16)
    M, Cc = x32.shape
    check(_lib.load().gcd_cast_f32_f16(x32.data_ptr(), _ld(x32), y16.data_ptr(), _ld(y16), M, Cc,
                                       _stream()), "gcd_cast_f32_f16")
    return y16


def cfg_euler_step(x, net, scale, sig, x_out, T: int):
    """x, x_out: [nx, C, H, W] fp32; net: [2*nx, C, H, W] fp32; sig: device [2] = (sigma, sigma_next)."""
    _need_gpu(x, net, scale, sig, x_out)
    assert x.is_contiguous() and net.is_contiguous() and x_out.is_contiguous()
    nx = x.shape[0]
    chw = x[0].numel()
    assert net.shape[0] == 2 * nx and net[0].numel() == chw
    check(_lib.load().gcd_cfg_euler_step(x.data_ptr(), net.data_ptr(), scale.data_ptr(),
                                         sig.data_ptr(), x_out.data_ptr(), nx, T, chw, _stream()),
          "gcd_cfg_euler_step")
    return x_out


def edm_scalings(sig, c_in, c_noise):
    _need_gpu(sig, c_in, c_noise)
    check(_lib.load().gcd_edm_scalings(sig.data_ptr(), c_in.data_ptr(), c_noise.data_ptr(),
                                       c_in.numel(), _stream()), "gcd_edm_scalings")


def timestep_embedding(t, emb, max_period: float = 10000.0):
    _need_gpu(t, emb)
    N, dim = emb.shape
    assert emb.is_contiguous() and t.numel() == N and t.dtype == torch.float32
    check(_lib.load().gcd_timestep_embedding(t.data_ptr(), emb.data_ptr(), N, dim, max_period,
                                             _stream()), "gcd_timestep_embedding")
    return emb
