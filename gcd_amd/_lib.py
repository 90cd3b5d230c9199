"""ctypes binding of libgcd_amd.so (the C ABI declared in include/gcd_amd.h).

The library is the product: there is no Python/CPU fallback.  If the shared object is missing or a
call fails, this module raises — loudly — instead of computing the result some other way.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import os

_PKG = Path(__file__).resolve().parent
# GCD_AMD_LIB: another build of the SAME sources (tools/libgcd_amd_*.so: ablation / A-B variants) for the benchmark
# tools; the product is the in-tree library next to this file.
LIB_PATH = Path(os.environ["GCD_AMD_LIB"]).resolve() if os.environ.get("GCD_AMD_LIB") else _PKG / "libgcd_amd.so"

ABI_VERSION = 9

# GEMM modes / output kinds (mirror include/gcd_amd.h)
GEMM_PLAIN, GEMM_CONV3X3, GEMM_TEMPORAL3 = 0, 1, 2
OUT_F32, OUT_F16, OUT_GEGLU = 0, 1, 2


class GcdError(RuntimeError):
    """A libgcd_amd entry point returned a non-zero status."""


class GemmDesc(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("W", C.c_void_p), ("out", C.c_void_p),
        ("lda", C.c_int64), ("ldo", C.c_int64),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("mode", C.c_int32),
        ("Cin", C.c_int32), ("Hi", C.c_int32), ("Wi", C.c_int32), ("Ho", C.c_int32),
        ("Wo", C.c_int32), ("stride", C.c_int32), ("upsample", C.c_int32), ("T", C.c_int32),
        ("HW", C.c_int32),
        ("bias", C.c_void_p), ("rowvec", C.c_void_p), ("ld_rowvec", C.c_int64),
        ("rows_per_vec", C.c_int32),
        ("R1", C.c_void_p), ("ldr1", C.c_int64), ("R2", C.c_void_p), ("ldr2", C.c_int64),
        ("s_acc", C.c_float), ("s_r1", C.c_float), ("s_r2", C.c_float),
        ("frame_alpha", C.c_void_p), ("rows_per_alpha", C.c_int32), ("r1_blend", C.c_int32),
        ("out_kind", C.c_int32), ("zero_page", C.c_void_p),
        ("ln_out16", C.c_void_p), ("ld_ln_out", C.c_int64), ("ln_gamma", C.c_void_p),
        ("ln_beta", C.c_void_p), ("ln_eps", C.c_float), ("ln_rows_per_vec", C.c_int32),
        ("ln_addvec", C.c_void_p), ("ld_ln_addvec", C.c_int64), ("ln_sum_out", C.c_void_p),
        ("ld_ln_sum", C.c_int64),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64), ("asym_pad", C.c_int32),
        ("colstats", C.c_void_p), ("out_blocked", C.c_int32), ("a_blocked", C.c_int32),
        ("operand_bf16", C.c_int32), ("sched", C.c_int32),
    ]


class FfDesc(C.Structure):
    """gcd_ff_desc (include/gcd_amd.h): the one-kernel FeedForward of the C = 320 level."""
    _fields_ = [
        ("X", C.c_void_p), ("ldx", C.c_int64), ("wp", C.c_void_p), ("b1", C.c_void_p), ("b2", C.c_void_p),
        ("R1", C.c_void_p), ("ldr1", C.c_int64), ("R2", C.c_void_p), ("ldr2", C.c_int64),
        ("out", C.c_void_p), ("ldo", C.c_int64), ("out_kind", C.c_int32),
        ("frame_alpha", C.c_void_p), ("rows_per_alpha", C.c_int32), ("s_acc", C.c_float), ("s_r2", C.c_float),
        ("M", C.c_int32), ("C", C.c_int32), ("hidden", C.c_int32), ("sched", C.c_int32),
        ("x32", C.c_void_p), ("ldx32", C.c_int64), ("ln_gamma", C.c_void_p), ("ln_beta", C.c_void_p),
        ("ln_eps", C.c_float), ("addvec", C.c_void_p), ("ld_addvec", C.c_int64), ("rows_per_vec", C.c_int32),
    ]


_vp, _i, _i64, _f = C.c_void_p, C.c_int, C.c_int64, C.c_float

# name -> (restype, argtypes); every symbol include/gcd_amd.h declares
SIGNATURES = {
    "gcd_abi_version": (_i, []),
    "gcd_last_error": (C.c_char_p, []),
    "gcd_device_info": (_i, [_i, C.c_char_p, _i, C.POINTER(_i), C.POINTER(C.c_size_t)]),
    "gcd_tune_set": (_i, [_i, _i]),
    "gcd_gemm_f16": (_i, [C.POINTER(GemmDesc), _vp]),
    "gcd_gemm_ln_fusable": (_i, [_i, _i, _i, _i]),
    "gcd_gemm_colstats_supported": (_i, [C.POINTER(GemmDesc)]),
    "gcd_gemm_hidden_blocked_supported": (_i, [_i, _i, _i]),
    "gcd_ff_packed_bytes": (_i64, []),
    "gcd_ff_pack_f16": (_i, [_vp, _vp, _vp, _i, _vp]),
    "gcd_ff_fused_supported": (_i, [_i, _i, _i]),
    "gcd_ff_fused_f16": (_i, [C.POINTER(FfDesc), _vp]),
    "gcd_lnqkv_packed_bytes": (_i64, [_i]),
    "gcd_lnqkv_supported": (_i, [_i, _i]),
    "gcd_lnqkv_pack_f16": (_i, [_vp, _i, _vp, _vp]),
    "gcd_lnqkv_f16": (_i, [_vp, _i64, _vp, _vp, _f, _vp, _vp, _i64, _i, _i, _i, _i, _vp]),
    "gcd_groupnorm_stats_from_colsums": (_i, [_vp, _i, _vp, _i, _i64, _i64, _f, _vp, _vp]),
    "gcd_linear_smallm_f32": (_i, [_vp, _i64, _vp, _vp, _vp, _i64, _i, _i, _i, _i, _vp]),
    "gcd_groupnorm_stats": (_i, [_vp, _i64, _i, _vp, _i64, _i, _i64, _i64, _f, _vp, _i, _vp, _vp]),
    "gcd_groupnorm_apply": (_i, [_vp, _i64, _i, _vp, _i64, _i, _i64, _i64, _vp, _vp, _vp, _i, _vp,
                                 _i64, _vp, _i64, _vp]),
    "gcd_layernorm_f16": (_i, [_vp, _i64, _i64, _i, _vp, _vp, _f, _vp, _i64, _i, _vp, _i64, _vp,
                               _i64, _i, _vp]),
    "gcd_attn_transpose_v": (_i, [_vp, _i64, _i, _i, _i, _vp, _i, _vp]),
    "gcd_attn_spatial_f16": (_i, [_vp, _i64, _vp, _i, _vp, _i64, _i, _i, _i, _i, _vp]),
    "gcd_attn_temporal_f16": (_i, [_vp, _i64, _vp, _i64, _i, _i, _i, _i, _vp]),
    "gcd_softmax_rows_f16": (_i, [_vp, _i64, _vp, _i64, _i64, _i, _vp]),
    "gcd_transpose_f16": (_i, [_vp, _i64, _vp, _i64, _i, _i, _vp]),
    "gcd_time_mix_unpack": (_i, [_vp, _i64, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "gcd_pack_input": (_i, [_vp, _i, _i, _vp, _i, _vp, _i, _i, _vp, _i, _vp]),
    "gcd_unpack_output": (_i, [_vp, _i64, _vp, _i, _i, _i, _vp]),
    "gcd_cast_f32_f16": (_i, [_vp, _i64, _vp, _i64, _i64, _i, _vp]),
    "gcd_cfg_euler_step": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i64, _vp]),
    "gcd_edm_scalings": (_i, [_vp, _vp, _vp, _i, _vp]),
    "gcd_timestep_embedding": (_i, [_vp, _vp, _i, _i, _f, _vp]),
    "gcd_im2col3x3_f16": (_i, [_vp, _i64, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "gcd_col2im3x3_f32": (_i, [_vp, _vp, _i64, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "gcd_im2col_t3_f16": (_i, [_vp, _i64, _vp, _i64, _i, _i, _i, _vp]),
    "gcd_col2im_t3_f32": (_i, [_vp, _vp, _i64, _i64, _i, _i, _i, _vp]),
    "gcd_rowblock_sum_f32": (_i, [_vp, _i64, _i64, _i, _i64, _vp, _vp]),
    "gcd_groupnorm_bwd_scratch_floats": (_i64, [_i, _i64, _i64]),
    "gcd_groupnorm_bwd": (_i, [_vp, _i64, _vp, _i64, _i, _i64, _i64, _vp, _vp, _vp, _i, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _vp]),
    "gcd_layernorm_bwd": (_i, [_vp, _i64, _vp, _i64, _i64, _i, _vp, _f, _vp, _i64, _vp, _vp, _vp, _i64, _vp]),
    "gcd_geglu_fwd_f32": (_i, [_vp, _i64, _vp, _i64, _i64, _i, _vp]),
    "gcd_geglu_fwd_f16": (_i, [_vp, _i64, _vp, _i64, _i64, _i, _vp]),
    "gcd_geglu_fwd_bf16": (_i, [_vp, _i64, _vp, _i64, _i64, _i, _vp]),
    "gcd_geglu_bwd_f32": (_i, [_vp, _i64, _vp, _i64, _vp, _i64, _i64, _i, _vp]),
    "gcd_softmax_bwd_rows": (_i, [_vp, _i64, _vp, _i64, _vp, _i64, _i64, _i, _f, _vp]),
    "gcd_attn_spatial_bwd_ws_bytes": (_i64, [_i, _i, _i]),
    "gcd_attn_spatial_bwd": (_i, [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _i, _i, _i, _f, _vp]),
    "gcd_attn_temporal_bwd": (_i, [_vp, _i64, _vp, _i64, _vp, _i64, _i, _i, _i, _i, _vp]),
    "gcd_cast_scale_f32_f16": (_i, [_vp, _i64, _vp, _i64, _i64, _i, _f, _vp]),
    "gcd_cast_f32_bf16": (_i, [_vp, _i64, _vp, _i64, _i64, _i, _vp]),
    "gcd_cast_colsum_f32": (_i, [_vp, _i64, _vp, _i64, _i64, _i, _i64, _vp, _i, _vp, _vp]),
    "gcd_cast_f16_bf16": (_i, [_vp, _i64, _vp, _i64, _i64, _i, _vp]),
    "gcd_adam_step": (_i, [_vp, _vp, _vp, _vp, _i64, _f, _f, _f, _f, _f, _i, _f, _vp]),
    "gcd_adam_step_multi": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _f, _f, _f, _f, _f, _i, _f, _vp]),
    "gcd_graph_begin_capture": (_i, [_vp]),
    "gcd_graph_end_capture": (_i, [_vp, C.POINTER(_vp)]),
    "gcd_graph_launch": (_i, [_vp, _vp]),
    "gcd_graph_destroy": (_i, [_vp]),
    "gcd_event_create": (_i, [C.POINTER(_vp)]),
    "gcd_event_record": (_i, [_vp, _vp]),
    "gcd_event_sync": (_i, [_vp]),
    "gcd_event_elapsed_ms": (_i, [_vp, _vp, C.POINTER(_f)]),
    "gcd_event_destroy": (_i, [_vp]),
    "gcd_stream_sync": (_i, [_vp]),
}

# libgcd_amd_train.so (include/gcd_amd_train.h): kernels of the fine-tune step only
TRAIN_LIB_PATH = _PKG / "libgcd_amd_train.so"
TRAIN_ABI_VERSION = 2


class PackEntry(C.Structure):
    """gcd_pack_entry (include/gcd_amd_train.h)."""
    _fields_ = [("src", _vp), ("dst_f", _vp), ("dst_t", _vp),
                ("f_ns", _i64), ("f_ts", _i64), ("t_cs", _i64), ("t_ts", _i64),
                ("N", C.c_int32), ("C", C.c_int32), ("taps", C.c_int32), ("mirror", C.c_int32),
                ("tile0", C.c_int32), ("tiles_c", C.c_int32)]


class SmallmProblem(C.Structure):
    """gcd_smallm_problem (include/gcd_amd_train.h)."""
    _fields_ = [("x", _vp), ("W", _vp), ("b", _vp), ("y", _vp), ("dx", _vp), ("dW", _vp), ("db", _vp),
                ("ldx", _i64), ("ldy", _i64), ("lddx", _i64),
                ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("flags", C.c_int32),
                ("block0", C.c_int32), ("reserved", C.c_int32)]


TRAIN_SIGNATURES = {
    "gcd_train_abi_version": (_i, []),
    "gcd_train_last_error": (C.c_char_p, []),
    "gcd_wgrad_tr_scratch_floats": (_i64, [_i64, _i, _i]),
    "gcd_wgrad_tr_f16": (_i, [_vp, _i64, _vp, _i64, _i64, _i, _i, _i, _vp, _i64, _vp, _i64, _vp]),
    "gcd_wgrad_tr_f16_ex": (_i, [_vp, _i64, _vp, _i64, _i64, _i, _i, _i, _vp, _i64, _i, _i, _i, _i, _vp, _i64, _vp]),
    "gcd_wgrad_conv_tr_f16": (_i, [_vp, _i64, _vp, _i64, _i64, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _i, _i, _i, _vp, _i64, _vp]),
    "gcd_train_pack_weights": (_i, [_vp, _i, _i, _i, _vp]),
    "gcd_blend_fwd_f32": (_i, [_vp, _i64, _vp, _i64, _vp, _i64, _i, _i64, _vp, _i64, _vp]),
    "gcd_blend_bwd_f32": (_i, [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _i, _i64, _vp, _i64, _i, _vp, _i64, _vp, _vp]),
    "gcd_gn_affine_grads": (_i, [_vp, _i, _i, _vp, _vp, _i, _vp]),
    "gcd_smallm_fwd": (_i, [_vp, _i, _i, _vp]),
    "gcd_smallm_dgrad": (_i, [_vp, _i, _i, _vp]),
    "gcd_smallm_wgrad": (_i, [_vp, _i, _i, _vp]),
}

_lib = None
_train = None


def load_train() -> C.CDLL:
    """Load libgcd_amd_train.so (once).  Raises if it has not been built — never falls back."""
    global _train
    if _train is not None:
        return _train
    if not TRAIN_LIB_PATH.exists():
        raise GcdError(f"{TRAIN_LIB_PATH} is missing: run `python -m gcd_amd.csrc.build` (needs hipcc)")
    lib = C.CDLL(str(TRAIN_LIB_PATH))
    for name, (res, args) in TRAIN_SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    v = lib.gcd_train_abi_version()
    if v != TRAIN_ABI_VERSION:
        raise GcdError(f"libgcd_amd_train ABI version {v} != expected {TRAIN_ABI_VERSION}; rebuild the library")
    _train = lib
    return lib


def check_train(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load_train().gcd_train_last_error().decode(errors="replace")
        raise GcdError(f"{what or 'libgcd_amd_train call'} failed (status {rc}): {msg}")


def load() -> C.CDLL:
    """Load libgcd_amd.so (once).  Raises if it has not been built — never falls back."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise GcdError(
            f"{LIB_PATH} is missing: the HIP extension has not been built. "
            "Run `python -m gcd_amd.csrc.build` (needs hipcc); gcd_amd has no CPU fallback."
        )
    lib = C.CDLL(str(LIB_PATH))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    v = lib.gcd_abi_version()
    if v != ABI_VERSION:
        raise GcdError(f"libgcd_amd ABI version {v} != expected {ABI_VERSION}; rebuild the library")
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().gcd_last_error().decode(errors="replace")
        raise GcdError(f"{what or 'libgcd_amd call'} failed (status {rc}): {msg}")
