"""Build libgcd_amd.so (the C-ABI HIP library) in-tree with hipcc for gfx950.

Usage: python -m gcd_amd.csrc.build [--force] [--save-temps]
The shared object lands next to the package (gcd_amd/libgcd_amd.so) so that it travels with the
source tree to the GPU box (it is git-ignored, not gpurun-ignored).
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys
from pathlib import Path

CSRC = Path(__file__).resolve().parent
PKG = CSRC.parent
ROOT = PKG.parent
LIB = PKG / "libgcd_amd.so"
STAMP = PKG / ".libgcd_amd.stamp"
SOURCES = ["runtime.hip", "gemm.hip", "gemm_pp.hip", "gemm_p8.hip", "conv_narrow.hip", "ff_fused.hip", "lnqkv.hip", "norm.hip", "attention.hip",
           "attn_bwd.hip", "elementwise.hip", "backward.hip"]
# per-source extra flags.  ff_fused.hip: hipcc's SLP pass pairs the GELU polynomial's scalar FMAs into v_pk_fma_f32, which
# does not issue in the shadow of an MFMA (tools/issue_probe; the kernel places every one of them behind an MFMA by hand)
EXTRA_FLAGS = {"ff_fused.hip": ["-fno-slp-vectorize"]}
# records of experiments that lost their A/B: compiled into the ablation / A-B libraries only (tools/libgcd_amd_*.so)
ABLATION_ONLY_SOURCES = ["gemm_p8x.hip"]
HEADERS = [CSRC / "common.h", CSRC / "gemm_common.h", CSRC / "ff_fused_kernel.h", ROOT / "include" / "gcd_amd.h"]
# libgcd_amd_train.so: kernels of the fine-tune step only (include/gcd_amd_train.h).  Its sources are NOT part of
# `sources_digest()`: they cannot change a kernel the sampler step launches.
LIB_TRAIN = PKG / "libgcd_amd_train.so"
STAMP_TRAIN = PKG / ".libgcd_amd_train.stamp"
TRAIN_SOURCES = ["train_wgrad.hip", "train_ops.hip"]
TRAIN_HEADERS = [ROOT / "include" / "gcd_amd_train.h", CSRC / "train_wgrad_kernel.h"]
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function",
         "-ffp-contract=fast"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (looked at $HIPCC, PATH and /opt/rocm/bin/hipcc)")


def _digest() -> str:
    h = hashlib.sha256()
    for p in [CSRC / s for s in SOURCES] + HEADERS:
        h.update(p.read_bytes())
    h.update(" ".join(FLAGS).encode())
    h.update(repr(sorted(EXTRA_FLAGS.items())).encode())
    return h.hexdigest()      # (ABLATION_ONLY_SOURCES are not in the product library)


def sources_digest() -> str:
    """Content hash of everything that decides which kernels a sampler step launches and how: the
    HIP sources + headers + flags and the host-side engine.  tools/pmc_traffic.py stamps a PMC
    profile with it and bench.py quotes `roofline.traffic` only while it matches."""
    h = hashlib.sha256(_digest().encode())
    for f in ("engine.py", "ops.py", "sampling.py"):
        h.update((PKG / f).read_bytes())
    return h.hexdigest()[:16]


def build_train(force: bool = False, verbose: bool = True) -> Path:
    """gcd_amd/libgcd_amd_train.so (the fine-tune step's own kernels), in-tree like the main library."""
    h = hashlib.sha256()
    for p in [CSRC / s for s in TRAIN_SOURCES] + TRAIN_HEADERS:
        h.update(p.read_bytes())
    h.update(" ".join(FLAGS).encode())
    digest = h.hexdigest()
    if not force and LIB_TRAIN.exists() and STAMP_TRAIN.exists() and STAMP_TRAIN.read_text().strip() == digest:
        return LIB_TRAIN
    hipcc = _hipcc()
    objdir = CSRC / "build"
    objdir.mkdir(exist_ok=True)
    objs = []
    for src in TRAIN_SOURCES:
        obj = objdir / (src + ".o")
        cmd = [hipcc, *FLAGS, "-c", str(CSRC / src), "-o", str(obj)]
        if verbose:
            print("[gcd_amd.build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd, cwd=str(objdir))
        objs.append(str(obj))
    cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", *objs, "-o", str(LIB_TRAIN)]
    if verbose:
        print("[gcd_amd.build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    STAMP_TRAIN.write_text(digest)
    return LIB_TRAIN


def build(force: bool = False, save_temps: bool = False, verbose: bool = True,
          ablation: bool = False) -> Path:
    """ablation=True builds tools/libgcd_amd_ablate.so instead: the same sources with
    -DGCD_ABLATION_BUILD, which adds the wrong-by-design kernel variants tools/gemm_bench times
    (GCD_TUNE_GEMM_IMPL >= 32).  The product library never contains them."""
    if ablation:
        return _build_ablation(verbose)
    build_train(force=force, verbose=verbose)
    digest = _digest()
    if not force and LIB.exists() and STAMP.exists() and STAMP.read_text().strip() == digest:
        return LIB
    hipcc = _hipcc()
    objdir = CSRC / "build"
    objdir.mkdir(exist_ok=True)
    procs = []
    for src in SOURCES:
        obj = objdir / (src + ".o")
        cmd = [hipcc, *FLAGS, *EXTRA_FLAGS.get(src, []), "-c", str(CSRC / src), "-o", str(obj)]
        if save_temps:
            cmd.insert(1, "-save-temps=obj")
        if verbose:
            print("[gcd_amd.build]", " ".join(cmd), flush=True)
        procs.append((src, obj, subprocess.Popen(cmd, cwd=str(objdir))))
    objs = []
    for src, obj, pr in procs:
        if pr.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
        objs.append(str(obj))
    cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", *objs, "-o", str(LIB)]
    if verbose:
        print("[gcd_amd.build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    STAMP.write_text(digest)
    return LIB


def _build_ablation(verbose: bool, name: str = "ablate", defines=("-DGCD_ABLATION_BUILD",)) -> Path:
    """A second library beside the product one (tools/libgcd_amd_<name>.so) from the same sources with extra defines:
    the ablation build, and the A/B variants of tools/ab_sweep.sh (loaded through GCD_AMD_LIB, see _lib.py)."""
    hipcc = _hipcc()
    objdir = CSRC / "build" / name
    objdir.mkdir(parents=True, exist_ok=True)
    out = ROOT / "tools" / f"libgcd_amd_{name}.so"
    procs = []
    if "-DGCD_ABLATION_BUILD" not in defines:
        defines = tuple(defines) + ("-DGCD_ABLATION_BUILD",)      # every A/B library carries the ablation-only variants
    for src in SOURCES + ABLATION_ONLY_SOURCES:
        obj = objdir / (src + ".o")
        cmd = [hipcc, *FLAGS, *EXTRA_FLAGS.get(src, []), *defines, "-c", str(CSRC / src), "-o", str(obj)]
        if verbose:
            print("[gcd_amd.build]", " ".join(cmd), flush=True)
        procs.append((src, obj, subprocess.Popen(cmd, cwd=str(objdir))))
    objs = []
    for src, obj, pr in procs:
        if pr.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
        objs.append(str(obj))
    subprocess.check_call([hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", *objs, "-o", str(out)])
    return out


if __name__ == "__main__":
    if "--ablation" in sys.argv:
        print(build(ablation=True))
    elif "--epi-wt" in sys.argv:      # write-through epilogue stores (gemm_common.h, GCD_EPI_WT): A/B library
        mask = int(sys.argv[sys.argv.index("--epi-wt") + 1])
        print(_build_ablation(True, f"wt{mask}", (f"-DGCD_EPI_WT={mask}",)))
    elif "--epi-nt" in sys.argv:      # non-temporal epilogue stores / residual loads (GCD_EPI_NT): A/B library
        mask = int(sys.argv[sys.argv.index("--epi-nt") + 1])
        print(_build_ablation(True, f"nt{mask}", (f"-DGCD_EPI_NT={mask}",)))
    else:
        build(force="--force" in sys.argv, save_temps="--save-temps" in sys.argv)
        print(LIB)
