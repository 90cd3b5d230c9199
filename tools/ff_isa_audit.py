#!/usr/bin/env python
"""Audit of the gfx950 code hipcc generates for the one-kernel FeedForward (gcd_amd/csrc/ff_fused.hip).

The kernel's MFMAs are inline asm (ff_fused_kernel.h explains why), so hipcc pads none of their hazards and counts none
of the asm loads.  The kernel keeps the distances itself (pins); this script checks the OUTPUT, after every rebuild, for
what a compiler scheduling or register-allocation change could silently break:

  1. no scratch (spill) access and no v_accvgpr_* shuttling inside the steady-state iteration of a kernel;
  2. no VALU / VMEM / LDS instruction writes a VGPR that an MFMA reads as its A, B or C operand within the 2
     instructions in front of that MFMA (VALU write -> MFMA source read needs wait states nothing inserts inside asm) —
     except LDS / VMEM loads, which the compiler's own s_waitcnt covers;
  3. no VALU instruction reads an MFMA's result within the 10 instructions behind it, unless the accumulate chain has
     taken the result over (a 4-pass MFMA's result needs 7 wait states before a VALU reads it; this is what a
     compiler-inserted copy of an accumulator — live-range splitting, a scalar "+v" pin — trips over);
  3b. no VALU instruction WRITES a register an MFMA reads as A, B (or as a C that is not its own result) within the 2 instructions behind that MFMA (the
     matrix pipe reads its sources over several cycles; hipcc believes an asm statement's inputs dead at the statement);
  4. no `s_waitcnt vmcnt(N)` with N < 40 inside the first iteration's code other than the kernel's own vmcnt(0)
     (a compiler-inserted counted wait there would drain the residual loads it cannot see: 9 instead of 60 memory
     instructions in flight per wave, measured 17 000 cycles per tile).

Usage: python tools/ff_isa_audit.py [file.s]     (without a file: compiles ff_fused.hip to assembly first)
Exit status 0 = clean.  tests/test_host.py runs it on the CPU (hipcc cross-compiles without a GPU).
"""
from __future__ import annotations

import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def compile_to_asm(out: Path) -> None:
    from gcd_amd.csrc import build as B
    cmd = [B._hipcc(), *B.FLAGS, *B.EXTRA_FLAGS.get("ff_fused.hip", []), "--cuda-device-only", "-S",
           str(B.CSRC / "ff_fused.hip"), "-o", str(out)]
    subprocess.check_call(cmd, stderr=subprocess.DEVNULL)


AUDIT_BACK = int(__import__("os").environ.get("FF_AUDIT_BACK", "2"))
AUDIT_FWD = int(__import__("os").environ.get("FF_AUDIT_FWD", "10"))


def regs(op: str):
    """Register set of one operand: v[10:13] -> {v10..v13}, a5 -> {a5}."""
    m = re.fullmatch(r"([va])\[(\d+):(\d+)\]", op)
    if m:
        return {f"{m.group(1)}{i}" for i in range(int(m.group(2)), int(m.group(3)) + 1)}
    m = re.fullmatch(r"([va])(\d+)", op)
    return {op} if m else set()


def parse(line: str):
    line = line.split(";")[0].strip()
    if not line or line.endswith(":") or line.startswith("."):
        return None
    parts = line.split(None, 1)
    ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
    return parts[0], ops


def audit_kernel(name: str, lines: list[str]) -> list[str]:
    errs = []
    ins = [(i, parse(l)) for i, l in enumerate(lines)]
    ins = [(i, p) for i, p in ins if p]
    # the steady-state iteration: the innermost loop (Depth=2) up to its back branch
    start = next((i for i, l in enumerate(lines) if "Depth=2" in l), None)
    end = None
    if start is not None:
        end = next((i for i in range(start, len(lines)) if re.search(r"s_cbranch_scc[01]\s", lines[i])), None)
    if start is None or end is None:
        return [f"{name}: steady-state loop not found"]
    for i in range(start, end):
        if "scratch_" in lines[i]:
            errs.append(f"{name}: scratch access inside the steady-state iteration: {lines[i].strip()}")
        if "v_accvgpr" in lines[i]:
            errs.append(f"{name}: v_accvgpr_* inside the steady-state iteration: {lines[i].strip()}")
    if any("scratch_" in l for l in lines):
        errs.append(f"{name}: the kernel spills ({sum('scratch_' in l for l in lines)} scratch accesses)")
    for k, (i, (op, ops)) in enumerate(ins):
        if not op.startswith("v_mfma"):
            continue
        src = set().union(*(regs(o) for o in ops[1:]))
        dst = regs(ops[0])
        for back in range(1, AUDIT_BACK + 1):
            if k - back < 0:
                break
            pop, pops = ins[k - back][1]
            if pop.startswith(("v_mfma", "s_", "ds_read", "global_load", "buffer_load", "ds_write", "global_store", ";;")):
                continue
            if pops and regs(pops[0]) & src and pop.startswith("v_"):
                errs.append(f"{name}: {pop} writes {pops[0]} {back} instruction(s) in front of the MFMA that reads it "
                            f"(line {i + 1})")
        ab = set().union(*(regs(o) for o in ops[1:4] if regs(o) != dst))      # A, B and a C that is not the accumulate chain
        for fwd in (1, 2):
            if k + fwd >= len(ins):
                break
            nop, nops = ins[k + fwd][1]
            if nop == "s_nop" and nops and int(nops[0]) >= 3:
                break      # wait states inside the asm string: whatever follows is far enough
            if nop.startswith("v_") and not nop.startswith("v_mfma") and nops and regs(nops[0]) & ab:
                errs.append(f"{name}: {nop} writes {nops[0]}, a source of the MFMA {fwd} instruction(s) in front of it "
                            f"(line {i + 1})")
        for fwd in range(1, AUDIT_FWD + 1):
            if k + fwd >= len(ins):
                break
            nop, nops = ins[k + fwd][1]
            if fwd == 1 and nop == "s_nop" and nops and int(nops[0]) >= 6:
                break      # the result's wait states are inside the asm string
            if nop.startswith("v_mfma") and nops and regs(nops[0]) == dst:
                break      # the accumulate chain took the result over: a later reader reads THAT MFMA's result
            if not nop.startswith("v_") or nop.startswith("v_mfma"):
                continue
            rd = set().union(*(regs(o) for o in nops[1:])) if len(nops) > 1 else set()
            if rd & dst:
                errs.append(f"{name}: {nop} reads the result of the MFMA {fwd} instruction(s) in front of it (line {i + 1})")
    # first iteration = from the first residual load into the accumulator file to the steady-state loop
    first = next((i for i, l in enumerate(lines) if re.search(r"global_load_dwordx4 a\[", l)), None)
    if first is not None and first < start:
        for i in range(first, start):
            m = re.search(r"s_waitcnt.*vmcnt\((\d+)\)", lines[i])
            if m and 0 < int(m.group(1)) < 40:
                errs.append(f"{name}: compiler-inserted vmcnt({m.group(1)}) inside the first iteration (line {i + 1})")
    return errs


def main() -> int:
    if len(sys.argv) > 1:
        text = Path(sys.argv[1]).read_text()
    else:
        with tempfile.TemporaryDirectory() as td:
            out = Path(td) / "ff_fused.s"
            compile_to_asm(out)
            text = out.read_text()
    lines = text.splitlines()
    kernels = {}
    cur = None
    for l in lines:
        m = re.match(r"^(_Z\w*ff_fused_kernel\w*):", l)
        if m:
            cur = m.group(1)
            kernels[cur] = []
        if cur:
            kernels[cur].append(l)
            if "s_endpgm" in l:
                cur = None
    if not kernels:
        print("no ff_fused_kernel in the assembly")
        return 2
    bad = 0
    for name, kl in kernels.items():
        errs = audit_kernel(name, kl)
        n_mfma = sum("v_mfma" in l for l in kl)
        print(f"{name}: {len(kl)} lines, {n_mfma} MFMAs, {len(errs)} finding(s)")
        for e in errs[:20]:
            print("   ", e)
        bad += len(errs)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
