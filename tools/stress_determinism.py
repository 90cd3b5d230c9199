"""tools/stress_determinism.py — race hunt for the kernels whose synchronisation is hand-counted.

The persistent ping-pong GEMM admits its epilogue's stores through counted `s_waitcnt vmcnt(8 + n)` while the
next tile's first LDS-DMA pieces are in flight (cross-tile prefetch), and the software-pipelined attention kernel
reads K / V^T fragments from a 4-stage LDS-DMA ring with one barrier per tile.  Both are deterministic, so any
launch whose output differs from the first launch's by a single bit is a race.  Runs each case `--iters` times
under memory pressure from a concurrent copy stream and reports the number of differing launches.

    python tools/stress_determinism.py [--iters 300]
"""
from __future__ import annotations

import argparse
import math
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=300)
    a = ap.parse_args()
    from gcd_amd import ops, packing
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    bad_total = 0

    # a copy stream that keeps HBM busy with a different access pattern while the kernels run
    side = torch.cuda.Stream()
    junk_a = torch.empty(64 << 20, device=dev, dtype=torch.float32)
    junk_b = torch.empty_like(junk_a)

    def pressure():
        with torch.cuda.stream(side):
            junk_b.copy_(junk_a, non_blocking=True)

    def repeat(name, launch, out):
        nonlocal bad_total
        launch()
        torch.cuda.synchronize()
        first = out.clone()
        bad = 0
        for i in range(a.iters):
            out.zero_()
            if i % 3 == 0:
                pressure()
            launch()
            torch.cuda.synchronize()
            if not torch.equal(out, first):
                bad += 1
        print(f"{name:58s} {a.iters} launches, {bad} differ from the first")
        bad_total += bad

    ops.tune_set(ops.TUNE_GEMM_IMPL, 2)
    # (M, N, K): > 256 tiles so the persistent walk with the cross-tile prefetch runs; K = 320 / 640 = few
    # sub-tiles per tile (the prefetch and the counted waits dominate), one long-K shape
    for (M, N, K) in [(256 * 90, 2560, 320), (256 * 150, 960, 320), (256 * 300, 320, 320),
                      (256 * 100, 640, 640), (256 * 70, 1280, 2560)]:
        av = (torch.randn(M, K, generator=g)).half().to(dev)
        w = (torch.randn(N, K, generator=g) / math.sqrt(K)).half()
        bias = torch.randn(N, generator=g)
        r1 = torch.randn(M, N, generator=g).to(dev)
        wg = w.to(dev)
        # fp32 result with an fp32 residual (40 counted stores per wave)
        out = torch.empty(M, N, device=dev)
        repeat(f"gemm {M}x{N}x{K} f32 + residual", lambda: ops.gemm(av, wg, out, M=M, bias=bias.to(dev), r1=r1), out)
        # fp16 rows (20 stores)
        out16 = torch.empty(M, N, device=dev, dtype=torch.float16)
        repeat(f"gemm {M}x{N}x{K} f16 rows", lambda: ops.gemm(av, wg, out16, M=M, out_kind=ops.OUT_F16), out16)
        # GEGLU (10 stores)
        wp, bp = packing.pack_geglu(w.float(), bias)
        wpg, bpg = wp.half().to(dev), bp.to(dev)
        hid = torch.empty(M, N // 2, device=dev, dtype=torch.float16)
        repeat(f"gemm {M}x{N}x{K} GEGLU",
               lambda: ops.gemm(av, wpg, hid, M=M, bias=bpg, out_kind=ops.OUT_GEGLU), hid)
    ops.tune_set(ops.TUNE_GEMM_IMPL, 0)

    # spatial attention, pipelined kernel: many tiles, ragged and whole last tiles
    for (frames, S, heads) in [(4, 2304, 5), (2, 9216, 2), (3, 1300, 3)]:
        C = heads * 64
        qkv = (torch.randn(frames * S, 3 * C, generator=g) * 1.2).half().to(dev)
        S_pad = (S + 63) // 64 * 64
        vt = torch.empty(frames * heads * 64 * S_pad, dtype=torch.float16, device=dev)
        ops.attn_transpose_v(qkv, frames, S, heads, vt, S_pad)
        out = torch.empty(frames * S, C, dtype=torch.float16, device=dev)
        repeat(f"attention {frames} x {S} x {heads} heads (pipelined)",
               lambda: ops.attn_spatial(qkv, vt, S_pad, out, frames, S, heads, q_prescaled=True), out)
    print("TOTAL differing launches:", bad_total)
    return 1 if bad_total else 0


if __name__ == "__main__":
    sys.exit(main())
