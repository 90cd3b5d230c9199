"""tools/chunk_probe.py — does frame-chunked scheduling pay?  (round 2 experiment)

At the 72x128 level one fp32 activation [258048, 320] is 330 MB, more than the 256 MB Infinity Cache:
every kernel of a transformer block streams its operands from HBM although the previous kernel has
just written them.  This probe runs a representative per-frame-independent chain of the spatial
transformer block (LayerNorm -> q|k|v GEMM -> out-projection GEMM with fp32 residual -> LayerNorm ->
GEGLU GEMM -> FF-out GEMM with residual -> GroupNorm stats + apply) through the product's own ops,
once over all 28 frames and once per chunk of 28 / n frames with chunk-sized (reused) intermediates,
and prints the wall time of each.  Run on the GPU box:  python tools/chunk_probe.py
"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from gcd_amd import ops, packing                      # noqa: E402
from gcd_amd._lib import OUT_F16, OUT_GEGLU           # noqa: E402


def main():
    dev = torch.device("cuda:0")
    frames, HW, C = 28, 72 * 128, 320
    M = frames * HW
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn(M, C, device=dev, generator=g)
    rnd = lambda *s: (torch.randn(*s, device=dev, generator=g) / s[-1] ** 0.5)   # noqa: E731
    wqkv = rnd(3 * C, C).half()
    wo = rnd(C, C).half()
    w1, b1 = packing.pack_geglu(rnd(8 * C, C), torch.zeros(8 * C, device=dev))
    w2 = rnd(C, 4 * C).half()
    bo = torch.zeros(C, device=dev)
    gam, bet = torch.ones(C, device=dev), torch.zeros(C, device=dev)

    def chain(nchunks):
        fc = frames // nchunks
        Mc = fc * HW
        a16 = torch.empty(Mc, C, device=dev, dtype=torch.float16)
        qkv = torch.empty(Mc, 3 * C, device=dev, dtype=torch.float16)
        hid = torch.empty(Mc, 4 * C, device=dev, dtype=torch.float16)
        nch = ops.gn_nchunks(HW, fc)
        partial = torch.empty(fc * nch * 64, device=dev, dtype=torch.float64)
        stats = torch.empty(fc * 64, device=dev)

        def run():
            for ci in range(nchunks):
                xs = x[ci * Mc:(ci + 1) * Mc]
                ops.layernorm(xs, gam, bet, a16)
                ops.gemm(a16, wqkv, qkv, M=Mc, out_kind=OUT_F16)
                ops.gemm(qkv[:, :C], wo, xs, M=Mc, bias=bo, r1=xs, s_acc=0.05)
                ops.layernorm(xs, gam, bet, a16)
                ops.gemm(a16, w1, hid, M=Mc, bias=b1, out_kind=OUT_GEGLU)
                ops.gemm(hid, w2, xs, M=Mc, bias=bo, r1=xs, s_acc=0.05)
                ops.groupnorm_stats(xs, None, HW, 1e-5, partial, stats, nch)
                ops.groupnorm_apply(xs, None, HW, stats, gam, bet, True, a16)
        return run

    for nchunks in (1, 2, 4, 7, 14, 28, 1, 4):
        run = chain(nchunks)
        run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for _ in range(5):
            e0.record()
            run()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ts.sort()
        print(f"chunks {nchunks:3d} ({frames // nchunks:2d} frames, {frames // nchunks * HW * C * 4 / 1e6:6.1f} MB fp32 per tensor): "
              f"median {ts[2]:8.3f} ms  min {ts[0]:8.3f} ms", flush=True)


if __name__ == "__main__":
    main()
