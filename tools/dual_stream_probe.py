"""tools/dual_stream_probe.py — does running the two classifier-free-guidance halves of a step as two
independent 14-frame UNet forwards on two HIP streams beat one 28-frame forward?  (Idea: the HBM-bound
kernels of one half — GroupNorm, LayerNorm, the K = 320 residual GEMM epilogues — could run under the
MFMA-bound kernels of the other.)  Both variants replay captured graphs; same weights, same work.

    python tools/dual_stream_probe.py [--iters 5]
"""
from __future__ import annotations

import argparse
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=5)
    a = ap.parse_args()
    import bench
    from gcd_amd.engine import UNetEngine, Workspace
    dev = torch.device("cuda:0")
    net = bench.build_model(dev)
    T, h, w = 14, 72, 128
    g = torch.Generator(device=dev).manual_seed(3)

    def inputs(n):
        x = torch.randn(n, 8, h, w, generator=g, device=dev)
        ts = torch.full((n,), 1.3, device=dev)
        ctx = torch.randn(n, 1, 1024, generator=g, device=dev)
        y = torch.randn(n, 896, generator=g, device=dev).clamp(-1, 1)
        ioi = torch.zeros(n // T, T, device=dev)
        return x, ts, ctx, y, ioi

    def make(engine, n):
        x, ts, ctx, y, ioi = inputs(n)
        out = torch.empty(n, 4, h, w, device=dev)
        alphas = None

        def run():
            engine.run(x, None, None, ts, ctx, y, T, ioi, out)
        return run, out

    eng28 = net.engine
    eng28.pack()
    run28, _ = make(eng28, 28)
    engs = []
    for _ in range(2):
        e = UNetEngine(net)
        e.packed = eng28.packed
        e.ws = Workspace(dev)
        engs.append(e)
    # NB: the shared `packed` dict caches the collapsed cross-attention per context tensor; give each
    # half its own cache by running them once eagerly (the cache key is the context pointer)
    run14 = [make(e, 14)[0] for e in engs]
    for f in [run28] + run14:
        f(); f()
    torch.cuda.synchronize()

    def capture(f, stream):
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.stream(stream):
            f()
            torch.cuda.synchronize()
            with torch.cuda.graph(gr, stream=stream):
                f()
        return gr

    s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
    g28 = capture(run28, s0)
    g14 = [capture(run14[0], s0), capture(run14[1], s1)]
    torch.cuda.synchronize()

    def timeit(fn):
        ts = []
        for _ in range(a.iters):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        return sorted(ts)[len(ts) // 2]

    def one28():
        with torch.cuda.stream(s0):
            g28.replay()
        torch.cuda.current_stream().wait_stream(s0)

    def two14_serial():
        with torch.cuda.stream(s0):
            g14[0].replay()
        torch.cuda.current_stream().wait_stream(s0)
        with torch.cuda.stream(s0):
            pass
        s1.wait_stream(s0)
        with torch.cuda.stream(s1):
            g14[1].replay()
        torch.cuda.current_stream().wait_stream(s1)

    def two14_concurrent():
        s0.wait_stream(torch.cuda.current_stream())
        s1.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s0):
            g14[0].replay()
        with torch.cuda.stream(s1):
            g14[1].replay()
        torch.cuda.current_stream().wait_stream(s0)
        torch.cuda.current_stream().wait_stream(s1)

    print(f"one 28-frame forward            : {timeit(one28):8.2f} ms")
    print(f"two 14-frame forwards, serial   : {timeit(two14_serial):8.2f} ms")
    print(f"two 14-frame forwards, 2 streams: {timeit(two14_concurrent):8.2f} ms")


if __name__ == "__main__":
    main()
