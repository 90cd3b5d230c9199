"""tools/graph_node_latency.py — what one hipGraph NODE costs on replay against one direct launch on a stream, for small
kernels.  The fine-tune step under two hipGraphs (GCD_TRAIN_GRAPH=1) measures 0.166 s against 0.136-0.142 s eager
(DESIGN.md section 11): ~2 200 launches per step, most of them a few microseconds long.  This tool takes the model out of the
question: N launches of a tiny kernel (gcd_cast_f32_f16 on 64 elements) and of a ~20 us kernel (the same cast on 8 M
elements), (a) enqueued directly on a stream, (b) captured once with torch.cuda.graph and replayed; wall time per launch with
the stream idle at the start, median of 7.

    python tools/graph_node_latency.py [N=2000]
"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from gcd_amd import ops  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    dev = torch.device("cuda:0")
    rows = []
    for label, numel in (("tiny (64 elements)", 64), ("~20 us (8 M elements)", 8 << 20)):
        x = torch.randn(numel // 64, 64, device=dev)
        y = torch.empty(numel // 64, 64, device=dev, dtype=torch.float16)

        def body():
            for _ in range(n):
                ops.cast_f16(x, y)

        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            body()
            torch.cuda.synchronize()
            eager = []
            for _ in range(7):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                body()
                torch.cuda.synchronize()
                eager.append(time.perf_counter() - t0)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                body()
            g.replay()
            torch.cuda.synchronize()
            rep = []
            for _ in range(7):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                g.replay()
                torch.cuda.synchronize()
                rep.append(time.perf_counter() - t0)
            # kernel time alone: HIP events around the eager launches measure the same wall; one launch in isolation:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.cast_f16(x, y)
            e1.record()
            torch.cuda.synchronize()
            one = e0.elapsed_time(e1) * 1e3
        me, mr = sorted(eager)[3], sorted(rep)[3]
        rows.append((label, one, me / n * 1e6, mr / n * 1e6))
    print(f"{n} launches per measurement; microseconds per launch")
    print(f"{'kernel':26s} {'one launch (events)':>20s} {'stream, back to back':>22s} {'hipGraph replay':>17s}")
    for label, one, e, r in rows:
        print(f"{label:26s} {one:20.1f} {e:22.2f} {r:17.2f}")


if __name__ == "__main__":
    main()
