"""tools/smallm_bench.py — the step's widest fp32 Linear (all 44 ResBlock emb_layers as one [N ~ 40 000, 1280] matrix, 28 rows;
gcd_amd/engine.py `_embeddings`) through gcd_linear_smallm_f32, HIP-event time per launch with flushed caches."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from gcd_amd import ops  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    big = torch.empty(1 << 28, device=dev)
    for (M, N, K) in ((28, 40000, 1280), (14, 40000, 1280), (28, 1280, 1280), (28, 1280, 320)):
        x = torch.randn(M, K, device=dev)
        w = torch.randn(N, K, device=dev) / K ** 0.5
        b = torch.randn(N, device=dev)
        y = torch.empty(M, N, device=dev)
        ts = []
        for it in range(8):
            big.zero_()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.linear_smallm(x, w, b, y, silu_in=True)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts = sorted(ts[2:])
        ref = torch.nn.functional.linear(torch.nn.functional.silu(x.double()), w.double(), b.double())
        err = float((y.double() - ref).norm() / ref.norm())
        print(f"M {M:3d} N {N:6d} K {K:5d}: {ts[len(ts) // 2]:8.1f} us (min {ts[0]:.1f})  W = {N * K * 4 / 1e6:6.1f} MB -> "
              f"{N * K * 4 / ts[len(ts) // 2] / 1e6:.2f} TB/s   rel-L2 vs fp64 {err:.2e}")


if __name__ == "__main__":
    main()
