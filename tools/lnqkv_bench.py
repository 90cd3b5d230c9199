"""tools/lnqkv_bench.py — LayerNorm + q | k | v at the 72 x 128 level (M = 258 048 tokens, C = 320, N = 960): the one-launch
kernel of gcd_amd/csrc/lnqkv.hip against gcd_layernorm_f16 + gcd_gemm_f16, HIP-event time per launch (caches flushed between
launches by a 1 GB fill).  GCD_LNQKV_FORM=0..3 selects the kernel form (bit 0: stores under the next chunk, bit 1: 4 waves x 64 tokens)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from gcd_amd import ops  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    M, C, N = 28 * 72 * 128, 320, 960
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn(M, C, device=dev, generator=g) * 1.3 + 0.4
    w16 = (torch.randn(N, C, device=dev, generator=g) / C ** 0.5).half()
    gamma, beta = torch.randn(C, device=dev, generator=g), torch.randn(C, device=dev, generator=g)
    wp = ops.lnqkv_pack(w16)
    out = torch.empty(M, N, device=dev, dtype=torch.float16)
    two = torch.empty(M, N, device=dev, dtype=torch.float16)
    x16 = torch.empty(M, C, device=dev, dtype=torch.float16)
    big = torch.empty(1 << 28, device=dev)

    def timed(fn):
        ts = []
        for it in range(9):
            big.zero_()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts = sorted(ts[2:])
        return ts[len(ts) // 2]

    t1 = timed(lambda: ops.lnqkv(x, gamma, beta, wp, out, M=M, N=N))
    tl = timed(lambda: ops.layernorm(x, gamma, beta, x16))
    tg = timed(lambda: ops.gemm(x16, w16, two, M=M, out_kind=ops.OUT_F16))
    err = float((out.float() - two.float()).norm() / two.float().norm())
    print(f"one launch {t1:7.1f} us = {(M * C * 4 + M * N * 2) / t1 / 1e6:.2f} TB/s | LayerNorm {tl:.1f} + GEMM {tg:.1f} = {tl + tg:.1f} us "
          f"| rel-L2 between them {err:.2e}")


if __name__ == "__main__":
    main()
