"""tools/conv_narrow_bench.py — the UNet's output head (Conv2d(320, 4, 3, padding=1) on 28 frames of 72 x 128, N padded to 16)
on the N == 16 kernel of gcd_amd/csrc/conv_narrow.hip against the tile kernels (GCD_TUNE_GEMM_IMPL = 1 / 3), HIP-event time
per launch.  Run on the GPU box:  python tools/conv_narrow_bench.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from gcd_amd import ops, packing  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    frames, Cin, H, W = 28, 320, 72, 128
    M = frames * H * W
    g = torch.Generator(device=dev).manual_seed(0)
    a = torch.randn(M, Cin, device=dev, generator=g).half()
    w = (torch.randn(4, Cin, 3, 3, device=dev, generator=g) / (9 * Cin) ** 0.5)
    wp = packing.pack_conv3x3(w.cpu(), cout_pad=16).to(dev)
    b = torch.zeros(16, device=dev)
    out = torch.empty(M, 16, device=dev)
    big = torch.empty(1 << 28, device=dev)      # 1 GB: flushes the caches between launches
    conv = dict(Cin=Cin, Hi=H, Wi=W, Ho=H, Wo=W, stride=1, upsample=0)
    for name, impl in (("narrow", 0), ("general 128-row", 1), ("256x320 tile", 3)):
        ops.tune_set(ops.TUNE_GEMM_IMPL, impl)
        ts = []
        for it in range(8):
            big.zero_()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.gemm(a, wp, out, M=M, mode=ops.GEMM_CONV3X3, bias=b, conv=conv)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts = sorted(ts[2:])
        print(f"{name:18s} {ts[len(ts) // 2]:8.1f} us (min {ts[0]:.1f})   input read once = {M * Cin * 2 / 1e6:.0f} MB "
              f"-> {M * Cin * 2 / ts[len(ts) // 2] / 1e6:.2f} TB/s")
    ops.tune_set(ops.TUNE_GEMM_IMPL, 0)


if __name__ == "__main__":
    main()
