"""tools/decoder_bench.py — time the HIP VideoDecoder on one MI355X.

    python tools/decoder_bench.py [--frames 14] [--h 72] [--w 128] [--iters 3] [--json out.json]

Decodes one synthetic clip of `frames` latents (random-init 128-channel SVD decoder with procedural
weights: no checkpoint is available offline), reports ms per decode, decoded frames/s and the GEMM
share measured with HIP events around every gcd_gemm_f16 launch (ops.start_profile).  Run it under
rocprofv3 --kernel-trace --stats for the per-kernel table (profiles/r01_decoder_*).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=14)
    ap.add_argument("--h", type=int, default=72)
    ap.add_argument("--w", type=int, default=128)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--json", type=str, default="")
    a = ap.parse_args()
    from gcd_amd import _lib, ops
    from gcd_amd.temporal_ae import VideoDecoder
    dev = torch.device("cuda:0")
    kw = dict(attn_type="vanilla", double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3,
              ch=128, ch_mult=[1, 2, 4, 4], num_res_blocks=2, attn_resolutions=[], dropout=0.0,
              video_kernel_size=[3, 1, 1])
    torch.manual_seed(0)
    dec = VideoDecoder(**kw)
    with torch.no_grad():
        for n, p in dec.named_parameters():          # un-zero the zero-initialised convs
            if p.dim() > 1 and float(p.abs().max()) == 0.0:
                p.normal_(0, (p[0].numel()) ** -0.5)
    dec = dec.to(dev).eval()
    z = torch.randn(a.frames, 4, a.h, a.w, device=dev)
    out = dec(z, timesteps=a.frames)                 # warm-up: packs weights, sizes the workspace
    torch.cuda.synchronize()
    assert bool(torch.isfinite(out).all())
    t = []
    for _ in range(a.iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        dec(z, timesteps=a.frames)
        e1.record()
        torch.cuda.synchronize()
        t.append(e0.elapsed_time(e1))
    ms = sorted(t)[len(t) // 2]
    prof = ops.start_profile()
    dec(z, timesteps=a.frames)
    torch.cuda.synchronize()
    ops.stop_profile()
    lib = _lib.load()
    gemm_ms, gemm_fl, rows = 0.0, 0.0, {}
    for r in prof:
        v = C.c_float()
        _lib.check(lib.gcd_event_elapsed_ms(r["start"], r["stop"], C.byref(v)))
        lib.gcd_event_destroy(r["start"])
        lib.gcd_event_destroy(r["stop"])
        gemm_ms += v.value
        gemm_fl += r["flops"]
        key = (r["M"], r["N"], r["K"], r["mode"])
        acc = rows.setdefault(key, [0, 0.0, 0.0])
        acc[0] += 1
        acc[1] += v.value
        acc[2] += r["flops"]
    res = dict(workload=f"VideoDecoder ch128 [1,2,4,4], {a.frames}x{a.h}x{a.w} latents -> "
                        f"{a.frames}x3x{8 * a.h}x{8 * a.w} frames",
               ms_per_decode=round(ms, 2), frames_per_s=round(a.frames / ms * 1e3, 2),
               gemm_ms=round(gemm_ms, 2), gemm_tflop=round(gemm_fl * 1e-12, 2),
               gemm_tflops=round(gemm_fl / gemm_ms * 1e-9, 1) if gemm_ms else None,
               workspace_gb=round(dec.engine.ws.nbytes() / 2 ** 30, 2))
    print(json.dumps(res))
    print(f"{'M':>9} {'N':>5} {'K':>5} mode  n   total ms   TF/s")
    for (M, N, K, mode), (n, tms, fl) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
        print(f"{M:9d} {N:5d} {K:5d} {mode:4d} {n:3d} {tms:9.2f} {fl / tms * 1e-9:7.1f}")
    if a.json:
        Path(a.json).parent.mkdir(parents=True, exist_ok=True)
        Path(a.json).write_text(json.dumps(res))


if __name__ == "__main__":
    main()
