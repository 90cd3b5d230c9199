"""tools/encoder_bench.py — time the HIP first-stage Encoder (the VAE half of the conditioner
front-end, model.py:487-601 via AutoencoderKLModeOnly.encode) on one MI355X.

    python tools/encoder_bench.py [--frames 14] [--h 576] [--w 1024] [--iters 3] [--json out.json]

Encodes `frames` synthetic RGB images (random-init 128-channel SVD encoder: no checkpoint is available offline) to the
mode of the posterior, as the conditioner does once per clip (the reference encodes the conditioning frames in chunks of
`en_and_decode_n_samples_a_time`, encoders/modules.py:1071-1114).  Reports ms per encode, images/s, the GEMM share
measured with HIP events around every gcd_gemm_f16 launch, and the workspace.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=14)
    ap.add_argument("--h", type=int, default=576)
    ap.add_argument("--w", type=int, default=1024)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--json", type=str, default="")
    a = ap.parse_args()
    from gcd_amd import _lib, ops
    from gcd_amd.ae_encoder import Encoder, encode_mode
    dev = torch.device("cuda:0")
    kw = dict(attn_type="vanilla", double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3,
              ch=128, ch_mult=[1, 2, 4, 4], num_res_blocks=2, attn_resolutions=[], dropout=0.0)
    torch.manual_seed(0)
    enc = Encoder(**kw).to(dev).eval()
    quant = torch.nn.Conv2d(8, 8, 1).to(dev)
    x = torch.rand(a.frames, 3, a.h, a.w, device=dev) * 2 - 1
    z = encode_mode(enc, x, quant)                   # warm-up: packs weights, sizes the workspace
    torch.cuda.synchronize()
    assert z.shape == (a.frames, 4, a.h // 8, a.w // 8) and bool(torch.isfinite(z).all())
    t = []
    for _ in range(a.iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        encode_mode(enc, x, quant)
        e1.record()
        torch.cuda.synchronize()
        t.append(e0.elapsed_time(e1))
    ms = sorted(t)[len(t) // 2]
    prof = ops.start_profile()
    encode_mode(enc, x, quant)
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
    eng = getattr(enc, "engine", None)
    res = dict(workload=f"Encoder ch128 [1,2,4,4] + quant_conv + mode, {a.frames}x3x{a.h}x{a.w} images -> "
                        f"{a.frames}x4x{a.h // 8}x{a.w // 8} latents",
               ms_per_encode=round(ms, 2), images_per_s=round(a.frames / ms * 1e3, 2),
               gemm_ms=round(gemm_ms, 2), gemm_tflop=round(gemm_fl * 1e-12, 2),
               gemm_tflops=round(gemm_fl / gemm_ms * 1e-9, 1) if gemm_ms else None,
               workspace_gb=round(eng.ws.nbytes() / 2 ** 30, 2) if eng is not None and eng.ws is not None else None)
    print(json.dumps(res))
    print(f"{'M':>9} {'N':>5} {'K':>5} mode  n   total ms   TF/s")
    for (M, N, K, mode), (n, tms, fl) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
        print(f"{M:9d} {N:5d} {K:5d} {mode:4d} {n:3d} {tms:9.2f} {fl / tms * 1e-9:7.1f}")
    if a.json:
        Path(a.json).parent.mkdir(parents=True, exist_ok=True)
        Path(a.json).write_text(json.dumps(res))


if __name__ == "__main__":
    main()
