"""tools/train_step_bench.py — one fine-tune step of the full-width Kubric VideoUNet at BASELINE.json
cfg4's per-GPU shape (batch 2 clips x 14 frames, 32x48 latents = 256x384 pixels, N = 28 frames, no CFG;
configs/train_kubric_max90.yaml:209-210,234) on the HIP training path: StandardDiffusionLoss forward,
backward through gcd_amd.autograd_ops, AdamHIP step.  Reports wall time per phase and the algorithmic
TFLOP/s (12.53 TFLOP forward at this shape; backward = 2x forward; the re-forward of the
activation checkpointing is executed but not counted).  Not the repo's headline metric.

    python tools/train_step_bench.py [--steps 3] [--latent 32x48]
"""
import argparse
import json
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bench import build_model  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--latent", default="32x48")
    ap.add_argument("--clips", type=int, default=2)
    ap.add_argument("--dtype", default="fp16", choices=["fp16", "bf16"],
                    help="operand type of the GEMM-family contractions, forward and backward (cfg4 names bf16)")
    a = ap.parse_args()
    from gcd_amd import autograd_ops as AO
    from gcd_amd import training as TR
    AO.set_train_dtype(a.dtype)
    dev = torch.device("cuda:0")
    T = 14
    h, w = (int(v) for v in a.latent.split("x"))
    BT = a.clips * T
    net = build_model(dev, seed=0).train()
    den = TR.TrainDenoiser({"target": "gcd_amd.denoiser_scaling.VScalingWithEDMcNoise"})
    loss_fn = TR.StandardDiffusionLoss(
        sigma_sampler_config={"target": "gcd_amd.training.EDMSampling", "params": {"p_mean": 1.0, "p_std": 1.6}},
        loss_weighting_config={"target": "gcd_amd.training.EDMWeighting", "params": {"sigma_data": 1.0}},
        focus_top=0.1, focus_steps=5000, batch2model_keys=["image_only_indicator", "num_video_frames"])
    opt = TR.AdamHIP(net.parameters(), lr=2e-5)
    g = torch.Generator(device=dev).manual_seed(1)
    x0 = torch.randn(BT, 4, h, w, generator=g, device=dev)
    cond = {"crossattn": torch.randn(BT, 1, 1024, generator=g, device=dev),
            "concat": torch.randn(BT, 4, h, w, generator=g, device=dev) * 0.8,
            "vector": torch.randn(BT, 896, generator=g, device=dev).clamp(-1, 1)}
    batch = {"global_step": 2500, "num_video_frames": T, "image_only_indicator": torch.zeros(a.clips, T, device=dev)}
    scale = 1024.0
    times = []
    for it in range(a.steps + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loss = loss_fn._forward(net, den, cond, x0, batch).mean()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        opt.zero_grad()
        (loss * scale).backward()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        opt.step(grad_scale=1.0 / scale)
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        if it:
            times.append((t1 - t0, t2 - t1, t3 - t2))
        finite = bool(torch.isfinite(loss))
    f, b, o = (sorted(t[i] for t in times)[len(times) // 2] for i in range(3))
    tf_fwd = 12.531 * (h * w) / (32 * 48) * a.clips / 2
    print(json.dumps({
        "what": "one fine-tune step, full-width Kubric VideoUNet, HIP training path", "gemm_operands": a.dtype,
        "frames": BT, "latent": [h, w], "forward_s": round(f, 3), "backward_s": round(b, 3), "adam_s": round(o, 3),
        "step_s": round(f + b + o, 3), "algorithmic_tflop": round(3 * tf_fwd, 2),
        "tflops": round(3 * tf_fwd / (f + b + o), 1), "loss_finite": finite,
        "peak_mem_gib": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)}))


if __name__ == "__main__":
    main()
