"""tools/train_step_bench.py — one fine-tune step of the full-width Kubric VideoUNet at BASELINE.json
cfg4's per-GPU shape (batch 2 clips x 14 frames, 32x48 latents = 256x384 pixels, N = 28 frames, no CFG;
configs/train_kubric_max90.yaml:209-210,234) on the HIP training path: StandardDiffusionLoss forward,
backward through gcd_amd.autograd_ops, AdamHIP step.  Reports wall time per phase and the algorithmic
TFLOP/s (12.53 TFLOP forward at this shape; backward = 2x forward; the re-forward of the
activation checkpointing is executed but not counted).  Not the repo's headline metric.

    python tools/train_step_bench.py [--steps 3] [--latent 32x48]
    torchrun --nproc-per-node N tools/train_step_bench.py --ddp      cfg4's data parallelism: one rank per GPU, the
        gradient exchange by training.GradBucketer on RCCL; reports the all-reduce time left EXPOSED after backward
"""
import argparse
import json
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bench import build_model  # noqa: E402


def _graph_mode(net):
    from gcd_amd import train_plan as TP
    p = net.__dict__.get("_gcd_train_plan")
    return None if p is None or not TP.USE_GRAPH else p.graphed.mode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--latent", default="32x48")
    ap.add_argument("--clips", type=int, default=2)
    ap.add_argument("--dtype", default="fp16", choices=["fp16", "bf16"],
                    help="operand type of the GEMM-family contractions, forward and backward (cfg4 names bf16)")
    ap.add_argument("--checkpoint", default="net", choices=["net", "on", "off"],
                    help="activation checkpointing: the network's own flag (True in every GCD config), forced on, or off "
                         "(keep every unit's contexts: no re-forward in the backward pass; 288 GB of HBM hold them)")
    ap.add_argument("--ddp", action="store_true", help="run under torchrun: GradBucketer over the nccl (= RCCL) backend")
    a = ap.parse_args()
    import os
    from gcd_amd import autograd_ops as AO
    from gcd_amd import training as TR
    AO.set_train_dtype(a.dtype)
    dist = None
    local = 0
    if a.ddp:
        import torch.distributed as dist
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        dist.init_process_group(os.environ.get("GCD_DIST_BACKEND", "nccl"))
    dev = torch.device(f"cuda:{local}")
    T = 14
    h, w = (int(v) for v in a.latent.split("x"))
    BT = a.clips * T
    net = build_model(dev, seed=0).train()
    den = TR.TrainDenoiser({"target": "gcd_amd.denoiser_scaling.VScalingWithEDMcNoise"},
                           use_checkpoint={"net": None, "on": True, "off": False}[a.checkpoint])
    loss_fn = TR.StandardDiffusionLoss(
        sigma_sampler_config={"target": "gcd_amd.training.EDMSampling", "params": {"p_mean": 1.0, "p_std": 1.6}},
        loss_weighting_config={"target": "gcd_amd.training.EDMWeighting", "params": {"sigma_data": 1.0}},
        focus_top=0.1, focus_steps=5000, batch2model_keys=["image_only_indicator", "num_video_frames"])
    opt = TR.AdamHIP(net.parameters(), lr=2e-5)
    bucketer = None
    if dist is not None:
        # the parameters behind the one-key cross-attentions are never reached by the graph: named
        # up front, every bucket launches during the FIRST backward pass already
        dead = [p for n, p in net.named_parameters() if ".attn2.to_q." in n or ".attn2.to_k." in n or ".norm2." in n]
        from gcd_amd import train_plan as TP
        late = TP.late_parameters(net) if TR.TRAIN_ENGINE == "planned" else None
        bucketer = TR.GradBucketer(net.parameters(), dist, unused=dead, late=late)
    g = torch.Generator(device=dev).manual_seed(1)
    x0 = torch.randn(BT, 4, h, w, generator=g, device=dev)
    cond = {"crossattn": torch.randn(BT, 1, 1024, generator=g, device=dev),
            "concat": torch.randn(BT, 4, h, w, generator=g, device=dev) * 0.8,
            "vector": torch.randn(BT, 896, generator=g, device=dev).clamp(-1, 1)}
    batch = {"global_step": 2500, "num_video_frames": T, "image_only_indicator": torch.zeros(a.clips, T, device=dev)}
    scale = 1024.0
    times, exposed, launched = [], [], []
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
        if bucketer is not None:
            bucketer.finish()
            torch.cuda.synchronize()
            exposed.append(time.perf_counter() - t2)
            launched.append(bucketer.launched_during_backward)
        opt.step(grad_scale=1.0 / scale)
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        if it:
            times.append((t1 - t0, t2 - t1, t3 - t2))
        finite = bool(torch.isfinite(loss))
    f, b, o = (sorted(t[i] for t in times)[len(times) // 2] for i in range(3))
    tf_fwd = 12.531 * (h * w) / (32 * 48) * a.clips / 2
    extra = {}
    if dist is not None:
        ex = sorted(exposed[1:])[len(exposed[1:]) // 2] if len(exposed) > 1 else exposed[0]
        extra = {"world": dist.get_world_size(), "backend": dist.get_backend(), "buckets": len(bucketer.buckets),
                 "buckets_launched_during_backward_last_step": launched[-1] - (launched[-2] if len(launched) > 1 else 0),
                 "exposed_allreduce_plus_writeback_s": round(ex, 4)}
        o += ex
        if dist.get_rank() != 0:
            dist.barrier()
            dist.destroy_process_group()
            return
    print(json.dumps({**extra, 
        "what": "one fine-tune step, full-width Kubric VideoUNet, HIP training path", "gemm_operands": a.dtype,
        "frames": BT, "latent": [h, w], "forward_s": round(f, 3), "backward_s": round(b, 3), "adam_s": round(o, 3),
        "step_s": round(f + b + o, 3), "algorithmic_tflop": round(3 * tf_fwd, 2),
        "tflops": round(3 * tf_fwd / (f + b + o), 1), "loss_finite": finite,
        "peak_mem_gib": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1), "engine": TR.TRAIN_ENGINE, "checkpoint": a.checkpoint, "clips": a.clips,
        "hipgraph": _graph_mode(net)}))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
