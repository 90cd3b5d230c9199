"""tools/wgrad_ab.py — the weight-gradient kernel that reads its operands through ds_read_b64_tr_b16
(gcd_wgrad_tr_f16, libgcd_amd_train.so) against the round-3 path (transposed copies + split-K gcd_gemm_f16), in ONE
process on one MI355X:

  1. unit check: dW = dY^T X of both paths against fp32 torch on the fine-tune step's shapes (fp16 and bf16, one ragged);
  2. the whole fine-tune step at cfg4's shape (tools/train_step_bench.py's setup), the two paths INTERLEAVED step by step
     on the same model / batch: median step time per path, and the parameter gradients of one path against the other.

    python tools/wgrad_ab.py [--steps 4] [--json out.json]
"""
import argparse
import json
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bench import build_model  # noqa: E402


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--json", default="")
    a = ap.parse_args()
    from gcd_amd import autograd_ops as AO
    from gcd_amd import training as TR
    dev = torch.device("cuda:0")
    out = {"unit": []}
    g = torch.Generator(device=dev).manual_seed(3)
    for (M, N, K, dt) in [(43008, 320, 1280, torch.float16), (43008, 2560, 320, torch.bfloat16),
                          (2688, 1280, 1280, torch.float16), (1000, 208, 336, torch.float16),
                          (10752, 640, 5760, torch.bfloat16)]:
        dy = (torch.randn(M, N, generator=g, device=dev) * 0.5).to(dt)
        x = torch.randn(M, K, generator=g, device=dev).to(dt)
        ref = dy.float().t() @ x.float()
        res = {}
        for impl in ("gemm", "tr"):
            AO.set_wgrad_impl(impl)
            dw = AO._wgrad(dy, x)
            torch.cuda.synchronize()
            res[impl] = rel(dw, ref)
        out["unit"].append(dict(M=M, N=N, K=K, dtype=str(dt).split(".")[-1], **{f"rel_l2_{k}": v for k, v in res.items()}))
        print(out["unit"][-1], flush=True)
        assert res["tr"] < 2e-3, res
    # ---- the step, interleaved ----
    T, h, w, clips = 14, 32, 48, 2
    BT = clips * T
    net = build_model(dev, seed=0).train()
    den = TR.TrainDenoiser({"target": "gcd_amd.denoiser_scaling.VScalingWithEDMcNoise"})
    loss_fn = TR.StandardDiffusionLoss(
        sigma_sampler_config={"target": "gcd_amd.training.EDMSampling", "params": {"p_mean": 1.0, "p_std": 1.6}},
        loss_weighting_config={"target": "gcd_amd.training.EDMWeighting", "params": {"sigma_data": 1.0}},
        focus_top=0.1, focus_steps=5000, batch2model_keys=["image_only_indicator", "num_video_frames"])
    gg = torch.Generator(device=dev).manual_seed(1)
    x0 = torch.randn(BT, 4, h, w, generator=gg, device=dev)
    cond = {"crossattn": torch.randn(BT, 1, 1024, generator=gg, device=dev),
            "concat": torch.randn(BT, 4, h, w, generator=gg, device=dev) * 0.8,
            "vector": torch.randn(BT, 896, generator=gg, device=dev).clamp(-1, 1)}
    batch = {"global_step": 2500, "num_video_frames": T, "image_only_indicator": torch.zeros(clips, T, device=dev)}
    scale = 1024.0

    def fwd_bwd(impl, seed):
        AO.set_wgrad_impl(impl)
        torch.manual_seed(seed)                       # the loss draws sigma / noise from the default generator
        for p in net.parameters():
            p.grad = None
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loss = loss_fn._forward(net, den, cond, x0, batch).mean()
        (loss * scale).backward()
        torch.cuda.synchronize()
        return time.perf_counter() - t0, float(loss)

    fwd_bwd("gemm", 0)                                # warm-up: packs, allocator
    fwd_bwd("tr", 0)
    times = {"gemm": [], "tr": []}
    for it in range(a.steps):
        for impl in ("gemm", "tr") if it % 2 == 0 else ("tr", "gemm"):
            dt, _ = fwd_bwd(impl, 100 + it)
            times[impl].append(dt)
    # gradients of the two paths on the same draw
    _, l0 = fwd_bwd("gemm", 7)
    ref = {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}
    _, l1 = fwd_bwd("tr", 7)
    num = den2 = 0.0
    for n, p in net.named_parameters():
        if p.grad is not None and n in ref:
            num += float((p.grad.double() - ref[n].double()).pow(2).sum())
            den2 += float(ref[n].double().pow(2).sum())
    med = {k: sorted(v)[len(v) // 2] for k, v in times.items()}
    out.update(forward_backward_s_gemm=round(med["gemm"], 4), forward_backward_s_tr=round(med["tr"], 4),
               all_s={k: [round(x, 4) for x in v] for k, v in times.items()},
               gradients_tr_vs_gemm_rel_l2=(num / max(den2, 1e-300)) ** 0.5, loss_gemm=l0, loss_tr=l1)
    print(json.dumps(out))
    if a.json:
        Path(a.json).parent.mkdir(parents=True, exist_ok=True)
        Path(a.json).write_text(json.dumps(out))


if __name__ == "__main__":
    main()
