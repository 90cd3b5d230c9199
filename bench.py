#!/usr/bin/env python
"""bench.py — BASELINE.json's metric on MI355X: UNet denoise steps/sec on 14-frame 576x1024 SVD
latents (14 x 72 x 128 x 4), N GPUs of one node.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one EulerEDM sampler_step on one 14-frame clip = EDM scalings + one VideoUNet forward
on 28 frames (classifier-free guidance doubles the batch, guiders.py:89-100) + guidance + Euler
update: 89.604 algorithmic TFLOP.  Every rank owns one clip (weak scaling: clips
are independent, the reference itself runs one replica per GPU with no communication,
scripts/test.py:1059-1084); the only collective is the RCCL all-gather of the final latents, after
the timed region.  Weights are random-init tensors of the Kubric architecture (the 61 zero-init
tensors re-drawn so the network is not identically 0), inputs synthetic and resident in HBM.

Prints ONE JSON line on rank 0 (see the driver contract), including
  roofline      dominant kernel family (gemm_pp_kernel / gemm_f16_kernel, the MFMA implicit GEMMs): algorithmic
                FLOPs of all its launches in one step / the sum of their durations, measured with
                HIP events on the launch stream in an instrumented eager step right after the timed
                region (the timed region itself replays a hipGraph, which has no per-kernel hooks);
  cpu_baseline  the CPU oracle (oracle/svd_unet_ref.py, a port of the reference's PyTorch path)
                timed on this box's host cores on a bounded sample (one step at 14 x 32 x 32
                latents), scaled to the metric's unit by the algorithmic FLOP ratio.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

import torch  # noqa: E402

STEP_TFLOP = {(72, 128): 89.604, (32, 48): 12.531, (32, 32): 8.250}   # SURVEY.md §8(d), N = 28
PEAK_MFMA_TFLOPS = 2500.0   # dense fp16/bf16 MFMA, MI355X_MICROARCH.md


def kubric_kwargs():
    """configs/infer_kubric.yaml:18-40 (attn type string is irrelevant to gcd_amd)."""
    return dict(adm_in_channels=768, num_classes="sequential", use_checkpoint=True, in_channels=8,
                out_channels=4, model_channels=320, attention_resolutions=[4, 2, 1],
                num_res_blocks=2, channel_mult=[1, 2, 4, 4], num_head_channels=64,
                use_linear_in_transformer=True, transformer_depth=1, context_dim=1024,
                spatial_transformer_attn_type="softmax-xformers", extra_ff_mix_layer=True,
                use_spatial_context=True, merge_strategy="learned_with_images",
                video_kernel_size=[3, 1, 1], aux_emb_dim=128, aux_zero_init=False)


def build_model(dev, seed=0):
    from gcd_amd.video_model import VideoUNet
    torch.manual_seed(seed)
    with torch.device(dev):
        net = VideoUNet(**kubric_kwargs())
    g = torch.Generator(device=dev).manual_seed(seed + 1)
    with torch.no_grad():
        for name, p in net.named_parameters():   # re-draw the zero_module tensors
            if float(p.abs().max()) == 0.0 and p.dim() >= 2:
                fan_in = p[0].numel()
                p.copy_(torch.randn(p.shape, generator=g, device=dev) / math.sqrt(fan_in))
    return net.eval()


def synth_inputs(dev, T, h, w, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    n = T
    noise = torch.randn(n, 4, h, w, generator=g, device=dev)
    c = {"crossattn": torch.randn(n, 1, 1024, generator=g, device=dev),
         "concat": torch.randn(n, 4, h, w, generator=g, device=dev) * 0.8,
         "vector": torch.randn(n, 896, generator=g, device=dev).clamp(-1, 1)}
    uc = {"crossattn": torch.zeros_like(c["crossattn"]), "concat": torch.zeros_like(c["concat"]),
          "vector": c["vector"].clone()}
    return noise, c, uc


def cpu_baseline(net, T, seed):
    """Oracle (port of the reference's CPU path) on the host cores.  Bounded sample: one sampler
    step at 14x16x16 latents first; if that took < 6 s, one more at 14x32x32 (the reported one).
    Threads are capped at 32: on a 256-core host torch's intra-op pool gets *slower* beyond that on
    these conv / GEMM sizes (measured 0.02 TFLOP/s with 256 threads vs 0.6 with 8)."""
    from oracle import svd_unet_ref as O
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    sd = {k: v.detach().float().cpu() for k, v in net.state_dict().items()}
    scale = O.guider_scale(T)
    ioi2 = torch.zeros(2, T)

    def one(hw):
        g = torch.Generator().manual_seed(seed)
        x = torch.randn(T, 4, hw, hw, generator=g) * 700.0
        c = {"crossattn": torch.randn(T, 1, 1024, generator=g),
             "concat": torch.randn(T, 4, hw, hw, generator=g),
             "vector": torch.randn(T, 896, generator=g).clamp(-1, 1)}
        uc = {"crossattn": torch.zeros_like(c["crossattn"]), "concat": torch.zeros_like(c["concat"]),
              "vector": c["vector"].clone()}
        t0 = time.perf_counter()
        with torch.no_grad():
            O.sampler_step(sd, O.KUBRIC, x, 700.0, 545.7, c, uc, T, ioi2, scale)
        return time.perf_counter() - t0

    hw, tf = 16, STEP_TFLOP[(32, 32)] * (16 * 16) / (32 * 32)    # FLOPs scale ~ with pixels here
    dt = one(16)
    if dt < 6.0:
        hw, tf = 32, STEP_TFLOP[(32, 32)]
        dt = one(32)
    tflops = tf / dt
    return dict(value=tflops / STEP_TFLOP[(72, 128)], unit="steps/s", cores=cores, kind="port",
                sample=f"1 EulerEDM step (UNet on 28 frames) at 14x{hw}x{hw} latents ~ {tf:.2f} TFLOP "
                       f"in {dt:.1f} s ({tflops:.2f} TFLOP/s fp32, {cores} threads), scaled to "
                       f"14x72x128 by algorithmic FLOPs ({STEP_TFLOP[(72, 128)]} TFLOP/step)",
                measured_steps_per_s_at_sample=1.0 / dt)


def _pmc_traffic():
    """HBM-side bytes per GEMM-family launch from the committed PMC passes of this same command
    (counters cannot be read from inside the process; None if the profile is not shipped)."""
    f = Path(__file__).resolve().parent / "profiles" / "r01v_hbm_traffic.json"
    try:
        return json.loads(f.read_text())["gemm"]["bytes_per_launch"]
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--latent", type=str, default="72x128", help="latent HxW (default 72x128)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--dump-profile", type=str, default=None,
                    help="write the per-launch HIP-event table of the instrumented step (JSON)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks "
                         f"(WORLD_SIZE={world})")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU path for the product")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    from gcd_amd import _lib, ops
    from gcd_amd.denoiser import Denoiser
    from gcd_amd.sampling import EulerEDMSampler, FusedDenoiser, FusedEulerLoop
    from gcd_amd.wrappers import OpenAIWrapper
    from gcd_amd.parallel import gather_clips
    lib = _lib.load()

    T = 14
    h, w = (int(v) for v in args.latent.split("x"))
    net = build_model(dev, seed=0)                      # same weights on every rank (replicas)
    noise, c, uc = synth_inputs(dev, T, h, w, seed=100 + rank)   # a different clip per rank
    nsched = max(args.steps + args.warmup, 2)
    sampler = EulerEDMSampler(
        discretization_config={"target": "gcd_amd.discretizer.EDMDiscretization",
                               "params": {"sigma_max": 700.0}},
        num_steps=nsched,
        guider_config={"target": "gcd_amd.guiders.LinearPredictionGuider",
                       "params": {"num_frames": T, "max_scale": 1.5, "min_scale": 1.0}},
        device="cuda")
    sampler.use_graph = not args.no_graph
    fd = FusedDenoiser(Denoiser({"target": "gcd_amd.denoiser_scaling.VScalingWithEDMcNoise"}),
                       OpenAIWrapper(net), num_video_frames=T,
                       image_only_indicator=torch.zeros(2, T, device=dev))
    assert sampler._can_fuse(fd, noise, c, uc)
    loop = FusedEulerLoop(sampler, fd, noise, c, uc)

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with loop:
        # warm-up: >= 2 steps so that the eager step and the graph capture are outside the timing
        nwarm = max(args.warmup, 2 if sampler.use_graph else 1)
        for i in range(nwarm):
            loop.step(i)
        barrier()
        t0 = time.perf_counter()
        e0.record(loop.side)
        for i in range(args.steps):
            loop.step(nwarm + i)
        e1.record(loop.side)
        barrier()
        wall = time.perf_counter() - t0
        ev_ms = e0.elapsed_time(e1)
        finite = bool(torch.isfinite(loop.x).all())

        # ---- instrumented eager step: per-launch HIP events on the launch stream ----
        prof = ops.start_profile()
        loop.sig.copy_(loop.sigmas[0:2])
        loop.launch_step()
        loop.side.synchronize()
        ops.stop_profile()
    kinds = {}
    table = []
    import ctypes as C
    for r in prof:
        ms = C.c_float()
        _lib.check(lib.gcd_event_elapsed_ms(r["start"], r["stop"], C.byref(ms)))
        k = kinds.setdefault(r["kind"], dict(flops=0.0, ms=0.0, launches=0))
        k["flops"] += r["flops"]
        k["ms"] += ms.value
        k["launches"] += 1
        table.append({kk: vv for kk, vv in r.items() if kk not in ("start", "stop")} | {"ms": ms.value})
        lib.gcd_event_destroy(r["start"])
        lib.gcd_event_destroy(r["stop"])
    loop.close()
    if args.dump_profile and rank == 0:
        Path(args.dump_profile).parent.mkdir(parents=True, exist_ok=True)
        Path(args.dump_profile).write_text(json.dumps(table))

    # ---- aggregate over ranks ----
    elapsed = torch.tensor([wall], device=dev, dtype=torch.float64)
    if dist is not None:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    t_gather0 = time.perf_counter()
    gathered = gather_clips(loop.x, dist)             # the one collective of the path
    torch.cuda.synchronize(dev)
    gather_ms = (time.perf_counter() - t_gather0) * 1e3
    elapsed_s = float(elapsed.item())

    if rank == 0:
        ms_per_step = elapsed_s * 1e3 / args.steps
        steps_per_s = world * args.steps / elapsed_s
        step_tf = STEP_TFLOP.get((h, w))
        gk = kinds.get("gemm", dict(flops=0.0, ms=1.0, launches=0))
        ak = kinds.get("attn_spatial", dict(flops=0.0, ms=1.0, launches=0))
        gemm_tflops = gk["flops"] / (gk["ms"] * 1e-3) / 1e12
        out = {
            "metric": "UNet denoise steps/sec, 14-frame 576x1024 SVD latents",
            "value": round(steps_per_s, 4), "unit": "steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": f"kubric_gradual_max90 VideoUNet (1.53 B params, random init), one "
                                   f"14-frame clip per GPU at {h}x{w} latents, EulerEDM step with "
                                   f"CFG (28 frames per UNet forward)",
                       "latent": [T, h, w, 4], "sampler": "EulerEDM + LinearPredictionGuider 1.0-1.5",
                       "parallelism": f"replica per GPU x{world}, RCCL all-gather of final latents",
                       "precision": "fp16 MFMA operands, fp32 accumulate / residual stream / softmax / norms",
                       "graph": sampler.use_graph},
            "roofline": {"bound": "mfma", "achieved": round(gemm_tflops, 2), "peak": PEAK_MFMA_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(gemm_tflops / PEAK_MFMA_TFLOPS, 4),
                         "traffic": _pmc_traffic(), "traffic_unit": "bytes per launch (fabric-side: HBM + Infinity Cache)",
                         "traffic_profile": "profiles/r01v_hbm_traffic.{txt,json}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command (tools/pmc_traffic.py; FETCH x2 per the gfx950 note of MI355X_MICROARCH.md), per shape: profiles/r01k_gemm_hbm_traffic.txt",
                         "kernel": "gemm_pp_kernel + gemm_f16_kernel (MFMA implicit-GEMM family: Linear, Conv2d 3x3/1x1, Conv3d (3,1,1))",
                         "launches_per_step": gk["launches"],
                         "algorithmic_tflop_per_step": round(gk["flops"] / 1e12, 3),
                         "kernel_ms_per_step": round(gk["ms"], 3)},
            "attention": {"kernel": "attn_spatial64_kernel (72x128, 36x64 tokens) + attn_spatial_kernel", "achieved": round(
                ak["flops"] / (ak["ms"] * 1e-3) / 1e12, 2), "unit": "TFLOP/s",
                "algorithmic_tflop_per_step": round(ak["flops"] / 1e12, 3),
                "kernel_ms_per_step": round(ak["ms"], 3), "launches_per_step": ak["launches"]},
            "frame_evals_per_s": round(steps_per_s * 2 * T, 2),
            "hip_event_ms_per_step_rank0": round(ev_ms / args.steps, 3),
            "gather_ms": round(gather_ms, 3), "gathered_shape": list(gathered.shape),
            "output_finite": finite,
            "workspace_gib": round(net.engine.ws.nbytes() / 2 ** 30, 2),
        }
        if step_tf is not None:
            out["step_tflops_per_gpu"] = round(step_tf * args.steps / elapsed_s, 2)
            out["step_frac_of_mfma_peak"] = round(step_tf * args.steps / elapsed_s / PEAK_MFMA_TFLOPS, 4)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(net, T, seed=5)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
