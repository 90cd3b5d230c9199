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
  roofline      dominant kernel family (gemm_p8_kernel / gemm_pp_kernel / gemm_f16_kernel, the MFMA implicit GEMMs): algorithmic
                FLOPs of all its launches in one step / the sum of their durations, measured with
                HIP events on the launch stream in an instrumented eager step right after the timed
                region (the timed region itself replays a hipGraph, which has no per-kernel hooks);
  cpu_baseline  the CPU oracle (oracle/svd_unet_ref.py, a port of the reference's PyTorch path)
                timed on this box's host cores on a bounded sample (one step at 14 x 32 x 32
                latents = BASELINE.json cfg0), scaled to the metric's unit by the algorithmic FLOP
                ratio; next to it the reference itself measured at the metric's shape;
  parity        the SAME cfg0 step (same weights, same inputs) run through the HIP path and compared
                with the oracle result of the cpu_baseline leg: rel-L2 of x_next; the bench exits
                non-zero if it exceeds 2e-3.

`python bench.py --gpus N` with N > 1 and no torchrun environment re-launches itself under
`torch.distributed.run` (one rank per GPU, RCCL); under the driver's own torchrun it just runs.
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


def pose_conditioner(dev, seed=7):
    """The `vector` half of GCD's conditioner (configs/infer_kubric.yaml:52-109): 3 x sinusoid(256)
    of fps_id / motion_bucket_id / cond_aug + SphericalEmbedder(13 -> 128) with seeded weights."""
    from gcd_amd.conditioning import GeneralConditioner
    P = "gcd_amd.conditioning."
    torch.manual_seed(seed)
    return GeneralConditioner([
        dict(input_key="fps_id", target=P + "ConcatTimestepEmbedderND", params=dict(outdim=256)),
        dict(input_key="motion_bucket_id", is_trainable=True, target=P + "ConcatTimestepEmbedderND",
             params=dict(outdim=256)),
        dict(input_key="cond_aug", target=P + "ConcatTimestepEmbedderND", params=dict(outdim=256)),
        dict(input_key="scaled_relative_angles", is_trainable=True, target=P + "SphericalEmbedder",
             params=dict(embed_dim=128, zero_init=False))]).to(dev)


def synth_inputs(dev, T, h, w, seed, cond=None):
    """What DiffusionEngine.sample_video hands the sampler (diffusion.py:522-543):
    noise, c, uc.  crossattn / concat are N(0,1)-shaped stand-ins for the CLIP token and the VAE
    latents of the conditioning frames; `vector` is the real thing: the conditioner's embedding of
    fps_id 12, motion_bucket_id 127, cond_aug 0.02 and the gradual camera trajectory
    0 -> (30 deg, 15 deg, 1 m) over 13 of the 14 frames (eval_utils.py:235-245, common.py:450-479)."""
    from gcd_amd.camera import scaled_relative_angles
    g = torch.Generator(device=dev).manual_seed(seed)
    n = T
    noise = torch.randn(n, 4, h, w, generator=g, device=dev)
    cond = cond if cond is not None else pose_conditioner(dev)
    batch = {"fps_id": torch.full((n,), 12.0, device=dev),
             "motion_bucket_id": torch.full((n,), 127.0, device=dev),
             "cond_aug": torch.full((n,), 0.02, device=dev),
             "scaled_relative_angles": scaled_relative_angles(30.0, 15.0, 1.0, num_frames=T, device=dev)}
    c = {"crossattn": torch.randn(n, 1, 1024, generator=g, device=dev),
         "concat": torch.randn(n, 4, h, w, generator=g, device=dev) * 0.8,
         "vector": cond(batch)["vector"].detach()}
    assert c["vector"].shape == (n, 896)
    uc = {"crossattn": torch.zeros_like(c["crossattn"]), "concat": torch.zeros_like(c["concat"]),
          "vector": c["vector"].clone()}
    return noise, c, uc


PARITY_TOL = 2e-3
PARITY_SIGMAS = (3.0, 2.0)     # a mid-schedule step: x and the denoised prediction weigh alike in x_next
# the unmodified reference modules, fp32, 8 threads, ONE step at the metric's shape
REFERENCE_AT_SHAPE = dict(value=0.0054, unit="steps/s", cores=8, seconds_per_step=185.1,
                          provenance="BASELINE.md §2 / SURVEY.md §6: reference VideoUNet + EulerEDM "
                                     "step via oracle/ref_shim.py at 14x72x128 latents on the build "
                                     "container's 8 cores (the reference tree does not exist on the GPU "
                                     "box); re-measured while generating tests/golden/"
                                     "unet_kubric_72x128.pt (forward only)",
                          full_loop=dict(
                              what="the UNMODIFIED reference stack running cfg1's whole 25-step loop at 14x72x128 "
                                   "(oracle/make_golden_loop72.py, the run that produced tests/golden/"
                                   "loop_kubric_72x128.pt)",
                              cores=6, seconds_per_step_median=268.5, seconds_per_step_min=208.2,
                              seconds_per_step_max=461.7, steps_per_s=round(1.0 / 268.5, 5), loop_seconds=7036))


# The oracle port TIMED AT THE METRIC'S OWN SHAPE on a GPU box's host cores (`bench.py --cpu-baseline-full`, round 4,
# profiles/r04_bench_cpu_full.json): a recorded constant with provenance, quoted in every default line beside the live
# FLOP-scaled sample (which overstates the CPU: the small shape runs at 0.64 TFLOP/s, the metric's shape at 0.40).
PORT_MEASURED_AT_SHAPE = dict(value=0.0045, unit="steps/s", cores=32, seconds_per_step=222.1, kind="port, measured at shape",
                              provenance="profiles/r04_bench_cpu_full.json: `python bench.py --cpu-baseline-full` on an "
                                         "MI355X box of this pool, ONE oracle EulerEDM step (UNet on 28 frames, 89.604 "
                                         "TFLOP) at 14x72x128 latents in 222.1 s = 0.40 TFLOP/s fp32 on 32 threads; "
                                         "re-measure with --cpu-baseline-full (minutes)")


def cpu_baseline_and_parity(net, sampler, fd, T, dev, seed, cond, full=False):
    """cfg0 of BASELINE.json, twice on the same weights and inputs:
      * the oracle (port of the reference's CPU path) on the host cores, timed -> cpu_baseline.
        Bounded sample: one sampler step at 14x16x16 latents first; if that took < 6 s, one more at
        14x32x32 (the reported one).  Threads are capped at 32: on a 256-core host torch's intra-op
        pool gets *slower* beyond that on these conv / GEMM sizes;
      * the HIP path (one fused step) -> parity = rel-L2 of x_next against the oracle's.
    full (--cpu-baseline-full): additionally ONE oracle step at the metric's own shape, 14x72x128 (minutes of CPU time
    on the GPU box's host cores) -> cpu_baseline.value becomes that MEASURED rate (kind "port, measured at shape") with
    the scaled small-shape figure kept beside it, and parity is reported at 14x72x128 too."""
    from gcd_amd.sampling import FusedEulerLoop
    from oracle import svd_unet_ref as O
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    sd = {k: v.detach().float().cpu() for k, v in net.state_dict().items()}
    scale = O.guider_scale(T)
    ioi2 = torch.zeros(2, T)
    sig, nxt = PARITY_SIGMAS

    def one(hw, hw2=None):
        noise, c, uc = synth_inputs(dev, T, hw, hw2 or hw, seed, cond)
        x = noise * (1.0 + sig ** 2) ** 0.5
        cpu = lambda d: {k: v.cpu() for k, v in d.items()}   # noqa: E731
        t0 = time.perf_counter()
        with torch.no_grad():
            ref = O.sampler_step(sd, O.KUBRIC, x.cpu(), sig, nxt, cpu(c), cpu(uc), T, ioi2, scale)
        dt = time.perf_counter() - t0
        loop = FusedEulerLoop(sampler, fd, noise.clone(), c, uc)
        with loop:
            loop.x.copy_(x)
            loop.sig.copy_(torch.tensor([sig, nxt], device=dev))
            loop.launch_step()
        loop.close()
        got = loop.x.detach().cpu().double()
        rel = float((got - ref.double()).norm() / ref.double().norm())
        return dt, rel

    hw, tf = 16, STEP_TFLOP[(32, 32)] * (16 * 16) / (32 * 32)    # FLOPs scale ~ with pixels here
    dt, rel = one(16)
    if dt < 6.0:
        hw, tf = 32, STEP_TFLOP[(32, 32)]
        dt, rel = one(32)
    tflops = tf / dt
    base = dict(value=tflops / STEP_TFLOP[(72, 128)], unit="steps/s", cores=cores, kind="port",
                sample=f"1 EulerEDM step (UNet on 28 frames) at 14x{hw}x{hw} latents ~ {tf:.2f} TFLOP "
                       f"in {dt:.1f} s ({tflops:.2f} TFLOP/s fp32, {cores} threads), scaled to "
                       f"14x72x128 by algorithmic FLOPs ({STEP_TFLOP[(72, 128)]} TFLOP/step)",
                measured_steps_per_s_at_sample=1.0 / dt,
                measured_at_shape=PORT_MEASURED_AT_SHAPE,
                reference_at_shape=REFERENCE_AT_SHAPE)
    parity = dict(shape=[T, hw, hw, 4], config="BASELINE.json cfg0 (one EulerEDM step, CFG, 28-frame UNet)",
                  sigma=sig, next_sigma=nxt, rel_l2=rel, tol=PARITY_TOL, ok=bool(rel <= PARITY_TOL),
                  against="oracle/svd_unet_ref.py on the same weights and inputs (pinned to the reference "
                          "at this shape by tests/test_oracle.py::test_oracle_full_width_cfg0_step...)")
    if full:
        dtf, relf = one(72, 128)
        base.update(scaled_from_sample=dict(value=base["value"], sample=base["sample"]),
                    value=1.0 / dtf, kind="port, measured at shape",
                    sample=f"1 EulerEDM step (UNet on 28 frames) at 14x72x128 latents = {STEP_TFLOP[(72, 128)]} TFLOP in "
                           f"{dtf:.1f} s ({STEP_TFLOP[(72, 128)] / dtf:.2f} TFLOP/s fp32, {cores} threads): the oracle "
                           f"port timed at the metric's own shape on this box's host cores")
        parity["at_metric_shape"] = dict(shape=[T, 72, 128, 4], rel_l2=relf, tol=PARITY_TOL, ok=bool(relf <= PARITY_TOL))
        parity["ok"] = bool(parity["ok"] and relf <= PARITY_TOL)
    return base, parity


def sources_digest():
    from gcd_amd.csrc import build as _b
    return _b.sources_digest()


def _pmc_traffic():
    """HBM-side bytes per GEMM-family launch from the PMC passes of this same command (counters
    cannot be read from inside the process): profiles/hbm_traffic_latest.json, written by
    tools/pmc_traffic.py with the digest of the sources it was collected on.  Returns
    (bytes_per_launch or None, note)."""
    f = ROOT / "profiles" / "hbm_traffic_latest.json"
    try:
        d = json.loads(f.read_text())
    except Exception:
        return None, "no profiles/hbm_traffic_latest.json shipped"
    have, want = d.get("sources_digest"), sources_digest()
    if have != want:
        return None, (f"profiles/hbm_traffic_latest.json was collected on sources {have}, this build is "
                      f"{want}: stale, not quoted (re-run tools/pmc_traffic.sh)")
    return d["gemm"]["bytes_per_launch"], (f"{d.get('source', 'profiles/hbm_traffic_latest.json')}: rocprofv3 --pmc "
                                           "FETCH_SIZE / WRITE_SIZE passes of this command on these sources "
                                           "(FETCH x2 per the gfx950 note of MI355X_MICROARCH.md)")


TRAIN_ALG_TFLOP = 37.593      # one fine-tune step at cfg4's per-GPU shape: 12.531 forward x 3


class _AbiCallCounter:
    """Counts the C-ABI calls of one step (every `gcd_*` entry point of both libraries launches one kernel — a few
    launch two: split-K reduce, two-pass statistics).  rocprofv3's dispatch count of the same command is the exact
    figure (profiles/r06_train_kernel_stats.txt)."""

    def __init__(self):
        from gcd_amd import _lib
        self.libs = [(_lib.load(), _lib.SIGNATURES), (_lib.load_train(), _lib.TRAIN_SIGNATURES)]
        self.n = 0
        self.saved = []

    def __enter__(self):
        skip = ("gcd_last_error", "gcd_abi_version", "gcd_train_abi_version", "gcd_tune_set", "gcd_event_", "gcd_stream_",
                "gcd_graph_", "_bytes", "_supported", "_fusable", "_floats", "gcd_device_info")
        for lib, sigs in self.libs:
            for name in sigs:
                if any(k in name for k in skip):
                    continue
                fn = getattr(lib, name)
                self.saved.append((lib, name, fn))

                def wrap(*a, _fn=fn):
                    self.n += 1
                    return _fn(*a)
                setattr(lib, name, wrap)
        return self

    def __exit__(self, *exc):
        for lib, name, fn in self.saved:
            setattr(lib, name, fn)
        return False


def train_leg(dev, dtype: str, steps: int, clips: int = 2, latent=(32, 48)):
    """BASELINE.json cfg4 on ONE GPU: the fine-tune step (StandardDiffusionLoss forward + backward + Adam,
    sgm/modules/diffusionmodules/loss.py:115-273, sgm/models/diffusion.py:412-431) of the full-width Kubric VideoUNet on
    `clips` clips x 14 frames of 32 x 48 latents (configs/train_kubric_max90.yaml:209-234), GEMM operands in `dtype`.
    Wall time per phase (median of `steps`), algorithmic TFLOP/s, C-ABI calls of one step."""
    from gcd_amd import autograd_ops as AO
    from gcd_amd import training as TR
    AO.set_train_dtype(dtype)
    AO.PACK.clear()
    T = 14
    h, w = latent
    BT = clips * T
    net = build_model(dev, seed=0).train()
    den = TR.TrainDenoiser({"target": "gcd_amd.denoiser_scaling.VScalingWithEDMcNoise"}, use_checkpoint=None)
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
    batch = {"global_step": 2500, "num_video_frames": T, "image_only_indicator": torch.zeros(clips, T, device=dev)}
    scale = 1024.0
    times, calls, finite = [], 0, True
    try:
        for it in range(steps + 2):
            counter = _AbiCallCounter() if it == 1 else None      # the second step: packing and planning are done
            if counter:
                counter.__enter__()
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            loss = loss_fn._forward(net, den, cond, x0, batch).mean()
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            opt.zero_grad()
            (loss * scale).backward()
            torch.cuda.synchronize(dev)
            t2 = time.perf_counter()
            opt.step(grad_scale=1.0 / scale)
            torch.cuda.synchronize(dev)
            t3 = time.perf_counter()
            if counter:
                counter.__exit__()
                calls = counter.n
            elif it >= 2:
                times.append((t1 - t0, t2 - t1, t3 - t2))
            finite = finite and bool(torch.isfinite(loss))
    finally:
        AO.set_train_dtype("fp16")
        AO.PACK.clear()
    f, b, o = (sorted(t[i] for t in times)[len(times) // 2] for i in range(3))
    tf = TRAIN_ALG_TFLOP * (h * w) / (32 * 48) * clips / 2
    step = f + b + o
    res = {"gemm_operands": dtype, "clips": clips, "frames": BT, "latent": [h, w], "forward_s": round(f, 4),
           "backward_s": round(b, 4), "adam_s": round(o, 4), "step_s": round(step, 4), "algorithmic_tflop": round(tf, 2),
           "tflops": round(tf / step, 1), "frac_of_2p5_pflops": round(tf / step / 2500.0, 4),
           "c_abi_calls_per_step": calls, "loss_finite": finite, "engine": TR.TRAIN_ENGINE,
           "peak_mem_gib": round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 1)}
    del net, opt
    torch.cuda.empty_cache()
    return res


def train_parity(dev, dtype: str):
    """The same step against the UNMODIFIED reference's fp32 autograd run (tests/golden/train_kubric_32x48.pt, generated
    by oracle/make_golden_cfg4.py; the statistics of tests/test_backward_gpu.py): loss ratio, denoiser output and sampled
    global gradient rel-L2.  The oracle package is imported HERE only, as the checker of an already-timed path."""
    gold = ROOT / "tests" / "golden" / "train_kubric_32x48.pt"
    if not gold.exists():
        return None
    from gcd_amd import autograd_ops as AO
    from gcd_amd import training as TR
    from gcd_amd.video_model import VideoUNet
    from oracle import svd_unet_ref as O, weights
    from oracle.make_golden_cfg4 import inputs
    from oracle.make_golden_fullres import sample
    G = torch.load(gold)
    with torch.device("meta"):
        net = VideoUNet(**O.KUBRIC.as_reference_kwargs())
    sd = weights.synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, G["salt"])
    net = net.to_empty(device=dev)
    net.load_state_dict(sd)
    del sd
    net.train()
    x0, noise, cond, sig = inputs()
    AO.set_train_dtype(dtype)
    AO.PACK.clear()
    try:
        den = TR.TrainDenoiser({"target": "gcd_amd.denoiser_scaling.VScalingWithEDMcNoise"}, use_checkpoint=True)
        loss_fn = TR.StandardDiffusionLoss(
            sigma_sampler_config={"target": "gcd_amd.training.EDMSampling", "params": {"p_mean": 1.0, "p_std": 1.6}},
            loss_weighting_config={"target": "gcd_amd.training.EDMWeighting", "params": {"sigma_data": 1.0}},
            focus_top=0.1, focus_steps=5000, batch2model_keys=["image_only_indicator", "num_video_frames"])
        sg = sig.to(dev)
        out = den(net, (x0 + noise * sig[:, None, None, None]).to(dev), sg, {k: v.to(dev) for k, v in cond.items()},
                  num_video_frames=G["T"], image_only_indicator=torch.zeros(G["B"], G["T"], device=dev))
        wgt = loss_fn.loss_weighting(sg)[:, None, None, None]
        loss = loss_fn.get_loss(out, x0.to(dev), wgt, {"global_step": G["step"]}).mean()
        (loss * 1024.0).backward()
        torch.cuda.synchronize(dev)
    finally:
        AO.set_train_dtype("fp16")
        AO.PACK.clear()

    def rel(a, b):
        a, b = a.double(), b.double()
        return float((a - b).norm() / b.norm().clamp_min(1e-30))
    num = den_ = 0.0
    total_ref = sum(v * v for v in G["grad_norms"].values()) ** 0.5
    for name, prm in net.named_parameters():
        if name in G["dead"] or G["grad_norms"][name] < 1e-7 * total_ref or prm.grad is None:
            continue
        ref_s = G["grad_samples"][name].double()
        got_s = sample(prm.grad.detach().float().cpu() / 1024.0, 128).double()
        num += float((got_s - ref_s).pow(2).sum())
        den_ += float(ref_s.pow(2).sum())
    res = {"fixture": "tests/golden/train_kubric_32x48.pt (unmodified reference, fp32 autograd, CPU)",
           "loss_ratio_minus_1": float(loss.detach()) / G["loss"] - 1.0,
           "output_rel_l2": rel(sample(out.detach().cpu(), 65536), G["out_samples"]),
           "gradient_global_rel_l2": (num / den_) ** 0.5}
    del net
    torch.cuda.empty_cache()
    return res


def main_train(args):
    """`python bench.py --train`: BASELINE.json cfg4 on one GPU — ONE JSON line, separate from the sampler line (whose
    format the driver parses): the fine-tune step in bf16 (what cfg4 names) and fp16, at 2 clips (cfg4's per-GPU batch)."""
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU path for the product")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    legs = {dt: train_leg(dev, dt, max(args.steps // 2, 3)) for dt in ("bf16", "fp16")}
    parity = None
    if not args.no_cpu_baseline:
        parity = {dt: train_parity(dev, dt) for dt in ("bf16", "fp16")}
    bars = ROOT / "tests" / "golden" / "autocast_bars.json"
    ref_bf16 = None
    if bars.exists():
        ref_bf16 = json.loads(bars.read_text()).get("cfg4_reference_bf16_autocast_vs_own_fp32")
        if ref_bf16:
            ref_bf16 = {k: ref_bf16[k] for k in ("output_rel_l2", "gradient_global_rel_l2", "loss_ratio_minus_1")}
    line = {"metric": "fine-tune steps/sec, SVD-UNet (loss_fn forward+backward+Adam), 2 clips x 14 frames of 32x48 latents, 1 MI355X",
            "value": round(1.0 / legs["bf16"]["step_s"], 3), "unit": "steps/s", "n_gpus": 1, "higher_is_better": True,
            "dtype": "bf16", "data": "synthetic", "vs_baseline": None,
            "config": {"workload": "BASELINE.json configs[4] per-GPU shape (train_kubric_max90.yaml:209-234); single GPU: "
                                   "no gradient exchange (tools/first_multi_gpu.sh runs the DDP form)"},
            "train": legs, "parity_vs_reference_fp32_autograd": parity,
            "reference_own_bf16_autocast_vs_its_fp32": ref_bf16,
            "roofline": {"bound": "mfma", "achieved": legs["bf16"]["tflops"], "peak": 2500.0, "unit": "TFLOP/s",
                         "frac": legs["bf16"]["frac_of_2p5_pflops"], "traffic": None}}
    print(json.dumps(line))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--latent", type=str, default="72x128", help="latent HxW (default 72x128)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-full", action="store_true",
                    help="also time ONE oracle step at 14x72x128 on the host cores (minutes) and report it as "
                         "cpu_baseline.value (kind 'port, measured at shape')")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--backend", default="nccl",
                    help="torch.distributed backend for N > 1 (nccl = RCCL over xGMI; gloo only to exercise "
                         "the multi-rank path on a box with fewer GPUs than ranks, with GCD_BENCH_SHARE_GPU=1)")
    ap.add_argument("--repeats", type=int, default=5,
                    help="timed regions of --steps steps each: the first is the reported one (driver "
                         "contract), all of them feed timing_stats (median)")
    ap.add_argument("--dump-profile", type=str, default=None,
                    help="write the per-launch HIP-event table of the instrumented step (JSON)")
    ap.add_argument("--train", action="store_true",
                    help="instead of the sampler step: BASELINE.json cfg4's fine-tune step on one GPU, bf16 and fp16 "
                         "operands, with its parity against the reference's fp32 autograd golden (one JSON line)")
    args = ap.parse_args()

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.train:
        return main_train(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # self-launch: one rank per GPU under torch.distributed.run (what the driver does itself)
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
               f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
               str(Path(__file__).resolve())] + sys.argv[1:]
        os.execv(sys.executable, cmd)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} ranks")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU path for the product")
    share = os.environ.get("GCD_BENCH_SHARE_GPU") == "1"     # test rig: every rank on cuda:0 (gloo only)
    dev = torch.device("cuda", 0 if share else local_rank)
    torch.cuda.set_device(dev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            if share:
                raise SystemExit("GCD_BENCH_SHARE_GPU=1 needs --backend gloo (RCCL wants one GPU per rank)")
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    from gcd_amd import _lib, ops
    from gcd_amd.denoiser import Denoiser
    from gcd_amd.sampling import EulerEDMSampler, FusedDenoiser, FusedEulerLoop
    from gcd_amd.wrappers import OpenAIWrapper
    from gcd_amd.parallel import gather_clips
    lib = _lib.load()

    T = 14
    h, w = (int(v) for v in args.latent.split("x"))
    net = build_model(dev, seed=0)                      # same weights on every rank (replicas)
    cond = pose_conditioner(dev)
    noise, c, uc = synth_inputs(dev, T, h, w, seed=100 + rank, cond=cond)   # a different clip per rank
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
        # further regions of the same length: spread of the measurement (the first one is `value`)
        regions = [wall]
        for r in range(1, max(1, args.repeats)):
            barrier()
            t1 = time.perf_counter()
            for i in range(args.steps):
                loop.step(nwarm + r * args.steps + i)
            barrier()
            regions.append(time.perf_counter() - t1)

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
    elapsed = torch.tensor(regions, device=dev, dtype=torch.float64)
    per_rank = [elapsed.clone()]
    if dist is not None:
        per_rank = [torch.empty_like(elapsed) for _ in range(world)]
        dist.all_gather(per_rank, elapsed)
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    # how many ranks RCCL actually connected: a sum of ones over the nccl (= RCCL) backend, executed, not inferred
    rccl_ranks = 0
    if dist is not None and args.backend == "nccl":
        one = torch.ones(1, device=dev)
        dist.all_reduce(one)
        rccl_ranks = int(one.item())
        if dist.get_backend() != "nccl" or rccl_ranks != args.gpus:
            raise SystemExit(f"bench.py: {rccl_ranks} ranks answered over '{dist.get_backend()}' for --gpus {args.gpus}")
    t_gather0 = time.perf_counter()
    gathered = gather_clips(loop.x, dist)             # the one collective of the path
    torch.cuda.synchronize(dev)
    gather_ms = (time.perf_counter() - t_gather0) * 1e3
    elapsed_s = float(elapsed[0].item())
    region_ms = sorted(float(v) * 1e3 / args.steps for v in elapsed.tolist())

    if rank == 0:
        ms_per_step = elapsed_s * 1e3 / args.steps
        steps_per_s = world * args.steps / elapsed_s
        step_tf = STEP_TFLOP.get((h, w))
        gk = kinds.get("gemm", dict(flops=0.0, ms=1.0, launches=0))
        ak = kinds.get("attn_spatial", dict(flops=0.0, ms=1.0, launches=0))
        gemm_tflops = gk["flops"] / (gk["ms"] * 1e-3) / 1e12
        traffic, traffic_note = _pmc_traffic()
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
                         "traffic": traffic, "traffic_unit": "bytes per launch (fabric-side: HBM + Infinity Cache)",
                         "traffic_profile": traffic_note, "sources_digest": sources_digest(),
                         "kernel": "gemm_p8_kernel (+ gemm_pp_kernel, gemm_f16_kernel for the shapes it does not take; ff_fused_kernel = LayerNorm + FeedForward and lnqkv_kernel = LayerNorm + q|k|v at C = 320, conv3x3_narrow_kernel = the output head; MFMA implicit-GEMM family: Linear, Conv2d 3x3/1x1, Conv3d (3,1,1))",
                         "launches_per_step": gk["launches"],
                         "algorithmic_tflop_per_step": round(gk["flops"] / 1e12, 3),
                         "kernel_ms_per_step": round(gk["ms"], 3)},
            "attention": {"kernel": "attn_spatial64p_kernel (72x128, 36x64 tokens) + attn_spatial_kernel", "achieved": round(
                ak["flops"] / (ak["ms"] * 1e-3) / 1e12, 2), "unit": "TFLOP/s",
                "algorithmic_tflop_per_step": round(ak["flops"] / 1e12, 3),
                "kernel_ms_per_step": round(ak["ms"], 3), "launches_per_step": ak["launches"]},
            "frame_evals_per_s": round(steps_per_s * 2 * T, 2),
            "hip_event_ms_per_step_rank0": round(ev_ms / args.steps, 3),
            "timing_stats": {"regions": len(region_ms), "steps_per_region": args.steps,
                             "ms_per_step_sorted": [round(v, 3) for v in region_ms],
                             "ms_per_step_median": round(region_ms[len(region_ms) // 2], 3),
                             "reported_region": "first (max over ranks)"},
            "per_rank_ms_per_step": [round(float(t[0]) * 1e3 / args.steps, 3) for t in per_rank],
            "rccl_ranks": rccl_ranks,
            "dist_backend": args.backend if dist is not None else None,
            "gather_ms": round(gather_ms, 3), "gathered_shape": list(gathered.shape),
            "output_finite": finite,
            "workspace_gib": round(net.engine.ws.nbytes() / 2 ** 30, 2),
        }
        if step_tf is not None:
            out["step_tflops_per_gpu"] = round(step_tf * args.steps / elapsed_s, 2)
            out["step_frac_of_mfma_peak"] = round(step_tf * args.steps / elapsed_s / PEAK_MFMA_TFLOPS, 4)
        parity_ok = True
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"], out["parity"] = cpu_baseline_and_parity(net, sampler, fd, T, dev, 5, cond,
                                                                             full=args.cpu_baseline_full)
            parity_ok = out["parity"]["ok"]
        elif world > 1:
            out["cpu_baseline_note"] = ("N > 1 lines carry no cpu_baseline / parity: both are measured by the N = 1 "
                                        "line only (BENCH contract: rank 0 at N = 1)")
        print(json.dumps(out), flush=True)
        if not parity_ok:
            print(f"bench.py: PARITY FAILED: rel-L2 {out['parity']['rel_l2']:.3e} > {PARITY_TOL}",
                  file=sys.stderr, flush=True)
            sys.exit(3)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
