"""ORACLE support (build container only): where the UNMODIFIED reference itself lands under the reduced-precision
autocast its own configs use, measured against its own fp32 run — the numbers the HIP path's tolerance bars are pinned to
(tests/golden/autocast_bars.json).

  cfg4   the full-width fine-tune step of oracle/make_golden_cfg4.py (same weights, same seeded batch) under
         torch.autocast("cpu", dtype=torch.bfloat16) — what `precision: bf16-mixed` of
         configs/train_kubric_max90.yaml does to the reference (Lightning wraps the step in autocast; loss.py:115-156 runs
         inside it) — compared with tests/golden/train_kubric_32x48.pt (the same step in fp32) by exactly the statistics of
         tests/test_backward_gpu.py::test_training_step_full_width_cfg4_vs_reference_golden: loss ratio, denoiser output
         rel-L2 over the 65 536 strided samples, sampled global gradient rel-L2, worst tensor, worst norm ratio.
  range  the GEGLU-range stress forward of oracle/make_golden_stress.py (unet_tiny_geglu_range.pt: Student-t weights, GEGLU
         gain 22) under torch.autocast("cpu", dtype=torch.float16) — the reference's own inference arithmetic
         (scripts/eval_utils.py:181-186) — compared with the fixture's fp32 output: rel-L2 of the output.  Also the Gaussian
         `unet_tiny.pt` forward, as the control.

  python -m oracle.make_autocast_bars cfg4     (~25 min, ~40 GiB)
  python -m oracle.make_autocast_bars range    (seconds)
Both merge their result into tests/golden/autocast_bars.json.
"""
from __future__ import annotations

import importlib
import json
import os
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from oracle import ref_shim, svd_unet_ref as O, weights  # noqa: E402

OUT = ROOT / "tests" / "golden"
BARS = OUT / "autocast_bars.json"


def rel_l2(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def merge(key, val):
    cur = json.loads(BARS.read_text()) if BARS.exists() else {}
    cur[key] = val
    BARS.write_text(json.dumps(cur, indent=1, sort_keys=True) + "\n")
    print(json.dumps({key: val}, indent=1))


def cfg4():
    from oracle.make_golden_cfg4 import B, T, SALT, inputs
    from oracle.make_golden_fullres import sample
    from oracle.make_golden_loss import CFG as LOSS_CFG
    G = torch.load(OUT / "train_kubric_32x48.pt")
    torch.manual_seed(0)
    torch.set_num_threads(int(os.environ.get("GOLDEN_THREADS", str(os.cpu_count()))))
    cfg = O.KUBRIC
    VideoUNet, OpenAIWrapper, Denoiser, _ = ref_shim.reference_classes()
    ref_shim.reference_diffusion_module()
    L = importlib.import_module("sgm.modules.diffusionmodules.loss")
    kw = cfg.as_reference_kwargs()
    kw["use_checkpoint"] = True
    with torch.device("meta"):
        net = VideoUNet(**kw)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    net = net.to_empty(device="cpu")
    net.load_state_dict(weights.synth_state_dict(shapes, SALT))
    net.train()
    x0, noise, cond, sig = inputs()
    den = Denoiser(ref_shim.DENOISER_CFG)
    model = OpenAIWrapper(net)
    loss_fn = L.StandardDiffusionLoss(**LOSS_CFG)
    noised = x0 + noise * sig[:, None, None, None]
    t0 = time.time()
    with torch.autocast("cpu", dtype=torch.bfloat16):
        out = den(model, noised, sig, cond, num_video_frames=T, image_only_indicator=torch.zeros(B, T))
        out = out.contiguous()
        w = loss_fn.loss_weighting(sig)[:, None, None, None]
        loss = loss_fn.get_loss(out.float(), x0, w, {"global_step": G["step"]}).mean()
    print(f"forward {time.time() - t0:.0f} s, loss {float(loss):.6f} (fp32 run: {G['loss']:.6f})", flush=True)
    t1 = time.time()
    loss.backward()
    print(f"backward {time.time() - t1:.0f} s", flush=True)
    e_out = rel_l2(sample(out.detach().float(), 65536), G["out_samples"])
    num = den_ = 0.0
    worst, worst_norm = ("", 0.0), ("", 0.0)
    total_ref = sum(v * v for v in G["grad_norms"].values()) ** 0.5
    seen = 0
    for name, prm in net.named_parameters():
        if name in G["dead"] or G["grad_norms"][name] < 1e-7 * total_ref or prm.grad is None:
            continue
        ref_s = G["grad_samples"][name].double()
        got = prm.grad.detach().float()
        got_s = sample(got, 128).double()
        num += float((got_s - ref_s).pow(2).sum())
        den_ += float(ref_s.pow(2).sum())
        seen += 1
        if got.numel() >= 64:
            e = float((got_s - ref_s).norm() / ref_s.norm().clamp_min(1e-30))
            if e > worst[1]:
                worst = (name, e)
            rn = abs(float(got.double().norm()) / G["grad_norms"][name] - 1.0)
            if rn > worst_norm[1]:
                worst_norm = (name, rn)
    merge("cfg4_reference_bf16_autocast_vs_own_fp32", {
        "what": "unmodified reference, one fine-tune step at cfg4's shape, torch.autocast(cpu, bfloat16) vs its fp32 run "
                "(tests/golden/train_kubric_32x48.pt); statistics of test_training_step_full_width_cfg4_vs_reference_golden",
        "loss": float(loss), "loss_fp32": G["loss"], "loss_ratio_minus_1": float(loss) / G["loss"] - 1.0,
        "output_rel_l2": e_out, "gradient_global_rel_l2": (num / den_) ** 0.5, "gradients_compared": seen,
        "worst_tensor": worst[0], "worst_tensor_rel_l2": worst[1],
        "worst_norm_ratio_tensor": worst_norm[0], "worst_norm_ratio_off": worst_norm[1],
        "cpu_seconds": time.time() - t0, "threads": torch.get_num_threads(), "torch": torch.__version__})


def stress_range():
    from oracle.make_golden import unet_inputs
    cfg = O.TINY
    VideoUNet, *_ = ref_shim.reference_classes()
    res = {}
    for fname in ("unet_tiny_geglu_range.pt", "unet_tiny_heavy.pt", "unet_tiny.pt"):
        G = torch.load(OUT / fname)
        torch.manual_seed(0)
        net = VideoUNet(**cfg.as_reference_kwargs()).eval()
        shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
        if "nu" in G:
            net.load_state_dict(weights.synth_state_dict_heavy(shapes, G["salt"], G["nu"], G["geglu_gain"]))
        else:
            net.load_state_dict(weights.synth_state_dict(shapes))      # make_golden.py: default salt
        x, ts, ctx, y, ioi = unet_inputs(cfg, G["T"], G["h"], G["w"], G["input_seed"])
        row = {}
        for dt in (torch.float16, torch.bfloat16):
            with torch.no_grad(), torch.autocast("cpu", dtype=dt):
                out = net(x, ts, context=ctx, y=y, num_video_frames=G["T"], image_only_indicator=ioi)
            row[str(dt).split(".")[-1]] = {"output_rel_l2": rel_l2(out.float(), G["out"]),
                                           "finite": bool(torch.isfinite(out).all())}
        with torch.no_grad():
            out32 = net(x, ts, context=ctx, y=y, num_video_frames=G["T"], image_only_indicator=ioi)
        row["float32_rerun"] = {"output_rel_l2": rel_l2(out32, G["out"])}
        res[fname] = row
    merge("tiny_forward_reference_autocast_vs_own_fp32", {
        "what": "unmodified reference VideoUNet forward (TINY width, the fixtures' weights and inputs) under "
                "torch.autocast(cpu, dtype) vs the fixture's fp32 output", "fixtures": res, "torch": torch.__version__})


if __name__ == "__main__":
    which = sys.argv[1:] or ["range"]
    if "range" in which:
        stress_range()
    if "cfg4" in which:
        cfg4()
