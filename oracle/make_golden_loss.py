"""ORACLE support (build container only): known answers of the training loss from the UNMODIFIED
reference classes — StandardDiffusionLoss.get_loss / _forward (loss.py:115-273), EDMSampling, EDMWeighting
— imported through oracle/ref_shim.py (LPIPS / Lightning stubbed).
Re-run with:  python -m oracle.make_golden_loss   ->  tests/golden/loss_kat.pt"""
from __future__ import annotations

import importlib
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from oracle import ref_shim  # noqa: E402

OUT = ROOT / "tests" / "golden"
CFG = dict(
    harmonize_sigmas=True, focus_top=0.1, focus_steps=5000,
    batch2model_keys=["image_only_indicator", "num_video_frames"],
    loss_weighting_config={"target": "sgm.modules.diffusionmodules.loss_weighting.EDMWeighting",
                           "params": {"sigma_data": 1.0}},
    sigma_sampler_config={"target": "sgm.modules.diffusionmodules.sigma_sampling.EDMSampling",
                          "params": {"p_mean": 1.0, "p_std": 1.6}})        # train_kubric_max90.yaml:140-157


def main():
    ref_shim.reference_diffusion_module()           # installs the stubs loss.py needs
    L = importlib.import_module("sgm.modules.diffusionmodules.loss")
    g = torch.Generator().manual_seed(41)
    BT, T = 8, 4
    out = torch.randn(BT, 4, 6, 10, generator=g)
    tgt = torch.randn(BT, 4, 6, 10, generator=g)
    rand = torch.randn(BT, generator=g)
    loss = L.StandardDiffusionLoss(**CFG)
    sig = loss.sigma_sampler(BT, rand=rand)
    w = loss.loss_weighting(sig)
    res = {"out": out, "tgt": tgt, "rand": rand, "sigmas": sig, "weights": w, "T": T, "cases": []}
    for loss_type in ("l2", "l1"):
        lo = L.StandardDiffusionLoss(**dict(CFG, loss_type=loss_type))
        for step in (0, 20, 1000, 2500, 5000, 9000):
            val = lo.get_loss(out, tgt, w[:, None, None, None], {"global_step": step})
            res["cases"].append({"loss_type": loss_type, "step": step, "loss": val.clone()})
    # _forward with a recording denoiser: harmonised sigmas, noised input, weighting
    seen = {}

    def denoiser(network, noised, sigmas, cond, **kw):
        seen.update(noised=noised.clone(), sigmas=sigmas.clone(), kw=dict(kw))
        return noised * 0.5

    torch.manual_seed(77)
    batch = {"global_step": 2500, "num_video_frames": T, "image_only_indicator": torch.zeros(2, T)}
    val = loss._forward(None, denoiser, {}, tgt, batch)
    res["forward"] = {"seed": 77, "loss": val.clone(), "sigmas": seen["sigmas"], "noised": seen["noised"],
                      "kw_keys": sorted(seen["kw"])}
    OUT.mkdir(parents=True, exist_ok=True)
    torch.save(res, OUT / "loss_kat.pt")
    print("cases", len(res["cases"]), "forward loss", val[:3])


if __name__ == "__main__":
    main()
