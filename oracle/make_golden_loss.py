"""ORACLE support (build container only): known answers of the training loss from the UNMODIFIED
reference classes — StandardDiffusionLoss.get_loss / _forward (loss.py:115-273), EDMSampling, EDMWeighting
— imported through oracle/ref_shim.py (LPIPS / Lightning stubbed).
Re-run with:  python -m oracle.make_golden_loss   ->  tests/golden/loss_kat.pt
              python -m oracle.make_golden_loss pd ->  tests/golden/loss_pd_kat.pt  (ParallelDomain class re-weighting,
                                                        loss.py:196-230, configs/train_pardom_semantic.yaml:145-146)"""
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


def pd_semantic_frames(BT: int, H: int, W: int, seed: int = 43) -> torch.Tensor:
    """Synthetic batch['jpg'] semantic maps in [-1, 1]: a random background, rectangles painted in exact class colours
    (pedestrian, car, bus, bicyclist, truck), one rectangle a hair inside the 0.02 matching threshold and one a hair
    outside it; rectangle edges are NOT aligned to the 8 x 8 latent squares, so the area-averaged masks are fractional."""
    g = torch.Generator().manual_seed(seed)
    img = torch.rand(BT, 3, H, W, generator=g) * 2.0 - 1.0
    paint = [((220, 20, 60), 0.0), ((0, 0, 142), 0.0), ((0, 60, 100), 0.0), ((64, 64, 64), 0.0), ((0, 0, 70), 0.0),
             ((220, 20, 60), 0.017), ((0, 0, 142), 0.023)]
    for b in range(BT):
        for i, (rgb, off) in enumerate(paint):
            y0 = int(torch.randint(0, H - 13, (1,), generator=g)); x0 = int(torch.randint(0, W - 21, (1,), generator=g))
            hh = int(torch.randint(3, 13, (1,), generator=g)); ww = int(torch.randint(5, 21, (1,), generator=g))
            col = torch.tensor(rgb, dtype=torch.float32) / 127.5 - 1.0 + off
            img[b, :, y0:y0 + hh, x0:x0 + ww] = col[:, None, None]
    return img


def main_pd():
    ref_shim.reference_diffusion_module()
    L = importlib.import_module("sgm.modules.diffusionmodules.loss")
    g = torch.Generator().manual_seed(47)
    BT = 6
    out = torch.randn(BT, 4, 6, 10, generator=g)
    tgt = torch.randn(BT, 4, 6, 10, generator=g)
    rand = torch.randn(BT, generator=g)
    jpg = pd_semantic_frames(BT, 48, 80)
    base = L.StandardDiffusionLoss(**CFG)
    w = base.loss_weighting(base.sigma_sampler(BT, rand=rand))
    res = {"out": out, "tgt": tgt, "rand": rand, "weights": w, "jpg_seed": 43, "jpg_hw": (48, 80), "cases": []}
    import contextlib
    import io
    for pw, vw in ((7.0, 3.0), (7.0, 1.0), (1.0, 3.0)):          # train_pardom_semantic.yaml uses (7, 3)
        for loss_type in ("l2", "l1"):
            lo = L.StandardDiffusionLoss(**dict(CFG, loss_type=loss_type, pd_person_weight=pw, pd_vehicle_weight=vw))
            for step in (1, 2501, 5001):                         # (odd steps: the reference prints tensors on round ones)
                with contextlib.redirect_stdout(io.StringIO()):
                    val = lo.get_loss(out, tgt, w[:, None, None, None], {"global_step": step, "jpg": jpg})
                res["cases"].append({"person": pw, "vehicle": vw, "loss_type": loss_type, "step": step,
                                     "loss": val.clone()})
    torch.save(res, OUT / "loss_pd_kat.pt")
    print("pd cases", len(res["cases"]), res["cases"][0]["loss"][:3], res["cases"][1]["loss"][:3])


if __name__ == "__main__":
    main_pd() if sys.argv[1:] == ["pd"] else main()
