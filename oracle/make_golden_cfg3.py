"""ORACLE support (build container only): BASELINE.json cfg3 golden — the ParDom network (no
aux_label_emb, y is 768 wide; configs/infer_pardom.yaml) through the UNMODIFIED reference plugin stack
(EulerEDMSampler + LinearPredictionGuider + Denoiser + OpenAIWrapper + VideoUNet, oracle/ref_shim.py)
for the full **50-step** EulerEDM loop on a 14-frame clip, full 1.5 B-parameter width, 16x16 latents
(so the fp32 CPU run takes minutes, not hours).  Re-run with:  python -m oracle.make_golden_cfg3

  sampler_pardom_50.pt   final latents + x after steps 1, 10, 20, 30, 40, 50
"""
from __future__ import annotations

import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from oracle import ref_shim, svd_unet_ref as O, weights  # noqa: E402

OUT = ROOT / "tests" / "golden"
SALT = 3
SEED = 93
KEEP = (1, 10, 20, 30, 40, 50)


def main():
    torch.manual_seed(0)
    cfg = O.PARDOM
    VideoUNet, OpenAIWrapper, Denoiser, EulerEDMSampler = ref_shim.reference_classes()
    with torch.device("meta"):
        net = VideoUNet(**cfg.as_reference_kwargs())
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    net = net.to_empty(device="cpu")
    net.load_state_dict(weights.synth_state_dict(shapes, SALT))
    net.eval()
    T, h, w, steps = 14, 16, 16, 50
    noise, c, uc = weights.synth_inputs(1, T, h, w, cfg.context_dim,
                                        cfg.adm_in_channels + cfg.aux_emb_dim, SEED)
    sampler = EulerEDMSampler(num_steps=steps, device="cpu", **ref_shim.SAMPLER_CFG)
    den = Denoiser(ref_shim.DENOISER_CFG)
    model = OpenAIWrapper(net)
    extra = {"num_video_frames": T, "image_only_indicator": torch.zeros(2, T)}

    def denoiser(inp, sigma, cc):
        return den(model, inp, sigma, cc, **extra)

    trace = {}
    orig = sampler.sampler_step
    count = [0]

    def traced(*a, **k):
        r = orig(*a, **k)
        count[0] += 1
        if count[0] in KEEP:
            trace[count[0]] = r.detach().clone()
        return r

    sampler.sampler_step = traced
    t0 = time.time()
    with torch.no_grad():
        final = sampler(denoiser, noise.clone(), cond=c, uc=uc)
    print(f"50-step ParDom loop: {time.time() - t0:.0f} s, final std {float(final.std()):.4f}")
    torch.save({"config": "PARDOM", "salt": SALT, "T": T, "h": h, "w": w, "steps": steps,
                "input_seed": SEED, "final": final, "trace": trace}, OUT / "sampler_pardom_50.pt")


if __name__ == "__main__":
    main()
