"""ORACLE support (build container only): goldens at the METRIC's own shapes, made by the UNMODIFIED
reference modules (oracle/ref_shim.py) with the full-width 1.53 B-parameter Kubric VideoUNet on the
procedural weights of oracle/weights.py (salt 2).  Re-run with:  python -m oracle.make_golden_fullres
(~6 min and ~16 GiB on 8 host cores; the fixtures are strided samples, < 1 MB together).

  unet_kubric_72x128.pt   one reference VideoUNet.forward on N = 28 frames of 72x128 latents — the
                          shape BASELINE.json's metric is quoted on (cfg1): 65 536 strided samples of the
                          output, 4 096 of every TimestepEmbedSequential output, and their norms
  step_kubric_32x32.pt    BASELINE.json cfg0: ONE EulerEDM sampler_step of the reference plugin stack
                          (EulerEDMSampler.sampler_step + LinearPredictionGuider + Denoiser +
                          OpenAIWrapper + VideoUNet) on a 14 x 32 x 32 x 4 latent, at a high and a
                          mid-schedule noise level: full x_next tensors
"""
from __future__ import annotations

import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from oracle import ref_shim, svd_unet_ref as O, weights  # noqa: E402
from oracle.make_golden import unet_inputs                # noqa: E402

OUT = ROOT / "tests" / "golden"
SALT = 2
FWD_SEED = 81
STEP_SEED = 82


def sample(t: torch.Tensor, n: int = 4096) -> torch.Tensor:
    """n strided samples; the index grid is computed in float64 (a float32 linspace rounds past the
    end of tensors with > 2^24 elements).  tests/test_unet_gpu.py uses the same grid."""
    f = t.reshape(-1)
    idx = torch.linspace(0, f.numel() - 1, min(n, f.numel()), dtype=torch.float64).long()
    return f[idx].clone()


STEP_SIGMAS = [(700.0, 545.7294921875), (3.0, 2.0)]     # (sigma, next_sigma): first step / mid schedule


def main():
    torch.manual_seed(0)
    cfg = O.KUBRIC
    VideoUNet, OpenAIWrapper, Denoiser, EulerEDMSampler = ref_shim.reference_classes()
    t0 = time.time()
    with torch.device("meta"):
        net = VideoUNet(**cfg.as_reference_kwargs())
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    net = net.to_empty(device="cpu")
    net.load_state_dict(weights.synth_state_dict(shapes, SALT))
    net.eval()
    print(f"reference VideoUNet built in {time.time() - t0:.0f} s", flush=True)

    # ---- cfg0: one sampler_step at 14 x 32 x 32 through the reference's own classes ----
    T, h, w = 14, 32, 32
    noise, c, uc = weights.synth_inputs(1, T, h, w, cfg.context_dim,
                                        cfg.adm_in_channels + cfg.aux_emb_dim, STEP_SEED)
    sampler = EulerEDMSampler(num_steps=25, device="cpu", **ref_shim.SAMPLER_CFG)
    den = Denoiser(ref_shim.DENOISER_CFG)
    model = OpenAIWrapper(net)
    extra = {"num_video_frames": T, "image_only_indicator": torch.zeros(2, T)}

    def denoiser(inp, sigma, cc):
        return den(model, inp, sigma, cc, **extra)

    steps = []
    for sig, nxt in ([] if "--forward-only" in sys.argv else STEP_SIGMAS):
        x = noise * (1.0 + sig ** 2) ** 0.5
        s_in = x.new_ones([x.shape[0]])
        t1 = time.time()
        with torch.no_grad():
            xn = sampler.sampler_step(s_in * sig, s_in * nxt, denoiser, x, c, uc, gamma=0.0)
        print(f"sampler_step sigma {sig} -> {nxt}: {time.time() - t1:.1f} s, std {float(xn.std()):.4f}",
              flush=True)
        steps.append({"sigma": sig, "next_sigma": nxt, "x_next": xn.clone()})
    if steps:
        torch.save({"config": "KUBRIC", "salt": SALT, "T": T, "h": h, "w": w, "input_seed": STEP_SEED,
                    "steps": steps}, OUT / "step_kubric_32x32.pt")

    # ---- cfg1's shape: one forward at 28 x 72 x 128 with per-block taps ----
    T, h, w = 14, 72, 128
    x, ts, ctx, y, ioi = unet_inputs(cfg, T, h, w, FWD_SEED)
    taps = {}
    hooks = []
    for name, mod in list(net.input_blocks.named_children()):
        hooks.append(mod.register_forward_hook(
            lambda m, i, o, n=f"input_blocks.{name}": taps.__setitem__(
                n, (sample(o.detach()), float(o.detach().double().norm()), tuple(o.shape)))))
    hooks.append(net.middle_block.register_forward_hook(
        lambda m, i, o: taps.__setitem__(
            "middle_block", (sample(o.detach()), float(o.detach().double().norm()), tuple(o.shape)))))
    for name, mod in list(net.output_blocks.named_children()):
        hooks.append(mod.register_forward_hook(
            lambda m, i, o, n=f"output_blocks.{name}": taps.__setitem__(
                n, (sample(o.detach()), float(o.detach().double().norm()), tuple(o.shape)))))
    t1 = time.time()
    with torch.no_grad():
        out = net(x, ts, context=ctx, y=y, num_video_frames=T, image_only_indicator=ioi)
    dt = time.time() - t1
    for hk in hooks:
        hk.remove()
    print(f"forward 28 x 72 x 128: {dt:.0f} s, out std {float(out.std()):.4f}", flush=True)
    torch.save({
        "config": "KUBRIC", "salt": SALT, "T": T, "h": h, "w": w, "input_seed": FWD_SEED,
        "out_samples": sample(out, 65536), "out_norm": float(out.double().norm()),
        "out_shape": tuple(out.shape),
        "tap_samples": {k: v[0] for k, v in taps.items()},
        "tap_norms": {k: v[1] for k, v in taps.items()},
        "tap_shapes": {k: v[2] for k, v in taps.items()},
        "reference_cpu_seconds": dt, "reference_cpu_threads": torch.get_num_threads(),
    }, OUT / "unet_kubric_72x128.pt")


if __name__ == "__main__":
    main()
