"""ORACLE support (build container only): generate tests/golden/*.pt by running the UNMODIFIED
reference modules from /root/reference (via oracle/ref_shim.py) on procedural weights / inputs
(oracle/weights.py).  The fixtures pin oracle/svd_unet_ref.py (tests/test_oracle.py) and, through it,
the HIP path.  Re-run with:  python -m oracle.make_golden

Fixtures (all fp32, TINY config = model_channels 64, otherwise the Kubric topology):
  unet_tiny.pt     one VideoUNet.forward on N = 2*T = 8 frames (T = 4), 16x16 latents:
                   output + every TimestepEmbedSequential output (strided samples + norms)
  sampler_tiny.pt  EulerEDMSampler + LinearPredictionGuider + Denoiser + OpenAIWrapper + VideoUNet,
                   5 steps, T = 4, 8x8 latents: x after every step
  kat.pt           analytic known answers: sigma schedule (n=25, sigma_max=700), guider scale,
                   VScalingWithEDMcNoise at sigma in {700, 1, 0.002}, timestep_embedding rows
"""
from __future__ import annotations

import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from oracle import ref_shim, svd_unet_ref as O, weights  # noqa: E402

OUT = ROOT / "tests" / "golden"


def build_reference_unet(cfg):
    VideoUNet, *_ = ref_shim.reference_classes()
    net = VideoUNet(**cfg.as_reference_kwargs()).eval()
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    net.load_state_dict(weights.synth_state_dict(shapes))
    return net, shapes


def unet_inputs(cfg, T, h, w, seed=2):
    noise, c, uc = weights.synth_inputs(1, T, h, w, cfg.context_dim,
                                        cfg.adm_in_channels + cfg.aux_emb_dim, seed)
    x = torch.cat([torch.cat([noise, uc["concat"]], 1), torch.cat([noise, c["concat"]], 1)])
    ts = torch.linspace(-1.5, 1.63, 2 * T)
    ctx = torch.cat([uc["crossattn"], c["crossattn"]])
    y = torch.cat([uc["vector"], c["vector"]])
    ioi = torch.zeros(2, T)
    return x, ts, ctx, y, ioi


def sample(t: torch.Tensor, n: int = 4096) -> torch.Tensor:
    f = t.reshape(-1)
    idx = torch.linspace(0, f.numel() - 1, min(n, f.numel())).long()
    return f[idx].clone()


def main():
    torch.manual_seed(0)
    OUT.mkdir(parents=True, exist_ok=True)
    cfg = O.TINY
    net, shapes = build_reference_unet(cfg)

    # ---- single forward with per-block taps ----
    T, h, w = 4, 16, 16
    x, ts, ctx, y, ioi = unet_inputs(cfg, T, h, w)
    taps = {}
    hooks = []
    for name, mod in list(net.input_blocks.named_children()):
        hooks.append(mod.register_forward_hook(
            lambda m, i, o, n=f"input_blocks.{name}": taps.__setitem__(n, o.detach())))
    hooks.append(net.middle_block.register_forward_hook(
        lambda m, i, o: taps.__setitem__("middle_block", o.detach())))
    for name, mod in list(net.output_blocks.named_children()):
        hooks.append(mod.register_forward_hook(
            lambda m, i, o, n=f"output_blocks.{name}": taps.__setitem__(n, o.detach())))
    with torch.no_grad():
        out = net(x, ts, context=ctx, y=y, num_video_frames=T, image_only_indicator=ioi)
    for hk in hooks:
        hk.remove()
    torch.save({
        "config": "TINY", "T": T, "h": h, "w": w, "input_seed": 2,
        "out": out, "tap_samples": {k: sample(v) for k, v in taps.items()},
        "tap_norms": {k: float(v.double().norm()) for k, v in taps.items()},
        "tap_shapes": {k: tuple(v.shape) for k, v in taps.items()},
        "state_dict_shapes": shapes,
    }, OUT / "unet_tiny.pt")
    print("unet_tiny: out std", float(out.std()), "taps", len(taps))

    # ---- sampler loop through the reference's own plugin stack ----
    VideoUNet, OpenAIWrapper, Denoiser, EulerEDMSampler = ref_shim.reference_classes()
    T, h, w, steps = 4, 8, 8, 5
    noise, c, uc = weights.synth_inputs(1, T, h, w, cfg.context_dim,
                                        cfg.adm_in_channels + cfg.aux_emb_dim, seed=3)
    scfg = dict(ref_shim.SAMPLER_CFG)
    scfg["guider_config"] = {"target": scfg["guider_config"]["target"],
                             "params": {"num_frames": T, "max_scale": 1.5, "min_scale": 1.0}}
    sampler = EulerEDMSampler(num_steps=steps, device="cpu", **scfg)
    den = Denoiser(ref_shim.DENOISER_CFG)
    model = OpenAIWrapper(net)
    extra = {"num_video_frames": T, "image_only_indicator": torch.zeros(2, T)}
    trace = []

    def denoiser(inp, sigma, cc):
        return den(model, inp, sigma, cc, **extra)

    orig_step = sampler.sampler_step

    def traced_step(*a, **k):
        r = orig_step(*a, **k)
        trace.append(r.detach().clone())
        return r

    sampler.sampler_step = traced_step
    with torch.no_grad():
        final = sampler(denoiser, noise.clone(), cond=c, uc=uc)
    torch.save({"config": "TINY", "T": T, "h": h, "w": w, "steps": steps, "input_seed": 3,
                "trace": torch.stack(trace), "final": final}, OUT / "sampler_tiny.pt")
    print("sampler_tiny: final std", float(final.std()))

    # ---- analytic known answers from the reference's own small classes ----
    from sgm.modules.diffusionmodules.discretizer import EDMDiscretization
    from sgm.modules.diffusionmodules.denoiser_scaling import VScalingWithEDMcNoise
    from sgm.modules.diffusionmodules.guiders import LinearPredictionGuider
    from sgm.modules.diffusionmodules.util import timestep_embedding
    sig = EDMDiscretization(sigma_max=700.0)(25)
    sc = VScalingWithEDMcNoise()
    s = torch.tensor([700.0, 1.0, 0.002])
    torch.save({
        "sigmas_25_700": sig,
        "guider_scale_14": LinearPredictionGuider(max_scale=1.5, num_frames=14).scale,
        "scaling_sigma": s, "scaling": torch.stack(sc(s)),
        "temb_t": torch.tensor([0.0, 1.6378, -1.5534, 13.0]),
        "temb_320": timestep_embedding(torch.tensor([0.0, 1.6378, -1.5534, 13.0]), 320),
    }, OUT / "kat.pt")
    print("kat: sigmas", sig[:3].tolist(), "...", sig[-3:].tolist())


if __name__ == "__main__":
    main()
