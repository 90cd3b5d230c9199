"""ORACLE support (build container only): the fp16 OPERAND-RANGE stress fixture.  Every other fixture uses Gaussian
weights of variance 1 / fan_in, under which no fp16 operand of the HIP path comes near the edges of the fp16 range.
Here the UNMODIFIED reference VideoUNet (oracle/ref_shim.py) runs the TINY width with
  * Student-t (nu = 3) weights — single weights tens of sigma out (oracle/weights.synth_tensor_heavy), and
  * the GEGLU projections scaled by GEGLU_GAIN so that the hidden tensor value * gelu(gate) — which the HIP path
    stores in fp16 (clamped to the fp16 range in the epilogue) — reaches thousands,
and the fixture records, beside the output and the per-block taps, the largest |hidden| the reference saw in any GEGLU
(forward hook on sgm.modules.attention.GEGLU, attention.py:87-94) so that the test can state how close to 65504 the
case sits.  tests/test_unet_gpu.py::test_unet_forward_heavy_tailed_weights_vs_reference_golden holds the HIP path to
"finite and < 2e-3" on it.

  python -m oracle.make_golden_stress        -> tests/golden/unet_tiny_heavy.pt
"""
from __future__ import annotations

import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from oracle import ref_shim, svd_unet_ref as O, weights  # noqa: E402
from oracle.make_golden import sample, unet_inputs       # noqa: E402

OUT = ROOT / "tests" / "golden"
SALT, NU, GEGLU_GAIN, SEED = 11, 3.0, 22.0, 57


def main():
    torch.manual_seed(0)
    cfg = O.TINY
    VideoUNet, *_ = ref_shim.reference_classes()
    from sgm.modules.attention import GEGLU
    net = VideoUNet(**cfg.as_reference_kwargs()).eval()
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    net.load_state_dict(weights.synth_state_dict_heavy(shapes, SALT, NU, GEGLU_GAIN))
    T, h, w = 4, 16, 16
    x, ts, ctx, y, ioi = unet_inputs(cfg, T, h, w, SEED)
    taps, hidden_max, hooks = {}, [], []
    for name, mod in net.named_modules():
        if isinstance(mod, GEGLU):
            hooks.append(mod.register_forward_hook(lambda m, i, o: hidden_max.append(float(o.abs().max()))))
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
    assert torch.isfinite(out).all()
    torch.save({
        "config": "TINY", "T": T, "h": h, "w": w, "input_seed": SEED, "salt": SALT, "nu": NU,
        "geglu_gain": GEGLU_GAIN, "out": out,
        "tap_samples": {k: sample(v) for k, v in taps.items()},
        "tap_norms": {k: float(v.double().norm()) for k, v in taps.items()},
        "geglu_hidden_absmax": max(hidden_max), "geglu_hidden_absmax_each": hidden_max,
        "weight_absmax_over_sigma": max(float(v.abs().max() / v.std()) for k, v in net.state_dict().items()
                                        if v.ndim > 1),
    }, OUT / "unet_tiny_heavy.pt")
    print("unet_tiny_heavy: out std", float(out.std()), "absmax", float(out.abs().max()),
          "| GEGLU hidden absmax", max(hidden_max), "| per GEGLU", [f"{v:.0f}" for v in hidden_max])


if __name__ == "__main__":
    main()
