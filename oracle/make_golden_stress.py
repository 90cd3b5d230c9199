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

Two fixtures, because the two stresses turned out to be different things (measured, round 5, tools/stress_diag.py):
  unet_tiny_heavy.pt        Student-t weights, GEGLU gain 1: heavy tails alone change nothing (1.64e-3 vs 1.57e-3 for
                            Gaussian weights at this width) -> held to the forward bar, 2e-3
  unet_tiny_geglu_range.pt  Student-t weights, GEGLU gain 22 (hidden peaks at 31 710 = 48 % of the fp16 range): finite, no
                            clamp event — but the forward error is 5.2e-3, and it is 4.9e-3 already at gain 6 (hidden
                            2 308) and 4.4e-3 with Gaussian weights at gain 22: NOT a range effect.  With the FeedForward
                            output ~500x the residual it lands on, the stream IS the fp16-operand product chain (48
                            FeedForwards x ~6e-4 operand rounding each), no longer diluted by an fp32 residual -> held to
                            "finite, no overflow, < 8e-3", and recorded in DESIGN.md section 5 as the limit of the 1e-3 claim

  python -m oracle.make_golden_stress        -> both fixtures
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
import os  # noqa: E402
SALT, SEED = 11, 57
# The two committed fixtures (NU, GEGLU_GAIN, file).  GOLDEN_STRESS_NU (>= 1e6: Gaussian weights) / GOLDEN_STRESS_GAIN /
# GOLDEN_STRESS_OUT make ONE diagnostic variant instead (tools/stress_diag.py: which stress an error comes from).
FIXTURES = [(3.0, 1.0, "unet_tiny_heavy.pt"), (3.0, 22.0, "unet_tiny_geglu_range.pt")]
if "GOLDEN_STRESS_OUT" in os.environ:
    FIXTURES = [(float(os.environ.get("GOLDEN_STRESS_NU", "3")), float(os.environ.get("GOLDEN_STRESS_GAIN", "22")),
                 os.environ["GOLDEN_STRESS_OUT"])]


def main(NU, GEGLU_GAIN, FNAME):
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
    }, OUT / FNAME)
    print(FNAME, ": out std", float(out.std()), "absmax", float(out.abs().max()),
          "| GEGLU hidden absmax", max(hidden_max), "| per GEGLU", [f"{v:.0f}" for v in hidden_max])


if __name__ == "__main__":
    for fx in FIXTURES:
        main(*fx)
