"""ORACLE support (build container only): generate tests/golden/encoder_tiny.pt by running the
UNMODIFIED reference `Encoder` (sgm/modules/diffusionmodules/model.py:487-601) from /root/reference on
procedural weights / images (oracle/weights.py).  The fixture pins oracle/vae_encoder_ref.py
(tests/test_oracle_encoder.py) and, through it, the HIP encoder.
Re-run with:  python -m oracle.make_golden_encoder

encoder_tiny.pt (fp32, TINY config = ch 32, otherwise the Kubric topology ch_mult [1,2,4,4],
num_res_blocks 2, double_z): one Encoder.forward on 3 images of 64x96 -> moments 3x8x8x12:
output + the output of conv_in, every ResnetBlock / Downsample / AttnBlock (strided samples + norms)
and the state_dict shapes.
"""
from __future__ import annotations

import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from oracle import ref_shim, vae_encoder_ref as E, weights  # noqa: E402

OUT = ROOT / "tests" / "golden"


def reference_encoder_class():
    ref_shim.install()
    from sgm.modules.diffusionmodules.model import Encoder
    return Encoder


def images(n: int, h: int, w: int, seed: int = 6) -> torch.Tensor:
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(n, 3, h, w, generator=g) * 2.0 - 1.0)


def sample(t: torch.Tensor, n: int = 4096) -> torch.Tensor:
    f = t.reshape(-1)
    idx = torch.linspace(0, f.numel() - 1, min(n, f.numel())).long()
    return f[idx].clone()


def main():
    torch.manual_seed(0)
    OUT.mkdir(parents=True, exist_ok=True)
    cfg = E.TINY
    enc = reference_encoder_class()(**cfg.as_reference_kwargs()).eval()
    shapes = {k: tuple(v.shape) for k, v in enc.state_dict().items()}
    enc.load_state_dict(weights.synth_state_dict(shapes, salt=2))
    n, h, w = 3, 64, 96
    x = images(n, h, w)
    taps = {}
    hooks = [enc.conv_in.register_forward_hook(lambda m, i, o: taps.__setitem__("conv_in", o.detach()))]
    for lv, down in enumerate(enc.down):
        for bi, blk in enumerate(down.block):
            hooks.append(blk.register_forward_hook(
                lambda m, i, o, nm=f"down.{lv}.block.{bi}": taps.__setitem__(nm, o.detach())))
        if hasattr(down, "downsample"):
            hooks.append(down.downsample.register_forward_hook(
                lambda m, i, o, nm=f"down.{lv}.downsample": taps.__setitem__(nm, o.detach())))
    for nm in ("block_1", "attn_1", "block_2"):
        hooks.append(getattr(enc.mid, nm).register_forward_hook(
            lambda m, i, o, k=f"mid.{nm}": taps.__setitem__(k, o.detach())))
    with torch.no_grad():
        out = enc(x)
    for hk in hooks:
        hk.remove()
    torch.save({
        "config": "TINY", "n": n, "h": h, "w": w, "input_seed": 6, "weight_salt": 2,
        "out": out, "tap_samples": {k: sample(v) for k, v in taps.items()},
        "tap_norms": {k: float(v.double().norm()) for k, v in taps.items()},
        "tap_shapes": {k: tuple(v.shape) for k, v in taps.items()},
        "state_dict_shapes": shapes,
    }, OUT / "encoder_tiny.pt")
    print("encoder_tiny: out", tuple(out.shape), "std", float(out.std()), "taps", len(taps))


if __name__ == "__main__":
    main()
