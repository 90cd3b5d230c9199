"""ORACLE support (build container only): generate tests/golden/decoder_tiny.pt by running the
UNMODIFIED reference `VideoDecoder` (sgm/modules/autoencoding/temporal_ae.py:293-349) from
/root/reference on procedural weights / latents (oracle/weights.py).  The fixture pins
oracle/vae_decoder_ref.py (tests/test_oracle_decoder.py) and, through it, the HIP decoder.
Re-run with:  python -m oracle.make_golden_decoder

decoder_tiny.pt (fp32, TINY config = ch 32, otherwise the Kubric topology ch_mult [1,2,4,4],
num_res_blocks 2, video_kernel_size [3,1,1], merge_strategy "learned"):
  one VideoDecoder.forward on N = 2 clips x T = 3 frames of 8x8 latents -> 64x64 frames:
  output + the output of conv_in, every VideoResBlock / AttnBlock / Upsample (strided samples +
  norms), the state_dict shapes, and the list of state_dict keys (checked against the product
  module's own keys).
"""
from __future__ import annotations

import sys
import types
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from oracle import ref_shim, vae_decoder_ref as D, weights  # noqa: E402

OUT = ROOT / "tests" / "golden"


def reference_decoder_class():
    ref_shim.install()
    name = "sgm.modules.autoencoding"
    if name not in sys.modules:
        m = types.ModuleType(name)
        m.__path__ = [str(ref_shim.REF / "sgm" / "modules" / "autoencoding")]
        m.__package__ = name
        sys.modules[name] = m
    from sgm.modules.autoencoding.temporal_ae import VideoDecoder
    return VideoDecoder


def decoder_latents(clips: int, T: int, h: int, w: int, zc: int = 4, seed: int = 5) -> torch.Tensor:
    g = torch.Generator().manual_seed(seed)
    return torch.randn(clips * T, zc, h, w, generator=g) * 1.5


def sample(t: torch.Tensor, n: int = 4096) -> torch.Tensor:
    f = t.reshape(-1)
    idx = torch.linspace(0, f.numel() - 1, min(n, f.numel())).long()
    return f[idx].clone()


def main():
    torch.manual_seed(0)
    OUT.mkdir(parents=True, exist_ok=True)
    cfg = D.TINY
    VideoDecoder = reference_decoder_class()
    dec = VideoDecoder(**cfg.as_reference_kwargs()).eval()
    shapes = {k: tuple(v.shape) for k, v in dec.state_dict().items()}
    dec.load_state_dict(weights.synth_state_dict(shapes, salt=1))
    clips, T, h, w = 2, 3, 8, 8
    z = decoder_latents(clips, T, h, w, cfg.z_channels)
    taps = {}
    hooks = [dec.conv_in.register_forward_hook(lambda m, i, o: taps.__setitem__("conv_in", o.detach()))]
    for nm in ("block_1", "attn_1", "block_2"):
        hooks.append(getattr(dec.mid, nm).register_forward_hook(
            lambda m, i, o, n=f"mid.{nm}": taps.__setitem__(n, o.detach())))
    for lv, up in enumerate(dec.up):
        for bi, blk in enumerate(up.block):
            hooks.append(blk.register_forward_hook(
                lambda m, i, o, n=f"up.{lv}.block.{bi}": taps.__setitem__(n, o.detach())))
        if hasattr(up, "upsample"):
            hooks.append(up.upsample.register_forward_hook(
                lambda m, i, o, n=f"up.{lv}.upsample": taps.__setitem__(n, o.detach())))
    with torch.no_grad():
        out = dec(z, timesteps=T)
    for hk in hooks:
        hk.remove()
    torch.save({
        "config": "TINY", "clips": clips, "T": T, "h": h, "w": w, "input_seed": 5, "weight_salt": 1,
        "out": out, "tap_samples": {k: sample(v) for k, v in taps.items()},
        "tap_norms": {k: float(v.double().norm()) for k, v in taps.items()},
        "tap_shapes": {k: tuple(v.shape) for k, v in taps.items()},
        "state_dict_shapes": shapes,
    }, OUT / "decoder_tiny.pt")
    print("decoder_tiny: out", tuple(out.shape), "std", float(out.std()), "taps", len(taps))


if __name__ == "__main__":
    main()
