"""ORACLE (test infrastructure, NOT product code): CPU fp32 restatement of the first-stage image
`Encoder` (the VAE half of the conditioner front-end).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module, and only
as the checker — gcd_amd never imports it.

A from-scratch functional restatement (plain torch fp32 ops over a flat reference-named state_dict) of
/root/reference/gcd-model/sgm/modules/diffusionmodules/model.py:487-601 (+ Downsample :76-91,
ResnetBlock :93-153, AttnBlock :164-209), each function citing the lines it follows.  Parity pin:
tests/test_oracle_encoder.py checks it against tests/golden/encoder_tiny.pt, produced by the
reference's own `Encoder` class (oracle/make_golden_encoder.py, run in the build container) — the
reference has no tests or golden vectors for this path.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional, Sequence

import torch
import torch.nn.functional as F

from .vae_decoder_ref import _attn_block, _conv2d, _gn, _resnet_block, _swish

SD = Dict[str, torch.Tensor]


@dataclass
class EncoderConfig:
    """Encoder kwargs (configs/infer_kubric.yaml:138-150 / the conditioner's ddconfig :89-101)."""
    ch: int = 128
    out_ch: int = 3
    ch_mult: Sequence[int] = (1, 2, 4, 4)
    num_res_blocks: int = 2
    z_channels: int = 4
    resolution: int = 256
    in_channels: int = 3
    double_z: bool = True

    def as_reference_kwargs(self) -> dict:
        return dict(attn_type="vanilla", double_z=self.double_z, z_channels=self.z_channels,
                    resolution=self.resolution, in_channels=self.in_channels, out_ch=self.out_ch,
                    ch=self.ch, ch_mult=list(self.ch_mult), num_res_blocks=self.num_res_blocks,
                    attn_resolutions=[], dropout=0.0)


KUBRIC = EncoderConfig()
TINY = EncoderConfig(ch=32, resolution=64)


def encoder_forward(sd: SD, cfg: EncoderConfig, x: torch.Tensor, taps: Optional[dict] = None) -> torch.Tensor:
    """Encoder.forward (model.py:573-601): x (N, 3, H, W) fp32 -> moments (N, 2*z, H/8, W/8)."""

    def tap(name, t):
        if taps is not None:
            taps[name] = t

    nres = len(cfg.ch_mult)
    h = _conv2d(sd, "conv_in", x, 1)
    tap("conv_in", h)
    for i_level in range(nres):
        for i_block in range(cfg.num_res_blocks):
            h = _resnet_block(sd, f"down.{i_level}.block.{i_block}", h)
            tap(f"down.{i_level}.block.{i_block}", h)
        if i_level != nres - 1:
            # Downsample (model.py:84-88): zero-pad right / bottom by one, conv k3 s2 p0
            h = F.pad(h, (0, 1, 0, 1), mode="constant", value=0)
            p = f"down.{i_level}.downsample.conv"
            h = F.conv2d(h, sd[p + ".weight"], sd[p + ".bias"], stride=2, padding=0)
            tap(f"down.{i_level}.downsample", h)
    h = _resnet_block(sd, "mid.block_1", h)
    tap("mid.block_1", h)
    h = _attn_block(sd, "mid.attn_1", h)
    tap("mid.attn_1", h)
    h = _resnet_block(sd, "mid.block_2", h)
    tap("mid.block_2", h)
    h = _swish(_gn(sd, "norm_out", h, 1e-6))
    return _conv2d(sd, "conv_out", h, 1)


def encode_mode(sd: SD, cfg: EncoderConfig, x: torch.Tensor, quant_w: Optional[torch.Tensor] = None,
                quant_b: Optional[torch.Tensor] = None) -> torch.Tensor:
    """AutoencoderKLModeOnly.encode (autoencoder.py:480-500,627-640): encoder -> 1x1 quant_conv ->
    DiagonalGaussianDistribution.mode() = the mean half (distributions.py:27,87-88)."""
    m = encoder_forward(sd, cfg, x)
    if quant_w is not None:
        m = F.conv2d(m, quant_w, quant_b)
    return torch.chunk(m, 2, dim=1)[0]
