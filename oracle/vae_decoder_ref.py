"""ORACLE (test infrastructure, NOT product code): CPU fp32 restatement of the first-stage
`VideoDecoder`, the step that follows the sampling loop.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module, and
only as the checker — gcd_amd never imports it.

A from-scratch functional restatement (plain torch fp32 ops over a flat reference-named state_dict)
of /root/reference/gcd-model/sgm/modules/autoencoding/temporal_ae.py and
sgm/modules/diffusionmodules/model.py, so that it runs on the GPU box where the reference tree does
not exist; each function cites the reference file:line it follows.  Parity pin:
tests/test_oracle_decoder.py checks it against tests/golden/decoder_tiny.pt, produced by the
reference's own `VideoDecoder` class (oracle/make_golden_decoder.py, run in the build container) —
the reference has no tests or golden vectors for this path.

Deliberately literal: every frame's attention is a full softmax(q k^T) v, the time_stack's skip and
the alpha merge are evaluated as written — so the product's fused epilogues are checked against the
un-fused math.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional, Sequence

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


@dataclass
class DecoderConfig:
    """VideoDecoder kwargs (configs/infer_kubric.yaml:150-164)."""
    ch: int = 128
    out_ch: int = 3
    ch_mult: Sequence[int] = (1, 2, 4, 4)
    num_res_blocks: int = 2
    z_channels: int = 4
    resolution: int = 256
    in_channels: int = 3
    merge_strategy: str = "learned"
    alpha: float = 0.0

    def as_reference_kwargs(self) -> dict:
        return dict(attn_type="vanilla", double_z=True, z_channels=self.z_channels,
                    resolution=self.resolution, in_channels=self.in_channels, out_ch=self.out_ch,
                    ch=self.ch, ch_mult=list(self.ch_mult), num_res_blocks=self.num_res_blocks,
                    attn_resolutions=[], dropout=0.0, video_kernel_size=[3, 1, 1],
                    alpha=self.alpha, merge_strategy=self.merge_strategy)


KUBRIC = DecoderConfig()
TINY = DecoderConfig(ch=32, resolution=64)


def _swish(x):                                   # model.py:47-49
    return x * torch.sigmoid(x)


def _gn(sd: SD, p: str, x: torch.Tensor, eps: float) -> torch.Tensor:
    return F.group_norm(x, 32, sd[p + ".weight"], sd[p + ".bias"], eps)


def _conv2d(sd: SD, p: str, x: torch.Tensor, pad: int) -> torch.Tensor:
    return F.conv2d(x, sd[p + ".weight"], sd[p + ".bias"], padding=pad)


def _resnet_block(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """ResnetBlock.forward with temb None (model.py:129-153); Normalize = GroupNorm(32, eps 1e-6)
    (model.py:52-55)."""
    h = _conv2d(sd, p + ".conv1", _swish(_gn(sd, p + ".norm1", x, 1e-6)), 1)
    h = _conv2d(sd, p + ".conv2", _swish(_gn(sd, p + ".norm2", h, 1e-6)), 1)
    if p + ".nin_shortcut.weight" in sd:
        x = _conv2d(sd, p + ".nin_shortcut", x, 0)
    return x + h


def _time_stack(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """ResBlock(dims=3, kernel (3,1,1), skip_t_emb=True, emb_channels 0).forward on (b c t h w)
    (openaimodel.py:331-357: in_layers, no embedding, out_layers, identity skip);
    GroupNorm32 = GroupNorm(32, c) at its default eps 1e-5 computed in fp32 (util.py:259-276)."""
    pad = (1, 0, 0)
    h = F.conv3d(F.silu(_gn(sd, p + ".in_layers.0", x, 1e-5)), sd[p + ".in_layers.2.weight"],
                 sd[p + ".in_layers.2.bias"], padding=pad)
    h = F.conv3d(F.silu(_gn(sd, p + ".out_layers.0", h, 1e-5)), sd[p + ".out_layers.3.weight"],
                 sd[p + ".out_layers.3.bias"], padding=pad)
    return x + h


def _video_resblock(sd: SD, p: str, x: torch.Tensor, T: int, merge_strategy: str) -> torch.Tensor:
    """VideoResBlock.forward (temporal_ae.py:63-81)."""
    x = _resnet_block(sd, p, x)
    n, c, hh, ww = x.shape
    x5 = x.reshape(n // T, T, c, hh, ww).permute(0, 2, 1, 3, 4)          # (b t) c h w -> b c t h w
    xt = _time_stack(sd, p + ".time_stack", x5)
    mix = sd[p + ".mix_factor"]
    alpha = mix if merge_strategy == "fixed" else torch.sigmoid(mix)     # temporal_ae.py:55-61
    out = alpha * xt + (1.0 - alpha) * x5
    return out.permute(0, 2, 1, 3, 4).reshape(n, c, hh, ww)


def _attn_block(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """AttnBlock.forward (model.py:180-209): one head of width c over the h*w tokens of each frame,
    softmax scale c ** -0.5 (the F.scaled_dot_product_attention default)."""
    n, c, hh, ww = x.shape
    h_ = _gn(sd, p + ".norm", x, 1e-6)
    q, k, v = (_conv2d(sd, p + "." + nm, h_, 0).reshape(n, c, hh * ww).transpose(1, 2)
               for nm in ("q", "k", "v"))
    w_ = torch.softmax(q @ k.transpose(1, 2) * (float(c) ** -0.5), dim=-1)
    o = (w_ @ v).transpose(1, 2).reshape(n, c, hh, ww)
    return x + _conv2d(sd, p + ".proj_out", o, 0)


def decoder_forward(sd: SD, cfg: DecoderConfig, z: torch.Tensor, T: int,
                    taps: Optional[dict] = None) -> torch.Tensor:
    """VideoDecoder.forward == Decoder.forward (model.py:703-748) with the conv-only video blocks:
    z (N, zc, h, w) fp32 -> (N, out_ch, 8h, 8w) for the 4-level configuration."""

    def tap(name, t):
        if taps is not None:
            taps[name] = t

    nres = len(cfg.ch_mult)
    h = _conv2d(sd, "conv_in", z, 1)
    tap("conv_in", h)
    h = _video_resblock(sd, "mid.block_1", h, T, cfg.merge_strategy)
    tap("mid.block_1", h)
    h = _attn_block(sd, "mid.attn_1", h)
    tap("mid.attn_1", h)
    h = _video_resblock(sd, "mid.block_2", h, T, cfg.merge_strategy)
    tap("mid.block_2", h)
    for i_level in reversed(range(nres)):
        for i_block in range(cfg.num_res_blocks + 1):
            h = _video_resblock(sd, f"up.{i_level}.block.{i_block}", h, T, cfg.merge_strategy)
            tap(f"up.{i_level}.block.{i_block}", h)
        if i_level != 0:
            # Upsample (model.py:68-73): nearest x2 then conv3x3
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = _conv2d(sd, f"up.{i_level}.upsample.conv", h, 1)
            tap(f"up.{i_level}.upsample", h)
    h = _swish(_gn(sd, "norm_out", h, 1e-6))
    # AE3DConv.forward (temporal_ae.py:99-107): Conv2d then Conv3d over (t, 1, 1)
    h = _conv2d(sd, "conv_out", h, 1)
    n, c, hh, ww = h.shape
    h5 = h.reshape(n // T, T, c, hh, ww).permute(0, 2, 1, 3, 4)
    h5 = F.conv3d(h5, sd["conv_out.time_mix_conv.weight"], sd["conv_out.time_mix_conv.bias"],
                  padding=(1, 0, 0))
    return h5.permute(0, 2, 1, 3, 4).reshape(n, c, hh, ww)


def decode_first_stage(sd: SD, cfg: DecoderConfig, z: torch.Tensor, scale_factor: float,
                       n_samples: Optional[int] = None) -> torch.Tensor:
    """DiffusionEngine.decode_first_stage (diffusion.py:233-251): rescale, decode in chunks of
    `en_and_decode_n_samples_a_time` frames, each chunk one clip (timesteps = chunk length)."""
    z = 1.0 / scale_factor * z
    n_samples = z.shape[0] if n_samples is None else n_samples
    outs = []
    for i in range(0, z.shape[0], n_samples):
        chunk = z[i:i + n_samples]
        outs.append(decoder_forward(sd, cfg, chunk, chunk.shape[0]))
    return torch.cat(outs, 0)
