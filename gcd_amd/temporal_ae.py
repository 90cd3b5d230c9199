"""Drop-in for the reference's first-stage decoder, `sgm.modules.autoencoding.temporal_ae.VideoDecoder`
(temporal_ae.py:293-349 on top of `Decoder`, diffusionmodules/model.py:604-748) — the step right
after the sampling loop (`DiffusionEngine.decode_first_stage`, models/diffusion.py:233-251; SURVEY.md
§8(f)-1).

Same constructor keywords, same `forward(z, timesteps=...)`, same parameter names and shapes (so the
`first_stage_model.decoder.*` keys of the published checkpoints load unchanged); the forward runs on
libgcd_amd kernels through `gcd_amd.decoder_engine.DecoderEngine`.  There is no CPU path.

Supported configuration = what GCD ships (configs/infer_kubric.yaml:150-164): `time_mode` "conv-only"
(VideoResBlock everywhere, plain single-head AttnBlock in the middle, AE3DConv at the end),
`attn_type` "vanilla" / "vanilla-xformers" (same arithmetic), `attn_resolutions: []`.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Union

import torch
import torch.nn as nn


def _zero(m: nn.Module) -> nn.Module:
    for p in m.parameters():
        nn.init.zeros_(p)
    return m


class _TimeStack(nn.Module):
    """Parameter holder of the `time_stack` ResBlock(dims=3, kernel (3,1,1), skip_t_emb)
    (openaimodel.py:213-357 as built at temporal_ae.py:32-44): GroupNorm32 -> SiLU -> Conv3d twice,
    identity skip, the last conv zero-initialised."""

    def __init__(self, channels: int, kernel_size: Sequence[int], dropout: float):
        super().__init__()
        pad = [k // 2 for k in kernel_size]
        self.in_layers = nn.Sequential(nn.GroupNorm(32, channels), nn.SiLU(),
                                       nn.Conv3d(channels, channels, tuple(kernel_size), padding=pad))
        self.out_layers = nn.Sequential(nn.GroupNorm(32, channels), nn.SiLU(), nn.Dropout(dropout),
                                        _zero(nn.Conv3d(channels, channels, tuple(kernel_size),
                                                        padding=pad)))


class VideoResBlock(nn.Module):
    """temporal_ae.py:18-81 (ResnetBlock, model.py:93-153, + time_stack + alpha merge)."""

    def __init__(self, in_channels: int, out_channels: int, dropout: float,
                 video_kernel_size: Sequence[int], alpha: float, merge_strategy: str):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.merge_strategy = merge_strategy
        if merge_strategy == "fixed":
            self.register_buffer("mix_factor", torch.Tensor([alpha]))
        elif merge_strategy == "learned":
            self.register_parameter("mix_factor", nn.Parameter(torch.Tensor([alpha])))
        else:
            raise ValueError(f"unknown merge strategy {merge_strategy}")
        self.norm1 = nn.GroupNorm(32, in_channels, eps=1e-6, affine=True)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, 1, 1)
        self.norm2 = nn.GroupNorm(32, out_channels, eps=1e-6, affine=True)
        self.dropout = nn.Dropout(dropout)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, 1, 1)
        if in_channels != out_channels:
            self.nin_shortcut = nn.Conv2d(in_channels, out_channels, 1, 1, 0)
        self.time_stack = _TimeStack(out_channels, video_kernel_size, dropout)

    def alpha(self) -> float:
        m = self.mix_factor.detach().float()
        return float(m if self.merge_strategy == "fixed" else torch.sigmoid(m))


class AttnBlock(nn.Module):
    """model.py:164-209: GroupNorm -> q, k, v 1x1 convs -> single-head SDPA over H*W -> proj_out."""

    def __init__(self, in_channels: int):
        super().__init__()
        self.in_channels = in_channels
        self.norm = nn.GroupNorm(32, in_channels, eps=1e-6, affine=True)
        self.q = nn.Conv2d(in_channels, in_channels, 1)
        self.k = nn.Conv2d(in_channels, in_channels, 1)
        self.v = nn.Conv2d(in_channels, in_channels, 1)
        self.proj_out = nn.Conv2d(in_channels, in_channels, 1)


class Upsample(nn.Module):
    """model.py:59-73: nearest x2 + Conv2d 3x3."""

    def __init__(self, in_channels: int, with_conv: bool):
        super().__init__()
        if not with_conv:
            raise NotImplementedError("gcd_amd VideoDecoder: resamp_with_conv=False")
        self.with_conv = with_conv
        self.conv = nn.Conv2d(in_channels, in_channels, 3, 1, 1)


class AE3DConv(nn.Conv2d):
    """temporal_ae.py:84-107: Conv2d followed by a Conv3d over time on its output."""

    def __init__(self, in_channels, out_channels, video_kernel_size=3, *args, **kwargs):
        super().__init__(in_channels, out_channels, *args, **kwargs)
        ks = list(video_kernel_size) if isinstance(video_kernel_size, (list, tuple)) else \
            [int(video_kernel_size)] * 3
        self.time_mix_conv = nn.Conv3d(out_channels, out_channels, tuple(ks),
                                       padding=[k // 2 for k in ks])


class VideoDecoder(nn.Module):
    available_time_modes = ["all", "conv-only", "attn-only"]

    def __init__(self, *, ch: int, out_ch: int, ch_mult=(1, 2, 4, 8), num_res_blocks: int,
                 attn_resolutions, dropout: float = 0.0, resamp_with_conv: bool = True,
                 in_channels: int, resolution: int, z_channels: int, give_pre_end: bool = False,
                 tanh_out: bool = False, use_linear_attn: bool = False, attn_type: str = "vanilla",
                 video_kernel_size: Union[int, list] = 3, alpha: float = 0.0,
                 merge_strategy: str = "learned", time_mode: str = "conv-only", **ignorekwargs):
        super().__init__()
        assert time_mode in self.available_time_modes, \
            f"time_mode parameter has to be in {self.available_time_modes}"
        if time_mode != "conv-only":
            raise NotImplementedError("gcd_amd VideoDecoder implements time_mode='conv-only' (the GCD "
                                      "configs' default)")
        if use_linear_attn or attn_type not in ("vanilla", "vanilla-xformers"):
            raise NotImplementedError(f"gcd_amd VideoDecoder: attn_type {attn_type!r}")
        if list(attn_resolutions):
            raise NotImplementedError("gcd_amd VideoDecoder: attn_resolutions must be [] (GCD configs)")
        if give_pre_end or tanh_out:
            raise NotImplementedError("gcd_amd VideoDecoder: give_pre_end / tanh_out")
        ks = list(video_kernel_size) if isinstance(video_kernel_size, (list, tuple)) else \
            [int(video_kernel_size)] * 3
        if ks != [3, 1, 1]:
            raise NotImplementedError("gcd_amd VideoDecoder: video_kernel_size must be [3, 1, 1]")
        self.ch, self.temb_ch = ch, 0
        self.num_resolutions = len(ch_mult)
        self.num_res_blocks = num_res_blocks
        self.resolution, self.in_channels = resolution, in_channels
        self.out_ch, self.z_channels = out_ch, z_channels
        self.video_kernel_size, self.alpha = ks, alpha
        self.merge_strategy, self.time_mode = merge_strategy, time_mode
        self.give_pre_end, self.tanh_out = give_pre_end, tanh_out

        def res(cin, cout):
            return VideoResBlock(cin, cout, dropout, ks, alpha, merge_strategy)

        block_in = ch * ch_mult[self.num_resolutions - 1]
        curr_res = resolution // 2 ** (self.num_resolutions - 1)
        self.z_shape = (1, z_channels, curr_res, curr_res)
        self.conv_in = nn.Conv2d(z_channels, block_in, 3, 1, 1)
        self.mid = nn.Module()
        self.mid.block_1 = res(block_in, block_in)
        self.mid.attn_1 = AttnBlock(block_in)
        self.mid.block_2 = res(block_in, block_in)
        self.up = nn.ModuleList()
        for i_level in reversed(range(self.num_resolutions)):
            block, attn = nn.ModuleList(), nn.ModuleList()
            block_out = ch * ch_mult[i_level]
            for _ in range(num_res_blocks + 1):
                block.append(res(block_in, block_out))
                block_in = block_out
            up = nn.Module()
            up.block, up.attn = block, attn
            if i_level != 0:
                up.upsample = Upsample(block_in, resamp_with_conv)
            self.up.insert(0, up)
        self.norm_out = nn.GroupNorm(32, block_in, eps=1e-6, affine=True)
        self.conv_out = AE3DConv(block_in, out_ch, video_kernel_size=ks, kernel_size=3, stride=1,
                                 padding=1)
        self._engine = None
        self.register_load_state_dict_post_hook(lambda module, incompatible: module.invalidate())

    # ---------------------------------------------------------------------------------------
    def invalidate(self) -> None:
        if self._engine is not None:
            self._engine.invalidate()

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self.invalidate()
        return out

    @property
    def engine(self):
        from .decoder_engine import DecoderEngine
        if self._engine is None:
            object.__setattr__(self, "_engine", DecoderEngine(self))
        return self._engine

    def get_last_layer(self, skip_time_mix=False, **kwargs):
        return self.conv_out.time_mix_conv.weight if not skip_time_mix else self.conv_out.weight

    def forward(self, z: torch.Tensor, timesteps: Optional[int] = None, skip_video: bool = False,
                **kwargs) -> torch.Tensor:
        """z (N, z_channels, h, w) -> (N, out_ch, 8h, 8w) for ch_mult of length 4; the N frames are
        `N // timesteps` clips of `timesteps` frames.  `DiffusionEngine.decode_first_stage` passes
        `timesteps = len(chunk)` only to decoders that pass its `isinstance(..., VideoDecoder)` test
        against the reference class (diffusion.py:242-245); called without it — as it calls this
        drop-in — the chunk is taken as one clip, i.e. the very value the reference would have
        passed, so the config-only swap needs no edit there."""
        if skip_video:
            raise NotImplementedError("gcd_amd VideoDecoder: skip_video=True")
        if timesteps is None:
            timesteps = z.shape[0]
        return self.engine.forward(z, int(timesteps))
