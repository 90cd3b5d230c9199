"""Drop-in `VideoUNet` for GCD's `network_config.target` socket.

Replaces sgm.modules.diffusionmodules.video_model.VideoUNet (reference video_model.py:84-540):
same constructor keywords, same `forward(x, timesteps, context, y, time_context,
num_video_frames, image_only_indicator)` signature, and the same `state_dict()` keys / shapes
(1432 tensors for the Kubric config), so the published `.ckpt` / `svd.safetensors` load unchanged
(diffusion.py:191-219).  What differs is everything below the parameters: the module tree here only
*holds* weights; `forward` hands them to `gcd_amd.engine.UNetEngine`, which runs the whole network
as hand-written gfx950 kernels over a token-major fp32 residual stream with fp16 MFMA operands.

There is no CPU path: calling forward with CPU tensors raises.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Union

import torch
import torch.nn as nn

_SUPPORTED_MERGE = ("learned_with_images", "learned", "fixed")


def _zero(m: nn.Module) -> nn.Module:
    for p in m.parameters():
        p.detach().zero_()
    return m


class AlphaBlender(nn.Module):
    """Parameter holder for diffusionmodules/util.py:312-340 (mix_factor only)."""

    def __init__(self, alpha: float, merge_strategy: str):
        super().__init__()
        if merge_strategy not in _SUPPORTED_MERGE:
            raise ValueError(f"unknown merge strategy {merge_strategy}")
        self.merge_strategy = merge_strategy
        if merge_strategy == "fixed":
            self.register_buffer("mix_factor", torch.tensor([float(alpha)]))
        else:
            self.mix_factor = nn.Parameter(torch.tensor([float(alpha)]))


class ResBlockParams(nn.Module):
    """Parameters of openaimodel.py:213-357 ResBlock (dims 2, or dims 3 with a (3,1,1) kernel)."""

    def __init__(self, channels: int, emb_channels: int, out_channels: int, dims: int):
        super().__init__()
        self.channels, self.out_channels, self.dims = channels, out_channels, dims
        if dims == 2:
            def conv(i, o):
                return nn.Conv2d(i, o, 3, padding=1)
        else:
            def conv(i, o):
                return nn.Conv3d(i, o, (3, 1, 1), padding=(1, 0, 0))
        self.in_layers = nn.Sequential(nn.GroupNorm(32, channels), nn.SiLU(),
                                       conv(channels, out_channels))
        self.emb_layers = nn.Sequential(nn.SiLU(), nn.Linear(emb_channels, out_channels))
        self.out_layers = nn.Sequential(nn.GroupNorm(32, out_channels), nn.SiLU(), nn.Dropout(0.0),
                                        _zero(conv(out_channels, out_channels)))
        if out_channels == channels:
            self.skip_connection = nn.Identity()
        else:
            self.skip_connection = nn.Conv2d(channels, out_channels, 1) if dims == 2 else \
                nn.Conv3d(channels, out_channels, 1)


class VideoResBlock(ResBlockParams):
    """video_model.py:12-60: spatial ResBlock + `time_stack` ResBlock(dims=3) + `time_mixer`."""

    def __init__(self, channels, emb_channels, out_channels, merge_strategy, merge_factor):
        super().__init__(channels, emb_channels, out_channels, dims=2)
        self.time_stack = ResBlockParams(out_channels, emb_channels, out_channels, dims=3)
        self.time_mixer = AlphaBlender(merge_factor, merge_strategy)


class _Attention(nn.Module):
    """Parameters of attention.py:255-278 CrossAttention."""

    def __init__(self, query_dim: int, context_dim: Optional[int], heads: int, dim_head: int):
        super().__init__()
        inner = heads * dim_head
        ctx = query_dim if context_dim is None else context_dim
        self.heads = heads
        self.to_q = nn.Linear(query_dim, inner, bias=False)
        self.to_k = nn.Linear(ctx, inner, bias=False)
        self.to_v = nn.Linear(ctx, inner, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner, query_dim), nn.Dropout(0.0))


class _GEGLU(nn.Module):
    def __init__(self, dim_in: int, dim_out: int):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)


class _FeedForward(nn.Module):
    """Parameters of attention.py:98-113 FeedForward(glu=True, mult=4)."""

    def __init__(self, dim: int, dim_out: Optional[int] = None):
        super().__init__()
        inner = dim * 4
        self.net = nn.Sequential(_GEGLU(dim, inner), nn.Dropout(0.0),
                                 nn.Linear(inner, dim if dim_out is None else dim_out))


class BasicTransformerBlock(nn.Module):
    """Parameters of attention.py:456-521."""

    def __init__(self, dim: int, heads: int, d_head: int, context_dim: int):
        super().__init__()
        self.attn1 = _Attention(dim, None, heads, d_head)
        self.ff = _FeedForward(dim)
        self.attn2 = _Attention(dim, context_dim, heads, d_head)
        self.norm1 = nn.LayerNorm(dim)
        self.norm2 = nn.LayerNorm(dim)
        self.norm3 = nn.LayerNorm(dim)


class VideoTransformerBlock(nn.Module):
    """Parameters of video_attention.py:15-97 with ff_in=True, inner_dim == dim."""

    def __init__(self, dim: int, heads: int, d_head: int, context_dim: int):
        super().__init__()
        self.norm_in = nn.LayerNorm(dim)
        self.ff_in = _FeedForward(dim, dim)
        self.attn1 = _Attention(dim, None, heads, d_head)
        self.ff = _FeedForward(dim, dim)
        self.norm2 = nn.LayerNorm(dim)
        self.attn2 = _Attention(dim, context_dim, heads, d_head)
        self.norm1 = nn.LayerNorm(dim)
        self.norm3 = nn.LayerNorm(dim)


class SpatialVideoTransformer(nn.Module):
    """Parameters of video_attention.py:146-228 (+ attention.py:629-700 base class)."""

    def __init__(self, in_channels, heads, d_head, depth, context_dim, merge_strategy, merge_factor,
                 max_time_embed_period):
        super().__init__()
        inner = heads * d_head
        assert inner == in_channels, "SVD transformers keep the channel count"
        self.in_channels, self.heads, self.depth = in_channels, heads, depth
        self.max_time_embed_period = max_time_embed_period
        self.norm = nn.GroupNorm(32, in_channels, eps=1e-6)
        self.proj_in = nn.Linear(in_channels, inner)
        self.transformer_blocks = nn.ModuleList(
            [BasicTransformerBlock(inner, heads, d_head, context_dim) for _ in range(depth)])
        self.proj_out = _zero(nn.Linear(inner, in_channels))
        self.time_stack = nn.ModuleList(
            [VideoTransformerBlock(inner, heads, d_head, context_dim) for _ in range(depth)])
        self.time_pos_embed = nn.Sequential(nn.Linear(in_channels, in_channels * 4), nn.SiLU(),
                                            nn.Linear(in_channels * 4, in_channels))
        self.time_mixer = AlphaBlender(merge_factor, merge_strategy)


class Downsample(nn.Module):
    """openaimodel.py:163-210 with use_conv=True, dims=2: conv 3x3 stride 2."""

    def __init__(self, channels: int, out_channels: int):
        super().__init__()
        self.channels, self.out_channels = channels, out_channels
        self.op = nn.Conv2d(channels, out_channels, 3, stride=2, padding=1)


class Upsample(nn.Module):
    """openaimodel.py:110-160 with use_conv=True, dims=2: nearest x2 then conv 3x3."""

    def __init__(self, channels: int, out_channels: int):
        super().__init__()
        self.channels, self.out_channels = channels, out_channels
        self.conv = nn.Conv2d(channels, out_channels, 3, padding=1)


class TimestepEmbedSequential(nn.Sequential):
    """Container with the reference's name (openaimodel.py:66-107); dispatch lives in the engine."""


class VideoUNet(nn.Module):
    def __init__(
        self,
        in_channels: int,
        model_channels: int,
        out_channels: int,
        num_res_blocks: int,
        attention_resolutions: Sequence[int],
        dropout: float = 0.0,
        channel_mult: Sequence[int] = (1, 2, 4, 8),
        conv_resample: bool = True,
        dims: int = 2,
        num_classes: Optional[Union[int, str]] = None,
        use_checkpoint: bool = False,
        num_heads: int = -1,
        num_head_channels: int = -1,
        num_heads_upsample: int = -1,
        use_scale_shift_norm: bool = False,
        resblock_updown: bool = False,
        transformer_depth: Union[List[int], int] = 1,
        transformer_depth_middle: Optional[int] = None,
        context_dim: Optional[int] = None,
        time_downup: bool = False,
        time_context_dim: Optional[int] = None,
        extra_ff_mix_layer: bool = False,
        use_spatial_context: bool = False,
        merge_strategy: str = "fixed",
        merge_factor: float = 0.5,
        spatial_transformer_attn_type: str = "softmax",
        video_kernel_size: Union[int, List[int]] = 3,
        use_linear_in_transformer: bool = False,
        adm_in_channels: Optional[int] = None,
        aux_emb_dim: int = 0,
        aux_zero_init: bool = False,
        disable_temporal_crossattention: bool = False,
        max_ddpm_temb_period: int = 10000,
    ):
        super().__init__()
        assert context_dim is not None
        # -- the SVD / GCD family is what the HIP path implements; anything else is refused loudly
        unsupported = []
        if dims != 2: unsupported.append("dims != 2")
        if num_classes != "sequential": unsupported.append("num_classes != 'sequential'")
        if use_scale_shift_norm: unsupported.append("use_scale_shift_norm")
        if resblock_updown: unsupported.append("resblock_updown")
        if time_downup: unsupported.append("time_downup")
        if not conv_resample: unsupported.append("conv_resample=False")
        if not use_linear_in_transformer: unsupported.append("use_linear_in_transformer=False")
        if not extra_ff_mix_layer: unsupported.append("extra_ff_mix_layer=False")
        if not use_spatial_context: unsupported.append("use_spatial_context=False")
        if disable_temporal_crossattention: unsupported.append("disable_temporal_crossattention")
        if list(video_kernel_size if not isinstance(video_kernel_size, int) else [video_kernel_size]) \
                != [3, 1, 1]:
            unsupported.append("video_kernel_size != [3,1,1]")
        if num_head_channels != 64: unsupported.append("num_head_channels != 64")
        if dropout != 0.0: unsupported.append("dropout > 0 (inference path)")
        if spatial_transformer_attn_type not in ("softmax", "softmax-xformers"):
            unsupported.append(f"attn type {spatial_transformer_attn_type}")
        if model_channels % 64: unsupported.append("model_channels % 64 != 0")
        if unsupported:
            raise NotImplementedError(
                "gcd_amd.VideoUNet implements the SVD/GCD configuration family only; unsupported: "
                + ", ".join(unsupported))
        assert adm_in_channels is not None

        self.in_channels, self.model_channels, self.out_channels = in_channels, model_channels, out_channels
        if isinstance(transformer_depth, int):
            transformer_depth = len(channel_mult) * [transformer_depth]
        transformer_depth_middle = transformer_depth[-1] if transformer_depth_middle is None \
            else transformer_depth_middle
        self.num_res_blocks = num_res_blocks
        self.attention_resolutions = list(attention_resolutions)
        self.channel_mult = list(channel_mult)
        self.num_classes = num_classes
        self.use_checkpoint = use_checkpoint            # accepted, meaningless without autograd
        self.num_head_channels = num_head_channels
        self.aux_emb_dim = aux_emb_dim
        self.adm_in_channels = adm_in_channels
        self.context_dim = context_dim
        self.max_ddpm_temb_period = max_ddpm_temb_period

        ted = model_channels * 4

        def mlp(i):
            return nn.Sequential(nn.Linear(i, ted), nn.SiLU(), nn.Linear(ted, ted))

        self.time_embed = mlp(model_channels)
        self.label_emb = nn.Sequential(mlp(adm_in_channels))
        if aux_emb_dim > 0:
            self.aux_label_emb = mlp(aux_emb_dim)
            if aux_zero_init:
                _zero(self.aux_label_emb)

        def res(cin, cout):
            return VideoResBlock(cin, ted, cout, merge_strategy, merge_factor)

        def attn(ch, depth):
            return SpatialVideoTransformer(ch, ch // num_head_channels, num_head_channels, depth,
                                           context_dim, merge_strategy, merge_factor,
                                           max_ddpm_temb_period)

        self.input_blocks = nn.ModuleList(
            [TimestepEmbedSequential(nn.Conv2d(in_channels, model_channels, 3, padding=1))])
        chans = [model_channels]
        ch, ds = model_channels, 1
        for level, mult in enumerate(channel_mult):
            for _ in range(num_res_blocks):
                layers: List[nn.Module] = [res(ch, mult * model_channels)]
                ch = mult * model_channels
                if ds in attention_resolutions:
                    layers.append(attn(ch, transformer_depth[level]))
                self.input_blocks.append(TimestepEmbedSequential(*layers))
                chans.append(ch)
            if level != len(channel_mult) - 1:
                ds *= 2
                self.input_blocks.append(TimestepEmbedSequential(Downsample(ch, ch)))
                chans.append(ch)
        self.middle_block = TimestepEmbedSequential(res(ch, ch), attn(ch, transformer_depth_middle),
                                                    res(ch, ch))
        self.output_blocks = nn.ModuleList([])
        for level, mult in list(enumerate(channel_mult))[::-1]:
            for i in range(num_res_blocks + 1):
                ich = chans.pop()
                layers = [res(ch + ich, model_channels * mult)]
                ch = model_channels * mult
                if ds in attention_resolutions:
                    layers.append(attn(ch, transformer_depth[level]))
                if level and i == num_res_blocks:
                    ds //= 2
                    layers.append(Upsample(ch, ch))
                self.output_blocks.append(TimestepEmbedSequential(*layers))
        self.out = nn.Sequential(nn.GroupNorm(32, ch), nn.SiLU(),
                                 _zero(nn.Conv2d(model_channels, out_channels, 3, padding=1)))

        self._engine = None
        self.register_load_state_dict_post_hook(lambda module, incompatible: module.invalidate())

    # ---------------------------------------------------------------------------------------
    def invalidate(self) -> None:
        """Drop the packed fp16 weights (call after mutating parameters in place)."""
        if self._engine is not None:
            self._engine.invalidate()

    def _apply(self, fn, *args, **kwargs):   # .to() / .cuda() / .half() move the parameters
        out = super()._apply(fn, *args, **kwargs)
        self.invalidate()
        return out

    @property
    def engine(self):
        from .engine import UNetEngine
        if self._engine is None:
            object.__setattr__(self, "_engine", UNetEngine(self))
        return self._engine

    def forward(
        self,
        x: torch.Tensor,
        timesteps: torch.Tensor,
        context: Optional[torch.Tensor] = None,
        y: Optional[torch.Tensor] = None,
        time_context: Optional[torch.Tensor] = None,
        num_video_frames: Optional[int] = None,
        image_only_indicator: Optional[torch.Tensor] = None,
    ) -> torch.Tensor:
        """x (N, in_channels, H, W), timesteps (N,), context (N, 1, context_dim), y (N, adm+aux),
        image_only_indicator (N // T, T) -> (N, out_channels, H, W) in x.dtype.
        Same contract as video_model.py:461-540."""
        assert (y is not None) == (self.num_classes is not None), \
            "must specify y if and only if the model is class-conditional"
        assert y.shape[0] == x.shape[0]
        if self.aux_emb_dim > 0:
            assert y.shape[-1] == self.adm_in_channels + self.aux_emb_dim
        if time_context is not None:
            raise NotImplementedError("explicit time_context: GCD configs use use_spatial_context")
        return self.engine.forward(x, timesteps, context, y, num_video_frames, image_only_indicator)
