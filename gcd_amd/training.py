"""The fine-tune step of GCD on libgcd_amd kernels (BASELINE.json cfg4).

Reference: `DiffusionEngine.training_step -> shared_step -> loss_fn` (diffusion.py:268-315),
`StandardDiffusionLoss._forward / get_loss` (loss.py:115-273), Lightning's backward + DDP gradient
all-reduce, `configure_optimizers` (diffusion.py:412-431).  Here:

  * `unet_forward_train`: VideoUNet.forward (video_model.py:461-540) as an autograd graph whose nodes
    are the HIP-backed operators of `gcd_amd.autograd_ops` — forward AND backward of every Linear,
    convolution, GroupNorm, LayerNorm, GEGLU and attention run on gfx950 kernels; torch provides the
    tape, the residual additions and the blends;
  * `TrainDenoiser`: Denoiser.forward + OpenAIWrapper.forward around it (denoiser.py:23-49,
    wrappers.py:23-34);
  * `StandardDiffusionLoss`: drop-in for the `loss_fn_config.target` socket: EDM sigma sampling
    harmonised per clip, noised input, denoiser call, L2 / L1, the annealed top-k "focal" loss and the
    EDM weighting.  Elementwise torch on device tensors (the reference's own code is that too); the
    ParallelDomain class re-weighting needs the RGB ground truth of a dataset that is not part of the
    path and raises NotImplementedError when requested;
  * `AdamHIP`: torch.optim.Adam semantics on `gcd_adam_step`;
  * `allreduce_gradients`: the data-parallel exchange (RCCL over xGMI; gloo in the CPU tests): flat
    fp32 buckets, asynchronous all-reduce, averaged in place.

  * `GradBucketer`: the same exchange launched from gradient hooks DURING the backward pass (what DDP does).

One step of the full-width network at cfg4's shape matches the reference's fp32 autograd to 7e-4 on the gradients and
takes 0.224 s on an MI355X; nothing here is on the measured inference path.
"""
from __future__ import annotations

import math
import os
from typing import Dict, Iterable, List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import autograd_ops as A
from . import ops
from .util import append_dims, instantiate_from_config
from .video_model import (Downsample, SpatialVideoTransformer, Upsample, VideoResBlock, VideoUNet)


# ------------------------------------------------------------------------------------------------
# token-major helpers: an activation [frames, C, H, W] is [frames*H*W, C]
# ------------------------------------------------------------------------------------------------
def _to_tokens(x: torch.Tensor) -> torch.Tensor:
    n, c, h, w = x.shape
    return x.permute(0, 2, 3, 1).reshape(n * h * w, c).contiguous()


def _from_tokens(t: torch.Tensor, n: int, h: int, w: int) -> torch.Tensor:
    return t.reshape(n, h, w, -1).permute(0, 3, 1, 2).contiguous()


def _per_frame(v: torch.Tensor, rows: int) -> torch.Tensor:
    """[frames, C] -> [frames*rows, C] (broadcast of a per-frame vector over its tokens)."""
    return v.repeat_interleave(rows, dim=0)


def _timestep_embedding(t: torch.Tensor, dim: int, max_period: float) -> torch.Tensor:
    emb = torch.empty(t.numel(), dim, device=t.device, dtype=torch.float32)
    ops.timestep_embedding(t.detach().float().contiguous(), emb, float(max_period))
    return emb


def _silu(x):
    return x * torch.sigmoid(x)


def _mlp(seq: nn.Sequential, x: torch.Tensor) -> torch.Tensor:
    """Sequential(Linear, SiLU, Linear) on a few rows (time / label / aux embeddings, time_pos_embed)."""
    return A.linear(_silu(A.linear(x, seq[0].weight, seq[0].bias)), seq[2].weight, seq[2].bias)


def _alpha(blender, ioi: torch.Tensor, frames: int) -> torch.Tensor:
    """AlphaBlender.get_alpha per frame (util.py:342-356) -> [frames, 1]."""
    if blender.merge_strategy == "fixed":
        a = blender.mix_factor.reshape(1).expand(frames)
    else:
        a = torch.sigmoid(blender.mix_factor).reshape(1).expand(frames)
        if blender.merge_strategy == "learned_with_images":
            a = torch.where(ioi.reshape(-1).bool(), torch.ones_like(a), a)
    return a[:, None]


# ------------------------------------------------------------------------------------------------
# blocks
# ------------------------------------------------------------------------------------------------
def _resblock_2d(rb, x, emb, frames, H, W):
    """ResBlock._forward, dims 2 (openaimodel.py:331-357): every GroupNorm+SiLU feeds its convolution as the
    16-bit operand it produces, the emb_layers vector and the skip ride in the convolutions' epilogues."""
    HW = H * W
    e = A.linear(_silu(emb), rb.emb_layers[1].weight, rb.emb_layers[1].bias)
    h = A.conv3x3(x, rb.in_layers[2].weight, rb.in_layers[2].bias, frames, H, W,
                  norm=("gn", rb.in_layers[0], HW, 1e-5, True), rowvec=(e, HW))
    if isinstance(rb.skip_connection, nn.Identity):
        skip = x
    else:
        skip = A.linear(x, rb.skip_connection.weight, rb.skip_connection.bias)
    return A.conv3x3(h, rb.out_layers[3].weight, rb.out_layers[3].bias, frames, H, W,
                     norm=("gn", rb.out_layers[0], HW, 1e-5, True), residual=skip)


def _resblock_time(ts, x, emb, T, HW):
    """The time_stack ResBlock, dims 3, kernel (3,1,1), GroupNorm over T*H*W per clip, per-frame emb
    (`exchange_temb_dims`, openaimodel.py:353-354)."""
    rows = T * HW
    e = A.linear(_silu(emb), ts.emb_layers[1].weight, ts.emb_layers[1].bias)
    h = A.conv_t3(x, ts.in_layers[2].weight, ts.in_layers[2].bias, T, HW,
                  norm=("gn", ts.in_layers[0], rows, 1e-5, True), rowvec=(e, HW))
    return A.conv_t3(h, ts.out_layers[3].weight, ts.out_layers[3].bias, T, HW,
                     norm=("gn", ts.out_layers[0], rows, 1e-5, True), residual=x)


def _video_resblock(rb: VideoResBlock, x, emb, frames, T, H, W, ioi):
    """VideoResBlock.forward (video_model.py:62-81)."""
    xs = _resblock_2d(rb, x, emb, frames, H, W)
    xt = _resblock_time(rb.time_stack, xs, emb, T, H * W)
    a = _per_frame(_alpha(rb.time_mixer, ioi, frames), H * W)
    return torch.lerp(xt, xs, a)            # alpha x_s + (1 - alpha) x_t   (util.py:358-369)


def _self_attention(att, x, kind, dims, norm=None, residual=None, rowvec=None):
    """attn1: q | k | v in one GEMM on the (LayerNorm'ed) input, the attention core, to_out (+ per-frame / per-clip
    vector: the one-key cross-attention that follows; + residual)."""
    qkv = A.qkv_linear(x, att.to_q.weight, att.to_k.weight, att.to_v.weight, norm=norm)
    if kind == "spatial":
        o = A.spatial_attention(qkv, *dims)
    else:
        o = A.temporal_attention(qkv, *dims)
    return A.linear(o, att.to_out[0].weight, att.to_out[0].bias, residual=residual, rowvec=rowvec)


def _cross_attention_one_key(att, ctx_rows: torch.Tensor) -> torch.Tensor:
    """attn2 with a single context token: softmax over one key is 1, so the output is
    to_out(to_v(ctx)) per frame / per clip, and to_q / to_k (and the LayerNorm in front of them)
    receive exactly zero gradient — as they do under torch.autograd in the reference."""
    return A.linear(A.linear(ctx_rows, att.to_v.weight, None), att.to_out[0].weight, att.to_out[0].bias)


_GEGLU_PROLOGUE = os.environ.get("GCD_TRAIN_GEGLU_PROLOGUE", "1") != "0"
# Which engine runs the VideoUNet of the fine-tune step: "planned" = gcd_amd/train_plan.py (round 5: no tape, one autograd
# node for the whole network, weight gradients written in place, grouped few-row Linears, one pack launch per step);
# "autograd" = unet_forward_train below (rounds 2-4: one torch.autograd node per operator).  Same kernels, same results.
TRAIN_ENGINE = "planned"


def set_train_engine(name: str) -> None:
    global TRAIN_ENGINE
    if name not in ("planned", "autograd"):
        raise ValueError(f"train engine must be 'planned' or 'autograd', got {name!r}")
    TRAIN_ENGINE = name


set_train_engine(os.environ.get("GCD_TRAIN_ENGINE", "planned"))
_ADAM_MULTI = os.environ.get("GCD_ADAM_MULTI", "1") != "0"


def _ff(ff, x, norm=None, residual=None):
    h = A.linear(x, ff.net[0].proj.weight, ff.net[0].proj.bias, norm=norm)
    if not _GEGLU_PROLOGUE:      # A/B switch: GEGLU as its own graph node with an fp32 result
        return A.linear(A.geglu(h), ff.net[2].weight, ff.net[2].bias, residual=residual)
    return A.linear(h, ff.net[2].weight, ff.net[2].bias, norm=("geglu",), residual=residual)


def _ln(m):
    return ("ln", m, 1e-5)


def _transformer(tr: SpatialVideoTransformer, x, context2d, frames, T, H, W, ioi):
    """SpatialVideoTransformer.forward (video_attention.py:230-301) on token-major rows.  Every LayerNorm is
    the prologue of the GEMM it feeds, every `+ x` the epilogue of the GEMM that produces the other addend."""
    HW = H * W
    clips = frames // T
    heads = tr.heads
    h = A.linear(x, tr.proj_in.weight, tr.proj_in.bias, norm=("gn", tr.norm, HW, 1e-6, False))
    fidx = torch.arange(T, device=x.device, dtype=torch.float32).repeat(clips)
    pos = _mlp(tr.time_pos_embed, _timestep_embedding(fidx, tr.in_channels, tr.max_time_embed_period))
    for sb, tb in zip(tr.transformer_blocks, tr.time_stack):
        # spatial BasicTransformerBlock (attention.py:551-572)
        ca = _cross_attention_one_key(sb.attn2, context2d)                  # [frames, C]
        h = _self_attention(sb.attn1, h, "spatial", (frames, HW, heads), norm=_ln(sb.norm1), residual=h,
                            rowvec=(ca, HW))
        h = _ff(sb.ff, h, norm=_ln(sb.norm3), residual=h)
        # temporal VideoTransformerBlock (video_attention.py:109-140) on x + frame position embedding
        xm = h + _per_frame(pos, HW)
        xm = _ff(tb.ff_in, xm, norm=_ln(tb.norm_in), residual=xm)
        ca = _cross_attention_one_key(tb.attn2, context2d[::T])             # [clips, C]: first frame's context
        xm = _self_attention(tb.attn1, xm, "temporal", (clips, T, HW, heads), norm=_ln(tb.norm1), residual=xm,
                             rowvec=(ca, T * HW))
        xm = _ff(tb.ff, xm, norm=_ln(tb.norm3), residual=xm)
        a = _per_frame(_alpha(tr.time_mixer, ioi, frames), HW)
        h = torch.lerp(xm, h, a)
    return A.linear(h, tr.proj_out.weight, tr.proj_out.bias, residual=x)


def unet_forward_train(unet: VideoUNet, x: torch.Tensor, timesteps: torch.Tensor, context: torch.Tensor,
                       y: torch.Tensor, num_video_frames: int, image_only_indicator: torch.Tensor,
                       use_checkpoint: Optional[bool] = None) -> torch.Tensor:
    """VideoUNet.forward (video_model.py:461-540) with gradients: x (N, 8, H, W) -> (N, 4, H, W).

    use_checkpoint (default: the network's own `use_checkpoint`, True in every GCD config): activation
    checkpointing at the reference's sites — every ResBlock (openaimodel.py:326-329) and every
    transformer (attention.py:544-546, video_attention.py:104-105; here one unit per
    SpatialVideoTransformer): only block inputs are kept, the block is re-run on HIP kernels during the
    backward pass.  Gradients equal the un-checkpointed run's up to the summation order of atomics."""
    ops._need_gpu(x, timesteps, context, y)
    A.PACK.attach(unet)          # packed operand forms are cached for registered parameters only
    T = num_video_frames
    N, _, H, W = x.shape
    assert N % T == 0 and context.dim() == 3
    if context.shape[1] != 1:
        raise NotImplementedError("gcd_amd implements the single-token (CLIP image) context of SVD / GCD")
    ioi = image_only_indicator.to(x.device)
    ctx2d = context.reshape(N, -1).float()
    emb = _mlp(unet.time_embed, _timestep_embedding(timesteps, unet.model_channels, unet.max_ddpm_temb_period
                                                    if hasattr(unet, "max_ddpm_temb_period") else 10000.0))
    adm = unet.adm_in_channels
    emb = emb + _mlp(unet.label_emb[0], y[:, :adm].float().contiguous())
    if unet.aux_emb_dim > 0:
        emb = emb + _mlp(unet.aux_label_emb, y[:, adm:].float().contiguous())

    st = dict(H=H, W=W)
    ckpt = bool(unet.use_checkpoint if use_checkpoint is None else use_checkpoint) and torch.is_grad_enabled()

    def block(fn, *args):
        if ckpt:
            from torch.utils.checkpoint import checkpoint
            return checkpoint(fn, *args, use_reentrant=False)
        return fn(*args)

    def run(seq, h):
        for m in seq:
            if isinstance(m, VideoResBlock):
                h = block(_video_resblock, m, h, emb, N, T, st["H"], st["W"], ioi)
            elif isinstance(m, SpatialVideoTransformer):
                h = block(_transformer, m, h, ctx2d, N, T, st["H"], st["W"], ioi)
            elif isinstance(m, Downsample):
                h = A.conv3x3(h, m.op.weight, m.op.bias, N, st["H"], st["W"], stride=2)
                st["H"], st["W"] = (st["H"] - 1) // 2 + 1, (st["W"] - 1) // 2 + 1
            elif isinstance(m, Upsample):
                h = A.conv3x3(h, m.conv.weight, m.conv.bias, N, st["H"], st["W"], upsample=True)
                st["H"], st["W"] = 2 * st["H"], 2 * st["W"]
            elif isinstance(m, nn.Conv2d):
                h = A.conv3x3(h, m.weight, m.bias, N, st["H"], st["W"])
            else:
                raise NotImplementedError(type(m).__name__)
        return h

    h = _to_tokens(x.float())
    hs: List[torch.Tensor] = []
    for blk in unet.input_blocks:
        h = run(blk, h)
        hs.append(h)
    h = run(unet.middle_block, h)
    for blk in unet.output_blocks:
        h = run(blk, torch.cat([h, hs.pop()], dim=1))
    h = A.conv3x3(h, unet.out[2].weight, unet.out[2].bias, N, H, W, norm=("gn", unet.out[0], H * W, 1e-5, True))
    return _from_tokens(h, N, H, W)


class TrainDenoiser(nn.Module):
    """Denoiser.forward (denoiser.py:23-49) over OpenAIWrapper.forward (wrappers.py:23-34) with the
    training-mode UNet underneath: `denoiser(network, input, sigma, cond, **kwargs)`."""

    def __init__(self, scaling_config: Dict, use_checkpoint: Optional[bool] = None, engine: Optional[str] = None):
        super().__init__()
        self.scaling = instantiate_from_config(scaling_config)
        self.use_checkpoint = use_checkpoint        # None: follow the network's `use_checkpoint`
        if engine not in (None, "planned", "autograd"):
            raise ValueError(f"train engine must be 'planned' or 'autograd', got {engine!r}")
        self.engine = engine                        # None: the process default (set_train_engine / GCD_TRAIN_ENGINE)

    def forward(self, network, input, sigma, cond, **additional_model_inputs):
        unet = getattr(network, "diffusion_model", network)
        sigma_shape = sigma.shape
        s = append_dims(sigma, input.ndim)
        c_skip, c_out, c_in, c_noise = self.scaling(s)
        x = input * c_in
        concat = cond.get("concat")
        if concat is not None and concat.numel() > 0:
            x = torch.cat((x, concat.type_as(x)), dim=1)
        if (self.engine or TRAIN_ENGINE) == "planned":
            from .train_plan import unet_forward_planned as run
        else:
            run = unet_forward_train
        out = run(unet, x, c_noise.reshape(sigma_shape), cond.get("crossattn"),
                  cond.get("vector"), additional_model_inputs["num_video_frames"],
                  additional_model_inputs["image_only_indicator"],
                  use_checkpoint=self.use_checkpoint)
        return out * c_out + input * c_skip


# ------------------------------------------------------------------------------------------------
# loss (loss.py, sigma_sampling.py, loss_weighting.py)
# ------------------------------------------------------------------------------------------------
class EDMSampling:
    """sigma = exp(p_mean + p_std * N(0, 1))   (sigma_sampling.py:6-13)."""

    def __init__(self, p_mean: float = -1.2, p_std: float = 1.2):
        self.p_mean, self.p_std = p_mean, p_std

    def __call__(self, n_samples: int, rand: Optional[torch.Tensor] = None) -> torch.Tensor:
        r = torch.randn((n_samples,)) if rand is None else rand
        return (self.p_mean + self.p_std * r).exp()


class EDMWeighting:
    """(sigma^2 + sigma_data^2) / (sigma sigma_data)^2   (loss_weighting.py:17-22)."""

    def __init__(self, sigma_data: float = 0.5):
        self.sigma_data = sigma_data

    def __call__(self, sigma: torch.Tensor) -> torch.Tensor:
        return (sigma ** 2 + self.sigma_data ** 2) / (sigma * self.sigma_data) ** 2


# ParallelDomain semantic-map palette of the classes the loss can up-weight (the dataset's colour coding, as listed at
# loss.py:16-33; "Train" is left out there because of its size).
PD_PERSON_RGB = {"Animal": (220, 20, 180), "Bicyclist": (64, 64, 64), "Motorcyclist": (128, 128, 128),
                 "OtherRider": (192, 192, 192), "Pedestrian": (220, 20, 60)}
PD_VEHICLE_RGB = {"Bus": (0, 60, 100), "Car": (0, 0, 142), "Caravan/RV": (0, 0, 90),
                  "ConstructionVehicle": (32, 32, 32), "Bicycle": (119, 11, 32), "Motorcycle": (0, 0, 230),
                  "OwnCar": (128, 230, 128), "Truck": (0, 0, 70), "WheeledSlow": (0, 64, 64)}


class StandardDiffusionLoss(nn.Module):
    """Drop-in for sgm.modules.diffusionmodules.loss.StandardDiffusionLoss (loss.py:57-273): same
    constructor keywords, `forward(network, denoiser, conditioner, input, batch)` and `_forward(network,
    denoiser, cond, input, batch)` -> per-frame loss (BT,)."""

    def __init__(self, sigma_sampler_config: dict, loss_weighting_config: dict, loss_type: str = "l2",
                 offset_noise_level: float = 0.0, harmonize_sigmas: bool = True, batch2model_keys=None,
                 pd_person_weight: float = 1.0, pd_vehicle_weight: float = 1.0, focus_top: float = 1.0,
                 focus_steps: int = -1):
        super().__init__()
        assert loss_type in ["l2", "l1", "lpips"]
        if loss_type == "lpips":
            raise NotImplementedError("loss_type 'lpips' needs the LPIPS network's torchvision weights")
        self.harmonize_sigmas = harmonize_sigmas
        self.sigma_sampler = instantiate_from_config(sigma_sampler_config)
        self.loss_weighting = instantiate_from_config(loss_weighting_config)
        self.loss_type = loss_type
        self.offset_noise_level = offset_noise_level
        if not batch2model_keys:
            batch2model_keys = []
        if isinstance(batch2model_keys, str):
            batch2model_keys = [batch2model_keys]
        self.batch2model_keys = set(batch2model_keys)
        self.pd_person_weight, self.pd_vehicle_weight = pd_person_weight, pd_vehicle_weight
        self.focus_top, self.focus_steps = focus_top, focus_steps

    def get_noised_input(self, sigmas_bc, noise, input):
        return input + noise * sigmas_bc

    def forward(self, network, denoiser, conditioner, input, batch):
        return self._forward(network, denoiser, conditioner(batch), input, batch)

    def _forward(self, network, denoiser, cond, input, batch):
        extra = {k: batch[k] for k in self.batch2model_keys.intersection(batch)}
        sigmas = self.sigma_sampler(input.shape[0]).to(input)
        if self.harmonize_sigmas:                      # one noise level per clip (loss.py:131-136)
            t = extra["num_video_frames"]
            sigmas = sigmas.reshape(-1, t)[:, 0:1].expand(-1, t).reshape(-1)
        noise = torch.randn_like(input)
        if self.offset_noise_level > 0.0:      # per (frame, channel) offsets (loss.py:141-150)
            noise = noise + self.offset_noise_level * append_dims(
                torch.randn((input.shape[0], input.shape[1]), device=input.device), input.ndim)
        noised = self.get_noised_input(append_dims(sigmas, input.ndim), noise, input)
        out = denoiser(network, noised, sigmas, cond, **extra)
        w = append_dims(self.loss_weighting(sigmas), input.ndim)
        return self.get_loss(out, input, w, batch)

    def _pd_class_weight_map(self, gt_rgb: torch.Tensor, latent_hw) -> torch.Tensor:
        """loss.py:196-230 (configs/train_pardom_semantic.yaml:145-146): per latent pixel, sum over the selected
        ParallelDomain classes of (class weight - 1) x the fraction of the pixels of its 8 x 8 square whose
        semantic-map colour matches the class (mean absolute channel difference < 0.02 on the [-1, 1] range).
        Returns (BT, 1, Hl, Wl); the reference adds `loss_raw * mask * (weight - 1)` class by class, this is the
        same sum with the class loop inside the map."""
        classes = []
        if self.pd_person_weight > 1.0:
            classes += [(rgb, self.pd_person_weight) for rgb in PD_PERSON_RGB.values()]
        if self.pd_vehicle_weight > 1.0:
            classes += [(rgb, self.pd_vehicle_weight) for rgb in PD_VEHICLE_RGB.values()]
        wmap = torch.zeros(gt_rgb.shape[0], 1, *latent_hw, dtype=torch.float32, device=gt_rgb.device)
        for rgb, weight in classes:
            colour = torch.tensor(rgb, dtype=torch.float32, device=gt_rgb.device) / 127.5 - 1.0
            match = ((gt_rgb - colour[None, :, None, None]).abs().mean(dim=1, keepdim=True) < 0.02).float()
            wmap += F.interpolate(match, tuple(latent_hw), mode="area") * (weight - 1.0)
        return wmap

    def get_loss(self, model_output, target, w, batch):
        """loss.py:163-273: L2 / L1; the ParallelDomain class re-weighting (half of it enters before the focal
        selection, half after); the annealed top-fraction focal loss (keep the `cur_top` largest per-pixel losses of
        every frame, 0.9 top + 0.1 mean); the EDM weighting."""
        diff = model_output - target
        BT = target.shape[0]
        loss_raw = diff ** 2 if self.loss_type == "l2" else diff.abs()
        if self.pd_person_weight > 1.0 or self.pd_vehicle_weight > 1.0:
            loss_bias = loss_raw * self._pd_class_weight_map(batch["jpg"].detach().to(loss_raw), target.shape[2:4])
            bias_mean = loss_bias.reshape(BT, -1).mean(dim=1)
            loss_all = loss_raw + loss_bias * 0.5
        else:
            bias_mean = 0.0
            loss_all = loss_raw
        cur_step = batch["global_step"]
        progress = min(max(cur_step / self.focus_steps, 0.0), 1.0) if self.focus_steps > 0 else 0.0
        loss_mean = loss_all.reshape(BT, -1).mean(dim=1)
        cur_top = (1.0 - progress) + self.focus_top * progress
        if cur_top < 1.0:
            flat = loss_all.reshape(BT, -1)
            keep = int(flat.shape[1] * cur_top)
            loss_focal = flat.topk(keep, dim=1)[0].mean(dim=1) * 0.9 + loss_mean * 0.1
        else:
            loss_focal = loss_mean
        return (loss_focal + bias_mean * 0.5) * w.flatten()


# ------------------------------------------------------------------------------------------------
# optimizer and the data-parallel gradient exchange
# ------------------------------------------------------------------------------------------------
class AdamHIP:
    """torch.optim.Adam (lr, betas, eps, weight_decay; no amsgrad) driven by gcd_adam_step.
    `grad_scale` undoes a static loss scale (and / or a 1 / world_size)."""

    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float = 2e-5, betas=(0.9, 0.999),
                 eps: float = 1e-8, weight_decay: float = 0.0):
        self.params = [p for p in params if p.requires_grad]
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.state = [(torch.zeros_like(p, dtype=torch.float32), torch.zeros_like(p, dtype=torch.float32))
                      for p in self.params]
        self.step_count = 0
        self._touched = {}       # id(p) -> the moments of p have been written at least once
        self._tables = None      # cached ctypes pointer tables of (params, m, v)

    def zero_grad(self):
        for p in self.params:
            p.grad = None

    @torch.no_grad()
    def step(self, grad_scale: float = 1.0):
        import ctypes as C
        from . import _lib
        self.step_count += 1
        ps, gs, ms, vs = [], [], [], []
        keep = []
        for p, (m, v) in zip(self.params, self.state):
            if p.grad is None:
                # A trainable parameter the graph never reached (attn2.to_q / to_k / norm2 behind the one-key
                # cross-attention identity): torch.autograd gives the reference exact ZERO gradients there and
                # torch.optim.Adam still steps them, which matters as soon as weight_decay != 0.  With no
                # decay and untouched moments the update is exactly zero and is skipped.
                if self.weight_decay == 0.0 and not self._touched.get(id(p), False):
                    continue
                g = torch.zeros_like(p, dtype=torch.float32)
            else:
                g = p.grad.detach().float().contiguous()
            assert p.is_contiguous() and p.dtype == torch.float32
            self._touched[id(p)] = True
            keep.append(g)
            ps.append(p.data)
            gs.append(g)
            ms.append(m)
            vs.append(v)
        n = len(ps)
        if n and not _ADAM_MULTI:          # A/B switch: one launch per tensor
            for p_, g_, m_, v_ in zip(ps, gs, ms, vs):
                A.adam_step(p_, g_, m_, v_, self.step_count, self.lr, self.betas, self.eps, self.weight_decay, grad_scale)
        elif n:
            ops._need_gpu(*ps[:1], *gs[:1])
            arr = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts])      # noqa: E731
            # parameter / moment addresses do not change between steps: their pointer tables are built once
            # keyed on the storage addresses themselves (the `p.data` wrappers above are temporaries whose ids recycle;
            # a resume that re-assigns `state` or moves a parameter must rebuild the tables)
            key = tuple((t.data_ptr(), m_.data_ptr(), v_.data_ptr()) for t, m_, v_ in zip(ps, ms, vs))
            if self._tables is None or self._tables[0] != key:
                self._tables = (key, arr(ps), arr(ms), arr(vs), (C.c_int64 * n)(*[t.numel() for t in ps]))
            _, ap, am, av, numel = self._tables
            _lib.check(_lib.load().gcd_adam_step_multi(
                n, ap, arr(gs), am, av, numel, self.lr, self.betas[0], self.betas[1], self.eps,
                self.weight_decay, self.step_count, grad_scale, torch.cuda.current_stream().cuda_stream),
                "gcd_adam_step_multi")
        # gcd_adam_step_multi writes the parameters through raw pointers, which torch's version counters do
        # not see: drop the fp16 operand forms that were packed from the old values
        A.PACK.clear()


class GradBucketer:
    """Gradient all-reduce OVERLAPPED with the backward pass — what DistributedDataParallel does for the reference
    under Lightning's DDPStrategy (main.py:826-843), built for xGMI: few, large buckets (a ring all-reduce over
    point-to-point xGMI links is per-link bound), each launched asynchronously on RCCL the moment its last
    gradient has been accumulated, so that the 6.1 GB of fp32 gradients travel while the rest of the backward pass still
    computes.

        bucketer = GradBucketer(net.parameters(), dist)        # once
        loss.backward()                                        # hooks fire; buckets launch as they fill
        bucketer.finish()                                      # wait, average, write back; then optimizer step

    Parameters are bucketed in REVERSE registration order (the order the backward pass reaches them, output blocks
    first).  A parameter the graph never reaches (attn2.to_q / to_k / norm2 behind the one-key cross-attention: three
    in every SpatialVideoTransformer block, so nearly every 256 MB bucket of the VideoUNet holds one) never fires its
    hook.  Such parameters count as ready from the start — the ones named in `unused`, plus (static_graph, what DDP
    calls it) the ones the FIRST synchronised backward pass did not reach — and enter their bucket as zeros, so every
    rank reduces the same buffers whatever its graph reached and every bucket still launches DURING the backward pass.
    If a parameter recorded as unused does fire later, its bucket is re-launched in `finish()` with the real gradient.

    Gradient accumulation (accumulate_grad_batches, main.py:950): run the first micro-batches under `no_sync()` — hooks
    are ignored, torch accumulates into p.grad — and the last one outside it: buckets pack the ACCUMULATED p.grad the
    moment the last micro-batch's hook fires.  A second synchronised backward() before finish() would reduce a stale
    first micro-batch; it raises instead.
    Results equal `allreduce_gradients` (mean over ranks); gloo world-2 test in tests/test_training_host.py."""

    def __init__(self, params: Iterable[torch.nn.Parameter], dist=None, group=None, bucket_bytes: int = 256 << 20,
                 unused: Optional[Iterable[torch.nn.Parameter]] = None, static_graph: bool = True,
                 late: Optional[Iterable[torch.nn.Parameter]] = None):
        """late: parameters whose gradient only becomes final at the very END of the backward pass — with the planned
        engine the few-row ones (`train_plan.late_parameters(net)`: every ResBlock's emb_layers, attn2.to_v / to_out, the
        embedding MLPs, time_pos_embed, the mix factors: finalised by three grouped launches after the unit walk).  They
        sit in nearly every 256 MB bucket of the registration order and would hold every all-reduce back until then; named
        here they get the LAST bucket(s) to themselves and the others launch while the backward pass still computes."""
        self.dist, self.group = dist, group
        self.active = dist is not None and dist.is_initialized() and dist.get_world_size(group) > 1
        self.world = dist.get_world_size(group) if self.active else 1
        late_ids = {id(p) for p in (late or ())}
        ordered = [p for p in params if p.requires_grad][::-1]
        self.params = [p for p in ordered if id(p) not in late_ids] + [p for p in ordered if id(p) in late_ids]
        n_early = sum(1 for p in ordered if id(p) not in late_ids)
        self.buckets: List[List[torch.nn.Parameter]] = [[]]
        size = 0
        for k, p in enumerate(self.params):
            nb = p.numel() * 4
            if self.buckets[-1] and (size + nb > bucket_bytes or (late_ids and k == n_early)):
                self.buckets.append([])
                size = 0
            self.buckets[-1].append(p)
            size += nb
        self._bucket_of = {id(p): i for i, b in enumerate(self.buckets) for p in b}
        self._flat: List[Optional[torch.Tensor]] = [None] * len(self.buckets)
        self._unused = {id(p) for p in (unused or ()) if id(p) in self._bucket_of}
        self._learn_unused = static_graph
        self._ready = [0] * len(self.buckets)
        self._seen = set()
        self._late = set()       # buckets launched without a gradient that arrived afterwards
        self._work: List[Optional[object]] = [None] * len(self.buckets)
        self._sync = True
        self.launched_during_backward = 0
        self._hooks = []
        self._reset_ready()
        if self.active:
            for p in self.params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
            # the planned engine (train_plan.py) writes .grad itself, so torch's hooks never fire for its parameters: it
            # calls the listeners below the moment a parameter's gradient is final
            from . import train_plan
            self._listener = self._on_grad
            train_plan.add_grad_listener(self.params, self._listener)

    def _reset_ready(self) -> None:
        for i, b in enumerate(self.buckets):
            self._ready[i] = sum(1 for p in b if id(p) in self._unused)

    def no_sync(self):
        """Context manager for the non-final micro-batches of gradient accumulation: hooks are ignored."""
        import contextlib

        @contextlib.contextmanager
        def ctx():
            old, self._sync = self._sync, False
            try:
                yield self
            finally:
                self._sync = old
        return ctx()

    def _launch(self, i: int) -> None:
        b = self.buckets[i]
        dev = b[0].device
        flat = self._flat[i]
        if flat is None or flat.device != dev:
            flat = self._flat[i] = torch.empty(sum(p.numel() for p in b), dtype=torch.float32, device=dev)
        off = 0
        for p in b:
            n = p.numel()
            if p.grad is None:
                flat[off:off + n].zero_()
            else:
                flat[off:off + n].copy_(p.grad.detach().reshape(-1))
            off += n
        self._work[i] = self.dist.all_reduce(flat, op=self.dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _on_grad(self, p: torch.nn.Parameter) -> None:
        if not self._sync:
            return
        i = self._bucket_of[id(p)]
        if id(p) in self._seen:
            # a second firing between two finish() calls: a shared parameter accumulating twice inside one backward
            # pass is harmless while its bucket still waits, but once the bucket is in flight the reduction has
            # already read the first value — that is a second backward() without no_sync() / finish()
            if self._work[i] is not None:
                raise RuntimeError("GradBucketer: a gradient arrived for a bucket whose all-reduce is already in "
                                   "flight — run all but the last micro-batch under bucketer.no_sync(), and call "
                                   "finish() after every synchronised backward()")
            return
        self._seen.add(id(p))
        if id(p) in self._unused:          # recorded as unreached, but this graph reached it
            self._unused.discard(id(p))
            if self._work[i] is not None:
                self._late.add(i)          # its bucket left with zeros: finish() reduces it again
            return
        self._ready[i] += 1
        if self._ready[i] == len(self.buckets[i]) and self._work[i] is None:
            self._launch(i)
            self.launched_during_backward += 1

    def finish(self) -> int:
        """Launch what the backward pass did not complete (buckets holding parameters not yet known to be unreached),
        wait for every bucket, write the averaged gradients back.  Returns the number of buckets."""
        if not self.active:
            return 0
        for i in sorted(self._late):       # same order on every rank only if the graphs agree; they do under DDP's
            self._work[i].wait()           # own contract (every rank runs the same module)
            self._launch(i)
        self._late.clear()
        for i in range(len(self.buckets)):
            if self._work[i] is None:
                self._launch(i)
        for i, b in enumerate(self.buckets):
            self._work[i].wait()
            flat = self._flat[i]
            flat.div_(self.world)
            off = 0
            for p in b:
                n = p.numel()
                g = flat[off:off + n].reshape(p.shape)
                if p.grad is None:
                    p.grad = g.clone()
                else:
                    p.grad.copy_(g)
                off += n
            self._work[i] = None
        if self._learn_unused and self._seen:
            # static graph: what this pass did not reach stays unreached.  Learned only from a pass that DID reach
            # something: a finish() behind no synchronised backward (a warm-up, a step run wholly under no_sync())
            # would otherwise mark every parameter unreached and no bucket would launch during the next backward.
            self._unused |= {id(p) for p in self.params if id(p) not in self._seen}
            self._learn_unused = False
        self._seen.clear()
        self._reset_ready()
        return len(self.buckets)

    def close(self) -> None:
        for h in self._hooks:
            h.remove()
        self._hooks = []
        if getattr(self, "_listener", None) is not None:
            from . import train_plan
            train_plan.remove_grad_listener(self.params, self._listener)
            self._listener = None


def allreduce_gradients(params: Iterable[torch.nn.Parameter], dist=None, group=None,
                        bucket_bytes: int = 256 << 20) -> int:
    """Average the gradients over the data-parallel ranks (what Lightning's DDPStrategy does for the
    reference, main.py:826-843): flat fp32 buckets of ~`bucket_bytes` (large, few: xGMI ring
    all-reduce is per-link bound), one asynchronous all-reduce each, results copied back
    divided by the world size.  Returns the number of buckets.  No-op for a single process.

    Buckets run over EVERY trainable parameter in the order given, a missing gradient entering as zeros (and
    coming back as the mean of the other ranks' — zero if nobody had one): the bucket layout is then the
    same on every rank even when ranks disagree on which parameters their graph reached, which would
    otherwise hang the collective."""
    if dist is None or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return 0
    world = dist.get_world_size(group)
    grads = []
    for p in params:
        if not p.requires_grad:
            continue
        if p.grad is None:
            p.grad = torch.zeros_like(p, dtype=torch.float32)
        grads.append(p.grad)
    buckets: List[List[torch.Tensor]] = [[]]
    size = 0
    for g in grads:
        nb = g.numel() * 4
        if buckets[-1] and size + nb > bucket_bytes:
            buckets.append([])
            size = 0
        buckets[-1].append(g)
        size += nb
    work = []
    for b in buckets:
        flat = torch.cat([g.detach().float().reshape(-1) for g in b])
        work.append((dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=True), flat, b))
    for h, flat, b in work:
        h.wait()
        off = 0
        for g in b:
            n = g.numel()
            g.copy_((flat[off:off + n] / world).reshape(g.shape))
            off += n
    return len(buckets)
