"""ORACLE (test infrastructure, NOT product code): CPU fp32 restatement of GCD's denoising hot path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module, and
only as the checker — gcd_amd never imports it.

This is a from-scratch functional restatement (plain torch fp32 ops over a flat reference-named
state_dict) of the algorithm in /root/reference/gcd-model/sgm, written so that it can run on the GPU
box where the reference tree does not exist.  Each function cites the reference file:line it
follows.  Parity pin: tests/test_oracle.py checks it against golden tensors produced by the
*reference modules themselves* (oracle/make_golden.py, run in the build container where
/root/reference is mounted) — the reference has no tests or golden vectors of its own for this path
(SURVEY.md §4, §8c), so those committed fixtures are the pin.

Everything here is deliberately literal: cross-attention runs the full q/k/softmax even though the
context has one key, the decoder concatenates skips, AlphaBlender is evaluated as written — so that
the product's algebraic shortcuts are checked against the un-shortcut math.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


# --------------------------------------------------------------------------------------------
# configuration (VideoUNet kwargs, video_model.py:85-120; values of configs/infer_kubric.yaml:18-40)
# --------------------------------------------------------------------------------------------
@dataclass
class UNetConfig:
    in_channels: int = 8
    model_channels: int = 320
    out_channels: int = 4
    num_res_blocks: int = 2
    attention_resolutions: Sequence[int] = (4, 2, 1)
    channel_mult: Sequence[int] = (1, 2, 4, 4)
    num_head_channels: int = 64
    context_dim: int = 1024
    adm_in_channels: int = 768
    aux_emb_dim: int = 128
    transformer_depth: int = 1
    max_ddpm_temb_period: int = 10000

    def as_reference_kwargs(self) -> dict:
        """kwargs for the reference VideoUNet (spatial_transformer_attn_type 'softmax': SURVEY §0.4)."""
        return dict(
            adm_in_channels=self.adm_in_channels, num_classes="sequential", use_checkpoint=False,
            in_channels=self.in_channels, out_channels=self.out_channels,
            model_channels=self.model_channels,
            attention_resolutions=list(self.attention_resolutions),
            num_res_blocks=self.num_res_blocks, channel_mult=list(self.channel_mult),
            num_head_channels=self.num_head_channels, use_linear_in_transformer=True,
            transformer_depth=self.transformer_depth, context_dim=self.context_dim,
            spatial_transformer_attn_type="softmax", extra_ff_mix_layer=True,
            use_spatial_context=True, merge_strategy="learned_with_images",
            video_kernel_size=[3, 1, 1], aux_emb_dim=self.aux_emb_dim, aux_zero_init=False,
            max_ddpm_temb_period=self.max_ddpm_temb_period)


KUBRIC = UNetConfig()                       # configs/infer_kubric.yaml:18-40
PARDOM = UNetConfig(aux_emb_dim=0)          # configs/infer_pardom.yaml (no aux_label_emb, y is 768)
TINY = UNetConfig(model_channels=64, context_dim=64, adm_in_channels=64, aux_emb_dim=64)


# --------------------------------------------------------------------------------------------
# topology (video_model.py:205-459): which layers each TimestepEmbedSequential holds
# --------------------------------------------------------------------------------------------
def topology(cfg: UNetConfig):
    """Returns (input_blocks, middle_block, output_blocks): lists of layer tuples
    ('conv', cin, cout) | ('res', cin, cout) | ('attn', ch) | ('down', ch) | ('up', ch)."""
    mc = cfg.model_channels
    inputs: List[List[tuple]] = [[("conv", cfg.in_channels, mc)]]
    chans = [mc]
    ch, ds = mc, 1
    nlev = len(cfg.channel_mult)
    for level, mult in enumerate(cfg.channel_mult):
        for _ in range(cfg.num_res_blocks):
            layers = [("res", ch, mult * mc)]
            ch = mult * mc
            if ds in cfg.attention_resolutions:
                layers.append(("attn", ch))
            inputs.append(layers)
            chans.append(ch)
        if level != nlev - 1:
            ds *= 2
            inputs.append([("down", ch)])
            chans.append(ch)
    middle = [("res", ch, ch), ("attn", ch), ("res", ch, ch)]
    outputs: List[List[tuple]] = []
    for level, mult in list(enumerate(cfg.channel_mult))[::-1]:
        for i in range(cfg.num_res_blocks + 1):
            ich = chans.pop()
            layers = [("res", ch + ich, mc * mult)]
            ch = mc * mult
            if ds in cfg.attention_resolutions:
                layers.append(("attn", ch))
            if level and i == cfg.num_res_blocks:
                ds //= 2
                layers.append(("up", ch))
            outputs.append(layers)
    return inputs, middle, outputs


# --------------------------------------------------------------------------------------------
# leaf ops
# --------------------------------------------------------------------------------------------
def timestep_embedding(t: torch.Tensor, dim: int, max_period: float = 10000) -> torch.Tensor:
    """diffusionmodules/util.py:207-231 (repeat_only=False, even dim)."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def _lin(sd: SD, p: str, x: torch.Tensor, bias: bool = True) -> torch.Tensor:
    return F.linear(x, sd[p + ".weight"], sd[p + ".bias"] if bias else None)


def _mlp(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """Sequential(linear, SiLU, linear): video_model.py:155-160 and friends."""
    return _lin(sd, p + ".2", F.silu(_lin(sd, p + ".0", x)))


def _gn(sd: SD, p: str, x: torch.Tensor, eps: float) -> torch.Tensor:
    """GroupNorm32 (util.py:259-276, eps 1e-5) / Normalize (attention.py:125-128, eps 1e-6)."""
    return F.group_norm(x.float(), 32, sd[p + ".weight"], sd[p + ".bias"], eps)


def _ln(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], 1e-5)


def _alpha(sd: SD, p: str, ioi: torch.Tensor, pattern_5d: bool) -> torch.Tensor:
    """AlphaBlender.get_alpha, merge_strategy 'learned_with_images' (util.py:342-356)."""
    a = torch.where(ioi.bool(), torch.ones(1, 1), torch.sigmoid(sd[p + ".mix_factor"])[..., None])
    if pattern_5d:                      # "b t -> b 1 t 1 1" (video_model.py:57)
        return a[:, None, :, None, None]
    return a.reshape(-1)[:, None, None]  # "b t -> (b t) 1 1" (util.py:316)


# --------------------------------------------------------------------------------------------
# ResBlock / VideoResBlock
# --------------------------------------------------------------------------------------------
def _resblock(sd: SD, p: str, x: torch.Tensor, emb: torch.Tensor, dims: int,
              exchange_temb_dims: bool = False) -> torch.Tensor:
    """ResBlock._forward, use_scale_shift_norm=False, no up/down (openaimodel.py:331-357)."""
    conv = F.conv2d if dims == 2 else F.conv3d
    pad = 1 if dims == 2 else (1, 0, 0)
    h = conv(F.silu(_gn(sd, p + ".in_layers.0", x, 1e-5)), sd[p + ".in_layers.2.weight"],
             sd[p + ".in_layers.2.bias"], padding=pad)
    emb_out = _lin(sd, p + ".emb_layers.1", F.silu(emb))
    while emb_out.dim() < h.dim():
        emb_out = emb_out[..., None]
    if exchange_temb_dims:               # "b t c ... -> b c t ..." (openaimodel.py:353-354)
        emb_out = emb_out.transpose(1, 2)
    h = h + emb_out
    h = conv(F.silu(_gn(sd, p + ".out_layers.0", h, 1e-5)), sd[p + ".out_layers.3.weight"],
             sd[p + ".out_layers.3.bias"], padding=pad)
    if p + ".skip_connection.weight" in sd:   # 1x1 conv when channels change (openaimodel.py:318)
        x = conv(x, sd[p + ".skip_connection.weight"], sd[p + ".skip_connection.bias"])
    return x + h


def _video_resblock(sd: SD, p: str, x: torch.Tensor, emb: torch.Tensor, T: int,
                    ioi: torch.Tensor) -> torch.Tensor:
    """VideoResBlock.forward (video_model.py:62-81)."""
    x = _resblock(sd, p, x, emb, dims=2)
    bt, c, h, w = x.shape
    x5 = x.reshape(bt // T, T, c, h, w).permute(0, 2, 1, 3, 4)             # b c t h w
    xt = _resblock(sd, p + ".time_stack", x5, emb.reshape(bt // T, T, -1), dims=3,
                   exchange_temb_dims=True)
    a = _alpha(sd, p + ".time_mixer", ioi, pattern_5d=True)
    out = a * x5 + (1.0 - a) * xt                                          # util.py:364-368
    return out.permute(0, 2, 1, 3, 4).reshape(bt, c, h, w)


# --------------------------------------------------------------------------------------------
# attention / transformer blocks
# --------------------------------------------------------------------------------------------
def _attention(sd: SD, p: str, x: torch.Tensor, context: Optional[torch.Tensor],
               heads: int) -> torch.Tensor:
    """CrossAttention.forward (attention.py:281-344): self-attention when context is None."""
    ctx = x if context is None else context
    q = F.linear(x, sd[p + ".to_q.weight"])
    k = F.linear(ctx, sd[p + ".to_k.weight"])
    v = F.linear(ctx, sd[p + ".to_v.weight"])
    b, n, _ = q.shape

    def split(t):
        return t.reshape(t.shape[0], t.shape[1], heads, -1).transpose(1, 2)

    q, k, v = split(q), split(k), split(v)
    att = torch.softmax(q @ k.transpose(-1, -2) * (q.shape[-1] ** -0.5), dim=-1)
    out = (att @ v).transpose(1, 2).reshape(b, n, -1)
    return _lin(sd, p + ".to_out.0", out)


def _feedforward(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """FeedForward with GEGLU (attention.py:87-113)."""
    a, gate = _lin(sd, p + ".net.0.proj", x).chunk(2, dim=-1)
    return _lin(sd, p + ".net.2", a * F.gelu(gate))


def _basic_block(sd: SD, p: str, x: torch.Tensor, context: torch.Tensor, heads: int) -> torch.Tensor:
    """BasicTransformerBlock._forward (attention.py:551-572)."""
    x = _attention(sd, p + ".attn1", _ln(sd, p + ".norm1", x), None, heads) + x
    x = _attention(sd, p + ".attn2", _ln(sd, p + ".norm2", x), context, heads) + x
    return _feedforward(sd, p + ".ff", _ln(sd, p + ".norm3", x)) + x


def _video_block(sd: SD, p: str, x: torch.Tensor, context: torch.Tensor, heads: int,
                 T: int) -> torch.Tensor:
    """VideoTransformerBlock._forward, ff_in=True, is_res=True (video_attention.py:109-140)."""
    bt, s, c = x.shape
    x = x.reshape(bt // T, T, s, c).transpose(1, 2).reshape(-1, T, c)       # (b s) t c
    x = _feedforward(sd, p + ".ff_in", _ln(sd, p + ".norm_in", x)) + x
    x = _attention(sd, p + ".attn1", _ln(sd, p + ".norm1", x), None, heads) + x
    x = _attention(sd, p + ".attn2", _ln(sd, p + ".norm2", x), context, heads) + x
    x = _feedforward(sd, p + ".ff", _ln(sd, p + ".norm3", x)) + x
    return x.reshape(bt // T, s, T, c).transpose(1, 2).reshape(bt, s, c)


def _spatial_video_transformer(sd: SD, p: str, x: torch.Tensor, context: torch.Tensor, T: int,
                               ioi: torch.Tensor, cfg: UNetConfig) -> torch.Tensor:
    """SpatialVideoTransformer.forward, use_linear, use_spatial_context (video_attention.py:230-301)."""
    bt, c, h, w = x.shape
    heads = c // cfg.num_head_channels
    x_in = x
    time_context = context[::T].repeat_interleave(h * w, dim=0)            # :244-253
    x = _gn(sd, p + ".norm", x, 1e-6).reshape(bt, c, h * w).transpose(1, 2)
    x = _lin(sd, p + ".proj_in", x)
    frames = torch.arange(T).repeat(bt // T)
    t_emb = timestep_embedding(frames, c, cfg.max_ddpm_temb_period)
    emb = _mlp(sd, p + ".time_pos_embed", t_emb)[:, None, :]
    for d in range(cfg.transformer_depth):
        x = _basic_block(sd, f"{p}.transformer_blocks.{d}", x, context, heads)
        x_mix = _video_block(sd, f"{p}.time_stack.{d}", x + emb, time_context, heads, T)
        a = _alpha(sd, p + ".time_mixer", ioi, pattern_5d=False)
        x = a * x + (1.0 - a) * x_mix
    x = _lin(sd, p + ".proj_out", x)
    return x.transpose(1, 2).reshape(bt, c, h, w) + x_in


# --------------------------------------------------------------------------------------------
# VideoUNet.forward
# --------------------------------------------------------------------------------------------
def unet_forward(sd: SD, cfg: UNetConfig, x: torch.Tensor, timesteps: torch.Tensor,
                 context: torch.Tensor, y: torch.Tensor, num_video_frames: int,
                 image_only_indicator: torch.Tensor,
                 taps: Optional[Dict[str, torch.Tensor]] = None) -> torch.Tensor:
    """VideoUNet.forward (video_model.py:461-540).  `taps`, if given, collects the output of every
    TimestepEmbedSequential under its reference module name."""
    T, ioi = num_video_frames, image_only_indicator
    inputs, middle, outputs = topology(cfg)
    emb = _mlp(sd, "time_embed", timestep_embedding(timesteps, cfg.model_channels))
    if cfg.aux_emb_dim == 0:
        emb = emb + _mlp(sd, "label_emb.0", y)
    else:
        assert y.shape[-1] == cfg.adm_in_channels + cfg.aux_emb_dim
        emb = emb + _mlp(sd, "label_emb.0", y[..., :cfg.adm_in_channels]) \
            + _mlp(sd, "aux_label_emb", y[..., cfg.adm_in_channels:])

    def run(prefix: str, layers, h):
        for j, layer in enumerate(layers):
            p = f"{prefix}.{j}"
            kind = layer[0]
            if kind == "conv":
                h = F.conv2d(h, sd[p + ".weight"], sd[p + ".bias"], padding=1)
            elif kind == "res":
                h = _video_resblock(sd, p, h, emb, T, ioi)
            elif kind == "attn":
                h = _spatial_video_transformer(sd, p, h, context, T, ioi, cfg)
            elif kind == "down":     # Downsample: conv 3x3 stride 2 pad 1 (openaimodel.py:199-210)
                h = F.conv2d(h, sd[p + ".op.weight"], sd[p + ".op.bias"], stride=2, padding=1)
            elif kind == "up":       # Upsample: nearest x2 then conv 3x3 (openaimodel.py:143-160)
                h = F.conv2d(F.interpolate(h, scale_factor=2, mode="nearest"),
                             sd[p + ".conv.weight"], sd[p + ".conv.bias"], padding=1)
        if taps is not None:
            taps[prefix] = h
        return h

    hs = []
    h = x.float()
    for i, layers in enumerate(inputs):
        h = run(f"input_blocks.{i}", layers, h)
        hs.append(h)
    h = run("middle_block", middle, h)
    for i, layers in enumerate(outputs):
        h = run(f"output_blocks.{i}", layers, torch.cat([h, hs.pop()], dim=1))
    h = F.silu(_gn(sd, "out.0", h, 1e-5))
    return F.conv2d(h, sd["out.2.weight"], sd["out.2.bias"], padding=1)


# --------------------------------------------------------------------------------------------
# sampler side: discretization, scalings, guider, Euler step, loop
# --------------------------------------------------------------------------------------------
def edm_sigmas(n: int, sigma_min: float = 0.002, sigma_max: float = 700.0, rho: float = 7.0) -> torch.Tensor:
    """EDMDiscretization.get_sigmas + append_zero (discretizer.py:17-39, sgm/util.py:188-189)."""
    ramp = torch.linspace(0, 1, n)
    min_inv, max_inv = sigma_min ** (1 / rho), sigma_max ** (1 / rho)
    sig = (max_inv + ramp * (min_inv - max_inv)) ** rho
    return torch.cat([sig, sig.new_zeros([1])])


def v_scaling_edm_cnoise(sigma: torch.Tensor):
    """VScalingWithEDMcNoise (denoiser_scaling.py:53-61) -> c_skip, c_out, c_in, c_noise."""
    c_skip = 1.0 / (sigma ** 2 + 1.0)
    c_out = -sigma / (sigma ** 2 + 1.0) ** 0.5
    c_in = 1.0 / (sigma ** 2 + 1.0) ** 0.5
    c_noise = 0.25 * sigma.log()
    return c_skip, c_out, c_in, c_noise


def denoise(sd: SD, cfg: UNetConfig, x: torch.Tensor, sigma: torch.Tensor, cond: dict, T: int,
            ioi: torch.Tensor) -> torch.Tensor:
    """Denoiser.forward + OpenAIWrapper.forward (denoiser.py:23-49, wrappers.py:23-34)."""
    s = sigma[:, None, None, None]
    c_skip, c_out, c_in, c_noise = v_scaling_edm_cnoise(s)
    net_in = torch.cat([x * c_in, cond["concat"]], dim=1)
    net = unet_forward(sd, cfg, net_in, c_noise.reshape(sigma.shape), cond["crossattn"],
                       cond["vector"], T, ioi)
    return net * c_out + x * c_skip


def guider_scale(T: int, max_scale: float = 1.5, min_scale: float = 1.0) -> torch.Tensor:
    """LinearPredictionGuider.scale (guiders.py:72)."""
    return torch.linspace(min_scale, max_scale, T)


def sampler_step(sd: SD, cfg: UNetConfig, x: torch.Tensor, sigma: float, sigma_next: float, c: dict,
                 uc: dict, T: int, ioi2: torch.Tensor, scale: torch.Tensor) -> torch.Tensor:
    """EDMSampler.sampler_step with gamma = 0 + EulerEDMSampler (sampling.py:101-121,225-230),
    LinearPredictionGuider (guiders.py:79-100), to_d / euler_step (sampling_utils.py:34-35,
    sampling.py:86-87)."""
    n = x.shape[0]
    s_in = x.new_ones([n])
    cc = {k: torch.cat((uc[k], c[k]), 0) for k in ("vector", "crossattn", "concat")}
    den = denoise(sd, cfg, torch.cat([x] * 2), torch.cat([s_in * sigma] * 2), cc, T, ioi2)
    x_u, x_c = den.chunk(2)
    sc = scale.repeat(n // T)[:, None, None, None]
    denoised = x_u + sc * (x_c - x_u)
    d = (x - denoised) / (s_in * sigma)[:, None, None, None]
    dt = (s_in * sigma_next - s_in * sigma)[:, None, None, None]
    return x + dt * d


def sample_loop(sd: SD, cfg: UNetConfig, noise: torch.Tensor, c: dict, uc: dict, T: int,
                num_steps: int, sigma_max: float = 700.0, max_scale: float = 1.5,
                min_scale: float = 1.0, trace: Optional[list] = None,
                ioi2: Optional[torch.Tensor] = None) -> torch.Tensor:
    """EDMSampler.__call__ (sampling.py:123-144) fed the way DiffusionEngine.sample_video does
    (diffusion.py:522-543): image_only_indicator (2B, T) = batch value after repeat_interleave(2),
    zeros unless given."""
    sigmas = edm_sigmas(num_steps, sigma_max=sigma_max)
    x = noise * torch.sqrt(1.0 + sigmas[0] ** 2.0)                         # sampling.py:54
    if ioi2 is None:
        ioi2 = torch.zeros(2 * noise.shape[0] // T, T)
    scale = guider_scale(T, max_scale, min_scale)
    for i in range(len(sigmas) - 1):
        x = sampler_step(sd, cfg, x, float(sigmas[i]), float(sigmas[i + 1]), c, uc, T, ioi2, scale)
        if trace is not None:
            trace.append(x.clone())
    return x


# --------------------------------------------------------------------------------------------
# conditioning producers (a22): camera-pose / scalar embedders
# --------------------------------------------------------------------------------------------
def spherical_embed(rel: torch.Tensor, w: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """SphericalEmbedder.forward (encoders/modules.py:255-287): rel = (d_azimuth, d_elev, d_radius)."""
    az, el, r = rel[..., 0], rel[..., 1], rel[..., 2]
    feats = []
    for ang in (az, el):
        for m in (1.0, 2.0, 4.0):
            feats += [torch.cos(ang * m), torch.sin(ang * m)]
    feats.append(r)
    return F.linear(torch.stack(feats, dim=-1), w, b)


def camera_embed(rt: torch.Tensor, w: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """CameraEmbedder.forward (encoders/modules.py:239-244): flatten 3x4 -> Linear(12, D)."""
    return F.linear(rt.reshape(*rt.shape[:-2], 12), w, b)


def concat_timestep_embed(v: torch.Tensor, outdim: int = 256) -> torch.Tensor:
    """ConcatTimestepEmbedderND.forward (encoders/modules.py:1008-1016)."""
    if v.dim() == 1:
        v = v[:, None]
    b, d = v.shape
    return timestep_embedding(v.reshape(-1), outdim).reshape(b, d * outdim)


def construct_trajectory(start, end, trajectory: str, T: int, move_time: int):
    """sgm/data/common.py:450-479: the source camera rests at `start`; the destination camera goes
    from `start` to `end` over the first `move_time` frames, then rests at `end`."""
    start = torch.as_tensor(start, dtype=torch.float32)
    end = torch.as_tensor(end, dtype=torch.float32)
    src = start[None].repeat(T, 1)
    dst = end[None].repeat(T, 1)
    for t in range(max(0, move_time)):
        if trajectory == "interpol_linear":
            a = t / move_time
        elif trajectory == "interpol_sine":
            a = (1.0 - math.cos(t / move_time * math.pi)) / 2.0
        else:
            raise ValueError(f"Unknown trajectory: {trajectory}")
        dst[t] = start * (1.0 - a) + end * a
    return src, dst


def scaled_relative_angles(az_deg: float, el_deg: float, r_m: float, T: int = 14,
                           trajectory: str = "interpol_linear", move_time: int = 13) -> torch.Tensor:
    """scripts/eval_utils.py:235-245: (dst - src) per frame, angles in radians."""
    src, dst = construct_trajectory([0.0, 0.0, 0.0], [az_deg, el_deg, r_m], trajectory, T, move_time)
    rel = dst - src
    rel[:, 0] *= math.pi / 180.0
    rel[:, 1] *= math.pi / 180.0
    return rel


def general_conditioner(embedded: Sequence[Tuple[str, torch.Tensor]],
                        force_zero: Sequence[str] = ()) -> Dict[str, torch.Tensor]:
    """GeneralConditioner.forward (encoders/modules.py:133-188) over already-embedded tensors:
    `embedded` = [(input_key, embedder output)] in embedder order.  Rank 2 -> 'vector' (cat dim 1),
    rank 3 -> 'crossattn' (cat dim 2), rank 4/5 -> 'concat' (cat dim 1); outputs whose input key is
    in `force_zero` are replaced by zeros (how sample_video builds uc, diffusion.py:522-524)."""
    dim2key = {2: "vector", 3: "crossattn", 4: "concat", 5: "concat"}
    catdim = {"vector": 1, "crossattn": 2, "concat": 1}
    out: Dict[str, torch.Tensor] = {}
    for key, emb in embedded:
        k = dim2key[emb.dim()]
        if key in force_zero:
            emb = torch.zeros_like(emb)
        out[k] = torch.cat((out[k], emb), catdim[k]) if k in out else emb
    return out
