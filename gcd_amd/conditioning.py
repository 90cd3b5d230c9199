"""Camera-pose / scalar conditioning embedders of GCD: drop-ins for
sgm.modules.encoders.modules.{SphericalEmbedder, CameraEmbedder, ConcatTimestepEmbedderND}
(reference encoders/modules.py:231-287, 1000-1016) with the same parameter names (`proj.weight`,
`proj.bias`) so the GCD checkpoints' `conditioner.embedders.N.*` tensors load.

They run once per clip and feed `c['vector']` (the 128-d pose embedding is what `aux_label_emb`
injects into the UNet, video_model.py:491-497).  The projections run on libgcd_amd's fp32 small-M
kernel and the sinusoids on gcd_timestep_embedding; like the rest of gcd_amd there is no CPU path.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops


def _project_hip(feats: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    """feats [n, k] fp32 (cuda) -> feats @ weight^T + bias via gcd_linear_smallm_f32 (K padded to 4, rows in chunks of 32)."""
    n, k = feats.shape
    kp = (k + 3) // 4 * 4
    x = torch.zeros(n, kp, device=feats.device, dtype=torch.float32)
    x[:, :k] = feats
    w = torch.zeros(weight.shape[0], kp, device=feats.device, dtype=torch.float32)
    w[:, :k] = weight.detach().float()
    b = bias.detach().float().contiguous()
    out = torch.empty(n, weight.shape[0], device=feats.device, dtype=torch.float32)
    for r0 in range(0, n, 32):
        ops.linear_smallm(x[r0:r0 + 32], w, b, out[r0:r0 + 32])
    return out


class _ProjectFn(torch.autograd.Function):
    """The embedder's projection as a graph node: the reference trains SphericalEmbedder.proj (`is_trainable: True` in the
    Kubric configs; encoders/modules.py:84-208 leaves trainable embedders in the graph), so proj(feats) must hand its
    weight and bias a gradient.  Forward on the HIP kernel; the backward of a [n <= a few dozen, k <= 16] x [128, k] product
    (a few kFLOP, once per step) is two torch matmuls on the device."""

    @staticmethod
    def forward(ctx, feats, weight, bias):
        ctx.save_for_backward(feats, weight)
        return _project_hip(feats, weight, bias)

    @staticmethod
    def backward(ctx, dy):
        feats, weight = ctx.saved_tensors
        dy = dy.float()
        d_feats = dy @ weight.float() if ctx.needs_input_grad[0] else None
        d_w = (dy.t() @ feats.float()).to(weight.dtype) if ctx.needs_input_grad[1] else None
        d_b = dy.sum(0).to(weight.dtype) if ctx.needs_input_grad[2] else None
        return d_feats, d_w, d_b


def _project(feats: torch.Tensor, proj: nn.Linear) -> torch.Tensor:
    ops._need_gpu(feats, proj.weight)
    if torch.is_grad_enabled() and (proj.weight.requires_grad or feats.requires_grad):
        return _ProjectFn.apply(feats, proj.weight, proj.bias)
    return _project_hip(feats, proj.weight, proj.bias)


class AbstractEmbModel(nn.Module):
    """Attribute surface GeneralConditioner expects (encoders/modules.py:25-81)."""

    def __init__(self):
        super().__init__()
        self.is_trainable = False
        self.ucg_rate = 0.0
        self.input_key = None


def _disabled_train(self, mode=True):
    """sgm/util.py `disabled_train`: frozen embedders never leave eval mode."""
    return self


def _expand_dims_like(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    while x.dim() != y.dim():
        x = x.unsqueeze(-1)
    return x


class GeneralConditioner(nn.Module):
    """Drop-in for `sgm.modules.encoders.modules.GeneralConditioner` (encoders/modules.py:84-208): the
    `conditioner_config.target` socket of DiffusionEngine (diffusion.py:110-112).

    Runs every embedder on `batch[embedder.input_key]` and assembles the sampler's conditioning
    dict by output rank: 2-D -> `vector` (concatenated on dim 1: 3 x 256 sinusoid ids + the 128-d
    camera-pose embedding = the 896 columns `label_emb` / `aux_label_emb` consume,
    video_model.py:491-497), 3-D -> `crossattn` (dim 2), 4/5-D -> `concat` (dim 1).
    `force_zero_embeddings` zeroes whole outputs by input key (how `sample_video` builds `uc`,
    diffusion.py:522-524); `ucg_rate` drops rows at training time.  Same embedder attributes
    (`is_trainable`, `ucg_rate`, `input_key[s]`, `legacy_ucg_val`), same `embedders.N.*` parameter
    names, same errors."""

    OUTPUT_DIM2KEYS = {2: "vector", 3: "crossattn", 4: "concat", 5: "concat"}
    KEY2CATDIM = {"vector": 1, "crossattn": 2, "concat": 1}

    def __init__(self, emb_models):
        super().__init__()
        from .util import instantiate_from_config
        built = []
        for cfg in emb_models:
            emb = instantiate_from_config(cfg)
            if not isinstance(emb, AbstractEmbModel):
                raise AssertionError(f"embedder model {type(emb).__name__} has to inherit from AbstractEmbModel")
            self._configure(emb, cfg)
            built.append(emb)
        self.embedders = nn.ModuleList(built)

    @staticmethod
    def _configure(emb: AbstractEmbModel, cfg) -> None:
        """The per-embedder switches of the YAML entry (encoders/modules.py:97-131)."""
        emb.is_trainable = cfg.get("is_trainable", False)
        emb.ucg_rate = cfg.get("ucg_rate", 0.0)
        if not emb.is_trainable:                    # frozen: no gradients, pinned to eval mode
            emb.requires_grad_(False)
            emb.train = _disabled_train.__get__(emb)
            emb.eval()
        single, several = cfg.get("input_key"), cfg.get("input_keys")
        if "input_key" in cfg:
            emb.input_key = single
        elif "input_keys" in cfg:
            emb.input_key, emb.input_keys = None, several
        else:
            raise KeyError(f"need either 'input_key' or 'input_keys' for embedder {type(emb).__name__}")
        emb.legacy_ucg_val = cfg.get("legacy_ucg_value", None)
        if emb.legacy_ucg_val is not None:
            import numpy as np
            emb.ucg_prng = np.random.RandomState()

    def possibly_get_ucg_val(self, embedder, batch):
        """Legacy classifier-free dropout: overwrite entries of the INPUT with `legacy_ucg_val` at rate `ucg_rate`."""
        assert embedder.legacy_ucg_val is not None
        rate, entries = embedder.ucg_rate, batch[embedder.input_key]
        for i in range(len(entries)):
            if embedder.ucg_prng.choice(2, p=[1 - rate, rate]):
                entries[i] = embedder.legacy_ucg_val
        return batch

    def _embed(self, emb, batch):
        """One embedder's outputs as a list (frozen embedders run under no_grad)."""
        key = getattr(emb, "input_key", None)
        with torch.set_grad_enabled(bool(emb.is_trainable) and torch.is_grad_enabled()):
            if key is not None:
                if emb.legacy_ucg_val is not None:
                    batch = self.possibly_get_ucg_val(emb, batch)
                out = emb(batch[key])
            else:
                out = emb(*(batch[k] for k in emb.input_keys))
        if not isinstance(out, (torch.Tensor, list, tuple)):
            raise AssertionError(f"encoder outputs must be tensors or a sequence, but got {type(out)}")
        return list(out) if isinstance(out, (list, tuple)) else [out]

    def forward(self, batch, force_zero_embeddings=None):
        zeroed = set(force_zero_embeddings or ())
        parts = {}                                      # output key -> tensors in embedder order
        for emb in self.embedders:
            key = getattr(emb, "input_key", None)
            for t in self._embed(emb, batch):
                if emb.ucg_rate > 0.0 and emb.legacy_ucg_val is None:
                    # drop whole rows of the batch (= frames) with probability ucg_rate
                    keep = torch.bernoulli(torch.full((t.shape[0],), 1.0 - emb.ucg_rate, device=t.device))
                    t = _expand_dims_like(keep, t) * t
                if key is not None and key in zeroed:
                    t = torch.zeros_like(t)
                parts.setdefault(self.OUTPUT_DIM2KEYS[t.dim()], []).append(t)
        return {k: (v[0] if len(v) == 1 else torch.cat(v, self.KEY2CATDIM[k])) for k, v in parts.items()}

    def get_unconditional_conditioning(self, batch_c, batch_uc=None, force_uc_zero_embeddings=None,
                                       force_cond_zero_embeddings=None):
        """(c, uc) with the row dropout switched off for both passes (encoders/modules.py:190-208)."""
        saved = [emb.ucg_rate for emb in self.embedders]
        for emb in self.embedders:
            emb.ucg_rate = 0.0
        try:
            c = self(batch_c, force_cond_zero_embeddings)
            uc = self(batch_c if batch_uc is None else batch_uc, force_uc_zero_embeddings or [])
        finally:
            for emb, rate in zip(self.embedders, saved):
                emb.ucg_rate = rate
        return c, uc


class IdentityEncoder(AbstractEmbModel):
    """encoders/modules.py:290-295."""

    def encode(self, x):
        return x

    def forward(self, x):
        return x


class FrozenOpenCLIPImagePredictionEmbedder(AbstractEmbModel):
    """Frame-axis plumbing around the CLIP image tower (encoders/modules.py:1117-1136): the tower
    itself (`open_clip_embedding_config`, third-party ViT-H/14 weights) stays whatever the config
    names — SURVEY.md §8(f)-3 keeps it on PyTorch-ROCm; this class only does the
    `(b t) d -> (b s) t d` rearrange that produces the single-token `crossattn` context."""

    def __init__(self, open_clip_embedding_config, n_cond_frames: int, n_copies: int):
        super().__init__()
        from .util import instantiate_from_config
        self.n_cond_frames = n_cond_frames
        self.n_copies = n_copies
        self.open_clip = instantiate_from_config(open_clip_embedding_config)

    def forward(self, vid):
        vid = self.open_clip(vid)
        d = vid.shape[-1]
        vid = vid.reshape(-1, self.n_cond_frames, d)                     # (b t) d -> b t d
        return vid.repeat_interleave(self.n_copies, dim=0)               # b t d -> (b s) t d


class SphericalEmbedder(AbstractEmbModel):
    """(d_azimuth, d_elevation, d_radius) -> [cos, sin](az * {1,2,4}), [cos, sin](el * {1,2,4}), r
    -> Linear(13, embed_dim)   (encoders/modules.py:247-287)."""

    def __init__(self, embed_dim=128, zero_init=False):
        super().__init__()
        self.proj = nn.Linear(13, embed_dim)
        if zero_init:
            self.proj.weight.data.zero_()
            self.proj.bias.data.zero_()

    def forward(self, x):
        assert x.shape[-1] == 3
        lead = x.shape[:-1]
        x = x.reshape(-1, 3).float()
        cols = []
        for ang in (x[:, 0], x[:, 1]):
            for m in (1.0, 2.0, 4.0):
                cols += [torch.cos(ang * m), torch.sin(ang * m)]
        cols.append(x[:, 2])
        return _project(torch.stack(cols, dim=-1), self.proj).reshape(*lead, -1)


class CameraEmbedder(AbstractEmbModel):
    """Flattened 3x4 extrinsics -> Linear(12, embed_dim)   (encoders/modules.py:231-244)."""

    def __init__(self, embed_dim=128, zero_init=False):
        super().__init__()
        self.proj = nn.Linear(12, embed_dim)
        if zero_init:
            self.proj.weight.data.zero_()
            self.proj.bias.data.zero_()

    def forward(self, x):
        assert x.shape[-2:] == (3, 4)
        lead = x.shape[:-2]
        return _project(x.reshape(-1, 12).float(), self.proj).reshape(*lead, -1)


class ConcatTimestepEmbedderND(AbstractEmbModel):
    """Sinusoid(outdim) of every scalar, concatenated (encoders/modules.py:1000-1016): fps_id,
    motion_bucket_id, cond_aug -> 3 x 256 = the 768 `adm_in_channels`."""

    def __init__(self, outdim):
        super().__init__()
        self.outdim = outdim

    def forward(self, x):
        if x.ndim == 1:
            x = x[:, None]
        assert len(x.shape) == 2
        b, dims = x.shape
        ops._need_gpu(x)
        flat = x.reshape(-1).float().contiguous()
        emb = torch.empty(b * dims, self.outdim, device=x.device, dtype=torch.float32)
        ops.timestep_embedding(flat, emb, 10000.0)
        return emb.reshape(b, dims * self.outdim)


class VideoPredictionEmbedderWithEncoder(AbstractEmbModel):
    """Drop-in for `sgm.modules.encoders.modules.VideoPredictionEmbedderWithEncoder`
    (encoders/modules.py:1037-1114): the conditioning frames -> first-stage latents that become
    `cond['concat']`.  Same constructor keywords and forward; the encoder named by
    `encoder_config` is typically `gcd_amd.ae_encoder.AutoencoderKLModeOnly` (is_ae=True).  The
    optional noise augmentation (`sigma_sampler_config`, training only) is plain torch, as in the
    reference."""

    def __init__(self, n_cond_frames: int, n_copies: int, encoder_config: dict,
                 sigma_sampler_config=None, sigma_cond_config=None, is_ae: bool = False,
                 scale_factor: float = 1.0, disable_encoder_autocast: bool = False,
                 en_and_decode_n_samples_a_time=None):
        super().__init__()
        from .util import instantiate_from_config
        self.n_cond_frames = n_cond_frames
        self.n_copies = n_copies
        self.encoder = instantiate_from_config(encoder_config)
        self.sigma_sampler = instantiate_from_config(sigma_sampler_config) \
            if sigma_sampler_config is not None else None
        self.sigma_cond = instantiate_from_config(sigma_cond_config) \
            if sigma_cond_config is not None else None
        self.is_ae = is_ae
        self.scale_factor = scale_factor
        self.disable_encoder_autocast = disable_encoder_autocast
        self.en_and_decode_n_samples_a_time = en_and_decode_n_samples_a_time

    def forward(self, vid: torch.Tensor):
        import math
        from .util import append_dims
        sigma_cond = None
        if self.sigma_sampler is not None:
            b = vid.shape[0] // self.n_cond_frames
            sigmas = self.sigma_sampler(b).to(vid.device)
            if self.sigma_cond is not None:
                sigma_cond = self.sigma_cond(sigmas)
                sigma_cond = sigma_cond.repeat_interleave(self.n_copies, dim=0)     # b d -> (b t) d
            sigmas = sigmas.repeat_interleave(self.n_cond_frames)                   # b -> (b t)
            vid = vid + torch.randn_like(vid) * append_dims(sigmas, vid.ndim)
        n_samples = self.en_and_decode_n_samples_a_time \
            if self.en_and_decode_n_samples_a_time is not None else vid.shape[0]
        n_rounds = math.ceil(vid.shape[0] / n_samples)
        all_out = []
        for n in range(n_rounds):
            chunk = vid[n * n_samples:(n + 1) * n_samples]
            all_out.append(self.encoder.encode(chunk) if self.is_ae else self.encoder(chunk))
        z = torch.cat(all_out, dim=0) * self.scale_factor
        bt, c, h, w = z.shape
        t = self.n_cond_frames
        z = z.reshape(bt // t, 1, t * c, h, w)                                       # (b t) c h w -> b () (t c) h w
        z = z.expand(-1, self.n_copies, -1, -1, -1).reshape(-1, t * c, h, w)         # b 1 c h w -> (b t) c h w
        return (z, sigma_cond) if sigma_cond is not None else z
