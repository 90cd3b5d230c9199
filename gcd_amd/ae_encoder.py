"""Drop-in for the reference's first-stage image encoder, `sgm.modules.diffusionmodules.model.Encoder`
(model.py:487-601) — the VAE half of the conditioner front-end that runs once per clip BEFORE the
sampling loop (`VideoPredictionEmbedderWithEncoder` encodes the conditioning frames
through `AutoencoderKLModeOnly`, encoders/modules.py:1071-1114, autoencoder.py:458-500,627-640).

Same constructor keywords, `forward(x) -> moments` (N, 2*z_channels, H/8, W/8 for the 4-level
configuration), parameter names and shapes (`encoder.*` in the checkpoints).  The 1x1 `quant_conv`
and the DiagonalGaussian `mode()` (= the first z_channels of the moments) that follow it stay in the
reference's own `AutoencodingEngineLegacy.encode` — `encode_mode` below restates them for callers that
do not go through that class.  The forward runs on libgcd_amd kernels (EncoderEngine): ResnetBlocks
and the single-head mid attention are the decoder's, `Downsample`'s asymmetric (0,1,0,1) padding is
the `asym_pad` geometry of gcd_gemm_f16.  There is no CPU path.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from . import _lib, ops, packing
from ._lib import GEMM_CONV3X3
from .decoder_engine import DecoderEngine
from .engine import CIN_PAD, COUT_PAD, Workspace, _f32
from .temporal_ae import AttnBlock


class ResnetBlock(nn.Module):
    """model.py:93-153 with temb_channels = 0 (parameter holder)."""

    def __init__(self, in_channels: int, out_channels: int, dropout: float):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.norm1 = nn.GroupNorm(32, in_channels, eps=1e-6, affine=True)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, 1, 1)
        self.norm2 = nn.GroupNorm(32, out_channels, eps=1e-6, affine=True)
        self.dropout = nn.Dropout(dropout)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, 1, 1)
        if in_channels != out_channels:
            self.nin_shortcut = nn.Conv2d(in_channels, out_channels, 1, 1, 0)


class Downsample(nn.Module):
    """model.py:76-91: F.pad(x, (0,1,0,1)) then Conv2d(k 3, stride 2, padding 0)."""

    def __init__(self, in_channels: int, with_conv: bool):
        super().__init__()
        if not with_conv:
            raise NotImplementedError("gcd_amd Encoder: resamp_with_conv=False (avg_pool2d)")
        self.with_conv = with_conv
        self.conv = nn.Conv2d(in_channels, in_channels, kernel_size=3, stride=2, padding=0)


class Encoder(nn.Module):
    def __init__(self, *, ch: int, out_ch: int, ch_mult=(1, 2, 4, 8), num_res_blocks: int,
                 attn_resolutions, dropout: float = 0.0, resamp_with_conv: bool = True,
                 in_channels: int, resolution: int, z_channels: int, double_z: bool = True,
                 use_linear_attn: bool = False, attn_type: str = "vanilla", **ignore_kwargs):
        super().__init__()
        if use_linear_attn or attn_type not in ("vanilla", "vanilla-xformers"):
            raise NotImplementedError(f"gcd_amd Encoder: attn_type {attn_type!r}")
        if list(attn_resolutions):
            raise NotImplementedError("gcd_amd Encoder: attn_resolutions must be [] (GCD configs)")
        if in_channels > 4:
            raise NotImplementedError("gcd_amd Encoder: in_channels <= 4 (RGB / RGBA)")
        self.ch, self.temb_ch = ch, 0
        self.num_resolutions = len(ch_mult)
        self.num_res_blocks = num_res_blocks
        self.resolution, self.in_channels = resolution, in_channels
        self.z_channels, self.double_z = z_channels, double_z
        self.conv_in = nn.Conv2d(in_channels, ch, 3, 1, 1)
        in_ch_mult = (1,) + tuple(ch_mult)
        self.in_ch_mult = in_ch_mult
        self.down = nn.ModuleList()
        block_in = ch
        for i_level in range(self.num_resolutions):
            block, attn = nn.ModuleList(), nn.ModuleList()
            block_in = ch * in_ch_mult[i_level]
            block_out = ch * ch_mult[i_level]
            for _ in range(num_res_blocks):
                block.append(ResnetBlock(block_in, block_out, dropout))
                block_in = block_out
            down = nn.Module()
            down.block, down.attn = block, attn
            if i_level != self.num_resolutions - 1:
                down.downsample = Downsample(block_in, resamp_with_conv)
            self.down.append(down)
        self.mid = nn.Module()
        self.mid.block_1 = ResnetBlock(block_in, block_in, dropout)
        self.mid.attn_1 = AttnBlock(block_in)
        self.mid.block_2 = ResnetBlock(block_in, block_in, dropout)
        self.norm_out = nn.GroupNorm(32, block_in, eps=1e-6, affine=True)
        self.conv_out = nn.Conv2d(block_in, 2 * z_channels if double_z else z_channels, 3, 1, 1)
        self._engine = None
        self.register_load_state_dict_post_hook(lambda module, incompatible: module.invalidate())

    def invalidate(self) -> None:
        if self._engine is not None:
            self._engine.invalidate()

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self.invalidate()
        return out

    @property
    def engine(self):
        if self._engine is None:
            object.__setattr__(self, "_engine", EncoderEngine(self))
        return self._engine

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.engine.forward(x)


class AutoencoderKLModeOnly(nn.Module):
    """Encode side of `sgm.models.autoencoder.AutoencoderKLModeOnly` (autoencoder.py:458-500,627-640) as
    the conditioner uses it (`VideoPredictionEmbedderWithEncoder`, is_ae=True): HIP `Encoder` ->
    1x1 `quant_conv` -> mode of the diagonal Gaussian.  Parameter names `encoder.*`, `quant_conv.*`,
    `post_quant_conv.*` as in the checkpoints; the image `decoder.*` tensors of that class are not
    needed for conditioning and are not instantiated (load with strict=False, as
    DiffusionEngine.init_from_ckpt does)."""

    def __init__(self, embed_dim: int, ddconfig: dict, **ignored):
        super().__init__()
        self.encoder = Encoder(**ddconfig)
        mult = 2 if ddconfig.get("double_z", True) else 1
        self.quant_conv = nn.Conv2d(mult * ddconfig["z_channels"], mult * embed_dim, 1)
        self.post_quant_conv = nn.Conv2d(embed_dim, ddconfig["z_channels"], 1)
        self.embed_dim = embed_dim

    def encode(self, x: torch.Tensor, return_reg_log: bool = False):
        z = encode_mode(self.encoder, x, self.quant_conv)
        return (z, dict()) if return_reg_log else z

    def forward(self, x: torch.Tensor):
        return self.encode(x)


def encode_mode(encoder: Encoder, x: torch.Tensor, quant_conv: Optional[nn.Conv2d] = None) -> torch.Tensor:
    """`AutoencoderKLModeOnly.encode` (autoencoder.py:480-500 with the `sample: False` regularizer,
    regularizers/__init__.py:13-31): moments -> optional 1x1 quant_conv -> the Gaussian's mode = the
    mean half of the channels."""
    moments = encoder(x)
    if quant_conv is not None:
        moments = _conv1x1_hip(moments, quant_conv)
    return torch.chunk(moments, 2, dim=1)[0]


def _conv1x1_hip(x: torch.Tensor, conv: nn.Conv2d) -> torch.Tensor:
    """A 1x1 convolution of an NCHW fp32 tensor on gcd_gemm_f16 (the 8 -> 8 channel `quant_conv`): channels
    zero-padded to the GEMM's 64-deep K granule / 16-row N granule, fp16 operands, fp32 accumulation and bias —
    the same arithmetic as every other contraction of the encoder.  torch only re-lays the tensor out."""
    from . import ops
    n, c, h, w = x.shape
    co = conv.out_channels
    assert tuple(conv.kernel_size) == (1, 1) and conv.in_channels == c
    kp, np_ = (c + 63) // 64 * 64, (co + 15) // 16 * 16
    M = n * h * w
    tok = torch.zeros(M, kp, dtype=torch.float16, device=x.device)
    tok[:, :c] = x.permute(0, 2, 3, 1).reshape(M, c)
    wq = torch.zeros(np_, kp, dtype=torch.float16, device=x.device)
    wq[:co, :c] = conv.weight.detach().reshape(co, c)
    b = torch.zeros(np_, dtype=torch.float32, device=x.device)
    if conv.bias is not None:
        b[:co] = conv.bias.detach().float()
    out = torch.empty(M, np_, dtype=torch.float32, device=x.device)
    ops.gemm(tok, wq, out, M=M, bias=b)
    return out[:, :co].reshape(n, h, w, co).permute(0, 3, 1, 2).contiguous()


class EncoderEngine(DecoderEngine):
    """Encoder.forward (model.py:573-601) as a sequence of libgcd_amd kernels; shares the GroupNorm,
    ResnetBlock and mid-attention code of the decoder engine."""

    def __init__(self, encoder):
        super().__init__(encoder)
        self.enc = encoder

    def pack(self) -> None:
        e = self.enc
        dev = e.conv_in.weight.device
        if dev.type != "cuda":
            raise _lib.GcdError("gcd_amd Encoder parameters are on the CPU: move the model to the GPU "
                                "(`.to('cuda')`); there is no CPU execution path")
        _lib.load()
        P = dict()
        P["conv_in_w"] = packing.pack_conv3x3(e.conv_in.weight, cin_pad=CIN_PAD)
        P["conv_in_b"] = _f32(e.conv_in.bias)
        P["down"] = []
        for lvl in e.down:
            lev = dict(blocks=[self._pack_resnet2d(b) for b in lvl.block])
            if hasattr(lvl, "downsample"):
                lev["dw"] = packing.pack_conv3x3(lvl.downsample.conv.weight)
                lev["db"] = _f32(lvl.downsample.conv.bias)
                lev["dc"] = lvl.downsample.conv.weight.shape[0]
            P["down"].append(lev)
        P["mid1"], P["mid2"] = self._pack_resnet2d(e.mid.block_1), self._pack_resnet2d(e.mid.block_2)
        P["attn"] = self._pack_attn(e.mid.attn_1)
        P["out_gn"] = (_f32(e.norm_out.weight), _f32(e.norm_out.bias))
        P["out_c"] = e.conv_out.weight.shape[1]
        cz = e.conv_out.weight.shape[0]
        P["cz"], P["cz_pad"] = cz, (cz + COUT_PAD - 1) // COUT_PAD * COUT_PAD
        P["out_w"] = packing.pack_conv3x3(e.conv_out.weight, cout_pad=P["cz_pad"])
        ob = torch.zeros(P["cz_pad"], dtype=torch.float32, device=dev)
        ob[:cz] = e.conv_out.bias.detach().float()
        P["out_b"] = ob
        self.packed = P
        if self.ws is None or self.ws.device != dev:
            self.ws = Workspace(dev)

    def _downsample(self, lev, h, st):
        ws, N = self.ws, st["N"]
        H, W = st["H"], st["W"]
        Ho, Wo = H // 2, W // 2
        Cc = lev["dc"]
        a16 = ws.alloc((N * H * W, Cc), torch.float16)
        ops.cast_f16(h, a16)
        ws.release(h)
        out = ws.alloc((N * Ho * Wo, Cc), torch.float32)
        ops.gemm(a16, lev["dw"], out, M=N * Ho * Wo, mode=GEMM_CONV3X3, bias=lev["db"],
                 conv=dict(Cin=Cc, Hi=H, Wi=W, Ho=Ho, Wo=Wo, stride=2, upsample=0, asym_pad=1))
        ws.release(a16)
        st["H"], st["W"] = Ho, Wo
        return out

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        ops._need_gpu(x)
        if self.packed is None:
            self.pack()
        P, ws, e = self.packed, self.ws, self.enc
        N, Cx, H, W = x.shape
        if Cx != e.in_channels:
            raise ValueError(f"expected {e.in_channels} image channels, got {Cx}")
        f = 1 << (e.num_resolutions - 1)
        if H % f or W % f:
            raise ValueError(f"image size {H}x{W} must be divisible by {f}")
        out_dtype = x.dtype
        x32 = x.detach().float().contiguous()
        ws.reset((N, H, W))
        st = dict(N=N, T=1, H=H, W=W)
        M = N * H * W
        xin = ws.alloc((M, CIN_PAD), torch.float16)
        ops.pack_input(x32, None, None, N, H * W, xin, CIN_PAD)
        h = ws.alloc((M, P["conv_in_w"].shape[0]), torch.float32)
        ops.gemm(xin, P["conv_in_w"], h, M=M, mode=GEMM_CONV3X3, bias=P["conv_in_b"],
                 conv=dict(Cin=CIN_PAD, Hi=H, Wi=W, Ho=H, Wo=W, stride=1, upsample=0),
                 alg_flops_scale=Cx / CIN_PAD)
        ws.release(xin)
        self._tap("conv_in", h, st)
        for li, lev in enumerate(P["down"]):
            for bi, L in enumerate(lev["blocks"]):
                h, h1 = self._resnet2d(L, h, st)
                ws.release(h1)
                self._tap(f"down.{li}.block.{bi}", h, st)
            if "dw" in lev:
                h = self._downsample(lev, h, st)
                self._tap(f"down.{li}.downsample", h, st)
        h, h1 = self._resnet2d(P["mid1"], h, st)
        ws.release(h1)
        self._tap("mid.block_1", h, st)
        h = self._attn(P["attn"], h, st)
        self._tap("mid.attn_1", h, st)
        h, h1 = self._resnet2d(P["mid2"], h, st)
        ws.release(h1)
        self._tap("mid.block_2", h, st)
        Ho, Wo = st["H"], st["W"]
        Mo = N * Ho * Wo
        a16, _ = self._gn(h, Ho * Wo, 1e-6, P["out_gn"], True)
        ws.release(h)
        tok = ws.alloc((Mo, P["cz_pad"]), torch.float32)
        ops.gemm(a16, P["out_w"], tok, M=Mo, mode=GEMM_CONV3X3, bias=P["out_b"],
                 conv=dict(Cin=P["out_c"], Hi=Ho, Wi=Wo, Ho=Ho, Wo=Wo, stride=1, upsample=0),
                 alg_flops_scale=P["cz"] / P["cz_pad"])
        ws.release(a16)
        out = torch.empty(N, P["cz"], Ho, Wo, device=x.device, dtype=torch.float32)
        ops.unpack_output(tok, out, P["cz"], N, Ho * Wo)
        ws.release(tok)
        ws.finish()
        return out if out_dtype == torch.float32 else out.to(out_dtype)
