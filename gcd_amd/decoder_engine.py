"""DecoderEngine — runs a `gcd_amd.temporal_ae.VideoDecoder` forward as a sequence of libgcd_amd
kernels (the first-stage decode that follows the sampling loop,
diffusion.py:233-251 -> temporal_ae.py:293-349 -> diffusionmodules/model.py:604-748).

Same HBM layout rules as the UNet engine: fp32 token-major residual stream [frames*H*W, C], fp16
token-major MFMA operands, fp16 [N, K] weights packed once per parameter version.

Layer mapping (every GEMM is gcd_gemm_f16):
  * conv_in 4 -> 512: implicit-GEMM 3x3 with the input channels zero-padded to 64 (gcd_pack_input);
  * VideoResBlock: GroupNorm(eps 1e-6)+SiLU -> conv3x3 -> GroupNorm+SiLU -> conv3x3 (+ 1x1
    nin_shortcut of the raw input as the residual), then the time_stack: GroupNorm32 over (T,H,W)
    +SiLU -> (3,1,1) conv, twice; the alpha merge  alpha*(x_s + f(x_s)) + (1-alpha)*x_s  is the
    epilogue  x_s + alpha*(acc + b)  of the last temporal GEMM (temporal_ae.py:69-81);
  * AttnBlock (one head of width C = 512 over H*W tokens): q|v and k projections as GEMMs, then per
    frame  S = q k^T / sqrt(C)  (GEMM, fp32 out, scale in the epilogue), row softmax -> fp16 P
    (gcd_softmax_rows_f16), V^T (gcd_transpose_f16), O = P V (GEMM), and proj_out with the residual
    in its epilogue.  The head is as wide as the model, so both contractions are plain MFMA GEMMs at
    K = 512 / K = H*W — no flash kernel needed;
  * Upsample: nearest x2 fused into the 3x3 conv's A gather (upsample = 1);
  * norm_out + SiLU -> conv_out 128 -> 3 (N padded to 16) -> time_mix_conv + NCHW unpack in one
    HBM-bound kernel (gcd_time_mix_unpack).
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib, ops, packing
from ._lib import GEMM_CONV3X3, GEMM_PLAIN, GEMM_TEMPORAL3, OUT_F16, OUT_F32
from .engine import CIN_PAD, COUT_PAD, Workspace, _f32


class DecoderEngine:
    def __init__(self, decoder):
        self.dec = decoder
        self.packed = None
        self.ws: Optional[Workspace] = None
        self.taps: Optional[dict] = None   # debug: name -> NCHW fp32 clone of block outputs

    def invalidate(self) -> None:
        self.packed = None

    # ------------------------------------------------------------------------------------------
    @staticmethod
    def _pack_resnet2d(rb) -> dict:
        """fp16 GEMM operands of a ResnetBlock (norm1, conv1, norm2, conv2, nin_shortcut)."""
        L = dict(cin=rb.in_channels, cout=rb.out_channels)
        for c in (L["cin"], L["cout"]):
            if c % 32:
                raise NotImplementedError(f"autoencoder channel count {c} is not a multiple of 32")
        L["gn1"] = (_f32(rb.norm1.weight), _f32(rb.norm1.bias))
        L["gn2"] = (_f32(rb.norm2.weight), _f32(rb.norm2.bias))
        L["w1"], L["b1"] = packing.pack_conv3x3(rb.conv1.weight), _f32(rb.conv1.bias)
        L["w2"], L["b2"] = packing.pack_conv3x3(rb.conv2.weight), _f32(rb.conv2.bias)
        if rb.in_channels != rb.out_channels:
            L["wskip"] = packing.pack_conv1x1(rb.nin_shortcut.weight)
            L["bskip"] = _f32(rb.nin_shortcut.bias)
        else:
            L["wskip"] = None
        return L

    @staticmethod
    def _pack_attn(a) -> dict:
        return dict(
            C=a.in_channels, gn=(_f32(a.norm.weight), _f32(a.norm.bias)),
            wqv=torch.cat([packing.pack_conv1x1(a.q.weight), packing.pack_conv1x1(a.v.weight)], 0).contiguous(),
            bqv=torch.cat([_f32(a.q.bias), _f32(a.v.bias)]).contiguous(),
            wk=packing.pack_conv1x1(a.k.weight), bk=_f32(a.k.bias),
            wo=packing.pack_conv1x1(a.proj_out.weight), bo=_f32(a.proj_out.bias))

    def pack(self) -> None:
        d = self.dec
        dev = d.conv_in.weight.device
        if dev.type != "cuda":
            raise _lib.GcdError("gcd_amd.VideoDecoder parameters are on the CPU: move the model to "
                                "the GPU (`.to('cuda')`); there is no CPU execution path")
        _lib.load()

        def gn(m):
            return (_f32(m.weight), _f32(m.bias))

        def res(rb):
            L = self._pack_resnet2d(rb)
            L["alpha"] = rb.alpha()
            ts = rb.time_stack
            L["tgn1"], L["tgn2"] = gn(ts.in_layers[0]), gn(ts.out_layers[0])
            L["tw1"], L["tb1"] = packing.pack_conv_t3(ts.in_layers[2].weight), _f32(ts.in_layers[2].bias)
            L["tw2"], L["tb2"] = packing.pack_conv_t3(ts.out_layers[3].weight), _f32(ts.out_layers[3].bias)
            return L

        P = dict()
        P["conv_in_w"] = packing.pack_conv3x3(d.conv_in.weight, cin_pad=CIN_PAD)
        P["conv_in_b"] = _f32(d.conv_in.bias)
        P["mid1"], P["mid2"] = res(d.mid.block_1), res(d.mid.block_2)
        P["attn"] = self._pack_attn(d.mid.attn_1)
        P["up"] = []
        for i_level in range(d.num_resolutions):
            up = d.up[i_level]
            lev = dict(blocks=[res(b) for b in up.block])
            if i_level != 0:
                lev["up_w"] = packing.pack_conv3x3(up.upsample.conv.weight)
                lev["up_b"] = _f32(up.upsample.conv.bias)
                lev["up_c"] = up.upsample.conv.weight.shape[0]
            P["up"].append(lev)
        P["out_gn"] = gn(d.norm_out)
        P["out_c"] = d.conv_out.weight.shape[1]
        if d.out_ch > 4:
            raise NotImplementedError("gcd_amd VideoDecoder: out_ch <= 4 (RGB / RGBA)")
        P["out_w"] = packing.pack_conv3x3(d.conv_out.weight, cout_pad=COUT_PAD)
        ob = torch.zeros(COUT_PAD, dtype=torch.float32, device=dev)
        ob[:d.out_ch] = d.conv_out.bias.detach().float()
        P["out_b"] = ob
        P["mix_w"] = _f32(d.conv_out.time_mix_conv.weight.reshape(d.out_ch, d.out_ch, 3))
        P["mix_b"] = _f32(d.conv_out.time_mix_conv.bias)
        self.packed = P
        if self.ws is None or self.ws.device != dev:
            self.ws = Workspace(dev)

    # ------------------------------------------------------------------------------------------
    def _gn(self, x, rows, eps, affine, silu, want_raw=False):
        ws = self.ws
        M, C = x.shape
        ninst = M // rows
        nch = ops.gn_nchunks(rows, ninst, cap=2048)
        partial = ws.alloc((ninst * nch * 64,), torch.float64)
        stats = ws.alloc((ninst * 64,), torch.float32)
        ops.groupnorm_stats(x, None, rows, eps, partial, stats, nch)
        y = ws.alloc((M, C), torch.float16)
        raw = ws.alloc((M, C), torch.float16) if want_raw else None
        ops.groupnorm_apply(x, None, rows, stats, affine[0], affine[1], silu, y, raw)
        ws.release(partial, stats)
        return y, raw

    def _resnet2d(self, L, x, st):
        """ResnetBlock.forward with temb None (model.py:129-153).  Consumes x."""
        ws, N = self.ws, st["N"]
        H, W = st["H"], st["W"]
        HW = H * W
        M = N * HW
        cin, cout = L["cin"], L["cout"]
        conv = dict(Cin=cin, Hi=H, Wi=W, Ho=H, Wo=W, stride=1, upsample=0)
        a16, raw16 = self._gn(x, HW, 1e-6, L["gn1"], True, L["wskip"] is not None)
        h1 = ws.alloc((M, cout), torch.float32)
        ops.gemm(a16, L["w1"], h1, M=M, mode=GEMM_CONV3X3, bias=L["b1"], conv=conv)
        ws.release(a16)
        a16, _ = self._gn(h1, HW, 1e-6, L["gn2"], True)
        conv2 = dict(conv, Cin=cout)
        if L["wskip"] is not None:
            xs = ws.alloc((M, cout), torch.float32)
            ops.gemm(raw16, L["wskip"], xs, M=M, bias=L["bskip"])
            ws.release(raw16, x)
            ops.gemm(a16, L["w2"], xs, M=M, mode=GEMM_CONV3X3, bias=L["b2"], r1=xs, conv=conv2)
        else:
            xs = x
            ops.gemm(a16, L["w2"], xs, M=M, mode=GEMM_CONV3X3, bias=L["b2"], r1=x, conv=conv2)
        ws.release(a16)
        return xs, h1

    def _resblock(self, L, x, st):
        """VideoResBlock.forward (temporal_ae.py:56-81): ResnetBlock (model.py:125-153, temb None)
        -> time_stack (openaimodel.py:331-357 with skip_t_emb) -> alpha merge.  Consumes x."""
        ws, N, T = self.ws, st["N"], st["T"]
        HW = st["H"] * st["W"]
        M = N * HW
        cout = L["cout"]
        xs, h1 = self._resnet2d(L, x, st)
        tconv = dict(Cin=cout, T=T, HW=HW)
        a16, _ = self._gn(xs, T * HW, 1e-5, L["tgn1"], True)
        ops.gemm(a16, L["tw1"], h1, M=M, mode=GEMM_TEMPORAL3, bias=L["tb1"], conv=tconv)
        ws.release(a16)
        a16, _ = self._gn(h1, T * HW, 1e-5, L["tgn2"], True)
        ws.release(h1)
        # alpha*(x_s + conv + b) + (1 - alpha)*x_s = x_s + alpha*(conv + b)
        ops.gemm(a16, L["tw2"], xs, M=M, mode=GEMM_TEMPORAL3, bias=L["tb2"], r1=xs,
                 s_acc=L["alpha"], conv=tconv)
        ws.release(a16)
        return xs

    def _attn(self, A, x, st):
        """AttnBlock.forward (model.py:180-209): x + proj_out(softmax(q k^T / sqrt(C)) v), one head of
        width C over the H*W tokens of each frame.  Updates x in place."""
        ws, N = self.ws, st["N"]
        S = st["H"] * st["W"]
        M = N * S
        Cc = A["C"]
        if S % 4 or S > 16384:
            raise NotImplementedError(f"VideoDecoder mid attention over {S} tokens per frame "
                                      "(needs H*W % 4 == 0 and <= 16384)")
        a16, _ = self._gn(x, S, 1e-6, A["gn"], False)
        qv = ws.alloc((M, 2 * Cc), torch.float16)
        ops.gemm(a16, A["wqv"], qv, M=M, bias=A["bqv"], out_kind=OUT_F16)
        k = ws.alloc((M, Cc), torch.float16)
        ops.gemm(a16, A["wk"], k, M=M, bias=A["bk"], out_kind=OUT_F16)
        ws.release(a16)
        o16 = ws.alloc((M, Cc), torch.float16)
        S_pad = (S + 31) // 32 * 32
        sc = ws.alloc((S, S), torch.float32)
        pr = ws.alloc((S, S_pad), torch.float16)
        vt = ws.alloc((Cc, S_pad), torch.float16)
        if S_pad != S:
            pr.zero_()
            vt.zero_()
        scale = float(Cc) ** -0.5
        for f in range(N):
            rows = slice(f * S, (f + 1) * S)
            ops.gemm(qv[rows, :Cc], k[rows], sc, M=S, s_acc=scale)
            ops.softmax_rows(sc, pr[:, :S])
            ops.transpose_f16(qv[rows, Cc:], vt[:, :S])
            ops.gemm(pr, vt, o16[rows], M=S, out_kind=OUT_F16)
        ws.release(qv, k, sc, pr, vt)
        ops.gemm(o16, A["wo"], x, M=M, bias=A["bo"], r1=x)
        ws.release(o16)
        return x

    def _upsample(self, lev, h, st):
        ws, N = self.ws, st["N"]
        H, W = st["H"], st["W"]
        Cc = lev["up_c"]
        a16 = ws.alloc((N * H * W, Cc), torch.float16)
        ops.cast_f16(h, a16)
        ws.release(h)
        out = ws.alloc((N * 4 * H * W, Cc), torch.float32)
        ops.gemm(a16, lev["up_w"], out, M=N * 4 * H * W, mode=GEMM_CONV3X3, bias=lev["up_b"],
                 conv=dict(Cin=Cc, Hi=H, Wi=W, Ho=2 * H, Wo=2 * W, stride=1, upsample=1))
        ws.release(a16)
        st["H"], st["W"] = 2 * H, 2 * W
        return out

    def _tap(self, name, h, st):
        if self.taps is not None:
            self.taps[name] = h.reshape(st["N"], st["H"], st["W"], -1).permute(0, 3, 1, 2).float().clone()

    # ------------------------------------------------------------------------------------------
    def forward(self, z: torch.Tensor, T: int) -> torch.Tensor:
        """VideoDecoder.forward / Decoder.forward (model.py:703-748): z [N, zc, h, w] -> [N, out_ch,
        h*2^(levels-1), w*2^(levels-1)], the N frames being N // T clips of T frames."""
        ops._need_gpu(z)
        if self.packed is None:
            self.pack()
        P, ws, d = self.packed, self.ws, self.dec
        N, zc, H, W = z.shape
        if zc != d.z_channels:
            raise ValueError(f"expected {d.z_channels} latent channels, got {zc}")
        if T <= 0 or N % T:
            raise ValueError(f"timesteps={T} must divide the {N} frames of the batch")
        out_dtype = z.dtype
        z32 = z.detach().float().contiguous()
        ws.reset((N, H, W, T))
        st = dict(N=N, T=T, H=H, W=W)
        M = N * H * W
        zin = ws.alloc((M, CIN_PAD), torch.float16)
        ops.pack_input(z32, None, None, N, H * W, zin, CIN_PAD)
        h = ws.alloc((M, P["conv_in_w"].shape[0]), torch.float32)
        ops.gemm(zin, P["conv_in_w"], h, M=M, mode=GEMM_CONV3X3, bias=P["conv_in_b"],
                 conv=dict(Cin=CIN_PAD, Hi=H, Wi=W, Ho=H, Wo=W, stride=1, upsample=0),
                 alg_flops_scale=zc / CIN_PAD)
        ws.release(zin)
        self._tap("conv_in", h, st)
        h = self._resblock(P["mid1"], h, st)
        self._tap("mid.block_1", h, st)
        h = self._attn(P["attn"], h, st)
        self._tap("mid.attn_1", h, st)
        h = self._resblock(P["mid2"], h, st)
        self._tap("mid.block_2", h, st)
        for i_level in reversed(range(d.num_resolutions)):
            lev = P["up"][i_level]
            for bi, L in enumerate(lev["blocks"]):
                h = self._resblock(L, h, st)
                self._tap(f"up.{i_level}.block.{bi}", h, st)
            if i_level != 0:
                h = self._upsample(lev, h, st)
                self._tap(f"up.{i_level}.upsample", h, st)
        Ho, Wo = st["H"], st["W"]
        Mo = N * Ho * Wo
        a16, _ = self._gn(h, Ho * Wo, 1e-6, P["out_gn"], True)
        ws.release(h)
        tok = ws.alloc((Mo, COUT_PAD), torch.float32)
        ops.gemm(a16, P["out_w"], tok, M=Mo, mode=GEMM_CONV3X3, bias=P["out_b"],
                 conv=dict(Cin=P["out_c"], Hi=Ho, Wi=Wo, Ho=Ho, Wo=Wo, stride=1, upsample=0),
                 alg_flops_scale=d.out_ch / COUT_PAD)
        ws.release(a16)
        out = torch.empty(N, d.out_ch, Ho, Wo, device=z.device, dtype=torch.float32)
        ops.time_mix_unpack(tok, P["mix_w"], P["mix_b"], out, d.out_ch, N, T, Ho * Wo)
        ws.release(tok)
        ws.finish()
        return out if out_dtype == torch.float32 else out.to(out_dtype)
