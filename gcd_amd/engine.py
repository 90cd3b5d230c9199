"""UNetEngine — runs a `gcd_amd.video_model.VideoUNet` forward as a sequence of libgcd_amd kernels.

Data layout in HBM (all row-major, leading dimension explicit):
  * residual stream / skip tensors: fp32 token-major [frames*H*W, C];
  * MFMA operands (normalised activations, q|k|v, attention output, FF hidden): fp16 token-major;
  * weights: fp16 [N, K] packed once per parameter version (`pack()`), biases / norm affine fp32;
  * per-frame vectors (timestep-embedding projections, collapsed cross-attention, frame-position
    embeddings, blend alphas): small fp32 matrices indexed by frame or clip in the GEMM epilogues.

Algebra used (all exact; checked against the un-shortcut oracle):
  * cross-attention has one key (context is (N, 1, D)): softmax == 1, so attn2(x) = to_out(to_v(ctx))
    is a per-frame (spatial) / per-clip (temporal) vector added in the to_out epilogue of attn1;
  * AlphaBlender: a*x_s + (1-a)*x_t is folded into the epilogue of the GEMM that produces x_t;
  * torch.cat([h, skip]) of the decoder is read as a virtual concat by GroupNorm.

Workspace: a deterministic first-fit slab pool (all sizes fixed by the input shape), so after one
warm-up call every buffer address is stable and the whole forward can be replayed from a hipGraph.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Tuple

import torch

from . import _lib, ops, packing
from ._lib import GEMM_CONV3X3, GEMM_PLAIN, GEMM_TEMPORAL3, OUT_F16, OUT_F32, OUT_GEGLU
from .video_model import (Downsample, SpatialVideoTransformer, Upsample, VideoResBlock)

CIN_PAD = 64    # first conv: input channels padded to the GEMM K granule
COUT_PAD = 16   # last conv: output channels padded to the GEMM N granule


_GN_REVERSE = os.environ.get("GCD_GN_REVERSE", "1") != "0"   # A/B switch (see ops.groupnorm_apply)
# Walk directions of the big streaming launches (GEMMs on the 8-phase kernel, LayerNorm, GroupNorm apply) — pure
# scheduling, results bit-identical.  What a launch wrote LAST is what the 256 MB Infinity Cache still holds when the
# next one starts, so a consumer that walks in the OPPOSITE direction of its producer turns part of its HBM reads into
# cache hits (gcd_gemm_desc.sched, the `order` of gcd_layernorm_f16 / gcd_groupnorm_apply).
#   0  everything front to back, except the GroupNorm apply behind a GEMM that left its column sums (back to front)
#   1  zig-zag: every big launch walks opposite to the previous one; the norms use the GEMM's region order (eight
#      contiguous shares walked concurrently); the attention cores count as front to back
#   2  as 1 with whole-tensor orders for the norms        3  as 0 with the region order for that GroupNorm
#   4  as 0 plus every LayerNorm back to front (region order)
# Measured on one box, interleaved (profiles/r04m_ab_sweep.txt, r04n_ab_sweep.txt): 0: 102.09 / 100.90 ms per step,
# 1: 101.84, 2: 101.76 / 100.57, 3: 101.97, 4: 102.04 — 2 is the default (-0.3 ms; the cache keeps less of a producer's
# tail than its size suggests, the rest of the step's traffic flows through it too).
_ZIGZAG = int(os.environ.get("GCD_ZIGZAG", "2"))
_ITEMSIZE = {torch.float16: 2, torch.float32: 4, torch.float64: 8, torch.uint8: 1}


class Workspace:
    """Deterministic slab pool on one device (torch owns the memory, we own the placement).

    The first run with a given signature records which slab served each allocation; later runs with
    the same signature replay that record, so every buffer address is identical from call to call
    (required for hipGraph replay) and no allocation can happen inside a stream capture."""

    def __init__(self, device):
        self.device = device
        self.slabs: List[torch.Tensor] = []
        self.free: List[bool] = []
        self.by_ptr: Dict[int, int] = {}
        self._sig = None
        self._trace: List[int] = []
        self._complete = False
        self._replay = False
        self._k = 0
        self._aux: Dict[int, torch.Tensor] = {}   # data_ptr of a buffer -> side buffer that lives with it

    def attach(self, t: torch.Tensor, aux: Optional[torch.Tensor]) -> None:
        """Tie a side buffer (the GroupNorm column sums of `t`) to `t`: released with it, replaced
        (and the old one released) when `t` is rewritten."""
        old = self._aux.pop(t.data_ptr(), None)
        if old is not None:
            self.free[self.by_ptr[old.data_ptr()]] = True
        if aux is not None:
            self._aux[t.data_ptr()] = aux

    def attached(self, t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
        return None if t is None else self._aux.get(t.data_ptr())

    def reset(self, signature=None) -> None:
        for i in range(len(self.free)):
            self.free[i] = True
        self._k = 0
        self._aux.clear()
        if signature is None or signature != self._sig or not self._complete:
            self._sig, self._trace, self._complete, self._replay = signature, [], False, False
        else:
            self._replay = True

    def finish(self) -> None:
        if not self._replay:
            self._complete = True

    def alloc(self, shape, dtype) -> torch.Tensor:
        n = 1
        for s in shape:
            n *= int(s)
        nbytes = n * _ITEMSIZE[dtype]
        need = (nbytes + 255) // 256 * 256
        if self._replay:
            best = self._trace[self._k]
            assert self.free[best] and self.slabs[best].numel() >= need, "workspace replay diverged"
        else:
            best = -1
            for i, slab in enumerate(self.slabs):
                if self.free[i] and slab.numel() >= need and \
                        (best < 0 or slab.numel() < self.slabs[best].numel()):
                    best = i
            if best < 0:
                self.slabs.append(torch.empty(need, dtype=torch.uint8, device=self.device))
                self.free.append(True)
                best = len(self.slabs) - 1
                self.by_ptr[self.slabs[best].data_ptr()] = best
            self._trace.append(best)
        self._k += 1
        self.free[best] = False
        return self.slabs[best][:nbytes].view(dtype).view(*shape)

    def release(self, *tensors) -> None:
        for t in tensors:
            if t is None:
                continue
            aux = self._aux.pop(t.data_ptr(), None)
            if aux is not None:
                self.free[self.by_ptr[aux.data_ptr()]] = True
            self.free[self.by_ptr[t.data_ptr()]] = True

    def nbytes(self) -> int:
        return sum(s.numel() for s in self.slabs)


def _f32(p: torch.Tensor) -> torch.Tensor:
    return p.detach().to(torch.float32).contiguous()


class UNetEngine:
    def __init__(self, unet):
        self.unet = unet
        self.packed = None
        self.ws: Optional[Workspace] = None
        self._graphs: Dict[tuple, dict] = {}
        self.use_graph = False
        # LayerNorm in the epilogue of the N == 320 GEMMs.  Measured neutral on MI355X (7.288 vs 7.297
        # steps/s in interleaved runs: the stand-alone LayerNorm already runs at 5.9 TB/s and the fused
        # epilogue pays for it in VALU and 8-byte fp16 stores), so it is off unless GCD_FUSE_LN=1.
        self.fuse_layernorm = os.environ.get("GCD_FUSE_LN", "0") == "1"
        self.taps: Optional[dict] = None   # debug: name -> NCHW fp32 clone of every block output
        self._last_dir = 1                  # direction of the last big launch (zig-zag schedule, _ZIGZAG)
        self._side_streams: Dict[int, "torch.cuda.Stream"] = {}

    def side_stream(self, device) -> "torch.cuda.Stream":
        """ONE side stream per (engine, device) for the fused sampling loops: a fresh stream per
        sampler call would walk through torch's 32-stream pool and pin a split-K scratch (84 MB,
        keyed by stream) for each of them."""
        idx = torch.device(device).index or 0
        s = self._side_streams.get(idx)
        if s is None:
            s = self._side_streams[idx] = torch.cuda.Stream(device=device)
        return s

    # ------------------------------------------------------------------------------------------
    def invalidate(self) -> None:
        self.packed = None
        self._destroy_graphs()

    def _destroy_graphs(self) -> None:
        for g in self._graphs.values():
            try:
                _lib.load().gcd_graph_destroy(g["exec"])
            except Exception:
                pass
        self._graphs = {}

    # ------------------------------------------------------------------------------------------
    # weight packing
    # ------------------------------------------------------------------------------------------
    def _device(self) -> torch.device:
        return self.unet.out[2].weight.device

    def pack(self) -> None:
        u = self.unet
        dev = self._device()
        if dev.type != "cuda":
            raise _lib.GcdError("gcd_amd.VideoUNet parameters are on the CPU: move the model to the "
                                "GPU (`.to('cuda')`); there is no CPU execution path")
        _lib.load()
        P = {}
        emb_w, emb_b, emb_off = [], [], [0]
        ca_wv, ca_mods = [], []
        blenders = []
        pos_mlps = []

        def add_emb(lin):
            emb_w.append(_f32(lin.weight))
            emb_b.append(_f32(lin.bias))
            off = emb_off[0]
            emb_off[0] += lin.weight.shape[0]
            return (off, lin.weight.shape[0])

        def add_ca(att, temporal):
            off = sum(w.shape[0] for w in ca_wv)
            ca_wv.append(_f32(att.to_v.weight))
            ca_mods.append(dict(off=off, C=att.to_v.weight.shape[0], wo=_f32(att.to_out[0].weight),
                                bo=_f32(att.to_out[0].bias), temporal=temporal))
            return len(ca_mods) - 1

        def add_blender(b):
            blenders.append(b)
            return len(blenders) - 1

        def pack_res2d(rb: VideoResBlock):
            d = dict(kind="res", cin=rb.channels, cout=rb.out_channels)
            d["gn1"] = (_f32(rb.in_layers[0].weight), _f32(rb.in_layers[0].bias))
            d["w1"], d["b1"] = packing.pack_conv3x3(rb.in_layers[2].weight), _f32(rb.in_layers[2].bias)
            d["emb"] = add_emb(rb.emb_layers[1])
            d["gn2"] = (_f32(rb.out_layers[0].weight), _f32(rb.out_layers[0].bias))
            d["w2"], d["b2"] = packing.pack_conv3x3(rb.out_layers[3].weight), _f32(rb.out_layers[3].bias)
            if isinstance(rb.skip_connection, torch.nn.Identity):
                d["wskip"] = None
            else:
                d["wskip"] = packing.pack_conv1x1(rb.skip_connection.weight)
                d["bskip"] = _f32(rb.skip_connection.bias)
            ts = rb.time_stack
            t = dict()
            t["gn1"] = (_f32(ts.in_layers[0].weight), _f32(ts.in_layers[0].bias))
            t["w1"], t["b1"] = packing.pack_conv_t3(ts.in_layers[2].weight), _f32(ts.in_layers[2].bias)
            t["emb"] = add_emb(ts.emb_layers[1])
            t["gn2"] = (_f32(ts.out_layers[0].weight), _f32(ts.out_layers[0].bias))
            t["w2"], t["b2"] = packing.pack_conv_t3(ts.out_layers[3].weight), _f32(ts.out_layers[3].bias)
            d["ts"] = t
            d["blend"] = add_blender(rb.time_mixer)
            return d

        def ln(m):
            return (_f32(m.weight), _f32(m.bias))

        def pack_ff(ff):
            w1, b1 = packing.pack_geglu(ff.net[0].proj.weight, ff.net[0].proj.bias)
            d = dict(w1=w1, b1=b1, w2=packing.pack_linear(ff.net[2].weight), b2=_f32(ff.net[2].bias))
            if w1.is_cuda and tuple(w1.shape) == (2560, 320) and tuple(d["w2"].shape) == (320, 1280):
                # the one-kernel LayerNorm + FeedForward's fragment-order weight stream (K order of the LayerNorm form)
                d["wp"] = ops.ff_pack(w1, d["w2"], for_ln=True)
            return d

        def pack_attn(att, q_scale=1.0):
            d = dict(wqkv=packing.pack_qkv(att.to_q.weight, att.to_k.weight, att.to_v.weight,
                                           q_scale=q_scale),
                     wo=packing.pack_linear(att.to_out[0].weight), bo=_f32(att.to_out[0].bias))
            if d["wqkv"].is_cuda and tuple(d["wqkv"].shape) == (960, 320):
                # the one-kernel LayerNorm + q | k | v's fragment-order weights (lnqkv.hip)
                d["wqkv_p"] = ops.lnqkv_pack(d["wqkv"])
            return d

        def pack_tr(tr: SpatialVideoTransformer):
            d = dict(kind="attn", C=tr.in_channels, heads=tr.heads, depth=tr.depth)
            d["gn"] = (_f32(tr.norm.weight), _f32(tr.norm.bias))
            d["win"], d["bin"] = packing.pack_linear(tr.proj_in.weight), _f32(tr.proj_in.bias)
            d["wout"], d["bout"] = packing.pack_linear(tr.proj_out.weight), _f32(tr.proj_out.bias)
            d["blocks"] = []
            for sb, tb in zip(tr.transformer_blocks, tr.time_stack):
                s = dict(ln1=ln(sb.norm1), attn=pack_attn(sb.attn1, ops.ATTN_Q_SCALE_LOG2),
                         ca=add_ca(sb.attn2, False),
                         ln3=ln(sb.norm3), ff=pack_ff(sb.ff))
                t = dict(ln_in=ln(tb.norm_in), ff_in=pack_ff(tb.ff_in), ln1=ln(tb.norm1),
                         attn=pack_attn(tb.attn1), ca=add_ca(tb.attn2, True), ln3=ln(tb.norm3),
                         ff=pack_ff(tb.ff))
                d["blocks"].append((s, t))
            pos_mlps.append(dict(C=tr.in_channels, period=float(tr.max_time_embed_period),
                                 w0=_f32(tr.time_pos_embed[0].weight), b0=_f32(tr.time_pos_embed[0].bias),
                                 w2=_f32(tr.time_pos_embed[2].weight), b2=_f32(tr.time_pos_embed[2].bias)))
            d["pos"] = len(pos_mlps) - 1
            d["blend"] = add_blender(tr.time_mixer)
            return d

        def pack_seq(seq):
            layers = []
            for m in seq:
                if isinstance(m, VideoResBlock):
                    layers.append(pack_res2d(m))
                elif isinstance(m, SpatialVideoTransformer):
                    layers.append(pack_tr(m))
                elif isinstance(m, Downsample):
                    layers.append(dict(kind="down", w=packing.pack_conv3x3(m.op.weight), b=_f32(m.op.bias),
                                       cin=m.channels, cout=m.out_channels))
                elif isinstance(m, Upsample):
                    layers.append(dict(kind="up", w=packing.pack_conv3x3(m.conv.weight), b=_f32(m.conv.bias),
                                       cin=m.channels, cout=m.out_channels))
                elif isinstance(m, torch.nn.Conv2d):
                    layers.append(dict(kind="conv_in", w=packing.pack_conv3x3(m.weight, cin_pad=CIN_PAD),
                                       b=_f32(m.bias), cout=m.out_channels))
                else:
                    raise NotImplementedError(f"layer {type(m).__name__}")
            return layers

        P["input"] = [pack_seq(s) for s in u.input_blocks]
        P["middle"] = pack_seq(u.middle_block)
        P["output"] = [pack_seq(s) for s in u.output_blocks]
        P["out_gn"] = (_f32(u.out[0].weight), _f32(u.out[0].bias))
        P["out_w"] = packing.pack_conv3x3(u.out[2].weight, cout_pad=COUT_PAD)
        ob = torch.zeros(COUT_PAD, dtype=torch.float32, device=dev)
        ob[:u.out_channels] = u.out[2].bias.detach().float()
        P["out_b"] = ob

        def mlp(seq):
            return dict(w0=_f32(seq[0].weight), b0=_f32(seq[0].bias), w2=_f32(seq[2].weight),
                        b2=_f32(seq[2].bias))

        P["time_embed"] = mlp(u.time_embed)
        P["label_emb"] = mlp(u.label_emb[0])
        P["aux_label_emb"] = mlp(u.aux_label_emb) if u.aux_emb_dim > 0 else None
        P["emb_w"] = torch.cat(emb_w, 0).contiguous()
        P["emb_b"] = torch.cat(emb_b, 0).contiguous()
        P["emb_total"] = emb_off[0]
        P["ca_wv"] = torch.cat(ca_wv, 0).contiguous()
        P["ca_mods"] = ca_mods
        P["ca_total"] = P["ca_wv"].shape[0]
        P["pos_mlps"] = pos_mlps
        mix = torch.cat([b.mix_factor.detach().float().reshape(1) for b in blenders])
        learned = torch.tensor([b.merge_strategy != "fixed" for b in blenders], device=mix.device)
        P["blend_alpha"] = torch.where(learned, torch.sigmoid(mix), mix).contiguous()   # util.py:342-346
        P["blend_with_images"] = torch.tensor(
            [b.merge_strategy == "learned_with_images" for b in blenders], device=mix.device)
        P["pos_cache"] = {}
        self.packed = P
        if self.ws is None or self.ws.device != dev:
            self.ws = Workspace(dev)

    # ------------------------------------------------------------------------------------------
    # small helpers
    # ------------------------------------------------------------------------------------------
    def _gn(self, x1, x2, rows, eps, affine, silu, want_raw):
        ws = self.ws
        M = x1.shape[0]
        C = x1.shape[1] + (0 if x2 is None else x2.shape[1])
        ninst = M // rows
        stats = ws.alloc((ninst * 64,), torch.float32)
        order = 0
        cs1, cs2 = ws.attached(x1), ws.attached(x2)
        if cs1 is not None and (x2 is None or cs2 is not None) and rows % 64 == 0:
            # the GEMM epilogues that wrote x1 (and x2) left per-64-row column sums behind: no
            # statistics pass over the tensor
            ops.groupnorm_stats_from_colsums(cs1, x1.shape[1], cs2, 0 if x2 is None else x2.shape[1],
                                             M, rows, eps, stats)
            partial = None
            # nothing has read the tensor since the GEMM wrote it front to back: start at its tail
            if _GN_REVERSE and x2 is None and _ZIGZAG in (0, 3, 4):
                order = 2 if _ZIGZAG == 3 else 1
        else:
            nch = ops.gn_nchunks(rows, ninst)
            partial = ws.alloc((ninst * nch * 64,), torch.float64)
            ops.groupnorm_stats(x1, x2, rows, eps, partial, stats, nch)
            self._walked(0)      # the statistics pass read the tensor front to back
        if _ZIGZAG in (1, 2):
            order = self._norm_order()
        y = ws.alloc((M, C), torch.float16)
        raw = ws.alloc((M, C), torch.float16) if want_raw else None
        ops.groupnorm_apply(x1, x2, rows, stats, affine[0], affine[1], silu, y, raw, order=order)
        ws.release(partial, stats)
        return y, raw

    def _gemm_gn(self, a16, w16, out, **kw):
        """`ops.gemm` for an fp32 output that a GroupNorm reads next: where the shape allows
        (gcd_gemm_colstats_supported) the epilogue also writes the per-64-row column sums and they
        travel with `out` (Workspace.attach); otherwise any stale sums of `out` are dropped."""
        ws = self.ws
        if ops.gemm(a16, w16, out, probe_colstats=True, **kw):
            cs = ws.alloc((2 * (kw["M"] // 64), w16.shape[0]), torch.float32)
            self._gemm(a16, w16, out, colstats=cs, **kw)
            ws.attach(out, cs)
        else:
            self._gemm(a16, w16, out, **kw)
            ws.attach(out, None)
        return out

    # ---- walk directions (see _ZIGZAG) ----
    def _walked(self, d: int) -> None:
        """Record the direction (0 front to back, 1 back to front) of a big launch that chose none."""
        self._last_dir = d

    def _next_dir(self) -> int:
        d = 1 - self._last_dir
        self._last_dir = d
        return d

    def _norm_order(self) -> int:
        d = self._next_dir()
        if _ZIGZAG == 1:
            return 2 if d else 3
        return d

    def _gemm(self, a16, w16, out, **kw):
        """`ops.gemm` with the walk direction of the zig-zag schedule (gcd_gemm_desc.sched bit 0)."""
        if _ZIGZAG in (1, 2):
            # only the 256 x 320 tile kernels honour the bit (automatic choice: >= 192 tiles and N >= 160, gemm.hip); a
            # small launch on the general kernel walks front to back whatever it is told and must not flip the record
            # the next big launch alternates against
            M, N = kw["M"], w16.shape[0]
            if ((M + 255) // 256) * ((N + 319) // 320) >= 192 and N >= 160:
                kw["sched"] = self._next_dir()
        return ops.gemm(a16, w16, out, **kw)

    def _ln(self, x, affine, addvec=None, rows_per_vec=1, sum_out=None):
        y = self.ws.alloc(tuple(x.shape), torch.float16)
        order = self._norm_order() if _ZIGZAG in (1, 2) else (2 if _ZIGZAG == 4 else 0)
        ops.layernorm(x, affine[0], affine[1], y, addvec=addvec, rows_per_vec=rows_per_vec,
                      sum_out=sum_out, order=order)
        return y

    def _mlp_small(self, x, m, out=None, accumulate=False):
        """Linear -> SiLU -> Linear on [N, K] fp32 rows (any N: gcd_linear_smallm_f32 chunks by 32)."""
        ws = self.ws
        n = x.shape[0]
        hid = ws.alloc((n, m["w0"].shape[0]), torch.float32)
        ops.linear_smallm(x, m["w0"], m["b0"], hid, silu_out=True)
        if out is None:
            out = ws.alloc((n, m["w2"].shape[0]), torch.float32)
        ops.linear_smallm(hid, m["w2"], m["b2"], out, accumulate=accumulate)
        ws.release(hid)
        return out

    # ------------------------------------------------------------------------------------------
    # layers
    # ------------------------------------------------------------------------------------------
    def _resblock(self, L, x1, x2, st):
        """VideoResBlock (video_model.py:62-81, openaimodel.py:331-357).  x = [x1 | x2] fp32."""
        ws, N, T = self.ws, st["N"], st["T"]
        H, W = st["H"], st["W"]
        HW = H * W
        M = N * HW
        cin, cout = L["cin"], L["cout"]
        emb_all = st["emb_all"]
        conv = dict(Cin=cin, Hi=H, Wi=W, Ho=H, Wo=W, stride=1, upsample=0)
        # --- spatial ResBlock ---
        a16, raw16 = self._gn(x1, x2, HW, 1e-5, L["gn1"], True, L["wskip"] is not None)
        h1 = ws.alloc((M, cout), torch.float32)
        off, n = L["emb"]
        self._gemm_gn(a16, L["w1"], h1, M=M, mode=GEMM_CONV3X3, bias=L["b1"],
                      rowvec=emb_all[:, off:off + n], rows_per_vec=HW, conv=conv)
        ws.release(a16)
        a16, _ = self._gn(h1, None, HW, 1e-5, L["gn2"], True, False)
        xs = ws.alloc((M, cout), torch.float32)
        conv2 = dict(conv, Cin=cout)
        if L["wskip"] is not None:
            self._gemm(raw16, L["wskip"], xs, M=M, bias=L["bskip"])
            ws.release(raw16)
            self._gemm_gn(a16, L["w2"], xs, M=M, mode=GEMM_CONV3X3, bias=L["b2"], r1=xs, conv=conv2)
        else:
            assert x2 is None
            self._gemm_gn(a16, L["w2"], xs, M=M, mode=GEMM_CONV3X3, bias=L["b2"], r1=x1, conv=conv2)
        ws.release(a16)
        # --- time_stack ResBlock (dims 3, kernel (3,1,1), GroupNorm over T*H*W) + AlphaBlender ---
        ts = L["ts"]
        tconv = dict(Cin=cout, T=T, HW=HW)
        a16, _ = self._gn(xs, None, T * HW, 1e-5, ts["gn1"], True, False)
        off, n = ts["emb"]
        self._gemm_gn(a16, ts["w1"], h1, M=M, mode=GEMM_TEMPORAL3, bias=ts["b1"],
                      rowvec=emb_all[:, off:off + n], rows_per_vec=HW, conv=tconv)
        ws.release(a16)
        a16, _ = self._gn(h1, None, T * HW, 1e-5, ts["gn2"], True, False)
        ws.release(h1)
        # out = x_s + (1 - alpha) * (conv + b): alpha*x_s + (1-alpha)*(x_s + conv + b), util.py:364-368
        self._gemm_gn(a16, ts["w2"], xs, M=M, mode=GEMM_TEMPORAL3, bias=ts["b2"], r1=xs,
                      frame_alpha=st["alphas"][L["blend"]], rows_per_alpha=HW, r1_blend=False, conv=tconv)
        ws.release(a16)
        return xs

    def _ff_ln(self, F, x32, affine, M, *, out, r2=None, frame_alpha=None, rows_per_alpha=1, out_kind=OUT_F32,
               addvec=None, rows_per_vec=1) -> bool:
        """x = ff(norm(x32 + addvec)) + (x32 + addvec) [blended with r2] as ONE launch (ff_fused_kernel.h, the LayerNorm
        form): the LayerNorm kernel, its fp16 output and the 660 MB hidden tensor never touch memory.  False = not taken
        (another level, too few tokens, switched off): the caller runs LayerNorm + two GEMMs."""
        if "wp" not in F or self.fuse_layernorm or not ops.ff_fused_ok(M, F["w1"].shape[1], F["w1"].shape[0] // 2):
            return False
        ln = dict(gamma=affine[0], beta=affine[1], addvec=addvec, rows_per_vec=rows_per_vec)
        kw = dict(r2=r2, out_kind=out_kind, frame_alpha=frame_alpha, rows_per_alpha=rows_per_alpha, ln=ln)
        if _ZIGZAG in (1, 2):
            kw["sched"] = self._next_dir()
        ops.ff_fused(x32, F["wp"], F["b1"], F["b2"], out, M=M, **kw)
        return True

    def _ln_qkv(self, A, x32, affine, M) -> torch.Tensor:
        """q | k | v = to_qkv(norm1(x32)) as fp16 [M, 3 C]: ONE launch where the model width is 320 and the tile count pays
        (lnqkv.hip: the fp32 rows are read once, the normalised operand never reaches memory), else LayerNorm + GEMM."""
        ws = self.ws
        N3 = A["wqkv"].shape[0]
        qkv = ws.alloc((M, N3), torch.float16)
        if "wqkv_p" in A and not self.fuse_layernorm and ops.lnqkv_ok(M, A["wqkv"].shape[1], N3):
            sched = self._next_dir() if _ZIGZAG in (1, 2) else 0
            ops.lnqkv(x32, affine[0], affine[1], A["wqkv_p"], qkv, M=M, N=N3, sched=sched)
            return qkv
        a16 = self._ln(x32, affine)
        self._gemm(a16, A["wqkv"], qkv, M=M, out_kind=OUT_F16)
        ws.release(a16)
        return qkv

    def _ff(self, F, a16, M, **epi):
        ws = self.ws
        hid = ws.alloc((M, F["w1"].shape[0] // 2), torch.float16)
        # the hidden tensor only lives between these two GEMMs: where both run on the ping-pong kernel it
        # is kept tile-blocked ([M/256][N/320][256][160]: whole 128-byte lines per store, §3.2 of DESIGN.md)
        blocked = epi.get("ln") is None and ops.gemm_hidden_blocked_ok(M, F["w1"].shape[0], F["w2"].shape[0])
        self._gemm(a16, F["w1"], hid, M=M, bias=F["b1"], out_kind=OUT_GEGLU, out_blocked=blocked)
        ws.release(a16)
        self._gemm(hid, F["w2"], epi.pop("out"), M=M, bias=F["b2"], a_blocked=blocked, **epi)
        ws.release(hid)

    def _transformer(self, L, x, st):
        """SpatialVideoTransformer.forward (video_attention.py:230-301); x fp32 [M, C], updated in place."""
        ws, N, T = self.ws, st["N"], st["T"]
        HW = st["H"] * st["W"]
        M = N * HW
        Cc, heads = L["C"], L["heads"]
        ca = st["ca"]
        a16, _ = self._gn(x, None, HW, 1e-6, L["gn"], False, False)
        xs = ws.alloc((M, Cc), torch.float32)
        # LayerNorm fused into the epilogue of the GEMM that writes its input (C == 320 rows fit one
        # 256x320 tile of the ping-pong kernel); otherwise the stand-alone kernel.
        fuse = self.fuse_layernorm and ops.gemm_ln_fusable(M, Cc, Cc)

        def ln_req(affine, **extra):
            if not fuse:
                return None, None
            y = ws.alloc((M, Cc), torch.float16)
            return y, dict(gamma=affine[0], beta=affine[1], out16=y, **extra)

        blocks = L["blocks"]
        nxt, req = ln_req(blocks[0][0]["ln1"])
        self._gemm(a16, L["win"], xs, M=M, bias=L["bin"], ln=req)
        ws.release(a16)
        pos = self._pos_embed(L["pos"], N, T)
        S_pad = (HW + 63) // 64 * 64
        last = None
        for bi, (sb, tb) in enumerate(blocks):
            # ---- spatial BasicTransformerBlock (attention.py:551-572) ----
            if fuse:
                qkv = ws.alloc((M, 3 * Cc), torch.float16)
                self._gemm(nxt, sb["attn"]["wqkv"], qkv, M=M, out_kind=OUT_F16)
                ws.release(nxt)
            else:
                qkv = self._ln_qkv(sb["attn"], xs, sb["ln1"], M)
            vt = ws.alloc((N * heads * 64 * S_pad,), torch.float16)
            ops.attn_transpose_v(qkv, N, HW, heads, vt, S_pad)
            ao = ws.alloc((M, Cc), torch.float16)
            ops.attn_spatial(qkv, vt, S_pad, ao, N, HW, heads, q_prescaled=True)
            self._walked(0)
            ws.release(qkv, vt)
            # x = attn1 + x ; x = attn2 + x  (attn2 == per-frame vector, one key)
            nxt, req = ln_req(sb["ln3"])
            self._gemm(ao, sb["attn"]["wo"], xs, M=M, bias=sb["attn"]["bo"], r1=xs,
                     rowvec=ca[sb["ca"]], rows_per_vec=HW, ln=req)
            ws.release(ao)
            # ---- temporal VideoTransformerBlock (video_attention.py:109-140) on x + frame pos-emb ----
            xm = ws.alloc((M, Cc), torch.float32)
            if not self._ff_ln(sb["ff"], xs, sb["ln3"], M, out=xs):                  # x = ff(norm3(x)) + x
                a16 = nxt if fuse else self._ln(xs, sb["ln3"])
                nxt, req = ln_req(tb["ln_in"], addvec=pos, rows_per_vec=HW, sum_out=xm)
                self._ff(sb["ff"], a16, M, out=xs, r1=xs, ln=req)
            # x_mix = x + pos; x_mix = ff_in(norm_in(x_mix)) + x_mix
            if not self._ff_ln(tb["ff_in"], xs, tb["ln_in"], M, out=xm, addvec=pos, rows_per_vec=HW):
                a16 = nxt if fuse else self._ln(xs, tb["ln_in"], addvec=pos, rows_per_vec=HW, sum_out=xm)
                nxt, req = ln_req(tb["ln1"])
                self._ff(tb["ff_in"], a16, M, out=xm, r1=xm, ln=req)
            if fuse:
                qkv = ws.alloc((M, 3 * Cc), torch.float16)
                self._gemm(nxt, tb["attn"]["wqkv"], qkv, M=M, out_kind=OUT_F16)
                ws.release(nxt)
            else:
                qkv = self._ln_qkv(tb["attn"], xm, tb["ln1"], M)
            ao = ws.alloc((M, Cc), torch.float16)
            ops.attn_temporal(qkv, ao, N // T, T, HW, heads)
            self._walked(0)
            ws.release(qkv)
            nxt, req = ln_req(tb["ln3"])
            self._gemm(ao, tb["attn"]["wo"], xm, M=M, bias=tb["attn"]["bo"], r1=xm,
                     rowvec=ca[tb["ca"]], rows_per_vec=T * HW, ln=req)
            ws.release(ao)
            # x = alpha*x + (1-alpha)*(ff(...) + x_mix)   (AlphaBlender, video_attention.py:289-293)
            final = bi == len(blocks) - 1
            if final:
                last = ws.alloc((M, Cc), torch.float16)
            if self._ff_ln(tb["ff"], xm, tb["ln3"], M, out=last if final else xs, r2=xs, out_kind=OUT_F16 if final else OUT_F32,
                           frame_alpha=st["alphas"][L["blend"]], rows_per_alpha=HW):
                ws.release(xm)
                continue
            a16 = nxt if fuse else self._ln(xm, tb["ln3"])
            if final:   # only proj_out reads the result: emit it as the fp16 GEMM operand directly
                self._ff(tb["ff"], a16, M, out=last, out_kind=OUT_F16, r1=xm, r2=xs,
                         frame_alpha=st["alphas"][L["blend"]], rows_per_alpha=HW, r1_blend=True)
            else:
                nxt, req = ln_req(blocks[bi + 1][0]["ln1"])
                self._ff(tb["ff"], a16, M, out=xs, r1=xm, r2=xs,
                         frame_alpha=st["alphas"][L["blend"]], rows_per_alpha=HW, r1_blend=True, ln=req)
            ws.release(xm)
        ws.release(xs)
        self._gemm_gn(last, L["wout"], x, M=M, bias=L["bout"], r1=x)   # + x_in; the next block's GroupNorm reads x
        ws.release(last)
        return x

    def _pos_embed(self, idx, N, T):
        """time_pos_embed(timestep_embedding(frame index)) — input independent, cached per (N, T)
        (video_attention.py:272-284)."""
        P = self.packed
        key = (idx, N, T)
        if key not in P["pos_cache"]:
            m = P["pos_mlps"][idx]
            dev = m["w0"].device
            frames = torch.arange(T, device=dev, dtype=torch.float32).repeat(N // T)
            temb = torch.empty(N, m["C"], device=dev, dtype=torch.float32)
            ops.timestep_embedding(frames, temb, m["period"])
            hid = torch.empty(N, m["w0"].shape[0], device=dev, dtype=torch.float32)
            ops.linear_smallm(temb, m["w0"], m["b0"], hid, silu_out=True)
            out = torch.empty(N, m["C"], device=dev, dtype=torch.float32)
            ops.linear_smallm(hid, m["w2"], m["b2"], out)
            P["pos_cache"][key] = out
        return P["pos_cache"][key]

    # ------------------------------------------------------------------------------------------
    # conditioning vectors
    # ------------------------------------------------------------------------------------------
    def _embeddings(self, timesteps, y, st):
        """emb = time_embed(t_emb) + label_emb(y[:, :adm]) + aux_label_emb(y[:, adm:])
        (video_model.py:483-497), then every ResBlock's Linear(SiLU(emb)) in one launch."""
        P, ws, u = self.packed, self.ws, self.unet
        N = st["N"]
        temb = ws.alloc((N, u.model_channels), torch.float32)
        ops.timestep_embedding(timesteps, temb, 10000.0)
        emb = self._mlp_small(temb, P["time_embed"])
        ws.release(temb)
        adm = u.adm_in_channels
        self._mlp_small(y[:, :adm], P["label_emb"], out=emb, accumulate=True)
        if P["aux_label_emb"] is not None:
            self._mlp_small(y[:, adm:], P["aux_label_emb"], out=emb, accumulate=True)
        emb_all = ws.alloc((N, P["emb_total"]), torch.float32)
        ops.linear_smallm(emb, P["emb_w"], P["emb_b"], emb_all, silu_in=True)
        ws.release(emb)
        return emb_all

    def _cross_attention_vectors(self, context_src, context, st):
        """attn2 with a single key: out = to_out(to_v(ctx)) per frame (spatial, attention.py:300-344)
        or per clip from the first frame's context (temporal, video_attention.py:244-253).

        The vectors depend on the context and the weights only, not on the noise level, so they are
        kept across calls for as long as the caller passes the SAME tensor object, unmodified — every
        step of the fused sampling loop after the first (`pack()` clears the cache when the weights
        change).  Identity, not address: the cache holds a reference to the tensor, so its storage
        cannot be freed and handed to a different context at the same address while the entry lives
        (a (data_ptr, _version) key would silently reuse the vectors of the previous clip there);
        in-place edits bump torch's version counter and miss."""
        P = self.packed
        N, T = st["N"], st["T"]
        # inference-mode tensors have no version counter (reading _version raises): they cannot be
        # edited in place behind our back, so identity alone is the key there
        ver = -1 if context_src.is_inference() else context_src._version
        key = (ver, tuple(context_src.shape), context_src.dtype, N, T)
        hit = P.get("ca_cache")
        if hit is not None and hit[0] is context_src and hit[1] == key:
            return hit[2]
        dev = context.device
        v_all = torch.empty((N, P["ca_total"]), dtype=torch.float32, device=dev)
        ops.linear_smallm(context, P["ca_wv"], None, v_all)
        outs = []
        for m in P["ca_mods"]:
            src = v_all[:, m["off"]:m["off"] + m["C"]]
            if m["temporal"]:
                src = src[::T]
            o = torch.empty((src.shape[0], m["C"]), dtype=torch.float32, device=dev)
            ops.linear_smallm(src, m["wo"], m["bo"], o)
            outs.append(o)
        P["ca_cache"] = (context_src, key, outs, v_all)
        return outs

    # ------------------------------------------------------------------------------------------
    # forward
    # ------------------------------------------------------------------------------------------
    def forward(self, x, timesteps, context, y, num_video_frames, image_only_indicator):
        """Reference-signature entry (NCHW in, NCHW out)."""
        ops._need_gpu(x, timesteps, context, y)
        if self.packed is None:
            self.pack()
        N, Cx, H, W = x.shape
        assert Cx == self.unet.in_channels
        out_dtype = x.dtype
        x32 = x.detach().float().contiguous()
        out = torch.empty(N, self.unet.out_channels, H, W, device=x.device, dtype=torch.float32)
        self.run(x32, None, None, timesteps.detach().float().contiguous(),
                 context, y, num_video_frames, image_only_indicator, out)
        return out if out_dtype == torch.float32 else out.to(out_dtype)

    def blend_alphas(self, image_only_indicator, N, T):
        """alpha per (blender, frame): 1 where image_only_indicator, else sigmoid(mix_factor)
        (util.py:342-356).  Returns fp32 [num_blenders, N]."""
        P = self.packed
        base = P["blend_alpha"][:, None].expand(-1, N)
        if image_only_indicator is None:
            if bool(P["blend_with_images"].any()):
                raise AssertionError("need image_only_indicator ...")
            return base.contiguous()
        ioi = image_only_indicator.reshape(-1).to(base.device)
        assert ioi.numel() == N, f"image_only_indicator has {ioi.numel()} entries for {N} frames"
        with_img = P["blend_with_images"][:, None]
        return torch.where(with_img & (ioi[None, :] != 0), torch.ones_like(base), base).contiguous()

    def run(self, x, concat, c_in, timesteps, context, y, T, image_only_indicator, out_nchw,
            alphas=None):
        """x: [nx, Cx, H, W] fp32 (nx == N, or N == 2*nx for the CFG-duplicated fused path with
        `concat` [N, Cc, H, W] and per-frame `c_in` [N]); writes out_nchw [N, out_ch, H, W] fp32."""
        if x.device.index is not None and x.device.index != torch.cuda.current_device():
            # kernels launch on the current device's current stream: make the tensors' device current
            with torch.cuda.device(x.device):
                return self.run(x, concat, c_in, timesteps, context, y, T, image_only_indicator,
                                out_nchw, alphas)
        if self.packed is not None and self._device() != x.device:
            raise _lib.GcdError(f"VideoUNet parameters are on {self._device()} but the input is on {x.device}")
        P, ws, u = self.packed, self.ws, self.unet
        N = timesteps.shape[0]
        H, W = x.shape[-2:]
        assert T is not None and N % T == 0, "num_video_frames must divide the batch of frames"
        assert context is not None and context.dim() == 3 and context.shape[0] == N, \
            f"n dims of spatial context should be 3 but are {None if context is None else context.ndim}"
        if context.shape[1] != 1:
            raise NotImplementedError("gcd_amd implements the single-token (CLIP image) context of "
                                      "SVD/GCD; got %d context tokens" % context.shape[1])
        nlev = len(u.channel_mult)
        if H % (1 << (nlev - 1)) or W % (1 << (nlev - 1)):
            raise ValueError(f"latent size {H}x{W} must be divisible by {1 << (nlev - 1)}")
        ws.reset((N, H, W, T))
        self._last_dir = 1
        st = dict(N=N, T=T, H=H, W=W)
        st["alphas"] = alphas if alphas is not None else self.blend_alphas(image_only_indicator, N, T)
        ctx2d = context.detach().float().reshape(N, -1).contiguous()
        y32 = y.detach().float().contiguous()
        st["emb_all"] = self._embeddings(timesteps, y32, st)
        st["ca"] = self._cross_attention_vectors(context, ctx2d, st)

        M = N * H * W
        xin = ws.alloc((M, CIN_PAD), torch.float16)
        ops.pack_input(x, concat, c_in, N, H * W, xin, CIN_PAD)

        hs: List[torch.Tensor] = []
        h = None
        for bi, layers in enumerate(P["input"]):
            for L in layers:
                k = L["kind"]
                if k == "conv_in":
                    h = ws.alloc((M, L["cout"]), torch.float32)
                    self._gemm_gn(xin, L["w"], h, M=M, mode=GEMM_CONV3X3, bias=L["b"],
                                  conv=dict(Cin=CIN_PAD, Hi=H, Wi=W, Ho=H, Wo=W, stride=1, upsample=0),
                                  alg_flops_scale=u.in_channels / CIN_PAD)
                    ws.release(xin)
                elif k == "res":
                    hn = self._resblock(L, h, None, st)
                    if not (hs and hs[-1] is h):
                        ws.release(h)
                    h = hn
                elif k == "attn":
                    h = self._transformer(L, h, st)
                elif k == "down":
                    h = self._downsample(L, h, st, keep_input=True)
            hs.append(h)
            self._tap(f"input_blocks.{bi}", h, st)
        for L in P["middle"]:
            if L["kind"] == "res":
                hn = self._resblock(L, h, None, st)
                if not (hs and hs[-1] is h):
                    ws.release(h)
                h = hn
            else:
                h = self._transformer(L, h, st)
        self._tap("middle_block", h, st)
        for bi, layers in enumerate(P["output"]):
            skip = hs.pop()
            first = True
            for L in layers:
                k = L["kind"]
                if k == "res":
                    assert first
                    hn = self._resblock(L, h, skip, st)
                    ws.release(h, skip)
                    h = hn
                elif k == "attn":
                    h = self._transformer(L, h, st)
                elif k == "up":
                    h = self._upsample(L, h, st)
                first = False
            self._tap(f"output_blocks.{bi}", h, st)
        # out: GroupNorm32 -> SiLU -> Conv3x3 (video_model.py:455-459)
        Hc, Wc = st["H"], st["W"]
        assert (Hc, Wc) == (H, W)
        a16, _ = self._gn(h, None, H * W, 1e-5, P["out_gn"], True, False)
        ws.release(h)
        tok = ws.alloc((M, COUT_PAD), torch.float32)
        self._gemm(a16, P["out_w"], tok, M=M, mode=GEMM_CONV3X3, bias=P["out_b"],
                 conv=dict(Cin=u.model_channels, Hi=H, Wi=W, Ho=H, Wo=W, stride=1, upsample=0),
                 alg_flops_scale=u.out_channels / COUT_PAD)
        ws.release(a16)
        ops.unpack_output(tok, out_nchw, u.out_channels, N, H * W)
        ws.release(tok)
        ws.finish()
        return out_nchw

    def _tap(self, name, h, st):
        if self.taps is not None:
            N, H, W = st["N"], st["H"], st["W"]
            self.taps[name] = h.reshape(N, H, W, -1).permute(0, 3, 1, 2).float().clone()

    def _downsample(self, L, h, st, keep_input):
        ws, N = self.ws, st["N"]
        H, W = st["H"], st["W"]
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        a16 = ws.alloc((N * H * W, L["cin"]), torch.float16)
        ops.cast_f16(h, a16)
        out = ws.alloc((N * Ho * Wo, L["cout"]), torch.float32)
        self._gemm_gn(a16, L["w"], out, M=N * Ho * Wo, mode=GEMM_CONV3X3, bias=L["b"],
                      conv=dict(Cin=L["cin"], Hi=H, Wi=W, Ho=Ho, Wo=Wo, stride=2, upsample=0))
        ws.release(a16)
        if not keep_input:
            ws.release(h)
        st["H"], st["W"] = Ho, Wo
        return out

    def _upsample(self, L, h, st):
        ws, N = self.ws, st["N"]
        H, W = st["H"], st["W"]
        Ho, Wo = 2 * H, 2 * W
        a16 = ws.alloc((N * H * W, L["cin"]), torch.float16)
        ops.cast_f16(h, a16)
        ws.release(h)
        out = ws.alloc((N * Ho * Wo, L["cout"]), torch.float32)
        self._gemm_gn(a16, L["w"], out, M=N * Ho * Wo, mode=GEMM_CONV3X3, bias=L["b"],
                      conv=dict(Cin=L["cin"], Hi=H, Wi=W, Ho=Ho, Wo=Wo, stride=1, upsample=1))
        ws.release(a16)
        st["H"], st["W"] = Ho, Wo
        return out
