"""GPU: parity of the HIP first-stage Encoder (through the C ABI) with the CPU oracle and the committed
reference golden, and of the asymmetric-padding stride-2 conv it adds to gcd_gemm_f16.

Tolerances (rel-L2 vs the fp32 reference; fp16 MFMA operands, fp32 accumulation / residual stream):
moments and per-block activations <= 2e-3."""
from pathlib import Path

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_l2
from oracle import vae_encoder_ref as E, weights

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).resolve().parent / "golden"
TOL = 2e-3


def images(n, h, w, seed=6):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(n, 3, h, w, generator=g) * 2.0 - 1.0


def _sample(t, n=4096):
    f = t.reshape(-1)
    idx = torch.linspace(0, f.numel() - 1, min(n, f.numel())).long()
    return f[idx]


def _build(cfg, gpu, salt=2):
    from gcd_amd.ae_encoder import Encoder
    with torch.device("meta"):
        enc = Encoder(**cfg.as_reference_kwargs())
    sd = weights.synth_state_dict({k: tuple(v.shape) for k, v in enc.state_dict().items()}, salt)
    enc = enc.to_empty(device=gpu)
    enc.load_state_dict(sd)
    return enc.eval(), sd


@pytest.fixture(scope="module")
def tiny(gpu):
    return _build(E.TINY, gpu)


# (the general kernel walks K in 64-channel chunks: the 32-channel case is not one of its cases)
@pytest.mark.parametrize("impl,frames,C,H,W", [(i, *c) for i in (0, 1, 2) for c in ((2, 64, 8, 12), (3, 320, 32, 32), (1, 32, 6, 10))
                                               if not (i == 1 and c[1] % 64)])
def test_conv3x3_stride2_asymmetric_padding(gpu, impl, frames, C, H, W):
    from gcd_amd import ops, packing
    ops.tune_set(ops.TUNE_GEMM_IMPL, impl)
    try:
        g = torch.Generator().manual_seed(C + H)
        x = (torch.randn(frames, C, H, W, generator=g)).half().float()
        w = (torch.randn(C, C, 3, 3, generator=g) / (3.0 * C ** 0.5)).half().float()
        b = torch.randn(C, generator=g)
        ref = F.conv2d(F.pad(x, (0, 1, 0, 1)), w, b, stride=2)
        Ho, Wo = H // 2, W // 2
        assert ref.shape[-2:] == (Ho, Wo)
        a = x.permute(0, 2, 3, 1).contiguous().reshape(frames * H * W, C).half().to(gpu)
        out = torch.empty(frames * Ho * Wo, C, device=gpu)
        ops.gemm(a, packing.pack_conv3x3(w).to(gpu), out, M=frames * Ho * Wo, mode=ops.GEMM_CONV3X3,
                 bias=b.to(gpu), conv=dict(Cin=C, Hi=H, Wi=W, Ho=Ho, Wo=Wo, stride=2, upsample=0, asym_pad=1))
        torch.cuda.synchronize()
        e = rel_l2(out, ref.permute(0, 2, 3, 1).reshape(frames * Ho * Wo, C))
        assert e < 1e-4, f"asym conv {C}ch {H}x{W}: rel-L2 {e:.3e}"
    finally:
        ops.tune_set(ops.TUNE_GEMM_IMPL, 0)


def test_encoder_vs_reference_golden(gpu, tiny):
    enc, sd = tiny
    g = torch.load(GOLD / "encoder_tiny.pt")
    x = images(g["n"], g["h"], g["w"], g["input_seed"])
    enc.engine.taps = {}
    out = enc(x.to(gpu))
    torch.cuda.synchronize()
    taps, enc.engine.taps = enc.engine.taps, None
    assert out.shape == g["out"].shape and out.dtype == torch.float32
    errs = {k: rel_l2(_sample(v.cpu()), g["tap_samples"][k]) for k, v in taps.items()}
    assert set(errs) == set(g["tap_samples"])
    worst = max(errs, key=errs.get)
    e = rel_l2(out, g["out"])
    print(f"encoder_tiny vs reference golden: rel-L2 {e:.3e}; worst block {worst} {errs[worst]:.3e}")
    assert errs[worst] < TOL and e < TOL


@pytest.mark.parametrize("n,h,w", [(1, 32, 32), (14, 64, 96), (2, 72, 128)])
def test_encoder_vs_oracle(gpu, tiny, n, h, w):
    enc, sd = tiny
    x = images(n, h, w, seed=20 + n)
    with torch.no_grad():
        ref = E.encoder_forward(sd, E.TINY, x)
    out = enc(x.to(gpu))
    e = rel_l2(out, ref)
    print(f"encoder tiny {n}x{h}x{w}: rel-L2 {e:.3e}")
    assert e < TOL


def test_full_width_encoder_and_mode(gpu):
    """The real 128-channel encoder on two 128x192 images, then quant_conv + mode as
    AutoencoderKLModeOnly.encode does (the latent the conditioner concatenates to the noise)."""
    from gcd_amd.ae_encoder import encode_mode
    enc, sd = _build(E.KUBRIC, gpu, salt=4)
    x = images(2, 128, 192, seed=9)
    quant = torch.nn.Conv2d(8, 8, 1)
    with torch.no_grad():
        ref = E.encode_mode(sd, E.KUBRIC, x, quant.weight, quant.bias)
    z = encode_mode(enc, x.to(gpu), quant.to(gpu))
    e = rel_l2(z, ref)
    print(f"encoder kubric 2x128x192 -> z: rel-L2 {e:.3e}")
    assert z.shape == (2, 4, 16, 24) and e < TOL
    assert torch.equal(enc(x.to(gpu)), enc(x.to(gpu)))       # bit-reproducible
    assert enc(x.half().to(gpu)).dtype == torch.float16
    with pytest.raises(ValueError):
        enc(x[..., :100].to(gpu))


def test_video_prediction_embedder_with_hip_encoder(gpu):
    """The conditioner socket that produces cond['concat']: VideoPredictionEmbedderWithEncoder over the
    HIP AutoencoderKLModeOnly (config of infer_kubric.yaml:69-101 at tiny width) vs the oracle."""
    from gcd_amd.conditioning import VideoPredictionEmbedderWithEncoder
    dd = dict(E.TINY.as_reference_kwargs(), attn_type="vanilla-xformers")
    emb = VideoPredictionEmbedderWithEncoder(
        n_cond_frames=1, n_copies=1, is_ae=True, disable_encoder_autocast=True,
        en_and_decode_n_samples_a_time=2, scale_factor=0.18215,
        encoder_config={"target": "gcd_amd.ae_encoder.AutoencoderKLModeOnly",
                        "params": {"embed_dim": 4, "monitor": "val/rec_loss", "ddconfig": dd,
                                   "lossconfig": {"target": "torch.nn.Identity"}}})
    shapes = {k: tuple(v.shape) for k, v in emb.state_dict().items()}
    assert len(shapes) == 106 + 4 and "encoder.quant_conv.weight" in shapes
    sd = weights.synth_state_dict(shapes, salt=6)
    emb.load_state_dict(sd)
    emb = emb.to(gpu).eval()
    vid = images(5, 64, 96, seed=33)                    # 5 frames, chunks of 2 + 2 + 1
    enc_sd = {k[len("encoder.encoder."):]: v for k, v in sd.items() if k.startswith("encoder.encoder.")}
    with torch.no_grad():
        ref = E.encode_mode(enc_sd, E.TINY, vid, sd["encoder.quant_conv.weight"],
                            sd["encoder.quant_conv.bias"]) * 0.18215
    out = emb(vid.to(gpu))
    e = rel_l2(out, ref)
    print(f"VideoPredictionEmbedderWithEncoder: rel-L2 {e:.3e}")
    assert out.shape == (5, 4, 8, 12) and e < TOL
