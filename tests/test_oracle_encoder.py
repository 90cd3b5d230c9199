"""CPU: pin the encoder oracle (oracle/vae_encoder_ref.py) against the golden tensors produced by the
reference's own Encoder (oracle/make_golden_encoder.py), and check the host side of the HIP drop-in
without a GPU.  Bar: fp32 round-off."""
from pathlib import Path

import pytest
import torch

from conftest import rel_l2
from oracle import vae_encoder_ref as E, weights

GOLD = Path(__file__).resolve().parent / "golden"


def images(n, h, w, seed=6):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(n, 3, h, w, generator=g) * 2.0 - 1.0


def _sample(t, n=4096):
    f = t.reshape(-1)
    idx = torch.linspace(0, f.numel() - 1, min(n, f.numel())).long()
    return f[idx]


@pytest.fixture(scope="module")
def gold():
    return torch.load(GOLD / "encoder_tiny.pt")


def test_encoder_oracle_matches_reference_golden(gold):
    g = gold
    sd = weights.synth_state_dict(g["state_dict_shapes"], salt=g["weight_salt"])
    x = images(g["n"], g["h"], g["w"], g["input_seed"])
    taps = {}
    with torch.no_grad():
        out = E.encoder_forward(sd, E.TINY, x, taps=taps)
    assert float(g["out"].std()) > 0.1, "golden output is degenerate"
    assert out.shape == g["out"].shape == (g["n"], 8, g["h"] // 8, g["w"] // 8)
    assert rel_l2(out, g["out"]) < 2e-5
    assert set(taps) == set(g["tap_samples"]) and len(taps) == 15
    for k, v in taps.items():
        assert tuple(v.shape) == g["tap_shapes"][k]
        assert rel_l2(_sample(v), g["tap_samples"][k]) < 2e-5, k
        assert abs(float(v.double().norm()) / g["tap_norms"][k] - 1) < 2e-5, k
    # mode of the posterior = mean half of the moments (after an optional 1x1 quant_conv)
    qw = torch.randn(8, 8, 1, 1, generator=torch.Generator().manual_seed(1)) * 0.3
    qb = torch.zeros(8)
    z = E.encode_mode(sd, E.TINY, x, qw, qb)
    assert z.shape == (g["n"], 4, g["h"] // 8, g["w"] // 8)
    assert rel_l2(z, torch.nn.functional.conv2d(out, qw, qb)[:, :4]) < 1e-6


def test_downsample_padding_is_asymmetric(gold):
    """(0,1,0,1) padding, not padding 1: the oracle's Downsample differs from a symmetric conv (guards
    the asym_pad geometry of the product against a silent fallback to the UNet's stride-2 conv)."""
    sd = weights.synth_state_dict(gold["state_dict_shapes"], salt=gold["weight_salt"])
    g = torch.Generator().manual_seed(3)
    h = torch.randn(1, 32, 8, 8, generator=g)
    w, b = sd["down.0.downsample.conv.weight"], sd["down.0.downsample.conv.bias"]
    asym = torch.nn.functional.conv2d(torch.nn.functional.pad(h, (0, 1, 0, 1)), w, b, stride=2)
    sym = torch.nn.functional.conv2d(h, w, b, stride=2, padding=1)
    assert asym.shape == sym.shape == (1, 32, 4, 4) and rel_l2(asym, sym) > 0.1


def test_product_encoder_has_the_reference_parameters(gold):
    from gcd_amd import _lib
    from gcd_amd.ae_encoder import Encoder, encode_mode
    enc = Encoder(**E.TINY.as_reference_kwargs())
    mine = {k: tuple(v.shape) for k, v in enc.state_dict().items()}
    assert list(mine) == list(gold["state_dict_shapes"]) and mine == gold["state_dict_shapes"]
    assert not any(enc.load_state_dict(weights.synth_state_dict(mine, salt=2)))
    with pytest.raises(_lib.GcdError, match="no CPU"):
        enc(torch.zeros(1, 3, 64, 64))
    with pytest.raises(_lib.GcdError, match="no CPU"):
        encode_mode(enc, torch.zeros(1, 3, 64, 64))
    kw = E.TINY.as_reference_kwargs()
    with pytest.raises(NotImplementedError):
        Encoder(**dict(kw, attn_resolutions=[16]))
    with pytest.raises(NotImplementedError):
        Encoder(**dict(kw, resamp_with_conv=False))


@pytest.mark.skipif(not Path("/root/reference/gcd-model").exists(), reason="reference tree not mounted")
def test_golden_is_reproducible_from_the_reference(gold):
    from oracle.make_golden_encoder import reference_encoder_class
    enc = reference_encoder_class()(**E.TINY.as_reference_kwargs()).eval()
    enc.load_state_dict(weights.synth_state_dict(gold["state_dict_shapes"], salt=gold["weight_salt"]))
    with torch.no_grad():
        out = enc(images(gold["n"], gold["h"], gold["w"], gold["input_seed"]))
    assert rel_l2(out, gold["out"]) < 1e-6


def test_video_prediction_embedder_host_logic_matches_the_reference_rearranges():
    """VideoPredictionEmbedderWithEncoder.forward (encoders/modules.py:1071-1114) with a stub encoder:
    chunking by en_and_decode_n_samples_a_time, scale_factor, and the two einops patterns
    "(b t) c h w -> b () (t c) h w" / "b 1 c h w -> (b t) c h w" restated with reshape / expand."""
    from einops import rearrange, repeat
    from gcd_amd.conditioning import VideoPredictionEmbedderWithEncoder

    class Stub(torch.nn.Module):
        calls = []

        def encode(self, x):
            Stub.calls.append(x.shape[0])
            return x[:, :2, ::2, ::2] * 3.0 + 1.0

    g = torch.Generator().manual_seed(4)
    for n_cond, n_copies, bsz, chunk in [(1, 1, 4, 2), (2, 3, 3, 4), (1, 5, 2, None)]:
        Stub.calls = []
        emb = VideoPredictionEmbedderWithEncoder(
            n_cond_frames=n_cond, n_copies=n_copies, is_ae=True, scale_factor=0.18215,
            en_and_decode_n_samples_a_time=chunk,
            encoder_config={"target": "torch.nn.Identity"})
        emb.encoder = Stub()
        vid = torch.randn(bsz * n_cond, 3, 8, 12, generator=g)
        out = emb(vid)
        ref = Stub().encode(vid) * 0.18215
        ref = rearrange(ref, "(b t) c h w -> b () (t c) h w", t=n_cond)
        ref = repeat(ref, "b 1 c h w -> (b t) c h w", t=n_copies)
        assert out.shape == ref.shape == (bsz * n_copies, 2 * n_cond, 4, 6)
        assert torch.allclose(out, ref, atol=1e-6)
        want = [vid.shape[0]] if chunk is None else [min(chunk, vid.shape[0] - i) for i in range(0, vid.shape[0], chunk)]
        assert Stub.calls[:len(want)] == want
