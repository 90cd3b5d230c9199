"""CPU: pin the decoder oracle (oracle/vae_decoder_ref.py) against the golden tensors produced by the
reference's own VideoDecoder (oracle/make_golden_decoder.py), and check the host side of the HIP
drop-in (parameter names / shapes, error behaviour) without a GPU.  Bar: fp32 round-off."""
from pathlib import Path

import pytest
import torch

from conftest import rel_l2
from oracle import vae_decoder_ref as D, weights

GOLD = Path(__file__).resolve().parent / "golden"


def latents(clips, T, h, w, zc=4, seed=5):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(clips * T, zc, h, w, generator=g) * 1.5


def _sample(t, n=4096):
    f = t.reshape(-1)
    idx = torch.linspace(0, f.numel() - 1, min(n, f.numel())).long()
    return f[idx]


@pytest.fixture(scope="module")
def gold():
    return torch.load(GOLD / "decoder_tiny.pt")


def test_decoder_oracle_matches_reference_golden(gold):
    g = gold
    sd = weights.synth_state_dict(g["state_dict_shapes"], salt=g["weight_salt"])
    z = latents(g["clips"], g["T"], g["h"], g["w"], seed=g["input_seed"])
    taps = {}
    with torch.no_grad():
        out = D.decoder_forward(sd, D.TINY, z, g["T"], taps=taps)
    assert float(g["out"].std()) > 0.1, "golden output is degenerate"
    assert out.shape == g["out"].shape == (g["clips"] * g["T"], 3, 8 * g["h"], 8 * g["w"])
    assert rel_l2(out, g["out"]) < 2e-5
    assert set(taps) == set(g["tap_samples"]) and len(taps) == 19
    for k, v in taps.items():
        assert tuple(v.shape) == g["tap_shapes"][k]
        assert rel_l2(_sample(v), g["tap_samples"][k]) < 2e-5, k
        assert abs(float(v.double().norm()) / g["tap_norms"][k] - 1) < 2e-5, k


def test_time_mixing_really_mixes_frames(gold):
    """The decoder is not frame-wise: decoding the 2 clips as one 6-frame clip differs (guards the
    oracle, and through it the product, against silently dropping the (3,1,1) convolutions)."""
    g = gold
    sd = weights.synth_state_dict(g["state_dict_shapes"], salt=g["weight_salt"])
    z = latents(g["clips"], g["T"], g["h"], g["w"], seed=g["input_seed"])
    with torch.no_grad():
        a = D.decoder_forward(sd, D.TINY, z, g["T"])
        b = D.decoder_forward(sd, D.TINY, z, g["clips"] * g["T"])
        c = D.decode_first_stage(sd, D.TINY, z * 0.18215, 0.18215, n_samples=g["T"])
    assert rel_l2(b, a) > 1e-2
    assert rel_l2(c, a) < 1e-5     # chunked decode == per-clip decode


def test_product_decoder_has_the_reference_parameters(gold):
    from gcd_amd.temporal_ae import VideoDecoder
    dec = VideoDecoder(**D.TINY.as_reference_kwargs())
    mine = {k: tuple(v.shape) for k, v in dec.state_dict().items()}
    assert list(mine) == list(gold["state_dict_shapes"])
    assert mine == gold["state_dict_shapes"]
    sd = weights.synth_state_dict(mine, salt=1)
    assert not any(dec.load_state_dict(sd))           # no missing / unexpected keys
    # the merge factor is the sigmoid of the learned logit (temporal_ae.py:55-61)
    rb = dec.mid.block_1
    assert abs(rb.alpha() - float(torch.sigmoid(sd["mid.block_1.mix_factor"]))) < 1e-7
    full = VideoDecoder(**D.KUBRIC.as_reference_kwargs())
    assert sum(p.numel() for p in full.parameters()) > 60e6      # the 128-channel SVD decoder


def test_product_decoder_has_no_cpu_path_and_rejects_other_modes():
    from gcd_amd import _lib
    from gcd_amd.first_stage import decode_first_stage
    from gcd_amd.temporal_ae import VideoDecoder
    kw = D.TINY.as_reference_kwargs()
    dec = VideoDecoder(**kw)
    with pytest.raises(_lib.GcdError, match="no CPU"):
        dec(torch.zeros(3, 4, 8, 8), timesteps=3)
    with pytest.raises(_lib.GcdError, match="no CPU"):
        decode_first_stage(dec, torch.zeros(3, 4, 8, 8))
    with pytest.raises(_lib.GcdError, match="no CPU"):     # timesteps defaults to the chunk length
        dec(torch.zeros(3, 4, 8, 8))                       # (config-only drop-in, diffusion.py:242-245)
    with pytest.raises(NotImplementedError):
        VideoDecoder(**dict(kw, time_mode="all"))
    with pytest.raises(AssertionError):
        VideoDecoder(**dict(kw, time_mode="bogus"))
    with pytest.raises(NotImplementedError):
        VideoDecoder(**dict(kw, video_kernel_size=3))


@pytest.mark.skipif(not Path("/root/reference/gcd-model").exists(), reason="reference tree not mounted")
def test_golden_is_reproducible_from_the_reference(gold):
    """Build container only: the committed fixture is what the reference class computes today."""
    from oracle.make_golden_decoder import reference_decoder_class
    cls = reference_decoder_class()
    dec = cls(**D.TINY.as_reference_kwargs()).eval()
    dec.load_state_dict(weights.synth_state_dict(gold["state_dict_shapes"], salt=gold["weight_salt"]))
    z = latents(gold["clips"], gold["T"], gold["h"], gold["w"], seed=gold["input_seed"])
    with torch.no_grad():
        out = dec(z, timesteps=gold["T"])
    assert rel_l2(out, gold["out"]) < 1e-6
