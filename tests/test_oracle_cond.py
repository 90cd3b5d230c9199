"""CPU: the conditioning producers of SURVEY.md §8a a22 — oracle restatements and the host-side
assembly of gcd_amd.conditioning.GeneralConditioner / gcd_amd.camera against the golden made by the
unmodified reference classes (oracle/make_golden_cond.py -> tests/golden/cond_tiny.pt)."""
from pathlib import Path

import pytest
import torch

from conftest import rel_l2
from oracle import svd_unet_ref as O, weights

GOLD = Path(__file__).resolve().parent / "golden" / "cond_tiny.pt"


@pytest.fixture(scope="module")
def g():
    return torch.load(GOLD)


def _batch():
    from oracle.make_golden_cond import inputs
    return inputs()


def test_trajectories_oracle_and_product(g):
    from gcd_amd import camera
    for name, t in g["trajectories"].items():
        az, el, r = t["end"]
        src, dst = O.construct_trajectory([0, 0, 0], t["end"], t["kind"], g["T"], t["move_time"])
        assert torch.equal(src, t["src"]) and torch.allclose(dst, t["dst"], rtol=1e-6, atol=1e-6), name
        rel = O.scaled_relative_angles(az, el, r, g["T"], t["kind"], t["move_time"])
        assert torch.allclose(rel, t["rel"], rtol=1e-6, atol=1e-6), name
        ps, pd = camera.construct_trajectory([0, 0, 0], t["end"], t["kind"], g["T"], t["move_time"])
        assert torch.equal(torch.tensor(pd), t["dst"]) and torch.equal(torch.tensor(ps), t["src"]), name
        prel = camera.scaled_relative_angles(az, el, r, g["T"], t["kind"], t["move_time"])
        assert torch.equal(prel, t["rel"]), name
    with pytest.raises(ValueError):
        camera.construct_trajectory([0, 0, 0], [1, 1, 1], "spline", 14, 13)


def test_oracle_embedders_vs_reference_golden(g):
    w = weights.synth_tensor("conditioner.embedders.5.proj.weight", (128, 13))
    b = weights.synth_tensor("conditioner.embedders.5.proj.bias", (128,))
    for name, t in g["trajectories"].items():
        e = rel_l2(O.spherical_embed(t["rel"], w, b), g["spherical"][name])
        assert e < 1e-6, f"spherical {name}: {e:.2e}"
    cw = weights.synth_tensor("camera.proj.weight", (128, 12))
    cb = weights.synth_tensor("camera.proj.bias", (128,))
    batch = _batch()
    assert rel_l2(O.camera_embed(batch["scaled_relative_pose"], cw, cb), g["camera"]) < 1e-6
    v = torch.stack([batch["fps_id"], batch["motion_bucket_id"], batch["cond_aug"]], 1)
    assert rel_l2(O.concat_timestep_embed(v, 256), g["timestep_nd"]) < 1e-6


def test_oracle_general_conditioner_vs_reference_golden(g):
    batch = _batch()
    rel = g["trajectories"]["gradual_linear"]["rel"]
    w = weights.synth_tensor("conditioner.embedders.5.proj.weight", (128, 13))
    b = weights.synth_tensor("conditioner.embedders.5.proj.bias", (128,))
    embedded = [("fps_id", O.concat_timestep_embed(batch["fps_id"])),
                ("motion_bucket_id", O.concat_timestep_embed(batch["motion_bucket_id"])),
                ("cond_frames_without_noise", batch["cond_frames_without_noise"]),
                ("cond_frames", batch["cond_frames"]),
                ("cond_aug", O.concat_timestep_embed(batch["cond_aug"])),
                ("scaled_relative_angles", O.spherical_embed(rel, w, b))]
    c = O.general_conditioner(embedded)
    uc = O.general_conditioner(embedded, force_zero=("cond_frames", "cond_frames_without_noise"))
    for k in ("vector", "crossattn", "concat"):
        assert c[k].shape == g["c"][k].shape
        assert rel_l2(c[k], g["c"][k]) < 1e-6, k
    assert rel_l2(uc["vector"], g["uc"]["vector"]) < 1e-6
    assert float(uc["crossattn"].abs().max()) == 0.0 == float(g["uc"]["crossattn"].abs().max())
    assert float(uc["concat"].abs().max()) == 0.0 == float(g["uc"]["concat"].abs().max())


def test_general_conditioner_host_assembly(g):
    """The assembly logic of the product class with embedders that need no kernel (IdentityEncoder):
    key routing by rank, concatenation order, force-zero, ucg restore, attribute surface, errors."""
    from gcd_amd.conditioning import AbstractEmbModel, GeneralConditioner
    I = "gcd_amd.conditioning.IdentityEncoder"
    cond = GeneralConditioner([dict(input_key="a", target=I), dict(input_key="b", target=I, ucg_rate=0.5),
                               dict(input_key="x", target=I), dict(input_key="f", target=I),
                               dict(input_key="f2", target=I, is_trainable=True)])
    gen = torch.Generator().manual_seed(1)
    batch = {"a": torch.randn(6, 3, generator=gen), "b": torch.randn(6, 2, generator=gen),
             "x": torch.randn(6, 1, 5, generator=gen), "f": torch.randn(6, 4, 2, 2, generator=gen),
             "f2": torch.randn(6, 1, 2, 2, generator=gen)}
    c, uc = cond.get_unconditional_conditioning(batch, batch_uc=batch, force_uc_zero_embeddings=["x", "f"])
    assert torch.equal(c["vector"], torch.cat([batch["a"], batch["b"]], 1))     # ucg off inside
    assert torch.equal(c["crossattn"], batch["x"])
    assert torch.equal(c["concat"], torch.cat([batch["f"], batch["f2"]], 1))
    assert float(uc["crossattn"].abs().max()) == 0.0
    assert torch.equal(uc["concat"], torch.cat([torch.zeros_like(batch["f"]), batch["f2"]], 1))
    assert cond.embedders[1].ucg_rate == 0.5                                     # restored
    torch.manual_seed(0)
    dropped = cond(batch)["vector"][:, 3:]                                       # ucg on: rows zeroed
    rows = dropped.abs().sum(1) == 0
    assert 0 < int(rows.sum()) < 6 and torch.equal(dropped[~rows], batch["b"][~rows])
    assert cond.embedders[0].train() is cond.embedders[0]            # frozen: train() is a no-op
    assert not any(p.requires_grad for p in cond.embedders[0].parameters())
    assert all(isinstance(e, AbstractEmbModel) for e in cond.embedders)
    with pytest.raises(KeyError):
        GeneralConditioner([dict(target=I)])
    with pytest.raises(AssertionError):
        GeneralConditioner([dict(input_key="a", target="torch.nn.Identity")])


def test_kubric_conditioner_state_dict_keys(g):
    """`conditioner.embedders.5.proj.*` of the GCD checkpoints must land in the drop-in unchanged."""
    from gcd_amd.conditioning import GeneralConditioner
    P = "gcd_amd.conditioning."
    cond = GeneralConditioner([
        dict(input_key="fps_id", target=P + "ConcatTimestepEmbedderND", params=dict(outdim=256)),
        dict(input_key="motion_bucket_id", is_trainable=True, target=P + "ConcatTimestepEmbedderND", params=dict(outdim=256)),
        dict(input_key="cond_frames_without_noise", target=P + "IdentityEncoder"),
        dict(input_key="cond_frames", target=P + "IdentityEncoder"),
        dict(input_key="cond_aug", target=P + "ConcatTimestepEmbedderND", params=dict(outdim=256)),
        dict(input_key="scaled_relative_angles", is_trainable=True, target=P + "SphericalEmbedder",
             params=dict(embed_dim=128, zero_init=False))])
    assert {k: tuple(v.shape) for k, v in cond.state_dict().items()} == g["state_dict_keys"]
    assert [p.requires_grad for p in cond.embedders[5].parameters()] == [True, True]


def test_oracle_cond_matches_live_reference(g):
    """In the build container (reference mounted) re-run the reference classes and compare with the
    committed golden: guards the fixture against drift."""
    from oracle import ref_shim
    if not ref_shim.available():
        pytest.skip("reference tree not mounted (GPU box)")
    from oracle import make_golden_cond as M
    mod = M.reference_encoder_module()
    sph = mod.SphericalEmbedder(embed_dim=128)
    sph.proj.weight.data.copy_(weights.synth_tensor("conditioner.embedders.5.proj.weight", (128, 13)))
    sph.proj.bias.data.copy_(weights.synth_tensor("conditioner.embedders.5.proj.bias", (128,)))
    with torch.no_grad():
        for name, t in g["trajectories"].items():
            assert torch.equal(sph(t["rel"]), g["spherical"][name])
