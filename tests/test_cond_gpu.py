"""GPU: the HIP-backed conditioning producers through the C ABI against the
golden made by the reference classes, and the camera-pose injection end to end:
(d_azimuth, d_elevation, d_radius) -> SphericalEmbedder -> GeneralConditioner `vector` ->
`aux_label_emb` -> VideoUNet output, against the oracle."""
from pathlib import Path

import pytest
import torch

from conftest import rel_l2
from oracle import svd_unet_ref as O, weights

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).resolve().parent / "golden" / "cond_tiny.pt"
P = "gcd_amd.conditioning."


@pytest.fixture(scope="module")
def g():
    return torch.load(GOLD)


def _batch(gpu, g):
    from oracle.make_golden_cond import inputs
    b = {k: v.to(gpu) for k, v in inputs().items()}
    b["scaled_relative_angles"] = g["trajectories"]["gradual_linear"]["rel"].to(gpu)
    return b


def _kubric_conditioner(gpu):
    from gcd_amd.conditioning import GeneralConditioner
    cond = GeneralConditioner([
        dict(input_key="fps_id", target=P + "ConcatTimestepEmbedderND", params=dict(outdim=256)),
        dict(input_key="motion_bucket_id", is_trainable=True, target=P + "ConcatTimestepEmbedderND", params=dict(outdim=256)),
        dict(input_key="cond_frames_without_noise", target=P + "IdentityEncoder"),
        dict(input_key="cond_frames", target=P + "IdentityEncoder"),
        dict(input_key="cond_aug", target=P + "ConcatTimestepEmbedderND", params=dict(outdim=256)),
        dict(input_key="scaled_relative_angles", is_trainable=True, target=P + "SphericalEmbedder",
             params=dict(embed_dim=128, zero_init=False))]).to(gpu)
    cond.load_state_dict({
        "embedders.5.proj.weight": weights.synth_tensor("conditioner.embedders.5.proj.weight", (128, 13)),
        "embedders.5.proj.bias": weights.synth_tensor("conditioner.embedders.5.proj.bias", (128,))})
    return cond


def test_embedders_vs_reference_golden(gpu, g):
    from gcd_amd.conditioning import CameraEmbedder, ConcatTimestepEmbedderND
    cond = _kubric_conditioner(gpu)
    sph = cond.embedders[5]
    for name, t in g["trajectories"].items():
        e = rel_l2(sph(t["rel"].to(gpu)), g["spherical"][name])
        assert e < 2e-6, f"SphericalEmbedder {name}: {e:.2e}"
    cam = CameraEmbedder(embed_dim=128).to(gpu)
    cam.load_state_dict({"proj.weight": weights.synth_tensor("camera.proj.weight", (128, 12)),
                         "proj.bias": weights.synth_tensor("camera.proj.bias", (128,))})
    b = _batch(gpu, g)
    assert rel_l2(cam(b["scaled_relative_pose"]), g["camera"]) < 2e-6
    v = torch.stack([b["fps_id"], b["motion_bucket_id"], b["cond_aug"]], 1)
    e = rel_l2(ConcatTimestepEmbedderND(256)(v), g["timestep_nd"])
    assert e < 2e-5, f"ConcatTimestepEmbedderND: {e:.2e}"      # sin/cos of arguments up to 127 in fp32
    with pytest.raises(Exception):                               # CPU tensors: no CPU path
        sph(g["trajectories"]["direct"]["rel"])


def test_trainable_pose_embedder_receives_gradients(gpu, g):
    """The Kubric configs train SphericalEmbedder.proj (`is_trainable: True`): the embedding must stay in the autograd graph
    (the projection used to run on detached weights, so a pose-embedder fine-tune would silently never learn).
    Forward = the HIP kernel as before; d proj.weight / d proj.bias against torch's own Linear on the same features."""
    cond = _kubric_conditioner(gpu)
    sph = cond.embedders[5]
    sph.proj.weight.requires_grad_(True)
    sph.proj.bias.requires_grad_(True)
    t = next(iter(g["trajectories"].values()))["rel"].to(gpu)
    out = sph(t)
    assert out.requires_grad
    wsum = torch.randn(out.shape, device=gpu, generator=torch.Generator(device=gpu).manual_seed(3))
    (out * wsum).sum().backward()
    gw, gb = sph.proj.weight.grad.clone(), sph.proj.bias.grad.clone()
    # torch reference: the features proj() is applied to = the pseudo-inverse of the forward through a second Linear
    ref = torch.nn.Linear(sph.proj.in_features, sph.proj.out_features).to(gpu)
    ref.load_state_dict(sph.proj.state_dict())
    feats = torch.linalg.lstsq(sph.proj.weight.detach().double(),
                               (out.detach().double() - sph.proj.bias.detach().double()).reshape(-1, out.shape[-1]).t()).solution
    o2 = ref(feats.t().float()).reshape(out.shape)
    assert rel_l2(o2, out.detach()) < 1e-5
    (o2 * wsum).sum().backward()
    assert rel_l2(gw, ref.weight.grad) < 1e-4 and rel_l2(gb, ref.bias.grad) < 1e-5
    with torch.no_grad():                      # inference: no graph node, same values
        assert torch.equal(sph(t), out.detach())


def test_general_conditioner_vs_reference_golden(gpu, g):
    cond = _kubric_conditioner(gpu)
    b = _batch(gpu, g)
    c, uc = cond.get_unconditional_conditioning(
        b, batch_uc=b, force_uc_zero_embeddings=["cond_frames", "cond_frames_without_noise"])
    for k in ("vector", "crossattn", "concat"):
        assert tuple(c[k].shape) == tuple(g["c"][k].shape) and c[k].is_cuda
        assert rel_l2(c[k], g["c"][k]) < 2e-5, k
    assert rel_l2(uc["vector"], g["uc"]["vector"]) < 2e-5
    assert float(uc["crossattn"].abs().max()) == 0.0 and float(uc["concat"].abs().max()) == 0.0


def test_camera_pose_reaches_unet_output(gpu):
    """(d_az, d_el, d_r) -> conditioner -> y -> aux_label_emb -> UNet output, vs the oracle doing the
    same from the angles; two different destinations must give different outputs."""
    from gcd_amd.camera import scaled_relative_angles
    from gcd_amd.conditioning import GeneralConditioner
    from gcd_amd.video_model import VideoUNet
    cfg = O.UNetConfig(model_channels=64, context_dim=64, adm_in_channels=96, aux_emb_dim=32)
    with torch.device("meta"):
        net = VideoUNet(**cfg.as_reference_kwargs())
    sd = weights.synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, salt=4)
    net = net.to_empty(device=gpu)
    net.load_state_dict(sd)
    net.eval()
    cond = GeneralConditioner([
        dict(input_key="fps_id", target=P + "ConcatTimestepEmbedderND", params=dict(outdim=32)),
        dict(input_key="motion_bucket_id", target=P + "ConcatTimestepEmbedderND", params=dict(outdim=32)),
        dict(input_key="cond_aug", target=P + "ConcatTimestepEmbedderND", params=dict(outdim=32)),
        dict(input_key="scaled_relative_angles", is_trainable=True, target=P + "SphericalEmbedder",
             params=dict(embed_dim=32))]).to(gpu)
    w = weights.synth_tensor("pose.proj.weight", (32, 13))
    b = weights.synth_tensor("pose.proj.bias", (32,))
    cond.load_state_dict({"embedders.3.proj.weight": w, "embedders.3.proj.bias": b})
    T, h, wd = 14, 8, 8
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(T, 8, h, wd, generator=gen)
    ts = torch.linspace(-1.0, 1.5, T)
    ctx = torch.randn(T, 1, 64, generator=gen)
    ioi = torch.zeros(1, T)
    outs = []
    for dest in ((30.0, 15.0, 1.0), (-80.0, 5.0, -2.0)):
        rel = scaled_relative_angles(*dest, num_frames=T)
        batch = {"fps_id": torch.full((T,), 12.0), "motion_bucket_id": torch.full((T,), 127.0),
                 "cond_aug": torch.full((T,), 0.02), "scaled_relative_angles": rel}
        y = cond({k: v.to(gpu) for k, v in batch.items()})["vector"]
        assert y.shape == (T, 128)
        y_ref = torch.cat([O.concat_timestep_embed(batch["fps_id"], 32),
                           O.concat_timestep_embed(batch["motion_bucket_id"], 32),
                           O.concat_timestep_embed(batch["cond_aug"], 32),
                           O.spherical_embed(O.scaled_relative_angles(*dest, T=T), w, b)], 1)
        assert rel_l2(y, y_ref) < 2e-5
        out = net(x.to(gpu), ts.to(gpu), context=ctx.to(gpu), y=y, num_video_frames=T,
                  image_only_indicator=ioi.to(gpu))
        with torch.no_grad():
            ref = O.unet_forward(sd, cfg, x, ts, ctx, y_ref, T, ioi)
        e = rel_l2(out, ref)
        assert e < 2e-3, f"pose {dest}: UNet rel-L2 {e:.3e}"
        outs.append(out)
    assert rel_l2(outs[0], outs[1]) > 1e-2, "the camera pose does not reach the UNet output"
