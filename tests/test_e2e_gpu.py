"""GPU: the whole inference tail the north_star names — EulerEDM + CFG sampling loop on the HIP
VideoUNet, then the HIP first-stage VideoDecoder — against the CPU oracles on identical noise and
inputs: latents within the loop's rel-L2 contract and **PSNR-equivalent decoded frames**.

PSNR is taken on the [-1, 1] pixel range the decoder emits (peak-to-peak 2): PSNR = 10 log10(4 / MSE)
between the frames decoded from the HIP latents by the HIP decoder and the frames decoded from the
oracle latents by the oracle decoder.  Bar: >= 55 dB (measured ~65 dB; the reference's own fp16
autocast path differs from its fp32 path by about as much)."""
import math

import pytest
import torch

from conftest import rel_l2
from oracle import svd_unet_ref as O, vae_decoder_ref as D, weights

pytestmark = pytest.mark.gpu


def _psnr(a, b):
    mse = float(((a.double().cpu() - b.double().cpu()) ** 2).mean())
    return 10.0 * math.log10(4.0 / max(mse, 1e-30))


def _unet(gpu):
    from gcd_amd.video_model import VideoUNet
    with torch.device("meta"):
        net = VideoUNet(**O.TINY.as_reference_kwargs())
    sd = weights.synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()})
    net = net.to_empty(device=gpu)
    net.load_state_dict(sd)
    return net.eval(), sd


def _decoder(gpu):
    from gcd_amd.temporal_ae import VideoDecoder
    with torch.device("meta"):
        dec = VideoDecoder(**D.TINY.as_reference_kwargs())
    sd = weights.synth_state_dict({k: tuple(v.shape) for k, v in dec.state_dict().items()}, salt=1)
    dec = dec.to_empty(device=gpu)
    dec.load_state_dict(sd)
    return dec.eval(), sd


@pytest.mark.parametrize("T,steps,h,w", [(4, 5, 8, 8), (14, 25, 8, 16)])
def test_sample_then_decode_frames_psnr(gpu, T, steps, h, w):
    from gcd_amd.denoiser import Denoiser
    from gcd_amd.first_stage import decode_first_stage
    from gcd_amd.sampling import EulerEDMSampler, FusedDenoiser
    from gcd_amd.wrappers import OpenAIWrapper
    net, sd_u = _unet(gpu)
    dec, sd_d = _decoder(gpu)
    noise, c, uc = weights.synth_inputs(1, T, h, w, O.TINY.context_dim,
                                        O.TINY.adm_in_channels + O.TINY.aux_emb_dim, 81 + T)
    # ---- oracle: loop (fp32, CPU) then decode, exactly DiffusionEngine.sample_video's tail ----
    with torch.no_grad():
        z_ref = O.sample_loop(sd_u, O.TINY, noise, c, uc, T, steps)
        frames_ref = D.decode_first_stage(sd_d, D.TINY, z_ref, 0.18215, n_samples=T)
    # ---- product ----
    den = Denoiser({"target": "gcd_amd.denoiser_scaling.VScalingWithEDMcNoise"})
    extra = {"num_video_frames": T, "image_only_indicator": torch.zeros(2, T, device=gpu)}
    fused = FusedDenoiser(den, OpenAIWrapper(net), **extra)
    sampler = EulerEDMSampler(
        discretization_config={"target": "gcd_amd.discretizer.EDMDiscretization", "params": {"sigma_max": 700.0}},
        num_steps=steps,
        guider_config={"target": "gcd_amd.guiders.LinearPredictionGuider",
                       "params": {"num_frames": T, "max_scale": 1.5, "min_scale": 1.0}},
        device="cuda")
    z = sampler(fused, noise.clone().to(gpu), cond={k: v.to(gpu) for k, v in c.items()},
                uc={k: v.to(gpu) for k, v in uc.items()})
    frames = decode_first_stage(dec, z, 0.18215, en_and_decode_n_samples_a_time=T)
    torch.cuda.synchronize()
    assert frames.shape == frames_ref.shape == (T, 3, 8 * h, 8 * w)
    ez, ef, psnr = rel_l2(z, z_ref), rel_l2(frames, frames_ref), _psnr(frames, frames_ref)
    print(f"T={T} steps={steps}: latents rel-L2 {ez:.3e}, frames rel-L2 {ef:.3e}, PSNR {psnr:.1f} dB")
    assert ez < (1e-3 if steps >= 25 else 1.5e-3)
    assert psnr >= 55.0 and ef < 4e-3
    # the decoder's own share: same (oracle) latents through both decoders
    frames_same_z = decode_first_stage(dec, z_ref.to(gpu), 0.18215, en_and_decode_n_samples_a_time=T)
    assert _psnr(frames_same_z, frames_ref) >= 58.0


# ------------------------------------------------------------------------------------------------
# The reference's own caller, restated: DiffusionEngine.__init__ plugin assembly (diffusion.py:76-112)
# + sample_video (diffusion.py:504-577) + decode_first_stage (:233-251), with every socket filled from
# a config dict whose `target:` strings point into gcd_amd — the YAML-only drop-in of INTEGRATION.md.
# (tests/test_dropin.py runs the UNMODIFIED reference sample_video on the CPU in the build container
# and checks that the closure it builds is recovered; here the same closure shape drives the GPU.)
# ------------------------------------------------------------------------------------------------
class _RestatedEngine:
    def __init__(self, network_config, denoiser_config, sampler_config, conditioner_config,
                 decoder_config, network_wrapper, scale_factor, en_and_decode_n_samples_a_time):
        from gcd_amd.util import get_obj_from_str, instantiate_from_config
        model = instantiate_from_config(network_config)
        self.model = get_obj_from_str(network_wrapper)(model, compile_model=False)
        self.denoiser = instantiate_from_config(denoiser_config)
        self.sampler = instantiate_from_config(sampler_config)
        self.conditioner = instantiate_from_config(conditioner_config)
        self.first_stage_decoder = instantiate_from_config(decoder_config)
        self.scale_factor = scale_factor
        self.en_and_decode_n_samples_a_time = en_and_decode_n_samples_a_time

    def decode_first_stage(self, z):                                  # diffusion.py:233-251
        z = 1.0 / self.scale_factor * z
        n = self.en_and_decode_n_samples_a_time or z.shape[0]
        # the reference's isinstance(..., sgm VideoDecoder) test is False for the drop-in: no kwargs
        return torch.cat([self.first_stage_decoder(z[i:i + n]) for i in range(0, z.shape[0], n)], 0)

    def sample_video(self, batch):                                     # diffusion.py:504-577
        c, uc = self.conditioner.get_unconditional_conditioning(
            batch, batch_uc=batch, force_uc_zero_embeddings=["cond_frames", "cond_frames_without_noise"])
        additional_model_inputs = {}
        additional_model_inputs["num_video_frames"] = batch["num_video_frames"]
        additional_model_inputs["image_only_indicator"] = \
            batch["image_only_indicator"].repeat_interleave(2, dim=0)

        def denoiser(input, sigma, c):
            return self.denoiser(self.model, input, sigma, c, **additional_model_inputs)

        BT, Cp, Hp, Wp = batch["cond_frames"].shape
        latent_noise = torch.randn((BT, 4, Hp // 8, Wp // 8), device=batch["cond_frames"].device)
        samples_z = self.sampler(denoiser, latent_noise, cond=c, uc=uc).detach()
        samples_x = self.decode_first_stage(samples_z).detach()
        return {"sampled_z": samples_z, "sampled_video": torch.clamp((samples_x + 1.0) / 2.0, 0.0, 1.0),
                "c": c, "uc": uc}


def test_sample_video_config_only_dropin(gpu):
    """frames + camera pose -> conditioner (HIP VAE encoder, HIP embedders) -> EulerEDM loop reached as
    the FUSED path through the sample_video closure -> HIP VideoDecoder, all built from `target:`
    strings, against the CPU oracles of every stage on the same noise."""
    from gcd_amd.camera import scaled_relative_angles
    from oracle import vae_encoder_ref as E
    T, steps, Hp, Wp = 14, 25, 64, 64
    ucfg = O.UNetConfig(model_channels=64, context_dim=64, adm_in_channels=96, aux_emb_dim=32)
    P = "gcd_amd.conditioning."
    enc_dd = dict(E.TINY.as_reference_kwargs(), attn_type="vanilla-xformers")
    eng = _RestatedEngine(
        network_config={"target": "gcd_amd.video_model.VideoUNet", "params": ucfg.as_reference_kwargs()},
        network_wrapper="gcd_amd.wrappers.OpenAIWrapper",
        denoiser_config={"target": "gcd_amd.denoiser.Denoiser", "params": {
            "scaling_config": {"target": "gcd_amd.denoiser_scaling.VScalingWithEDMcNoise"}}},
        sampler_config={"target": "gcd_amd.sampling.EulerEDMSampler", "params": {
            "num_steps": steps,
            "discretization_config": {"target": "gcd_amd.discretizer.EDMDiscretization",
                                      "params": {"sigma_max": 700.0}},
            "guider_config": {"target": "gcd_amd.guiders.LinearPredictionGuider",
                              "params": {"num_frames": T, "max_scale": 1.5, "min_scale": 1.0}}}},
        conditioner_config={"target": P + "GeneralConditioner", "params": {"emb_models": [
            dict(input_key="fps_id", target=P + "ConcatTimestepEmbedderND", params=dict(outdim=32)),
            dict(input_key="motion_bucket_id", is_trainable=True, target=P + "ConcatTimestepEmbedderND",
                 params=dict(outdim=32)),
            dict(input_key="cond_frames_without_noise", target=P + "IdentityEncoder"),   # CLIP token stand-in
            dict(input_key="cond_frames", target=P + "VideoPredictionEmbedderWithEncoder", params=dict(
                n_cond_frames=1, n_copies=1, is_ae=True, disable_encoder_autocast=True,
                en_and_decode_n_samples_a_time=2, scale_factor=0.18215,
                encoder_config={"target": "gcd_amd.ae_encoder.AutoencoderKLModeOnly",
                                "params": {"embed_dim": 4, "monitor": "val/rec_loss", "ddconfig": enc_dd,
                                           "lossconfig": {"target": "torch.nn.Identity"}}})),
            dict(input_key="cond_aug", target=P + "ConcatTimestepEmbedderND", params=dict(outdim=32)),
            dict(input_key="scaled_relative_angles", is_trainable=True, target=P + "SphericalEmbedder",
                 params=dict(embed_dim=32))]}},
        decoder_config={"target": "gcd_amd.temporal_ae.VideoDecoder", "params": D.TINY.as_reference_kwargs()},
        scale_factor=0.18215, en_and_decode_n_samples_a_time=T)
    # seeded weights for every stage, loaded through the reference-named state_dicts
    sd_u = weights.synth_state_dict({k: tuple(v.shape) for k, v in eng.model.diffusion_model.state_dict().items()}, 11)
    sd_c = weights.synth_state_dict({k: tuple(v.shape) for k, v in eng.conditioner.state_dict().items()}, 12)
    sd_d = weights.synth_state_dict({k: tuple(v.shape) for k, v in eng.first_stage_decoder.state_dict().items()}, 13)
    eng.model.diffusion_model.load_state_dict(sd_u)
    eng.conditioner.load_state_dict(sd_c)
    eng.first_stage_decoder.load_state_dict(sd_d)
    for m in (eng.model, eng.conditioner, eng.first_stage_decoder):
        m.to(gpu).eval()
    g = torch.Generator().manual_seed(55)
    frames = torch.rand(T, 3, Hp, Wp, generator=g) * 2.0 - 1.0
    batch_cpu = {"cond_frames": frames + 0.02 * torch.randn(T, 3, Hp, Wp, generator=g),
                 "cond_frames_without_noise": torch.randn(T, 1, 64, generator=g),
                 "fps_id": torch.full((T,), 12.0), "motion_bucket_id": torch.full((T,), 127.0),
                 "cond_aug": torch.full((T,), 0.02),
                 "scaled_relative_angles": scaled_relative_angles(30.0, 15.0, 1.0, num_frames=T),
                 "image_only_indicator": torch.zeros(1, T)}
    batch = {k: v.to(gpu) for k, v in batch_cpu.items()}
    batch["num_video_frames"] = T
    torch.manual_seed(4321)
    noise = torch.randn((T, 4, Hp // 8, Wp // 8), device=gpu).cpu()
    torch.manual_seed(4321)
    out = eng.sample_video(batch)
    torch.cuda.synchronize()
    assert eng.sampler.last_path == "fused", "config-only drop-in did not reach the fused loop"
    # ---- oracle of every stage ----
    enc_sd = {k[len("embedders.3.encoder.encoder."):]: v for k, v in sd_c.items()
              if k.startswith("embedders.3.encoder.encoder.")}
    with torch.no_grad():
        concat = E.encode_mode(enc_sd, E.TINY, batch_cpu["cond_frames"],
                               sd_c["embedders.3.encoder.quant_conv.weight"],
                               sd_c["embedders.3.encoder.quant_conv.bias"]) * 0.18215
        vec = torch.cat([O.concat_timestep_embed(batch_cpu["fps_id"], 32),
                         O.concat_timestep_embed(batch_cpu["motion_bucket_id"], 32),
                         O.concat_timestep_embed(batch_cpu["cond_aug"], 32),
                         O.spherical_embed(batch_cpu["scaled_relative_angles"],
                                           sd_c["embedders.5.proj.weight"], sd_c["embedders.5.proj.bias"])], 1)
        embedded = [("v", vec), ("cond_frames_without_noise", batch_cpu["cond_frames_without_noise"]),
                    ("cond_frames", concat)]
        c = O.general_conditioner(embedded)
        uc = O.general_conditioner(embedded, force_zero=("cond_frames", "cond_frames_without_noise"))
        z_ref = O.sample_loop(sd_u, ucfg, noise, c, uc, T, steps)
        x_ref = D.decode_first_stage(sd_d, D.TINY, z_ref, 0.18215, n_samples=T)
        vid_ref = torch.clamp((x_ref + 1.0) / 2.0, 0.0, 1.0)
    e_cat, e_vec = rel_l2(out["c"]["concat"], c["concat"]), rel_l2(out["c"]["vector"], c["vector"])
    ez = rel_l2(out["sampled_z"], z_ref)
    mse = float(((out["sampled_video"].double().cpu() - vid_ref.double()) ** 2).mean())
    psnr = 10.0 * math.log10(1.0 / max(mse, 1e-30))                  # [0, 1] range
    print(f"sample_video drop-in: concat {e_cat:.2e}, vector {e_vec:.2e}, latents {ez:.3e}, video PSNR {psnr:.1f} dB")
    assert e_cat < 2e-3 and e_vec < 2e-5
    assert float(out["uc"]["concat"].abs().max()) == 0.0 and float(out["uc"]["crossattn"].abs().max()) == 0.0
    # the conditioning latents themselves carry fp16-operand error (1e-3) into the loop: 2e-3 here,
    # the 1e-3 loop contract on identical inputs is held by tests/test_unet_gpu.py
    assert ez < 2e-3 and psnr >= 55.0
