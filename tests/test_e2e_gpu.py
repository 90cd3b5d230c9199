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
