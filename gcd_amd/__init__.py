"""gcd_amd — MI355X-native (gfx950 / CDNA4) implementation of GCD's denoising hot path: the Stable
Video Diffusion VideoUNet forward and the EulerEDM sampling loop, behind the reference's own sgm
plugin surface.  YAML `target:` strings that pointed at `sgm.modules.diffusionmodules.*` point at:

    network_config.target        gcd_amd.video_model.VideoUNet
    network_wrapper              gcd_amd.wrappers.OpenAIWrapper
    denoiser_config.target       gcd_amd.denoiser.Denoiser
    scaling_config.target        gcd_amd.denoiser_scaling.VScalingWithEDMcNoise
    sampler_config.target        gcd_amd.sampling.EulerEDMSampler
    guider_config.target         gcd_amd.guiders.LinearPredictionGuider
    discretization_config.target gcd_amd.discretizer.EDMDiscretization
    conditioner_config.target    gcd_amd.conditioning.GeneralConditioner
    (conditioner embedders)      gcd_amd.conditioning.{SphericalEmbedder,CameraEmbedder,ConcatTimestepEmbedderND,
                                                       VideoPredictionEmbedderWithEncoder,...}
    first-stage decoder_config   gcd_amd.temporal_ae.VideoDecoder
    loss_fn_config.target        gcd_amd.training.StandardDiffusionLoss          (fine-tune step, a vertical slice)

Beside the sockets: gcd_amd.parallel (clip sharding over GPUs), gcd_amd.camera (pose trajectories),
gcd_amd.metrics (PSNR / SSIM of the evaluation script), gcd_amd.eval_io (its frame / video writer), gcd_amd.autograd_ops (HIP-backed autograd operators).

All arithmetic runs in libgcd_amd.so (hand-written HIP, C ABI in include/gcd_amd.h); importing the
package does not load it, using any op does — and raises if it is missing.
"""
__version__ = "0.2.0"
