"""Drop-in for sgm.modules.diffusionmodules.sampling.EulerEDMSampler (reference sampling.py:26-144,
225-230): the `sampler_config.target` socket.

    sampler = EulerEDMSampler(discretization_config, num_steps, guider_config, verbose, device)
    x0 = sampler(denoiser, x, cond, uc=None, num_steps=None)

Two execution paths with identical results:
  * generic — any `denoiser(input, sigma, c)` callable (e.g. the closure DiffusionEngine.sample_video
    builds, diffusion.py:531-532): the reference's step arithmetic in torch, without the reference's
    per-step device->host sync (sampling.py:131 compares a device tensor; with s_churn == 0 the
    comparison cannot change gamma, so it is skipped; otherwise it uses a host copy of the sigmas);
  * fused — when the denoiser is a `FusedDenoiser` around gcd_amd's own Denoiser / OpenAIWrapper /
    VideoUNet and the guider is a LinearPredictionGuider: the whole step (EDM scalings, CFG batch
    assembly, UNet, guidance, Euler update) runs as libgcd_amd kernels on static buffers, one
    hipGraph replay per step, sigma fed from a device table.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import torch

from . import _lib, ops
from .denoiser import Denoiser
from .denoiser_scaling import VScalingWithEDMcNoise
from .guiders import LinearPredictionGuider
from .util import append_dims, default, instantiate_from_config
from .video_model import VideoUNet
from .wrappers import OpenAIWrapper

DEFAULT_GUIDER = {"target": "gcd_amd.guiders.IdentityGuider"}


class FusedDenoiser:
    """`denoiser(input, sigma, c)` exactly like the closure in DiffusionEngine.sample_video
    (diffusion.py:526-532), but with its captured objects visible so the sampler may fuse."""

    def __init__(self, denoiser: Denoiser, network, **additional_model_inputs):
        self.denoiser = denoiser
        self.network = network
        self.additional_model_inputs = additional_model_inputs

    def __call__(self, input, sigma, c):
        return self.denoiser(self.network, input, sigma, c, **self.additional_model_inputs)


class BaseDiffusionSampler:
    def __init__(self, discretization_config, num_steps: Optional[int] = None, guider_config=None,
                 verbose: bool = False, device: str = "cuda"):
        self.num_steps = num_steps
        self.discretization = instantiate_from_config(discretization_config)
        self.guider = instantiate_from_config(default(guider_config, DEFAULT_GUIDER))
        self.verbose = verbose
        self.device = device

    def prepare_sampling_loop(self, x, cond, uc=None, num_steps=None):
        sigmas = self.discretization(self.num_steps if num_steps is None else num_steps,
                                     device=self.device)
        uc = default(uc, cond)
        x *= torch.sqrt(1.0 + sigmas[0] ** 2.0)          # in place, like sampling.py:54
        num_sigmas = len(sigmas)
        s_in = x.new_ones([x.shape[0]])
        return x, s_in, sigmas, num_sigmas, cond, uc

    def denoise(self, x, denoiser, sigma, cond, uc):
        denoised = denoiser(*self.guider.prepare_inputs(x, sigma, cond, uc))
        return self.guider(denoised, sigma)


class EDMSampler(BaseDiffusionSampler):
    def __init__(self, s_churn=0.0, s_tmin=0.0, s_tmax=float("inf"), s_noise=1.0, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.s_churn, self.s_tmin, self.s_tmax, self.s_noise = s_churn, s_tmin, s_tmax, s_noise
        self.use_graph = True      # fused path: replay one captured step per iteration
        self.last_path = None      # "fused" | "generic" (introspection for tests / bench)

    def euler_step(self, x, d, dt):
        return x + dt * d

    def possible_correction_step(self, euler_step, x, d, dt, next_sigma, denoiser, cond, uc):
        raise NotImplementedError

    def sampler_step(self, sigma, next_sigma, denoiser, x, cond, uc=None, gamma=0.0):
        sigma_hat = sigma * (gamma + 1.0)
        if gamma > 0:
            eps = torch.randn_like(x) * self.s_noise
            x = x + eps * append_dims(sigma_hat ** 2 - sigma ** 2, x.ndim) ** 0.5
        denoised = self.denoise(x, denoiser, sigma_hat, cond, uc)
        d = (x - denoised) / append_dims(sigma_hat, x.ndim)          # to_d, sampling_utils.py:34-35
        dt = append_dims(next_sigma - sigma_hat, x.ndim)
        return self.possible_correction_step(self.euler_step(x, d, dt), x, d, dt, next_sigma,
                                             denoiser, cond, uc)

    # -------------------------------------------------------------------------------------------
    def __call__(self, denoiser, x, cond, uc=None, num_steps=None):
        if self._can_fuse(denoiser, x, cond, uc):
            self.last_path = "fused"
            return self._call_fused(denoiser, x, cond, default(uc, cond), num_steps)
        self.last_path = "generic"
        x, s_in, sigmas, num_sigmas, cond, uc = self.prepare_sampling_loop(x, cond, uc, num_steps)
        sig_host = sigmas.detach().cpu().tolist() if self.s_churn > 0 else None
        for i in range(num_sigmas - 1):
            gamma = 0.0
            if sig_host is not None and self.s_tmin <= sig_host[i] <= self.s_tmax:
                gamma = min(self.s_churn / (num_sigmas - 1), 2 ** 0.5 - 1)
            x = self.sampler_step(s_in * sigmas[i], s_in * sigmas[i + 1], denoiser, x, cond, uc, gamma)
        return x

    # -------------------------------------------------------------------------------------------
    def _can_fuse(self, denoiser, x, cond, uc) -> bool:
        if not isinstance(denoiser, FusedDenoiser) or not isinstance(self, EulerEDMSampler):
            return False
        if type(denoiser.denoiser) is not Denoiser or \
                not isinstance(denoiser.denoiser.scaling, VScalingWithEDMcNoise):
            return False
        net = denoiser.network
        if type(net) is not OpenAIWrapper or not isinstance(net.diffusion_model, VideoUNet):
            return False
        if type(self.guider) is not LinearPredictionGuider or self.guider.additional_cond_keys:
            return False
        if self.s_churn != 0 or not x.is_cuda or x.dtype != torch.float32:
            return False
        if set(cond.keys()) != {"vector", "crossattn", "concat"}:
            return False
        extra = set(denoiser.additional_model_inputs) - {"num_video_frames", "image_only_indicator"}
        T = denoiser.additional_model_inputs.get("num_video_frames")
        if extra or T is None or T != self.guider.num_frames or x.shape[0] % T:
            return False
        unet = net.diffusion_model
        return x.shape[1] + cond["concat"].shape[1] == unet.in_channels

    def _call_fused(self, fd: FusedDenoiser, x, cond, uc, num_steps):
        unet: VideoUNet = fd.network.diffusion_model
        eng = unet.engine
        if eng.packed is None:
            eng.pack()
        T = fd.additional_model_inputs["num_video_frames"]
        ioi = fd.additional_model_inputs.get("image_only_indicator")
        dev = x.device
        sigmas = self.discretization(self.num_steps if num_steps is None else num_steps,
                                     device=self.device).to(device=dev, dtype=torch.float32)
        x *= torch.sqrt(1.0 + sigmas[0] ** 2.0)          # caller's tensor, like sampling.py:54
        nx = x.shape[0]
        N = 2 * nx
        # ---- static buffers (guiders.py:89-100: batch = [uc | c]) ----
        f32 = dict(device=dev, dtype=torch.float32)
        xs = x.detach().clone().contiguous()
        concat2 = torch.cat((uc["concat"], cond["concat"]), 0).to(**f32).contiguous()
        ctx2 = torch.cat((uc["crossattn"], cond["crossattn"]), 0).to(**f32).contiguous()
        y2 = torch.cat((uc["vector"], cond["vector"]), 0).to(**f32).contiguous()
        net_out = torch.empty(N, unet.out_channels, *x.shape[2:], **f32)
        c_in, c_noise = torch.empty(N, **f32), torch.empty(N, **f32)
        sig = torch.empty(2, **f32)
        scale = self.guider.scale.reshape(-1).to(**f32).contiguous()
        alphas = eng.blend_alphas(ioi, N, T)
        lib = _lib.load()

        def step():
            ops.edm_scalings(sig, c_in, c_noise)
            eng.run(xs, concat2, c_in, c_noise, ctx2, y2, T, None, net_out, alphas=alphas)
            ops.cfg_euler_step(xs, net_out, scale, sig, xs, T)

        n = len(sigmas) - 1
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        graph = C.c_void_p()
        try:
            with torch.cuda.stream(side):
                for i in range(n):
                    sig.copy_(sigmas[i:i + 2])
                    if not self.use_graph or i == 0 or n < 3:
                        step()                                   # eager (also warms the workspace)
                    elif i == 1:
                        _lib.check(lib.gcd_graph_begin_capture(side.cuda_stream), "graph capture")
                        try:
                            step()
                        finally:
                            _lib.check(lib.gcd_graph_end_capture(side.cuda_stream, C.byref(graph)),
                                       "graph instantiate")
                        _lib.check(lib.gcd_graph_launch(graph, side.cuda_stream), "graph launch")
                    else:
                        _lib.check(lib.gcd_graph_launch(graph, side.cuda_stream), "graph launch")
                side.synchronize()
        finally:
            if graph:
                lib.gcd_graph_destroy(graph)
        torch.cuda.current_stream(dev).wait_stream(side)
        return xs


class EulerEDMSampler(EDMSampler):
    """Deterministic Euler steps, no correction (sampling.py:225-230)."""

    def possible_correction_step(self, euler_step, x, d, dt, next_sigma, denoiser, cond, uc):
        return euler_step
