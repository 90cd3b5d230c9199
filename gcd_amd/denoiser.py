"""Drop-in for sgm.modules.diffusionmodules.denoiser.Denoiser (reference denoiser.py:11-49): the
`denoiser_config.target` socket.  forward(network, input, sigma, cond, **additional_model_inputs)."""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn as nn

from .util import append_dims, instantiate_from_config


class Denoiser(nn.Module):
    def __init__(self, scaling_config: Dict):
        super().__init__()
        self.scaling = instantiate_from_config(scaling_config)

    def possibly_quantize_sigma(self, sigma: torch.Tensor) -> torch.Tensor:
        return sigma

    def possibly_quantize_c_noise(self, c_noise: torch.Tensor) -> torch.Tensor:
        return c_noise

    def forward(self, network: nn.Module, input: torch.Tensor, sigma: torch.Tensor, cond: Dict,
                **additional_model_inputs) -> torch.Tensor:
        sigma = self.possibly_quantize_sigma(sigma)
        sigma_shape = sigma.shape
        sigma = append_dims(sigma, input.ndim)
        c_skip, c_out, c_in, c_noise = self.scaling(sigma)
        c_noise = self.possibly_quantize_c_noise(c_noise.reshape(sigma_shape))
        return network(input * c_in, c_noise, cond, **additional_model_inputs) * c_out + input * c_skip
