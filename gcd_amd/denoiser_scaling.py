"""Drop-in for sgm.modules.diffusionmodules.denoiser_scaling (reference denoiser_scaling.py:53-61):
the `scaling_config.target` socket.  __call__(sigma) -> (c_skip, c_out, c_in, c_noise)."""
from __future__ import annotations

from typing import Tuple

import torch


class VScalingWithEDMcNoise:
    def __call__(self, sigma: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
        var = sigma ** 2 + 1.0
        c_skip = 1.0 / var
        c_out = -sigma / var ** 0.5
        c_in = 1.0 / var ** 0.5
        c_noise = 0.25 * sigma.log()
        return c_skip, c_out, c_in, c_noise
