"""Drop-in for sgm.modules.diffusionmodules.discretizer.EDMDiscretization (reference
discretizer.py:16-39): the `discretization_config.target` socket.
__call__(n, do_append_zero=True, device="cpu", flip=False) -> sigmas."""
from __future__ import annotations

import torch

from .util import append_zero


class Discretization:
    def __call__(self, n, do_append_zero=True, device="cpu", flip=False):
        sigmas = self.get_sigmas(n, device=device)
        sigmas = append_zero(sigmas) if do_append_zero else sigmas
        return sigmas if not flip else torch.flip(sigmas, (0,))

    def get_sigmas(self, n, device):
        raise NotImplementedError


class EDMDiscretization(Discretization):
    """sigma_i = (smax^(1/rho) + i/(n-1) (smin^(1/rho) - smax^(1/rho)))^rho  (Karras et al. 2022)."""

    def __init__(self, sigma_min=0.002, sigma_max=80.0, rho=7.0):
        self.sigma_min, self.sigma_max, self.rho = sigma_min, sigma_max, rho

    def get_sigmas(self, n, device="cpu"):
        ramp = torch.linspace(0, 1, n, device=device)
        lo, hi = self.sigma_min ** (1 / self.rho), self.sigma_max ** (1 / self.rho)
        return (hi + ramp * (lo - hi)) ** self.rho
