"""Drop-in for sgm.modules.diffusionmodules.wrappers (reference wrappers.py:1-34): the
`network_wrapper` socket of DiffusionEngine (diffusion.py:51,77-79).  Must expose
`.diffusion_model` (diffusion.py:128,135,141)."""
from __future__ import annotations

import torch
import torch.nn as nn

OPENAIUNETWRAPPER = "gcd_amd.wrappers.OpenAIWrapper"


class IdentityWrapper(nn.Module):
    def __init__(self, diffusion_model, compile_model: bool = False):
        super().__init__()
        # compile_model is accepted for signature parity; the network is already hand-written kernels
        self.diffusion_model = diffusion_model

    def forward(self, *args, **kwargs):
        return self.diffusion_model(*args, **kwargs)


class OpenAIWrapper(IdentityWrapper):
    """forward(x, t, c, **kwargs): channel-concat c['concat'], map crossattn -> context and
    vector -> y (wrappers.py:23-34)."""

    def forward(self, x: torch.Tensor, t: torch.Tensor, c: dict, **kwargs) -> torch.Tensor:
        concat = c.get("concat", None)
        if concat is not None and concat.numel() > 0:
            x = torch.cat((x, concat.type_as(x)), dim=1)
        return self.diffusion_model(x, timesteps=t, context=c.get("crossattn", None),
                                    y=c.get("vector", None), **kwargs)
