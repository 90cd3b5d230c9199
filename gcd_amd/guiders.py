"""Drop-in for sgm.modules.diffusionmodules.guiders (reference guiders.py:13-100): the
`guider_config.target` socket.  prepare_inputs(x, s, c, uc) / __call__(x, sigma)."""
from __future__ import annotations

from typing import List, Optional, Union

import torch

from .util import append_dims


class IdentityGuider:
    def __call__(self, x, sigma):
        return x

    def prepare_inputs(self, x, s, c, uc):
        return x, s, {k: c[k] for k in c}


class LinearPredictionGuider:
    """Per-frame CFG scale linspace(min_scale, max_scale, num_frames) (guiders.py:60-100)."""

    def __init__(self, max_scale: float, num_frames: int, min_scale: float = 1.0,
                 additional_cond_keys: Optional[Union[List[str], str]] = None):
        self.min_scale, self.max_scale, self.num_frames = min_scale, max_scale, num_frames
        self.scale = torch.linspace(min_scale, max_scale, num_frames).unsqueeze(0)
        if additional_cond_keys is None:
            additional_cond_keys = []
        elif isinstance(additional_cond_keys, str):
            additional_cond_keys = [additional_cond_keys]
        self.additional_cond_keys = additional_cond_keys

    def __call__(self, x: torch.Tensor, sigma: torch.Tensor) -> torch.Tensor:
        x_u, x_c = x.chunk(2)
        t = self.num_frames
        x_u = x_u.reshape(-1, t, *x_u.shape[1:])
        x_c = x_c.reshape(-1, t, *x_c.shape[1:])
        scale = append_dims(self.scale.expand(x_u.shape[0], t), x_u.ndim).to(x_u.device)
        out = x_u + scale * (x_c - x_u)
        return out.reshape(-1, *out.shape[2:])

    def prepare_inputs(self, x, s, c, uc):
        c_out = dict()
        for k in c:
            if k in ["vector", "crossattn", "concat"] + self.additional_cond_keys:
                c_out[k] = torch.cat((uc[k], c[k]), 0)
            elif "hijack" not in k:
                assert c[k] == uc[k]
                c_out[k] = c[k]
        return torch.cat([x] * 2), torch.cat([s] * 2), c_out
