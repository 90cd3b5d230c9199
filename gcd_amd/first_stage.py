"""`DiffusionEngine.decode_first_stage` (sgm/models/diffusion.py:233-251) for the HIP `VideoDecoder`:
latents off the sampling loop -> frames.  Same arithmetic and chunking rule as the reference: the
latents are divided by `scale_factor`, decoded `en_and_decode_n_samples_a_time` frames at a time,
and every chunk is one clip for the time-mixing layers (`timesteps = len(chunk)`), so — exactly as
in the reference — the chunk length has to match the clip length for the temporal convolutions to
see whole clips (scripts/infer.py:83: `--decoding_t 14` for 14-frame clips).
"""
from __future__ import annotations

import math
from typing import Optional

import torch


@torch.no_grad()
def decode_first_stage(decoder, z: torch.Tensor, scale_factor: float = 0.18215,
                       en_and_decode_n_samples_a_time: Optional[int] = None) -> torch.Tensor:
    z = 1.0 / scale_factor * z
    n_samples = z.shape[0] if en_and_decode_n_samples_a_time is None else \
        int(en_and_decode_n_samples_a_time)
    n_rounds = math.ceil(z.shape[0] / n_samples)
    all_out = []
    for n in range(n_rounds):
        chunk = z[n * n_samples:(n + 1) * n_samples]
        all_out.append(decoder(chunk, timesteps=len(chunk)))
    return torch.cat(all_out, dim=0)
