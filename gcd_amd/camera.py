"""Camera trajectories that produce `batch['scaled_relative_angles']`, the input of
`SphericalEmbedder` (SURVEY.md §8a a22 / §8d "Synthetic inputs").

Restates `construct_trajectory` (reference gcd-model/sgm/data/common.py:450-479) and the batch
assembly around it (scripts/eval_utils.py:235-245, sgm/data/kubric_arbit.py:565-647): the source
camera stays put, the destination camera moves from `start` to `end` over the first `move_time`
frames ('interpol_linear' or 'interpol_sine') and then rests; the model is conditioned on
destination - source per frame, with the two angles converted to radians.
"""
from __future__ import annotations

import math

import numpy as np
import torch


def construct_trajectory(spherical_start, spherical_end, trajectory: str, model_frames: int,
                         move_time: int):
    """(3,), (3,) float32 (azimuth deg, elevation deg, radius m) -> (src, dst), each (T, 3) float32."""
    start = np.asarray(spherical_start, dtype=np.float32)
    end = np.asarray(spherical_end, dtype=np.float32)
    T = int(model_frames)
    src = np.repeat(start[None], T, axis=0)
    dst = np.repeat(end[None], T, axis=0)
    for t in range(max(0, int(move_time))):
        if trajectory == "interpol_linear":
            alpha = t / move_time
        elif trajectory == "interpol_sine":
            alpha = (1.0 - np.cos(t / move_time * np.pi)) / 2.0
        else:
            raise ValueError(f"Unknown trajectory: {trajectory}")
        dst[t] = start * (1.0 - alpha) + end * alpha
    return src, dst


def scaled_relative_angles(azimuth_deg: float, elevation_deg: float, radius_m: float,
                           num_frames: int = 14, trajectory: str = "interpol_linear",
                           move_time: int = 13, device=None) -> torch.Tensor:
    """The (T, 3) fp32 tensor GCD feeds `SphericalEmbedder` for a camera that travels to
    (azimuth, elevation, radius) relative to the input view (eval_utils.py:235-245)."""
    src, dst = construct_trajectory(np.zeros(3, np.float32),
                                    np.array([azimuth_deg, elevation_deg, radius_m], np.float32),
                                    trajectory, num_frames, move_time)
    rel = dst - src
    rel[:, 0] *= math.pi / 180.0
    rel[:, 1] *= math.pi / 180.0
    return torch.tensor(rel, dtype=torch.float32, device=device)
