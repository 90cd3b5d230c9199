"""ORACLE (test infrastructure, NOT product code): restatement of GCD's training loss
(sgm/modules/diffusionmodules/loss.py:115-273, sigma_sampling.py:6-13, loss_weighting.py:17-22) as
plain functions.  Pinned to tests/golden/loss_kat.pt, produced by the unmodified reference classes
(oracle/make_golden_loss.py)."""
from __future__ import annotations

import torch


def edm_sigmas(rand: torch.Tensor, p_mean: float, p_std: float) -> torch.Tensor:
    return (p_mean + p_std * rand).exp()


def harmonize(sigmas: torch.Tensor, T: int) -> torch.Tensor:
    """loss.py:131-136: every frame of a clip gets the clip's first sigma."""
    r = sigmas.reshape(-1, T)
    return r[:, 0:1].expand(r.shape).reshape(-1)


def edm_weighting(sigma: torch.Tensor, sigma_data: float) -> torch.Tensor:
    return (sigma ** 2 + sigma_data ** 2) / (sigma * sigma_data) ** 2


def get_loss(model_output, target, w, cur_step, loss_type="l2", focus_top=1.0, focus_steps=-1):
    """loss.py:163-273 without the ParallelDomain class weights."""
    diff = model_output - target
    BT = target.shape[0]
    raw = diff ** 2 if loss_type == "l2" else diff.abs()
    progress = min(max(cur_step / focus_steps, 0.0), 1.0) if focus_steps > 0 else 0.0
    mean = raw.reshape(BT, -1).mean(1)
    cur_top = (1.0 - progress) + focus_top * progress
    if cur_top < 1.0:
        flat = raw.reshape(BT, -1)
        keep = int(flat.shape[1] * cur_top)
        focal = flat.topk(keep, dim=1)[0].mean(1) * 0.9 + mean * 0.1
    else:
        focal = mean
    return focal * w.flatten()
