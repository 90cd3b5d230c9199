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


# ParallelDomain class colours of the semantic maps (loss.py:16-33)
PD_PERSON = [[220, 20, 180], [64, 64, 64], [128, 128, 128], [192, 192, 192], [220, 20, 60]]
PD_VEHICLE = [[0, 60, 100], [0, 0, 142], [0, 0, 90], [32, 32, 32], [119, 11, 32], [0, 0, 230], [128, 230, 128],
              [0, 0, 70], [0, 64, 64]]


def get_loss_pd(model_output, target, w, cur_step, gt_rgb, person_w, vehicle_w, loss_type="l2", focus_top=1.0,
                focus_steps=-1):
    """loss.py:163-273 WITH the ParallelDomain class weights (loss.py:196-230), literal: one pass per class colour,
    `loss_bias += loss_raw * area-downsampled mask * (weight - 1)`; half of the bias joins the per-pixel loss before the
    focal top-fraction, the other half is added as a per-frame mean after it."""
    diff = model_output - target
    BT = target.shape[0]
    raw = diff ** 2 if loss_type == "l2" else diff.abs()
    bias = torch.zeros_like(raw)
    todo = ([(c, person_w) for c in PD_PERSON] if person_w > 1.0 else []) + \
           ([(c, vehicle_w) for c in PD_VEHICLE] if vehicle_w > 1.0 else [])
    for rgb, weight in todo:
        col = (torch.tensor(rgb, dtype=torch.float32) / 127.5 - 1.0)[None, :, None, None]
        mask = ((gt_rgb - col).abs().mean(dim=1, keepdim=True) < 0.02).float()
        bias = bias + raw * torch.nn.functional.interpolate(mask, tuple(target.shape[2:4]), mode="area") * (weight - 1.0)
    bias_mean = bias.reshape(BT, -1).mean(1)
    allv = raw + bias * 0.5
    progress = min(max(cur_step / focus_steps, 0.0), 1.0) if focus_steps > 0 else 0.0
    mean = allv.reshape(BT, -1).mean(1)
    cur_top = (1.0 - progress) + focus_top * progress
    if cur_top < 1.0:
        flat = allv.reshape(BT, -1)
        focal = flat.topk(int(flat.shape[1] * cur_top), dim=1)[0].mean(1) * 0.9 + mean * 0.1
    else:
        focal = mean
    return (focal + bias_mean * 0.5) * w.flatten()
