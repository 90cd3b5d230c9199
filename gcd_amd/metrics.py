"""Evaluation metrics of GCD's test script: per-frame PSNR / SSIM of the sampled
videos against ground truth, their visible / occluded variants from the re-projection mask, and the
sample diversity — `scripts/test.py:346-496` (calculate_metrics) and `scripts/eval_utils.py:571-666`
(masked_ssim).  Host-side numpy / scipy work, exactly like the reference (frames leave the GPU as
numpy arrays, test.py:334-338); not on the denoising hot path.

The reference calls scikit-image 0.22.0 (`requirements_versions.txt:40`), which is not installed
here; the two functions it uses are restated from the published algorithm:
  * peak_signal_noise_ratio = 10 log10(data_range^2 / MSE)                      (skimage.metrics)
  * structural_similarity (Wang et al. 2004 as implemented by skimage 0.22.0): 7x7 uniform window,
    sample covariance (NP / (NP - 1)), K1 = 0.01, K2 = 0.03, mean over the image cropped by the
    window radius, channels averaged (`channel_axis=0`).
`masked_ssim` is the reference's own variant: the same SSIM map averaged over the pixels of an
arbitrary mask eroded by the window radius.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
from scipy.ndimage import binary_erosion, uniform_filter


def peak_signal_noise_ratio(image_test: np.ndarray, image_true: np.ndarray, data_range: float = 1.0) -> float:
    a = np.asarray(image_true, dtype=np.float64)
    b = np.asarray(image_test, dtype=np.float64)
    if a.shape != b.shape:
        raise ValueError("Input images must have the same dimensions.")
    mse = np.mean((a - b) ** 2, dtype=np.float64)
    with np.errstate(divide="ignore"):
        return float(10.0 * np.log10((data_range ** 2) / mse))


def _ssim_map(im1: np.ndarray, im2: np.ndarray, win_size: int, K1: float, K2: float, data_range: float):
    """SSIM map of one channel (2-D arrays) and the crop radius."""
    if win_size % 2 != 1:
        raise ValueError("Window size must be odd.")
    if im1.shape != im2.shape:
        raise ValueError("Input images must have the same dimensions.")
    if min(im1.shape) < win_size:
        raise ValueError("win_size exceeds image extent.")
    ft = np.float32 if im1.dtype == np.float32 else np.float64      # skimage _supported_float_type
    x = im1.astype(ft, copy=False)
    y = im2.astype(ft, copy=False)
    NP = win_size ** x.ndim
    cov_norm = NP / (NP - 1)                                       # sample covariance
    ux = uniform_filter(x, size=win_size)
    uy = uniform_filter(y, size=win_size)
    uxx = uniform_filter(x * x, size=win_size)
    uyy = uniform_filter(y * y, size=win_size)
    uxy = uniform_filter(x * y, size=win_size)
    vx = cov_norm * (uxx - ux * ux)
    vy = cov_norm * (uyy - uy * uy)
    vxy = cov_norm * (uxy - ux * uy)
    C1 = (K1 * data_range) ** 2
    C2 = (K2 * data_range) ** 2
    S = ((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux ** 2 + uy ** 2 + C1) * (vx + vy + C2))
    return S, (win_size - 1) // 2


def _crop(a: np.ndarray, pad: int) -> np.ndarray:
    return a[tuple(slice(pad, s - pad) for s in a.shape)] if pad else a


def structural_similarity(im1: np.ndarray, im2: np.ndarray, data_range: float = 1.0,
                          channel_axis: Optional[int] = 0, win_size: int = 7, K1: float = 0.01,
                          K2: float = 0.03) -> float:
    im1, im2 = np.asarray(im1), np.asarray(im2)
    if channel_axis is not None:
        a = np.moveaxis(im1, channel_axis, 0)
        b = np.moveaxis(im2, channel_axis, 0)
        return float(np.mean([structural_similarity(a[c], b[c], data_range, None, win_size, K1, K2)
                              for c in range(a.shape[0])]))
    S, pad = _ssim_map(im1, im2, win_size, K1, K2, data_range)
    return float(np.mean(_crop(S, pad), dtype=np.float64))


def masked_ssim(im1: np.ndarray, im2: np.ndarray, mask: np.ndarray, win_size: int = 7, K1: float = 0.01,
                K2: float = 0.03, channel_axis: Optional[int] = 0) -> np.ndarray:
    """(mssim over the whole cropped image, mssim over the eroded mask) — eval_utils.py:571-666
    (data_range 1, uniform window).  An empty eroded mask gives nan, as np.mean of nothing does."""
    im1, im2 = np.asarray(im1), np.asarray(im2)
    mask = np.asarray(mask).astype(bool)
    if channel_axis is not None:
        a = np.moveaxis(im1, channel_axis, 0)
        b = np.moveaxis(im2, channel_axis, 0)
        return np.mean([masked_ssim(a[c], b[c], mask, win_size, K1, K2, None) for c in range(a.shape[0])], axis=0)
    S, pad = _ssim_map(im1, im2, win_size, K1, K2, 1.0)
    S_crop = _crop(S, pad)
    mask_crop = _crop(binary_erosion(mask, iterations=pad) if pad else mask, pad)
    with np.errstate(invalid="ignore"), __import__("warnings").catch_warnings():
        __import__("warnings").simplefilter("ignore", RuntimeWarning)
        m_all = np.mean(S_crop, dtype=np.float64)
        m_mask = np.mean(S_crop[mask_crop], dtype=np.float64)
    return np.array([m_all, m_mask])


def calculate_metrics(gt_rgb: np.ndarray, reproject_rgb: Optional[np.ndarray],
                      pred_samples: Sequence[Dict[str, np.ndarray]]) -> Tuple[Dict[str, np.ndarray], np.ndarray]:
    """scripts/test.py:346-496.  gt_rgb / reproject_rgb: (T, 3, H, W) float32 in [0, 1];
    pred_samples: dicts with 'sampled_rgb' (T, 3, H, W).  Returns (metrics dict, per-pixel
    uncertainty (T, H, W)) with the reference's keys."""
    S = len(pred_samples)
    pred = np.stack([p["sampled_rgb"] for p in pred_samples], axis=0) if S >= 1 else np.zeros((0,) + gt_rgb.shape)
    have_mask = reproject_rgb is not None
    if have_mask:
        occluded = (np.sum(np.abs(reproject_rgb), axis=1) <= 1e-7).astype(np.uint8)
        visible = 1 - occluded
        vis_bc = np.tile(visible[:, None].astype(bool), (1, 3, 1, 1))
        occ_bc = np.tile(occluded[:, None].astype(bool), (1, 3, 1, 1))
    names = ["psnr", "ssim"] + (["psnr_vis", "ssim_vis", "psnr_occ", "ssim_occ"] if have_mask else [])
    frame: Dict[str, List[List[float]]] = {n: [] for n in names}
    for out in pred:
        T = out.shape[0]
        cur: Dict[str, List[float]] = {n: [] for n in names}
        for t in range(T):
            cur["psnr"].append(peak_signal_noise_ratio(out[t], gt_rgb[t], 1.0))
            cur["ssim"].append(structural_similarity(out[t], gt_rgb[t], 1.0, 0))
            if have_mask:
                for tag, m in (("vis", vis_bc[t]), ("occ", occ_bc[t])):
                    if m.any():
                        cur["psnr_" + tag].append(peak_signal_noise_ratio(out[t][m], gt_rgb[t][m], 1.0))
                        cur["ssim_" + tag].append(float(masked_ssim(out[t], gt_rgb[t], m[0])[1]))
                    else:
                        cur["psnr_" + tag].append(np.nan)
                        cur["ssim_" + tag].append(np.nan)
        for n in names:
            frame[n].append(cur[n])
    import warnings
    md: Dict[str, np.ndarray] = {}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)              # nanmean of all-nan rows
        for n in names:
            arr = np.array(frame[n], dtype=np.float64).reshape(S, -1)
            md["frame_" + n] = arr
            md["mean_" + n] = np.nanmean(arr, axis=1) if arr.size else np.zeros((S,))
        uncertainty = np.nanmean(np.std(pred, axis=0), axis=1)       # (T, H, W)
        md["frame_diversity"] = np.nanmean(uncertainty, axis=(1, 2))
        md["mean_diversity"] = np.nanmean(md["frame_diversity"])
        if have_mask:
            T = gt_rgb.shape[0]
            for tag, m in (("vis", vis_bc), ("occ", occ_bc)):
                per_t = [np.stack([x[t][m[t]] for x in pred], axis=0) for t in range(T)]
                fd = np.array([np.nanmean(np.std(x, axis=0)) if x.size else np.nan for x in per_t])
                md["frame_diversity_" + tag] = fd
                md["mean_diversity_" + tag] = np.nanmean(fd)
    return md, uncertainty
