"""ORACLE (test infrastructure, NOT product code): brute-force restatement of the image metrics GCD's
evaluation uses (scripts/test.py:386-420 -> skimage.metrics 0.22.0; scripts/eval_utils.py:571-666).

PARITY PIN (round 3): the SSIM arithmetic is pinned to the REFERENCE'S OWN CODE — scripts/eval_utils.py:571-666
(`masked_ssim`: skimage 0.22.0's structural_similarity as adapted by the reference authors) executed unmodified by
oracle/make_golden_metrics.py (scikit-image, absent from this image, only contributes four one-line helpers, stubbed
from their documented behaviour) -> tests/golden/metrics_kat.pt; tests/test_metrics.py holds gcd_amd.metrics to it
at 1e-12.  This file remains the independent brute-force restatement (explicit window loops from Wang et al. 2004:
7x7 uniform window, 'reflect' boundary as scipy.ndimage.uniform_filter's default, sample covariance, crop by the
window radius) that the same tests also compare against.  Unpinned by reference outputs: only skimage's own entry
points (`peak_signal_noise_ratio`, argument handling of `structural_similarity`), not the arithmetic.
"""
from __future__ import annotations

import numpy as np


def psnr(test, true, data_range=1.0):
    d = np.asarray(true, np.float64) - np.asarray(test, np.float64)
    return 10.0 * np.log10(data_range ** 2 / np.mean(d * d))


def _reflect(i, n):
    # scipy 'reflect' (d c b a | a b c d | d c b a)
    while i < 0 or i >= n:
        i = -i - 1 if i < 0 else 2 * n - 1 - i
    return i


def ssim_map_2d(x, y, win=7, K1=0.01, K2=0.03, R=1.0):
    x = np.asarray(x, np.float64)
    y = np.asarray(y, np.float64)
    H, W = x.shape
    r = win // 2
    NP = win * win
    S = np.zeros((H, W))
    C1, C2 = (K1 * R) ** 2, (K2 * R) ** 2
    for i in range(H):
        rows = [_reflect(i + d, H) for d in range(-r, r + 1)]
        for j in range(W):
            cols = [_reflect(j + d, W) for d in range(-r, r + 1)]
            a = x[np.ix_(rows, cols)].ravel()
            b = y[np.ix_(rows, cols)].ravel()
            ux, uy = a.mean(), b.mean()
            vx = ((a * a).mean() - ux * ux) * NP / (NP - 1)
            vy = ((b * b).mean() - uy * uy) * NP / (NP - 1)
            vxy = ((a * b).mean() - ux * uy) * NP / (NP - 1)
            S[i, j] = ((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux * ux + uy * uy + C1) * (vx + vy + C2))
    return S


def ssim(im1, im2, win=7):
    """channel_axis=0 mean of the cropped per-channel maps."""
    r = win // 2
    return float(np.mean([ssim_map_2d(a, b, win)[r:-r, r:-r].mean() for a, b in zip(im1, im2)]))


def erode(mask, iterations):
    """binary erosion with the 4-connected cross (scipy.ndimage.binary_erosion default), border = 0."""
    m = np.asarray(mask, bool)
    for _ in range(iterations):
        p = np.pad(m, 1, constant_values=False)
        m = p[1:-1, 1:-1] & p[:-2, 1:-1] & p[2:, 1:-1] & p[1:-1, :-2] & p[1:-1, 2:]
    return m


def masked_ssim(im1, im2, mask, win=7):
    r = win // 2
    me = erode(mask, r)[r:-r, r:-r]
    vals_all, vals_mask = [], []
    for a, b in zip(im1, im2):
        S = ssim_map_2d(a, b, win)[r:-r, r:-r]
        vals_all.append(S.mean())
        vals_mask.append(S[me].mean() if me.any() else np.nan)
    return np.array([np.mean(vals_all), np.mean(vals_mask)])
