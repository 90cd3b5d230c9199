"""ORACLE support (build container only): known answers of the evaluation SSIM from the reference's OWN code.

scripts/eval_utils.py:571-666 (`masked_ssim`) is, in the reference authors' words, skimage 0.22.0's
`structural_similarity` adapted to arbitrary masks: the whole algorithm (uniform 7x7 window, sample covariance,
K1 / K2, crop by the window radius) is in the reference tree; scikit-image — absent from this image — only supplies
four one-line helpers.  This script executes that function's source UNMODIFIED (it is cut out of eval_utils.py by
line, because importing the file pulls in cv2 / imageio / plotly) with those helpers stubbed from their documented
behaviour, and stores its results.  With an all-true mask its first return value is what
`skimage.metrics.structural_similarity(..., data_range=1, channel_axis=0)` — the call at scripts/test.py:386-420 —
computes (same code path before the masking).  Re-run with:  python -m oracle.make_golden_metrics
    -> tests/golden/metrics_kat.pt
"""
from __future__ import annotations

import functools
import sys
import types
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
SRC = Path("/root/reference/gcd-model/scripts/eval_utils.py")
OUT = ROOT / "tests" / "golden"


def _install_skimage_helper_stubs():
    """The four helpers masked_ssim imports from scikit-image, from their documented behaviour."""
    def slice_at_axis(sl, axis):
        return (slice(None),) * axis + (sl,) + (...,)

    def _supported_float_type(dtype):
        dtype = np.dtype(dtype)
        return np.float32 if dtype in (np.dtype(np.float16), np.dtype(np.float32)) else np.float64

    def check_shape_equality(*images):
        if not all(images[0].shape == im.shape for im in images[1:]):
            raise ValueError("Input images must have the same dimensions.")

    def crop(ar, crop_width):
        return ar[tuple(slice(crop_width, s - crop_width) for s in ar.shape)] if crop_width else ar

    utils = types.ModuleType("skimage._shared.utils")
    utils.slice_at_axis, utils._supported_float_type = slice_at_axis, _supported_float_type
    utils.check_shape_equality, utils.warn = check_shape_equality, (lambda *a, **k: None)
    shared = types.ModuleType("skimage._shared")
    shared.utils = utils
    arraycrop = types.ModuleType("skimage.util.arraycrop")
    arraycrop.crop = crop
    util = types.ModuleType("skimage.util")
    util.arraycrop = arraycrop
    sk = types.ModuleType("skimage")
    sk._shared, sk.util = shared, util
    for name, m in [("skimage", sk), ("skimage._shared", shared), ("skimage._shared.utils", utils),
                    ("skimage.util", util), ("skimage.util.arraycrop", arraycrop)]:
        sys.modules[name] = m


def reference_masked_ssim():
    lines = SRC.read_text().splitlines()
    start = next(i for i, l in enumerate(lines) if l.startswith("def masked_ssim("))
    end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith("def "))
    ns = {"np": np, "functools": functools}
    _install_skimage_helper_stubs()
    exec(compile("\n".join(lines[start:end]), str(SRC), "exec"), ns)
    return ns["masked_ssim"]


def cases():
    """Seeded image pairs in [0, 1] and masks: (name, im1, im2, mask, kwargs)."""
    g = np.random.default_rng(2024)
    out = []
    for name, shape, dt, noise in [("f32_rgb", (3, 40, 56), np.float32, 0.08), ("f64_rgb", (3, 33, 47), np.float64, 0.2),
                                   ("f32_small_noise", (3, 64, 64), np.float32, 0.01)]:
        base = g.random(shape)
        # smooth it a little so that the structure term matters
        base = (base + np.roll(base, 1, -1) + np.roll(base, 1, -2) + np.roll(base, 2, -1)) / 4.0
        im1 = base.astype(dt)
        im2 = np.clip(base + noise * g.standard_normal(shape), 0.0, 1.0).astype(dt)
        yy, xx = np.mgrid[:shape[1], :shape[2]]
        blob = ((yy - shape[1] * 0.45) ** 2 + (xx - shape[2] * 0.55) ** 2) < (0.3 * min(shape[1:])) ** 2
        out.append((name + "_blob", im1, im2, blob, {}))
        out.append((name + "_full", im1, im2, np.ones(shape[1:], bool), {}))
    im1, im2 = out[0][1], out[0][2]
    out.append(("f32_rgb_win11", im1, im2, out[0][3], {"win_size": 11}))
    out.append(("f32_gray_2d", im1[0], im2[0], out[0][3], {"channel_axis": None}))
    out.append(("f32_channels_last", np.moveaxis(im1, 0, -1).copy(), np.moveaxis(im2, 0, -1).copy(), out[0][3],
                {"channel_axis": -1}))
    return out


def main():
    f = reference_masked_ssim()
    res = {}
    for name, a, b, m, kw in cases():
        v = f(a, b, m, **kw)
        res[name] = np.asarray(v, dtype=np.float64)
        print(f"{name:24s} mssim_all {v[0]:.12f}  mssim_mask {v[1]:.12f}")
    OUT.mkdir(parents=True, exist_ok=True)
    torch.save({"values": {k: torch.from_numpy(v) for k, v in res.items()}, "rng_seed": 2024,
                "source": "scripts/eval_utils.py masked_ssim, executed unmodified"}, OUT / "metrics_kat.pt")


if __name__ == "__main__":
    main()
