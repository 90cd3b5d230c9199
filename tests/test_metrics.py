"""CPU: gcd_amd.metrics (scripts/test.py:346-496, eval_utils.py:571-666) against the
brute-force oracle (oracle/metrics_ref.py) and closed-form known answers.  scikit-image is absent
here, see the oracle's header for what pins these."""
import numpy as np
import pytest

from gcd_amd import metrics as M
from oracle import metrics_ref as R


def _pair(seed, h=20, w=26, noise=0.08):
    g = np.random.default_rng(seed)
    a = g.random((3, h, w)).astype(np.float32)
    b = np.clip(a + noise * g.standard_normal((3, h, w)).astype(np.float32), 0, 1)
    return a, b


def test_psnr_known_answers():
    a = np.full((3, 8, 8), 0.5, np.float32)
    assert M.peak_signal_noise_ratio(a + 0.1, a) == pytest.approx(20.0, abs=1e-5)
    assert np.isinf(M.peak_signal_noise_ratio(a, a))
    x, y = _pair(0)
    assert M.peak_signal_noise_ratio(y, x) == pytest.approx(R.psnr(y, x), rel=1e-12)
    with pytest.raises(ValueError):
        M.peak_signal_noise_ratio(a, a[:, :4])


def test_ssim_closed_form_and_bruteforce():
    a = np.full((3, 16, 16), 0.3)
    b = np.full((3, 16, 16), 0.6)
    C1 = 0.01 ** 2
    # constant images: all (co)variances are 0 -> S = (2ab + C1) / (a^2 + b^2 + C1)
    assert M.structural_similarity(a, b) == pytest.approx((2 * 0.18 + C1) / (0.09 + 0.36 + C1), rel=1e-9)
    assert M.structural_similarity(a, a) == pytest.approx(1.0)
    for seed in (1, 2):
        x, y = _pair(seed)
        assert M.structural_similarity(y, x) == pytest.approx(R.ssim(y.astype(np.float64), x.astype(np.float64)), abs=2e-6)
    x, y = _pair(3)
    assert M.structural_similarity(y.astype(np.float64), x.astype(np.float64)) == \
        pytest.approx(R.ssim(y, x), abs=1e-12)
    with pytest.raises(ValueError):
        M.structural_similarity(x, y, win_size=6)
    with pytest.raises(ValueError):
        M.structural_similarity(x[:, :5], y[:, :5])


def test_masked_ssim_vs_bruteforce_and_full_mask():
    x, y = _pair(4, 24, 30)
    x, y = x.astype(np.float64), y.astype(np.float64)
    mask = np.zeros((24, 30), bool)
    mask[4:20, 6:25] = True
    mask[10:14, 12:16] = False                                  # a hole: erosion must respect it
    got = M.masked_ssim(y, x, mask)
    want = R.masked_ssim(y, x, mask)
    assert got == pytest.approx(want, abs=1e-12)
    assert got[0] == pytest.approx(M.structural_similarity(y, x), abs=1e-12)
    full = M.masked_ssim(y, x, np.ones((24, 30), bool))
    # an all-true mask erodes from the border by the window radius and is cropped by it again
    assert np.isfinite(full).all()
    assert np.isnan(M.masked_ssim(y, x, np.zeros((24, 30), bool))[1])


def test_calculate_metrics_keys_shapes_and_masks():
    T, H, W = 3, 16, 20
    g = np.random.default_rng(5)
    gt = g.random((T, 3, H, W)).astype(np.float32)
    samples = [{"sampled_rgb": np.clip(gt + 0.05 * g.standard_normal(gt.shape).astype(np.float32), 0, 1)}
               for _ in range(2)]
    md, unc = M.calculate_metrics(gt, None, samples)
    assert set(md) == {"frame_psnr", "frame_ssim", "frame_diversity", "mean_psnr", "mean_ssim", "mean_diversity"}
    assert md["frame_psnr"].shape == (2, T) and unc.shape == (T, H, W)
    assert md["mean_psnr"][0] == pytest.approx(np.mean([R.psnr(samples[0]["sampled_rgb"][t], gt[t]) for t in range(T)]))
    reproj = gt.copy()
    reproj[:, :, :, :8] = 0.0                                    # left part occluded (no re-projected colour)
    reproj[2] = 0.0                                              # frame 2 fully occluded
    md, _ = M.calculate_metrics(gt, reproj, samples)
    for k in ("frame_psnr_vis", "frame_ssim_vis", "frame_psnr_occ", "frame_ssim_occ", "frame_diversity_vis",
              "frame_diversity_occ", "mean_psnr_vis", "mean_ssim_occ", "mean_diversity_vis", "mean_diversity_occ"):
        assert k in md
    assert np.isnan(md["frame_psnr_vis"][0, 2]) and np.isfinite(md["frame_psnr_occ"][0, 2])
    vis = np.tile((np.abs(reproj[0]).sum(0) > 1e-7)[None], (3, 1, 1))
    assert md["frame_psnr_vis"][1, 0] == pytest.approx(
        R.psnr(samples[1]["sampled_rgb"][0][vis], gt[0][vis]))
    assert md["frame_ssim_vis"][1, 0] == pytest.approx(
        R.masked_ssim(samples[1]["sampled_rgb"][0].astype(np.float64), gt[0].astype(np.float64), vis[0])[1], abs=2e-6)
    assert md["mean_diversity"] > 0


# ---------------------------------------------------------------------------------------------------------------
# Pinned to the REFERENCE'S OWN SSIM code: scripts/eval_utils.py:571-666 (masked_ssim, the reference authors'
# adaptation of skimage 0.22.0's structural_similarity) executed unmodified by oracle/make_golden_metrics.py
# ---------------------------------------------------------------------------------------------------------------
def test_ssim_and_masked_ssim_vs_reference_code_golden():
    from pathlib import Path
    import torch
    from gcd_amd import metrics as M
    from oracle.make_golden_metrics import cases
    g = torch.load(Path(__file__).resolve().parent / "golden" / "metrics_kat.pt")["values"]
    seen = 0
    for name, a, b, m, kw in cases():
        want = g[name].numpy()
        got = M.masked_ssim(a, b, m, **kw)
        tol = 1e-6 if a.dtype == np.float32 else 1e-12
        assert np.allclose(got, want, rtol=0, atol=tol), (name, got, want)
        # the un-masked value is what skimage.metrics.structural_similarity returns (test.py:386-420)
        s = M.structural_similarity(a, b, data_range=1.0, channel_axis=kw.get("channel_axis", 0),
                                    win_size=kw.get("win_size", 7))
        assert abs(s - want[0]) <= tol, (name, s, want[0])
        if name.endswith("_full"):
            assert abs(want[0] - want[1]) <= 1e-12          # an all-true mask changes nothing
        seen += 1
    assert seen == len(g) == 9
