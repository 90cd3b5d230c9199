"""gcd_amd.eval_io — the frame / video writers of the evaluation scripts (scripts/eval_utils.py:510-568)."""
import os

import numpy as np
import pytest

from gcd_amd import eval_io


def test_uint8_conversion_truncates_like_the_reference():
    x = np.array([[[[0.0, 0.5, 1.0]]]], dtype=np.float32)          # (T=1, H=1, W=1, 3)
    assert eval_io.to_uint8_frames(x).tolist() == [[[[0, 127, 255]]]]   # 127.5 -> 127: astype truncates
    u = np.arange(12, dtype=np.uint8).reshape(1, 2, 2, 3)
    assert eval_io.to_uint8_frames(u) is u or np.array_equal(eval_io.to_uint8_frames(u), u)
    assert eval_io.to_uint8_frames([u[0], u[0]]).shape == (2, 2, 2, 3)


def test_crop_to_multiple():
    x = np.zeros((3, 37, 50, 3), np.uint8)
    assert eval_io.crop_to_multiple(x, 16).shape == (3, 32, 48, 3)
    assert eval_io.crop_to_multiple(x, None).shape == x.shape
    assert eval_io.crop_to_multiple(x, 1).shape == x.shape


def test_frames_round_trip_and_paths(tmp_path, capsys):
    PIL = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(0)
    frames = rng.random((4, 24, 40, 3)).astype(np.float32)
    # dst_fp only: frames go to <dst_fp without extension>_frames (eval_utils.py:514-517)
    out = eval_io.write_video_and_frames(frames, dst_fp=str(tmp_path / "clip"), save_mp4=False)
    assert [os.path.basename(p) for p in out["frames"]] == ["0000.png", "0001.png", "0002.png", "0003.png"]
    assert os.path.dirname(out["frames"][0]) == str(tmp_path / "clip_frames")
    want = (frames * 255.0).astype(np.uint8)
    for i, fp in enumerate(out["frames"]):
        got = np.asarray(PIL.open(fp).convert("RGB"))
        assert np.array_equal(got, want[i])
    # dst_dp only: the video path defaults to it; a missing imageio is reported, never raised
    out2 = eval_io.write_video_and_frames(frames, dst_dp=str(tmp_path / "d"), save_images=False,
                                          crop_multiple=16, max_attempts=1, retry_sleep=0.0)
    assert out2["frames"] == []
    try:
        import imageio  # noqa: F401
    except ImportError:
        assert out2["video"] is None
        assert "Error saving video" in capsys.readouterr().out
    with pytest.raises(AssertionError):
        eval_io.write_video_and_frames(frames)
