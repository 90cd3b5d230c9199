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
        # no imageio / ffmpeg: the .mp4 is still written, as Motion-JPEG in an MP4 container (gcd_amd/mp4_mjpeg.py),
        # cropped to the macro block like the reference's (24 x 40 -> 16 x 32), same frame count and fps
        from gcd_amd.mp4_mjpeg import read_mp4_mjpeg
        assert out2["video"] == str(tmp_path / "d") + ".mp4" and os.path.exists(out2["video"])
        assert "Motion-JPEG" in capsys.readouterr().out
        back = read_mp4_mjpeg(out2["video"])
        assert (back["width"], back["height"], len(back["frames"]), back["fps"], back["oti"]) == (32, 16, 4, 10.0, 0x6C)
        for a, b in zip(back["frames"], want[:, :16, :32]):
            assert np.abs(a.astype(int) - b.astype(int)).mean() < 24.0       # random-noise frames through JPEG
    with pytest.raises(AssertionError):
        eval_io.write_video_and_frames(frames)


def test_mp4_mjpeg_container_structure(tmp_path):
    """gcd_amd/mp4_mjpeg.py: box tree, sample table and decoder configuration of the written file; smooth frames come
    back within JPEG accuracy."""
    pytest.importorskip("PIL.Image")
    import struct
    from gcd_amd.mp4_mjpeg import _children, read_mp4_mjpeg, write_mp4_mjpeg
    yy, xx = np.mgrid[:72, :104]
    frames = [np.stack([(xx * 2 + 5 * t) % 256, (yy * 3 + t) % 256, (xx + yy + 9 * t) % 256], -1).astype(np.uint8)
              for t in range(14)]
    p = str(tmp_path / "v.mp4")
    info = write_mp4_mjpeg(p, frames, fps=7.5, quality=8)
    assert info["frames"] == 14 and info["fps"] == 7.5 and os.path.getsize(p) == info["bytes"]
    buf = open(p, "rb").read()
    assert [k for k, _, _ in _children(buf, 0, len(buf))] == [b"ftyp", b"mdat", b"moov"]
    assert struct.unpack(">I", buf[:4])[0] == 28 and buf[4:12] == b"ftypisom"
    back = read_mp4_mjpeg(p)
    assert back["oti"] == 0x6C and back["fps"] == 7.5 and abs(back["duration_s"] - 14 / 7.5) < 1e-9
    assert (back["width"], back["height"]) == (104, 72) and len(back["frames"]) == 14
    for a, b in zip(back["frames"], frames):
        assert a.shape == b.shape and np.abs(a.astype(int) - b.astype(int)).mean() < 2.0
    with pytest.raises(ValueError):
        write_mp4_mjpeg(p, [], fps=10)
    with pytest.raises(ValueError):
        write_mp4_mjpeg(p, [frames[0], frames[1][:, :50]], fps=10)
