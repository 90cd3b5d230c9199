"""Frame / video writers of the evaluation scripts (the IO half).

`write_video_and_frames` mirrors `gcd-model/scripts/eval_utils.py:510-568`: same signature, the same files
(`<dst_dp>/0000.png ...` via `matplotlib.pyplot.imsave`, `<dst_fp>.mp4` via `imageio.mimwrite(format='ffmpeg')`),
the same float -> uint8 conversion (`(x * 255).astype(uint8)`: truncation, not rounding) and the same crop to
a multiple of the codec's macro block AFTER the frames were saved.  Like the reference it does not raise when the
video cannot be written (there: 10 attempts, each failure printed).  Without `imageio` / ffmpeg the `.mp4` is still
written — same frames, same fps — as Motion-JPEG in an MP4 container (`gcd_amd/mp4_mjpeg.py`).  CPU / numpy work,
not on the HIP path.
"""
from __future__ import annotations

import os
import pathlib
import time
from typing import Optional, Sequence, Union

import numpy as np

__all__ = ["to_uint8_frames", "crop_to_multiple", "write_video_and_frames"]


def to_uint8_frames(images: Union[np.ndarray, Sequence[np.ndarray]]) -> np.ndarray:
    """(T, H, W, 3) float in [0, 1] or uint8 -> uint8, as eval_utils.py:520-523 (a list is stacked first)."""
    if isinstance(images, (list, tuple)):
        images = np.stack(images)
    images = np.asarray(images)
    if images.dtype in (np.float16, np.float32, np.float64):
        images = (images * 255.0).astype(np.uint8)
    return images


def crop_to_multiple(images: np.ndarray, crop_multiple: Optional[int]) -> np.ndarray:
    """Top-left crop of (T, H, W, C) frames to multiples of `crop_multiple` (eval_utils.py:535-540)."""
    if crop_multiple is not None and crop_multiple > 1:
        H, W = images.shape[1:3]
        images = images[:, 0:crop_multiple * (H // crop_multiple), 0:crop_multiple * (W // crop_multiple)]
    return images


def write_video_and_frames(images, dst_dp: Optional[str] = None, dst_fp: Optional[str] = None, fps: float = 10,
                           save_images: bool = True, save_mp4: bool = True,
                           crop_multiple: Optional[int] = None, quality: int = 8,
                           max_attempts: int = 10, retry_sleep: float = 1.0) -> dict:
    """eval_utils.py:510-568.  Returns what was written: {"frames": [paths], "video": path or None}."""
    if dst_dp is not None and dst_fp is None:
        dst_fp = dst_dp
    elif dst_fp is not None and dst_dp is None:
        dst_dp = os.path.splitext(dst_fp)[0] + "_frames"
    assert dst_dp is not None and dst_fp is not None
    images = to_uint8_frames(images)
    written = {"frames": [], "video": None}

    if save_images:
        import matplotlib
        matplotlib.use("Agg", force=False)
        import matplotlib.pyplot as plt
        os.makedirs(dst_dp, exist_ok=True)
        print(f"Saving frames as images to: {dst_dp}")
        for i, image in enumerate(images):
            fp = os.path.join(dst_dp, f"{i:04d}.png")
            plt.imsave(fp, image)
            written["frames"].append(fp)

    if save_mp4:
        parent = os.path.dirname(dst_fp)
        if parent:
            os.makedirs(parent, exist_ok=True)
        os.makedirs(str(pathlib.Path(dst_fp).parent), exist_ok=True)
        frames = list(crop_to_multiple(images, crop_multiple))
        try:
            import imageio
        except ImportError as e:
            # No imageio / ffmpeg (the reference's H.264 writer): still produce `<dst_fp>.mp4` — the same frames at
            # the same fps as Motion-JPEG in an MP4 container (gcd_amd/mp4_mjpeg.py, PIL for the JPEGs).
            print(f"imageio is not available ({e}): writing Motion-JPEG video to: {dst_fp}.mp4")
            try:
                from .mp4_mjpeg import write_mp4_mjpeg
                info = write_mp4_mjpeg(dst_fp + ".mp4", frames, fps=float(fps), quality=quality)
                written["video"] = dst_fp + ".mp4"
                written["video_codec"] = info["codec"]
            except Exception as e2:   # noqa: BLE001 — like the reference, a failed video never raises
                print(f"Error saving video: {e2}")
            return written
        if hasattr(os, "sched_setaffinity"):   # eval_utils.py:546-549: ffmpeg should see every core
            try:
                os.sched_setaffinity(0, set(range(os.cpu_count() or 1)))
            except OSError:
                pass
        print(f"Saving video to: {dst_fp}.mp4")
        for i in range(max_attempts):
            try:
                imageio.mimwrite(dst_fp + ".mp4", frames, format="ffmpeg", fps=float(fps),
                                 macro_block_size=crop_multiple, quality=quality)
                written["video"] = dst_fp + ".mp4"
                break
            except Exception as e:   # noqa: BLE001 — the reference catches everything and retries
                print(f"Error saving video: {e}")
                if i <= max_attempts - 2:
                    print(f"Retrying (attempt {i + 1})...")
                time.sleep(retry_sleep)
    return written
