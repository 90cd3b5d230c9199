"""A minimal MP4 (ISO/IEC 14496-12) writer for Motion-JPEG video, used by `eval_io.write_video_and_frames` when
imageio / ffmpeg — what the reference writes its H.264 `.mp4` with (scripts/eval_utils.py:553-566) — is not installed.

One video track, every frame a baseline JPEG (PIL) and a sync sample, carried as an `mp4v` sample entry whose
`esds` DecoderConfigDescriptor names objectTypeIndication 0x6C (Visual ISO/IEC 10918-1, JPEG) — the registered way to
put JPEG into MP4, which ffmpeg / VLC / browsers' demuxers understand.  Layout: ftyp | mdat (all samples, one chunk)
| moov.  `read_mp4_mjpeg` parses such a file back (tests; it is not a general MP4 reader).  Host-side IO, no HIP.
"""
from __future__ import annotations

import io
import struct
from typing import List, Sequence

import numpy as np


def _box(kind: bytes, payload: bytes) -> bytes:
    return struct.pack(">I4s", 8 + len(payload), kind) + payload


def _full(kind: bytes, version: int, flags: int, payload: bytes) -> bytes:
    return _box(kind, struct.pack(">I", (version << 24) | flags) + payload)


def _descr(tag: int, payload: bytes) -> bytes:
    n = len(payload)
    assert n < (1 << 28)
    size = bytes([0x80 | ((n >> 21) & 0x7F), 0x80 | ((n >> 14) & 0x7F), 0x80 | ((n >> 7) & 0x7F), n & 0x7F])
    return bytes([tag]) + size + payload


_MATRIX = struct.pack(">9I", 0x00010000, 0, 0, 0, 0x00010000, 0, 0, 0, 0x40000000)


def encode_jpegs(frames: Sequence[np.ndarray], quality: int = 8) -> List[bytes]:
    """uint8 (H, W, 3) frames -> baseline JPEGs.  `quality` is imageio's 0..10 scale (the reference passes 8)."""
    from PIL import Image
    q = int(min(95, max(10, round(10 + 8.5 * float(quality)))))
    out = []
    for f in frames:
        f = np.ascontiguousarray(f)
        assert f.dtype == np.uint8 and f.ndim == 3 and f.shape[2] == 3, "frames must be uint8 (H, W, 3)"
        buf = io.BytesIO()
        Image.fromarray(f, "RGB").save(buf, format="JPEG", quality=q, subsampling=0, optimize=False)
        out.append(buf.getvalue())
    return out


def write_mp4_mjpeg(path: str, frames: Sequence[np.ndarray], fps: float = 10.0, quality: int = 8) -> dict:
    frames = list(frames)
    if not frames:
        raise ValueError("write_mp4_mjpeg: no frames")
    H, W = frames[0].shape[:2]
    if any(f.shape[:2] != (H, W) for f in frames):
        raise ValueError("write_mp4_mjpeg: frames differ in size")
    if W >= 65536 or H >= 65536:
        raise ValueError("write_mp4_mjpeg: frame too large for an MP4 visual sample entry")
    jpegs = encode_jpegs(frames, quality)
    n = len(jpegs)
    delta = 1000
    timescale = max(1, int(round(float(fps) * delta)))
    duration = n * delta
    ftyp = _box(b"ftyp", b"isom" + struct.pack(">I", 0x200) + b"isomiso2mp41")
    mdat_payload = b"".join(jpegs)
    if len(ftyp) + 8 + len(mdat_payload) >= (1 << 32) - (1 << 20):
        raise ValueError("write_mp4_mjpeg: > 4 GiB of samples (64-bit boxes are not implemented)")
    mdat = _box(b"mdat", mdat_payload)
    chunk_offset = len(ftyp) + 8
    sizes = [len(j) for j in jpegs]
    total_bits = 8 * sum(sizes)
    avg_bitrate = int(total_bits * timescale / max(duration, 1))
    dcd = _descr(0x04, bytes([0x6C, 0x11]) + struct.pack(">I", max(sizes))[1:] +
                 struct.pack(">II", int(8 * max(sizes) * timescale / delta), avg_bitrate))
    esds = _full(b"esds", 0, 0, _descr(0x03, struct.pack(">HB", 0, 0) + dcd + _descr(0x06, b"\x02")))
    name = b"gcd_amd MJPEG"
    visual = (b"\x00" * 6 + struct.pack(">H", 1) + b"\x00" * 16 + struct.pack(">HH", W, H) +
              struct.pack(">II", 0x00480000, 0x00480000) + struct.pack(">I", 0) + struct.pack(">H", 1) +
              bytes([len(name)]) + name + b"\x00" * (31 - len(name)) + struct.pack(">Hh", 0x0018, -1) + esds)
    stsd = _full(b"stsd", 0, 0, struct.pack(">I", 1) + _box(b"mp4v", visual))
    stts = _full(b"stts", 0, 0, struct.pack(">III", 1, n, delta))
    stsc = _full(b"stsc", 0, 0, struct.pack(">IIII", 1, 1, n, 1))
    stsz = _full(b"stsz", 0, 0, struct.pack(">II", 0, n) + struct.pack(f">{n}I", *sizes))
    stco = _full(b"stco", 0, 0, struct.pack(">II", 1, chunk_offset))
    stbl = _box(b"stbl", stsd + stts + stsc + stsz + stco)
    dinf = _box(b"dinf", _full(b"dref", 0, 0, struct.pack(">I", 1) + _full(b"url ", 0, 1, b"")))
    vmhd = _full(b"vmhd", 0, 1, struct.pack(">HHHH", 0, 0, 0, 0))
    minf = _box(b"minf", vmhd + dinf + stbl)
    hdlr = _full(b"hdlr", 0, 0, struct.pack(">I4sIII", 0, b"vide", 0, 0, 0) + b"VideoHandler\x00")
    mdhd = _full(b"mdhd", 0, 0, struct.pack(">IIIIHH", 0, 0, timescale, duration, 0x55C4, 0))
    mdia = _box(b"mdia", mdhd + hdlr + minf)
    tkhd = _full(b"tkhd", 0, 3, struct.pack(">IIIII", 0, 0, 1, 0, duration) + struct.pack(">IIhhhH", 0, 0, 0, 0, 0, 0) +
                 _MATRIX + struct.pack(">II", W << 16, H << 16))
    trak = _box(b"trak", tkhd + mdia)
    mvhd = _full(b"mvhd", 0, 0, struct.pack(">IIII", 0, 0, timescale, duration) + struct.pack(">IH", 0x00010000, 0x0100) +
                 b"\x00" * 10 + _MATRIX + b"\x00" * 24 + struct.pack(">I", 2))
    moov = _box(b"moov", mvhd + trak)
    with open(path, "wb") as f:
        f.write(ftyp)
        f.write(mdat)
        f.write(moov)
    return {"frames": n, "width": W, "height": H, "fps": timescale / delta, "bytes": len(ftyp) + len(mdat) + len(moov),
            "codec": "mjpeg (mp4v, objectTypeIndication 0x6C)"}


# ---------------------------------------------------------------------------------------------------------------
def _children(buf: bytes, start: int, end: int):
    pos = start
    while pos + 8 <= end:
        size, kind = struct.unpack(">I4s", buf[pos:pos + 8])
        if size < 8 or pos + size > end:
            raise ValueError(f"bad box {kind!r} at {pos}")
        yield kind, pos + 8, pos + size
        pos += size


def _find(buf: bytes, path: Sequence[bytes], start: int, end: int):
    for kind, s, e in _children(buf, start, end):
        if kind == path[0]:
            return (s, e) if len(path) == 1 else _find(buf, path[1:], s, e)
    raise KeyError(path[0])


def read_mp4_mjpeg(path: str) -> dict:
    """Parse a file written by `write_mp4_mjpeg`: {"frames": [uint8 (H, W, 3)], "fps", "width", "height", "oti"}."""
    from PIL import Image
    buf = open(path, "rb").read()
    top = {k: (s, e) for k, s, e in _children(buf, 0, len(buf))}
    assert buf[top[b"ftyp"][0]:top[b"ftyp"][0] + 4] == b"isom"
    ms, me = top[b"moov"]
    s, _ = _find(buf, [b"trak", b"mdia", b"mdhd"], ms, me)
    timescale, duration = struct.unpack(">II", buf[s + 12:s + 20])
    stbl = _find(buf, [b"trak", b"mdia", b"minf", b"stbl"], ms, me)
    s, e = _find(buf, [b"stsd"], *stbl)
    assert buf[s + 12:s + 16] == b"mp4v"
    W, H = struct.unpack(">HH", buf[s + 16 + 24:s + 16 + 28])
    esds = buf.index(b"esds", s, e)
    oti = buf[buf.index(b"\x04\x80\x80\x80", esds, e) + 5]
    s, _ = _find(buf, [b"stts"], *stbl)
    _, n, delta = struct.unpack(">III", buf[s + 4:s + 16])
    s, _ = _find(buf, [b"stsz"], *stbl)
    fixed, count = struct.unpack(">II", buf[s + 4:s + 12])
    sizes = list(struct.unpack(f">{count}I", buf[s + 12:s + 12 + 4 * count])) if fixed == 0 else [fixed] * count
    s, _ = _find(buf, [b"stco"], *stbl)
    (off,) = struct.unpack(">I", buf[s + 8:s + 12])
    assert count == n and top[b"mdat"][0] == off
    frames = []
    for sz in sizes:
        frames.append(np.asarray(Image.open(io.BytesIO(buf[off:off + sz])).convert("RGB")))
        off += sz
    assert off == top[b"mdat"][1]
    return {"frames": frames, "fps": timescale / delta, "width": W, "height": H, "oti": oti, "duration_s": duration / timescale}
