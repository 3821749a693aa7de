"""Framebuffer dump.  The reference has no image writer: its frame only ever reaches the
window (RayTracer.cs:135); the closest things are ColorHelper.ColorToInt (ColorHelper.cs:16-23, clamp then
round(c*255)) and Surface's pixel array (surface.cs:13-24).  The library's RGBA8 plane already applies
that quantisation rule on the GPU; these writers only fix the row order -- the library's row 0 is the
BOTTOM row (gl_FragCoord), image files start at the top -- and encode.  No dependencies beyond zlib.
"""
import struct
import zlib

import numpy as np


def _top_down(rgba8):
    a = np.asarray(rgba8)
    if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] not in (3, 4):
        raise ValueError("expected an [H, W, 3|4] uint8 array (the rgba8 plane of RenderGL)")
    return np.ascontiguousarray(a[::-1, :, :3])


def write_ppm(path, rgba8):
    """Binary PPM (P6), top row first."""
    px = _top_down(rgba8)
    with open(path, "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (px.shape[1], px.shape[0]))
        f.write(px.tobytes())


def write_png(path, rgba8, level=6):
    """8-bit RGB PNG (filter 0 on every scanline), top row first."""
    px = _top_down(rgba8)
    h, w = px.shape[:2]
    raw = np.concatenate([np.zeros((h, 1), np.uint8), px.reshape(h, w * 3)], axis=1).tobytes()

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n")
        f.write(chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)))
        f.write(chunk(b"IDAT", zlib.compress(raw, level)))
        f.write(chunk(b"IEND", b""))


def quantise(rgb):
    """The RGBA8 rule of include/rt_mi355x.h on the host (for float planes fetched with want=("rgb",)):
    round-half-even(clamp(c, 0, 1) * 255), NaN -> 0, alpha 255."""
    c = np.nan_to_num(np.asarray(rgb, np.float32), nan=0.0)
    q = np.rint(np.clip(c, 0.0, 1.0) * np.float32(255.0)).astype(np.uint8)
    return np.concatenate([q, np.full(q.shape[:2] + (1,), 255, np.uint8)], axis=2)
