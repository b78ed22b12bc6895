"""Deterministic, integer-only synthetic RGB8 images (same bytes on every machine).

``mix32`` is a 32-bit avalanche hash; ``noise`` is uniform bytes, ``natural`` is a set of
colour ramps with +-8 of hashed dither, which compresses like a smooth photograph.
"""
import numpy as np


def mix32(x):
    x = x.astype(np.uint32, copy=True)
    x ^= x >> np.uint32(16)
    x *= np.uint32(0x7FEB352D)
    x ^= x >> np.uint32(15)
    x *= np.uint32(0x846CA68B)
    x ^= x >> np.uint32(16)
    return x


def _rows(width, y0, y1):
    ys = np.arange(y0, y1, dtype=np.uint32)[:, None, None]
    xs = np.arange(width, dtype=np.uint32)[None, :, None]
    cs = np.arange(3, dtype=np.uint32)[None, None, :]
    return ys, xs, cs


def noise(width, height, seed=42, y0=0, y1=None, full_width=None):
    """byte(x, y, c) = mix32(seed ^ (3*(y*W + x) + c)) >> 24."""
    y1 = height if y1 is None else y1
    fw = width if full_width is None else full_width
    with np.errstate(over="ignore"):
        ys, xs, cs = _rows(width, y0, y1)
        idx = (ys * np.uint32(fw) + xs) * np.uint32(3) + cs
        return (mix32(np.uint32(seed) ^ idx) >> np.uint32(24)).astype(np.uint8)


def natural(width, height, seed=1, y0=0, y1=None, full_width=None):
    """Triangle-wave colour ramps plus a 4-bit hashed dither, clamped to 0..255."""
    y1 = height if y1 is None else y1
    fw = width if full_width is None else full_width
    with np.errstate(over="ignore"):
        ys, xs, cs = _rows(width, y0, y1)
        idx = (ys * np.uint32(fw) + xs) * np.uint32(3) + cs
        t = (xs * (np.uint32(3) + cs) + ys * (np.uint32(5) - cs)) >> np.uint32(3)
        lo = (t & np.uint32(255)).astype(np.int32)
        tri = np.where(((t >> np.uint32(8)) & np.uint32(1)) != 0, 255 - lo, lo)
        dither = (mix32(np.uint32(seed) ^ idx) >> np.uint32(28)).astype(np.int32) - 8
        return np.clip(tri + dither, 0, 255).astype(np.uint8)


def mixed_band(height):
    """Rows of the "mixed" image that hold noise: a 128th of the rows, from the middle down."""
    return height // 2, height // 2 + height // 128


def generate(kind, width, height, seed=None, chunk_rows=1024):
    """Whole image as uint8 [H, W, 3], produced in row chunks to bound temporaries.

    "mixed": natural content (seed 1) with a narrow band of noise (seed 42) across the middle -- an
    image whose mean statistics say "photographic" while a few of its tiles are as dense as noise."""
    if kind == "mixed":
        out = generate("natural", width, height, 1, chunk_rows)
        y0, y1 = mixed_band(height)
        if y1 > y0:
            out[y0:y1] = noise(width, height, 42, y0, y1)
        return out
    fn = {"noise": noise, "natural": natural}[kind]
    if seed is None:
        seed = 42 if kind == "noise" else 1
    out = np.empty((height, width, 3), np.uint8)
    for y0 in range(0, height, chunk_rows):
        y1 = min(height, y0 + chunk_rows)
        out[y0:y1] = fn(width, height, seed, y0, y1)
    return out
