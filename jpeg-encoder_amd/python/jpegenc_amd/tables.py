"""Sampling schemes and quantization tables of the encode path.

Sampling schemes mirror ``SamplingDefinition`` (reference SamplingScheme.h:25-94): one
``(widthFactor, widthMethod, heightFactor, heightMethod)`` tuple per channel, method
0 = Subsampling (keep the first sample of each group), 1 = Average.  Quantization tables
are 64 floats in natural (row-major) order as in ``QTable`` (QuantizationTables.h:6-110).
"""
import numpy as np

SUBSAMPLING = 0
AVERAGE = 1


def scheme(y, cb, cr):
    return np.array(list(y) + list(cb) + list(cr), np.int32)


_FULL = (1, SUBSAMPLING, 1, SUBSAMPLING)
SCHEME_444 = scheme(_FULL, _FULL, _FULL)                                             # SamplingScheme.h:48-51
SCHEME_422 = scheme(_FULL, (2, SUBSAMPLING, 1, SUBSAMPLING), (2, SUBSAMPLING, 1, SUBSAMPLING))   # :53-56
SCHEME_411 = scheme(_FULL, (4, SUBSAMPLING, 1, SUBSAMPLING), (4, SUBSAMPLING, 1, SUBSAMPLING))   # :58-61
SCHEME_420 = scheme(_FULL, (2, AVERAGE, 2, AVERAGE), (2, AVERAGE, 2, AVERAGE))       # :63-66
SCHEME_422_AVERAGE = scheme(_FULL, (2, AVERAGE, 1, SUBSAMPLING), (2, AVERAGE, 1, SUBSAMPLING))   # :70-73
SCHEME_422_HEIGHT = scheme(_FULL, (1, SUBSAMPLING, 2, SUBSAMPLING), (1, SUBSAMPLING, 2, SUBSAMPLING))  # :75-78
SCHEME_422_HEIGHT_AVERAGE = scheme(_FULL, (1, SUBSAMPLING, 2, AVERAGE), (1, SUBSAMPLING, 2, AVERAGE))  # :80-83
SCHEME_311 = scheme(_FULL, (3, SUBSAMPLING, 1, SUBSAMPLING), (3, SUBSAMPLING, 1, SUBSAMPLING))   # :85-88
SCHEME_321 = scheme(_FULL, (3, SUBSAMPLING, 2, SUBSAMPLING), (3, SUBSAMPLING, 2, SUBSAMPLING))   # :90-93

SCHEMES = {
    "444": SCHEME_444, "422": SCHEME_422, "411": SCHEME_411, "420": SCHEME_420,
    "422avg": SCHEME_422_AVERAGE, "422h": SCHEME_422_HEIGHT, "422havg": SCHEME_422_HEIGHT_AVERAGE,
    "311": SCHEME_311, "321": SCHEME_321,
}


def _lcm(a, b):
    return a * b // int(np.gcd(a, b))


def step_size(sch):
    """MCU size in pixels (SamplingScheme.h:36-38)."""
    sch = np.asarray(sch).reshape(3, 4)
    w = _lcm(_lcm(int(sch[0, 0]), int(sch[1, 0])), int(sch[2, 0])) * 8
    h = _lcm(_lcm(int(sch[0, 2]), int(sch[1, 2])), int(sch[2, 2])) * 8
    return w, h


def inverse_factors(sch):
    """JPEG H and V sampling factors per channel (SamplingScheme.h:39-42)."""
    sch = np.asarray(sch).reshape(3, 4)
    w, h = step_size(sch)
    return [(w // int(sch[c, 0]) // 8, h // int(sch[c, 2]) // 8) for c in range(3)]


def padded_size(width, height, sch):
    """Image::Image, Image.cpp:8-10."""
    sw, sh = step_size(sch)
    return (width + sw - 1) // sw * sw, (height + sh - 1) // sh * sh


# QuantizationTables.h:40-60 (Annex K.1 / K.2)
LUMINANCE = np.array([
    16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55,
    14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87, 80, 62,
    18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92,
    49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99], np.float32)
CHROMINANCE = np.array([
    17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99,
    24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99] + [99] * 32, np.float32)
# QuantizationTables.h:69-110
PHOTOSHOP_LUMINANCE12 = np.array([
    1, 1, 1, 1, 1, 1, 1, 2, 1, 1, 1, 1, 1, 1, 1, 2, 1, 1, 1, 1, 1, 1, 2, 2, 1, 1, 1, 1, 1, 2, 2, 3,
    1, 1, 1, 1, 2, 2, 3, 3, 1, 1, 1, 2, 2, 3, 3, 3, 1, 1, 2, 2, 3, 3, 3, 3, 2, 2, 2, 3, 3, 3, 3, 3], np.float32)
PHOTOSHOP_CHROMINANCE12 = np.array([
    1, 1, 1, 2, 3, 3, 3, 3, 1, 1, 1, 2, 3, 3, 3, 3, 1, 1, 2, 3, 3, 3, 3, 3, 2, 2, 3, 3, 3, 3, 3, 3] + [3] * 32,
    np.float32)
PHOTOSHOP_LUMINANCE5 = np.array([
    12, 8, 13, 21, 26, 32, 34, 17, 8, 9, 12, 20, 27, 23, 12, 12, 13, 12, 16, 26, 23, 12, 12, 12,
    21, 20, 26, 23, 12, 12, 12, 12, 26, 27, 23, 12, 12, 12, 12, 12, 32, 23, 12, 12, 12, 12, 12, 12,
    34, 12, 12, 12, 12, 12, 12, 12, 17, 12, 12, 12, 12, 12, 12, 12], np.float32)
PHOTOSHOP_CHROMINANCE5 = np.array([
    13, 13, 17, 27, 20, 20, 17, 17, 13, 14, 17, 14, 14, 12, 12, 12, 17, 17, 14, 14, 12, 12, 12, 12,
    27, 14, 14, 12, 12, 12, 12, 12, 20, 14, 12, 12, 12, 12, 12, 12, 20, 12, 12, 12, 12, 12, 12, 12,
    17, 12, 12, 12, 12, 12, 12, 12, 17, 12, 12, 12, 12, 12, 12, 12], np.float32)


def qtable_set(luminance, chrominance, chrominance2=None):
    """``QTableSet(lum, chrom[, chrom2])`` (QuantizationTables.h:11-27) as float32 [3, 64]."""
    if chrominance2 is None:
        chrominance2 = chrominance
    return np.stack([np.asarray(luminance, np.float32).reshape(64), np.asarray(chrominance, np.float32).reshape(64),
                     np.asarray(chrominance2, np.float32).reshape(64)])


def quality_tables(quality):
    """Annex-K tables scaled by the IJG quality rule, clamped to 1..255.

    The reference has only fixed tables; its ``QTableSet`` accepts any, so a quality
    setting is expressed as a table pair.  quality 100 gives all ones.
    """
    quality = int(quality)
    if not 1 <= quality <= 100:
        raise ValueError("quality must be in 1..100")
    scale = 5000 // quality if quality < 50 else 200 - 2 * quality

    def one(base):
        t = (base.astype(np.int64) * scale + 50) // 100
        return np.clip(t, 1, 255).astype(np.float32)

    return qtable_set(one(LUMINANCE), one(CHROMINANCE))


STANDARD_TABLES = qtable_set(LUMINANCE, CHROMINANCE)
NAMED_TABLES = {
    "annexk": STANDARD_TABLES,
    "ps12": qtable_set(PHOTOSHOP_LUMINANCE12, PHOTOSHOP_CHROMINANCE12),
    "ps5": qtable_set(PHOTOSHOP_LUMINANCE5, PHOTOSHOP_CHROMINANCE5),
    "ones": qtable_set(np.ones(64), np.ones(64)),
}

ZIGZAG = np.array([
    0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47,
    55, 62, 63], np.int32)   # ZigZag.h:10-19
