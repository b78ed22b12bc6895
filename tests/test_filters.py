"""Pixel filters in front of the transform (Image::applySepia, Encoder::multiplyColorChannelBy on RGB)."""
import numpy as np
import pytest

import oracle as O
from jpegenc_amd import binding as B, synth, tables as T

pytestmark = pytest.mark.gpu
F32 = np.float32


@pytest.fixture(scope="module")
def ctx():
    c = B.Context(0)
    yield c
    c.close()


def _sepia(rgb):
    """applySepiaFilterAVXImpl, SIMD256.h:218-244, in float32 with every step rounded."""
    r, g, b = rgb[..., 0], rgb[..., 1], rgb[..., 2]
    m = [(0.3930, 0.7690, 0.1890), (0.3490, 0.6860, 0.1680), (0.2720, 0.5340, 0.1310)]
    out = np.zeros_like(rgb)
    for c, (a0, a1, a2) in enumerate(m):
        t = F32(a0) * r
        t = t + F32(a1) * g
        t = t + F32(a2) * b
        out[..., c] = t + F32(0.0)
    return out


@pytest.mark.parametrize("from_bytes", [True, False])
def test_sepia_and_channel_gain_bit_exact(ctx, from_bytes):
    w, h = 123, 77
    img8 = synth.generate("natural", w, h, seed=4)
    imgf = img8.astype(F32) / F32(255.0)
    src = img8 if from_bytes else imgf
    qt = T.quality_tables(90)
    for scheme in ("444", "420"):
        prm = B.make_params(w, h, T.SCHEMES[scheme], qt)
        lay = B.compute_layout(prm)
        # sepia, then the green channel times 0.75, then the blue channel times 1.5 (may leave [0, 1])
        want_px = _sepia(imgf)
        want_px[..., 1] = want_px[..., 1] * F32(0.75)
        want_px[..., 2] = want_px[..., 2] * F32(1.5)
        ctx.begin(prm)
        ctx.upload_pixels(src)
        ctx.filter_pixels(B.FILTER_SEPIA)
        ctx.filter_pixels(B.FILTER_MULTIPLY, 1, 0.75)
        ctx.filter_pixels(B.FILTER_MULTIPLY, 2, 1.5)
        got_px = ctx.get_pixels(w, h)
        assert np.array_equal(got_px.view(np.uint32), want_px.view(np.uint32)), scheme
        ctx.transform()
        enc = O.encode(want_px, T.SCHEMES[scheme], qt)
        for c in range(3):
            assert np.array_equal(ctx.coefficients(c, lay), enc.coef[c]), (scheme, c)


def test_filters_need_pixels_and_valid_arguments(ctx):
    prm = B.make_params(16, 16, T.SCHEME_444, T.STANDARD_TABLES)
    ctx.begin(prm)
    with pytest.raises(B.JpegHipError) as e:
        ctx.filter_pixels(B.FILTER_SEPIA)
    assert e.value.status == B.ERR_BAD_STATE
    ctx.upload_pixels(np.zeros((16, 16, 3), np.uint8))
    with pytest.raises(B.JpegHipError) as e:
        ctx.filter_pixels(7)
    assert e.value.status == B.ERR_INVALID_ARGUMENT
    with pytest.raises(B.JpegHipError) as e:
        ctx.filter_pixels(B.FILTER_MULTIPLY, 3, 2.0)
    assert e.value.status == B.ERR_INVALID_ARGUMENT


@pytest.mark.parametrize("scheme", ["444", "420", "422", "411"])
def test_channel_gain_after_colour_conversion(ctx, scheme):
    # Encoder::multiplyColorChannelBy between convertToYCbCr and reduceResolutionBySchema
    # (stage 0) and between reduceResolutionBySchema and applyDCT (stage 1), Encoder.cpp:35-38.
    w, h = 123, 77
    img = synth.generate("natural", w, h, seed=9)
    qt = T.quality_tables(90)
    ops = [(1, 0.5, False), (0, 1.25, False), (2, 0.8, True), (1, 1.5, True)]
    prm = B.make_params(w, h, T.SCHEMES[scheme], qt)
    lay = B.compute_layout(prm)
    ctx.begin(prm)
    ctx.upload_pixels(img)
    for ch, val, after in ops:
        ctx.channel_gain(ch, val, after)
    ctx.transform()
    O.set_channel_gains(ops)
    try:
        enc = O.encode(img, T.SCHEMES[scheme], qt)
    finally:
        O.set_channel_gains(())
    for c in range(3):
        assert np.array_equal(ctx.coefficients(c, lay), enc.coef[c]), (scheme, c)
    ctx.build_huffman(ctx.histograms())
    assert ctx.emit_scan_host() == enc.scan
    # begin() forgets the gains: the same pixels now give the plain coefficients
    ctx.begin(prm)
    ctx.upload_pixels(img)
    ctx.transform()
    plain = O.encode(img, T.SCHEMES[scheme], qt)
    for c in range(3):
        assert np.array_equal(ctx.coefficients(c, lay), plain.coef[c]), (scheme, c)


def test_channel_gain_argument_checks(ctx):
    prm = B.make_params(16, 16, T.SCHEME_444, T.STANDARD_TABLES)
    ctx.begin(prm)
    with pytest.raises(B.JpegHipError):
        ctx.channel_gain(3, 1.0)
    for _ in range(4):
        ctx.channel_gain(0, 1.0)
    with pytest.raises(B.JpegHipError):
        ctx.channel_gain(0, 1.0)
