"""Randomised HIP-vs-oracle comparisons: sizes, schemes, tables, restart intervals, content mixes."""
import os

import numpy as np
import pytest

import oracle as O
from jpegenc_amd import binding as B, synth, tables as T

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = B.Context(0)
    yield c
    c.close()


def _content(rng, w, h):
    kind = rng.randint(0, 5)
    if kind == 0:
        return synth.generate("natural", w, h, seed=int(rng.randint(1, 1000)))
    if kind == 1:
        return synth.generate("noise", w, h, seed=int(rng.randint(1, 1000)))
    if kind == 2:      # flat with a noisy patch: short and very long blocks side by side
        img = np.full((h, w, 3), int(rng.randint(0, 256)), np.uint8)
        y0, x0 = rng.randint(0, h), rng.randint(0, w)
        img[y0:, x0:] = rng.randint(0, 256, (h - y0, w - x0, 3)).astype(np.uint8)
        img[y0 + (h - y0) // 4:, :] = img[y0 + (h - y0) // 4:, :] // 64 * 64   # and a posterised part
        return img
    if kind == 3:      # extremes: saturated checkerboards
        yy, xx = np.indices((h, w))
        v = (((yy // int(rng.randint(1, 4))) + (xx // int(rng.randint(1, 4)))) & 1) * 255
        return np.stack([v, 255 - v, v], -1).astype(np.uint8)
    g = np.linspace(0, 255, w)[None, :] * np.ones((h, 1))       # smooth gradient
    return np.stack([g, g[:, ::-1], g * 0.5], -1).astype(np.uint8)


# JPEGENC_FUZZ_SEEDS / JPEGENC_FUZZ_MAX_SIDE widen the search for an occasional long run.
_SEEDS = int(os.environ.get("JPEGENC_FUZZ_SEEDS", "24"))
_MAX_SIDE = int(os.environ.get("JPEGENC_FUZZ_MAX_SIDE", "400"))


@pytest.mark.parametrize("seed", range(_SEEDS))
def test_random_configurations_byte_identical(ctx, seed):
    rng = np.random.RandomState(1000 + seed)
    names = sorted(T.SCHEMES)
    for _ in range(6):
        w, h = int(rng.randint(1, _MAX_SIDE)), int(rng.randint(1, max(2, _MAX_SIDE * 3 // 4)))
        scheme = T.SCHEMES[names[rng.randint(len(names))]]
        img = _content(rng, w, h)
        which = rng.randint(0, 4)
        if which == 0:
            qt = T.quality_tables(int(rng.randint(1, 101)))
        elif which == 1:
            qt = T.NAMED_TABLES["ones"]
        elif which == 2:
            qt = rng.randint(1, 256, (3, 64)).astype(np.float32)
        else:
            qt = T.STANDARD_TABLES
        mcus = ((w + T.step_size(scheme)[0] - 1) // T.step_size(scheme)[0]) * ((h + T.step_size(scheme)[1] - 1) // T.step_size(scheme)[1])
        ri = 0 if rng.randint(0, 2) else int(rng.randint(1, max(2, mcus + 2)))
        pad = bool(rng.randint(0, 2))
        want = O.encode(img, scheme, qt, restart_interval=ri, pad_ones=pad).jpeg
        prm = B.make_params(w, h, scheme, qt, restart_interval=ri, pad_ones=pad)
        got = ctx.encode_host(img, prm)
        assert got == want, (seed, w, h, ri, pad, which)


def test_divisors_that_are_not_integers(ctx):
    # QTable takes any float (QuantizationTables.h:6-27): the transform then has to divide for
    # real (the reciprocal form is exact for integers only), and DQT stores (byte)q
    # (JPEGSegments.h:152-167).  Byte and float pixels, every scheme family, ragged sizes.
    rng = np.random.RandomState(77)
    for k, (w, h, name) in enumerate([(97, 61, "420"), (64, 48, "444"), (333, 40, "422"), (50, 130, "411"), (1, 1, "420")]):
        scheme = T.SCHEMES[name]
        qt = (1.0 + rng.rand(3, 64) * 254.9).astype(np.float32)
        if k == 1:
            qt = (T.STANDARD_TABLES * np.float32(0.37) + np.float32(1.0)).astype(np.float32)
        img = _content(rng, w, h)
        want = O.encode(img, scheme, qt).jpeg
        assert ctx.encode_host(img, B.make_params(w, h, scheme, qt)) == want, (w, h, name)
        imgf = (img.astype(np.float32) / np.float32(255.0)).astype(np.float32)
        wantf = O.encode(imgf, scheme, qt).jpeg
        assert ctx.encode_host(imgf, B.make_params(w, h, scheme, qt)) == wantf, (w, h, name, "float")
    # out of range stays an error
    bad = T.STANDARD_TABLES.copy()
    bad[0, 5] = 256.0
    with pytest.raises(B.JpegHipError):
        ctx.begin(B.make_params(16, 16, T.SCHEME_444, bad))
    bad[0, 5] = 0.5
    with pytest.raises(B.JpegHipError):
        ctx.begin(B.make_params(16, 16, T.SCHEME_444, bad))


@pytest.mark.parametrize("split", ["0", "1000000000"])
@pytest.mark.parametrize("groups", [False, True])
def test_both_layouts_of_the_coefficient_buffer(ctx, monkeypatch, split, groups):
    # The library keeps a block's 128 bytes together, or groups 0..3 of all blocks in one plane and
    # groups 4..7 in another (large images).  Files, per-channel coefficients and the scan-ordered
    # device buffer have to be the same either way, with the one-lane-per-block kernels and with
    # the kernels on groups (switched on for these small images).
    import ctypes as C
    monkeypatch.setenv("JPEGENC_SPLIT_MIN_BLOCKS", split)
    if groups:
        monkeypatch.setenv("JPEGENC_GROUP_MIN_BLOCKS", "0")
        monkeypatch.setenv("JPEGENC_HIST_TILES_MIN_BLOCKS", "0")
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    rng = np.random.RandomState(5)
    for (w, h, name, ri, kind, q) in [(640, 480, "420", 0, "natural", 90), (333, 211, "444", 7, "natural", 75),
                                      (1000, 700, "422", 0, "noise", 100), (97, 61, "411", 3, "natural", 100),
                                      (1, 1, "420", 0, "natural", 90)]:
        scheme, qt = T.SCHEMES[name], T.quality_tables(q)
        img = synth.generate(kind, w, h, seed=int(rng.randint(1, 100)))
        want = O.encode(img, scheme, qt, restart_interval=ri)
        prm = B.make_params(w, h, scheme, qt, restart_interval=ri)
        assert ctx.encode_host(img, prm) == want.jpeg, (w, h, name)
        lay = B.compute_layout(prm)
        for c in range(3):
            assert np.array_equal(ctx.coefficients(c, lay), want.coef[c]), (w, h, name, c)
        # scan order: the channels' blocks interleaved per MCU
        total = int(lay.total_blocks)
        ptr = B.lib().jpeg_hip_coefficients_device(ctx._h)
        assert ptr
        got = np.zeros(total * 64, np.int16)
        assert hip.hipMemcpy(got.ctypes.data, ptr, got.nbytes, 2) == 0   # device to host
        got = got.reshape(total, 64)
        n_mcus = int(lay.mcus_per_row) * int(lay.mcu_rows)
        per = [int(lay.blocks[c]) // n_mcus for c in range(3)]
        assert sum(per) == int(lay.blocks_per_mcu)
        want_scan = np.concatenate([np.concatenate([want.coef[c][m * per[c]:(m + 1) * per[c]] for c in range(3)])
                                    for m in range(n_mcus)])
        assert np.array_equal(got, want_scan), (w, h, name, "device buffer")


def test_mostly_flat_image_with_long_blocks(ctx):
    # Mean block length far below 128 bits (the short-block bit writer is chosen), but a
    # band of noise at q=100 whose blocks overflow the private words and take the second walk.
    w, h = 1024, 768
    img = np.full((h, w, 3), 120, np.uint8)
    img[300:340, :, :] = synth.generate("noise", w, 40, seed=3)
    for scheme in ("420", "444"):
        for ri in (0, 64):
            qt = T.NAMED_TABLES["ones"]
            want = O.encode(img, T.SCHEMES[scheme], qt, restart_interval=ri).jpeg
            got = ctx.encode_host(img, B.make_params(w, h, T.SCHEMES[scheme], qt, restart_interval=ri))
            assert got == want, (scheme, ri)


def test_extreme_aspect_ratios(ctx):
    qt = T.quality_tables(90)
    for (w, h) in [(65535, 1), (1, 65535), (65535, 9), (17, 40000)]:
        img = synth.generate("natural", w, h, seed=2)
        for scheme in ("420", "444"):
            want = O.encode(img, T.SCHEMES[scheme], qt).jpeg
            got = ctx.encode_host(img, B.make_params(w, h, T.SCHEMES[scheme], qt))
            assert got == want, (w, h, scheme)


@pytest.mark.parametrize("seed", range(max(4, _SEEDS // 4)))
def test_random_configurations_float_pixels(ctx, seed):
    # Float pixels (Image::setRawPixelData) take a path of their own through the transform:
    # arbitrary values in 0..1, every scheme, aligned and unaligned row lengths.
    rng = np.random.RandomState(5000 + seed)
    names = sorted(T.SCHEMES)
    for _ in range(6):
        w, h = int(rng.randint(1, _MAX_SIDE)), int(rng.randint(1, max(2, _MAX_SIDE * 3 // 4)))
        if rng.randint(0, 2):
            w = (w + 3) // 4 * 4          # rows start on 16-byte boundaries
        scheme = T.SCHEMES[names[rng.randint(len(names))]]
        if rng.randint(0, 2):
            img = rng.rand(h, w, 3).astype(np.float32)
        else:
            img = (_content(rng, w, h).astype(np.float32) / np.float32(255.0)) * np.float32(rng.rand())
        qt = T.quality_tables(int(rng.randint(1, 101))) if rng.randint(0, 2) else rng.randint(1, 256, (3, 64)).astype(np.float32)
        want = O.encode(img, scheme, qt).jpeg
        got = ctx.encode_host(img, B.make_params(w, h, scheme, qt))
        assert got == want, (seed, w, h)


@pytest.mark.parametrize("seed", range(max(4, _SEEDS // 4)))
def test_random_device_buffers(ctx, seed):
    # Caller-owned device pixels: any row stride, any first address.
    import torch
    rng = np.random.RandomState(9000 + seed)
    names = sorted(T.SCHEMES)
    d_out = torch.zeros(8 << 20, dtype=torch.uint8, device="cuda")
    for _ in range(6):
        w, h = int(rng.randint(1, _MAX_SIDE)), int(rng.randint(1, max(2, _MAX_SIDE * 3 // 4)))
        scheme = T.SCHEMES[names[rng.randint(len(names))]]
        img = _content(rng, w, h)
        pad, off = int(rng.randint(0, 9)), int(rng.randint(0, 8))
        stride = w * 3 + pad
        buf = rng.randint(0, 256, off + stride * h + 16).astype(np.uint8)
        buf[off:off + stride * h].reshape(h, stride)[:, :w * 3] = img.reshape(h, w * 3)
        d_buf = torch.from_numpy(buf).cuda()
        torch.cuda.synchronize()
        qt = T.quality_tables(int(rng.randint(1, 101)))
        ri = 0 if rng.randint(0, 2) else int(rng.randint(1, 50))
        want = O.encode(img, scheme, qt, restart_interval=ri).jpeg
        prm = B.make_params(w, h, scheme, qt, restart_interval=ri)
        n = ctx.encode(d_buf.data_ptr() + off, B.RGB8, stride, prm, d_out.data_ptr(), d_out.numel())
        assert d_out[:n].cpu().numpy().tobytes() == want, (seed, w, h, pad, off)
