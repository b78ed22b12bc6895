"""HIP path vs CPU oracle through the C ABI (needs an MI355X)."""
import io
import hashlib

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


SIZES = [(16, 16), (17, 17), (37, 29), (48, 32), (64, 64), (250, 130), (8, 8), (1, 1), (9, 500), (500, 9)]


def _staged(ctx, img, scheme, qt, ri=0, pad=False):
    prm = B.make_params(img.shape[1], img.shape[0], scheme, qt, restart_interval=ri, pad_ones=pad)
    lay = B.compute_layout(prm)
    ctx.begin(prm)
    ctx.upload_pixels(img)
    ctx.transform()
    coef = [ctx.coefficients(c, lay) for c in range(3)]
    hist = ctx.histograms()
    return prm, lay, coef, hist


@pytest.mark.parametrize("scheme", sorted(T.SCHEMES))
@pytest.mark.parametrize("kind", ["natural", "noise"])
def test_small_images_bit_exact(ctx, scheme, kind):
    for (w, h) in SIZES:
        img = synth.generate(kind, w, h, seed=w * 131 + h)
        for tname in (["annexk", "ones"] if kind == "noise" else ["annexk", "ps12", "ps5"]):
            qt = T.NAMED_TABLES[tname]
            enc = O.encode(img, T.SCHEMES[scheme], qt)
            prm, lay, coef, hist = _staged(ctx, img, T.SCHEMES[scheme], qt)
            for c in range(3):
                assert np.array_equal(coef[c], enc.coef[c]), (scheme, kind, w, h, tname, c)
            assert np.array_equal(hist, enc.hist), (scheme, kind, w, h, tname)
            got = ctx.encode_host(img, prm)
            assert got == enc.jpeg, (scheme, kind, w, h, tname)


@pytest.mark.parametrize("ri", [1, 2, 3, 7, 64])
def test_restart_intervals(ctx, ri):
    for scheme in ("444", "420", "411"):
        for pad in (False, True):
            img = synth.generate("noise", 123, 77, seed=ri)
            qt = T.quality_tables(90)
            enc = O.encode(img, T.SCHEMES[scheme], qt, restart_interval=ri, pad_ones=pad)
            prm = B.make_params(123, 77, T.SCHEMES[scheme], qt, restart_interval=ri, pad_ones=pad)
            assert ctx.encode_host(img, prm) == enc.jpeg, (scheme, ri, pad)


def test_float_pixels(ctx):
    rng = np.random.RandomState(3)
    img = rng.rand(45, 70, 3).astype(np.float32)
    for scheme in ("444", "420"):
        enc = O.encode(img, T.SCHEMES[scheme], T.STANDARD_TABLES)
        prm = B.make_params(70, 45, T.SCHEMES[scheme], T.STANDARD_TABLES)
        assert ctx.encode_host(img, prm) == enc.jpeg


@pytest.mark.parametrize("w,h", [(4, 4), (16, 16), (64, 48), (332, 270), (1028, 36), (70, 45)])
def test_float_pixels_420_row_pair_form(ctx, w, h):
    # Rows of w*12 bytes are 16-byte aligned when w is a multiple of 4: those sizes take the
    # kernel that asks for whole row pairs ahead; 70 takes the general one.  Values outside
    # [0,1] reach the clamp of the level shift, ragged sizes the edge replication.
    rng = np.random.RandomState(w * 7 + h)
    img = (rng.rand(h, w, 3) * 1.4 - 0.2).astype(np.float32)
    for ri in (0, 5):
        enc = O.encode(img, T.SCHEMES["420"], T.STANDARD_TABLES, restart_interval=ri)
        prm = B.make_params(w, h, T.SCHEMES["420"], T.STANDARD_TABLES, restart_interval=ri)
        assert ctx.encode_host(img, prm) == enc.jpeg, (w, h, ri)


def test_float_planes_like_the_reference_image(ctx):
    # Three planes with the reference's layout: padded to whole MCUs ("simulated" size), the
    # border filled with anything -- only the image's own samples may be read.
    import torch
    rng = np.random.RandomState(11)
    d_out = torch.zeros(1 << 23, dtype=torch.uint8, device="cuda")
    for (w, h, scheme) in ((70, 45, "420"), (64, 48, "444"), (333, 17, "422"), (1930, 1085, "420")):
        img = rng.rand(h, w, 3).astype(np.float32)
        enc = O.encode(img, T.SCHEMES[scheme], T.quality_tables(75))
        sw, sh = T.padded_size(w, h, T.SCHEMES[scheme])
        planes = rng.rand(3, sh, sw).astype(np.float32)
        planes[:, :h, :w] = img.transpose(2, 0, 1)
        prm = B.make_params(w, h, T.SCHEMES[scheme], T.quality_tables(75))
        ctx.begin(prm)
        ctx.upload_planes(planes[0], planes[1], planes[2])
        ctx.transform()
        ctx.build_huffman(ctx.histograms())
        n = ctx.emit_scan(d_out.data_ptr(), d_out.numel())
        assert d_out[:n].cpu().numpy().tobytes() == enc.scan, (w, h, scheme)
    with pytest.raises(B.JpegHipError):
        ctx.begin(B.make_params(8, 8, T.SCHEME_444, T.STANDARD_TABLES, standard_huffman=True))
        ctx.upload_planes(planes[0], planes[1], planes[2])


def test_all_byte_values_and_saturated_blocks(ctx):
    # every byte value in every channel, plus flat black / white / checkerboard extremes
    ramp = np.arange(256, dtype=np.uint8)
    img = np.stack([np.tile(ramp, (64, 1)), np.tile(ramp[::-1], (64, 1)), np.tile((ramp * 7 + 3).astype(np.uint8), (64, 1))], -1)
    img[1::2] = img[1::2, ::-1]
    chk = (((np.indices((64, 256)).sum(0)) & 1) * 255).astype(np.uint8)
    for pic in (img, np.zeros_like(img), np.full_like(img, 255), np.stack([chk, chk, chk], -1)):
        for scheme in ("444", "420"):
            for tn in ("ones", "annexk"):
                enc = O.encode(pic, T.SCHEMES[scheme], T.NAMED_TABLES[tn])
                prm = B.make_params(256, 64, T.SCHEMES[scheme], T.NAMED_TABLES[tn])
                assert ctx.encode_host(pic, prm) == enc.jpeg


def test_row_stride_and_staged_emit(ctx):
    import torch
    img = synth.generate("natural", 100, 60)
    padded = np.zeros((60, 100 * 3 + 20), np.uint8)
    padded[:, :300] = img.reshape(60, 300)
    enc = O.encode(img, T.SCHEME_420, T.STANDARD_TABLES)
    prm = B.make_params(100, 60, T.SCHEME_420, T.STANDARD_TABLES)
    d_rgb = torch.from_numpy(padded).cuda()
    d_out = torch.zeros(1 << 20, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    n = ctx.encode(d_rgb.data_ptr(), B.RGB8, 320, prm, d_out.data_ptr(), d_out.numel())
    assert d_out[:n].cpu().numpy().tobytes() == enc.jpeg
    # staged: tables from the oracle's histograms, scan only
    ctx.begin(prm)
    ctx.bind_pixels(d_rgb.data_ptr(), B.RGB8, 320)
    ctx.transform()
    ctx.build_huffman(ctx.histograms())
    lengths, codes = ctx.get_huffman()
    assert np.array_equal(lengths, enc.lengths) and np.array_equal(codes, enc.codes)
    n = ctx.emit_scan(d_out.data_ptr(), d_out.numel())
    assert d_out[:n].cpu().numpy().tobytes() == enc.scan


def test_output_capacity_is_checked(ctx):
    import torch
    img = synth.generate("noise", 64, 64)
    prm = B.make_params(64, 64, T.SCHEME_444, T.NAMED_TABLES["ones"])
    d_rgb = torch.from_numpy(img).cuda()
    d_out = torch.zeros(4096, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    with pytest.raises(B.JpegHipError) as e:
        ctx.encode(d_rgb.data_ptr(), B.RGB8, 192, prm, d_out.data_ptr(), 2000)
    assert e.value.status == -3


def test_1080p_420_standard_tables(ctx):
    from PIL import Image
    img = synth.generate("natural", 1920, 1080)
    enc = O.encode(img, T.SCHEME_420, T.STANDARD_TABLES)
    prm = B.make_params(1920, 1080, T.SCHEME_420, T.STANDARD_TABLES)
    got = ctx.encode_host(img, prm)
    assert got == enc.jpeg
    dec = np.array(Image.open(io.BytesIO(got)).convert("RGB"))
    ref = np.array(Image.open(io.BytesIO(enc.jpeg)).convert("RGB"))
    assert np.array_equal(dec, ref)   # decoded PSNR is identical by construction


def test_4k_444_q90(ctx):
    img = synth.generate("natural", 3840, 2160)
    qt = T.quality_tables(90)
    enc = O.encode(img, T.SCHEME_444, qt)
    prm, lay, coef, hist = _staged(ctx, img, T.SCHEME_444, qt)
    mism = sum(int(np.count_nonzero(coef[c] != enc.coef[c])) for c in range(3))
    assert mism == 0
    assert ctx.encode_host(img, prm) == enc.jpeg


def test_noise_q100_worst_case(ctx):
    img = synth.generate("noise", 1024, 1024)
    qt = T.quality_tables(100)
    for scheme in ("444", "420"):
        enc = O.encode(img, T.SCHEMES[scheme], qt)
        prm = B.make_params(1024, 1024, T.SCHEMES[scheme], qt)
        got = ctx.encode_host(img, prm)
        assert hashlib.sha256(got).hexdigest() == hashlib.sha256(enc.jpeg).hexdigest()
        assert enc.lengths.max() == 16


def test_whole_image_band_without_summed_histograms(ctx):
    # finish_band(None): the band is the whole image and its histograms are used as
    # analyze_band counted them (the kernel leaves a copy with the host).  Interleaved with
    # the explicit form and with different sizes, so a stale host copy would show.
    import torch
    qt = T.quality_tables(90)
    d_hist = torch.zeros((4, 256), dtype=torch.int64, device="cuda")
    d_out = torch.zeros(4 << 20, dtype=torch.uint8, device="cuda")
    for k, (w, h, scheme) in enumerate([(333, 270, "420"), (64, 48, "444"), (800, 601, "420"), (37, 29, "422")]):
        img = synth.generate("natural" if k % 2 == 0 else "noise", w, h, seed=30 + k)
        d_rgb = torch.from_numpy(img).cuda()
        prm = B.make_params(w, h, T.SCHEMES[scheme], qt)
        want = O.encode(img, T.SCHEMES[scheme], qt).jpeg
        ctx.analyze_band(d_rgb.data_ptr(), B.RGB8, w * 3, prm, d_hist.data_ptr())
        n = ctx.finish_band(None, d_out.data_ptr(), d_out.numel())
        assert d_out[:n].cpu().numpy().tobytes() == want, (w, h, scheme, "own")
        n = ctx.finish_band(d_hist.data_ptr(), d_out.data_ptr(), d_out.numel())
        assert d_out[:n].cpu().numpy().tobytes() == want, (w, h, scheme, "explicit")
    # after histograms counted through the staged call there is no host copy to fall back on
    ctx.begin(prm)
    ctx.upload_pixels(img)
    ctx.transform()
    ctx.histograms()
    with pytest.raises(B.JpegHipError):
        ctx.finish_band(None, d_out.data_ptr(), d_out.numel())


def test_staged_table_sequences(ctx):
    # Tables reach the device lazily (the first kernel of the entropy stage fetches them from
    # pinned memory): repeated emits, tables replaced between emits, tables of one image
    # reused for the next, and a whole-file encode in between must all see the right ones.
    import torch
    qt = T.quality_tables(90)
    a = synth.generate("natural", 120, 88, seed=1)
    b = synth.generate("noise", 120, 88, seed=2)
    prm = B.make_params(120, 88, T.SCHEME_420, qt)
    d_out = torch.zeros(1 << 20, dtype=torch.uint8, device="cuda")
    ea, eb = O.encode(a, T.SCHEME_420, qt), O.encode(b, T.SCHEME_420, qt)

    def emit():
        n = ctx.emit_scan(d_out.data_ptr(), d_out.numel())
        return d_out[:n].cpu().numpy().tobytes()

    ctx.begin(prm)
    ctx.upload_pixels(a)
    ctx.transform()
    ctx.build_huffman(ctx.histograms())
    assert emit() == ea.scan
    assert emit() == ea.scan                       # second emit: the device copy of the tables
    # image a under tables that are not its own (the Annex K ones have a code for every symbol),
    # then back under its own
    std_lengths, std_codes = B.standard_huffman()
    ctx.set_huffman(std_lengths, std_codes)
    O.set_fixed_tables(std_lengths, std_codes)
    try:
        a_under_std = O.encode(a, T.SCHEME_420, qt).scan
    finally:
        O.set_fixed_tables()
    assert emit() == a_under_std
    # image b's tables lack symbols that image a has: an error, not a stream with holes
    assert ((ea.hist > 0) & (eb.lengths == 0)).any()
    ctx.set_huffman(eb.lengths, eb.codes)
    with pytest.raises(B.JpegHipError) as err:
        emit()
    assert err.value.status == B.ERR_INVALID_ARGUMENT
    ctx.set_huffman(ea.lengths, ea.codes)
    assert emit() == ea.scan
    # a whole-file encode of b installs b's tables and header; a staged emit afterwards uses them
    assert ctx.encode_host(b, prm) == eb.jpeg
    assert emit() == eb.scan
    assert ctx.emit_scan_host() == eb.scan


def test_finish_band_in_two_halves_and_images_in_flight(ctx):
    """jpeg_hip_finish_band_start / _wait: the same bytes as the one-call form, with three
    images in flight on three contexts, and the calls that must wait their turn refused."""
    import torch
    qt = T.quality_tables(85)
    imgs = [synth.generate("natural", 640, 360, seed=s) for s in (5, 6, 7, 8, 9)]
    prm = B.make_params(640, 360, T.SCHEME_420, qt)
    want = [ctx.encode_host(im, prm) for im in imgs]
    lanes = [B.Context(0) for _ in range(3)]
    d_imgs = [torch.from_numpy(im).cuda() for im in imgs]
    outs = [torch.zeros(1 << 20, dtype=torch.uint8, device="cuda") for _ in lanes]
    torch.cuda.synchronize()
    got = [None] * len(imgs)
    for i in range(len(imgs) + 2):
        if i < len(imgs):
            lanes[i % 3].analyze_band(d_imgs[i].data_ptr(), B.RGB8, 640 * 3, prm, None)
        if 1 <= i <= len(imgs):
            lanes[(i - 1) % 3].finish_band_start(None, outs[(i - 1) % 3].data_ptr(), outs[0].numel())
        if i >= 2:
            k = (i - 2) % 3
            n = lanes[k].finish_band_wait()
            got[i - 2] = outs[k][:n].cpu().numpy().tobytes()
    assert got == want
    c = lanes[0]
    c.analyze_band(d_imgs[0].data_ptr(), B.RGB8, 640 * 3, prm, None)
    c.finish_band_start(None, outs[0].data_ptr(), outs[0].numel())
    for refused in (lambda: c.begin(prm), lambda: c.finish_band_start(None, outs[0].data_ptr(), outs[0].numel()),
                    lambda: c.encode_host(imgs[0], prm), lambda: c.transform()):
        with pytest.raises(B.JpegHipError):
            refused()
    n = c.finish_band_wait()
    assert outs[0][:n].cpu().numpy().tobytes() == want[0]
    with pytest.raises(B.JpegHipError):
        c.finish_band_wait()
    for c in lanes:
        c.close()


def test_device_rows_off_dword_boundaries(ctx):
    """Caller-owned byte pixels with an odd row stride and an odd first address."""
    import torch
    rng = np.random.RandomState(21)
    d_out = torch.zeros(1 << 22, dtype=torch.uint8, device="cuda")
    for (w, h, scheme, pad, off) in ((1001, 333, "420", 0, 0), (1366, 200, "444", 1, 3), (77, 50, "422", 2, 1), (640, 91, "420", 3, 2)):
        img = synth.generate("natural", w, h, seed=w)
        stride = w * 3 + pad
        buf = rng.randint(0, 256, off + stride * h + 8).astype(np.uint8)
        view = buf[off:off + stride * h].reshape(h, stride)
        view[:, :w * 3] = img.reshape(h, w * 3)
        d_buf = torch.from_numpy(buf).cuda()
        torch.cuda.synchronize()
        qt = T.quality_tables(80)
        want = O.encode(img, T.SCHEMES[scheme], qt).jpeg
        prm = B.make_params(w, h, T.SCHEMES[scheme], qt)
        n = ctx.encode(d_buf.data_ptr() + off, B.RGB8, stride, prm, d_out.data_ptr(), d_out.numel())
        assert d_out[:n].cpu().numpy().tobytes() == want, (w, h, scheme, pad, off)


@pytest.mark.parametrize("writer", ["blocks", "groups"])
def test_tables_from_a_histogram_that_undercounts(ctx, writer, monkeypatch):
    """The histograms the tables are built from also tell how much of the stream buffer has
    to be cleared.  Tables from counts that do not cover the image (a flat image's tables kept
    for a busy one, a caller's own counts) must not leave a stream with stale bytes in it:
    the stage notices that it wrote past the cleared part and runs again."""
    import torch
    if writer == "groups":
        monkeypatch.setenv("JPEGENC_GROUP_MIN_BLOCKS", "0")
    qt = T.quality_tables(90)
    busy = synth.generate("noise", 640, 480, seed=5)
    prm = B.make_params(640, 480, T.SCHEME_420, qt)
    d_out = torch.full((4 << 20,), 0xFF, dtype=torch.uint8, device="cuda")
    c = B.Context(0)
    try:
        # leave stale bytes in the context's stream buffer: a long stream first
        assert c.encode_host(busy, prm) == O.encode(busy, T.SCHEME_420, qt).jpeg
        c.begin(prm)
        c.upload_pixels(busy)
        c.transform()
        hist = c.histograms()
        c.build_huffman(np.minimum(hist, 1))        # every symbol once: the same symbols, a tiny bit count
        lengths, codes = c.get_huffman()
        O.set_fixed_tables(lengths, codes)
        try:
            want = O.encode(busy, T.SCHEME_420, qt).scan
        finally:
            O.set_fixed_tables()
        n = c.emit_scan(d_out.data_ptr(), d_out.numel())
        assert d_out[:n].cpu().numpy().tobytes() == want
        # new pixels under the tables of the image before: the bound of that image is forgotten
        flat = np.full((480, 640, 3), 128, np.uint8)
        c.begin(prm)
        c.upload_pixels(flat)
        c.transform()
        c.build_huffman(c.histograms())
        c.emit_scan(d_out.data_ptr(), d_out.numel())
        std_lengths, std_codes = B.standard_huffman()
        c.set_huffman(std_lengths, std_codes)
        c.upload_pixels(busy)
        c.transform()
        O.set_fixed_tables(std_lengths, std_codes)
        try:
            want = O.encode(busy, T.SCHEME_420, qt).scan
        finally:
            O.set_fixed_tables()
        n = c.emit_scan(d_out.data_ptr(), d_out.numel())
        assert d_out[:n].cpu().numpy().tobytes() == want
    finally:
        c.close()


def test_stuffing_by_waves_on_small_streams(monkeypatch):
    """k_stuff_waves, which long streams get (a wave per 4 KiB chunk, group sums in front), on pictures of every
    size: streams that end in the middle of a row of a chunk, markers of restart intervals inside chunks, runs of
    0xFF bytes (saturated blocks), and a stream of a single chunk.  BitBuffer.cpp:128-204 is what the bytes must be."""
    monkeypatch.setenv("JPEGENC_STUFF_FORM", "1")
    monkeypatch.setenv("JPEGENC_GROUP_SUMS_FROM", "1")
    ctx = B.Context(0)
    try:
        for (w, h, kind, scheme, q, ri) in [(640, 480, "noise", "444", 100, 0), (333, 270, "noise", "420", 100, 7), (1000, 700, "natural", "422", 90, 63),
                                            (64, 40, "noise", "444", 100, 1), (1920, 1080, "noise", "420", 95, 120), (2049, 130, "natural", "444", 75, 0)]:
            img = synth.generate(kind, w, h, seed=w + h)
            qt = T.quality_tables(q)
            want = O.encode(img, T.SCHEMES[scheme], qt, restart_interval=ri).jpeg
            prm = B.make_params(w, h, T.SCHEMES[scheme], qt, restart_interval=ri)
            assert ctx.encode_host(img, prm) == want, (w, h, kind, scheme, q, ri)
        # every byte 0xFF for long stretches: white noise through tables of ones has runs of them, a flat picture none
        img = np.full((256, 256, 3), 255, np.uint8)
        img[::2, ::3] = 0
        qt = T.NAMED_TABLES["ones"]
        assert ctx.encode_host(img, B.make_params(256, 256, T.SCHEMES["444"], qt)) == O.encode(img, T.SCHEMES["444"], qt).jpeg
    finally:
        ctx.close()
