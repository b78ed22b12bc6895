"""The bit writer for long blocks (k_encode_dense, k_encode_dense_rest) on inputs the library would give to
its other writers: the diagnostic switch JPEGENC_WRITER=dense sends everything to it, so that photographic
content (tiles with runs of sixteen zeros and more: the list and the second kernel), restart intervals of
every length (tiles with boundaries, written from their tokens once their start is known), ragged sizes,
batches and bands that do not begin their image all pass through it.  Whole files against the CPU checker;
what the file must be is Encoder::serialize's output (Encoder.cpp:439-488, BitBuffer.cpp:128-204)."""
import numpy as np
import pytest
import torch

import oracle as O
from jpegenc_amd import binding as B, sharding, synth, tables as T

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["device_counter", "per_xcd"])
def ctx(monkeypatch, request):
    """Both ways the writer's workgroups come by their tile numbers: from one counter (what pictures of this size
    get), and by XCD in chunks (what large pictures get; the switch sends every picture there)."""
    monkeypatch.setenv("JPEGENC_WRITER", "dense")
    if request.param == "per_xcd":
        monkeypatch.setenv("JPEGENC_DENSE_PER_XCD_FROM", "0")
    c = B.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("kind,quality", [("natural", 90), ("noise", 100), ("noise", 75)])
@pytest.mark.parametrize("scheme", ["444", "420", "422"])
def test_whole_files(ctx, kind, quality, scheme):
    qt = T.quality_tables(quality)
    for (w, h) in ((333, 270), (640, 480), (1000, 37), (8, 8), (2049, 130)):
        img = synth.generate(kind, w, h, seed=w + h)
        lay = B.compute_layout(B.make_params(w, h, T.SCHEMES[scheme], qt))
        for ri in (0, 1, 7, lay.mcus_per_row):
            prm = B.make_params(w, h, T.SCHEMES[scheme], qt, restart_interval=ri)
            want = O.encode(img, T.SCHEMES[scheme], qt, restart_interval=ri).jpeg
            assert ctx.encode_host(img, prm) == want, (kind, quality, scheme, w, h, ri)


def test_one_bits_as_padding_and_a_second_image_in_the_same_context(ctx):
    qt = T.quality_tables(95)
    a = synth.generate("noise", 777, 333, seed=3)
    b = synth.generate("natural", 777, 333, seed=4)
    for img in (a, b, a):
        for pad_ones in (False, True):
            prm = B.make_params(777, 333, T.SCHEME_420, qt, restart_interval=5, pad_ones=pad_ones)
            assert ctx.encode_host(img, prm) == O.encode(img, T.SCHEME_420, qt, restart_interval=5, pad_ones=pad_ones).jpeg


def test_frames_of_a_batch(ctx):
    qt = T.quality_tables(100)
    frames = [synth.generate("noise" if k % 2 else "natural", 320, 240, seed=50 + k) for k in range(5)]
    prm = B.make_params(320, 240, T.SCHEME_444, qt)
    got = ctx.encode_batch_host(frames, prm, 1 << 20)
    for k, f in enumerate(frames):
        assert got[k] == O.encode(f, T.SCHEME_444, qt).jpeg, k


def test_bands_that_do_not_begin_the_image(ctx):
    """Three bands in one process: pieces two and three begin with the RSTn of their first interval."""
    w, h, world = 500, 300, 3
    img = synth.generate("noise", w, h, seed=9)
    scheme, qt = T.SCHEME_420, T.quality_tables(100)
    dev = torch.device("cuda", 0)
    encs = [sharding.HipBandEncoder(B.Context(0), B, w, h, scheme, qt, world, r) for r in range(world)]
    try:
        hists, pixels = [], []
        for e in encs:
            y0, y1 = e.pixel_rows
            d_rgb = torch.from_numpy(np.ascontiguousarray(img[y0:y1])).to(dev)
            d_hist = torch.zeros((4, 256), dtype=torch.int64, device=dev)
            torch.cuda.synchronize()
            e.analyze(d_rgb.data_ptr(), B.RGB8, w * 3, d_hist.data_ptr())
            torch.cuda.ExternalStream(e.ctx.stream).synchronize()
            hists.append(d_hist)
            pixels.append(d_rgb)
        total = sum(hists[1:], hists[0].clone())
        torch.cuda.synchronize()
        file = b""
        for e in encs:
            d_out = torch.zeros(1 << 21, dtype=torch.uint8, device=dev)
            torch.cuda.synchronize()
            n = e.finish(total.data_ptr(), d_out.data_ptr(), d_out.numel())
            file += d_out[:n].cpu().numpy().tobytes()
        lay = B.compute_layout(B.make_params(w, h, scheme, qt))
        assert file == O.encode(img, scheme, qt, restart_interval=lay.mcus_per_row).jpeg
    finally:
        for e in encs:
            e.ctx.close()
