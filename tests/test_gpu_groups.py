"""The bit writer on groups of eight coefficients (k_encode_groups) against the oracle.

The library takes it for images of a few hundred thousand blocks and more; here a diagnostic
switch makes it take small ones too, with every tile size, so that its paths (partial
tiles, interval ends inside tiles and chunks, stage overflow and the second run, batches,
bands) are compared byte for byte like the one-lane-per-block writer is elsewhere."""
import os

import numpy as np
import pytest

import oracle as O
from jpegenc_amd import binding as B, synth, tables as T

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    os.environ["JPEGENC_GROUP_MIN_BLOCKS"] = "0"
    os.environ["JPEGENC_SLOTS_MIN_BLOCKS"] = "0"      # tiles of 256 blocks leave their bits in slots (deferred placement)
    os.environ["JPEGENC_HIST_TILES_MIN_BLOCKS"] = "0"  # the histogram pass on groups as well (k_histogram_tiles)
    c = B.Context(0)
    yield c
    c.close()
    os.environ.pop("JPEGENC_GROUP_MIN_BLOCKS", None)
    os.environ.pop("JPEGENC_SLOTS_MIN_BLOCKS", None)
    os.environ.pop("JPEGENC_HIST_TILES_MIN_BLOCKS", None)
    os.environ.pop("JPEGENC_GROUP_TILE", None)


def _tile(n):
    if n is None:
        os.environ.pop("JPEGENC_GROUP_TILE", None)
    else:
        os.environ["JPEGENC_GROUP_TILE"] = str(n)


@pytest.mark.parametrize("tile", [None, 256, 64, 16])
@pytest.mark.parametrize("scheme", ["444", "420", "422", "411"])
def test_files_bit_exact(ctx, scheme, tile):
    _tile(tile)
    for (w, h, kind, tname) in [(640, 360, "natural", "annexk"), (333, 77, "noise", "annexk"), (250, 130, "noise", "ones"),
                                (1, 1, "natural", "ps12"), (64, 64, "natural", "ps5"), (9, 500, "noise", "annexk")]:
        img = synth.generate(kind, w, h, seed=w + h)
        qt = T.NAMED_TABLES[tname]
        enc = O.encode(img, T.SCHEMES[scheme], qt)
        prm = B.make_params(w, h, T.SCHEMES[scheme], qt)
        assert ctx.encode_host(img, prm) == enc.jpeg, (scheme, tile, w, h, kind, tname)


@pytest.mark.parametrize("scheme", ["444", "420", "422", "411"])
def test_histograms_counted_on_groups(ctx, scheme):
    # the staged call: k_histogram_tiles' counts against the oracle's, restart intervals of every kind,
    # sizes that end in partial tiles and MCUs, content with zero runs beyond sixteen and without
    for (w, h, kind, tname, ri) in [(640, 360, "natural", "annexk", 0), (333, 77, "noise", "annexk", 1), (250, 130, "noise", "ones", 7),
                                    (1, 1, "natural", "ps12", 0), (1030, 9, "natural", "ps5", 3), (9, 500, "noise", "annexk", 40)]:
        img = synth.generate(kind, w, h, seed=w * 3 + h)
        qt = T.NAMED_TABLES[tname]
        enc = O.encode(img, T.SCHEMES[scheme], qt, restart_interval=ri)
        prm = B.make_params(w, h, T.SCHEMES[scheme], qt, restart_interval=ri)
        ctx.begin(prm)
        ctx.upload_pixels(img)
        ctx.transform()
        assert np.array_equal(ctx.histograms(), enc.hist), (scheme, w, h, kind, tname, ri)


@pytest.mark.parametrize("scheme", ["444", "420", "422"])
def test_histograms_of_tiles_without_an_empty_group(ctx, scheme):
    """Dense content: k_histogram_tiles takes the tiles all of whose groups hold something block by block
    (histogram_full_tile).  Noise through tables of ones (hardly a zero), through fine tables (zeros here and there,
    counted one by one), tiles that end in the middle, intervals that restart the DC prediction inside tiles, and a
    picture whose lower part is photographic so that both kinds of tiles meet in one histogram."""
    for (w, h, tname, ri) in [(512, 512, "ones", 0), (517, 261, "ones", 5), (250, 130, "ones", 1), (640, 200, "ps12", 0),
                              (331, 203, "ps12", 9), (1040, 72, "ones", 0)]:
        img = synth.generate("noise", w, h, seed=w + 7 * h)
        if (w, h) == (512, 512):
            img[300:] = synth.generate("natural", w, h - 300, seed=5)
        qt = T.NAMED_TABLES[tname]
        enc = O.encode(img, T.SCHEMES[scheme], qt, restart_interval=ri)
        prm = B.make_params(w, h, T.SCHEMES[scheme], qt, restart_interval=ri)
        ctx.begin(prm)
        ctx.upload_pixels(img)
        ctx.transform()
        assert np.array_equal(ctx.histograms(), enc.hist), (scheme, w, h, tname, ri)
        assert ctx.encode_host(img, prm) == enc.jpeg, (scheme, w, h, tname, ri)
    # frames of a batch: every frame's tiles count into the frame's own histograms
    qt = T.NAMED_TABLES["ones"]
    frames = np.stack([synth.generate("noise" if k != 1 else "natural", 264, 200, seed=90 + k) for k in range(4)])
    prm = B.make_params(264, 200, T.SCHEMES[scheme], qt)
    got = ctx.encode_batch_host(frames, prm, 1 << 19)
    for k in range(4):
        assert got[k] == O.encode(frames[k], T.SCHEMES[scheme], qt).jpeg, (scheme, k)


@pytest.mark.parametrize("tile", [256, 16])
@pytest.mark.parametrize("ri", [3, 8, 40, 1000])
def test_restart_intervals(ctx, ri, tile):
    # intervals of 16 blocks and more go through the group writer (ends inside tiles and
    # chunks, several per tile); shorter ones fall back to the other writer
    _tile(tile)
    for scheme in ("444", "420"):
        for pad in (False, True):
            img = synth.generate("noise", 331, 203, seed=ri)
            qt = T.quality_tables(90)
            enc = O.encode(img, T.SCHEMES[scheme], qt, restart_interval=ri, pad_ones=pad)
            prm = B.make_params(331, 203, T.SCHEMES[scheme], qt, restart_interval=ri, pad_ones=pad)
            assert ctx.encode_host(img, prm) == enc.jpeg, (scheme, ri, pad, tile)


def test_stage_overflow_runs_again_with_smaller_tiles(ctx):
    # noise at q=100: some 400 bits per block; tiles of 256 blocks do not fit their stage,
    # the kernel says so and the stage runs again with 64 and then 16 blocks per tile
    _tile(256)
    img = synth.generate("noise", 512, 256, seed=3)
    qt = T.quality_tables(100)
    enc = O.encode(img, T.SCHEMES["444"], qt)
    prm = B.make_params(512, 256, T.SCHEMES["444"], qt)
    assert ctx.encode_host(img, prm) == enc.jpeg


def test_batch_and_standard_tables(ctx):
    import torch
    _tile(None)
    qt = T.quality_tables(85)
    w, h, n = 200, 120, 5
    frames = np.stack([synth.generate("natural" if k % 2 else "noise", w, h, seed=40 + k) for k in range(n)])
    for std in (False, True):
        prm = B.make_params(w, h, T.SCHEMES["420"], qt, standard_huffman=std)
        want = [ctx.encode_host(frames[k], prm) for k in range(n)]
        got = ctx.encode_batch_host(frames, prm, 1 << 17)
        assert got == want, std
    # the single encodes above ran through the group writer too; check one against the oracle
    prm = B.make_params(w, h, T.SCHEMES["420"], qt)
    assert ctx.encode_host(frames[1], prm) == O.encode(frames[1], T.SCHEMES["420"], qt).jpeg


def test_bands_concatenate_to_the_whole_file(ctx):
    import torch
    _tile(64)
    qt = T.quality_tables(90)
    w, h = 640, 480
    img = synth.generate("natural", w, h, seed=77)
    scheme = T.SCHEMES["420"]
    mcus_per_row = (w + 15) // 16
    whole = O.encode(img, scheme, qt, restart_interval=mcus_per_row).jpeg
    rows = (h + 15) // 16
    dev = torch.device("cuda", 0)
    pieces = []
    hists = []
    ctxs = [B.Context(0) for _ in range(3)]
    try:
        cuts = [0, 7, 19, rows]
        bufs = []
        for k, c in enumerate(ctxs):
            prm = B.make_params(w, h, scheme, qt, restart_interval=mcus_per_row, mcu_rows=(cuts[k], cuts[k + 1]))
            band = torch.from_numpy(np.ascontiguousarray(img[cuts[k] * 16:])).to(dev)
            hist = torch.zeros((4, 256), dtype=torch.int64, device=dev)
            c.analyze_band(band.data_ptr(), B.RGB8, w * 3, prm, hist.data_ptr())
            bufs.append((band, hist))
        torch.cuda.synchronize()
        total = sum(b[1] for b in bufs)
        for k, c in enumerate(ctxs):
            out = torch.zeros(1 << 20, dtype=torch.uint8, device=dev)
            n = c.finish_band(total.data_ptr(), out.data_ptr(), out.numel())
            pieces.append(out[:n].cpu().numpy().tobytes())
    finally:
        for c in ctxs:
            c.close()
    assert b"".join(pieces) == whole


def test_tiles_that_chain_with_a_look_back(ctx):
    """The other way to place a tile's bits (the tile waits for its start and writes them itself),
    which small images and the smaller tiles take."""
    _tile(256)
    os.environ["JPEGENC_SLOTS_MIN_BLOCKS"] = "1000000000"
    try:
        for (w, h, kind, ri, pad) in [(640, 360, "natural", 0, False), (331, 203, "noise", 8, True), (331, 203, "natural", 40, False)]:
            img = synth.generate(kind, w, h, seed=w + h)
            qt = T.quality_tables(90)
            enc = O.encode(img, T.SCHEMES["420"], qt, restart_interval=ri, pad_ones=pad)
            prm = B.make_params(w, h, T.SCHEMES["420"], qt, restart_interval=ri, pad_ones=pad)
            assert ctx.encode_host(img, prm) == enc.jpeg, (w, h, kind, ri, pad)
    finally:
        os.environ["JPEGENC_SLOTS_MIN_BLOCKS"] = "0"
