"""The benchmark-size workloads, byte for byte: the HIP encoder's file, coefficients and histograms
against length and SHA-256 of what the CPU oracle produced for the same inputs
(tests/golden/fullsize_hashes.json, written by tests/golden/make_fullsize_hashes.py after the scalar
and the AVX2 restatement had agreed on every one of them).  No oracle runs here: the inputs come
from the library's generator (the same bytes as the host generator, see
test_generator_matches_the_host_generators) and only digests are compared.  What the file must be
is Encoder::serialize's output (Encoder.cpp:439-488) for that input as the oracle states it."""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

from jpegenc_amd import binding as B, tables as T

pytestmark = pytest.mark.gpu

GOLDEN = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fullsize_hashes.json")))


def _tables(name):
    return T.NAMED_TABLES["annexk"] if name == "annexk" else T.quality_tables(int(name[1:]))


def _sha(t):
    return hashlib.sha256(t.cpu().numpy().tobytes()).hexdigest()


@pytest.fixture(scope="module")
def ctx():
    c = B.Context(0)
    yield c
    c.close()


SINGLE = [k for k in sorted(GOLDEN) if "_frame" not in k]


@pytest.mark.parametrize("name", SINGLE)
def test_file_coefficients_and_histograms_equal_the_oracles(ctx, name):
    g = GOLDEN[name]
    w, h = g["width"], g["height"]
    scheme, qt = T.SCHEMES[g["scheme"]], _tables(g["tables"])
    dev = torch.device("cuda", 0)
    rgb = torch.empty(h * w * 3, dtype=torch.uint8, device=dev)
    if g["kind"] == "mixed":
        from jpegenc_amd import synth
        ctx.synthetic_images("natural", w, 0, h, 1, rgb.data_ptr(), w * 3)
        y0, y1 = synth.mixed_band(h)
        ctx.synthetic_images("noise", w, y0, y1, 42, rgb.data_ptr() + y0 * w * 3, w * 3)
    else:
        ctx.synthetic_images(g["kind"], w, 0, h, g["seed"], rgb.data_ptr(), w * 3)
    prm = B.make_params(w, h, scheme, qt, restart_interval=g["restart_interval"])
    lay = B.compute_layout(prm)
    out = torch.empty(g["length"] + (1 << 20), dtype=torch.uint8, device=dev)
    ctx.set_profiling(True)
    n = ctx.encode(rgb.data_ptr(), B.RGB8, w * 3, prm, out.data_ptr(), out.numel())
    assert n == g["length"], name
    assert _sha(out[:n]) == g["sha256"], name
    # the form of the bit writer is chosen from the histogram pass's statistics, not found by running it again
    assert ctx.profile_sums()[3] == 1, name
    ctx.set_profiling(False)
    # the same again: nothing of the first encode may leak into the second (workspace reuse)
    out.fill_(0x5A)
    n = ctx.encode(rgb.data_ptr(), B.RGB8, w * 3, prm, out.data_ptr(), out.numel())
    assert n == g["length"] and _sha(out[:n]) == g["sha256"], name
    assert bool((out[n:n + 4096] == 0x5A).all()), "bytes written behind the file"
    # the stages on their own: coefficients (per channel, MCU order, zig-zag) and the four histograms
    ctx.begin(prm)
    ctx.bind_pixels(rgb.data_ptr(), B.RGB8, w * 3)
    ctx.transform()
    hs = hashlib.sha256()
    for c in range(3):
        hs.update(ctx.coefficients(c, lay).tobytes())
    assert hs.hexdigest() == g["coef_sha256"], name
    assert hashlib.sha256(ctx.histograms().tobytes()).hexdigest() == g["hist_sha256"], name


def test_frames_of_the_batch_workload(ctx):
    """256 frames in one call (bench.py's batch workload: frame k has seed 1000 + k)."""
    w, h, n = 1920, 1080, 256
    dev = torch.device("cuda", 0)
    prm = B.make_params(w, h, T.SCHEME_420, T.quality_tables(90))
    rgb = torch.empty(n * w * h * 3, dtype=torch.uint8, device=dev)
    ctx.synthetic_images("natural", w, 0, h, 1000, rgb.data_ptr(), w * 3, w * h * 3, n)
    slot = 1 << 20
    out = torch.empty(n * slot, dtype=torch.uint8, device=dev)
    lens = ctx.encode_batch(rgb.data_ptr(), B.RGB8, w * 3, w * h * 3, n, prm, out.data_ptr(), slot)
    for k in (0, 1, 255):
        g = GOLDEN["1080p_420_q90_frame%d" % k]
        assert int(lens[k]) == g["length"] and _sha(out[k * slot:k * slot + int(lens[k])]) == g["sha256"], k


def test_one_gpus_share_of_ten_thousand_frames(ctx):
    """Ten thousand 1080p frames over eight GPUs are 1250 per GPU and call: the last rank's share
    (frames 8750..9999), whose last file must be the oracle's file of frame 9999; every other file
    is compared with a single encode of a sample of the frames."""
    w, h, n, first = 1920, 1080, 1250, 8750
    dev = torch.device("cuda", 0)
    prm = B.make_params(w, h, T.SCHEME_420, T.quality_tables(90))
    rgb = torch.empty(n * w * h * 3, dtype=torch.uint8, device=dev)
    ctx.synthetic_images("natural", w, 0, h, 1000 + first, rgb.data_ptr(), w * 3, w * h * 3, n)
    slot = 512 << 10
    out = torch.empty(n * slot, dtype=torch.uint8, device=dev)
    lens = ctx.encode_batch(rgb.data_ptr(), B.RGB8, w * 3, w * h * 3, n, prm, out.data_ptr(), slot)
    g = GOLDEN["1080p_420_q90_frame9999"]
    assert int(lens[n - 1]) == g["length"]
    assert _sha(out[(n - 1) * slot:(n - 1) * slot + g["length"]]) == g["sha256"]
    assert lens.min() > 300000 and lens.max() < 360000
    one = torch.empty(slot, dtype=torch.uint8, device=dev)
    for k in (0, 1, 127, 128, 624, 625, 1000, 1248):
        m = ctx.encode(rgb.data_ptr() + k * w * h * 3, B.RGB8, w * 3, prm, one.data_ptr(), slot)
        assert m == int(lens[k]) and torch.equal(one[:m], out[k * slot:k * slot + m]), k
