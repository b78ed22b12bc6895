"""Standard (ITU-T T.81 Annex K.3) Huffman tables: the table constants on the CPU, the
histogram-free encode paths on the GPU."""
import io
import json
import os

import numpy as np
import pytest

import oracle as O
from jpegenc_amd import binding as B, synth, tables as T

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "annexk_huffman.json")


def _canonical(bits, values):
    """Annex C code assignment, written independently of the library."""
    lengths = np.zeros(256, np.uint8)
    codes = np.zeros(256, np.uint16)
    code, k = 0, 0
    for ln in range(1, 17):
        for _ in range(bits[ln - 1]):
            lengths[values[k]] = ln
            codes[values[k]] = code
            code += 1
            k += 1
        code <<= 1
    return lengths, codes


def _golden_tables():
    g = json.load(open(GOLDEN))
    ln = np.zeros((4, 256), np.uint8)
    cd = np.zeros((4, 256), np.uint16)
    for t, tab in enumerate(g["tables"]):
        ln[t], cd[t] = _canonical(tab["bits"], tab["values"])
    return ln, cd


def test_tables_match_libjpeg_fixture():
    ln, cd = B.standard_huffman()          # pure host function of the library
    gl, gc = _golden_tables()
    assert np.array_equal(ln, gl)
    assert np.array_equal(cd, gc)
    # prefix-free and complete enough: Kraft sum below 1, no all-ones code of maximal length
    for t in range(4):
        used = ln[t] > 0
        assert (2.0 ** -ln[t][used].astype(float)).sum() < 1.0
        assert not any(int(cd[t][s]) == (1 << int(ln[t][s])) - 1 for s in np.nonzero(used)[0])


def test_dht_segment_of_standard_table_round_trips():
    ln, cd = B.standard_huffman()
    g = json.load(open(GOLDEN))
    for t, (tid, kind) in enumerate([(0, 0), (1, 0), (0, 1), (1, 1)]):
        seg = B.write_dht(ln[t], cd[t], tid, kind)
        assert seg[:2] == b"\xff\xc4"
        assert seg[4] == (kind << 4 | tid)
        assert list(seg[5:21]) == g["tables"][t]["bits"]
        assert list(seg[21:]) == g["tables"][t]["values"]


@pytest.fixture(scope="module")
def ctx():
    c = B.Context(0)
    yield c
    c.close()


@pytest.fixture()
def fixed_oracle():
    ln, cd = _golden_tables()
    O.set_fixed_tables(ln, cd)
    yield
    O.set_fixed_tables()


@pytest.mark.gpu
@pytest.mark.parametrize("scheme", ["444", "420", "422", "411"])
def test_standard_tables_byte_identical(ctx, fixed_oracle, scheme):
    for (w, h) in [(16, 16), (37, 29), (250, 130), (1, 1), (500, 9)]:
        for kind, tn in [("natural", "annexk"), ("noise", "ones"), ("noise", "ps5")]:
            img = synth.generate(kind, w, h, seed=w + h)
            qt = T.NAMED_TABLES[tn]
            for ri in (0, 5):
                want = O.encode(img, T.SCHEMES[scheme], qt, restart_interval=ri).jpeg
                prm = B.make_params(w, h, T.SCHEMES[scheme], qt, restart_interval=ri, standard_huffman=True)
                assert ctx.encode_host(img, prm) == want, (scheme, w, h, kind, tn, ri)


@pytest.mark.gpu
def test_standard_tables_decode_like_optimised(ctx):
    from PIL import Image
    img = synth.generate("natural", 320, 200)
    qt = T.quality_tables(90)
    a = ctx.encode_host(img, B.make_params(320, 200, T.SCHEME_420, qt))
    b = ctx.encode_host(img, B.make_params(320, 200, T.SCHEME_420, qt, standard_huffman=True))
    assert len(b) >= len(a)            # optimised tables never lose
    pa = np.asarray(Image.open(io.BytesIO(a)).convert("RGB"))
    pb = np.asarray(Image.open(io.BytesIO(b)).convert("RGB"))
    assert np.array_equal(pa, pb)      # same coefficients, different entropy code


@pytest.mark.gpu
def test_standard_tables_device_paths(ctx, fixed_oracle):
    import torch
    w, h, n = 200, 120, 5
    qt = T.quality_tables(90)
    imgs = [synth.generate("natural", w, h, seed=k) for k in range(n)]
    want = [O.encode(im, T.SCHEME_420, qt).jpeg for im in imgs]
    prm = B.make_params(w, h, T.SCHEME_420, qt, standard_huffman=True)
    # one-shot device encode with a tight capacity
    d_rgb = torch.from_numpy(np.stack(imgs)).cuda()
    out = torch.zeros(1 << 18, dtype=torch.uint8, device="cuda")
    nbytes = ctx.encode(d_rgb[0].data_ptr(), B.RGB8, w * 3, prm, out.data_ptr(), out.numel())
    assert out[:nbytes].cpu().numpy().tobytes() == want[0]
    # batch: one shared table set, no host round trip
    slot = 1 << 16
    outs = torch.full((n * slot,), 0x55, dtype=torch.uint8, device="cuda")
    lens = ctx.encode_batch(d_rgb.data_ptr(), B.RGB8, w * 3, w * h * 3, n, prm, outs.data_ptr(), slot)
    host = outs.cpu().numpy()
    for k in range(n):
        assert host[k * slot:k * slot + int(lens[k])].tobytes() == want[k], k
    # float pixels are refused with these tables
    with pytest.raises(B.JpegHipError) as e:
        ctx.encode_host(np.zeros((8, 8, 3), np.float32), B.make_params(8, 8, T.SCHEME_444, qt, standard_huffman=True))
    assert e.value.status == B.ERR_INVALID_ARGUMENT
