"""C-ABI library on a machine without a GPU: symbols, host-side tables and headers."""
import ctypes
import os
import re

import numpy as np
import pytest

import oracle as O
from jpegenc_amd import binding as B, synth, tables as T

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "jpegenc_hip.h")).read()
    declared = set(re.findall(r"\b(jpeg_hip_[a-z0-9_]+)\s*\(", header))
    assert declared == set(B.EXPORTS)
    lib = ctypes.CDLL(B.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name


def test_status_strings():
    assert B.status_string(0) == "ok"
    assert "argument" in B.status_string(-1)


def _random_hist(rng, kind):
    h = np.zeros(256, np.uint64)
    if kind == "few":
        idx = rng.choice(256, rng.randint(1, 6), replace=False)
        h[idx] = rng.randint(1, 100, idx.size)
    elif kind == "ties":
        idx = rng.choice(256, rng.randint(17, 200), replace=False)
        h[idx] = rng.randint(1, 4, idx.size)       # many equal frequencies, more than 16 elements
    elif kind == "geometric":
        idx = rng.choice(256, rng.randint(20, 163), replace=False)
        h[idx] = (2.0 ** rng.uniform(0, 30, idx.size)).astype(np.uint64) + 1   # forces the 16-bit limit
    else:
        h[:] = rng.randint(0, 1000, 256)
    return h


@pytest.mark.parametrize("kind", ["few", "ties", "geometric", "dense"])
def test_huffman_tables_match_oracle(kind):
    rng = np.random.RandomState(hash(kind) % 1000)
    for _ in range(40):
        h = _random_hist(rng, kind)
        lo, co, dht = O.huffman(h, 1, 1)
        lp, cp = B.huffman_from_histogram(h)
        assert np.array_equal(lo, lp) and np.array_equal(co, cp)
        assert lp.max() <= 16
        assert B.write_dht(lp, cp, 1, 1) == dht
        # Kraft sum with the reserved all-ones code is exactly one.
        kraft = sum(2.0 ** -int(l) for l in lp if l) + 2.0 ** -int(lp.max())
        assert abs(kraft - 1.0) < 1e-12 or kind == "few"


@pytest.mark.parametrize("limit", [9, 10, 12, 16, 20, 33, 64])
def test_huffman_tables_of_other_code_word_lengths_match_oracle(limit):
    # HuffmanTable<byte>::create(codeWordLength, ...) for the limits the reference accepts (9..64).
    rng = np.random.RandomState(limit)
    for kind in ("few", "ties", "geometric", "dense"):
        for _ in range(8):
            h = _random_hist(rng, kind)
            lo, co = O.huffman_limit(h, limit)
            lp, cp = B.huffman_from_histogram_limit(h, limit)
            assert np.array_equal(lo, lp) and np.array_equal(co, cp), (limit, kind)
            assert lp.max() <= limit
            if limit == 16:
                l16, c16 = B.huffman_from_histogram(h)
                assert np.array_equal(l16, lp) and np.array_equal(c16.astype(np.uint64), cp)
            # a prefix code that leaves the all-ones word free: Kraft sum below one
            used = [int(l) for l in lp if l]
            assert sum(2.0 ** -l for l in used) <= 1.0 - 2.0 ** -max(used) + 1e-12
    for bad in (0, 8, 65):
        assert O.huffman_limit(np.ones(256), bad) is None
        with pytest.raises(B.JpegHipError):
            B.huffman_from_histogram_limit(np.ones(256), bad)


def test_huffman_reference_kat_through_abi():
    cnt = np.bincount([0, 1] + [2] * 3 + [3] * 5 + [4] * 6 + [5] * 11 + [6] * 13, minlength=256)
    lengths, codes = B.huffman_from_histogram(cnt)
    assert B.write_dht(lengths, codes, 3, 1).hex() == "ffc4001a130003010101010000000000000000000006050403020100"


@pytest.mark.parametrize("scheme", ["444", "420", "422", "411", "321"])
def test_headers_match_oracle(scheme):
    img = synth.generate("natural", 61, 43)
    for ri in (0, 7):
        enc = O.encode(img, T.SCHEMES[scheme], T.quality_tables(75), restart_interval=ri)
        prm = B.make_params(61, 43, T.SCHEMES[scheme], T.quality_tables(75), restart_interval=ri)
        assert B.write_headers(prm, enc.lengths, enc.codes) == enc.jpeg[:enc.scan_offset]


def test_layout_matches_oracle_dimensions():
    for scheme in sorted(T.SCHEMES):
        for (w, h) in [(16, 16), (17, 17), (37, 29), (250, 130)]:
            prm = B.make_params(w, h, T.SCHEMES[scheme], T.STANDARD_TABLES)
            lay = B.compute_layout(prm)
            enc = O.encode(synth.generate("noise", w, h), T.SCHEMES[scheme], T.STANDARD_TABLES)
            assert (lay.padded_width, lay.padded_height) == (enc.sim_w, enc.sim_h)
            assert (lay.padded_width, lay.padded_height) == T.padded_size(w, h, T.SCHEMES[scheme])
            for c in range(3):
                assert lay.blocks[c] == enc.coef[c].shape[0]
            assert [(lay.h_factor[c], lay.v_factor[c]) for c in range(3)] == T.inverse_factors(T.SCHEMES[scheme])


def test_invalid_parameters_are_rejected():
    bad = B.make_params(0, 10, T.SCHEME_444, T.STANDARD_TABLES)
    with pytest.raises(B.JpegHipError):
        B.compute_layout(bad)
    q = T.STANDARD_TABLES.copy()
    q[0, 0] = 0
    with pytest.raises(B.JpegHipError):
        B.compute_layout(B.make_params(8, 8, T.SCHEME_444, q))
    # a band without restart markers cannot be stitched
    with pytest.raises(B.JpegHipError):
        B.compute_layout(B.make_params(64, 64, T.SCHEME_444, T.STANDARD_TABLES, mcu_rows=(0, 4)))
    # a band must start on a restart-interval boundary
    with pytest.raises(B.JpegHipError):
        B.compute_layout(B.make_params(64, 64, T.SCHEME_444, T.STANDARD_TABLES, restart_interval=3, mcu_rows=(1, 8)))


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(B.JpegHipError):
        B.Context(0)


def test_quality_tables():
    assert np.all(T.quality_tables(100) == 1)
    q90 = T.quality_tables(90)
    assert q90[0, 0] == 3 and q90[1, 63] == 20
    assert np.array_equal(T.quality_tables(50), T.STANDARD_TABLES)
