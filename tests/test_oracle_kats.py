"""The CPU oracle against the reference's own known-answer vectors (CPU only)."""
import io
import json
import os

import numpy as np
import pytest

import oracle as O
from jpegenc_amd import synth, tables as T

KATS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_kats.json")))


def f32(hexstr):
    return np.float32(float.fromhex(hexstr))


def test_aan_constants_match_reference_hex():
    c = O.constants()
    k = KATS["aan_constants_hex"]
    want = [f32(k["a1"]), f32(k["a2"]), f32(k["a4"]), f32(k["a5"])] + [f32(s) for s in k["s"]]
    assert [float(x).hex() for x in c] == [float(x).hex() for x in want]


def test_colour_kats_bit_exact():
    for kat in KATS["colour"]:
        if "rgb_hex" in kat:
            rgb = [f32(v) for v in kat["rgb_hex"]]
        elif "rgb_f32" in kat:
            rgb = [np.float32(v) for v in kat["rgb_f32"]]
        else:
            rgb = [np.float32(v) / np.float32(255.0) for v in kat["rgb_u8"]]
        got = O.color_convert(*rgb)
        assert [float(x).hex() for x in got] == [float(f32(v)).hex() for v in kat["ycbcr_hex"]]


def test_dct_kat_two_decimals():
    blk = np.array(KATS["dct_block_input"], np.float32)
    got = O.dct_block(blk)
    want = np.array(KATS["dct_block_output_2dec_rows01"])
    assert np.array_equal(np.round(got[:2].astype(np.float64), 2), want)


def test_dct_matches_direct_dct():
    # The reference's testDCT compares its AAN path with the direct O(n^4) transform
    # (JPEG-Encoder-SIMD.cpp:305-379); same check in float64, tolerance for fp32 rounding.
    rng = np.random.RandomState(0)
    n = np.arange(8)
    basis = np.cos((2 * n[None, :] + 1) * n[:, None] * np.pi / 16)
    basis[0] *= np.sqrt(0.5)
    for _ in range(20):
        blk = rng.randint(0, 256, (8, 8)).astype(np.float32)
        want = 0.25 * basis @ blk.astype(np.float64) @ basis.T
        want[0, 0] -= 1024
        assert np.abs(O.dct_block(blk) - want).max() < 2e-3


def test_dct_quant_kats_exact():
    blk = np.array(KATS["dct_block_input"], np.float32) / np.float32(255.0)
    assert np.array_equal(O.dct_quant_block(blk, T.LUMINANCE).astype(int), np.array(KATS["dct_quant_luminance"]))
    assert np.array_equal(O.dct_quant_block(blk, T.CHROMINANCE).astype(int), np.array(KATS["dct_quant_chrominance"]))


def _codes(lengths, codes):
    return {str(s): format(int(codes[s]), "0%db" % lengths[s]) for s in range(256) if lengths[s]}


def test_huffman_kat():
    k = KATS["huffman"]
    cnt = np.bincount(k["symbols"], minlength=256)
    lengths, codes, dht = O.huffman(cnt, 3, 1)
    assert _codes(lengths, codes) == k["codes"]
    assert dht.hex() == k["dht_id3_ac_hex"]
    enc = "".join(_codes(lengths, codes)[str(s)] for s in (3, 0, 4))
    assert enc == k["encode_3_0_4"]


def test_huffman_tie_kat():
    k = KATS["huffman_tie"]
    cnt = np.zeros(256, np.int64)
    cnt[k["symbols_each_3_times"]] = 3
    lengths, codes, _ = O.huffman(cnt)
    assert _codes(lengths, codes) == k["codes"]
    assert "111" not in _codes(lengths, codes).values()   # all-ones code is never assigned


def test_bitbuffer_stuffing_matches_simple_writer():
    # BitBuffer::pushBitsEscaped semantics: a 0x00 follows every completed 0xFF byte,
    # the tail is padded with zero bits.
    rng = np.random.RandomState(1)
    for trial in range(50):
        pushes = []
        for _ in range(rng.randint(1, 200)):
            nb = int(rng.randint(0, 17))
            v = int(rng.randint(0, 1 << 16)) if trial % 2 else (1 << 16) - 1
            pushes.append((nb, v & ((1 << nb) - 1) if nb else 0))
        got, bitlen = O.bitbuffer_replay(pushes)
        bits = "".join(format(v, "0%db" % nb) for nb, v in pushes if nb)
        bits += "0" * (-len(bits) % 8)
        raw = bytes(int(bits[i:i + 8], 2) for i in range(0, len(bits), 8))
        want = raw.replace(b"\xff", b"\xff\x00")
        assert got == want


@pytest.mark.parametrize("scheme", sorted(T.SCHEMES))
def test_oracle_files_decode(scheme):
    # Independent decoder (Pillow) accepts the stream and reproduces the picture.
    from PIL import Image
    img = synth.generate("natural", 250, 130)
    for ri in (0, 5):
        enc = O.encode(img, T.SCHEMES[scheme], T.quality_tables(90), restart_interval=ri)
        dec = np.array(Image.open(io.BytesIO(enc.jpeg)).convert("RGB"))
        assert dec.shape == img.shape
        mse = np.mean((dec.astype(np.float64) - img) ** 2)
        assert 10 * np.log10(255 ** 2 / mse) > 33.0


def test_oracle_header_bytes():
    img = synth.generate("natural", 33, 17)
    enc = O.encode(img, T.SCHEME_420, T.STANDARD_TABLES)
    j = enc.jpeg
    assert j[:2] == b"\xff\xd8" and j[-2:] == b"\xff\xd9"
    assert j[2:20].hex() == KATS["headers"]["app0_hex"]
    assert j[20:30] == bytes([0xff, 0xc0, 0, 0x11, 8, 0, 17, 0, 33, 3])
    assert [j[31], j[34], j[37]] == KATS["headers"]["sof0_420_factors"]
    assert j[enc.scan_offset - 14:enc.scan_offset].hex() == KATS["headers"]["sos_hex"]


def test_padding_is_edge_replication():
    # An image and the same image explicitly padded by edge replication give the same
    # coefficients (Image::setRawPixelData clamps to the last row/column).
    img = synth.generate("noise", 37, 29)
    pad = np.pad(img, ((0, 3), (0, 11), (0, 0)), mode="edge")
    a = O.encode(img, T.SCHEME_420, T.STANDARD_TABLES)
    b = O.encode(pad, T.SCHEME_420, T.STANDARD_TABLES)
    for c in range(3):
        assert np.array_equal(a.coef[c], b.coef[c])


def test_division_replacements_are_exact(tmp_path):
    # byte/255.0f and value/q (integer q <= 255) as two-term reciprocal products: the fast
    # kernels rely on these being bit-identical to IEEE division.
    import subprocess
    src = os.path.join(os.path.dirname(__file__), "..", "oracle", "tools", "verify_exact_division.c")
    exe = str(tmp_path / "verify")
    subprocess.check_call(["gcc", "-O2", "-mfma", "-ffp-contract=off", "-fopenmp", src, "-o", exe, "-lm"])
    out = subprocess.check_output([exe, "1"], text=True)
    assert "mismatches=0" in out
    assert "hi=0x1.010102p-8 lo=-0x1.fdfdfep-33" in out


def test_oracle_output_hashes_unchanged():
    """The oracle's whole-file output is locked by hashes (tests/golden/make_oracle_hashes.py made them)."""
    import importlib.util
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("make_oracle_hashes", os.path.join(here, "golden", "make_oracle_hashes.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    want = json.load(open(os.path.join(here, "golden", "oracle_hashes.json")))
    got = {"%dx%d_%s_%s_%s_ri%d" % c: mod.digest(*c) for c in mod.cases()}
    assert got == want
    # the AVX2 + OpenMP restatement (bench.py's CPU baseline) writes the same files
    got = {"%dx%d_%s_%s_%s_ri%d" % c: mod.digest(*c, variant="avx2") for c in mod.cases()}
    assert got == want


def test_fullsize_fixture_is_what_the_oracle_writes():
    """tests/golden/fullsize_hashes.json (what the GPU tests compare benchmark-size files with) holds
    an entry for every case of the script that made it, and its 1080p and 4K entries are what both
    restatements write today (the larger ones take minutes and are checked when the fixture is made)."""
    import importlib.util
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("make_fullsize_hashes", os.path.join(here, "golden", "make_fullsize_hashes.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    want = json.load(open(os.path.join(here, "golden", "fullsize_hashes.json")))
    assert sorted(want) == sorted(mod.CASES)
    for name in ("1080p_420_annexk_natural", "1080p_420_q90_natural", "4k_444_q90_natural", "1080p_420_q90_frame9999"):
        assert mod.entry(name) == want[name], name


@pytest.mark.parametrize("scheme", sorted(T.SCHEMES))
def test_vector_restatement_equals_scalar_one(scheme):
    """Larger and ragged images, every scheme (those with factors other than 1 and 2 take the
    scalar stages inside the vector library), restart intervals, one-bit padding, float pixels."""
    for (w, h, kind, ri, pad) in [(1031, 517, "natural", 0, False), (640, 368, "noise", 7, True), (9, 5, "natural", 1, False)]:
        img = synth.generate(kind, w, h, seed=w + h)
        for pix in (img, (img.astype(np.float32) / np.float32(255.0))):
            a = O.encode(pix, T.SCHEMES[scheme], T.quality_tables(90), restart_interval=ri, pad_ones=pad)
            b = O.encode(pix, T.SCHEMES[scheme], T.quality_tables(90), restart_interval=ri, pad_ones=pad, variant="avx2")
            assert a.jpeg == b.jpeg
            assert all(np.array_equal(x, y) for x, y in zip(a.coef, b.coef))
            assert np.array_equal(a.hist, b.hist)
            assert (a.scan_offset, a.scan_len, a.channel_sizes) == (b.scan_offset, b.scan_len, b.channel_sizes)


def test_vector_restatement_with_fixed_tables_and_band_arguments():
    img = synth.generate("natural", 320, 200, seed=3)
    whole = O.encode(img, T.SCHEME_420, T.quality_tables(90), restart_interval=20)
    for variant in ("scalar", "avx2"):
        band = O.encode(img[96:], T.SCHEME_420, T.quality_tables(90), restart_interval=20, hist_override=whole.hist, rst_base=6,
                        header_height=200, variant=variant)
        assert whole.scan.endswith(band.scan)
    from jpegenc_amd import binding as B
    lengths, codes = B.standard_huffman()
    O.set_fixed_tables(lengths, codes)
    try:
        a = O.encode(img, T.SCHEME_420, T.quality_tables(90))
        b = O.encode(img, T.SCHEME_420, T.quality_tables(90), variant="avx2")
        assert a.jpeg == b.jpeg
    finally:
        O.set_fixed_tables()


# ---- vectors assembled from literals in the reference's sources (tests/golden/make_literal_kats.py) ----

LIT = KATS["literals"]


def _segments(j):
    """[(marker byte, segment bytes incl. marker and length)] up to and including SOS."""
    out, pos = [(j[1], j[:2])], 2
    while True:
        assert j[pos] == 0xFF
        n = (j[pos + 2] << 8) | j[pos + 3]
        out.append((j[pos + 1], j[pos:pos + 2 + n]))
        pos += 2 + n
        if out[-1][0] == 0xDA:
            return out


def test_headers_of_the_reference_ad_hoc_frame_from_literals():
    """1680 x 900, 4:2:2, Annex-K tables: the frame the reference's own bitBufferTest builds
    (JPEG-Encoder-SIMD.cpp:73-86).  Oracle file and the product's host header writer against
    the bytes that follow from the reference's struct initialisers."""
    from jpegenc_amd import binding as B
    img = synth.generate("natural", 1680, 900)
    enc = O.encode(img, T.SCHEME_422, T.STANDARD_TABLES)
    segs = _segments(enc.jpeg)
    assert [f"{m:02x}" for m, _ in segs] == LIT["segment_order"]["markers"]
    assert segs[0][1].hex() == LIT["soi_hex"]["hex"]
    assert segs[1][1].hex() == LIT["app0_hex"]["hex"] == KATS["headers"]["app0_hex"]
    assert segs[2][1].hex() == LIT["sof0_1680x900_422_hex"]["hex"]
    assert [s[4] for _, s in segs[3:7]] == LIT["dht_layout"]["encoder_infos"]
    for _, s in segs[3:7]:
        assert s[:2].hex() == LIT["dht_layout"]["marker_hex"]
        assert ((s[2] << 8) | s[3]) == 2 + 1 + 16 + sum(s[5:21]) == len(s) - 2
    assert segs[7][1].hex() == LIT["dqt_luminance_id0_hex"]["hex"]
    assert segs[8][1].hex() == LIT["dqt_chrominance_id1_hex"]["hex"]
    assert segs[9][1].hex() == LIT["dqt_chrominance_id2_hex"]["hex"]
    assert segs[10][1].hex() == LIT["sos_hex"]["hex"] == KATS["headers"]["sos_hex"]
    assert enc.jpeg[-2:].hex() == LIT["eoi_hex"]["hex"]
    # the product's host side writes the same header bytes (no GPU involved)
    prm = B.make_params(1680, 900, T.SCHEME_422, T.STANDARD_TABLES)
    assert B.write_headers(prm, enc.lengths, enc.codes) == enc.jpeg[:enc.scan_offset]


def test_dht_of_seven_symbols_has_the_length_the_reference_struct_gives():
    # bitBufferTest's table: id 3, AC, symbols 0..6 (JPEG-Encoder-SIMD.cpp:77-83)
    from jpegenc_amd import binding as B
    counts = np.zeros(256, np.uint64)
    for s, n in zip(range(7), (1, 1, 3, 5, 6, 11, 13)):
        counts[s] = n
    lengths, codes, dht = O.huffman(counts, 3, 1)
    assert dht[:2].hex() == LIT["dht_layout"]["marker_hex"]
    assert (dht[2] << 8) | dht[3] == LIT["dht_layout"]["length_for_7_symbols"]
    assert dht[4] == LIT["dht_layout"]["info_id3_ac"]
    assert sorted(dht[21:]) == list(range(7))
    assert B.write_dht(lengths, codes, 3, 1) == bytes(dht)


@pytest.mark.parametrize("scheme", ["444", "422", "420", "411"])
def test_sof0_sampling_bytes_from_literals(scheme):
    enc = O.encode(synth.generate("natural", 40, 24), T.SCHEMES[scheme], T.STANDARD_TABLES)
    j = enc.jpeg
    assert [j[31], j[34], j[37]] == LIT["sof0_factor_bytes"][scheme]


def _category(v):
    for k, lo, hi in LIT["bit_category_ranges"]["ranges"]:
        if lo <= abs(v) <= hi:
            return k
    raise AssertionError(v)


def _amplitude(v, cat):
    pattern = v if v >= 0 else (~(-v)) & 0xFFFF
    return format(pattern & ((1 << cat) - 1), f"0{cat}b") if cat else ""


def test_amplitude_rule_examples():
    for v, cat, bits in LIT["amplitude_examples"]["cases"]:
        assert _category(v) == cat and _amplitude(v, cat) == bits


def test_scan_rebuilt_from_the_category_and_run_length_rules():
    """The oracle's entropy-coded segment against a plain restatement of Encoder.cpp:262-352
    (DC difference, run/size symbols with ZRL and EOB, amplitude bits) driven by the category
    table from the literals, on the oracle's own coefficients and code tables."""
    img = synth.generate("noise", 48, 32, seed=7)
    enc = O.encode(img, T.SCHEME_444, T.quality_tables(50))
    nblocks = enc.coef[0].shape[0]
    bits = []
    last = [0, 0, 0]
    for b in range(nblocks):
        for c in range(3):
            dc_t, ac_t = (0, 2) if c == 0 else (1, 3)
            blk = enc.coef[c][b].astype(int)
            diff = int(blk[0]) - last[c]
            last[c] = int(blk[0])
            cat = _category(diff)
            bits.append(format(int(enc.codes[dc_t][cat]), f"0{int(enc.lengths[dc_t][cat])}b"))
            bits.append(_amplitude(diff, cat))
            zeros = 0
            for i in range(1, 64):
                v = int(blk[i])
                if v == 0:
                    zeros += 1
                    continue
                while zeros > 15:
                    bits.append(format(int(enc.codes[ac_t][0xF0]), f"0{int(enc.lengths[ac_t][0xF0])}b"))
                    zeros -= 16
                cat = _category(v)
                sym = (zeros << 4) | cat
                bits.append(format(int(enc.codes[ac_t][sym]), f"0{int(enc.lengths[ac_t][sym])}b"))
                bits.append(_amplitude(v, cat))
                zeros = 0
            if zeros:
                bits.append(format(int(enc.codes[ac_t][0]), f"0{int(enc.lengths[ac_t][0])}b"))
    s = "".join(bits)
    s += "0" * (-len(s) % 8)                       # fillToByteBorder pads with zero bits, BitBuffer.cpp:12-16
    raw = bytes(int(s[i:i + 8], 2) for i in range(0, len(s), 8))
    scan = enc.jpeg[enc.scan_offset:enc.scan_offset + enc.scan_len]
    assert scan == raw.replace(b"\xff", b"\xff\x00")
