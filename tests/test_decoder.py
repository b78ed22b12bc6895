"""Verification decoder: header parsing on the CPU; on the GPU the decoded coefficients
against the encoder's, the pixels against the numpy restatement and an independent
decoder, and PSNR."""
import io

import numpy as np
import pytest

import oracle as O
import decode_ref as D
from jpegenc_amd import binding as B, synth, tables as T


def test_decode_info_parses_oracle_files():
    img = synth.generate("natural", 123, 77)
    for scheme in ("444", "420", "411", "422"):
        for ri in (0, 7):
            enc = O.encode(img, T.SCHEMES[scheme], T.STANDARD_TABLES, restart_interval=ri)
            info = B.decode_info(enc.jpeg)
            assert (info.width, info.height, info.restart_interval) == (123, 77, ri)
            inv = T.inverse_factors(T.SCHEMES[scheme])
            assert [(info.h_samp[c], info.v_samp[c]) for c in range(3)] == inv
            n = sum(len(enc.coef[c]) for c in range(3))
            assert info.total_blocks == n


def test_decode_info_rejects_garbage():
    for bad in (b"", b"\xff\xd8", b"notajpeg" * 10, b"\xff\xd8\xff\xe0\x00\x10" + b"\0" * 30):
        with pytest.raises(B.JpegHipError):
            B.decode_info(bad)


@pytest.fixture(scope="module")
def ctx():
    c = B.Context(0)
    yield c
    c.close()


def _scan_order(enc, scheme):
    """Oracle coefficients (per channel, MCU order) -> one array in scan order."""
    inv = T.inverse_factors(T.SCHEMES[scheme])
    per = [h * v for (h, v) in inv]
    n_mcu = len(enc.coef[0]) // per[0]
    rows = []
    for c in range(3):
        rows.append(enc.coef[c].reshape(n_mcu, per[c], 64))
    return np.concatenate(rows, axis=1).reshape(-1, 64)


@pytest.mark.gpu
@pytest.mark.parametrize("scheme", ["444", "420", "422", "411", "422h"])
def test_decoded_coefficients_equal_encoder_coefficients(ctx, scheme):
    for (w, h, kind, tn, ri, std) in [(37, 29, "natural", "annexk", 0, False), (250, 130, "noise", "ones", 0, False),
                                      (123, 77, "noise", "ps5", 5, False), (64, 48, "natural", "annexk", 1, True),
                                      (1, 1, "natural", "annexk", 0, False)]:
        img = synth.generate(kind, w, h, seed=3)
        qt = T.NAMED_TABLES[tn]
        enc = O.encode(img, T.SCHEMES[scheme], qt, restart_interval=ri)
        prm = B.make_params(w, h, T.SCHEMES[scheme], qt, restart_interval=ri, standard_huffman=std)
        jpg = ctx.encode_host(img, prm)
        rgb, info = ctx.decode_host(jpg)
        got = ctx.decoded_coefficients(info)
        assert np.array_equal(got, _scan_order(enc, scheme)), (scheme, w, h, kind, tn, ri, std)
        assert rgb.shape == (h, w, 3)


@pytest.mark.gpu
@pytest.mark.parametrize("scheme", ["444", "420", "411"])
def test_pixels_match_numpy_restatement(ctx, scheme):
    for (w, h, kind) in [(48, 32, "natural"), (37, 29, "noise")]:
        img = synth.generate(kind, w, h, seed=9)
        qt = T.quality_tables(90)
        enc = O.encode(img, T.SCHEMES[scheme], qt)
        inv = T.inverse_factors(T.SCHEMES[scheme])
        want = D.decode_from_coefficients(enc.coef, np.asarray(qt, np.float32).reshape(3, 64), w, h,
                                          [i[0] for i in inv], [i[1] for i in inv])
        got, _ = ctx.decode_host(enc.jpeg)
        assert np.array_equal(got, want), (scheme, w, h, kind, int(np.abs(got.astype(int) - want.astype(int)).max()))


@pytest.mark.gpu
def test_against_independent_decoder_and_psnr(ctx):
    import torch
    from PIL import Image
    w, h = 320, 200
    img = synth.generate("natural", w, h)
    qt = T.quality_tables(90)
    d_img = torch.from_numpy(img).cuda()
    results = {}
    for name, kw in [("opt", {}), ("std", {"standard_huffman": True}), ("rst", {"restart_interval": 20})]:
        jpg = ctx.encode_host(img, B.make_params(w, h, T.SCHEME_444, qt, **kw))
        mine, _ = ctx.decode_host(jpg)
        pil = np.asarray(Image.open(io.BytesIO(jpg)).convert("RGB"))
        # float direct IDCT here, integer IDCT and integer colour conversion there
        assert np.abs(mine.astype(int) - pil.astype(int)).max() <= 3, name
        d_dec = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda")
        ctx.decode(jpg, d_dec.data_ptr(), w * 3, d_dec.numel())
        assert np.array_equal(d_dec.cpu().numpy(), mine)
        mse, psnr = ctx.psnr(d_img.data_ptr(), w * 3, d_dec.data_ptr(), w * 3, w, h)
        ref_mse = float(((img.astype(np.float64) - mine) ** 2).mean())
        assert abs(mse - ref_mse) < 1e-9 * max(1.0, ref_mse)
        assert psnr > 35.0
        results[name] = (mse, mine)
    # same coefficients whichever entropy code or restart interval: same pixels, same PSNR
    assert np.array_equal(results["opt"][1], results["std"][1])
    assert np.array_equal(results["opt"][1], results["rst"][1])
    assert results["opt"][0] == results["std"][0] == results["rst"][0]
    # identical images: infinite PSNR
    mse, psnr = ctx.psnr(d_img.data_ptr(), w * 3, d_img.data_ptr(), w * 3, w, h)
    assert mse == 0.0 and np.isinf(psnr)


@pytest.mark.gpu
def test_subsampled_file_decodes_close_to_independent_decoder(ctx):
    from PIL import Image
    w, h = 256, 160
    img = synth.generate("natural", w, h, seed=5)
    jpg = ctx.encode_host(img, B.make_params(w, h, T.SCHEME_420, T.quality_tables(90)))
    mine, _ = ctx.decode_host(jpg)
    pil = np.asarray(Image.open(io.BytesIO(jpg)).convert("RGB")).astype(np.float64)
    # sample replication here, smoothing upsampler there: close, not equal
    mse = ((mine - pil) ** 2).mean()
    assert 10 * np.log10(255.0 ** 2 / max(mse, 1e-12)) > 38.0


@pytest.mark.gpu
def test_damaged_stream_is_reported(ctx):
    img = synth.generate("noise", 64, 64)
    jpg = bytearray(ctx.encode_host(img, B.make_params(64, 64, T.SCHEME_444, T.NAMED_TABLES["ones"])))
    with pytest.raises(B.JpegHipError):
        ctx.decode_host(bytes(jpg[:len(jpg) // 2]))       # no EOI


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["4k_444", "8k_band_420_restarts"])
def test_round_trip_at_benchmark_sizes(ctx, case):
    """encode -> decode gives back the very coefficients the encoder held, and a sane PSNR."""
    import torch
    from jpegenc_amd import synth_torch
    if case == "4k_444":
        w, h, scheme, ri = 3840, 2160, T.SCHEME_444, 0
    else:
        w, h, scheme, ri = 8192, 2048, T.SCHEME_420, 8192 // 16
    qt = T.quality_tables(90)
    rgb = synth_torch.generate("natural", w, h, device="cuda")
    prm = B.make_params(w, h, scheme, qt, restart_interval=ri)
    lay = B.compute_layout(prm)
    out = torch.zeros(w * h * 2, dtype=torch.uint8, device="cuda")
    n = ctx.encode(rgb.data_ptr(), B.RGB8, w * 3, prm, out.data_ptr(), out.numel())
    held = np.concatenate([ctx.coefficients(c, lay).reshape(-1, lay.blocks[c] // (lay.total_blocks // lay.blocks_per_mcu), 64)
                           for c in range(3)], axis=1).reshape(-1, 64)
    jpg = out[:n].cpu().numpy().tobytes()
    dec = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda")
    info = ctx.decode(jpg, dec.data_ptr(), w * 3, dec.numel())
    assert (info.width, info.height, info.restart_interval) == (w, h, ri)
    assert np.array_equal(ctx.decoded_coefficients(info), held)
    mse, psnr = ctx.psnr(rgb.data_ptr(), w * 3, dec.data_ptr(), w * 3, w, h)
    assert psnr > 33.0, psnr
