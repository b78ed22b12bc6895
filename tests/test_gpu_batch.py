"""Batch entry point vs the CPU oracle, image by image (needs an MI355X)."""
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


def _run_batch(ctx, imgs, scheme, qt, slot=None, image_pad=0, row_pad=0, pad_ones=False):
    import torch
    n, h, w = len(imgs), imgs[0].shape[0], imgs[0].shape[1]
    stride = w * 3 + row_pad
    image_stride = stride * h + image_pad
    host = np.zeros(n * image_stride, np.uint8)
    for k, im in enumerate(imgs):
        view = host[k * image_stride:k * image_stride + stride * h].reshape(h, stride)
        view[:, :w * 3] = im.reshape(h, w * 3)
    d_rgb = torch.from_numpy(host).cuda()
    prm = B.make_params(w, h, scheme, qt, pad_ones=pad_ones)
    if slot is None:
        slot = int(B.compute_layout(prm).max_jpeg_bytes)
    d_out = torch.full((n * slot,), 0xAA, dtype=torch.uint8, device="cuda")
    lens = ctx.encode_batch(d_rgb.data_ptr(), B.RGB8, stride, image_stride, n, prm, d_out.data_ptr(), slot)
    out = d_out.cpu().numpy()
    return [out[k * slot:k * slot + int(lens[k])].tobytes() for k in range(n)], out, lens, slot


@pytest.mark.parametrize("scheme", ["420", "444", "422", "411"])
def test_batch_matches_oracle_per_image(ctx, scheme):
    # different content per image -> different Huffman tables and stream lengths per slot
    for (w, h, n) in [(64, 48, 5), (37, 29, 9), (250, 130, 3), (16, 16, 33), (8, 8, 70)]:
        imgs = [synth.generate("noise" if k % 3 == 0 else "natural", w, h, seed=17 * k + w) for k in range(n)]
        if n > 2:
            imgs[1] = np.zeros_like(imgs[1])            # flat image: shortest possible stream
            imgs[2] = np.full_like(imgs[2], 255)
        qt = T.quality_tables(90)
        files, out, lens, slot = _run_batch(ctx, imgs, T.SCHEMES[scheme], qt)
        for k, im in enumerate(imgs):
            assert files[k] == O.encode(im, T.SCHEMES[scheme], qt).jpeg, (scheme, w, h, k)
            # nothing written behind the file inside its slot
            assert np.all(out[k * slot + int(lens[k]):(k + 1) * slot] == 0xAA), (scheme, w, h, k)


def test_batch_strides_unaligned_and_pad_ones(ctx):
    imgs = [synth.generate("noise", 123, 77, seed=k) for k in range(6)]
    qt = T.NAMED_TABLES["ones"]
    for row_pad, image_pad in [(0, 0), (3, 0), (1, 5), (4, 64)]:
        files, _, _, _ = _run_batch(ctx, imgs, T.SCHEME_420, qt, image_pad=image_pad, row_pad=row_pad, pad_ones=True)
        for k, im in enumerate(imgs):
            assert files[k] == O.encode(im, T.SCHEME_420, qt, pad_ones=True).jpeg, (row_pad, image_pad, k)


def test_batch_equals_single_encodes_1080p(ctx):
    import torch
    w, h, n = 1920, 1080, 6
    imgs = [synth.generate("natural", w, h, seed=k) for k in range(n)]
    qt = T.quality_tables(90)
    files, _, _, _ = _run_batch(ctx, imgs, T.SCHEME_420, qt, slot=2 << 20)
    prm = B.make_params(w, h, T.SCHEME_420, qt)
    assert files[0] == O.encode(imgs[0], T.SCHEME_420, qt).jpeg
    for k in range(1, n):
        assert files[k] == ctx.encode_host(imgs[k], prm), k


def test_batch_of_one_and_errors(ctx):
    import torch
    img = synth.generate("natural", 100, 60)
    qt = T.STANDARD_TABLES
    files, _, _, _ = _run_batch(ctx, [img], T.SCHEME_444, qt)
    assert files[0] == O.encode(img, T.SCHEME_444, qt).jpeg
    # slot too small for the file
    with pytest.raises(B.JpegHipError) as e:
        _run_batch(ctx, [synth.generate("noise", 100, 60, seed=k) for k in range(3)], T.SCHEME_444,
                   T.NAMED_TABLES["ones"], slot=4096)
    assert e.value.status == B.ERR_BUFFER_TOO_SMALL
    # restart intervals are not part of batch mode
    prm = B.make_params(100, 60, T.SCHEME_444, qt, restart_interval=4)
    buf = torch.zeros(1 << 20, dtype=torch.uint8, device="cuda")
    with pytest.raises(B.JpegHipError) as e:
        ctx.encode_batch(buf.data_ptr(), B.RGB8, 300, 18000, 2, prm, buf.data_ptr(), 1 << 18)
    assert e.value.status == B.ERR_INVALID_ARGUMENT
    # the context still works for single images afterwards
    prm = B.make_params(100, 60, T.SCHEME_444, qt)
    assert ctx.encode_host(img, prm) == O.encode(img, T.SCHEME_444, qt).jpeg


def test_host_batch_small_frames_and_strides(ctx):
    w, h, n = 123, 77, 9
    imgs = [synth.generate("noise" if k % 2 else "natural", w, h, seed=50 + k) for k in range(n)]
    qt = T.quality_tables(90)
    prm = B.make_params(w, h, T.SCHEME_420, qt)
    files = ctx.encode_batch_host(np.stack(imgs), prm, slot_bytes=1 << 16)
    for k, im in enumerate(imgs):
        assert files[k] == O.encode(im, T.SCHEME_420, qt).jpeg, k
    # padded rows and padded images (a view into a larger array)
    big = np.zeros((n, h + 3, w + 5, 3), np.uint8)
    big[:, :h, :w] = np.stack(imgs)
    view = big[:, :h, :w]
    fmt, out, lens = B.RGB8, np.zeros(n << 16, np.uint8), np.zeros(n, np.int64)
    import ctypes as C
    ctx._check(B.lib().jpeg_hip_encode_batch_host(ctx._h, C.c_void_p(view.ctypes.data), fmt, view.strides[1], view.strides[0], n,
                                                  C.byref(prm), C.c_void_p(out.ctypes.data), 1 << 16,
                                                  C.c_void_p(lens.ctypes.data)))
    for k, im in enumerate(imgs):
        assert out[(k << 16):(k << 16) + int(lens[k])].tobytes() == files[k], k
    # float frames
    fl = (np.stack(imgs[:3]).astype(np.float32) / np.float32(255.0))
    got = ctx.encode_batch_host(fl, prm, slot_bytes=1 << 16)
    for k in range(3):
        assert got[k] == O.encode(fl[k], T.SCHEME_420, qt).jpeg, k


def test_host_batch_pipelines_several_passes(ctx):
    # 1080p frames: about 30 fit one staging block, so 70 frames take three passes
    w, h, n = 1920, 1080, 70
    base = [synth.generate("natural", w, h, seed=k) for k in range(4)]
    frames = np.stack([base[k % 4] for k in range(n)])
    for k in range(n):                                          # make every frame different
        frames[k, :16, :16, :] = 3 * k
    qt = T.quality_tables(90)
    prm = B.make_params(w, h, T.SCHEME_420, qt)
    files = ctx.encode_batch_host(frames, prm, slot_bytes=1 << 20)
    for k in (0, 1, 29, 30, 31, 59, 60, 69):
        assert files[k] == ctx.encode_host(frames[k], prm), k
    assert files[0] == O.encode(frames[0], T.SCHEME_420, qt).jpeg
    assert len(set(files)) == n


def test_device_batch_split_into_passes(ctx):
    # 70 frames of 3840x2160 4:2:0 are 13.6 M blocks: more than one pass of the batch entry point
    import torch
    from jpegenc_amd import synth_torch
    w, h, n = 3840, 2160, 70
    base = synth_torch.generate("natural", w, h, device="cuda")
    frames = base.reshape(1, h, w, 3).repeat(n, 1, 1, 1).contiguous()
    for k in range(n):
        frames[k, 8 * (k % 200):8 * (k % 200) + 64, :256, :] = (3 * k) & 255
    qt = T.quality_tables(90)
    for std in (False, True):
        prm = B.make_params(w, h, T.SCHEME_420, qt, standard_huffman=std)
        slot = 4 << 20
        out = torch.zeros(n * slot, dtype=torch.uint8, device="cuda")
        lens = ctx.encode_batch(frames.data_ptr(), B.RGB8, w * 3, w * h * 3, n, prm, out.data_ptr(), slot)
        one = torch.zeros(slot, dtype=torch.uint8, device="cuda")
        for k in (0, 1, 33, 66, 67, 68, 69):
            m = ctx.encode(frames[k].data_ptr(), B.RGB8, w * 3, prm, one.data_ptr(), slot)
            assert m == int(lens[k]), (std, k)
            assert torch.equal(one[:m], out[k * slot:k * slot + m]), (std, k)


def test_long_stream_form_of_the_stuffing_pass(ctx, monkeypatch):
    # Streams of more than 16384 chunks (64 MiB) get the stuffing counts of 64 chunks added
    # up by a kernel of their own; the switch lowers that limit so that small inputs take the
    # same route: a batch with image boundaries inside chunks and a single noisy image with
    # restart markers.
    monkeypatch.setenv("JPEGENC_GROUP_SUMS_FROM", "2")
    qt = T.NAMED_TABLES["ones"]
    imgs = [synth.generate("noise", 200, 120, seed=k) for k in range(20)]
    files, out, lens, slot = _run_batch(ctx, imgs, T.SCHEME_420, qt)
    for k, im in enumerate(imgs):
        assert files[k] == O.encode(im, T.SCHEME_420, qt).jpeg, k
        assert np.all(out[k * slot + int(lens[k]):(k + 1) * slot] == 0xAA), k
    big = synth.generate("noise", 1333, 811, seed=5)
    prm = B.make_params(1333, 811, T.SCHEME_444, qt, restart_interval=7)
    assert ctx.encode_host(big, prm) == O.encode(big, T.SCHEME_444, qt, restart_interval=7).jpeg
