"""Sharded encode (one band of MCU rows per rank, histogram all-reduce, RSTn stitching).

CPU: world_size-2 gloo run in which the oracle stands in for the GPU kernels, so the
product's host logic (band split, all-reduce plumbing, table builder, header writer, piece
order) is what is under test.  GPU: two ranks sharing cuda:0 through the C ABI.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle as O
from jpegenc_amd import binding as B, sharding, synth, tables as T

W, H = 200, 150
SCHEME = "420"


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _init(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)


def _cpu_worker(rank, world, port, q):
    _init(rank, world, port)
    try:
        img = synth.generate("natural", W, H)
        scheme, qt = T.SCHEMES[SCHEME], T.quality_tables(90)
        whole = B.compute_layout(B.make_params(W, H, scheme, qt))
        r0, r1 = sharding.band_rows(whole.mcu_rows, world, rank)
        ri = sharding.row_restart_interval(whole.mcus_per_row)
        band = img[r0 * whole.mcu_height:min(H, r1 * whole.mcu_height)]
        # "analyze": the band's own histograms (oracle in place of the kernels)
        own = O.encode(band, scheme, qt, restart_interval=ri)
        hist = torch.from_numpy(own.hist.view(np.int64).copy())
        sharding.all_reduce_histograms(hist)
        hsum = sharding.as_uint64(hist.numpy())
        # "finish": tables and headers from the product's host code, scan from the oracle
        lengths = np.zeros((4, 256), np.uint8)
        codes = np.zeros((4, 256), np.uint16)
        for t in range(4):
            lengths[t], codes[t] = B.huffman_from_histogram(hsum[t])
        enc = O.encode(band, scheme, qt, restart_interval=ri, hist_override=hsum, rst_base=r0, header_height=H)
        assert np.array_equal(enc.lengths, lengths) and np.array_equal(enc.codes, codes)
        piece = b""
        if rank == 0:
            prm = B.make_params(W, H, scheme, qt, restart_interval=ri)
            piece += B.write_headers(prm, lengths, codes)
        else:
            piece += bytes([0xFF, 0xD0 + ((r0 - 1) & 7)])   # marker in front of the band's first interval
        piece += enc.scan
        if rank == world - 1:
            piece += b"\xff\xd9"
        pieces = [None] * world
        dist.all_gather_object(pieces, piece)
        if rank == 0:
            q.put(sharding.concatenate_pieces(pieces))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_sharded_encode_matches_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_cpu_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    img = synth.generate("natural", W, H)
    scheme, qt = T.SCHEMES[SCHEME], T.quality_tables(90)
    lay = B.compute_layout(B.make_params(W, H, scheme, qt))
    want = O.encode(img, scheme, qt, restart_interval=lay.mcus_per_row).jpeg
    assert got == want


def _cpu_failing_worker(rank, world, port, q):
    """As _cpu_worker, but rank 1's output buffer is too small for its piece: the product's rule is
    that every rank takes part in every collective and that one more small collective tells all of
    them whether the file is complete."""
    _init(rank, world, port)
    try:
        img = synth.generate("natural", W, H)
        scheme, qt = T.SCHEMES[SCHEME], T.quality_tables(90)
        whole = B.compute_layout(B.make_params(W, H, scheme, qt))
        r0, r1 = sharding.band_rows(whole.mcu_rows, world, rank)
        ri = sharding.row_restart_interval(whole.mcus_per_row)
        band = img[r0 * whole.mcu_height:min(H, r1 * whole.mcu_height)]
        own = O.encode(band, scheme, qt, restart_interval=ri)
        hist = torch.from_numpy(own.hist.view(np.int64).copy())
        sharding.all_reduce_histograms(hist)            # nobody leaves before the exchange
        hsum = sharding.as_uint64(hist.numpy())
        enc = O.encode(band, scheme, qt, restart_interval=ri, hist_override=hsum, rst_base=r0, header_height=H)
        capacity = 10 if rank == 1 else 1 << 20
        local_ok = len(enc.scan) + 2 <= capacity
        failed = sharding.failed_ranks(local_ok)
        q.put((rank, local_ok, failed))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_one_rank_fails_and_every_rank_learns_it():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_cpu_failing_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0          # nobody hung, nobody crashed
    assert got == [(0, True, 1), (1, False, 1)]


def test_band_rows_partition():
    for rows in (1, 7, 64, 1024):
        for world in (1, 2, 3, 4, 8):
            if world > rows:
                with pytest.raises(ValueError):
                    sharding.band_rows(rows, world, 0)
                continue
            bands = [sharding.band_rows(rows, world, r) for r in range(world)]
            assert bands[0][0] == 0 and bands[-1][1] == rows
            assert all(a[1] == b[0] for a, b in zip(bands, bands[1:]))
            sizes = [b - a for a, b in bands]
            assert max(sizes) - min(sizes) <= 1 and min(sizes) >= 1


def _gpu_worker(rank, world, port, q):
    _init(rank, world, port)
    try:
        img = synth.generate("natural", 333, 270)
        scheme, qt = T.SCHEMES[SCHEME], T.quality_tables(90)
        ctx = B.Context(0)
        enc = sharding.HipBandEncoder(ctx, B, 333, 270, scheme, qt, world, rank)
        y0, y1 = enc.pixel_rows
        d_rgb = torch.from_numpy(np.ascontiguousarray(img[y0:y1])).cuda()
        d_hist = torch.zeros((4, 256), dtype=torch.int64, device="cuda")
        d_out = torch.zeros(1 << 20, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        enc.analyze(d_rgb.data_ptr(), B.RGB8, 333 * 3, d_hist.data_ptr())
        ext = torch.cuda.ExternalStream(ctx.stream)
        ext.synchronize()
        h = d_hist.cpu()                      # gloo: the reduction goes through host memory
        sharding.all_reduce_histograms(h)
        d_hist.copy_(h)
        torch.cuda.synchronize()
        n = enc.finish(d_hist.data_ptr(), d_out.data_ptr(), d_out.numel())
        piece = d_out[:n].cpu().numpy().tobytes()
        pieces = [None] * world
        dist.all_gather_object(pieces, piece)
        if rank == 0:
            q.put(sharding.concatenate_pieces(pieces))
        ctx.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_ranks_sharing_one_gpu_match_oracle(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gpu_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=240)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    img = synth.generate("natural", 333, 270)
    scheme, qt = T.SCHEMES[SCHEME], T.quality_tables(90)
    lay = B.compute_layout(B.make_params(333, 270, scheme, qt))
    want = O.encode(img, scheme, qt, restart_interval=lay.mcus_per_row).jpeg
    assert got == want
    from PIL import Image
    import io
    dec = np.array(Image.open(io.BytesIO(got)).convert("RGB"))
    assert dec.shape == img.shape


@pytest.mark.gpu
def test_full_size_bands_equal_whole_image_and_round_trip():
    """The 16384 x 16384 4:2:0 q=90 benchmark image, far beyond what the CPU checker encodes in
    seconds, through properties that do not depend on the size: the pieces of eight bands put
    together are byte for byte the file of one whole-image call with the same restart interval;
    that file decodes to exactly the coefficients the encoder held; the decoded pixels are close
    to the source."""
    from jpegenc_amd import synth_torch
    w = h = 16384
    world = 8
    scheme, qt = T.SCHEMES["420"], T.quality_tables(90)
    rgb = synth_torch.generate("natural", w, h, device="cuda")
    ctxs = [B.Context(0) for _ in range(world)]
    encs = [sharding.HipBandEncoder(ctxs[r], B, w, h, scheme, qt, world, r) for r in range(world)]
    hists = [torch.zeros((4, 256), dtype=torch.int64, device="cuda") for _ in range(world)]
    torch.cuda.synchronize()
    for r, enc in enumerate(encs):
        y0, y1 = enc.pixel_rows
        enc.analyze(rgb[y0:y1].data_ptr(), B.RGB8, w * 3, hists[r].data_ptr())
    for c in ctxs:
        torch.cuda.ExternalStream(c.stream).synchronize()
    total = torch.stack(hists).sum(0).contiguous()      # what the all-reduce leaves on every rank
    torch.cuda.synchronize()
    out = torch.zeros(96 << 20, dtype=torch.uint8, device="cuda")
    pieces = []
    for enc in encs:
        n = enc.finish(total.data_ptr(), out.data_ptr(), out.numel())
        pieces.append(out[:n].clone())
    sharded = torch.cat(pieces)
    for c in ctxs[1:]:
        c.close()

    ctx = ctxs[0]
    lay0 = B.compute_layout(B.make_params(w, h, scheme, qt))
    prm = B.make_params(w, h, scheme, qt, restart_interval=lay0.mcus_per_row)
    lay = B.compute_layout(prm)
    n = ctx.encode(rgb.data_ptr(), B.RGB8, w * 3, prm, out.data_ptr(), out.numel())
    assert n == sharded.numel()
    assert torch.equal(out[:n], sharded)
    del sharded, pieces

    held = np.concatenate([ctx.coefficients(c, lay).reshape(-1, lay.blocks[c] // (lay.total_blocks // lay.blocks_per_mcu), 64)
                           for c in range(3)], axis=1).reshape(-1, 64)
    jpg = out[:n].cpu().numpy().tobytes()
    dec = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda")
    info = ctx.decode(jpg, dec.data_ptr(), w * 3, dec.numel())
    assert (info.width, info.height, info.restart_interval) == (w, h, lay0.mcus_per_row)
    assert np.array_equal(ctx.decoded_coefficients(info), held)
    mse, psnr = ctx.psnr(rgb.data_ptr(), w * 3, dec.data_ptr(), w * 3, w, h)
    assert psnr > 33.0, psnr
    ctx.close()


@pytest.mark.gpu
def test_sharded_entry_point_over_an_rccl_communicator_of_one():
    """jpeg_hip_encode_sharded (analyze, ncclAllReduce on the context's stream, finish) with a
    world of one rank: the same bytes as jpeg_hip_encode with the same restart interval, and as
    the oracle; then the exchange step alone between the two halves."""
    dev = torch.device("cuda", 0)
    qt = T.quality_tables(90)
    comm = B.Communicator(B.comm_unique_id(), 0, 1, 0)
    ctx, plain = B.Context(0), B.Context(0)
    try:
        for (w, h, scheme, std) in [(640, 360, "420", False), (333, 210, "444", False), (640, 360, "420", True)]:
            img = synth.generate("natural", w, h, seed=w)
            enc = sharding.HipBandEncoder(ctx, B, w, h, T.SCHEMES[scheme], qt, 1, 0)
            prm = enc.params
            prm.standard_huffman = 1 if std else 0
            rgb = torch.from_numpy(img).to(dev)
            out_a = torch.zeros(1 << 20, dtype=torch.uint8, device=dev)
            out_b = torch.zeros(1 << 20, dtype=torch.uint8, device=dev)
            n_a = ctx.encode_sharded(comm, rgb.data_ptr(), B.RGB8, w * 3, prm, out_a.data_ptr(), out_a.numel())
            n_b = plain.encode(rgb.data_ptr(), B.RGB8, w * 3, prm, out_b.data_ptr(), out_b.numel())
            got = out_a[:n_a].cpu().numpy().tobytes()
            assert got == out_b[:n_b].cpu().numpy().tobytes(), (w, h, scheme, std)
            if not std:
                assert got == O.encode(img, T.SCHEMES[scheme], qt, restart_interval=prm.restart_interval).jpeg
                hist = torch.zeros((4, 256), dtype=torch.int64, device=dev)
                ctx.analyze_band(rgb.data_ptr(), B.RGB8, w * 3, prm, hist.data_ptr())
                ctx.allreduce_histograms(comm, hist.data_ptr())
                n_c = ctx.finish_band(hist.data_ptr(), out_a.data_ptr(), out_a.numel())
                assert out_a[:n_c].cpu().numpy().tobytes() == got
    finally:
        ctx.close()
        plain.close()
        comm.close()


@pytest.mark.gpu
def test_sharded_entry_point_takes_part_in_the_exchange_when_its_own_step_fails():
    """A rank whose local step fails (here: a row stride smaller than a row, which
    jpeg_hip_analyze_band refuses) must still enter the all-reduce, with its failure flag up, and
    return its own error afterwards; the communicator and the context serve the next image as if
    nothing had happened.  With a world of one the rank's peers are absent, so what shows is that
    the call returns the local status (not the peers' one) and leaves everything usable."""
    dev = torch.device("cuda", 0)
    qt = T.quality_tables(90)
    comm = B.Communicator(B.comm_unique_id(), 0, 1, 0)
    ctx = B.Context(0)
    try:
        w, h = 640, 360
        img = synth.generate("natural", w, h, seed=5)
        enc = sharding.HipBandEncoder(ctx, B, w, h, T.SCHEMES["420"], qt, 1, 0)
        rgb = torch.from_numpy(img).to(dev)
        out = torch.zeros(1 << 20, dtype=torch.uint8, device=dev)
        want = O.encode(img, T.SCHEMES["420"], qt, restart_interval=enc.params.restart_interval).jpeg
        for _ in range(2):
            with pytest.raises(B.JpegHipError) as e:
                ctx.encode_sharded(comm, rgb.data_ptr(), B.RGB8, w * 3 - 1, enc.params, out.data_ptr(), out.numel())
            assert e.value.status == B.ERR_INVALID_ARGUMENT and "stride" in str(e.value)
            n = ctx.encode_sharded(comm, rgb.data_ptr(), B.RGB8, w * 3, enc.params, out.data_ptr(), out.numel())
            assert out[:n].cpu().numpy().tobytes() == want
        # a failure behind the exchange is local ...
        with pytest.raises(B.JpegHipError) as e:
            ctx.encode_sharded(comm, rgb.data_ptr(), B.RGB8, w * 3, enc.params, out.data_ptr(), 4096)
        assert e.value.status == B.ERR_BUFFER_TOO_SMALL
        # ... and the status collective tells every rank about it
        assert ctx.allreduce_status(comm, e.value.status) == 1
        assert ctx.allreduce_status(comm, B.OK) == 0
        n = ctx.encode_sharded(comm, rgb.data_ptr(), B.RGB8, w * 3, enc.params, out.data_ptr(), out.numel())
        assert out[:n].cpu().numpy().tobytes() == want
    finally:
        ctx.close()
        comm.close()
