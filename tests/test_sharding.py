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
