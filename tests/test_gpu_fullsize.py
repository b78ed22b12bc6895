"""Full-size checks of the noise workload (8192 x 8192, q=100, all-ones tables), whose stream is
far beyond what the CPU checker encodes in seconds -- 196 MB in 4:4:4, 1.56 * 10^9 bit
positions, some 48 thousand stuffing chunks (the group sums of the stuffing pass are taken without
any switch) -- through properties that do not depend on the size.  BitBuffer semantics as in
BitBuffer.cpp:128-204."""
import numpy as np
import pytest
import torch

from jpegenc_amd import binding as B, sharding, synth, tables as T

pytestmark = pytest.mark.gpu


def test_generator_matches_the_host_generators():
    ctx = B.Context(0)
    try:
        for kind in ("noise", "natural"):
            w, h = 203, 117
            out = torch.zeros((3, h, w, 3), dtype=torch.uint8, device="cuda")
            ctx.synthetic_images(kind, w, 0, h, 1000, out.data_ptr(), w * 3, w * h * 3, 3)
            for k in range(3):
                assert np.array_equal(out[k].cpu().numpy(), synth.generate(kind, w, h, seed=1000 + k)), (kind, k)
            # a band of rows
            band = torch.zeros((40, w, 3), dtype=torch.uint8, device="cuda")
            ctx.synthetic_images(kind, w, 30, 70, 7, band.data_ptr(), w * 3)
            assert np.array_equal(band.cpu().numpy(), synth.generate(kind, w, h, seed=7)[30:70])
    finally:
        ctx.close()


def test_a_tensor_torch_is_still_filling_can_be_handed_over():
    """The context works on its own stream; the binding orders it behind torch's current stream
    whenever it is handed a device pointer (jpeg_hip_wait_for_stream).  Without that order the
    zero fill below, queued behind a gigabyte-sized one, would run after the generator and wipe
    its output (this happened once, with nothing but a small fill in the way)."""
    ctx = B.Context(0)
    try:
        w, h = 2048, 1024
        want = synth.generate("noise", w, h, seed=9)
        for _ in range(4):
            busy = torch.zeros(1 << 30, dtype=torch.uint8, device="cuda")     # keeps torch's stream busy for a while
            out = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda")    # queued behind it
            ctx.synthetic_images("noise", w, 0, h, 9, out.data_ptr(), w * 3)
            assert np.array_equal(out.cpu().numpy(), want)
            del busy
        # and the other way round: a result of a call that returns early, read by torch at once
        prm = B.make_params(w, h, T.SCHEME_420, T.quality_tables(90))
        rgb = torch.from_numpy(want).cuda()
        hist = torch.zeros((4, 256), dtype=torch.int64, device="cuda")
        ctx.analyze_band(rgb.data_ptr(), B.RGB8, w * 3, prm, hist.data_ptr())
        ctx.release_to_torch()
        got = hist.cpu().numpy()
        assert np.array_equal(got.view(np.uint64), ctx.histograms())
    finally:
        ctx.close()


@pytest.mark.parametrize("sname", ["444", "420"])
def test_noise_q100_full_size(sname):
    w = h = 8192
    scheme, qt = T.SCHEMES[sname], T.quality_tables(100)
    dev = torch.device("cuda", 0)
    ctx = B.Context(0)
    rgb = torch.empty(h * w * 3, dtype=torch.uint8, device=dev)
    ctx.synthetic_images("noise", w, 0, h, 42, rgb.data_ptr(), w * 3)
    out = torch.zeros(224 << 20, dtype=torch.uint8, device=dev)

    # 1. One scan, as the reference writes it.  Its length follows from the histograms and the
    # tables (code length + amplitude bits per symbol, padded to a byte) plus one 0x00 behind every
    # 0xFF of the entropy-coded data -- and no 0xFF of the scan may lack its 0x00.
    prm = B.make_params(w, h, scheme, qt)
    lay = B.compute_layout(prm)
    n = ctx.encode(rgb.data_ptr(), B.RGB8, w * 3, prm, out.data_ptr(), out.numel())
    lengths, codes = ctx.get_huffman()
    ctx.begin(prm)
    ctx.bind_pixels(rgb.data_ptr(), B.RGB8, w * 3)
    ctx.transform()
    hist = ctx.histograms().astype(np.int64)
    extra = np.stack([np.arange(256), np.arange(256), np.arange(256) & 15, np.arange(256) & 15])
    bits = int((hist * (lengths.astype(np.int64) + extra)).sum())
    assert bits > (1500000000 if sname == "444" else 500000000)   # 1.56 * 10^9 bit positions in 4:4:4
    header = len(B.write_headers(prm, lengths, codes))
    scan = out[header:n - 2]
    assert out[n - 2:n].cpu().tolist() == [0xFF, 0xD9]
    ff = scan == 0xFF
    pairs = int((ff[:-1] & (scan[1:] == 0)).sum())
    assert int(ff.sum()) == pairs and not bool(ff[-1])
    assert scan.numel() == (bits + 7) // 8 + pairs
    whole_single_scan_len = n
    del scan, ff

    # 2. With a restart interval of one MCU row: eight bands put together are the file of one call.
    world = 8
    ctxs = [ctx] + [B.Context(0) for _ in range(world - 1)]
    encs = [sharding.HipBandEncoder(ctxs[r], B, w, h, scheme, qt, world, r) for r in range(world)]
    hists = [torch.zeros((4, 256), dtype=torch.int64, device=dev) for _ in range(world)]
    for r, enc in enumerate(encs):
        y0, y1 = enc.pixel_rows
        enc.analyze(rgb[y0 * w * 3:].data_ptr(), B.RGB8, w * 3, hists[r].data_ptr())
    for c in ctxs:
        torch.cuda.ExternalStream(c.stream).synchronize()
    total = torch.stack(hists).sum(0).contiguous()
    # (the DC statistics differ from the single scan's: the predictor restarts with every interval)
    assert np.array_equal(total.cpu().numpy()[2:], hist[2:])
    torch.cuda.synchronize()
    pieces = []
    for enc in encs:
        m = enc.finish(total.data_ptr(), out.data_ptr(), out.numel())
        pieces.append(out[:m].clone())
    sharded = torch.cat(pieces)
    del pieces
    for c in ctxs[1:]:
        c.close()
    prm_ri = B.make_params(w, h, scheme, qt, restart_interval=lay.mcus_per_row)
    n_ri = ctx.encode(rgb.data_ptr(), B.RGB8, w * 3, prm_ri, out.data_ptr(), out.numel())
    assert n_ri == sharded.numel() and torch.equal(out[:n_ri], sharded)
    # markers, padding and restarting predictors (other DC tables) change the size only a little
    assert abs(n_ri - whole_single_scan_len) < whole_single_scan_len // 1000
    del sharded

    # 3. The file decodes to exactly the coefficients the encoder held.
    lay_ri = B.compute_layout(prm_ri)
    per_mcu = [int(lay_ri.blocks[c]) // (int(lay_ri.total_blocks) // lay_ri.blocks_per_mcu) for c in range(3)]
    held = np.concatenate([ctx.coefficients(c, lay_ri).reshape(-1, per_mcu[c], 64) for c in range(3)], axis=1).reshape(-1, 64)
    jpg = out[:n_ri].cpu().numpy().tobytes()
    dec = torch.zeros((h, w, 3), dtype=torch.uint8, device=dev)
    info = ctx.decode(jpg, dec.data_ptr(), w * 3, dec.numel())
    assert (info.width, info.height, info.restart_interval) == (w, h, lay.mcus_per_row)
    assert np.array_equal(ctx.decoded_coefficients(info), held)
    ctx.close()


@pytest.mark.parametrize("blocks", [140000, 160000, 240000, 260000, 390000, 410000, 490000, 510000])
def test_sizes_on_either_side_of_the_kernel_thresholds(blocks):
    """Images just below and just above the sizes at which the library changes kernels (bit writer
    on groups from 150 k blocks, histogram pass on groups from 250 k, the two halves of the blocks in
    two planes from 400 k, deferred placement from 500 k), with their real sizes and no diagnostic switch: whole files against the CPU checker."""
    import oracle as O
    w = 2560
    mcus = -(-blocks // 6)                     # 4:2:0: six blocks per MCU
    h = -(-mcus // (w // 16)) * 16 - 5         # ragged last MCU row
    img = synth.generate("natural", w, h, seed=blocks % 97)
    qt = T.quality_tables(90)
    ctx = B.Context(0)
    try:
        for ri in (0, w // 16):
            prm = B.make_params(w, h, T.SCHEME_420, qt, restart_interval=ri)
            assert abs(int(B.compute_layout(prm).total_blocks) - blocks) < 2000
            assert ctx.encode_host(img, prm) == O.encode(img, T.SCHEME_420, qt, restart_interval=ri).jpeg, (blocks, ri)
    finally:
        ctx.close()
