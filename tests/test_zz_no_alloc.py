"""jpeg_hip_encode_static (the entry point that allocates no device memory) against
jpeg_hip_encode, through the C ABI (needs an MI355X).

Collected last on purpose: the file name sorts behind every other test file."""
import numpy as np
import pytest
import torch

from jpegenc_amd import binding as B, synth, tables as T

pytestmark = pytest.mark.gpu


def _encode_both(ctx_static, ctx_plain, img, scheme, qt, **kw):
    h, w = img.shape[:2]
    prm = B.make_params(w, h, T.SCHEMES[scheme], qt, **kw)
    dev = torch.device("cuda", 0)
    rgb = torch.from_numpy(img).to(dev)
    cap = int(B.compute_layout(prm).max_jpeg_bytes)
    cap = min(cap, 3 * w * h + (1 << 20))
    out_a = torch.full((cap,), 0xAA, dtype=torch.uint8, device=dev)
    out_b = torch.full((cap,), 0x55, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    n_a = ctx_static.encode_static(rgb.data_ptr(), B.RGB8, w * 3, prm, out_a.data_ptr(), cap)
    n_b = ctx_plain.encode(rgb.data_ptr(), B.RGB8, w * 3, prm, out_b.data_ptr(), cap)
    torch.cuda.synchronize()
    return out_a[:n_a].cpu().numpy().tobytes(), out_b[:n_b].cpu().numpy().tobytes()


def test_static_workspace_matches_encode_twice_in_a_row():
    assert B.lib().jpeg_hip_static_workspace_bytes() == 32 << 20
    a, b = B.Context(0), B.Context(0)
    try:
        qt = T.quality_tables(90)
        # Two calls in a row on the same array, the second with a smaller input: whatever the
        # first left behind (coefficients, tile states, stream bytes, counters) must not show.
        big = synth.generate("noise", 1920, 1080, seed=7)
        got, want = _encode_both(a, b, big, "420", qt)
        assert got == want and got[:2] == b"\xff\xd8" and got[-2:] == b"\xff\xd9"
        small = synth.generate("natural", 203, 117, seed=8)
        got, want = _encode_both(a, b, small, "420", qt)
        assert got == want
        # Nothing is allocated: the device's free memory is the same before and after a call
        # (inputs and outputs made beforehand, the plain context left alone).
        prm = B.make_params(203, 117, T.SCHEMES["420"], qt)
        dev = torch.device("cuda", 0)
        rgb = torch.from_numpy(small).to(dev)
        out = torch.zeros(1 << 20, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        free_before = torch.cuda.mem_get_info(0)[0]
        for _ in range(3):
            n = a.encode_static(rgb.data_ptr(), B.RGB8, 203 * 3, prm, out.data_ptr(), out.numel())
        assert torch.cuda.mem_get_info(0)[0] == free_before
        assert out[:n].cpu().numpy().tobytes() == want
        # Other shapes of the path: 4:4:4, restart intervals, a second context on the same
        # array, the standard tables, an odd width (rows off 4-byte boundaries).
        got, want = _encode_both(b, a, synth.generate("natural", 640, 360, seed=9), "444", qt, restart_interval=5)
        assert got == want
        got, want = _encode_both(a, b, synth.generate("natural", 333, 77, seed=10), "422", qt, standard_huffman=True)
        assert got == want
        # The context still serves the ordinary entry points afterwards.
        img = synth.generate("natural", 64, 48, seed=11)
        prm = B.make_params(64, 48, T.SCHEMES["420"], qt)
        assert a.encode_host(img, prm) == b.encode_host(img, prm)
    finally:
        a.close()
        b.close()


def test_static_workspace_serves_threads_one_at_a_time():
    """Two threads, two contexts, one GPU and so one array: the calls take turns inside the library."""
    import threading
    qt = T.quality_tables(90)
    imgs = [synth.generate("natural", 640, 360, seed=21), synth.generate("noise", 500, 300, seed=22)]
    plain = B.Context(0)
    want = [plain.encode_host(im, B.make_params(im.shape[1], im.shape[0], T.SCHEMES["420"], qt)) for im in imgs]
    plain.close()
    dev = torch.device("cuda", 0)
    bad = []

    def work(k):
        ctx = B.Context(0)
        try:
            im = imgs[k]
            h, w = im.shape[:2]
            prm = B.make_params(w, h, T.SCHEMES["420"], qt)
            rgb = torch.from_numpy(im).to(dev)
            out = torch.zeros(1 << 20, dtype=torch.uint8, device=dev)
            torch.cuda.synchronize()
            for _ in range(40):
                n = ctx.encode_static(rgb.data_ptr(), B.RGB8, w * 3, prm, out.data_ptr(), out.numel())
                if out[:n].cpu().numpy().tobytes() != want[k]:
                    bad.append(k)
                    break
        except Exception as e:      # reported through the list: an assertion in a thread is lost
            bad.append((k, repr(e)))
        finally:
            ctx.close()

    threads = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(120)
    assert not bad, bad


def test_static_workspace_rejects_what_does_not_fit():
    ctx = B.Context(0)
    try:
        qt = T.quality_tables(90)
        w = h = 4096   # 393216 blocks of 4:2:0: 48 MiB of coefficients alone
        prm = B.make_params(w, h, T.SCHEMES["420"], qt)
        dev = torch.device("cuda", 0)
        rgb = torch.zeros(w * h * 3, dtype=torch.uint8, device=dev)
        out = torch.zeros(1 << 20, dtype=torch.uint8, device=dev)
        with pytest.raises(B.JpegHipError) as e:
            ctx.encode_static(rgb.data_ptr(), B.RGB8, w * 3, prm, out.data_ptr(), out.numel())
        assert e.value.status == B.ERR_BUFFER_TOO_SMALL and "static workspace" in str(e.value)
        # with the standard tables the array has to hold the worst-case stream: 1080p 4:4:4 does not fit
        prm = B.make_params(1920, 1080, T.SCHEMES["444"], qt, standard_huffman=True)
        with pytest.raises(B.JpegHipError) as e:
            ctx.encode_static(rgb.data_ptr(), B.RGB8, 1920 * 3, prm, out.data_ptr(), out.numel())
        assert e.value.status == B.ERR_BUFFER_TOO_SMALL and "worst-case" in str(e.value)
    finally:
        ctx.close()
