"""Sequential single-image encodes from several host threads, one context (stream) each: how
much of the per-image latency (host waits, launch gaps) do concurrent contexts hide?"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "jpeg-encoder_amd", "python"))
import torch  # noqa: E402

from jpegenc_amd import binding as B, synth_torch, tables as T  # noqa: E402

dev = torch.device("cuda", 0)
for (w, h, sname) in [(1920, 1080, "420"), (3840, 2160, "444")]:
    scheme, qt = T.SCHEMES[sname], T.quality_tables(90)
    prm = B.make_params(w, h, scheme, qt)
    rgb = synth_torch.generate("natural", w, h, device=dev)
    for nthreads in (1, 2, 4, 8):
        ctxs = [B.Context(0) for _ in range(nthreads)]
        outs = [torch.empty(w * h * 3, dtype=torch.uint8, device=dev) for _ in range(nthreads)]
        torch.cuda.synchronize()
        per = 200

        def work(i):
            for _ in range(per):
                ctxs[i].encode(rgb.data_ptr(), B.RGB8, w * 3, prm, outs[i].data_ptr(), outs[i].numel())

        for i in range(nthreads):
            ctxs[i].encode(rgb.data_ptr(), B.RGB8, w * 3, prm, outs[i].data_ptr(), outs[i].numel())
        ts = [threading.Thread(target=work, args=(i,)) for i in range(nthreads)]
        t0 = time.perf_counter()
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        dt = time.perf_counter() - t0
        print("%dx%d %s, %d contexts: %.0f images/s (%.3f ms per image overall)" %
              (w, h, sname, nthreads, nthreads * per / dt, dt / (nthreads * per) * 1e3), flush=True)
        for c in ctxs:
            c.close()
