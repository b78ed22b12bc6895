"""What the stage timers cost a step: the 16k image and a 1080p frame encoded with and without them.
    python3 scripts/profiling_cost.py   (GPU box)"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "jpeg-encoder_amd", "python"))
from jpegenc_amd import binding as B, tables as T

dev = torch.device("cuda", 0)
for (w, h, ri) in ((16384, 16384, 1024), (1920, 1080, 0)):
    qt = T.quality_tables(90)
    prm = B.make_params(w, h, T.SCHEME_420, qt, restart_interval=ri)
    ctx = B.Context(0)
    rgb = torch.empty(w * h * 3, dtype=torch.uint8, device=dev)
    ctx.synthetic_images("natural", w, 0, h, 1, rgb.data_ptr(), w * 3)
    cap = w * h + (1 << 20)
    out = torch.empty(cap, dtype=torch.uint8, device=dev)
    for mode in ("off", "on", "on+read", "off", "on", "on+read"):
        ctx.set_profiling(mode != "off")
        for _ in range(10):
            ctx.encode(rgb.data_ptr(), B.RGB8, w * 3, prm, out.data_ptr(), cap)
        torch.cuda.synchronize()
        n = 200
        t0 = time.perf_counter()
        for _ in range(n):
            ctx.encode(rgb.data_ptr(), B.RGB8, w * 3, prm, out.data_ptr(), cap)
            if mode == "on+read":
                ctx.timings()
        torch.cuda.synchronize()
        print("%dx%d timers %-8s %.4f ms per image" % (w, h, mode, (time.perf_counter() - t0) / n * 1e3))
    ctx.close()
