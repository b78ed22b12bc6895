"""Transform time for caller-owned byte pixels whose rows do not start on 4-byte boundaries."""
import os, sys
sys.path.insert(0, os.path.join(os.getcwd(), "jpeg-encoder_amd", "python"))
import torch
from jpegenc_amd import binding as B, synth_torch, tables as T
dev = torch.device("cuda", 0)
ctx = B.Context(0); ctx.set_profiling(True)
qt = T.quality_tables(90)
for w, h in ((8190, 8192), (8192, 8192)):
    rgb = synth_torch.generate("natural", w, h, device=dev)
    out = torch.empty(w * h * 2, dtype=torch.uint8, device=dev)
    prm = B.make_params(w, h, T.SCHEMES["420"], qt)
    import time
    for _ in range(20):
        ctx.encode(rgb.data_ptr(), B.RGB8, w * 3, prm, out.data_ptr(), out.numel())
    t0 = time.perf_counter()
    for _ in range(20):
        ctx.encode(rgb.data_ptr(), B.RGB8, w * 3, prm, out.data_ptr(), out.numel())
    dt = (time.perf_counter() - t0) / 20
    print(w, "stride", w * 3, "%.3f ms per image" % (dt * 1e3), [round(float(x), 3) for x in ctx.timings()], flush=True)
