import os, sys, time
sys.path.insert(0, os.path.join(os.getcwd(), "jpeg-encoder_amd", "python"))
import torch
from jpegenc_amd import binding as B, synth_torch, tables as T
dev = torch.device("cuda", 0)
w = h = 8192
rgb = synth_torch.generate("natural", w, h, device=dev)
rgbf = (rgb.to(torch.float32) / 255.0).contiguous()
out = torch.empty(w * h * 2, dtype=torch.uint8, device=dev)
ctx = B.Context(0); ctx.set_profiling(True)
qt = T.quality_tables(90)
for name in ("420", "444", "422", "411"):
    for fmt, buf, stride in ((B.RGB8, rgb, w * 3), (B.RGBF32, rgbf, w * 12)):
        prm = B.make_params(w, h, T.SCHEMES[name], qt)
        for _ in range(3):
            ctx.encode(buf.data_ptr(), fmt, stride, prm, out.data_ptr(), out.numel())
        t0 = time.perf_counter()
        for _ in range(10):
            ctx.encode(buf.data_ptr(), fmt, stride, prm, out.data_ptr(), out.numel())
        dt = (time.perf_counter() - t0) / 10
        print(name, "u8" if fmt == B.RGB8 else "f32", "%.3f ms" % (dt * 1e3), [round(float(x), 3) for x in ctx.timings()], flush=True)
