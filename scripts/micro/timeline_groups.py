"""Phase times of the tiles of the bit writer on groups (needs a library built with the
timeline instrumentation, which is not part of the product)."""
import sys, ctypes as C, numpy as np, torch
sys.path.insert(0, "jpeg-encoder_amd/python")
from jpegenc_amd import binding as B, tables as T
w = h = 16384
ctx = B.Context(0)
lay0 = B.compute_layout(B.make_params(w, h, T.SCHEMES["420"], T.quality_tables(90)))
prm = B.make_params(w, h, T.SCHEMES["420"], T.quality_tables(90), restart_interval=lay0.mcus_per_row)
rgb = torch.empty(w * h * 3, dtype=torch.uint8, device="cuda")
ctx.synthetic_images("natural", w, 0, h, 1, rgb.data_ptr(), w * 3)
out = torch.empty(w * h, dtype=torch.uint8, device="cuda")
for _ in range(3):
    n = ctx.encode(rgb.data_ptr(), B.RGB8, w * 3, prm, out.data_ptr(), out.numel())
torch.cuda.synchronize()
tl = np.zeros((32768, 8), np.uint64)
print("rc", B.lib().jpeg_hip_debug_timeline(tl.ctypes.data_as(C.c_void_p)), "bytes", n)
used = tl[:, 0] != 0
tl = tl[used].astype(np.int64)
print("tiles", len(tl))
real = (tl[:, 7] - tl[:, 7].min()) / 100.0
print("kernel span by the 100 MHz clock: %.1f us (starts only)" % real.max())
names = ["tables", "first load", "first place", "first flush", "rest"]
d = np.diff(tl[:, :6], axis=1)
life = tl[:, 5] - tl[:, 0]
print("tile life (shader cycles): median %d p10 %d p90 %d" % (np.median(life), np.percentile(life, 10), np.percentile(life, 90)))
for k, nme in enumerate(names):
    c = d[:, k]
    print("%-10s median %7d  p10 %7d  p90 %7d  share of life %.1f%%" % (nme, np.median(c), np.percentile(c, 10), np.percentile(c, 90), 100.0 * c.sum() / life.sum()))
items = tl[:, 6]
print("records per tile: median %d p10 %d p90 %d max %d" % (np.median(items), np.percentile(items, 10), np.percentile(items, 90), items.max()))
for q in (0, 10, 25, 50, 75, 90, 100):
    print("start of the tile at percentile %3d: %.1f us" % (q, np.percentile(real, q)))
