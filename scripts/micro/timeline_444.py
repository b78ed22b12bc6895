import sys, os, ctypes as C, numpy as np, torch
sys.path.insert(0, "jpeg-encoder_amd/python")
from jpegenc_amd import binding as B, tables as T
w, h = 3840, 2160
ctx = B.Context(0)
prm = B.make_params(w, h, T.SCHEMES["444"], T.quality_tables(90))
rgb = torch.empty(w * h * 3, dtype=torch.uint8, device="cuda")
ctx.synthetic_images("natural", w, 0, h, 1, rgb.data_ptr(), w * 3)
out = torch.empty(w * h * 2, dtype=torch.uint8, device="cuda")
for _ in range(5):
    ctx.encode(rgb.data_ptr(), B.RGB8, w * 3, prm, out.data_ptr(), out.numel())
torch.cuda.synchronize()
hist = torch.zeros((4, 256), dtype=torch.int64, device="cuda")
ctx.analyze_band(rgb.data_ptr(), B.RGB8, w * 3, prm, hist.data_ptr())
torch.cuda.synchronize()
tl = np.zeros((4096, 8), np.uint64)
L = B.lib()
print("rc", L.jpeg_hip_debug_timeline(tl.ctypes.data_as(C.c_void_p)))
tl = tl[:2025].astype(np.int64)
real = tl[:, 7]                      # constant 100 MHz counter, common to the chip
print("dispatch spread of the workgroups (100 MHz ticks): first %d last %d -> %.2f us" % (real.min(), real.max(), (real.max() - real.min()) / 100.0))
names = ["start", "loads issued", "converted", "Y done", "Cb done", "Cr issued", "stores drained"]
life = tl[:, 6] - tl[:, 0]
print("wave life in shader cycles: median %d p10 %d p90 %d max %d" % (np.median(life), np.percentile(life, 10), np.percentile(life, 90), life.max()))
d = np.diff(tl[:, :7], axis=1)
for k in range(6):
    c = d[:, k]
    print("phase %-14s median %7d  p10 %7d p90 %7d" % (names[k + 1], np.median(c), np.percentile(c, 10), np.percentile(c, 90)))
# start order: ticks since the first workgroup, by deciles of blockIdx
rel = (real - real.min()) / 100.0
for q in (0, 10, 25, 50, 75, 90, 100):
    print("start of the workgroup at percentile %3d: %.2f us" % (q, np.percentile(rel, q)))

hw = tl[:, 5]
xcc = (hw >> 32) & 0xf
hwid = hw & 0xffffffff
wave = hwid & 0xf; simd = (hwid >> 4) & 3; cu = (hwid >> 8) & 0xf; sh = (hwid >> 12) & 1; se = (hwid >> 13) & 7
early = rel < 8.0
import collections
print("early workgroups:", int(early.sum()), "of", len(rel))
per_cu = collections.Counter(zip(xcc[early].tolist(), se[early].tolist(), sh[early].tolist(), cu[early].tolist()))
print("CUs seen:", len(per_cu), "early waves per CU histogram:", sorted(collections.Counter(per_cu.values()).items()))
per_simd = collections.Counter(zip(xcc[early].tolist(), se[early].tolist(), sh[early].tolist(), cu[early].tolist(), simd[early].tolist()))
print("SIMDs seen:", len(per_simd), "early waves per SIMD histogram:", sorted(collections.Counter(per_simd.values()).items()))
print("per XCC early:", sorted(collections.Counter(xcc[early].tolist()).items()))
print("per XCC all:", sorted(collections.Counter(xcc.tolist()).items()))
print("wave slot ids used:", sorted(collections.Counter(wave.tolist()).items()))
