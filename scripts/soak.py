"""Encodes the same inputs over and over and compares every output with the first one: a race in
the tile chains, the deferred placement or the counters would show as a different file sooner or later.
    python3 scripts/soak.py [seconds per workload]   (GPU box)"""
import hashlib
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "jpeg-encoder_amd", "python"))
from jpegenc_amd import binding as B, tables as T

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 15.0
dev = torch.device("cuda", 0)
cases = [("16k natural 420 rst", 16384, 16384, "420", 90, "natural", 1024, 1),
         ("16k natural 420", 16384, 16384, "420", 90, "natural", 0, 1),
         ("4k natural 444", 3840, 2160, "444", 90, "natural", 0, 1),
         ("2.5k natural 420 (look-back, groups)", 2560, 2048, "420", 90, "natural", 0, 1),
         ("1080p natural 420", 1920, 1080, "420", 90, "natural", 0, 1),
         ("1080p natural 420 rst 7", 1920, 1080, "420", 90, "natural", 7, 1),
         ("8k noise 444 q100", 8192, 8192, "444", 100, "noise", 0, 1),
         ("4k noise 420 q100", 4096, 4096, "420", 100, "noise", 64, 1),
         ("8k noise 420 q100 rst 3", 8192, 8192, "420", 100, "noise", 3, 1),
         ("4k noise 444 q100 rst 5 (the smallest picture with tile numbers by XCD)", 4096, 4096, "444", 100, "noise", 5, 1),
         ("batch 16 x 1080p noise q100 444", 1920, 1080, "444", 100, "noise", 0, 16),
         ("batch 64 x 1080p", 1920, 1080, "420", 90, "natural", 0, 64),
         ("batch 12 x 640x480 444", 640, 480, "444", 75, "natural", 0, 12)]
bad = 0
for name, w, h, sname, q, kind, ri, frames in cases:
    ctx = B.Context(0)
    prm = B.make_params(w, h, T.SCHEMES[sname], T.quality_tables(q), restart_interval=ri)
    stride, image = w * 3, w * h * 3
    rgb = torch.empty(frames * image, dtype=torch.uint8, device=dev)
    ctx.synthetic_images(kind, w, 0, h, 7, rgb.data_ptr(), stride, image, frames)
    cap = int(w * h * (3.2 if kind == "noise" else 1.0)) + (1 << 20)
    out = torch.empty(cap * frames, dtype=torch.uint8, device=dev)

    def once():
        out.zero_()
        if frames > 1:
            lens = ctx.encode_batch(rgb.data_ptr(), B.RGB8, stride, image, frames, prm, out.data_ptr(), cap)
            hs = hashlib.sha256()
            host = out.cpu().numpy()
            for k, n in enumerate(lens):
                hs.update(host[k * cap:k * cap + int(n)].tobytes())
            return hs.hexdigest()
        n = ctx.encode(rgb.data_ptr(), B.RGB8, stride, prm, out.data_ptr(), cap)
        return hashlib.sha256(out[:n].cpu().numpy().tobytes()).hexdigest()

    first = once()
    runs, t0 = 1, time.time()
    while time.time() - t0 < budget:
        if once() != first:
            bad += 1
            print("MISMATCH in run %d of %s" % (runs, name), flush=True)
        runs += 1
    print("%-40s %5d runs, all equal: %s" % (name, runs, bad == 0), flush=True)
    ctx.close()
sys.exit(1 if bad else 0)
