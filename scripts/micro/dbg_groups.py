import os, sys
sys.path.insert(0, "jpeg-encoder_amd/python"); sys.path.insert(0, "oracle")
os.environ["JPEGENC_GROUP_MIN_BLOCKS"] = "0"
import numpy as np
import oracle as O
from jpegenc_amd import binding as B, synth, tables as T
for tile in ("256", "64", "16"):
    os.environ["JPEGENC_GROUP_TILE"] = tile
    ctx = B.Context(0)
    for (w, h, kind, tname) in [(640, 360, "natural", "annexk"), (333, 77, "noise", "annexk"), (250, 130, "noise", "ones")]:
        img = synth.generate(kind, w, h, seed=w + h)
        qt = T.NAMED_TABLES[tname]
        enc = O.encode(img, T.SCHEMES["444"], qt)
        prm = B.make_params(w, h, T.SCHEMES["444"], qt)
        try:
            got = ctx.encode_host(img, prm)
            print(tile, w, h, kind, tname, len(got), len(enc.jpeg), got == enc.jpeg, "bits/block", 8 * enc.scan_len / sum(c.shape[0] for c in enc.coef))
        except Exception as e:
            print(tile, w, h, kind, tname, "error", e)
    ctx.close()
