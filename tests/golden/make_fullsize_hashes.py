"""Writes fullsize_hashes.json: length and SHA-256 of the CPU oracle's file for the benchmark-size
workloads (too large to keep as files, and too large for the GPU box to re-encode on its CPU inside
a test).  The GPU tests compare the HIP encoder's file with these without running the oracle.

For every workload the scalar restatement (oracle/jpeg_oracle.cpp) and the AVX2 + OpenMP one
(oracle/jpeg_oracle_avx2.cpp) are both run and must agree on file, coefficients and histograms
before anything is written.  CPU only; takes a few minutes and about 12 GB of memory.

    python tests/golden/make_fullsize_hashes.py
"""
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "jpeg-encoder_amd", "python"))
import numpy as np  # noqa: E402

import oracle as O  # noqa: E402
from jpegenc_amd import synth, tables as T  # noqa: E402

# name: (width, height, scheme, tables, image kind, seed, restart interval in MCUs)
CASES = {
    # the driver's hard-coded case (JPEG-Encoder-SIMD.cpp:401-402: Scheme420, Annex-K tables) at 1080p
    "1080p_420_annexk_natural": (1920, 1080, "420", "annexk", "natural", 1, 0),
    "1080p_420_q90_natural": (1920, 1080, "420", "q90", "natural", 1, 0),
    "4k_444_q90_natural": (3840, 2160, "444", "q90", "natural", 1, 0),
    # one restart interval per MCU row: what the ranks of a sharded encode write between them
    "16k_420_q90_natural_ri1024": (16384, 16384, "420", "q90", "natural", 1, 1024),
    "8k_444_q100_noise": (8192, 8192, "444", "q100", "noise", 42, 0),
    "8k_420_q100_noise": (8192, 8192, "420", "q100", "noise", 42, 0),
    # middle of the density range (150 to 300 bits per block)
    "4k_square_444_q75_noise": (4096, 4096, "444", "q75", "noise", 42, 0),
    "4k_square_420_q75_noise": (4096, 4096, "420", "q75", "noise", 42, 0),
    # smooth content with a narrow band of noise: the mean block is short, a few tiles are dense
    "4k_square_420_q90_mixed": (4096, 4096, "420", "q90", "mixed", 0, 0),
    # frames of the batch workload: frame k has seed 1000 + k
    "1080p_420_q90_frame0": (1920, 1080, "420", "q90", "natural", 1000, 0),
    "1080p_420_q90_frame1": (1920, 1080, "420", "q90", "natural", 1001, 0),
    "1080p_420_q90_frame255": (1920, 1080, "420", "q90", "natural", 1255, 0),
    "1080p_420_q90_frame9999": (1920, 1080, "420", "q90", "natural", 10999, 0),
}


def tables_named(name):
    if name == "annexk":
        return T.NAMED_TABLES["annexk"]
    return T.quality_tables(int(name[1:]))


def entry(name):
    w, h, scheme, tn, kind, seed, ri = CASES[name]
    img = synth.generate(kind, w, h, seed=seed if kind != "mixed" else None)
    qt = tables_named(tn)
    t0 = time.time()
    fast = O.encode(img, T.SCHEMES[scheme], qt, restart_interval=ri, variant="avx2")
    t1 = time.time()
    slow = O.encode(img, T.SCHEMES[scheme], qt, restart_interval=ri, variant="scalar")
    t2 = time.time()
    assert fast.jpeg == slow.jpeg, name + ": the two restatements disagree on the file"
    for c in range(3):
        assert np.array_equal(fast.coef[c], slow.coef[c]), name + ": coefficients differ"
    assert np.array_equal(fast.hist, slow.hist), name + ": histograms differ"
    coef = hashlib.sha256()
    for c in range(3):
        coef.update(np.ascontiguousarray(fast.coef[c]).tobytes())
    print("%-30s %10d bytes  avx2 %.2f s  scalar %.2f s" % (name, len(fast.jpeg), t1 - t0, t2 - t1), flush=True)
    return {
        "width": w, "height": h, "scheme": scheme, "tables": tn, "kind": kind, "seed": seed, "restart_interval": ri,
        "length": len(fast.jpeg),
        "sha256": hashlib.sha256(fast.jpeg).hexdigest(),
        "coef_sha256": coef.hexdigest(),          # channels Y, Cb, Cr, each in its MCU order, zig-zag
        "hist_sha256": hashlib.sha256(np.ascontiguousarray(fast.hist).tobytes()).hexdigest(),
        "scan_offset": int(fast.scan_offset),
    }


def main():
    out = {name: entry(name) for name in CASES}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fullsize_hashes.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote %d entries to %s" % (len(out), path))


if __name__ == "__main__":
    main()
