"""PCIe-inclusive rate of jpeg_hip_encode_batch_host: frames and files in pageable host memory."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "jpeg-encoder_amd", "python"))
import numpy as np  # noqa: E402

from jpegenc_amd import binding as B, synth, tables as T  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
w, h = 1920, 1080
base = [synth.generate("natural", w, h, seed=k) for k in range(4)]
frames = np.stack([base[k % 4] for k in range(n)])
prm = B.make_params(w, h, T.SCHEME_420, T.quality_tables(90))
ctx = B.Context(0)
ctx.encode_batch_host(frames[:8], prm, 1 << 20)
for rep in range(3):
    t0 = time.perf_counter()
    files = ctx.encode_batch_host(frames, prm, 1 << 20)
    dt = time.perf_counter() - t0
    print("%d frames in %.1f ms: %.0f frames/s, %.0f MP/s, %.1f GB/s of pixels" %
          (n, dt * 1e3, n / dt, n * w * h / 1e6 / dt, frames.nbytes / dt / 1e9))
ctx.close()
