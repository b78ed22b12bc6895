"""Where a sharded step spends its time on each rank: analyze, all-reduce, finish (host clock).
Run under scripts/rehearse_ranks.sh-like environment (RANK, WORLD_SIZE, MASTER_*), gloo or nccl."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "jpeg-encoder_amd", "python"))
import torch, torch.distributed as dist
from jpegenc_amd import binding as B, sharding, synth_torch, tables as T

world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
local = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
dist.init_process_group(backend, rank=rank, world_size=world)
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
w = h = int(os.environ.get("SIDE", "16384"))
ctx = B.Context(local)
enc = sharding.HipBandEncoder(ctx, B, w, h, T.SCHEMES["420"], T.quality_tables(90), world, rank)
y0, y1 = enc.pixel_rows
rgb = synth_torch.generate("natural", w, h, device=dev, y0=y0, y1=y1)
hist = torch.zeros((4, 256), dtype=torch.int64, device=dev)
out = torch.empty((y1 - y0 + 16) * w + (1 << 20), dtype=torch.uint8, device=dev)
ext = torch.cuda.ExternalStream(ctx.stream, device=dev)
torch.cuda.synchronize()
for i in range(6):
    dist.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    ctx.analyze_band(rgb.data_ptr(), B.RGB8, w * 3, enc.params, hist.data_ptr())
    t1 = time.perf_counter()
    cur = torch.cuda.current_stream(); cur.wait_stream(ext)
    if backend == "nccl":
        dist.all_reduce(hist)
    else:
        hh = hist.cpu(); dist.all_reduce(hh); hist.copy_(hh)
    ext.wait_stream(cur)
    t2 = time.perf_counter()
    n = ctx.finish_band(hist.data_ptr(), out.data_ptr(), out.numel())
    t3 = time.perf_counter()
    print("rank %d step %d: analyze %.3f ms, all-reduce %.3f ms, finish %.3f ms (%d bytes)"
          % (rank, i, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, n), flush=True)
ctx.close()
dist.destroy_process_group()
