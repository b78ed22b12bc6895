#!/usr/bin/env python3
"""Benchmark of the MI355X JPEG encode hot path.

One step = one full encode of the workload image (RGB8 resident in HBM -> JFIF bytes
resident in HBM): fused transform, histograms, [histogram all-reduce], Huffman tables,
bit writer, byte stuffing.  With N > 1 ranks the image is sharded by restart intervals,
one band of MCU rows per GPU (strong scaling), and the only collective is the 8 KiB
histogram all-reduce.

  python bench.py --gpus N --steps K --warmup W

Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "jpeg-encoder_amd", "python"))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from jpegenc_amd import binding as B, sharding, synth, synth_torch, tables as T  # noqa: E402

WORKLOADS = {
    # name: (width, height, scheme, quality, image kind, description)
    "16k_420_q90": (16384, 16384, "420", 90, "natural",
                    "16384x16384 synthetic natural-like RGB8, 4:2:0, q=90, restart interval = 1 MCU row, "
                    "sharded by MCU rows over the ranks"),
    "4k_444_q90": (3840, 2160, "444", 90, "natural", "3840x2160 synthetic natural-like RGB8, 4:4:4, q=90"),
    "8k_noise_q100_444": (8192, 8192, "444", 100, "noise", "8192x8192 uniform noise, 4:4:4, q=100 (all-ones tables)"),
    "8k_noise_q100_420": (8192, 8192, "420", 100, "noise", "8192x8192 uniform noise, 4:2:0, q=100 (all-ones tables)"),
    "1080p_420_q90": (1920, 1080, "420", 90, "natural", "1920x1080 synthetic natural-like RGB8, 4:2:0, q=90"),
    # the middle of the density range (about 200 bits per block), between photographic content and the noise at q=100
    "4k2_noise_q75_444": (4096, 4096, "444", 75, "noise", "4096x4096 uniform noise, 4:4:4, q=75"),
    "4k2_noise_q75_420": (4096, 4096, "420", 75, "noise", "4096x4096 uniform noise, 4:2:0, q=75"),
    "4k2_mixed_q90_420": (4096, 4096, "420", 90, "mixed", "4096x4096 natural-like with a band of uniform noise (a 128th of the rows), 4:2:0, q=90"),
    "4k2_noise_q50_420": (4096, 4096, "420", 50, "noise", "4096x4096 uniform noise, 4:2:0, q=50"),
    "4k2_noise_q90_420": (4096, 4096, "420", 90, "noise", "4096x4096 uniform noise, 4:2:0, q=90"),
    "batch_1080p_420_q90": (1920, 1080, "420", 90, "natural",
                            "batch of 1920x1080 synthetic natural-like RGB8 frames (different seeds), 4:2:0, q=90, "
                            "one call per step, frames split evenly over the ranks (throughput mode)"),
}
HBM_PEAK_GBPS = 8000.0


def algorithmic_bytes(lay):
    """Fused transform: RGB8 read once + int16 coefficients written once (band)."""
    px = lay.padded_width * lay.band_mcu_rows * lay.mcu_height
    return 3 * px + 2 * 64 * int(lay.total_blocks)


def measured_traffic(workload, world, frames):
    """HBM bytes the fused transform moves per step, from the committed PMC passes (N=1 and the
    default frame count only): FETCH_SIZE doubled for gfx950 plus WRITE_SIZE, per launch, times the
    launches of a step (a batch pass runs the transform in two launches)."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    if world != 1 or not os.path.exists(path) or (workload.startswith("batch_") and frames != 256):
        return None
    entry = json.load(open(path)).get(workload)
    return entry["traffic_bytes_per_launch"] * entry.get("launches_per_step", 1) if entry else None


def cpu_baseline(rgb, width, height, frames, scheme, qt, restart_interval, gpu_file):
    """The AVX2 + OpenMP restatement of the reference path (oracle/jpeg_oracle_avx2.cpp) timed on
    the host cores over the images the GPU encoded: the whole image, or the first frames of a
    batch.  Its first output is compared with the GPU's file on the way (checker use only)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle
    threads = int(oracle.lib_avx2().orcv_threads())
    host = rgb.cpu().numpy()
    per_image = width * height * 3
    sample = min(max(frames, 1), 64)
    best, runs, same = None, 0, None
    t_end = time.time() + 20.0
    while runs < 2 or (time.time() < t_end and runs < 5):
        total = 0.0
        for i in range(sample):
            img = host[i * per_image:(i + 1) * per_image].reshape(height, width, 3)
            tm = []
            enc = oracle.encode(img, scheme, qt, restart_interval=restart_interval, variant="avx2", timing=tm)
            total += tm[0]
            if runs == 0 and i == 0:
                same = bool(enc.jpeg == gpu_file(0))
            del enc
        best = total if best is None else min(best, total)
        runs += 1
    what = ("the whole %dx%d workload image" % (width, height)) if not frames else \
        ("the first %d of the %d frames, one after the other" % (sample, frames))
    return {"value": round(sample * width * height / 1e6 / best, 2), "unit": "MP/s", "cores": threads,
            "kind": "port", "implementation": "AVX2 + OpenMP restatement of the reference's path (oracle/jpeg_oracle_avx2.cpp)",
            "sample": "%s, same scheme, tables and restart interval; encode call only (pixels in host memory -> file "
                      "in host memory), best of %d runs, OpenMP threads=%d" % (what, runs, threads),
            "file_equals_gpu_file": same}


def stage_rooflines(lay, images, out_bytes, kern_ms):
    """Achieved GB/s and fraction of the HBM peak per stage and for the three stages together.
    Bytes a stage has to move: transform 3 B/pixel in + 2 B/coefficient out; histogram
    2 B/coefficient in; entropy 2 B/coefficient in + the stream written, read and written
    again stuffed (3 x the output size)."""
    coef = 128 * int(lay.total_blocks) * images
    need = {"transform": algorithmic_bytes(lay) * images, "histogram": coef, "entropy": coef + 3 * out_bytes}
    ms = {"transform": float(kern_ms[0]), "histogram": float(kern_ms[1]), "entropy": float(kern_ms[2])}
    res = {}
    for k in need:
        if ms[k] > 0:
            gbs = need[k] / (ms[k] * 1e-3) / 1e9
            res[k] = {"bytes": int(need[k]), "ms": round(ms[k], 4), "achieved": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBPS, 4)}
    total_ms = sum(ms[k] for k in res)
    if total_ms > 0:
        gbs = sum(need[k] for k in res) / (total_ms * 1e-3) / 1e9
        res["all"] = {"bytes": int(sum(need[k] for k in res)), "ms": round(total_ms, 4), "achieved": round(gbs, 1),
                      "frac": round(gbs / HBM_PEAK_GBPS, 4)}
    return res


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher around it: this process starts the N ranks
    itself (one per GPU, environment as torch.distributed.run would set it), never touches a
    GPU, relays rank 0's JSON line and fails if any rank fails."""
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    have = torch.cuda.device_count()   # counting devices does not initialise the GPU
    if backend == "nccl" and have < n:
        print("bench.py: --gpus %d asked for but %d GPU(s) are visible; not running a smaller job under that name"
              % (n, have), file=sys.stderr)
        return 2
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # All ranks are watched: when one dies (before or inside a collective the others would wait
    # in for ever) or the job runs out of time, the rest are ended -- first asked, then killed.
    limit = float(os.environ.get("BENCH_RANKS_TIMEOUT", "900"))
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    t_end = time.time() + limit
    timed_out = False
    while any(p.poll() is None for p in procs):
        if any(p.poll() not in (None, 0) for p in procs) or time.time() > t_end:
            timed_out = time.time() > t_end
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            t_kill = time.time() + 10.0
            while any(p.poll() is None for p in procs) and time.time() < t_kill:
                time.sleep(0.1)
            for p in procs:
                if p.poll() is None:
                    p.kill()
            break
        time.sleep(0.05)
    codes = [p.wait() for p in procs]
    reader.join(5.0)
    out = b"".join(c for c in chunks if c)
    if timed_out:
        print("bench.py: the ranks did not finish within %.0f s and were ended" % limit, file=sys.stderr)
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    if bad:
        print("bench.py: ranks failed (rank, exit code): %s" % bad, file=sys.stderr)
        return 1
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="16k_420_q90", choices=sorted(WORKLOADS))
    ap.add_argument("--frames", type=int, default=256, help="frames per GPU and step (batch workload)")
    ap.add_argument("--standard-huffman", action="store_true",
                    help="Annex K tables instead of per-image optimised ones: no histogram pass, no host step")
    ap.add_argument("--band-of", type=int, default=0,
                    help="diagnostic: encode only rank 0's band of an image sharded over this many ranks")
    ap.add_argument("--pipeline-depth", type=int, default=1,
                    help="images in flight per rank in the timed region, 1 to 3 (sharded workloads)")
    ap.add_argument("--pipelined-steps", type=int, default=50,
                    help="steps of the extra measurement with three images in flight, reported as \"pipelined\" (0: skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--analyze-only", action="store_true",
                    help="diagnostic: stop after the transform and the histogram pass (no file is written)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL ("nccl") with one GPU per rank; BENCH_DIST_BACKEND=gloo lets several ranks share
        # one GPU when rehearsing the multi-rank path on a single-GPU box.
        backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
            local_rank = local_rank % max(torch.cuda.device_count(), 1)
    if args.gpus != world:
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE %d: the launcher and the arguments disagree" % (args.gpus, world),
                  file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # The histogram all-reduce runs inside the library (jpeg_hip_encode_sharded /
    # jpeg_hip_allreduce_histograms) on RCCL communicators the library creates: rank 0 makes the
    # ids, torch.distributed only carries them to the other ranks (and the barriers and the
    # timing reduction of this harness).  One communicator per image in flight.
    use_rccl = world > 1 and dist.get_backend() == "nccl"

    rccl_error = None

    def make_comm():
        # Byte 128 says whether rank 0 got an id at all, so that no rank waits for a broadcast
        # that never comes.
        msg = torch.zeros(129, dtype=torch.uint8, device=dev)
        if rank == 0:
            try:
                msg[:128] = torch.from_numpy(B.comm_unique_id()).to(dev)
                msg[128] = 1
            except Exception as e:      # reported below, through the fall-back
                print("bench.py: no RCCL id: %s: %s" % (type(e).__name__, e), file=sys.stderr)
        dist.broadcast(msg, 0)
        host = msg.cpu().numpy()
        if host[128] != 1:
            raise RuntimeError("rank 0 could not create an RCCL id")
        return B.Communicator(host[:128], rank, world, local_rank)

    comms = [None]
    if use_rccl:
        # If the library's own communicator cannot be set up on every rank, all ranks fall back
        # together to torch.distributed's all-reduce of the histograms (the same exchange through
        # another RCCL communicator) and the line says so.
        try:
            comms = [make_comm()]
            ok = 1
        except Exception as e:
            rccl_error = "%s: %s" % (type(e).__name__, e)
            ok = 0
        agreed = torch.tensor([ok], dtype=torch.int32, device=dev)
        dist.all_reduce(agreed, op=dist.ReduceOp.MIN)
        if int(agreed.item()) == 0:
            for c in comms:
                if c is not None:
                    c.close()
            comms = [None]
            use_rccl = False
            rccl_error = rccl_error or "another rank could not create its communicator"
    # What the exchange step ran on; with one rank there is none (the image is one band).
    collective = {"backend": "none (one rank, no exchange step)", "world_size": 1,
                  "devices": [torch.cuda.get_device_name(local_rank)]}
    if world > 1:
        names = [None] * world
        dist.all_gather_object(names, "cuda:%d %s" % (local_rank, torch.cuda.get_device_name(local_rank)))
        collective = {"backend": "rccl: ncclAllReduce of uint64[4][256] on the encoder's stream (libjpegenc_hip), "
                                 "torch.distributed backend %s for set-up and timing" % dist.get_backend()
                      if use_rccl else
                      ("torch.distributed all_reduce of uint64[4][256] (backend nccl = RCCL) ordered against the encoder's "
                       "stream; the library's own communicator failed: %s" % rccl_error
                       if dist.get_backend() == "nccl" else "%s through host memory (rehearsal)" % dist.get_backend()),
                      "world_size": dist.get_world_size(), "devices": names}

    width, height, sname, quality, kind, desc = WORKLOADS[args.workload]
    scheme, qt = T.SCHEMES[sname], T.quality_tables(quality)
    ctx = B.Context(local_rank)
    ctx.set_profiling(True)

    batch = args.frames if args.workload.startswith("batch_") else 0
    sharded = not batch and (world > 1 or args.workload == "16k_420_q90")
    if batch:
        params = B.make_params(width, height, scheme, qt)
        lay = B.compute_layout(params)
        y0, y1 = 0, height
    elif sharded:
        enc = sharding.HipBandEncoder(ctx, B, width, height, scheme, qt, args.band_of or world, rank)
        params, lay = enc.params, enc.layout
        y0, y1 = enc.pixel_rows
    else:
        params = B.make_params(width, height, scheme, qt)
        lay = B.compute_layout(params)
        y0, y1 = 0, height
    if args.standard_huffman:
        params.standard_huffman = 1
    stride = width * 3
    if batch:
        image_stride = stride * height
        # The inputs come from the library's own generator, one launch for all frames (the same
        # bytes as synth / synth_torch, which the tests compare it with).
        rgb = torch.empty(batch * image_stride, dtype=torch.uint8, device=dev)
        ctx.synthetic_images(kind, width, 0, height, 1000 + rank * batch, rgb.data_ptr(), stride, image_stride, batch)
    else:
        rgb = torch.empty((y1 - y0) * stride, dtype=torch.uint8, device=dev)
        if kind == "mixed":
            # natural content with a narrow band of uniform noise across the middle (a 128th of the rows):
            # the histograms of the whole image say "photographic", one tile in 128 is as dense as noise
            ctx.synthetic_images("natural", width, y0, y1, 1, rgb.data_ptr(), stride)
            b0, b1 = synth.mixed_band(height)
            n0, n1 = max(y0, b0), min(y1, b1)
            if n1 > n0:
                ctx.synthetic_images("noise", width, n0, n1, 42, rgb.data_ptr() + (n0 - y0) * stride, stride)
        else:
            ctx.synthetic_images(kind, width, y0, y1, 42 if kind == "noise" else 1, rgb.data_ptr(), stride)
    hist = torch.zeros((4, 256), dtype=torch.int64, device=dev)
    # Output capacity: the worst case bound is huge; 1.5 bytes per pixel of the band plus
    # slack covers every workload here (noise at q=100 needs about 1.1 B/px in 4:2:0 and
    # 2.9 B/px in 4:4:4, which gets its own bound).
    per_px = 3.2 if kind in ("noise", "mixed") else (0.5 if args.standard_huffman else 1.0)
    cap = int((y1 - y0 + lay.mcu_height) * lay.padded_width * per_px) + (1 << 20)
    out = torch.empty(cap * max(batch, 1), dtype=torch.uint8, device=dev)
    ext = torch.cuda.ExternalStream(ctx.stream, device=dev)
    torch.cuda.synchronize()

    host_reduce = world > 1 and dist.get_backend() != "nccl"

    def step():
        if batch:
            lens = ctx.encode_batch(rgb.data_ptr(), B.RGB8, stride, image_stride, batch, params, out.data_ptr(), cap)
            return int(lens.sum())
        if use_rccl and not args.analyze_only:
            return ctx.encode_sharded(comms[0], rgb.data_ptr(), B.RGB8, stride, params, out.data_ptr(), cap)
        ctx.analyze_band(rgb.data_ptr(), B.RGB8, stride, params, hist.data_ptr())
        if args.analyze_only:
            return 0
        if world > 1 and not args.standard_huffman:
            cur = torch.cuda.current_stream()
            cur.wait_stream(ext)
            if host_reduce:      # gloo rehearsal: through host memory, like tests/test_sharding.py
                h = hist.cpu()
                dist.all_reduce(h)
                hist.copy_(h)
            else:
                dist.all_reduce(hist)
            ext.wait_stream(cur)
        # One rank: the band's own histograms are the image's, no summed copy to hand back.
        return ctx.finish_band(hist.data_ptr() if world > 1 else None, out.data_ptr(), cap)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # A band of an image sharded over several ranks is a fraction of a millisecond of kernels
    # between waits of the host (the all-reduce, the histograms, the result).  The timed region
    # encodes one image at a time (depth 1), which is what one large image costs.  A stream of
    # images can be pipelined over `depth` contexts.  Depth 3: while image i is transformed and
    # counted, image i - 1 is reduced and its bit writer launched, and the result of image
    # i - 2 is collected.  Depth 2: image i + 1 is transformed while image i is reduced and
    # written.  Every image is still encoded completely and in order, and every rank issues
    # its all-reduces in the same order.
    depth = min(3, max(1, args.pipeline_depth)) if (not batch and not args.analyze_only) else 1
    lanes = [(ctx, hist, ext, out)]

    def add_lanes(total):
        while use_rccl and len(comms) < total:
            comms.append(make_comm())
        while len(lanes) < total:
            c = B.Context(local_rank)
            c.set_profiling(True)
            lanes.append((c, torch.zeros((4, 256), dtype=torch.int64, device=dev),
                          torch.cuda.ExternalStream(c.stream, device=dev), torch.empty(cap, dtype=torch.uint8, device=dev)))
        torch.cuda.synchronize()

    add_lanes(depth)

    def analyze(k):
        c, h, _, _ = lanes[k]
        c.analyze_band(rgb.data_ptr(), B.RGB8, stride, params, h.data_ptr())

    def reduce(k):
        c, h, e, _ = lanes[k]
        if use_rccl and not args.standard_huffman:
            c.allreduce_histograms(comms[k], h.data_ptr())
        elif world > 1 and not args.standard_huffman:
            cur = torch.cuda.current_stream()
            cur.wait_stream(e)
            if host_reduce:
                hh = h.cpu()
                dist.all_reduce(hh)
                h.copy_(hh)
            else:
                dist.all_reduce(h)
            e.wait_stream(cur)

    def start(k):
        reduce(k)
        c, h, _, o = lanes[k]
        c.finish_band_start(h.data_ptr() if world > 1 else None, o.data_ptr(), cap)

    def wait(k, acc):
        return lanes[k][0].finish_band_wait()

    def run(count, acc, depth=depth):
        n = 0
        if depth == 1:
            for _ in range(count):
                n = step()
                if acc is not None and batch:
                    acc += ctx.timings()
            return n
        if depth == 2:
            for i in range(count + 1):
                if i < count:
                    analyze(i % 2)
                if i >= 1:
                    start((i - 1) % 2)
                    n = wait((i - 1) % 2, acc)
            return n
        for i in range(count + 2):
            if i < count:
                analyze(i % 3)
            if 1 <= i <= count:
                start((i - 1) % 3)
            if i >= 2:
                n = wait((i - 2) % 3, acc)
        return n

    # One step before the warm-up proper: workspaces are allocated and, with several ranks, the
    # communicator is set up by the first collective -- neither belongs to a timed step even
    # when the caller asks for no warm-up.
    n_out = run(1, None)
    n_out = run(args.warmup, None)
    sync()
    kern_ms = np.zeros(4)
    host_tl = np.zeros(8)
    entropy_runs = [0, 0]       # launches of the entropy stage, images (equal unless a bit writer had to be run again)
    for lane in lanes:
        lane[0].set_profiling(True)      # clears the sums the library keeps per context
    t0 = time.perf_counter()
    n_out = run(args.steps, kern_ms)
    sync()
    elapsed = time.perf_counter() - t0
    if not batch:       # single images: the library has added up stage times and host stations itself
        for lane in lanes:
            ms, us, imgs, runs = lane[0].profile_sums()
            kern_ms += ms
            host_tl += us
            entropy_runs[0] += runs
            entropy_runs[1] += imgs
    own_elapsed = elapsed
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())
    kern_ms /= max(args.steps, 1)
    host_tl /= max(args.steps, 1)
    # Every rank's own figures, for reading a scaling curve: the kernels of its band, the
    # stations of its host, its step time.
    mine = {"rank": rank, "ms_per_step": round(own_elapsed / max(args.steps, 1) * 1e3, 4),
            "stage_ms": {"transform": round(float(kern_ms[0]), 4), "histogram": round(float(kern_ms[1]), 4),
                         "entropy": round(float(kern_ms[2]), 4), "host_tables": round(float(kern_ms[3]), 4)},
            "host_us": dict(zip(["launch_analyze", "to_finish (exchange)", "wait_histograms", "tables_header", "launch_entropy",
                                 "to_wait", "wait_result", "between_steps"], [round(float(v), 1) for v in host_tl]))}
    per_rank = [mine]
    if world > 1:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)

    if rank == 0:
        mp = width * height / 1e6 * (batch * world if batch else 1)
        ms_per_step = elapsed / args.steps * 1e3
        alg = algorithmic_bytes(lay) * max(batch, 1)
        achieved = alg / (kern_ms[0] * 1e-3) / 1e9 if kern_ms[0] > 0 else 0.0
        res = {
            "metric": "megapixels/sec encoded (4:2:0, q=90) at 1/2/4/8 MI355X; achieved HBM GB/s",
            "value": round(mp / (elapsed / args.steps), 1),
            "unit": "MP/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "weak" if batch else "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": desc, "name": args.workload, "width": width, "height": height,
                       "sampling": sname, "quality": quality,
                       "restart_interval_mcus": int(params.restart_interval),
                       "huffman": "standard (Annex K)" if args.standard_huffman else "optimised per image",
                       "frames_per_gpu_per_step": batch or None,
                       "parallelism": ("frames-per-gpu x%d" if batch else "band-per-gpu x%d") % world,
                       "images_in_flight": depth,
                       "jpeg_bytes_rank0": int(n_out)},
            "collective": collective,
            "roofline": {"bound": "hbm", "kernel": "fused transform (colour, subsample, DCT, quantize, zig-zag)",
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4),
                         "traffic": measured_traffic(args.workload, world, args.frames),
                         "algorithmic_bytes_per_launch": int(alg), "kernel_ms": round(float(kern_ms[0]), 4)},
            # Every stage against the same HBM peak (bytes each stage has to move / its time); the
            # histogram and bit-writer passes are bound by instruction issue, not by memory.
            "roofline_stages": stage_rooflines(lay, max(batch, 1), int(n_out), kern_ms),
            "stage_ms": {"transform": round(float(kern_ms[0]), 4), "histogram": round(float(kern_ms[1]), 4),
                         "entropy": round(float(kern_ms[2]), 4), "host_tables": round(float(kern_ms[3]), 4)},
            "entropy_stage_runs_per_image": round(entropy_runs[0] / entropy_runs[1], 3) if entropy_runs[1] else None,
            "per_rank": per_rank,
        }
        if not args.no_cpu_baseline and world == 1:
            def gpu_file(i):
                if not batch:
                    return out[:n_out].cpu().numpy().tobytes()
                one = torch.empty(cap, dtype=torch.uint8, device=dev)
                n = ctx.encode(rgb.data_ptr() + i * image_stride, B.RGB8, stride, params, one.data_ptr(), cap)
                return one[:n].cpu().numpy().tobytes()
            res["cpu_baseline"] = cpu_baseline(rgb, width, height, batch, scheme, qt, int(params.restart_interval), gpu_file)
        elif not args.no_cpu_baseline:
            res["cpu_baseline"] = None

    # The extra measurement below must never cost the line above: if it has not come back
    # after two minutes (it never took more than a second), every rank leaves and rank 0
    # prints what it has.
    def give_up():
        if rank == 0:
            res["pipelined"] = {"error": "no result within 120 s"}
            print(json.dumps(res), flush=True)
        os._exit(0)

    watchdog = threading.Timer(120.0, give_up)
    watchdog.daemon = True
    watchdog.start()
    # For information beside the headline: the same images three in flight per rank (see
    # above).  Kernels of neighbouring images then share the GPU, so their individual
    # durations say nothing about the kernels; the roofline figures stay those of the timed
    # region above, where one image is in flight.
    in_flight = None
    try:
        if depth == 1 and not batch and not args.analyze_only and not args.band_of and args.pipelined_steps > 0:
            add_lanes(3)
            run(min(args.warmup, 5), None, 3)
            sync()
            t1 = time.perf_counter()
            run(args.pipelined_steps, None, 3)
            sync()
            tp = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=dev)
            if world > 1:
                dist.all_reduce(tp, op=dist.ReduceOp.MAX)
            dt = float(tp.item()) / args.pipelined_steps
            in_flight = {"images_in_flight": 3, "steps": args.pipelined_steps, "ms_per_step": round(dt * 1e3, 4),
                         "value": round(width * height / 1e6 / dt, 1), "unit": "MP/s"}
    except Exception as e:      # reported, not fatal: the headline stands on its own
        in_flight = {"error": "%s: %s" % (type(e).__name__, e)}
    watchdog.cancel()

    if rank == 0:
        if in_flight is not None:
            res["pipelined"] = in_flight
        print(json.dumps(res), flush=True)
    for lane in lanes:
        lane[0].close()
    for c in comms:
        if c is not None:
            c.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
