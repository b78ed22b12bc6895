"""Copies what scripts/gpu_final_profiles.sh left under gpurun_out/final into profiles/ under a tag:
kernel statistics, bench lines, counter passes reduced to the library's kernels, and traffic.json
(HBM bytes per launch of the transform kernel = (2 x FETCH_SIZE + WRITE_SIZE) x 1024, the fetch
figure doubled for gfx950 as the microarchitecture guide prescribes).

    python3 scripts/collect_profiles.py r02g
"""
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1]
src = "gpurun_out/final"
names = {"16k_420_q90": "16k_420", "4k_444_q90": "4k_444", "batch_1080p_420_q90": "batch256",
         "1080p_420_q90": "1080p_420", "8k_noise_q100_420": "8k_noise_420", "8k_noise_q100_444": "8k_noise_444"}
launches = {wl: 2 if wl.startswith("batch_") else 1 for wl in names}
traffic = {}
for wl, short in names.items():
    # (a directory may hold earlier runs as well: the newest file counts)
    stats = sorted(glob.glob("%s/stats_%s/*/*kernel_stats.csv" % (src, wl)), key=os.path.getmtime)
    if stats:
        shutil.copy(stats[-1], "profiles/%s_%s_kernel_stats.csv" % (tag, short))
    line = "%s/bench_line_%s.json" % (src, wl)
    if os.path.exists(line) and os.path.getsize(line) > 0:
        shutil.copy(line, "profiles/%s_%s_bench_line.json" % (tag, short))
    per_kernel = {}
    for pas in ("fetch", "write", "sq", "sq2"):
        found = sorted(glob.glob("%s/pmc_%s_%s/*/*counter_collection.csv" % (src, wl, pas)), key=os.path.getmtime)
        if not found:
            continue
        rows = [r for r in csv.DictReader(open(found[-1])) if "jpeghip" in r["Kernel_Name"]]
        if not rows:
            continue
        with open("profiles/%s_%s_pmc_%s.csv" % (tag, short, pas), "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
            w.writeheader()
            w.writerows(rows)
        if pas in ("fetch", "write"):
            for r in rows:
                k = r["Kernel_Name"].split("(")[0].split("::")[-1]
                if "k_transform" in k:
                    per_kernel.setdefault(k, {}).setdefault(pas, []).append(float(r["Counter_Value"]))
    for k, d in per_kernel.items():
        if "fetch" in d and "write" in d:
            fetch = sum(d["fetch"]) / len(d["fetch"])
            write = sum(d["write"]) / len(d["write"])
            traffic[wl] = {"kernel": k, "fetch_size_kib": round(fetch, 1), "write_size_kib": round(write, 1),
                           "fetch_correction": 2.0, "launches_per_step": launches[wl],
                           "traffic_bytes_per_launch": int((2.0 * fetch + write) * 1024),
                           "source": "profiles/%s_%s_pmc_fetch.csv, profiles/%s_%s_pmc_write.csv" % (tag, short, tag, short)}
default = "%s/bench_16k_default.json" % src
if os.path.exists(default) and os.path.getsize(default) > 0:
    shutil.copy(default, "profiles/%s_16k_420_bench_default_run.json" % tag)
for wl in ("4k_444_q90", "batch_1080p_420_q90", "1080p_420_q90", "8k_noise_q100_420", "8k_noise_q100_444"):
    p = "%s/bench_%s.json" % (src, wl)
    if os.path.exists(p) and os.path.getsize(p) > 0:
        shutil.copy(p, "profiles/%s_%s_bench_plain.json" % (tag, wl))
if traffic:
    json.dump(traffic, open("profiles/traffic.json", "w"), indent=1)
print("traffic:", {k: v["traffic_bytes_per_launch"] for k, v in traffic.items()})
