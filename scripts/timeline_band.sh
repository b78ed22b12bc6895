# Kernel timeline of rank 0's band of the 16k image sharded N ways (no all-reduce): N=8 bash scripts/timeline_band.sh  (GPU box)
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
N=${N:-8}
timeout -k 10 200 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/tl_band$N -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --pipelined-steps 0 --band-of $N > gpurun_out/tl_band$N.log 2>&1
tail -1 gpurun_out/tl_band$N.log | cut -c1-400
python3 - <<PY
import csv, glob
ev = []
for f in glob.glob('gpurun_out/tl_band$N/*/*kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'K ' + r['Kernel_Name'].split('(')[0][-44:]))
for f in glob.glob('gpurun_out/tl_band$N/*/*memory_copy_trace.csv'):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'C ' + r['Direction']))
ev.sort()
idx = [i for i, e in enumerate(ev) if 'k_transform' in e[2]]
start = idx[-3]
t0 = ev[start][0]
pe = t0
for s, e, n in ev[start:idx[-1] + 1]:
    print('%8.1f us  dur %7.1f  gap %6.1f  %s' % ((s - t0) / 1e3, (e - s) / 1e3, (s - pe) / 1e3, n))
    pe = max(pe, e)
PY
