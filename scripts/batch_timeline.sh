# Kernel and copy timeline of the last batch pass: bash scripts/batch_timeline.sh  (GPU box)
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 200 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/batch_tl -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --pipelined-steps 0 --workload batch_1080p_420_q90 > gpurun_out/batch_tl.log 2>&1
python3 - <<'PY'
import csv, glob
ev = []
for f in glob.glob('gpurun_out/batch_tl/*/*kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'K ' + r['Kernel_Name'].split('(')[0][-44:], r.get('Stream_Id', r.get('Queue_Id', ''))))
for f in glob.glob('gpurun_out/batch_tl/*/*memory_copy_trace.csv'):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'C ' + r['Direction'] , ''))
ev.sort()
idx = [i for i, e in enumerate(ev) if 'k_transform' in e[2]]
# a pass has two transform launches: the last pass starts at the second-to-last one
start = idx[-2]
t0 = ev[start][0]
for s, e, n, q in ev[start:]:
    print('%8.1f us  dur %7.1f  %s %s' % ((s - t0) / 1e3, (e - s) / 1e3, n, q))
PY
