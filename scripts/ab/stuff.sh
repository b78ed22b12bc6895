# Kernel averages of the stuffing pass: STUFF="0 4096 5120" bash scripts/ab/stuff.sh <workload>...   (JPEGENC_STUFF_WAVES values)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for WL in "$@"; do
for w in ${STUFF:-0 5120}; do
  export JPEGENC_STUFF_WAVES=$w
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abs_$w -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --pipelined-steps 0 --workload $WL > gpurun_out/abs.log 2>&1 || { tail -5 gpurun_out/abs.log; exit 1; }
  python3 - $w $WL <<'PY'
import csv,glob,sys,json
f=sorted(glob.glob('gpurun_out/abs_%s/*/*kernel_stats.csv' % sys.argv[1]))[-1]
out=[]
for r in csv.DictReader(open(f)):
    n=r['Name']
    if 'jpeghip' in n and ('stuff' in n or 'group_sums' in n):
        out.append('%s %.1f' % (n.split('(')[0].split('::')[-1][:22], float(r['AverageNs'])/1e3))
d=json.loads([l for l in open('gpurun_out/abs.log') if l.startswith('{')][-1])
print(sys.argv[2].ljust(20), ('waves=%s' % sys.argv[1]).ljust(12), ' | '.join(out), ' step %.4f entropy %.4f' % (d['ms_per_step'], d['stage_ms']['entropy']))
PY
done
done
