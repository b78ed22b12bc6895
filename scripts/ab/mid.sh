# The entropy stage of the middle-density pictures with the writer the library picks and with the dense writer forced.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for WL in "$@"; do
for w in auto dense; do
  if [ $w = dense ]; then export JPEGENC_WRITER=dense; else unset JPEGENC_WRITER; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abm_$w -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --pipelined-steps 0 --workload $WL > gpurun_out/abm.log 2>&1 || { tail -5 gpurun_out/abm.log; exit 1; }
  python3 - $w $WL <<'PY'
import csv,glob,sys,json
f=sorted(glob.glob('gpurun_out/abm_%s/*/*kernel_stats.csv' % sys.argv[1]))[-1]
out=[]
for r in csv.DictReader(open(f)):
    n=r['Name']
    if 'jpeghip' in n and not any(k in n for k in ('transform','synthetic','histogram')):
        out.append('%s %.1f' % (n.split('(')[0].split('::')[-1][:20], float(r['AverageNs'])/1e3))
d=json.loads([l for l in open('gpurun_out/abm.log') if l.startswith('{')][-1])
print(sys.argv[2].ljust(20), sys.argv[1].ljust(6), 'step %.4f entropy %.4f |' % (d['ms_per_step'], d['stage_ms']['entropy']), ' | '.join(out))
PY
done
done
