# Kernel averages and fetched bytes of the 16k workload (GPU box): bash scripts/split_probe.sh <tag>
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-probe}
OUT=gpurun_out/$TAG
rm -rf $OUT && mkdir -p $OUT
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --pipelined-steps 0 > $OUT/stats.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 bench.py --pipelined-steps 0 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/fetch.log 2>&1
python3 - <<PY
import csv,glob,collections
f=glob.glob('$OUT/stats/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if 'jpeghip' in r['Name'] and 'synthetic' not in r['Name']:
        print('%-44s %8.1f us' % (r['Name'].split('(')[0].replace('jpeghip::','')[-44:], float(r['AverageNs'])/1e3))
f=glob.glob('$OUT/fetch/*/*counter_collection.csv')[0]
agg=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if 'jpeghip' in r['Kernel_Name']: agg[r['Kernel_Name'].split('(')[0][-30:]].append(float(r['Counter_Value']))
for k,v in agg.items(): print('fetch MB (x2 corrected)', k, round(sum(v)/len(v)*2*1024/1e6,1))
PY
