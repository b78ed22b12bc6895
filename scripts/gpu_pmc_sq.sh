set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmcx/sq -- python3 bench.py --pipelined-steps 0 --steps 2 --warmup 1 --no-cpu-baseline --workload 16k_420_q90 > gpurun_out/pmcx.log 2>&1
python3 - <<PY
import csv,glob,collections
fs=glob.glob('gpurun_out/pmcx/sq/*/*counter_collection.csv')
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(fs[0])):
    k=r['Kernel_Name']
    if 'jpeghip' not in k: continue
    agg[k.split('(')[0][-40:]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,d in agg.items():
    print(k, {c: round(sum(v)/len(v)/1e6,2) for c,v in d.items()})
PY
