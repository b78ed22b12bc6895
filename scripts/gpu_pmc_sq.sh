# SQ counters of every library kernel of a workload, two passes: WL=16k_420_q90 bash scripts/gpu_pmc_sq.sh  (GPU box)
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
WL=${WL:-16k_420_q90}
for PASS in "a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "b SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM"; do
  set -- $PASS
  NAME=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmcsq_${WL}_$NAME -- python3 bench.py --pipelined-steps 0 --steps 2 --warmup 1 --no-cpu-baseline --workload $WL > gpurun_out/pmcsq_${WL}_$NAME.log 2>&1
  python3 - <<PY
import csv,glob,collections
fs=glob.glob('gpurun_out/pmcsq_${WL}_$NAME/*/*counter_collection.csv')
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(fs[0])):
    k=r['Kernel_Name']
    if 'jpeghip' not in k: continue
    agg[k.split('(')[0][-32:]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,d in agg.items():
    print(k, {c.replace('SQ_',''): round(sum(v)/len(v)/1e6,2) for c,v in d.items()})
PY
done
