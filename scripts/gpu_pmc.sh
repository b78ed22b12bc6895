# PMC passes for one workload (each its own run): bash scripts/gpu_pmc.sh <workload> <tag>
set -e
WL=${1:-16k_420_q90}; TAG=${2:-pmc}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { # name counters...
  local name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/$TAG/$name -- python3 bench.py --pipelined-steps 0 --steps 2 --warmup 1 --no-cpu-baseline --workload $WL > gpurun_out/$TAG.$name.log 2>&1
}
run sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
run sq2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM
run fetch FETCH_SIZE
run write WRITE_SIZE
python3 - <<PY
import csv,glob,collections
for name in ('sq','sq2','fetch','write'):
    fs=glob.glob('gpurun_out/$TAG/%s/*/*counter_collection.csv'%name)
    if not fs: print(name,'no csv'); continue
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k=r['Kernel_Name']
        if 'jpeghip' not in k: continue
        agg[k.split('(')[0][-40:]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,d in agg.items():
        print(name, k, {c: round(sum(v)/len(v),1) for c,v in d.items()})
PY
