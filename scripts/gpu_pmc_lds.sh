# LDS counters of the library's kernels for one workload: WL=16k_420_q90 bash scripts/gpu_pmc_lds.sh <tag>
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
WL=${WL:-16k_420_q90}; TAG=${1:-lds}
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d gpurun_out/pmc_$TAG -- python3 bench.py --pipelined-steps 0 --steps 2 --warmup 1 --no-cpu-baseline --workload $WL > gpurun_out/pmc_$TAG.log 2>&1
python3 - $TAG <<'PY'
import csv,glob,collections,sys
fs=glob.glob('gpurun_out/pmc_%s/*/*counter_collection.csv' % sys.argv[1])
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(fs[0])):
    k=r['Kernel_Name']
    if 'jpeghip' not in k: continue
    agg[k.split('(')[0][-32:]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,d in agg.items():
    if 'hist' in k or 'encode' in k or 'transform' in k:
        print(k, {c.replace('SQ_',''): round(sum(v)/len(v)/1e6,2) for c,v in d.items()})
PY
