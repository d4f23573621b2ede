#!/bin/bash
# GPU box: hardware counters of the builder's kernels, one rocprofv3 --pmc pass per counter set
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out/run
i=0
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL"; do
  i=$((i+1))
  timeout -s KILL 90 rocprofv3 --pmc $set --output-format csv -d gpurun_out/run/pmc_${TAG:-x}_$i -- python3 tools/builder_trace.py > gpurun_out/run/pmc_${TAG:-x}_$i.log 2>&1 || { tail -5 gpurun_out/run/pmc_${TAG:-x}_$i.log; exit 1; }
  echo "pass $i done"
done
python3 - <<'PY'
import csv, glob, os, collections
tag=os.environ.get('TAG','x')
for i in (1,2):
    f=sorted(glob.glob('gpurun_out/run/pmc_%s_%d/*/*_counter_collection.csv'%(tag,i)))[-1]
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ','').split('(')[0][:48]
        agg[k][r['Counter_Name']]+=float(r['Counter_Value']); 
        cnt[(k,r['Counter_Name'])]+=1
    for k,v in agg.items():
        if 'scatter' in k or 'histogram' in k or 'emit' in k or 'segments' == k:
            print(k, {c: '%.3g'%(x/cnt[(k,c)]) for c,x in v.items()})
PY
