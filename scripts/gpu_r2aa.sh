#!/bin/bash
# which SQ counters exist; then what the lock-step column kernels wait for
mkdir -p gpurun_out/r2aa
R=/root/repo
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --force-pipeline --streams 2 --pipe-batch 4 --steps 8 --warmup 2"
export FALCON_HIP_STAGE_GRAPH=0
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace -d $R/gpurun_out/r2aa -o a -- $CMD > $R/gpurun_out/r2aa/a.log 2>&1; echo "a rc=$?"
timeout 600 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --kernel-trace -d $R/gpurun_out/r2aa -o b -- $CMD > $R/gpurun_out/r2aa/b.log 2>&1; echo "b rc=$?"
cd $R
for n in a b; do
db=$(find gpurun_out/r2aa -name "${n}*results.db" | head -1)
python - <<PY
import sqlite3
from collections import defaultdict
try:
    c=sqlite3.connect("$db").cursor()
    acc=defaultdict(lambda: defaultdict(float)); n=defaultdict(lambda: defaultdict(int))
    for name,cn,val in c.execute("select kernel_name, counter_name, value from counters_collection"):
        k=name.replace("(anonymous namespace)::","").split("(")[0].replace("void ","")[:44]; acc[k][cn]+=val; n[k][cn]+=1
    for k,v in acc.items():
        if 'cols' in k or 'gemv_ln' in k: print(k, {a: round(b/max(n[k][a],1)/1e3,1) for a,b in v.items()}, '(thousands per launch)')
except Exception as e: print('no db', e)
PY
done
find gpurun_out/r2aa -name "*.db" -delete
