#!/bin/bash
# PMC passes over a 2048-token prompt: HBM traffic and matrix-pipe occupancy of the prefill attention / GEMM
mkdir -p gpurun_out/r2z
R=/root/repo
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --prompt 2048 --n-ctx 4096 --steps 2 --warmup 1 --no-cpu --no-north-star --no-lock-step --prefill-long 0 --no-graph"
for c in FETCH_SIZE WRITE_SIZE; do
  n=$(echo $c | tr 'A-Z' 'a-z' | sed 's/_size//')
  timeout 600 rocprofv3 --pmc $c --kernel-trace -d $R/gpurun_out/r2z -o $n -- $CMD > $R/gpurun_out/r2z/$n.log 2>&1; echo "$c rc=$?"
done
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_INSTS_LDS --kernel-trace -d $R/gpurun_out/r2z -o mfma -- $CMD > $R/gpurun_out/r2z/mfma.log 2>&1; echo "mfma rc=$?"
cd $R
fdb=$(find gpurun_out/r2z -name "fetch*results.db" | head -1); wdb=$(find gpurun_out/r2z -name "write*results.db" | head -1); mdb=$(find gpurun_out/r2z -name "mfma*results.db" | head -1)
python scripts/pmc_summary.py $fdb $wdb gpurun_out/r2z/pmc_traffic.json | head -6
python - <<PY
import sqlite3
c=sqlite3.connect("$mdb").cursor()
from collections import defaultdict
acc=defaultdict(lambda: defaultdict(float)); cnt=defaultdict(int)
for name,cn,val in c.execute("select kernel_name, counter_name, value from counters_collection"):
    k=name.split("(")[0].replace("void ","")[:40]; acc[k][cn]+=val
for k,v in acc.items():
    if 'attention' in k or 'gemm' in k: print(k, {a: round(b/1e6,2) for a,b in v.items()})
PY
find gpurun_out/r2z -name "*.db" -delete
