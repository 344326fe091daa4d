#!/bin/bash
# counters of the prefill attention launches (71 heads, 2048 tokens): matrix-pipe busy, VALU, waits; kernel durations
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; export TMPDIR=/tmp PYTHONUNBUFFERED=1
T=${1:-r05d}; mkdir -p gpurun_out/$T/pmc
cd /tmp
FORMS=${FORMS:-32,1} timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$T/trace -o t -- python $R/scripts/gpu_attn_forms.py 2048 > $R/gpurun_out/$T/trace.log 2>&1
FORMS=${FORMS:-32,1} timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace -d $R/gpurun_out/$T/pmc -o m -- python $R/scripts/gpu_attn_forms.py 2048 > $R/gpurun_out/$T/pmc.log 2>&1
FORMS=${FORMS:-32,1} timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAVES --kernel-trace -d $R/gpurun_out/$T/pmc2 -o m -- python $R/scripts/gpu_attn_forms.py 2048 > $R/gpurun_out/$T/pmc2.log 2>&1
cd $R
python - <<'PY' "$T"
import sqlite3, sys, glob
from collections import defaultdict
T = sys.argv[1]
for pat in ("pmc", "pmc2"):
    for db in glob.glob("gpurun_out/%s/%s/**/*results.db" % (T, pat), recursive=True):
        c = sqlite3.connect(db).cursor()
        acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
        for name, counter, val in c.execute("select kernel_name, counter_name, value from counters_collection"):
            k = name.split("(")[0].replace("void ", "")[:40]
            acc[k][counter][0] += 1; acc[k][counter][1] += val
        for k, d in acc.items():
            if "attention" in k or "pack" in k:
                print(k, {cn: round(v[1] / v[0]) for cn, v in sorted(d.items())})
for db in glob.glob("gpurun_out/%s/trace/**/*results.db" % T, recursive=True):
    c = sqlite3.connect(db).cursor()
    try:
        for row in c.execute("select name, count(*), avg(end-start), min(end-start) from kernels group by name order by 3 desc limit 8"):
            print(row)
    except Exception as e:
        print("trace:", e)
        for (n,) in c.execute("select name from sqlite_master where type in ('table','view')"): print(" ", n)
PY
find gpurun_out/$T -name "*.db" -size +20M -delete
