#!/bin/bash
# counters of the 128-token tile GEMM (Falcon-7B Wdown / Wqkv shapes, configurations 2 and 7): where do the waves of a launch spend their cycles, what does the L2 see
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; export TMPDIR=/tmp PYTHONUNBUFFERED=1
T=${1:-gemm_pmc}; mkdir -p gpurun_out/$T
cd /tmp
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_BUSY_CU_CYCLES --kernel-trace -d $R/gpurun_out/$T/sq -o m -- python $R/scripts/gpu_gemm_seg_time.py 128 > $R/gpurun_out/$T/sq.log 2>&1
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --kernel-trace -d $R/gpurun_out/$T/tcc -o m -- python $R/scripts/gpu_gemm_seg_time.py 128 > $R/gpurun_out/$T/tcc.log 2>&1
timeout 300 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum --kernel-trace -d $R/gpurun_out/$T/tcp -o m -- python $R/scripts/gpu_gemm_seg_time.py 128 > $R/gpurun_out/$T/tcp.log 2>&1
cd $R
python - <<'PY' "$T"
import sqlite3, sys, glob
from collections import defaultdict
T = sys.argv[1]
for pat in ("sq", "tcc", "tcp"):
    for db in glob.glob("gpurun_out/%s/%s/**/*results.db" % (T, pat), recursive=True):
        c = sqlite3.connect(db).cursor()
        acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
        for name, gx, counter, val in c.execute("select kernel_name, grid_size_x, counter_name, value from counters_collection"):
            if "k_gemm_q" not in name: continue
            k = name.split("(")[0].replace("void ", "")[:28] + " grid %d" % gx
            acc[k][counter][0] += 1; acc[k][counter][1] += val
        for k, d in acc.items():
            print(pat, k, {cn: round(v[1] / v[0]) for cn, v in sorted(d.items())})
PY
find gpurun_out/$T -name "*.db" -delete
