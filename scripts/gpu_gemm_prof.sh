#!/bin/bash
# rocprofv3 kernel trace of scripts/gpu_gemm_time.py; prints the average k_gemm_q duration per (shape, debug mode) group of 13 calls
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
cd /tmp
timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/prof -o gemm -- python $R/scripts/gpu_gemm_time.py ${1:-128} > $R/gpurun_out/gemm_time.log 2>&1
cd $R
python - <<PY
import sqlite3
c = sqlite3.connect("gpurun_out/prof/gemm_results.db").cursor()
rows = c.execute("select name, (end - start) / 1000.0, grid_x, workgroup_x from kernels where name like '%k_gemm_q%' order by start").fetchall()
names = ["qkv", "wo", "up", "down"]; modes = ["full", "no loads", "no math", "neither"]
for i in range(0, len(rows), 13):
    grp = rows[i:i + 13][3:]
    k = i // 13
    print("%-5s %-9s %-28s grid %5d x %4d: avg %8.1f us  min %8.1f" % (names[k // 4] if k // 4 < 4 else "?", modes[k % 4], grp[0][0][5:30], grp[0][2] // grp[0][3], grp[0][3], sum(r[1] for r in grp) / len(grp), min(r[1] for r in grp)))
PY
