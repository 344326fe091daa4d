#!/bin/bash
# where a 2048-token prompt spends its time (rocprofv3 kernel trace)
mkdir -p gpurun_out/r2u
R=/root/repo
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r2u/prof -o p2048 -- python $R/bench.py --prompt 2048 --n-ctx 4096 --steps 8 --warmup 2 --no-cpu --no-north-star --no-lock-step --prefill-long 0 > $R/gpurun_out/r2u/run.log 2>&1
cd $R
db=$(find gpurun_out/r2u/prof -name "*results.db" | head -1)
python scripts/prof_summary.py $db gpurun_out/r2u/prefill2048_7b_q4_0 > /dev/null 2>&1
head -14 gpurun_out/r2u/prefill2048_7b_q4_0_kernel_stats.md | cut -c1-180
find gpurun_out/r2u/prof -name "*.db" -delete
