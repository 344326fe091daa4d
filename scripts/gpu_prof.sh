#!/bin/bash
# rocprofv3 kernel trace of a short decode bench (plain launches: stream capture under the tracer crashes rocprofv3 7.2)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o ${1:-r1} -- python $R/bench.py --steps ${2:-48} --no-cpu --no-ref-order --no-graph > $R/gpurun_out/prof_run.log 2>&1
echo "rocprof exit $?"
cd $R
find gpurun_out/prof -type f | head -20
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -30 "$f"
tail -c 1500 gpurun_out/prof_run.log
