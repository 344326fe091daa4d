#!/bin/bash
# rocprofv3 kernel trace of an arbitrary bench.py command line: scripts/gpu_prof_cmd.sh <tag> <bench.py args...>
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
tag=$1; shift
cd /tmp
timeout 900 rocprofv3 --kernel-trace -d $R/gpurun_out/prof -o $tag -- python $R/bench.py "$@" --no-cpu --no-ref-order --no-graph > $R/gpurun_out/prof_$tag.log 2>&1
cd $R
python scripts/prof_summary.py gpurun_out/prof/${tag}_results.db gpurun_out/prof/${tag}_kernel_stats.csv | head -14
