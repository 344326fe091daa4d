#!/bin/bash
# rocprofv3 kernel trace of any command; prints the per-kernel summary.  usage: scripts/gpu_prof_any.sh <tag> <command...>
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
export TMPDIR=/tmp PYTHONUNBUFFERED=1
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof -o trace -- "$@" > $R/$OUT/run.log 2>&1
echo "rocprof rc=$?"
cd $R
db=$(find $OUT/prof -name "*results.db" | head -1)
[ -n "$db" ] && python scripts/prof_summary.py $db $OUT/kernels > /dev/null 2>&1 && head -${PROF_LINES:-16} $OUT/kernels_kernel_stats.md | cut -c1-200
find $OUT/prof -name "*.db" -delete
tail -3 $OUT/run.log
