#!/bin/bash
# rocprofv3 kernel traces of the decode step in the default order and in the fast reference order (same box, same model): profiles/<tag>_decode_order{0,2}_*
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
export TMPDIR=/tmp
TAG=${1:-r06}
mkdir -p gpurun_out/prof
cd /tmp
for mode in 0 2; do
  timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o ${TAG}_order${mode} -- python $R/scripts/gpu_decode_mode.py $mode ${2:-32} ${3:-48} > $R/gpurun_out/prof_order${mode}.log 2>&1
  echo "rocprof mode $mode exit $?"
done
cd $R
for mode in 0 2; do
  f=$(find gpurun_out/prof -name "${TAG}_order${mode}_results.db" | head -1)
  [ -n "$f" ] && python scripts/prof_summary.py $f gpurun_out/${TAG}_decode_order${mode} | head -14
done
