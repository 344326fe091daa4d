#!/bin/bash
# the three-launch form (k_gemv_ln | k_attn_decode | k_gemv_out) in the default order and the fast reference order: isolates each REF kernel's cost
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
cd /tmp
for M in 0 2; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o ${1:-r06}_3l_m$M -- python $R/scripts/gpu_decode_mode.py $M 8 32 1 > $R/gpurun_out/prof_3l.log 2>&1
  f=$(find $R/gpurun_out/prof -name "${1:-r06}_3l_m${M}_results.db" | head -1)
  echo "== order $M, three launches per block"
  [ -n "$f" ] && python $R/scripts/prof_summary.py $f /tmp/x | grep -E "k_attn_decode|k_gemv_out|k_gemv_ln" | head -4
done
