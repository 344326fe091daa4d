#!/bin/bash
# rocprofv3 kernel traces of the fast reference order's decode launches with pieces compiled out (FQ_REF_DBG / FQ_RING_DEBUG: timing only, results garbage)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
export TMPDIR=/tmp
TAG=${1:-r06}
mkdir -p gpurun_out/prof
cd /tmp
for cfg in ${CFGS:-0_0 7_0 0_128}; do
  set -- ${cfg/_/ }
  M=2; [ "$1" = "d" ] && { M=0; set -- 0 0; }      # cfg "d_0": the default order, same model, for calibration
  FQ_REF_DBG=$1 FQ_RING_DEBUG=$2 timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o ${TAG}_dbg$1_$2_m$M -- python $R/scripts/gpu_decode_mode.py $M ${LAYERS:-8} 32 > $R/gpurun_out/prof_dbg.log 2>&1
  f=$(find $R/gpurun_out/prof -name "${TAG}_dbg$1_$2_m${M}_results.db" | head -1)
  echo "== order $M FQ_REF_DBG=$1 FQ_RING_DEBUG=$2"
  [ -n "$f" ] && python $R/scripts/prof_summary.py $f /tmp/x | grep -E "k_attn_out|k_gemv_ln" | head -3
done
