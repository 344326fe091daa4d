#!/bin/bash
# per-kernel times (rocprofv3 kernel trace, 8 blocks of Falcon-40B, plain launches) of the k-quant decode launches, ring forms on and off
# usage: scripts/gpu_kq_kernels.sh <tag> [formats ...]
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
export TMPDIR=/tmp PYTHONUNBUFFERED=1
TAG=${1:-r04x}; shift || true
FMTS=${*:-q4_k q2_k q3_k q5_k q6_k}
OUT=gpurun_out/$TAG
mkdir -p $OUT
for q in $FMTS; do
  for ring in 1 0; do
    cd /tmp
    FALCON_HIP_RING=$ring timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof_k -o trace -- python $R/bench.py --model 40b --quant $q --layers 8 --no-cpu --no-ref-order --no-cli --no-lock-step --no-north-star --prefill-long 0 --no-graph --steps 16 --warmup 2 --repeats 1 > $R/$OUT/prof_k.log 2>&1
    cd $R
    db=$(find $OUT/prof_k -name "*results.db" | head -1)
    [ -n "$db" ] && python scripts/prof_summary.py $db $OUT/kern_${q}_ring$ring > /dev/null 2>&1 && echo "== $q ring=$ring" && grep -E "k_ring|k_gemv_ln|k_gemv_out|k_attn_decode|k_quantize" $OUT/kern_${q}_ring${ring}_kernel_stats.md | grep -v " x 1 " | head -6 | cut -c1-110
    rm -rf $OUT/prof_k
  done
done
