#!/bin/bash
# k-quant single-stream decode on Falcon-40B shapes: tok/s per format (all 60 blocks), ring forms on / off in ONE call, + a kernel trace (12 blocks) per format
# usage: scripts/gpu_kq_decode.sh <tag> [formats ...]      -> gpurun_out/<tag>/kq_decode.txt, decode_40b_<fmt>_kernel_stats.{csv,md}
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
export TMPDIR=/tmp PYTHONUNBUFFERED=1
TAG=${1:-r04x}; shift || true
FMTS=${*:-q4_k q2_k q3_k q5_k q6_k}
OUT=gpurun_out/$TAG
mkdir -p $OUT
B="--model 40b --no-cpu --no-ref-order --no-cli --no-lock-step --no-north-star --prefill-long 0 --steps 32 --warmup 4 --repeats 3"
for q in $FMTS; do
  for ring in 1 0; do
    v=$(FALCON_HIP_RING=$ring timeout 600 python bench.py $B --quant $q 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f tok/s  %.3f ms  step_frac %.3f  launch_frac %.3f' % (d['value'], d['ms_per_step'], d['roofline']['step_frac'], d['roofline']['frac'] or 0))")
    echo "40b $q ring=$ring: $v" | tee -a $OUT/kq_decode.txt
  done
  if [ "${TRACE:-1}" = 1 ]; then
    cd /tmp
    timeout 600 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof_$q -o trace -- python $R/bench.py $B --quant $q --layers 12 --no-graph --steps 16 --warmup 2 --repeats 1 > $R/$OUT/prof_$q.log 2>&1
    cd $R
    db=$(find $OUT/prof_$q -name "*results.db" | head -1)
    [ -n "$db" ] && python scripts/prof_summary.py $db $OUT/decode_40b_$q > /dev/null 2>&1 && head -9 $OUT/decode_40b_${q}_kernel_stats.md | cut -c1-150
    rm -rf $OUT/prof_$q
  fi
done
