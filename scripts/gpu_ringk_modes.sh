#!/bin/bash
# tuning aid: per-kernel time of the k-quant ring forms with pieces compiled out at run time (FQ_RING_DEBUG bits: 1 = the loader does not wait for the
# prologue's loads, 2 = no row dots (results are garbage: timing only), 32 = generic consumer loop instead of the fast path)
# usage: scripts/gpu_ringk_modes.sh <tag> <quant> [modes ...]
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
export TMPDIR=/tmp PYTHONUNBUFFERED=1
TAG=${1:-r04x}; Q=${2:-q4_k}; shift 2 || true
MODES=${*:-0 2}
OUT=gpurun_out/$TAG
mkdir -p $OUT
for m in $MODES; do
  cd /tmp
  FQ_RING_DEBUG=$m timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof_m$m -o trace -- python $R/bench.py --model 40b --quant $Q --layers 8 --no-cpu --no-ref-order --no-cli --no-lock-step --no-north-star --prefill-long 0 --no-graph --steps 16 --warmup 2 --repeats 1 > $R/$OUT/prof_m$m.log 2>&1
  cd $R
  db=$(find $OUT/prof_m$m -name "*results.db" | head -1)
  [ -n "$db" ] && python scripts/prof_summary.py $db $OUT/modes_${Q}_m$m > /dev/null 2>&1 && echo "== $Q FQ_RING_DEBUG=$m" && grep -E "k_ring|k_gemv|k_attn_decode" $OUT/modes_${Q}_m${m}_kernel_stats.md | head -5 | cut -c1-120
  rm -rf $OUT/prof_m$m
done
