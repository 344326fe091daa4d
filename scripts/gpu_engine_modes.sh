#!/bin/bash
# One gpurun call: the persistent engine's timeline under its tuning modes (FALCON_HIP_ENGINE_DEBUG_MODE bits, kernels.h).
# usage: scripts/gpu_engine_modes.sh <tag> [spec] [modes...]
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp PYTHONUNBUFFERED=1
TAG=${1:-eng}; SPEC=${2:-q4_0:7b:32}; shift 2 || true
MODES=${@:-0 60 2 62 1}
OUT=gpurun_out/$TAG; mkdir -p $OUT
for m in $MODES; do
  echo "#### debug mode $m" | tee -a $OUT/engine_modes.log
  FALCON_HIP_ENGINE_DEBUG_MODE=$m timeout 300 python scripts/gpu_engine_debug.py $SPEC >> $OUT/engine_modes.log 2>&1
  echo "rc=$?" >> $OUT/engine_modes.log
done
grep -E "####|tok/s|whole block|LN  |A rows|FF sweep|B1 rows|att sweep|B2 rows|cbar|loader|consumer 0|failure|failed" $OUT/engine_modes.log | head -150
