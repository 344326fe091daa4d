#!/bin/bash
# One gpurun call: engine tests first (bounded), then the timeline under tuning modes.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp PYTHONUNBUFFERED=1
TAG=${1:-r03eng}; shift || true
OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_engine.py -x -q --no-header -p no:cacheprovider > $OUT/pytest_engine.log 2>&1; echo "pytest engine rc=$?"; tail -15 $OUT/pytest_engine.log
timeout 300 python scripts/gpu_engine_debug.py q4_0:7b:3 > $OUT/dbg3.log 2>&1; echo "dbg3 rc=$?"; head -40 $OUT/dbg3.log
for m in ${@:-0}; do
  echo "#### debug mode $m" | tee -a $OUT/engine_modes.log
  FALCON_HIP_ENGINE_DEBUG_MODE=$m timeout 300 python scripts/gpu_engine_debug.py q4_0:7b:32 >> $OUT/engine_modes.log 2>&1
  echo "rc=$?" >> $OUT/engine_modes.log
done
cat $OUT/engine_modes.log | cut -c1-220 | head -230
