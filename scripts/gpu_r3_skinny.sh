#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp PYTHONUNBUFFERED=1
TAG=${1:-r03sk}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_mul_mat.py -x -q --no-header -p no:cacheprovider -k "prefill_gemm" > $OUT/pytest_gemm.log 2>&1; echo "pytest gemm rc=$?"; tail -12 $OUT/pytest_gemm.log
echo "== skinny"; timeout 300 python scripts/gpu_skinny_time.py 5 8 16 2>&1 | tee $OUT/skinny.log | tail -20
echo "== tile gemm"; FQ_GEMM_SKINNY=0 timeout 300 python scripts/gpu_skinny_time.py 8 16 2>&1 | tee $OUT/tile.log | tail -14
