#!/bin/bash
# round 2, call A: new parity tests first, then the whole gpu suite, smoke, the bench line
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r2a
mkdir -p $OUT
nproc > $OUT/host.txt; lscpu | grep -E "Model name|Socket|Core|Thread" >> $OUT/host.txt
t0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_falcon.py -m gpu -q -x --no-header -p no:cacheprovider -s > $OUT/test_gpu_falcon.log 2>&1; echo "falcon exit $? $(( $(date +%s) - t0 ))s" | tee -a $OUT/summary.txt
timeout 900 python -m pytest tests/test_gpu_configs.py -m gpu -q --no-header -p no:cacheprovider -s --durations=0 > $OUT/test_gpu_configs.log 2>&1; echo "configs exit $? $(( $(date +%s) - t0 ))s" | tee -a $OUT/summary.txt
timeout 900 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider --deselect tests/test_gpu_falcon.py --deselect tests/test_gpu_configs.py > $OUT/test_rest.log 2>&1; echo "rest exit $? $(( $(date +%s) - t0 ))s" | tee -a $OUT/summary.txt
timeout 300 python -c "import __graft_entry__ as e; e.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $? $(( $(date +%s) - t0 ))s" | tee -a $OUT/summary.txt
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $? $(( $(date +%s) - t0 ))s" | tee -a $OUT/summary.txt
tail -4 $OUT/test_gpu_falcon.log; tail -15 $OUT/test_gpu_configs.log; tail -3 $OUT/test_rest.log; tail -2 $OUT/smoke.log; tail -c 3000 $OUT/bench.json; tail -3 $OUT/bench.err
