#!/bin/bash
# One gpurun call: GPU parity tests, smoke, bench, rocprof. Everything lands in gpurun_out/.
# usage: scripts/gpu_round.sh [tests|bench|prof|all]   (default all)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
WHAT=${1:-all}
rocm-smi --showproductname 2>/dev/null | head -8 > $OUT/gpu.txt
nproc >> $OUT/gpu.txt; lscpu | grep -E "Model name|Socket|Core|Thread" >> $OUT/gpu.txt
if [[ $WHAT == all || $WHAT == tests ]]; then
  for f in test_gpu_quant test_gpu_mul_mat test_gpu_block_ops test_gpu_falcon test_gpu_shim test_gpu_wquant test_gpu_model_quantize; do
    timeout 600 python -m pytest tests/$f.py -m gpu -q -x --no-header -p no:cacheprovider -s > $OUT/$f.log 2>&1
    echo "$f exit $?" | tee -a $OUT/summary.txt
    tail -3 $OUT/$f.log
  done
  timeout 300 python -c "import __graft_entry__ as e; e.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?" | tee -a $OUT/summary.txt; tail -2 $OUT/smoke.log
fi
if [[ $WHAT == all || $WHAT == bench ]]; then
  timeout 300 python bench.py --layers 4 --steps 32 --no-cpu > $OUT/bench_4layers.json 2> $OUT/bench_4layers.err; echo "bench4 exit $?" | tee -a $OUT/summary.txt; tail -c 1500 $OUT/bench_4layers.json
  timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?" | tee -a $OUT/summary.txt; tail -c 2500 $OUT/bench.json; tail -5 $OUT/bench.err
  timeout 600 python bench.py --no-graph --no-cpu > $OUT/bench_nograph.json 2> $OUT/bench_nograph.err; echo "bench-nograph exit $?" | tee -a $OUT/summary.txt; tail -c 1200 $OUT/bench_nograph.json
fi
if [[ $WHAT == all || $WHAT == prof ]]; then
  cd /tmp
  timeout 600 rocprofv3 --kernel-trace --stats -d $OLDPWD/$OUT/prof -o r1 -- python $OLDPWD/bench.py --steps 64 --no-cpu > $OLDPWD/$OUT/prof_run.log 2>&1
  echo "rocprof exit $?" | tee -a $OLDPWD/$OUT/summary.txt
  cd $OLDPWD
  find $OUT/prof -name "*stats*" | head; 
  for f in $(find $OUT/prof -name "*kernel_stats*csv" | head -1); do head -25 $f; done
fi
