#!/bin/bash
# One gpurun call of a round: every -m gpu test, smoke, the bench line the driver reads; optionally rocprofv3 kernel trace + PMC passes.
# usage: scripts/gpu_round.sh <tag> [tests|bench|prof|pmc|all ...]
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
export TMPDIR=/tmp PYTHONUNBUFFERED=1
TAG=${1:-r03x}; shift || true
WHAT=" ${*:-tests bench} "
OUT=gpurun_out/$TAG
mkdir -p $OUT
has() { [[ $WHAT == *" $1 "* || $WHAT == *" all "* ]]; }
if has tests; then
  timeout 1500 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -x > $OUT/pytest_gpu.log 2>&1; echo "pytest -m gpu exit $?" | tee -a $OUT/summary.txt
  tail -5 $OUT/pytest_gpu.log
  timeout 300 python -c "import __graft_entry__ as e; e.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?" | tee -a $OUT/summary.txt; tail -2 $OUT/smoke.log
fi
if has bench; then
  ( time timeout 1200 python bench.py ${BENCH_ARGS:-} ) > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?" | tee -a $OUT/summary.txt; python scripts/bench_brief.py < $OUT/bench.json 2>/dev/null || tail -c 1500 $OUT/bench.json; tail -4 $OUT/bench.err
fi
if has prof; then
  cd /tmp
  timeout 600 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof -o decode -- python $R/bench.py --steps 64 --no-cpu --no-graph --no-north-star --no-lock-step > $R/$OUT/prof_run.log 2>&1
  echo "rocprof exit $?" | tee -a $R/$OUT/summary.txt
  cd $R
  db=$(find $OUT/prof -name "*results.db" | head -1)
  [ -n "$db" ] && python scripts/prof_summary.py $db $OUT/decode_7b_q4_0 > /dev/null 2>&1 && head -12 $OUT/decode_7b_q4_0_kernel_stats.md | cut -c1-170
  find $OUT/prof -name "*.db" -delete
fi
if has pmc; then
  mkdir -p $OUT/pmc
  cd /tmp
  for c in FETCH_SIZE WRITE_SIZE; do
    n=$(echo $c | tr 'A-Z' 'a-z' | sed 's/_size//')
    timeout 900 rocprofv3 --pmc $c --kernel-trace -d $R/$OUT/pmc -o $n -- python $R/bench.py --steps 8 --warmup 2 --no-cpu --no-graph --no-north-star --no-lock-step > $R/$OUT/pmc/$n.log 2>&1
    echo "$c rc=$?"
  done
  cd $R
  fdb=$(find $OUT/pmc -name "fetch*results.db" | head -1); wdb=$(find $OUT/pmc -name "write*results.db" | head -1)
  python scripts/pmc_summary.py $fdb $wdb $OUT/pmc_traffic.json > $OUT/pmc_summary.log 2>&1; tail -5 $OUT/pmc_summary.log
  find $OUT/pmc -name "*.db" -delete
fi
cat $OUT/summary.txt
