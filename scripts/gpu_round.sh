#!/bin/bash
# One gpurun call of a round: every -m gpu test, smoke, the bench line the driver reads; optionally rocprofv3 kernel trace + PMC passes.
# usage: scripts/gpu_round.sh <tag> [tests|bench|prof|pmc|mfma|prefill2048|prefill128|b40|lockstep|lockstep40|q4kpmc|all ...]
#   prof         rocprofv3 kernel trace of the decode bench             -> <tag>/decode_7b_q4_0_kernel_stats.{csv,md}
#   pmc          FETCH_SIZE / WRITE_SIZE passes (HBM traffic per launch)  -> <tag>/pmc_traffic.json
#   mfma         SQ matrix-pipe / VALU counters (decode + 128- and 2048-token prefill in one run) -> <tag>/pmc_mfma.json
#   prefill2048  kernel trace of a 2048-token prompt                     -> <tag>/prefill2048_7b_q4_0_kernel_stats.{csv,md}
#   prefill128   kernel trace of the bench's 128-token prompt            -> <tag>/prefill128_7b_q4_0_kernel_stats.{csv,md}
#   b40          Falcon-40B Q4_K, all 60 blocks, one GPU: bench line + kernel trace -> <tag>/bench_40b_q4_k.json, <tag>/decode_40b_q4_k_kernel_stats.{csv,md}
#   lockstep     kernel trace of 16 lock-step streams per weight pass    -> <tag>/lockstep_b16_kernel_stats.{csv,md}
#   q4kpmc       counters of the Q4_K small-batch launches on Falcon-40B shapes (16 columns): VALU / matrix pipe, HBM fetch / write -> <tag>/q4k_pmc_mfma.json, <tag>/q4k_pmc_traffic.json
#   kq           k-quant single-stream decode on Falcon-40B (60 blocks): tok/s per format, ring form on / off, + a kernel trace per format -> <tag>/kq_decode.txt, <tag>/decode_40b_<fmt>_kernel_stats.{csv,md}
#   kqpmc        counters of the k-quant decode launches (8 blocks): VALU busy + HBM fetch / write per format -> <tag>/kq_<fmt>_pmc_{mfma,traffic}.json
#   lockstep40   Falcon-40B Q4_K, 60 blocks: lock-step streams 4..128 per pass + kernel trace of 16 per pass (12 blocks) -> <tag>/lockstep_40b_q4_k.txt, <tag>/lockstep40_b16_kernel_stats.{csv,md}
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
  timeout 600 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof -o decode -- python $R/bench.py --steps 64 --no-cpu --no-ref-order --no-graph --no-north-star --no-lock-step > $R/$OUT/prof_run.log 2>&1
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
    timeout 900 rocprofv3 --pmc $c --kernel-trace -d $R/$OUT/pmc -o $n -- python $R/bench.py --steps 8 --warmup 2 --no-cpu --no-ref-order --no-graph --no-north-star --no-lock-step > $R/$OUT/pmc/$n.log 2>&1
    echo "$c rc=$?"
  done
  cd $R
  fdb=$(find $OUT/pmc -name "fetch*results.db" | head -1); wdb=$(find $OUT/pmc -name "write*results.db" | head -1)
  python scripts/pmc_summary.py $fdb $wdb $OUT/pmc_traffic.json > $OUT/pmc_summary.log 2>&1; tail -5 $OUT/pmc_summary.log
  find $OUT/pmc -name "*.db" -delete
fi
trace() {   # trace <name> <command...>: kernel trace -> $OUT/<name>_kernel_stats.{csv,md}
  local name=$1; shift
  cd /tmp
  timeout 900 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof_$name -o trace -- "$@" > $R/$OUT/prof_$name.log 2>&1
  echo "rocprof $name exit $?" | tee -a $R/$OUT/summary.txt
  cd $R
  local db=$(find $OUT/prof_$name -name "*results.db" | head -1)
  [ -n "$db" ] && python scripts/prof_summary.py $db $OUT/$name > /dev/null 2>&1 && head -10 $OUT/${name}_kernel_stats.md | cut -c1-170
  rm -rf $OUT/prof_$name
}
if has mfma; then
  mkdir -p $OUT/pmc
  cd /tmp
  timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_I8 SQ_INSTS_VALU SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace -d $R/$OUT/pmc -o mfma -- python $R/bench.py --steps 4 --warmup 1 --repeats 1 --no-cpu --no-ref-order --no-graph --no-north-star --no-lock-step --no-cli > $R/$OUT/pmc/mfma.log 2>&1
  echo "mfma counters exit $?" | tee -a $R/$OUT/summary.txt
  cd $R
  mdb=$(find $OUT/pmc -name "mfma*results.db" | head -1)
  python scripts/pmc_mfma_summary.py $mdb $OUT/pmc_mfma.json > $OUT/pmc_mfma_summary.log 2>&1; tail -6 $OUT/pmc_mfma_summary.log
  find $OUT/pmc -name "*.db" -delete
fi
if has prefill2048; then
  trace prefill2048_7b_q4_0 python $R/bench.py --prompt 2048 --n-ctx 4096 --steps 8 --warmup 2 --repeats 1 --no-cpu --no-ref-order --no-north-star --no-lock-step --no-cli --prefill-long 0
fi
if has prefill128; then
  trace prefill128_7b_q4_0 python $R/bench.py --prompt 128 --steps 8 --warmup 2 --repeats 1 --no-cpu --no-ref-order --no-north-star --no-lock-step --no-cli --prefill-long 0
fi
if has b40; then
  timeout 900 python bench.py --model 40b --quant q4_k --no-cpu --no-ref-order --no-cli --no-lock-step --no-north-star --prefill-long 0 --steps 32 --warmup 4 --repeats 1 > $OUT/bench_40b_q4_k.json 2> $OUT/bench_40b.err; echo "bench 40b exit $?" | tee -a $OUT/summary.txt
  python scripts/bench_brief.py < $OUT/bench_40b_q4_k.json
  trace decode_40b_q4_k python $R/bench.py --model 40b --quant q4_k --layers 12 --no-cpu --no-ref-order --no-cli --no-graph --no-lock-step --no-north-star --prefill-long 0 --steps 16 --warmup 2 --repeats 1
fi
if has lockstep; then
  FALCON_HIP_STAGE_GRAPH=0 trace lockstep_b16 python $R/bench.py --force-pipeline --streams 1 --pipe-batch 16 --steps 16 --warmup 4 --no-cpu --no-ref-order --no-cli
fi
if has lockstep40; then
  LOCKSTEP_MODEL=40b_q4_k timeout 600 python scripts/gpu_lockstep.py 1 2 4 8 12 16 32 48 64 80 128 2>&1 | grep "streams per pass" | tee $OUT/lockstep_40b_q4_k.txt
  LOCKSTEP_MODEL=40b_q4_k LOCKSTEP_LAYERS=12 trace lockstep40_b16 python $R/scripts/gpu_lockstep.py 16
fi
if has kq; then
  TRACE=1 scripts/gpu_kq_decode.sh $TAG q4_k q2_k q3_k q5_k q6_k
fi
if has kqpmc; then
  mkdir -p $OUT/pmc
  KB="--model 40b --layers 8 --no-cpu --no-ref-order --no-cli --no-lock-step --no-north-star --prefill-long 0 --no-graph --steps 8 --warmup 2 --repeats 1"
  for q in q4_k q2_k q3_k q5_k q6_k; do
    cd /tmp
    timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_I8 SQ_INSTS_VALU SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace -d $R/$OUT/pmc -o kq${q}mfma -- python $R/bench.py $KB --quant $q > $R/$OUT/pmc/kq${q}mfma.log 2>&1
    for c in FETCH_SIZE WRITE_SIZE; do
      n=kq${q}$(echo $c | tr 'A-Z' 'a-z' | sed 's/_size//')
      timeout 300 rocprofv3 --pmc $c --kernel-trace -d $R/$OUT/pmc -o $n -- python $R/bench.py $KB --quant $q > $R/$OUT/pmc/$n.log 2>&1
    done
    cd $R
    python scripts/pmc_mfma_summary.py $(find $OUT/pmc -name "kq${q}mfma*results.db" | head -1) $OUT/kq_${q}_pmc_mfma.json 2>&1 | grep -E "k_ring|k_gemv" | cut -c1-170
    python scripts/pmc_summary.py $(find $OUT/pmc -name "kq${q}fetch*results.db" | head -1) $(find $OUT/pmc -name "kq${q}write*results.db" | head -1) $OUT/kq_${q}_pmc_traffic.json 2>&1 | grep -E "k_ring|k_gemv" | cut -c1-170
    find $OUT/pmc -name "*.db" -delete
  done
fi
if has q4kpmc; then
  mkdir -p $OUT/pmc
  cd /tmp
  timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_I8 SQ_INSTS_VALU SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace -d $R/$OUT/pmc -o q4kmfma -- python $R/scripts/gpu_q4k_skinny.py 16 > $R/$OUT/pmc/q4kmfma.log 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do
    n=q4k$(echo $c | tr 'A-Z' 'a-z' | sed 's/_size//')
    timeout 300 rocprofv3 --pmc $c --kernel-trace -d $R/$OUT/pmc -o $n -- python $R/scripts/gpu_q4k_skinny.py 16 > $R/$OUT/pmc/$n.log 2>&1
  done
  cd $R
  python scripts/pmc_mfma_summary.py $(find $OUT/pmc -name "q4kmfma*results.db" | head -1) $OUT/q4k_pmc_mfma.json | cut -c1-170
  python scripts/pmc_summary.py $(find $OUT/pmc -name "q4kfetch*results.db" | head -1) $(find $OUT/pmc -name "q4kwrite*results.db" | head -1) $OUT/q4k_pmc_traffic.json 2>&1 | tail -8
  find $OUT/pmc -name "*.db" -delete
fi
cat $OUT/summary.txt
