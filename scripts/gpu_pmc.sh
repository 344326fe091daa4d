#!/bin/bash
# HBM traffic of the decode kernels from the L2's memory-side counters: one rocprofv3 --pmc pass per counter (FETCH_SIZE and
# WRITE_SIZE do not fit one pass), kernel trace only. Output: gpurun_out/pmc/{fetch,write}_results.db
# usage: scripts/gpu_pmc.sh [steps]
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  n=$(echo $c | tr 'A-Z' 'a-z' | sed 's/_size//')
  timeout 900 rocprofv3 --pmc $c --kernel-trace -d $R/gpurun_out/pmc -o $n -- python $R/bench.py --steps ${1:-8} --warmup 2 --no-cpu --no-ref-order --no-graph > $R/gpurun_out/pmc/$n.log 2>&1
  echo "$c rc=$?"
done
# matrix-pipe evidence for the prefill GEMM (one pass, SQ block only)
timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_I8 SQ_INSTS_VALU SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace -d $R/gpurun_out/pmc -o mfma -- python $R/bench.py --steps 2 --warmup 1 --no-cpu --no-ref-order --no-graph > $R/gpurun_out/pmc/mfma.log 2>&1
echo "MFMA pass rc=$?"
cd $R
find gpurun_out/pmc -type f | head
