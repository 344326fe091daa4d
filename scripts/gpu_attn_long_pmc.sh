#!/bin/bash
# round 6: the prefill attention of a 512-token batch at 8192 keys (BASELINE config 5's batches), scratch form (32) against k_attention_flash's LONG form (0):
# time per launch (kernel trace) and HBM traffic (FETCH_SIZE / WRITE_SIZE, one pass each) -> profiles/<tag>_attn_8k_*
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; export TMPDIR=/tmp PYTHONUNBUFFERED=1
T=${1:-r06}; mkdir -p gpurun_out/$T
export HEADS=${HEADS:-71} KV_HEADS=${KV_HEADS:-1} NPAST=${NPAST:-7680} FORMS=32,0
cd /tmp
timeout 300 python $R/scripts/gpu_attn_forms.py 512 > $R/gpurun_out/$T/attn_8k_time.txt 2>&1
FQ_ATTN_SCRATCH_GB=16 NPAST=0 timeout 300 python $R/scripts/gpu_attn_forms.py 4096 >> $R/gpurun_out/$T/attn_8k_time.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$T/trace -o t -- python $R/scripts/gpu_attn_forms.py 512 > $R/gpurun_out/$T/trace.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  n=$(echo $c | tr 'A-Z' 'a-z' | sed 's/_size//')
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d $R/gpurun_out/$T/pmc -o $n -- python $R/scripts/gpu_attn_forms.py 512 > $R/gpurun_out/$T/pmc_$n.log 2>&1
done
cd $R
cat gpurun_out/$T/attn_8k_time.txt
f=$(find gpurun_out/$T/pmc -name "fetch_results.db" | head -1); w=$(find gpurun_out/$T/pmc -name "write_results.db" | head -1)
[ -n "$f" ] && [ -n "$w" ] && timeout 120 python scripts/pmc_summary.py $f $w gpurun_out/$T/attn_8k_pmc_traffic.json | grep -E "attention|pack"
t=$(find gpurun_out/$T/trace -name "t_results.db" | head -1)
[ -n "$t" ] && timeout 120 python scripts/prof_summary.py $t gpurun_out/$T/attn_8k | grep -E "attention|pack" | head -6
find gpurun_out/$T -name "*.db" -size +20M -delete
