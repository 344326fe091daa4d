#!/bin/bash
# round 2, call q: lock-step pipeline (one GPU) after the batched decode attention + where a batched pass spends its time
mkdir -p gpurun_out/r2q
cd /root/repo
python -m pytest tests/test_gpu_pipeline.py -x -q > gpurun_out/r2q/test_pipeline.log 2>&1
tail -3 gpurun_out/r2q/test_pipeline.log
for b in 2 4; do
  timeout 600 python bench.py --force-pipeline --streams 2 --pipe-batch $b --steps 64 --warmup 8 > gpurun_out/r2q/pipe_7b_b$b.json 2> gpurun_out/r2q/pipe_7b_b$b.err
  python -c "
import json
d=json.loads(open('gpurun_out/r2q/pipe_7b_b$b.json').read().strip().splitlines()[-1]); print('b=$b', round(d['value'],1), 'tok/s', round(d['ms_per_step'],3), 'ms/round')"
done
cd /tmp && export TMPDIR=/tmp
FALCON_HIP_STAGE_GRAPH=0 timeout 900 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r2q/prof -o pipe_b4 -- python /root/repo/bench.py --force-pipeline --streams 2 --pipe-batch 4 --steps 16 --warmup 4 > /root/repo/gpurun_out/r2q/prof_run.log 2>&1
cd /root/repo
db=$(find gpurun_out/r2q/prof -name "*results.db" | head -1)
python scripts/prof_summary.py $db gpurun_out/r2q/pipe_b4 > /dev/null 2>&1
head -16 gpurun_out/r2q/pipe_b4_kernel_stats.md | cut -c1-200
find gpurun_out/r2q/prof -name "*.db" -delete
