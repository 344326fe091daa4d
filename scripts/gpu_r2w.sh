#!/bin/bash
mkdir -p gpurun_out/r2w
cd /root/repo
for b in 5 8 12; do
  timeout 600 python bench.py --force-pipeline --streams 2 --pipe-batch $b --steps 32 --warmup 4 > gpurun_out/r2w/pipe_7b_b$b.json 2> gpurun_out/r2w/pipe_7b_b$b.err
  python -c "
import json
d=json.loads(open('gpurun_out/r2w/pipe_7b_b$b.json').read().strip().splitlines()[-1]); print('b=$b', round(d['value'],1), 'tok/s', round(d['ms_per_step']/2,3), 'ms/pass')" || tail -3 gpurun_out/r2w/pipe_7b_b$b.err
done
