#!/bin/bash
# two-branch prefill blocks (attention chain || MLP chain on two streams): parity + prompt timings, A/B by FALCON_HIP_PAR2_MAX_N
mkdir -p gpurun_out/r2ac
cd /root/repo
export PYTHONUNBUFFERED=1
python -m pytest tests/test_gpu_falcon.py tests/test_gpu_configs.py -x -q > gpurun_out/r2ac/tests.log 2>&1; tail -4 gpurun_out/r2ac/tests.log
for m in 512 0 512 0; do
  for P in 128 512; do
  FALCON_HIP_PAR2_MAX_N=$m python bench.py --prompt $P --no-cpu --steps 8 --repeats 1 --no-north-star --no-lock-step --prefill-long 0 > gpurun_out/r2ac/bench_$m_$P.json 2> gpurun_out/r2ac/bench_$m_$P.err
  python -c "
import json
d=json.loads(open('gpurun_out/r2ac/bench_$m_$P.json').read().strip().splitlines()[-1]); print('PAR2_MAX_N=$m prompt $P: %.2f ms (%.0f tok/s)' % (d['prefill_ms'], d['prefill_tok_s']))"
  done
done
