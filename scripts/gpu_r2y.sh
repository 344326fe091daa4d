#!/bin/bash
# prefill GEMM shape landscape: FQ_GEMM_CFG overrides (0 <1,4>, 1 <4,1>, 2 <4,4>, 3 <2,4>, 4 <4,2>, 5 <2,2>, 6 <2,4,2>, 7 <4,4,2>)
mkdir -p gpurun_out/r2y
cd /root/repo
for cfg in default 0 1 2 3 4 5 6 7; do
  if [ $cfg == default ]; then unset FQ_GEMM_CFG; else export FQ_GEMM_CFG=$cfg; fi
  python bench.py --no-cpu --steps 8 --repeats 1 --no-north-star --no-lock-step > gpurun_out/r2y/bench_$cfg.json 2> gpurun_out/r2y/bench_$cfg.err
  python -c "
import json
d=json.loads(open('gpurun_out/r2y/bench_$cfg.json').read().strip().splitlines()[-1]); print('cfg $cfg', 'prefill128 %.2f ms (%.0f tok/s)  prefill2048 %.1f ms (%.0f tok/s)' % (d['prefill_ms'], d['prefill_tok_s'], d['prefill_roofline']['long']['ms'], d['prefill_roofline']['long']['tok_s']))" || tail -2 gpurun_out/r2y/bench_$cfg.err
done
