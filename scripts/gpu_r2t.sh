#!/bin/bash
# prefill attention on the f32 matrix pipe: parity tests + prompt timings (FQ_ATTN_MFMA=0: the previous kernel)
mkdir -p gpurun_out/r2t
cd /root/repo
export PYTHONUNBUFFERED=1
python -m pytest tests/test_gpu_block_ops.py tests/test_gpu_falcon.py tests/test_gpu_configs.py -x -q > gpurun_out/r2t/tests.log 2>&1; tail -15 gpurun_out/r2t/tests.log
for m in 1 0; do
  W=512; [ $m == 2 ] && W=1000000
  FQ_ATTN_MFMA_WIDE=$W FQ_ATTN_MFMA=$(( m > 0 ? 1 : 0 )) python bench.py --no-cpu --steps 16 --repeats 1 --no-north-star --no-lock-step > gpurun_out/r2t/bench_mfma$m.json 2> gpurun_out/r2t/bench_mfma$m.err
  python -c "
import json
d=json.loads(open('gpurun_out/r2t/bench_mfma$m.json').read().strip().splitlines()[-1]); print('FQ_ATTN_MFMA=$m', 'prefill128 %.2f ms (%.0f tok/s)  prefill2048 %.1f ms (%.0f tok/s)' % (d['prefill_ms'], d['prefill_tok_s'], d['prefill_roofline']['long']['ms'], d['prefill_roofline']['long']['tok_s']))"
done
