#!/bin/bash
mkdir -p gpurun_out/r2v
cd /root/repo
for lib in libggml_hip.so libggml_hip_fma.so libggml_hip.so libggml_hip_fma.so; do
  GGLLM_HIP_LIB=/root/repo/ggllm.cpp_amd/$lib python bench.py --no-cpu --steps 16 --repeats 1 --no-north-star --no-lock-step > gpurun_out/r2v/bench_$lib.json 2> gpurun_out/r2v/bench_$lib.err
  python -c "
import json
d=json.loads(open('gpurun_out/r2v/bench_$lib.json').read().strip().splitlines()[-1]); print('$lib', 'prefill128 %.2f ms (%.0f tok/s)  prefill2048 %.1f ms (%.0f tok/s)' % (d['prefill_ms'], d['prefill_tok_s'], d['prefill_roofline']['long']['ms'], d['prefill_roofline']['long']['tok_s']))"
done
