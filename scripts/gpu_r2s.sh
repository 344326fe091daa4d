#!/bin/bash
# A/B of the prefill GEMM's scaling epilogue: packed f32 operations (default build) vs scalar (libggml_hip_np.so)
mkdir -p gpurun_out/r2s
cd /root/repo
python -m pytest tests/test_gpu_mul_mat.py tests/test_gpu_falcon.py -x -q > gpurun_out/r2s/tests.log 2>&1; tail -3 gpurun_out/r2s/tests.log
for lib in libggml_hip.so libggml_hip_np.so libggml_hip.so libggml_hip_np.so; do
  GGLLM_HIP_LIB=/root/repo/ggllm.cpp_amd/$lib python bench.py --no-cpu --steps 16 --repeats 1 --no-north-star --no-lock-step > gpurun_out/r2s/bench_$lib.json 2> gpurun_out/r2s/bench_$lib.err
  python -c "
import json
d=json.loads(open('gpurun_out/r2s/bench_$lib.json').read().strip().splitlines()[-1]); print('$lib', 'prefill128 %.2f ms (%.0f tok/s)  prefill2048 %.1f ms (%.0f tok/s)' % (d['prefill_ms'], d['prefill_tok_s'], d['prefill_roofline']['long']['ms'], d['prefill_roofline']['long']['tok_s']))"
done
