#!/bin/bash
mkdir -p gpurun_out/r2ae
cd /root/repo
export PYTHONUNBUFFERED=1
python -m pytest tests/test_gpu_mul_mat.py tests/test_gpu_falcon.py tests/test_gpu_configs.py -x -q > gpurun_out/r2ae/tests.log 2>&1; tail -4 gpurun_out/r2ae/tests.log
for lib in libggml_hip.so libggml_hip_p2.so libggml_hip.so libggml_hip_p2.so; do
  echo $lib; GGLLM_HIP_LIB=/root/repo/ggllm.cpp_amd/$lib python scripts/gpu_par2_ab.py 2>&1 | grep "N=" | cut -c1-64
done
