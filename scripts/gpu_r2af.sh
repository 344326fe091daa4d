#!/bin/bash
mkdir -p gpurun_out/r2af
cd /root/repo
export PYTHONUNBUFFERED=1
python -m pytest tests/test_gpu_mul_mat.py tests/test_gpu_falcon.py tests/test_gpu_configs.py tests/test_gpu_pipeline.py -x -q > gpurun_out/r2af/tests.log 2>&1; tail -6 gpurun_out/r2af/tests.log
for n in 256 0 256 0; do
  echo "FQ_GEMM_STREAM_N=$n"; FQ_GEMM_STREAM_N=$n python scripts/gpu_par2_ab.py 2>&1 | grep "N=" | cut -c1-64
done
