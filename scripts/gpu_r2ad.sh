#!/bin/bash
mkdir -p gpurun_out/r2ad
cd /root/repo
export PYTHONUNBUFFERED=1
python -m pytest tests/test_gpu_mul_mat.py tests/test_gpu_falcon.py tests/test_gpu_configs.py tests/test_gpu_pipeline.py -x -q > gpurun_out/r2ad/tests.log 2>&1; tail -6 gpurun_out/r2ad/tests.log
for k in 1 0 1 0; do
  echo "FQ_GEMM_KHALF=$k"; FQ_GEMM_KHALF=$k python scripts/gpu_par2_ab.py 2>&1 | grep "N=  16\|N= 128\|N= 512" | cut -c1-70
done
