#!/bin/bash
mkdir -p gpurun_out/r2ab
cd /root/repo
export FALCON_PIPE_SAME_DEVICE=1
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 2 --model tiny --steps 4 --warmup 2 > gpurun_out/r2ab/bench2.log 2>&1
echo "rc=$?"; grep -v "alt_rsmi\|^$\|iommu\|^W0\|socket.cpp\|amdgpu.ids" gpurun_out/r2ab/bench2.log | head -20 | cut -c1-220
