#!/bin/bash
# two ranks on ONE GPU: how far does the RCCL transport get? (unique id over gloo, dlopen, ncclCommInitRank)
mkdir -p gpurun_out/r2x
cd /root/repo
export FALCON_PIPE_SAME_DEVICE=1 NCCL_DEBUG=WARN
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 scripts/rccl_pipeline_smoke.py > gpurun_out/r2x/smoke.log 2>&1
echo "smoke rc=$?"; tail -25 gpurun_out/r2x/smoke.log
