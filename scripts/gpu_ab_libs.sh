#!/bin/bash
# A/B/A/B of two builds of the library inside ONE gpurun call (boxes of the pool differ by +-1 %): decode tok/s + 128- / 2048-token prompts
# usage: scripts/gpu_ab_libs.sh <libA.so> <libB.so> [tag]
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
A=$1; B=$2; T=${3:-ab}; mkdir -p gpurun_out/$T
for rep in 1 2; do for lib in $A $B; do
  GGLLM_HIP_LIB=$PWD/ggllm.cpp_amd/$lib timeout 600 python bench.py --steps 128 --repeats 3 --no-cpu --no-ref-order --no-north-star --no-lock-step --no-cli 2> /dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib rep $rep: decode %.1f tok/s | 128-token prompt %.2f ms | 2048 tokens %.2f ms' % (d['value'], d['prefill_ms'], d['prefill_roofline']['long']['ms']))" | tee -a gpurun_out/$T/ab.txt
done; done
