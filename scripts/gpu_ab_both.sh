#!/bin/bash
# A/B/A/B of two builds of the library inside ONE gpurun call: decode tok/s in the default order and in the fast reference order (graph replay) + the ring launch's duration
# usage: scripts/gpu_ab_both.sh <libA.so> <libB.so> [tag]
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
A=$1; B=$2; T=${3:-abboth}; mkdir -p gpurun_out/$T
for rep in 1 2; do for lib in $A $B; do
  echo -n "$lib rep $rep: " | tee -a gpurun_out/$T/ab.txt
  GGLLM_HIP_LIB=$PWD/ggllm.cpp_amd/$lib timeout 300 python scripts/gpu_ref_fast_ab.py ${AB_ARGS:-} 2>&1 | grep -E "^rep 1 mode|^mode [02]: " | sed -e 's/prefill128 [0-9.]* ms ([0-9]* tok.s), //' | tr '\n' ' ' | tee -a gpurun_out/$T/ab.txt; echo | tee -a gpurun_out/$T/ab.txt
done; done
