#!/bin/bash
# A/B/A/B of two builds of the library inside ONE gpurun call: the fast reference order's decode (ggml_hip_reference_order(2)), graph replay
# usage: scripts/gpu_ab_ref.sh <libA.so> <libB.so> [tag]
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
A=$1; B=$2; T=${3:-abref}; mkdir -p gpurun_out/$T
for rep in 1 2; do for lib in $A $B; do
  echo -n "$lib rep $rep: " | tee -a gpurun_out/$T/ab.txt
  GGLLM_HIP_LIB=$PWD/ggllm.cpp_amd/$lib timeout 300 python scripts/gpu_ref_fast_ab.py 2>&1 | grep -E "^rep 1 mode 2|mode 2: " | tr '\n' ' ' | tee -a gpurun_out/$T/ab.txt; echo | tee -a gpurun_out/$T/ab.txt
done; done
