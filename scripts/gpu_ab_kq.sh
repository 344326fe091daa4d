#!/bin/bash
# A/B/A/B of two builds of the library on the k-quant single-stream decode (Falcon-40B shapes, 60 blocks, ring forms on) inside ONE gpurun call
# usage: scripts/gpu_ab_kq.sh <libA.so> <libB.so> <tag> [formats ...]     -> gpurun_out/<tag>/ab_kq.txt
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
A=$1; B=$2; T=$3; shift 3; FMTS=${*:-q2_k q3_k q4_k}
mkdir -p gpurun_out/$T
BA="--model 40b --no-cpu --no-ref-order --no-other-order --no-cli --no-lock-step --no-north-star --prefill-long 0 --steps 32 --warmup 4 --repeats 3"
for q in $FMTS; do for rep in 1 2; do for lib in $A $B; do
  v=$(GGLLM_HIP_LIB=$PWD/ggllm.cpp_amd/$lib timeout 600 python bench.py $BA --order 0 --quant $q 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f tok/s  %.3f ms  step_frac %.3f  launch_frac %.3f' % (d['value'], d['ms_per_step'], d['roofline']['step_frac'], d['roofline']['frac'] or 0))")
  echo "40b $q $lib rep $rep: $v" | tee -a gpurun_out/$T/ab_kq.txt
done; done; done
