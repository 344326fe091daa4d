#!/bin/bash
# A/B (one gpurun call): Falcon-40B-width decode, three launches per block against the merged attention + output launch run as a
# two-round grid with the attention workgroups first (FQ_ATTN_OUT_ROUNDS=2).  usage: scripts/gpu_ab_40b_rounds.sh [quant...]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for q in ${*:-q4_k q5_1}; do
  for r in 1 2 1 2; do
    FQ_ATTN_OUT_ROUNDS=$r timeout 600 python bench.py --model 40b --quant $q --layers 12 --no-cpu --no-ref-order --no-cli --no-lock-step --no-north-star --prefill-long 0 --steps 64 --warmup 8 --repeats 3 2>/dev/null \
      | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$q rounds=$r: %.1f us/token, %s' % (d['ms_per_step']*1e3, d['repeat_ms_per_step']))"
  done
done
