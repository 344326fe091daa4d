#!/bin/bash
# A/B/A/B of environment settings of ONE build inside one gpurun call, decode only (the bench's headline): tok/s of the Falcon-7B Q4_0 greedy decode
# usage: scripts/gpu_ab_decode.sh <tag> "<ENV=.. settings A>" "<settings B>" ...     ("-" = no setting)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
T=$1; shift; mkdir -p gpurun_out/$T
for rep in 1 2; do for cfg in "$@"; do
  [ "$cfg" = "-" ] && e="" || e="$cfg"
  env $e timeout 600 python bench.py --steps 256 --repeats 3 --no-cpu --no-ref-order --no-north-star --no-lock-step --no-cli --prefill-long 0 2> /dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('[$cfg] rep $rep: decode %.1f tok/s' % d['value'])" | tee -a gpurun_out/$T/ab_decode.txt
done; done
