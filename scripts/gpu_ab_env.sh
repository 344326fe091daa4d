#!/bin/bash
# A/B/A/B of environment settings of ONE build inside one gpurun call: decode tok/s + 128- / 2048-token prompts
# usage: scripts/gpu_ab_env.sh <tag> "<ENV=.. settings A>" "<settings B>" ...     ("-" = no setting)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
T=$1; shift; mkdir -p gpurun_out/$T
for rep in 1 2; do for cfg in "$@"; do
  [ "$cfg" = "-" ] && e="" || e="$cfg"
  env $e timeout 600 python bench.py --steps 128 --repeats 3 --no-cpu --no-ref-order --no-north-star --no-lock-step --no-cli 2> /dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('[$cfg] rep $rep: decode %.1f tok/s | 128-token prompt %.2f ms | 2048 tokens %.2f ms' % (d['value'], d['prefill_ms'], d['prefill_roofline']['long']['ms']))" | tee -a gpurun_out/$T/ab.txt
done; done
