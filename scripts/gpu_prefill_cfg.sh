#!/bin/bash
# the 128-token prompt under the tile GEMM's shapes (FQ_GEMM_CFG: 2 <4,4> = the default there, 1 <4,1>, 4 <4,2>, 5 <2,2>, 3 <2,4>), one box
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
for rep in 1 2; do for cfg in ${CFGS:-"" 1 4 5}; do
  echo -n "rep $rep FQ_GEMM_CFG=${cfg:-default}: "
  env ${cfg:+FQ_GEMM_CFG=$cfg} ${EXTRA_ENV:-} timeout 300 python bench.py --no-cpu --no-ref-order --no-north-star --no-lock-step --no-cli --prefill-long 0 --steps 8 --warmup 2 --repeats 1 --order 0 --prompt ${PROMPT:-128} 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('prompt %.3f ms (%.0f tok/s)' % (d['prefill_ms'], d['prefill_tok_s']))"
done; done
