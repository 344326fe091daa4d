#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r2l
mkdir -p $OUT
timeout 200 python -m pytest tests/test_gpu_engine.py tests/test_gpu_falcon.py -m gpu -q --no-header -p no:cacheprovider 2>&1 | tail -4
timeout 300 python bench.py --no-cpu --prefill-long 0 > $OUT/bench_base.json 2> $OUT/bench_base.err; echo "bench base exit $?"
FALCON_HIP_ENGINE=1 timeout 300 python bench.py --no-cpu --prefill-long 0 > $OUT/bench_engine.json 2> $OUT/bench_engine.err; echo "bench engine exit $?"; tail -3 $OUT/bench_engine.err
python - <<'PY'
import json
for n in ("base", "engine"):
    try:
        d = json.loads(open(f"gpurun_out/r2l/bench_{n}.json").read().strip().splitlines()[-1])
        print(n, "tok/s %.1f" % d["value"], "ms/step %.4f" % d["ms_per_step"], "kernel roofline", d["roofline"].get("frac"), "step_frac %.4f" % d["roofline"]["step_frac"], "prefill tok/s %.0f" % d["prefill_tok_s"])
    except Exception as e:
        print(n, "no line:", e)
PY
FALCON_HIP_ENGINE_DEBUG_MODE=1 timeout 300 python scripts/gpu_engine_debug.py q4_0:7b:32 2>&1 | grep "steps\|loader"
timeout 300 python scripts/gpu_engine_debug.py q4_0:7b:32 > $OUT/engine_debug.log 2>&1; cat $OUT/engine_debug.log
