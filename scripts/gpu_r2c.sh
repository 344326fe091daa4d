#!/bin/bash
# round 2, call C: first run of the persistent decode engine
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r2c
mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_engine.py -m gpu -v --no-header -p no:cacheprovider -s > $OUT/test_engine.log 2>&1; echo "engine tests exit $?" | tee -a $OUT/summary.txt
tail -25 $OUT/test_engine.log
FALCON_HIP_ENGINE=1 timeout 300 python bench.py --no-cpu --prefill-long 0 > $OUT/bench_engine.json 2> $OUT/bench_engine.err; echo "bench engine exit $?" | tee -a $OUT/summary.txt
tail -c 1800 $OUT/bench_engine.json; tail -5 $OUT/bench_engine.err
timeout 300 python bench.py --no-cpu --prefill-long 0 > $OUT/bench_base.json 2> $OUT/bench_base.err; echo "bench base exit $?" | tee -a $OUT/summary.txt
python - <<'PY'
import json
for n in ("engine", "base"):
    try:
        d = json.loads(open(f"gpurun_out/r2c/bench_{n}.json").read().strip().splitlines()[-1])
        print(n, "tok/s %.1f" % d["value"], "ms/step %.4f" % d["ms_per_step"], "roofline", d["roofline"].get("frac"), "step_frac %.4f" % d["roofline"]["step_frac"])
    except Exception as e:
        print(n, "no line:", e)
PY
true
