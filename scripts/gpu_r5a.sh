#!/bin/bash
# round 5: the new prefill attention form -- parity with the scratch form, per-launch time, prompt times
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
T=${1:-r05a}
mkdir -p gpurun_out/$T
python scripts/gpu_exp_boundary.py > gpurun_out/$T/fq_exp_fix.h 2> gpurun_out/$T/exp_boundary.err; head -4 gpurun_out/$T/fq_exp_fix.h | cut -c1-200; tail -2 gpurun_out/$T/exp_boundary.err
python -c "import ggllm_cpp_amd as g; g.init(0); print('exp formula mismatches', g.load().ggml_hip_exp_formula_mismatches())"
SEEDS=0,1,2 python scripts/gpu_attn_bisect.py 2>&1 | tee gpurun_out/$T/bisect.txt
FORMS=32,1 timeout 300 python scripts/gpu_attn_forms.py 2048 1024 512 256 128 2>&1 | tee gpurun_out/$T/attn_forms.txt
FQ_ATTN_PACK_MIN_N=100000 FORMS=32,1 timeout 300 python scripts/gpu_attn_forms.py 2048 2>&1 | tee gpurun_out/$T/attn_forms_nopack.txt
timeout 600 python -m pytest tests/test_gpu_block_ops.py -m gpu -q --no-header -p no:cacheprovider -x -k "attention or softmax or exp" 2>&1 | tail -5 | tee gpurun_out/$T/pytest_attention.log
