timeout 900 python -m pytest tests/test_gpu_mul_mat.py tests/test_gpu_falcon.py -m gpu -x -q -k "q4k_small or 40b_shaped or 180b_shaped or prefill_gemm_vs_oracle" 2>&1 | tail -5
python scripts/gpu_q4k_skinny.py 16 2>&1 | tail -4
LOCKSTEP_MODEL=40b_q4_k LOCKSTEP_LAYERS=12 python scripts/gpu_lockstep.py 8 16 32 2>&1 | tail -3
