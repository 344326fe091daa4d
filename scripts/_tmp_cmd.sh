cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp PYTHONUNBUFFERED=1; mkdir -p gpurun_out/r06w
for rep in 1 2; do for v in 1 0; do echo "FQ_GEMM_SEQ16=$v rep $rep" | tee -a gpurun_out/r06w/ab_seq16_lengths.txt; PROMPT_ORDER=2 FQ_GEMM_SEQ16=$v timeout 600 python scripts/gpu_prompt_lengths.py 40 64 128 256 512 2048 2>&1 | grep prompt | tee -a gpurun_out/r06w/ab_seq16_lengths.txt; done; done
