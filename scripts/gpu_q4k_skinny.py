"""tuning aid (GPU): the Q4_K small-batch mat-mul on Falcon-40B shapes: python scripts/gpu_q4k_skinny.py [N ...]   (env FQ_SKINNY_Q4K=0: the tile GEMM;
FQ_KQ_T / FQ_KQ_NBW / FQ_KQ_SEG: launch shape; FQ_DBG: fq_gemm_debug_mode bits 4 no column DMA, 8 no weight DMA, 16 no arithmetic)"""
import sys, os, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ggllm_cpp_amd as g
from ggllm_cpp_amd import synth
g.init(0); L = g.load()
Ns = [int(a) for a in sys.argv[1:]] or [16]
L.ggml_hip_debug_gemm_mode.argtypes = [C.c_int]
rng = np.random.default_rng(0)
t = g.Q4_K
for name, K, M in (("qkv", 8192, 9216), ("wo", 8192, 8192), ("up", 8192, 32768), ("down", 32768, 8192)):
    blocks = synth.random_blocks(t, M, K, rng)
    w = g.Weight(t, blocks, K, M)
    for N in Ns:
        x = rng.standard_normal((N, K)).astype(np.float32)
        xb, yb = g.DevBuf(host=x), g.DevBuf(N * M * 4)
        L.ggml_hip_debug_gemm_mode(int(os.environ.get("FQ_DBG", "0")))
        for _ in range(3): L.ggml_hip_mul_mat_q(w.h, xb.ptr, K, N, yb.ptr, M)
        e0, e1 = L.ggml_hip_event_create(), L.ggml_hip_event_create()
        L.ggml_hip_event_record(e0)
        for _ in range(20): L.ggml_hip_mul_mat_q(w.h, xb.ptr, K, N, yb.ptr, M)
        L.ggml_hip_event_record(e1); L.ggml_hip_synchronize()
        us = L.ggml_hip_event_elapsed_ms(e0, e1) * 50
        L.ggml_hip_debug_gemm_mode(0)
        print("q4_k %-5s N=%3d %9.1f us (with the column quantizer)  %6.2f TB/s" % (name, N, us, M * K * 0.5625 / us / 1e6), flush=True)
        xb.free(); yb.free()
    w.free()
