"""A/B aid: time the prefill GEMM of the k-quant formats on Falcon-40B shapes with the library named by FQ_AB_LIB.
python scripts/gpu_kgemm_ab.py [N ...]"""
import sys, os, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ggllm_cpp_amd as g
from ggllm_cpp_amd import synth
if os.environ.get("FQ_AB_LIB"):
    g.LIB_PATH = os.environ["FQ_AB_LIB"]
g.init(0); L = g.load()
Ns = [int(a) for a in sys.argv[1:]] or [128, 512]
MODES = [int(m) for m in os.environ.get("FQ_AB_MODES", "0").split(",")]      # ggml_hip_debug_gemm_mode: 1 no loads, 2 no math
L.ggml_hip_debug_gemm_mode.argtypes = [C.c_int]
rng = np.random.default_rng(0)
for t, tn in ((g.Q2_K, "q2_k"), (g.Q3_K, "q3_k"), (g.Q4_K, "q4_k"), (g.Q5_K, "q5_k"), (g.Q6_K, "q6_k"), (g.Q4_0, "q4_0")):
    for name, K, M in (("up", 8192, 32768), ("down", 32768, 8192)):
        blocks = synth.random_blocks(t, M, K, rng)
        w = g.Weight(t, blocks, K, M)
        for N in Ns:
            x = rng.standard_normal((N, K)).astype(np.float32)
            xb, yb = g.DevBuf(host=x), g.DevBuf(N * M * 4)
            for mode in MODES:
                L.ggml_hip_debug_gemm_mode(mode)
                for _ in range(3): L.ggml_hip_mul_mat_q(w.h, xb.ptr, K, N, yb.ptr, M)
                e0, e1 = L.ggml_hip_event_create(), L.ggml_hip_event_create()
                L.ggml_hip_event_record(e0)
                for _ in range(10): L.ggml_hip_mul_mat_q(w.h, xb.ptr, K, N, yb.ptr, M)
                L.ggml_hip_event_record(e1); L.ggml_hip_synchronize()
                us = L.ggml_hip_event_elapsed_ms(e0, e1) * 100
                print("%s %-5s N=%4d mode %d %9.1f us  %6.1f TOP/s" % (tn, name, N, mode, us, 2.0 * M * K * N / us / 1e6), flush=True)
            L.ggml_hip_debug_gemm_mode(0)
            xb.free(); yb.free()
        w.free()
