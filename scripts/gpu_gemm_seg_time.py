"""GEMM-only time (activations quantized beforehand) of a long-K mat-mul and of its K thirds / halves under the tile configurations a K split could use (tuning aid).
python scripts/gpu_gemm_seg_time.py [N]"""
import sys, os, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ggllm_cpp_amd as g
from ggllm_cpp_amd import synth
g.init(0); L = g.load()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
rng = np.random.default_rng(0)
M = 4544
for K in (18176, 9216, 6144, 4544):
    w = g.Weight(g.Q4_0, synth.random_blocks(g.Q4_0, M, K, rng), K, M)
    x = rng.standard_normal((N, K)).astype(np.float32)
    xb, yb = g.DevBuf(host=x), g.DevBuf(N * M * 4)
    a = L.ggml_hip_acts_alloc(g.Q8_0, K, N)
    L.ggml_hip_quantize_acts(a, xb.ptr, K, N)
    for cfg in ("2", "7", "6", "3"):
        os.environ["FQ_GEMM_CFG"] = cfg
        for _ in range(3): L.ggml_hip_mul_mat_q_acts(w.h, a, N, yb.ptr, M, 0, None, None)
        e0, e1 = L.ggml_hip_event_create(), L.ggml_hip_event_create()
        L.ggml_hip_event_record(e0)
        for _ in range(20): L.ggml_hip_mul_mat_q_acts(w.h, a, N, yb.ptr, M, 0, None, None)
        L.ggml_hip_event_record(e1); L.ggml_hip_synchronize()
        print("K=%5d M=%d N=%d cfg %s (%s): %7.1f us" % (K, M, N, cfg, {"default": "the launcher's choice (K segments where the rule gives them)", "2": "<4,4,1> 32 rows, 16 waves", "7": "<4,4,2> 64 rows, 16 waves", "6": "<2,4,2> 64 rows, 8 waves", "3": "<2,4,1> 32 rows, 8 waves"}[cfg], L.ggml_hip_event_elapsed_ms(e0, e1) * 50), flush=True)
    os.environ.pop("FQ_GEMM_CFG", None)
    L.ggml_hip_acts_free(a); w.free(); xb.free(); yb.free()
