"""time the prefill GEMM for the four Falcon-7B shapes (tuning aid): python scripts/gpu_gemm_time.py [N]"""
import sys, os, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ggllm_cpp_amd as g
from ggllm_cpp_amd import synth
g.init(0); L = g.load()
L.ggml_hip_debug_gemm_mode.argtypes = [C.c_int]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
rng = np.random.default_rng(0)
for name, K, M in (("qkv", 4544, 4672), ("wo", 4544, 4544), ("up", 4544, 18176), ("down", 18176, 4544)):
    blocks = synth.random_blocks(g.Q4_0, M, K, rng)
    w = g.Weight(g.Q4_0, blocks, K, M)
    x = rng.standard_normal((N, K)).astype(np.float32)
    xb, yb = g.DevBuf(host=x), g.DevBuf(N * M * 4)
    for mode in [int(m) for m in os.environ.get('MODES', '0,2').split(',')]:
        L.ggml_hip_debug_gemm_mode(mode)
        for _ in range(3): L.ggml_hip_mul_mat_q(w.h, xb.ptr, K, N, yb.ptr, M)
        e0, e1 = L.ggml_hip_event_create(), L.ggml_hip_event_create()
        L.ggml_hip_event_record(e0)
        for _ in range(10): L.ggml_hip_mul_mat_q(w.h, xb.ptr, K, N, yb.ptr, M)
        L.ggml_hip_event_record(e1); L.ggml_hip_synchronize()
        us = L.ggml_hip_event_elapsed_ms(e0, e1) * 100
        print("%-5s K=%5d M=%5d N=%d mode %d (%s): %8.1f us  (weights %.1f MB)" % (name, K, M, N, mode, {0: "full", 2: "no math", 8: "staggered K (timing only)", 10: "staggered, no math"}.get(mode, "?"), us, M * K * 18 / 32 / 1e6))
    L.ggml_hip_debug_gemm_mode(0)
    w.free(); xb.free(); yb.free()
