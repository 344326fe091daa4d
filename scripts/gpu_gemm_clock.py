"""What clock do the CUs of a tile-GEMM launch run at? (a -DGQ_STAMPS=2 tuning build of kernels_gemm.hip: shader cycles and 10 ns ticks around the K loop of two workgroups)
python scripts/gpu_gemm_clock.py"""
import sys, os, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ggllm_cpp_amd as g
from ggllm_cpp_amd import synth
g.init(0); L = g.load()
L.ggml_hip_debug_stamps.argtypes = [C.c_int, C.c_void_p]
rng = np.random.default_rng(0)
for name, K, M, N, cfg in (("Wdown 128 tokens", 18176, 4544, 128, "2"), ("Wdown 128 tokens, 64-row", 18176, 4544, 128, "7"), ("a third of Wdown's K, 64-row", 6144, 4544, 128, "7"),
                           ("Wup 128 tokens", 4544, 18176, 128, "3"), ("Wup 2048 tokens", 4544, 18176, 2048, "6"), ("Wdown 2048 tokens", 18176, 4544, 2048, "6")):
    w = g.Weight(g.Q4_0, synth.random_blocks(g.Q4_0, M, K, rng), K, M)
    x = rng.standard_normal((N, K)).astype(np.float32)
    xb, yb = g.DevBuf(host=x), g.DevBuf(N * M * 4)
    a = L.ggml_hip_acts_alloc(g.Q8_0, K, N)
    L.ggml_hip_quantize_acts(a, xb.ptr, K, N)
    os.environ["FQ_GEMM_CFG"] = cfg
    L.ggml_hip_debug_stamps(1, None)
    for _ in range(5): L.ggml_hip_mul_mat_q_acts(w.h, a, N, yb.ptr, M, 0, None, None)
    buf = np.zeros(2 * 4096 * 8, np.int64)
    L.ggml_hip_debug_stamps(1, buf.ctypes.data)
    for i in range(2):
        cyc, ticks = buf[8000 + 2 * i], buf[8001 + 2 * i]
        if ticks: print("%-30s cfg %s workgroup %2d: %9d shader cycles in %8.2f us = %6.0f MHz" % (name, cfg, 50 * i, cyc, ticks / 100.0, cyc / (ticks / 100.0)), flush=True)
    os.environ.pop("FQ_GEMM_CFG", None)
    L.ggml_hip_acts_free(a); w.free(); xb.free(); yb.free()
