"""time the small-batch mat-mul for the Falcon-7B shapes (tuning aid): python scripts/gpu_skinny_time.py [N ...]   (FQ_GEMM_SKINNY=0: the tile GEMM)"""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ggllm_cpp_amd as g
from ggllm_cpp_amd import synth
g.init(0); L = g.load()
Ns = [int(a) for a in sys.argv[1:]] or [8, 16]
import ctypes as C
L.ggml_hip_debug_gemm_mode.argtypes = [C.c_int]
L.ggml_hip_debug_gemm_mode(int(os.environ.get("SK_DBG", "0")))
rng = np.random.default_rng(0)
tot = {n: 0.0 for n in Ns}
for name, K, M in (("qkv", 4544, 4672), ("wo", 4544, 4544), ("up", 4544, 18176), ("down", 18176, 4544), ("head", 4544, 65024)):
    blocks = synth.random_blocks(g.Q4_0, M, K, rng)
    w = g.Weight(g.Q4_0, blocks, K, M)
    for N in Ns:
        x = rng.standard_normal((N, K)).astype(np.float32)
        xb, yb = g.DevBuf(host=x), g.DevBuf(N * M * 4)
        for _ in range(3): L.ggml_hip_mul_mat_q(w.h, xb.ptr, K, N, yb.ptr, M)
        e0, e1 = L.ggml_hip_event_create(), L.ggml_hip_event_create()
        L.ggml_hip_event_record(e0)
        for _ in range(20): L.ggml_hip_mul_mat_q(w.h, xb.ptr, K, N, yb.ptr, M)
        L.ggml_hip_event_record(e1); L.ggml_hip_synchronize()
        us = L.ggml_hip_event_elapsed_ms(e0, e1) * 50
        mb = M * K * 18 / 32 / 1e6
        print("%-5s K=%5d M=%5d N=%2d: %8.1f us incl. the activation quantizer  (weights %.1f MB -> %.2f TB/s)" % (name, K, M, N, us, mb, mb / us))
        if name != "head": tot[N] += us
        xb.free(); yb.free()
    w.free()
for N in Ns:
    print("block total N=%d: %.1f us -> 32 blocks %.2f ms" % (N, tot[N], tot[N] * 32 / 1000))
