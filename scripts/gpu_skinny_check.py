"""tuning aid (GPU): the streaming small-batch mat-mul against the tile GEMM with the same K split, bit for bit, over K / N / format"""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ggllm_cpp_amd as g
from ggllm_cpp_amd import synth
g.init(0); L = g.load()
rng = np.random.default_rng(1)
bad = 0
for t in (g.Q4_0, g.Q4_1, g.Q5_0, g.Q5_1, g.Q8_0):
    for K in (512, 1024, 1056, 2048, 2080, 3072, 4544, 18176):
        for M, N in ((64, 16), (70, 9)):
            blocks = synth.random_blocks(t, M, K, rng)
            w = g.Weight(t, blocks, K, M)
            x = rng.standard_normal((N, K)).astype(np.float32)
            for seq in (0, 1):
                L.ggml_hip_gemm_sequential(seq)
                os.environ.pop("FQ_GEMM_CFG", None)
                a = w.mul_mat(x)
                os.environ["FQ_GEMM_CFG"] = "0" if seq else "2"
                b = w.mul_mat(x)
                os.environ.pop("FQ_GEMM_CFG", None)
                if not np.array_equal(a.view(np.uint32), b.view(np.uint32)):
                    bad += 1
                    d = np.nan_to_num(np.abs(a - b), nan=1e9); nz = np.argwhere(a.view(np.uint32) != b.view(np.uint32))
                    print("MISMATCH type %d K %d M %d N %d S %d: %d of %d differ, max %.3e; first at (n, m) = %s; rows differing: %s" %
                          (t, K, M, N, 1 if seq else 4, len(nz), a.size, d.max(), nz[0], sorted(set(nz[:, 1].tolist()))[:12]))
            L.ggml_hip_gemm_sequential(0)
            w.free()
print("mismatches:", bad)
