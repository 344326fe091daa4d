"""Phase stamps of k_gemm_q's K pipeline (a -DGQ_STAMPS=1 -DGQ_PAIR=0 tuning build of kernels_gemm.hip -- the stamps sit in the one-stage-per-barrier loop --, GGLLM_HIP_LIB=...): per wave of workgroups 0 and 100, stages 8..23 of one
launch -- when the wave passed the top of a stage, finished issuing its loads, finished the stage's arithmetic, finished writing the next stage into LDS (then: barrier).
python scripts/gpu_gemm_stamps.py [N] [K] [M]"""
import sys, os, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ggllm_cpp_amd as g
from ggllm_cpp_amd import synth
g.init(0); L = g.load()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
K = int(sys.argv[2]) if len(sys.argv) > 2 else 4544
M = int(sys.argv[3]) if len(sys.argv) > 3 else 4544
rng = np.random.default_rng(0)
w = g.Weight(g.Q4_0, synth.random_blocks(g.Q4_0, M, K, rng), K, M)
x = rng.standard_normal((N, K)).astype(np.float32)
xb, yb = g.DevBuf(host=x), g.DevBuf(N * M * 4)
L.ggml_hip_debug_stamps.argtypes = [C.c_int, C.c_void_p]
for _ in range(3): L.ggml_hip_mul_mat_q(w.h, xb.ptr, K, N, yb.ptr, M)
L.ggml_hip_debug_stamps(1, None)
L.ggml_hip_mul_mat_q(w.h, xb.ptr, K, N, yb.ptr, M)
buf = np.zeros(2 * 4096 * 8, np.int64)
L.ggml_hip_debug_stamps(1, buf.ctypes.data)
t = buf[:2 * 16 * 8 * 8].reshape(2, 16, 8, 8).astype(np.float64) * 10.0       # ns (100 MHz clock)
names = ["top", "issued", "computed", "committed", "top'", "issued'", "computed'", "committed'"]
for wg in (0, 1):
    live = [wv for wv in range(16) if t[wg, wv].any()]
    if not live: continue
    t0 = min(t[wg, wv, 0, 0] for wv in live)
    print("workgroup %d (%d waves): ns since the first wave's top of stage 8; per wave the 8 phases of stage pairs (8,9) (10,11) ..." % (0 if wg == 0 else 100, len(live)))
    for it in range(4):
        print("  stage pair %d:" % it)
        for wv in live:
            print("    wave %2d: " % wv + "  ".join("%s %6.0f" % (names[s][:6], t[wg, wv, it, s] - t0) for s in range(8)))
    # summary: per phase, mean duration over waves and iterations
    d = np.diff(t[wg, live][:, :, :], axis=2)           # [wave][it][7]
    print("  mean ns per phase (issue, compute, commit, barrier, issue', compute', commit'): " + "  ".join("%.0f" % v for v in d.mean(axis=(0, 1))))
    for wv in live: print("    wave %2d: " % wv + "  ".join("%5.0f" % v for v in d[live.index(wv)].mean(axis=0)) + "   | pair period %.0f ns" % ((t[wg, wv, 7, 0] - t[wg, wv, 0, 0]) / 7))
