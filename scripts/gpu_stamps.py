"""phase timeline of the fused decode kernels (tuning aid): python scripts/gpu_stamps.py [layers]
(k_gemv_ln: stamp 5 = the workgroup's first wave has streamed its rows, slot 6 = its LAST wave has)"""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ggllm_cpp_amd as g
from ggllm_cpp_amd import synth
g.init(0); L = g.load()
hp = dict(synth.HP_7B); hp["n_layer"] = int(sys.argv[1]) if len(sys.argv) > 1 else 4
w = synth.make_model_fast(hp, g.Q4_0)
m = g.FalconModel(w, n_ctx=512, n_batch=8)
toks = synth.tokens(8, hp["n_vocab"])
m.eval(toks, 0)
out = m.decode_greedy(1, 8, 150)            # warm, n_past ~ 158
L.ggml_hip_debug_stamps(1, None)
m.decode_greedy(int(out[-1]), 158, 4)
st = np.zeros(2 * 4096 * 8, np.int64)
L.ggml_hip_debug_stamps(1, st.ctypes.data)
st = st.reshape(2, 4096, 8)
for name, k, b0, nb in (("k_gemv_ln", 0, 0, 239), ("k_attn_out mat-vec role", 1, 0, 190), ("k_attn_out attention role", 1, 2048, 36)):
    s = st[k, b0:b0 + nb].astype(np.float64)
    t0 = s[:, 0].min()
    rel = (s - t0) / 100.0          # wall clock = 100 MHz -> us
    print(name, "blocks", nb, "start spread us: min %.2f med %.2f max %.2f" % (rel[:, 0].min(), np.median(rel[:, 0]), rel[:, 0].max()))
    for slot in range(8):
        if (s[:, slot] > 0).all():
            print("   stamp %d: med %.2f us  p10 %.2f p90 %.2f max %.2f" % (slot, np.median(rel[:, slot]), np.percentile(rel[:, slot], 10), np.percentile(rel[:, slot], 90), rel[:, slot].max()))
