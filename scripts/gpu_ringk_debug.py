"""tuning aid (GPU): phase timelines of the k-quant ring launches (k_ring_ln_k, k_ring_out_sys / k_ring_out) on Falcon-40B shapes, medians over the workgroups
usage: python scripts/gpu_ringk_debug.py [quant=q4_k] [layers=4]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ggllm_cpp_amd as g
from ggllm_cpp_amd import synth
g.init(0); L = g.load()
q = sys.argv[1] if len(sys.argv) > 1 else "q4_k"
tname = {v: k for k, v in g.TYPE_NAME.items()}
wtype = tname[q if q in tname else q.replace("_k", "_K")]
hp = dict(synth.HP_40B); hp["n_layer"] = int(sys.argv[2]) if len(sys.argv) > 2 else 4; hp["n_vocab"] = 4096
w = synth.make_model_fast(hp, wtype)
m = g.FalconModel(w, n_ctx=512, n_batch=8)
m.set_fused(5)
toks = synth.tokens(8, hp["n_vocab"])
m.eval(toks, 0)
out = m.decode_greedy(1, 8, 100)
L.ggml_hip_debug_stamps(1, None)
m.decode_greedy(int(out[-1]), 108, 4)
st = np.zeros(2 * 4096 * 8, np.int64)
L.ggml_hip_debug_stamps(1, st.ctypes.data)
st = st.reshape(2, 4096, 8).astype(np.float64)
names = [{(0, 0): "workgroup starts", (0, 1): "loader starts", (0, 2): "last piece issued", (0, 3): "all landed",
          (1, 1): "epilogue wave: mean known", (1, 2): "epilogue wave: image done", (1, 3): "epilogue wave: epilogues done",
          (2, 1): "consumer 0: mean known", (2, 2): "consumer 0: image done", (2, 3): "consumer 0: up rows done", (2, 4): "consumer 0: qkv rows done",
          (3, 3): "consumer 9: up rows done", (3, 4): "consumer 9: qkv rows done"},
         {(0, 0): "workgroup starts", (0, 1): "loader starts", (0, 2): "last piece issued", (0, 3): "all landed",
          (1, 2): "helper 0: images done", (1, 3): "helper 0: Wdown rows done (row form)", (1, 4): "helper 0: done",
          (2, 2): "helper 1: images done", (2, 4): "helper 1: done"}]
for region, title in ((0, "LayerNorm mat-vec launch [Wup | Wqkv]"), (1, "output mat-vec launch [Wdown, Wo]")):
    s = st[region]
    t0 = s[:256, 0][s[:256, 0] > 0].min()
    print("== %s (%s)" % (title, q))
    for (role, slot), nm in names[region].items():
        v = s[role * 256:(role + 1) * 256, slot]
        v = (v[v > 0] - t0) / 100.0
        if len(v):
            print("%-40s med %6.2f us   p10 %6.2f  p90 %6.2f  max %6.2f   (%d workgroups)" % (nm, np.median(v), np.percentile(v, 10), np.percentile(v, 90), v.max(), len(v)))
m.free()
