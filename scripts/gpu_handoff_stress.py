"""stress of the in-launch hand-offs (k_attn_out, and k_attn_out_ln with mode 3): full Falcon-7B Q4_0, long greedy decodes, every
mode must produce the same tokens as the 3-launch form and no sweep may time out:  python scripts/gpu_handoff_stress.py [steps]"""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ggllm_cpp_amd as g
from ggllm_cpp_amd import synth
g.init(0)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 768
hp = dict(synth.HP_7B)
w = synth.make_model_fast(hp, g.Q4_0, seed=1234)
toks = synth.tokens(64, hp["n_vocab"], seed=42)
res = {}
for mode in (1, 2, 3):
    m = g.FalconModel(w, n_ctx=1024, n_batch=64)
    m.set_fused(mode)
    m.eval(toks, 0, logits_all=False)
    out = []
    n_past, first = 64, int(toks[-1])
    for chunk in range(0, steps, 256):                     # several graph captures, different base positions
        n = min(256, steps - chunk)
        o = m.decode_greedy(first, n_past, n, use_graph=True)
        out.append(o); n_past += n; first = int(o[-1])
    res[mode] = np.concatenate(out)
    print("mode", mode, "sync_error", m.sync_error(), "tokens", res[mode][:6], "...", flush=True)
    assert m.sync_error() == 0
    m.free()
for mode in (2, 3):
    same = np.array_equal(res[1], res[mode])
    print("mode %d == mode 1: %s" % (mode, same))
    assert same
print("ok")
