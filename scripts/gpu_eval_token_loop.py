"""tuning aid (GPU): the per-token cost of the C-ABI single-token path -- falcon_hip_eval_token + falcon_hip_get_logits + a host argmax, one token at a time, as the
reference's CLI drives it through the wrap -- against the device-side greedy loop (falcon_hip_decode_greedy, no host round trip)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ggllm_cpp_amd as g
from ggllm_cpp_amd import synth
g.init(0); L = g.load()
hp = dict(synth.HP_7B)
w = synth.make_model_fast(hp, g.Q4_0)
m = g.FalconModel(w, n_ctx=2048, n_batch=128)
toks = synth.tokens(128, hp["n_vocab"])
lg = m.eval(toks, 0, logits_all=False)
cur = int(lg[0].argmax())
n = 128
for rep in range(2):
    L.ggml_hip_synchronize()
    t0 = time.perf_counter(); t_eval = 0.0; t_get = 0.0
    c = cur
    for i in range(n):
        a = time.perf_counter()
        m.eval_token(c, 128 + i)
        b = time.perf_counter()
        row = m.logits()
        d = time.perf_counter()
        c = int(row.argmax())
        t_eval += b - a; t_get += d - b
    dt = time.perf_counter() - t0
    print("eval_token loop: %.1f us per token (eval call %.1f us, get_logits wait %.1f us, host argmax %.1f us) = %.1f tok/s" % (dt / n * 1e6, t_eval / n * 1e6, t_get / n * 1e6, (dt - t_eval - t_get) / n * 1e6, n / dt))
out = m.decode_greedy(cur, 128, 16, use_graph=True)
L.ggml_hip_synchronize()
t0 = time.perf_counter(); m.decode_greedy(cur, 128, n, use_graph=True); L.ggml_hip_synchronize(); dt = time.perf_counter() - t0
print("device-side greedy loop: %.1f us per token = %.1f tok/s" % (dt / n * 1e6, n / dt))
m.free()
