import os, sys, time
sys.path.insert(0, "/root/repo")
import ggllm_cpp_amd as g
from ggllm_cpp_amd import synth
g.init(0); L = g.load()
hp = dict(synth.HP_7B)
m = g.FalconModel(synth.make_model_fast(hp, 2, seed=1234), n_ctx=2048, n_batch=128)
toks = synth.tokens(128, hp["n_vocab"], seed=42)
for _ in range(6):
    L.falcon_hip_eval(m.ctx, toks.ctypes.data, 128, 0, 0)
L.ggml_hip_synchronize()
