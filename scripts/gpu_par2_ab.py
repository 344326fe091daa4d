#!/usr/bin/env python3
"""A/B of the two-branch prefill blocks inside ONE process (same clocks, same weights): contexts created with
FALCON_HIP_PAR2_MAX_N = 0 and 512 over the same resident Falcon-7B Q4_0 model, prompts of 64 .. 512 tokens, interleaved."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ggllm_cpp_amd as g
from ggllm_cpp_amd import synth

g.init(0)
L = g.load()
hp = dict(synth.HP_7B)
m = g.FalconModel(synth.make_model_fast(hp, 2, seed=1234), n_ctx=2048, n_batch=2048)
ctx = {}
for mode in (0, 4096):
    os.environ["FALCON_HIP_PAR2_MAX_N"] = str(mode)
    ctx[mode] = L.falcon_hip_context_create(m.m, 2048, 2048, 0)
toks = synth.tokens(2048, hp["n_vocab"], seed=42)
for _ in range(3):                                  # warm the clocks
    L.falcon_hip_eval(ctx[0], toks.ctypes.data, 512, 0, 0)
e0, e1 = L.ggml_hip_event_create(), L.ggml_hip_event_create()
for N in (16, 32, 128, 512, 1024, 2048):
    res = {0: [], 4096: []}
    for rep in range(4):
        for mode in (0, 4096):
            L.falcon_hip_eval(ctx[mode], toks.ctypes.data, N, 0, 0)
            L.ggml_hip_event_record(e0)
            L.falcon_hip_eval(ctx[mode], toks.ctypes.data, N, 0, 0)
            L.ggml_hip_event_record(e1)
            L.ggml_hip_synchronize()
            res[mode].append(L.ggml_hip_event_elapsed_ms(e0, e1))
    print("N=%4d  one stream %.2f ms   two branches %.2f ms   (min of 4; all: %s | %s)" % (N, min(res[0]), min(res[4096]),
          " ".join("%.2f" % v for v in res[0]), " ".join("%.2f" % v for v in res[4096])), flush=True)
