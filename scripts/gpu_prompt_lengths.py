#!/usr/bin/env python3
"""tuning aid (GPU): prompt evaluation time by prompt length on one resident model, events around falcon_hip_eval (no logits copy in the timed region):
python scripts/gpu_prompt_lengths.py [N ...]    PROMPT_MODEL=7b_q4_0 (default) | 40b_q4_k | ...; PROMPT_LAYERS=n keeps the first n blocks; PROMPT_ORDER=2: the fast reference order"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ggllm_cpp_amd as g
from ggllm_cpp_amd import synth

g.init(0)
L = g.load()
size, _, fmt = os.environ.get("PROMPT_MODEL", "7b_q4_0").partition("_")
hp = dict(synth.HP_7B if size == "7b" else synth.HP_40B)
if os.environ.get("PROMPT_LAYERS"):
    hp["n_layer"] = int(os.environ["PROMPT_LAYERS"])
wtype = {"q4_0": g.Q4_0, "q4_1": g.Q4_1, "q5_0": g.Q5_0, "q5_1": g.Q5_1, "q8_0": g.Q8_0, "q2_k": g.Q2_K, "q3_k": g.Q3_K, "q4_k": g.Q4_K, "q5_k": g.Q5_K, "q6_k": g.Q6_K}[fmt]
Ns = [int(a) for a in sys.argv[1:]] or [8, 16, 32, 48, 64, 80, 96, 128]
NCTX = max(512, max(Ns))
m = g.FalconModel(synth.make_model_fast(hp, wtype, seed=1234), n_ctx=NCTX, n_batch=max(Ns))
ctx = L.falcon_hip_context_create(m.m, NCTX, max(Ns), 0)
L.ggml_hip_reference_order(int(os.environ.get("PROMPT_ORDER", "0")))
toks = synth.tokens(max(Ns), hp["n_vocab"], seed=42)
for _ in range(2):
    L.falcon_hip_eval(ctx, toks.ctypes.data, max(Ns), 0, 0)
e0, e1 = L.ggml_hip_event_create(), L.ggml_hip_event_create()
for N in Ns:
    res = []
    for rep in range(4):
        L.falcon_hip_eval(ctx, toks.ctypes.data, N, 0, 0)
        L.ggml_hip_event_record(e0)
        L.falcon_hip_eval(ctx, toks.ctypes.data, N, 0, 0)
        L.ggml_hip_event_record(e1)
        L.ggml_hip_synchronize()
        res.append(L.ggml_hip_event_elapsed_ms(e0, e1))
    print("%s x %d blocks: %4d-token prompt %8.2f ms  (%7.0f tok/s; min of 4)" % (os.environ.get("PROMPT_MODEL", "7b_q4_0"), hp["n_layer"], N, min(res), N / min(res) * 1e3), flush=True)
