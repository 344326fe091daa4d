"""A/B on one box: Falcon-7B Q4_0 (32 blocks), 128-token prompt + greedy decode through the hipGraph, default order against the FAST reference order
(ggml_hip_reference_order(2), csrc/fq_ref_chain.h), and the fast reference order's logits against the one-thread-per-output instrument (mode 1) at full depth."""
import os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ggllm_cpp_amd as g
from ggllm_cpp_amd import synth

g.init(0)
L = g.load()
quant = sys.argv[1] if len(sys.argv) > 1 else "q4_0"
model_name = sys.argv[2] if len(sys.argv) > 2 else "7b"
tname = {v: k for k, v in g.TYPE_NAME.items()}
hp = dict({"7b": synth.HP_7B, "40b": synth.HP_40B}[model_name])
if len(sys.argv) > 3:
    hp["n_layer"] = int(sys.argv[3])
w = synth.make_model_fast(hp, tname[quant if quant in tname else quant.replace("_k", "_K")], seed=1234)
m = g.FalconModel(w, n_ctx=2048, n_batch=128)
toks = synth.tokens(136, hp["n_vocab"], seed=42)
K = 64
out = {}
LEGACY = quant in ('q4_0', 'q4_1', 'q5_0', 'q5_1', 'q8_0', 'q2_k', 'q3_k', 'q4_k', 'q5_k', 'q6_k')      # formats with a fast form of the reference's association (k-quants: single-token mat-vecs, kernels_kqref.hip)
MODES = (0, 2) if LEGACY else (0,)
e0, e1 = L.ggml_hip_event_create(), L.ggml_hip_event_create()
for rep in range(2):
    for mode in MODES:
        L.ggml_hip_reference_order(mode)
        m.eval(toks[:128], 0, logits_all=False)
        L.ggml_hip_event_record(e0)
        lg = m.eval(toks[:128], 0, logits_all=False)
        L.ggml_hip_event_record(e1)
        pre_ms = L.ggml_hip_event_elapsed_ms(e0, e1)
        first = int(lg[0].argmax())
        ow = m.decode_greedy(first, 128, 5, use_graph=True)
        L.ggml_hip_synchronize()
        t0 = time.perf_counter()
        o = m.decode_greedy(int(ow[-1]), 133, K, use_graph=True)
        L.ggml_hip_synchronize()
        dt = time.perf_counter() - t0
        assert m.sync_error() == 0
        print(f"rep {rep} mode {mode}: prefill128 {pre_ms:.2f} ms ({128 / pre_ms * 1e3:.0f} tok/s), decode {K / dt:.1f} tok/s ({dt / K * 1e6:.1f} us/token)", flush=True)
        out.setdefault(str(mode), []).append({"prefill_ms": pre_ms, "decode_tok_s": K / dt})
        L.ggml_hip_reference_order(0)
# parity at full depth: mode 2 == mode 1 (prefill logits of the last prompt token, then two decode steps)
res = {}
for mode in ((1, 2) if LEGACY else ()):
    L.ggml_hip_reference_order(mode)
    a = m.eval(toks[:128], 0, logits_all=False)[0].copy()
    b = m.eval(toks[128:129], 128, logits_all=False)[0].copy()
    c = m.eval(toks[129:130], 129, logits_all=False)[0].copy()
    res[mode] = (a, b, c)
    L.ggml_hip_reference_order(0)
eq = [bool(np.array_equal(x, y)) for x, y in zip(res[1], res[2])] if LEGACY else None
# the one-thread-per-output instrument's decode speed, for scale
L.ggml_hip_reference_order(1)
cur = 5
L.ggml_hip_synchronize(); t0 = time.perf_counter()
for i in range(4):
    cur = int(m.eval(np.array([cur], np.int32), 130 + i, logits_all=False)[0].argmax())
L.ggml_hip_synchronize(); out["mode1_decode_tok_s"] = 4 / (time.perf_counter() - t0)
L.ggml_hip_reference_order(0)
print("mode 1 decode: %.1f tok/s" % out["mode1_decode_tok_s"], flush=True)
print("mode 2 == mode 1 (prefill, step 128, step 129):", eq, flush=True)
out["mode2_equals_mode1"] = eq
# per-launch timing table of one decode step in each mode
for mode in MODES:
    L.ggml_hip_reference_order(mode)
    L.ggml_hip_profile_begin()
    m.decode_greedy(5, 133, 16, use_graph=False)
    import ctypes as C
    nl, us, by = C.c_int64(), C.c_double(), C.c_double()
    L.ggml_hip_profile_end(C.byref(nl), C.byref(us), C.byref(by))
    print(f"mode {mode}: {nl.value} fused launches, avg {us.value / max(1, nl.value):.2f} us, {by.value / max(1e-9, us.value) / 1e3:.1f} GB/s", flush=True)
    out.setdefault("launch_avg_us", {})[str(mode)] = us.value / max(1, nl.value)
    L.ggml_hip_reference_order(0)
print(json.dumps(out))
