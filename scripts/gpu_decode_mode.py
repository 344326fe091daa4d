"""Falcon-7B Q4_0 (or argv[2] blocks of it): 128-token prompt + argv[3] greedy decode steps as PLAIN launches in summation order argv[1] (0 default, 2 fast
reference order) -- the workload of scripts/gpu_prof_modes.sh (rocprofv3 --kernel-trace: stream capture under the tracer crashes rocprofv3 7.2)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ggllm_cpp_amd as g
from ggllm_cpp_amd import synth
mode = int(sys.argv[1]); layers = int(sys.argv[2]) if len(sys.argv) > 2 else 32; steps = int(sys.argv[3]) if len(sys.argv) > 3 else 48
g.init(0)
L = g.load()
hp = dict({"7b": synth.HP_7B, "40b": synth.HP_40B}[os.environ.get("MODEL", "7b")]); hp["n_layer"] = layers
tname = {v: k for k, v in g.TYPE_NAME.items()}
w = synth.make_model_fast(hp, tname[os.environ.get("QUANT", "q4_0")], seed=1234)      # MODEL=40b QUANT=q4_K: the k-quant forms
m = g.FalconModel(w, n_ctx=2048, n_batch=128)
if len(sys.argv) > 4:
    m.set_fused(int(sys.argv[4]))          # 1 = three launches per block (k_gemv_ln | k_attn_decode | k_gemv_out)
toks = synth.tokens(136, hp["n_vocab"], seed=42)
L.ggml_hip_reference_order(mode)
lg = m.eval(toks[:128], 0, logits_all=False)
o = m.decode_greedy(int(lg[0].argmax()), 128, steps, use_graph=False)
assert m.sync_error() == 0
print("tokens", list(o[:8]))
