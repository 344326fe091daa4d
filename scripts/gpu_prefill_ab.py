"""tuning aid (GPU): prompt times of the resident Falcon-7B Q4_0 at several lengths: python scripts/gpu_prefill_ab.py [N ...]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ggllm_cpp_amd as g
from ggllm_cpp_amd import synth
g.init(0); L = g.load()
hp = dict(synth.HP_7B)
w = synth.make_model_fast(hp, g.Q4_0, seed=1234)
Ns = [int(x) for x in sys.argv[1:]] or [128, 512, 1024, 2048]
m = g.FalconModel(w, n_ctx=2048, n_batch=max(Ns))
for N in Ns:
    toks = synth.tokens(N, hp["n_vocab"], seed=42)
    m.eval(toks, 0); L.ggml_hip_synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); m.eval(toks, 0); L.ggml_hip_synchronize(); ts.append(time.perf_counter() - t0)
    print("N = %4d: %.2f ms (%.0f tok/s)" % (N, min(ts) * 1e3, N / min(ts)), flush=True)
m.free()
