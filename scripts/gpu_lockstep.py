"""tuning aid (GPU): lock-step decode streams on one resident Falcon-7B Q4_0: python scripts/gpu_lockstep.py [B ...]
(FALCON_HIP_COLS_MAX_N=4: 5 and more sequences per pass through the mat-mul kernels instead of the 4-column mat-vec chunks)"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ggllm_cpp_amd as g
from ggllm_cpp_amd import synth
g.init(0); L = g.load()
hp = dict(synth.HP_7B)
w = synth.make_model_fast(hp, g.Q4_0, seed=1234)
model = g.FalconModel(w, n_ctx=512, n_batch=256)
for B in [int(x) for x in sys.argv[1:]] or [4, 8, 16]:
    G, R = 2, 32
    pipe = g.Pipeline(model, 0, 1, G, B, 512)
    pipe.set_tokens(synth.tokens(G * B, hp["n_vocab"], seed=42))
    pipe.run(8, 0)
    L.ggml_hip_synchronize()
    t0 = time.perf_counter()
    pipe.run(R, 8)
    L.ggml_hip_synchronize()
    dt = time.perf_counter() - t0
    pipe.free()
    print("B = %2d streams per pass (x %d groups): %.3f ms per weight pass, %.0f tok/s" % (B, G, dt / (R * G) * 1e3, R * G * B / dt), flush=True)
model.free()
