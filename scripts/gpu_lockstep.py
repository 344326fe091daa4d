"""tuning aid (GPU): lock-step decode streams on one resident model: python scripts/gpu_lockstep.py [B ...]
LOCKSTEP_MODEL=7b_q4_0 (default) | 40b_q4_k | 40b_q5_1 ...; LOCKSTEP_LAYERS=n keeps the first n blocks (quicker set-up);
(FALCON_HIP_COLS_MAX_N=4: 5 and more sequences per pass through the mat-mul kernels instead of the 4-column mat-vec chunks)"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ggllm_cpp_amd as g
from ggllm_cpp_amd import synth
g.init(0); L = g.load()
size, _, fmt = os.environ.get("LOCKSTEP_MODEL", "7b_q4_0").partition("_")
hp = dict(synth.HP_7B if size == "7b" else synth.HP_40B)
if os.environ.get("LOCKSTEP_LAYERS"):
    hp["n_layer"] = int(os.environ["LOCKSTEP_LAYERS"])
wtype = {"q4_0": g.Q4_0, "q4_1": g.Q4_1, "q5_0": g.Q5_0, "q5_1": g.Q5_1, "q8_0": g.Q8_0, "q2_k": g.Q2_K, "q3_k": g.Q3_K, "q4_k": g.Q4_K, "q5_k": g.Q5_K, "q6_k": g.Q6_K}[fmt]
w = synth.make_model_fast(hp, wtype, seed=1234)
model = g.FalconModel(w, n_ctx=512, n_batch=256)
del w
for B in [int(x) for x in sys.argv[1:]] or [4, 8, 16]:
    G, R = 2, (32 if size == "7b" else 8)
    pipe = g.Pipeline(model, 0, 1, G, B, 512)
    pipe.set_tokens(synth.tokens(G * B, hp["n_vocab"], seed=42))
    pipe.run(4, 0)
    L.ggml_hip_synchronize()
    t0 = time.perf_counter()
    pipe.run(R, 4)
    L.ggml_hip_synchronize()
    dt = time.perf_counter() - t0
    pipe.free()
    print("%s x %d blocks, B = %3d streams per pass (x %d groups): %.3f ms per weight pass, %.0f tok/s" % (os.environ.get("LOCKSTEP_MODEL", "7b_q4_0"), hp["n_layer"], B, G, dt / (R * G) * 1e3, R * G * B / dt), flush=True)
model.free()
