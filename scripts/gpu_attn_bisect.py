"""which prefill attention form differs from the others, and is each one reproducible? python scripts/gpu_attn_bisect.py"""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ggllm_cpp_amd as g
g.init(0); L = g.load()
FORMS = [int(f) for f in os.environ.get("FORMS", "32,16,1").split(",")]
H, HKV, D = 2, 1, 64
SEEDS = [int(x) for x in os.environ.get('SEEDS', '0').split(',')]
for N, seed in [(n, sd) for sd in SEEDS for n in (160, 256, 288, 320, 512)]:
    rng = np.random.default_rng(N + seed)
    qkv = rng.standard_normal((N, H + 2 * HKV, D)).astype(np.float32)
    kc = rng.standard_normal((N, HKV, D)).astype(np.float32)
    vc = rng.standard_normal((N, HKV, D)).astype(np.float32)
    qb, kb, vb, ob_ = g.DevBuf(host=qkv), g.DevBuf(host=kc), g.DevBuf(host=vc), g.DevBuf(N * H * D * 4)
    outs = {}
    for rep in range(3):
        for form in FORMS:
            L.ggml_hip_debug_attention_form(form)
            L.ggml_hip_memset(ob_.ptr, 0xFF, N * H * D * 4)
            L.ggml_hip_attention(qb.ptr, N, H, HKV, D, 0, kb.ptr, vb.ptr, ob_.ptr)
            outs[(form, rep)] = ob_.to_host(np.float32, (N, H, D))
    L.ggml_hip_debug_attention_form(0)
    ref = outs[(FORMS[0], 2)]
    msg = []
    for form in FORMS:
        for rep in range(3):
            bad = outs[(form, rep)] != ref
            if bad.any():
                rows = np.nonzero(bad.any(axis=(1, 2)))[0]
                msg.append("form %d rep %d: %d rows differ from form %d rep 2 (tiles %s)" % (form, rep, len(rows), FORMS[0], sorted(set((rows // 32).tolist()))[:8]))
    print("N=%4d seed %d: %s" % (N, seed, "; ".join(msg) if msg else "all forms, all repetitions identical"), flush=True)
    for b in (qb, kb, vb, ob_): b.free()
