"""time the prefill attention kernels on a Falcon-7B block's shape (71 heads, MQA): python scripts/gpu_attn_forms.py [N ...]
form 32 = k_attention_mfma (32-token tiles, scores through the global scratch), 1 = k_attention_flash (round 5: 32-token tiles, K.Q twice, fp16 probabilities in
LDS, nothing leaves the CU; the default while it fits), 16 = k_attention_mfma16 (f32 scores in LDS), 17 = k_attention_mfma16h (fp16 probabilities in LDS, K.Q twice,
two workgroups per CU); FORMS=32,1 selects; also checks that they agree bit for bit with the first one"""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ggllm_cpp_amd as g
g.init(0); L = g.load()
H, HKV, D = int(os.environ.get("HEADS", "71")), int(os.environ.get("KV_HEADS", "1")), 64
NPAST = int(os.environ.get("NPAST", "0"))       # a batch of N tokens behind NPAST cached positions (round 6: the LONG flash form beyond 2368 keys, e.g. NPAST=7680 N=512)
FORMS = [int(f) for f in os.environ.get("FORMS", "32,1,16,17").split(",")]
for N in [int(a) for a in sys.argv[1:]] or [2048, 512, 128]:
    rng = np.random.default_rng(N)
    qkv = rng.standard_normal((N, H + 2 * HKV, D)).astype(np.float32)
    kc = rng.standard_normal((NPAST + N, HKV, D)).astype(np.float32)
    vc = rng.standard_normal((NPAST + N, HKV, D)).astype(np.float32)
    qb, kb, vb, ob_ = g.DevBuf(host=qkv), g.DevBuf(host=kc), g.DevBuf(host=vc), g.DevBuf(N * H * D * 4)
    ref = None
    for form in FORMS:
        L.ggml_hip_debug_attention_form(form)
        for _ in range(2): L.ggml_hip_attention(qb.ptr, N, H, HKV, D, NPAST, kb.ptr, vb.ptr, ob_.ptr)
        e0, e1 = L.ggml_hip_event_create(), L.ggml_hip_event_create()
        L.ggml_hip_event_record(e0)
        for _ in range(8): L.ggml_hip_attention(qb.ptr, N, H, HKV, D, NPAST, kb.ptr, vb.ptr, ob_.ptr)
        L.ggml_hip_event_record(e1); L.ggml_hip_synchronize()
        ms = L.ggml_hip_event_elapsed_ms(e0, e1) / 8
        out = ob_.to_host(np.float32, (N, H * D))
        if ref is None: ref = out
        print("N=%5d n_past=%5d form %2d: %8.3f ms per launch   bit-identical to the first form: %s" % (N, NPAST, form, ms, bool(np.array_equal(out, ref))), flush=True)
    L.ggml_hip_debug_attention_form(0)
    for b in (qb, kb, vb, ob_): b.free()
