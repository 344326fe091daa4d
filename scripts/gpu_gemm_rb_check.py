"""64-row GEMM tiles (FQ_GEMM_CFG 6 / 7) against the oracle with the same association, then timings per cfg.
python scripts/gpu_gemm_rb_check.py check|time"""
import sys, os, subprocess, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
what = sys.argv[1] if len(sys.argv) > 1 else "check"
if what == "all":
    for cfg, order in ((6, 4), (7, 3)):
        subprocess.run([sys.executable, __file__, "check"], env=dict(os.environ, FQ_GEMM_CFG=str(cfg), ORC_ORDER=str(order)))
    for cfg in ("3", "6", "7", "2"):
        subprocess.run([sys.executable, __file__, "time"], env=dict(os.environ, FQ_GEMM_CFG=cfg))
    sys.exit(0)
import ggllm_cpp_amd as g
from ggllm_cpp_amd import synth
g.init(0); L = g.load()
cfg = os.environ.get("FQ_GEMM_CFG", "default")
if what == "check":
    from oracle import binding as ob
    import synth as tsynth
    ob.build_oracle(); orc = ob.Oracle()
    order = int(os.environ.get("ORC_ORDER", "4"))
    bad = 0
    for t in ob.WEIGHT_TYPES:
        if cfg == "7" and t in ob.KQUANTS: continue
        for K, M, N in ((512, 37, 9), (4544, 200, 33), (8192, 129, 130), (1024, 300, 257), (18176, 70, 40)):
            if K % ob.BLCK[t]: continue
            rng = np.random.default_rng(K + M + N + t)
            w = tsynth.quantized_matrix(orc, t, M, K, rng)
            x = rng.standard_normal((N, K)).astype(np.float32)
            dw = g.Weight(t, w, K, M); got = dw.mul_mat(x); dw.free()
            orc.lib.orc_set_sum_order(order)
            try: exp = orc.mul_mat(t, w, K, M, x, 8)
            finally: orc.lib.orc_set_sum_order(0)
            rel = float(np.abs(got.astype(np.float64) - exp).max() / (np.sqrt((exp.astype(np.float64) ** 2).mean()) + 1e-30))
            exact = np.array_equal(got, exp)
            ok = exact if t in ob.LEGACY else rel <= 2e-5
            if not ok: bad += 1
            print("cfg %s type %-5s K=%5d M=%3d N=%3d  %s rel %.2e" % (cfg, ob.TYPE_NAME[t], K, M, N, "bit-exact" if exact else ("ok" if ok else "MISMATCH"), rel), flush=True)
    print("cfg", cfg, "mismatches:", bad)
else:
    rng = np.random.default_rng(0)
    cases = [(g.Q4_0, "q4_0", "7b-up", 4544, 18176), (g.Q4_0, "q4_0", "7b-down", 18176, 4544), (g.Q4_0, "q4_0", "7b-qkv", 4544, 4672),
             (g.Q5_1, "q5_1", "40b-up", 8192, 32768), (g.Q4_K, "q4_k", "40b-up", 8192, 32768), (g.Q2_K, "q2_k", "40b-down", 32768, 8192), (g.Q6_K, "q6_k", "40b-up", 8192, 32768)]
    for t, tn, name, K, M in cases:
        if cfg == "7" and tn.endswith("_k"): continue
        w = g.Weight(t, synth.random_blocks(t, M, K, rng), K, M)
        for N in (256, 512, 2048):
            x = rng.standard_normal((N, K)).astype(np.float32)
            xb, yb = g.DevBuf(host=x), g.DevBuf(N * M * 4)
            for _ in range(2): L.ggml_hip_mul_mat_q(w.h, xb.ptr, K, N, yb.ptr, M)
            e0, e1 = L.ggml_hip_event_create(), L.ggml_hip_event_create()
            L.ggml_hip_event_record(e0)
            for _ in range(5): L.ggml_hip_mul_mat_q(w.h, xb.ptr, K, N, yb.ptr, M)
            L.ggml_hip_event_record(e1); L.ggml_hip_synchronize()
            us = L.ggml_hip_event_elapsed_ms(e0, e1) * 200
            print("cfg %s %s %-8s N=%4d %9.1f us  %6.1f TOP/s" % (cfg, tn, name, N, us, 2.0 * M * K * N / us / 1e6), flush=True)
            xb.free(); yb.free()
        w.free()
