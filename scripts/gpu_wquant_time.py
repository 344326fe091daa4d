"""throughput of the device weight quantizers on a Falcon-sized matrix (rows x K f32 resident in HBM -> ggml blocks),
next to the reference build's quantizer on one host core for a 1/64 sample (when oracle/_ref travelled)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ggllm_cpp_amd as g  # noqa: E402

L = g.load()
g.init(0)
rng = np.random.default_rng(1)
rows = 18176
for t in g.LEGACY + g.KQUANTS:
    K = 4608 if t in g.KQUANTS else 4544
    x = (rng.standard_normal((rows, K)) * 0.02).astype(np.float32)
    xb = g.DevBuf(host=x)
    ob_ = g.DevBuf(rows * (K // g.BLCK[t]) * g.TSIZE[t])
    L.ggml_hip_quantize_rows(t, xb.ptr, K, rows, ob_.ptr, None)
    L.ggml_hip_synchronize()
    e0, e1 = L.ggml_hip_event_create(), L.ggml_hip_event_create()
    L.ggml_hip_event_record(e0)
    for _ in range(5):
        L.ggml_hip_quantize_rows(t, xb.ptr, K, rows, ob_.ptr, None)
    L.ggml_hip_event_record(e1)
    ms = L.ggml_hip_event_elapsed_ms(e0, e1) / 5
    line = "%-5s %d x %d: %.3f ms  %.1f G weights/s  (%.0f GB/s of f32 read)" % (g.TYPE_NAME[t], rows, K, ms, rows * K / ms / 1e6, rows * K * 4 / ms / 1e6)
    try:
        from oracle import binding as ob
        if ob.Ref.available():
            R = ob.Ref()
            sample = x[:rows // 64]
            t0 = time.perf_counter()
            R.quantize_chunk(t, sample)
            dt = time.perf_counter() - t0
            line += " | reference, 1 core: %.1f M weights/s" % (sample.size / dt / 1e6)
    except Exception as e:  # noqa: BLE001
        line += " | (reference not timed: %s)" % e
    print(line, flush=True)
    xb.free(); ob_.free()
