#!/usr/bin/env python3
"""tuning aid (GPU): failure records and the phase timeline of the persistent decode engine"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["FALCON_HIP_ENGINE_DEBUG"] = "1"
import ggllm_cpp_amd as g          # noqa: E402
from ggllm_cpp_amd import synth    # noqa: E402

CODES = {1: "loader: ring space", 2: "consumer: row landed", 4: "attention: qkv granules", 5: "gather: x granules", 6: "group rows", 7: "gather: GELU image",
         8: "gather: attention image", 12: "LN statistics flag", 13: "LN image counter", 14: "phase A counter", 15: "Wdown counter", 16: "Wo counter",
         17: "GELU image counter", 18: "attention image counter"}


def run(tname, shape, layers, steps=8, timeline=True):
    t = {v: k for k, v in g.TYPE_NAME.items()}[tname]
    hp = dict(synth.HP_7B); hp["n_layer"] = layers; hp["n_vocab"] = 4096 if layers < 32 else 65024
    if shape == "7b2n":
        hp.update(n_embd=4608, n_head=72, n_head_kv=2, n_ff=18432, two_norms=True)
    w = synth.make_model_fast(hp, t, seed=5)
    toks = synth.tokens(16, hp["n_vocab"], seed=9)
    m = g.FalconModel(w, n_ctx=512, n_batch=16)
    m.set_fused(4)
    print("==== %s %s %d blocks: engine active %s" % (tname, shape, layers, m.engine_active()), flush=True)
    m.eval(toks, 0)
    try:
        m.eval(toks[-1:], 16)
    except RuntimeError as e:
        print("  step failed:", e)
    rec, st = m.engine_debug()
    if rec is not None and len(rec):
        print("  %d failure records (code, workgroup, wave, block, x0, x1):" % len(rec))
        for r in rec[:24]:
            print("   ", CODES.get(int(r[0]), r[0]), "| wg", int(r[1]), "wave", int(r[2]), "block", int(r[3]), "x0", int(r[4]), "x1", int(r[5]))
    elif timeline:
        L = g.load()
        out = m.decode_greedy(int(toks[-1]), 17, steps, use_graph=True)
        L.ggml_hip_synchronize()
        t0 = time.perf_counter()
        out = m.decode_greedy(int(out[-1]), 17 + steps, 64, use_graph=True)
        L.ggml_hip_synchronize()
        dt = time.perf_counter() - t0
        print("  64 steps: %.1f us / token, %.1f tok/s" % (dt / 64 * 1e6, 64 / dt))
        rec, st = m.engine_debug()
        gs = m.engine_gstamps
        n_attn = (hp["n_head"] + 1) // 2
        s = st[n_attn:256].astype(np.float64) / 100.0      # us (100 MHz wall clock)
        gg = gs[n_attn:256].astype(np.float64) / 100.0
        cn = ["start", "x chunk in", "LN stats", "image done", "A rows done", "GELU image in", "Wdown rows done", "attention image in (Wo starts)"]
        gn = ["start", "x chunk in", "LN stats", "image done", "epilogues done", "GELU image in", "-", "attention image in"]
        for bi, bl in enumerate(["block 0", "block 1", "block 2", "last block"]):
            d = s[:, bi, :]
            ok = d[:, 0] > 0
            if not ok.any():
                continue
            d = d[ok]; dg = gg[ok][:, bi, :]
            t0 = d[:, 0].min()
            print("  %s: start spread %.1f us; stamps (us after the first workgroup starts the block) median [min..max] over %d workgroups: consumer 0 | gatherer" % (bl, d[:, 0].max() - t0, len(d)))
            for k in range(8):
                x = d[:, k] - t0; y = dg[:, k] - t0
                print("     %-32s %6.2f [%6.2f .. %6.2f]   | %-22s %6.2f [%6.2f .. %6.2f]" % (cn[k], np.median(x), x.min(), x.max(), gn[k], np.median(y), y.min(), y.max()))
        for bi in range(3):
            a0, a1 = s[:, bi, 0], s[:, bi + 1, 0]
            if a0.max() > 0 and a1.max() > 0 and bi + 1 < 3:
                print("  block %d start -> block %d start: %.2f us (first workgroup), %.2f us (median)" % (bi, bi + 1, a1[a1 > 0].min() - a0[a0 > 0].min(), np.median(a1[a1 > 0]) - np.median(a0[a0 > 0])))
        sa = st[:n_attn].astype(np.float64) / 100.0
        for bi, bl in enumerate(["block 0", "block 1", "block 2", "last block"]):
            d = sa[:, bi, :]
            if d[:, 0].max() > 0 and s[:, bi, 0].max() > 0:
                t0 = s[s[:, bi, 0] > 0, bi, 0].min()
                print("  attention workgroups, %s (us after the block's first streaming workgroup starts): wait begins %.1f, q/k/v gathered %.1f [%.1f..%.1f], image published %.1f [%.1f..%.1f]"
                      % (bl, np.median(d[:, 0]) - t0, np.median(d[:, 1]) - t0, d[:, 1].min() - t0, d[:, 1].max() - t0, np.median(d[:, 2]) - t0, d[:, 2].min() - t0, d[:, 2].max() - t0))
        cnt = m.engine_counters[n_attn:256].astype(np.float64)
        clk = float(np.median(cnt[:, 0])) / (dt / 64 * 1e6)                # s_memtime ticks per us, from the loader's whole-launch count
        med = lambda k: np.median(cnt[:, k]) / clk
        print("  loader (per token, median over workgroups): total %.0f us, blocked on ring space %.0f us in %.0f refills, waiting in vmcnt(32) %.0f us; %.1f MB"
              % (med(0), med(1), np.median(cnt[:, 3]), med(2), np.median(cnt[:, 4]) / 1e6))
        print("  consumer 0 (per token): rows to land %.0f us, dots %.0f us (%d rows), waits: LN statistics %.0f, image %.0f, GELU image %.0f, attention image %.0f; own gathers %.0f us"
              % (med(5), med(6), np.median(cnt[:, 7]), med(8), med(9), med(10), med(11), med(12)))
        print("  gatherer (per token): gathers %.0f us, epilogues (incl. waiting for rows) %.0f us, wait for LN statistics %.0f us" % (med(13), med(14), med(15)))
    m.free()


if __name__ == "__main__":
    g.init(0)
    which = sys.argv[1:] or ["q5_1:7b:3", "q8_0:7b2n:3", "q4_0:7b:32"]
    for spec in which:
        tn, sh, nl = spec.split(":")
        run(tn, sh, int(nl))
