"""how well-conditioned do the tiny synthetic models get? default order / CPU reference / f64 evaluation, by out_gain and seed"""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import ggllm_cpp_amd as g
from oracle import binding as ob
import synth
g.init(0)
oracle = ob.Oracle()
def rel(a, b): return float(np.abs(a.astype(np.float64) - b.astype(np.float64)).max() / np.sqrt((b.astype(np.float64) ** 2).mean()))
for name, hp0, t in (("mqa_q4_0", synth.HP_TINY_MQA, ob.Q4_0), ("gqa_q5_1", synth.HP_TINY_GQA, ob.Q5_1), ("gqa_q8_0", synth.HP_TINY_GQA, ob.Q8_0)):
    for lg in (6, 9, 12):
        for seed in (77, 78, 79):
            hp = dict(hp0); hp["n_layer"] = 6
            w = synth.make_model(oracle, hp, t, seed=seed, out_gain=2.0 ** -lg)
            toks = synth.tokens(14, hp["n_vocab"], seed=3)
            m = g.FalconModel(w, n_ctx=32, n_batch=8)
            fast = [m.eval(toks[:8], 0, logits_all=True)] + [m.eval(toks[i:i + 1], i, logits_all=True) for i in range(8, 14)]
            m.free()
            def run(order):
                oracle.lib.orc_set_sum_order(order)
                try:
                    mo = oracle.model(w, 32)
                    return [mo.eval(toks[:8], 0, 4)] + [mo.eval(toks[i:i + 1], i, 4) for i in range(8, 14)]
                finally:
                    oracle.lib.orc_set_sum_order(0)
            cpu, yard = run(0), run(6)
            worst = lambda xs, ys: max(rel(a, b) for a, b in zip(xs, ys))
            print("%s gain 2^-%-2d seed %d: default vs cpu %.2e | default vs f64 %.2e | cpu vs f64 %.2e" % (name, lg, seed, worst(fast, cpu), worst(fast, yard), worst(cpu, yard)), flush=True)
