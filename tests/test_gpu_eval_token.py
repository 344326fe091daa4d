"""GPU: falcon_hip_eval_token (one token through the captured graph, logits fetched lazily -- what the falcon_eval wrap uses for
n_tokens = 1) and falcon_hip_context_set_rope_n_ctx (the per-call n_max_real_ctx of the reference, libfalcon.cpp:2229-2230)."""
import numpy as np
import pytest

import ggllm_cpp_amd as g
from oracle import binding as ob
import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _init():
    g.init(0)


@pytest.mark.parametrize("hp,t", [(synth.HP_TINY_MQA, ob.Q4_0), (synth.HP_TINY_GQA, ob.Q4_K)])
def test_eval_token_equals_eval(oracle, hp, t):
    w = synth.make_model(oracle, hp, t, seed=77)
    toks = synth.tokens(12, hp["n_vocab"], seed=3)
    m = g.FalconModel(w, n_ctx=64, n_batch=8)
    m.eval(toks[:4], 0)
    ref = [m.eval(toks[i:i + 1], i)[0] for i in range(4, 12)]
    m.eval(toks[:4], 0)
    for k, i in enumerate(range(4, 12)):
        m.eval_token(toks[i], i)
        if k % 2 == 0:                                    # (the row is only copied when asked for; skipped steps leave nothing behind)
            assert np.array_equal(m.logits(), ref[k]), i
    assert m.sync_error() == 0
    # an ordinary eval after pending single tokens still returns its own logits
    assert np.array_equal(m.eval(toks[11:12], 11)[0], ref[-1])
    m.free()


def test_rope_context_can_change_per_call(oracle):
    hp = synth.HP_TINY_MQA
    w = synth.make_model(oracle, hp, ob.Q4_0, seed=78)
    toks = synth.tokens(6, hp["n_vocab"], seed=4)
    big = g.FalconModel(w, n_ctx=4096, n_batch=8)                       # rope table of a 4096 context: NTK factor 3
    small = g.FalconModel(w, n_ctx=4096, n_batch=8, rope_n_ctx=40)      # ... handed n_max_real_ctx = 40 at creation: factor 1
    a = big.eval(toks, 0)
    b = small.eval(toks, 0)
    assert not np.array_equal(a, b)
    big.set_rope_n_ctx(40)
    assert np.array_equal(big.eval(toks, 0), b)
    big.set_rope_n_ctx(0)                                               # back to the context's own n_ctx
    assert np.array_equal(big.eval(toks, 0), a)
    big.free(); small.free()
