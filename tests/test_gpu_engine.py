"""GPU: the persistent decode engine (csrc/kernels_engine.hip, falcon_hip_context_set_fused(ctx, 4): ONE launch per token --
LDS-DMA loader wave + consumer waves per CU, attention workgroups, tagged-granule hand-offs) reproduces the op-by-op launch
list bit for bit: logits, hidden states of every block, KV cache (through later steps), greedy tokens through the hipGraph."""
import numpy as np
import pytest

import ggllm_cpp_amd as g
from oracle import binding as ob
import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _init():
    # the engine is an opt-in build since round 5 (make -C ggllm.cpp_amd/csrc ENGINE=1): measured 36 % slower than the two-launch default
    if not g.load().falcon_hip_engine_compiled():
        pytest.skip("libggml_hip.so was built without the persistent engine (make ENGINE=1)")
    g.init(0)


@pytest.mark.parametrize("name,hp,t", [("mqa_q4_0", synth.HP_TINY_MQA, ob.Q4_0), ("gqa_q5_1", synth.HP_TINY_GQA, ob.Q5_1),
                                       ("gqa_q4_1", synth.HP_TINY_GQA, ob.Q4_1), ("mqa_q5_0", synth.HP_TINY_MQA, ob.Q5_0),
                                       ("gqa_q8_0", synth.HP_TINY_GQA, ob.Q8_0)])
def test_engine_bit_identical_to_op_list(oracle, name, hp, t):
    w = synth.make_model(oracle, hp, t, seed=21)
    toks = synth.tokens(11, hp["n_vocab"], seed=6)
    outs = []
    for mode in (0, 4):
        m = g.FalconModel(w, n_ctx=32, n_batch=4)
        m.set_fused(mode)
        assert m.engine_active() == (mode == 4)
        m.eval(toks[:4], 0)
        r = [m.eval(toks[i:i + 1], i, want_hidden=True) for i in range(4, 11)]
        assert m.sync_error() == 0
        outs.append(r)
        m.free()
    for (la, ha), (lb, hb) in zip(*outs):
        assert np.array_equal(ha, hb)
        assert np.array_equal(la, lb)


def test_engine_outside_its_scope_falls_back(oracle):
    """k-quant weights are outside the engine's scope: mode 4 then runs the two-launch path, same bits"""
    hp = synth.HP_TINY_GQA
    w = synth.make_model(oracle, hp, ob.Q4_K, seed=3)
    toks = synth.tokens(6, hp["n_vocab"], seed=1)
    res = []
    for mode in (2, 4):
        m = g.FalconModel(w, n_ctx=32, n_batch=4)
        m.set_fused(mode)
        assert not m.engine_active()
        m.eval(toks[:4], 0)
        res.append([m.eval(toks[i:i + 1], i) for i in range(4, 6)])
        m.free()
    for a, b in zip(*res):
        assert np.array_equal(a, b)


WIDE = {"7b": dict(), "7b2n": dict(n_embd=4608, n_head=72, n_head_kv=2, n_ff=18432, two_norms=True)}
_wide_ref = {}


def _wide_run(t, shape, mode):
    hp = dict(synth.HP_7B); hp["n_layer"] = 3; hp["n_vocab"] = 4096
    hp.update(WIDE[shape])
    w = synth.make_model_fast(hp, t, seed=5)
    toks = synth.tokens(16, hp["n_vocab"], seed=9)
    m = g.FalconModel(w, n_ctx=256, n_batch=16)
    m.set_fused(mode)
    assert m.engine_active() == (mode == 4)
    m.eval(toks, 0)
    one = m.eval(toks[-1:], 16)                       # a single plain-launch step first (errors surface here, not inside a graph)
    assert m.sync_error() == 0
    m.eval(toks, 0)
    out = m.decode_greedy(int(toks[-1]), 16, 96, use_graph=True)
    lg = m.eval(out[-1:], 16 + 96)
    assert m.sync_error() == 0
    m.free()
    return one, out, lg


@pytest.mark.parametrize("t,shape", [(ob.Q4_0, "7b"), (ob.Q5_1, "7b"), (ob.Q5_1, "7b2n"), (ob.Q8_0, "7b2n")])
def test_engine_full_width_reference_run(oracle, t, shape):
    """the three-launch form at Falcon-7B width (n_embd 4544 / 71 heads MQA, and a two-norm GQA variant): the reference the
    engine is compared with below"""
    _wide_ref[(t, shape)] = _wide_run(t, shape, 1)


@pytest.mark.parametrize("t,shape", [(ob.Q4_0, "7b"), (ob.Q5_1, "7b"), (ob.Q5_1, "7b2n"), (ob.Q8_0, "7b2n")])
def test_engine_full_width_greedy(oracle, t, shape):
    """3 blocks at full width with every CU streaming: one plain step, then 96 greedy steps through the hipGraph -- same
    logits, same tokens and same final logits as the three-launch form, no wait ever gave up"""
    if (t, shape) not in _wide_ref:
        pytest.skip("reference run failed")
    one, out, lg = _wide_run(t, shape, 4)
    r1, rout, rlg = _wide_ref[(t, shape)]
    assert np.array_equal(one, r1)
    assert np.array_equal(out, rout)
    assert np.array_equal(lg, rlg)
