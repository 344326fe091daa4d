"""GPU: the ring form of the fused decode launches (csrc/kernels_ring.hip, falcon_hip_context_set_fused(ctx, 5): an LDS-DMA loader wave
+ consumers out of an LDS ring inside each launch) reproduces the op-by-op launch list bit for bit: logits, hidden states of every
block, greedy tokens through the hipGraph."""
import numpy as np
import pytest

import ggllm_cpp_amd as g
from oracle import binding as ob
import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _init():
    g.init(0)


@pytest.mark.parametrize("name,hp,t", [("mqa_q4_0", synth.HP_TINY_MQA, ob.Q4_0), ("gqa_q5_1", synth.HP_TINY_GQA, ob.Q5_1),
                                       ("gqa_q4_1", synth.HP_TINY_GQA, ob.Q4_1), ("mqa_q5_0", synth.HP_TINY_MQA, ob.Q5_0),
                                       ("gqa_q8_0", synth.HP_TINY_GQA, ob.Q8_0)])
def test_ring_bit_identical_to_op_list(oracle, name, hp, t):
    w = synth.make_model(oracle, hp, t, seed=21)
    toks = synth.tokens(11, hp["n_vocab"], seed=6)
    outs = []
    for mode in (0, 5):
        m = g.FalconModel(w, n_ctx=32, n_batch=4)
        m.set_fused(mode)
        m.eval(toks[:4], 0)
        r = [m.eval(toks[i:i + 1], i, want_hidden=True) for i in range(4, 11)]
        assert m.sync_error() == 0
        outs.append(r)
        m.free()
    for (la, ha), (lb, hb) in zip(*outs):
        assert np.array_equal(ha, hb)
        assert np.array_equal(la, lb)


WIDE = {"7b": dict(), "7b2n": dict(n_embd=4608, n_head=72, n_head_kv=2, n_ff=18432, two_norms=True)}


@pytest.mark.parametrize("shape,t", [("7b", ob.Q4_0), ("7b", ob.Q5_1), ("7b2n", ob.Q8_0), ("7b2n", ob.Q4_1), ("7b", ob.Q5_0)])
def test_ring_full_width_blocks(shape, t):
    """three Falcon-7B-wide blocks (one norm: the real shape; two norms at a 256-divisible width): 24 greedy steps through the hipGraph
    and step-by-step logits + hidden states, ring form against the register-streaming k_gemv_ln"""
    hp = dict(synth.HP_7B); hp["n_layer"] = 3; hp["n_vocab"] = 4096
    hp.update(WIDE[shape])
    w = synth.make_model_fast(hp, t, seed=5)
    toks = synth.tokens(12, hp["n_vocab"], seed=9)
    res = {}
    for mode in (1, 5):                                              # (1 = three launches per block: k_gemv_ln in its register-streaming form)
        m = g.FalconModel(w, n_ctx=64, n_batch=16)
        m.set_fused(mode)
        m.eval(toks, 0)
        lg, hid = m.eval(toks[-1:], 12, want_hidden=True)
        lg2, hid2 = m.eval(np.array([int(lg[0].argmax())], np.int32), 13, want_hidden=True)
        dev = m.decode_greedy(int(lg2[0].argmax()), 14, 24, use_graph=True)
        assert m.sync_error() == 0
        res[mode] = (lg, hid, lg2, hid2, dev)
        m.free()
    for a, b in zip(res[1], res[5]):
        assert np.array_equal(a, b)
