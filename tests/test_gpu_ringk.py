"""GPU: the ring forms of csrc/kernels_ringk.hip -- k_ring_ln_k (the k-quants' LayerNorm mat-vec launch with the weights streamed by an
LDS-DMA loader wave through the LayerNorm + Q8_K prologue; gelu(up)'s Q8_K image rides on the attention launch) and k_ring_out (the
output mat-vec launch, all ten formats) -- reproduce the op-by-op launch list and the register-streaming fused kernels bit for bit:
logits, the hidden state of every block, greedy tokens through the hipGraph. Arithmetic: ggml_vec_dot_q*_K_q8_K (k_quants.c:1267-1306,
1684-1746, 1999-2055, 2340-2400, 2748-2789), quantize_row_q8_K_reference (k_quants.c:899-934), falcon_eval_internal's block
(libfalcon.cpp:2160-2400)."""
import numpy as np
import pytest

import ggllm_cpp_amd as g
from oracle import binding as ob
import synth

pytestmark = pytest.mark.gpu

KQ = [ob.Q2_K, ob.Q3_K, ob.Q4_K, ob.Q5_K, ob.Q6_K]


@pytest.fixture(scope="module", autouse=True)
def _init():
    g.init(0)


def _tiny(hp):
    """the tiny shapes at a super-block-sized width (the k-quants need n_embd % 256 == 0)"""
    hp = dict(hp)
    if hp["n_embd"] % 256:
        hp["n_embd"] = 256 * ((hp["n_embd"] + 255) // 256)
        hp["n_head"] = hp["n_embd"] // 64
        hp["n_head_kv"] = 1 if hp["n_head_kv"] == 1 else 2
        hp["n_ff"] = 4 * hp["n_embd"]
    return hp


@pytest.mark.parametrize("t", KQ)
@pytest.mark.parametrize("base", ["mqa", "gqa"])
def test_ringk_tiny_models_equal_op_list(oracle, base, t):
    hp = _tiny(synth.HP_TINY_MQA if base == "mqa" else synth.HP_TINY_GQA)
    w = synth.make_model(oracle, hp, t, seed=21)
    toks = synth.tokens(11, hp["n_vocab"], seed=6)
    outs = []
    for mode in (0, 1, 5):                      # op list | three launches, register-streaming kernels | ring forms
        m = g.FalconModel(w, n_ctx=32, n_batch=4)
        m.set_fused(mode)
        m.eval(toks[:4], 0)
        r = [m.eval(toks[i:i + 1], i, want_hidden=True) for i in range(4, 11)]
        assert m.sync_error() == 0
        outs.append(r)
        m.free()
    for other in outs[1:]:
        for (la, ha), (lb, hb) in zip(outs[0], other):
            assert np.array_equal(ha, hb)
            assert np.array_equal(la, lb)


W40 = dict(n_embd=8192, n_head=128, n_head_kv=8, n_ff=32768, two_norms=True)
WIDE = {"40b": W40, "1norm": dict(n_embd=4608, n_head=72, n_head_kv=1, n_ff=18432, two_norms=False)}


@pytest.mark.parametrize("shape,t", [("40b", ob.Q4_K), ("40b", ob.Q2_K), ("40b", ob.Q3_K), ("40b", ob.Q5_K), ("40b", ob.Q6_K), ("1norm", ob.Q4_K), ("1norm", ob.Q6_K),
                                     ("40b", ob.Q5_1), ("40b", ob.Q4_0), ("40b", ob.Q8_0)])
def test_ringk_full_width_blocks(shape, t):
    """two Falcon-40B-wide blocks (GQA 128 / 8, two norms: the grid of the merged launch does not fit the chip, so a block is three launches)
    and a one-norm 4608-wide shape (the merged two-launch form): step-by-step logits + hidden states and 24 greedy steps through the
    hipGraph, ring forms (mode 5; legacy formats: k_gemv_ln_ring + k_ring_out) against the register-streaming kernels (mode 1)"""
    hp = dict(synth.HP_40B); hp["n_layer"] = 2; hp["n_vocab"] = 4096
    hp.update(WIDE[shape])
    w = synth.make_model_fast(hp, t, seed=5)
    toks = synth.tokens(12, hp["n_vocab"], seed=9)
    res = {}
    for mode in (1, 5):
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


def test_ringk_against_the_oracle_at_40b_width(oracle):
    """one Falcon-40B-wide Q4_K block, decode steps: the ring forms' logits == the oracle evaluating the same steps in backend order"""
    hp = dict(synth.HP_40B); hp["n_layer"] = 1; hp["n_vocab"] = 512
    w = synth.make_model(oracle, hp, ob.Q4_K, seed=3)
    toks = synth.tokens(6, hp["n_vocab"], seed=4)
    m = g.FalconModel(w, n_ctx=16, n_batch=4)
    m.set_fused(5)
    m.eval(toks[:3], 0)
    got = [m.eval(toks[i:i + 1], i)[0] for i in range(3, 6)]
    assert m.sync_error() == 0
    m.free()
    oracle.lib.orc_set_sum_order(2)
    try:
        mo = oracle.model(w, 16)
        mo.eval(toks[:3], 0, 16)
        want = [mo.eval(toks[i:i + 1], i, 16)[0] for i in range(3, 6)]
    finally:
        oracle.lib.orc_set_sum_order(0)
    for a, b in zip(got, want):
        assert np.array_equal(a, b)


_SYS_SCRIPT = r"""
import sys, numpy as np
sys.path[:0] = [%(root)r, %(tests)r]
import ggllm_cpp_amd as g
import synth
from oracle import binding as ob
g.init(0)
hp = dict(synth.HP_40B); hp["n_layer"] = 2; hp["n_vocab"] = 4096
t = int(sys.argv[1])
w = synth.make_model_fast(hp, t, seed=5)
toks = synth.tokens(12, hp["n_vocab"], seed=9)
res = {}
for mode in (1, 5):
    m = g.FalconModel(w, n_ctx=64, n_batch=16)
    m.set_fused(mode)
    m.eval(toks, 0)
    lg, hid = m.eval(toks[-1:], 12, want_hidden=True)
    dev = m.decode_greedy(int(lg[0].argmax()), 13, 16, use_graph=True)
    assert m.sync_error() == 0
    res[mode] = (lg, hid, dev)
    m.free()
for a, b in zip(res[1], res[5]):
    assert np.array_equal(a, b)
print("systolic ok")
"""


@pytest.mark.parametrize("t", [ob.Q4_K, ob.Q2_K])
def test_systolic_output_form_is_bit_identical(tmp_path, t):
    """k_ring_out_sys (the systolic chain of consumers, opt-in: FQ_RING_OUT_SYS=1 -- measured slower) still reproduces the register-streaming
    kernels: run in a child process, the switch is read once per process"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "sys.py"
    script.write_text(_SYS_SCRIPT % {"root": root, "tests": os.path.join(root, "tests")})
    env = dict(os.environ, FQ_RING_OUT_SYS="1", FALCON_HIP_RING_OUT="1")
    r = subprocess.run([sys.executable, str(script), str(t)], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "systolic ok" in r.stdout
