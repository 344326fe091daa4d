"""GPU: the k-quant mat-vec in the reference's OWN association at wave speed (csrc/kernels_kqref.hip, ggml_hip_reference_order(2) for all five k-quants):
one term per super-block (Q2_K, k_quants.c:1267-1306) / eight float lanes of the elements e = l (mod 8) plus the mins' chain (Q4_K, Q5_K: k_quants.c:1999-2055,
2340-2400), every f32 operation the reference's in the reference's order. Bit-identical (==) with the oracle's order 0 (the restatement pinned to the reference's
scalar build), with the one-thread-per-output instrument (mode 1) and -- whole tiny models -- with the logits captured from the real reference."""
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


@pytest.mark.parametrize("t", KQ)
@pytest.mark.parametrize("K,M", [(256, 5), (512, 37), (2048, 300), (8192, 129), (32768, 66), (4608, 71)])
@pytest.mark.parametrize("N", [1, 2, 3, 7])
def test_kq_mat_vec_in_reference_order_vs_oracle_order0(oracle, t, K, M, N):
    rng = np.random.default_rng(K * 31 + M + N + t)
    w = synth.quantized_matrix(oracle, t, M, K, rng)
    x = (rng.standard_normal((N, K)) * rng.uniform(0.2, 3.0)).astype(np.float32)
    dw = g.Weight(t, w, K, M)
    L = g.load()
    res = {}
    for mode in (1, 2):
        L.ggml_hip_reference_order(mode)
        try:
            res[mode] = dw.mul_mat(x)
        finally:
            L.ggml_hip_reference_order(0)
    dw.free()
    exp = oracle.mul_mat(t, w, K, M, x, 4)                      # order 0 = the reference's scalar build
    assert np.array_equal(res[1], exp)
    assert np.array_equal(res[2], exp)


@pytest.mark.parametrize("name,hp,t,gfile", [("gqa_q4_K", synth.HP_TINY_GQA, ob.Q4_K, "tiny_models"), ("gqa_q2_K", synth.HP_TINY_GQA, ob.Q2_K, "tiny_models_all"),
                                             ("gqa_q5_K", synth.HP_TINY_GQA, ob.Q5_K, "tiny_models_all"), ("gqa_q6_K", synth.HP_TINY_GQA, ob.Q6_K, "tiny_models_all"), ("gqa_q3_K", synth.HP_TINY_GQA, ob.Q3_K, "tiny_models_all")])
def test_kq_models_in_fast_reference_order_equal_the_real_reference(oracle, golden, name, hp, t, gfile):
    """tiny Falcon models: prefill (8 tokens: the column-by-column form of the same kernel) and four decode steps under mode 2 == the reference's scalar build"""
    gt = golden[gfile]
    w = synth.make_model(oracle, hp, t, seed=1234)
    toks = gt[f"{name}_tokens"]
    m = g.FalconModel(w, n_ctx=64, n_batch=8)
    g.load().ggml_hip_reference_order(2)
    try:
        lr, hr = m.eval(toks[:8], 0, logits_all=True, want_hidden=True)
        dr = np.concatenate([m.eval(toks[i:i + 1], i, logits_all=True) for i in range(8, 12)])
    finally:
        g.load().ggml_hip_reference_order(0)
    m.free()
    assert np.array_equal(hr, gt[f"{name}_prefill_hidden_scalar"])
    assert np.array_equal(lr, gt[f"{name}_prefill_logits_scalar"])
    assert np.array_equal(dr, gt[f"{name}_decode_logits_scalar"])


@pytest.mark.parametrize("t", [ob.Q4_K, ob.Q2_K])
def test_kq_falcon40b_width_decode_mode2_equals_mode1(t):
    """three blocks at Falcon-40B width (8192 / 32768, GQA 128 / 8, two norms): a 12-token prompt (mode 1 for both: the prompt is not this kernel's case) and decode
    steps + greedy tokens: mode 2 == mode 1, bit for bit"""
    hp = dict(synth.HP_40B); hp["n_layer"] = 3; hp["n_vocab"] = 4096
    w = synth.make_model_fast(hp, t, seed=5)
    toks = synth.tokens(12, hp["n_vocab"], seed=9)
    res = {}
    for mode in (1, 2):
        m = g.FalconModel(w, n_ctx=64, n_batch=16)
        g.load().ggml_hip_reference_order(mode)
        try:
            m.eval(toks, 0)
            lg, hid = m.eval(toks[-1:], 12, want_hidden=True)
            lg2, hid2 = m.eval(np.array([int(lg[0].argmax())], np.int32), 13, want_hidden=True)
            dev = m.decode_greedy(int(lg2[0].argmax()), 14, 6, use_graph=True)
            assert m.sync_error() == 0
        finally:
            g.load().ggml_hip_reference_order(0)
        res[mode] = (lg, hid, lg2, hid2, dev)
        m.free()
    for a, b in zip(res[1], res[2]):
        assert np.array_equal(a, b)
