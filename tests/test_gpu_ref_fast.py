"""GPU: the FAST reference order (ggml_hip_reference_order(2), csrc/fq_ref_chain.h) -- the fused decode launches, the ring form and the int8-MFMA GEMM with
every row's per-block terms added LEFT TO RIGHT as the reference's scalar build adds them (ggml.c:2591-2609, 2719-2735, 2951-2972, 3207-3228, 3317-3329;
caller ggml.c:11484-11516), decode attention dots in f64 (ggml.c:2296-2300). Bit-identical (==) with
  * the logits captured from the REAL reference's scalar build (tests/golden/tiny_models*.npz),
  * the oracle in order 0 (the restatement pinned to that build),
  * the one-thread-per-output parity instrument (ggml_hip_reference_order(1)), at Falcon-7B and Falcon-40B widths, through the hipGraph replay too."""
import numpy as np
import pytest

import ggllm_cpp_amd as g
from oracle import binding as ob
import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _init():
    g.init(0)


class order:
    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        g.load().ggml_hip_reference_order(self.mode)

    def __exit__(self, *a):
        g.load().ggml_hip_reference_order(0)


LEGACY = [("mqa_q4_0", synth.HP_TINY_MQA, ob.Q4_0, "tiny_models"), ("gqa_q5_1", synth.HP_TINY_GQA, ob.Q5_1, "tiny_models"),
          ("mqa_q8_0", synth.HP_TINY_MQA, ob.Q8_0, "tiny_models"), ("gqa_q4_1", synth.HP_TINY_GQA, ob.Q4_1, "tiny_models_all"),
          ("gqa_q5_0", synth.HP_TINY_GQA, ob.Q5_0, "tiny_models_all")]


@pytest.mark.parametrize("fused", [2, 1, 0])
@pytest.mark.parametrize("name,hp,t,gfile", LEGACY)
def test_fast_reference_order_equals_the_real_reference(golden, name, hp, t, gfile, fused, oracle):
    """tiny Falcon models of the five legacy formats: prefill hidden states + logits and four decode steps == the reference's scalar build, whichever launch
    form the N = 1 steps take (2: merged attention + output launch, 1: three launches per block, 0: op list)"""
    gt = golden[gfile]
    w = synth.make_model(oracle, hp, t, seed=1234)
    toks = gt[f"{name}_tokens"]
    ref_h, ref_l, ref_d = gt[f"{name}_prefill_hidden_scalar"], gt[f"{name}_prefill_logits_scalar"], gt[f"{name}_decode_logits_scalar"]
    m = g.FalconModel(w, n_ctx=64, n_batch=8)
    m.set_fused(fused)
    with order(2):
        lr, hr = m.eval(toks[:8], 0, logits_all=True, want_hidden=True)
        dr = np.concatenate([m.eval(toks[i:i + 1], i, logits_all=True) for i in range(8, 12)])
        assert m.sync_error() == 0
    m.free()
    assert np.array_equal(hr, ref_h), "prefill hidden states differ from the reference's"
    assert np.array_equal(lr, ref_l), "prefill logits differ from the reference's"
    assert np.array_equal(dr, ref_d), "decode logits differ from the reference's"


WIDE = {"7b": dict(), "7b2n": dict(n_embd=4608, n_head=72, n_head_kv=2, n_ff=18432, two_norms=True),
        "40b": dict(n_embd=8192, n_head=128, n_head_kv=8, n_ff=32768, two_norms=True)}


@pytest.mark.parametrize("shape,t", [("7b", ob.Q4_0), ("7b", ob.Q5_1), ("7b2n", ob.Q8_0), ("7b2n", ob.Q4_1), ("7b", ob.Q5_0), ("40b", ob.Q5_1), ("40b", ob.Q4_0)])
def test_fast_reference_order_full_width(shape, t):
    """three blocks at Falcon-7B / 40B widths (ring form, merged or three-launch output, lm_head): a 12-token prompt, step-by-step logits + hidden states of
    every block, and 24 greedy steps through the hipGraph: mode 2 == mode 1 (the one-thread-per-output instrument, itself == the reference), bit for bit"""
    hp = dict(synth.HP_7B); hp["n_layer"] = 3; hp["n_vocab"] = 4096
    hp.update(WIDE[shape])
    w = synth.make_model_fast(hp, t, seed=5)
    toks = synth.tokens(12, hp["n_vocab"], seed=9)
    res = {}
    for mode in (1, 2):
        m = g.FalconModel(w, n_ctx=64, n_batch=16)
        with order(mode):
            lp, hp_ = m.eval(toks, 0, logits_all=True, want_hidden=True)
            lg, hid = m.eval(toks[-1:], 12, want_hidden=True)
            lg2, hid2 = m.eval(np.array([int(lg[0].argmax())], np.int32), 13, want_hidden=True)
            dev = m.decode_greedy(int(lg2[0].argmax()), 14, 24, use_graph=True)
            lg3 = m.eval(np.array([int(dev[-1])], np.int32), 38)
            assert m.sync_error() == 0
        res[mode] = (lp, hp_, lg, hid, lg2, hid2, dev, lg3)
        m.free()
    names = ("prefill logits", "prefill hidden", "step 12 logits", "step 12 hidden", "step 13 logits", "step 13 hidden", "greedy tokens", "step 38 logits")
    for n, a, b in zip(names, res[1], res[2]):
        assert np.array_equal(a, b), n


def test_fast_reference_order_vs_oracle_order0(oracle):
    """two Falcon-7B-wide Q4_0 blocks against the oracle in order 0 (the restatement pinned to the reference's scalar build): prefill + decode logits =="""
    hp = dict(n_vocab=2048, n_embd=4544, n_head=71, n_head_kv=1, n_layer=2, n_ff=18176, two_norms=False)
    w = synth.make_model(oracle, hp, ob.Q4_0, seed=1)
    toks = synth.tokens(6, hp["n_vocab"], seed=42)
    m = g.FalconModel(w, n_ctx=32, n_batch=4)
    with order(2):
        got = [m.eval(toks[:3], 0), m.eval(toks[3:4], 3), m.eval(toks[4:5], 4), m.eval(toks[5:6], 5)]
        assert m.sync_error() == 0
    m.free()
    oracle.lib.orc_set_sum_order(0)
    mo = oracle.model(w, 32)
    want = [mo.eval(toks[:3], 0, 8), mo.eval(toks[3:4], 3, 8), mo.eval(toks[4:5], 4, 8), mo.eval(toks[5:6], 5, 8)]
    for a, b in zip(got, want):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("t", [ob.Q4_0, ob.Q4_1, ob.Q5_0, ob.Q5_1, ob.Q8_0])
@pytest.mark.parametrize("K,M", [(512, 37), (4544, 70), (18176, 33)])
@pytest.mark.parametrize("N", [1, 2, 3, 4, 5, 17, 40])
def test_fast_reference_order_mul_mat_vs_oracle_order0(oracle, t, K, M, N):
    """op level, every column count: N = 1 the stand-alone mat-vec k_gemv_legacy_ref (csrc/kernels_kqref.hip: a block's term per lane, the strip [row][block] added left to
    right; the resident model's single-token steps run the fused launches instead), N >= 2 the int8-MFMA GEMM / the streaming small-batch forms with ONE left-to-right sum per
    row -- == the oracle's order 0 (the reference's scalar build)"""
    rng = np.random.default_rng(K * 31 + M + N + t)
    w = synth.quantized_matrix(oracle, t, M, K, rng)
    x = rng.standard_normal((N, K)).astype(np.float32)
    dw = g.Weight(t, w, K, M)
    with order(2):
        got = dw.mul_mat(x)
    dw.free()
    assert np.array_equal(got, oracle.mul_mat(t, w, K, M, x, 4))


@pytest.mark.parametrize("hp,t,B", [(synth.HP_TINY_MQA, ob.Q4_0, 3), (synth.HP_TINY_GQA, ob.Q5_1, 7), (synth.HP_TINY_MQA, ob.Q8_0, 12), (synth.HP_TINY_GQA, ob.Q4_1, 20),
                                    (synth.HP_TINY_GQA, ob.Q4_K, 5), (synth.HP_TINY_GQA, ob.Q2_K, 9)])
def test_fast_reference_order_is_batch_invariant_in_lock_step_contexts(oracle, hp, t, B):
    """the reference's association makes a row's result independent of the rows beside it (ggml.c:11484-11516: one vec_dot per row and column), so in the fast
    reference order B lock-step sequences through ONE weight pass per step give, for ANY B, exactly the logits of B contexts of their own (the default order promises
    that for the column mat-vec kernels only: B <= 4) -- and those are the one-thread-per-output instrument's, i.e. the reference's"""
    hp = dict(hp); hp["n_layer"] = 2
    if t in ob.KQUANTS and hp["n_embd"] % 256:
        hp["n_embd"] = 256 * ((hp["n_embd"] + 255) // 256); hp["n_head"] = hp["n_embd"] // 64; hp["n_head_kv"] = 2; hp["n_ff"] = 4 * hp["n_embd"]
    w = synth.make_model(oracle, hp, t, seed=17)
    streams = [synth.tokens(6, hp["n_vocab"], seed=50 + b) for b in range(B)]
    m = g.FalconModel(w, n_ctx=32, n_batch=8)
    with order(1):
        inst = np.stack([m.eval(streams[0][i:i + 1], i)[0] for i in range(6)])
    with order(2):
        singles = [np.stack([m.eval(streams[b][i:i + 1], i)[0] for i in range(6)]) for b in range(B)]
        sc = g.SeqContext(m, 32, B)
        rows = [sc.eval([int(streams[b][i]) for b in range(B)], i) for i in range(6)]
        sc.free()
        assert m.sync_error() == 0
    m.free()
    assert np.array_equal(singles[0], inst)
    for i in range(6):
        for b in range(B):
            assert np.array_equal(rows[i][b], singles[b][i]), (b, i)
