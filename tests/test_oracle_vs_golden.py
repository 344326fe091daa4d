"""Pins the oracle (oracle/oracle_*.c) against golden vectors captured from the REAL reference
(tests/golden/*.npz, written by oracle/gen_golden.py from oracle/_ref). CPU only."""
import hashlib

import numpy as np
import pytest

from oracle import binding as ob
import synth


def relrms(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.sqrt((b ** 2).mean()) + 1e-30))


@pytest.mark.parametrize("t", ob.LEGACY)
@pytest.mark.parametrize("data", ["cos", "gau"])
def test_legacy_weight_quantizers_bit_exact(oracle, golden, t, data):
    g = golden["quant_fns"]
    nm = ob.TYPE_NAME[t]
    assert np.array_equal(oracle.quantize(t, g["x_" + data]), g[f"{nm}_{data}_q"])


@pytest.mark.parametrize("t", ob.WEIGHT_TYPES)
@pytest.mark.parametrize("data", ["cos", "gau"])
def test_dequantize_bit_exact(oracle, golden, t, data):
    g = golden["quant_fns"]
    nm = ob.TYPE_NAME[t]
    assert np.array_equal(oracle.dequantize(t, g[f"{nm}_{data}_q"], 4096), g[f"{nm}_{data}_deq"])


@pytest.mark.parametrize("t", ob.WEIGHT_TYPES)
def test_activation_quantizers_bit_exact(oracle, golden, t):
    """reference flavour == the reference's scalar build; AVX flavour == its AVX2 build."""
    g = golden["quant_fns"]
    nm = ob.TYPE_NAME[t]
    at = ob.VEC_DOT[t]
    assert np.array_equal(oracle.quantize_act(at, g["x_cos1"], ob.ROUND_REFERENCE), g[f"{nm}_act_scalar"])
    assert np.array_equal(oracle.quantize_act(at, g["x_cos1"], ob.ROUND_AVX), g[f"{nm}_act_avx"])


@pytest.mark.parametrize("t", ob.WEIGHT_TYPES)
def test_vec_dot(oracle, golden, t):
    g = golden["quant_fns"]
    nm = ob.TYPE_NAME[t]
    for data in ("cos", "gau"):
        got = oracle.vec_dot(t, 4096, g[f"{nm}_{data}_q"], g[f"{nm}_act_scalar"])
        exp_s, exp_a = float(g[f"{nm}_{data}_dot_scalar"]), float(g[f"{nm}_{data}_dot_avx"])
        assert got == exp_s                          # every format: the reference's scalar branch, bit for bit
        scale = max(abs(exp_s), 1e-3 * 4096 * 0.02)
        assert abs(got - exp_a) <= 2e-4 * scale      # different activation rounding flavour + 8-lane sums


@pytest.mark.parametrize("K", [4544, 18176])
@pytest.mark.parametrize("t", ob.WEIGHT_TYPES)
def test_vec_dot_falcon_row_lengths(oracle, golden, t, K):
    if K % ob.BLCK[t]:
        pytest.skip("k-quants need K % 256 == 0 (libfalcon.cpp:3626-3636)")
    g = golden["quant_fns"]
    nm = ob.TYPE_NAME[t]
    act = oracle.quantize_act(ob.VEC_DOT[t], g[f"x_{K}"], ob.ROUND_REFERENCE)
    got = np.array([oracle.vec_dot(t, K, g[f"{nm}_{K}_q"][r], act) for r in range(3)], np.float32)
    ref = g[f"{nm}_{K}_dot_scalar"]
    norm = 0.02 * np.sqrt(K)          # rms of a dot of N(0,.02^2) weights with N(0,1) activations
    assert np.array_equal(got, ref)
    assert np.abs(got - g[f"{nm}_{K}_dot_avx"]).max() <= 1e-3 * norm


@pytest.mark.parametrize("t", ob.WEIGHT_TYPES)
def test_mul_mat_graph(oracle, golden, t):
    g = golden["mul_mat"]
    nm = ob.TYPE_NAME[t]
    y = oracle.mul_mat(t, g[f"{nm}_w"], 512, 48, g[f"{nm}_x"], 3, ob.ROUND_REFERENCE)
    assert np.array_equal(y, g[f"{nm}_y_scalar"])
    ya = oracle.mul_mat(t, g[f"{nm}_w"], 512, 48, g[f"{nm}_x"], 2, ob.ROUND_AVX)
    assert relrms(ya, g[f"{nm}_y_avx"]) < 1e-5


def test_norm_gelu_rope_softmax_bit_exact(oracle, golden):
    g = golden["block_ops"]
    assert np.array_equal(oracle.norm(g["norm_x"]), g["norm_y"])
    assert np.array_equal(oracle.gelu(g["gelu_x"]), g["gelu_y"])
    tab = oracle.gelu_table()
    assert np.array_equal(tab[g["gelu_all_in_bits"]], g["gelu_all_out_bits"])
    for n_ctx in (2048, 8192):
        assert np.array_equal(oracle.rope(g[f"rope_x_{n_ctx}"], 64, 5, 3, 1021, n_ctx), g[f"rope_y_{n_ctx}"])
    kq = g["sm_kq"] * np.float32(0.125)
    n_past = int(g["sm_n_past"])
    for j in range(kq.shape[1]):
        kq[:, j, n_past + j + 1:] = -np.inf
    assert np.array_equal(oracle.softmax_rows(kq), g["sm_p"])


def test_rope_table_matches_rope(oracle):
    """the host-side cos/sin table the product uploads reproduces the in-loop cosf/sinf exactly"""
    rng = np.random.default_rng(5)
    for n_ctx in (2048, 8192):
        x = rng.standard_normal((4, 3, 64)).astype(np.float32)
        cs = oracle.rope_table(64, 40, n_ctx)
        y = oracle.rope(x, 64, 3, 4, 30, n_ctx)
        for t in range(4):
            c, s = cs[30 + t, :, 0], cs[30 + t, :, 1]
            x0, x1 = x[t, :, :32], x[t, :, 32:]
            assert np.array_equal(y[t, :, :32], x0 * c - x1 * s)
            assert np.array_equal(y[t, :, 32:], x0 * s + x1 * c)


def _digest(w):
    h = hashlib.sha256()
    for name in ("tok_emb", "lm_head", "out_norm_w", "out_norm_b"):
        h.update(np.ascontiguousarray(w[name]).tobytes())
    for lw in w["layers"]:
        for name in sorted(lw):
            h.update(np.ascontiguousarray(lw[name]).tobytes())
    return np.frombuffer(h.digest(), np.uint8)


CASES = [("mqa_q4_0", synth.HP_TINY_MQA, ob.Q4_0), ("gqa_q5_1", synth.HP_TINY_GQA, ob.Q5_1),
         ("gqa_q4_K", synth.HP_TINY_GQA, ob.Q4_K), ("mqa_q8_0", synth.HP_TINY_MQA, ob.Q8_0)]


@pytest.mark.parametrize("name,hp,t", CASES)
def test_tiny_falcon_end_to_end(oracle, golden, name, hp, t):
    """whole decoder stack (prefill 8 + 4 decode steps) against the reference's scalar build"""
    g = golden["tiny_models"]
    w = synth.make_model(oracle, hp, t, seed=1234)
    assert np.array_equal(_digest(w), g[f"{name}_digest"]), "synthetic weights drifted from the fixture's"
    toks = g[f"{name}_tokens"]
    m = oracle.model(w, 64)
    lg, hid = m.eval(toks[:8], 0, 2, ob.ROUND_REFERENCE, want_hidden=True)
    dec = np.concatenate([m.eval(toks[i:i + 1], i, 2, ob.ROUND_REFERENCE) for i in range(8, 12)])
    assert np.array_equal(hid, g[f"{name}_prefill_hidden_scalar"])
    assert np.array_equal(lg, g[f"{name}_prefill_logits_scalar"])
    assert np.array_equal(dec, g[f"{name}_decode_logits_scalar"])


CASES_ALL = [("gqa_q4_1", ob.Q4_1), ("gqa_q5_0", ob.Q5_0), ("gqa_q2_K", ob.Q2_K), ("gqa_q3_K", ob.Q3_K), ("gqa_q5_K", ob.Q5_K), ("gqa_q6_K", ob.Q6_K)]


@pytest.mark.parametrize("name,t", CASES_ALL)
def test_tiny_falcon_remaining_formats_end_to_end(oracle, golden, name, t):
    """the other six weight formats (tests/golden/tiny_models_all.npz, same recipe): bit-identical with the reference's
    scalar build -- so all ten formats are pinned at model level against logits of the real reference"""
    g = golden["tiny_models_all"]
    w = synth.make_model(oracle, synth.HP_TINY_GQA, t, seed=1234)
    assert np.array_equal(_digest(w), g[f"{name}_digest"]), "synthetic weights drifted from the fixture's"
    toks = g[f"{name}_tokens"]
    m = oracle.model(w, 64)
    lg, hid = m.eval(toks[:8], 0, 2, ob.ROUND_REFERENCE, want_hidden=True)
    dec = np.concatenate([m.eval(toks[i:i + 1], i, 2, ob.ROUND_REFERENCE) for i in range(8, 12)])
    assert np.array_equal(hid, g[f"{name}_prefill_hidden_scalar"])
    assert np.array_equal(lg, g[f"{name}_prefill_logits_scalar"])
    assert np.array_equal(dec, g[f"{name}_decode_logits_scalar"])
