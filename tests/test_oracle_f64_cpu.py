"""CPU: the f64 yardstick of the oracle (orc_set_sum_order(6), round 5): every reduction of the path accumulated in f64 from exactly converted terms, the
integer dots / activation quantizers / elementwise f32 steps unchanged. It is NOT an order any build runs -- it is what the f32 associations (the reference's
scalar and AVX2 builds, the backend's default order) are measured against (bench.py parity.err_vs_f64, tests/test_gpu_parity_f64.py). Here: it equals an
independent numpy f64 evaluation of the same sums, it sits within the association spread of the reference order, and the statistics the GPU test relies on
hold for the restated orders (ggml.c:2591-2609, k_quants.c:1999-2055)."""
import numpy as np
import pytest

from oracle import binding as ob
import synth


def _rows(oracle, t, M, K, seed):
    rng = np.random.default_rng(seed)
    w = np.ascontiguousarray(synth.quantized_matrix(oracle, t, M, K, rng))
    x = rng.standard_normal((3, K)).astype(np.float32)
    return w, x


@pytest.mark.parametrize("t", [ob.Q4_0, ob.Q4_1, ob.Q5_0, ob.Q5_1, ob.Q8_0])
def test_f64_order_equals_numpy_f64_of_the_same_terms_legacy(oracle, t):
    K, M = 4544 - 4544 % 32, 24
    w, x = _rows(oracle, t, M, K, 11 + t)
    oracle.lib.orc_set_sum_order(6)
    try:
        got = oracle.mul_mat(t, w, K, M, x)
    finally:
        oracle.lib.orc_set_sum_order(0)
    at = ob.VEC_DOT[t]
    wf = oracle.dequantize(t, w, K * M).reshape(M, K).astype(np.float64)          # d_w * q (+ m_w): exact in f64 per element
    for n in range(x.shape[0]):
        a = oracle.quantize_act(at, x[n])
        xq = oracle.dequantize(at, a, K).astype(np.float64) if at == ob.Q8_0 else None
        if xq is None:                                                             # Q8_1: d (f32) * q
            raw = np.frombuffer(a, np.uint8).reshape(K // 32, 40)
            d = raw[:, :4].copy().view(np.float32)[:, 0].astype(np.float64)
            xq = (raw[:, 8:].view(np.int8).astype(np.float64) * d[:, None]).reshape(K)
        want = (wf * xq[None, :]).sum(axis=1)
        # Q4_1 / Q5_1 add m_w * s_x with s_x = d_x * sum(q) rounded to f32 by the activation quantizer: identical up to that rounding
        tol = 0.0 if t in (ob.Q4_0, ob.Q5_0, ob.Q8_0) else 4e-7
        scale = np.sqrt((want ** 2).mean())
        assert np.abs(got[n].astype(np.float64) - want).max() <= tol * scale + np.abs(np.spacing(want.astype(np.float32))).max()


@pytest.mark.parametrize("t", [ob.Q4_0, ob.Q5_1, ob.Q2_K, ob.Q3_K, ob.Q4_K, ob.Q5_K, ob.Q6_K])
def test_every_f32_order_sits_within_the_association_spread_of_the_f64_sums(oracle, t):
    K, M = (4608 if t in ob.KQUANTS else 4544), 96
    w, x = _rows(oracle, t, M, K, 5 + t)
    res = {}
    for order in (6, 0, 1, 3, 4, 5):
        oracle.lib.orc_set_sum_order(order)
        try:
            res[order] = oracle.mul_mat(t, w, K, M, x).astype(np.float64)
        finally:
            oracle.lib.orc_set_sum_order(0)
    rms = np.sqrt((res[6] ** 2).mean())
    err = {o: np.sqrt(((res[o] - res[6]) ** 2).mean()) / rms for o in res if o != 6}
    assert all(e <= 2e-5 for e in err.values()), err
    # 64 partial sums (the mat-vec kernels' order 1) round less than one left-to-right chain (the reference's scalar order 0): the statistic behind the
    # GPU test's "the default order is no further from the f64 sums than the reference's own builds"
    assert err[1] <= err[0] * 1.05, err
