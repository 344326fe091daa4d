"""GPU parity: weight re-tiling + dequantize_row, activation quantizers -- bit-exact against the oracle."""
import numpy as np
import pytest

import ggllm_cpp_amd as g
from oracle import binding as ob
import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _init():
    g.init(0)


@pytest.mark.parametrize("t", ob.WEIGHT_TYPES)
def test_dequantize_rows_bit_exact(oracle, t):
    rng = np.random.default_rng(11 + t)
    K, M = 1024, 37
    w = synth.quantized_matrix(oracle, t, M, K, rng)
    dw = g.Weight(t, w, K, M)
    got = dw.dequantize()
    exp = np.stack([oracle.dequantize(t, w[r], K) for r in range(M)])
    assert np.array_equal(got, exp)
    rows = [36, 0, 5, 5, 17]                       # get_rows semantics (ggml.c:11975): arbitrary, repeated indices
    assert np.array_equal(dw.dequantize(rows), exp[rows])
    dw.free()


@pytest.mark.parametrize("t", ob.WEIGHT_TYPES)
def test_dequantize_golden(golden, t):
    gq = golden["quant_fns"]
    nm = ob.TYPE_NAME[t]
    for data in ("cos", "gau"):
        dw = g.Weight(t, gq[f"{nm}_{data}_q"], 4096, 1)
        assert np.array_equal(dw.dequantize()[0], gq[f"{nm}_{data}_deq"])
        dw.free()


@pytest.mark.parametrize("K", [4544, 18176])
def test_dequantize_falcon_row_lengths(oracle, K):
    rng = np.random.default_rng(K)
    for t in ob.LEGACY:
        w = synth.quantized_matrix(oracle, t, 3, K, rng)
        dw = g.Weight(t, w, K, 3)
        assert np.array_equal(dw.dequantize(), np.stack([oracle.dequantize(t, w[r], K) for r in range(3)]))
        dw.free()


@pytest.mark.parametrize("at", [ob.Q8_0, ob.Q8_1, ob.Q8_K])
@pytest.mark.parametrize("K", [256, 4608, 8192, 18176 + 256 * 1])
def test_activation_quantizers_bit_exact(oracle, at, K):
    if K % ob.BLCK[at]:
        pytest.skip("block size")
    rng = np.random.default_rng(K + at)
    x = rng.standard_normal((5, K)).astype(np.float32) * rng.uniform(0.01, 30, size=(5, 1)).astype(np.float32)
    x[1] = 0.0                                     # all-zero row: d = 0 path
    x[2, :64] = 0.0                                # one all-zero block
    x[3, 7] = 1e-30                                # tiny values
    x[4] = np.clip(x[4], -100, 100)
    x[4, ::2] = np.round(x[4, ::2] * 4) / 4        # many exact .5 ties after scaling
    x[4, 0] = 127.0
    x[4, 1] = -127.0                               # equal magnitudes, first one wins (Q8_K sign rule)
    got = g.quantize_acts(at, x)
    exp = np.stack([oracle.quantize_act(at, x[i], ob.ROUND_REFERENCE) for i in range(5)])
    assert np.array_equal(got, exp)


@pytest.mark.parametrize("t", [ob.Q4_0, ob.Q4_1, ob.Q4_K])
def test_activation_quantizers_golden(golden, t):
    gq = golden["quant_fns"]
    got = g.quantize_acts(ob.VEC_DOT[t], gq["x_cos1"][None, :])
    assert np.array_equal(got[0], gq[f"{ob.TYPE_NAME[t]}_act_scalar"])


def test_activation_quantizer_4544(oracle):
    rng = np.random.default_rng(3)
    x = rng.standard_normal((3, 4544)).astype(np.float32)
    for at in (ob.Q8_0, ob.Q8_1):
        assert np.array_equal(g.quantize_acts(at, x), np.stack([oracle.quantize_act(at, x[i]) for i in range(3)]))
