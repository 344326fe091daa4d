"""GPU: ggml_hip_quantize_rows / ggml_hip_weight_quantize (kernels_wquant.hip) against the reference's vectors and,
at model-sized matrices, against the host build of the same arithmetic + round-trip properties."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import ggllm_cpp_amd as g
from oracle import binding as ob

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
KINDS = ["gauss", "uniform", "sparse", "heavy", "edges", "positive"]


@pytest.fixture(scope="module", autouse=True)
def _init():
    g.init(0)


@pytest.fixture(scope="module")
def host_wq():
    src = os.path.join(ROOT, "tests", "host", "wquant_harness.cpp")
    out = os.path.join(ROOT, "tests", "host", "libwquant_host.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-I",
                           os.path.join(ROOT, "ggllm.cpp_amd", "csrc"), "-o", out, src])
    L = C.CDLL(out)
    L.wquant_rows.argtypes = [C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]

    def run(t, x):
        x = np.ascontiguousarray(x, np.float32).ravel()
        out = np.zeros(ob.row_bytes(t, x.size), np.uint8)
        h = np.zeros(16, np.int64)
        L.wquant_rows(t, x.ctypes.data, x.size, out.ctypes.data, h.ctypes.data)
        return out, h
    return run


@pytest.mark.parametrize("t", ob.WEIGHT_TYPES)
def test_reference_vectors(t):
    gold = np.load(os.path.join(GOLD, "wquant.npz"))
    nm = ob.TYPE_NAME[t]
    for kind in KINDS:
        q, h = g.quantize_rows(t, gold[f"x_{kind}"].reshape(1, -1), hist=True)
        assert np.array_equal(q.ravel(), gold[f"{nm}_{kind}_q"]), kind
        if t not in (ob.Q5_0, ob.Q5_1):                       # tests/test_wquant_cpu.py on the reference's Q5 histogram
            assert np.array_equal(h, gold[f"{nm}_{kind}_hist"]), kind
    qf = np.load(os.path.join(GOLD, "quant_fns.npz"))
    for xn in ("cos", "gau"):
        assert np.array_equal(g.quantize_rows(t, qf[f"x_{xn}"].reshape(1, -1)).ravel(), qf[f"{nm}_{xn}_q"])
    for K in (4544, 18176):
        if K % ob.BLCK[t] == 0:
            assert np.array_equal(g.quantize_rows(t, qf[f"w_{K}"]), qf[f"{nm}_{K}_q"])


@pytest.mark.parametrize("t", ob.WEIGHT_TYPES)
def test_matrix_matches_host_build_and_ragged_tail(host_wq, t):
    """a matrix whose block count is not a multiple of the workgroup's share (ragged last workgroup)"""
    rng = np.random.default_rng(t)
    K = 4608 if t in ob.KQUANTS else 4544
    M = 37
    x = (rng.standard_normal((M, K)) * 0.02).astype(np.float32)
    x[3] = 0
    x[5, :512] = 1.5
    q, h = g.quantize_rows(t, x, hist=True)
    hq, hh = host_wq(t, x)
    assert np.array_equal(q.ravel(), hq)
    assert np.array_equal(h, hh)


@pytest.mark.parametrize("t", ob.WEIGHT_TYPES)
def test_quantized_weight_round_trip(oracle, t):
    """quantize + re-tile on the device, read back through dequantize_rows == oracle dequantize of the block bytes; the
    quantization error passes the reference's own check (tests/test-quantize-fns.cpp:17-19, 30-38, 129-133: sqrt(sum of
    squared errors) / n over n = 4096 values of 0.1 + 2 cos(i) below 0.002, 0.0075 for Q2_K, 0.004 for Q3_K)"""
    rng = np.random.default_rng(50 + t)
    K, M = (2048, 24)
    x = (0.1 + 2.0 * np.cos(np.arange(M * K, dtype=np.float32))).astype(np.float32).reshape(M, K)
    w = g.Weight.quantize(t, x)
    y = w.dequantize()
    blocks = g.quantize_rows(t, x)
    for r in (0, 7, 23):
        assert np.array_equal(y[r], oracle.dequantize(t, blocks[r], K))
    n = 4096
    err = np.sqrt(np.sum((y.ravel()[:n].astype(np.float64) - x.ravel()[:n]) ** 2)) / n
    assert err < (0.0075 if t == ob.Q2_K else 0.004 if t == ob.Q3_K else 0.002), err
    # and the quantized matrix multiplies like the uploaded one
    w2 = g.Weight(t, blocks, K, M)
    xin = rng.standard_normal((2, K)).astype(np.float32)
    assert np.array_equal(w.mul_mat(xin), w2.mul_mat(xin))


def test_bad_row_length_is_refused():
    with pytest.raises(ValueError):
        g.quantize_rows(ob.Q4_K, np.zeros((1, 4544), np.float32))
