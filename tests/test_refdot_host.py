"""CPU: the reference-order row dot the backend compiles into kernels_ref.hip (ggllm.cpp_amd/csrc/fq_ref_dot.h) is
host-compiled with g++ and must reproduce, BIT FOR BIT, the dots captured from the real reference's scalar build
(tests/golden/quant_fns.npz) -- all ten formats, the Falcon row lengths included."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import binding as ob

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def harness():
    src = os.path.join(ROOT, "tests", "host", "refdot_harness.cpp")
    out = os.path.join(ROOT, "tests", "host", "librefdot_host.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-I", os.path.join(ROOT, "ggllm.cpp_amd", "csrc"), "-o", out, src])
    L = C.CDLL(out)
    L.refdot_row.restype = C.c_float
    L.refdot_row.argtypes = [C.c_int, C.c_int64, C.c_void_p, C.c_void_p]
    return L


@pytest.mark.parametrize("t", ob.WEIGHT_TYPES)
def test_reference_order_dot_equals_reference_scalar_build(harness, golden, t):
    g = golden["quant_fns"]
    nm = ob.TYPE_NAME[t]
    act = np.ascontiguousarray(g[f"{nm}_act_scalar"])
    for data in ("cos", "gau"):
        row = np.ascontiguousarray(g[f"{nm}_{data}_q"])
        got = harness.refdot_row(t, 4096, row.ctypes.data, act.ctypes.data)
        assert np.float32(got) == np.float32(g[f"{nm}_{data}_dot_scalar"])


@pytest.mark.parametrize("K", [4544, 18176])
@pytest.mark.parametrize("t", ob.WEIGHT_TYPES)
def test_reference_order_dot_falcon_rows(harness, oracle, golden, t, K):
    if K % ob.BLCK[t]:
        pytest.skip("k-quants need K % 256 == 0 (libfalcon.cpp:3626-3636)")
    g = golden["quant_fns"]
    nm = ob.TYPE_NAME[t]
    act = oracle.quantize_act(ob.VEC_DOT[t], g[f"x_{K}"], ob.ROUND_REFERENCE)
    for r in range(3):
        row = np.ascontiguousarray(g[f"{nm}_{K}_q"][r])
        got = harness.refdot_row(t, K, row.ctypes.data, act.ctypes.data)
        assert np.float32(got) == g[f"{nm}_{K}_dot_scalar"][r]
