"""CPU: the per-format unit decoders the GEMV kernels compile (ggllm.cpp_amd/csrc/fq_units.h) are host-compiled with
g++ (tests/host/units_harness.cpp) and checked against the oracle -- index math of all ten formats, no GPU."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import binding as ob
import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def harness():
    src = os.path.join(ROOT, "tests", "host", "units_harness.cpp")
    out = os.path.join(ROOT, "tests", "host", "libunits_host.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-I", os.path.join(ROOT, "ggllm.cpp_amd", "csrc"), "-o", out, src])
    L = C.CDLL(out)
    L.units_row_dot.restype = C.c_float
    L.units_row_dot.argtypes = [C.c_int, C.c_int64, C.c_void_p, C.c_void_p]
    L.units_kdot_mismatches.restype = C.c_int
    L.units_kdot_mismatches.argtypes = [C.c_int, C.c_int64, C.c_void_p, C.c_void_p]
    return L


@pytest.mark.parametrize("t", ob.KQUANTS)
@pytest.mark.parametrize("K", [4096, 8192, 32768])
def test_ring_consumer_unit_dots_equal_the_generic_ones(oracle, harness, t, K):
    """csrc/fq_kdot.h (lane-constant index math hoisted, activation slices pre-loaded, packed scale decode: what the ring consumers of
    kernels_ringk.hip run) gives the SAME f32 term as fq_unit<TYPE>::dot for every unit of rows of whole columns"""
    rng = np.random.default_rng(7 * K + t)
    w = synth.quantized_matrix(oracle, t, 4, K, rng)
    for r in range(4):
        x = (rng.standard_normal(K) * (1.0 + 3.0 * r)).astype(np.float32)
        act = oracle.quantize_act(ob.VEC_DOT[t], x)
        row = np.ascontiguousarray(w[r])
        assert harness.units_kdot_mismatches(t, K, row.ctypes.data, act.ctypes.data) == 0


@pytest.mark.parametrize("t", ob.WEIGHT_TYPES)
@pytest.mark.parametrize("K", [256, 4544, 8192, 18176])
def test_unit_decode_matches_oracle(oracle, harness, t, K):
    if K % ob.BLCK[t]:
        pytest.skip("block size")
    rng = np.random.default_rng(K + t)
    w = synth.quantized_matrix(oracle, t, 3, K, rng)
    x = rng.standard_normal(K).astype(np.float32)
    act = oracle.quantize_act(ob.VEC_DOT[t], x)
    for r in range(3):
        row = np.ascontiguousarray(w[r])
        got = harness.units_row_dot(t, K, row.ctypes.data, act.ctypes.data)
        exp = oracle.vec_dot(t, K, row, act)
        norm = np.abs(oracle.dequantize(t, row, K)).mean() * np.sqrt(K)
        assert abs(got - exp) <= 3e-6 * max(norm, 1e-6)
        if t in ob.LEGACY:
            assert got == exp            # same left-to-right f32 association as the reference's scalar loop
