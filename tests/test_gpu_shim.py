"""GPU: the ggml-cuda.h boundary (include/dropin/ggml-cuda.h) behaves like the reference backend for its callers:
weights uploaded by ggml_cuda_transform_tensor, ggml_cuda_compute_forward called by every pool thread in every phase."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import ggllm_cpp_amd as g
from oracle import binding as ob
import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def shim():
    out = os.path.join(ROOT, "tests", "host", "libshim_harness.so")
    lib = os.path.dirname(g.LIB_PATH)
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-pthread", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "host", "shim_harness.cpp"), "-o", out, "-L" + lib, "-lggml_hip", "-Wl,-rpath," + lib])
    L = C.CDLL(out)
    L.shim_mul_mat.restype = C.c_int
    L.shim_mul_mat.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int]
    return L


@pytest.mark.parametrize("t", [ob.Q4_0, ob.Q5_1, ob.Q4_K, ob.Q6_K])
@pytest.mark.parametrize("N", [1, 3, 40])
def test_shim_mul_mat(oracle, shim, t, N):
    rng = np.random.default_rng(t + N)
    K, M = 1024, 96
    w = np.ascontiguousarray(synth.quantized_matrix(oracle, t, M, K, rng))
    x = rng.standard_normal((N, K)).astype(np.float32)
    y = np.zeros((N, M), np.float32)
    rc = shim.shim_mul_mat(t, w.ctypes.data, w.shape[1], K, M, x.ctypes.data, N, y.ctypes.data, 4)
    assert rc == 0
    exp = oracle.mul_mat(t, w, K, M, x, 4)
    assert float(np.abs(y - exp).max() / np.sqrt((exp.astype(np.float64) ** 2).mean())) <= 2e-5
