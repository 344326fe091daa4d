"""CPU: the weight quantizers' arithmetic (ggllm.cpp_amd/csrc/fq_wquant.h, what kernels_wquant.hip runs per thread) is
host-compiled (tests/host/wquant_harness.cpp) and must reproduce the reference's model-file quantizers byte for byte:
against the golden vectors of the reference build (tests/golden/wquant.npz, quant_fns.npz) and, where oracle/_ref exists,
against the reference itself on fresh inputs."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import binding as ob

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
KINDS = ["gauss", "uniform", "sparse", "heavy", "edges", "positive"]


@pytest.fixture(scope="module")
def wq():
    src = os.path.join(ROOT, "tests", "host", "wquant_harness.cpp")
    out = os.path.join(ROOT, "tests", "host", "libwquant_host.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-I",
                           os.path.join(ROOT, "ggllm.cpp_amd", "csrc"), "-o", out, src])
    L = C.CDLL(out)
    L.wquant_rows.argtypes = [C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
    L.wquant_f2h.restype = C.c_uint16
    L.wquant_f2h.argtypes = [C.c_float]

    def run(t, x, hist=False):
        x = np.ascontiguousarray(x, np.float32).ravel()
        out = np.zeros(ob.row_bytes(t, x.size), np.uint8)
        h = np.zeros(16, np.int64)
        assert L.wquant_rows(t, x.ctypes.data, x.size, out.ctypes.data, h.ctypes.data if hist else None) == 0
        return (out, h) if hist else out
    run.lib = L
    return run


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLD, "wquant.npz"))


@pytest.mark.parametrize("t", ob.WEIGHT_TYPES)
@pytest.mark.parametrize("kind", KINDS)
def test_blocks_and_histogram_match_reference_vectors(wq, gold, t, kind):
    q, h = wq(t, gold[f"x_{kind}"], hist=True)
    nm = ob.TYPE_NAME[t]
    assert np.array_equal(q, gold[f"{nm}_{kind}_q"])
    if t in (ob.Q5_0, ob.Q5_1):
        # the reference's Q5 histogram loop indexes qh with the ELEMENT-PAIR counter (ggml.c:19411-19413: bit j and bit
        # j + 16 for qs[j / 2], shifts past bit 31 from j = 16 on): not a function of the stored values. Ours bins the
        # stored 5-bit value / 2, which is what that code's comment ("cast to 16 bins") describes.
        assert h.sum() == gold[f"x_{kind}"].size
    else:
        assert np.array_equal(h, gold[f"{nm}_{kind}_hist"])      # all zero for the k-quants (they never count)


@pytest.mark.parametrize("t", ob.WEIGHT_TYPES)
def test_reference_test_vectors_of_quant_fns(wq, t):
    """the inputs tests/test-quantize-fns.cpp uses (0.1 + 2 cos(i)) and Falcon row lengths, quantized by the reference"""
    g = np.load(os.path.join(GOLD, "quant_fns.npz"))
    nm = ob.TYPE_NAME[t]
    for xn in ("cos", "gau"):
        assert np.array_equal(wq(t, g[f"x_{xn}"]), g[f"{nm}_{xn}_q"])
    for K in (4544, 18176):
        if K % ob.BLCK[t] == 0:
            assert np.array_equal(wq(t, g[f"w_{K}"]).reshape(3, -1), g[f"{nm}_{K}_q"])


def test_fp32_to_fp16_matches_oracle(wq, oracle):
    rng = np.random.default_rng(3)
    vals = rng.integers(0, 2 ** 32, 20000, dtype=np.uint64).astype(np.uint32).view(np.float32)
    vals = np.concatenate([vals, np.float32([0, -0.0, 65504, 65519.99, 65520, 1e-8, 5.96e-8, 2.98e-8, 2.9802322e-8, 6.1e-5, np.inf, -np.inf])])
    for v in vals:
        if np.isnan(v):
            continue
        assert wq.lib.wquant_f2h(float(v)) == oracle.fp32_to_fp16(float(v)), v


@pytest.mark.skipif(not ob.Ref.available(), reason="oracle/_ref not built (reference sources absent)")
@pytest.mark.parametrize("t", ob.WEIGHT_TYPES)
def test_fresh_inputs_against_the_reference_build(wq, t):
    R = ob.Ref()
    rng = np.random.default_rng(100 + t)
    for scale in (0.02, 1.0, 50.0):
        x = (rng.standard_normal(8192) * scale).astype(np.float32)
        q, h = wq(t, x, hist=True)
        rq, rh = R.quantize_chunk(t, x)
        assert np.array_equal(q, rq)
        assert np.array_equal(h, rh) or t in (ob.Q5_0, ob.Q5_1)   # see the note on the Q5 histogram above


@pytest.mark.skipif(not ob.Ref.available(), reason="oracle/_ref not built (reference sources absent)")
@pytest.mark.parametrize("t", [ob.Q2_K, ob.Q4_K, ob.Q5_K])
def test_only_known_difference_is_the_stale_levels_quirk(wq, t):
    """make_qkx1_quants compares its first pass with the PREVIOUS super-block's levels (never-cleared array,
    k_quants.c:235-241): a super-block that repeats its predecessor stops refining early in the reference. fq_wquant.h
    always refines (documented there); nothing else may differ."""
    R = ob.Ref()
    rng = np.random.default_rng(7)
    x = (rng.standard_normal(4096) * 0.02).astype(np.float32)
    x[1280:1536] = x[1024:1280]
    ours, theirs = wq(t, x), R.quantize(t, x)
    ts = ob.TSIZE[t]
    assert set(np.nonzero(ours != theirs)[0] // ts) <= {5}
