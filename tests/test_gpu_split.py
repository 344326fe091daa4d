"""GPU: row-split tensor parallelism (csrc/split_tp.hip, the reference's -ts / GGML_BACKEND_GPU_SPLIT): every rank's row range
of a quantized matrix uploaded on its own (ggml_hip_weight_upload_rows, the bytes ggml_cuda_transform_tensor would send it),
the parts multiplied one by one in this process -- the assembled result is the oracle's mat-mul (backend order) and the unsplit
mat-mul, bit for bit."""
import ctypes as C

import numpy as np
import pytest

import ggllm_cpp_amd as g
from oracle import binding as ob
import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _init():
    g.init(0)


@pytest.mark.parametrize("ts", [[1, 1], [3, 1, 2], [0, 0, 0, 0], [1, 0, 1], [1] * 8])
@pytest.mark.parametrize("t,N", [(ob.Q4_0, 1), (ob.Q4_0, 40), (ob.Q5_1, 3), (ob.Q4_K, 1), (ob.Q6_K, 40), (ob.Q8_0, 2)])
def test_row_split_equals_unsplit(oracle, ts, t, N):
    _row_split(oracle, ts, t, N, 1024, 1000 if t not in ob.KQUANTS else 777)


@pytest.mark.parametrize("ts", [[3, 1, 2], [1] * 8])
@pytest.mark.parametrize("t,N", [(ob.Q4_K, 8), (ob.Q4_K, 40), (ob.Q2_K, 12), (ob.Q6_K, 20), (ob.Q3_K, 3), (ob.Q5_K, 300), (ob.Q4_0, 12)])
def test_row_split_of_a_long_row_matrix(oracle, ts, t, N):
    """K = 16384 (64 super-blocks: two / four segments of the k-quants' small-batch form, whose partial sums are a different association from the tile GEMM's),
    M = 1040 (a multiple of 16, so the unsplit matrix takes that form at 5..80 / 112 columns) cut into ranges that are NOT multiples of 16 (the reference does
    not round them, ggml-cuda.cu:3046-3047): a part takes the form and the K split of the matrix it belongs to (fq_weight::form_M, zero rows up to a whole
    tile) -- the assembled rows equal the unsplit mat-mul and the oracle bit for bit"""
    _row_split(oracle, ts, t, N, 16384, 1040)


def _row_split(oracle, ts, t, N, K, M):
    L = g.load()
    rng = np.random.default_rng(t * 100 + N + len(ts))
    w = np.ascontiguousarray(synth.quantized_matrix(oracle, t, M, K, rng))
    x = rng.standard_normal((N, K)).astype(np.float32)
    whole = g.Weight(t, w, K, M)
    want = whole.mul_mat(x)
    n = len(ts)
    lo, hi = (C.c_int64 * n)(), (C.c_int64 * n)()
    L.ggml_hip_tensor_split_rows((C.c_float * n)(*ts), n, M, lo, hi)
    parts = (C.c_void_p * n)()
    for r in range(n):
        parts[r] = L.ggml_hip_weight_upload_rows(t, w.ctypes.data, K, M, lo[r], hi[r])
        assert bool(parts[r]) == (hi[r] > lo[r])
    xd, yd = g.DevBuf(x.nbytes), g.DevBuf(N * M * 4)
    L.ggml_hip_memcpy_h2d(xd.ptr, x.ctypes.data, x.nbytes)
    L.ggml_hip_memset(yd.ptr, 0xFF, N * M * 4)
    assert L.ggml_hip_mul_mat_q_split_local(parts, n, xd.ptr, K, N, yd.ptr, M, lo, hi) == 0
    got = yd.to_host(np.float32, (N, M))
    # the degenerate communicator (world 1) takes the same path as a rank of a real job, minus the exchange
    comm = L.ggml_hip_split_comm_create(0, 1, None)
    one_lo, one_hi = (C.c_int64 * 1)(0), (C.c_int64 * 1)(M)
    y1 = g.DevBuf(N * M * 4)
    assert L.ggml_hip_mul_mat_q_split(comm, whole.h, xd.ptr, K, N, y1.ptr, M, one_lo, one_hi) == 0
    got1 = y1.to_host(np.float32, (N, M))
    ts_bytes = (C.c_float * n)(*ts)
    assert L.ggml_hip_split_comm_agree(comm, ts_bytes, 4 * n) == 0      # one rank agrees with itself
    L.ggml_hip_split_comm_free(comm)
    for r in range(n):
        if parts[r]:
            L.ggml_hip_weight_free(parts[r])
    for b in (xd, yd, y1):
        b.free()
    whole.free()
    # against the oracle, not only against ourselves: the backend's association for this shape (wave order for N <= 4, the GEMM's K
    # split above; a part has fewer rows than the matrix but the same split: 4 partial sums below 1024 tiles)
    oracle.lib.orc_set_sum_order(2)
    try:
        exp = oracle.mul_mat(t, w, K, M, x, 8)
    finally:
        oracle.lib.orc_set_sum_order(0)
    assert np.array_equal(got, exp)
    assert np.array_equal(got, want)
    assert np.array_equal(got1, want)
