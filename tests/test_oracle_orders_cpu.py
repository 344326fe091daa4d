"""CPU: the oracle's restated summation orders (orc_set_sum_order: 1 = the mat-vec kernels' unit-per-lane order, 2 = as the
backend picks per mat-mul, 3 / 4 = the GEMM's four / two interleaved partial sums) are re-associations of the SAME terms
as the reference order 0: every order agrees with order 0 within the f32 association tolerance, for all ten formats, and
where every term is an integer below 2^24 they agree exactly."""
import numpy as np
import pytest

from oracle import binding as ob
import synth

TOL = 2e-5


def relrms(a, b):
    return float(np.abs(a.astype(np.float64) - b).max() / (np.sqrt((b.astype(np.float64) ** 2).mean()) + 1e-30))


@pytest.mark.parametrize("t", ob.WEIGHT_TYPES)
@pytest.mark.parametrize("K,M,N", [(512, 9, 1), (1024, 7, 3), (4608, 5, 6), (8192, 3, 40)])
def test_orders_are_reassociations(oracle, t, K, M, N):
    if K % ob.BLCK[t]:
        pytest.skip("k-quants need K % 256 == 0")
    rng = np.random.default_rng(K + M + N + t)
    w = synth.quantized_matrix(oracle, t, M, K, rng)
    x = rng.standard_normal((N, K)).astype(np.float32)
    ref = oracle.mul_mat(t, w, K, M, x, 2)
    for order in (1, 2, 3, 4):
        oracle.lib.orc_set_sum_order(order)
        try:
            got = oracle.mul_mat(t, w, K, M, x, 2)
        finally:
            oracle.lib.orc_set_sum_order(0)
        assert relrms(got, ref) <= TOL, (order, relrms(got, ref))
        assert not np.array_equal(got, np.zeros_like(got))


@pytest.mark.parametrize("t", [ob.Q4_0, ob.Q8_0])
def test_orders_exact_on_integer_terms(oracle, t):
    """scale-1 blocks: every partial sum is an integer < 2^24, so every order gives the same bits"""
    K, M = 1024, 5
    rng = np.random.default_rng(t)
    x = rng.integers(-127, 128, size=(2, K)).astype(np.float32)
    x[:, ::32] = 127.0
    if t == ob.Q8_0:
        blk = np.zeros((M, K // 32, 34), np.uint8)
        blk[:, :, 0:2] = np.frombuffer(np.float16(1.0).tobytes(), np.uint8)
        blk[:, :, 2:] = rng.integers(-20, 21, size=(M, K // 32, 32)).astype(np.int8).view(np.uint8)
    else:
        blk = np.zeros((M, K // 32, 18), np.uint8)
        blk[:, :, 0:2] = np.frombuffer(np.float16(1.0).tobytes(), np.uint8)
        blk[:, :, 2:] = rng.integers(0, 256, size=(M, K // 32, 16), dtype=np.uint8)
    w = blk.reshape(M, -1)
    ref = oracle.mul_mat(t, w, K, M, x, 1)
    for order in (1, 2, 3, 4):
        oracle.lib.orc_set_sum_order(order)
        try:
            got = oracle.mul_mat(t, w, K, M, x, 1)
        finally:
            oracle.lib.orc_set_sum_order(0)
        assert np.array_equal(got, ref), order


@pytest.mark.parametrize("t", [ob.Q4_K, ob.Q5_K, ob.Q2_K, ob.Q3_K, ob.Q6_K])
def test_backend_mode_segments_for_q4k_small_batches(oracle, t):
    """mode 2 for Q4_K / Q5_K at model widths and 5..80 columns = the backend's small-batch form (k_gemm_skinny_q4k): four interleaved partial
    sums per segment of 32 super-blocks, the segments' values added left to right -- i.e. the order-3 results of the two K halves of a 64-super-block
    row, added; one segment (K = 8192) is order 3 itself (Q2_K / Q3_K: segments of 16, K = 4096); 4 columns and 113 columns keep the other rules (unit
    order / the tile GEMM's split)"""
    rng = np.random.default_rng(7 + t)
    K, M = (16384, 32) if t in (ob.Q4_K, ob.Q5_K) else (8192, 32)        # two segments: 32 super-blocks each (Q2_K: 16)
    w = synth.random_blocks(t, M, K, rng)
    x = rng.standard_normal((113, K)).astype(np.float32)
    half, rb = K // 2, w.shape[1] // 2
    try:
        oracle.lib.orc_set_sum_order(2)
        seg = oracle.mul_mat(t, w, K, M, x[:5], 2)
        four_cols = oracle.mul_mat(t, w, K, M, x[:4], 2)
        many = oracle.mul_mat(t, w, K, M, x, 2)
        one_seg = oracle.mul_mat(t, np.ascontiguousarray(w[:, :rb]), half, M, np.ascontiguousarray(x[:5, :half]), 2)
        oracle.lib.orc_set_sum_order(3)
        a = oracle.mul_mat(t, np.ascontiguousarray(w[:, :rb]), half, M, np.ascontiguousarray(x[:5, :half]), 2)
        b = oracle.mul_mat(t, np.ascontiguousarray(w[:, rb:]), half, M, np.ascontiguousarray(x[:5, half:]), 2)
        whole3 = oracle.mul_mat(t, w, K, M, x, 2)
        oracle.lib.orc_set_sum_order(1)
        unit = oracle.mul_mat(t, w, K, M, x[:4], 2)
    finally:
        oracle.lib.orc_set_sum_order(0)
    assert np.array_equal(seg, (a + b).astype(np.float32))
    assert np.array_equal(one_seg, a)
    assert np.array_equal(four_cols, unit)
    assert np.array_equal(many, whole3)              # 113 columns: the tile GEMM's four sums over the whole row (M = 32: few tiles)
    assert not np.array_equal(seg, whole3[:5])


def test_backend_mode_pass_limits_per_format(oracle):
    """the small-batch forms take up to 80 columns (Q4_K / Q5_K / Q6_K) or 112 (Q2_K / Q3_K: kernels.h fq_skinny_kq_max_cols); beyond, mode 2 is the
    tile GEMM's split over the whole row again"""
    K, M, N = 8192, 32, 100
    for t, segmented in ((ob.Q2_K, True), (ob.Q3_K, True), (ob.Q6_K, False), (ob.Q4_K, False)):
        rng = np.random.default_rng(11 + t)
        w = synth.random_blocks(t, M, K, rng)
        x = rng.standard_normal((N, K)).astype(np.float32)
        try:
            oracle.lib.orc_set_sum_order(2)
            got = oracle.mul_mat(t, w, K, M, x, 2)
            oracle.lib.orc_set_sum_order(3)
            whole = oracle.mul_mat(t, w, K, M, x, 2)
        finally:
            oracle.lib.orc_set_sum_order(0)
        assert np.array_equal(got, whole) != segmented, ob.TYPE_NAME[t]


@pytest.mark.parametrize("t", [ob.Q4_K, ob.Q2_K])
def test_backend_mode_from_three_columns_in_lock_step_contexts(oracle, t):
    """a lock-step context takes the small-batch form from 3 sequences up (fq_mul_mat_q_acts_from3): orc_set_kq_min_cols(3) moves mode 2's threshold; a
    column's value in that form does not depend on the other columns (3 columns = the first 3 of 5), and the default threshold keeps 3 columns in unit order"""
    rng = np.random.default_rng(21 + t)
    K, M = 16384, 32
    w = synth.random_blocks(t, M, K, rng)
    x = rng.standard_normal((5, K)).astype(np.float32)
    try:
        oracle.lib.orc_set_sum_order(2)
        five = oracle.mul_mat(t, w, K, M, x, 2)
        three_default = oracle.mul_mat(t, w, K, M, x[:3], 2)
        oracle.lib.orc_set_kq_min_cols(3)
        three = oracle.mul_mat(t, w, K, M, x[:3], 2)
        two = oracle.mul_mat(t, w, K, M, x[:2], 2)
        oracle.lib.orc_set_backend_batch(4)
        one_of_four = oracle.mul_mat(t, w, K, M, x[:1], 2)
        oracle.lib.orc_set_backend_batch(0)
        oracle.lib.orc_set_sum_order(1)
        unit = oracle.mul_mat(t, w, K, M, x[:3], 2)
    finally:
        oracle.lib.orc_set_sum_order(0); oracle.lib.orc_set_kq_min_cols(5); oracle.lib.orc_set_backend_batch(0)
    assert np.array_equal(three, five[:3])
    assert np.array_equal(one_of_four, five[:1])
    assert np.array_equal(three_default, unit)
    assert np.array_equal(two, unit[:2])
    assert not np.array_equal(three, unit)
