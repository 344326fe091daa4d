"""GPU parity: quantized mat-mul (ggml_compute_forward_mul_mat_q_f32) against the oracle and the golden vectors."""
import numpy as np
import pytest

import ggllm_cpp_amd as g
from oracle import binding as ob
import synth

pytestmark = pytest.mark.gpu

# fp32 tolerance: results may differ from the reference only by the association of the f32 sum over blocks.
# max |diff| <= TOL * rms(reference row); north_star asks for 1e-3 on logits, the kernels are held to 2e-5.
TOL = 2e-5


@pytest.fixture(scope="module", autouse=True)
def _init():
    g.init(0)


def relrms(a, b):
    return float(np.abs(a.astype(np.float64) - b).max() / (np.sqrt((b.astype(np.float64) ** 2).mean()) + 1e-30))


@pytest.mark.parametrize("t", ob.WEIGHT_TYPES)
@pytest.mark.parametrize("K,M", [(512, 37), (4544, 70), (8192, 129), (18176, 33)])
@pytest.mark.parametrize("N", [1, 2, 3, 5])
def test_mul_mat_vs_oracle(oracle, t, K, M, N):
    if K % ob.BLCK[t]:
        pytest.skip("k-quants need K % 256 == 0")
    rng = np.random.default_rng(K * 31 + M + N + t)
    w = synth.quantized_matrix(oracle, t, M, K, rng)
    x = rng.standard_normal((N, K)).astype(np.float32)
    dw = g.Weight(t, w, K, M)
    got = dw.mul_mat(x)
    exp = oracle.mul_mat(t, w, K, M, x, 4)
    dw.free()
    assert relrms(got, exp) <= TOL, relrms(got, exp)
    # and bit for bit against the oracle run with the backend's association of the same terms (N <= 4: the mat-vec kernels,
    # one f32 term per unit of fq_units.h, 64 lanes, ascending units per lane, xor butterfly; N = 5: the GEMM's split order)
    oracle.lib.orc_set_sum_order(2)
    try:
        exp_wave = oracle.mul_mat(t, w, K, M, x, 4)
    finally:
        oracle.lib.orc_set_sum_order(0)
    assert np.array_equal(got, exp_wave)


@pytest.mark.parametrize("t", ob.WEIGHT_TYPES)
def test_mul_mat_golden(golden, t):
    gm = golden["mul_mat"]
    nm = ob.TYPE_NAME[t]
    dw = g.Weight(t, gm[f"{nm}_w"], 512, 48)
    got = dw.mul_mat(gm[f"{nm}_x"])
    dw.free()
    assert relrms(got, gm[f"{nm}_y_scalar"]) <= TOL
    assert relrms(got, gm[f"{nm}_y_avx"]) <= 1e-3          # other rounding flavour of the activation quantizer


def _integer_case(t, K, M, rng):
    """weights/activations whose scales are exactly 1 so that every partial sum is an integer < 2^24:
    the result is then independent of summation order and must be BIT-EXACT."""
    x = rng.integers(-127, 128, size=(2, K)).astype(np.float32)
    x[:, ::32] = 127.0                                  # amax = 127 in every block -> d = 1 (Q8_0/Q8_1)
    if t == ob.Q8_0:
        blk = np.zeros((M, K // 32, 34), np.uint8)
        blk[:, :, 0:2] = np.frombuffer(np.float16(1.0).tobytes(), np.uint8)
        blk[:, :, 2:] = rng.integers(-20, 21, size=(M, K // 32, 32)).astype(np.int8).view(np.uint8)
        return blk.reshape(M, -1), x
    if t == ob.Q4_0:
        blk = np.zeros((M, K // 32, 18), np.uint8)
        blk[:, :, 0:2] = np.frombuffer(np.float16(1.0).tobytes(), np.uint8)
        blk[:, :, 2:] = rng.integers(0, 256, size=(M, K // 32, 16), dtype=np.uint8)
        return blk.reshape(M, -1), x
    raise AssertionError


@pytest.mark.parametrize("t", [ob.Q4_0, ob.Q8_0])
@pytest.mark.parametrize("K", [64, 4544, 18176])
def test_integer_dot_bit_exact(oracle, t, K):
    rng = np.random.default_rng(K + t)
    M = 67
    w, x = _integer_case(t, K, M, rng)
    dw = g.Weight(t, w, K, M)
    got = dw.mul_mat(x)
    dw.free()
    exp = oracle.mul_mat(t, w, K, M, x, 2)
    assert np.array_equal(got, exp)
    # and against plain integer arithmetic
    deq = np.stack([oracle.dequantize(t, w[r], K) for r in range(M)]).astype(np.int64)
    assert np.array_equal(got.astype(np.int64), x.astype(np.int64) @ deq.T)


GEMM_TYPES = [ob.Q4_0, ob.Q4_1, ob.Q5_0, ob.Q5_1, ob.Q8_0, ob.Q4_K, ob.Q5_K]


@pytest.mark.parametrize("t", ob.WEIGHT_TYPES)
@pytest.mark.parametrize("K,M,N", [(512, 37, 9), (4544, 200, 33), (8192, 129, 128), (1024, 300, 257), (18176, 70, 40),
                                   (4544, 200, 16), (18176, 70, 12), (8192, 129, 5), (4544, 100, 7), (4672, 33, 13),      # (N <= 16: the streaming form, kernels_gemm_skinny.hip)
                                   (4544, 200, 17), (8192, 129, 32), (18176, 70, 29),                              # (17..32: two passes of it)
                                   (4736, 70, 40), (4992, 70, 130), (4864, 40, 64)])                               # (37 / 39 / 38 K stages: the tails of the two-stages-per-barrier loop)
def test_prefill_gemm_vs_oracle(oracle, t, K, M, N):
    """N > 4 columns through the int8 MFMA GEMM, three orders, each bit-exact against the oracle's restatement of it:
      * default (4 or 2 interleaved K-split partial sums, picked per shape)      == orc_set_sum_order(2)
      * ggml_hip_gemm_sequential(1) (one partial sum: the legacy formats' blocks left to right = the reference's scalar
        vec_dot; k-quants one d * isum - dmin * msum term per super-block)        == orc_set_sum_order(5)
      * ggml_hip_reference_order(1) (kernels_ref.hip, the reference's scalar branches for all ten formats: Q3_K .. Q6_K keep
        eight float lanes)                                                        == order 0 == the reference's scalar build"""
    if K % ob.BLCK[t]:
        pytest.skip("k-quants need K % 256 == 0")
    rng = np.random.default_rng(K + M + N + t)
    w = synth.quantized_matrix(oracle, t, M, K, rng)
    x = rng.standard_normal((N, K)).astype(np.float32)
    dw = g.Weight(t, w, K, M)
    got = dw.mul_mat(x)
    exp = oracle.mul_mat(t, w, K, M, x, 8)
    g.load().ggml_hip_gemm_sequential(1)
    try:
        seq = dw.mul_mat(x)
    finally:
        g.load().ggml_hip_gemm_sequential(0)
    g.load().ggml_hip_reference_order(1)
    try:
        ref = dw.mul_mat(x)
    finally:
        g.load().ggml_hip_reference_order(0)
    oracle.lib.orc_set_sum_order(2)          # the backend's choice for this shape: 4 or 2 interleaved partial sums
    try:
        exp_split = oracle.mul_mat(t, w, K, M, x, 8)
        oracle.lib.orc_set_sum_order(5)
        exp_seq = oracle.mul_mat(t, w, K, M, x, 8)
    finally:
        oracle.lib.orc_set_sum_order(0)
    assert np.array_equal(ref, exp)
    assert np.array_equal(seq, exp_seq)
    if t in ob.LEGACY or t == ob.Q2_K:
        assert np.array_equal(seq, exp)      # there the single left-to-right sum IS the reference's order
    assert np.array_equal(got, exp_split)
    assert relrms(got, exp) <= TOL and relrms(seq, exp) <= TOL
    # and the same columns through the mat-vec kernel agree within the association tolerance
    g.load().ggml_hip_debug_force_gemv(1)
    try:
        via_gemv = dw.mul_mat(x)
    finally:
        g.load().ggml_hip_debug_force_gemv(0)
    dw.free()
    assert relrms(got, via_gemv) <= TOL


@pytest.mark.parametrize("K,M,N", [(2048, 64, 7), (2560, 48, 16), (8192, 144, 5), (10240, 32, 13), (16384, 16, 16), (8192, 144, 29), (4096, 64, 77), (4096, 32, 100), (2048, 20480, 6),
                                   (10240, 20480, 5)])
@pytest.mark.parametrize("t", [ob.Q4_K, ob.Q5_K, ob.Q2_K, ob.Q3_K, ob.Q6_K])
def test_q4k_small_batch_vs_oracle(oracle, t, K, M, N):
    """Q4_K and Q5_K (the same kernel plus the plane of fifth bits) Q2_K / Q3_K (k_gemm_skinny_q2k: all four shares per wave, the sub-block scales
    folded into the matrix operand, segments of 16 super-blocks) and Q6_K (k_gemm_skinny_q6k: the same frame, two half operands per group), 5..16 columns (17..80: passes) at model widths: the share-pair streaming form (k_gemm_skinny_q4k + k_skinny_sum4) -- four
    interleaved partial sums per segment of 32 super-blocks whatever the shape, the segments added left to right == orc_set_sum_order(2), which
    follows the same rule (== order 3 for rows of one segment). Shapes: one device column, a partial last column (10 super-blocks), one full segment,
    two segments (40 super-blocks), two full segments on one tile, two passes, more row blocks than workgroup slots, and the latter with two segments."""
    rng = np.random.default_rng(K + M + N + t)
    w = synth.random_blocks(t, M, K, rng) if M > 1024 else synth.quantized_matrix(oracle, t, M, K, rng)
    x = rng.standard_normal((N, K)).astype(np.float32)
    dw = g.Weight(t, w, K, M)
    got = dw.mul_mat(x)
    g.load().ggml_hip_gemm_sequential(1)
    try:
        seq = dw.mul_mat(x)
    finally:
        g.load().ggml_hip_gemm_sequential(0)
    dw.free()
    oracle.lib.orc_set_sum_order(2)
    try:
        exp_split = oracle.mul_mat(t, w, K, M, x, 8)
        oracle.lib.orc_set_sum_order(3)
        exp4 = oracle.mul_mat(t, w, K, M, x, 8)
        oracle.lib.orc_set_sum_order(5)
        exp_seq = oracle.mul_mat(t, w, K, M, x, 8)
    finally:
        oracle.lib.orc_set_sum_order(0)
    if K <= (8192 if t in (ob.Q4_K, ob.Q5_K) else 4096):
        assert np.array_equal(exp_split, exp4)      # one segment: exactly the tile GEMM's four-sum order
    assert np.array_equal(got, exp_split)
    assert np.array_equal(seq, exp_seq)
    assert relrms(got, exp_seq) <= TOL


@pytest.mark.parametrize("cfg,order", [(6, 4), (7, 3)])
@pytest.mark.parametrize("t", ob.WEIGHT_TYPES)
def test_prefill_gemm_64_row_workgroups(oracle, t, cfg, order, monkeypatch):
    """The 64-row workgroup forms (<2,4,2> = two partial sums per row, <4,4,2> = four; picked by the launcher from 32 x #CU
    tiles upwards, forced here through FQ_GEMM_CFG): the same sums in the same association as the 32-row forms, so the
    oracle run with that association pins them -- bit-exactly for the legacy formats."""
    if cfg == 7 and t in ob.KQUANTS:
        pytest.skip("<4,4,2> is not used for k-quants (register budget)")
    monkeypatch.setenv("FQ_GEMM_CFG", str(cfg))
    for K, M, N in [(512, 37, 9), (4544, 200, 33), (8192, 129, 130), (1024, 300, 257)]:
        if K % ob.BLCK[t]:
            continue
        rng = np.random.default_rng(K + M + N + t)
        w = synth.quantized_matrix(oracle, t, M, K, rng)
        x = rng.standard_normal((N, K)).astype(np.float32)
        dw = g.Weight(t, w, K, M)
        got = dw.mul_mat(x)
        dw.free()
        oracle.lib.orc_set_sum_order(order)
        try:
            exp = oracle.mul_mat(t, w, K, M, x, 8)
        finally:
            oracle.lib.orc_set_sum_order(0)
        assert np.array_equal(got, exp), (K, M, N)


def test_epilogues(oracle):
    L = g.load()
    rng = np.random.default_rng(5)
    K, M, N = 512, 100, 2
    t = ob.Q4_0
    w = synth.quantized_matrix(oracle, t, M, K, rng)
    x = rng.standard_normal((N, K)).astype(np.float32) * 3
    a1 = rng.standard_normal((N, M)).astype(np.float32)
    a2 = rng.standard_normal((N, M)).astype(np.float32)
    base = oracle.mul_mat(t, w, K, M, x, 2)
    dw = g.Weight(t, w, K, M)
    acts = L.ggml_hip_acts_alloc(ob.Q8_0, K, N)
    xb, yb, b1, b2 = g.DevBuf(host=x), g.DevBuf(N * M * 4), g.DevBuf(host=a1), g.DevBuf(host=a2)
    L.ggml_hip_quantize_acts(acts, xb.ptr, K, N)
    L.ggml_hip_mul_mat_q_acts(dw.h, acts, N, yb.ptr, M, 0, None, None)
    plain = yb.to_host(np.float32, (N, M))
    assert relrms(plain, base) <= TOL
    L.ggml_hip_mul_mat_q_acts(dw.h, acts, N, yb.ptr, M, 1, None, None)
    assert np.array_equal(yb.to_host(np.float32, (N, M)), oracle.gelu(plain))
    L.ggml_hip_mul_mat_q_acts(dw.h, acts, N, yb.ptr, M, 2, b1.ptr, b2.ptr)
    assert np.array_equal(yb.to_host(np.float32, (N, M)), (plain + a1) + a2)
    # in place: dst aliases add2 (the residual stream update of a decoder block)
    L.ggml_hip_mul_mat_q_acts(dw.h, acts, N, b2.ptr, M, 2, b1.ptr, b2.ptr)
    assert np.array_equal(b2.to_host(np.float32, (N, M)), (plain + a1) + a2)
    L.ggml_hip_acts_free(acts)
    dw.free()


@pytest.mark.parametrize("t,K,M", [(ob.Q4_0, 4544, 4672), (ob.Q4_0, 18176, 4544), (ob.Q4_K, 8192, 9216), (ob.Q6_K, 32768, 1024), (ob.Q2_K, 8192, 4100)])
def test_falcon_shapes_properties(oracle, t, K, M):
    """BASELINE-size shapes: sampled rows against the oracle + batch-invariance (N=1 vs N=4 columns bit-identical)"""
    rng = np.random.default_rng(K + M)
    w = synth.quantized_matrix(oracle, t, M, K, rng) if t in ob.KQUANTS else \
        oracle.quantize(t, (rng.standard_normal((M, K)) * 0.02).astype(np.float32)).reshape(M, -1)
    x = rng.standard_normal((4, K)).astype(np.float32)
    dw = g.Weight(t, w, K, M)
    y4 = dw.mul_mat(x)
    y1 = np.concatenate([dw.mul_mat(x[i:i + 1]) for i in range(4)])
    dw.free()
    assert np.array_equal(y4, y1)
    rows = rng.choice(M, size=24, replace=False)
    exp = oracle.mul_mat(t, np.ascontiguousarray(w[rows]), K, len(rows), x, 4)
    assert relrms(y4[:, rows], exp) <= TOL
