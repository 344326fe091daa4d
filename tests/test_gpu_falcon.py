"""GPU parity: the device-resident Falcon stack against the oracle and against logits captured from the reference."""
import os

import numpy as np
import pytest

import ggllm_cpp_amd as g
from oracle import binding as ob
import synth

pytestmark = pytest.mark.gpu

# north_star: logits within 1e-3 relative of the CPU reference. Metric: max |diff| / rms(reference logits).
LOGIT_TOL = 1e-3


@pytest.fixture(scope="module", autouse=True)
def _init():
    g.init(0)


def relrms(a, b):
    return float(np.abs(a.astype(np.float64) - b).max() / (np.sqrt((b.astype(np.float64) ** 2).mean()) + 1e-30))


CASES = [("mqa_q4_0", synth.HP_TINY_MQA, ob.Q4_0, "tiny_models"), ("gqa_q5_1", synth.HP_TINY_GQA, ob.Q5_1, "tiny_models"),
         ("gqa_q4_K", synth.HP_TINY_GQA, ob.Q4_K, "tiny_models"), ("mqa_q8_0", synth.HP_TINY_MQA, ob.Q8_0, "tiny_models"),
         ("gqa_q4_1", synth.HP_TINY_GQA, ob.Q4_1, "tiny_models_all"), ("gqa_q5_0", synth.HP_TINY_GQA, ob.Q5_0, "tiny_models_all"),
         ("gqa_q2_K", synth.HP_TINY_GQA, ob.Q2_K, "tiny_models_all"), ("gqa_q3_K", synth.HP_TINY_GQA, ob.Q3_K, "tiny_models_all"),
         ("gqa_q5_K", synth.HP_TINY_GQA, ob.Q5_K, "tiny_models_all"), ("gqa_q6_K", synth.HP_TINY_GQA, ob.Q6_K, "tiny_models_all")]

# Whole-model parity, all ten weight formats, against logits captured from the REAL reference (tests/golden/tiny_models.npz,
# tiny_models_all.npz: the reference's graph executor, scalar and AVX2 builds).
#   (1) REFERENCE ORDER (ggml_hip_reference_order(1)): hidden states, prefill logits and decode logits are BIT-IDENTICAL with
#       the reference's scalar build -- every mat-mul walks its rows in the reference's own block / lane order
#       (csrc/fq_ref_dot.h), the attention accumulates in f64 like the portable ggml_vec_dot_f32. This is the north-star's
#       "logits within 1e-3 of the CPU reference", met with 0.
#   (2) DEFAULT ORDER (what the benchmarks run): bit-identical with the oracle run with the backend's association of the
#       same terms (orc_set_sum_order(2): unit-per-lane partial sums + xor butterfly for the mat-vec kernels, interleaved
#       K-split partial sums for the MFMA GEMM, f32 fused multiply-add chains in the attention) -- this pins every fast kernel.
#   (3) The distance between (1) and (2) is a property of the MODEL, not of a kernel: the decoder stack is chaotic at the
#       1e-7 level -- re-associating an f32 block sum (which every vectorised build of the reference does as well) now and
#       then flips one 8-bit activation rounding, and one flip moves logits by 1e-3..1e-2. The reference's own AVX2 and
#       scalar builds differ by up to 2.8e-2 on these fixtures (both are in the golden files). For the legacy formats the
#       default order is asserted within max(1e-3, 2 x that spread) of the reference, for the k-quants (whose two reference
#       builds share one association) within 3.6e-2 = twice the largest distance measured on MI355X (Q2_K 1.8e-2; round 5: was 5.6e-2); it is printed for all.


@pytest.mark.parametrize("name,hp,t,gfile", CASES)
def test_tiny_falcon_vs_reference_fixture(oracle, golden, name, hp, t, gfile):
    gt = golden[gfile]
    w = synth.make_model(oracle, hp, t, seed=1234)
    toks = gt[f"{name}_tokens"]
    ref_h, ref_l, ref_d = gt[f"{name}_prefill_hidden_scalar"], gt[f"{name}_prefill_logits_scalar"], gt[f"{name}_decode_logits_scalar"]
    m = g.FalconModel(w, n_ctx=64, n_batch=8)
    # (1) reference order == the real reference, bit for bit
    g.load().ggml_hip_reference_order(1)
    try:
        lr, hr = m.eval(toks[:8], 0, logits_all=True, want_hidden=True)
        dr = np.concatenate([m.eval(toks[i:i + 1], i, logits_all=True) for i in range(8, 12)])
    finally:
        g.load().ggml_hip_reference_order(0)
    assert np.array_equal(hr, ref_h), "prefill hidden states differ from the reference's"
    assert np.array_equal(lr, ref_l), "prefill logits differ from the reference's"
    assert np.array_equal(dr, ref_d), "decode logits differ from the reference's"
    # (2) default order == the oracle with the backend's association
    lg, hid = m.eval(toks[:8], 0, logits_all=True, want_hidden=True)
    dec = np.concatenate([m.eval(toks[i:i + 1], i, logits_all=True) for i in range(8, 12)])
    m.free()
    oracle.lib.orc_set_sum_order(2)
    try:
        mo = oracle.model(w, 64)
        lo, ho = mo.eval(toks[:8], 0, 2, want_hidden=True)
        do = np.concatenate([mo.eval(toks[i:i + 1], i, 2) for i in range(8, 12)])
    finally:
        oracle.lib.orc_set_sum_order(0)
    assert np.array_equal(hid, ho) and np.array_equal(lg, lo) and np.array_equal(dec, do)
    # (3) the association spread, next to the reference's own
    spread = max(relrms(gt[f"{name}_prefill_logits_avx"], ref_l), relrms(gt[f"{name}_decode_logits_avx"], ref_d))
    e_l, e_d = relrms(lg, ref_l), relrms(dec, ref_d)
    print(name, "default order vs reference logits: prefill %.2e decode %.2e (reference AVX2-vs-scalar spread %.2e)" % (e_l, e_d, spread))
    if t in ob.LEGACY:
        assert max(e_l, e_d) <= max(LOGIT_TOL, 2 * spread)
    else:
        # k-quants: the reference's two builds keep the same eight float lanes, so THEIR spread is ~1e-6 and says nothing about the
        # model's sensitivity; the default order's distance is the backend's own re-association (one term per lane-unit + butterfly for
        # N <= 4, K-split partial sums in the GEMM), amplified by the same activation-rounding flips as on the legacy models, whose
        # builds differ by up to 2.8e-2 on these fixtures. Measured on MI355X: 1.1e-2 (Q4_K) .. 1.8e-2 (Q2_K), one flipped Q8_K rounding in 2 blocks of a
        # tiny model; the stated bound is twice the largest (round 5; it was 5.6e-2)
        assert max(e_l, e_d) <= 3.6e-2, (name, e_l, e_d)


@pytest.mark.parametrize("name,hp,t", [("mqa_q4_0", synth.HP_TINY_MQA, ob.Q4_0), ("gqa_q5_1", synth.HP_TINY_GQA, ob.Q5_1),
                                       ("gqa_q4_K", synth.HP_TINY_GQA, ob.Q4_K), ("gqa_q6_K", synth.HP_TINY_GQA, ob.Q6_K)])
def test_fused_decode_bit_identical_to_op_list(oracle, name, hp, t):
    """the fused decode kernels -- 3 launches per block, 2 (attention + output mat-vec in one launch with an in-launch
    hand-off) and 1 (the next block's LayerNorm mat-vec as a second phase of that launch) -- reproduce the op-by-op launch
    list bit for bit (logits, hidden, KV cache)"""
    w = synth.make_model(oracle, hp, t, seed=21)
    toks = synth.tokens(9, hp["n_vocab"], seed=6)
    outs = []
    for mode in (0, 1, 2, 3):
        m = g.FalconModel(w, n_ctx=32, n_batch=4)
        m.set_fused(mode)
        m.eval(toks[:4], 0)                                       # prefill is the same code on all
        r = [m.eval(toks[i:i + 1], i, want_hidden=(i % 2 == 0)) for i in range(4, 9)]      # (the hidden-state hook keeps mode 3 at two launches)
        r = [x if isinstance(x, tuple) else (x, None) for x in r]
        assert m.sync_error() == 0
        outs.append(r)
        m.free()
    for other in outs[1:]:
        for (la, ha), (lb, hb) in zip(outs[0], other):
            assert ha is None or np.array_equal(ha, hb)
            assert np.array_equal(la, lb)


@pytest.mark.parametrize("t", [ob.Q4_0, ob.Q5_1, ob.Q4_K])
def test_in_launch_handoff_full_width(oracle, t):
    """k_attn_out at Falcon-7B width (214 workgroups, 24 of them attention producers, every CU streaming weights while
    the consumers poll): 96 greedy steps through the hipGraph give the same tokens and the same final logits as the
    3-launch form, and no poll ever timed out"""
    hp = dict(synth.HP_7B); hp["n_layer"] = 3; hp["n_vocab"] = 4096
    if t in ob.KQUANTS:                                 # super-blocks of 256: 72 heads, two norms, 2 kv heads
        hp.update(n_embd=4608, n_head=72, n_head_kv=2, n_ff=18432, two_norms=True)
    w = synth.make_model_fast(hp, t, seed=5)
    toks = synth.tokens(16, hp["n_vocab"], seed=9)
    res = []
    for mode in (1, 2, 3):
        m = g.FalconModel(w, n_ctx=256, n_batch=16)
        m.set_fused(mode)
        m.eval(toks, 0)
        out = m.decode_greedy(int(toks[-1]), 16, 96, use_graph=True)
        lg = m.eval(out[-1:], 16 + 96)
        assert m.sync_error() == 0
        res.append((out, lg))
        m.free()
    for other in res[1:]:
        assert np.array_equal(res[0][0], other[0])
        assert np.array_equal(res[0][1], other[1])


def test_prefill_equals_incremental_and_graph(oracle):
    """size-independent properties: batch-invariance (N tokens at once == one at a time, bit-identical) and
    greedy decode through the captured hipGraph == plain launches == oracle greedy"""
    hp = synth.HP_TINY_MQA
    w = synth.make_model(oracle, hp, ob.Q4_0, seed=3)
    toks = synth.tokens(9, hp["n_vocab"], seed=8)
    m = g.FalconModel(w, n_ctx=64, n_batch=9)
    g.load().ggml_hip_debug_force_gemv(1)          # same kernel family for both -> bit-identical
    try:
        full = m.eval(toks, 0)
        inc = np.concatenate([m.eval(toks[i:i + 1], i) for i in range(9)])
    finally:
        g.load().ggml_hip_debug_force_gemv(0)
    assert np.array_equal(full, inc)
    # (the MFMA GEMM prefill associates the block sum differently; that path is pinned bit-exactly against the oracle in
    #  test_tiny_falcon_vs_reference_fixture -- comparing the two associations with each other only measures the chaos)
    first = int(full[-1].argmax())
    plain = m.decode_greedy(first, 9, 12, use_graph=False)
    m.eval(toks, 0)                                   # rewind the KV cache to the same state
    graph = m.decode_greedy(first, 9, 12, use_graph=True)
    assert np.array_equal(plain, graph)
    mo = oracle.model(w, 64)
    mo.eval(toks, 0, 2)
    cur, ref = first, []
    for i in range(12):
        cur = int(mo.eval(np.array([cur], np.int32), 9 + i, 2)[0].argmax())
        ref.append(cur)
    assert list(plain) == ref
    m.free()


def _both_orders(oracle, w, toks, n_pre, n_ctx):
    """GPU vs oracle for a prefill of n_pre tokens + one decode step: default order against the oracle's backend
    association, reference order against the oracle's order 0 (= the reference's scalar build, test_oracle_vs_golden.py).
    Returns the default-vs-reference-order distance of the prefill logits."""
    m = g.FalconModel(w, n_ctx=n_ctx, n_batch=n_pre)
    out = {}
    for order in (0, 2):
        g.load().ggml_hip_reference_order(1 if order == 0 else 0)
        oracle.lib.orc_set_sum_order(order)
        try:
            mo = oracle.model(w, n_ctx)
            lo, ho = mo.eval(toks[:n_pre], 0, 8, want_hidden=True)
            do = mo.eval(toks[n_pre:n_pre + 1], n_pre, 8)
            lg, hid = m.eval(toks[:n_pre], 0, want_hidden=True)
            d = m.eval(toks[n_pre:n_pre + 1], n_pre)
        finally:
            oracle.lib.orc_set_sum_order(0)
            g.load().ggml_hip_reference_order(0)
        assert np.array_equal(hid, ho), "hidden states, order %d" % order
        assert np.array_equal(lg, lo), "prefill logits, order %d" % order
        assert np.array_equal(d, do), "decode logits, order %d" % order
        out[order] = lg
    m.free()
    return relrms(out[2], out[0])


@pytest.mark.parametrize("hp,t", [(synth.HP_TINY_MQA, ob.Q4_0), (synth.HP_TINY_GQA, ob.Q5_1), (synth.HP_TINY_GQA, ob.Q4_K)])
@pytest.mark.parametrize("n_past,N", [(0, 32), (0, 33), (0, 45), (0, 100), (5, 64), (40, 33), (37, 95), (64, 32)])
def test_matrix_pipe_attention_ragged_prompts(oracle, hp, t, n_past, N):
    """prompts of 32 tokens and more take the prefill attention on the f32 matrix pipe (k_attention_mfma: 32-token query tiles,
    32-key tiles): ragged last tiles, a context that does not start at 0 (the first chunk through whichever kernel its length
    selects), MQA and GQA -- hidden states and logits of the chunk bit-identical to the oracle's restatement of the sequential
    multiply-add chains (oracle_falcon.c dot_qk_mfma / dot_pv_mfma)"""
    w = synth.make_model(oracle, hp, t, seed=31)
    toks = synth.tokens(n_past + N, hp["n_vocab"], seed=12)
    m = g.FalconModel(w, n_ctx=160, n_batch=100)
    if n_past:
        m.eval(toks[:n_past], 0)
    lg, hid = m.eval(toks[n_past:], n_past, want_hidden=True)
    m.free()
    oracle.lib.orc_set_sum_order(2)
    try:
        mo = oracle.model(w, 160)
        if n_past:
            mo.eval(toks[:n_past], 0, 8)
        lo, ho = mo.eval(toks[n_past:], n_past, 8, want_hidden=True)
    finally:
        oracle.lib.orc_set_sum_order(0)
    assert np.array_equal(hid, ho)
    assert np.array_equal(lg, lo)


def test_eval_rejects_bad_batches(oracle):
    """an empty batch, a batch larger than n_batch, a position past n_ctx and a token id outside the vocabulary are refused with
    falcon_eval's non-zero return (libfalcon.cpp:4588-4591) -- nothing is evaluated, the context stays usable"""
    hp = synth.HP_TINY_MQA
    w = synth.make_model(oracle, hp, ob.Q4_0, seed=5)
    m = g.FalconModel(w, n_ctx=16, n_batch=4)
    toks = synth.tokens(8, hp["n_vocab"], seed=1)
    good = m.eval(toks[:4], 0)
    L = g.load()
    for bad_toks, n_past in ((toks[:0], 0), (toks[:5], 0), (toks[:4], 13), (toks[:1], 16), (np.array([hp["n_vocab"]], np.int32), 4), (np.array([-1], np.int32), 4)):
        t = np.ascontiguousarray(bad_toks, np.int32)
        rc = L.falcon_hip_eval(m.ctx, t.ctypes.data if t.size else None, t.size, n_past, 1)
        assert rc in (1, 2), (t.size, n_past, rc)
        with pytest.raises(RuntimeError):
            m.eval(bad_toks, n_past)
    assert np.array_equal(m.eval(toks[:4], 0), good)
    m.free()


def test_falcon7b_shaped_layer_vs_oracle(oracle):
    """one block with the real 7B dimensions (n_embd 4544, 71 heads MQA, n_ff 18176), small vocab: bit-exact in both orders"""
    hp = dict(n_vocab=1024, n_embd=4544, n_head=71, n_head_kv=1, n_layer=1, n_ff=18176, two_norms=False)
    w = synth.make_model(oracle, hp, ob.Q4_0, seed=9)
    print("7B-shaped block: association spread %.2e" % _both_orders(oracle, w, synth.tokens(4, 1024, seed=2), 3, 16))


@pytest.mark.parametrize("n_pre", [9, 16, 23])
def test_falcon7b_shaped_layer_small_batch_vs_oracle(oracle, n_pre):
    """the same block with prompts of 9, 16 and 23 tokens: Wqkv and Wup in ONE launch of the small-batch mat-mul with the columns resident
    in LDS, Wdown as one K share per workgroup (kernels_gemm_skinny.hip; 23 = two passes) -- bit-exact in both orders"""
    hp = dict(n_vocab=1024, n_embd=4544, n_head=71, n_head_kv=1, n_layer=1, n_ff=18176, two_norms=False)
    w = synth.make_model(oracle, hp, ob.Q4_0, seed=9)
    _both_orders(oracle, w, synth.tokens(n_pre + 1, 1024, seed=3), n_pre, 32)


def test_falcon40b_shaped_layer_vs_oracle(oracle):
    """one block with the 40B geometry (n_embd 8192, 128 heads, 8 kv heads, two norms), Q4_K weights: bit-exact in both orders"""
    hp = dict(n_vocab=512, n_embd=8192, n_head=128, n_head_kv=8, n_layer=1, n_ff=32768, two_norms=True)
    w = synth.make_model(oracle, hp, ob.Q4_K, seed=10)
    print("40B-shaped block: association spread %.2e" % _both_orders(oracle, w, synth.tokens(3, 512, seed=4), 2, 16))


@pytest.mark.parametrize("t,n_pre", [(ob.Q4_K, 6), (ob.Q4_K, 16), (ob.Q4_K, 21), (ob.Q5_K, 9), (ob.Q2_K, 11), (ob.Q3_K, 7), (ob.Q6_K, 13)])
def test_falcon40b_shaped_layer_small_batch_vs_oracle(oracle, t, n_pre):
    """the 40B-shaped Q4_K block with prompts of 6, 16 and 21 tokens: every mat-mul through the share-pair small-batch form (k_gemm_skinny_q4k;
    21 = two passes), Wup's sum launch applying GELU and writing Wdown's Q8_K image, Wo and Wdown (four K segments) sharing one sum launch with
    the residual -- bit-exact in both orders (the oracle's mode 2 restates the segmented four-sum order)"""
    hp = dict(n_vocab=512, n_embd=8192, n_head=128, n_head_kv=8, n_layer=1, n_ff=32768, two_norms=True)
    w = synth.make_model_fast(hp, t, seed=11)
    _both_orders(oracle, w, synth.tokens(n_pre + 1, 512, seed=5), n_pre, 32)


@pytest.mark.parametrize("t", [ob.Q4_0, ob.Q4_K])
def test_falcon180b_shaped_layer_vs_oracle(oracle, t):
    """one block with the 180B geometry (libfalcon.cpp:1578-1582: n_embd 14848, 232 heads, 8 kv heads, n_ff 59392 -- the
    K = 59392 down projection; at this width k_gemv_ln's LayerNorm leaves its register path): bit-exact in both orders"""
    hp = dict(n_vocab=512, n_embd=14848, n_head=232, n_head_kv=8, n_layer=1, n_ff=59392, two_norms=True)
    w = synth.make_model_fast(hp, t, seed=12)
    print("180B-shaped block %s: association spread %.2e" % (ob.TYPE_NAME[t], _both_orders(oracle, w, synth.tokens(7, 512, seed=4), 6, 16)))


@pytest.mark.parametrize("name,hp,t", [("mqa_q4_0", synth.HP_TINY_MQA, ob.Q4_0), ("gqa_q5_1", synth.HP_TINY_GQA, ob.Q5_1),
                                       ("gqa_q4_K", synth.HP_TINY_GQA, ob.Q4_K), ("gqa_q6_K", synth.HP_TINY_GQA, ob.Q6_K), ("mqa_q8_0", synth.HP_TINY_MQA, ob.Q8_0)])
def test_ggcc_file_loader(oracle, golden, tmp_path, name, hp, t):
    """falcon_hip_model_load_ggcc on a GGCC v10 file (the reference's model format; the file is byte-identical to the one
    the real libfalcon.cpp loaded when tests/golden/ggcc_models.npz was captured): same logits as the in-memory upload of
    the same weights, bit for bit; in reference order (ggml_hip_reference_order) prefill AND decode logits are the logits of the
    REFERENCE's own libfalcon.cpp (its loader, graph builder and falcon_eval on the same file), bit for bit, k-quants included;
    a pipeline stage loads only its own blocks"""
    import ggcc_writer
    gg = golden["ggcc_models"]
    w = synth.make_model(oracle, hp, t, seed=4321)
    path = str(tmp_path / (name + ".ggcc"))
    ggcc_writer.write_ggcc(path, w)
    toks = gg[f"{name}_tokens"]
    a = g.FalconModel.from_ggcc(path, n_ctx=64, n_batch=16)
    b = g.FalconModel(w, n_ctx=64, n_batch=16)
    assert a.hp["n_embd"] == hp["n_embd"] and a.hp["two_norms"] == bool(hp.get("two_norms"))
    la, lb = a.eval(toks[:9], 0), b.eval(toks[:9], 0)
    assert np.array_equal(la, lb)
    da = np.concatenate([a.eval(toks[i:i + 1], i) for i in range(9, 12)])
    db = np.concatenate([b.eval(toks[i:i + 1], i) for i in range(9, 12)])
    assert np.array_equal(da, db)
    ref_pre, ref_dec = gg[f"{name}_prefill_logits"], gg[f"{name}_decode_logits"]
    g.load().ggml_hip_reference_order(1)                             # the reference's block / lane order, attention dots in f64
    try:
        ls = a.eval(toks[:9], 0)
        ds = np.concatenate([a.eval(toks[i:i + 1], i) for i in range(9, 12)])
    finally:
        g.load().ggml_hip_reference_order(0)
    assert np.array_equal(ls, ref_pre)
    assert np.array_equal(ds, ref_dec)
    print(name, "default order vs libfalcon.cpp logits: prefill %.2e decode %.2e" % (relrms(la, ref_pre), relrms(da, ref_dec)))
    if t in ob.LEGACY:                                               # default order: DESIGN.md section 2, the reference's own spread
        assert relrms(la, ref_pre) <= max(1e-3, 2 * 2.8e-2) and relrms(da, ref_dec) <= max(1e-3, 2 * 2.8e-2)
    assert a.weight_bytes() == b.weight_bytes()
    a.free(); b.free()
    # second pipeline stage: block 1 only (+ ln_f, lm_head)
    s1 = g.FalconModel.from_ggcc(path, n_ctx=64, n_batch=16, layer_begin=1, layer_end=hp["n_layer"])
    assert s1.n_local == hp["n_layer"] - 1
    s1.free()


@pytest.mark.parametrize("name,hp,t", [("mqa_q4_0", synth.HP_TINY_MQA, ob.Q4_0), ("gqa_q5_1", synth.HP_TINY_GQA, ob.Q5_1)])
def test_perplexity_loop(oracle, golden, name, hp, t):
    """falcon_hip_perplexity = the reference's perplexity loop (falcon_perplexity.cpp:28-124): chunks of n_ctx 32 in batches
    of 8, NLL of the second half of every chunk. The fixture was produced by driving the same loop over the REAL reference's
    falcon_eval (oracle/gen_golden.py); in reference order (ggml_hip_reference_order) the prefill logits are bit-identical to the
    reference's, so the NLL is too (same libm); the default order (partial sums) moves it within the reference's own
    build-to-build spread"""
    gg = golden["ggcc_models"]
    w = synth.make_model(oracle, hp, t, seed=4321)
    m = g.FalconModel(w, n_ctx=64, n_batch=16)
    nll, count = m.perplexity(gg[f"{name}_ppl_tokens"], n_ctx=32, n_batch=8)
    g.load().ggml_hip_reference_order(1)
    try:
        nll_seq, count_seq = m.perplexity(gg[f"{name}_ppl_tokens"], n_ctx=32, n_batch=8)
    finally:
        g.load().ggml_hip_reference_order(0)
    m.free()
    assert count == count_seq == int(gg[f"{name}_ppl_count"]) == 45
    assert abs(nll_seq - float(gg[f"{name}_ppl_nll"])) <= 1e-9 * abs(nll_seq)
    assert abs(nll - float(gg[f"{name}_ppl_nll"])) <= 2e-3 * abs(nll)


@pytest.mark.parametrize("graph", [1, 0])
@pytest.mark.parametrize("hp,t,cut", [(synth.HP_TINY_GQA, ob.Q5_1, 1), (synth.HP_TINY_MQA, ob.Q4_0, 1)])
def test_pipeline_stage_steps_match_whole_model(oracle, monkeypatch, hp, t, cut, graph):
    """the stage API of the layer pipeline (falcon_hip_stage_step: device-resident inputs / outputs, no host sync; captured
    into one hipGraph per stage by default) chained over two stages in ONE process reproduces the whole-model greedy decode"""
    monkeypatch.setenv("FALCON_HIP_STAGE_GRAPH", str(graph))
    L = g.load()
    w = synth.make_model(oracle, hp, t, seed=11)
    toks = synth.tokens(6, hp["n_vocab"], seed=3)
    whole = g.FalconModel(w, n_ctx=64, n_batch=8)
    whole.eval(toks, 0)
    want = whole.decode_greedy(int(toks[-1]), 6, 10)
    whole.free()
    s0 = g.FalconModel(w, n_ctx=64, n_batch=8, layer_begin=0, layer_end=cut)
    s1 = g.FalconModel(w, n_ctx=64, n_batch=8, layer_begin=cut, layer_end=hp["n_layer"])
    E = hp["n_embd"]
    tok, hid, nxt = g.DevBuf(4), g.DevBuf(E * 4), g.DevBuf(4)
    got = []
    seq = list(toks) + [int(toks[-1])]                     # prompt token by token, then feed back the sampled one
    for pos in range(6 + 10):
        cur = np.array([seq[pos] if pos < len(seq) else got[-1]], np.int32)
        L.ggml_hip_memcpy_h2d(tok.ptr, cur.ctypes.data, 4)
        L.falcon_hip_stage_step(s0.ctx, tok.ptr, None, pos, hid.ptr, None)
        L.falcon_hip_stage_step(s1.ctx, None, hid.ptr, pos, None, nxt.ptr)
        if pos >= 6:
            got.append(int(nxt.to_host(np.int32, (1,))[0]))
    s0.free(); s1.free()
    for b in (tok, hid, nxt):
        b.free()
    assert got == [int(x) for x in want]


def test_text_in_text_out_example(oracle, tmp_path):
    """examples/falcon_generate.py end to end on a GGCC file that carries a real byte-level BPE vocabulary: tokenizer ->
    prefill -> greedy decode on the device -> detokenizer; the generated ids are the oracle's greedy continuation"""
    import importlib.util
    import bpe_fixture
    import ggcc_writer
    vocab, merges = bpe_fixture.build(n_merges=308)                  # 12 + 256 + 308 = 576 tokens
    hp = dict(synth.HP_TINY_MQA)
    hp["n_vocab"] = len(vocab)
    w = synth.make_model(oracle, hp, ob.Q4_0, seed=321)
    path = str(tmp_path / "tiny_bpe.ggcc")
    ggcc_writer.write_ggcc(path, w, vocab, merges)
    spec = importlib.util.spec_from_file_location("falcon_generate", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "falcon_generate.py"))
    ex = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ex)
    prompt = "The quick brown fox didn't jump"
    ids, out, text = ex.generate(path, prompt, 6, n_ctx=64)
    assert 4 < ids.size < len(prompt)                                # merges applied
    oracle.lib.orc_set_sum_order(2)
    try:
        mo = oracle.model(w, 64)
        lg = mo.eval(ids, 0, 2)
        exp = [int(lg[-1].argmax())]
        for i in range(5):
            exp.append(int(mo.eval(np.array(exp[-1:], np.int32), ids.size + i, 2)[-1].argmax()))
    finally:
        oracle.lib.orc_set_sum_order(0)
    eos = 11
    if eos in exp:
        exp = exp[:exp.index(eos)]
    assert out.tolist() == exp
    v = g.Vocab(path)
    assert v.detokenize(out) == text and v.detokenize(ids) == prompt.encode()
    v.free()


def test_text_perplexity_example(oracle, tmp_path):
    """examples/falcon_perplexity.py: tokenizer (bos first, as the reference's tool) + the perplexity loop on a GGCC file"""
    import importlib.util
    import math
    import bpe_fixture
    import ggcc_writer
    vocab, merges = bpe_fixture.build(n_merges=308)
    hp = dict(synth.HP_TINY_MQA)
    hp["n_vocab"] = len(vocab)
    w = synth.make_model(oracle, hp, ob.Q4_0, seed=322)
    path = str(tmp_path / "tiny_bpe.ggcc")
    ggcc_writer.write_ggcc(path, w, vocab, merges)
    spec = importlib.util.spec_from_file_location("falcon_perplexity", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "falcon_perplexity.py"))
    ex = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ex)
    text = (bpe_fixture.CORPUS * 2).encode("utf-8")
    ppl, n, total = ex.perplexity(path, text, n_ctx=32, n_batch=8)
    v = g.Vocab(path)
    ids = v.tokenize(text, add_bos=True)
    v.free()
    assert ids[0] == 11 and total == ids.size and n == (ids.size // 32) * (32 - 1 - 16)
    m = g.FalconModel.from_ggcc(path, n_ctx=32, n_batch=8)
    nll, n2 = m.perplexity(ids, 32, 8)
    m.free()
    assert n2 == n and ppl == math.exp(nll / n) and 1.0 < ppl < 10.0 * len(vocab)


def test_batched_eval_graph_replay_equals_plain_launches(oracle, monkeypatch):
    """batches (N > 4) are replayed from a hipGraph cached per (size, keys): the same logits as plain launches, on first use (capture),
    on replays, at other positions, and after a global switch changed the launch list"""
    hp = synth.HP_TINY_GQA
    w = synth.make_model(oracle, hp, ob.Q5_1, seed=31)
    toks = synth.tokens(40, hp["n_vocab"], seed=4)
    monkeypatch.setenv("FALCON_HIP_PREFILL_GRAPH", "0")
    plain = g.FalconModel(w, n_ctx=64, n_batch=16)
    monkeypatch.setenv("FALCON_HIP_PREFILL_GRAPH", "1")
    graph = g.FalconModel(w, n_ctx=64, n_batch=16)
    for rep in range(3):
        for n_past, n in ((0, 12), (12, 9), (21, 12), (0, 6)):
            a = plain.eval(toks[n_past:n_past + n], n_past, logits_all=True)
            b = graph.eval(toks[n_past:n_past + n], n_past, logits_all=True)
            assert np.array_equal(a, b), (rep, n_past, n)
        g.load().ggml_hip_debug_force_gemv(rep == 0)           # second round: the mat-muls through the mat-vec kernel
    g.load().ggml_hip_debug_force_gemv(0)
    plain.free(); graph.free()
