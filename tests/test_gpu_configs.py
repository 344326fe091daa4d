"""GPU: the BASELINE.json configurations at their OWN sizes (config 3: 2048-token prefill on k-quant weights; config 4:
Falcon-40B Q5_1 layer-sharded; config 5: Falcon-40B Q2_K at 8k context, perplexity) and Falcon-180B's widths.

A whole 2048-token prompt through a Falcon-40B-sized block is 1.4e12 multiply-adds -- minutes for the CPU oracle. The
checks here are therefore SAMPLED, and still bit-exact: the GPU keeps every block's input rows for all tokens
(falcon_hip_context_keep_hidden); for each block the oracle recomputes, from those inputs, the K / V rows of ALL tokens and
then the complete block (LayerNorms, Q rows, RoPE, attention over every earlier position, Wo, MLP, residual) for a handful of
sampled tokens (oracle/oracle_falcon.c: orc_falcon_block_sampled -- verified on the CPU to reproduce orc_falcon_eval's rows,
tests/test_oracle_sampled_cpu.py). A sampled output row must equal the GPU's next-block input row bit for bit, the
sampled logits rows must equal the GPU's. Every weight byte and every position's K / V enters some sampled row's result.
"""
import os

import numpy as np
import pytest

import ggllm_cpp_amd as g
from oracle import binding as ob
import synth

pytestmark = pytest.mark.gpu
NT = max(4, min(32, (os.cpu_count() or 8)))


@pytest.fixture(scope="module", autouse=True)
def _init():
    g.init(0)


HP_7B_K = dict(n_vocab=512, n_embd=4608, n_head=72, n_head_kv=2, n_layer=2, n_ff=18432, two_norms=True)     # 7B-like, 256-divisible
HP_40B_2 = dict(n_vocab=512, n_embd=8192, n_head=128, n_head_kv=8, n_layer=2, n_ff=32768, two_norms=True)


def gemm_split(M, N, n_cu=256):
    """the oracle mode that restates the prefill GEMM's K-split for an [M x N] result (kernels_gemm.hip: four partial sums
    below 4 x #CU 32 x 32 tiles, two above)"""
    tiles = ((M + 31) // 32) * ((N + 31) // 32)
    return 3 if tiles < 4 * n_cu else 4


def check_blocks_sampled(oracle, w, hid, logits, samples, n_batch, reference_order=False):
    hp = w["hparams"]
    L = hp["n_layer"]
    mo = oracle.model(w, 8, rope_n_ctx=w.get("rope_n_ctx"))
    modes = {gemm_split(M, n_batch) for M in ((hp["n_head"] + 2 * hp["n_head_kv"]) * 64, hp["n_embd"], hp["n_ff"])}
    assert len(modes) == 1, "one K-split for every mat-mul of a block at this batch size"
    oracle.lib.orc_set_sum_order(0 if reference_order else modes.pop())
    try:
        for il in range(L):
            out = mo.block_sampled(oracle.lib, il, hid[il], samples, n_threads=NT)
            assert np.array_equal(out, hid[il + 1][samples]), "block %d: sampled output rows differ from the oracle's" % il
        oracle.lib.orc_set_sum_order(0 if reference_order else gemm_split(hp["n_vocab"], n_batch))
        lo = mo.head_rows(oracle.lib, hid[L][samples], n_threads=NT)
        assert np.array_equal(lo, logits[samples]), "sampled logits rows differ from the oracle's"
    finally:
        oracle.lib.orc_set_sum_order(0)
    return lo


@pytest.mark.parametrize("shape,hp,t", [("7b-like", HP_7B_K, ob.Q4_K), ("7b-like", HP_7B_K, ob.Q6_K), ("40b", HP_40B_2, ob.Q4_K)])
def test_config3_kquant_prefill_2048(oracle, shape, hp, t):
    """BASELINE config 3 (k_quants path, 2048-token prefill) on k-quant super-blocks of 256: Falcon-7B's own n_embd 4544 is
    not a multiple of 256 and the reference refuses to quantize it (libfalcon.cpp:3626-3636), so the widths are the
    256-divisible 4608 ("7b-like") and Falcon-40B's. One 2048-token eval through the MFMA GEMM + prefill attention."""
    N = 2048
    w = synth.make_model_fast(hp, t, seed=33)
    toks = synth.tokens(N, hp["n_vocab"], seed=3)
    m = g.FalconModel(w, n_ctx=N, n_batch=N)
    lg, hid = m.eval(toks, 0, want_hidden=True)
    samples = [0, 1, 31, 32, 777, 2047]
    check_blocks_sampled(oracle, w, hid, lg, samples, N)
    # size-independent property: the last prompt token evaluated ALONE against the same KV cache (mat-vec kernels, decode
    # attention) is the same computation in another association -- close, and the same greedy choice
    one = m.eval(toks[-1:], N - 1)
    rel = float(np.abs(one[0] - lg[-1]).max() / np.sqrt((lg[-1].astype(np.float64) ** 2).mean()))
    print("%s %s: 2048-token prefill sampled rows bit-exact; last token as a decode step: %.2e" % (shape, ob.TYPE_NAME[t], rel))
    assert rel < 5e-2
    if shape == "7b-like" and t == ob.Q4_K:
        # the same prompt in the reference's own order (per-thread scalar mat-mul, f64 attention) against the oracle's
        # order 0, which is the reference's scalar build bit for bit
        g.load().ggml_hip_reference_order(1)
        try:
            lr, hr = m.eval(toks[:256], 0, want_hidden=True)
        finally:
            g.load().ggml_hip_reference_order(0)
        check_blocks_sampled(oracle, w, hr, lr, [0, 100, 255], 256, reference_order=True)
    m.free()


def test_config2_falcon7b_q4_0_full_depth(oracle):
    """BASELINE config 2 at its OWN size and depth: Falcon-7B Q4_0, all 32 blocks, vocabulary 65024, a 128-token prompt as one batch,
    then 128 greedy decode steps (fused decode kernels). Every block of both phases is checked bit for bit on sampled tokens against
    the oracle in the backend's order (prompt: the GEMM's K split; decode: the mat-vec kernels' wave order), the decode phase against
    the K / V rows the prompt phase left, and the sampled logits rows of both phases against the oracle's head."""
    hp = dict(synth.HP_7B)
    w = synth.make_model_fast(hp, ob.Q4_0, seed=1234)
    NP, ND, L = 128, 128, hp["n_layer"]
    toks = synth.tokens(NP, hp["n_vocab"], seed=42)
    m = g.FalconModel(w, n_ctx=NP + ND + 8, n_batch=NP)
    lg_p, hid_p = m.eval(toks, 0, logits_all=True, want_hidden=True)         # [NP, V], [L + 1, NP, E]
    cur, hid_d, lg_d, seq = int(lg_p[-1].argmax()), [], [], []
    for i in range(ND):
        lg, h = m.eval(np.array([cur], np.int32), NP + i, want_hidden=True)
        hid_d.append(h[:, 0, :]); lg_d.append(lg[0]); seq.append(cur)
        cur = int(lg[0].argmax())
    assert m.sync_error() == 0
    # the same 128 steps through the device-side greedy loop (hipGraph replay): the same tokens
    m.eval(toks, 0)
    dev = m.decode_greedy(seq[0], NP, ND, use_graph=True)
    assert np.array_equal(dev, np.array(seq[1:] + [cur], np.int32))
    m.free()
    hid_d = np.stack(hid_d, axis=1)                                          # [L + 1, ND, E]
    lg_d = np.stack(lg_d)
    mo = oracle.model(w, 8)
    sp, sd = [0, 63, 127], [0, 1, 64, 127]
    # at 128 columns the mat-muls of a block do not share one K split (Wup's 568 x 4 tiles get two partial sums, the others
    # four): mode 2 decides per mat-mul like the backend, told that the sampled rows stand for a batch of NP
    try:
        for il in range(L):
            oracle.lib.orc_set_sum_order(2); oracle.lib.orc_set_backend_batch(NP)
            out, ko, vo = mo.block_sampled(oracle.lib, il, hid_p[il], sp, n_threads=NT, want_kv=True)
            assert np.array_equal(out, hid_p[il + 1][sp]), "prompt, block %d" % il
            oracle.lib.orc_set_backend_batch(1)             # decode steps: one column each (wave order, the decode attention's chains)
            out = mo.block_sampled(oracle.lib, il, hid_d[il], sd, pos0=NP, k_prev=ko, v_prev=vo, n_threads=NT)
            assert np.array_equal(out, hid_d[il + 1][sd]), "decode, block %d" % il
        oracle.lib.orc_set_sum_order(gemm_split(hp["n_vocab"], NP))
        assert np.array_equal(mo.head_rows(oracle.lib, hid_p[L][sp], n_threads=NT), lg_p[sp])
        oracle.lib.orc_set_sum_order(1)
        assert np.array_equal(mo.head_rows(oracle.lib, hid_d[L][sd], n_threads=NT), lg_d[sd])
    finally:
        oracle.lib.orc_set_sum_order(0); oracle.lib.orc_set_backend_batch(0)


def test_config4_falcon40b_q5_1_two_stages(oracle):
    """BASELINE config 4 (Falcon-40B Q5_1 layer-sharded) at Falcon-40B's width: 4 blocks cut into two pipeline stages that
    run in ONE process through the stage API (falcon_hip_stage_step: device-resident hand-off of the residual row, as the
    RCCL pipeline moves it between GPUs). The chained stages reproduce the whole-model greedy decode, and the whole model
    reproduces the oracle's greedy continuation (the backend's association)."""
    hp = dict(HP_40B_2); hp["n_layer"] = 4
    w = synth.make_model_fast(hp, ob.Q5_1, seed=44)
    n_prompt, n_gen = 6, 8
    toks = synth.tokens(n_prompt, hp["n_vocab"], seed=7)
    L = g.load()
    whole = g.FalconModel(w, n_ctx=64, n_batch=8)
    lg = whole.eval(toks, 0)
    want = whole.decode_greedy(int(toks[-1]), n_prompt, n_gen)
    whole.free()
    s0 = g.FalconModel(w, n_ctx=64, n_batch=8, layer_begin=0, layer_end=2)
    s1 = g.FalconModel(w, n_ctx=64, n_batch=8, layer_begin=2, layer_end=4)
    E = hp["n_embd"]
    tok, hid, nxt = g.DevBuf(4), g.DevBuf(E * 4), g.DevBuf(4)
    got = []
    seq = list(toks) + [int(toks[-1])]
    for pos in range(n_prompt + n_gen):
        cur = np.array([seq[pos] if pos < len(seq) else got[-1]], np.int32)
        L.ggml_hip_memcpy_h2d(tok.ptr, cur.ctypes.data, 4)
        L.falcon_hip_stage_step(s0.ctx, tok.ptr, None, pos, hid.ptr, None)
        L.falcon_hip_stage_step(s1.ctx, None, hid.ptr, pos, None, nxt.ptr)
        if pos >= n_prompt:
            got.append(int(nxt.to_host(np.int32, (1,))[0]))
    assert s0.sync_error() == 0 and s1.sync_error() == 0
    s0.free(); s1.free()
    for b in (tok, hid, nxt):
        b.free()
    assert got == [int(x) for x in want]
    oracle.lib.orc_set_sum_order(2)
    try:
        mo = oracle.model(w, 64)
        lo = mo.eval(toks, 0, NT)
        assert np.array_equal(lg, lo)
        cur, ref = int(toks[-1]), []
        for i in range(n_gen):
            cur = int(mo.eval(np.array([cur], np.int32), n_prompt + i, NT)[0].argmax())
            ref.append(cur)
    finally:
        oracle.lib.orc_set_sum_order(0)
    assert got == ref


def test_config5_falcon40b_q2_k_8k_context_perplexity(oracle):
    """BASELINE config 5 (Falcon-40B Q2_K at 8k context, perplexity vs the CPU reference) at Falcon-40B's width, 2 blocks:
    n_ctx 8192 -> dynamic-NTK RoPE with alpha = 7.45 (ggml.c:12875-12898), the perplexity loop's 16 batches of 512 against a
    growing KV cache. falcon_hip_perplexity's NLL is recomputed from the logits of the same 16 evals; those evals are pinned
    against the oracle by sampled rows (K / V of all 8192 positions, attention over up to 8192 keys), and so are the NLL terms
    of the sampled positions."""
    n_ctx, n_batch = 8192, 512
    hp = dict(HP_40B_2)
    w = synth.make_model_fast(hp, ob.Q2_K, seed=55)
    w["rope_n_ctx"] = n_ctx
    toks = synth.tokens(n_ctx, hp["n_vocab"], seed=5)
    m = g.FalconModel(w, n_ctx=n_ctx, n_batch=n_batch)
    nll, count = m.perplexity(toks, n_ctx, n_batch)
    assert count == n_ctx - 1 - 512
    L, E, V = hp["n_layer"], hp["n_embd"], hp["n_vocab"]
    hid = np.empty((L + 1, n_ctx, E), np.float32)
    lg = np.empty((n_ctx, V), np.float32)
    for j in range(n_ctx // n_batch):
        a, b = j * n_batch, (j + 1) * n_batch
        lg[a:b], h = m.eval(toks[a:b], a, want_hidden=True)
        hid[:, a:b] = h
    m.free()

    def nll_terms(logits, pos):          # falcon_perplexity.cpp:12-27, 104-117 in numpy (float expf, double sum)
        out = []
        for j in pos:
            l = logits[j]
            e = np.exp((l - l.max()).astype(np.float32)).astype(np.float32)
            out.append(-np.log(np.float32(e[toks[j + 1]] / e.astype(np.float64).sum())))
        return np.array(out, np.float64)

    scored = range(512, n_ctx - 1)
    nll_host = float(nll_terms(lg, scored).sum())
    assert abs(nll_host - nll) <= 1e-6 * abs(nll), "falcon_hip_perplexity scored other logits than the 16 evals produced"
    samples = [0, 511, 512, 4097, 8190, 8191]
    lo = check_blocks_sampled(oracle, w, hid, lg, samples, n_batch)
    pos = [s for s in samples if 512 <= s < n_ctx - 1]
    idx = [samples.index(s) for s in pos]
    lo_full = lg.copy(); lo_full[pos] = lo[idx]
    assert np.array_equal(nll_terms(lo_full, pos), nll_terms(lg, pos))
    print("40B-width Q2_K, n_ctx 8192: perplexity %.3f over %d tokens; sampled rows bit-exact" % (float(np.exp(nll / count)), count))
