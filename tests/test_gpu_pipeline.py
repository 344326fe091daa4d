"""GPU: lock-step sequence contexts (falcon_hip_context_create_seqs) and the C++ layer pipeline (csrc/falcon_pipeline.hip)
with its local transport -- every rank of the job in this process on the one GPU, the schedule, stage contexts, hipGraphs
and hand-off buffers being the ones the RCCL transport uses."""
import numpy as np
import pytest

import ggllm_cpp_amd as g
from oracle import binding as ob
import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _init():
    g.init(0)


def _model(oracle, hp, t, n_layer, seed=17):
    hp = dict(hp)
    hp["n_layer"] = n_layer
    return hp, synth.make_model(oracle, hp, t, seed=seed)


@pytest.mark.parametrize("hp,t,B", [(synth.HP_TINY_MQA, ob.Q4_0, 3), (synth.HP_TINY_GQA, ob.Q5_1, 4), (synth.HP_TINY_GQA, ob.Q4_K, 2),
                                    (synth.HP_TINY_MQA, ob.Q8_0, 7), (synth.HP_TINY_GQA, ob.Q4_1, 8), (synth.HP_TINY_GQA, ob.Q4_1, 12), (synth.HP_TINY_MQA, ob.Q5_0, 14),
                                    (synth.HP_TINY_GQA, ob.Q4_K, 7), (synth.HP_TINY_GQA, ob.Q6_K, 12)])
def test_lock_step_sequences_equal_contexts_of_their_own(oracle, hp, t, B):
    """B sequences through ONE pass over the weights per step. Through the column mat-vec kernels (B <= 4; k-quants up to 12 in
    chunks of 4 columns) each sequence's logits are those of a context of its own, bit for bit (the same mat-vec per column); legacy
    formats with B >= 5 go through the streaming small-batch mat-mul (kernels_gemm_skinny.hip: the GEMM's split sums, pinned against
    the oracle's split orders in test_gpu_mul_mat.py): within the documented association spread"""
    hp, w = _model(oracle, hp, t, 2)
    m = g.FalconModel(w, n_ctx=32, n_batch=8)
    streams = [synth.tokens(9, hp["n_vocab"], seed=50 + b) for b in range(B)]
    singles = []
    for b in range(B):
        rows = [m.eval(streams[b][i:i + 1], i)[0] for i in range(9)]       # (a context is reusable from position 0)
        singles.append(np.stack(rows))
    sc = g.SeqContext(m, 32, B)
    for i in range(9):
        lg = sc.eval([int(streams[b][i]) for b in range(B)], i)
        for b in range(B):
            if B <= 4 or (t in ob.KQUANTS and B <= 12):
                assert np.array_equal(lg[b], singles[b][i]), (b, i)
            else:
                ref = singles[b][i]
                assert float(np.abs(lg[b] - ref).max() / np.sqrt((ref.astype(np.float64) ** 2).mean())) < 5e-2
    sc.free()
    m.free()


@pytest.mark.parametrize("t", [ob.Q4_K, ob.Q2_K])
def test_lock_step_at_falcon40b_width(oracle, t):
    """one 40B-shaped block (n_embd 8192, n_ff 32768, GQA 128/8, two norms; Q4_K, Q2_K): a context of 2 sequences runs the column mat-vec kernels
    (the output launch two columns at a time: four would not fit its LDS) -- one context's bits per sequence; contexts of 3 and more run the k-quant
    small-batch forms (a pass of 16 columns costs less than the column kernels' pass of 4 at this width): a row of those mat-muls does not depend
    on the other rows, so a 4-sequence context gives the bits of the first 4 sequences of an 8-sequence one, within the association spread of the
    single contexts"""
    hp = dict(n_vocab=512, n_embd=8192, n_head=128, n_head_kv=8, n_layer=1, n_ff=32768, two_norms=True)
    w = synth.make_model_fast(hp, t, seed=10)
    m = g.FalconModel(w, n_ctx=16, n_batch=8)
    B = 8
    streams = [synth.tokens(5, hp["n_vocab"], seed=70 + b) for b in range(B)]
    singles = [np.stack([m.eval(streams[b][i:i + 1], i)[0] for i in range(5)]) for b in range(B)]
    out = {}
    for nb in (2, 4, 8):
        sc = g.SeqContext(m, 16, nb)
        out[nb] = [sc.eval([int(streams[b][i]) for b in range(nb)], i) for i in range(5)]
        sc.free()
    m.free()
    for i in range(5):
        for b in range(2):
            assert np.array_equal(out[2][i][b], singles[b][i]), (b, i)
        for b in range(4):
            assert np.array_equal(out[4][i][b], out[8][i][b]), (b, i)
        for b in range(8):
            ref = singles[b][i].astype(np.float64)
            assert float(np.abs(out[8][i][b] - ref).max() / (np.sqrt((ref ** 2).mean()) + 1e-30)) <= 5e-2, (b, i)


@pytest.mark.parametrize("t,B", [(ob.Q4_K, 4), (ob.Q2_K, 3)])
def test_lock_step_small_batch_form_against_the_oracle(oracle, t, B):
    """contexts of 3 and 4 lock-step sequences at Falcon-40B width: every mat-mul of a step (Wqkv, Wup, Wo, Wdown, lm_head) runs the k-quants' small-batch form
    (fq_mul_mat_q_acts_from3, from 3 columns up) -- each sequence's logits equal the oracle evaluating that sequence alone with the backend's rule for such a
    context (orc_set_backend_batch(B) + orc_set_kq_min_cols(3): four partial sums per segment of 32 / 16 super-blocks), bit for bit"""
    hp = dict(synth.HP_40B); hp["n_layer"] = 1; hp["n_vocab"] = 512
    w = synth.make_model(oracle, hp, t, seed=13)
    streams = [synth.tokens(3, hp["n_vocab"], seed=80 + b) for b in range(B)]
    m = g.FalconModel(w, n_ctx=16, n_batch=8)
    sc = g.SeqContext(m, 16, B)
    got = [sc.eval([int(streams[b][i]) for b in range(B)], i) for i in range(3)]
    sc.free()
    m.free()
    oracle.lib.orc_set_sum_order(2); oracle.lib.orc_set_backend_batch(B); oracle.lib.orc_set_kq_min_cols(3)
    try:
        for b in range(B):
            mo = oracle.model(w, 16)
            for i in range(3):
                want = mo.eval(streams[b][i:i + 1], i, 16)[0]
                assert np.array_equal(got[i][b], want), (b, i)
    finally:
        oracle.lib.orc_set_sum_order(0); oracle.lib.orc_set_backend_batch(0); oracle.lib.orc_set_kq_min_cols(5)


def _greedy_reference(w, hp, first, rounds):
    m = g.FalconModel(w, n_ctx=64, n_batch=4)
    out = np.stack([m.decode_greedy(int(t), 0, rounds) for t in first], axis=1)      # [round][sequence]
    m.free()
    return out


@pytest.mark.parametrize("world,groups,batch", [(1, 2, 2), (2, 2, 2), (2, 4, 1), (3, 3, 2), (3, 6, 2), (4, 9, 1)])
@pytest.mark.parametrize("hp,t", [(synth.HP_TINY_MQA, ob.Q4_0), (synth.HP_TINY_GQA, ob.Q5_1)])
def test_cpp_pipeline_reproduces_single_process_greedy_decode(oracle, hp, t, world, groups, batch):
    """world stages x groups x batch sequences: the tokens the last stage samples, round by round, are the greedy decode of
    each sequence on the whole model in one process -- for the simple schedule (groups < 2 x world), the overlapped one, and
    for two consecutive run calls (warm-up + timed region)"""
    import bench_pipeline as bp
    hp, w = _model(oracle, hp, t, 5)
    rounds = 7
    first = synth.tokens(groups * batch, hp["n_vocab"], seed=8)
    want = _greedy_reference(w, hp, first, rounds)
    parts = bp.partition(hp["n_layer"], world)
    stages = [g.FalconModel(w, n_ctx=8, n_batch=1, layer_begin=lb, layer_end=le) for lb, le in parts]
    ranks = [g.Pipeline(stages[r], r, world, groups, batch, 16, local=True) for r in range(world)]
    ranks[0].set_tokens(first)
    g.Pipeline.run_local(ranks, 3, 0)
    g.Pipeline.run_local(ranks, rounds - 3, 3)
    got = ranks[-1].history(0, rounds)
    for r in range(world - 1):
        assert ranks[r].history(0, rounds) is None or world == 1
    for p in ranks:
        p.free()
    for s in stages:
        s.free()
    assert np.array_equal(got, want)


def test_lock_step_groups_beyond_64_sequences(oracle):
    """80 sequences per weight pass (the int8-MFMA tile GEMM's range): every sequence reads its own token id and KV cache -- 80
    streams started from 5 distinct tokens fall into 5 classes of identical streams (a row of the mat-mul does not depend on the
    other rows), each the stream of a 16-sequence pass started from the same token"""
    hp, w = _model(oracle, synth.HP_TINY_GQA, ob.Q4_0, 2)
    m = g.FalconModel(w, n_ctx=16, n_batch=128)
    first5 = synth.tokens(5, hp["n_vocab"], seed=21)
    B = 80
    first = np.array([first5[b % 5] for b in range(B)], np.int32)
    p = g.Pipeline(m, 0, 1, 1, B, 16)
    p.set_tokens(first)
    p.run(6, 0)
    got = p.history(0, 6)                                    # [round][sequence]
    p.free()
    q = g.Pipeline(m, 0, 1, 1, 20, 16)                       # 20 > 16 columns: the same mat-mul kernel, the same per-row arithmetic
    q.set_tokens(first[:20])
    q.run(6, 0)
    want = q.history(0, 6)
    q.free()
    m.free()
    for b in range(B):
        assert np.array_equal(got[:, b], want[:, b % 5]), b


def test_cpp_pipeline_world_1_run_without_rccl(oracle):
    """falcon_hip_pipeline_create with world 1 needs no communicator: falcon_hip_pipeline_run advances groups x batch
    sequences on one GPU (what bench.py --force-pipeline times)"""
    hp, w = _model(oracle, synth.HP_TINY_MQA, ob.Q4_0, 2)
    first = synth.tokens(6, hp["n_vocab"], seed=9)
    want = _greedy_reference(w, hp, first, 5)
    m = g.FalconModel(w, n_ctx=8, n_batch=1)
    p = g.Pipeline(m, 0, 1, 3, 2, 16)
    p.set_tokens(first)
    p.run(5, 0)
    got = p.history(0, 5)
    p.free(); m.free()
    assert np.array_equal(got, want)
