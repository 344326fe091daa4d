"""CPU: orc_falcon_block_sampled / orc_falcon_head_rows (the sampled-token form of the oracle that the full-size GPU tests
use, tests/test_gpu_configs.py) reproduce orc_falcon_eval's rows bit for bit -- first batch and a later batch that attends
to earlier positions, reference order and the backend's GEMM order."""
import numpy as np
import pytest

from oracle import binding as ob
import synth


@pytest.mark.parametrize("t", [ob.Q4_K, ob.Q5_1, ob.Q2_K])
@pytest.mark.parametrize("mode", [0, 4])
def test_sampled_block_equals_full_eval(oracle, t, mode):
    hp = synth.HP_TINY_GQA
    w = synth.make_model(oracle, hp, t, seed=5)
    toks = synth.tokens(24, hp["n_vocab"], seed=1)
    oracle.lib.orc_set_sum_order(mode)
    try:
        m = oracle.model(w, 64)
        lg, hid = m.eval(toks[:20], 0, 2, want_hidden=True)
        samp = [0, 3, 19]
        for il in range(hp["n_layer"]):
            assert np.array_equal(m.block_sampled(oracle.lib, il, hid[il], samp), hid[il + 1][samp])
        assert np.array_equal(m.head_rows(oracle.lib, hid[-1][samp]), lg[samp])
        lg2, hid2 = m.eval(toks[20:24], 20, 2, want_hidden=True)
        for il in range(hp["n_layer"]):
            _, k0, v0 = m.block_sampled(oracle.lib, il, hid[il], [0], want_kv=True)
            out = m.block_sampled(oracle.lib, il, hid2[il], [0, 3], pos0=20, k_prev=k0, v_prev=v0)
            assert np.array_equal(out, hid2[il + 1][[0, 3]])
    finally:
        oracle.lib.orc_set_sum_order(0)


@pytest.mark.parametrize("t", [ob.Q4_0, ob.Q4_K])
def test_sampled_block_in_backend_mode_is_told_the_batch(oracle, t):
    """mode 2 ("as the backend") decides per mat-mul from its shape: a few sampled rows OF a 40-token batch must be evaluated with
    the batch's decision (orc_set_backend_batch), and a decode step's K / V rows with one column's (wave order, decode attention)"""
    hp = synth.HP_TINY_GQA
    w = synth.make_model(oracle, hp, t, seed=6)
    toks = synth.tokens(44, hp["n_vocab"], seed=2)
    oracle.lib.orc_set_sum_order(2)
    try:
        m = oracle.model(w, 64)
        lg, hid = m.eval(toks[:40], 0, 2, want_hidden=True)          # 40 columns: GEMM split order, MFMA-order attention
        steps = [m.eval(toks[40 + i:41 + i], 40 + i, 2, want_hidden=True)[1] for i in range(4)]      # single columns
        hid_d = np.concatenate(steps, axis=1)
        samp = [0, 17, 39]
        for il in range(hp["n_layer"]):
            oracle.lib.orc_set_backend_batch(40)
            out, k0, v0 = m.block_sampled(oracle.lib, il, hid[il], samp, want_kv=True)
            assert np.array_equal(out, hid[il + 1][samp])
            oracle.lib.orc_set_backend_batch(1)
            out = m.block_sampled(oracle.lib, il, hid_d[il], [0, 1, 3], pos0=40, k_prev=k0, v_prev=v0)
            assert np.array_equal(out, hid_d[il + 1][[0, 1, 3]])
    finally:
        oracle.lib.orc_set_sum_order(0); oracle.lib.orc_set_backend_batch(0)
