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
