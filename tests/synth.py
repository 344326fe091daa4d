"""Synthetic tensors / models shared by the tests and smoke().

Weights: f32 N(0, 0.02^2) quantized with the oracle's legacy quantizers, or -- for k-quants, whose
quantizers are a "next" row -- random VALID super-blocks (ggllm_cpp_amd.synth). Everything that needs no checker code
(shapes, tokens, fast random-block models) is defined in ggllm.cpp_amd/synth.py and re-exported here.
"""
import numpy as np

from oracle import binding as ob
from ggllm_cpp_amd.synth import *  # noqa: F401,F403
from ggllm_cpp_amd.synth import random_kquant_rows, _pool_bytes  # noqa: F401

Q = ob


def quantized_matrix(oracle, t, rows, k, rng, std=0.02):
    """[rows, row_bytes] uint8 of type t (ggml row-major layout: row r = output row r, k/blck blocks)."""
    if t in ob.KQUANTS:
        return random_kquant_rows(t, rows, k, rng)
    w = (rng.standard_normal((rows, k)) * std).astype(np.float32)
    return oracle.quantize(t, w).reshape(rows, ob.row_bytes(t, k))


def make_model(oracle, hp, wtype, seed=1234, emb_type=None, out_gain=1.0):
    """dict of numpy arrays describing a Falcon model with random-init weights (ggml block bytes). out_gain < 1 (legacy formats): wo and down -- the
    matrices that write the residual stream -- are drawn that much smaller: a residual-dominated, well-conditioned model (ggllm_cpp_amd.synth.make_model_fast)"""
    E, H, HKV, L, FF, V = hp["n_embd"], hp["n_head"], hp["n_head_kv"], hp["n_layer"], hp["n_ff"], hp["n_vocab"]
    D = E // H
    idx = [0]

    def rng():
        idx[0] += 1
        return np.random.default_rng(seed + idx[0])

    def mat(rows, k, gain=1.0):
        return quantized_matrix(oracle, wtype, rows, k, rng(), std=0.02 * gain)

    def ln():
        r = rng()
        return ((1.0 + 0.02 * r.standard_normal(E)).astype(np.float32), (0.02 * r.standard_normal(E)).astype(np.float32))

    m = dict(hparams=dict(hp), wtype=wtype, layers=[])
    m["tok_emb"] = mat(V, E)
    for _ in range(L):
        lw = dict(qkv=mat((H + 2 * HKV) * D, E), wo=mat(E, E, out_gain), up=mat(FF, E), down=mat(E, FF, out_gain))
        lw["ln_w"], lw["ln_b"] = ln()
        if hp.get("two_norms"):
            lw["ln2_w"], lw["ln2_b"] = ln()
        m["layers"].append(lw)
    m["out_norm_w"], m["out_norm_b"] = ln()
    m["lm_head"] = mat(V, E)
    return m




def make_model_float(hp, seed=1234, f16=False):
    """the same model shape with unquantized weights (f32, or f16 for the matrices): the input of falcon_quantize"""
    E, H, HKV, L, FF, V = hp["n_embd"], hp["n_head"], hp["n_head_kv"], hp["n_layer"], hp["n_ff"], hp["n_vocab"]
    D = E // H
    idx = [0]

    def rng():
        idx[0] += 1
        return np.random.default_rng(seed + idx[0])

    def mat(rows, k):
        w = (rng().standard_normal((rows, k)) * 0.02).astype(np.float32)
        return w.astype(np.float16) if f16 else w

    def ln():
        r = rng()
        return ((1.0 + 0.02 * r.standard_normal(E)).astype(np.float32), (0.02 * r.standard_normal(E)).astype(np.float32))

    m = dict(hparams=dict(hp), wtype=1 if f16 else 0, layers=[])
    m["tok_emb"] = mat(V, E)
    for _ in range(L):
        lw = dict(qkv=mat((H + 2 * HKV) * D, E), wo=mat(E, E), up=mat(FF, E), down=mat(E, FF))
        lw["ln_w"], lw["ln_b"] = ln()
        if hp.get("two_norms"):
            lw["ln2_w"], lw["ln2_b"] = ln()
        m["layers"].append(lw)
    m["out_norm_w"], m["out_norm_b"] = ln()
    m["lm_head"] = mat(V, E)
    return m
