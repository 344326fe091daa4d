"""Synthetic Falcon models for bench.py / bench_pipeline.py / tests WITHOUT any checker code: shapes of the
BASELINE configs, token streams, and random VALID blocks drawn directly in quantized space (every bit pattern is a legal
block; scales chosen so that the de-quantized weights are roughly zero-mean with std ~0.02-0.05). The oracle-quantized
variants used by the parity tests live in tests/synth.py."""
import numpy as np

from . import (Q4_0, Q4_1, Q5_0, Q5_1, Q8_0, Q2_K, Q3_K, Q4_K, Q5_K, Q6_K, BLCK, TSIZE, KQUANTS, LEGACY)  # noqa: F401

_K_SCALES = {  # (d, dmin/d)
    Q2_K: (4e-3, 1.5), Q3_K: (5e-4, 0.0), Q4_K: (1e-4, 7.5), Q5_K: (5e-5, 15.5), Q6_K: (2e-5, 0.0),
}
# byte offsets of (d, dmin) inside a super-block
_K_DOFF = {Q2_K: (80, 82), Q3_K: (108, None), Q4_K: (0, 2), Q5_K: (0, 2), Q6_K: (208, None)}


def random_kquant_rows(t, rows, k, rng):
    """rows x (k/256) random valid blocks of k-quant type t -> uint8 [rows, row_bytes]."""
    nb = k // 256
    ts = TSIZE[t]
    blk = rng.integers(0, 256, size=(rows, nb, ts), dtype=np.uint8)
    d0, ratio = _K_SCALES[t]
    d = (d0 * rng.uniform(0.5, 1.5, size=(rows, nb))).astype(np.float16)
    doff, moff = _K_DOFF[t]
    blk[:, :, doff:doff + 2] = d.view(np.uint8).reshape(rows, nb, 2)
    if moff is not None:
        dm = (d.astype(np.float32) * ratio * rng.uniform(0.8, 1.2, size=(rows, nb))).astype(np.float16)
        blk[:, :, moff:moff + 2] = dm.view(np.uint8).reshape(rows, nb, 2)
    return blk.reshape(rows, nb * ts)


HP_7B = dict(n_vocab=65024, n_embd=4544, n_head=71, n_head_kv=1, n_layer=32, n_ff=18176, two_norms=False)
HP_40B = dict(n_vocab=65024, n_embd=8192, n_head=128, n_head_kv=8, n_layer=60, n_ff=32768, two_norms=True)
# tiny models for parity tests (head_dim is always 64 in Falcon)
HP_TINY_MQA = dict(n_vocab=512, n_embd=256, n_head=4, n_head_kv=1, n_layer=2, n_ff=1024, two_norms=False)
HP_TINY_GQA = dict(n_vocab=512, n_embd=512, n_head=8, n_head_kv=2, n_layer=2, n_ff=2048, two_norms=True)


def tokens(n, n_vocab, seed=42):
    return np.random.default_rng(seed).integers(0, n_vocab, size=n, dtype=np.int32)


# ---- fast generators for full-size models (bench.py): random VALID blocks drawn directly in quantized space -------
_LEGACY_FAST = {  # type: (delta for std~0.02 weights, offset of d, offset of m or None, m/d)
    Q4_0: (0.02 / 4.6, 0, None, 0.0), Q4_1: (0.02 / 4.6, 0, 2, -7.5), Q5_0: (0.02 / 9.2, 0, None, 0.0),
    Q5_1: (0.02 / 9.2, 0, 2, -15.5), Q8_0: (0.02 / 74.0, 0, None, 0.0),
}


_POOL = None


def _pool_bytes(nbytes, rng):
    """nbytes pseudo-random bytes, fast: a 32 MiB PCG64 pool re-read from a per-call random offset (host RNG speed
    would otherwise dominate the build of a 4-40 GB synthetic model; per-byte statistics are unchanged)."""
    global _POOL
    if _POOL is None:
        _POOL = np.frombuffer(np.random.default_rng(987654321).bytes(32 << 20), np.uint8)
    out = np.empty(nbytes, np.uint8)
    pos = 0
    while pos < nbytes:
        off = int(rng.integers(0, _POOL.size - 1))
        n = min(nbytes - pos, _POOL.size - off)
        out[pos:pos + n] = _POOL[off:off + n]
        pos += n
    return out


def random_blocks(t, rows, k, rng, gain=1.0):
    """[rows, row_bytes] uint8: uniformly random quants, fp16 scales ~U(0.5,1.5)*delta -> zero-mean weights, std ~0.02 * gain (legacy formats)"""
    if t in KQUANTS:
        return random_kquant_rows(t, rows, k, rng)
    nb, ts = k // 32, TSIZE[t]
    blk = _pool_bytes(rows * nb * ts, rng).reshape(rows, nb, ts)
    d0, doff, moff, ratio = _LEGACY_FAST[t]
    d0 = d0 * gain
    sel = _pool_bytes(rows * nb, rng).reshape(rows, nb)                # 256 scale levels in [0.5, 1.5) * d0
    dtab = (d0 * (0.5 + np.arange(256) / 256.0)).astype(np.float16)
    blk[:, :, doff:doff + 2] = dtab.view(np.uint8).reshape(256, 2)[sel]
    if moff is not None:
        mtab = (dtab.astype(np.float32) * ratio).astype(np.float16)
        blk[:, :, moff:moff + 2] = mtab.view(np.uint8).reshape(256, 2)[sel]
    return blk.reshape(rows, nb * ts)


def make_model_fast(hp, wtype, seed=1234, layers=None, out_gain=1.0):
    """like make_model but with random_blocks for every matrix; `layers` = iterable of layer ids to materialise.
    out_gain < 1 (legacy formats): the two matrices that WRITE the residual stream (wo, down) are drawn out_gain times smaller -- a residual-dominated,
    well-conditioned model: with std-0.02 blocks the block outputs (O(1) per element) swamp the embedding (0.02), and one flipped 8-bit activation
    rounding anywhere moves the logits by 1e-3..1e-2; with out_gain = 2^-12 no such flip showed above 1e-5 on any model tried (profiles/r05_wellcond.txt)"""
    E, H, HKV, L, FF, V = hp["n_embd"], hp["n_head"], hp["n_head_kv"], hp["n_layer"], hp["n_ff"], hp["n_vocab"]
    want = set(range(L)) if layers is None else set(layers)

    def rng(i):
        return np.random.default_rng(seed + i)        # per-tensor seed = 1234 + tensor index

    def ln(i):
        r = rng(i)
        return ((1.0 + 0.02 * r.standard_normal(E)).astype(np.float32), (0.02 * r.standard_normal(E)).astype(np.float32))

    m = dict(hparams=dict(hp), wtype=wtype, layers=[])
    m["tok_emb"] = random_blocks(wtype, V, E, rng(0))
    for il in range(L):
        if il not in want:
            m["layers"].append(None)
            continue
        b = 10 + il * 8
        lw = dict(qkv=random_blocks(wtype, (H + 2 * HKV) * 64, E, rng(b)), wo=random_blocks(wtype, E, E, rng(b + 1), gain=out_gain),
                  up=random_blocks(wtype, FF, E, rng(b + 2)), down=random_blocks(wtype, E, FF, rng(b + 3), gain=out_gain))
        lw["ln_w"], lw["ln_b"] = ln(b + 4)
        if hp.get("two_norms"):
            lw["ln2_w"], lw["ln2_b"] = ln(b + 5)
        m["layers"].append(lw)
    m["out_norm_w"], m["out_norm_b"] = ln(5)
    m["lm_head"] = random_blocks(wtype, V, E, rng(1))
    return m
