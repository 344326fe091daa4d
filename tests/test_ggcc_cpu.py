"""GGCC v10 model files, the parts that need no GPU:
  * tests/ggcc_writer.py writes the reference's format byte for byte: the sha256 of each regenerated file equals the one
    recorded when the REAL reference (libfalcon.cpp, oracle/_ref/libfalcon_ref.so) loaded and evaluated that file in the
    build container (tests/golden/ggcc_models.npz, made by oracle/gen_golden.py);
  * the oracle reproduces the reference's logits on those models (bit for bit for the legacy formats);
  * the library's host-side parser (falcon_hip_ggcc_scan) returns the header and the tensor directory the writer put
    there, and refuses damaged files."""
import hashlib
import os
import struct

import numpy as np
import pytest

import ggllm_cpp_amd as g
import ggcc_writer
import synth
from oracle import binding as ob

CASES = [("mqa_q4_0", synth.HP_TINY_MQA, ob.Q4_0), ("gqa_q5_1", synth.HP_TINY_GQA, ob.Q5_1), ("gqa_q4_K", synth.HP_TINY_GQA, ob.Q4_K),
         ("gqa_q6_K", synth.HP_TINY_GQA, ob.Q6_K), ("mqa_q8_0", synth.HP_TINY_MQA, ob.Q8_0)]


@pytest.fixture(scope="module")
def files(oracle, tmp_path_factory):
    d = tmp_path_factory.mktemp("ggcc")
    out = {}
    for name, hp, t in CASES:
        w = synth.make_model(oracle, hp, t, seed=4321)
        p = str(d / (name + ".ggcc"))
        ggcc_writer.write_ggcc(p, w)
        out[name] = (p, w)
    return out


@pytest.mark.parametrize("name", [c[0] for c in CASES])
def test_writer_reproduces_the_file_the_reference_loaded(files, golden, name):
    p, _ = files[name]
    gg = golden["ggcc_models"]
    assert os.path.getsize(p) == int(gg[f"{name}_bytes"])
    assert hashlib.sha256(open(p, "rb").read()).digest() == bytes(gg[f"{name}_sha256"])


@pytest.mark.parametrize("name,hp,t", CASES)
def test_oracle_matches_reference_model_path(oracle, files, golden, name, hp, t):
    """libfalcon.cpp's own loader + graph + falcon_eval (scalar build) vs the oracle on the same weights"""
    _, w = files[name]
    gg = golden["ggcc_models"]
    toks = gg[f"{name}_tokens"]
    m = oracle.model(w, n_ctx=64)
    pre = m.eval(toks[:9], 0, n_threads=1)
    dec = np.concatenate([m.eval(toks[i:i + 1], i, n_threads=1) for i in range(9, 12)])
    if t in ob.LEGACY:
        assert np.array_equal(pre, gg[f"{name}_prefill_logits"])
        assert np.array_equal(dec, gg[f"{name}_decode_logits"])
    else:
        # k-quants: integers exact, the 8-lane float epilogue is associated differently (1e-6 per mat-mul); through two
        # blocks that either stays 1e-6 (Q6_K here) or flips one 8-bit activation rounding (Q4_K here: 4.5e-3) -- the
        # decoder's chaos, DESIGN.md section 2
        for a, b in ((pre, gg[f"{name}_prefill_logits"]), (dec, gg[f"{name}_decode_logits"])):
            assert np.abs(a - b).max() <= 5e-2 * np.sqrt((b.astype(np.float64) ** 2).mean())


@pytest.mark.parametrize("name,hp,t", CASES[:3])
def test_scan_returns_header_and_directory(files, name, hp, t):
    p, w = files[name]
    h, ftype, rows = g.ggcc_scan(p)
    for k in ("n_vocab", "n_embd", "n_head", "n_head_kv", "n_layer", "n_ff"):
        assert h[k] == hp[k]
    assert h["two_norms"] == bool(hp.get("two_norms"))
    assert ftype == ggcc_writer.FTYPE_OF[t]
    want = ggcc_writer.tensor_list(w)
    assert [r[0] for r in rows] == [x[0] for x in want]
    blob = open(p, "rb").read()
    for (name_, ty, ne0, ne1, off, size), (wn, wt, wne, data) in zip(rows, want):
        assert ty == wt and ne0 == wne[0] and ne1 == (wne[1] if len(wne) > 1 else 1)
        assert off % 32 == 0 and size == np.ascontiguousarray(data).nbytes
        assert blob[off:off + size] == np.ascontiguousarray(data).tobytes()


def test_scan_refuses_damaged_files(files, tmp_path):
    p, _ = files["mqa_q4_0"]
    blob = open(p, "rb").read()
    bad = tmp_path / "bad.ggcc"
    for mutated in (struct.pack("<I", 0x67676a74) + blob[4:],        # GGJT magic
                    blob[:4] + struct.pack("<I", 3) + blob[8:],       # wrong version
                    blob[:len(blob) - 1000],                          # truncated tensor data
                    blob[:30],                                        # truncated header
                    blob[:8] + struct.pack("<I", 0) + blob[12:],      # n_vocab 0
                    blob[:8] + struct.pack("<I", 0x7FFFFFFF) + blob[12:],   # n_vocab absurd
                    blob[:24] + struct.pack("<I", 0) + blob[28:],     # n_layer 0
                    blob[:24] + struct.pack("<I", 1 << 20) + blob[28:]):    # n_layer absurd
        bad.write_bytes(mutated)
        with pytest.raises(RuntimeError):
            g.ggcc_scan(str(bad))


def test_stage_planner(oracle, tmp_path):
    """falcon_hip_plan_stages (host-only): contiguous block ranges that minimise the bytes the slowest stage streams (the
    last one also streams lm_head), device bytes per stage, the capacity check"""
    hp = dict(synth.HP_TINY_MQA); hp["n_layer"] = 7
    w = synth.make_model(oracle, hp, ob.Q4_0, seed=9)
    p = str(tmp_path / "seven.ggcc")
    ggcc_writer.write_ggcc(p, w)
    _, _, rows = g.ggcc_scan(p)
    blk = [sum(r[5] for r in rows if r[0].startswith("transformer.h.%d." % i)) for i in range(7)]
    emb = next(r[5] for r in rows if r[0] == "transformer.word_embeddings.weight")
    head = sum(r[5] for r in rows if r[0] in ("lm_head.weight", "transformer.ln_f.weight", "transformer.ln_f.bias"))
    for P in (1, 2, 3, 7):
        parts, bytes_, fits = g.plan_stages(p, P, n_ctx=64, n_batch=4, n_streams=2)
        assert fits and len(parts) == P and parts[0][0] == 0 and parts[-1][1] == 7
        assert all(a[1] == b[0] for a, b in zip(parts, parts[1:])) and all(e > b for b, e in parts)
        load = [sum(blk[b:e]) + (head if i == P - 1 else 0) for i, (b, e) in enumerate(parts)]
        # optimal: no other contiguous partition has a smaller slowest stage (brute force over the cut positions)
        import itertools
        best = min(max(sum(blk[a:b]) + (head if k == P - 1 else 0) for k, (a, b) in enumerate(zip((0,) + cuts, cuts + (7,))))
                   for cuts in itertools.combinations(range(1, 7), P - 1))
        assert max(load) == best
        kv = lambda nb: 2 * nb * 64 * hp["n_head_kv"] * 64 * 4
        for i, (b, e) in enumerate(parts):
            assert bytes_[i] >= sum(blk[b:e]) + (emb if i == 0 else 0) + (head if i == P - 1 else 0) + 2 * kv(e - b)
    # lm_head weighs 0.67 blocks here: with 2 stages the last one gets fewer blocks
    parts, bytes_, _ = g.plan_stages(p, 2, n_ctx=64)
    assert parts[1][1] - parts[1][0] <= parts[0][1] - parts[0][0]
    assert g.plan_stages(p, 2, n_ctx=64, vram_per_gpu=max(bytes_) - 1)[2] is False
    with pytest.raises(RuntimeError):
        g.plan_stages(p, 8)                                   # more stages than blocks
