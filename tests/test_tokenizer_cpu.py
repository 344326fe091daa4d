"""CPU: falcon_hip_tokenize against the real reference's falcon_tokenize (tests/golden/tokenizer.npz, captured by
oracle/gen_golden.py tokenizer from the reference build on the vocabulary of tests/bpe_fixture.py), the code-point classes
the pre-tokenizer uses against the reference's for all of Unicode, and -- when the reference build is here -- live against
the reference on fresh random strings."""
import ctypes as C
import os

import numpy as np
import pytest

import ggllm_cpp_amd as g
import bpe_fixture
import ggcc_writer
import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def tok_file(tmp_path_factory):
    vocab, merges = bpe_fixture.build()
    hp = dict(synth.HP_TINY_MQA)
    hp["n_vocab"] = len(vocab)
    path = str(tmp_path_factory.mktemp("tok") / "tok.ggcc")
    ggcc_writer.write_ggcc(path, synth.make_model_float(hp, seed=97), vocab, merges)
    return path, vocab, merges


@pytest.fixture(scope="module")
def vocab(tok_file):
    v = g.Vocab(tok_file[0])
    yield v
    v.free()


def test_vocab_loads(vocab, tok_file, golden):
    gt = golden["tokenizer"]
    assert vocab.n_vocab == len(tok_file[1]) == int(gt["n_vocab"])
    assert vocab.n_merges == len(tok_file[2]) == int(gt["n_merges"])
    assert vocab.token_bytes(11) == b"<|endoftext|>" and vocab.token_bytes(12) == b"\x00" and vocab.token_bytes(12 + 0x41) == b"A"
    L = g.load()
    assert L.falcon_hip_token_bos() == 11 and L.falcon_hip_token_eos() == 11
    with pytest.raises(ValueError):
        g.Vocab(os.path.join(ROOT, "README.md"))


def test_token_ids_match_the_reference(vocab, golden):
    gt = golden["tokenizer"]
    n = int(gt["n_texts"])
    assert n >= 190
    bad = []
    for i in range(n):
        raw = gt[f"s{i}"].tobytes()
        for bos in (0, 1):
            got = vocab.tokenize(raw, add_bos=bool(bos))
            if not np.array_equal(got, gt[f"t{i}_{bos}"]):
                bad.append((i, bos, raw, got.tolist(), gt[f"t{i}_{bos}"].tolist()))
    assert not bad, bad[:3]
    assert vocab.tokenize("The quick brown fox", n_max=2) == int(gt["too_small_rc"]) < 0      # minus the count, as falcon_tokenize
    assert vocab.tokenize("").size == 0 and vocab.tokenize("", add_bos=True).size == 0        # empty text: not even bos


def test_round_trip(vocab, golden):
    """byte-level BPE loses nothing: the tokens' bytes concatenate to the text"""
    gt = golden["tokenizer"]
    for i in range(int(gt["n_texts"])):
        raw = gt[f"s{i}"].tobytes()
        assert vocab.detokenize(vocab.tokenize(raw)) == raw


def test_code_point_classes_match_the_reference(golden):
    """letter / digit / whitespace of every code point (tables generated from unicodedata by scripts/gen_unicode_tables.py)
    against the reference's cmpnct_unicode.cpp classes (0 digit, 1 letter, 2 whitespace, others)"""
    import unicodedata
    ref = golden["tokenizer"]["code_class"]
    assert ref.size == 0x110000
    cps = np.arange(0x110000)
    mine = np.full(0x110000, 3, np.uint8)
    chars = [chr(c) for c in cps]
    mine[[ch.isspace() for ch in chars]] = 2
    mine[[ch.isdigit() for ch in chars]] = 0
    mine[[unicodedata.category(ch)[0] == "L" for ch in chars]] = 1
    refc = np.where(ref > 2, 3, ref)
    assert np.array_equal(mine, refc), np.nonzero(mine != refc)[0][:10]
    # and the generated header is what the generator writes today
    hdr = open(os.path.join(ROOT, "ggllm.cpp_amd", "csrc", "fq_unicode_tables.h")).read()
    assert "unicodedata %s" % unicodedata.unidata_version in hdr


def test_live_against_the_reference_build(tok_file, vocab):
    so = os.path.join(ROOT, "oracle", "_ref", "libfalcon_ref.so")
    if not os.path.exists(so):
        pytest.skip("reference build not present (GPU box)")
    L = C.CDLL(so)
    L.reff_load.restype = C.c_void_p; L.reff_load.argtypes = [C.c_char_p, C.c_int, C.c_int]
    L.reff_tokenize.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int, C.c_int]
    L.reff_free.argtypes = [C.c_void_p]
    ctx = L.reff_load(tok_file[0].encode(), 64, 8)
    assert ctx
    rng = np.random.default_rng(2024)
    atoms = list("abcdefghijklmnopqrstuvwxyzABCDEFG0123456789") + [" "] * 8 + ["'", "'", "\n", "\t", ".", ",", "!", "-", "é", "ж", "語", "😀", "²",
             " ", ">>TITLE<<", "<|endoftext|>", ">>", "the", " of", "'re", "'ll", "n't"]
    buf = (C.c_int * 8192)()
    try:
        for _ in range(400):
            k = int(rng.integers(1, 60))
            raw = "".join(atoms[int(i)] for i in rng.integers(0, len(atoms), size=k)).encode("utf-8")
            n = L.reff_tokenize(ctx, raw, buf, 8192, 0)
            assert np.array_equal(vocab.tokenize(raw), np.array(buf[:n], np.int32)), raw
    finally:
        L.reff_free(ctx)
