"""Writer for the reference's model file format, GGCC v10 (what falcon_quantize emits and falcon_main loads):
layout as parsed by falcon_file_loader (libfalcon.cpp:770-973): magic 0x67676363, version 10, eight u32 hparams
{n_vocab, n_embd, n_head, n_head_kv, n_layer, n_falcon_type, ftype, n_bpe_merges}, the vocabulary (u32 len, bytes,
f32 score per token), the BPE merges (u32 count, then two length-prefixed strings each), then until EOF one record per
tensor: u32 n_dims, u32 name_len, u32 ggml_type, u32 ne[n_dims], name, zero padding to a multiple of 32, data.
Test infrastructure: produces synthetic model files for the loader tests (and for the real reference, which
oracle/gen_golden.py runs on them in the build container)."""
import struct

import numpy as np

from oracle import binding as ob

GGCC_MAGIC, GGCC_VERSION = 0x67676363, 10
FTYPE_OF = {0: 0, 1: 1, ob.Q4_0: 2, ob.Q4_1: 3, ob.Q8_0: 7, ob.Q5_0: 8, ob.Q5_1: 9, ob.Q2_K: 10, ob.Q3_K: 12, ob.Q4_K: 15, ob.Q5_K: 17, ob.Q6_K: 18}
F32 = 0
NAMES_7B = {"ln_w": "input_layernorm.weight", "ln_b": "input_layernorm.bias"}
NAMES_40B = {"ln_w": "ln_mlp.weight", "ln_b": "ln_mlp.bias", "ln2_w": "ln_attn.weight", "ln2_b": "ln_attn.bias"}


def tensor_list(weights):
    """[(name, ggml type, ne (ne0 = row length first), bytes)] in the converter's order (falcon_convert.py)"""
    hp, wt = weights["hparams"], weights["wtype"]
    E, H, HKV, FF, V = hp["n_embd"], hp["n_head"], hp["n_head_kv"], hp["n_ff"], hp["n_vocab"]
    out = [("transformer.word_embeddings.weight", wt, (E, V), weights["tok_emb"])]
    names = NAMES_40B if hp.get("two_norms") else NAMES_7B
    for i, lw in enumerate(weights["layers"]):
        p = f"transformer.h.{i}."
        for k, leaf in names.items():
            out.append((p + leaf, F32, (E,), lw[k]))
        out.append((p + "self_attention.query_key_value.weight", wt, (E, (H + 2 * HKV) * 64), lw["qkv"]))
        out.append((p + "self_attention.dense.weight", wt, (E, E), lw["wo"]))
        out.append((p + "mlp.dense_h_to_4h.weight", wt, (E, FF), lw["up"]))
        out.append((p + "mlp.dense_4h_to_h.weight", wt, (FF, E), lw["down"]))
    out.append(("transformer.ln_f.weight", F32, (E,), weights["out_norm_w"]))
    out.append(("transformer.ln_f.bias", F32, (E,), weights["out_norm_b"]))
    out.append(("lm_head.weight", wt, (E, V), weights["lm_head"]))
    return out


def write_ggcc(path, weights, vocab=None, merges=None):
    """vocab: list of token byte strings (len == n_vocab), merges: list of (bytes, bytes) pairs in the printable BPE alphabet;
    default: unique dummy tokens and no merges (the tokenizer is not on the eval path)"""
    hp, wt = weights["hparams"], weights["wtype"]
    assert hp["n_ff"] == 4 * hp["n_embd"], "the format does not store n_ff: the loader assumes 4 * n_embd (libfalcon.cpp:1598)"
    with open(path, "wb") as f:
        f.write(struct.pack("<II", GGCC_MAGIC, GGCC_VERSION))
        f.write(struct.pack("<8I", hp["n_vocab"], hp["n_embd"], hp["n_head"], hp["n_head_kv"], hp["n_layer"],
                            40 if hp.get("two_norms") else 7, FTYPE_OF[wt], len(merges) if merges else 0))
        assert vocab is None or len(vocab) == hp["n_vocab"]
        for i in range(hp["n_vocab"]):
            w = vocab[i] if vocab is not None else ("<%d>" % i).encode()
            f.write(struct.pack("<I", len(w))); f.write(w); f.write(struct.pack("<f", 0.0))
        f.write(struct.pack("<I", len(merges) if merges else 0))        # BPE merges
        for a, b in (merges or []):
            f.write(struct.pack("<I", len(a))); f.write(a); f.write(struct.pack("<I", len(b))); f.write(b)
        for name, t, ne, data in tensor_list(weights):
            nb = name.encode()
            f.write(struct.pack("<III", len(ne), len(nb), t))
            f.write(struct.pack("<%dI" % len(ne), *ne))
            f.write(nb)
            f.write(b"\0" * (-f.tell() & 31))
            f.write(np.ascontiguousarray(data).tobytes())
