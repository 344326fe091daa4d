#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REAL reference (oracle/_ref, built by oracle/Makefile from
/root/reference). Runs only in the dev container; the fixtures it writes are data (inputs + expected
outputs) and are committed. Re-run: `python oracle/gen_golden.py`.

Every fixture records results of both reference builds:
  *_avx     libggml_ref.so         (-march=x86-64-v3: the AVX2 branches, what a real x86 run executes)
  *_scalar  libggml_ref_scalar.so  (SIMD disabled: the reference's portable #else branches)
The spread between the two is the legitimate float-association / rounding-flavour spread of the reference.
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import binding as ob  # noqa: E402
import synth  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def weights_digest(w):
    h = hashlib.sha256()
    for name in ("tok_emb", "lm_head", "out_norm_w", "out_norm_b"):
        h.update(np.ascontiguousarray(w[name]).tobytes())
    for lw in w["layers"]:
        for name in sorted(lw):
            h.update(np.ascontiguousarray(lw[name]).tobytes())
    return h.hexdigest()


def gen_ggcc(O):
    """5. the reference's MODEL path on model FILES: synthetic GGCC v10 files (tests/ggcc_writer.py) loaded and evaluated
    by the real libfalcon.cpp (falcon_init_from_file -> its loader, graph builder, falcon_eval; scalar build,
    oracle/_ref/libfalcon_ref.so). The fixture holds the file's sha256 (the writer is deterministic), the tokens and the
    reference's logits for a 9-token prompt and 3 single-token steps."""
    import ctypes as C
    import tempfile
    import ggcc_writer
    so = os.path.join(ROOT, "oracle", "_ref", "libfalcon_ref.so")
    if not os.path.exists(so):
        print("oracle/_ref/libfalcon_ref.so not built (make -C oracle ref_falcon): ggcc_models.npz not regenerated")
        return
    L = C.CDLL(so)
    L.reff_load.restype = C.c_void_p; L.reff_load.argtypes = [C.c_char_p, C.c_int, C.c_int]
    L.reff_eval.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.reff_free.argtypes = [C.c_void_p]
    L.reff_token_nll.restype = C.c_double; L.reff_token_nll.argtypes = [C.c_void_p, C.c_int, C.c_int]
    d = {}
    for name, hp, t in (("mqa_q4_0", synth.HP_TINY_MQA, ob.Q4_0), ("gqa_q5_1", synth.HP_TINY_GQA, ob.Q5_1),
                        ("gqa_q4_K", synth.HP_TINY_GQA, ob.Q4_K), ("gqa_q6_K", synth.HP_TINY_GQA, ob.Q6_K), ("mqa_q8_0", synth.HP_TINY_MQA, ob.Q8_0)):
        w = synth.make_model(O, hp, t, seed=4321)
        with tempfile.TemporaryDirectory() as td:
            path = os.path.join(td, name + ".ggcc")
            ggcc_writer.write_ggcc(path, w)
            d[f"{name}_sha256"] = np.frombuffer(hashlib.sha256(open(path, "rb").read()).digest(), np.uint8)
            d[f"{name}_bytes"] = np.int64(os.path.getsize(path))
            ctx = L.reff_load(path.encode(), 64, 16)
            assert ctx, name
            toks = synth.tokens(12, hp["n_vocab"], seed=77)
            d[f"{name}_tokens"] = toks
            lg = np.zeros((9, hp["n_vocab"]), np.float32)
            assert L.reff_eval(ctx, toks[:9].ctypes.data, 9, 0, 1, lg.ctypes.data) == 0
            d[f"{name}_prefill_logits"] = lg
            dec = []
            for i in range(9, 12):
                one = np.zeros((1, hp["n_vocab"]), np.float32)
                assert L.reff_eval(ctx, toks[i:i + 1].ctypes.data, 1, i, 1, one.ctypes.data) == 0
                dec.append(one)
            d[f"{name}_decode_logits"] = np.concatenate(dec)
            if name in ("mqa_q4_0", "gqa_q5_1"):
                # the perplexity loop (falcon_perplexity.cpp:28-124) driven by hand over the reference's falcon_eval:
                # 3 chunks of n_ctx 32 in batches of 8, NLL of positions 16..30 of every chunk
                n_ctx, n_batch = 32, 8
                stream = synth.tokens(3 * n_ctx + 5, hp["n_vocab"], seed=99)
                nll, count = 0.0, 0
                for i in range(len(stream) // n_ctx):
                    start = i * n_ctx
                    lgs = np.zeros((n_ctx, hp["n_vocab"]), np.float32)
                    for j in range(n_ctx // n_batch):
                        b0 = start + j * n_batch
                        part = np.zeros((n_batch, hp["n_vocab"]), np.float32)
                        assert L.reff_eval(ctx, stream[b0:b0 + n_batch].ctypes.data, n_batch, j * n_batch, 1, part.ctypes.data) == 0
                        lgs[j * n_batch:(j + 1) * n_batch] = part
                    for j in range(min(512, n_ctx // 2), n_ctx - 1):
                        nll += L.reff_token_nll(lgs[j].ctypes.data, hp["n_vocab"], int(stream[start + j + 1]))
                        count += 1
                d[f"{name}_ppl_tokens"] = stream
                d[f"{name}_ppl_nll"] = np.float64(nll)
                d[f"{name}_ppl_count"] = np.int64(count)
            L.reff_free(ctx)
    np.savez_compressed(os.path.join(OUT, "ggcc_models.npz"), **d)


QUANT_FTYPES = {"gqa_f32": (0, 1, 2, 3, 7, 8, 9, 10, 12, 15, 17, 18), "mqa_f16": (0, 2, 9), "gqa_f32_keep_output": (15,)}


def gen_model_quantize():
    """6. falcon_model_quantize of the real reference (one thread) on synthetic f32 / f16 GGCC files: sha256 + size of
    every output file; tests/test_gpu_falcon.py quantizes the same inputs on the device and compares the files."""
    import ctypes as C
    import tempfile
    import ggcc_writer
    so = os.path.join(ROOT, "oracle", "_ref", "libfalcon_ref.so")
    if not os.path.exists(so):
        print("oracle/_ref/libfalcon_ref.so not built: model_quantize.npz not regenerated")
        return
    L = C.CDLL(so)
    L.reff_quantize.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int]
    d = {}
    for name, ftypes in QUANT_FTYPES.items():
        hp = synth.HP_TINY_GQA if name.startswith("gqa") else synth.HP_TINY_MQA
        w = synth.make_model_float(hp, seed=2468, f16=name.endswith("f16"))
        keep = name.endswith("keep_output")
        with tempfile.TemporaryDirectory() as td:
            src = os.path.join(td, "in.ggcc")
            ggcc_writer.write_ggcc(src, w)
            d[f"{name}_src_sha256"] = np.frombuffer(hashlib.sha256(open(src, "rb").read()).digest(), np.uint8)
            for ft in ftypes:
                dst = os.path.join(td, "out_%d.ggcc" % ft)
                rc = L.reff_quantize(src.encode(), dst.encode(), ft, 0 if keep else 1, 0)
                assert rc == 0, (name, ft)
                d[f"{name}_ft{ft}_sha256"] = np.frombuffer(hashlib.sha256(open(dst, "rb").read()).digest(), np.uint8)
                d[f"{name}_ft{ft}_bytes"] = np.int64(os.path.getsize(dst))
            if name == "gqa_f32":                               # requantize: the Q8_0 file again to Q4_0 and Q6_K
                q8 = os.path.join(td, "out_7.ggcc")
                assert L.reff_quantize(q8.encode(), os.path.join(td, "rq.ggcc").encode(), 2, 1, 0) == 1   # refused by default
                for ft in (2, 18):
                    dst = os.path.join(td, "rq_%d.ggcc" % ft)
                    assert L.reff_quantize(q8.encode(), dst.encode(), ft, 1, 1) == 0
                    d[f"{name}_requant_ft{ft}_sha256"] = np.frombuffer(hashlib.sha256(open(dst, "rb").read()).digest(), np.uint8)
    np.savez_compressed(os.path.join(OUT, "model_quantize.npz"), **d)


def wquant_inputs():
    """inputs of the weight-quantizer vectors (tests/golden/wquant.npz): ordinary weights plus the branches the reference's
    quantizers special-case -- all-zero and constant (sub-)blocks, sparse rows, heavy tails, huge and tiny magnitudes"""
    rng = np.random.default_rng(20240926)
    n = 2048
    x = {}
    x["gauss"] = (rng.standard_normal(n) * 0.02).astype(np.float32)
    x["uniform"] = rng.uniform(-1, 1, n).astype(np.float32)
    v = (rng.standard_normal(n) * 0.05).astype(np.float32); v[rng.random(n) < 0.3] = 0
    x["sparse"] = v
    x["heavy"] = (rng.standard_normal(n).astype(np.float32) ** 3).astype(np.float32)
    v = (rng.standard_normal(n) * 0.02).astype(np.float32)
    v[:256] = 0; v[512:528] = 0; v[768:800] = 0.5; v[1024:1280] = -0.25; v[1500] = 1e30; v[1600] = 3.0; v[1792:1808] = 1e-30
    x["edges"] = v
    x["positive"] = np.abs(rng.standard_normal(n) * 0.1).astype(np.float32)
    return x


def gen_wquant():
    """ggml_quantize_chunk of the reference build on wquant_inputs(): blocks + histogram per format"""
    R = ob.Ref()
    d = {}
    for name, xv in wquant_inputs().items():
        d[f"x_{name}"] = xv
        for t in ob.WEIGHT_TYPES:
            q, h = R.quantize_chunk(t, xv)
            d[f"{ob.TYPE_NAME[t]}_{name}_q"], d[f"{ob.TYPE_NAME[t]}_{name}_hist"] = q, h
    np.savez_compressed(os.path.join(OUT, "wquant.npz"), **d)


def fuzz_texts(n=160, seed=99):
    """seeded random strings over an alphabet that hits every branch of the pre-tokenizer"""
    rng = np.random.default_rng(seed)
    atoms = list("abcdefgstmdrvlexyzABC") + list("0123456789") + [" ", " ", " ", "  ", "\n", "\t", "'", "'", ".", ",", "!", "?", "-", "_", "(", ")", "\"",
             "é", "ß", "ж", "語", "한", "😀", "½", "²", "٣", "\u00a0", "\u2003", ">>TITLE<<", "<|endoftext|>", ">>", "<|", "the", " the", " and", "ing", "tion"]
    out = []
    for _ in range(n):
        k = int(rng.integers(1, 24))
        out.append("".join(atoms[int(i)] for i in rng.integers(0, len(atoms), size=k)))
    return out


def gen_tokenizer():
    """7. falcon_tokenize of the real reference on the BPE vocabulary of tests/bpe_fixture.py (stored in a synthetic GGCC file
    as the reference's loader expects it): token ids for the fixture's texts and for seeded random strings, with and
    without bos; and the reference's class of every code point (what the pre-tokenizer's letter / digit / whitespace tests
    see). tests/test_tokenizer_cpu.py runs falcon_hip_tokenize on the same file."""
    import ctypes as C
    import tempfile
    import ggcc_writer
    import bpe_fixture
    so = os.path.join(ROOT, "oracle", "_ref", "libfalcon_ref.so")
    if not os.path.exists(so):
        print("oracle/_ref/libfalcon_ref.so not built: tokenizer.npz not regenerated")
        return
    L = C.CDLL(so)
    L.reff_load.restype = C.c_void_p; L.reff_load.argtypes = [C.c_char_p, C.c_int, C.c_int]
    L.reff_tokenize.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int, C.c_int]
    L.reff_free.argtypes = [C.c_void_p]
    vocab, merges = bpe_fixture.build()
    hp = dict(synth.HP_TINY_MQA); hp["n_vocab"] = len(vocab)
    w = synth.make_model_float(hp, seed=97)
    d = {"n_vocab": np.int64(len(vocab)), "n_merges": np.int64(len(merges))}
    texts = list(bpe_fixture.TEXTS) + fuzz_texts()
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "tok.ggcc")
        ggcc_writer.write_ggcc(path, w, vocab, merges)
        d["file_head_sha256"] = np.frombuffer(hashlib.sha256(open(path, "rb").read()[:1 << 16]).digest(), np.uint8)
        ctx = L.reff_load(path.encode(), 64, 8)
        assert ctx
        buf = (C.c_int * 4096)()
        for i, t in enumerate(texts):
            raw = t.encode("utf-8")
            d[f"s{i}"] = np.frombuffer(raw, np.uint8)
            for bos in (0, 1):
                n = L.reff_tokenize(ctx, raw, buf, 4096, bos)
                assert n >= 0, (t, n)
                d[f"t{i}_{bos}"] = np.array(buf[:n], np.int32)
        n_small = L.reff_tokenize(ctx, b"The quick brown fox", buf, 2, 0)          # too small: minus the count
        d["too_small_rc"] = np.int64(n_small)
        L.reff_free(ctx)
    d["n_texts"] = np.int64(len(texts))
    L.reff_code_type.argtypes = [C.c_int]
    cls = np.fromiter((L.reff_code_type(c) for c in range(0x110000)), np.uint8, count=0x110000)
    d["code_class"] = cls
    np.savez_compressed(os.path.join(OUT, "tokenizer.npz"), **d)
    print("tokenizer.npz:", len(texts), "texts,", len(vocab), "tokens,", len(merges), "merges")


TINY_ALL = [("gqa_q4_1", ob.Q4_1), ("gqa_q5_0", ob.Q5_0), ("gqa_q2_K", ob.Q2_K), ("gqa_q3_K", ob.Q3_K), ("gqa_q5_K", ob.Q5_K), ("gqa_q6_K", ob.Q6_K)]


def gen_tiny_all():
    """4b. the six formats tiny_models.npz does not hold, same recipe (whole tiny GQA stack through the reference's graph
    executor, AVX2 and scalar builds): tests/golden/tiny_models_all.npz. With it every one of the ten weight formats has
    model-level logits of the REAL reference to be bit-identical with."""
    ob.build_oracle()
    O, R, RS = ob.Oracle(), ob.Ref(), ob.Ref(scalar=True)
    d = {}
    hp = synth.HP_TINY_GQA
    for name, t in TINY_ALL:
        w = synth.make_model(O, hp, t, seed=1234)
        d[f"{name}_digest"] = np.frombuffer(bytes.fromhex(weights_digest(w)), np.uint8)
        toks = synth.tokens(12, hp["n_vocab"], seed=42)
        d[f"{name}_tokens"] = toks
        for tag, lib in (("avx", R), ("scalar", RS)):
            m = lib.model(w, 64)
            lg, hid = m.eval(toks[:8], 0, 2, want_hidden=True)
            d[f"{name}_prefill_logits_{tag}"] = lg
            if tag == "scalar":
                d[f"{name}_prefill_hidden_{tag}"] = hid
            d[f"{name}_decode_logits_{tag}"] = np.concatenate([m.eval(toks[i:i + 1], i, 2) for i in range(8, 12)])
    np.savez_compressed(os.path.join(OUT, "tiny_models_all.npz"), **d)
    print("tiny_models_all.npz", os.path.getsize(os.path.join(OUT, "tiny_models_all.npz")))


CLI_PROMPT = "The quick brown fox didn't jump"
CLI_MAIN_ARGS = ["-n", "8", "--temp", "0", "-t", "2", "-c", "64", "-b", "8", "--ignore-eos", "-s", "1"]
CLI_PPL_ARGS = ["-t", "2", "-c", "32", "-b", "8", "-s", "1"]
# -c 4096 with a short real context: falcon_main sets n_max_real_ctx = min(n_ctx, prompt + n_predict) (falcon_main.cpp:836), so the
# rope sees a small n_ctx (NTK factor 1) although the context was created for 4096 (factor 3)
CLI_MAIN_ARGS_C4096 = ["-n", "8", "--temp", "0", "-t", "2", "-c", "4096", "-b", "8", "--ignore-eos", "-s", "1"]


def cli_model(O, path):
    """the tiny byte-level-BPE model the CLI fixtures run on (also built by tests/test_gpu_dropin.py)"""
    import bpe_fixture
    import ggcc_writer
    vocab, merges = bpe_fixture.build(n_merges=308)
    hp = dict(synth.HP_TINY_MQA)
    hp["n_vocab"] = len(vocab)
    w = synth.make_model(O, hp, ob.Q4_0, seed=321)
    ggcc_writer.write_ggcc(path, w, vocab, merges)
    return bpe_fixture


def gen_cli():
    """8. the reference's own command-line tools, built WITHOUT any back-end from its unchanged sources (scalar objects of
    `make -C oracle ref_falcon` + examples/falcon/falcon_main.cpp, examples/falcon_perplexity/falcon_perplexity.cpp,
    examples/falcon_common.cpp; build-info.h written by the reference's scripts/build-info.sh) and run on a tiny GGCC file:
    the bytes falcon_main prints for a prompt (greedy, its default repetition penalty) and the chunk perplexities
    falcon_perplexity prints for a text. tests/test_gpu_dropin.py runs the same tools linked against libggml_hip.so."""
    import subprocess
    import tempfile
    ref = "/root/reference"
    obj = os.path.join(ROOT, "oracle", "_ref", "obj")
    sc = "-march=x86-64 -mno-sse3 -mno-ssse3 -mno-avx -mno-avx2 -mno-fma -mno-f16c".split()
    ob.build_oracle()
    O = ob.Oracle()
    d = {}
    with tempfile.TemporaryDirectory() as td:
        subprocess.check_call("sh %s/scripts/build-info.sh > build-info.h 2>/dev/null" % ref, shell=True, cwd=td)
        objs = [os.path.join(obj, f) for f in ("libfalcon.o", "cmpnct_unicode.o", "ggml.o", "k_quants.o")]
        for tool, src in (("falcon_main", "examples/falcon/falcon_main.cpp"), ("falcon_perplexity", "examples/falcon_perplexity/falcon_perplexity.cpp")):
            subprocess.check_call(["g++", "-O2", "-std=c++11", "-pthread", *sc, "-DGGML_USE_K_QUANTS", "-I" + ref, "-I" + ref + "/examples", "-I" + td,
                                   os.path.join(ref, src), os.path.join(ref, "examples/falcon_common.cpp"), *objs, "-lm", "-o", os.path.join(td, tool)],
                                  stderr=subprocess.DEVNULL)
        path = os.path.join(td, "tiny_bpe.ggcc")
        bf = cli_model(O, path)
        r = subprocess.run([os.path.join(td, "falcon_main"), "-m", path, "-p", CLI_PROMPT, *CLI_MAIN_ARGS], capture_output=True)
        assert r.returncode == 0, r.stderr[-2000:]
        d["main_stdout"] = np.frombuffer(r.stdout, np.uint8)
        r = subprocess.run([os.path.join(td, "falcon_main"), "-m", path, "-p", CLI_PROMPT, *CLI_MAIN_ARGS_C4096], capture_output=True)
        assert r.returncode == 0, r.stderr[-2000:]
        d["main_c4096_stdout"] = np.frombuffer(r.stdout, np.uint8)
        print("falcon_main -c 4096:", r"%r" % bytes(d["main_c4096_stdout"]))
        txt = os.path.join(td, "corpus.txt")
        open(txt, "wb").write((bf.CORPUS * 2).encode("utf-8"))
        r = subprocess.run([os.path.join(td, "falcon_perplexity"), "-m", path, "-f", txt, *CLI_PPL_ARGS], capture_output=True)
        assert r.returncode == 0, r.stderr[-2000:]
        d["ppl_stdout"] = np.frombuffer(r.stdout, np.uint8)
        print("falcon_main:", r"%r" % bytes(d["main_stdout"]))
        print("falcon_perplexity:", r"%r" % bytes(d["ppl_stdout"]))
    np.savez_compressed(os.path.join(OUT, "cli.npz"), **d)


def main():
    if len(sys.argv) > 1 and sys.argv[1] in ("wquant", "model_quantize", "tokenizer", "tiny_all", "cli"):
        os.makedirs(OUT, exist_ok=True)
        return {"wquant": gen_wquant, "model_quantize": gen_model_quantize, "tokenizer": gen_tokenizer, "tiny_all": gen_tiny_all, "cli": gen_cli}[sys.argv[1]]()
    ob.build_oracle()
    O, R, RS = ob.Oracle(), ob.Ref(), ob.Ref(scalar=True)
    os.makedirs(OUT, exist_ok=True)

    # ---- 1. quantize / dequantize / vec_dot (the functions tests/test-quantize-fns.cpp pins) -------------
    d = {}
    n = 4096
    x_cos = (0.1 + 2.0 * np.cos(np.arange(n, dtype=np.float32) + 0.0)).astype(np.float32)   # test-quantize-fns.cpp:24-30
    x_cos1 = (0.1 + 2.0 * np.cos(np.arange(n, dtype=np.float32) + 1.0)).astype(np.float32)
    rng = np.random.default_rng(20240917)
    x_gau = (rng.standard_normal(n) * 0.02).astype(np.float32)
    d["x_cos"], d["x_cos1"], d["x_gau"] = x_cos, x_cos1, x_gau
    for t in ob.WEIGHT_TYPES:
        nm = ob.TYPE_NAME[t]
        for xn, xv in (("cos", x_cos), ("gau", x_gau)):
            q = R.quantize(t, xv)                       # quantize_row_q_reference (also for k-quants)
            d[f"{nm}_{xn}_q"] = q
            d[f"{nm}_{xn}_deq"] = R.dequantize(t, q, n)
            act_avx = R.quantize_dot(t, x_cos1)
            act_sc = RS.quantize_dot(t, x_cos1)
            d[f"{nm}_act_avx"], d[f"{nm}_act_scalar"] = act_avx, act_sc
            d[f"{nm}_{xn}_dot_avx"] = np.float32(R.vec_dot(t, n, q, act_avx))
            d[f"{nm}_{xn}_dot_scalar"] = np.float32(RS.vec_dot(t, n, q, act_sc))
    # Falcon-7B row lengths (legacy types only: 4544 % 256 != 0) and 18176
    for K in (4544, 18176):
        xk = rng.standard_normal(K).astype(np.float32)
        wk = (rng.standard_normal((3, K)) * 0.02).astype(np.float32)
        d[f"x_{K}"], d[f"w_{K}"] = xk, wk
        for t in ob.WEIGHT_TYPES:
            if K % ob.BLCK[t]:
                continue
            nm = ob.TYPE_NAME[t]
            q = np.stack([R.quantize(t, wk[r]) for r in range(3)])
            d[f"{nm}_{K}_q"] = q
            a_sc = RS.quantize_dot(t, xk)
            d[f"{nm}_{K}_dot_avx"] = np.array([R.vec_dot(t, K, q[r], R.quantize_dot(t, xk)) for r in range(3)], np.float32)
            d[f"{nm}_{K}_dot_scalar"] = np.array([RS.vec_dot(t, K, q[r], a_sc) for r in range(3)], np.float32)
    np.savez_compressed(os.path.join(OUT, "quant_fns.npz"), **d)

    # ---- 2. mul_mat through ggml_graph_compute ---------------------------------------------------------
    d = {}
    for t in ob.WEIGHT_TYPES:
        nm = ob.TYPE_NAME[t]
        K, M, N = 512, 48, 5
        r2 = np.random.default_rng(100 + t)
        w = R.quantize(t, (r2.standard_normal((M, K)) * 0.02).astype(np.float32))
        x = r2.standard_normal((N, K)).astype(np.float32)
        d[f"{nm}_w"], d[f"{nm}_x"] = w, x
        d[f"{nm}_y_avx"] = R.mul_mat(t, w, K, M, x, 3)
        d[f"{nm}_y_scalar"] = RS.mul_mat(t, w, K, M, x, 2)
    np.savez_compressed(os.path.join(OUT, "mul_mat.npz"), **d)

    # ---- 3. block ops ----------------------------------------------------------------------------------
    d = {}
    r3 = np.random.default_rng(7)
    xn = (r3.standard_normal((5, 4544)) * 3 + 0.5).astype(np.float32)
    d["norm_x"], d["norm_y"] = xn, RS.norm(xn)
    assert np.array_equal(R.norm(xn), d["norm_y"])
    gx = np.concatenate([r3.standard_normal(4096).astype(np.float32) * 3,
                         np.array([0, -0.0, 1e-8, 11.0, -11.0, 60000, -60000], np.float32)])
    d["gelu_x"], d["gelu_y"] = gx, RS.gelu(gx)
    assert np.array_equal(R.gelu(gx), d["gelu_y"])
    # the full fp16 GELU table through the op (all 65536 halves as inputs), and exp through soft_max is
    # covered by softmax below; the gelu table is stored packed as fp16 bits.
    allh = np.arange(1 << 16, dtype=np.uint16).view(np.float16).astype(np.float32)
    finite = np.isfinite(allh)
    gy = RS.gelu(allh[finite])
    d["gelu_all_in_bits"] = np.arange(1 << 16, dtype=np.uint16)[finite]
    d["gelu_all_out_bits"] = gy.astype(np.float16).view(np.uint16)   # exact: outputs are fp16-representable
    assert np.array_equal(gy.astype(np.float16).astype(np.float32), gy)
    for n_ctx in (2048, 8192):
        xr = r3.standard_normal((3, 5, 64)).astype(np.float32)
        d[f"rope_x_{n_ctx}"] = xr
        d[f"rope_y_{n_ctx}"] = RS.rope(xr, 64, 5, 3, 1021, n_ctx)
        assert np.array_equal(R.rope(xr, 64, 5, 3, 1021, n_ctx), d[f"rope_y_{n_ctx}"])
    kq = (r3.standard_normal((4, 3, 40)) * 6).astype(np.float32)
    d["sm_kq"], d["sm_n_past"] = kq, np.int32(37)
    d["sm_p"] = RS.scale_mask_softmax(kq, 37, 0.125)
    assert np.array_equal(R.scale_mask_softmax(kq, 37, 0.125), d["sm_p"])
    np.savez_compressed(os.path.join(OUT, "block_ops.npz"), **d)

    # ---- 4. whole tiny Falcon models (weights regenerated from the seed; digest pins them) --------------
    d = {}
    cases = [("mqa_q4_0", synth.HP_TINY_MQA, ob.Q4_0), ("gqa_q5_1", synth.HP_TINY_GQA, ob.Q5_1),
             ("gqa_q4_K", synth.HP_TINY_GQA, ob.Q4_K), ("mqa_q8_0", synth.HP_TINY_MQA, ob.Q8_0)]
    for name, hp, t in cases:
        w = synth.make_model(O, hp, t, seed=1234)
        d[f"{name}_digest"] = np.frombuffer(bytes.fromhex(weights_digest(w)), np.uint8)
        toks = synth.tokens(12, hp["n_vocab"], seed=42)
        d[f"{name}_tokens"] = toks
        for tag, lib in (("avx", R), ("scalar", RS)):
            m = lib.model(w, 64)
            lg, hid = m.eval(toks[:8], 0, 2, want_hidden=True)
            d[f"{name}_prefill_logits_{tag}"] = lg
            d[f"{name}_prefill_hidden_{tag}"] = hid
            dec = [m.eval(toks[i:i + 1], i, 2) for i in range(8, 12)]
            d[f"{name}_decode_logits_{tag}"] = np.concatenate(dec)
    np.savez_compressed(os.path.join(OUT, "tiny_models.npz"), **d)
    gen_tiny_all()
    gen_ggcc(O)
    gen_wquant()
    gen_model_quantize()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
