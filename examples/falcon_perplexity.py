#!/usr/bin/env python3
"""Perplexity of a text file, as the reference's falcon_perplexity tool computes it (examples/falcon_perplexity/
falcon_perplexity.cpp:28-124): the text is tokenized with bos in front (`falcon_tokenize(ctx, prompt, true)`), cut into
chunks of n_ctx tokens, every chunk is evaluated from an empty context in batches of n_batch, and the tokens of a chunk's
second half are scored (from min(512, n_ctx / 2) on).

    python examples/falcon_perplexity.py --model falcon-40b-q2_k.ggcc -f wiki.test.raw -c 2048 -b 512
"""
import argparse
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ggllm_cpp_amd as g  # noqa: E402


def perplexity(model_path, text, n_ctx=2048, n_batch=512, device=0):
    """returns (perplexity, scored tokens, token count)"""
    g.init(device)
    vocab = g.Vocab(model_path)
    ids = vocab.tokenize(text, add_bos=True)
    vocab.free()
    if ids.size < n_ctx:
        raise ValueError("the text has %d tokens, fewer than one chunk of n_ctx = %d" % (ids.size, n_ctx))
    n_batch = min(n_batch, n_ctx)
    model = g.FalconModel.from_ggcc(model_path, n_ctx=n_ctx, n_batch=n_batch)
    try:
        nll, n = model.perplexity(ids, n_ctx, n_batch)
    finally:
        model.free()
    return math.exp(nll / n), n, int(ids.size)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", required=True)
    ap.add_argument("-f", "--file", required=True)
    ap.add_argument("-c", "--n-ctx", type=int, default=2048)
    ap.add_argument("-b", "--n-batch", type=int, default=512)
    ap.add_argument("--device", type=int, default=0)
    a = ap.parse_args()
    text = open(a.file, "rb").read()
    ppl, n, total = perplexity(a.model, text, a.n_ctx, a.n_batch, a.device)
    print("perplexity %.4f over %d scored tokens (%d tokens, %d chunks of %d)" % (ppl, n, total, total // a.n_ctx, a.n_ctx))


if __name__ == "__main__":
    main()
