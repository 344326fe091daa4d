"""A small, self-consistent byte-level BPE vocabulary for the tokenizer tests: the 12 special tokens of a Falcon vocabulary
(ids 0..11), the 256 single bytes, and the merges a tiny BPE trainer learns from the corpus below (GPT-2 conventions: words
are pre-split, bytes are written in the printable alphabet, a merged token's id follows the merge's rank). Deterministic:
oracle/gen_golden.py and the tests build the same GGCC file from it."""
import re

SPECIAL = [b">>TITLE<<", b">>ABSTRACT<<", b">>INTRODUCTION<<", b">>SUMMARY<<", b">>COMMENT<<", b">>ANSWER<<", b">>QUESTION<<",
           b">>DOMAIN<<", b">>PREFIX<<", b">>SUFFIX<<", b">>MIDDLE<<", b"<|endoftext|>"]

CORPUS = """The quick brown fox jumps over the lazy dog. The dog didn't mind; it's a lazy dog, and they're friends.
Falcon models are decoder-only transformers: the attention uses multi-query heads and rotary positions.
In 2023 the 7B and 40B models were released; 1,000,000,000,000 tokens of RefinedWeb went into them.
We'll see what you've done, I'm sure he'd say: "that's all, folks!"   Tabs\tand  spaces   matter.
Die Größe der Bäume ändert sich über die Jahre; naïve café déjà vu. Ελληνικά γράμματα, кириллица, 日本語のテキスト, 한국어.
def tokenize(text):\n    return [t for t in text.split() if t]\n\n# numbers: 3.14159 2.71828 42 007 1e-9
the the the and and of of to to in in is is that that it it was was for for on on are are as as with with
"""


def byte_alphabet():
    keep = list(range(0x21, 0x7F)) + list(range(0xA1, 0xAD)) + list(range(0xAE, 0x100))
    enc, nxt = {}, 0x100
    for b in range(256):
        if b in keep:
            enc[b] = chr(b)
        else:
            enc[b] = chr(nxt); nxt += 1
    return enc


def build(n_merges=300):
    enc = byte_alphabet()
    dec = {v: k for k, v in enc.items()}
    words = re.findall(r"'s|'t|'re|'ve|'m|'ll|'d| ?[^\W\d_]+| ?\d+| ?[^\s\w]+|\s+(?!\S)|\s+", CORPUS)
    seqs = [[enc[b] for b in w.encode("utf-8")] for w in words]
    merges = []
    for _ in range(n_merges):
        counts = {}
        for s in seqs:
            for a, b in zip(s, s[1:]):
                counts[(a, b)] = counts.get((a, b), 0) + 1
        if not counts:
            break
        best = max(sorted(counts), key=lambda k: counts[k])           # deterministic tie-break: first in sorted order
        if counts[best] < 1:
            break
        merges.append(best)
        a, b = best
        for s in seqs:
            i = 0
            while i + 1 < len(s):
                if s[i] == a and s[i + 1] == b:
                    s[i:i + 2] = [a + b]
                else:
                    i += 1
    vocab = list(SPECIAL) + [bytes([b]) for b in range(256)]
    for a, b in merges:
        vocab.append(bytes(dec[ch] for ch in a + b))
    merges_b = [(a.encode("utf-8"), b.encode("utf-8")) for a, b in merges]
    return vocab, merges_b


TEXTS = [
    "Hello world", " Hello  world ", "The quick brown fox jumps over the lazy dog.", "it's what they're saying, isn't it? we'll see; I've won",
    "'round here 'twas ever thus, y'all", "a'xe b'yl c'q", "numbers 123 4567 3.14 1,000 2023年", "tabs\tand\nnewlines\n\n  and   spaces   ",
    "   leading and trailing   ", "x", " ", "  ", "\n", "ends with space ", "ends with punct!", "ends with digit 7", "!!!???...", "a.b,c;d",
    ">>TITLE<<Falcon>>ABSTRACT<< text<|endoftext|>", "no special > > here >>TITL", "<|endoftext|>", "<|endoftext|><|endoftext|> x",
    "Größe naïve café déjà vu", "Ελληνικά кириллица 日本語 한국어", "emoji 😀 and 👍🏽 mixed", "math: ∑ x² ≤ ½·π", "mixed١٢٣digits٤", "tab\there",
    "def f(x):\n    return x + 1  # comment", "UPPER lower MiXeD", "under_score-dash", "http://example.com/a?b=c&d=e", "quote \"double\" 'single'",
    "the the the and of to in is that it was for on are as with", "a" * 40, "ab " * 20, "\x01\x02 control\x7f", "nbsp\u00a0here\u2003em",
]
