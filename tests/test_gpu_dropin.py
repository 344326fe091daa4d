"""GPU: the drop-in, end to end, with the REFERENCE'S OWN code on top of libggml_hip.so (artifacts of `make -C oracle
ref_falcon_hip`, built in the dev container from the reference's unchanged sources and shipped like oracle/_ref):

  libfalcon_ref_shim.so   the reference's loader, graph builder and graph executor (libfalcon.cpp, ggml.c; -DGGML_USE_CUBLAS);
                          weights offloaded through ggml_cuda_transform_tensor, every mat-mul through ggml_cuda_compute_forward
  libfalcon_ref_hip.so    the same plus csrc/falcon_wrap.cpp: falcon_eval runs device-resident (falcon_hip_eval)
  falcon_main_hip, falcon_perplexity_hip   the reference's command-line tools, unchanged, with the wrap

In reference order (GGML_HIP_REFERENCE_ORDER=1) all of them must reproduce what the pure-CPU reference produced -- logits of
tests/golden/ggcc_models.npz bit for bit, the bytes falcon_main / falcon_perplexity print (tests/golden/cli.npz)."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

import ggllm_cpp_amd as g
from oracle import binding as ob
import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFDIR = os.path.join(ROOT, "oracle", "_ref")


def _need(name):
    p = os.path.join(REFDIR, name)
    if not os.path.exists(p):
        pytest.skip("%s was not built (make -C oracle ref_falcon_hip in the dev container)" % name)
    return p


_LOGITS_SCRIPT = r"""
import ctypes as C, sys, numpy as np
so, path, nv, ngl, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
toks = np.load(sys.argv[6])
L = C.CDLL(so)
L.reff_load_ngl.restype = C.c_void_p; L.reff_load_ngl.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int]
L.reff_eval.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
L.reff_free.argtypes = [C.c_void_p]
ctx = L.reff_load_ngl(path.encode(), 64, 16, ngl)
assert ctx
pre = np.zeros((9, nv), np.float32)
assert L.reff_eval(ctx, toks[:9].ctypes.data, 9, 0, 2, pre.ctypes.data) == 0
dec = []
for i in range(9, 12):
    one = np.zeros((1, nv), np.float32)
    assert L.reff_eval(ctx, toks[i:i + 1].ctypes.data, 1, i, 2, one.ctypes.data) == 0
    dec.append(one)
H = C.CDLL(sys.argv[7])                                # libggml_hip.so: the handle the reference code is linked against
st = (C.c_size_t * 3)()
H.ggml_hip_shim_pool_stats(C.byref(st, 0), C.byref(st, C.sizeof(C.c_size_t)), C.byref(st, 2 * C.sizeof(C.c_size_t)))
np.savez(out, pre=pre, dec=np.concatenate(dec), pool=np.array(list(st), np.int64))
L.reff_free(ctx)
"""


# order: GGML_HIP_REFERENCE_ORDER = 1 (the one-thread-per-output parity instrument) or 2 (round 6, the FAST reference order: the fused decode launches and the
# sequential-sum GEMM in the reference's association, csrc/fq_ref_chain.h; formats without a fast form -- the k-quants -- run mode 1's kernels under it)
@pytest.mark.parametrize("order", ["1", "2"])
@pytest.mark.parametrize("so", ["libfalcon_ref_shim.so", "libfalcon_ref_hip.so"])
@pytest.mark.parametrize("name,hp,t", [("mqa_q4_0", synth.HP_TINY_MQA, ob.Q4_0), ("gqa_q5_1", synth.HP_TINY_GQA, ob.Q5_1),
                                       ("gqa_q4_K", synth.HP_TINY_GQA, ob.Q4_K), ("gqa_q6_K", synth.HP_TINY_GQA, ob.Q6_K)])
def test_reference_code_on_libggml_hip_reproduces_reference_logits(oracle, golden, tmp_path, so, name, hp, t, order):
    """the reference's falcon_init_from_file / falcon_eval / falcon_get_logits, with the offload (shim) or the device-resident
    evaluation (wrap) underneath: prefill and decode logits == the pure-CPU reference's (run in a child process: the
    reference's ggml_init starts its own CUDA-init thread and owns process-wide state)"""
    import ggcc_writer
    lib = _need(so)
    gg = golden["ggcc_models"]
    w = synth.make_model(oracle, hp, t, seed=4321)
    path = str(tmp_path / (name + ".ggcc"))
    ggcc_writer.write_ggcc(path, w)
    toks = str(tmp_path / "toks.npy")
    np.save(toks, gg[f"{name}_tokens"].astype(np.int32))
    out = str(tmp_path / "out.npz")
    script = str(tmp_path / "run.py")
    open(script, "w").write(_LOGITS_SCRIPT)
    env = dict(os.environ, GGML_HIP_REFERENCE_ORDER=order)
    r = subprocess.run([sys.executable, script, lib, path, str(hp["n_vocab"]), "100", out, toks, g.LIB_PATH], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    if so.endswith("_hip.so"):
        assert "resident on the device" in r.stderr
    res = np.load(out)
    assert np.array_equal(res["pre"], gg[f"{name}_prefill_logits"])
    assert np.array_equal(res["dec"], gg[f"{name}_decode_logits"])
    n_alloc, n_reuse, n_free = (int(v) for v in res["pool"])
    if so.endswith("_shim.so"):
        # the per-node staging buffers come from the pool (ggml-cuda.cu:1738-1816's role): 4 evals x offloaded mat-muls x 2
        # buffers were handed out, only a handful were ever allocated, and all are back in the pool
        assert n_alloc + n_reuse >= 4 * 4 * 2 and 0 < n_alloc <= 8 and n_free == n_alloc, (n_alloc, n_reuse, n_free)
    else:
        assert n_alloc == 0 and n_reuse == 0                  # nothing crosses the per-op boundary on the resident path


def _cli_model(oracle, path):
    import bpe_fixture
    import ggcc_writer
    vocab, merges = bpe_fixture.build(n_merges=308)
    hp = dict(synth.HP_TINY_MQA)
    hp["n_vocab"] = len(vocab)
    w = synth.make_model(oracle, hp, ob.Q4_0, seed=321)
    ggcc_writer.write_ggcc(path, w, vocab, merges)
    return bpe_fixture


@pytest.mark.parametrize("order", ["1", "2"])
def test_falcon_main_unchanged_on_the_fast_path(oracle, golden, tmp_path, order):
    """the reference's falcon_main (examples/falcon/falcon_main.cpp, unchanged: its argument parser, tokenizer, repetition
    penalty, greedy sampler, detokenizer) with falcon_eval on the device prints the same bytes as the pure-CPU build"""
    exe = _need("falcon_main_hip")
    path = str(tmp_path / "tiny_bpe.ggcc")
    _cli_model(oracle, path)
    env = dict(os.environ, GGML_HIP_REFERENCE_ORDER=order)
    r = subprocess.run([exe, "-m", path, "-p", "The quick brown fox didn't jump", "-n", "8", "--temp", "0", "-t", "2", "-c", "64", "-b", "8", "--ignore-eos", "-s", "1"],
                       capture_output=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert b"resident on the device" in r.stderr and b"device-resident path" in r.stderr
    assert r.stdout == bytes(golden["cli"]["main_stdout"])


def test_falcon_main_rope_context_follows_n_max_real_ctx(oracle, golden, tmp_path):
    """-c 4096 with a short prompt: falcon_main hands falcon_eval n_max_real_ctx = min(n_ctx, prompt + n_predict)
    (falcon_main.cpp:836), the reference ropes with THAT (libfalcon.cpp:2229-2230: NTK factor 1, not the 4096-context's 3);
    the wrap follows it per call (falcon_hip_context_set_rope_n_ctx) -- same bytes as the pure-CPU build"""
    exe = _need("falcon_main_hip")
    if "main_c4096_stdout" not in golden["cli"]:
        pytest.skip("fixture predates the -c 4096 run (python oracle/gen_golden.py cli)")
    path = str(tmp_path / "tiny_bpe.ggcc")
    _cli_model(oracle, path)
    env = dict(os.environ, GGML_HIP_REFERENCE_ORDER="1")
    r = subprocess.run([exe, "-m", path, "-p", "The quick brown fox didn't jump", "-n", "8", "--temp", "0", "-t", "2", "-c", "4096", "-b", "8", "--ignore-eos", "-s", "1"],
                       capture_output=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert r.stdout == bytes(golden["cli"]["main_c4096_stdout"])


def test_session_files_fail_loudly_on_the_fast_path(oracle, tmp_path):
    """--prompt-cache would restore / save the reference context's host KV cache, which the device path does not use: the wrap makes
    llama_load_session_file fail with a message instead of generating from an empty cache (falcon_main.cpp:425 then exits 1)"""
    exe = _need("falcon_main_hip")
    path = str(tmp_path / "tiny_bpe.ggcc")
    _cli_model(oracle, path)
    sess = str(tmp_path / "s.bin")
    open(sess, "wb").write(b"\0" * 64)                       # (any existing file: the wrap refuses before it is parsed)
    r = subprocess.run([exe, "-m", path, "-p", "The quick", "-n", "2", "--temp", "0", "-t", "2", "-c", "64", "-b", "8", "--prompt-cache", sess, "-s", "1"],
                       capture_output=True, timeout=600)
    assert r.returncode != 0
    assert b"llama_load_session_file is not supported for a context evaluated on the device" in r.stderr


def test_ngl_0_stays_on_the_host(oracle, golden, tmp_path):
    """BASELINE config 1's command: `falcon_main -ngl 0` means the reference's CPU path (libfalcon.cpp:1813-1826) -- the wrap creates no
    device side, says so, and the tool prints the CPU reference's bytes (default order: nothing of ours computes)"""
    exe = _need("falcon_main_hip")
    path = str(tmp_path / "tiny_bpe.ggcc")
    _cli_model(oracle, path)
    r = subprocess.run([exe, "-m", path, "-p", "The quick brown fox didn't jump", "-n", "8", "--temp", "0", "-t", "2", "-c", "64", "-b", "8", "--ignore-eos", "-s", "1", "-ngl", "0"],
                       capture_output=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert b"n_gpu_layers = 0" in r.stderr and b"stays on the host" in r.stderr
    assert b"resident on the device" not in r.stderr and b"device-resident path" not in r.stderr
    assert r.stdout == bytes(golden["cli"]["main_stdout"])


_LORA_SCRIPT = r"""
import ctypes as C, sys
so, path, ngl = sys.argv[1], sys.argv[2], int(sys.argv[3])
L = C.CDLL(so)
L.reff_load_ngl.restype = C.c_void_p; L.reff_load_ngl.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int]
L.reff_apply_lora.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
L.reff_free.argtypes = [C.c_void_p]
ctx = L.reff_load_ngl(path.encode(), 64, 16, ngl)
assert ctx
print("lora rc", L.reff_apply_lora(ctx, b"/nonexistent/adapter.bin", 1))
L.reff_free(ctx)
"""


def test_lora_fails_loudly_on_a_device_context(oracle, tmp_path):
    """llama_apply_lora_from_file (libfalcon.h:187-191) would patch the host tensors while the device copy keeps the unpatched weights:
    for a context with a device side the wrap refuses before the adapter file is even opened"""
    lib = _need("libfalcon_ref_hip.so")
    import ggcc_writer
    w = synth.make_model(oracle, synth.HP_TINY_MQA, ob.Q4_0, seed=4321)
    path = str(tmp_path / "m.ggcc")
    ggcc_writer.write_ggcc(path, w)
    script = str(tmp_path / "lora.py")
    open(script, "w").write(_LORA_SCRIPT)
    r = subprocess.run([sys.executable, script, lib, path, "100"], capture_output=True, text=True, timeout=600)
    if "undefined symbol: reff_apply_lora" in r.stderr:
        pytest.skip("oracle/_ref/libfalcon_ref_hip.so predates reff_apply_lora (make -C oracle ref_falcon_hip)")
    assert r.returncode == 0, r.stderr[-3000:]
    assert "lora rc 1" in r.stdout
    assert "llama_apply_lora_from_file is not supported for a context evaluated on the device" in r.stderr


@pytest.mark.parametrize("order", ["1", "2"])
def test_falcon_perplexity_unchanged_on_the_fast_path(oracle, golden, tmp_path, order):
    """the reference's falcon_perplexity tool, unchanged, with falcon_eval on the device: the chunk perplexities it prints"""
    exe = _need("falcon_perplexity_hip")
    path = str(tmp_path / "tiny_bpe.ggcc")
    bf = _cli_model(oracle, path)
    txt = str(tmp_path / "corpus.txt")
    open(txt, "wb").write((bf.CORPUS * 2).encode("utf-8"))
    env = dict(os.environ, GGML_HIP_REFERENCE_ORDER=order)
    r = subprocess.run([exe, "-m", path, "-f", txt, "-t", "2", "-c", "32", "-b", "8", "-s", "1"], capture_output=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert r.stdout == bytes(golden["cli"]["ppl_stdout"])


def test_debug_timings_prints_the_launch_table(oracle, golden, tmp_path):
    """--debug-timings 3 (every eval): where the reference prints its ggml graph's nodes (libfalcon.cpp:2506-2520 -> ggml_graph_print_impl) the resident path
    prints one line per launch site of the eval (falcon_hip_eval_debug_timings: event brackets around every fq_launch_*) -- and the generated bytes stay the
    CPU build's (the timed evals run as plain launches, same kernels)"""
    exe = _need("falcon_main_hip")
    path = str(tmp_path / "tiny_bpe.ggcc")
    _cli_model(oracle, path)
    env = dict(os.environ, GGML_HIP_REFERENCE_ORDER="1")
    r = subprocess.run([exe, "-m", path, "-p", "The quick brown fox didn't jump", "-n", "8", "--temp", "0", "-t", "2", "-c", "64", "-b", "8", "--ignore-eos", "-s", "1", "--debug-timings", "3"],
                       capture_output=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    err = r.stderr.decode("utf-8", "replace")
    assert err.count("falcon-hip: launches of this eval") >= 8 and "launch site" in err and "attention" in err and "mul_mat_ref" in err
    assert r.stdout == bytes(golden["cli"]["main_stdout"])


def test_eval_debug_timings_through_the_c_abi(oracle, capfd):
    """falcon_hip_eval_debug_timings == falcon_hip_eval (same logits) + the table on stderr, default order, a prompt batch and a single token"""
    g.init(0)
    hp = dict(synth.HP_TINY_GQA)
    w = synth.make_model(oracle, hp, ob.Q5_1, seed=9)
    toks = synth.tokens(40, hp["n_vocab"], seed=2)
    m = g.FalconModel(w, n_ctx=64, n_batch=40)
    want = [m.eval(toks[:36], 0, logits_all=False).copy(), m.eval(toks[36:37], 36, logits_all=False).copy()]
    L = g.load()
    got = []
    for lo, n, past in ((0, 36, 0), (36, 1, 36)):
        t = np.ascontiguousarray(toks[lo:lo + n], np.int32)
        assert L.falcon_hip_eval_debug_timings(m.ctx, t.ctypes.data, n, past, 0) == 0
        got.append(m.logits().reshape(1, -1).copy())
    m.free()
    err = capfd.readouterr().err
    assert err.count("falcon-hip: launches of this eval") == 2 and "gemm" in err and "attention" in err
    assert all(np.array_equal(a, b) for a, b in zip(got, want))
