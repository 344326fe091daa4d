#!/usr/bin/env python3
"""bench.py -- Falcon decode throughput on MI355X through libggml_hip.so (BASELINE.json metric).

  python bench.py --gpus 1 --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

Workload (config.workload): Falcon-7B Q4_0, full offload, 128-token prompt then greedy decode (BASELINE configs[1]).
A STEP is one decoded token = one pass of the whole hot path (32 blocks + ln_f + lm_head + greedy argmax) over a batch
of one token, everything resident in HBM. W untimed warm-up tokens, then exactly K timed tokens between
barrier + device synchronise; `value` = K / time (tokens/s); prefill tok/s of the 128-token prompt is reported beside it.

  roofline     dominant kernels = the two fused quantized mat-vec launches of a block (k_gemv_ln: LayerNorm + [Wqkv | Wup];
               k_attn_out: attention + [Wdown, Wo] + residual): achieved = algorithmic weight bytes per launch / average
               launch duration, both measured live with hipEvents (stamped by the dispatch itself) on every such launch of an instrumented repeat of
               the timed decode steps (same stream, plain launches); `step_*` = whole-token view B_tok * tok/s of the timed
               (hipGraph) region. profiles/ holds the rocprofv3 kernel-trace summary of the same command.
  cpu_baseline the same decode step on the host cores, bounded sample: oracle/_ref (the real reference, "reference")
               when its .so travelled, else the oracle port.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

# the host driver of these boxes only supports dmabuf IPC: without this, RCCL / cross-process device memory sharing fails
# with hipIpcGetMemHandle: invalid argument (must be set before the HIP runtime initialises)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured with a float4 copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=128)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--model", default="7b", choices=["7b", "40b", "tiny"])
    ap.add_argument("--quant", default="q4_0")
    ap.add_argument("--prompt", type=int, default=128)
    ap.add_argument("--n-ctx", type=int, default=2048)
    ap.add_argument("--layers", type=int, default=0, help="debug: truncate the model to this many blocks (marks the line invalid)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cpu-tokens", type=int, default=32, help="timed CPU decode steps after the CPU prompt (cpu_baseline leg)")
    ap.add_argument("--prefill-long", type=int, default=2048, help="also time one prompt of this many tokens (0 = skip)")
    ap.add_argument("--repeats", type=int, default=3, help="the K timed steps are run this many times (same positions); value = the median run")
    ap.add_argument("--force-pipeline", action="store_true", help="run the multi-GPU pipeline driver even with one GPU (testing)")
    ap.add_argument("--streams", type=int, default=2, help="groups of decode streams in flight for --force-pipeline at one GPU")
    ap.add_argument("--pipe-batch", type=int, default=128, help="pipeline: lock-step streams per group (one weight pass serves them; 1..256; 5..16 through the small-batch mat-muls, 17..32 in two passes of them, beyond through the int8-MFMA tile GEMM with a block's two branches on two streams -- round 6: 128 per pass is the best operating point of both bench models on one GPU (Falcon-7B Q4_0 16.5 k tok/s against 5.9 k at 16; Falcon-40B Q4_K 2.96 k against 1.6 k), and a stage's slot is long against the hand-off)")
    ap.add_argument("--no-north-star", action="store_true", help="skip the extra keys of the north-star configuration (Falcon-40B Q4_K, all 60 blocks, on the same GPUs)")
    ap.add_argument("--no-lock-step", action="store_true", help="skip the extra keys of the multi-stream (lock-step) decode measurement")
    ap.add_argument("--no-ref-order", action="store_true", help="skip the reference_order key (prompt + 16 decode steps in the reference's scalar summation order)")
    ap.add_argument("--no-cli", action="store_true", help="skip the reference_cli key (the reference's own falcon_main, linked against libggml_hip.so, on a written GGCC file)")
    ap.add_argument("--no-other-order", action="store_true", help="tuning runs: do not time the other summation order beside the timed one")
    ap.add_argument("--order", type=int, default=-1, choices=[-1, 0, 2],
                    help="summation order of the TIMED region: 2 = the fast reference order (ggml_hip_reference_order(2): logits bit-identical to the reference's scalar build; "
                         "legacy formats), 0 = the default order; -1 (default) = 2 where the format has it, else 0. The other order is timed beside it (key other_order)")
    return ap.parse_args()


def spawn_ranks(n, argv=None, script=None):
    """start n ranks of this script on this node (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT as torch.distributed.run
    sets them) and wait; rank 0 prints the JSON line. Returns the largest exit code."""
    import socket
    import subprocess
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, script or os.path.abspath(__file__), *(sys.argv[1:] if argv is None else argv)], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for p in procs:
        rc = max(rc, p.wait())
    if rc:
        raise SystemExit(rc)
    return rc


def kv_bytes_per_token(hp, n_past):
    # n_layer * 2 * (n_past+1) * n_head_kv * 64 * 4 read + n_layer * 2 * n_head_kv * 64 * 4 written
    return hp["n_layer"] * 2 * (n_past + 1) * hp["n_head_kv"] * 64 * 4 + hp["n_layer"] * 2 * hp["n_head_kv"] * 64 * 4


def host_topology():
    """{socket: [one logical cpu per physical core, in core order]} of this host (first hardware thread of every core)"""
    cores = {}
    try:
        cpu = pid = cid = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("processor"):
                cpu = int(line.split(":")[1])
            elif line.startswith("physical id"):
                pid = int(line.split(":")[1])
            elif line.startswith("core id"):
                cid = int(line.split(":")[1])
            elif not line.strip():
                if cpu is not None and pid is not None and cid is not None:
                    cores.setdefault(pid, {}).setdefault(cid, cpu)
                cpu = pid = cid = None
    except OSError:
        pass
    if not cores:
        n = max(1, (os.cpu_count() or 2) // 2)
        return {0: list(range(n))}
    return {p: [c[k] for k in sorted(c)] for p, c in cores.items()}


def host_cores():
    """(physical cores of ONE socket, sockets) of this host"""
    topo = host_topology()
    return max(1, len(topo[min(topo)])), len(topo)


class pinned:
    """run the body with this thread (and the worker threads it creates: they inherit the mask) restricted to `n` physical cores of
    socket 0, one logical cpu per core -- the reference creates its spin-barrier pool per graph compute (ggml.c:17251+), so its
    threads land on distinct cores of ONE socket instead of roaming over both (threads pinned)"""
    def __init__(self, n):
        topo = host_topology()
        avail = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else set()
        want = [c for c in topo[min(topo)] if not avail or c in avail]
        self.cpus = set(want[:max(1, n)]) if want else None
    def __enter__(self):
        self.old = os.sched_getaffinity(0) if self.cpus and hasattr(os, "sched_getaffinity") else None
        if self.old is not None:
            try:
                os.sched_setaffinity(0, self.cpus)
            except OSError:
                self.old = None
        return self
    def __exit__(self, *exc):
        if self.old is not None:
            os.sched_setaffinity(0, self.old)
        return False


def cpu_baseline(weights, hp, wbytes, prompt, n_tokens, tokens):
    """The same workload on the host cores: the prompt as one batch, then n_tokens timed decode steps -- at the
    reference's default -t 4 (examples/falcon_common.cpp:115-118) and at 8 / 16 / 32 / all physical cores of ONE socket, every run
    pinned to that many distinct cores of socket 0. `value` = the best of them (which one: `cores`). Runs the REAL reference
    (oracle/_ref/libggml_ref.so, its AVX2 build: what a user of the reference executes) when the .so travelled, else the oracle port."""
    from oracle import binding as ob
    cores, sockets = host_cores()
    ob.build_oracle()
    if ob.Ref.available():
        runner, kind = ob.Ref().model(weights, prompt + n_tokens + 8), "reference"
    else:
        runner, kind = ob.Oracle().model(weights, prompt + n_tokens + 8), "port"
    with pinned(cores):
        t0 = time.time()
        lg = runner.eval(tokens[:prompt], 0, cores)
        t_prompt = time.time() - t0
        cur = int(lg[-1].argmax())
        seq = []
        for i in range(3):                                               # 3 warm-up steps
            cur = int(runner.eval(np.array([cur], np.int32), prompt + i, cores)[0].argmax()); seq.append(cur)
    res = {}
    for threads in sorted({4, 8, 16, 32, cores} & set(range(1, cores + 1)) | {min(4, cores)}):
        with pinned(threads):
            t0 = time.time()
            c = seq[-1]
            for i in range(n_tokens):
                c = int(runner.eval(np.array([c], np.int32), prompt + 3 + i, threads)[0].argmax())
            res[threads] = n_tokens / (time.time() - t0)
    best = max(res, key=lambda k: res[k])
    t4 = res[min(4, cores)]
    return dict(value=res[best], unit="tokens/s", cores=best, kind=kind,
                effective_GBs=wbytes * res[best] / 1e9,
                t4_value=t4, t4_effective_GBs=wbytes * t4 / 1e9,
                by_threads={str(k): v for k, v in sorted(res.items())}, pinned="one logical cpu per physical core of socket 0 (sched_setaffinity before the reference creates its threads)",
                prefill_tok_s=prompt / t_prompt, sockets=sockets, socket_cores=cores,
                sample=f"{prompt}-token prompt as one batch ({cores} threads), 3 warm-up + {n_tokens} timed greedy decode steps of the same synthetic model per thread count "
                       f"{sorted(res)}; value = the best ({best} threads), t4_value = the reference's default -t 4")


def reference_cli(g, weights, hp, n_prompt, n_predict, n_ctx):
    """The reference's OWN entry point: examples/falcon/falcon_main.cpp, unchanged, linked against libggml_hip.so (csrc/falcon_wrap.cpp:
    its falcon_eval runs on the device) -- on this model written as a GGCC v10 file, default summation order, batch 128; the numbers are
    the ones its falcon_print_timings prints (libfalcon.cpp:4700-4714: batch eval = calls with more than one token, eval = one token)."""
    import re
    import shutil
    import subprocess
    import tempfile
    exe = os.path.join(ROOT, "oracle", "_ref", "falcon_main_hip")
    if not os.path.exists(exe):
        return None
    import bpe_fixture
    import ggcc_writer
    vocab, merges = bpe_fixture.build(n_merges=308)
    vocab = list(vocab) + [b"<|unused%d|>" % i for i in range(len(vocab), hp["n_vocab"])]      # (ids the byte-level BPE never produces; the model's logits cover them)
    td = tempfile.mkdtemp(prefix="falcon_cli_", dir=os.environ.get("TMPDIR", "/tmp"))
    try:
        path = os.path.join(td, "falcon_synth.ggcc")
        t0 = time.time()
        ggcc_writer.write_ggcc(path, weights, vocab, merges)
        t_write = time.time() - t0
        # a prompt of exactly n_prompt tokens: the longest prefix of the (repeated) fixture corpus that tokenizes to it
        voc = g.Vocab(path)
        text = (bpe_fixture.CORPUS * 8)
        lo, hi = 1, len(text)
        while lo < hi:
            mid = (lo + hi + 1) // 2
            if len(voc.tokenize(text[:mid])) <= n_prompt:
                lo = mid
            else:
                hi = mid - 1
        prompt = text[:lo]
        n_tok = len(voc.tokenize(prompt))
        voc.free() if hasattr(voc, "free") else None
        def run(order):
            env = dict(os.environ)
            env.pop("GGML_HIP_REFERENCE_ORDER", None)
            if order:
                env["GGML_HIP_REFERENCE_ORDER"] = str(order)
            t0 = time.time()
            r = subprocess.run([exe, "-m", path, "-p", prompt, "-n", str(n_predict), "--temp", "0", "-t", "4", "-c", str(n_ctx), "-b", "128", "--ignore-eos", "-s", "1"],
                               capture_output=True, env=env, timeout=600)
            wall = time.time() - t0
            err = r.stderr.decode("utf-8", "replace")
            if r.returncode != 0:
                return {"error": "falcon_main_hip exited %d" % r.returncode, "stderr_tail": err[-600:]}, None
            out = {"summation_order": order, "wall_s": wall, "on_device": "resident on the device" in err}
            m = re.search(r"batch eval time\s*=\s*([0-9.]+) ms /\s*(\d+) tokens \(\s*([0-9.]+) ms per token,\s*([0-9.]+) tokens per second", err)
            if m:
                out.update(batch_eval_ms=float(m.group(1)), batch_eval_tokens=int(m.group(2)), batch_eval_tok_s=float(m.group(4)))
            m = re.search(r"[^h] eval time\s*=\s*([0-9.]+) ms /\s*(\d+) runs\s*\(\s*([0-9.]+) ms per token,\s*([0-9.]+) tokens per second", err)
            if m:
                out.update(eval_ms=float(m.group(1)), eval_runs=int(m.group(2)), eval_ms_per_token=float(m.group(3)), eval_tok_s=float(m.group(4)))
            return out, r.stdout
        out, text0 = run(0)
        out["workload"] = (f"oracle/_ref/falcon_main_hip -m <this model as GGCC v10> -p <{n_tok} tokens> -n {n_predict} --temp 0 -t 4 -c {n_ctx} -b 128 --ignore-eos "
                           f"(the reference's falcon_main.cpp + libfalcon.cpp + ggml.c, unchanged, with falcon_eval on the device; default order)")
        out["prompt_tokens"] = n_tok
        out["ggcc_write_s"] = t_write
        # the same command with GGML_HIP_REFERENCE_ORDER=2: the reference's own CLI at the fast reference order's speed (its logits are then the CPU build's, bit for bit:
        # tests/test_gpu_dropin.py compares the bytes it prints with the pure-CPU build's on the small fixture)
        o2, text2 = run(2)
        if text0 is not None and text2 is not None:
            o2["same_text_as_default_order"] = text0 == text2
        out["fast_reference_order"] = o2
        return out
    finally:
        shutil.rmtree(td, ignore_errors=True)


def parity_sample(model, weights, toks, L, timed_order=0):
    """logits of two decode steps (n_past 0 and 1) of `model` on the device, in both summation orders, against the reference on the host -- its scalar build
    (what parity is pinned to) and its AVX2 build (what a user runs) -- and all four against the f64 YARDSTICK (oracle order 6: every reduction of the path
    accumulated in f64, same integer dots, quantizers and table look-ups): who is how far from the sums an exact evaluation would give.
    rel(a, b) = max |a - b| / rms(b) over the vocabulary, worst of the two steps."""
    from oracle import binding as ob
    ob.build_oracle()
    th = min(32, os.cpu_count() or 4)
    have_ref = ob.Ref.available(scalar=True)
    if have_ref:
        runner, kind = ob.Ref(scalar=True).model(weights, 8), "reference (scalar build)"
    else:
        runner, kind = ob.Oracle().model(weights, 8), "port (order 0 == the reference's scalar build, tests/test_oracle_vs_golden.py)"
    ref = [runner.eval(toks[i:i + 1], i, th)[0] for i in range(2)]
    simd = None
    if have_ref and ob.Ref.available():
        r2 = ob.Ref().model(weights, 8)
        simd = [r2.eval(toks[i:i + 1], i, th)[0] for i in range(2)]
    orc = ob.Oracle()
    orc.lib.orc_set_sum_order(6)
    try:
        mo = orc.model(weights, 8)
        f64 = [mo.eval(toks[i:i + 1], i, th)[0] for i in range(2)]
    finally:
        orc.lib.orc_set_sum_order(0)

    def rel(a, b):
        return max(float(np.abs(a[i].astype(np.float64) - b[i]).max() / np.sqrt((b[i].astype(np.float64) ** 2).mean())) for i in range(2))
    fast = [model.eval(toks[i:i + 1], i, logits_all=False)[0].copy() for i in range(2)]
    L.ggml_hip_reference_order(1)
    try:
        exact = [model.eval(toks[i:i + 1], i, logits_all=False)[0].copy() for i in range(2)]
    finally:
        L.ggml_hip_reference_order(0)
    timed = fast
    if timed_order:
        L.ggml_hip_reference_order(timed_order)
        try:
            timed = [model.eval(toks[i:i + 1], i, logits_all=False)[0].copy() for i in range(2)]
        finally:
            L.ggml_hip_reference_order(0)
    return dict(timed_order=timed_order, timed_order_vs_cpu=rel(timed, ref), timed_order_bit_identical=all(bool(np.array_equal(timed[i], ref[i])) for i in range(2)),
                default_order_vs_cpu=rel(fast, ref), reference_order_vs_cpu=rel(exact, ref),
                reference_avx2_vs_scalar_spread=(rel(simd, ref) if simd else None),
                default_order_vs_reference_avx2=(rel(fast, simd) if simd else None),
                err_vs_f64=dict(default_order=rel(fast, f64), reference_order=rel(exact, f64), reference_scalar_build=rel(ref, f64),
                                reference_avx2_build=(rel(simd, f64) if simd else None),
                                yardstick="oracle order 6 (orc_set_sum_order(6)): every reduction in f64 from exactly converted terms; integer dots, activation "
                                          "quantizers, fp16 tables and elementwise f32 steps are the reference's"),
                cpu=kind)


def pmc_decode_traffic(order=0):
    """(HBM bytes per launch of the decode's fused mat-vec launches IN SUMMATION ORDER `order`, source) from the NEWEST committed profiles/*pmc_traffic.json that
    holds those kernels (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this command, scripts/gpu_round.sh <tag> pmc, corrected by
    scripts/pmc_summary.py as MI355X_MICROARCH.md prescribes), or None. Files of other workloads (small-batch launches ...) are skipped."""
    pdir = os.path.join(ROOT, "profiles")
    if not os.path.isdir(pdir):
        return None

    def mine(k):
        # the launches of this order: default = k_gemv_ln_ring<T, NS, TWO, false> / k_gemv_ln<..> / k_attn_out<T>; fast reference order = the <.., true> ring form,
        # k_gemv_ln_ref, k_attn_out_ref
        ring_ref = False
        if k.startswith("k_gemv_ln_ring<"):                       # <TYPE, NSLOT, TWO, REF> since round 6 (three arguments before)
            targs = [t.strip() for t in k[k.index("<") + 1:k.rindex(">")].split(",")]
            ring_ref = len(targs) == 4 and targs[3] == "true"
        if order == 2:
            return k.startswith(("k_attn_out_ref", "k_gemv_ln_ref")) or ring_ref
        return k.startswith(("k_attn_out<", "k_gemv_ln<")) or (k.startswith("k_gemv_ln_ring") and not ring_ref)
    for f in sorted((f for f in os.listdir(pdir) if f.endswith("pmc_traffic.json")), reverse=True):
        try:
            d = json.load(open(os.path.join(pdir, f)))
        except (OSError, ValueError):
            continue
        ks = {k: v for k, v in d.items() if mine(k)}
        if not (any(k.startswith("k_gemv_ln") for k in ks) and any(k.startswith("k_attn_out") for k in ks)):      # (the merged attention + output launch: only the headline workload has it)
            continue
        n = sum(v["launches"] for v in ks.values())
        if n:
            return (sum(v["launches"] * v["hbm_bytes_per_launch"] for v in ks.values()) / n,
                    "profiles/" + f + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, 2 x FETCH_SIZE per the gfx950 note; bytes per launch over " + ", ".join(sorted(ks)) + ")")
    return None


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(a.gpus)                 # `python bench.py --gpus N` on its own: one process per GPU, as torch.distributed.run would start them
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(1, a.gpus):
        sys.stderr.write(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: the launcher must start exactly --gpus ranks\n")
        raise SystemExit(2)
    if a.gpus > 1 or world > 1 or a.force_pipeline:
        import bench_pipeline                      # layer-sharded multi-GPU path (RCCL hand-off)
        return bench_pipeline.main(a, rank, world, local)

    import ggllm_cpp_amd as g
    from ggllm_cpp_amd import synth

    tname = {v: k for k, v in g.TYPE_NAME.items()}
    wtype = tname[a.quant if a.quant in tname else a.quant.replace("_k", "_K")]
    hp = dict({"7b": synth.HP_7B, "40b": synth.HP_40B, "tiny": synth.HP_TINY_MQA}[a.model])
    if a.layers:
        hp["n_layer"] = a.layers
    if not os.path.exists(g.LIB_PATH):
        g.build()
    g.init(local)
    L = g.load()

    t0 = time.time()
    weights = synth.make_model_fast(hp, wtype, seed=1234)
    t_gen = time.time() - t0
    t0 = time.time()
    model = g.FalconModel(weights, n_ctx=max(a.n_ctx, a.prefill_long), n_batch=max(a.prompt, a.prefill_long, 1))
    t_up = time.time() - t0
    wbytes = model.weight_bytes()

    toks = synth.tokens(max(a.prompt, a.prefill_long) + 8, hp["n_vocab"], seed=42)
    legacy = a.quant in ("q4_0", "q4_1", "q5_0", "q5_1", "q8_0")
    kq_fast = a.quant.lower() in ("q2_k", "q3_k", "q4_k", "q5_k", "q6_k")         # k-quants with a wave-speed form of the reference's association (csrc/kernels_kqref.hip: ~0.5-0.7 x the default order's decode)
    # THE TIMED ORDER. 2 = the fast reference order (round 6, csrc/fq_ref_chain.h): the fused decode launches and the prefill GEMM add every row's per-block
    # terms left to right as the reference's scalar build does -- logits bit-identical to the CPU reference (north_star: within 1e-3; measured below on THIS
    # model: 0.0). The k-quants have no fast form of that association: their timed order is the default one (0).
    order = a.order if a.order >= 0 else (2 if legacy else 0)
    if order == 2 and not (legacy or kq_fast):
        sys.stderr.write("bench.py: --order 2 needs a format with a fast form of the reference's association (every format has one since round 6): timing the default order\n")
        order = 0
    other = 0 if order == 2 else (2 if (legacy or kq_fast) else None)
    if a.no_other_order:
        other = None
    # ---- parity, measured on the benchmark's own N(0, 0.02^2) model: the timed order, the default order, the one-thread-per-output instrument (mode 1) and
    # the reference's AVX2 build against the reference's scalar build on the host, all of them also against the f64 yardstick. (A residual-dominated variant
    # -- wo / down drawn 2^-4 times smaller, block scales still normal fp16 numbers -- is reported under its own key, never as the headline.)
    parity = None
    if not a.no_cpu:
        parity = {"benchmark_model": parity_sample(model, weights, toks, L, order)}
        if legacy:
            w2 = synth.make_model_fast(hp, wtype, seed=1234, out_gain=2.0 ** -4)
            m2 = g.FalconModel(w2, n_ctx=8, n_batch=1)
            parity["residual_dominated_model"] = parity_sample(m2, w2, toks, L, order)
            parity["residual_dominated_model"]["model"] = "same shapes and seeds, wo and down drawn 2^-4 times smaller (synth.make_model_fast(out_gain=2**-4)); not the headline"
            m2.free()
            del w2, m2

    e0, e1 = L.ggml_hip_event_create(), L.ggml_hip_event_create()
    # ---- the one-thread-per-output parity instrument (mode 1) timed on the same prompt and decode steps: what the reference's association cost before round 6
    ref_order = None
    if not a.no_ref_order:
        L.ggml_hip_reference_order(1)
        try:
            model.eval(toks[:a.prompt], 0, logits_all=False)
            L.ggml_hip_event_record(e0)
            lg_r = model.eval(toks[:a.prompt], 0, logits_all=False)
            L.ggml_hip_event_record(e1)
            r_prefill_ms = L.ggml_hip_event_elapsed_ms(e0, e1)
            cur = int(lg_r[0].argmax())
            K_r = min(a.steps, 8)
            cur = int(model.eval(np.array([cur], np.int32), a.prompt, logits_all=False)[0].argmax())       # warm
            L.ggml_hip_synchronize()
            t0 = time.perf_counter()
            for i in range(K_r):
                cur = int(model.eval(np.array([cur], np.int32), a.prompt + 1 + i, logits_all=False)[0].argmax())
            L.ggml_hip_synchronize()
            r_dt = time.perf_counter() - t0
            ref_order = {"mode": "ggml_hip_reference_order(1): every mat-mul one thread per output in the reference's scalar association (csrc/fq_ref_dot.h), f64 attention dots, "
                                 "op-by-op launches, one logits row D2H + host argmax per step -- the parity instrument the fast reference order (mode 2) is tested against",
                         "decode_tok_s": K_r / r_dt, "decode_ms_per_token": r_dt / K_r * 1e3, "decode_steps": K_r,
                         "prefill_tok_s": a.prompt / (r_prefill_ms * 1e-3), "prefill_ms": r_prefill_ms, "prompt": a.prompt}
        finally:
            L.ggml_hip_reference_order(0)

    use_graph = not a.no_graph

    def timed_run(mode, repeats, with_long, with_profile):
        """the whole measured workload in summation order `mode`: long prompt, the prompt (hipEvents), W warm-up + K timed decode steps (x repeats), and -- with_profile --
        the instrumented plain-launch repeat for the kernel-level roofline"""
        L.ggml_hip_reference_order(mode)
        try:
            r = {"order": mode}
            if with_long and a.prefill_long:
                model.eval(toks[:a.prefill_long], 0, logits_all=False)
                L.ggml_hip_event_record(e0)
                model.eval(toks[:a.prefill_long], 0, logits_all=False)
                L.ggml_hip_event_record(e1)
                r["long_ms"] = L.ggml_hip_event_elapsed_ms(e0, e1)
            model.eval(toks[:a.prompt], 0, logits_all=False)                   # warm (allocations, code load)
            L.ggml_hip_event_record(e0)
            lg = model.eval(toks[:a.prompt], 0, logits_all=False)
            L.ggml_hip_event_record(e1)
            r["prefill_ms"] = L.ggml_hip_event_elapsed_ms(e0, e1)
            first = int(lg[0].argmax())
            n_past = a.prompt
            # ---- warm-up decode steps (also captures the graph)
            out_w = model.decode_greedy(first, n_past, max(a.warmup, 1), use_graph=use_graph)
            n_past += max(a.warmup, 1)
            # ---- timed region: exactly K decode steps between device synchronisations; run `repeats` times over the SAME positions
            # (the KV entries are rewritten with the same values) and report the median run, all runs listed beside it
            runs = []
            for _ in range(max(1, repeats)):
                L.ggml_hip_synchronize()
                t0 = time.perf_counter()
                out = model.decode_greedy(int(out_w[-1]), n_past, a.steps, use_graph=use_graph)
                L.ggml_hip_synchronize()
                runs.append(time.perf_counter() - t0)
            r["runs"] = runs
            r["dt"] = sorted(runs)[len(runs) // 2]
            r["tok_s"] = a.steps / r["dt"]
            r["n_past"] = n_past
            r["tokens"] = [int(t) for t in out[:8]]
            if with_profile and hasattr(L, "ggml_hip_profile_begin"):
                import ctypes as C
                L.ggml_hip_profile_begin()
                model.decode_greedy(int(out[-1]), n_past + a.steps, min(a.steps, 32), use_graph=False)
                nl, us, by = C.c_int64(), C.c_double(), C.c_double()
                L.ggml_hip_profile_end(C.byref(nl), C.byref(us), C.byref(by))
                r["profile"] = (nl.value, us.value, by.value)
            return r
        finally:
            L.ggml_hip_reference_order(0)

    # the other order first (its numbers are reported beside the timed one), the timed order last. The long prompt runs in the default order only: the
    # prefill numbers that are priced against the MFMA roof are the default order's (int8 MFMA tiles with a K split, f32 attention on the matrix pipe);
    # the reference association's prompt (one left-to-right sum per row: ggml_hip_gemm_sequential, f64 attention dots) is reported next to it
    other_run = timed_run(other, 1, other == 0, True) if other is not None else None
    tr = timed_run(order, a.repeats, order == 0, True)
    runs, dt, tok_s, n_past = tr["runs"], tr["dt"], tr["tok_s"], tr["n_past"]
    run0 = tr if (order == 0 or other_run is None) else other_run                              # the default order's run: prefill_* keys
    run2 = tr if order == 2 else other_run                              # the fast reference order's run (None for the k-quants)
    long_ms, prefill_ms = run0.get("long_ms"), run0["prefill_ms"]
    n_mid = n_past + a.steps // 2
    b_tok = wbytes + kv_bytes_per_token(hp, n_mid)
    kname = ("k_gemv_ln_ring<.., REF> (lm_head: k_gemv_ln_ref) + k_attn_out_ref (the fused quantized mat-vec launches in the reference's association; lm_head included)" if order == 2 else
             "k_gemv_ln_ring (lm_head: k_gemv_ln) + k_attn_out (fused quantized mat-vec launches; lm_head included)")

    # ---- kernel-level roofline from the instrumented repeat (hipEvents stamped by every fused mat-vec dispatch)
    roof = dict(bound="hbm", achieved=None, peak=HBM_PEAK_GBS, unit="GB/s", frac=None, traffic=None)

    def launch_roof(prof):
        nl, us, by = prof
        if not nl:
            return {}
        avg_us = us / nl
        ach = (by / nl) / (avg_us * 1e-6) / 1e9
        return dict(launch_achieved=ach, launch_frac=ach / HBM_PEAK_GBS, launches=nl, avg_launch_us=avg_us, bytes_per_launch=by / nl)
    if tr.get("profile"):
        # the two events of a launch are handed to hipExtLaunchKernelGGL: the runtime stamps them with the dispatch's own
        # begin / end (the clock rocprofv3's kernel trace reads) -- no marker packets, nothing to subtract
        roof.update(launch_roof(tr["profile"]), kernel=kname, empty_event_pair_us=L.ggml_hip_profile_bracket_overhead_us())
    # HBM traffic per launch from the PMC counters: collected off-line (scripts/gpu_pmc.sh: one rocprofv3 --pmc pass per
    # counter over this same command, corrected by scripts/pmc_summary.py as MI355X_MICROARCH.md prescribes) and committed
    if a.model == "7b" and a.quant == "q4_0" and a.layers == 0:
        tr_pmc = pmc_decode_traffic(order)
        if tr_pmc:
            roof["traffic"], roof["traffic_source"] = tr_pmc
        else:                                                           # loud, but the line survives: the driver reads it
            roof["traffic_error"] = "no profiles/*pmc_traffic.json holds the decode launches (k_gemv_ln* + k_attn_out*): run scripts/gpu_round.sh <tag> pmc and commit its summary"
            sys.stderr.write("bench.py: ERROR: roofline.traffic is null -- " + roof["traffic_error"] + "\n")
    step_gbs = b_tok * tok_s / 1e9
    # `achieved` / `frac` = the TIMED (hipGraph) region: algorithmic bytes per token x tokens/s -- launch boundaries, attention and argmax included;
    # `launch_*` = the dominant kernels alone (algorithmic weight bytes per launch / average launch duration, events stamped by the dispatches of an
    # instrumented plain-launch repeat; agrees with profiles/*decode_7b_q4_0_kernel_stats.md)
    roof.update(achieved=step_gbs, frac=step_gbs / HBM_PEAK_GBS, step_achieved=step_gbs, step_frac=step_gbs / HBM_PEAK_GBS, bytes_per_token=b_tok,
                note="frac = whole timed step (B_tok x tok/s over 8 TB/s); launch_frac = the two fused mat-vec launches + lm_head alone")
    # what this chip streams at the same launch size: the pure-read micro-benchmark's rate (scripts/microbench/mb_stream.hip: a 58 MB launch back to back,
    # 10.0-10.5 us = 5.6-5.8 TB/s; NOTEBOOK section 4), so that the line states both the fraction of the spec sheet and of what the part can stream
    roof["measured_stream_GBs"] = 5700.0
    roof["frac_of_measured_stream"] = step_gbs / 5700.0
    roof["measured_stream_source"] = "scripts/microbench/mb_stream.hip on MI355X: pure 58 MB read launches back to back, 10.0-10.5 us (NOTEBOOK section 4); MI355X_MICROARCH.md float4 copy: 6.29 TB/s"

    # ---- prefill: flops / time against the matrix pipe (2 N sum(ne00 ne01) + attention 4 64 n_head n_layer N(N+1)/2)
    def prefill_roof(N, ms):
        E, H, HKV, FF, V, NL = hp["n_embd"], hp["n_head"], hp["n_head_kv"], hp["n_ff"], hp["n_vocab"], hp["n_layer"]
        per_tok = NL * (E * (H + 2 * HKV) * 64 + E * E + 2 * E * FF) + E * V
        fl = 2.0 * N * per_tok + 4.0 * 64 * H * NL * N * (N + 1) / 2
        tf = fl / (ms * 1e-3) / 1e12
        return dict(bound="mfma", tokens=N, achieved=tf, peak=2500.0, unit="TFLOP/s", frac=tf / 2500.0, frac_of_int8_peak=tf / 3944.0, tok_s=N / (ms * 1e-3), ms=ms,
                    note="peak = dense fp16/bf16 MFMA (SURVEY 8d); the GEMM runs on the int8 MFMA (>= 3944 TOPS dense) with exact per-32-group f32 scaling, "
                         "which bounds it by VALU issue; MFMA-busy counters: profiles/*pmc_mfma.json")
    prefill = {"prompt": prefill_roof(a.prompt, prefill_ms)}
    if long_ms:
        prefill["long"] = prefill_roof(a.prefill_long, long_ms)

    # ---- extra keys (not `value`): lock-step decode streams on the same weights -- B sequences advance together, ONE pass over the
    # weights serves B tokens (falcon_hip_pipeline at world 1: csrc/falcon_pipeline.hip; B <= 4 columns per mat-vec launch
    # (kernels_cols.hip, up to 8 in two chunks), 9..16 the small-batch streaming mat-mul, beyond that the int8-MFMA tile GEMM), same greedy sampler
    lock_step = None
    if not a.no_lock_step:
        G, R = 2, min(a.steps, 32)
        rows = {}
        for B in (8, 16, 64, 128, 256):
            pipe = g.Pipeline(model, 0, 1, G, B, min(a.n_ctx, 512))
            pipe.set_tokens(synth.tokens(G * B, hp["n_vocab"], seed=42))
            pipe.run(8, 0)
            L.ggml_hip_synchronize()
            t0 = time.perf_counter()
            pipe.run(R, 8)
            L.ggml_hip_synchronize()
            dtp = time.perf_counter() - t0
            pipe.history(8, R)
            pipe.free()
            ls_tok_s = R * G * B / dtp
            ls_bytes = wbytes / B + kv_bytes_per_token(hp, 8 + R // 2)
            rows[B] = {"tok_s": ls_tok_s, "ms_per_weight_pass": dtp / (R * G) * 1e3, "vs_single_stream": ls_tok_s / tok_s, "effective_GBs": ls_bytes * ls_tok_s / 1e9}
        best = max(rows, key=lambda b: rows[b]["tok_s"])
        lock_step = {"workload": f"{G} groups x B lock-step greedy decode streams on the same resident weights (one weight pass serves B tokens), positions 8..{8 + R}",
                     "value": rows[best]["tok_s"], "unit": "tokens/s", "streams_per_pass": best, "streams": G * best,
                     "ms_per_weight_pass": rows[best]["ms_per_weight_pass"], "vs_single_stream": rows[best]["vs_single_stream"],
                     "by_streams_per_pass": {str(b): rows[b] for b in rows}}

    cpu = None
    if not a.no_cpu:
        cpu = cpu_baseline(weights, hp, wbytes, a.prompt, a.cpu_tokens, toks)

    valid = (a.layers == 0)
    order_name = {0: "default order (per-lane partial sums + wave butterfly, K-split GEMM, f32 attention chains)",
                  2: "fast reference order (ggml_hip_reference_order(2): every row's block terms added left to right, f64 attention dots -- logits bit-identical to the reference's scalar build)"}

    def order_keys(r):
        """the numbers of one order's run, for the key that sits beside the timed one"""
        if r is None:
            return None
        k = {"order": r["order"], "order_name": order_name[r["order"]], "decode_tok_s": r["tok_s"], "ms_per_step": r["dt"] / a.steps * 1e3,
             "step_frac_of_hbm_peak": (wbytes + kv_bytes_per_token(hp, r["n_past"] + a.steps // 2)) * r["tok_s"] / 1e9 / HBM_PEAK_GBS,
             "prefill_ms": r["prefill_ms"], "prefill_tok_s": a.prompt / (r["prefill_ms"] * 1e-3), "first_tokens": r["tokens"]}
        if r.get("profile"):
            k.update(launch_roof(r["profile"]))
        return k
    pb = parity["benchmark_model"] if parity else None
    line = {
        "metric": "decode tokens/sec (+ prefill tok/s), Falcon-7B Q4_0 @1 GPU; % HBM roofline" if a.model == "7b" and a.quant == "q4_0"
                  else f"decode tokens/sec, Falcon-{a.model} {a.quant}",
        "value": tok_s, "unit": "tokens/s", "n_gpus": 1, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": dt / a.steps * 1e3, "repeat_ms_per_step": [r / a.steps * 1e3 for r in runs], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "int8", "data": "synthetic (random-init blocks, seeds 1234+i; tokens mt(42))",
        "config": {"workload": f"Falcon-{a.model.upper()} {a.quant.upper()} full offload, {a.prompt}-token prompt + greedy decode, n_ctx {a.n_ctx}"
                               + ("" if valid else f" [TRUNCATED to {a.layers} blocks: not the benchmark config]"),
                   "summation_order": order, "summation_order_name": order_name[order],
                   "n_past_timed": [n_past, n_past + a.steps], "hipgraph": use_graph, "weight_bytes_per_token": wbytes},
        # prefill: the default order's prompt (what the MFMA roofline prices); the reference association's prompt (the one in front of the timed decode steps
        # when summation_order is 2) beside it
        "prefill_tok_s": a.prompt / (prefill_ms * 1e-3), "prefill_ms": prefill_ms, "prefill_order": 0,
        "prefill_reference_order": ({"order": 2, "prefill_ms": run2["prefill_ms"], "prefill_tok_s": a.prompt / (run2["prefill_ms"] * 1e-3),
                                     "form": "int8-MFMA GEMM with ONE left-to-right sum per row (ggml_hip_gemm_sequential) + f64 attention dots: logits == the reference's scalar build"}
                                    if run2 else None),
        "prefill_roofline": prefill,
        "roofline": roof, "cpu_baseline": cpu,
        # north_star: logits within 1e-3 of the CPU reference -- measured on THE TIMED CONFIGURATION: the benchmark's own model, the timed summation order,
        # two decode steps against the reference's scalar build run on the host in this process (max |diff| / rms over the vocabulary)
        "max_rel_logit_err_vs_cpu": pb["timed_order_vs_cpu"] if pb else None,
        "max_rel_logit_err_mode": (f"the TIMED order ({order}: {order_name[order]}) on the benchmark's own N(0, 0.02^2) model against the reference's scalar build on the host "
                                   "(parity.benchmark_model.timed_order_vs_cpu; timed_order_bit_identical says whether every logit has the same bits). The default order's "
                                   "distance on the same model is parity.benchmark_model.default_order_vs_cpu (one flipped 8-bit activation rounding moves logits by 1e-2; "
                                   "the reference's own AVX2 and scalar builds differ by reference_avx2_vs_scalar_spread); err_vs_f64 places all of them against exact sums") if pb else None,
        "max_rel_logit_err_vs_cpu_default_order": pb["default_order_vs_cpu"] if pb else None,
        "max_rel_logit_err_vs_cpu_reference_order": max(v["reference_order_vs_cpu"] for v in parity.values()) if parity else None,    # ggml_hip_reference_order(1), the instrument: 0.0
        "parity": parity, "reference_order": ref_order,
        "other_order": order_keys(other_run),                           # the same prompt + decode steps in the other summation order, timed in this run
        "setup_s": {"synthesize": t_gen, "upload": t_up},
        "lock_step_streams": lock_step,
    }
    model.free()
    # ---- extra key: the same model through the reference's own CLI on top of the library
    if not a.no_cli and a.model == "7b" and valid:
        try:
            line["reference_cli"] = reference_cli(g, weights, hp, a.prompt, 128, a.n_ctx)
        except Exception as e:                                          # (never lose the bench line to the extra key)
            line["reference_cli"] = {"error": repr(e)}
    del weights
    # ---- extra keys: the north-star configuration on this GPU (Falcon-40B Q4_K, all 60 blocks resident, single-stream greedy decode)
    if not a.no_north_star and a.model == "7b" and a.quant == "q4_0" and valid:
        line["north_star_1gpu"] = north_star_1gpu(g, synth, L, tname, a)
    print(json.dumps(line))


def north_star_1gpu(g, synth, L, tname, a):
    hp = dict(synth.HP_40B)
    t0 = time.time()
    weights = synth.make_model_fast(hp, tname["q4_K"], seed=1234)
    model = g.FalconModel(weights, n_ctx=512, n_batch=128)
    del weights
    t_setup = time.time() - t0
    wbytes = model.weight_bytes()
    toks = synth.tokens(136, hp["n_vocab"], seed=42)
    e0, e1 = L.ggml_hip_event_create(), L.ggml_hip_event_create()
    model.eval(toks[:128], 0, logits_all=False)
    L.ggml_hip_event_record(e0)
    lg = model.eval(toks[:128], 0, logits_all=False)
    L.ggml_hip_event_record(e1)
    prefill_ms = L.ggml_hip_event_elapsed_ms(e0, e1)
    out_w = model.decode_greedy(int(lg[0].argmax()), 128, 4, use_graph=not a.no_graph)
    K = 32
    L.ggml_hip_synchronize()
    t0 = time.perf_counter()
    model.decode_greedy(int(out_w[-1]), 132, K, use_graph=not a.no_graph)
    L.ggml_hip_synchronize()
    dt = time.perf_counter() - t0
    # kernel-level roofline of the fused mat-vec launches (events stamped by the dispatches, as in the headline's `roofline`)
    kroof = None
    if hasattr(L, "ggml_hip_profile_begin"):
        import ctypes as C
        L.ggml_hip_profile_begin()
        model.decode_greedy(int(out_w[-1]), 132 + K, 8, use_graph=False)
        nl, us, by = C.c_int64(), C.c_double(), C.c_double()
        L.ggml_hip_profile_end(C.byref(nl), C.byref(us), C.byref(by))
        if nl.value:
            ach = (by.value / nl.value) / (us.value / nl.value * 1e-6) / 1e9
            kroof = dict(bound="hbm", achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s", frac=ach / HBM_PEAK_GBS, launches=nl.value, avg_launch_us=us.value / nl.value,
                         bytes_per_launch=by.value / nl.value,
                         kernel="the block's LayerNorm mat-vec launch [Wqkv | Wup] (ring form) + its output mat-vec launch [Wdown, Wo] + lm_head; algorithmic bytes = the matrices once; "
                                "rocprofv3 traces of the same launches: profiles/*decode_40b_q4_k_kernel_stats.md")
    # lock-step decode streams on the same resident model: one weight pass serves B tokens (B <= 4: the k-quant column mat-vec kernels;
    # 5..80: passes of the Q4_K small-batch form, 16 columns each; beyond: the int8-MFMA tile GEMM)
    ls = {}
    for B in (8, 16, 32, 64, 128):
        pipe = g.Pipeline(model, 0, 1, 1, B, 64)
        pipe.set_tokens(synth.tokens(B, hp["n_vocab"], seed=42))
        pipe.run(2, 0)
        L.ggml_hip_synchronize()
        t1 = time.perf_counter()
        pipe.run(6, 2)
        L.ggml_hip_synchronize()
        dl = time.perf_counter() - t1
        pipe.free()
        ls[str(B)] = {"tok_s": 6 * B / dl, "ms_per_weight_pass": dl / 6 * 1e3}
    # the same model in the reference's own association (round 6, csrc/kernels_kqref.hip: Q4_K's eight float lanes per super-block at wave speed; op-by-op launches):
    # decode steps after the same prompt, the first tokens against the one-thread-per-output instrument's (mode 1; bit-identity of the logits: tests/test_gpu_kqref.py)
    ref2 = None
    try:
        L.ggml_hip_reference_order(2)
        lg2 = model.eval(toks[:128], 0, logits_all=False)
        ow2 = model.decode_greedy(int(lg2[0].argmax()), 128, 2, use_graph=not a.no_graph)
        L.ggml_hip_synchronize()
        t2 = time.perf_counter()
        o2 = model.decode_greedy(int(ow2[-1]), 130, 16, use_graph=not a.no_graph)
        L.ggml_hip_synchronize()
        d2 = time.perf_counter() - t2
        L.ggml_hip_reference_order(1)
        cur = int(ow2[-1])
        t1 = time.perf_counter()
        o1 = []
        for i in range(2):
            cur = int(model.eval(np.array([cur], np.int32), 130 + i, logits_all=False)[0].argmax()); o1.append(cur)
        L.ggml_hip_synchronize()
        d1 = time.perf_counter() - t1
        ref2 = {"order": 2, "decode_tok_s": 16 / d2, "ms_per_step": d2 / 16 * 1e3, "mode1_decode_tok_s": 2 / d1,
                "first_tokens_equal_mode1": [int(t) for t in o2[:2]] == o1,
                "form": "ggml_hip_reference_order(2): single-token mat-vecs through k_gemv_kq_ref (the reference's scalar association: eight float lanes + the mins' chain per row), "
                        "f64 attention dots, op-by-op launches; the prompt column by column through the same kernel"}
    except Exception as e:                                              # (never lose the line to the extra key)
        ref2 = {"error": repr(e)}
    finally:
        L.ggml_hip_reference_order(0)
    model.free()
    tok_s = K / dt
    b_tok = wbytes + kv_bytes_per_token(hp, 132 + K // 2)
    return {"workload": "Falcon-40B Q4_K (60 blocks, GQA 128/8, 8192 wide) fully resident on ONE GPU, 128-token prompt + 32 greedy decode steps",
            "fast_reference_order": ref2,
            "value": tok_s, "unit": "tokens/s", "ms_per_step": dt / K * 1e3, "prefill_tok_s": 128 / (prefill_ms * 1e-3),
            "weight_bytes_per_token": wbytes, "step_achieved_GBs": b_tok * tok_s / 1e9, "step_frac": b_tok * tok_s / 1e9 / HBM_PEAK_GBS, "setup_s": t_setup,
            "roofline": kroof,
            "lock_step_by_streams_per_pass": ls}


if __name__ == "__main__":
    main()
