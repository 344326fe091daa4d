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
    ap.add_argument("--cpu-tokens", type=int, default=6)
    ap.add_argument("--force-pipeline", action="store_true", help="run the multi-GPU pipeline driver even with one GPU (testing)")
    ap.add_argument("--streams", type=int, default=2, help="decode streams in flight for --force-pipeline at one GPU")
    return ap.parse_args()


def kv_bytes_per_token(hp, n_past):
    # n_layer * 2 * (n_past+1) * n_head_kv * 64 * 4 read + n_layer * 2 * n_head_kv * 64 * 4 written
    return hp["n_layer"] * 2 * (n_past + 1) * hp["n_head_kv"] * 64 * 4 + hp["n_layer"] * 2 * hp["n_head_kv"] * 64 * 4


def cpu_baseline(weights, hp, n_tokens, first_logits_gpu, tokens):
    """decode steps on the host: the real reference if its .so is here, else the oracle port"""
    from oracle import binding as ob
    cores = os.cpu_count() or 1
    try:
        import psutil
        cores = psutil.cpu_count(logical=False) or cores
    except Exception:
        pass
    threads = max(1, min(cores, 32))
    ob.build_oracle()
    if ob.Ref.available():
        runner, kind = ob.Ref().model(weights, 64), "reference"
    else:
        runner, kind = ob.Oracle().model(weights, 64), "port"
    lg0 = runner.eval(tokens[:1], 0, threads)              # warm-up + parity sample
    err = float(np.abs(lg0[0] - first_logits_gpu).max() / np.sqrt((lg0[0].astype(np.float64) ** 2).mean()))
    t0 = time.time()
    for i in range(1, 1 + n_tokens):
        runner.eval(tokens[i:i + 1], i, threads)
    dt = time.time() - t0
    return dict(value=n_tokens / dt, unit="tokens/s", cores=threads, kind=kind,
                sample=f"{n_tokens} decode steps (N=1, n_past 1..{n_tokens}) of the same synthetic model, {threads} threads"), err


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
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
    model = g.FalconModel(weights, n_ctx=a.n_ctx, n_batch=max(a.prompt, 1))
    t_up = time.time() - t0
    wbytes = model.weight_bytes()

    toks = synth.tokens(a.prompt + 8, hp["n_vocab"], seed=42)
    # ---- parity sample for the cpu leg: logits of the first token at n_past 0
    first_logits = model.eval(toks[:1], 0, logits_all=False)[0].copy()

    # ---- prefill of the prompt (timed with hipEvents, reported beside the decode number)
    e0, e1 = L.ggml_hip_event_create(), L.ggml_hip_event_create()
    model.eval(toks[:a.prompt], 0, logits_all=False)                   # warm (allocations, code load)
    L.ggml_hip_event_record(e0)
    lg = model.eval(toks[:a.prompt], 0, logits_all=False)
    L.ggml_hip_event_record(e1)
    prefill_ms = L.ggml_hip_event_elapsed_ms(e0, e1)
    first = int(lg[0].argmax())

    use_graph = not a.no_graph
    n_past = a.prompt
    # ---- warm-up decode steps (also captures the graph)
    out_w = model.decode_greedy(first, n_past, max(a.warmup, 1), use_graph=use_graph)
    n_past += max(a.warmup, 1)
    # ---- timed region: exactly K decode steps
    L.ggml_hip_synchronize()
    t0 = time.perf_counter()
    out = model.decode_greedy(int(out_w[-1]), n_past, a.steps, use_graph=use_graph)
    L.ggml_hip_synchronize()
    dt = time.perf_counter() - t0
    tok_s = a.steps / dt
    n_mid = n_past + a.steps // 2
    b_tok = wbytes + kv_bytes_per_token(hp, n_mid)

    # ---- instrumented repeat for the kernel-level roofline (hipEvents around every GEMV launch)
    roof = dict(bound="hbm", achieved=None, peak=HBM_PEAK_GBS, unit="GB/s", frac=None, traffic=None)
    if hasattr(L, "ggml_hip_profile_begin"):
        import ctypes as C
        L.ggml_hip_profile_begin()
        model.decode_greedy(int(out[-1]), n_past + a.steps, min(a.steps, 32), use_graph=False)
        nl, us, by = C.c_int64(), C.c_double(), C.c_double()
        L.ggml_hip_profile_end(C.byref(nl), C.byref(us), C.byref(by))
        if nl.value:
            # the two events of a launch are handed to hipExtLaunchKernelGGL: the runtime stamps them with the dispatch's own
            # begin / end (the clock rocprofv3's kernel trace reads) -- no marker packets, nothing to subtract
            ovh = L.ggml_hip_profile_bracket_overhead_us()
            avg_us = us.value / nl.value
            ach = (by.value / nl.value) / (avg_us * 1e-6) / 1e9
            roof.update(achieved=ach, frac=ach / HBM_PEAK_GBS, kernel="k_gemv_ln + k_attn_out (fused quantized mat-vec launches; lm_head included)",
                        launches=nl.value, avg_launch_us=avg_us, empty_event_pair_us=ovh, bytes_per_launch=by.value / nl.value)
    # HBM traffic per launch from the PMC counters: collected off-line (scripts/gpu_pmc.sh: one rocprofv3 --pmc pass per
    # counter over this same command, corrected by scripts/pmc_summary.py as MI355X_MICROARCH.md prescribes) and committed
    pmc = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("pmc_traffic.json")) if os.path.isdir(os.path.join(ROOT, "profiles")) else []
    if pmc and a.model == "7b" and a.quant == "q4_0" and a.layers == 0:
        d = json.load(open(os.path.join(ROOT, "profiles", pmc[-1])))
        ks = [v for k, v in d.items() if k.startswith("k_gemv_ln") or k.startswith("k_attn_out") or k.startswith("k_gemv_out")]
        if ks:
            roof["traffic"] = sum(v["launches"] * v["hbm_bytes_per_launch"] for v in ks) / sum(v["launches"] for v in ks)
            roof["traffic_source"] = "profiles/" + pmc[-1] + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, 2 x FETCH_SIZE per the gfx950 note; bytes per launch, same launch mix)"
    step_gbs = b_tok * tok_s / 1e9
    roof.update(step_achieved=step_gbs, step_frac=step_gbs / HBM_PEAK_GBS, bytes_per_token=b_tok)

    cpu = None
    err = None
    if not a.no_cpu:
        cpu, err = cpu_baseline(weights, hp, a.cpu_tokens, first_logits, toks)

    valid = (a.layers == 0)
    line = {
        "metric": "decode tokens/sec (+ prefill tok/s), Falcon-7B Q4_0 @1 GPU; % HBM roofline" if a.model == "7b" and a.quant == "q4_0"
                  else f"decode tokens/sec, Falcon-{a.model} {a.quant}",
        "value": tok_s, "unit": "tokens/s", "n_gpus": 1, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "int8", "data": "synthetic (random-init blocks, seeds 1234+i; tokens mt(42))",
        "config": {"workload": f"Falcon-{a.model.upper()} {a.quant.upper()} full offload, {a.prompt}-token prompt + greedy decode, n_ctx {a.n_ctx}"
                               + ("" if valid else f" [TRUNCATED to {a.layers} blocks: not the benchmark config]"),
                   "n_past_timed": [n_past, n_past + a.steps], "hipgraph": use_graph, "weight_bytes_per_token": wbytes},
        "prefill_tok_s": a.prompt / (prefill_ms * 1e-3), "prefill_ms": prefill_ms,
        "roofline": roof, "cpu_baseline": cpu, "max_rel_logit_err_vs_cpu": err,
        "setup_s": {"synthesize": t_gen, "upload": t_up},
    }
    print(json.dumps(line))
    model.free()


if __name__ == "__main__":
    main()
