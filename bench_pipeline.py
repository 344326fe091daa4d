#!/usr/bin/env python3
"""Layer-pipelined multi-GPU decode: one process per GPU, rank r owns a contiguous range of Falcon blocks,
the ONLY exchange is the residual row [n_embd] f32 between neighbouring stages (RCCL send/recv over xGMI, one link per
hop) plus the 4-byte greedy-sampled token from the last stage back to stage 0. No all-reduce, no all-gather.

A single decode stream is serial through the stages (capacity scaling only), so S = 2 x world independent decode streams
are kept in flight, each with its own KV cache on every stage; a ROUND advances every stream by one token:

    stage 0     :  token_s (from the last stage)  -> embed + blocks                          -> hidden_s to stage 1
    stage r     :  hidden_s from stage r-1        -> blocks                                  -> hidden_s to stage r+1
    last stage  :  hidden_s                       -> blocks + ln_f + lm_head + greedy argmax -> token_s to stage 0
(slot schedule and deadlock-freedom: see PipelineRunner)

Everything is enqueued asynchronously on the library's HIP stream (falcon_hip_stage_step has no host sync; the RCCL calls
are torch.distributed P2P ops issued with that stream current), so stage r works on stream s+1 while stage r+1 works on s.

bench.py --gpus N (N > 1) lands here: W warm-up rounds, K timed rounds, value = K * S / time (tokens/s over all GPUs).
The host logic (partition, schedule, token ring) is backend-agnostic and is tested on CPU with gloo and a mock stage
(tests/test_pipeline_gloo.py).
"""
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


def partition(n_layer, world, head_units=0.0):
    """contiguous block ranges [(begin, end)] per rank, balanced by the bytes a stage streams per token: a block is one
    unit, the last stage also owns ln_f + lm_head = head_units blocks' worth of weights (Falcon-7B Q4_0: 166 MB / 122 MB =
    1.4; 40B: 0.8). Minimises the slowest stage -- the pipeline's throughput is its reciprocal. head_units = 0: the plain
    split (40B: 60 -> 15/15/15/15 or 8,8,8,8,7,7,7,7)."""
    if world == 1:
        return [(0, n_layer)]
    if head_units <= 0:
        base, extra = divmod(n_layer, world)
        out, b = [], 0
        for r in range(world):
            e = b + base + (1 if r < extra else 0)
            out.append((b, e))
            b = e
        return out
    best = None
    for last in range(1, n_layer - (world - 1) + 1):            # blocks of the last stage; the others split the rest evenly
        rest = n_layer - last
        if rest < world - 1:
            continue
        front = -(-rest // (world - 1))                          # ceil
        cost = max(front, last + head_units)
        if best is None or cost < best[0] - 1e-9 or (abs(cost - best[0]) <= 1e-9 and last > best[1]):
            best = (cost, last)
    last = best[1]
    base, extra = divmod(n_layer - last, world - 1)
    counts = [base + (1 if r < extra else 0) for r in range(world - 1)] + [last]
    out, b = [], 0
    for c in counts:
        out.append((b, b + c))
        b += c
    return out


def head_units(hp, wtype_bits_per_weight=None):
    """lm_head's weight bytes in units of one block's weight bytes (same format for both: the ratio is format-free)"""
    E, FF, V = hp["n_embd"], hp["n_ff"], hp["n_vocab"]
    qkv = (hp["n_head"] + 2 * hp["n_head_kv"]) * 64
    per_block = E * qkv + E * E + 2 * E * FF
    return (V * E) / per_block


class PipelineRunner:
    """Backend-agnostic schedule. `engine.step(stream, n_past)` runs this stage for one stream on pre-bound buffers;
    `comm.exchange(send, recv)` performs one grouped P2P exchange. Both may be asynchronous w.r.t. the host.

    The ranks form a ring (stage P-1 returns the sampled token to stage 0). Work item w = k*S + s is stream s in round
    k; rank r computes item t - r in slot t. At the START of slot t every rank does ONE grouped exchange
        send  the result of its slot t-1 (item t-1-r) to the next rank      (last rank: the token, to rank 0)
        recv  the input of item t-r from the previous rank                  (rank 0: the token of item t-P, into the
                                                                             token buffer of that item's stream)
    then computes. Send and receive are posted together (ncclGroup / async gloo ops), so the ring cannot deadlock, and
    in steady state all P stages compute concurrently on P different streams. S >= P keeps stage 0 from waiting for
    a token that is still in flight. With S >= 2P streams (what bench.py runs) the schedule is stretched to one extra slot
    per hop so that every transfer overlaps a compute slot instead of preceding it (_run_overlapped).
    """

    def __init__(self, rank, world, n_streams, engine, comm):
        self.rank, self.world, self.S, self.engine, self.comm = rank, world, n_streams, engine, comm
        self.first, self.last = rank == 0, rank == world - 1
        assert world == 1 or n_streams >= world, "need at least one decode stream per stage"

    def run(self, rounds, n_past0):
        """advance every stream by `rounds` tokens starting at position n_past0 (initial tokens are in engine.tok_in)"""
        P, S, r = self.world, self.S, self.rank
        W = rounds * S
        if P == 1:
            for w in range(W):
                self.engine.step(w % S, n_past0 + w // S)
                self.engine.feed_back_token(w % S)
            return
        if S >= 2 * P and hasattr(self.comm, "post"):
            return self._run_overlapped(W, n_past0)
        for t in range(W + P):
            sends, recvs = [], []
            w_prev = t - 1 - r                              # item computed in the previous slot
            if 0 <= w_prev < W:
                sends.append(("token", w_prev % S, 0) if self.last else ("hidden", w_prev % S, r + 1))
            w = t - r                                       # item to compute now
            if self.first:
                if 0 <= t - P < W:
                    recvs.append(("token", (t - P) % S, P - 1))
            elif 0 <= w < W:
                recvs.append(("hidden", w % S, r - 1))
            if sends or recvs:
                self.comm.exchange(sends, recvs)
            if 0 <= w < W:
                self.engine.step(w % S, n_past0 + w // S)

    def _run_overlapped(self, W, n_past0):
        """S >= 2P streams: a hop gets a whole slot. Rank r computes item t - 2r in slot t; the exchange posted at the start
        of slot t carries the result of slot t-1 to the next rank and brings in the input of slot t+1, and is only waited
        for at the start of slot t+1 -- the transfer overlaps the compute of slot t. (The last rank's token of item x
        travels in the exchange of slot x + 2P - 1 and is needed by rank 0 for item x + S in slot x + S >= x + 2P.)"""
        P, S, r = self.world, self.S, self.rank
        pending = None
        for t in range(W + 2 * P):
            sends, recvs = [], []
            w_prev = t - 1 - 2 * r
            if 0 <= w_prev < W:
                sends.append(("token", w_prev % S, 0) if self.last else ("hidden", w_prev % S, r + 1))
            if self.first:
                x = t - 2 * P + 1                           # the item whose token the last rank sends in this exchange
                if 0 <= x < W:
                    recvs.append(("token", x % S, P - 1))
            else:
                w_next = t + 1 - 2 * r
                if 0 <= w_next < W:
                    recvs.append(("hidden", w_next % S, r - 1))
            posted = self.comm.post(sends, recvs) if (sends or recvs) else None
            if pending is not None:
                self.comm.finish(pending)
            pending = posted
            w = t - 2 * r
            if 0 <= w < W:
                self.engine.step(w % S, n_past0 + w // S)
        if pending is not None:
            self.comm.finish(pending)


# ------------------------------------------------------------------------------------------------ GPU backend
class TorchComm:
    """torch.distributed P2P (backend nccl = RCCL on ROCm, or gloo on CPU) on the engine's buffers; one grouped
    batch_isend_irecv per slot"""

    def __init__(self, dist, engine):
        self.dist, self.e = dist, engine

    def post(self, sends, recvs):
        ops = []
        for kind, s, peer in sends:
            ops.append(self.dist.P2POp(self.dist.isend, self.e.tok_out[s] if kind == "token" else self.e.hidden_out[s], peer))
        for kind, s, peer in recvs:
            ops.append(self.dist.P2POp(self.dist.irecv, self.e.tok_in[s] if kind == "token" else self.e.hidden_in[s], peer))
        return self.dist.batch_isend_irecv(ops)

    def finish(self, works):
        for w in works:                  # nccl: the current stream waits (not the host); gloo: the host waits
            w.wait()

    def exchange(self, sends, recvs):
        self.finish(self.post(sends, recvs))


def _watchdog(seconds, what):
    """the RCCL path cannot be exercised where this is developed (one GPU per box): if `what` has not finished after
    `seconds`, say so and leave instead of hanging the job"""
    import threading
    done = threading.Event()

    def run():
        if not done.wait(seconds):
            sys.stderr.write(f"bench_pipeline: {what} did not finish within {seconds} s -- giving up\n")
            sys.stderr.flush()
            os._exit(3)
    threading.Thread(target=run, daemon=True).start()
    return done


_PREFLIGHT = r"""
import sys
sys.path.insert(0, %r)
import ggllm_cpp_amd as g
rank, world, dev = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[4])
sys.exit(g.load().falcon_hip_rccl_selftest(rank, world, bytes.fromhex(sys.argv[3]), dev))
"""


def rccl_preflight(g, dist, torch, rank, world, local, timeout_s=None):
    """RCCL's send / recv between THESE ranks, tried in a child process per rank under a time-out (falcon_hip_rccl_selftest: a ring of one small message over a
    fresh communicator) before anything is built on it: on the boxes this library is developed on RCCL refuses (two ranks on one device), and it has never run
    between two GPUs there -- a refusal or a hang must cost the job a fall-back to the host-staged transport, not its life. Returns (ok on EVERY rank, reason)."""
    import subprocess
    timeout_s = float(os.environ.get("FALCON_PIPE_PREFLIGHT_S", "120")) if timeout_s is None else timeout_s
    if timeout_s <= 0:
        return True, "skipped"
    box = [None]
    if rank == 0:
        try:
            box[0] = g.Pipeline.unique_id().hex()
        except Exception as e:                                      # noqa: BLE001
            box[0] = "!" + repr(e)
    dist.broadcast_object_list(box, src=0)
    if box[0].startswith("!"):
        return False, "no RCCL unique id: " + box[0][1:]
    why, ok = "", 0
    try:
        p = subprocess.Popen([sys.executable, "-c", _PREFLIGHT % ROOT, str(rank), str(world), box[0], str(local)], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
        try:
            _, err = p.communicate(timeout=timeout_s)
            ok = 1 if p.returncode == 0 else 0
            if not ok:
                why = "rank %d: exit %d: %s" % (rank, p.returncode, err.decode("utf-8", "replace").strip().splitlines()[-1][-160:] if err.strip() else "")
        except subprocess.TimeoutExpired:
            p.kill()
            p.communicate()
            why = "rank %d: no answer within %.0f s" % (rank, timeout_s)
    except Exception as e:                                          # noqa: BLE001
        why = "rank %d: %r" % (rank, e)
    t = torch.tensor([ok], dtype=torch.int32)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    reasons = [None] * world
    dist.all_gather_object(reasons, why)
    return bool(int(t.item())), "; ".join(r for r in reasons if r) or "a peer failed"


def run_cpp(a, rank, world, local, hp, wtype, model_name, quant_name, dist, torch, groups, batch, n_ctx, steps, warmup):
    """one timed run of the C++ pipeline (csrc/falcon_pipeline.hip): returns (tokens/s over the job, weight bytes over all
    ranks, blocks per stage, setup seconds) on every rank"""
    import ggllm_cpp_amd as g
    from ggllm_cpp_amd import synth
    L = g.load()
    parts = partition(hp["n_layer"], world, head_units(hp))
    lb, le = parts[rank]
    t0 = time.time()
    weights = synth.make_model_fast(hp, wtype, seed=1234, layers=range(lb, le))
    model = g.FalconModel(weights, n_ctx=8, n_batch=1, layer_begin=lb, layer_end=le)
    del weights
    def make_pipe():
        uid = None
        if world > 1:
            box = [g.Pipeline.unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)               # 128 bytes over the launcher's (gloo) store
            uid = box[0]
        return g.Pipeline(model, rank, world, groups, batch, n_ctx, unique_id=uid)
    try:
        pipe = make_pipe()
    except RuntimeError:
        # the device-to-device fall-back (ipc) gives itself up on EVERY rank together when one rank cannot export / open a mailbox (csrc/falcon_pipeline.hip):
        # the job goes on with the host-staged form, under a fresh id
        if world > 1 and os.environ.get("FALCON_PIPE_TRANSPORT") == "ipc":
            os.environ["FALCON_PIPE_TRANSPORT"] = "shm"
            run_cpp.ipc_refused = True
            if rank == 0:
                sys.stderr.write("bench_pipeline: the device-to-device (ipc) transport is not available between these ranks: host-staged transport (shm)\n")
            pipe = make_pipe()
        else:
            raise
    rccl_ranks = int(L.falcon_hip_pipeline_rccl_ranks(pipe.p))      # ncclCommCount of the communicator the hand-offs use
    run_cpp.transport = pipe.transport()
    t_setup = time.time() - t0
    pipe.set_tokens(synth.tokens(groups * batch, hp["n_vocab"], seed=42))
    done = _watchdog(600, f"warm-up of the {world}-rank pipeline (RCCL channel set-up)")
    wr = max(warmup, 1)
    pipe.run(wr, 0)                                              # warm-up rounds (also build the RCCL P2P channels and the stage graphs)
    L.ggml_hip_synchronize()
    done.set()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pipe.run(steps, wr)                                          # K rounds = K * groups * batch tokens, including pipeline fill and drain
    L.ggml_hip_synchronize()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    hist = pipe.history(wr, steps)                               # (last rank) also surfaces hand-off time-outs
    if hist is not None and (hist < 0).any():
        raise RuntimeError("pipeline produced invalid tokens")
    dump = os.environ.get("FALCON_PIPE_DUMP_HISTORY")            # tests: the last rank's sampled tokens [round][sequence] of the timed rounds
    if dump and hist is not None and world > 1:
        np.save(dump, hist)
    dt_t = torch.tensor([dt], dtype=torch.float64)
    wb_t = torch.tensor([float(model.weight_bytes())], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(dt_t, op=dist.ReduceOp.MAX)
        dist.all_reduce(wb_t, op=dist.ReduceOp.SUM)
    pipe.free()
    model.free()
    run_cpp.rccl_ranks = rccl_ranks
    return steps * groups * batch / float(dt_t.item()), float(wb_t.item()), [e - b for b, e in parts], t_setup, float(dt_t.item())


def main(a, rank, world, local):
    import torch
    import torch.distributed as dist
    import ggllm_cpp_amd as g
    from ggllm_cpp_amd import synth

    tname = {v: k for k, v in g.TYPE_NAME.items()}
    wtype = tname[a.quant if a.quant in tname else a.quant.replace("_k", "_K")]
    hp = dict({"7b": synth.HP_7B, "40b": synth.HP_40B, "tiny": synth.HP_TINY_MQA}[a.model])
    if a.layers:
        hp["n_layer"] = a.layers
    if not os.path.exists(g.LIB_PATH):
        g.build()
    if os.environ.get("FALCON_PIPE_SAME_DEVICE") == "1":         # debug: every rank on GPU 0 (RCCL normally refuses two ranks on one device)
        local = 0
    torch.cuda.set_device(local)
    g.init(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)      # host-side control only (id hand-out, barrier, max over ranks); the data path is RCCL inside libggml_hip.so
    transport_note = None
    forced = os.environ.get("FALCON_PIPE_TRANSPORT")              # shm / ipc forced by the caller (the one-GPU boxes' tests): not a fall-back
    fallback = False
    if world > 1 and forced not in ("shm", "ipc"):
        ok, why = rccl_preflight(g, dist, torch, rank, world, local)
        if not ok:
            # LOUD: a multi-GPU number that never touched RCCL must not look like one that did. First choice: device-to-device copies into the peer's
            # IPC-exported mailboxes (over xGMI between two GPUs); if a rank cannot export / open them, host shared memory.
            fallback = True
            os.environ["FALCON_PIPE_TRANSPORT"] = "ipc"
            transport_note = ("RCCL pre-flight FAILED (" + why + "): the hand-offs do NOT run over RCCL in this job -- FALL-BACK to device-to-device copies into the "
                              "peer's IPC-exported mailboxes (FALCON_PIPE_TRANSPORT=ipc; host shared memory if a rank cannot open them), same ranks, schedule and stage code")
            if rank == 0:
                sys.stderr.write("bench_pipeline: WARNING: " + transport_note + "\n")
    batch = max(1, min(int(getattr(a, "pipe_batch", 4)), 256))
    groups = max(2 * world, 2) if world > 1 else max(1, getattr(a, "streams", 2))
    n_ctx = min(a.n_ctx, 512)
    if a.warmup + a.steps + 1 > n_ctx:
        raise SystemExit(f"--warmup + --steps must stay below {n_ctx} positions")
    tok_s, wbytes, blocks, t_setup, dt = run_cpp(a, rank, world, local, hp, wtype, a.model, a.quant, dist, torch, groups, batch, n_ctx, a.steps, a.warmup)
    rccl_ranks = getattr(run_cpp, "rccl_ranks", None)
    transport = getattr(run_cpp, "transport", None)
    env_tr = os.environ.get("FALCON_PIPE_TRANSPORT")
    shm = env_tr in ("shm", "ipc")                               # ranks on one node exchanging through mailboxes (host shared memory, or the peer's device memory)
    if world > 1 and not shm and rccl_ranks != world:
        raise SystemExit(f"bench_pipeline: RCCL communicator has {rccl_ranks} ranks, launched {world}")
    if world > 1 and shm and not str(transport).startswith(env_tr):
        raise SystemExit(f"bench_pipeline: FALCON_PIPE_TRANSPORT={env_tr} but the pipeline reports transport {transport!r}")
    if getattr(run_cpp, "ipc_refused", False) and transport_note:
        transport_note += " -- the ipc transport was refused as well: host shared memory"
    extra = {"rccl_ranks": rccl_ranks, "transport": transport, "transport_fallback": bool(fallback), "transport_note": transport_note,
             "ranks_share_device_0": os.environ.get("FALCON_PIPE_SAME_DEVICE") == "1"}

    def one_gpu_same_workload(hp1, wt1, mname, qname, steps, warmup):
        # the denominator of the scaling figure: the SAME workload (groups x batch lock-step streams, all blocks) on rank 0's GPU alone,
        # measured in the same job while the other ranks wait (None when the model does not fit one GPU's budget)
        val = None
        if rank == 0:
            try:
                t1, _, _, _, _ = run_cpp(a, 0, 1, local, hp1, wt1, mname, qname, dist, torch, groups, batch, n_ctx, steps, warmup)
                val = t1
            except Exception as e:                                   # (out of memory on a smaller part: report, do not fail the line)
                sys.stderr.write(f"bench_pipeline: one-GPU run of {mname} {qname} failed: {e}\n")
        if world > 1:
            dist.barrier()
        return val
    if world > 1 and not a.layers:
        extra["same_workload_1gpu_tok_s"] = one_gpu_same_workload(hp, wtype, a.model, a.quant, max(8, a.steps // 2), max(2, a.warmup // 2))
        # THE scaling figure of this line: the same streams on one GPU of this job, measured in this job (`value` of the --gpus 1 line is ONE stream: another workload)
        extra["scaling_vs_1gpu"] = (tok_s / extra["same_workload_1gpu_tok_s"]) if extra["same_workload_1gpu_tok_s"] else None
    if not getattr(a, "no_north_star", False) and world > 1 and a.model == "7b" and not a.layers:
        # the north-star configuration next to the headline line: Falcon-40B Q4_K, all 60 blocks, over the same GPUs
        hp40 = dict(synth.HP_40B)
        ns_tok_s, ns_wb, ns_blocks, ns_setup, _ = run_cpp(a, rank, world, local, hp40, tname["q4_K"], "40b", "q4_k", dist, torch, groups, batch, n_ctx,
                                                          max(8, a.steps // 4), max(2, a.warmup // 2))
        extra["north_star"] = {"workload": f"Falcon-40B Q4_K, 60 blocks, layer-pipelined over {world} GPU(s) ({ns_blocks} blocks per stage), "
                                           f"{groups} groups x {batch} lock-step greedy decode streams", "value": ns_tok_s, "unit": "tokens/s",
                               "weight_bytes_per_token": ns_wb, "effective_GBs": ns_wb * ns_tok_s / batch / 1e9, "setup_s": ns_setup}
        ns1 = one_gpu_same_workload(hp40, tname["q4_K"], "40b", "q4_k", max(8, a.steps // 4), max(2, a.warmup // 2))
        extra["north_star"]["same_workload_1gpu_tok_s"] = ns1
        extra["north_star"]["scaling_vs_1gpu"] = (ns_tok_s / ns1) if ns1 else None
    # cpu_baseline: the N = 1 leg's measurement -- the reference itself (oracle/_ref) on this box's host cores, one decode stream of the same model
    # (the reference has no multi-stream mode); rank 0 synthesizes the whole model on the host for it while the other ranks wait
    cpu = None
    if not getattr(a, "no_cpu", False) and not a.layers and a.model != "40b":
        if rank == 0:
            try:
                from bench import cpu_baseline
                wall = synth.make_model_fast(hp, wtype, seed=1234)
                cpu = cpu_baseline(wall, hp, wbytes, 32, getattr(a, "cpu_tokens", 32), synth.tokens(40, hp["n_vocab"], seed=42))
                cpu["sample"] += " -- ONE decode stream (the reference has no lock-step mode); same measurement as the --gpus 1 line's leg, with a 32-token prompt"
                del wall
            except Exception as e:                                   # (never lose the line to the baseline leg)
                cpu = {"error": repr(e)}
        if world > 1:
            dist.barrier()
    if rank == 0:
        from bench import kv_bytes_per_token, HBM_PEAK_GBS
        S = groups * batch
        # a weight pass serves `batch` tokens: bytes per TOKEN = weights / batch + that token's KV traffic
        b_tok = wbytes / batch + kv_bytes_per_token(hp, a.warmup + a.steps // 2)
        gbs = b_tok * tok_s / 1e9
        print(json.dumps({
            "metric": f"decode tokens/sec, Falcon-{a.model.upper()} {a.quant.upper()} layer-pipelined over {world} GPU(s), {groups * batch} lock-step decode streams in flight "
                      f"(MULTI-STREAM; the --gpus 1 line times ONE stream -- scale with scaling_vs_1gpu = value / same_workload_1gpu_tok_s); % of the summed HBM roofline",
            "value": tok_s, "unit": "tokens/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int8", "data": "synthetic (random-init blocks, seeds 1234+i; tokens mt(42))",
            "config": {"workload": f"Falcon-{a.model.upper()} {a.quant.upper()} layer-pipelined over {world} GPU(s) "
                                   f"({blocks} blocks per stage, balanced by bytes incl. lm_head), {groups} groups x {batch} lock-step greedy decode streams "
                                   f"in flight (MULTI-STREAM: {S} sequences, one weight pass serves {batch} tokens -- NOT the single-stream workload of --gpus 1; "
                                   f"compare with same_workload_1gpu_tok_s, the same streams on one GPU), a step = one round (one token per stream); residual rows "
                                   f"and sampled tokens by " + ({"shm": "host shared memory between the ranks (FALCON_PIPE_TRANSPORT=shm)",
                                                                 "ipc": "device-to-device copies into the peer's IPC-exported mailboxes (FALCON_PIPE_TRANSPORT=ipc)"}[env_tr] if shm else "RCCL ncclSend/ncclRecv")
                                   + " (csrc/falcon_pipeline.hip)" + ("" if not a.layers else f" [TRUNCATED to {a.layers} blocks: not the benchmark config]"),
                       "streams": S, "groups": groups, "batch": batch, "weight_bytes_per_pass": wbytes, "n_past_timed": [a.warmup, a.warmup + a.steps]},
            "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS * world, "unit": "GB/s", "frac": gbs / (HBM_PEAK_GBS * world),
                         "traffic": None, "note": "whole-job view: (weight bytes / batch + KV bytes) per token x tokens/s over the summed peak of all GPUs"},
            "cpu_baseline": cpu, "setup_s": t_setup, **extra,
        }))
    if world > 1:
        dist.destroy_process_group()
