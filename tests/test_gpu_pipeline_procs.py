"""GPU: the layer pipeline as REAL PROCESSES on the one-GPU box (takes the role of the reference's device loop with peer copies,
ggml-cuda.cu:2586-2608, 2713-2732). `python bench.py --gpus N` is run exactly as the driver runs it -- bench.py spawns the N ranks (RANK / WORLD_SIZE /
MASTER_* as torch.distributed.run sets them), the ranks form their gloo control group, rank 0 hands out the 128-byte unique id, every rank uploads its own
block range, creates its falcon_hip_pipeline, replays its stage graphs under the slot schedule, the last rank checks the token history, rank 0 prints the
JSON line -- with two switches that make it possible where RCCL refuses ("Duplicate GPU detected"): FALCON_PIPE_SAME_DEVICE=1 puts every rank on GPU 0, and
FALCON_PIPE_TRANSPORT=shm sends the hand-offs (residual rows [B][n_embd] f32 stage to stage, B token ids back to stage 0) through host shared memory
(csrc/falcon_pipeline.hip: exchange_shm) instead of ncclSend / ncclRecv. Everything except RCCL's p2p itself is the code path of --gpus 8.
The sampled tokens of every sequence must be those of the same streams on ONE stage in this process."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import ggllm_cpp_amd as g
from ggllm_cpp_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _single_process_tokens(hp, wtype, groups, batch, n_ctx, warmup, steps):
    g.init(0)
    w = synth.make_model_fast(hp, wtype, seed=1234)
    m = g.FalconModel(w, n_ctx=8, n_batch=1)
    pipe = g.Pipeline(m, 0, 1, groups, batch, n_ctx)
    pipe.set_tokens(synth.tokens(groups * batch, hp["n_vocab"], seed=42))
    pipe.run(warmup, 0)
    pipe.run(steps, warmup)
    hist = pipe.history(warmup, steps)
    pipe.free()
    m.free()
    return hist


# transport "ipc" (round 6): the same mailbox protocol with the payloads in the RECEIVING rank's device memory, exported by hipIpcGetMemHandle -- a send is one
# device-to-device copy into the peer's mailbox (over xGMI between two GPUs; on-device here, every rank sharing GPU 0): the first fall-back of a job whose RCCL
# pre-flight fails, ahead of the host-staged form
@pytest.mark.parametrize("world,batch,quant,transport", [(2, 2, "q4_0", "shm"), (3, 2, "q4_0", "shm"), (2, 16, "q5_1", "shm"), (3, 1, "q4_0", "shm"),
                                                         (2, 2, "q4_0", "ipc"), (3, 2, "q4_0", "ipc"), (2, 16, "q5_1", "ipc")])
def test_bench_gpus_n_as_processes_on_one_gpu(tmp_path, world, batch, quant, transport):
    layers, steps, warmup = 5, 6, 2
    dump = str(tmp_path / "hist.npy")
    env = dict(os.environ, FALCON_PIPE_TRANSPORT=transport, FALCON_PIPE_SAME_DEVICE="1", FALCON_PIPE_DUMP_HISTORY=dump, FALCON_PIPE_SHM_TIMEOUT_S="120",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--model", "tiny", "--quant", quant, "--layers", str(layers),
                        "--steps", str(steps), "--warmup", str(warmup), "--pipe-batch", str(batch), "--no-north-star", "--no-cpu"],
                       env=env, capture_output=True, timeout=900)
    assert r.returncode == 0, r.stderr.decode("utf-8", "replace")[-3000:]
    line = json.loads(r.stdout.decode().strip().splitlines()[-1])
    assert line["n_gpus"] == world and line["transport"].startswith(transport) and line["ranks_share_device_0"] is True
    assert line["transport_fallback"] is False                      # (the transport was asked for, not fallen back to)
    assert f"over {world} GPU" in line["metric"] and "MULTI-STREAM" in line["metric"]
    assert line["config"]["groups"] == 2 * world and line["config"]["batch"] == batch and line["value"] > 0
    assert "MULTI-STREAM" in line["config"]["workload"]
    got = np.load(dump)
    hp = dict(synth.HP_TINY_MQA); hp["n_layer"] = layers
    tname = {v: k for k, v in g.TYPE_NAME.items()}
    want = _single_process_tokens(hp, tname[quant], 2 * world, batch, 512, warmup, steps)
    assert got.shape == want.shape == (steps, 2 * world * batch)
    # 2 sequences per pass keep the column kernels (bit-identical to a stream of its own); 16 per pass run the small-batch mat-muls in BOTH jobs
    assert np.array_equal(got, want)


def test_a_rank_that_never_comes_ends_the_job_with_a_message(tmp_path):
    """rank 1 of 2 is never started: rank 0 must give up on the segment's attach count within the time-out, not hang"""
    env = dict(os.environ, FALCON_PIPE_TRANSPORT="shm", FALCON_PIPE_SAME_DEVICE="1", FALCON_PIPE_SHM_TIMEOUT_S="3", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    code = ("import sys; sys.path[:0] = [%r]\n"
            "import ggllm_cpp_amd as g\nfrom ggllm_cpp_amd import synth\n"
            "g.init(0)\nhp = dict(synth.HP_TINY_MQA); hp['n_layer'] = 4\n"
            "w = synth.make_model_fast(hp, 2, seed=1234, layers=range(0, 2))\n"
            "m = g.FalconModel(w, n_ctx=8, n_batch=1, layer_begin=0, layer_end=2)\n"
            "try:\n    g.Pipeline(m, 0, 2, 4, 2, 32, unique_id=g.Pipeline.unique_id())\n    print('created')\n"
            "except RuntimeError as e:\n    print('refused:', e)\n") % ROOT
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, timeout=300)
    assert b"refused" in r.stdout and b"every rank to attach" in r.stderr, (r.stdout[-500:], r.stderr[-1500:])


def test_rccl_preflight_refusal_falls_back_loudly(tmp_path):
    """no transport forced: every rank on GPU 0 makes RCCL refuse ("Duplicate GPU detected") in the pre-flight child processes (falcon_hip_rccl_selftest), and the
    job must run on with the host-staged transport and say so in its line -- the insurance for the first multi-GPU node, where RCCL between these ranks runs for
    the first time: a refusal or a hang there costs a fall-back, not the SCALE run"""
    dump = str(tmp_path / "hist.npy")
    env = dict(os.environ, FALCON_PIPE_SAME_DEVICE="1", FALCON_PIPE_DUMP_HISTORY=dump, FALCON_PIPE_SHM_TIMEOUT_S="120", FALCON_PIPE_PREFLIGHT_S="90", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "FALCON_PIPE_TRANSPORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--model", "tiny", "--quant", "q4_0", "--layers", "4", "--steps", "4", "--warmup", "2",
                        "--pipe-batch", "2", "--no-north-star", "--no-cpu"], env=env, capture_output=True, timeout=900)
    assert r.returncode == 0, r.stderr.decode("utf-8", "replace")[-3000:]
    line = json.loads(r.stdout.decode().strip().splitlines()[-1])
    # loud: the line says the hand-offs did not run over RCCL, and which transport carried them (device-to-device mailboxes first, host shared memory behind them)
    assert line["transport_fallback"] is True and "pre-flight FAILED" in line["transport_note"]
    assert line["transport"].startswith(("ipc", "shm"))
    hp = dict(synth.HP_TINY_MQA); hp["n_layer"] = 4
    want = _single_process_tokens(hp, g.Q4_0, 4, 2, 512, 2, 4)
    assert np.array_equal(np.load(dump), want)
