"""CPU: the N > 1 host logic of bench_pipeline.py (layer partition, round schedule, hidden-state hand-off, token ring)
run for real over torch.distributed/gloo with world_size 2 and 3 and a mock stage in place of the HIP stage step."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import bench_pipeline as bp

E, V = 16, 97


def _stage_fn(h, layer):                 # one "block": deterministic, order-sensitive
    return np.tanh(h * (1.0 + 0.01 * layer) + 0.1 * (layer + 1)).astype(np.float32)


def _embed(tok):
    return np.cos(np.arange(E, dtype=np.float32) * (tok + 1) * 0.37).astype(np.float32)


def _head(h):
    return int(np.abs(h * 1000).sum()) % V


class MockEngine:
    def __init__(self, rank, world, n_layer, n_streams):
        self.lb, self.le = bp.partition(n_layer, world)[rank]
        self.first, self.last = rank == 0, rank == world - 1
        self.hidden_in = [torch.zeros(E) for _ in range(n_streams)]
        self.hidden_out = [torch.zeros(E) for _ in range(n_streams)]
        self.tok_in = [torch.zeros(1, dtype=torch.int32) for _ in range(n_streams)]
        self.tok_out = [torch.zeros(1, dtype=torch.int32) for _ in range(n_streams)]
        self.trace = [[] for _ in range(n_streams)]

    def step(self, s, n_past):
        h = _embed(int(self.tok_in[s][0]) + n_past) if self.first else self.hidden_in[s].numpy().copy()
        for layer in range(self.lb, self.le):
            h = _stage_fn(h, layer)
        if self.last:
            t = _head(h)
            self.tok_out[s][0] = t
            self.trace[s].append(t)
        else:
            self.hidden_out[s].copy_(torch.from_numpy(h))

    def feed_back_token(self, s):
        self.tok_in[s].copy_(self.tok_out[s])


def _reference(n_layer, n_streams, rounds, init):
    out = []
    for s in range(n_streams):
        tok, tr = int(init[s]), []
        for k in range(rounds):
            h = _embed(tok + k)
            for layer in range(n_layer):
                h = _stage_fn(h, layer)
            tok = _head(h)
            tr.append(tok)
        out.append(tr)
    return out


def _worker(rank, world, port, n_layer, n_streams, rounds, init, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    eng = MockEngine(rank, world, n_layer, n_streams)
    for s in range(n_streams):
        eng.tok_in[s][0] = int(init[s])
    runner = bp.PipelineRunner(rank, world, n_streams, eng, bp.TorchComm(dist, eng))
    runner.run(2, 0)               # two calls in a row, like warm-up + timed region
    runner.run(rounds - 2, 2)
    dist.barrier()
    if rank == world - 1:
        q.put(eng.trace)
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_partition():
    assert bp.partition(60, 4) == [(0, 15), (15, 30), (30, 45), (45, 60)]
    assert [e - b for b, e in bp.partition(60, 8)] == [8, 8, 8, 8, 7, 7, 7, 7]
    assert bp.partition(32, 1) == [(0, 32)]
    for L, w in ((32, 8), (80, 8), (5, 3)):
        for hu in (0.0, 0.8, 1.4):
            parts = bp.partition(L, w, hu)
            assert len(parts) == w and all(e > b for b, e in parts)
            assert parts[0][0] == 0 and parts[-1][1] == L and all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
    # balanced by the bytes a stage streams: the last stage also owns lm_head (1.4 blocks' worth for Falcon-7B)
    from ggllm_cpp_amd import synth
    hu7, hu40 = bp.head_units(synth.HP_7B), bp.head_units(synth.HP_40B)
    assert 1.3 < hu7 < 1.5 and 0.7 < hu40 < 0.9
    def slowest(parts, hu):
        return max([e - b for b, e in parts[:-1]] + [parts[-1][1] - parts[-1][0] + hu])
    for w in (2, 4, 8):
        assert slowest(bp.partition(32, w, hu7), hu7) <= slowest(bp.partition(32, w), hu7)
    assert [e - b for b, e in bp.partition(32, 8, hu7)] == [5, 4, 4, 4, 4, 4, 4, 3]
    assert [e - b for b, e in bp.partition(32, 2, hu7)] == [17, 15]
    assert [e - b for b, e in bp.partition(60, 8, hu40)] == [8, 8, 8, 8, 7, 7, 7, 7]


@pytest.mark.parametrize("world,n_layer,n_streams", [(2, 5, 4), (3, 7, 6), (2, 4, 2), (3, 3, 3)])
def test_pipeline_rounds_over_gloo(world, n_layer, n_streams):
    rounds = 5
    init = [3 + 7 * s for s in range(n_streams)]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_layer, n_streams, rounds, init, q)) for r in range(world)]
    for p in procs:
        p.start()
    trace = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert trace == _reference(n_layer, n_streams, rounds, init)


def test_single_rank_runner_feeds_tokens_back():
    eng = MockEngine(0, 1, 3, 2)
    init = [5, 9]
    for s in range(2):
        eng.tok_in[s][0] = init[s]
    r = bp.PipelineRunner(0, 1, 2, eng, None)
    r.run(4, 0)
    assert eng.trace == _reference(3, 2, 4, init)
