"""CPU: the slot schedule of the C++ pipeline (csrc/falcon_pipeline.hip, falcon_hip_pipeline_schedule -- host code of
libggml_hip.so, no device needed): (1) slot for slot the exchange ops and stage steps bench_pipeline.PipelineRunner issues
(whose data flow the gloo tests run for real), (2) executed with a mock stage and the local transport's phase order (sends of a
slot, its receives, its stage steps) it produces the tokens of the unpipelined loop."""
import numpy as np
import pytest

import bench_pipeline as bp
import ggllm_cpp_amd as g
from test_pipeline_gloo import E, _embed, _head, _reference, _stage_fn


class _RecEngine:
    def __init__(self, log):
        self.log = log

    def step(self, s, n_past):
        self.log.append(("compute", s, n_past))

    def feed_back_token(self, s):
        pass


class _RecComm:
    def __init__(self, log):
        self.log = log

    def post(self, sends, recvs):
        self.log.append(("exchange", tuple(sends), tuple(recvs)))
        return None

    def finish(self, posted):
        pass

    def exchange(self, sends, recvs):
        self.post(sends, recvs)


CASES = [(2, 2, 3), (2, 3, 4), (2, 4, 3), (3, 3, 2), (3, 5, 3), (3, 6, 4), (4, 4, 2), (4, 8, 3), (4, 9, 2), (8, 16, 2), (1, 3, 4)]


@pytest.mark.parametrize("world,groups,rounds", CASES)
def test_cpp_schedule_is_the_python_runners(world, groups, rounds):
    n_past0 = 5
    for rank in range(world):
        log = []
        bp.PipelineRunner(rank, world, groups, _RecEngine(log), _RecComm(log)).run(rounds, n_past0)
        mine = []
        for ops, grp, rnd in g.pipeline_schedule(rank, world, groups, rounds):
            sends = tuple((k[5:], s, peer) for k, s, peer in ops if k.startswith("send"))
            recvs = tuple((k[5:], s, peer) for k, s, peer in ops if k.startswith("recv"))
            if sends or recvs:
                mine.append(("exchange", sends, recvs))
            if grp is not None:
                mine.append(("compute", grp, n_past0 + rnd))
        assert mine == log, (rank, world, groups)


@pytest.mark.parametrize("world,groups,rounds", CASES)
def test_cpp_schedule_moves_the_right_data(world, groups, rounds):
    n_layer = 7
    parts = bp.partition(n_layer, world)
    init = [(3 * s + 1) % 97 for s in range(groups)]
    sched = [g.pipeline_schedule(r, world, groups, rounds) for r in range(world)]
    n_slots = len(sched[0])
    assert all(len(s) == n_slots for s in sched)
    hid_in = [[np.zeros(E, np.float32) for _ in range(groups)] for _ in range(world)]
    hid_out = [[np.zeros(E, np.float32) for _ in range(groups)] for _ in range(world)]
    tok_in = [[0] * groups for _ in range(world)]
    tok_out = [[0] * groups for _ in range(world)]
    mb_h = [[None] * groups for _ in range(world)]
    mb_t = [[None] * groups for _ in range(world)]
    tok_in[0] = list(init)
    trace = [[] for _ in range(groups)]
    for t in range(n_slots):
        posted = {}
        for r in range(world):
            for kind, grp, peer in sched[r][t][0]:
                if kind == "send_hidden":
                    mb_h[peer][grp] = hid_out[r][grp].copy(); posted[(peer, "h", grp, r)] = 1
                elif kind == "send_token":
                    mb_t[peer][grp] = tok_out[r][grp]; posted[(peer, "t", grp, r)] = 1
        for r in range(world):
            for kind, grp, peer in sched[r][t][0]:
                if kind == "recv_hidden":
                    assert posted.pop((r, "h", grp, peer))                 # the matching send is in the same slot's exchange
                    hid_in[r][grp] = mb_h[r][grp]
                elif kind == "recv_token":
                    assert posted.pop((r, "t", grp, peer))
                    tok_in[r][grp] = mb_t[r][grp]
        assert not posted                                                    # no send without its receive
        for r in range(world):
            _, grp, rnd = sched[r][t]
            if grp is None:
                continue
            h = _embed(tok_in[r][grp] + rnd) if r == 0 else hid_in[r][grp].copy()
            for layer in range(*parts[r]):
                h = _stage_fn(h, layer)
            if r == world - 1:
                tok_out[r][grp] = _head(h)
                trace[grp].append(tok_out[r][grp])
                if world == 1:
                    tok_in[0][grp] = tok_out[0][grp]
            else:
                hid_out[r][grp] = h
    assert trace == _reference(n_layer, groups, rounds, init)
