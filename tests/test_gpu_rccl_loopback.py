"""GPU: the library's RCCL binding and call sequence EXECUTED on a box with one GPU (RCCL refuses two ranks on one device, so the
multi-process jobs of scripts/rccl_*_smoke.py need a multi-GPU node).

  * layer pipeline (csrc/falcon_pipeline.hip): the in-process job of 1..3 stages with every hand-off -- residual rows [B][n_embd]
    f32 from stage to stage, B token ids from the last stage back to stage 0 -- sent through a communicator of ONE rank:
    ncclGetUniqueId / ncclCommInitRank through rccl_dyn.h's dlopen binding, grouped ncclSend / ncclRecv addressed to rank 0 itself
    on the pipeline's second stream, ordered against the stage steps by the events of the overlapped schedule. The sampled tokens
    must be those of the device-copy transport (falcon_hip_pipeline_run_local) and of the single-process greedy decode.
  * row-split tensor parallelism (csrc/split_tp.hip): pack -> grouped exchange -> unpack of ggml_hip_mul_mat_q_split over the same
    kind of communicator; the assembled matrix == the unsplit mat-mul.

Each case runs in a CHILD process under a time-out (the watchdog: a hung collective kills the child, not the test session); the child
prints `rccl_ranks: 1` = ncclCommCount of the communicator the bytes went through. Takes the role of the reference's device loop with
peer copies (ggml-cuda.cu:2713-2732, 2779-2788)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_PIPE = r"""
import sys, numpy as np
sys.path[:0] = [%(root)r, %(tests)r]
import ggllm_cpp_amd as g
from oracle import binding as ob
import synth, bench_pipeline as bp
world, groups, batch, wtype = (int(v) for v in sys.argv[1:5])
g.init(0)
oracle = ob.Oracle()
hp = dict(synth.HP_TINY_GQA if wtype == ob.Q5_1 else synth.HP_TINY_MQA); hp["n_layer"] = 5
w = synth.make_model(oracle, hp, wtype, seed=5)
rounds = 7
first = synth.tokens(groups * batch, hp["n_vocab"], seed=8)
m = g.FalconModel(w, n_ctx=64, n_batch=4)
want = np.stack([m.decode_greedy(int(t), 0, rounds) for t in first], axis=1)
m.free()
def run(rccl):
    parts = bp.partition(hp["n_layer"], world)
    stages = [g.FalconModel(w, n_ctx=8, n_batch=1, layer_begin=lb, layer_end=le) for lb, le in parts]
    ranks = [g.Pipeline(stages[r], r, world, groups, batch, 16, local=True) for r in range(world)]
    n = 0
    if rccl:
        g.Pipeline.attach_rccl(ranks)
        n = ranks[-1].rccl_ranks()
        assert all(p.rccl_ranks() == n for p in ranks)
    ranks[0].set_tokens(first)
    g.Pipeline.run_local(ranks, 3, 0)
    g.Pipeline.run_local(ranks, rounds - 3, 3)
    got = ranks[-1].history(0, rounds)
    for p in ranks: p.free()
    for s in stages: s.free()
    return got, n
copies, _ = run(False)
over_rccl, n = run(True)
print("rccl_ranks: %%d" %% n)
assert n == 1
assert np.array_equal(copies, want), "device-copy transport"
assert np.array_equal(over_rccl, want), "RCCL loop-back transport"
print("pipeline over RCCL ok")
"""

_SPLIT = r"""
import sys, ctypes as C, numpy as np
sys.path[:0] = [%(root)r, %(tests)r]
import ggllm_cpp_amd as g
from oracle import binding as ob
import synth
t, N = int(sys.argv[1]), int(sys.argv[2])
ts = [float(v) for v in sys.argv[3].split(",")]
g.init(0)
L = g.load()
oracle = ob.Oracle()
rng = np.random.default_rng(7 * t + N)
K, M = 1024, 1000 if t not in ob.KQUANTS else 777
w = np.ascontiguousarray(synth.quantized_matrix(oracle, t, M, K, rng))
x = rng.standard_normal((N, K)).astype(np.float32)
whole = g.Weight(t, w, K, M)
want = whole.mul_mat(x)
n = len(ts)
lo, hi = (C.c_int64 * n)(), (C.c_int64 * n)()
L.ggml_hip_tensor_split_rows((C.c_float * n)(*ts), n, M, lo, hi)
parts = (C.c_void_p * n)()
for r in range(n):
    parts[r] = L.ggml_hip_weight_upload_rows(t, w.ctypes.data, K, M, lo[r], hi[r])
xd, yd = g.DevBuf(x.nbytes), g.DevBuf(N * M * 4)
L.ggml_hip_memcpy_h2d(xd.ptr, x.ctypes.data, x.nbytes)
L.ggml_hip_memset(yd.ptr, 0xFF, N * M * 4)
comm = L.ggml_hip_split_comm_create_loopback(n)
assert comm, "no one-rank RCCL communicator"
ranks = L.ggml_hip_split_comm_rccl_ranks(comm)
print("rccl_ranks: %%d" %% ranks)
assert ranks == 1
assert L.ggml_hip_mul_mat_q_split_loopback(comm, parts, xd.ptr, K, N, yd.ptr, M, lo, hi) == 0
got = yd.to_host(np.float32, (N, M))
L.ggml_hip_split_comm_free(comm)
assert np.array_equal(got, want)
print("row split over RCCL ok")
"""


def _child(tmp_path, name, src, args, timeout=240):
    script = tmp_path / name
    script.write_text(src % {"root": ROOT, "tests": os.path.join(ROOT, "tests")})
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    try:
        r = subprocess.run([sys.executable, str(script), *[str(a) for a in args]], capture_output=True, text=True, env=env, timeout=timeout)
    except subprocess.TimeoutExpired as e:                        # the watchdog: the child (and its RCCL kernels) is killed, the session goes on
        pytest.fail("the RCCL exchange did not finish within %d s: %s" % (timeout, (e.stderr or b"")[-2000:]))
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert "rccl_ranks: 1" in r.stdout, r.stdout
    return r.stdout


@pytest.mark.parametrize("world,groups,batch,t", [(1, 2, 2, 2), (2, 2, 2, 2), (2, 4, 1, 7), (3, 6, 2, 2)])
def test_pipeline_hand_offs_through_a_one_rank_rccl_communicator(tmp_path, world, groups, batch, t):
    out = _child(tmp_path, "pipe.py", _PIPE, [world, groups, batch, t])
    assert "pipeline over RCCL ok" in out


@pytest.mark.parametrize("t,N,ts", [(2, 1, "1,1"), (2, 40, "3,1,2"), (12, 3, "1,0,1"), (7, 2, "1,1,1,1")])
def test_row_split_exchange_through_a_one_rank_rccl_communicator(tmp_path, t, N, ts):
    out = _child(tmp_path, "split.py", _SPLIT, [t, N, ts])
    assert "row split over RCCL ok" in out
