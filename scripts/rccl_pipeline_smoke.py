#!/usr/bin/env python3
"""RCCL smoke of the C++ layer pipeline, for the day two (or more) GPUs are reachable in one box:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 scripts/rccl_pipeline_smoke.py

Every rank holds its share of a tiny Falcon model (5 blocks, Q4_0 and Q5_1 GQA), rank 0 makes the RCCL unique id and hands it
out over the launcher's gloo group, falcon_hip_pipeline_run moves residual rows / sampled tokens with ncclSend / ncclRecv, and
the last rank checks the sampled tokens against the greedy decode of the whole model in its own process -- for the simple
schedule (groups = world) and the overlapped one (groups = 2 x world). The same check runs on ONE GPU with the local transport
in tests/test_gpu_pipeline.py; this script adds the transport."""
import os
import sys

import numpy as np

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)


def main():
    import torch
    import torch.distributed as dist
    import bench_pipeline as bp
    import ggllm_cpp_amd as g
    from ggllm_cpp_amd import synth
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if os.environ.get("FALCON_PIPE_SAME_DEVICE") == "1":         # debug: every rank on GPU 0
        local = 0
    torch.cuda.set_device(local)
    g.init(local)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ok = True
    for name, hp0, t in (("mqa q4_0", synth.HP_TINY_MQA, 2), ("gqa q5_1", synth.HP_TINY_GQA, 7)):
        hp = dict(hp0)
        hp["n_layer"] = max(5, world)
        parts = bp.partition(hp["n_layer"], world)
        lb, le = parts[rank]
        for groups, batch in ((world, 2), (2 * world, 2), (2 * world + 1, 1)):
            stage = g.FalconModel(synth.make_model_fast(hp, t, seed=1234, layers=range(lb, le)), n_ctx=8, n_batch=1, layer_begin=lb, layer_end=le)
            box = [g.Pipeline.unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            pipe = g.Pipeline(stage, rank, world, groups, batch, 32, unique_id=box[0])
            first = synth.tokens(groups * batch, hp["n_vocab"], seed=8)
            pipe.set_tokens(first)
            pipe.run(3, 0)
            pipe.run(6, 3)
            got = pipe.history(0, 9)
            if rank == world - 1:
                whole = g.FalconModel(synth.make_model_fast(hp, t, seed=1234), n_ctx=64, n_batch=4)
                want = np.stack([whole.decode_greedy(int(x), 0, 9) for x in first], axis=1)
                whole.free()
                same = bool(np.array_equal(got, want))
                ok &= same
                print(f"{name}: {world} stages, {groups} groups x {batch}: {'ok' if same else 'MISMATCH'}", flush=True)
            pipe.free()
            stage.free()
            dist.barrier()
    flag = torch.tensor([1 if ok else 0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    dist.destroy_process_group()
    sys.exit(0 if int(flag.item()) else 1)


if __name__ == "__main__":
    main()
