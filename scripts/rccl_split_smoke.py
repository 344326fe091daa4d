#!/usr/bin/env python3
"""RCCL smoke of the row-split tensor parallelism (csrc/split_tp.hip), for a box with two or more GPUs:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29518 scripts/rccl_split_smoke.py

Every rank uploads its row range of the same quantized matrices (-ts proportions 3:1:... ), multiplies, exchanges rows with
ncclSend / ncclRecv and compares the gathered result with the unsplit mat-mul computed on its own device: bit-identical on
every rank. (The ranges, the uploads and the per-part mat-muls are tested on one GPU in tests/test_gpu_split.py.)"""
import ctypes as C
import os
import sys

import numpy as np

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    import ggllm_cpp_amd as g
    from ggllm_cpp_amd import synth
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    torch.cuda.set_device(local)
    g.init(local)
    L = g.load()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    box = [g.Pipeline.unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    comm = L.ggml_hip_split_comm_create(rank, world, box[0])
    assert comm
    ts = [3.0] + [1.0] * (world - 1)
    ok = True
    for t, K, M, N in ((2, 4544, 4672, 1), (2, 4544, 18176, 40), (7, 1024, 999, 3), (12, 1024, 777, 1)):
        rng = np.random.default_rng(t + N)
        w = synth.random_blocks(t, M, K, rng)
        x = rng.standard_normal((N, K)).astype(np.float32)
        whole = g.Weight(t, w, K, M)
        want = whole.mul_mat(x)
        lo, hi = (C.c_int64 * world)(), (C.c_int64 * world)()
        L.ggml_hip_tensor_split_rows((C.c_float * world)(*ts), world, M, lo, hi)
        part = L.ggml_hip_weight_upload_rows(t, w.ctypes.data, K, M, lo[rank], hi[rank])
        xd, yd = g.DevBuf(host=x), g.DevBuf(N * M * 4)
        assert L.ggml_hip_mul_mat_q_split(comm, part, xd.ptr, K, N, yd.ptr, M, lo, hi) == 0
        same = bool(np.array_equal(yd.to_host(np.float32, (N, M)), want))
        ok &= same
        print(f"rank {rank}: type {t} [{M} x {K}] N={N}: {'ok' if same else 'MISMATCH'}", flush=True)
    L.ggml_hip_split_comm_free(comm)
    flag = torch.tensor([1 if ok else 0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    dist.destroy_process_group()
    sys.exit(0 if int(flag.item()) else 1)


if __name__ == "__main__":
    main()
