// split_tp.hip -- the reference's row-split tensor parallelism (`-ts`, GGML_BACKEND_GPU_SPLIT) for one process per
// GPU: every rank holds a ROW RANGE of a quantized matrix, multiplies it with the full activations, and the ranks exchange
// their output rows with RCCL (grouped ncclSend / ncclRecv: the ranges may differ in size) instead of the reference's peer
// copies into the main device's buffer (ggml-cuda.cu:2779-2788).
//   row ranges   ggml_cuda_set_tensor_split (ggml-cuda.cu:2050-2077) + ggml_cuda_transform_tensor (:3044-3052), same float arithmetic
//   upload       rows [row_low, row_high) of the ggml block bytes (offset row_low * nb1, :3057-3066)
//   mat-mul      ggml_hip_mul_mat_q on the range, written straight into its rows of dst -- a row's dot products do not
//                depend on which rank computes them, so the gathered result is bit-identical to the unsplit mat-mul: the part
//                takes the form and the K split the WHOLE matrix's shape selects (fq_weight::form_M, fq_weight_upload_part)
// A local form runs every rank's part in one process on one device (tests/test_gpu_split.py).
#include "../../include/ggml-hip-ops.h"
#include "fq_device.h"
#include "hip_context.h"
#include "rccl_dyn.h"

#include <stdio.h>
#include <string.h>
#include <vector>

struct ggml_hip_split_comm {
    int rank = 0, world = 1;
    int virtual_world = 0;                                     // > 0: a loop-back communicator (one RCCL rank standing for this many ranks)
    float * tmp = nullptr; size_t tmp_bytes = 0;              // loop-back: the parts' results before they travel
    ncclComm_t comm = nullptr;
    float * stage = nullptr; size_t stage_bytes = 0;          // packed rows: [own part | one slot per peer]
};

extern "C" {

// host only. tensor_split: the n_devices proportions of `-ts` (all zero: even split, the reference's default table
// proportional to free VRAM there, ggml-cuda.cu:1999-2012: even on identical GPUs). row_low / row_high: n_devices entries.
void ggml_hip_tensor_split_rows(const float * tensor_split, int n_devices, int64_t nrows, int64_t * row_low, int64_t * row_high) {
    std::vector<float> start((size_t) n_devices + 1, 0.0f);
    bool all_zero = true;
    for (int i = 0; i < n_devices; ++i) if (tensor_split && tensor_split[i] != 0.0f) { all_zero = false; break; }
    if (all_zero) {
        for (int i = 0; i < n_devices; ++i) start[(size_t) i] = (float) i / (float) n_devices;
    } else {
        float split_sum = 0.0f;
        for (int i = 0; i < n_devices; ++i) { start[(size_t) i] = split_sum; split_sum += tensor_split[i]; }
        for (int i = 0; i < n_devices; ++i) {
            const float prop = tensor_split[i] / split_sum;
            if (prop == 0.0f) start[(size_t) i] = 1.0f; else start[(size_t) i] /= split_sum;       // ggml-cuda.cu:2067-2075
        }
    }
    for (int id = 0; id < n_devices; ++id) {
        const int lo = id == 0 ? 0 : (int)((float)(int) nrows * start[(size_t) id]);                  // int * float -> float -> int, :3046-3047
        const int hi = id == n_devices - 1 ? (int) nrows : (int)((float)(int) nrows * start[(size_t) id + 1]);
        row_low[id] = lo; row_high[id] = hi > lo ? hi : lo;
    }
}

ggml_hip_weight * ggml_hip_weight_upload_rows(int type, const void * host_blocks, int64_t K, int64_t nrows, int64_t row_low, int64_t row_high) {
    if (row_low < 0 || row_high > nrows || row_low >= row_high) return nullptr;                       // an empty range holds nothing (:3053-3055)
    const fq_type_desc d = fq_desc(type);
    if (d.blck == 0 || K % d.blck != 0) { fprintf(stderr, "ggml-hip: split upload: type %d with K=%lld unsupported\n", type, (long long) K); return nullptr; }
    const size_t nb1 = (size_t)(K / d.blck) * d.tsize;
    return fq_weight_upload_part(type, (const uint8_t *) host_blocks + (size_t) row_low * nb1, K, row_high - row_low, nrows);
}

ggml_hip_split_comm * ggml_hip_split_comm_create(int rank, int world, const void * unique_id) {
    if (world < 1 || rank < 0 || rank >= world) return nullptr;
    ggml_hip_split_comm * c = new ggml_hip_split_comm();
    c->rank = rank; c->world = world;
    if (world > 1) {
        rccl_api * R = fq_rccl();
        if (!R || !unique_id) { fprintf(stderr, "ggml-hip: split: %s\n", R ? "no unique id" : "RCCL is not available"); delete c; return nullptr; }
        ncclUniqueId id; memcpy(&id, unique_id, sizeof(id));
        const ncclResult_t rc = R->ncclCommInitRank(&c->comm, world, id, rank);
        if (rc != ncclSuccess) { fprintf(stderr, "ggml-hip: split: ncclCommInitRank(rank %d of %d): %s\n", rank, world, R->ncclGetErrorString(rc)); delete c; return nullptr; }
    }
    return c;
}
ggml_hip_split_comm * ggml_hip_split_comm_create_loopback(int virtual_world) {
    if (virtual_world < 1) return nullptr;
    rccl_api * R = fq_rccl();
    if (!R) { fprintf(stderr, "ggml-hip: split: RCCL is not available\n"); return nullptr; }
    ggml_hip_split_comm * c = new ggml_hip_split_comm();
    c->rank = 0; c->world = 1; c->virtual_world = virtual_world;
    ncclUniqueId id;
    ncclResult_t rc = R->ncclGetUniqueId(&id);
    if (rc == ncclSuccess) rc = R->ncclCommInitRank(&c->comm, 1, id, 0);
    if (rc != ncclSuccess) { fprintf(stderr, "ggml-hip: split: one-rank communicator: %s\n", R->ncclGetErrorString(rc)); delete c; return nullptr; }
    return c;
}
int ggml_hip_split_comm_rccl_ranks(ggml_hip_split_comm * c) {
    if (!c || !c->comm) return 0;
    int n = -1;
    if (fq_rccl()->ncclCommCount(c->comm, &n) != ncclSuccess) return -1;
    return n;
}
void ggml_hip_split_comm_free(ggml_hip_split_comm * c) {
    if (!c) return;
    HIP_CHECK(hipStreamSynchronize(fq_ctx().stream));
    if (c->comm) fq_rccl()->ncclCommDestroy(c->comm);
    if (c->stage) HIP_CHECK(hipFree(c->stage));
    if (c->tmp) HIP_CHECK(hipFree(c->tmp));
    delete c;
}

// collective: 0 when every rank of the job passed the same `n` bytes (the shim checks the `-ts` proportions once per job: row
// ranges computed from different proportions would exchange the wrong rows without any error), 1 otherwise
int ggml_hip_split_comm_agree(ggml_hip_split_comm * c, const void * bytes, size_t n) {
    if (!c || c->world == 1 || n == 0) return 0;
    hipStream_t st = fq_ctx().stream;
    uint8_t * dev = nullptr;
    HIP_CHECK(hipMalloc((void **) &dev, n * (size_t)(c->world + 1)));
    HIP_CHECK(hipMemcpyAsync(dev, bytes, n, hipMemcpyHostToDevice, st));
    RCCL_CHECK(fq_rccl()->ncclAllGather(dev, dev + n, n, ncclUint8, c->comm, st));
    std::vector<uint8_t> all(n * (size_t) c->world);
    HIP_CHECK(hipMemcpyAsync(all.data(), dev + n, all.size(), hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    HIP_CHECK(hipFree(dev));
    for (int r = 0; r < c->world; ++r) if (memcmp(all.data() + n * (size_t) r, bytes, n) != 0) return 1;
    return 0;
}

// dst[N][M] (row-major by token, ldd = M) = the unsplit mat-mul, on EVERY rank: this rank computes rows [row_low[rank],
// row_high[rank]) with its part `w_rows` (nullptr for an empty range), then the ranks exchange their rows. row_low / row_high:
// world entries (ggml_hip_tensor_split_rows). Returns 0. Synchronous like ggml_hip_mul_mat_q.
int ggml_hip_mul_mat_q_split(ggml_hip_split_comm * c, const ggml_hip_weight * w_rows, const float * x_dev, int64_t K, int64_t N,
                             float * dst_dev, int64_t M, const int64_t * row_low, const int64_t * row_high) {
    hipStream_t st = fq_ctx().stream;
    const int64_t lo = row_low[c->rank], rows = row_high[c->rank] - lo;
    if (rows > 0) {
        if (!w_rows) { fprintf(stderr, "ggml-hip: split: rank %d owns %lld rows but holds no weight part\n", c->rank, (long long) rows); return 1; }
        ggml_hip_mul_mat_q(w_rows, x_dev, K, N, dst_dev + lo, M);                                     // its rows of every token, in place
    }
    if (c->world == 1) return 0;
    rccl_api * R = fq_rccl();
    // pack -> grouped exchange -> unpack (a token's rows of one rank are contiguous only for N = 1)
    int64_t max_rows = 0;
    for (int r = 0; r < c->world; ++r) if (row_high[r] - row_low[r] > max_rows) max_rows = row_high[r] - row_low[r];
    const size_t slot = (size_t) N * (size_t) max_rows * 4, need = slot * (size_t) c->world;
    if (need > c->stage_bytes) {
        if (c->stage) HIP_CHECK(hipFree(c->stage));
        HIP_CHECK(hipMalloc((void **) &c->stage, need)); c->stage_bytes = need;
    }
    float * mine = (float *)((uint8_t *) c->stage + slot * (size_t) c->rank);
    if (rows > 0) HIP_CHECK(hipMemcpy2DAsync(mine, (size_t) rows * 4, dst_dev + lo, (size_t) M * 4, (size_t) rows * 4, (size_t) N, hipMemcpyDeviceToDevice, st));
    RCCL_CHECK(R->ncclGroupStart());
    for (int p = 0; p < c->world; ++p) {
        if (p == c->rank) continue;
        const int64_t prow = row_high[p] - row_low[p];
        if (rows > 0) RCCL_CHECK(R->ncclSend(mine, (size_t)(N * rows), ncclFloat32, p, c->comm, st));
        if (prow > 0) RCCL_CHECK(R->ncclRecv((uint8_t *) c->stage + slot * (size_t) p, (size_t)(N * prow), ncclFloat32, p, c->comm, st));
    }
    RCCL_CHECK(R->ncclGroupEnd());
    for (int p = 0; p < c->world; ++p) {
        const int64_t prow = row_high[p] - row_low[p];
        if (p == c->rank || prow <= 0) continue;
        HIP_CHECK(hipMemcpy2DAsync(dst_dev + row_low[p], (size_t) M * 4, (uint8_t *) c->stage + slot * (size_t) p, (size_t) prow * 4, (size_t) prow * 4, (size_t) N, hipMemcpyDeviceToDevice, st));
    }
    HIP_CHECK(hipStreamSynchronize(st));
    return 0;
}

// every rank's part in THIS process on one device, the exchange run for real over a one-rank RCCL communicator: part r's rows go into
// a private result matrix, are packed like ggml_hip_mul_mat_q_split packs them, travel as a grouped ncclSend / ncclRecv addressed to
// rank 0 itself, and are unpacked into dst (which nothing else writes)
int ggml_hip_mul_mat_q_split_loopback(ggml_hip_split_comm * c, ggml_hip_weight * const * parts, const float * x_dev, int64_t K, int64_t N,
                                      float * dst_dev, int64_t M, const int64_t * row_low, const int64_t * row_high) {
    if (!c || !c->comm || c->virtual_world < 1) return 1;
    const int W = c->virtual_world;
    hipStream_t st = fq_ctx().stream;
    rccl_api * R = fq_rccl();
    int64_t max_rows = 0;
    for (int r = 0; r < W; ++r) if (row_high[r] - row_low[r] > max_rows) max_rows = row_high[r] - row_low[r];
    const size_t slot = (size_t) N * (size_t) max_rows * 4, need = 2 * slot * (size_t) W, full = (size_t) N * (size_t) M * 4;
    if (need > c->stage_bytes) {
        if (c->stage) HIP_CHECK(hipFree(c->stage));
        HIP_CHECK(hipMalloc((void **) &c->stage, need ? need : 16)); c->stage_bytes = need;
    }
    if (full > c->tmp_bytes) {
        if (c->tmp) HIP_CHECK(hipFree(c->tmp));
        HIP_CHECK(hipMalloc((void **) &c->tmp, full)); c->tmp_bytes = full;
    }
    uint8_t * out_slots = (uint8_t *) c->stage, * in_slots = out_slots + slot * (size_t) W;
    for (int r = 0; r < W; ++r) {
        const int64_t rows = row_high[r] - row_low[r];
        if (rows <= 0) continue;
        if (!parts[r]) return 1;
        ggml_hip_mul_mat_q(parts[r], x_dev, K, N, c->tmp + row_low[r], M);
        HIP_CHECK(hipMemcpy2DAsync(out_slots + slot * (size_t) r, (size_t) rows * 4, c->tmp + row_low[r], (size_t) M * 4, (size_t) rows * 4, (size_t) N, hipMemcpyDeviceToDevice, st));
    }
    RCCL_CHECK(R->ncclGroupStart());
    for (int r = 0; r < W; ++r) {
        const int64_t rows = row_high[r] - row_low[r];
        if (rows <= 0) continue;
        RCCL_CHECK(R->ncclSend(out_slots + slot * (size_t) r, (size_t)(N * rows), ncclFloat32, 0, c->comm, st));
        RCCL_CHECK(R->ncclRecv(in_slots + slot * (size_t) r, (size_t)(N * rows), ncclFloat32, 0, c->comm, st));
    }
    RCCL_CHECK(R->ncclGroupEnd());
    for (int r = 0; r < W; ++r) {
        const int64_t rows = row_high[r] - row_low[r];
        if (rows <= 0) continue;
        HIP_CHECK(hipMemcpy2DAsync(dst_dev + row_low[r], (size_t) M * 4, in_slots + slot * (size_t) r, (size_t) rows * 4, (size_t) rows * 4, (size_t) N, hipMemcpyDeviceToDevice, st));
    }
    HIP_CHECK(hipStreamSynchronize(st));
    return 0;
}

// every rank's part in THIS process on one device (no exchange needed: all parts write into the same dst)
int ggml_hip_mul_mat_q_split_local(ggml_hip_weight * const * parts, int world, const float * x_dev, int64_t K, int64_t N,
                                   float * dst_dev, int64_t M, const int64_t * row_low, const int64_t * row_high) {
    for (int r = 0; r < world; ++r) {
        const int64_t rows = row_high[r] - row_low[r];
        if (rows <= 0) continue;
        if (!parts[r]) return 1;
        ggml_hip_mul_mat_q(parts[r], x_dev, K, N, dst_dev + row_low[r], M);
    }
    return 0;
}

}
