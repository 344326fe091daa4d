// ggml_cuda_shim.hip -- the reference's device-backend boundary (ggml-cuda.h, 27 C entry points) on top of the HIP
// kernels, so that ggml.c / libfalcon.cpp / the CLIs built with -DGGML_USE_CUBLAS link against libggml_hip.so unchanged.
// Reference behaviour per function: ggml-cuda.cu (lines cited inline). No CUDA headers, no CUDA/HIP dual path.
//
// Data flow of the per-node hook is the reference's (activations live on the host in a ggml graph): src1 is copied to
// the device, the quantized mat-mul runs with the CPU arithmetic, dst is copied back. The device-RESIDENT fast path
// (no per-op PCIe round trip) is falcon_hip_eval in falcon-hip.h; INTEGRATION.md shows where libfalcon would call it.
#define GGML_HIP_STANDALONE_ABI 1
#include "../../include/dropin/ggml-cuda.h"
#include "../../include/ggml-hip-ops.h"
#include "fq_device.h"
#include "hip_context.h"

#include <atomic>
#include <map>
#include <mutex>
#include <string.h>

static GPUStatus        g_status;                 // ggml-cuda.cu:1876 g_system_gpu_status
static std::atomic<bool> g_inited{false};
static std::mutex       g_init_mutex;
static int              g_main_device = 0;
static int              g_max_gpus = GGML_CUDA_MAX_DEVICES;
static int64_t          g_vram_reserved_mb = 0;
static float            g_tensor_split[GGML_CUDA_MAX_DEVICES] = {0};
static size_t           g_scratch_size = 0;

static bool is_quantized(int t) {
    return t == GGML_TYPE_Q4_0 || t == GGML_TYPE_Q4_1 || t == GGML_TYPE_Q5_0 || t == GGML_TYPE_Q5_1 || t == GGML_TYPE_Q8_0 ||
           (t >= GGML_TYPE_Q2_K && t <= GGML_TYPE_Q6_K);
}
static bool on_device(const ggml_tensor * t) { return t && (t->backend == GGML_BACKEND_GPU || t->backend == GGML_BACKEND_GPU_SPLIT); }

// ---------------------------------------------------------------------------------------------- status / init
extern "C" void ggml_cuda_update_gpu_status(int device_id) {                       // ggml-cuda.cu:1886-1943
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
    if (n > GGML_CUDA_MAX_DEVICES) n = GGML_CUDA_MAX_DEVICES;
    if (n > g_max_gpus) n = g_max_gpus;
    g_status.num_devices = n;
    g_status.max_gpus = g_max_gpus;
    g_status.main_device_id = g_main_device < n ? g_main_device : 0;
    int cur = 0; (void) hipGetDevice(&cur);
    for (int id = 0; id < n; ++id) {
        if (device_id >= 0 && id != device_id) continue;
        size_t fr = 0, tot = 0;
        if (hipSetDevice(id) == hipSuccess && hipMemGetInfo(&fr, &tot) == hipSuccess) {
            g_status.device_vram_free[id] = fr; g_status.device_vram_total[id] = tot;
        }
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, id) == hipSuccess) {
            snprintf(g_status.device_props[id].name, sizeof(g_status.device_props[id].name), "%s", p.name);
            g_status.device_props[id].totalGlobalMem = p.totalGlobalMem;
            g_status.device_props[id].multiProcessorCount = p.multiProcessorCount;
            g_status.device_props[id].clockRate = p.clockRate;
            g_status.device_props[id].major = p.major; g_status.device_props[id].minor = p.minor;
        }
    }
    (void) hipSetDevice(cur);
    g_status.total_vram = 0; g_status.total_free_vram = 0;
    for (int id = 0; id < n; ++id) { g_status.total_vram += g_status.device_vram_total[id]; g_status.total_free_vram += g_status.device_vram_free[id]; }
}

extern "C" const GPUStatus * ggml_cuda_get_system_gpu_status(void) { return &g_status; }

extern "C" bool ggml_init_cublas(bool check_only) {                                 // ggml-cuda.cu:1982-2041
    if (check_only) return g_inited.load();                                         // must not block (polled every 50 ms)
    std::lock_guard<std::mutex> lock(g_init_mutex);
    if (g_inited.load()) return true;
    ggml_hip_init(g_main_device);                                                   // aborts without a device: no CPU fallback
    ggml_cuda_update_gpu_status(-1);
    for (int id = 0; id < g_status.num_devices; ++id)
        fprintf(stderr, "%s: HIP device %d: %s, %.1f GB\n", __func__, id, g_status.device_props[id].name, g_status.device_vram_total[id] / 1e9);
    g_inited.store(true);
    return true;
}

extern "C" void ggml_cuda_print_gpu_status(const GPUStatus * st, bool print_summary) {   // ggml-cuda.cu:1945-1972
    if (!st) return;
    fprintf(stderr, "+-----+------------------------------------+----------------+----------------+\n");
    fprintf(stderr, "| ID  | Device                             | VRAM total MB  | VRAM free MB   |\n");
    fprintf(stderr, "+-----+------------------------------------+----------------+----------------+\n");
    for (int id = 0; id < st->num_devices; ++id)
        fprintf(stderr, "| %2d%c | %-34.34s | %14zu | %14zu |\n", id, id == st->main_device_id ? '*' : ' ', st->device_props[id].name,
                st->device_vram_total[id] / D_MB, st->device_vram_free[id] / D_MB);
    fprintf(stderr, "+-----+------------------------------------+----------------+----------------+\n");
    if (print_summary) fprintf(stderr, "  total %zu MB, free %zu MB, %d device(s)\n", st->total_vram / D_MB, st->total_free_vram / D_MB, st->num_devices);
}

extern "C" void ggml_cuda_set_max_gpus(int n)             { g_max_gpus = n > 0 ? n : 1; }
extern "C" void ggml_cuda_set_main_device(int d)          { g_main_device = d; g_status.main_device_id = d; }
extern "C" void ggml_cuda_set_vram_reserved(int64_t mb)   { g_vram_reserved_mb = mb; for (int i = 0; i < GGML_CUDA_MAX_DEVICES; ++i) g_status.device_vram_reserved[i] = mb; }
extern "C" void ggml_cuda_set_tensor_split_prepare(const float * ts, int n) { for (int i = 0; i < GGML_CUDA_MAX_DEVICES; ++i) g_tensor_split[i] = (ts && i < n) ? ts[i] : 0.0f; }
extern "C" void ggml_cuda_set_tensor_split(const float * ts) { if (ts) memcpy(g_tensor_split, ts, sizeof(g_tensor_split)); }
extern "C" void ggml_cuda_set_scratch_size(size_t s)      { g_scratch_size = s; }
extern "C" void ggml_cuda_free_scratch(void)              { g_scratch_size = 0; }
// ---------------------------------------------------------------------------------------------- staging-buffer pool
// The per-op staging buffers (src1 in, dst out) come from a pool instead of a hipMalloc / hipFree pair per graph node --
// the role of ggml_cuda_pool_malloc / ggml_cuda_pool_free in the reference (ggml-cuda.cu:1738-1816). Free buffers are kept
// in a size-ordered map: a request takes the smallest free buffer that holds it (and is not more than 4x too large, so
// a prefill-sized buffer is not pinned under a decode-sized request), otherwise allocates with 1/16 head-room rounded up to
// 256 KiB so that slowly growing requests (n_past) reuse the buffer. Every hand-out counts as an access; falcon_eval's
// housekeeping (libfalcon.cpp:4573-4587) purges free buffers nobody asked for since the last reset.
namespace {
struct pool_buf { void * ptr; size_t size; int access; };
struct shim_pool {
    std::mutex mu;
    std::multimap<size_t, pool_buf> free_;                  // by size
    std::map<void *, pool_buf> busy_;
    size_t n_alloc = 0, n_reuse = 0;
};
shim_pool & pool() { static shim_pool p; return p; }

void * pool_get(size_t bytes) {
    if (bytes == 0) bytes = 16;
    shim_pool & P = pool();
    std::lock_guard<std::mutex> lk(P.mu);
    auto it = P.free_.lower_bound(bytes);
    if (it != P.free_.end() && it->first <= 4 * bytes + (1u << 20)) {
        pool_buf b = it->second; P.free_.erase(it);
        ++b.access; ++P.n_reuse;
        P.busy_[b.ptr] = b;
        return b.ptr;
    }
    pool_buf b;
    b.size = (bytes + bytes / 16 + 262143) & ~(size_t) 262143;
    b.ptr = ggml_hip_malloc(b.size); b.access = 1;
    ++P.n_alloc;
    P.busy_[b.ptr] = b;
    return b.ptr;
}
void pool_put(void * p) {
    if (!p) return;
    shim_pool & P = pool();
    std::lock_guard<std::mutex> lk(P.mu);
    auto it = P.busy_.find(p);
    if (it == P.busy_.end()) { fprintf(stderr, "ggml-hip: pool_put of a pointer the pool did not hand out\n"); abort(); }
    P.free_.emplace(it->second.size, it->second);
    P.busy_.erase(it);
}
}   // namespace

extern "C" void ggml_cuda_pool_reset_all_counters(int device_id) {                                  // ggml-cuda.cu:1843-1853
    if (device_id != g_status.main_device_id) return;
    shim_pool & P = pool();
    std::lock_guard<std::mutex> lk(P.mu);
    for (auto & kv : P.free_) kv.second.access = 0;
    for (auto & kv : P.busy_) kv.second.access = 0;
}
extern "C" int ggml_cuda_pool_purge_buffers_with_access_count(int min_access_count, int device_id) { // ggml-cuda.cu:1818-1841
    if (device_id != g_status.main_device_id) return 0;
    shim_pool & P = pool();
    std::lock_guard<std::mutex> lk(P.mu);
    int purged = 0;
    for (auto it = P.free_.begin(); it != P.free_.end(); ) {
        if (it->second.access < min_access_count) { ggml_hip_free(it->second.ptr); it = P.free_.erase(it); ++purged; }
        else ++it;
    }
    return purged;
}
// test hook: buffers allocated / hand-outs served from the pool / buffers currently free
extern "C" void ggml_hip_shim_pool_stats(size_t * n_alloc, size_t * n_reuse, size_t * n_free) {
    shim_pool & P = pool();
    std::lock_guard<std::mutex> lk(P.mu);
    if (n_alloc) *n_alloc = P.n_alloc;
    if (n_reuse) *n_reuse = P.n_reuse;
    if (n_free)  *n_free  = P.free_.size();
}

// ---------------------------------------------------------------------------------------------- pinned host memory
extern "C" void * ggml_cuda_host_malloc(size_t size) {                              // ggml-cuda.cu:2079-2098
    if (getenv("GGML_CUDA_NO_PINNED") != nullptr) return nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0) return nullptr;              // caller falls back to pageable memory
    void * p = nullptr;
    if (hipHostMalloc(&p, size, hipHostMallocDefault) != hipSuccess) {
        (void) hipGetLastError();
        fprintf(stderr, "WARNING: failed to allocate %.2f MB of pinned memory\n", size / 1024.0 / 1024.0);
        return nullptr;
    }
    return p;
}
extern "C" void ggml_cuda_host_free(void * p) { if (p) HIP_CHECK(hipHostFree(p)); }

// ---------------------------------------------------------------------------------------------- weights
struct shim_extra : ggml_tensor_extra_gpu { bool is_weight; size_t bytes; bool split; ggml_hip_split_comm * comm; int64_t row_low[GGML_CUDA_MAX_DEVICES], row_high[GGML_CUDA_MAX_DEVICES]; };

// Row-split tensor parallelism over several PROCESSES (one per GPU, every process running the same reference graph): once
// ggml_hip_split_configure has joined this process to a job, GGML_BACKEND_GPU_SPLIT tensors are uploaded as this rank's row
// range of the `-ts` proportions (ggml_cuda_set_tensor_split) and their mat-muls exchange output rows by RCCL
// (csrc/split_tp.hip). Not configured (the default): the whole matrix lives on this process's device.
static ggml_hip_split_comm * g_split = nullptr;
static int g_split_rank = 0, g_split_world = 1;
static bool g_split_ts_checked = false;
static int g_split_live = 0;            // split tensors uploaded under the current communicator and not yet freed
extern "C" int ggml_hip_split_configure(int rank, int world, const void * unique_id) {
    // a split tensor holds this rank's rows of the job it was uploaded under: the job cannot change below it
    if (g_split_live) { fprintf(stderr, "ggml-hip: split: %d split tensors are alive -- free them (ggml_cuda_free_data) before reconfiguring\n", g_split_live); return 1; }
    if (g_split) { ggml_hip_split_comm_free(g_split); g_split = nullptr; }
    g_split_rank = 0; g_split_world = 1;
    if (world <= 1) return 0;
    if (world > GGML_CUDA_MAX_DEVICES) { fprintf(stderr, "ggml-hip: split: at most %d ranks\n", GGML_CUDA_MAX_DEVICES); return 1; }
    ggml_init_cublas(false);
    g_split = ggml_hip_split_comm_create(rank, world, unique_id);
    if (!g_split) return 1;
    g_split_rank = rank; g_split_world = world; g_split_ts_checked = false;
    return 0;
}

extern "C" void ggml_cuda_transform_tensor(void * data, ggml_tensor * t) {          // ggml-cuda.cu:3030-3073
    if (!on_device(t)) return;
    ggml_init_cublas(false);
    shim_extra * ex = new shim_extra();
    memset(ex->data_device, 0, sizeof(ex->data_device));
    ex->split = false; ex->comm = nullptr;
    if (is_quantized(t->type)) {
        const int64_t nrows = t->ne[1] * t->ne[2] * t->ne[3];
        if (t->backend == GGML_BACKEND_GPU_SPLIT && g_split) {
            // this rank's rows of the -ts split (ggml-cuda.cu:3044-3066); an empty range holds nothing
            if (!g_split_ts_checked) {      // once per job, collectively (every rank uploads the same tensors in the same order)
                if (ggml_hip_split_comm_agree(g_split, g_tensor_split, sizeof(g_tensor_split)) != 0) { fprintf(stderr, "ggml-hip: split: rank %d: the ranks of this job were given different -ts proportions\n", g_split_rank); exit(1); }
                g_split_ts_checked = true;
            }
            ggml_hip_tensor_split_rows(g_tensor_split, g_split_world, nrows, ex->row_low, ex->row_high);
            ex->data_device[0] = ggml_hip_weight_upload_rows((int) t->type, data, t->ne[0], nrows, ex->row_low[g_split_rank], ex->row_high[g_split_rank]);
            ex->split = true; ex->comm = g_split; ++g_split_live;
        } else {
            // the whole matrix on this process's device: the north star shards by LAYER (one process per GPU)
            ex->data_device[0] = ggml_hip_weight_upload((int) t->type, data, t->ne[0], nrows);
        }
        ex->is_weight = true;
    } else if (t->type == GGML_TYPE_F32) {
        const size_t n = (size_t) t->ne[0] * t->ne[1] * t->ne[2] * t->ne[3] * 4;
        void * d = ggml_hip_malloc(n);
        ggml_hip_memcpy_h2d(d, data, n);
        ex->data_device[0] = d; ex->is_weight = false; ex->bytes = n;
    } else {
        fprintf(stderr, "ggml-hip: transform_tensor: tensor '%s' of type %d cannot be offloaded (f16 weights are not on the Falcon path)\n", t->name, (int) t->type);
        exit(1);
    }
    t->extra = ex;
}

extern "C" void ggml_cuda_free_data(ggml_tensor * t) {                              // ggml-cuda.cu:3075-3092
    if (!t || !on_device(t) || !t->extra) return;
    shim_extra * ex = (shim_extra *) t->extra;
    if (ex->split && g_split_live > 0) --g_split_live;
    if (ex->is_weight) { if (ex->data_device[0]) ggml_hip_weight_free((ggml_hip_weight *) ex->data_device[0]); }
    else               ggml_hip_free(ex->data_device[0]);
    delete ex;
    t->extra = nullptr;
}

// libfalcon sets the VRAM scratch size to 0 (libfalcon.cpp:1744-1745), which makes these no-ops in the reference too
// (ggml-cuda.cu:3095-3097): activations of a ggml graph stay on the host.
extern "C" void ggml_cuda_assign_buffers(ggml_tensor *) {}
extern "C" void ggml_cuda_assign_buffers_no_scratch(ggml_tensor *) {}

// ---------------------------------------------------------------------------------------------- compute
extern "C" bool ggml_cuda_can_mul_mat(const ggml_tensor * src0, const ggml_tensor * src1, ggml_tensor * dst) {   // ggml-cuda.cu:2842-2866
    if (!src0 || !src1 || !dst) return false;
    if (dst->meta.cuda_op_directive == 0) return false;
    // only weights that already live in HBM: re-uploading host weights per call (what the reference does for CPU-backend
    // src0) would stream the matrix over PCIe every time
    return on_device(src0) && is_quantized(src0->type) && src1->type == GGML_TYPE_F32 && dst->type == GGML_TYPE_F32;
}

__global__ void k_mul_rows(const float * __restrict__ a, const float * __restrict__ w, float * __restrict__ y, int64_t n, int64_t ne10) {
    for (int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t) gridDim.x * blockDim.x) y[i] = a[i] * w[i % ne10];
}

extern "C" void ggml_cuda_mul(const ggml_tensor * src0, const ggml_tensor * src1, ggml_tensor * dst) {           // ggml-cuda.cu:2151-2200
    // dst = src0 * src1 with src1 (a [ne10] f32 weight resident in HBM) broadcast over rows; src0/dst are host f32
    hip_context & c = fq_ctx();
    const int64_t n = src0->ne[0] * src0->ne[1] * src0->ne[2] * src0->ne[3];
    float * a = (float *) pool_get((size_t) n * 4), * y = (float *) pool_get((size_t) n * 4);
    ggml_hip_memcpy_h2d(a, src0->data, (size_t) n * 4);
    const float * w = (const float *) ((shim_extra *) src1->extra)->data_device[0];
    hipLaunchKernelGGL(k_mul_rows, dim3((unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256)), dim3(256), 0, c.stream, a, w, y, n, src1->ne[0]);
    ggml_hip_memcpy_d2h(dst->data, y, (size_t) n * 4);
    pool_put(a); pool_put(y);
}

static void shim_mul_mat(const ggml_tensor * src0, const ggml_tensor * src1, ggml_tensor * dst) {                 // ggml-cuda.cu:2931-2951 + 2520-2820
    const int64_t K = src0->ne[0], M = src0->ne[1], N = src1->ne[1] * src1->ne[2] * src1->ne[3];
    if (src1->ne[0] != K || src1->nb[0] != 4 || dst->nb[0] != 4 || src1->nb[1] != (size_t) K * 4 || dst->nb[1] != (size_t) M * 4) {
        fprintf(stderr, "ggml-hip: mul_mat '%s': non-contiguous src1/dst are not supported by the shim\n", dst->name); exit(1);
    }
    const shim_extra * ex = (const shim_extra *) src0->extra;
    const ggml_hip_weight * w = (const ggml_hip_weight *) ex->data_device[0];
    float * x = (float *) pool_get((size_t) N * K * 4), * y = (float *) pool_get((size_t) N * M * 4);
    ggml_hip_memcpy_h2d(x, src1->data, (size_t) N * K * 4);                         // reference: H2D of src1 every op (ggml-cuda.cu:2717)
    if (ex->split) { if (!ex->comm || ex->comm != g_split) { fprintf(stderr, "ggml-hip: mul_mat '%s': split tensor without its communicator\n", dst->name); exit(1); }
                     if (ggml_hip_mul_mat_q_split(ex->comm, w, x, K, N, y, M, ex->row_low, ex->row_high) != 0) exit(1); }   // rows exchanged by RCCL (reference: peer copies, :2779-2788)
    else ggml_hip_mul_mat_q(w, x, K, N, y, M);
    ggml_hip_memcpy_d2h(dst->data, y, (size_t) N * M * 4);                          // reference: D2H of dst every op (ggml-cuda.cu:2787-2791)
    pool_put(x); pool_put(y);
    dst->meta.cuda_perf_mal_mul_type = 1;                                           // "quantized kernel" tag of the timing table
}

extern "C" bool ggml_cuda_compute_forward(ggml_compute_params * params, ggml_tensor * t) {                        // ggml-cuda.cu:3193-3292
    const bool any_on_device = on_device(t) || on_device(t->src0) || on_device(t->src1);
    bool handled = false;
    switch ((int) t->op) {
        case GGML_OP_MUL_MAT: handled = any_on_device || ggml_cuda_can_mul_mat(t->src0, t->src1, t); break;
        case GGML_OP_MUL:     handled = any_on_device; break;
        case GGML_OP_RESHAPE: case GGML_OP_VIEW: case GGML_OP_PERMUTE: case GGML_OP_TRANSPOSE: handled = any_on_device; break;   // no-ops on the device
        default:
            if (any_on_device) { fprintf(stderr, "ggml-hip: op %d on device-resident tensor '%s' is not part of the Falcon path\n", (int) t->op, t->name); exit(1); }
            return false;                                                           // CPU computes it
    }
    if (!handled) return false;
    if (params->ith != 0) return true;                                              // single writer (ggml-cuda.cu:3284)
    if (params->type == GGML_TASK_INIT || params->type == GGML_TASK_FINALIZE) return true;   // ggml-cuda.cu:3287
    if (t->op == GGML_OP_MUL_MAT) {
        if (!on_device(t->src0) || !is_quantized(t->src0->type)) { fprintf(stderr, "ggml-hip: mul_mat '%s': src0 must be a quantized weight in HBM\n", t->name); exit(1); }
        shim_mul_mat(t->src0, t->src1, t);
    } else if (t->op == GGML_OP_MUL) {
        if (on_device(t->src1) && !on_device(t->src0)) ggml_cuda_mul(t->src0, t->src1, t);
        else { fprintf(stderr, "ggml-hip: mul '%s': only host-activation x device-weight is on the Falcon path\n", t->name); exit(1); }
    }
    return true;
}

extern "C" size_t ggml_cuda_mul_mat_get_wsize(const ggml_tensor *, const ggml_tensor *, ggml_tensor *) { return 0; }
extern "C" void   ggml_cuda_mul_mat(const ggml_tensor *, const ggml_tensor *, ggml_tensor *, void *, size_t) {
    fprintf(stderr, "ggml-hip: ggml_cuda_mul_mat(wdata) is declared by the reference but never defined or called there\n"); exit(1);
}
