// Drives the ggml-cuda.h boundary of libggml_hip.so the way ggml.c / libfalcon.cpp do: loader -> 
// ggml_cuda_transform_tensor, scheduler -> ggml_cuda_can_mul_mat, every worker thread / phase -> ggml_cuda_compute_forward.
#define GGML_HIP_STANDALONE_ABI 1
#include "dropin/ggml-cuda.h"
#include <cstring>
#include <cstdio>
#include <thread>
#include <vector>

static ggml_tensor make2d(int type, int64_t ne0, int64_t ne1, size_t row_bytes, void * data, const char * name) {
    ggml_tensor t; memset(&t, 0, sizeof(t));
    t.type = (ggml_type) type; t.backend = GGML_BACKEND_CPU; t.n_dims = 2;
    t.ne[0] = ne0; t.ne[1] = ne1; t.ne[2] = 1; t.ne[3] = 1;
    t.nb[0] = type == GGML_TYPE_F32 ? 4 : 0; t.nb[1] = row_bytes; t.nb[2] = row_bytes * ne1; t.nb[3] = t.nb[2];
    t.data = data; t.meta.cuda_op_directive = -1; t.meta.layer_id = -1;
    snprintf(t.name, sizeof(t.name), "%s", name);
    return t;
}

// returns 0 on success; y receives [N][M]
extern "C" int shim_mul_mat(int type, const void * wblocks, size_t wrow_bytes, int64_t K, int64_t M, const float * x, int64_t N, float * y, int n_threads) {
    while (!ggml_init_cublas(true)) ggml_init_cublas(false);            // libfalcon polls check_only (libfalcon.cpp:1947)
    const GPUStatus * st = ggml_cuda_get_system_gpu_status();
    if (st->num_devices < 1 || st->total_vram == 0) return 10;
    ggml_tensor W = make2d(type, K, M, wrow_bytes, (void *) wblocks, "w");
    W.backend = GGML_BACKEND_GPU_SPLIT;                                  // libfalcon.cpp:1857
    ggml_cuda_transform_tensor((void *) wblocks, &W);                    // libfalcon.cpp:1251
    if (!W.extra) return 11;
    ggml_tensor X = make2d(GGML_TYPE_F32, K, N, (size_t) K * 4, (void *) x, "x");
    ggml_tensor Y = make2d(GGML_TYPE_F32, M, N, (size_t) M * 4, y, "y");
    Y.op = GGML_OP_MUL_MAT; Y.src0 = &W; Y.src1 = &X;
    if (!ggml_cuda_can_mul_mat(&W, &X, &Y)) return 12;                   // ggml.c:17412 -> n_tasks = 1
    // every thread of the pool calls the hook in every phase (ggml.c:15779-15790); only ith == 0 / COMPUTE acts
    int rc = 0;
    for (int phase = 0; phase < 3; ++phase) {
        std::vector<std::thread> th;
        std::vector<int> ok((size_t) n_threads, 0);
        for (int i = 0; i < n_threads; ++i) th.emplace_back([&, i]() {
            ggml_compute_params p; p.type = (ggml_task_type) phase; p.ith = i; p.nth = n_threads; p.wsize = 0; p.wdata = nullptr;
            ok[(size_t) i] = ggml_cuda_compute_forward(&p, &Y) ? 1 : 0; });
        for (auto & t : th) t.join();
        for (int v : ok) if (!v) rc = 13;                                // "handled, skip CPU" for every thread
    }
    if (Y.meta.cuda_perf_mal_mul_type != 1) rc = rc ? rc : 14;
    // a CPU-only node must be declined so that ggml.c runs it
    ggml_tensor Z = make2d(GGML_TYPE_F32, M, N, (size_t) M * 4, y, "z"); Z.op = GGML_OP_SOFT_MAX; Z.src0 = &Y;
    ggml_compute_params p; p.type = GGML_TASK_COMPUTE; p.ith = 0; p.nth = 1; p.wsize = 0; p.wdata = nullptr;
    if (ggml_cuda_compute_forward(&p, &Z)) rc = rc ? rc : 15;
    ggml_cuda_free_data(&W);
    if (W.extra) rc = rc ? rc : 16;
    void * pin = ggml_cuda_host_malloc(1 << 20);                         // llama-util.h:462
    if (!pin) rc = rc ? rc : 17; else ggml_cuda_host_free(pin);
    return rc;
}
