/*
 * ggml-cuda.h -- drop-in replacement of cmp-nct/ggllm.cpp's ggml-cuda.h (its device-backend boundary) for the
 * MI355X/HIP backend in libggml_hip.so. Copy (or -I) this file over the reference's header, build the tree with
 * -DGGML_USE_CUBLAS exactly as before and link libggml_hip.so instead of ggml-cuda.o/cudart/cublas.
 *
 * Differences from the reference header (ggml-cuda.h:1-64), all source-compatible with ggml.c / libfalcon.cpp /
 * llama-util.h / the CLIs:
 *   - no <cuda_runtime.h>: GPUStatus::device_props is a plain struct (callers only read the other fields; the
 *     reference itself only uses .name, ggml-cuda.cu:1957-1961)
 *   - the two declarations the reference never defines nor calls (ggml_cuda_mul_mat, ggml_cuda_mul_mat_get_wsize,
 *     ggml-cuda.h:43-44) are kept for source compatibility and exported as aborting stubs
 */
#pragma once
#ifdef GGML_HIP_STANDALONE_ABI
#include "../ggml-abi.h"       /* building libggml_hip.so itself */
#else
#include "ggml.h"              /* inside a ggml tree: the real definitions */
#endif

#ifdef  __cplusplus
extern "C" {
#endif

#define GGML_CUDA_MAX_DEVICES       16
#define D_MB                        (1024*1024)

struct ggml_tensor_extra_gpu {                       /* ggml-cuda.h:13-15 */
    void * data_device[GGML_CUDA_MAX_DEVICES];       /* [0]: the backend's weight handle for this tensor */
};

struct ggml_hip_device_props {                       /* stands where struct cudaDeviceProp stood */
    char   name[256];
    size_t totalGlobalMem;
    int    multiProcessorCount;
    int    clockRate;
    int    major, minor;
};

typedef struct {                                     /* ggml-cuda.h:16-27, same field order */
    int     max_gpus;
    int     num_devices;
    int     main_device_id;
    size_t  total_vram;
    size_t  total_free_vram;
    size_t  device_vram_free[GGML_CUDA_MAX_DEVICES];
    size_t  device_vram_total[GGML_CUDA_MAX_DEVICES];
    int64_t device_vram_reserved[GGML_CUDA_MAX_DEVICES];
    struct ggml_hip_device_props device_props[GGML_CUDA_MAX_DEVICES];
} GPUStatus;

/* device enumeration / VRAM accounting (ggml-cuda.cu:1876-2041) */
const GPUStatus * ggml_cuda_get_system_gpu_status(void);
bool   ggml_init_cublas(bool check_only);
void   ggml_cuda_update_gpu_status(int device_id);
void   ggml_cuda_print_gpu_status(const GPUStatus * status, bool print_summary);
/* setters callable before init (falcon_common.cpp:385-460, libfalcon.cpp:1653, 1745) */
void   ggml_cuda_set_max_gpus(int max_gpus);
void   ggml_cuda_set_main_device(int main_device);
void   ggml_cuda_set_vram_reserved(int64_t vram_reserved);
void   ggml_cuda_set_tensor_split_prepare(const float * tensor_split, int num_devices);
void   ggml_cuda_set_tensor_split(const float * tensor_split);
void   ggml_cuda_set_scratch_size(size_t scratch_size);
void   ggml_cuda_free_scratch(void);
/* pinned host memory (llama-util.h:462, 477) */
void * ggml_cuda_host_malloc(size_t size);
void   ggml_cuda_host_free(void * ptr);
/* buffer-pool housekeeping (libfalcon.cpp:4573-4587) */
void   ggml_cuda_pool_reset_all_counters(int device_id);
int    ggml_cuda_pool_purge_buffers_with_access_count(int min_access_count, int device_id);
/* weights (libfalcon.cpp:1251, 533-536) */
void   ggml_cuda_transform_tensor(void * data, struct ggml_tensor * tensor);
void   ggml_cuda_free_data(struct ggml_tensor * tensor);
/* graph-builder offload hooks (libfalcon.cpp:2144-2157, 1377-1380) */
void   ggml_cuda_assign_buffers(struct ggml_tensor * tensor);
void   ggml_cuda_assign_buffers_no_scratch(struct ggml_tensor * tensor);
/* the per-node hook (ggml.c:15779-15790) and the scheduler's question (ggml.c:17412) */
bool   ggml_cuda_compute_forward(struct ggml_compute_params * params, struct ggml_tensor * tensor);
bool   ggml_cuda_can_mul_mat(const struct ggml_tensor * src0, const struct ggml_tensor * src1, struct ggml_tensor * dst);
void   ggml_cuda_mul(const struct ggml_tensor * src0, const struct ggml_tensor * src1, struct ggml_tensor * dst);
/* declared by the reference, never defined or called there */
size_t ggml_cuda_mul_mat_get_wsize(const struct ggml_tensor * src0, const struct ggml_tensor * src1, struct ggml_tensor * dst);
void   ggml_cuda_mul_mat(const struct ggml_tensor * src0, const struct ggml_tensor * src1, struct ggml_tensor * dst, void * wdata, size_t wsize);

#ifdef  __cplusplus
}
#endif
