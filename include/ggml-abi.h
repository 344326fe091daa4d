/*
 * ggml-abi.h -- the slice of ggml.h's ABI that crosses the backend boundary, restated so that libggml_hip.so can be
 * built WITHOUT the reference tree. Field order, types and sizes mirror cmp-nct/ggllm.cpp ggml.h:
 *   enum ggml_type (247-268), enum ggml_backend (269-273), enum ggml_op (300-366), tensor_meta (385-403),
 *   struct ggml_tensor (421-459), enum ggml_task_type / struct ggml_compute_params (500-516).
 * The offsets are pinned by static_asserts below; tests/test_dropin_link.py re-derives them from the reference's own
 * header when /root/reference is present. When this header is used INSIDE a ggml tree, include the real ggml.h instead
 * (include/dropin/ggml-cuda.h does that).
 */
#ifndef GGML_ABI_H
#define GGML_ABI_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GGML_MAX_DIMS 4
#define GGML_MAX_OPT  4
#define GGML_MAX_NAME 64

enum ggml_type {
    GGML_TYPE_F32 = 0, GGML_TYPE_F16 = 1, GGML_TYPE_Q4_0 = 2, GGML_TYPE_Q4_1 = 3, GGML_TYPE_Q5_0 = 6, GGML_TYPE_Q5_1 = 7,
    GGML_TYPE_Q8_0 = 8, GGML_TYPE_Q8_1 = 9, GGML_TYPE_Q2_K = 10, GGML_TYPE_Q3_K = 11, GGML_TYPE_Q4_K = 12, GGML_TYPE_Q5_K = 13,
    GGML_TYPE_Q6_K = 14, GGML_TYPE_Q8_K = 15, GGML_TYPE_I8, GGML_TYPE_I16, GGML_TYPE_I32, GGML_TYPE_COUNT,
};
enum ggml_backend { GGML_BACKEND_CPU = 0, GGML_BACKEND_GPU = 10, GGML_BACKEND_GPU_SPLIT = 20 };

/* only the operators the backend looks at are named; values are positions in the reference enum */
enum ggml_op {
    GGML_OP_NONE = 0, GGML_OP_ADD = 2, GGML_OP_MUL = 6, GGML_OP_REPEAT = 14, GGML_OP_GELU = 23, GGML_OP_NORM = 27,
    GGML_OP_MUL_MAT = 30, GGML_OP_SCALE = 32, GGML_OP_SET = 33, GGML_OP_CPY = 34, GGML_OP_CONT = 35, GGML_OP_RESHAPE = 36,
    GGML_OP_VIEW = 37, GGML_OP_PERMUTE = 38, GGML_OP_TRANSPOSE = 39, GGML_OP_GET_ROWS = 40, GGML_OP_DIAG_MASK_INF = 43,
    GGML_OP_SOFT_MAX = 45, GGML_OP_ROPE = 47,
    GGML_OP_ABI_FORCE_INT = 0x7fffffff
};

typedef struct {
    int8_t  layer_id;
    char    short_name[GGML_MAX_NAME];
    int8_t  cuda_op_directive;          /* -1 default, 0 never on the device, 1 force (ggml.h:391) */
    int8_t  cuda_info_op_on_device;
    uint8_t cuda_perf_mal_mul_type;     /* 0 none, 1 quantized kernel, 16/32 fp16/fp32 BLAS (ggml.h:394) */
    float   f_custom[4];
    int     i_custom[4];
    uint8_t debug_flag;
    char    padding[15];
} tensor_meta;

struct ggml_tensor {
    enum ggml_type    type;
    enum ggml_backend backend;
    int     n_dims;
    int64_t ne[GGML_MAX_DIMS];
    size_t  nb[GGML_MAX_DIMS];
    enum ggml_op op;
    bool    is_param;
    struct ggml_tensor * grad;
    struct ggml_tensor * src0;
    struct ggml_tensor * src1;
    struct ggml_tensor * opt[GGML_MAX_OPT];
    int     n_tasks;
    int     perf_runs;
    int64_t perf_cycles;
    int64_t perf_time_us;
    void *  data;
    char    name[GGML_MAX_NAME];
    void *  extra;
    tensor_meta meta;
    char    padding[4];
};

enum ggml_task_type { GGML_TASK_INIT = 0, GGML_TASK_COMPUTE, GGML_TASK_FINALIZE };
struct ggml_compute_params {
    enum ggml_task_type type;
    int    ith, nth;
    size_t wsize;
    void * wdata;
};

#ifdef __cplusplus
}
static_assert(offsetof(ggml_tensor, ne) == 16 && offsetof(ggml_tensor, nb) == 48 && offsetof(ggml_tensor, op) == 80, "ggml_tensor ABI");
static_assert(offsetof(ggml_tensor, src0) == 96 && offsetof(ggml_tensor, src1) == 104 && offsetof(ggml_tensor, data) == 168, "ggml_tensor ABI");
static_assert(offsetof(ggml_tensor, name) == 176 && offsetof(ggml_tensor, extra) == 240 && offsetof(ggml_tensor, meta) == 248, "ggml_tensor ABI");
static_assert(sizeof(ggml_tensor) == 368 && sizeof(tensor_meta) == 116 && offsetof(tensor_meta, cuda_op_directive) == 65, "ggml_tensor ABI");
static_assert(offsetof(tensor_meta, cuda_perf_mal_mul_type) == 67 && offsetof(tensor_meta, f_custom) == 68 && offsetof(tensor_meta, i_custom) == 84, "tensor_meta ABI");
static_assert(sizeof(ggml_compute_params) == 32 && offsetof(ggml_compute_params, wsize) == 16, "ggml_compute_params ABI");
#endif
#endif
