// rccl_dyn.h -- RCCL bound at run time: dlopen of librccl.so.1 (the copy already in the process if there is one, e.g. PyTorch's)
// at the first multi-GPU call, so that single-GPU users of libggml_hip.so need no RCCL at all. Shared by the layer pipeline
// (falcon_pipeline.hip) and the row-split tensor parallelism (split_tp.hip); the loader itself lives in falcon_pipeline.hip.
#pragma once
// The few RCCL types this library touches, declared here so that it builds on a ROCm install without the RCCL development
// headers (values as in rccl/rccl.h of ROCm 7.x; the library itself is only dlopen'ed): the real header is used when present.
#if defined(__has_include)
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#define FQ_HAVE_RCCL_H 1
#endif
#endif
#ifndef FQ_HAVE_RCCL_H
typedef struct ncclComm * ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4, ncclInvalidUsage = 5 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5, ncclFloat16 = 6, ncclFloat32 = 7, ncclFloat64 = 8 } ncclDataType_t;
#endif

struct rccl_api {
    void * lib = nullptr;
    ncclResult_t (*ncclGetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*ncclCommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*ncclCommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*ncclCommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*ncclSend)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*ncclRecv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*ncclAllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*ncclGroupStart)() = nullptr;
    ncclResult_t (*ncclGroupEnd)() = nullptr;
    const char * (*ncclGetErrorString)(ncclResult_t) = nullptr;
};
rccl_api * fq_rccl();          // nullptr (after a message on stderr) when RCCL cannot be loaded

#define RCCL_CHECK(call) do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) { \
    fprintf(stderr, "ggml-hip: %s failed: %s (%s:%d)\n", #call, fq_rccl()->ncclGetErrorString(r_), __FILE__, __LINE__); abort(); } } while (0)
