// fq_device.h -- small device-side helpers (gfx950, wave64) and the host-side HIP error check.
#pragma once
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "fq_types.h"

// Errors are fatal, as in the reference backend (CUDA_CHECK -> exit(1), ggml-cuda.cu:22-51).
#define HIP_CHECK(expr)                                                                              \
    do {                                                                                             \
        hipError_t e_ = (expr);                                                                      \
        if (e_ != hipSuccess) {                                                                      \
            fprintf(stderr, "ggml-hip: HIP error %d (%s) at %s:%d: %s\n", (int) e_, hipGetErrorString(e_), \
                    __FILE__, __LINE__, #expr);                                                      \
            exit(1);                                                                                 \
        }                                                                                            \
    } while (0)

#define FQ_WAVE 64

#if defined(__HIPCC__)
__device__ __forceinline__ float  wave_sum(float v)  {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o); return v; }
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o); return v; }
__device__ __forceinline__ float  wave_max(float v)  {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o)); return v; }

__device__ __forceinline__ uint16_t f2h_bits(float f) { return __builtin_bit_cast(uint16_t, (_Float16) f); }   // RNE, keeps subnormals
__device__ __forceinline__ float    h2f_bits(uint16_t h) { return (float) __builtin_bit_cast(_Float16, h); }

// block-wide reductions for 256-thread blocks (4 waves); `scratch` = >= 4 elements of LDS
template <typename T>
__device__ __forceinline__ T block_sum(T v, T * scratch) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if (lane == 0) scratch[wid] = v;
    __syncthreads();
    T t = scratch[0];
    for (int i = 1; i < nw; ++i) t += scratch[i];
    return t;
}
__device__ __forceinline__ float block_max(float v, float * scratch) {
    v = wave_max(v);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if (lane == 0) scratch[wid] = v;
    __syncthreads();
    float t = scratch[0];
    for (int i = 1; i < nw; ++i) t = fmaxf(t, scratch[i]);
    return t;
}
#endif
