// kernels_wquant.hip -- weight quantizers on the device (gfx950): f32 rows -> ggml blocks, bit-identical with the
// reference's quantize_row_q*_reference (ggml.c:927-1129, k_quants.c:275-844) = what ggml_quantize_chunk
// (ggml.c:19479-19560) writes into a model file. The arithmetic lives in fq_wquant.h (also compiled on the host).
//
//  * k_wquant_legacy  one thread = one 32-weight block; the workgroup's blocks are assembled in LDS and leave as one
//                     contiguous, coalesced store (ggml blocks are 18..34 bytes and only 2-byte aligned)
//  * k_wquant_k       one thread = one sub-block (16 or 32 weights) of a 256-weight super-block: fit -> (LDS) -> one
//                     thread per super-block packs the 4/6/8-bit scales -> every thread re-quantizes its sub-block
//                     with the dequantized scale -> the super-block's bytes are packed cooperatively
//  * k_f16_to_f32 / k_f32_to_f16   ggml_fp16_to_fp32_row / ggml_fp32_to_fp16_row (ggml.c:370-391): an f16 model file's
//                     weights before quantizing, and the F16 output type of ggml_quantize_chunk
//
// The fits are sequential, data-dependent loops (up to 5 refinement passes per sub-block): the kernel is bound by its
// per-thread instruction stream, not by HBM (4 bytes read + ~0.6 written per weight).
//
// Built with -ffp-contract=off.
#include <hip/hip_runtime.h>
#include "fq_wquant.h"
#include "kernels.h"

template <int N>
__device__ __forceinline__ void load_row_piece(const float * __restrict__ p, float (&v)[N]) {
#pragma unroll
    for (int i = 0; i < N; i += 4) {
        const float4 t = *reinterpret_cast<const float4 *>(p + i);
        v[i] = t.x; v[i + 1] = t.y; v[i + 2] = t.z; v[i + 3] = t.w;
    }
}
// n_bytes (even) from LDS to global, 2 bytes per thread and step
__device__ __forceinline__ void store_staged(uint8_t * __restrict__ dst, const uint8_t * src_lds, int n_bytes) {
    uint16_t * d = reinterpret_cast<uint16_t *>(dst);
    const uint16_t * s = reinterpret_cast<const uint16_t *>(src_lds);
    for (int i = threadIdx.x; i < n_bytes / 2; i += blockDim.x) d[i] = s[i];
}

template <int TYPE>
__global__ void __launch_bounds__(256) k_wquant_legacy(const float * __restrict__ x, int64_t nblocks, uint8_t * __restrict__ out,
                                                        unsigned long long * __restrict__ hist) {
    constexpr int TS = fq_desc(TYPE).tsize;
    __shared__ __attribute__((aligned(16))) uint8_t s_out[256 * TS];
    __shared__ int s_hist[16];
    if (threadIdx.x < 16) s_hist[threadIdx.x] = 0;
    __syncthreads();
    const bool want_hist = hist != nullptr;
    for (int64_t b0 = (int64_t) blockIdx.x * 256; b0 < nblocks; b0 += (int64_t) gridDim.x * 256) {
        const int64_t b = b0 + threadIdx.x;
        if (b < nblocks) {
            float v[32];
            load_row_piece<32>(x + 32 * b, v);
            wq_block_legacy<TYPE>(v, s_out + threadIdx.x * TS, [&](int bin) { if (want_hist) atomicAdd(&s_hist[bin], 1); });
        }
        __syncthreads();
        const int64_t left = nblocks - b0;
        store_staged(out + (size_t) b0 * TS, s_out, (int)(left < 256 ? left : 256) * TS);
        __syncthreads();
    }
    if (want_hist && threadIdx.x < 16 && s_hist[threadIdx.x]) atomicAdd(&hist[threadIdx.x], (unsigned long long) s_hist[threadIdx.x]);
}

template <int TYPE>
__global__ void __launch_bounds__(256) k_wquant_k(const float * __restrict__ x, int64_t nsb, uint8_t * __restrict__ out) {
    constexpr int N = wq_geom<TYPE>::N, NSB = wq_geom<TYPE>::NSB, SBW = 256 / NSB, TS = fq_desc(TYPE).tsize;
    __shared__ float s_scale[256], s_min[256];
    __shared__ uint8_t s_hdr[SBW][20];
    __shared__ uint8_t s_L[SBW][256];
    __shared__ __attribute__((aligned(16))) uint8_t s_out[SBW * TS];
    const int sbl = threadIdx.x / NSB, j = threadIdx.x % NSB;
    for (int64_t sb0 = (int64_t) blockIdx.x * SBW; sb0 < nsb; sb0 += (int64_t) gridDim.x * SBW) {
        const int64_t sb = sb0 + sbl < nsb ? sb0 + sbl : nsb - 1;   // surplus threads redo the last super-block, never stored
        float v[N];
        load_row_piece<N>(x + 256 * sb + N * j, v);
        int L[N];
        float sc, mn;
        wq_fit<TYPE>(v, L, sc, mn);
        s_scale[threadIdx.x] = sc; s_min[threadIdx.x] = mn;
        __syncthreads();
        if (j == 0) wq_header<TYPE>(s_scale + sbl * NSB, s_min + sbl * NSB, s_hdr[sbl]);
        __syncthreads();
        wq_requant<TYPE>(s_hdr[sbl], j, v, L);
#pragma unroll
        for (int i = 0; i < N; ++i) s_L[sbl][N * j + i] = (uint8_t) L[i];
        __syncthreads();
        for (int i = j; i < TS; i += NSB) s_out[sbl * TS + i] = wq_pack_byte<TYPE>(s_hdr[sbl], s_L[sbl], i);
        __syncthreads();
        const int64_t left = nsb - sb0;
        store_staged(out + (size_t) sb0 * TS, s_out, (int)(left < SBW ? left : SBW) * TS);
        __syncthreads();
    }
}

__global__ void k_f16_to_f32(const uint16_t * __restrict__ src, float * __restrict__ dst, int64_t n) {
    for (int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t) gridDim.x * blockDim.x) dst[i] = fq_h2f(src[i]);
}

__global__ void k_f32_to_f16(const float * __restrict__ src, uint16_t * __restrict__ dst, int64_t n) {
    for (int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t) gridDim.x * blockDim.x) dst[i] = fq_f2h(src[i]);
}

template <int TYPE>
static void launch_legacy(const float * x, int64_t nblocks, uint8_t * out, unsigned long long * hist, hipStream_t st) {
    const int64_t wgs = (nblocks + 255) / 256;
    hipLaunchKernelGGL(k_wquant_legacy<TYPE>, dim3((unsigned)(wgs > 16384 ? 16384 : wgs)), dim3(256), 0, st, x, nblocks, out, hist);
}
template <int TYPE>
static void launch_k(const float * x, int64_t nsb, uint8_t * out, hipStream_t st) {
    constexpr int SBW = 256 / wq_geom<TYPE>::NSB;
    const int64_t wgs = (nsb + SBW - 1) / SBW;
    hipLaunchKernelGGL(k_wquant_k<TYPE>, dim3((unsigned)(wgs > 16384 ? 16384 : wgs)), dim3(256), 0, st, x, nsb, out);
}

// n_elems f32 (a whole number of blocks) -> ggml blocks at out; hist: 16 counters (legacy formats only) or nullptr
bool fq_launch_wquant(int type, const float * x, int64_t n_elems, uint8_t * out, unsigned long long * hist, hipStream_t st) {
    FQ_TL(st, "wquant");
    if (n_elems <= 0) return true;
    switch (type) {
        case FQ_Q4_0: launch_legacy<FQ_Q4_0>(x, n_elems / 32, out, hist, st); return true;
        case FQ_Q4_1: launch_legacy<FQ_Q4_1>(x, n_elems / 32, out, hist, st); return true;
        case FQ_Q5_0: launch_legacy<FQ_Q5_0>(x, n_elems / 32, out, hist, st); return true;
        case FQ_Q5_1: launch_legacy<FQ_Q5_1>(x, n_elems / 32, out, hist, st); return true;
        case FQ_Q8_0: launch_legacy<FQ_Q8_0>(x, n_elems / 32, out, hist, st); return true;
        case FQ_Q2_K: launch_k<FQ_Q2_K>(x, n_elems / 256, out, st); return true;
        case FQ_Q3_K: launch_k<FQ_Q3_K>(x, n_elems / 256, out, st); return true;
        case FQ_Q4_K: launch_k<FQ_Q4_K>(x, n_elems / 256, out, st); return true;
        case FQ_Q5_K: launch_k<FQ_Q5_K>(x, n_elems / 256, out, st); return true;
        case FQ_Q6_K: launch_k<FQ_Q6_K>(x, n_elems / 256, out, st); return true;
    }
    return false;
}
void fq_launch_f16_to_f32(const uint16_t * src, float * dst, int64_t n, hipStream_t st) {
    FQ_TL(st, "f16_to_f32");
    if (n <= 0) return;
    const int64_t wgs = (n + 255) / 256;
    hipLaunchKernelGGL(k_f16_to_f32, dim3((unsigned)(wgs > 8192 ? 8192 : wgs)), dim3(256), 0, st, src, dst, n);
}
void fq_launch_f32_to_f16(const float * src, uint16_t * dst, int64_t n, hipStream_t st) {
    FQ_TL(st, "f32_to_f16");
    if (n <= 0) return;
    const int64_t wgs = (n + 255) / 256;
    hipLaunchKernelGGL(k_f32_to_f16, dim3((unsigned)(wgs > 8192 ? 8192 : wgs)), dim3(256), 0, st, src, dst, n);
}
