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
// ---- wave64 butterfly reductions, pairing order xor 1, 2, 4, 8, 16, 32 (ascending; oracle knob orc_set_sum_order(1)
// uses the same association). Steps 1..8 stay inside a 16-lane row and use DPP (register-to-register, no LDS
// crossbar): quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror -- after the first two steps every
// lane of a quad holds the quad's value, so mirror pairings combine exactly the operands xor 4 / xor 8 would. Steps
// 16 and 32 combine the four row results through v_readlane: (r0 + r1) + (r2 + r3).
// FQ_SHFL_REDUCE selects plain __shfl_xor for every step (same association; used by the self-test as the reference).
// (old = 0 with bound_ctrl: every source lane of these patterns exists, so the value is the same, and the compiler may fold
//  the move into the consuming instruction, e.g. v_add_f32_dpp)
template <int CTRL> __device__ __forceinline__ int dpp_i32(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }
template <int CTRL> __device__ __forceinline__ float  dpp_mov(float v)  { return __builtin_bit_cast(float, dpp_i32<CTRL>(__builtin_bit_cast(int, v))); }
template <int CTRL> __device__ __forceinline__ int    dpp_mov(int v)    { return dpp_i32<CTRL>(v); }
template <int CTRL> __device__ __forceinline__ double dpp_mov(double v) {
    const long long b = __builtin_bit_cast(long long, v);
    const unsigned lo = (unsigned) dpp_i32<CTRL>((int)(unsigned) b), hi = (unsigned) dpp_i32<CTRL>((int)(unsigned)(b >> 32));
    return __builtin_bit_cast(double, (long long)(((unsigned long long) hi << 32) | lo));
}
__device__ __forceinline__ float  lane_get(float v, int l)  { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); }
__device__ __forceinline__ int    lane_get(int v, int l)    { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ double lane_get(double v, int l) {
    const long long b = __builtin_bit_cast(long long, v);
    const unsigned lo = (unsigned) __builtin_amdgcn_readlane((int)(unsigned) b, l), hi = (unsigned) __builtin_amdgcn_readlane((int)(unsigned)(b >> 32), l);
    return __builtin_bit_cast(double, (long long)(((unsigned long long) hi << 32) | lo));
}
struct op_add { template <typename T> __device__ __forceinline__ T operator()(T a, T b) const { return a + b; } };
// maxNum of two floats in ONE instruction (fmaxf() adds a canonicalising v_max_f32 x, x per operand)
__device__ __forceinline__ float fq_max_f32(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
struct op_max { __device__ __forceinline__ float operator()(float a, float b) const { return fq_max_f32(a, b); } };

template <typename T, typename OP>
__device__ __forceinline__ T wave_reduce_shfl(T v, OP op) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) v = op(v, __shfl_xor(v, o));
    return v;
}
template <typename T, typename OP>
__device__ __forceinline__ T wave_reduce(T v, OP op) {
#if defined(FQ_SHFL_REDUCE)
    return wave_reduce_shfl(v, op);
#else
    v = op(v, dpp_mov<0xB1>(v));        // quad_perm [1,0,3,2]  == xor 1
    v = op(v, dpp_mov<0x4E>(v));        // quad_perm [2,3,0,1]  == xor 2
    v = op(v, dpp_mov<0x141>(v));       // row_half_mirror      (pairs quad q with quad q^1)
    v = op(v, dpp_mov<0x140>(v));       // row_mirror           (pairs 8-lane half with the other half)
    const T r0 = lane_get(v, 0), r1 = lane_get(v, 16), r2 = lane_get(v, 32), r3 = lane_get(v, 48);
    return op(op(r0, r1), op(r2, r3));
#endif
}
// butterflies over aligned groups of 8 / 16 / 32 lanes (every lane of the group ends with the group's result)
template <typename T, typename OP> __device__ __forceinline__ T reduce8(T v, OP op) {
    v = op(v, dpp_mov<0xB1>(v)); v = op(v, dpp_mov<0x4E>(v)); return op(v, dpp_mov<0x141>(v));
}
template <typename T, typename OP> __device__ __forceinline__ T reduce16(T v, OP op) { v = reduce8(v, op); return op(v, dpp_mov<0x140>(v)); }
template <typename T, typename OP> __device__ __forceinline__ T reduce32(T v, OP op) { v = reduce16(v, op); return op(v, __shfl_xor(v, 16)); }
__device__ __forceinline__ float  wave_sum(float v)  { return wave_reduce(v, op_add()); }
__device__ __forceinline__ double wave_sum(double v) { return wave_reduce(v, op_add()); }
__device__ __forceinline__ int    wave_sum(int v)    { return wave_reduce(v, op_add()); }
__device__ __forceinline__ float  wave_max(float v)  { return wave_reduce(v, op_max()); }

__device__ __forceinline__ uint16_t f2h_bits(float f) { return __builtin_bit_cast(uint16_t, (_Float16) f); }   // RNE, keeps subnormals
__device__ __forceinline__ float    h2f_bits(uint16_t h) { return (float) __builtin_bit_cast(_Float16, h); }

// fp16 EXP "table" without the table: table_exp_f16[h] = fp16(expf(fp32(h))) (ggml.c:4276-4290) recomputed in registers.
// ggml_hip_init compares the recomputation against the host-built table for all 63488 non-NaN inputs and only then lets
// the attention kernels use it (a dependent gather costs a memory round trip, 1-2 us while the chip streams weights).
// Fast path: f32 with a compensated exponent (x log2(e) as a hi + lo pair, v_exp_f32, first-order correction: ~1.5 ulp),
// accepted only where rounding to fp16 gives the same bits 4 f32-ulps to either side; the few inputs that sit closer
// than that to an fp16 rounding boundary are looked up in a short constant list (fq_exp_fix.h). ~20 instructions.
// exp_f16_fast: the value the f32 evaluation gives and whether it DECIDES the table entry: true where rounding to fp16 gives the same bits 4 f32-ulps to either
// side (-16 < x < 11), and for the two ranges whose entries follow from the input's bits alone -- x <= -16 (incl. -inf): the fp16 subnormals 2, 1 or 0
// (exp(-16) = 1.9 x 2^-24, exp(-17.33) = half of 2^-24; round 5: a real prompt's far-away keys all land here), x >= 11.09375 (incl. +inf): +inf.
__device__ __forceinline__ bool exp_f16_fast(uint16_t hbits, uint16_t & out) {
    const float x = h2f_bits(hbits);
    const float L2E_HI = 1.44269502162933349609375f, L2E_LO = 1.925963033500011e-08f;      // log2(e) = hi + lo
    const float t_hi = x * L2E_HI;
    const float t_lo = __builtin_fmaf(x, L2E_HI, -t_hi) + x * L2E_LO;
    const float p = __builtin_amdgcn_exp2f(t_hi);
    const float r = __builtin_fmaf(p * 0.693147182464599609375f, t_lo, p);
    const uint16_t h = f2h_bits(r);
    const bool tiny = hbits >= 0xCC00u && hbits <= 0xFC00u, big = hbits >= 0x498Cu && hbits <= 0x7C00u;
    const uint16_t hr = tiny ? (hbits <= 0xCC0Eu ? (uint16_t) 2 : (hbits <= 0xCC55u ? (uint16_t) 1 : (uint16_t) 0)) : (uint16_t) 0x7C00u;
    const bool robust = f2h_bits(r * 1.00000048f) == h && f2h_bits(r * 0.99999952f) == h && x > -16.0f && x < 11.0f;
    out = (tiny || big) ? hr : h;
    return robust || tiny || big;
}
// The inputs the fast path cannot decide sit within 4 f32-ulps of an fp16 rounding boundary: a FIXED set for this hardware's v_exp_f32 (FQ_EXP_FIX_N inputs of
// 63488; listed by ggml_hip_debug_exp_boundary on an MI355X, scripts/gpu_exp_boundary.py) with their table entries -- a search through a constant list instead
// of an f64 exp() inlined into every soft_max loop (round 5: ~150 instructions and a dozen registers per call site went with it). ggml_hip_init compares formula
// and table for EVERY input before the kernels may use the formula: an input missing from the list (another chip's v_exp_f32) sends them back to the table.
#include "fq_exp_fix.h"
__device__ __forceinline__ uint16_t exp_f16_undecided(uint16_t hbits, uint16_t h) {
#pragma unroll 1
    for (int i = 0; i < FQ_EXP_FIX_N; ++i) if ((fq_exp_fix[i] >> 16) == hbits) h = (uint16_t)(fq_exp_fix[i] & 0xFFFFu);
    return h;
}
__device__ __forceinline__ uint16_t exp_f16_formula(uint16_t hbits) {
    uint16_t h;
    if (__builtin_expect(exp_f16_fast(hbits, h), 1)) return h;
    return exp_f16_undecided(hbits, h);
}
__device__ __forceinline__ float soft_max_exp(const uint16_t * __restrict__ exp_tab, float x) {      // exp_tab == nullptr: verified formula
    const uint16_t hb = f2h_bits(x);
    return h2f_bits(exp_tab ? exp_tab[hb] : exp_f16_formula(hb));
}

// block-wide reductions (<= 16 waves); `scratch` = >= 16 elements of LDS. The per-wave partials are combined in wave
// order 0, 1, 2, ... : each lane fetches one partial with a single LDS read and the chain runs over v_readlane (a loop of
// dependent LDS reads costs ~70 ns per wave, which is most of a LayerNorm at 12 waves).
template <typename T, typename OP>
__device__ __forceinline__ T waves_combine(const T * scratch, int nw, OP op) {      // partials of waves 0..nw-1, in wave order
    const int lane = threadIdx.x & 63;
    const T mine = scratch[lane < nw ? lane : 0];
    T t = lane_get(mine, 0);
    for (int i = 1; i < nw; ++i) t = op(t, lane_get(mine, i));
    return t;
}
template <typename T, typename OP>
__device__ __forceinline__ T block_combine(const T * scratch, OP op) { return waves_combine(scratch, (int)(blockDim.x >> 6), op); }
// the same over a GROUP of gnt threads (a multiple of 64) of a larger workgroup whose groups run in lockstep: gtid is the
// thread's index in its group, scratch the group's own LDS words; the barriers are workgroup barriers
__device__ __forceinline__ double group_sum(double v, double * scratch, int gtid, int gnt) {
    v = wave_sum(v);
    __syncthreads();
    if ((gtid & 63) == 0) scratch[gtid >> 6] = v;
    __syncthreads();
    return waves_combine(scratch, gnt >> 6, op_add());
}
__device__ __forceinline__ float group_max(float v, float * scratch, int gtid, int gnt) {
    v = wave_max(v);
    __syncthreads();
    if ((gtid & 63) == 0) scratch[gtid >> 6] = v;
    __syncthreads();
    return waves_combine(scratch, gnt >> 6, op_max());
}
template <typename T>
__device__ __forceinline__ T block_sum(T v, T * scratch) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) scratch[wid] = v;
    __syncthreads();
    return block_combine(scratch, op_add());
}
__device__ __forceinline__ float block_max(float v, float * scratch) {
    v = wave_max(v);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) scratch[wid] = v;
    __syncthreads();
    return block_combine(scratch, op_max());
}
#endif
