// fq_block_dev.h -- device building blocks shared by the stand-alone kernels (kernels_quant.hip, kernels_block.hip)
// and by the fused decode kernels (kernels_decode.hip): identical arithmetic => bit-identical results on both paths.
#pragma once
#include "fq_device.h"
#include "fq_units.h"

// pointers into one activation image (fq_types.h), in global memory or LDS
struct act_image_ptr { int8_t * qs; float * d; uint8_t * aux; };

__device__ __forceinline__ act_image_ptr act_image_at(uint8_t * base, int act_type, int64_t K) {
    return { (int8_t *) base, (float *)(base + fq_act_d_off(act_type, K)), base + fq_act_aux_off(act_type, K) };
}

// roundf (round half away from zero) in 4 VALU ops: v - trunc(v) is exact, so the comparison with 0.5 is exact too
__device__ __forceinline__ int round_half_away(float v) {
    const float t = truncf(v);
    return (int) t + (fabsf(v - t) >= 0.5f ? (v < 0.0f ? -1 : 1) : 0);
}

// ---- Q8_0 / Q8_1 (ggml.c:1106-1129, 1292-1325): one thread = 4 consecutive elements (quad q4 of the row), the 8
// threads of a 32-block are 8 consecutive lanes. All 8 lanes of a group must call together.
template <int ACT>
__device__ __forceinline__ void quant_q8_quad(const float4 v, int64_t q4, const act_image_ptr & o, bool live) {
    float amax = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
    amax = reduce8(amax, op_max());
    const float d  = amax / 127.0f;                       // ggml.c:1116 / 1302
    const float id = d ? 1.0f / d : 0.0f;
    const int q0 = round_half_away(v.x * id), q1 = round_half_away(v.y * id), q2 = round_half_away(v.z * id), q3 = round_half_away(v.w * id);
    int s = q0 + q1 + q2 + q3;
    s = reduce8(s, op_add());
    if (live) {
        *(uint32_t *)(o.qs + 4 * q4) = (uint32_t)(q0 & 0xFF) | ((uint32_t)(q1 & 0xFF) << 8) | ((uint32_t)(q2 & 0xFF) << 16) | ((uint32_t)(q3 & 0xFF) << 24);
        if ((q4 & 7) == 0) {
            const int64_t b = q4 >> 3;
            if constexpr (ACT == FQ_Q8_0) {
                o.d[b] = h2f_bits(f2h_bits(d));                           // the block stores d as fp16 (ggml.c:1120)
                ((int32_t *) o.aux)[b] = s;
            } else {
                o.d[b] = d;                                               // ggml.c:1306
                ((float *) o.aux)[b] = (float) s * d;                     // ggml.c:1323  y.s = sum*d
            }
        }
    }
}

// ---- Q8_K (k_quants.c:899-934): one WAVE = one 256-element super-block sb, lane l holds elements 4l..4l+3.
__device__ __forceinline__ void quant_q8K_wave(const float4 v, int lane, int64_t sb, const act_image_ptr & o) {
    // element of largest magnitude, FIRST one on ties (strict '>' scan, k_quants.c:906-911)
    float ax = fabsf(v.x), mx = v.x; int idx = 4 * lane;
    if (fabsf(v.y) > ax) { ax = fabsf(v.y); mx = v.y; idx = 4 * lane + 1; }
    if (fabsf(v.z) > ax) { ax = fabsf(v.z); mx = v.z; idx = 4 * lane + 2; }
    if (fabsf(v.w) > ax) { ax = fabsf(v.w); mx = v.w; idx = 4 * lane + 3; }
    // over the 64 lanes: the largest magnitude by a plain max butterfly, then the LOWEST lane that holds it (elements are in lane order, and each
    // lane kept its first on ties) hands out its signed value -- the strict '>' scan's winner with a third of the instructions of a
    // (magnitude, value, index) butterfly
    const float amax = wave_max(ax);
    const unsigned long long holders = __ballot(ax == amax);
    const int first = __builtin_amdgcn_readfirstlane(holders ? (int) __builtin_ctzll(holders) : 0);      // (no holder: a NaN in the block, the result is garbage either way)
    mx = lane_get(mx, first);
    ax = amax;
    (void) idx;
    int8_t  * qo = o.qs + 256 * sb + 4 * lane;
    int16_t * bs = (int16_t *) o.aux + 16 * sb;
    if (ax == 0.0f) {
        *(uint32_t *) qo = 0u;
        if ((lane & 3) == 0) bs[lane >> 2] = 0;
        if (lane == 0) o.d[sb] = 0.0f;
        return;
    }
    const float iscale = -128.0f / mx;
    int q0 = (int) __builtin_rintf(iscale * v.x), q1 = (int) __builtin_rintf(iscale * v.y);     // nearest_int: round-half-even (k_quants.c:50-55)
    int q2 = (int) __builtin_rintf(iscale * v.z), q3 = (int) __builtin_rintf(iscale * v.w);
    q0 = q0 > 127 ? 127 : q0; q1 = q1 > 127 ? 127 : q1; q2 = q2 > 127 ? 127 : q2; q3 = q3 > 127 ? 127 : q3;
    *(uint32_t *) qo = (uint32_t)(q0 & 0xFF) | ((uint32_t)(q1 & 0xFF) << 8) | ((uint32_t)(q2 & 0xFF) << 16) | ((uint32_t)(q3 & 0xFF) << 24);
    int s = q0 + q1 + q2 + q3;
    s += dpp_mov<0xB1>(s); s += dpp_mov<0x4E>(s);
    if ((lane & 3) == 0) bs[lane >> 2] = (int16_t) s;
    if (lane == 0) o.d[sb] = 1.0f / iscale;
}

// N super-blocks at once (sb[n] < 0: none): the same arithmetic per super-block, the N max-butterflies -- chains of dependent cross-lane steps -- advance
// side by side, so a wave that owns several super-blocks (the ring forms' prologues: 3 per wave, two norms) pays the chain's latency once
template <int N>
__device__ __forceinline__ void quant_q8K_wave_n(const float4 (&v)[N], int lane, const int64_t (&sb)[N], const act_image_ptr (&o)[N]) {
    float ax[N], mx[N]; int idx[N];
#pragma unroll
    for (int n = 0; n < N; ++n) {
        ax[n] = fabsf(v[n].x); mx[n] = v[n].x; idx[n] = 4 * lane;
        if (fabsf(v[n].y) > ax[n]) { ax[n] = fabsf(v[n].y); mx[n] = v[n].y; idx[n] = 4 * lane + 1; }
        if (fabsf(v[n].z) > ax[n]) { ax[n] = fabsf(v[n].z); mx[n] = v[n].z; idx[n] = 4 * lane + 2; }
        if (fabsf(v[n].w) > ax[n]) { ax[n] = fabsf(v[n].w); mx[n] = v[n].w; idx[n] = 4 * lane + 3; }
    }
#pragma unroll
    for (int n = 0; n < N; ++n) {                                          // (see quant_q8K_wave: max butterfly, then the lowest holder's signed value)
        const float amax = wave_max(ax[n]);
        const unsigned long long holders = __ballot(ax[n] == amax);
        const int first = __builtin_amdgcn_readfirstlane(holders ? (int) __builtin_ctzll(holders) : 0);      // (no holder: a NaN in the block, the result is garbage either way)
        mx[n] = lane_get(mx[n], first);
        ax[n] = amax;
        (void) idx[n];
    }
#pragma unroll
    for (int n = 0; n < N; ++n) {
        if (sb[n] < 0) continue;                                           // wave-uniform
        int8_t  * qo = o[n].qs + 256 * sb[n] + 4 * lane;
        int16_t * bs = (int16_t *) o[n].aux + 16 * sb[n];
        if (ax[n] == 0.0f) {
            *(uint32_t *) qo = 0u;
            if ((lane & 3) == 0) bs[lane >> 2] = 0;
            if (lane == 0) o[n].d[sb[n]] = 0.0f;
            continue;
        }
        const float iscale = -128.0f / mx[n];
        int q0 = (int) __builtin_rintf(iscale * v[n].x), q1 = (int) __builtin_rintf(iscale * v[n].y);
        int q2 = (int) __builtin_rintf(iscale * v[n].z), q3 = (int) __builtin_rintf(iscale * v[n].w);
        q0 = q0 > 127 ? 127 : q0; q1 = q1 > 127 ? 127 : q1; q2 = q2 > 127 ? 127 : q2; q3 = q3 > 127 ? 127 : q3;
        *(uint32_t *) qo = (uint32_t)(q0 & 0xFF) | ((uint32_t)(q1 & 0xFF) << 8) | ((uint32_t)(q2 & 0xFF) << 16) | ((uint32_t)(q3 & 0xFF) << 24);
        int s = q0 + q1 + q2 + q3;
        s += dpp_mov<0xB1>(s); s += dpp_mov<0x4E>(s);
        if ((lane & 3) == 0) bs[lane >> 2] = (int16_t) s;
        if (lane == 0) o[n].d[sb[n]] = 1.0f / iscale;
    }
}

// quantize one f32 row of length K held in LDS (or global) into an image, by a whole 256-thread workgroup
template <int ACT>
__device__ __forceinline__ void quantize_row_block(const float * __restrict__ row, int64_t K, const act_image_ptr & o) {
    const int tid = threadIdx.x;
    if constexpr (ACT == FQ_Q8_K) {
        const int lane = tid & 63;
        for (int64_t sb = tid >> 6; sb < (K >> 8); sb += (blockDim.x >> 6))
            quant_q8K_wave(*(const float4 *)(row + 256 * sb + 4 * lane), lane, sb, o);
    } else {
        const int64_t quads = K >> 2;
        for (int64_t q4 = tid; q4 < ((quads + 63) & ~(int64_t) 63); q4 += blockDim.x) {
            const bool live = q4 < quads;
            quant_q8_quad<ACT>(*(const float4 *)(row + 4 * (live ? q4 : quads - 1)), live ? q4 : quads - 1, o, live);
        }
    }
}

// ---- layer norm of one row by a 256-thread workgroup (ggml.c:10540-10594 + libfalcon.cpp:2166-2188).
// x: global, n % 4 == 0. Result (norm * w + b, or plain norm when w == nullptr) is left in `row` (LDS, n floats).
// Split in two so that a fused kernel can issue the row's loads FIRST, then its weight-stream loads, and only then
// wait for the row (vmcnt counts in order: loads issued after the row do not delay it).
template <int NLN> struct ln_row_regs { float4 t[NLN]; };       // NLN * blockDim float4 of the row live in registers

template <int NLN>
__device__ __forceinline__ void layer_norm_issue(const float * __restrict__ x, int64_t n, ln_row_regs<NLN> & r) {
    const int tid = threadIdx.x, nt = blockDim.x;
    const int64_t nv = n >> 2;
#pragma unroll
    for (int k = 0; k < NLN; ++k) { const int64_t i = (int64_t) k * nt + tid; r.t[k] = ((const float4 *) x)[i < nv ? i : nv - 1]; }
}

// the rest of the norm once the row is in LDS (element i written by the thread that reads it here: i = tid, tid + nt, ..) and every thread holds the f64 sum of
// ITS elements, added in ascending i: mean, deviations, variance, scale, * w + b (w == nullptr: the plain norm) -> out (LDS; may be row). Ends in a workgroup barrier.
// out != row: row keeps (x - mean) * scale, so that a second norm of the same row (Falcon-40B's ln_attn beside ln_mlp) only repeats the last step (layer_norm_apply).
__device__ __forceinline__ void layer_norm_from_row(double s, int64_t n, const float * __restrict__ w, const float * __restrict__ b, float * row, float * out, double * red) {
    const int tid = threadIdx.x, nt = blockDim.x;
    const int64_t nv = n >> 2;
    s = block_sum(s, red);
    const float mean = (float)(s / (double) n);
    double s2 = 0.0;
    for (int64_t i = tid; i < nv; i += nt) {
        float4 v = ((float4 *) row)[i];
        v.x -= mean; v.y -= mean; v.z -= mean; v.w -= mean;
        ((float4 *) row)[i] = v;
        s2 += (double)(v.x * v.x); s2 += (double)(v.y * v.y); s2 += (double)(v.z * v.z); s2 += (double)(v.w * v.w);
    }
    s2 = block_sum(s2, red);
    const float variance = (float)(s2 / (double) n);
    const float scale = 1.0f / sqrtf(variance + 1e-5f);
    for (int64_t i = tid; i < nv; i += nt) {
        float4 v = ((float4 *) row)[i];
        v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
        if (out != row) ((float4 *) row)[i] = v;
        if (w) {
            const float4 ww = ((const float4 *) w)[i], bb = ((const float4 *) b)[i];
            v.x = v.x * ww.x + bb.x; v.y = v.y * ww.y + bb.y; v.z = v.z * ww.z + bb.z; v.w = v.w * ww.w + bb.w;
        }
        ((float4 *) out)[i] = v;
    }
    __syncthreads();
}
// out = row * w + b for a row that holds (x - mean) * scale (layer_norm_from_row with out != row). Ends in a workgroup barrier; the caller puts one in FRONT of it when
// `out` is still being read.
__device__ __forceinline__ void layer_norm_apply(int64_t n, const float * __restrict__ w, const float * __restrict__ b, const float * row, float * out) {
    const int tid = threadIdx.x, nt = blockDim.x;
    const int64_t nv = n >> 2;
    for (int64_t i = tid; i < nv; i += nt) {
        float4 v = ((const float4 *) row)[i];
        const float4 ww = ((const float4 *) w)[i], bb = ((const float4 *) b)[i];
        v.x = v.x * ww.x + bb.x; v.y = v.y * ww.y + bb.y; v.z = v.z * ww.z + bb.z; v.w = v.w * ww.w + bb.w;
        ((float4 *) out)[i] = v;
    }
    __syncthreads();
}

template <int NLN>
__device__ __forceinline__ void layer_norm_finish(const ln_row_regs<NLN> & r, const float * __restrict__ x, int64_t n,
                                                  const float * __restrict__ w, const float * __restrict__ b, float * row, double * red) {
    const int tid = threadIdx.x, nt = blockDim.x;
    const int64_t nv = n >> 2;
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < NLN; ++k) {
        const int64_t i = (int64_t) k * nt + tid;
        if (i < nv) { ((float4 *) row)[i] = r.t[k]; s += (double) r.t[k].x; s += (double) r.t[k].y; s += (double) r.t[k].z; s += (double) r.t[k].w; }
    }
    for (int64_t base = (int64_t) NLN * nt; base < nv; base += 8 * nt) {   // the part of the row that did not fit the registers
        float4 t[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { const int64_t i = base + (int64_t) k * nt + tid; t[k] = ((const float4 *) x)[i < nv ? i : nv - 1]; }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int64_t i = base + (int64_t) k * nt + tid;
            if (i < nv) { ((float4 *) row)[i] = t[k]; s += (double) t[k].x; s += (double) t[k].y; s += (double) t[k].z; s += (double) t[k].w; }
        }
    }
    layer_norm_from_row(s, n, w, b, row, row, red);
}

// ---- LayerNorm + activation quantizer of one row entirely in registers (the fused decode prologue), run by the first
// LNW waves of the workgroup only (one per SIMD for Falcon-7B: the prologue is instruction-issue bound, and the other
// waves are busy requesting weights). Thread t < nt = 64 LNW holds float4 number k*nt + t of the row (k < NLN), so the
// 8 lanes of a Q8 block / the wave of a Q8_K super-block already sit next to each other and the f32 row never visits
// LDS. Same arithmetic and per-thread order as layer_norm_finish + quantize_row_block. Requires n/4 <= NLN * nt.
// Three stages, the CALLER puts a workgroup barrier between them (every wave must reach it). red: 32 doubles of LDS.
template <int NLN>
__device__ __forceinline__ void ln_regs_issue(const float * __restrict__ x, const float * __restrict__ w, const float * __restrict__ b,
                                              int64_t n, int nt, ln_row_regs<NLN> & xr, ln_row_regs<NLN> & wr, ln_row_regs<NLN> & br, int tid_ = -1) {
    const int tid = tid_ < 0 ? (int) threadIdx.x : tid_;      // tid_: index among the nt threads that run the LayerNorm
    const int64_t nv = n >> 2;
#pragma unroll
    for (int k = 0; k < NLN; ++k) { const int64_t i = (int64_t) k * nt + tid; xr.t[k] = ((const float4 *) x)[i < nv ? i : nv - 1]; }
#pragma unroll
    for (int k = 0; k < NLN; ++k) {
        const int64_t i = (int64_t) k * nt + tid, j = i < nv ? i : nv - 1;
        wr.t[k] = ((const float4 *) w)[j]; br.t[k] = ((const float4 *) b)[j];
    }
}
// the same row fetched from a hand-off buffer instead: one 8-byte {epoch tag, f32 bits} granule per element, written by
// other workgroups of the SAME launch with agent-scope stores; every wave re-reads its granules until all carry `epoch`
// (bounded; *err is set if it gives up). w and b are plain loads and can be requested before this.
template <int NLN>
__device__ __forceinline__ void ln_regs_issue_wb(const float * __restrict__ w, const float * __restrict__ b, int64_t n, int nt,
                                                 ln_row_regs<NLN> & wr, ln_row_regs<NLN> & br, int tid_ = -1) {
    const int tid = tid_ < 0 ? (int) threadIdx.x : tid_;
    const int64_t nv = n >> 2;
#pragma unroll
    for (int k = 0; k < NLN; ++k) {
        const int64_t i = (int64_t) k * nt + tid, j = i < nv ? i : nv - 1;
        wr.t[k] = ((const float4 *) w)[j]; br.t[k] = ((const float4 *) b)[j];
    }
}
template <int NLN>
__device__ __forceinline__ void ln_regs_sweep_x(const unsigned long long * gran, unsigned epoch, int64_t n, int nt, ln_row_regs<NLN> & xr, unsigned * err) {
    const int tid = threadIdx.x;
    const int64_t nv = n >> 2;
#pragma unroll
    for (int k = 0; k < NLN; ++k) {
        const int64_t i = (int64_t) k * nt + tid, j = i < nv ? i : nv - 1;
        const unsigned long long * g = gran + 4 * j;
        unsigned v0, v1, v2, v3;
        for (unsigned spins = 0;; ++spins) {
            const unsigned long long a0 = __hip_atomic_load(g + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), a1 = __hip_atomic_load(g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long a2 = __hip_atomic_load(g + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), a3 = __hip_atomic_load(g + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            v0 = (unsigned) a0; v1 = (unsigned) a1; v2 = (unsigned) a2; v3 = (unsigned) a3;
            const bool ok = (unsigned)(a0 >> 32) == epoch && (unsigned)(a1 >> 32) == epoch && (unsigned)(a2 >> 32) == epoch && (unsigned)(a3 >> 32) == epoch;
            if (__all(ok)) break;
            if (spins > (1u << 20)) { if ((tid & 63) == 0) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            __builtin_amdgcn_s_sleep(4);
        }
        xr.t[k] = make_float4(__builtin_bit_cast(float, v0), __builtin_bit_cast(float, v1), __builtin_bit_cast(float, v2), __builtin_bit_cast(float, v3));
    }
}

template <int NLN>
__device__ __forceinline__ void ln_regs_stage1(const ln_row_regs<NLN> & r, int64_t n, int nt, double * red, int tid_ = -1) {
    const int tid = tid_ < 0 ? (int) threadIdx.x : tid_;
    const int64_t nv = n >> 2;
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < NLN; ++k) {
        const int64_t i = (int64_t) k * nt + tid;
        if (i < nv) { s += (double) r.t[k].x; s += (double) r.t[k].y; s += (double) r.t[k].z; s += (double) r.t[k].w; }
    }
    s = wave_sum(s);
    if ((tid & 63) == 0) red[tid >> 6] = s;
}
template <int NLN>
__device__ __forceinline__ void ln_regs_stage2(ln_row_regs<NLN> & r, int64_t n, int nt, double * red, int tid_ = -1) {
    const int tid = tid_ < 0 ? (int) threadIdx.x : tid_;
    const int64_t nv = n >> 2;
    const double s = waves_combine(red, nt >> 6, op_add());
    const float mean = (float)(s / (double) n);
    double s2 = 0.0;
#pragma unroll
    for (int k = 0; k < NLN; ++k) {
        const int64_t i = (int64_t) k * nt + tid;
        float4 v = r.t[k];
        v.x -= mean; v.y -= mean; v.z -= mean; v.w -= mean;
        r.t[k] = v;
        if (i < nv) { s2 += (double)(v.x * v.x); s2 += (double)(v.y * v.y); s2 += (double)(v.z * v.z); s2 += (double)(v.w * v.w); }
    }
    s2 = wave_sum(s2);
    if ((tid & 63) == 0) red[16 + (tid >> 6)] = s2;
}
template <int ACT, int NLN>
__device__ __forceinline__ void ln_regs_stage3(const ln_row_regs<NLN> & r, const ln_row_regs<NLN> & wr, const ln_row_regs<NLN> & br,
                                               int64_t n, int nt, const act_image_ptr & o, const double * red, int tid_ = -1) {
    const int tid = tid_ < 0 ? (int) threadIdx.x : tid_;
    const int64_t nv = n >> 2;
    const double s2 = waves_combine(red + 16, nt >> 6, op_add());
    const float variance = (float)(s2 / (double) n);
    const float scale = 1.0f / sqrtf(variance + 1e-5f);
#pragma unroll
    for (int k = 0; k < NLN; ++k) {
        const int64_t q4 = (int64_t) k * nt + tid;
        float4 v = r.t[k];
        const float4 ww = wr.t[k], bb = br.t[k];
        v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
        v.x = v.x * ww.x + bb.x; v.y = v.y * ww.y + bb.y; v.z = v.z * ww.z + bb.z; v.w = v.w * ww.w + bb.w;
        if constexpr (ACT == FQ_Q8_K) {
            if ((q4 >> 6) < (n >> 8)) quant_q8K_wave(v, tid & 63, q4 >> 6, o);             // wave-uniform
        } else {
            if ((q4 & ~(int64_t) 63) < nv) { const bool live = q4 < nv; quant_q8_quad<ACT>(v, live ? q4 : nv - 1, o, live); }
        }
    }
}

__device__ __forceinline__ void layer_norm_row_block(const float * __restrict__ x, int64_t n, const float * __restrict__ w,
                                                     const float * __restrict__ b, float * row, double * red) {
    ln_row_regs<8> r;
    layer_norm_issue(x, n, r);
    layer_norm_finish(r, x, n, w, b, row, red);
}
