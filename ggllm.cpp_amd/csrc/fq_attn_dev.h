// fq_attn_dev.h -- one (head, token) of Falcon attention by 256 threads (a workgroup, or one of several lockstep
// 256-thread groups of a larger workgroup: tid = index in the group, every barrier is a workgroup barrier); shared by k_attention (prefill /
// op-by-op path) and k_attn_decode (fused N = 1 path) so that both produce the same bits.
//
//   scores  K.Q (ggml.c:11049-11088) * 1/sqrt(64) (libfalcon.cpp:2313-2317); keys j >= n_kv are masked (ggml.c:12341)
//   softmax max, exp through the fp16 table, f64 sum, scale by (float)(1/sum)  (ggml.c:12389-12456)
//   V.P     out[d] = sum_j V[j][d] * p[j]
// Two arithmetic variants of the two dot products (template parameter F64):
//   F64 = false (default, every fused kernel): f32 fused multiply-add chains, as the reference's SIMD builds accumulate
//           (ggml_vec_dot_f32 with GGML_F32_VEC_FMA, ggml.c:2270-2294): per lane 8 dims, then a butterfly over the 8 lanes;
//           per value-row class (j mod 16) one chain, the 16 classes added in order. One instruction per multiply-add.
//   F64 = true  (ggml_hip_reference_order): f32 products accumulated in f64 like the reference's portable
//           ggml_vec_dot_f32 (ggml.c:2296-2300); the f64 sums are associated as above, which changes the f32 result with
//           probability ~1e-9. Three instructions per multiply-add (v_mul_f32, v_cvt_f64_f32, v_add_f64): 4-6 x slower.
// The oracle models both (orc_set_sum_order).
// (One thread per key row would reproduce the reference's order exactly and needs a third of the instructions, but its
// 64 scattered 16-byte requests per load instruction cost more than they save: measured +3 us per decode attention.)
//
// Keys/values [0, n_cached) come from the cache ([pos][HKV][64] f32); an optional newest key/value (index n_cached)
// comes from LDS (the fused decode kernel has not written it to HBM for other workgroups to see).
// Thread map. Scores: 8 lanes per key row (lane s8 owns dims 4 s8.. and 32 + 4 s8..: every load instruction reads runs of
// 128 contiguous bytes), 3 DPP steps per row. V.P: sub = tid & 15 owns 4 consecutive head dims, rowi = tid >> 4 owns
// value rows j == rowi (mod 16). Loads in steps of 128 rows per workgroup, one step ahead of their use.
#pragma once
#include "fq_device.h"

#define FQ_ATTN_STAMP(dbg, slot) do { if ((dbg) && threadIdx.x == 0) (dbg)[(size_t) blockIdx.x * 8 + (slot)] = (long long) wall_clock64(); } while (0)

// workgroup barriers attn_head_block executes (scores+maxima, sums, probabilities, V.P partials): waves of the workgroup that
// do NOT take part in an attention group must execute as many (k_attn_out_ln), see attn_decode_group_idle
#define FQ_ATTN_HEAD_BARRIERS 3

struct attn_lds {
    float  * redf;     // >= 16 floats
    double * red;      // 16 x 64 doubles
    float  * p;        // >= n_kv floats
};

__device__ __forceinline__ size_t attn_lds_bytes(int max_n_kv) { return 16 * 4 + 16 * 64 * 8 + (((size_t) max_n_kv * 4 + 15) & ~(size_t) 15); }

__device__ __forceinline__ attn_lds attn_lds_carve(uint8_t * base) {       // base 16-byte aligned
    attn_lds a;
    a.redf = (float *) base;
    a.red  = (double *)(base + 64);
    a.p    = (float *)(base + 64 + 16 * 64 * 8);
    return a;
}

// The first 128 key rows and the first 128 value rows (8 per thread each) can be requested before q is known (the decode
// kernel issues them together with the q/k/v loads of the rope, one memory round trip earlier); later batches are
// requested one batch ahead of their use. The arithmetic does not depend on any of this.
typedef float f32x4 __attribute__((ext_vector_type(4)));      // (arrays of HIP's float4 struct are not promoted to registers)
// Loads run TWO steps (of 128 rows) ahead of their use: decode attention at a few hundred keys is a chain of memory round
// trips (each 1.5-2 us while the rest of the chip streams weights), so keys [0, 256) and values [0, 128) are requested
// before q is even known (the decode kernel issues them first thing), values [128, 256) as soon as the scores are done.
struct attn_pre { f32x4 k[16], v[8]; };                       // k[8 s + ..]: step s (s = 0, 1)

__device__ __forceinline__ void attn_load_k(const float * __restrict__ kc, int HKV, int hk, int row_limit, int j0, int tid, f32x4 * k8) {
    const int s8 = tid & 7, rowg = tid >> 3, last = row_limit > 0 ? row_limit - 1 : 0;      // 8 lanes per row: float4 s8 and s8 + 8
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int j = j0 + 32 * b + rowg;
        const f32x4 * r = (const f32x4 *)(kc + ((int64_t)(j < row_limit ? j : last) * HKV + hk) * 64);
        k8[2 * b] = r[s8]; k8[2 * b + 1] = r[s8 + 8];
    }
}
__device__ __forceinline__ void attn_load_v(const float * __restrict__ vc, int HKV, int hk, int row_limit, int j0, int tid, f32x4 * v8) {
    const int sub = tid & 15, rowi = tid >> 4, last = row_limit > 0 ? row_limit - 1 : 0;    // 16 lanes per row: float4 sub
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        const int j = j0 + 16 * b + rowi;
        v8[b] = *(const f32x4 *)(vc + ((int64_t)(j < row_limit ? j : last) * HKV + hk) * 64 + 4 * sub);
    }
}
// row_limit: rows [0, row_limit) exist in the cache (the cache's length when n_cached is not known yet, else n_cached)
__device__ __forceinline__ void attn_prefetch(const float * __restrict__ kc, const float * __restrict__ vc, int HKV, int hk, int row_limit,
                                              int tid, attn_pre & P) {
    attn_load_k(kc, HKV, hk, row_limit, 0, tid, P.k);
    attn_load_v(vc, HKV, hk, row_limit, 0, tid, P.v);
    attn_load_k(kc, HKV, hk, row_limit, 128, tid, P.k + 8);
}

// scores of the 128 rows [j0, j0 + 128) held in k8
// one lane's share of a 64-dim dot (dims 4 s8.. and 32 + 4 s8..), then the sum over the row's 8 lanes
template <bool F64>
__device__ __forceinline__ float attn_dot8(const f32x4 ka, const f32x4 kb, const f32x4 qa, const f32x4 qb) {
    if constexpr (F64) {
        double s = (double)(ka.x * qa.x); s += (double)(ka.y * qa.y); s += (double)(ka.z * qa.z); s += (double)(ka.w * qa.w);
        s += (double)(kb.x * qb.x); s += (double)(kb.y * qb.y); s += (double)(kb.z * qb.z); s += (double)(kb.w * qb.w);
        return (float) reduce8(s, op_add());
    } else {
        float s = ka.x * qa.x; s = __builtin_fmaf(ka.y, qa.y, s); s = __builtin_fmaf(ka.z, qa.z, s); s = __builtin_fmaf(ka.w, qa.w, s);
        s = __builtin_fmaf(kb.x, qb.x, s); s = __builtin_fmaf(kb.y, qb.y, s); s = __builtin_fmaf(kb.z, qb.z, s); s = __builtin_fmaf(kb.w, qb.w, s);
        return reduce8(s, op_add());
    }
}
template <bool F64> struct attn_acc { typedef float t; };
template <> struct attn_acc<true> { typedef double t; };
template <bool F64>
__device__ __forceinline__ void attn_mac(typename attn_acc<F64>::t & a, float v, float p) {
    if constexpr (F64) a += (double)(v * p); else a = __builtin_fmaf(v, p, a);
}

// the newest key / value of a decode step, not yet in the cache for other workgroups to see: either 64 floats each in LDS (new_k / new_v pointers) or, REGS,
// this thread's own dims in registers (attn_new: dims 4 s8 .. and 32 + 4 s8 .. of the rotated q and key, dims 4 sub .. of the value)
struct attn_new { f32x4 qa, qb, ka, kb, v4; };
template <bool F64 = false, bool REGS = false>
__device__ __forceinline__ void attn_score_step(const f32x4 * k8, int j0, int n_cached, int n_kv, const float * new_k, const f32x4 qa, const f32x4 qb,
                                                int tid, float * p, float & lmax, const f32x4 nka = f32x4{0, 0, 0, 0}, const f32x4 nkb = f32x4{0, 0, 0, 0}) {
    const int s8 = tid & 7, rowg = tid >> 3;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int j = j0 + 32 * b + rowg;
        f32x4 ka = k8[2 * b], kb = k8[2 * b + 1];
        if constexpr (REGS) { if (j == n_cached) { ka = nka; kb = nkb; } }
        else if (new_k && j == n_cached) { ka = *(const f32x4 *)(new_k + 4 * s8); kb = *(const f32x4 *)(new_k + 32 + 4 * s8); }
        const float sc = attn_dot8<F64>(ka, kb, qa, qb) * 0.125f;
        if (j < n_kv && s8 == 0) p[j] = sc;
        lmax = fq_max_f32(lmax, j < n_kv ? sc : -INFINITY);
    }
}
// SCALE: p holds the soft_max's exp() values and the probability is formed here, p[j] * inv -- the f32 product ggml_vec_scale_f32 stores (ggml.c:12441-12450),
// so the same number, without a pass over p and the barrier behind it
template <bool F64 = false, bool SCALE = false>
__device__ __forceinline__ void attn_pv_step(const f32x4 * v8, int j0, int n_cached, int tid, const float * p, typename attn_acc<F64>::t & a0,
                                             typename attn_acc<F64>::t & a1, typename attn_acc<F64>::t & a2, typename attn_acc<F64>::t & a3, float inv = 1.0f) {
    const int rowi = tid >> 4;
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        const int j = j0 + 16 * b + rowi;
        float pv = p[j < n_cached ? j : 0];
        if constexpr (SCALE) pv *= inv;
        const float pj = j < n_cached ? pv : 0.0f;                // (a row beyond the end is a clamped, finite re-read: v * 0 adds nothing)
        const f32x4 v4 = v8[b];
        attn_mac<F64>(a0, v4.x, pj); attn_mac<F64>(a1, v4.y, pj); attn_mac<F64>(a2, v4.z, pj); attn_mac<F64>(a3, v4.w, pj);
    }
}

// q: 64 floats (rotated) in LDS or global; returns out[d] for d = tid (valid for tid < 64). REGS: q, the newest key and value come in `nw` (registers), the
// three pointers are unused and the newest key / value always exist
template <bool F64 = false, bool REGS = false>
__device__ __forceinline__ float attn_head_block(const float * __restrict__ q, const float * __restrict__ kc, const float * __restrict__ vc,
                                                 int HKV, int hk, int n_cached, const float * new_k, const float * new_v,
                                                 const uint16_t * __restrict__ exp_tab, const attn_lds & L, const int tid, attn_pre & P,
                                                 long long * dbg = nullptr, const attn_new * nw = nullptr) {
    constexpr int NT = 256;
    const int sub = tid & 15, rowi = tid >> 4;
    const int n_kv = n_cached + ((REGS || new_k) ? 1 : 0);

    // ---- scores: 8 lanes per key row (two float4 each), 4 rows per thread and step, 128 rows per step
    float lmax = -INFINITY;
    {
        const int s8 = tid & 7;
        f32x4 qa, qb, nka = f32x4{0, 0, 0, 0}, nkb = f32x4{0, 0, 0, 0};
        if constexpr (REGS) { qa = nw->qa; qb = nw->qb; nka = nw->ka; nkb = nw->kb; }
        else { qa = *(const f32x4 *)(q + 4 * s8); qb = *(const f32x4 *)(q + 32 + 4 * s8); }
        for (int j0 = 0; j0 < n_kv; j0 += 256) {
            attn_score_step<F64, REGS>(P.k, j0, n_cached, n_kv, new_k, qa, qb, tid, L.p, lmax, nka, nkb);
            if (j0 + 256 < n_kv) attn_load_k(kc, HKV, hk, n_cached, j0 + 256, tid, P.k);
            if (j0 + 128 < n_kv) {
                attn_score_step<F64, REGS>(P.k + 8, j0 + 128, n_cached, n_kv, new_k, qa, qb, tid, L.p, lmax, nka, nkb);
                if (j0 + 384 < n_kv) attn_load_k(kc, HKV, hk, n_cached, j0 + 384, tid, P.k + 8);
            }
        }
    }
    f32x4 v1[8];                                                   // values of the odd steps (the key registers are free now)
    if (128 < n_cached) attn_load_v(vc, HKV, hk, n_cached, 128, tid, v1);
    FQ_ATTN_STAMP(dbg, 3);
    // ---- soft_max (two barriers: scores + maxima visible, exp() values + sums visible; the scaling by 1 / sum is applied where V.P reads the values)
    lmax = wave_max(lmax);
    if ((tid & 63) == 0) L.redf[tid >> 6] = lmax;
    __syncthreads();
    const float mx = waves_combine(L.redf, NT >> 6, op_max());
    double lsum = 0.0;
    for (int j = tid; j < n_kv; j += NT) {
        const float e = soft_max_exp(exp_tab, L.p[j] - mx);
        L.p[j] = e;
        lsum += (double) e;
    }
    lsum = wave_sum(lsum);
    // (the per-wave sums sit in the half of `red` the f32 V.P partials below never touch: no barrier separates their readers from those writers any more)
    double * const sred = F64 ? L.red : L.red + 512;
    if ((tid & 63) == 0) sred[tid >> 6] = lsum;
    __syncthreads();
    const double sum = waves_combine(sred, NT >> 6, op_add());
    const float inv = (float)(1.0 / sum);
    FQ_ATTN_STAMP(dbg, 4);
    // ---- V.P
    typedef typename attn_acc<F64>::t acc_t;
    acc_t a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    acc_t * const pvred = (acc_t *) L.red;                         // 16 row classes x 64 dims
    for (int j0 = 0; j0 < n_cached; j0 += 256) {
        attn_pv_step<F64, true>(P.v, j0, n_cached, tid, L.p, a0, a1, a2, a3, inv);
        if (j0 + 256 < n_cached) attn_load_v(vc, HKV, hk, n_cached, j0 + 256, tid, P.v);
        if (j0 + 128 < n_cached) {
            attn_pv_step<F64, true>(v1, j0 + 128, n_cached, tid, L.p, a0, a1, a2, a3, inv);
            if (j0 + 384 < n_cached) attn_load_v(vc, HKV, hk, n_cached, j0 + 384, tid, v1);
        }
    }
    if ((REGS || new_v) && rowi == (n_cached & 15)) {
        float4 v;
        if constexpr (REGS) v = make_float4(nw->v4.x, nw->v4.y, nw->v4.z, nw->v4.w); else v = *(const float4 *)(new_v + 4 * sub);
        const float pj = L.p[n_cached] * inv;
        attn_mac<F64>(a0, v.x, pj); attn_mac<F64>(a1, v.y, pj); attn_mac<F64>(a2, v.z, pj); attn_mac<F64>(a3, v.w, pj);
    }
    FQ_ATTN_STAMP(dbg, 5);
    if constexpr (F64) __syncthreads();                            // (f64 partials fill all of `red`, the sums included: every reader of those must be through)
    pvred[rowi * 64 + 4 * sub + 0] = a0; pvred[rowi * 64 + 4 * sub + 1] = a1;
    pvred[rowi * 64 + 4 * sub + 2] = a2; pvred[rowi * 64 + 4 * sub + 3] = a3;
    __syncthreads();
    float out = 0.0f;
    if (tid < 64) {
        acc_t o = pvred[tid];
#pragma unroll
        for (int r = 1; r < 16; ++r) o += pvred[r * 64 + tid];
        out = (float) o;
    }
    return out;
}
template <bool F64 = false>
__device__ __forceinline__ float attn_head_block(const float * __restrict__ q, const float * __restrict__ kc, const float * __restrict__ vc,
                                                 int HKV, int hk, int n_cached, const float * new_k, const float * new_v,
                                                 const uint16_t * __restrict__ exp_tab, const attn_lds & L) {
    attn_pre P;
    attn_prefetch(kc, vc, HKV, hk, n_cached, (int) threadIdx.x, P);
    return attn_head_block<F64>(q, kc, vc, HKV, hk, n_cached, new_k, new_v, exp_tab, L, (int) threadIdx.x, P);
}

// ---- prefill: R consecutive tokens of one head by one 256-thread workgroup --------------------------------------------
// Every number is produced exactly as attn_head_block produces it for one token (same lane -> dimension map, same f64
// partial sums, same reduction trees: the results are bit-identical), but a tile of 128 key rows / value rows is loaded
// into registers ONCE and used for all R tokens: with one workgroup per (head, token) a 2048-token prompt re-reads the
// head's keys and values 2048 times from L2 (72 GB per block of Falcon-7B), with R = 8 an eighth of that.
//   p: R rows of p_stride floats in LDS (scores, then probabilities); token t0 + r sees keys [0, n_past + t0 + r + 1)
template <int R, bool F64>
__device__ __forceinline__ void attn_rows_block(const float * __restrict__ qkv, int heads, int h, int t0, int nrows, int n_past,
                                                const float * __restrict__ kc, const float * __restrict__ vc, int HKV, int hk,
                                                const uint16_t * __restrict__ exp_tab, float * redf, double * red, float * p, int p_stride,
                                                float * __restrict__ att, int H) {
    constexpr int NT = 256;
    const int tid = (int) threadIdx.x, s8 = tid & 7, rowg = tid >> 3, sub = tid & 15, rowi = tid >> 4;
    const int n_kv_max = n_past + t0 + nrows;                       // keys the block's last token sees
    // ---- scores
    f32x4 qa[R], qb[R];
    float lmax[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const float * q = qkv + ((int64_t)(t0 + (r < nrows ? r : nrows - 1)) * heads + h) * 64;
        qa[r] = *(const f32x4 *)(q + 4 * s8); qb[r] = *(const f32x4 *)(q + 32 + 4 * s8);
        lmax[r] = -INFINITY;
    }
    {
        f32x4 k8[8], kn[8];
        attn_load_k(kc, HKV, hk, n_kv_max, 0, tid, k8);
        for (int j0 = 0; j0 < n_kv_max; j0 += 128) {
            if (j0 + 128 < n_kv_max) attn_load_k(kc, HKV, hk, n_kv_max, j0 + 128, tid, kn);
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int j = j0 + 32 * b + rowg;
                const f32x4 ka = k8[2 * b], kb = k8[2 * b + 1];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const float sc = attn_dot8<F64>(ka, kb, qa[r], qb[r]) * 0.125f;
                    const bool vis = r < nrows && j < n_past + t0 + r + 1;      // (selects, not branches: a branch per
                    if (vis && s8 == 0) p[r * p_stride + j] = sc;               //  accumulator update costs a copy of every live accumulator)
                    lmax[r] = fq_max_f32(lmax[r], vis ? sc : -INFINITY);
                }
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) k8[i] = kn[i];
        }
    }
    // ---- soft_max, one token after the other with all 256 threads (three barriers each, as in attn_head_block)
    for (int r = 0; r < nrows; ++r) {
        const int n_kv = n_past + t0 + r + 1;
        float * pr = p + r * p_stride;
        float lm = lmax[0];
#pragma unroll
        for (int q = 1; q < R; ++q) if (q == r) lm = lmax[q];
        lm = wave_max(lm);
        if ((tid & 63) == 0) redf[tid >> 6] = lm;
        __syncthreads();
        const float mx = waves_combine(redf, NT >> 6, op_max());
        double lsum = 0.0;
        for (int j = tid; j < n_kv; j += NT) {
            const float e = soft_max_exp(exp_tab, pr[j] - mx);
            pr[j] = e;
            lsum += (double) e;
        }
        lsum = wave_sum(lsum);
        if ((tid & 63) == 0) red[tid >> 6] = lsum;
        __syncthreads();
        const double sum = waves_combine(red, NT >> 6, op_add());
        const float inv = (float)(1.0 / sum);
        for (int j = tid; j < n_kv; j += NT) pr[j] *= inv;
        __syncthreads();
    }
    // ---- V.P
    typedef typename attn_acc<F64>::t acc_t;
    acc_t * const pvred = (acc_t *) red;
    acc_t a[R][4];
#pragma unroll
    for (int r = 0; r < R; ++r) { a[r][0] = 0; a[r][1] = 0; a[r][2] = 0; a[r][3] = 0; }
    {
        f32x4 v8[8], vn[8];
        attn_load_v(vc, HKV, hk, n_kv_max, 0, tid, v8);
        for (int j0 = 0; j0 < n_kv_max; j0 += 128) {
            if (j0 + 128 < n_kv_max) attn_load_v(vc, HKV, hk, n_kv_max, j0 + 128, tid, vn);
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const int j = j0 + 16 * b + rowi;
                const f32x4 v4 = v8[b];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    // an invisible key contributes v * 0 (v is a clamped, finite re-read): the accumulator keeps its value
                    const bool vis = r < nrows && j < n_past + t0 + r + 1;
                    const float pj = vis ? p[r * p_stride + j] : 0.0f;
                    attn_mac<F64>(a[r][0], v4.x, pj); attn_mac<F64>(a[r][1], v4.y, pj); attn_mac<F64>(a[r][2], v4.z, pj); attn_mac<F64>(a[r][3], v4.w, pj);
                }
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) v8[i] = vn[i];
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (r < nrows) {                                            // (uniform)
            pvred[rowi * 64 + 4 * sub + 0] = a[r][0]; pvred[rowi * 64 + 4 * sub + 1] = a[r][1];
            pvred[rowi * 64 + 4 * sub + 2] = a[r][2]; pvred[rowi * 64 + 4 * sub + 3] = a[r][3];
            __syncthreads();
            if (tid < 64) {
                acc_t o = pvred[tid];
#pragma unroll
                for (int q = 1; q < 16; ++q) o += pvred[q * 64 + tid];
                att[(int64_t)(t0 + r) * H * 64 + (int64_t) h * 64 + tid] = (float) o;
            }
            __syncthreads();
        }
    }
}
