// fq_attn_dev.h -- one (head, token) of Falcon attention by a 256-thread workgroup; shared by k_attention (prefill /
// op-by-op path) and k_attn_decode (fused N = 1 path) so that both produce the same bits.
//
//   scores  K.Q (ggml.c:11049-11088) * 1/sqrt(64) (libfalcon.cpp:2313-2317); keys j >= n_kv are masked (ggml.c:12341)
//   softmax max, exp through the fp16 table, f64 sum, scale by (float)(1/sum)  (ggml.c:12389-12456)
//   V.P     out[d] = sum_j V[j][d] * p[j]
// f32 products are accumulated in f64 like the reference's portable ggml_vec_dot_f32 (ggml.c:2296-2300); the f64 sums
// are associated differently (16 lanes / 16 row classes), which changes the f32 result with probability ~1e-9.
//
// Keys/values [0, n_cached) come from the cache ([pos][HKV][64] f32); an optional newest key/value (index n_cached)
// comes from LDS (the fused decode kernel has not written it to HBM for other workgroups to see).
// Thread map: sub = tid & 15 owns 4 consecutive head dims, rowi = tid >> 4 owns key rows j == rowi (mod 16); global
// loads are issued in batches of 8 rows per thread (128 rows per workgroup step).
#pragma once
#include "fq_device.h"

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

// q: 64 floats (rotated) in LDS or global; returns out[d] for d = tid (valid for tid < 64)
__device__ __forceinline__ float attn_head_block(const float * __restrict__ q, const float * __restrict__ kc, const float * __restrict__ vc,
                                                 int HKV, int hk, int n_cached, const float * new_k, const float * new_v,
                                                 const uint16_t * __restrict__ exp_tab, const attn_lds & L) {
    constexpr int D = 64;
    const int tid = threadIdx.x, sub = tid & 15, rowi = tid >> 4;
    const int n_kv = n_cached + (new_k ? 1 : 0);
    const float4 q4 = *(const float4 *)(q + 4 * sub);
    const int last = n_cached > 0 ? n_cached - 1 : 0;

    // ---- scores
    float lmax = -INFINITY;
    for (int j0 = 0; j0 < n_kv; j0 += 128) {
        float4 k4[8];
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const int j = j0 + 16 * b + rowi;
            k4[b] = *(const float4 *)(kc + ((int64_t)(j < n_cached ? j : last) * HKV + hk) * D + 4 * sub);
        }
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const int j = j0 + 16 * b + rowi;
            if (new_k && j == n_cached) k4[b] = *(const float4 *)(new_k + 4 * sub);
            double s = (double)(k4[b].x * q4.x); s += (double)(k4[b].y * q4.y); s += (double)(k4[b].z * q4.z); s += (double)(k4[b].w * q4.w);
            s = reduce16(s, op_add());
            const float sc = (float) s * 0.125f;
            if (j < n_kv) { if (sub == 0) L.p[j] = sc; lmax = fmaxf(lmax, sc); }
        }
    }
    const float mx = block_max(lmax, L.redf);
    __syncthreads();
    // ---- soft_max
    double lsum = 0.0;
    for (int j = tid; j < n_kv; j += blockDim.x) {
        const float e = h2f_bits(exp_tab[f2h_bits(L.p[j] - mx)]);
        L.p[j] = e;
        lsum += (double) e;
    }
    const double sum = block_sum(lsum, L.red);
    const float inv = (float)(1.0 / sum);
    __syncthreads();
    for (int j = tid; j < n_kv; j += blockDim.x) L.p[j] *= inv;
    __syncthreads();
    // ---- V.P
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    for (int j0 = 0; j0 < n_cached; j0 += 128) {
        float4 v4[8];
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const int j = j0 + 16 * b + rowi;
            v4[b] = *(const float4 *)(vc + ((int64_t)(j < n_cached ? j : last) * HKV + hk) * D + 4 * sub);
        }
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const int j = j0 + 16 * b + rowi;
            if (j < n_cached) {
                const float pj = L.p[j];
                a0 += (double)(v4[b].x * pj); a1 += (double)(v4[b].y * pj); a2 += (double)(v4[b].z * pj); a3 += (double)(v4[b].w * pj);
            }
        }
    }
    if (new_v && rowi == (n_cached & 15)) {
        const float4 v = *(const float4 *)(new_v + 4 * sub);
        const float pj = L.p[n_cached];
        a0 += (double)(v.x * pj); a1 += (double)(v.y * pj); a2 += (double)(v.z * pj); a3 += (double)(v.w * pj);
    }
    L.red[rowi * 64 + 4 * sub + 0] = a0; L.red[rowi * 64 + 4 * sub + 1] = a1;
    L.red[rowi * 64 + 4 * sub + 2] = a2; L.red[rowi * 64 + 4 * sub + 3] = a3;
    __syncthreads();
    float out = 0.0f;
    if (tid < 64) {
        double o = L.red[tid];
#pragma unroll
        for (int r = 1; r < 16; ++r) o += L.red[r * 64 + tid];
        out = (float) o;
    }
    return out;
}
