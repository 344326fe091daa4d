// fq_ref_dot.h -- the row dot products of ggml_compute_forward_mul_mat_q_f32 in the REFERENCE'S OWN association
// (`ggml_hip_reference_order(1)`): one thread walks one weight row against one activation column block by block, exactly
// like the scalar branches of ggml_vec_dot_q*_q8_* -- legacy formats: one f32 term per 32-block added left to right
// (ggml.c:2591-2609, 2716-2735, 2951-2972, 3207-3228, 3317-3329); Q2_K: one term per super-block (k_quants.c:1267-1306);
// Q3_K .. Q6_K: eight f32 lanes, element e of a super-block feeds lane e mod 8, lanes added 0..7 after the last
// super-block, the mins subtracted per super-block (k_quants.c:1684-1746, 1999-2055, 2340-2400, 2748-2789).
// Nothing here is fast: it exists so that logits can be compared with the reference's scalar build bit for bit.
// Host-compilable (tests/host/refdot_harness.cpp runs it against the golden vectors of the real reference).
#pragma once
#include "fq_types.h"
#include <string.h>

#if defined(__HIPCC__)
#define FQ_REF_HD __host__ __device__ inline
#else
#define FQ_REF_HD static inline
#endif

FQ_REF_HD float fq_ref_h2f(const uint8_t * p) {            // IEEE half -> float, portable integer form
    const uint32_t h = (uint32_t) p[0] | ((uint32_t) p[1] << 8);
    const uint32_t s = (h & 0x8000u) << 16, e = (h >> 10) & 31u; uint32_t m = h & 0x3FFu, b;
    if (e == 0) { if (!m) b = s; else { int k = -1; do { m <<= 1; ++k; } while (!(m & 0x400u)); b = s | (uint32_t)(112 - k) << 23 | (m & 0x3FFu) << 13; } }
    else if (e == 31) b = s | 0x7F800000u | m << 13;
    else b = s | (e + 112u) << 23 | m << 13;
    float f; memcpy(&f, &b, 4); return f;
}

// one ggml block (array-of-structs bytes, fq_types.h header comment) re-assembled from the interleaved device layout
FQ_REF_HD void fq_ref_gather_block(const fq_type_desc & d, const uint8_t * row, int64_t nblk, int64_t b, uint8_t * blk) {
    for (int p = 0; p < d.nplanes; ++p) {
        const uint8_t * src = row + fq_il_offset(d, p, nblk, b);
        for (int i = 0; i < d.plane[p].bytes; ++i) blk[d.plane[p].src_off + i] = src[i];
    }
}

// the activation image of one column: [int8 x K | f32 d | aux] (fq_types.h)
struct fq_ref_act {
    const int8_t * qs; const uint8_t * d; const uint8_t * aux;
};
FQ_REF_HD float fq_ref_f32(const uint8_t * p) { float f; memcpy(&f, p, 4); return f; }
FQ_REF_HD int   fq_ref_i16(const uint8_t * p) { int16_t v; memcpy(&v, p, 2); return v; }

// integer weight of element e of a k-quant super-block and the scale / min of its 16-element block (SURVEY 8a')
template <int TYPE>
FQ_REF_HD void fq_ref_k_scales(const uint8_t * w, int * sc16, int * mn16) {
    if (TYPE == FQ_Q2_K) {
        for (int b = 0; b < 16; ++b) { sc16[b] = w[b] & 15; mn16[b] = w[b] >> 4; }
    } else if (TYPE == FQ_Q3_K) {                                       // k_quants.c:491-496 / 1718-1723
        const uint8_t * s = w + 96;
        for (int b = 0; b < 16; ++b) {
            const int lo = (b < 8) ? (s[b] & 15) : (s[b - 8] >> 4);
            const int hi = (s[8 + (b & 3)] >> (2 * (b >> 2))) & 3;
            sc16[b] = (lo | (hi << 4)) - 32; mn16[b] = 0;
        }
    } else if (TYPE == FQ_Q4_K || TYPE == FQ_Q5_K) {                    // get_scale_min_k4, k_quants.c:264-272
        const uint8_t * q = w + 4;
        for (int b = 0; b < 16; ++b) {
            const int j = b >> 1;
            int sc, m;
            if (j < 4) { sc = q[j] & 63; m = q[j + 4] & 63; }
            else       { sc = (q[j + 4] & 0xF) | ((q[j - 4] >> 6) << 4); m = (q[j + 4] >> 4) | ((q[j] >> 6) << 4); }
            sc16[b] = sc; mn16[b] = m;
        }
    } else {                                                            // Q6_K: int8 scales
        for (int b = 0; b < 16; ++b) { sc16[b] = (int8_t) w[192 + b]; mn16[b] = 0; }
    }
}
template <int TYPE>
FQ_REF_HD int fq_ref_k_quant(const uint8_t * w, int e) {
    if (TYPE == FQ_Q2_K) return (w[16 + (e / 128) * 32 + e % 32] >> (2 * ((e % 128) / 32))) & 3;
    if (TYPE == FQ_Q3_K) {
        const int lo = (w[32 + (e / 128) * 32 + e % 32] >> (2 * ((e % 128) / 32))) & 3;
        return lo - (((w[e % 32] >> (e / 32)) & 1) ? 0 : 4);
    }
    if (TYPE == FQ_Q4_K || TYPE == FQ_Q5_K) {
        const int qs_off = (TYPE == FQ_Q4_K) ? 16 : 48;
        const int byte = w[qs_off + 32 * (e / 64) + e % 32];
        int q = ((e % 64) / 32) ? (byte >> 4) : (byte & 15);
        if (TYPE == FQ_Q5_K) q += ((w[16 + e % 32] >> (e / 32)) & 1) ? 16 : 0;
        return q;
    }
    const int h = e / 128, t = (e % 128) / 32, l = e % 32;              // Q6_K
    const int byte = w[64 * h + 32 * (t & 1) + l];
    const int lo = (t < 2) ? (byte & 15) : (byte >> 4);
    const int hi = (w[128 + 32 * h + l] >> (2 * t)) & 3;
    return (int)(int8_t)(lo | (hi << 4)) - 32;
}

// row = the weight row in the device layout, nblk ggml blocks long
template <int TYPE>
FQ_REF_HD float fq_ref_row_dot(const uint8_t * row, int64_t nblk, const fq_ref_act & a) {
    constexpr fq_type_desc d = fq_desc(TYPE);
    uint8_t w[212];
    float sumf = 0.0f;
    if (d.blck == 32) {
        for (int64_t i = 0; i < nblk; ++i) {
            fq_ref_gather_block(d, row, nblk, i, w);
            const int8_t * q8 = a.qs + 32 * i;
            const float da = fq_ref_f32(a.d + 4 * i);                   // Q8_0: the fp16-rounded delta; Q8_1: f32 delta
            int sumi = 0;
            if (TYPE == FQ_Q4_0) {
                for (int j = 0; j < 16; ++j) sumi += ((w[2 + j] & 15) - 8) * q8[j] + ((w[2 + j] >> 4) - 8) * q8[j + 16];
                sumf += (float) sumi * fq_ref_h2f(w) * da;
            } else if (TYPE == FQ_Q4_1) {
                for (int j = 0; j < 16; ++j) sumi += (w[4 + j] & 15) * q8[j] + (w[4 + j] >> 4) * q8[j + 16];
                sumf += (fq_ref_h2f(w) * da) * (float) sumi + fq_ref_h2f(w + 2) * fq_ref_f32(a.aux + 4 * i);
            } else if (TYPE == FQ_Q5_0) {
                uint32_t qh; memcpy(&qh, w + 2, 4);
                for (int j = 0; j < 16; ++j)
                    sumi += (((w[6 + j] & 15) | (((qh >> j) & 1) << 4)) - 16) * q8[j]
                          + (((w[6 + j] >> 4) | (((qh >> (j + 16)) & 1) << 4)) - 16) * q8[j + 16];
                sumf += (fq_ref_h2f(w) * da) * (float) sumi;
            } else if (TYPE == FQ_Q5_1) {
                uint32_t qh; memcpy(&qh, w + 4, 4);
                for (int j = 0; j < 16; ++j)
                    sumi += ((w[8 + j] & 15) | (((qh >> j) & 1) << 4)) * q8[j]
                          + ((w[8 + j] >> 4) | (((qh >> (j + 16)) & 1) << 4)) * q8[j + 16];
                sumf += (fq_ref_h2f(w) * da) * (float) sumi + fq_ref_h2f(w + 2) * fq_ref_f32(a.aux + 4 * i);
            } else {                                                    // Q8_0
                for (int j = 0; j < 32; ++j) sumi += (int)(int8_t) w[2 + j] * q8[j];
                sumf += (float) sumi * (fq_ref_h2f(w) * da);
            }
        }
        return sumf;
    }
    float lanes[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    for (int64_t i = 0; i < nblk; ++i) {
        fq_ref_gather_block(d, row, nblk, i, w);
        const int8_t * q8 = a.qs + 256 * i;
        const float dy = fq_ref_f32(a.d + 4 * i);
        int sc16[16], mn16[16];
        fq_ref_k_scales<TYPE>(w, sc16, mn16);
        int msum = 0;
        if (TYPE == FQ_Q2_K || TYPE == FQ_Q4_K || TYPE == FQ_Q5_K)
            for (int b = 0; b < 16; ++b) msum += fq_ref_i16(a.aux + 2 * (16 * i + b)) * mn16[b];
        if (TYPE == FQ_Q2_K) {
            int isum = 0;
            for (int e = 0; e < 256; ++e) isum += sc16[e / 16] * (fq_ref_k_quant<TYPE>(w, e) * q8[e]);
            const float dall = dy * fq_ref_h2f(w + 80), dmin = dy * fq_ref_h2f(w + 82);
            sumf += dall * (float) isum - dmin * (float) msum;
            continue;
        }
        int aux32[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
        for (int e = 0; e < 256; ++e) aux32[e & 7] += sc16[e / 16] * (fq_ref_k_quant<TYPE>(w, e) * q8[e]);
        const float dd = fq_ref_h2f(w + (TYPE == FQ_Q3_K ? 108 : (TYPE == FQ_Q6_K ? 208 : 0))) * dy;
        for (int l = 0; l < 8; ++l) lanes[l] += dd * (float) aux32[l];
        if (TYPE == FQ_Q4_K || TYPE == FQ_Q5_K) sumf -= (fq_ref_h2f(w + 2) * dy) * (float) msum;
    }
    if (TYPE != FQ_Q2_K) for (int l = 0; l < 8; ++l) sumf += lanes[l];
    return sumf;
}
