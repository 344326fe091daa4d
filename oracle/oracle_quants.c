/*
 * oracle_quants.c -- TEST INFRASTRUCTURE (see oracle.h). Scalar CPU restatement of the reference's
 * block formats, (de)quantizers, integer dot products and quantized mat-mul.
 *
 * Written against the bit-level format description in SURVEY.md 8(a'); blocks are addressed by byte
 * offset (no structs) so the layout knowledge is explicit:
 *
 *   Q4_0 18 B: d:f16 @0, qs[16] @2            Q4_1 20 B: d @0, m @2, qs[16] @4
 *   Q5_0 22 B: d @0, qh:u32 @2, qs[16] @6     Q5_1 24 B: d @0, m @2, qh:u32 @4, qs[16] @8
 *   Q8_0 34 B: d:f16 @0, qs[32]:i8 @2         Q8_1 40 B: d:f32 @0, s:f32 @4, qs[32] @8
 *   Q2_K 84 B: scales[16] @0, qs[64] @16, d @80, dmin @82
 *   Q3_K 110 B: hmask[32] @0, qs[64] @32, scales[12] @96, d @108
 *   Q4_K 144 B: d @0, dmin @2, scales[12] @4, qs[128] @16
 *   Q5_K 176 B: d @0, dmin @2, scales[12] @4, qh[32] @16, qs[128] @48
 *   Q6_K 210 B: ql[128] @0, qh[64] @128, scales[16]:i8 @192, d @208
 *   Q8_K 292 B: d:f32 @0, qs[256]:i8 @4, bsums[16]:i16 @260
 */
#include "oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

/* ------------------------------------------------------------------ fp16 <-> fp32 (IEEE, RNE) */
/* The reference uses F16C (_cvtss_sh / _cvtsh_ss, ggml.c:345-351) or the bit-twiddling fallback
 * (ggml.c:360-410); both are IEEE round-to-nearest-even with subnormals. */
float orc_fp16_to_fp32(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    const uint32_t exp  = (h >> 10) & 0x1Fu;
    uint32_t man        = h & 0x3FFu;
    uint32_t bits;
    if (exp == 0) {
        if (man == 0) {
            bits = sign;
        } else {                       /* subnormal half -> normal float */
            int e = -1;
            do { man <<= 1; ++e; } while ((man & 0x400u) == 0);
            bits = sign | (uint32_t)(127 - 15 - e) << 23 | (man & 0x3FFu) << 13;
        }
    } else if (exp == 31) {
        bits = sign | 0x7F800000u | man << 13;
    } else {
        bits = sign | (exp + 112u) << 23 | man << 13;
    }
    float f; memcpy(&f, &bits, 4); return f;
}

uint16_t orc_fp32_to_fp16(float f) {
    uint32_t x; memcpy(&x, &f, 4);
    const uint16_t sign = (uint16_t)((x >> 16) & 0x8000u);
    const uint32_t ax = x & 0x7FFFFFFFu;
    if (ax >= 0x7F800000u) {                         /* inf / nan */
        return (uint16_t)(sign | 0x7C00u | (ax > 0x7F800000u ? (0x200u | ((ax >> 13) & 0x3FFu)) : 0u));
    }
    if (ax >= 0x477FF000u) {                         /* >= 65520 rounds to inf */
        return (uint16_t)(sign | 0x7C00u);
    }
    if (ax < 0x33000001u) {                          /* <= 2^-25 rounds to zero (tie at 2^-25 -> even = 0) */
        return sign;
    }
    const int e = (int)(ax >> 23) - 127;
    uint32_t man = (ax & 0x7FFFFFu) | 0x800000u;     /* 24-bit significand */
    int shift;                                       /* bits to drop */
    uint32_t hexp;
    if (e < -14) { shift = 13 + (-14 - e); hexp = 0; }
    else         { shift = 13;             hexp = (uint32_t)(e + 15); }
    const uint32_t kept = man >> shift;
    const uint32_t rem  = man & ((1u << shift) - 1u);
    const uint32_t half = 1u << (shift - 1);
    uint32_t r = kept;
    if (rem > half || (rem == half && (kept & 1u))) r += 1;
    uint32_t out;
    if (hexp == 0) out = r;                          /* subnormal; carry into exponent field is the right encoding */
    else           out = ((hexp << 10) + (r - 0x400u)); /* r includes the implicit bit; carry bumps exponent */
    return (uint16_t)(sign | out);
}

static inline float    rd_f16(const uint8_t * p) { uint16_t h; memcpy(&h, p, 2); return orc_fp16_to_fp32(h); }
static inline void     wr_f16(uint8_t * p, float f) { const uint16_t h = orc_fp32_to_fp16(f); memcpy(p, &h, 2); }
static inline float    rd_f32(const uint8_t * p) { float f; memcpy(&f, p, 4); return f; }
static inline void     wr_f32(uint8_t * p, float f) { memcpy(p, &f, 4); }
static inline uint32_t rd_u32(const uint8_t * p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline int16_t  rd_i16(const uint8_t * p) { int16_t v; memcpy(&v, p, 2); return v; }

/* ------------------------------------------------------------------ type table (ggml.c:3342-3418) */
int orc_blck_size(int type) {
    switch (type) {
        case ORC_F32: case ORC_F16: return 1;
        case ORC_Q4_0: case ORC_Q4_1: case ORC_Q5_0: case ORC_Q5_1: case ORC_Q8_0: case ORC_Q8_1: return 32;
        case ORC_Q2_K: case ORC_Q3_K: case ORC_Q4_K: case ORC_Q5_K: case ORC_Q6_K: case ORC_Q8_K: return 256;
    }
    return 0;
}
size_t orc_type_size(int type) {
    switch (type) {
        case ORC_F32: return 4;   case ORC_F16: return 2;
        case ORC_Q4_0: return 18; case ORC_Q4_1: return 20; case ORC_Q5_0: return 22; case ORC_Q5_1: return 24;
        case ORC_Q8_0: return 34; case ORC_Q8_1: return 40;
        case ORC_Q2_K: return 84; case ORC_Q3_K: return 110; case ORC_Q4_K: return 144; case ORC_Q5_K: return 176;
        case ORC_Q6_K: return 210; case ORC_Q8_K: return 292;
    }
    return 0;
}
int orc_vec_dot_type(int wtype) {
    switch (wtype) {
        case ORC_Q4_0: case ORC_Q5_0: case ORC_Q8_0: return ORC_Q8_0;
        case ORC_Q4_1: case ORC_Q5_1: case ORC_Q8_1: return ORC_Q8_1;
        case ORC_Q2_K: case ORC_Q3_K: case ORC_Q4_K: case ORC_Q5_K: case ORC_Q6_K: return ORC_Q8_K;
    }
    return -1;
}
size_t orc_row_bytes(int type, int64_t k) { return (size_t)(k / orc_blck_size(type)) * orc_type_size(type); }

/* ------------------------------------------------------------------ legacy weight quantizers */
/* Symmetric formats: scale from the signed element of largest magnitude (ggml.c:927-962 Q4_0,
 * 1009-1048 Q5_0). levels = 8 (4 bit) or 16 (5 bit); the float->int step is trunc(v*id + levels + .5)
 * through an int8 cast, clamped at 2*levels-1. */
static void quant_sym(const float * x, uint8_t * blk, int levels, uint8_t * qs, uint32_t * qh_out) {
    float amax = 0.0f, vmax = 0.0f;
    for (int j = 0; j < 32; ++j) {
        const float v = x[j];
        if (amax < fabsf(v)) { amax = fabsf(v); vmax = v; }
    }
    const float d  = vmax / (float)(-levels);
    const float id = d ? 1.0f / d : 0.0f;
    wr_f16(blk, d);
    uint32_t qh = 0;
    const int top = 2 * levels - 1;
    for (int j = 0; j < 16; ++j) {
        int a = (int8_t)(x[j]      * id + ((float) levels + 0.5f));
        int b = (int8_t)(x[j + 16] * id + ((float) levels + 0.5f));
        if (a > top) a = top;
        if (b > top) b = top;
        const uint8_t ua = (uint8_t) a, ub = (uint8_t) b;
        qs[j] = (uint8_t)((ua & 0x0F) | ((ub & 0x0F) << 4));
        qh |= (uint32_t)((ua >> 4) & 1u) << j;
        qh |= (uint32_t)((ub >> 4) & 1u) << (j + 16);
    }
    if (qh_out) *qh_out = qh;
}

/* Affine formats (ggml.c:968-1003 Q4_1, 1054-1096 Q5_1): d = (max-min)/(2^bits-1), q = trunc((v-min)*id + .5) */
static void quant_affine(const float * x, uint8_t * blk, int bits, uint8_t * qs, uint32_t * qh_out) {
    float mn = FLT_MAX, mx = -FLT_MAX;
    for (int j = 0; j < 32; ++j) {
        if (x[j] < mn) mn = x[j];
        if (x[j] > mx) mx = x[j];
    }
    const float d  = (mx - mn) / (float)((1 << bits) - 1);
    const float id = d ? 1.0f / d : 0.0f;
    wr_f16(blk, d);
    wr_f16(blk + 2, mn);
    uint32_t qh = 0;
    for (int j = 0; j < 16; ++j) {
        const float a = (x[j]      - mn) * id;
        const float b = (x[j + 16] - mn) * id;
        uint8_t ua, ub;
        if (bits == 4) {           /* int8 cast then clamp to 15 */
            int ia = (int8_t)(a + 0.5f), ib = (int8_t)(b + 0.5f);
            ua = (uint8_t)(ia > 15 ? 15 : ia);
            ub = (uint8_t)(ib > 15 ? 15 : ib);
        } else {                   /* straight uint8 cast, no clamp (ggml.c:1080-1081) */
            ua = (uint8_t)(a + 0.5f);
            ub = (uint8_t)(b + 0.5f);
        }
        qs[j] = (uint8_t)((ua & 0x0F) | ((ub & 0x0F) << 4));
        qh |= (uint32_t)((ua >> 4) & 1u) << j;
        qh |= (uint32_t)((ub >> 4) & 1u) << (j + 16);
    }
    if (qh_out) *qh_out = qh;
}

void orc_quantize_row(int type, const float * x, void * out, int64_t k) {
    uint8_t * o = (uint8_t *) out;
    const int64_t nb = k / 32;
    for (int64_t i = 0; i < nb; ++i, x += 32) {
        uint32_t qh;
        switch (type) {
            case ORC_Q4_0: quant_sym(x, o, 8, o + 2, NULL);            o += 18; break;
            case ORC_Q4_1: quant_affine(x, o, 4, o + 4, NULL);         o += 20; break;
            case ORC_Q5_0: quant_sym(x, o, 16, o + 6, &qh);  memcpy(o + 2, &qh, 4); o += 22; break;
            case ORC_Q5_1: quant_affine(x, o, 5, o + 8, &qh); memcpy(o + 4, &qh, 4); o += 24; break;
            case ORC_Q8_0: orc_quantize_act(ORC_Q8_0, x, o, 32, ORC_ROUND_REFERENCE); o += 34; break;
            default: abort();
        }
    }
}

/* ------------------------------------------------------------------ activation quantizers */
/* round-half-even for |v| < 2^22 via the 1.5*2^23 magic constant (k_quants.c:50-55 nearest_int) */
static inline int nearest_even(float v) {
    const float t = v + 12582912.0f;
    int32_t i; memcpy(&i, &t, 4);
    return (i & 0x007FFFFF) - 0x00400000;
}

static void quant_q8_block(const float * x, float * d_out, int8_t * qs, int * isum, int flavour) {
    float amax = 0.0f;
    for (int j = 0; j < 32; ++j) { const float a = fabsf(x[j]); if (a > amax) amax = a; }
    const float d = amax / 127.0f;
    float id;
    if (flavour == ORC_ROUND_AVX) id = (amax != 0.0f) ? 127.0f / amax : 0.0f;    /* ggml.c:1205 */
    else                          id = d ? 1.0f / d : 0.0f;                      /* ggml.c:1118 */
    int s = 0;
    for (int j = 0; j < 32; ++j) {
        const float v = x[j] * id;
        const int q = (flavour == ORC_ROUND_AVX) ? nearest_even(v) : (int) roundf(v);
        qs[j] = (int8_t) q;
        s += qs[j];
    }
    *d_out = d;
    *isum = s;
}

/* Q8_K (k_quants.c:899-934): scale from the signed max-magnitude element, iscale = -128/max,
 * q = min(127, nearest_even(iscale*x)), d = 1/iscale, 16 partial sums of 16. */
static void quant_q8_K_block(const float * x, uint8_t * blk) {
    float vmax = 0.0f, amax = 0.0f;
    for (int j = 0; j < 256; ++j) {
        const float a = fabsf(x[j]);
        if (a > amax) { amax = a; vmax = x[j]; }
    }
    int8_t * qs = (int8_t *)(blk + 4);
    if (amax == 0.0f) {
        /* reference leaves bsums untouched here (k_quants.c:913-918); we zero them: d == 0 nulls every use */
        memset(blk, 0, 292);
        return;
    }
    const float iscale = -128.0f / vmax;
    for (int j = 0; j < 256; ++j) {
        int v = nearest_even(iscale * x[j]);
        qs[j] = (int8_t)(v > 127 ? 127 : v);
    }
    for (int g = 0; g < 16; ++g) {
        int s = 0;
        for (int j = 0; j < 16; ++j) s += qs[g * 16 + j];
        const int16_t s16 = (int16_t) s;
        memcpy(blk + 260 + 2 * g, &s16, 2);
    }
    wr_f32(blk, 1.0f / iscale);
}

void orc_quantize_act(int act_type, const float * x, void * out, int64_t k, int flavour) {
    uint8_t * o = (uint8_t *) out;
    if (act_type == ORC_Q8_K) {
        for (int64_t i = 0; i < k / 256; ++i) quant_q8_K_block(x + i * 256, o + i * 292);
        return;
    }
    for (int64_t i = 0; i < k / 32; ++i, x += 32) {
        float d; int s;
        if (act_type == ORC_Q8_0) {
            quant_q8_block(x, &d, (int8_t *)(o + 2), &s, flavour);
            wr_f16(o, d);
            o += 34;
        } else if (act_type == ORC_Q8_1) {
            quant_q8_block(x, &d, (int8_t *)(o + 8), &s, flavour);
            wr_f32(o, d);
            wr_f32(o + 4, (float) s * d);        /* ggml.c:1323 y.s = sum*d (AVX: d*sum, same product) */
            o += 40;
        } else abort();
    }
}

/* ------------------------------------------------------------------ integer decode of one block */
/* Produces the integer code q[e] and the (scale, offset) so that w[e] = scale[e]*q[e] - offset[e];
 * callers either apply the floats (dequantize) or keep q for the integer dot. */

static inline int q5_bit(uint32_t qh, int e) { return (int)((qh >> e) & 1u); }   /* bit e of qh = 5th bit of element e */

/* 6-bit (scale, min) pair j of a Q4_K/Q5_K block, k_quants.c:264-272 */
static void k4_scale_min(const uint8_t * s, int j, int * sc, int * mn) {
    if (j < 4) { *sc = s[j] & 63;  *mn = s[j + 4] & 63; }
    else       { *sc = (s[j + 4] & 0x0F) | ((s[j - 4] >> 6) << 4);
                 *mn = (s[j + 4] >> 4)   | ((s[j]     >> 6) << 4); }
}

/* sixteen 6-bit Q3_K scales out of 12 bytes (k_quants.c:491-496), value still biased by +32 */
static void q3_scales(const uint8_t * s12, int sc[16]) {
    for (int j = 0; j < 16; ++j) {
        const int lo = (j < 8) ? (s12[j] & 0x0F) : (s12[j - 8] >> 4);
        const int hi = (s12[8 + (j & 3)] >> (2 * (j >> 2))) & 3;
        sc[j] = lo | (hi << 4);
    }
}

void orc_dequantize_row(int type, const void * in, float * y, int64_t k) {
    const uint8_t * b = (const uint8_t *) in;
    if (orc_blck_size(type) == 32) {
        for (int64_t i = 0; i < k / 32; ++i, y += 32, b += orc_type_size(type)) {
            switch (type) {
                case ORC_Q4_0: { const float d = rd_f16(b);                          /* ggml.c:1509-1527 */
                    for (int j = 0; j < 16; ++j) { y[j] = (float)((b[2 + j] & 15) - 8) * d; y[j + 16] = (float)((b[2 + j] >> 4) - 8) * d; } } break;
                case ORC_Q4_1: { const float d = rd_f16(b), m = rd_f16(b + 2);       /* ggml.c:1529-1548 */
                    for (int j = 0; j < 16; ++j) { y[j] = (float)(b[4 + j] & 15) * d + m; y[j + 16] = (float)(b[4 + j] >> 4) * d + m; } } break;
                case ORC_Q5_0: { const float d = rd_f16(b); const uint32_t qh = rd_u32(b + 2);   /* ggml.c:1550-1574 */
                    for (int j = 0; j < 16; ++j) {
                        y[j]      = (float)(((b[6 + j] & 15) | (q5_bit(qh, j)      << 4)) - 16) * d;
                        y[j + 16] = (float)(((b[6 + j] >> 4) | (q5_bit(qh, j + 16) << 4)) - 16) * d; } } break;
                case ORC_Q5_1: { const float d = rd_f16(b), m = rd_f16(b + 2); const uint32_t qh = rd_u32(b + 4);  /* ggml.c:1576-1601 */
                    for (int j = 0; j < 16; ++j) {
                        y[j]      = (float)((b[8 + j] & 15) | (q5_bit(qh, j)      << 4)) * d + m;
                        y[j + 16] = (float)((b[8 + j] >> 4) | (q5_bit(qh, j + 16) << 4)) * d + m; } } break;
                case ORC_Q8_0: { const float d = rd_f16(b);                          /* ggml.c:1603-1619 */
                    for (int j = 0; j < 32; ++j) y[j] = (float)((const int8_t *) b)[2 + j] * d; } break;
                default: abort();
            }
        }
        return;
    }
    for (int64_t i = 0; i < k / 256; ++i, y += 256, b += orc_type_size(type)) {
        switch (type) {
            case ORC_Q2_K: {                                                         /* k_quants.c:344-375 */
                const float d = rd_f16(b + 80), dmin = rd_f16(b + 82);
                for (int e = 0; e < 256; ++e) {
                    const int sb = (e / 128) * 8 + ((e % 128) / 32) * 2 + ((e % 32) / 16);
                    const int sc = b[sb];
                    const int q  = (b[16 + (e / 128) * 32 + e % 32] >> (2 * ((e % 128) / 32))) & 3;
                    const float dl = d * (float)(sc & 0x0F), ml = dmin * (float)(sc >> 4);
                    y[e] = dl * (float) q - ml;
                } } break;
            case ORC_Q3_K: {                                                         /* k_quants.c:472-521 */
                const float d = rd_f16(b + 108);
                int sc[16]; q3_scales(b + 96, sc);
                for (int e = 0; e < 256; ++e) {
                    const int sb = (e / 128) * 8 + ((e % 128) / 32) * 2 + ((e % 32) / 16);
                    const int lo = (b[32 + (e / 128) * 32 + e % 32] >> (2 * ((e % 128) / 32))) & 3;
                    const int hb = (b[e % 32] >> (e / 32)) & 1;
                    const float dl = d * (float)(sc[sb] - 32);
                    y[e] = dl * (float)(lo - (hb ? 0 : 4));
                } } break;
            case ORC_Q4_K: {                                                         /* k_quants.c:607-631 */
                const float d = rd_f16(b), dmin = rd_f16(b + 2);
                for (int e = 0; e < 256; ++e) {
                    const int c = e / 64, hi = (e % 64) / 32;
                    int sc, mn; k4_scale_min(b + 4, 2 * c + hi, &sc, &mn);
                    const int byte = b[16 + 32 * c + e % 32];
                    const int q = hi ? (byte >> 4) : (byte & 15);
                    const float dl = d * (float) sc, ml = dmin * (float) mn;
                    y[e] = dl * (float) q - ml;
                } } break;
            case ORC_Q5_K: {                                                         /* k_quants.c:734-760 */
                const float d = rd_f16(b), dmin = rd_f16(b + 2);
                for (int e = 0; e < 256; ++e) {
                    const int c = e / 64, hi = (e % 64) / 32;
                    int sc, mn; k4_scale_min(b + 4, 2 * c + hi, &sc, &mn);
                    const int byte = b[48 + 32 * c + e % 32];
                    const int q = (hi ? (byte >> 4) : (byte & 15)) + (((b[16 + e % 32] >> (e / 32)) & 1) ? 16 : 0);
                    const float dl = d * (float) sc, ml = dmin * (float) mn;
                    y[e] = dl * (float) q - ml;
                } } break;
            case ORC_Q6_K: {                                                         /* k_quants.c:845-876 */
                const float d = rd_f16(b + 208);
                const int8_t * sc = (const int8_t *)(b + 192);
                for (int e = 0; e < 256; ++e) {
                    const int h = e / 128, t = (e % 128) / 32, l = e % 32;
                    const int byte = b[64 * h + 32 * (t & 1) + l];
                    const int lo = (t < 2) ? (byte & 15) : (byte >> 4);
                    const int hi = (b[128 + 32 * h + l] >> (2 * t)) & 3;
                    const int q  = (int)(int8_t)(lo | (hi << 4)) - 32;
                    y[e] = d * (float) sc[8 * h + 2 * t + l / 16] * (float) q;      /* (d*sc)*q, left to right */
                } } break;
            case ORC_Q8_K: {                                                         /* k_quants.c:936-945 */
                const float d = rd_f32(b);
                for (int e = 0; e < 256; ++e) y[e] = d * (float)((const int8_t *) b)[4 + e];
                } break;
            default: abort();
        }
    }
}

/* ------------------------------------------------------------------ integer dot products */
/* Test knob: association of the f32 sum over blocks. 0 = the reference's scalar left-to-right loop; 1 = 64 strided
 * partial sums + xor butterfly in the order 1,2,4,..,32 (what the kernels' 64-lane wave reduction does). Used only to MEASURE how far a legitimate re-association moves
 * the logits of a whole model (tests/test_oracle_spread.py); the oracle proper always runs with 0. */
static int g_sum_order = 0;       /* effective order for the current mat-mul */
static int g_sum_mode  = 0;       /* 0 scalar, 1 wave, 2 = as the HIP backend (below), 3 = four interleaved partial sums */
/* order 2 = what the backend's prefill GEMM does (ggllm.cpp_amd/csrc/kernels_gemm.hip): g_split interleaved partial sums,
 * P_s = blocks s, s + g_split, ... left to right, result ((P0 + P1) + P2) + P3; 4 of them on matrices with fewer than
 * 4 x 256 (CUs of an MI355X) 32 x 32 tiles, 2 above. mode 2 picks per mat-mul like the backend: wave order for N <= 4
 * columns, order 2 for GEMMs (Q4_K with 5..80 columns at model widths: four partial sums per segment of 32 super-blocks, the small-batch form). mode 3 / 4 / 5: order 2 with 4 / 2 / 1 partial sums for every mat-mul (5 = ggml_hip_gemm_sequential: the
 * legacy formats' reference order; for the k-quants one d * isum - dmin * msum term per super-block, left to right). */
static int g_split = 4;
/* order 2 only, > 0: the partial sums restart every g_kseg super-blocks and the segments' values ((P0 + P1) + P2) + P3 are added left to right
 * (the backend's small-batch form for Q4_K, k_gemm_skinny_q4k: segments of 32 super-blocks; set per mat-mul by mode 2) */
static int g_kseg = 0;
/* mode 2 only: > 0 = decide as if the mat-mul had this many columns (the sampled-token checks evaluate a few tokens OF a batch:
 * the backend chose its order for the whole batch) */
static int g_backend_batch = 0;
void orc_set_backend_batch(int n) { g_backend_batch = n > 0 ? n : 0; }
/* mode 2 only: the fewest columns at which the k-quants' small-batch form is taken. 5 = a mat-mul on its own (ggml_hip_mul_mat_q, falcon_hip_eval: up to 4 columns
 * go through the mat-vec kernels); 3 = a lock-step context (falcon_hip_context_create_seqs: fq_mul_mat_q_acts_from3, ggllm.cpp_amd/csrc/ggml_hip_ops.hip) */
static int g_kq_min_cols = 5;
void orc_set_kq_min_cols(int n) { g_kq_min_cols = n >= 1 ? n : 5; }
int  orc_backend_batch(void) { return g_backend_batch; }
int orc_attn_backend_order(void) { return g_sum_mode >= 2 && g_sum_mode <= 5; }   /* modes 2, 3, 4, 5 = "as the backend" for the attention too */
/* mode 6 (round 5) = the YARDSTICK, not an order any build runs: every reduction of the path -- a row's sum over its blocks, the attention's two dot products,
 * LayerNorm's and soft_max's sums -- accumulated in f64 from exactly converted terms (the integer dots, table look-ups, activation quantizers and every
 * elementwise f32 step stay the reference's), rounded to f32 once where the reference stores an f32 tensor. What the different f32 associations (the reference's
 * scalar and AVX2 builds, the backend's default and reference orders) are measured against: bench.py `parity.err_vs_f64`, tests/test_gpu_parity_f64.py. */
int orc_exact_f64(void) { return g_sum_mode == 6; }
void orc_set_sum_order(int mode) {
    g_sum_mode = mode; g_sum_order = (mode == 1) ? 1 : ((mode == 3 || mode == 4 || mode == 5) ? 2 : (mode == 6 ? 3 : 0));
    g_split = (mode == 4) ? 2 : ((mode == 5) ? 1 : 4);
    g_kseg = 0;
}

/* ------------------------------------------------------------------ k-quant dots against Q8_K
 * A super-block decoded to integers: qv[e] = the integer weight of element e (the reference's aux8[]), sc16[b] / mn16[b] =
 * integer scale / min of the 16-element block b (a 32-element sub-block's pair repeated), d / dmin = its fp16 factors.
 * Integer parts are exact; only the association of the float sums differs between the orders below (and between the
 * reference's scalar and AVX2 branches). With i16[b] = the unscaled integer dot of block b and bs16[b] = Q8_K's block sum:
 * order 0 = the reference's scalar branches, bit for bit: Q2_K sumf += d dy * sum_b sc16 i16 - dmin dy * sum_b mn16 bs16
 * left to right over the super-blocks (k_quants.c:1303); Q3_K .. Q6_K eight float lanes, element e -> lane e mod 8,
 * sums[l] += (d dy) * aux32[l] per super-block, mins subtracted from sumf per super-block, lanes added 0..7 at the end
 * (k_quants.c:1733-1743, 2044-2053, 2389-2398, 2780-2787).
 * order 2 (the backend's prefill GEMM): the super-block's eight 32-element groups are dealt to g_split partial sums
 * (group g -> g mod g_split), each adds (d dy) * (its groups' integer sum) per super-block, the last one with the mins
 * term, (d dy) isum_s - (dmin dy) msum; result ((P0 + P1) + P2) + P3.
 * order 1 (the backend's mat-vec kernels, fq_units.h): the row is cut into UNITS of 2 or 4 blocks (below), unit u goes
 * to lane u mod 64 as one f32 term (d dy) isum_u - (dmin dy) msum_u, lanes add their units in ascending order and are
 * combined by the xor butterfly 1, 2, 4, .., 32. */
typedef struct { int8_t qv[256]; int16_t sc16[16], mn16[16]; float d, dmin; int has_min; } kq_sb;

static void kq_decode_sb(int wtype, const uint8_t * w, kq_sb * o) {
    memset(o->mn16, 0, sizeof(o->mn16));
    o->has_min = 0; o->dmin = 0.0f;
    switch (wtype) {
        case ORC_Q2_K: {                                                         /* k_quants.c:1267-1306 */
            for (int b = 0; b < 16; ++b) { o->sc16[b] = w[b] & 15; o->mn16[b] = w[b] >> 4; }
            for (int e = 0; e < 256; ++e)
                o->qv[e] = (int8_t)((w[16 + (e / 128) * 32 + e % 32] >> (2 * ((e % 128) / 32))) & 3);
            o->d = rd_f16(w + 80); o->dmin = rd_f16(w + 82); o->has_min = 1; } break;
        case ORC_Q3_K: {                                                         /* k_quants.c:1684-1746 */
            int sc[16]; q3_scales(w + 96, sc);
            for (int b = 0; b < 16; ++b) o->sc16[b] = (int16_t)(sc[b] - 32);
            for (int e = 0; e < 256; ++e) {
                const int lo = (w[32 + (e / 128) * 32 + e % 32] >> (2 * ((e % 128) / 32))) & 3;
                const int hb = (w[e % 32] >> (e / 32)) & 1;
                o->qv[e] = (int8_t)(lo - (hb ? 0 : 4));
            }
            o->d = rd_f16(w + 108); } break;
        case ORC_Q4_K: case ORC_Q5_K: {                                          /* k_quants.c:1999-2055, 2340-2400 */
            const int qs_off = (wtype == ORC_Q4_K) ? 16 : 48;
            for (int b = 0; b < 16; ++b) { int sc, mn; k4_scale_min(w + 4, b / 2, &sc, &mn); o->sc16[b] = (int16_t) sc; o->mn16[b] = (int16_t) mn; }
            for (int e = 0; e < 256; ++e) {
                const int c = e / 64, hi = (e % 64) / 32;
                const int byte = w[qs_off + 32 * c + e % 32];
                int q = hi ? (byte >> 4) : (byte & 15);
                if (wtype == ORC_Q5_K) q += (((w[16 + e % 32] >> (e / 32)) & 1) ? 16 : 0);
                o->qv[e] = (int8_t) q;
            }
            o->d = rd_f16(w); o->dmin = rd_f16(w + 2); o->has_min = 1; } break;
        case ORC_Q6_K: {                                                         /* k_quants.c:2748-2789 */
            const int8_t * sc = (const int8_t *)(w + 192);
            for (int b = 0; b < 16; ++b) o->sc16[b] = sc[b];
            for (int e = 0; e < 256; ++e) {
                const int h = e / 128, t = (e % 128) / 32, l = e % 32;
                const int byte = w[64 * h + 32 * (t & 1) + l];
                const int lo = (t < 2) ? (byte & 15) : (byte >> 4);
                const int hi = (w[128 + 32 * h + l] >> (2 * t)) & 3;
                o->qv[e] = (int8_t)((int)(int8_t)(lo | (hi << 4)) - 32);
            }
            o->d = rd_f16(w + 208); } break;
        default: abort();
    }
}

/* a = the column's Q8_K blocks (292 bytes each) */
static float kq_dot_row(int wtype, int64_t nsb, const kq_sb * row, const uint8_t * a) {
    if (g_sum_order == 3) {                                                   /* the f64 yardstick: sum_i d dy isum_i - dmin dy msum_i, every product and sum in f64 */
        double acc = 0.0;
        for (int64_t i = 0; i < nsb; ++i, a += 292) {
            const kq_sb * sb = &row[i];
            const double dy = (double) rd_f32(a);
            const int8_t * q8 = (const int8_t *)(a + 4);
            long is = 0, ms = 0;
            for (int b = 0; b < 16; ++b) {
                int t = 0;
                for (int j = 0; j < 16; ++j) t += sb->qv[16 * b + j] * q8[16 * b + j];
                is += (long) sb->sc16[b] * t;
                ms += (long) sb->mn16[b] * rd_i16(a + 260 + 2 * b);
            }
            acc += (double) sb->d * dy * (double) is;
            if (sb->has_min) acc -= (double) sb->dmin * dy * (double) ms;
        }
        return (float) acc;
    }
    float sumf = 0.0f;
    float part[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    float lane[64] = {0};
    float lanes8[8] = {0};
    int   eight = 0;
    float seg_tot = 0.0f; int have_seg = 0;
    const int split = (g_sum_order == 2 && nsb > 0) ? g_split : 0;
    const int wave  = (g_sum_order == 1 && nsb > 0);
    int64_t unit = 0;
    for (int64_t i = 0; i < nsb; ++i, a += 292) {
        const kq_sb * sb = &row[i];
        const float  dy = rd_f32(a);
        const int8_t * q8 = (const int8_t *)(a + 4);
        const int16_t * sc16 = sb->sc16, * mn16 = sb->mn16;
        const int has_min = sb->has_min;
        int i16[16], bs16[16];
        /* Q2_K: dall = y.d * d, dmin = y.d * dmin (k_quants.c:1282-1283); the others d * y.d -- the same product */
        const float dd = (wtype == ORC_Q2_K) ? dy * sb->d : sb->d * dy;
        const float dmn = (wtype == ORC_Q2_K) ? dy * sb->dmin : sb->dmin * dy;
        for (int b = 0; b < 16; ++b) bs16[b] = rd_i16(a + 260 + 2 * b);
        for (int b = 0; b < 16; ++b) {
            int t = 0;
            for (int j = 0; j < 16; ++j) t += sb->qv[16 * b + j] * q8[16 * b + j];
            i16[b] = t;
        }
        if (wave) {
            /* units (fq_units.h): Q2_K / Q3_K: 4 per super-block, unit (hf, g) = blocks 8 hf + 2 j + g, j = 0..3;
             * Q4_K / Q5_K: 8, unit (c, g) = blocks 4 c + g and 4 c + g + 2; Q6_K: 8, unit (h, t, g) = blocks 8 h + 2 t + g, + 4 */
            const int four = (wtype == ORC_Q2_K || wtype == ORC_Q3_K);
            for (int uu = 0; uu < (four ? 4 : 8); ++uu, ++unit) {
                int blk[4], nb;
                if (four) { nb = 4; for (int j = 0; j < 4; ++j) blk[j] = 8 * (uu >> 1) + 2 * j + (uu & 1); }
                else if (wtype == ORC_Q6_K) { nb = 2; blk[0] = 8 * (uu >> 2) + 2 * ((uu >> 1) & 1) + (uu & 1); blk[1] = blk[0] + 4; }
                else { nb = 2; blk[0] = 4 * (uu >> 1) + (uu & 1); blk[1] = blk[0] + 2; }
                int is = 0, ms = 0;
                for (int j = 0; j < nb; ++j) { is += sc16[blk[j]] * i16[blk[j]]; ms += mn16[blk[j]] * bs16[blk[j]]; }
                const float v = has_min ? dd * (float) is - dmn * (float) ms : dd * (float) is;
                lane[unit & 63] += v;
            }
            continue;
        }
        int msum = 0;
        for (int b = 0; b < 16; ++b) msum += mn16[b] * bs16[b];
        if (split) {
            for (int sp = 0; sp < split; ++sp) {
                int is = 0;
                for (int g = sp; g < 8; g += split) is += sc16[2 * g] * i16[2 * g] + sc16[2 * g + 1] * i16[2 * g + 1];
                const float A = dd * (float) is;
                part[sp] = part[sp] + ((has_min && sp == split - 1) ? (A - dmn * (float) msum) : A);
            }
            if (g_kseg > 0 && (i + 1) % g_kseg == 0 && i + 1 < nsb) {         /* a segment ends here and another follows */
                const float one = ((part[0] + part[1]) + part[2]) + part[3];
                seg_tot = have_seg ? seg_tot + one : one; have_seg = 1;
                part[0] = part[1] = part[2] = part[3] = 0.0f;
            }
        } else if (wtype == ORC_Q2_K) {                                       /* k_quants.c:1303 */
            int is = 0;
            for (int b = 0; b < 16; ++b) is += sc16[b] * i16[b];
            sumf += dd * (float) is - dmn * (float) msum;
        } else {
            int aux32[8] = {0};
            for (int b = 0; b < 16; ++b)
                for (int h = 0; h < 2; ++h)
                    for (int l = 0; l < 8; ++l) aux32[l] += sc16[b] * (sb->qv[16 * b + 8 * h + l] * q8[16 * b + 8 * h + l]);
            for (int l = 0; l < 8; ++l) lanes8[l] += dd * (float) aux32[l];
            if (has_min) sumf -= dmn * (float) msum;
            eight = 1;
        }
    }
    if (eight) for (int l = 0; l < 8; ++l) sumf += lanes8[l];
    if (wave) {
        for (int o = 1; o < 64; o <<= 1) { float t[64]; for (int l = 0; l < 64; ++l) t[l] = lane[l] + lane[l ^ o]; memcpy(lane, t, sizeof(t)); }
        return lane[0];
    }
    if (split) { const float one = ((part[0] + part[1]) + part[2]) + part[3]; return have_seg ? seg_tot + one : one; }
    return sumf;
}

/* one legacy block against its Q8_0 / Q8_1 block: d_w * d_x (one f32 product), the exact integer sum (offsets folded in: Q4_0 / Q5_0), and for Q4_1 / Q5_1 the
 * pair (m_w, s_x) -- the operands of the FMA form above */
static void orc_legacy_block_parts(int wtype, const uint8_t * w, const uint8_t * a, float * dd, int * sumi_out, float * m, float * sx) {
    const int at = orc_vec_dot_type(wtype);
    const int8_t * q8 = (const int8_t *)(a + (at == ORC_Q8_0 ? 2 : 8));
    int sumi = 0;
    switch (wtype) {
        case ORC_Q4_0:
            for (int j = 0; j < 16; ++j) sumi += ((w[2 + j] & 15) - 8) * q8[j] + ((w[2 + j] >> 4) - 8) * q8[j + 16];
            *dd = rd_f16(w) * rd_f16(a); break;
        case ORC_Q4_1:
            for (int j = 0; j < 16; ++j) sumi += (w[4 + j] & 15) * q8[j] + (w[4 + j] >> 4) * q8[j + 16];
            *dd = rd_f16(w) * rd_f32(a); *m = rd_f16(w + 2); *sx = rd_f32(a + 4); break;
        case ORC_Q5_0: { const uint32_t qh = rd_u32(w + 2);
            for (int j = 0; j < 16; ++j)
                sumi += (((w[6 + j] & 15) | (q5_bit(qh, j) << 4)) - 16) * q8[j] + (((w[6 + j] >> 4) | (q5_bit(qh, j + 16) << 4)) - 16) * q8[j + 16];
            *dd = rd_f16(w) * rd_f16(a); } break;
        case ORC_Q5_1: { const uint32_t qh = rd_u32(w + 4);
            for (int j = 0; j < 16; ++j)
                sumi += ((w[8 + j] & 15) | (q5_bit(qh, j) << 4)) * q8[j] + ((w[8 + j] >> 4) | (q5_bit(qh, j + 16) << 4)) * q8[j + 16];
            *dd = rd_f16(w) * rd_f32(a); *m = rd_f16(w + 2); *sx = rd_f32(a + 4); } break;
        case ORC_Q8_0:
            for (int j = 0; j < 32; ++j) sumi += ((const int8_t *) w)[2 + j] * q8[j];
            *dd = rd_f16(w) * rd_f16(a); break;
        default: abort();
    }
    *sumi_out = sumi;
}

float orc_vec_dot(int wtype, int64_t n, const void * wv, const void * av) {
    const uint8_t * w = (const uint8_t *) wv;
    const uint8_t * a = (const uint8_t *) av;
    float sumf = 0.0f;
    if (orc_blck_size(wtype) == 32 && g_sum_order == 3) {                     /* the f64 yardstick: sum_i d_w d_x isum_i (+ m_w s_x), every product and sum in f64 */
        const int at = orc_vec_dot_type(wtype);
        double acc = 0.0;
        for (int64_t i = 0; i < n / 32; ++i) {
            const uint8_t * wb = w + i * orc_type_size(wtype), * ab = a + i * orc_type_size(at);
            float dd, ms_m = 0.0f, ms_s = 0.0f; int sumi;
            orc_legacy_block_parts(wtype, wb, ab, &dd, &sumi, &ms_m, &ms_s);      /* (dd is not used: the two scales are multiplied in f64 below) */
            const double dw = (double) rd_f16(wb), dx = (at == ORC_Q8_0) ? (double) rd_f16(ab) : (double) rd_f32(ab);
            acc += dw * dx * (double) sumi;
            if (wtype == ORC_Q4_1 || wtype == ORC_Q5_1) acc += (double) ms_m * (double) ms_s;
        }
        return (float) acc;
    }
    if (orc_blck_size(wtype) == 32 && g_sum_order == 1 && n > 32) {
        const int at = orc_vec_dot_type(wtype);
        float lane[64] = {0};
        for (int64_t i = 0; i < n / 32; ++i) {
            const float one = orc_vec_dot(wtype, 32, w + i * orc_type_size(wtype), a + i * orc_type_size(at));
            lane[i & 63] += one;
        }
        /* xor butterfly, pairing order 1, 2, 4, ..., 32 (every lane ends with the total; lane 0 is returned) */
        for (int o = 1; o < 64; o <<= 1) { float t[64]; for (int l = 0; l < 64; ++l) t[l] = lane[l] + lane[l ^ o]; memcpy(lane, t, sizeof(t)); }
        return lane[0];
    }
    if (orc_blck_size(wtype) == 32 && g_sum_order == 2 && n > 32) {
        const int at = orc_vec_dot_type(wtype);
        float part[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int64_t i = 0; i < n / 32; ++i) {
            const uint8_t * wb = w + i * orc_type_size(wtype), * ab = a + i * orc_type_size(at);
            float * P = &part[i & (g_split - 1)];
            if (g_split > 1) {
                /* the K-split partial sums accumulate as the reference's AVX2 build does (ggml.c:2415-2438: acc = _mm256_fmadd_ps(d, q, acc)): one fused
                 * multiply-add of (d_w * d_x) and the block's integer sum -- and one of (m_w, s_x) for the formats with a minimum -- per block
                 * (kernels_gemm.hip / kernels_gemm_skinny.hip, S > 1). The single sum (g_split == 1: the reference ORDER) keeps the scalar build's
                 * two roundings per term. */
                float dd, ms_m = 0.0f, ms_s = 0.0f; int sumi;
                orc_legacy_block_parts(wtype, wb, ab, &dd, &sumi, &ms_m, &ms_s);
                *P = fmaf(dd, (float) sumi, *P);
                if (wtype == ORC_Q4_1 || wtype == ORC_Q5_1) *P = fmaf(ms_m, ms_s, *P);
            } else {
                const float one = orc_vec_dot(wtype, 32, wb, ab);
                *P = *P + one;
            }
        }
        return ((part[0] + part[1]) + part[2]) + part[3];
    }
    if (orc_blck_size(wtype) == 32) {
        const int at = orc_vec_dot_type(wtype);
        for (int64_t i = 0; i < n / 32; ++i, w += orc_type_size(wtype), a += orc_type_size(at)) {
            const int8_t * q8 = (const int8_t *)(a + (at == ORC_Q8_0 ? 2 : 8));
            int sumi = 0;
            switch (wtype) {
                case ORC_Q4_0:                                                       /* ggml.c:2591-2609 */
                    for (int j = 0; j < 16; ++j) sumi += ((w[2 + j] & 15) - 8) * q8[j] + ((w[2 + j] >> 4) - 8) * q8[j + 16];
                    sumf += (float) sumi * rd_f16(w) * rd_f16(a);
                    break;
                case ORC_Q4_1:                                                       /* ggml.c:2716-2735 */
                    for (int j = 0; j < 16; ++j) sumi += (w[4 + j] & 15) * q8[j] + (w[4 + j] >> 4) * q8[j + 16];
                    sumf += (rd_f16(w) * rd_f32(a)) * (float) sumi + rd_f16(w + 2) * rd_f32(a + 4);
                    break;
                case ORC_Q5_0: { const uint32_t qh = rd_u32(w + 2);                  /* ggml.c:2951-2972 */
                    for (int j = 0; j < 16; ++j)
                        sumi += (((w[6 + j] & 15) | (q5_bit(qh, j) << 4)) - 16) * q8[j]
                              + (((w[6 + j] >> 4) | (q5_bit(qh, j + 16) << 4)) - 16) * q8[j + 16];
                    sumf += (rd_f16(w) * rd_f16(a)) * (float) sumi; } break;
                case ORC_Q5_1: { const uint32_t qh = rd_u32(w + 4);                  /* ggml.c:3207-3228 */
                    for (int j = 0; j < 16; ++j)
                        sumi += ((w[8 + j] & 15) | (q5_bit(qh, j) << 4)) * q8[j]
                              + ((w[8 + j] >> 4) | (q5_bit(qh, j + 16) << 4)) * q8[j + 16];
                    sumf += (rd_f16(w) * rd_f32(a)) * (float) sumi + rd_f16(w + 2) * rd_f32(a + 4); } break;
                case ORC_Q8_0:                                                       /* ggml.c:3317-3329 */
                    for (int j = 0; j < 32; ++j) sumi += ((const int8_t *) w)[2 + j] * q8[j];
                    sumf += (float) sumi * (rd_f16(w) * rd_f16(a));
                    break;
                default: abort();
            }
        }
        return sumf;
    }
    /* k-quants against Q8_K: decode the row's super-blocks once (kq_decode_sb), then kq_dot_row */
    {
        const int64_t nsb = n / 256;
        kq_sb * row = (kq_sb *) malloc(sizeof(kq_sb) * (size_t)(nsb ? nsb : 1));
        for (int64_t i = 0; i < nsb; ++i) kq_decode_sb(wtype, w + (size_t) i * orc_type_size(wtype), &row[i]);
        sumf = kq_dot_row(wtype, nsb, row, a);
        free(row);
        return sumf;
    }
}

/* ------------------------------------------------------------------ quantized mat-mul (ggml.c:11318-11529) */
typedef struct {
    int wtype; const uint8_t * w; int64_t K, M, N; const uint8_t * act; size_t act_row; float * dst;
    int ith, nth;
} mm_job;

static void * mm_worker(void * arg) {
    const mm_job * j = (const mm_job *) arg;
    const size_t wrow = orc_row_bytes(j->wtype, j->K);
    const int64_t dr = (j->M + j->nth - 1) / j->nth;               /* rows of W split evenly over threads */
    const int64_t r0 = dr * j->ith, r1 = (r0 + dr < j->M) ? r0 + dr : j->M;
    if (orc_blck_size(j->wtype) == 256) {                          /* k-quants: decode a weight row once for all N columns */
        const int64_t nsb = j->K / 256;
        kq_sb * row = (kq_sb *) malloc(sizeof(kq_sb) * (size_t)(nsb ? nsb : 1));
        for (int64_t r = r0; r < r1; ++r) {
            for (int64_t i = 0; i < nsb; ++i) kq_decode_sb(j->wtype, j->w + (size_t) r * wrow + (size_t) i * orc_type_size(j->wtype), &row[i]);
            for (int64_t n = 0; n < j->N; ++n) j->dst[n * j->M + r] = kq_dot_row(j->wtype, nsb, row, j->act + (size_t) n * j->act_row);
        }
        free(row);
        return NULL;
    }
    for (int64_t n = 0; n < j->N; ++n) {
        const uint8_t * col = j->act + (size_t) n * j->act_row;
        for (int64_t r = r0; r < r1; ++r) {
            j->dst[n * j->M + r] = orc_vec_dot(j->wtype, j->K, j->w + (size_t) r * wrow, col);
        }
    }
    return NULL;
}

void orc_mul_mat_q(int wtype, const void * w, int64_t K, int64_t M, const float * x, int64_t N,
                   float * dst, int n_threads, int flavour) {
    const int at = orc_vec_dot_type(wtype);
    const size_t act_row = orc_row_bytes(at, K);
    if (g_sum_mode == 2) {
        const int64_t Nb = g_backend_batch > 0 ? g_backend_batch : N;
        g_sum_order = (Nb <= 4) ? 1 : 2;
        g_split = (((M + 31) / 32) * ((Nb + 31) / 32) < 4 * 256) ? 4 : 2;
        /* Q4_K / Q5_K with 5..80 columns: the backend's small-batch form (k_gemm_skinny_q4k): four partial sums per segment of 32 super-blocks */
        g_kseg = 0;
        if ((wtype == ORC_Q4_K || wtype == ORC_Q5_K) && Nb >= g_kq_min_cols && Nb <= 80 && M % 16 == 0 && K / 256 >= 8 &&
            ((K / 256 + 31) / 32) * 64 * ((M + 63) / 64 * 64) <= ((int64_t) 16 << 20)) { g_split = 4; g_kseg = 32; g_sum_order = 2; }
        /* Q2_K, Q3_K (k_gemm_skinny_q2k) and Q6_K (k_gemm_skinny_q6k, up to 80 columns) likewise: segments of 16 super-blocks */
        if ((wtype == ORC_Q2_K || wtype == ORC_Q3_K || (wtype == ORC_Q6_K && Nb <= 80)) && Nb >= g_kq_min_cols && Nb <= 112 && M % 16 == 0 && K / 256 >= 8 &&
            ((K / 256 + 15) / 16) * 64 * ((M + 63) / 64 * 64) <= ((int64_t) 16 << 20)) { g_split = 4; g_kseg = 16; g_sum_order = 2; }
    }
    uint8_t * act = (uint8_t *) malloc(act_row * (size_t) N);
    /* INIT phase: every src1 row is quantized by one thread (ggml.c:11462-11476) */
    for (int64_t n = 0; n < N; ++n) orc_quantize_act(at, x + n * K, act + (size_t) n * act_row, K, flavour);
    if (n_threads < 1) n_threads = 1;
    mm_job   * jobs = (mm_job *) malloc(sizeof(mm_job) * (size_t) n_threads);
    pthread_t * th  = (pthread_t *) malloc(sizeof(pthread_t) * (size_t) n_threads);
    for (int t = 0; t < n_threads; ++t) {
        jobs[t] = (mm_job){ wtype, (const uint8_t *) w, K, M, N, act, act_row, dst, t, n_threads };
        if (t > 0) pthread_create(&th[t], NULL, mm_worker, &jobs[t]);
    }
    mm_worker(&jobs[0]);
    for (int t = 1; t < n_threads; ++t) pthread_join(th[t], NULL);
    free(th); free(jobs); free(act);
}
