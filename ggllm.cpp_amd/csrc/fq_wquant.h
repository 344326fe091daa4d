// fq_wquant.h -- weight quantizers: f32 rows -> ggml blocks, the arithmetic of the reference's
// quantize_row_q*_reference restated per block so that one GPU thread fits one (sub-)block. Host + device (the host
// build is tests/host/wquant_harness.cpp, checked against golden vectors of the reference and the reference build itself).
//
//   legacy formats   quantize_row_q4_0/q4_1/q5_0/q5_1/q8_0_reference   ggml.c:927-1129
//   k-quants         quantize_row_q{2,3,4,5,6}_K_reference             k_quants.c:275-343, 396-471, 542-606, 652-733, 781-844
//                    over the three sub-block fits make_qkx1_quants / make_q3_quants / make_qx_quants (k_quants.c:57-262)
//
// A k-quant super-block is quantized in four steps, each a pure function here:
//   1. fit       one sub-block (16 or 32 weights) -> f32 scale (and min) + provisional levels L
//   2. header    the 8 / 16 sub-block scales of a super-block -> packed 4/6/8-bit scales + fp16 d (dmin)
//   3. requant   one sub-block again with the header's dequantized scale -> final levels L
//   4. pack      byte i of the ggml block from the header and the 256 levels
// Everything is IEEE f32 without contraction (the reference is ISO C: two roundings per a*b+c); division is correctly
// rounded on the device (hipcc default) as on the host.
//
// One deliberate difference: make_qkx1_quants compares its first iteration's levels with whatever its caller's L array
// held before (the previous super-block's levels, or uninitialised stack: k_quants.c:235-241 with L declared at 279 / 546 /
// 656 and never cleared). Here the first iteration always counts as "changed". The two differ only when a sub-block's
// first-iteration levels coincide with that stale content in all 16 / 32 places, where the reference's own output
// depends on what was quantized before on the same thread.
#pragma once
#include "fq_types.h"
#include "fq_units.h"      // fq_h2f
#include <math.h>
#include <string.h>

FQ_HD uint32_t wq_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
FQ_HD float    wq_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

// float -> IEEE half, round to nearest even (ggml_fp32_to_fp16 = F16C _cvtss_sh on the reference's x86 builds, ggml.c:320)
FQ_HD uint16_t fq_f2h(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
    // explicit v_cvt_f16_f32: `(_Float16)(a * b)` is otherwise selected as v_fma_mixlo_f16(a, b, +0), which turns a -0
    // product into +0 (an all-zero Q4_0 block stores d = 0 / -8 = -0 = 0x8000 in the reference)
    uint32_t r;
    asm("v_cvt_f16_f32 %0, %1" : "=v"(r) : "v"(f));
    return (uint16_t) r;
#else
    uint32_t x = wq_bits(f);
    const uint16_t sign = (uint16_t)((x >> 16) & 0x8000u);
    x &= 0x7FFFFFFFu;
    if (x >= 0x47800000u) return (uint16_t)(sign | (x > 0x7F800000u ? 0x7E00u : 0x7C00u));    // >= 65536, inf, NaN
    if (x < 0x38800000u) {                                    // below the smallest normal half: the f32 adder rounds for us
        const float m = wq_float(x) + 0.5f;
        return (uint16_t)(sign | (wq_bits(m) - 0x3F000000u));
    }
    const uint32_t odd = (x >> 13) & 1u;
    x += 0xC8000FFFu;                                         // rebias exponent by (15 - 127) << 23, add 0xFFF
    x += odd;
    return (uint16_t)(sign | (x >> 13));
#endif
}

FQ_HD int wq_nearest_int(float v) {                           // k_quants.c:50-55
    const float t = v + 12582912.0f;
    return (int)(wq_bits(t) & 0x007FFFFFu) - 0x00400000;
}
FQ_HD int wq_clamp(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// ================================================================================================ legacy formats
// one 32-weight block -> ggml block bytes at out; returns nothing. The 16 bins of the reference's quantize histogram
// (ggml_quantize_q4_0 ... q8_0, ggml.c:19352-19477) are reported through hist(bin), one call per weight. Q5_0 / Q5_1: the
// bin is the stored 5-bit value / 2 (the reference's loop reads the wrong qh bits there, ggml.c:19411-19413, so its
// printed Q5 histogram is not reproducible from the stored values; the blocks themselves are identical).
FQ_HD void wq_put16(uint8_t * o, uint16_t v) { o[0] = (uint8_t) v; o[1] = (uint8_t)(v >> 8); }
FQ_HD void wq_put32(uint8_t * o, uint32_t v) { o[0] = (uint8_t) v; o[1] = (uint8_t)(v >> 8); o[2] = (uint8_t)(v >> 16); o[3] = (uint8_t)(v >> 24); }

template <int TYPE, class HIST>
FQ_HD void wq_block_legacy(const float (&x)[32], uint8_t * out, HIST && hist) {
    if constexpr (TYPE == FQ_Q4_0 || TYPE == FQ_Q5_0) {       // symmetric: scale from the signed largest-magnitude weight
        constexpr int LEV = (TYPE == FQ_Q4_0) ? 8 : 16, TOP = 2 * LEV - 1;
        float amax = 0.0f, vmax = 0.0f;
#pragma unroll
        for (int j = 0; j < 32; ++j) { const float a = fabsf(x[j]); if (amax < a) { amax = a; vmax = x[j]; } }
        const float d = vmax / (float)(-LEV);
        const float id = d ? 1.0f / d : 0.0f;
        wq_put16(out, fq_f2h(d));
        uint32_t qh = 0;
        uint8_t * qs = out + (TYPE == FQ_Q4_0 ? 2 : 6);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            int a = (int)(int8_t)(int)(x[j] * id + ((float) LEV + 0.5f));
            int b = (int)(int8_t)(int)(x[j + 16] * id + ((float) LEV + 0.5f));
            a = a > TOP ? TOP : a; b = b > TOP ? TOP : b;
            const uint32_t ua = (uint32_t) a & 0xFFu, ub = (uint32_t) b & 0xFFu;
            qs[j] = (uint8_t)((ua & 0xFu) | ((ub & 0xFu) << 4));
            if constexpr (TYPE == FQ_Q5_0) {
                qh |= ((ua >> 4) & 1u) << j; qh |= ((ub >> 4) & 1u) << (j + 16);
                hist((int)((ua & 0x1Fu) >> 1)); hist((int)((ub & 0x1Fu) >> 1));
            } else { hist((int)(ua & 0xFu)); hist((int)(ub & 0xFu)); }
        }
        if constexpr (TYPE == FQ_Q5_0) wq_put32(out + 2, qh);
    } else if constexpr (TYPE == FQ_Q4_1 || TYPE == FQ_Q5_1) {   // affine: d = (max - min) / (2^bits - 1), m = min
        constexpr int TOP = (TYPE == FQ_Q4_1) ? 15 : 31;
        float mn = 3.402823466e+38f, mx = -3.402823466e+38f;
#pragma unroll
        for (int j = 0; j < 32; ++j) { if (x[j] < mn) mn = x[j]; if (x[j] > mx) mx = x[j]; }
        const float d = (mx - mn) / (float) TOP;
        const float id = d ? 1.0f / d : 0.0f;
        wq_put16(out, fq_f2h(d)); wq_put16(out + 2, fq_f2h(mn));
        uint32_t qh = 0;
        uint8_t * qs = out + (TYPE == FQ_Q4_1 ? 4 : 8);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float fa = (x[j] - mn) * id, fb = (x[j + 16] - mn) * id;
            uint32_t ua, ub;
            if constexpr (TYPE == FQ_Q4_1) {                  // int8 cast, then clamp (ggml.c:993-994)
                const int a = (int)(int8_t)(int)(fa + 0.5f), b = (int)(int8_t)(int)(fb + 0.5f);
                ua = (uint32_t)(a > 15 ? 15 : a) & 0xFFu; ub = (uint32_t)(b > 15 ? 15 : b) & 0xFFu;
                hist((int)(ua & 0xFu)); hist((int)(ub & 0xFu));
            } else {                                          // uint8 cast, no clamp (ggml.c:1080-1081)
                ua = (uint32_t)(int)(fa + 0.5f) & 0xFFu; ub = (uint32_t)(int)(fb + 0.5f) & 0xFFu;
                qh |= ((ua >> 4) & 1u) << j; qh |= ((ub >> 4) & 1u) << (j + 16);
                hist((int)((ua & 0x1Fu) >> 1)); hist((int)((ub & 0x1Fu) >> 1));
            }
            qs[j] = (uint8_t)((ua & 0xFu) | ((ub & 0xFu) << 4));
        }
        if constexpr (TYPE == FQ_Q5_1) wq_put32(out + 4, qh);
    } else {                                                  // Q8_0: d = amax / 127, roundf (ggml.c:1106-1129)
        float amax = 0.0f;
#pragma unroll
        for (int j = 0; j < 32; ++j) { const float a = fabsf(x[j]); if (a > amax) amax = a; }
        const float d = amax / 127.0f;
        const float id = d ? 1.0f / d : 0.0f;
        wq_put16(out, fq_f2h(d));
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            const int q = (int) roundf(x[j] * id);
            out[2 + j] = (uint8_t)(int8_t) q;
            hist((((int)(int8_t) q) / 16 + 8) & 15);
        }
    }
}

// ================================================================================================ k-quants, step 1: fits
// make_qkx1_quants(n, nmax, x, L, &min, ntry = 5) (k_quants.c:214-262): levels 0..nmax over [min, max], min <= 0
template <int N>
FQ_HD float wq_fit_minmax(int nmax, const float (&x)[N], int (&L)[N], float & the_min) {
    float mn = x[0], mx = x[0];
#pragma unroll
    for (int i = 1; i < N; ++i) { if (x[i] < mn) mn = x[i]; if (x[i] > mx) mx = x[i]; }
    if (mx == mn) {
#pragma unroll
        for (int i = 0; i < N; ++i) L[i] = 0;
        the_min = 0.0f;
        return 0.0f;
    }
    if (mn > 0.0f) mn = 0.0f;
#pragma unroll
    for (int i = 0; i < N; ++i) L[i] = 0;
    float iscale = (float) nmax / (mx - mn);
    float scale = 1.0f / iscale;
    for (int itry = 0; itry < 5; ++itry) {
        float sumlx = 0.0f; int suml2 = 0;
        bool changed = (itry == 0);                            // see the header comment
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int l = wq_clamp(wq_nearest_int(iscale * (x[i] - mn)), 0, nmax);
            changed = changed || (l != L[i]);
            L[i] = l;
            sumlx += (x[i] - mn) * (float) l;
            suml2 += l * l;
        }
        scale = sumlx / (float) suml2;
        float sum = 0.0f;
#pragma unroll
        for (int i = 0; i < N; ++i) sum += x[i] - scale * (float) L[i];
        mn = sum / (float) N;
        if (mn > 0.0f) mn = 0.0f;
        iscale = 1.0f / scale;
        if (!changed) break;
    }
    the_min = -mn;
    return scale;
}

// make_q3_quants(16, 4, x, L, do_rmse = true) (k_quants.c:151-212): levels -4..3 stored +4, weights x^2
FQ_HD float wq_fit_q3(const float (&x)[16], int (&L)[16]) {
    float vmax = 0.0f, amax = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) { const float a = fabsf(x[i]); if (a > amax) { amax = a; vmax = x[i]; } }
    if (!amax) {
#pragma unroll
        for (int i = 0; i < 16; ++i) L[i] = 0;
        return 0.0f;
    }
    const float iscale = -4.0f / vmax;
    float sumlx = 0.0f, suml2 = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int l = wq_clamp(wq_nearest_int(iscale * x[i]), -4, 3);
        L[i] = l;
        const float w = x[i] * x[i];
        sumlx += w * x[i] * (float) l;
        suml2 += w * (float) l * (float) l;
    }
    for (int itry = 0; itry < 5; ++itry) {
        int n_changed = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float w = x[i] * x[i];
            float slx = sumlx - w * x[i] * (float) L[i];
            if (slx > 0.0f) {
                float sl2 = suml2 - w * (float) L[i] * (float) L[i];
                const int nl = wq_clamp(wq_nearest_int(x[i] * sl2 / slx), -4, 3);
                if (nl != L[i]) {
                    slx += w * x[i] * (float) nl;
                    sl2 += w * (float) nl * (float) nl;
                    if (sl2 > 0.0f && slx * slx * suml2 > sumlx * sumlx * sl2) { L[i] = nl; sumlx = slx; suml2 = sl2; ++n_changed; }
                }
            }
        }
        if (!n_changed) break;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) L[i] += 4;
    return sumlx / suml2;
}

// make_qx_quants(16, 32, x, L, rmse_type = 1) (k_quants.c:57-149): levels -32..31 stored +32, weights x^2
FQ_HD float wq_fit_q6(const float (&x)[16], int (&L)[16]) {
    float vmax = 0.0f, amax = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) { const float a = fabsf(x[i]); if (a > amax) { amax = a; vmax = x[i]; } }
    if (!amax) {
#pragma unroll
        for (int i = 0; i < 16; ++i) L[i] = 0;
        return 0.0f;
    }
    float iscale = -32.0f / vmax;
    float sumlx = 0.0f, suml2 = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int l = wq_clamp(wq_nearest_int(iscale * x[i]), -32, 31);
        L[i] = l + 32;
        const float w = x[i] * x[i];
        sumlx += w * x[i] * (float) l;
        suml2 += w * (float) l * (float) l;
    }
    float scale = sumlx / suml2;
    float best = scale * sumlx;
    for (int itry = 0; itry < 3; ++itry) {
        iscale = 1.0f / scale;
        float slx = 0.0f, sl2 = 0.0f;
        bool changed = false;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int l = wq_clamp(wq_nearest_int(iscale * x[i]), -32, 31);
            if (l + 32 != L[i]) changed = true;
            const float w = x[i] * x[i];
            slx += w * x[i] * (float) l;
            sl2 += w * (float) l * (float) l;
        }
        if (!changed || sl2 == 0.0f || slx * slx <= best * sl2) break;
#pragma unroll
        for (int i = 0; i < 16; ++i) L[i] = 32 + wq_clamp(wq_nearest_int(iscale * x[i]), -32, 31);
        sumlx = slx; suml2 = sl2;
        scale = sumlx / suml2;
        best = scale * sumlx;
    }
    for (int itry = 0; itry < 5; ++itry) {
        int n_changed = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float w = x[i] * x[i];
            const int l = L[i] - 32;
            float slx = sumlx - w * x[i] * (float) l;
            if (slx > 0.0f) {
                float sl2 = suml2 - w * (float) l * (float) l;
                const int nl = wq_clamp(wq_nearest_int(x[i] * sl2 / slx), -32, 31);
                if (nl != l) {
                    slx += w * x[i] * (float) nl;
                    sl2 += w * (float) nl * (float) nl;
                    if (sl2 > 0.0f && slx * slx * suml2 > sumlx * sumlx * sl2) {
                        L[i] = 32 + nl; sumlx = slx; suml2 = sl2;
                        scale = sumlx / suml2; best = scale * sumlx;
                        ++n_changed;
                    }
                }
            }
        }
        if (!n_changed) break;
    }
    return scale;
}

// sub-block geometry of a k-quant format
template <int TYPE> struct wq_geom {
    static constexpr int N   = (TYPE == FQ_Q4_K || TYPE == FQ_Q5_K) ? 32 : 16;    // weights per sub-block
    static constexpr int NSB = 256 / N;                                           // sub-blocks per super-block
    static constexpr int HDR = (TYPE == FQ_Q2_K) ? 20 : (TYPE == FQ_Q3_K) ? 14 : (TYPE == FQ_Q6_K) ? 18 : 16;
};

template <int TYPE>
FQ_HD void wq_fit(const float (&x)[wq_geom<TYPE>::N], int (&L)[wq_geom<TYPE>::N], float & scale, float & mn) {
    mn = 0.0f;
    if constexpr (TYPE == FQ_Q2_K)      scale = wq_fit_minmax<16>(3, x, L, mn);
    else if constexpr (TYPE == FQ_Q3_K) scale = wq_fit_q3(x, L);
    else if constexpr (TYPE == FQ_Q4_K) scale = wq_fit_minmax<32>(15, x, L, mn);
    else if constexpr (TYPE == FQ_Q5_K) scale = wq_fit_minmax<32>(31, x, L, mn);
    else                                scale = wq_fit_q6(x, L);
}

// ================================================================================================ step 2: header
// hdr layouts (the block's non-quant fields, in the order pack() copies them):
//   Q2_K  [0,16) scales (lo nibble scale, hi nibble min)  [16,18) d  [18,20) dmin
//   Q3_K  [0,12) packed 6-bit scales                      [12,14) d
//   Q4_K / Q5_K  [0,2) d  [2,4) dmin  [4,16) packed 6-bit scales / mins
//   Q6_K  [0,16) int8 scales                              [16,18) d
template <int TYPE>
FQ_HD void wq_header(const float * scales, const float * mins, uint8_t * hdr) {
    constexpr int NSB = wq_geom<TYPE>::NSB;
    if constexpr (TYPE == FQ_Q2_K) {                          // k_quants.c:287-320
        float max_scale = 0.0f, max_min = 0.0f;
        for (int j = 0; j < NSB; ++j) { if (scales[j] > max_scale) max_scale = scales[j]; if (mins[j] > max_min) max_min = mins[j]; }
        if (max_scale > 0.0f) {
            const float iscale = 15.0f / max_scale;
            for (int j = 0; j < NSB; ++j) hdr[j] = (uint8_t) wq_nearest_int(iscale * scales[j]);
            wq_put16(hdr + 16, fq_f2h(max_scale / 15.0f));
        } else {
            for (int j = 0; j < NSB; ++j) hdr[j] = 0;
            wq_put16(hdr + 16, fq_f2h(0.0f));
        }
        if (max_min > 0.0f) {
            const float iscale = 15.0f / max_min;
            for (int j = 0; j < NSB; ++j) hdr[j] = (uint8_t)(hdr[j] | (uint8_t)(wq_nearest_int(iscale * mins[j]) << 4));
            wq_put16(hdr + 18, fq_f2h(max_min / 15.0f));
        } else wq_put16(hdr + 18, fq_f2h(0.0f));
    } else if constexpr (TYPE == FQ_Q3_K) {                   // k_quants.c:404-433
        float max_scale = 0.0f, amax = 0.0f;
        for (int j = 0; j < NSB; ++j) { const float a = fabsf(scales[j]); if (a > amax) { amax = a; max_scale = scales[j]; } }
        for (int j = 0; j < 12; ++j) hdr[j] = 0;
        if (max_scale) {
            const float iscale = -32.0f / max_scale;
            for (int j = 0; j < NSB; ++j) {
                int l = (int)(int8_t) wq_nearest_int(iscale * scales[j]);
                l = wq_clamp(l, -32, 31) + 32;
                if (j < 8) hdr[j] = (uint8_t)(l & 0xF); else hdr[j - 8] = (uint8_t)(hdr[j - 8] | ((l & 0xF) << 4));
                l >>= 4;
                hdr[j % 4 + 8] = (uint8_t)(hdr[j % 4 + 8] | (l << (2 * (j / 4))));
            }
            wq_put16(hdr + 12, fq_f2h(1.0f / iscale));
        } else wq_put16(hdr + 12, fq_f2h(0.0f));
    } else if constexpr (TYPE == FQ_Q4_K || TYPE == FQ_Q5_K) {   // k_quants.c:565-584, 675-694
        float max_scale = 0.0f, max_min = 0.0f;
        for (int j = 0; j < NSB; ++j) { if (scales[j] > max_scale) max_scale = scales[j]; if (mins[j] > max_min) max_min = mins[j]; }
        const float inv_scale = max_scale > 0.0f ? 63.0f / max_scale : 0.0f;
        const float inv_min   = max_min   > 0.0f ? 63.0f / max_min   : 0.0f;
        uint8_t * sc = hdr + 4;
        for (int j = 0; j < NSB; ++j) {
            uint8_t ls = (uint8_t) wq_nearest_int(inv_scale * scales[j]);
            uint8_t lm = (uint8_t) wq_nearest_int(inv_min * mins[j]);
            ls = ls > 63 ? 63 : ls; lm = lm > 63 ? 63 : lm;
            if (j < 4) { sc[j] = ls; sc[j + 4] = lm; }
            else {
                sc[j + 4] = (uint8_t)((ls & 0xF) | ((lm & 0xF) << 4));
                sc[j - 4] = (uint8_t)(sc[j - 4] | ((ls >> 4) << 6));
                sc[j]     = (uint8_t)(sc[j]     | ((lm >> 4) << 6));
            }
        }
        wq_put16(hdr, fq_f2h(max_scale / 63.0f));
        wq_put16(hdr + 2, fq_f2h(max_min / 63.0f));
    } else {                                                  // Q6_K, k_quants.c:790-812
        float max_scale = 0.0f, amax = 0.0f;
        for (int j = 0; j < NSB; ++j) { const float a = fabsf(scales[j]); if (a > amax) { amax = a; max_scale = scales[j]; } }
        const float iscale = -128.0f / max_scale;
        wq_put16(hdr + 16, fq_f2h(1.0f / iscale));
        for (int j = 0; j < NSB; ++j) { const int l = wq_nearest_int(iscale * scales[j]); hdr[j] = (uint8_t)(int8_t)(l > 127 ? 127 : l); }
    }
}

// ================================================================================================ step 3: requantize
// sub-block j of the super-block again, now with the scale the block will dequantize with; a zero scale keeps the fit's
// levels (the reference's `if (!d) continue;`)
template <int TYPE>
FQ_HD void wq_requant(const uint8_t * hdr, int j, const float (&x)[wq_geom<TYPE>::N], int (&L)[wq_geom<TYPE>::N]) {
    constexpr int N = wq_geom<TYPE>::N;
    auto h16 = [&](int o) { return fq_h2f((uint16_t)(hdr[o] | (hdr[o + 1] << 8))); };
    if constexpr (TYPE == FQ_Q2_K) {                          // k_quants.c:321-330
        const float d = h16(16) * (float)(hdr[j] & 0xF);
        if (!d) return;
        const float dm = h16(18) * (float)(hdr[j] >> 4);
#pragma unroll
        for (int i = 0; i < N; ++i) L[i] = wq_clamp(wq_nearest_int((x[i] + dm) / d), 0, 3);
    } else if constexpr (TYPE == FQ_Q3_K) {                   // k_quants.c:435-447
        int sc = j < 8 ? (hdr[j] & 0xF) : (hdr[j - 8] >> 4);
        sc = (int)(int8_t)((sc | (((hdr[8 + j % 4] >> (2 * (j / 4))) & 3) << 4)) - 32);
        const float d = h16(12) * (float) sc;
        if (!d) return;
#pragma unroll
        for (int i = 0; i < N; ++i) L[i] = wq_clamp(wq_nearest_int(x[i] / d), -4, 3) + 4;
    } else if constexpr (TYPE == FQ_Q4_K || TYPE == FQ_Q5_K) {   // k_quants.c:588-599, 698-709; get_scale_min_k4 264-272
        const uint8_t * q = hdr + 4;
        int sc, m;
        if (j < 4) { sc = q[j] & 63; m = q[j + 4] & 63; }
        else { sc = (q[j + 4] & 0xF) | ((q[j - 4] >> 6) << 4); m = (q[j + 4] >> 4) | ((q[j] >> 6) << 4); }
        const float d = h16(0) * (float) sc;
        if (!d) return;
        const float dm = h16(2) * (float) m;
        constexpr int TOP = (TYPE == FQ_Q4_K) ? 15 : 31;
#pragma unroll
        for (int i = 0; i < N; ++i) L[i] = wq_clamp(wq_nearest_int((x[i] + dm) / d), 0, TOP);
    } else {                                                  // Q6_K, k_quants.c:814-824
        const float d = h16(16) * (float)(int)(int8_t) hdr[j];
        if (!d) return;
#pragma unroll
        for (int i = 0; i < N; ++i) L[i] = wq_clamp(wq_nearest_int(x[i] / d), -32, 31) + 32;
    }
}

// ================================================================================================ step 4: pack
// byte i of the ggml block (k_quants.h:20-74) from the header and the super-block's 256 levels
template <int TYPE>
FQ_HD uint8_t wq_pack_byte(const uint8_t * hdr, const uint8_t * L, int i) {
    if constexpr (TYPE == FQ_Q2_K) {                          // scales[16] qs[64] d dmin
        if (i < 16) return hdr[i];
        if (i >= 80) return hdr[16 + (i - 80)];
        const int q = i - 16, h = q >> 5, l = q & 31;
        const uint8_t * p = L + 128 * h + l;
        return (uint8_t)(p[0] | (p[32] << 2) | (p[64] << 4) | (p[96] << 6));
    } else if constexpr (TYPE == FQ_Q3_K) {                   // hmask[32] qs[64] scales[12] d
        if (i < 32) {
            int m = 0;
            for (int b = 0; b < 8; ++b) m |= (L[32 * b + i] > 3 ? 1 : 0) << b;
            return (uint8_t) m;
        }
        if (i < 96) {
            const int q = i - 32, h = q >> 5, l = q & 31;
            const uint8_t * p = L + 128 * h + l;
            return (uint8_t)((p[0] & 3) | ((p[32] & 3) << 2) | ((p[64] & 3) << 4) | ((p[96] & 3) << 6));
        }
        return hdr[i - 96];
    } else if constexpr (TYPE == FQ_Q4_K) {                   // d dmin scales[12] qs[128]
        if (i < 16) return hdr[i];
        const int q = i - 16, c = q >> 5, l = q & 31;
        return (uint8_t)(L[64 * c + l] | (L[64 * c + l + 32] << 4));
    } else if constexpr (TYPE == FQ_Q5_K) {                   // d dmin scales[12] qh[32] qs[128]
        if (i < 16) return hdr[i];
        if (i < 48) {
            const int j = i - 16; int m = 0;
            for (int n = 0; n < 4; ++n) m |= ((L[64 * n + j] > 15 ? 1 : 0) << (2 * n)) | ((L[64 * n + j + 32] > 15 ? 1 : 0) << (2 * n + 1));
            return (uint8_t) m;
        }
        const int q = i - 48, c = q >> 5, l = q & 31;
        return (uint8_t)((L[64 * c + l] & 15) | ((L[64 * c + l + 32] & 15) << 4));
    } else {                                                  // Q6_K: ql[128] qh[64] scales[16] d
        if (i < 128) {
            const int h = i >> 6, r = i & 63, l = r & 31;
            const uint8_t * p = L + 128 * h + l + (r >= 32 ? 32 : 0);
            return (uint8_t)((p[0] & 0xF) | ((p[64] & 0xF) << 4));
        }
        if (i < 192) {
            const int q = i - 128, h = q >> 5, l = q & 31;
            const uint8_t * p = L + 128 * h + l;
            return (uint8_t)((p[0] >> 4) | ((p[32] >> 4) << 2) | ((p[64] >> 4) << 4) | ((p[96] >> 4) << 6));
        }
        return hdr[i - 192];
    }
}
