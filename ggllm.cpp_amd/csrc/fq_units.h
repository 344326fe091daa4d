// fq_units.h -- per-format "unit" decode + integer dot, shared by the GEMV kernels (device) and by a host
// unit test (tests/test_units_host.py compiles this header with g++ and checks it against the oracle).
//
// A UNIT is the work one lane does per step: one 16-byte quant group of plane0 (32 bytes for Q8_0) and the
// matching slice of the 8-bit activations. Integer arithmetic is exact (int32); each unit returns its float
// contribution with the SAME per-block expression order as the reference's scalar dot (cited per type), so a
// row's result differs from the reference only by the association of the float sum over units.
#pragma once
#include "fq_types.h"

struct fq_u4 { uint32_t x, y, z, w; };

FQ_HD int fq_dot4_ref(uint32_t a, uint32_t b, int c) {
    for (int i = 0; i < 4; ++i) c += (int)(int8_t)(a >> (8 * i)) * (int)(int8_t)(b >> (8 * i));
    return c;
}
FQ_HD float fq_h2f_ref(uint16_t h) {              // IEEE half -> float, portable
    const uint32_t s = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 31u; uint32_t m = h & 0x3FFu, b;
    if (e == 0) { if (!m) b = s; else { int k = -1; do { m <<= 1; ++k; } while (!(m & 0x400u)); b = s | (uint32_t)(112 - k) << 23 | (m & 0x3FFu) << 13; } }
    else if (e == 31) b = s | 0x7F800000u | m << 13;
    else b = s | (e + 112u) << 23 | m << 13;
    float f; __builtin_memcpy(&f, &b, 4); return f;
}
FQ_HD int fq_dot4(uint32_t a, uint32_t b, int c) {      // 4 x int8 . int8 + c
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_sdot4((int) a, (int) b, c, false);
#else
    return fq_dot4_ref(a, b, c);
#endif
}
FQ_HD float fq_h2f(uint16_t h) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (float) __builtin_bit_cast(_Float16, h);
#else
    return fq_h2f_ref(h);
#endif
}

// pointers to one weight row's planes (interleaved formats: p0 = the row, everything is addressed through fq_at)
struct fq_wrow { const uint8_t * p0, * p1, * p2, * p3; int64_t nblk; };

template <int TYPE>
FQ_HD fq_wrow fq_row(const fq_weight & w, int64_t r) {
    const fq_type_desc d = fq_desc(TYPE);
    fq_wrow o;
    o.nblk = w.nblk;
    if (fq_interleaved(TYPE)) { o.p0 = w.plane[0] + (size_t) r * w.row_stride; o.p1 = o.p2 = o.p3 = nullptr; return o; }
    o.p0 = w.plane[0] + (size_t) r * w.nblk * d.plane[0].bytes;
    o.p1 = w.plane[1] + (size_t) r * w.nblk * d.plane[1].bytes;
    o.p2 = d.nplanes > 2 ? w.plane[2] + (size_t) r * w.nblk * d.plane[2].bytes : nullptr;
    o.p3 = d.nplanes > 3 ? w.plane[3] + (size_t) r * w.nblk * d.plane[3].bytes : nullptr;
    return o;
}

// plane P's chunk of block b of an INTERLEAVED row (fq_types.h)
template <int TYPE, int P>
FQ_HD const uint8_t * fq_at(const fq_wrow & r, int64_t b) {
    constexpr int PB0 = (TYPE == FQ_Q8_0) ? 32 : 16;
    constexpr int CB  = 1024 / PB0;
    constexpr int TS  = TYPE == FQ_Q4_0 ? 18 : TYPE == FQ_Q4_1 ? 20 : TYPE == FQ_Q5_0 ? 22 : TYPE == FQ_Q5_1 ? 24 : 34;
    constexpr int PB1 = (TYPE == FQ_Q4_0 || TYPE == FQ_Q8_0) ? 2 : 4;                 // d | d,m | qh | qh | d
    constexpr int PB2 = TYPE == FQ_Q5_0 ? 2 : 4;                                      // d | d,m (Q5_0 / Q5_1 only)
    constexpr int PRE = P == 0 ? 0 : (P == 1 ? PB0 : PB0 + PB1);
    constexpr int PB  = P == 0 ? PB0 : (P == 1 ? PB1 : PB2);
    return r.p0 + fq_il_offset(CB, TS, PRE, PB, r.nblk, b);
}

// one activation column as seen by the dot (LDS on the device)
struct fq_actcol { const int8_t * qs; const float * d; const void * aux; };

// weight bytes are read ONCE per token by exactly one CU: non-temporal loads keep them from displacing what IS re-read
// (activation images, KV rows, LayerNorm inputs) -- FQ_NT_WEIGHTS=0 at compile time restores plain loads
#ifndef FQ_NT_WEIGHTS
#define FQ_NT_WEIGHTS 1
#endif
FQ_HD fq_u4    ld_u4 (const void * p) { return *(const fq_u4 *) p; }
FQ_HD fq_u4    ld_w4 (const void * p) {            // 16 bytes of a WEIGHT plane (global memory)
#if defined(__HIP_DEVICE_COMPILE__) && FQ_NT_WEIGHTS
    typedef unsigned int u32x4_nt __attribute__((ext_vector_type(4)));
    const u32x4_nt v = __builtin_nontemporal_load((const u32x4_nt *) p);
    return fq_u4{ v.x, v.y, v.z, v.w };
#else
    return *(const fq_u4 *) p;
#endif
}
FQ_HD uint32_t ld_u32(const void * p) { return *(const uint32_t *) p; }
FQ_HD uint16_t ld_u16(const void * p) { return *(const uint16_t *) p; }

FQ_HD uint32_t spread4(uint32_t bits4) { return ((bits4 & 0xFu) * 0x00204081u) & 0x01010101u; }   // bit k -> byte k bit 0

FQ_HD int dot16(const fq_u4 & a, const int8_t * x) {     // 16 int8 x 16 int8
    const fq_u4 b = ld_u4(x);
    int s = fq_dot4(a.x, b.x, 0); s = fq_dot4(a.y, b.y, s); s = fq_dot4(a.z, b.z, s); return fq_dot4(a.w, b.w, s);
}
FQ_HD fq_u4 and4(const fq_u4 & a, uint32_t m) { return { a.x & m, a.y & m, a.z & m, a.w & m }; }
FQ_HD fq_u4 shr4(const fq_u4 & a, int s)      { return { a.x >> s, a.y >> s, a.z >> s, a.w >> s }; }
FQ_HD fq_u4 or4 (const fq_u4 & a, const fq_u4 & b) { return { a.x | b.x, a.y | b.y, a.z | b.z, a.w | b.w }; }
FQ_HD fq_u4 shl4(const fq_u4 & a, int s)      { return { a.x << s, a.y << s, a.z << s, a.w << s }; }

// registers a lane holds for one unit between "load" and "dot"
struct fq_unit_regs {
    fq_u4    q;        // plane0 group
    fq_u4    q2;       // second group (Q8_0 second half, Q3_K/Q5_K/Q6_K high-bit plane)
    uint32_t s0, s1, s2;   // packed scales
    uint32_t dm;       // d (low 16) | m or dmin (high 16)
};

template <int TYPE> struct fq_unit;

// ---------------------------------------------------------------- Q4_0  (ggml.c:2591-2609)
template <> struct fq_unit<FQ_Q4_0> {
    static constexpr int ELEMS = 32;
    FQ_HDM static fq_unit_regs load(const fq_wrow & w, int u) {
        fq_unit_regs r{}; r.q = ld_w4(fq_at<FQ_Q4_0, 0>(w, u)); r.dm = ld_u16(fq_at<FQ_Q4_0, 1>(w, u)); return r;
    }
    FQ_HDM static float dot(const fq_unit_regs & r, const fq_actcol & a, int u) {
        const int8_t * x = a.qs + 32 * (size_t) u;
        int s = dot16(and4(r.q, 0x0F0F0F0Fu), x) + dot16(and4(shr4(r.q, 4), 0x0F0F0F0Fu), x + 16);
        s -= 8 * ((const int32_t *) a.aux)[u];
        return ((float) s * fq_h2f((uint16_t) r.dm)) * a.d[u];
    }
};
// ---------------------------------------------------------------- Q4_1  (ggml.c:2716-2735)
template <> struct fq_unit<FQ_Q4_1> {
    static constexpr int ELEMS = 32;
    FQ_HDM static fq_unit_regs load(const fq_wrow & w, int u) {
        fq_unit_regs r{}; r.q = ld_w4(fq_at<FQ_Q4_1, 0>(w, u)); r.dm = ld_u32(fq_at<FQ_Q4_1, 1>(w, u)); return r;
    }
    FQ_HDM static float dot(const fq_unit_regs & r, const fq_actcol & a, int u) {
        const int8_t * x = a.qs + 32 * (size_t) u;
        const int s = dot16(and4(r.q, 0x0F0F0F0Fu), x) + dot16(and4(shr4(r.q, 4), 0x0F0F0F0Fu), x + 16);
        return (fq_h2f((uint16_t) r.dm) * a.d[u]) * (float) s + fq_h2f((uint16_t)(r.dm >> 16)) * ((const float *) a.aux)[u];
    }
};
// 5th bits of a Q5 block: element j <- bit j of qh, element j+16 <- bit j+16   (ggml.c:1550-1574)
FQ_HD fq_u4 q5_hi(uint32_t qh, int base) {
    return { spread4(qh >> (base + 0)) << 4, spread4(qh >> (base + 4)) << 4, spread4(qh >> (base + 8)) << 4, spread4(qh >> (base + 12)) << 4 };
}
// ---------------------------------------------------------------- Q5_0  (ggml.c:2951-2972)
template <> struct fq_unit<FQ_Q5_0> {
    static constexpr int ELEMS = 32;
    FQ_HDM static fq_unit_regs load(const fq_wrow & w, int u) {
        fq_unit_regs r{}; r.q = ld_w4(fq_at<FQ_Q5_0, 0>(w, u)); r.s0 = ld_u32(fq_at<FQ_Q5_0, 1>(w, u)); r.dm = ld_u16(fq_at<FQ_Q5_0, 2>(w, u)); return r;
    }
    FQ_HDM static float dot(const fq_unit_regs & r, const fq_actcol & a, int u) {
        const int8_t * x = a.qs + 32 * (size_t) u;
        int s = dot16(or4(and4(r.q, 0x0F0F0F0Fu), q5_hi(r.s0, 0)), x) + dot16(or4(and4(shr4(r.q, 4), 0x0F0F0F0Fu), q5_hi(r.s0, 16)), x + 16);
        s -= 16 * ((const int32_t *) a.aux)[u];
        return (fq_h2f((uint16_t) r.dm) * a.d[u]) * (float) s;
    }
};
// ---------------------------------------------------------------- Q5_1  (ggml.c:3207-3228)
template <> struct fq_unit<FQ_Q5_1> {
    static constexpr int ELEMS = 32;
    FQ_HDM static fq_unit_regs load(const fq_wrow & w, int u) {
        fq_unit_regs r{}; r.q = ld_w4(fq_at<FQ_Q5_1, 0>(w, u)); r.s0 = ld_u32(fq_at<FQ_Q5_1, 1>(w, u)); r.dm = ld_u32(fq_at<FQ_Q5_1, 2>(w, u)); return r;
    }
    FQ_HDM static float dot(const fq_unit_regs & r, const fq_actcol & a, int u) {
        const int8_t * x = a.qs + 32 * (size_t) u;
        const int s = dot16(or4(and4(r.q, 0x0F0F0F0Fu), q5_hi(r.s0, 0)), x) + dot16(or4(and4(shr4(r.q, 4), 0x0F0F0F0Fu), q5_hi(r.s0, 16)), x + 16);
        return (fq_h2f((uint16_t) r.dm) * a.d[u]) * (float) s + fq_h2f((uint16_t)(r.dm >> 16)) * ((const float *) a.aux)[u];
    }
};
// ---------------------------------------------------------------- Q8_0  (ggml.c:3317-3329)  unit = whole block (2 x 16 B)
template <> struct fq_unit<FQ_Q8_0> {
    static constexpr int ELEMS = 32;
    FQ_HDM static fq_unit_regs load(const fq_wrow & w, int u) {
        fq_unit_regs r{}; const uint8_t * q = fq_at<FQ_Q8_0, 0>(w, u); r.q = ld_w4(q); r.q2 = ld_w4(q + 16); r.dm = ld_u16(fq_at<FQ_Q8_0, 1>(w, u)); return r;
    }
    FQ_HDM static float dot(const fq_unit_regs & r, const fq_actcol & a, int u) {
        const int8_t * x = a.qs + 32 * (size_t) u;
        const int s = dot16(r.q, x) + dot16(r.q2, x + 16);
        return (float) s * (fq_h2f((uint16_t) r.dm) * a.d[u]);
    }
};
// ---------------------------------------------------------------- Q2_K  (k_quants.c:1267-1306)
// unit u: super-block sb=u>>2, 128-half hf=(u>>1)&1, 16-byte group g=u&1; covers elements 128hf+32j+16g+l, j=0..3
template <> struct fq_unit<FQ_Q2_K> {
    static constexpr int ELEMS = 64;
    FQ_HDM static fq_unit_regs load(const fq_wrow & w, int u) {
        fq_unit_regs r{}; const size_t sb = (size_t)(u >> 2); const int hf = (u >> 1) & 1;
        r.q = ld_w4(w.p0 + 16 * (size_t) u);
        r.s0 = ld_u32(w.p1 + 16 * sb + 8 * hf); r.s1 = ld_u32(w.p1 + 16 * sb + 8 * hf + 4);
        r.dm = ld_u32(w.p2 + 4 * sb); return r;
    }
    FQ_HDM static float dot(const fq_unit_regs & r, const fq_actcol & a, int u) {
        const int sb = u >> 2, hf = (u >> 1) & 1, g = u & 1;
        const int8_t  * x  = a.qs + 256 * (size_t) sb + 128 * hf + 16 * g;
        const int16_t * bs = (const int16_t *) a.aux + 16 * (size_t) sb + 8 * hf + g;
        const uint64_t sc8 = (uint64_t) r.s0 | ((uint64_t) r.s1 << 32);       // scales[8hf .. 8hf+7]
        int isum = 0, msum = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t sc = (uint32_t)(sc8 >> (8 * (2 * j + g))) & 0xFFu;
            isum += (int)(sc & 0xFu) * dot16(and4(shr4(r.q, 2 * j), 0x03030303u), x + 32 * j);
            msum += (int)(sc >> 4) * (int) bs[2 * j];
        }
        const float dy = a.d[sb];
        return (dy * fq_h2f((uint16_t) r.dm)) * (float) isum - (dy * fq_h2f((uint16_t)(r.dm >> 16))) * (float) msum;
    }
};
// sixteen 6-bit Q3_K scales packed in 12 bytes (k_quants.c:491-496); returns scale `is` still biased by +32
FQ_HD int q3_scale(uint32_t s0, uint32_t s1, uint32_t s2, int is) {
    const uint32_t lo_src = (is & 4) ? s1 : s0;                       // bytes 0-3 / 4-7 hold the low nibbles of is%8
    const uint32_t byte = (lo_src >> (8 * (is & 3))) & 0xFFu;
    const uint32_t lo = (is < 8) ? (byte & 0xFu) : (byte >> 4);
    const uint32_t hi = ((s2 >> (8 * (is & 3))) >> (2 * (is >> 2))) & 3u;
    return (int)(lo | (hi << 4));
}
// ---------------------------------------------------------------- Q3_K  (k_quants.c:1684-1746)
template <> struct fq_unit<FQ_Q3_K> {
    static constexpr int ELEMS = 64;
    FQ_HDM static fq_unit_regs load(const fq_wrow & w, int u) {
        fq_unit_regs r{}; const size_t sb = (size_t)(u >> 2); const int g = u & 1;
        r.q  = ld_w4(w.p0 + 16 * (size_t) u);
        r.q2 = ld_w4(w.p1 + 32 * sb + 16 * g);                          // hmask bytes of this 16-byte group
        r.s0 = ld_u32(w.p2 + 12 * sb); r.s1 = ld_u32(w.p2 + 12 * sb + 4); r.s2 = ld_u32(w.p2 + 12 * sb + 8);
        r.dm = ld_u16(w.p3 + 2 * sb); return r;
    }
    FQ_HDM static float dot(const fq_unit_regs & r, const fq_actcol & a, int u) {
        const int sb = u >> 2, hf = (u >> 1) & 1, g = u & 1;
        const int8_t  * x  = a.qs + 256 * (size_t) sb + 128 * hf + 16 * g;
        const int16_t * bs = (const int16_t *) a.aux + 16 * (size_t) sb + 8 * hf + g;
        int isum = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const fq_u4 lo = and4(shr4(r.q, 2 * j), 0x03030303u);
            const fq_u4 hb = shl4(and4(shr4(r.q2, 4 * hf + j), 0x01010101u), 2);     // high bit set -> +4, then -4 for all
            const int d16 = dot16(or4(lo, hb), x + 32 * j) - 4 * (int) bs[2 * j];
            isum += (q3_scale(r.s0, r.s1, r.s2, 8 * hf + 2 * j + g) - 32) * d16;
        }
        return (fq_h2f((uint16_t) r.dm) * a.d[sb]) * (float) isum;
    }
};
// 6-bit (scale, min) pair j of a Q4_K / Q5_K block from its 12 scale bytes (k_quants.c:264-272)
FQ_HD void k4_scale_min(uint32_t s0, uint32_t s1, uint32_t s2, int j, int & sc, int & mn) {
    // bytes 0-3 = s0, 4-7 = s1, 8-11 = s2; no local arrays (runtime-indexed arrays would live in scratch)
    const int sh = 8 * (j & 3);
    const uint32_t b0 = (s0 >> sh) & 0xFFu, b1 = (s1 >> sh) & 0xFFu, b2 = (s2 >> sh) & 0xFFu;
    if (j < 4) { sc = (int)(b0 & 63u); mn = (int)(b1 & 63u); }
    else       { sc = (int)((b2 & 0xFu) | ((b0 >> 6) << 4)); mn = (int)((b2 >> 4) | ((b1 >> 6) << 4)); }
}
// ---------------------------------------------------------------- Q4_K  (k_quants.c:1999-2055)
// unit u: sb=u>>3, 64-chunk c=(u>>1)&3, group g=u&1; low nibbles -> elements 64c+16g+l (sub-block 2c), high -> +32 (2c+1)
template <> struct fq_unit<FQ_Q4_K> {
    static constexpr int ELEMS = 32;
    FQ_HDM static fq_unit_regs load(const fq_wrow & w, int u) {
        fq_unit_regs r{}; const size_t sb = (size_t)(u >> 3);
        r.q = ld_w4(w.p0 + 16 * (size_t) u);
        r.s0 = ld_u32(w.p1 + 12 * sb); r.s1 = ld_u32(w.p1 + 12 * sb + 4); r.s2 = ld_u32(w.p1 + 12 * sb + 8);
        r.dm = ld_u32(w.p2 + 4 * sb); return r;
    }
    FQ_HDM static float dot(const fq_unit_regs & r, const fq_actcol & a, int u) {
        const int sb = u >> 3, c = (u >> 1) & 3, g = u & 1;
        const int8_t  * x  = a.qs + 256 * (size_t) sb + 64 * c + 16 * g;
        const int16_t * bs = (const int16_t *) a.aux + 16 * (size_t) sb + 4 * c + g;
        int sc0, mn0, sc1, mn1;
        k4_scale_min(r.s0, r.s1, r.s2, 2 * c, sc0, mn0); k4_scale_min(r.s0, r.s1, r.s2, 2 * c + 1, sc1, mn1);
        const int isum = sc0 * dot16(and4(r.q, 0x0F0F0F0Fu), x) + sc1 * dot16(and4(shr4(r.q, 4), 0x0F0F0F0Fu), x + 32);
        const int msum = mn0 * (int) bs[0] + mn1 * (int) bs[2];
        const float dy = a.d[sb];
        return (fq_h2f((uint16_t) r.dm) * dy) * (float) isum - (fq_h2f((uint16_t)(r.dm >> 16)) * dy) * (float) msum;
    }
};
// ---------------------------------------------------------------- Q5_K  (k_quants.c:2340-2400)
template <> struct fq_unit<FQ_Q5_K> {
    static constexpr int ELEMS = 32;
    FQ_HDM static fq_unit_regs load(const fq_wrow & w, int u) {
        fq_unit_regs r{}; const size_t sb = (size_t)(u >> 3); const int g = u & 1;
        r.q  = ld_w4(w.p0 + 16 * (size_t) u);
        r.q2 = ld_w4(w.p1 + 32 * sb + 16 * g);                          // qh bytes of this group
        r.s0 = ld_u32(w.p2 + 12 * sb); r.s1 = ld_u32(w.p2 + 12 * sb + 4); r.s2 = ld_u32(w.p2 + 12 * sb + 8);
        r.dm = ld_u32(w.p3 + 4 * sb); return r;
    }
    FQ_HDM static float dot(const fq_unit_regs & r, const fq_actcol & a, int u) {
        const int sb = u >> 3, c = (u >> 1) & 3, g = u & 1;
        const int8_t  * x  = a.qs + 256 * (size_t) sb + 64 * c + 16 * g;
        const int16_t * bs = (const int16_t *) a.aux + 16 * (size_t) sb + 4 * c + g;
        int sc0, mn0, sc1, mn1;
        k4_scale_min(r.s0, r.s1, r.s2, 2 * c, sc0, mn0); k4_scale_min(r.s0, r.s1, r.s2, 2 * c + 1, sc1, mn1);
        const fq_u4 lo = or4(and4(r.q, 0x0F0F0F0Fu),          shl4(and4(shr4(r.q2, 2 * c),     0x01010101u), 4));
        const fq_u4 hi = or4(and4(shr4(r.q, 4), 0x0F0F0F0Fu), shl4(and4(shr4(r.q2, 2 * c + 1), 0x01010101u), 4));
        const int isum = sc0 * dot16(lo, x) + sc1 * dot16(hi, x + 32);
        const int msum = mn0 * (int) bs[0] + mn1 * (int) bs[2];
        const float dy = a.d[sb];
        return (fq_h2f((uint16_t) r.dm) * dy) * (float) isum - (fq_h2f((uint16_t)(r.dm >> 16)) * dy) * (float) msum;
    }
};
// ---------------------------------------------------------------- Q6_K  (k_quants.c:2748-2789)
// unit u: sb=u>>3, half h=(u>>2)&1, t01=(u>>1)&1, g=u&1; low nibbles -> quarter t01, high nibbles -> quarter t01+2
template <> struct fq_unit<FQ_Q6_K> {
    static constexpr int ELEMS = 32;
    FQ_HDM static fq_unit_regs load(const fq_wrow & w, int u) {
        fq_unit_regs r{}; const size_t sb = (size_t)(u >> 3); const int h = (u >> 2) & 1, g = u & 1;
        r.q  = ld_w4(w.p0 + 16 * (size_t) u);
        r.q2 = ld_w4(w.p1 + 64 * sb + 32 * h + 16 * g);
        r.s0 = ld_u32(w.p2 + 16 * sb + 8 * h); r.s1 = ld_u32(w.p2 + 16 * sb + 8 * h + 4);   // int8 scales[8h .. 8h+7]
        r.dm = ld_u16(w.p3 + 2 * sb); return r;
    }
    FQ_HDM static float dot(const fq_unit_regs & r, const fq_actcol & a, int u) {
        const int sb = u >> 3, h = (u >> 2) & 1, t = (u >> 1) & 1, g = u & 1;
        const int8_t  * x  = a.qs + 256 * (size_t) sb + 128 * h + 32 * t + 16 * g;
        const int16_t * bs = (const int16_t *) a.aux + 16 * (size_t) sb + 8 * h + 2 * t + g;
        const uint64_t sc8 = (uint64_t) r.s0 | ((uint64_t) r.s1 << 32);
        const int sc_lo = (int)(int8_t)(sc8 >> (8 * (2 * t + g)));
        const int sc_hi = (int)(int8_t)(sc8 >> (8 * (2 * (t + 2) + g)));
        const fq_u4 lo = or4(and4(r.q, 0x0F0F0F0Fu),          shl4(and4(shr4(r.q2, 2 * t),     0x03030303u), 4));
        const fq_u4 hi = or4(and4(shr4(r.q, 4), 0x0F0F0F0Fu), shl4(and4(shr4(r.q2, 2 * t + 4), 0x03030303u), 4));
        const int isum = sc_lo * (dot16(lo, x) - 32 * (int) bs[0]) + sc_hi * (dot16(hi, x + 64) - 32 * (int) bs[4]);
        return (fq_h2f((uint16_t) r.dm) * a.d[sb]) * (float) isum;
    }
};

// ---- unit (64 c + lane) of a row for the streaming loops: column c is wave-uniform, so for the interleaved formats the
// column's base and plane offsets are scalar work and only the lane's own offset is per-lane; units beyond the row's end
// are clamped to its last unit (the callers mask their contribution)
template <int TYPE>
FQ_HD fq_unit_regs fq_unit_load_col(const fq_wrow & w, int c, int lane, int units) {
    if constexpr (TYPE == FQ_Q4_0 || TYPE == FQ_Q4_1 || TYPE == FQ_Q5_0 || TYPE == FQ_Q5_1) {
        constexpr int TS = TYPE == FQ_Q4_0 ? 18 : TYPE == FQ_Q4_1 ? 20 : TYPE == FQ_Q5_0 ? 22 : 24;
        constexpr int PB1 = TYPE == FQ_Q4_0 ? 2 : 4;
        const int last = (units - 1) >> 6;
        const int cc = c < last ? c : last;
        const int rem = units - 64 * cc;
        const int nbc = rem < 64 ? rem : 64;
        const int jl = lane < nbc ? lane : nbc - 1;
        const uint8_t * base = w.p0 + (size_t) cc * (64 * TS);
        fq_unit_regs r{};
        r.q = ld_w4(base + 16 * jl);
        const uint8_t * p1 = base + 16 * nbc + PB1 * jl;
        if constexpr (TYPE == FQ_Q4_0)      r.dm = ld_u16(p1);
        else if constexpr (TYPE == FQ_Q4_1) r.dm = ld_u32(p1);
        else {
            r.s0 = ld_u32(p1);
            const uint8_t * p2 = base + (16 + PB1) * nbc + (TYPE == FQ_Q5_0 ? 2 : 4) * jl;
            if constexpr (TYPE == FQ_Q5_0) r.dm = ld_u16(p2); else r.dm = ld_u32(p2);
        }
        return r;
    } else {
        const int u = 64 * c + lane;
        return fq_unit<TYPE>::load(w, u < units ? u : units - 1);
    }
}

