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
// the FIRST dot of a chain: c + a . b with c a value that must survive (or the constant 0). With clamp = false the builtin always selects the two-operand form
// v_dot4c_i32_i8 (the accumulator is the destination), which costs a v_mov_b32 per chain to set it up (round 6: 6 of the 69 vector instructions of a Q2_K unit are such
// moves); with clamp = true it selects the three-operand VOP3P form v_dot4_i32_i8 v, a, b, src2 (inline 0 or a live register; the clamp saturates at the int32 range, which
// no sum here comes near: the same integers, scripts/microbench/mb_dot_clamp.hip). FQ_DOT_VOP3P=1 selects that form -- and is OFF: measured +0.6 % (Q2_K) .. +3 % (Q3_K, with
// the in-place decode) on the Falcon-40B decode, but k_gemv_kq_ref<Q3_K> then reads a STALE register in `sc * dot` (v_mul_lo_u32 three instructions behind the VOP3P dot, the
// distance the compiler's hazard table asks for; garbage on the box, tests/test_gpu_kqref.py) while every ring kernel passes. An instruction form whose latency the compiler
// misjudges somewhere is not one to ship for 1 %: NOTEBOOK 10.15. (An inline-assembly form is wrong for the same reason, without any s_nop at all.)
#ifndef FQ_DOT_VOP3P
#define FQ_DOT_VOP3P 0
#endif
FQ_HD int fq_dot4z(uint32_t a, uint32_t b) {             // 4 x int8 . int8
#if defined(__HIP_DEVICE_COMPILE__) && FQ_DOT_VOP3P
    return __builtin_amdgcn_sdot4((int) a, (int) b, 0, true);
#else
    return fq_dot4(a, b, 0);
#endif
}
FQ_HD int fq_dot4s(uint32_t a, uint32_t b, int c) {      // 4 x int8 . int8 + c, c stays alive
#if defined(__HIP_DEVICE_COMPILE__) && FQ_DOT_VOP3P
    return __builtin_amdgcn_sdot4((int) a, (int) b, c, true);
#else
    return fq_dot4(a, b, c);
#endif
}
FQ_HD float fq_h2f(uint16_t h) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (float) __builtin_bit_cast(_Float16, h);
#else
    return fq_h2f_ref(h);
#endif
}

// one weight row (fq_types.h: planes interleaved per column)
struct fq_wrow { const uint8_t * p0; int64_t nblk; };

template <int TYPE>
FQ_HD fq_wrow fq_row(const fq_weight & w, int64_t r) { return { w.plane[0] + (size_t) r * w.row_stride, w.nblk }; }

// layout constants of a format: unit = 16 bytes of plane 0 (32 for Q8_0)
template <int TYPE> struct fq_lay {
    static constexpr int UB  = (TYPE == FQ_Q8_0) ? 32 : 16;
    static constexpr int PB0 = fq_desc(TYPE).plane[0].bytes;
    static constexpr int TS  = fq_desc(TYPE).tsize;
    static constexpr int CB  = 1024 / PB0;                 // blocks per column
    static constexpr int UPS = PB0 / UB;                   // units per block
    static constexpr int UPC = CB * UPS;                   // units per column: 64 (Q8_0: 32)
};
// one column of a row: nbc blocks (CB, fewer in the row's last column), planes packed back to back
struct fq_col { const uint8_t * base; int nbc; };
template <int TYPE>
FQ_HD fq_col fq_col_at(const fq_wrow & r, int64_t c) {
    const int64_t rem = r.nblk - c * fq_lay<TYPE>::CB;
    return { r.p0 + (size_t) c * (size_t)(fq_lay<TYPE>::CB * fq_lay<TYPE>::TS), (int)(rem < fq_lay<TYPE>::CB ? rem : fq_lay<TYPE>::CB) };
}
// plane P's chunk of block bj of the column
template <int TYPE, int P>
FQ_HD const uint8_t * fq_cp(const fq_col & k, int bj) {
    constexpr int PRE = fq_plane_pre(fq_desc(TYPE), P), PB = fq_desc(TYPE).plane[P].bytes;
    return k.base + k.nbc * PRE + bj * PB;
}
// plane P's chunk of block b of the row
template <int TYPE, int P>
FQ_HD const uint8_t * fq_at(const fq_wrow & r, int64_t b) {
    const int64_t c = b / fq_lay<TYPE>::CB;
    return fq_cp<TYPE, P>(fq_col_at<TYPE>(r, c), (int)(b - c * fq_lay<TYPE>::CB));
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
// NT = false: the same 16 bytes out of LDS (the ring forms stage raw weight rows there: plain loads)
template <bool NT> FQ_HD fq_u4 ld_q4(const void * p) { if constexpr (NT) return ld_w4(p); else return ld_u4(p); }
FQ_HD uint32_t ld_u32(const void * p) { return *(const uint32_t *) p; }
FQ_HD uint16_t ld_u16(const void * p) { return *(const uint16_t *) p; }

FQ_HD uint32_t spread4(uint32_t bits4) { return ((bits4 & 0xFu) * 0x00204081u) & 0x01010101u; }   // bit k -> byte k bit 0

FQ_HD int dot16r(const fq_u4 & a, const fq_u4 & b) {     // 16 int8 x 16 int8, both in registers
    int s = fq_dot4z(a.x, b.x); s = fq_dot4(a.y, b.y, s); s = fq_dot4(a.z, b.z, s); return fq_dot4(a.w, b.w, s);
}
FQ_HD int dot16rs(const fq_u4 & a, const fq_u4 & b, int c) {     // c + 16 int8 x 16 int8 (c: a register that stays alive, e.g. a pre-negated block-sum term)
    int s = fq_dot4s(a.x, b.x, c); s = fq_dot4(a.y, b.y, s); s = fq_dot4(a.z, b.z, s); return fq_dot4(a.w, b.w, s);
}
FQ_HD int dot16(const fq_u4 & a, const int8_t * x) { return dot16r(a, ld_u4(x)); }     // 16 int8 x 16 int8
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

// the activation slice of one 32-element unit of the legacy formats, in registers: the 32 int8, the block's d, and its aux word
// (Q8_0: i32 sum of the block's quants; Q8_1: f32 s = d * sum). dot_x() below is THE arithmetic of each legacy format's unit; dot()
// loads the slice and calls it, the persistent engine loads all slices of a pass first (one LDS round trip) and calls it directly.
struct fq_act32 { fq_u4 x0, x1; float d; uint32_t aux; };
FQ_HD fq_act32 fq_act32_load(const fq_actcol & a, int u) {
    const int8_t * x = a.qs + 32 * (size_t) u;
    return { ld_u4(x), ld_u4(x + 16), a.d[u], ((const uint32_t *) a.aux)[u] };
}
FQ_HD float fq_bits2f(uint32_t b) { float f; __builtin_memcpy(&f, &b, 4); return f; }

// ---------------------------------------------------------------- Q4_0  (ggml.c:2591-2609)
template <> struct fq_unit<FQ_Q4_0> {
    static constexpr int ELEMS = 32;
    template <bool NT = true> FQ_HDM static fq_unit_regs load_at(const fq_col & k, int ju) {
        fq_unit_regs r{}; r.q = ld_q4<NT>(fq_cp<FQ_Q4_0, 0>(k, ju)); r.dm = ld_u16(fq_cp<FQ_Q4_0, 1>(k, ju)); return r;
    }
    FQ_HDM static float dot_x(const fq_unit_regs & r, const fq_act32 & y) {
        int s = dot16r(and4(r.q, 0x0F0F0F0Fu), y.x0) + dot16r(and4(shr4(r.q, 4), 0x0F0F0F0Fu), y.x1);
        s -= 8 * (int32_t) y.aux;
        return ((float) s * fq_h2f((uint16_t) r.dm)) * y.d;
    }
    FQ_HDM static float dot(const fq_unit_regs & r, const fq_actcol & a, int u) { return dot_x(r, fq_act32_load(a, u)); }
};
// ---------------------------------------------------------------- Q4_1  (ggml.c:2716-2735)
template <> struct fq_unit<FQ_Q4_1> {
    static constexpr int ELEMS = 32;
    template <bool NT = true> FQ_HDM static fq_unit_regs load_at(const fq_col & k, int ju) {
        fq_unit_regs r{}; r.q = ld_q4<NT>(fq_cp<FQ_Q4_1, 0>(k, ju)); r.dm = ld_u32(fq_cp<FQ_Q4_1, 1>(k, ju)); return r;
    }
    FQ_HDM static float dot_x(const fq_unit_regs & r, const fq_act32 & y) {
        const int s = dot16r(and4(r.q, 0x0F0F0F0Fu), y.x0) + dot16r(and4(shr4(r.q, 4), 0x0F0F0F0Fu), y.x1);
        return (fq_h2f((uint16_t) r.dm) * y.d) * (float) s + fq_h2f((uint16_t)(r.dm >> 16)) * fq_bits2f(y.aux);
    }
    FQ_HDM static float dot(const fq_unit_regs & r, const fq_actcol & a, int u) { return dot_x(r, fq_act32_load(a, u)); }
};
// 5th bits of a Q5 block: element j <- bit j of qh, element j+16 <- bit j+16   (ggml.c:1550-1574)
FQ_HD fq_u4 q5_hi(uint32_t qh, int base) {
    return { spread4(qh >> (base + 0)) << 4, spread4(qh >> (base + 4)) << 4, spread4(qh >> (base + 8)) << 4, spread4(qh >> (base + 12)) << 4 };
}
// ---------------------------------------------------------------- Q5_0  (ggml.c:2951-2972)
template <> struct fq_unit<FQ_Q5_0> {
    static constexpr int ELEMS = 32;
    template <bool NT = true> FQ_HDM static fq_unit_regs load_at(const fq_col & k, int ju) {
        fq_unit_regs r{}; r.q = ld_q4<NT>(fq_cp<FQ_Q5_0, 0>(k, ju)); r.s0 = ld_u32(fq_cp<FQ_Q5_0, 1>(k, ju)); r.dm = ld_u16(fq_cp<FQ_Q5_0, 2>(k, ju)); return r;
    }
    FQ_HDM static float dot_x(const fq_unit_regs & r, const fq_act32 & y) {
        int s = dot16r(or4(and4(r.q, 0x0F0F0F0Fu), q5_hi(r.s0, 0)), y.x0) + dot16r(or4(and4(shr4(r.q, 4), 0x0F0F0F0Fu), q5_hi(r.s0, 16)), y.x1);
        s -= 16 * (int32_t) y.aux;
        return (fq_h2f((uint16_t) r.dm) * y.d) * (float) s;
    }
    FQ_HDM static float dot(const fq_unit_regs & r, const fq_actcol & a, int u) { return dot_x(r, fq_act32_load(a, u)); }
};
// ---------------------------------------------------------------- Q5_1  (ggml.c:3207-3228)
template <> struct fq_unit<FQ_Q5_1> {
    static constexpr int ELEMS = 32;
    template <bool NT = true> FQ_HDM static fq_unit_regs load_at(const fq_col & k, int ju) {
        fq_unit_regs r{}; r.q = ld_q4<NT>(fq_cp<FQ_Q5_1, 0>(k, ju)); r.s0 = ld_u32(fq_cp<FQ_Q5_1, 1>(k, ju)); r.dm = ld_u32(fq_cp<FQ_Q5_1, 2>(k, ju)); return r;
    }
    FQ_HDM static float dot_x(const fq_unit_regs & r, const fq_act32 & y) {
        const int s = dot16r(or4(and4(r.q, 0x0F0F0F0Fu), q5_hi(r.s0, 0)), y.x0) + dot16r(or4(and4(shr4(r.q, 4), 0x0F0F0F0Fu), q5_hi(r.s0, 16)), y.x1);
        return (fq_h2f((uint16_t) r.dm) * y.d) * (float) s + fq_h2f((uint16_t)(r.dm >> 16)) * fq_bits2f(y.aux);
    }
    FQ_HDM static float dot(const fq_unit_regs & r, const fq_actcol & a, int u) { return dot_x(r, fq_act32_load(a, u)); }
};
// ---------------------------------------------------------------- Q8_0  (ggml.c:3317-3329)  unit = whole block (2 x 16 B)
template <> struct fq_unit<FQ_Q8_0> {
    static constexpr int ELEMS = 32;
    template <bool NT = true> FQ_HDM static fq_unit_regs load_at(const fq_col & k, int ju) {
        fq_unit_regs r{}; const uint8_t * q = fq_cp<FQ_Q8_0, 0>(k, ju); r.q = ld_q4<NT>(q); r.q2 = ld_q4<NT>(q + 16); r.dm = ld_u16(fq_cp<FQ_Q8_0, 1>(k, ju)); return r;
    }
    FQ_HDM static float dot_x(const fq_unit_regs & r, const fq_act32 & y) {
        const int s = dot16r(r.q, y.x0) + dot16r(r.q2, y.x1);
        return (float) s * (fq_h2f((uint16_t) r.dm) * y.d);
    }
    FQ_HDM static float dot(const fq_unit_regs & r, const fq_actcol & a, int u) { return dot_x(r, fq_act32_load(a, u)); }
};
// ---------------------------------------------------------------- Q2_K  (k_quants.c:1267-1306)
// unit u: super-block sb=u>>2, 128-half hf=(u>>1)&1, 16-byte group g=u&1; covers elements 128hf+32j+16g+l, j=0..3
template <> struct fq_unit<FQ_Q2_K> {
    static constexpr int ELEMS = 64;
    template <bool NT = true> FQ_HDM static fq_unit_regs load_at(const fq_col & k, int ju) {       // ju: unit inside the column (4 per super-block)
        fq_unit_regs r{}; const int sb = ju >> 2, hf = (ju >> 1) & 1;
        r.q = ld_q4<NT>(fq_cp<FQ_Q2_K, 0>(k, sb) + 16 * (ju & 3));
        const uint8_t * sc = fq_cp<FQ_Q2_K, 1>(k, sb) + 8 * hf;
        r.s0 = ld_u32(sc); r.s1 = ld_u32(sc + 4);
        r.dm = ld_u32(fq_cp<FQ_Q2_K, 2>(k, sb)); return r;
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
    template <bool NT = true> FQ_HDM static fq_unit_regs load_at(const fq_col & k, int ju) {
        fq_unit_regs r{}; const int sb = ju >> 2, g = ju & 1;
        r.q  = ld_q4<NT>(fq_cp<FQ_Q3_K, 0>(k, sb) + 16 * (ju & 3));
        r.q2 = ld_q4<NT>(fq_cp<FQ_Q3_K, 1>(k, sb) + 16 * g);                // hmask bytes of this 16-byte group
        const uint8_t * sc = fq_cp<FQ_Q3_K, 2>(k, sb);
        r.s0 = ld_u32(sc); r.s1 = ld_u32(sc + 4); r.s2 = ld_u32(sc + 8);
        r.dm = ld_u16(fq_cp<FQ_Q3_K, 3>(k, sb)); return r;
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
    template <bool NT = true> FQ_HDM static fq_unit_regs load_at(const fq_col & k, int ju) {       // 8 units per super-block
        fq_unit_regs r{}; const int sb = ju >> 3;
        r.q = ld_q4<NT>(fq_cp<FQ_Q4_K, 0>(k, sb) + 16 * (ju & 7));
        const uint8_t * sc = fq_cp<FQ_Q4_K, 1>(k, sb);
        r.s0 = ld_u32(sc); r.s1 = ld_u32(sc + 4); r.s2 = ld_u32(sc + 8);
        r.dm = ld_u32(fq_cp<FQ_Q4_K, 2>(k, sb)); return r;
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
    template <bool NT = true> FQ_HDM static fq_unit_regs load_at(const fq_col & k, int ju) {
        fq_unit_regs r{}; const int sb = ju >> 3, g = ju & 1;
        r.q  = ld_q4<NT>(fq_cp<FQ_Q5_K, 0>(k, sb) + 16 * (ju & 7));
        r.q2 = ld_q4<NT>(fq_cp<FQ_Q5_K, 1>(k, sb) + 16 * g);                // qh bytes of this group
        const uint8_t * sc = fq_cp<FQ_Q5_K, 2>(k, sb);
        r.s0 = ld_u32(sc); r.s1 = ld_u32(sc + 4); r.s2 = ld_u32(sc + 8);
        r.dm = ld_u32(fq_cp<FQ_Q5_K, 3>(k, sb)); return r;
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
    template <bool NT = true> FQ_HDM static fq_unit_regs load_at(const fq_col & k, int ju) {
        fq_unit_regs r{}; const int sb = ju >> 3, h = (ju >> 2) & 1, g = ju & 1;
        r.q  = ld_q4<NT>(fq_cp<FQ_Q6_K, 0>(k, sb) + 16 * (ju & 7));
        r.q2 = ld_q4<NT>(fq_cp<FQ_Q6_K, 1>(k, sb) + 32 * h + 16 * g);
        const uint8_t * sc = fq_cp<FQ_Q6_K, 2>(k, sb) + 8 * h;
        r.s0 = ld_u32(sc); r.s1 = ld_u32(sc + 4);                       // int8 scales[8h .. 8h+7]
        r.dm = ld_u16(fq_cp<FQ_Q6_K, 3>(k, sb)); return r;
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

// ---- generic entry points over load_at
// unit u of a row
template <int TYPE>
FQ_HD fq_unit_regs fq_unit_load(const fq_wrow & w, int u) {
    constexpr int UPC = fq_lay<TYPE>::UPC;
    const int c = u / UPC;
    return fq_unit<TYPE>::load_at(fq_col_at<TYPE>(w, c), u - c * UPC);
}
// unit (64 c + lane) of a row for the streaming loops: c is wave-uniform, so the column's base and the plane offsets are
// scalar work and only the lane's own offset is per-lane; units beyond the row's end are clamped to its last unit (the
// callers mask their contribution)
template <int TYPE>
FQ_HD fq_unit_regs fq_unit_load_col(const fq_wrow & w, int c, int lane, int units) {
    if constexpr (fq_lay<TYPE>::UPC == 64) {
        const int last = (units - 1) >> 6;
        const fq_col k = fq_col_at<TYPE>(w, c < last ? c : last);
        const int nu = k.nbc * fq_lay<TYPE>::UPS;
        return fq_unit<TYPE>::load_at(k, lane < nu ? lane : nu - 1);
    } else {                                                            // Q8_0: 32 units per column
        const int u = 64 * c + lane;
        return fq_unit_load<TYPE>(w, u < units ? u : units - 1);
    }
}

