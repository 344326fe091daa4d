// fq_kdot.h -- the k-quant unit dots of fq_units.h restated for the ring consumers (kernels_ringk.hip), which see the SAME lane in the same
// unit slot of every row: everything that only depends on the lane's slot ju (which 16-byte group of which super-block of a column, shift
// amounts and masks of its packed scales, where its activation slice sits) is computed once per wave (lane_t), the activation slice of a
// (pass, lane) -- 32 or 64 int8, the Q8_K block sums it needs, the super-block's d -- is loaded once and kept in registers (act_t), and the
// packed 6-bit / 4-bit scale decode works on both sub-blocks of a unit at a time. Integer arithmetic is exact, so any order gives the
// reference's integers; the float expression of a unit is fq_unit<TYPE>::dot's, operand for operand -- the same f32 term, bit for bit
// (tests/test_units_host.py compiles this header on the host and compares the two on random blocks).
//   ggml_vec_dot_q2_K_q8_K k_quants.c:1267-1306   q3_K :1684-1746   q4_K :1999-2055   q5_K :2340-2400   q6_K :2748-2789
//   get_scale_min_k4 k_quants.c:264-272, the Q3_K scale unpack k_quants.c:491-496
// A column here is always FULL (CB super-blocks): rows whose length is not a whole number of columns take the generic path.
#pragma once
#include "fq_units.h"

FQ_HD int fq_dot2(uint32_t a, uint32_t b, int c) {         // 2 x int16 . int16 + c
#if defined(__HIP_DEVICE_COMPILE__)
    typedef short fq_s2 __attribute__((ext_vector_type(2)));
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(fq_s2, a), __builtin_bit_cast(fq_s2, b), c, false);
#else
    return c + (int)(int16_t)(a & 0xFFFFu) * (int)(int16_t)(b & 0xFFFFu) + (int)(int16_t)(a >> 16) * (int)(int16_t)(b >> 16);
#endif
}
FQ_HD int fq_dot2z(uint32_t a, uint32_t b) {                // the first dot of a chain (fq_units.h: fq_dot4z)
#if defined(__HIP_DEVICE_COMPILE__) && FQ_DOT_VOP3P
    typedef short fq_s2z __attribute__((ext_vector_type(2)));
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(fq_s2z, a), __builtin_bit_cast(fq_s2z, b), 0, true);
#else
    return fq_dot2(a, b, 0);
#endif
}
FQ_HD uint32_t fq_pack16(int lo, int hi) {                  // the low 16 bits of two integers side by side (v_perm_b32)
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_perm((uint32_t) hi, (uint32_t) lo, 0x05040100u);
#else
    return ((uint32_t) lo & 0xFFFFu) | ((uint32_t) hi << 16);
#endif
}
FQ_HD uint32_t fq_pk_sub16(uint32_t a, uint32_t b) {         // two int16 lanes: a - b per lane (v_pk_sub_i16)
#if defined(__HIP_DEVICE_COMPILE__)
    typedef short fq_s2v __attribute__((ext_vector_type(2)));
    const fq_s2v r = __builtin_bit_cast(fq_s2v, a) - __builtin_bit_cast(fq_s2v, b);
    return __builtin_bit_cast(uint32_t, r);
#else
    const uint32_t lo = (uint32_t)((int)(int16_t)(a & 0xFFFFu) - (int)(int16_t)(b & 0xFFFFu)) & 0xFFFFu;
    const uint32_t hi = (uint32_t)((int)(int16_t)(a >> 16) - (int)(int16_t)(b >> 16)) & 0xFFFFu;
    return lo | (hi << 16);
#endif
}
FQ_HD int fq_sext8(uint32_t v, int sh) { return (int)(int8_t)(v >> sh); }
FQ_HD int fq_mul24(int a, int b) {                            // a * b for operands inside 24 signed bits (v_mul_i32_i24 / v_mad_i32_i24: full rate; a 32-bit multiply is quarter rate)
#if defined(__HIP_DEVICE_COMPILE__)
    return __mul24(a, b);
#else
    return a * b;
#endif
}

template <int TYPE> struct fq_kdot { static constexpr bool ok = false; };

// ---------------------------------------------------------------- Q4_K / Q5_K: unit ju of a column: super-block ju >> 3, 64-chunk c = (ju >> 1) & 3, group g = ju & 1
template <int TYPE> struct fq_kdot45 {
    static constexpr bool ok = true;
    static constexpr bool Q5 = TYPE == FQ_Q5_K;
    static constexpr int CB = fq_lay<TYPE>::CB;                          // 8 super-blocks per column
    static constexpr int O_QH = CB * 128, O_SC = CB * (Q5 ? 160 : 128), O_DM = O_SC + CB * 12;      // plane offsets inside a full column
    struct lane_t { uint32_t sh, m1, m2, lo, hsh; int qoff, hoff, soff, doff, aoff, boff, sbl; };
    struct act_t { fq_u4 x0, x1; uint32_t bs; float dy; };
    struct w_t { fq_u4 q, qh; uint32_t s0, s1, s2, dm; };
    FQ_HDM static lane_t lane_init(int ju) {
        lane_t L{};
        const int sbl = ju >> 3, c = (ju >> 1) & 3, g = ju & 1;
        L.sbl = sbl; L.lo = c < 2; L.sh = 16u * (uint32_t)(c & 1); L.m1 = c < 2 ? 0x3F3Fu : 0x0F0Fu; L.m2 = c < 2 ? 0u : 0x3030u; L.hsh = 2u * (uint32_t) c;
        L.qoff = 16 * ju; L.hoff = O_QH + 32 * sbl + 16 * g; L.soff = O_SC + 12 * sbl; L.doff = O_DM + 4 * sbl;
        L.aoff = 256 * sbl + 64 * c + 16 * g; L.boff = 16 * sbl + 4 * c + g;
        return L;
    }
    FQ_HDM static act_t act_load(const fq_actcol & a, int p, const lane_t & L) {          // the slice of unit 64 p + ju
        act_t y;
        const int8_t * x = a.qs + 256 * CB * (size_t) p + L.aoff;
        const int16_t * bs = (const int16_t *) a.aux + 16 * CB * (size_t) p + L.boff;
        y.x0 = ld_u4(x); y.x1 = ld_u4(x + 32);
        y.bs = fq_pack16(bs[0], bs[2]);
        y.dy = a.d[CB * p + L.sbl];
        return y;
    }
    FQ_HDM static w_t w_load(const uint8_t * col, const lane_t & L) {
        w_t w{};
        w.q = ld_u4(col + L.qoff);
        if (Q5) w.qh = ld_u4(col + L.hoff);
        const uint8_t * sc = col + L.soff;
        w.s0 = ld_u32(sc); w.s1 = ld_u32(sc + 4); w.s2 = ld_u32(sc + 8);
        w.dm = ld_u32(col + L.doff);
        return w;
    }
    FQ_HDM static w_t from_regs(const fq_unit_regs & r) { return w_t{ r.q, r.q2, r.s0, r.s1, r.s2, r.dm }; }      // what fq_unit<TYPE>::load_at fetched for the same unit
    FQ_HDM static float dot(const w_t & w, const act_t & y, const lane_t & L) {
        // the (scale, min) pairs 2c and 2c + 1 as byte pairs: bytes (2c) & 3 and + 1 of the three scale words
        const uint32_t w0 = (w.s0 >> L.sh) & 0xFFFFu, w1 = (w.s1 >> L.sh) & 0xFFFFu, w2 = (w.s2 >> L.sh) & 0xFFFFu;
        const uint32_t X = L.lo ? w0 : w2, Y = L.lo ? w1 : (w2 >> 4);
        const uint32_t psc = (X & L.m1) | ((w0 >> 2) & L.m2);             // j < 4: b0 & 63          j >= 4: (b2 & 15) | ((b0 >> 6) << 4)
        const uint32_t pmn = (Y & L.m1) | ((w1 >> 2) & L.m2);             // j < 4: b1 & 63          j >= 4: (b2 >> 4) | ((b1 >> 6) << 4)
        fq_u4 lo = and4(w.q, 0x0F0F0F0Fu), hi = and4(shr4(w.q, 4), 0x0F0F0F0Fu);
        if (Q5) {
            const fq_u4 t = shr4(w.qh, (int) L.hsh);                      // bit 0 of a byte: element of sub-block 2c, bit 1: of 2c + 1
            lo = or4(lo, shl4(and4(t, 0x01010101u), 4));
            hi = or4(hi, shl4(and4(t, 0x02020202u), 3));
        }
        const int isum = fq_mul24((int)(psc & 0xFFu), dot16r(lo, y.x0)) + fq_mul24((int)(psc >> 8), dot16r(hi, y.x1));      // (|dot| <= 16 * 31 * 128)
        const int msum = fq_dot2z((pmn | (pmn << 8)) & 0x00FF00FFu, y.bs);
        return (fq_h2f((uint16_t) w.dm) * y.dy) * (float) isum - (fq_h2f((uint16_t)(w.dm >> 16)) * y.dy) * (float) msum;
    }
};
template <> struct fq_kdot<FQ_Q4_K> : fq_kdot45<FQ_Q4_K> {};
template <> struct fq_kdot<FQ_Q5_K> : fq_kdot45<FQ_Q5_K> {};

// ---------------------------------------------------------------- Q6_K: unit ju: super-block ju >> 3, half h = (ju >> 2) & 1, t = (ju >> 1) & 1, g = ju & 1
template <> struct fq_kdot<FQ_Q6_K> {
    static constexpr bool ok = true;
    static constexpr int CB = fq_lay<FQ_Q6_K>::CB;
    static constexpr int O_QH = CB * 128, O_SC = CB * 192, O_D = CB * 208;
    struct lane_t { uint32_t ssh, tsh; int qoff, hoff, soff, doff, aoff, boff, sbl; };
    struct act_t { fq_u4 x0, x1; int b0, b1; float dy; };               // b0, b1 = -32 x the two block sums (the dot chains start from them)
    struct w_t { fq_u4 q, qh; uint32_t s0, s1, d; };
    FQ_HDM static lane_t lane_init(int ju) {
        lane_t L{};
        const int sbl = ju >> 3, h = (ju >> 2) & 1, t = (ju >> 1) & 1, g = ju & 1;
        L.sbl = sbl; L.ssh = 8u * (uint32_t)(2 * t + g); L.tsh = 2u * (uint32_t) t;
        L.qoff = 16 * ju; L.hoff = O_QH + 64 * sbl + 32 * h + 16 * g; L.soff = O_SC + 16 * sbl + 8 * h; L.doff = O_D + 2 * sbl;
        L.aoff = 256 * sbl + 128 * h + 32 * t + 16 * g; L.boff = 16 * sbl + 8 * h + 2 * t + g;
        return L;
    }
    FQ_HDM static act_t act_load(const fq_actcol & a, int p, const lane_t & L) {
        act_t y;
        const int8_t * x = a.qs + 256 * CB * (size_t) p + L.aoff;
        const int16_t * bs = (const int16_t *) a.aux + 16 * CB * (size_t) p + L.boff;
        y.x0 = ld_u4(x); y.x1 = ld_u4(x + 64);
        y.b0 = -32 * (int) bs[0]; y.b1 = -32 * (int) bs[4];
        y.dy = a.d[CB * p + L.sbl];
        return y;
    }
    FQ_HDM static w_t w_load(const uint8_t * col, const lane_t & L) {
        w_t w{};
        w.q = ld_u4(col + L.qoff); w.qh = ld_u4(col + L.hoff);
        const uint8_t * sc = col + L.soff;
        w.s0 = ld_u32(sc); w.s1 = ld_u32(sc + 4);
        w.d = ld_u16(col + L.doff);
        return w;
    }
    FQ_HDM static w_t from_regs(const fq_unit_regs & r) { return w_t{ r.q, r.q2, r.s0, r.s1, r.dm }; }
    FQ_HDM static float dot(const w_t & w, const act_t & y, const lane_t & L) {
        const int sc_lo = fq_sext8(w.s0, (int) L.ssh), sc_hi = fq_sext8(w.s1, (int) L.ssh);        // scales[8h + 2t + g], scales[8h + 2(t + 2) + g]
        const fq_u4 t = shr4(w.qh, (int) L.tsh);                          // bits 0-1 of a byte: quarter t, bits 4-5: quarter t + 2
        const fq_u4 lo = or4(and4(w.q, 0x0F0F0F0Fu), shl4(and4(t, 0x03030303u), 4));
        const fq_u4 hi = or4(and4(shr4(w.q, 4), 0x0F0F0F0Fu), and4(t, 0x30303030u));
        const int isum = fq_mul24(sc_lo, dot16rs(lo, y.x0, y.b0)) + fq_mul24(sc_hi, dot16rs(hi, y.x1, y.b1));      // (|dot - 32 bsum| < 2^18)
        return (fq_h2f((uint16_t) w.d) * y.dy) * (float) isum;
    }
};

// ---------------------------------------------------------------- Q2_K / Q3_K: unit ju: super-block ju >> 2, 128-half hf = (ju >> 1) & 1, group g = ju & 1; 64 elements
template <> struct fq_kdot<FQ_Q2_K> {
    static constexpr bool ok = true;
    static constexpr int CB = fq_lay<FQ_Q2_K>::CB;                       // 16 super-blocks per column
    static constexpr int O_SC = CB * 64, O_DM = CB * 80;
    struct lane_t { uint32_t gsh; int qoff, soff, doff, aoff, boff, sbl; };
    struct act_t { fq_u4 x[4]; uint32_t bs01, bs23; float dy; };
    struct w_t { fq_u4 q; uint32_t s0, s1, dm; };
    FQ_HDM static lane_t lane_init(int ju) {
        lane_t L{};
        const int sbl = ju >> 2, hf = (ju >> 1) & 1, g = ju & 1;
        L.sbl = sbl; L.gsh = 8u * (uint32_t) g;
        L.qoff = 16 * ju; L.soff = O_SC + 16 * sbl + 8 * hf; L.doff = O_DM + 4 * sbl;
        L.aoff = 256 * sbl + 128 * hf + 16 * g; L.boff = 16 * sbl + 8 * hf + g;
        return L;
    }
    FQ_HDM static act_t act_load(const fq_actcol & a, int p, const lane_t & L) {
        act_t y;
        const int8_t * x = a.qs + 256 * CB * (size_t) p + L.aoff;
        const int16_t * bs = (const int16_t *) a.aux + 16 * CB * (size_t) p + L.boff;
        y.x[0] = ld_u4(x); y.x[1] = ld_u4(x + 32); y.x[2] = ld_u4(x + 64); y.x[3] = ld_u4(x + 96);
        y.bs01 = fq_pack16(bs[0], bs[2]); y.bs23 = fq_pack16(bs[4], bs[6]);
        y.dy = a.d[CB * p + L.sbl];
        return y;
    }
    FQ_HDM static w_t w_load(const uint8_t * col, const lane_t & L) {
        w_t w{};
        w.q = ld_u4(col + L.qoff);
        const uint8_t * sc = col + L.soff;
        w.s0 = ld_u32(sc); w.s1 = ld_u32(sc + 4);
        w.dm = ld_u32(col + L.doff);
        return w;
    }
    FQ_HDM static w_t from_regs(const fq_unit_regs & r) { return w_t{ r.q, r.s0, r.s1, r.dm }; }
    FQ_HDM static float dot(const w_t & w, const act_t & y, const lane_t & L) {
        // scales[8hf + 2j + g]: bytes g, 2 + g of s0 (j = 0, 1) and of s1 (j = 2, 3); low nibble = scale, high nibble = min
        const uint32_t b0 = w.s0 >> L.gsh, b1 = w.s1 >> L.gsh;
        const uint32_t sc01 = b0 & 0x000F000Fu, sc23 = b1 & 0x000F000Fu, mn01 = (b0 >> 4) & 0x000F000Fu, mn23 = (b1 >> 4) & 0x000F000Fu;
        // the four 2-bit fields stay where they are: q & (3 << 2j) is 4^j times the field (the last one one bit lower: 0x60 keeps the byte a positive int8), the
        // 16-element dot is shifted back exactly; scales and dots then meet as int16 pairs in two v_dot2_i32_i16 (|dot| <= 16 * 3 * 128)
        const int d0 = dot16r(and4(w.q, 0x03030303u), y.x[0]);
        const int d1 = dot16r(and4(w.q, 0x0C0C0C0Cu), y.x[1]) >> 2;
        const int d2 = dot16r(and4(w.q, 0x30303030u), y.x[2]) >> 4;
        const int d3 = dot16r(and4(shr4(w.q, 1), 0x60606060u), y.x[3]) >> 5;
        const int isum = fq_dot2(sc23, fq_pack16(d2, d3), fq_dot2z(sc01, fq_pack16(d0, d1)));
        const int msum = fq_dot2(mn23, y.bs23, fq_dot2z(mn01, y.bs01));
        return (y.dy * fq_h2f((uint16_t) w.dm)) * (float) isum - (y.dy * fq_h2f((uint16_t)(w.dm >> 16))) * (float) msum;
    }
};
template <> struct fq_kdot<FQ_Q3_K> {
    static constexpr bool ok = true;
    static constexpr int CB = fq_lay<FQ_Q3_K>::CB;
    static constexpr int O_HM = CB * 64, O_SC = CB * 96, O_D = CB * 108;
    struct lane_t { uint32_t ssh, hsh; int qoff, hoff, soff, doff, aoff, boff, sbl; };
    struct act_t { fq_u4 x[4]; int b4[4]; float dy; };                  // b4[j] = -4 x block sum j x the factor its dot carries (1, 4, 1, 16): the dot chains start from them
    struct w_t { fq_u4 q, hm; uint32_t s0, s1, s2, d; };
    FQ_HDM static lane_t lane_init(int ju) {
        lane_t L{};
        const int sbl = ju >> 2, hf = (ju >> 1) & 1, g = ju & 1;
        L.sbl = sbl; L.ssh = 8u * (uint32_t) g + 4u * (uint32_t) hf; L.hsh = 4u * (uint32_t) hf;
        L.qoff = 16 * ju; L.hoff = O_HM + 32 * sbl + 16 * g; L.soff = O_SC + 12 * sbl; L.doff = O_D + 2 * sbl;
        L.aoff = 256 * sbl + 128 * hf + 16 * g; L.boff = 16 * sbl + 8 * hf + g;
        return L;
    }
    FQ_HDM static act_t act_load(const fq_actcol & a, int p, const lane_t & L) {
        act_t y;
        const int8_t * x = a.qs + 256 * CB * (size_t) p + L.aoff;
        const int16_t * bs = (const int16_t *) a.aux + 16 * CB * (size_t) p + L.boff;
        y.x[0] = ld_u4(x); y.x[1] = ld_u4(x + 32); y.x[2] = ld_u4(x + 64); y.x[3] = ld_u4(x + 96);
        y.b4[0] = -4 * (int) bs[0]; y.b4[1] = -16 * (int) bs[2]; y.b4[2] = -4 * (int) bs[4]; y.b4[3] = -64 * (int) bs[6];
        y.dy = a.d[CB * p + L.sbl];
        return y;
    }
    FQ_HDM static w_t w_load(const uint8_t * col, const lane_t & L) {
        w_t w{};
        w.q = ld_u4(col + L.qoff); w.hm = ld_u4(col + L.hoff);
        const uint8_t * sc = col + L.soff;
        w.s0 = ld_u32(sc); w.s1 = ld_u32(sc + 4); w.s2 = ld_u32(sc + 8);
        w.d = ld_u16(col + L.doff);
        return w;
    }
    FQ_HDM static w_t from_regs(const fq_unit_regs & r) { return w_t{ r.q, r.q2, r.s0, r.s1, r.s2, r.dm }; }
    FQ_HDM static float dot(const w_t & w, const act_t & y, const lane_t & L) {
        // scale is = 8hf + 2j + g (k_quants.c:491-496): low 4 bits = nibble hf of byte (2j + g) & 3 of s0 (j < 2) / s1 (j >= 2); high 2 bits = bits
        // 2 (is >> 2), +1 of byte (2j + g) & 3 of s2, is >> 2 = 2hf + (j >> 1)
        const uint32_t n0 = (w.s0 >> L.ssh) & 0x000F000Fu, n1 = (w.s1 >> L.ssh) & 0x000F000Fu;      // j = 0 | 1 << 16,  j = 2 | 3 << 16
        const uint32_t t = w.s2 >> L.ssh;
        const uint32_t s01 = n0 | ((t & 0x00030003u) << 4), s23 = n1 | (((t >> 2) & 0x00030003u) << 4);
        const fq_u4 u = shr4(w.hm, (int) L.hsh);                          // bit j of a byte: the high bit of the element in group j of this half
        // groups 1 and 3 stay where a single shift of u serves both: 4 x (low pair at bits 2-3, high bit at 4) and 16 x (low pair at 4-5, high bit at 6: <= 112, a
        // positive int8); their 16-element dots start from -16 / -64 x the block sum and are shifted back exactly
        const fq_u4 u3 = shl4(u, 3);
        const fq_u4 q0 = or4(and4(w.q, 0x03030303u),          and4(shl4(u, 2), 0x04040404u));
        const fq_u4 q1 = or4(and4(w.q, 0x0C0C0C0Cu),          and4(u3, 0x10101010u));
        const fq_u4 q2 = or4(and4(shr4(w.q, 4), 0x03030303u), and4(u, 0x04040404u));
        const fq_u4 q3 = or4(and4(shr4(w.q, 2), 0x30303030u), and4(u3, 0x40404040u));
        // (scale - 32) of both 16-bit lanes at once, the four sub-block dots (|dot - 4 bsum| <= 22 464) as int16 pairs: two v_dot2_i32_i16
        const uint32_t c01 = fq_pk_sub16(s01, 0x00200020u), c23 = fq_pk_sub16(s23, 0x00200020u);
        const int e0 = dot16rs(q0, y.x[0], y.b4[0]), e1 = dot16rs(q1, y.x[1], y.b4[1]) >> 2, e2 = dot16rs(q2, y.x[2], y.b4[2]), e3 = dot16rs(q3, y.x[3], y.b4[3]) >> 4;
        const int isum = fq_dot2(c23, fq_pack16(e2, e3), fq_dot2z(c01, fq_pack16(e0, e1)));
        return (fq_h2f((uint16_t) w.d) * y.dy) * (float) isum;
    }
};
