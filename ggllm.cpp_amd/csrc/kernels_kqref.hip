// kernels_kqref.hip -- the stand-alone mat-vec in the REFERENCE'S OWN association at wave speed (round 6; ggml_hip_reference_order(2)): the five k-quants (k_gemv_kq_ref,
// below) and, at the end of the file, the legacy formats' op-level form (k_gemv_legacy_ref).
//
// The scalar branches of ggml_vec_dot_q{2,4,5}_K_q8_K (k_quants.c:1267-1306, 1999-2055, 2340-2400; caller ggml.c:11484-11516) do not add one term per
// 32-element group the way the legacy formats do:
//   Q2_K          one term per 256-element super-block, `sumf += dall * isum - dmin * summs` (isum, summs: integers over the whole super-block);
//   Q4_K / Q5_K   EIGHT float lanes, `sums[l] += d * aux32[l]` per super-block with aux32[l] = sum over the super-block's elements e = l (mod 8) of
//                 scale(e) * q(e) * q8(e), a ninth chain `sumf -= dmin * sumi`, and `sumf += sums[l]`, l = 0..7, after the last super-block.
// The fast kernels' 16-byte units cover 32 CONSECUTIVE elements: their integer sums are another partition of the super-block's. Here a lane still loads one unit
// with one global_load_dwordx4 (the coalesced stream of every other kernel, fq_units.h), but
//   * Q4_K / Q5_K: the unit's bytes are transposed in registers (v_perm_b32) so that one dot4 holds ONE residue class -- [w(i), w(i + 8)] against the activation
//     pair [x(i), x(i + 8)], which the workgroup has laid out, zero-padded, in LDS -- giving the unit's share of all eight aux32[l]; the eight units of a super-block
//     (eight neighbouring lanes) add their shares by a transposing exchange (xor 4, 2, 1: lane t ends with residue t's total): integer adds, exact in any order;
//   * Q2_K: isum and summs of the four units of a super-block are added across four lanes;
//   * the lane that holds a total forms the reference's f32 term (`d * (float) aux32`, `dmin * (float) sumi`, `dall * isum - dmin * summs`) and drops it into an LDS
//     strip [row][chain][super-block]; after the row's last unit lanes = (row, chain) add each strip left to right (fq_ref_chain.h) and the row's first lane adds the
//     eight lane sums in order.
// Every f32 operation is the reference's, in the reference's order: the results are bit-identical to k_mul_mat_ref (mode 1) and to the reference's scalar build
// (tests/test_gpu_kqref.py). The integer work is ~2 x the default kernels' per unit (perms, sixteen half-filled dot4, the exchange), so this form runs at roughly half
// their speed -- against one thread per output in mode 1. Q6_K (k_quants.c:2748-2789: int8 scales per 16 elements) has Q4_K's unit shape; Q3_K (k_quants.c:1684-1746) has
// units of four 16-element scale blocks and super-blocks of four lanes, which end the exchange with two residues each.
#include "fq_block_dev.h"
#include "fq_units.h"
#include "fq_ref_chain.h"
#include "kernels.h"
#include "hip_context.h"

namespace {

constexpr int KQ_R = 2;                                                     // rows of a run (per wave)

template <int TYPE> struct kq_ref_fmt {
    static constexpr bool LANES8 = (TYPE == FQ_Q4_K || TYPE == FQ_Q5_K || TYPE == FQ_Q6_K);
    static constexpr bool MINS = (TYPE == FQ_Q4_K || TYPE == FQ_Q5_K);      // the ninth chain, sumf -= dmin * sumi
    static constexpr bool Q3 = (TYPE == FQ_Q3_K);                           // eight float lanes too, but a unit = four 16-element scale blocks and a super-block = four lanes
    static constexpr int  NCH = (LANES8 || Q3) ? (MINS ? 9 : 8) : 1;        // chains per row: eight float lanes (+ the mins), or the one sum
    static constexpr int  UPS = LANES8 ? 8 : 4;                             // units (lanes) per super-block
};

__device__ __forceinline__ int swz_xor4(int v) { return __builtin_amdgcn_ds_swizzle(v, 0x101F); }      // lane ^ 4 (bit-mask mode: and 0x1F, or 0, xor 4)
__device__ __forceinline__ int xor1(int v) { return dpp_mov<0xB1>(v); }     // quad_perm [1,0,3,2]
__device__ __forceinline__ int xor2(int v) { return dpp_mov<0x4E>(v); }     // quad_perm [2,3,0,1]

// LDS bytes: the activation column (Q4_K / Q5_K: four planes of K / 32 uint4 = 2 K bytes, the residue pairs zero-padded; Q2_K: the K int8 as they are),
// the column's d (K / 256 floats) and bsums (K / 16 int16), and per wave a strip of KQ_R rows x NCH chains
__host__ __device__ inline size_t kq_ref_act_bytes(int type, int64_t K) { return (size_t)((type == FQ_Q2_K) ? K : 2 * K); }      // (Q3_K: eight planes of K / 64 uint4 = 2 K bytes as well)
__host__ __device__ inline size_t kq_ref_lds(int type, int64_t K, int nw) {
    const int nsb = (int)(K / 256), nch = (type == FQ_Q2_K) ? 1 : ((type == FQ_Q6_K || type == FQ_Q3_K) ? 8 : 9);
    return kq_ref_act_bytes(type, K) + (((size_t) nsb * 36 + 15) & ~(size_t) 15) + (size_t) nw * KQ_R * nch * fq_ref_strip_stride(nsb) * 4;
}

}   // namespace

template <int TYPE>
__global__ void __launch_bounds__(512) k_gemv_kq_ref(fq_weight w, fq_act act, float * dst, int64_t ldd, fq_gemv_epi ep, int rows_per_wave) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    typedef kq_ref_fmt<TYPE> F;
    constexpr int NCH = F::NCH, R = KQ_R;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, nw = blockDim.x >> 6, nt = blockDim.x;
    const int64_t K = w.K, M = w.M, col = blockIdx.y;
    const int nsb = (int)(K >> 8), units = (int)(K / fq_unit<TYPE>::ELEMS), ncolu = (units + 63) >> 6;
    const unsigned SW = fq_ref_strip_stride(nsb);
    uint8_t * actx = smem;
    float   * dxs  = (float *)(smem + kq_ref_act_bytes(TYPE, K));
    int16_t * bss  = (int16_t *)(dxs + nsb);
    float   * strip_w = (float *)((uint8_t *) dxs + (((size_t) nsb * 36 + 15) & ~(size_t) 15)) + (size_t) wid * R * NCH * SW;      // (16-byte aligned: the chains read float4)
    // ---- the activation column -> LDS
    const uint8_t * img = act.base + (size_t) col * fq_act_col_bytes(FQ_Q8_K, K);
    {
        const float * dsrc = (const float *)(img + fq_act_d_off(FQ_Q8_K, K));
        const int16_t * bsrc = (const int16_t *)(img + fq_act_aux_off(FQ_Q8_K, K));
        for (int i = tid; i < nsb; i += nt) dxs[i] = dsrc[i];
        for (int i = tid; i < 16 * nsb; i += nt) bss[i] = bsrc[i];
        if constexpr (F::LANES8) {
            // unit u = (sb, c, g), half h (low / high nibbles): the 16 elements x[0..15] at 256 sb + 64 c + 32 h + 16 g, as eight dwords, residue r = 0..7:
            // r even: [x(r), x(r + 8), 0, 0], r odd: [0, 0, x(r), x(r + 8)] -- plane (2 k4 + h) holds residues 4 k4 .. 4 k4 + 3 of every unit as one uint4
            const int U = units;
            uint4 * xa = (uint4 *) actx;
            for (int i = tid; i < 2 * U; i += nt) {
                const int u = i >> 1, h = i & 1;
                const int sb = u >> 3, c = (u >> 1) & 3, g = u & 1;
                // (Q6_K: unit = (sb, 128-half c >> 1, quarter c & 1, g); its low nibbles are quarter (c & 1), its high nibbles quarter (c & 1) + 2: 64 elements on)
                const size_t e0 = (TYPE == FQ_Q6_K) ? 256 * (size_t) sb + 128 * (c >> 1) + 32 * (c & 1) + 64 * h + 16 * g : 256 * (size_t) sb + 64 * c + 32 * h + 16 * g;
                const uint4 x = *(const uint4 *)(img + e0);
                const unsigned xw[4] = { x.x, x.y, x.z, x.w };
                auto xb = [&](int e) { return (xw[e >> 2] >> (8 * (e & 3))) & 0xFFu; };
                unsigned o[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) { const unsigned p = xb(r) | (xb(r + 8) << 8); o[r] = (r & 1) ? p << 16 : p; }
                xa[(size_t)(0 + h) * U + u] = make_uint4(o[0], o[1], o[2], o[3]);
                xa[(size_t)(2 + h) * U + u] = make_uint4(o[4], o[5], o[6], o[7]);
            }
        } else if constexpr (F::Q3) {
            // unit u = (sb, hf, g), block j = 0..3 (the unit's 2-bit field j): the 16 elements at 256 sb + 128 hf + 32 j + 16 g -> plane 2 j + k4 as above
            const int U = units;
            uint4 * xa = (uint4 *) actx;
            for (int i = tid; i < 4 * U; i += nt) {
                const int u = i >> 2, j = i & 3;
                const int sb = u >> 2, hf = (u >> 1) & 1, g = u & 1;
                const uint4 x = *(const uint4 *)(img + 256 * (size_t) sb + 128 * hf + 32 * j + 16 * g);
                const unsigned xw[4] = { x.x, x.y, x.z, x.w };
                auto xb = [&](int e) { return (xw[e >> 2] >> (8 * (e & 3))) & 0xFFu; };
                unsigned o[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) { const unsigned p = xb(r) | (xb(r + 8) << 8); o[r] = (r & 1) ? p << 16 : p; }
                xa[(size_t)(2 * j + 0) * U + u] = make_uint4(o[0], o[1], o[2], o[3]);
                xa[(size_t)(2 * j + 1) * U + u] = make_uint4(o[4], o[5], o[6], o[7]);
            }
        } else {
            for (int64_t i = tid; i < (K >> 4); i += nt) ((uint4 *) actx)[i] = ((const uint4 *) img)[i];
        }
    }
    __syncthreads();

    const int64_t wrow0 = ((int64_t) blockIdx.x * nw + wid) * rows_per_wave;
    for (int64_t row0 = wrow0; row0 < wrow0 + rows_per_wave && row0 < M; row0 += R) {
        fq_wrow rows[R];
#pragma unroll
        for (int r = 0; r < R; ++r) { const int64_t row = row0 + r; rows[r] = fq_row<TYPE>(w, row < M ? row : M - 1); }
        // (the next unit column is requested before the current one's arithmetic: two columns of R rows in flight per wave)
        fq_unit_regs nxt[R];
#pragma unroll
        for (int r = 0; r < R; ++r) nxt[r] = fq_unit_load_col<TYPE>(rows[r], 0, lane, units);
        for (int c0 = 0; c0 < ncolu; ++c0) {
            fq_unit_regs regs[R];
#pragma unroll
            for (int r = 0; r < R; ++r) regs[r] = nxt[r];
            if (c0 + 1 < ncolu) {
#pragma unroll
                for (int r = 0; r < R; ++r) nxt[r] = fq_unit_load_col<TYPE>(rows[r], c0 + 1, lane, units);
            }
            const int u = 64 * c0 + lane;
            const bool ok = u < units;                                      // (K % 256 == 0: a super-block's lanes are in or out together)
            const int uc = ok ? u : units - 1;
            if constexpr (F::LANES8) {
                const int sb = uc >> 3, c = (uc >> 1) & 3, g = uc & 1, t = lane & 7;
                const uint4 * xa = (const uint4 *) actx;
                const uint4 xl0 = xa[(size_t) 0 * units + uc], xh0 = xa[(size_t) 1 * units + uc], xl1 = xa[(size_t) 2 * units + uc], xh1 = xa[(size_t) 3 * units + uc];
                const int bs0 = F::MINS ? bss[16 * sb + 4 * c + g] : 0, bs1 = F::MINS ? bss[16 * sb + 4 * c + g + 2] : 0;
                const float dy = dxs[sb];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const fq_unit_regs & q = regs[r];
                    int sc0, mn0 = 0, sc1, mn1 = 0;
                    fq_u4 lo, hi;
                    if constexpr (TYPE == FQ_Q6_K) {                        // k_quants.c:2756-2771: q = (low 4 | high 2 << 4) - 32, int8 scales per 16 elements
                        const int t01 = c & 1;
                        const uint64_t sc8 = (uint64_t) q.s0 | ((uint64_t) q.s1 << 32);
                        sc0 = (int)(int8_t)(sc8 >> (8 * (2 * t01 + g)));
                        sc1 = (int)(int8_t)(sc8 >> (8 * (2 * (t01 + 2) + g)));
                        const fq_u4 l6 = or4(and4(q.q, 0x0F0F0F0Fu),          shl4(and4(shr4(q.q2, 2 * t01),     0x03030303u), 4));
                        const fq_u4 h6 = or4(and4(shr4(q.q, 4), 0x0F0F0F0Fu), shl4(and4(shr4(q.q2, 2 * t01 + 4), 0x03030303u), 4));
                        // per byte v - 32 as int8, borrow-free: ((v | 0x80) - 0x20) ^ 0x80
                        auto m32 = [](uint32_t v) { return ((v | 0x80808080u) - 0x20202020u) ^ 0x80808080u; };
                        lo = fq_u4{ m32(l6.x), m32(l6.y), m32(l6.z), m32(l6.w) };
                        hi = fq_u4{ m32(h6.x), m32(h6.y), m32(h6.z), m32(h6.w) };
                    } else {
                        k4_scale_min(q.s0, q.s1, q.s2, 2 * c, sc0, mn0); k4_scale_min(q.s0, q.s1, q.s2, 2 * c + 1, sc1, mn1);
                        lo = and4(q.q, 0x0F0F0F0Fu); hi = and4(shr4(q.q, 4), 0x0F0F0F0Fu);
                        if constexpr (TYPE == FQ_Q5_K) {                    // the 5th bits: bit 2c / 2c + 1 of the unit's qh bytes (k_quants.c:2369-2378)
                            lo = or4(lo, shl4(and4(shr4(q.q2, 2 * c),     0x01010101u), 4));
                            hi = or4(hi, shl4(and4(shr4(q.q2, 2 * c + 1), 0x01010101u), 4));
                        }
                    }
                    // [a.b0, b.b0, a.b1, b.b1] and [a.b2, b.b2, a.b3, b.b3] of the dword pairs (0, 2) and (1, 3): one residue class per dot4 half
                    auto p01 = [](unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x05010400u); };
                    auto p23 = [](unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x07030602u); };
                    auto lanes8 = [&](const fq_u4 & v, const uint4 & x0, const uint4 & x1, int (&A)[8]) {
                        const unsigned e01 = p01(v.x, v.z), e23 = p23(v.x, v.z), o01 = p01(v.y, v.w), o23 = p23(v.y, v.w);
                        A[0] = fq_dot4z(e01, x0.x); A[1] = fq_dot4z(e01, x0.y); A[2] = fq_dot4z(e23, x0.z); A[3] = fq_dot4z(e23, x0.w);
                        A[4] = fq_dot4z(o01, x1.x); A[5] = fq_dot4z(o01, x1.y); A[6] = fq_dot4z(o23, x1.z); A[7] = fq_dot4z(o23, x1.w);
                    };
                    int Al[8], Ah[8], a[8];
                    lanes8(lo, xl0, xl1, Al); lanes8(hi, xh0, xh1, Ah);
#pragma unroll
                    for (int k = 0; k < 8; ++k) a[k] = ok ? sc0 * Al[k] + sc1 * Ah[k] : 0;      // aux32[l] += scale * aux16[l] (k_quants.c:2036-2043): integers
                    int ms = (F::MINS && ok) ? mn0 * bs0 + mn1 * bs1 : 0;                         // sumi's share (k_quants.c:2026-2027)
                    // the super-block's eight lanes add their shares: lane t ends with residue t
                    const bool b2 = t & 4, b1 = t & 2, b0 = t & 1;
                    int a1[4], a2[2];
#pragma unroll
                    for (int k = 0; k < 4; ++k) { const int keep = b2 ? a[4 + k] : a[k], send = b2 ? a[k] : a[4 + k]; a1[k] = keep + swz_xor4(send); }
#pragma unroll
                    for (int k = 0; k < 2; ++k) { const int keep = b1 ? a1[2 + k] : a1[k], send = b1 ? a1[k] : a1[2 + k]; a2[k] = keep + xor2(send); }
                    const int tot = (b0 ? a2[1] : a2[0]) + xor1(b0 ? a2[0] : a2[1]);
                    if constexpr (F::MINS) { ms += swz_xor4(ms); ms += xor2(ms); ms += xor1(ms); }
                    if (ok) {
                        const float d = fq_h2f((uint16_t) q.dm) * dy;                            // k_quants.c:2045 / 2781: d = fp16(x.d) * y.d
                        strip_w[(size_t)(r * NCH + t) * SW + sb] = d * (float) tot;              // sums[l] += d * aux32[l]
                        if constexpr (F::MINS) { if (t == 0) strip_w[(size_t)(r * NCH + 8) * SW + sb] = -((fq_h2f((uint16_t)(q.dm >> 16)) * dy) * (float) ms); }      // sumf -= dmin * sumi
                    }
                }
            } else if constexpr (F::Q3) {                                  // Q3_K (k_quants.c:1684-1746)
                const int sb = uc >> 2, hf = (uc >> 1) & 1, g = uc & 1, t = lane & 3;
                const uint4 * xa = (const uint4 *) actx;
                const float dy = dxs[sb];
                auto p01 = [](unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x05010400u); };
                auto p23 = [](unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x07030602u); };
                auto m4 = [](uint32_t v) { return ((v | 0x80808080u) - 0x04040404u) ^ 0x80808080u; };      // per byte v - 4 as int8, borrow-free
                int a[R][8];
#pragma unroll
                for (int r = 0; r < R; ++r) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) a[r][k] = 0;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint4 x0 = xa[(size_t)(2 * j) * units + uc], x1 = xa[(size_t)(2 * j + 1) * units + uc];
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const fq_unit_regs & q = regs[r];
                        const fq_u4 l2 = and4(shr4(q.q, 2 * j), 0x03030303u);
                        const fq_u4 hb = shl4(and4(shr4(q.q2, 4 * hf + j), 0x01010101u), 2);                 // high bit set -> + 4, then - 4 for all (k_quants.c:1701-1716)
                        const fq_u4 v = { m4(l2.x | hb.x), m4(l2.y | hb.y), m4(l2.z | hb.z), m4(l2.w | hb.w) };
                        const int sc = q3_scale(q.s0, q.s1, q.s2, 8 * hf + 2 * j + g) - 32;
                        const unsigned e01 = p01(v.x, v.z), e23 = p23(v.x, v.z), o01 = p01(v.y, v.w), o23 = p23(v.y, v.w);
                        a[r][0] += sc * fq_dot4z(e01, x0.x); a[r][1] += sc * fq_dot4z(e01, x0.y); a[r][2] += sc * fq_dot4z(e23, x0.z); a[r][3] += sc * fq_dot4z(e23, x0.w);
                        a[r][4] += sc * fq_dot4z(o01, x1.x); a[r][5] += sc * fq_dot4z(o01, x1.y); a[r][6] += sc * fq_dot4z(o23, x1.z); a[r][7] += sc * fq_dot4z(o23, x1.w);
                    }
                }
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    // the super-block's four lanes add their shares: lane t ends with residues 2 t and 2 t + 1
                    const bool b1 = t & 2, b0 = t & 1;
                    int a1[4], a2[2];
#pragma unroll
                    for (int k = 0; k < 4; ++k) { const int own = ok ? a[r][(b1 ? 4 : 0) + k] : 0, oth = ok ? a[r][(b1 ? 0 : 4) + k] : 0; a1[k] = own + xor2(oth); }
#pragma unroll
                    for (int k = 0; k < 2; ++k) { const int own = b0 ? a1[2 + k] : a1[k], oth = b0 ? a1[k] : a1[2 + k]; a2[k] = own + xor1(oth); }
                    if (ok) {
                        const float d = fq_h2f((uint16_t) regs[r].dm) * dy;                       // k_quants.c:1739: d = fp16(x.d) * y.d
                        strip_w[(size_t)(r * NCH + 2 * t) * SW + sb] = d * (float) a2[0];          // sums[l] += d * aux32[l]
                        strip_w[(size_t)(r * NCH + 2 * t + 1) * SW + sb] = d * (float) a2[1];
                    }
                }
            } else {                                                       // Q2_K (k_quants.c:1267-1306)
                const int sb = uc >> 2, hf = (uc >> 1) & 1, g = uc & 1, t = lane & 3;
                const int8_t  * x  = (const int8_t *) actx + 256 * (size_t) sb + 128 * hf + 16 * g;
                const int16_t * bs = bss + 16 * (size_t) sb + 8 * hf + g;
                const float dy = dxs[sb];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const fq_unit_regs & q = regs[r];
                    const uint64_t sc8 = (uint64_t) q.s0 | ((uint64_t) q.s1 << 32);
                    int isum = 0, msum = 0;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const uint32_t sc = (uint32_t)(sc8 >> (8 * (2 * j + g))) & 0xFFu;
                        isum += (int)(sc & 0xFu) * dot16(and4(shr4(q.q, 2 * j), 0x03030303u), x + 32 * j);
                        msum += (int)(sc >> 4) * (int) bs[2 * j];
                    }
                    isum = ok ? isum : 0; msum = ok ? msum : 0;
                    isum += xor1(isum); isum += xor2(isum);
                    msum += xor1(msum); msum += xor2(msum);
                    if (ok && t == 0) {
                        const float dall = dy * fq_h2f((uint16_t) q.dm), dmin = dy * fq_h2f((uint16_t)(q.dm >> 16));
                        strip_w[(size_t)(r * NCH) * SW + sb] = dall * (float) isum - dmin * (float) msum;
                    }
                }
            }
        }
        // ---- the chains: lane = (row of the run, chain), left to right over the super-blocks (the term stores above are ahead in this wave's LDS queue)
        const int cl = lane < R * NCH ? lane : 0;
        float v = fq_ref_chain(strip_w + (size_t) cl * SW, nsb, 0.0f);
        if constexpr (F::LANES8 || F::Q3) {
            // sumf (the mins' chain, lane 9 r + 8) += sums[0], .., sums[7] in this order (k_quants.c:2052-2053)
            const int rr = cl / NCH;
            float s = F::MINS ? __shfl(v, rr * NCH + 8) : 0.0f;
#pragma unroll
            for (int l = 0; l < 8; ++l) s += __shfl(v, rr * NCH + l);
            v = s;
        }
        const int64_t row = row0 + lane / NCH;
        if (lane < R * NCH && lane % NCH == 0 && row < M) {
            if (ep.mode == FQ_EPI_GELU)      v = h2f_bits(ep.gelu_table[f2h_bits(v)]);                                          // ggml.c:3477-3484
            else if (ep.mode == FQ_EPI_ADD2) v = (v + ep.add1[col * ep.ld_add + row]) + ep.add2[col * ep.ld_add + row];         // libfalcon.cpp:2399-2400
            dst[col * ldd + row] = v;
        }
    }
}


// ---- the legacy formats' stand-alone mat-vec in the reference's association (the op-level API and the ggml-cuda.h shim under mode 2; the resident model's single-token steps
// run the fused launches, kernels_ring.hip / kernels_decode.hip): a lane's unit IS a block, its term the reference's (fq_units.h), the strip [row][block] is added left to right
template <int TYPE>
__global__ void __launch_bounds__(512) k_gemv_legacy_ref(fq_weight w, fq_act act, float * dst, int64_t ldd, fq_gemv_epi ep, int rows_per_wave) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr int ACT = fq_act_of(TYPE), R = KQ_R;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, nw = blockDim.x >> 6, nt = blockDim.x;
    const int64_t K = w.K, M = w.M, col = blockIdx.y;
    const int units = (int)(K / 32), ncolu = (units + 63) >> 6;
    const unsigned SW = fq_ref_strip_stride(units);
    const size_t imgb = fq_act_col_bytes(ACT, K);
    float * strip_w = (float *)(smem + imgb) + (size_t) wid * R * SW;
    {
        const uint4 * src = (const uint4 *)(act.base + (size_t) col * imgb);
        for (int64_t i = tid; i < (int64_t)(imgb >> 4); i += nt) ((uint4 *) smem)[i] = src[i];
    }
    __syncthreads();
    const fq_actcol acol = { (const int8_t *) smem, (const float *)(smem + fq_act_d_off(ACT, K)), (const void *)(smem + fq_act_aux_off(ACT, K)) };
    const int64_t wrow0 = ((int64_t) blockIdx.x * nw + wid) * rows_per_wave;
    for (int64_t row0 = wrow0; row0 < wrow0 + rows_per_wave && row0 < M; row0 += R) {
        fq_wrow rows[R];
#pragma unroll
        for (int r = 0; r < R; ++r) { const int64_t row = row0 + r; rows[r] = fq_row<TYPE>(w, row < M ? row : M - 1); }
        for (int c0 = 0; c0 < ncolu; ++c0) {
            fq_unit_regs regs[R];
#pragma unroll
            for (int r = 0; r < R; ++r) regs[r] = fq_unit_load_col<TYPE>(rows[r], c0, lane, units);
            const int u = 64 * c0 + lane, uc = u < units ? u : units - 1;      // (a lane beyond the row holds the clamped re-read of the last unit: the same term into the same word)
#pragma unroll
            for (int r = 0; r < R; ++r) strip_w[(size_t) r * SW + uc] = fq_unit<TYPE>::dot(regs[r], acol, uc);
        }
        const int cl = lane < R ? lane : 0;
        float v = fq_ref_chain(strip_w + (size_t) cl * SW, units, 0.0f);          // ggml.c:2594-2609: sumf = 0; sumf += term_i, i ascending
        const int64_t row = row0 + lane;
        if (lane < R && row < M) {
            if (ep.mode == FQ_EPI_GELU)      v = h2f_bits(ep.gelu_table[f2h_bits(v)]);
            else if (ep.mode == FQ_EPI_ADD2) v = (v + ep.add1[col * ep.ld_add + row]) + ep.add2[col * ep.ld_add + row];
            dst[col * ldd + row] = v;
        }
    }
}

static bool legacy_ref_launch(const fq_weight & w, const fq_act & act, int64_t N, float * dst, int64_t ldd, const fq_gemv_epi & ep, hipStream_t st) {
    const int ACT = fq_desc(w.type).act_type;
    if (w.K % 32 || act.type != ACT || act.K != w.K) return false;
    const int units = (int)(w.K / 32), nw = 8;
    const size_t lds = fq_act_col_bytes(ACT, w.K) + (size_t) nw * KQ_R * fq_ref_strip_stride(units) * 4;
    if (lds > 160 * 1024) return false;
    FQ_TL(st, "gemv_legacy_ref");
    const int n_cu = fq_ctx().n_cu;
    int64_t rpw = (w.M + (int64_t) 4 * n_cu * nw - 1) / ((int64_t) 4 * n_cu * nw);
    rpw = ((rpw + KQ_R - 1) / KQ_R) * KQ_R;
    if (rpw < KQ_R) rpw = KQ_R;
    const unsigned blocks = (unsigned)((w.M + rpw * nw - 1) / (rpw * nw));
#define FQ_CASE(T) case T: { static size_t g = 0; if (lds > 64 * 1024 && lds > g) { HIP_CHECK(hipFuncSetAttribute((const void *) k_gemv_legacy_ref<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds)); g = lds; } \
        hipLaunchKernelGGL((k_gemv_legacy_ref<T>), dim3(blocks, (unsigned) N), dim3(64 * nw), lds, st, w, act, dst, ldd, ep, (int) rpw); } break;
    switch (w.type) { FQ_CASE(FQ_Q4_0) FQ_CASE(FQ_Q4_1) FQ_CASE(FQ_Q5_0) FQ_CASE(FQ_Q5_1) FQ_CASE(FQ_Q8_0) default: return false; }
#undef FQ_CASE
    return true;
}

bool fq_gemv_kq_ref_supported(const fq_weight & w) {
    return (w.type == FQ_Q2_K || w.type == FQ_Q3_K || w.type == FQ_Q4_K || w.type == FQ_Q5_K || w.type == FQ_Q6_K) && w.K % 256 == 0 && w.K >= 256 && kq_ref_lds(w.type, w.K, 4) <= 160 * 1024;
}

// dst[col * ldd + row], col < N: the mat-vec per column (N > 1: the columns one after the other over the same weights -- short batches; a prompt re-reads the matrix per token)
bool fq_launch_gemv_kq_ref(const fq_weight & w, const fq_act & act, int64_t N, float * dst, int64_t ldd, const fq_gemv_epi & ep, hipStream_t st) {
    if (fq_desc(w.type).blck == 32) return legacy_ref_launch(w, act, N, dst, ldd, ep, st);      // the legacy formats' form
    if (!fq_gemv_kq_ref_supported(w) || act.type != FQ_Q8_K || act.K != w.K) return false;
    FQ_TL(st, "gemv_kq_ref");
    int nw = 8;
    while (nw > 4 && kq_ref_lds(w.type, w.K, nw) > 150 * 1024) nw -= 2;
    const size_t lds = kq_ref_lds(w.type, w.K, nw);
    // rows per wave: a multiple of the run, sized for ~4 workgroups per CU over the matrix
    const int n_cu = fq_ctx().n_cu;
    int64_t rpw = (w.M + (int64_t) 4 * n_cu * nw - 1) / ((int64_t) 4 * n_cu * nw);
    rpw = ((rpw + KQ_R - 1) / KQ_R) * KQ_R;
    if (rpw < KQ_R) rpw = KQ_R;
    const unsigned blocks = (unsigned)((w.M + rpw * nw - 1) / (rpw * nw));
#define FQ_CASE(T) case T: { static size_t g = 0; if (lds > 64 * 1024 && lds > g) { HIP_CHECK(hipFuncSetAttribute((const void *) k_gemv_kq_ref<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds)); g = lds; } \
        hipLaunchKernelGGL((k_gemv_kq_ref<T>), dim3(blocks, (unsigned) N), dim3(64 * nw), lds, st, w, act, dst, ldd, ep, (int) rpw); } break;
    switch (w.type) { FQ_CASE(FQ_Q2_K) FQ_CASE(FQ_Q3_K) FQ_CASE(FQ_Q4_K) FQ_CASE(FQ_Q5_K) FQ_CASE(FQ_Q6_K) default: return false; }
#undef FQ_CASE
    return true;
}
