// kernels_gemm.hip -- prefill-time quantized mat-mul on the int8 matrix cores (gfx950 v_mfma_i32_32x32x32_i8).
//
// Replaces ggml_compute_forward_mul_mat_q_f32 for N > 4 columns (ggml.c:11318-11529; CUDA twin: dequantize to fp16 +
// cublasGemmEx, ggml-cuda.cu:2353-2403) with the CPU path's arithmetic: activations are the Q8 images the decode path
// uses, every 32-element group is ONE 32x32x32 int8 MFMA whose int32 result is exact, and the per-group scales are
// applied in f32 afterwards with the reference's own per-block expression, in block order (bit-identical to the
// reference's scalar vec_dot for the legacy formats):  dst[n][m] += (dW[m][g] * dX[n][g]) * C[n][m] (+ minW[m][g] * sX[n][g]).
// (The reference's fp16 GEMM path does NOT reproduce its own CPU results -- SURVEY hard part 1; this does, up to the
// association of the f32 sum over groups.)
//
// What bounds it. The exact per-group scaling costs ~3-4 VALU operations per (token, row, group): 16 values per lane and
// MFMA, ~50 VALU instructions (200 cycles) against the MFMA's 32 -- the kernel is VALU-issue bound, the matrix pipe is
// nearly free, and the f32 sum over groups is sequential per output (no split-K). The design therefore spends MFMAs to
// buy parallelism: a workgroup is 128 tokens x 32 weight rows; wave w owns the 32 tokens of tile w & 3, and S waves share
// each tile -- all S issue the tile's MFMA, each applies the scales to its own 16/S of the 16 result registers (S = 4 for
// the 4544-row matrices of Falcon-7B: 142 workgroups of 16 waves; S = 1 for the 18176-row one: 568 workgroups of 4).
// Per K-stage of 128 (4 groups), software-pipelined: the next stage's global loads are in flight while this stage's
// MFMAs + scaling run out of the current LDS buffer (two buffers, one barrier per stage):
//   weights  : 16-byte quant groups -> registers -> sign-corrected int8 -> LDS  Wq[row][144] (+ dW, minW per group)
//   tokens   : int8 image rows                                        -> LDS  Xq[tok][144] (+ dX, sX per group)
//   per group: A = ds_read_b128 Xq (tokens), B = ds_read_b128 Wq (rows), MFMA, cvt + mul + mul + add per value
// MFMA operand convention used: A lane l -> (row i = l & 31, 16 consecutive k bytes of half l >> 5), B likewise for
// column j; any k permutation is harmless as long as A and B agree. C: col = l & 31, row = (r&3) + 8*(r>>2) + 4*(l>>5).
// Row stride 144 B (128 + 16 pad) makes the 16-lane groups of ds_read_b128 conflict-free.
#include "fq_block_dev.h"
#include "kernels.h"
#include <type_traits>

// GQ_PACKED=1: the scaling epilogue with v_pk_mul_f32 / v_pk_add_f32, two results per instruction (356 instead of 640 f32 VALU
// instructions in <Q4_0,2,4,2>, the same bits). Measured on MI355X (Falcon-7B Q4_0, A/B of two builds in one gpurun call):
// 128-token prompt 9.90 ms against 9.72 scalar, 2048 tokens 145.5 ms against 140.2 -- the packed f32 operations issue at half
// rate next to the MFMAs, so the instruction count they save buys nothing. OFF.
#ifndef GQ_PACKED
#define GQ_PACKED 0
#endif
// GQ_FMA (= FQ_SPLIT_FMA of fq_types.h, shared with kernels_gemm_skinny.hip; round 4: the default order of the legacy formats' K-split sums):
// acc = fma(dw * dx, (float) c, acc) -- FMA form, oracle-pinned (the reference's AVX2 builds also fuse, ggml.c:2415-2438, but on 8 per-lane integer
// sums, so this is not their bits either) -- 3 instead of 4 VALU operations per result; measured 3-5 % of a prompt. The sequential sum
// (S == 1, ggml_hip_gemm_sequential / reference order) keeps the scalar build's two roundings per term and its bit-identity with the reference.
// -DFQ_SPLIT_FMA=0 at compile time restores the unfused K-split sums in BOTH files (the oracle then needs ORC_SPLIT_FMA=0 too).
#define GQ_FMA FQ_SPLIT_FMA
#ifndef GQ_OUTER
#define GQ_OUTER 0        // 1: the scales' products d_w * d_x of the FMA form on the f32 matrix pipe instead of 16 v_mul_f32 per group (see compute()). Round 5, bit-identical
                          // (tests/test_gpu_mul_mat.py: 332 passed with it) and SLOWER, A/B/A/B of two builds in one gpurun call: 2048-token prompt 110.8 against 99.3 ms,
                          // 128 tokens 9.0 against 8.7 -- a third fewer vector instructions per result do not help: the 64-cycle f32 matrix instruction sits in front of
                          // the group's 16 fused multiply-adds, and a wave's group is a dependent chain (LDS -> matrix pipe -> scaling), not a VALU-issue budget. OFF.
#endif
#ifndef GQ_MAGIC
#define GQ_MAGIC 0        // 1: the matrix instruction's int32 sums come out as the BITS of the float 1.5 * 2^23 + c (its C operand is the constant 0x4B400000; |c| <= 32 * 127 * 127 < 2^22, so
                          // the sum cannot leave the binade whose ulp is 1) and (float) c is one exact v_sub_f32 instead of v_cvt_f32_i32. Same value, bit for bit. Why: on MI355X
                          // (scripts/microbench/mb_mfma_i8_mix.hip) 16 v_cvt_f32_i32 cost a SIMD 28.7 ns against 16.9 for 16 v_mul_f32 / v_sub_f32, and the packed forms run at half rate
                          // (8 v_pk_fma_f32: 17.7 ns for the 16 results 16 v_fma_f32 take 20.5 for) -- the conversion is the dearest of the three operations per result.
#endif
#if GQ_MAGIC
#define GQ_C0 0x4B400000
#define GQ_CF(v) (__int_as_float(v) - 12582912.0f)
#else
#define GQ_C0 0
#define GQ_CF(v) ((float)(v))
#endif
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef int v4i  __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

#define GQ_TM      32           // weight rows per workgroup
#define GQ_STRIDE  144          // LDS bytes per staged row (128 payload + 16 pad)
#define GQ_GROUPS  4            // 32-element groups per K stage

__device__ __forceinline__ uint32_t bytes_sub(uint32_t x, uint32_t c4) { return ((x | 0x80808080u) - c4) ^ 0x80808080u; }   // per-byte x - c, x in [0,127], c <= 64

// one 32-element group of a weight row: load() only issues the global loads (so that they can fly during the previous
// stage's MFMAs), finish() turns them into 32 int8 (lo = elements 0..15, hi = 16..31), an f32 scale and an f32 min term
struct gemm_raw { fq_u4 a, b, c, d; uint32_t s0, s1, s2, s3; };
// SUB = scale sub-groups per 32-element group: 1, or 2 for the formats whose sub-blocks hold 16 elements (Q2_K, Q3_K, Q6_K:
// the group is then TWO MFMAs, each with the other half's weight bytes zeroed, and two scalings)
template <int TYPE> struct gemm_group;

template <> struct gemm_group<FQ_Q4_0> {
    static constexpr int SUB = 1;            // ggml.c:1509-1527
    static constexpr bool HAS_MIN = false;
    __device__ static void load(const fq_wrow & r, int g, gemm_raw & w) { w.a = ld_w4(fq_at<FQ_Q4_0, 0>(r, g)); w.s0 = ld_u16(fq_at<FQ_Q4_0, 1>(r, g)); }
    __device__ static void finish(const gemm_raw & w, int, v4i & lo, v4i & hi, float & sc, float & mn) {
        const uint32_t v[4] = { w.a.x, w.a.y, w.a.z, w.a.w };
#pragma unroll
        for (int i = 0; i < 4; ++i) { lo[i] = (int) bytes_sub(v[i] & 0x0F0F0F0Fu, 0x08080808u); hi[i] = (int) bytes_sub((v[i] >> 4) & 0x0F0F0F0Fu, 0x08080808u); }
        sc = fq_h2f((uint16_t) w.s0); mn = 0.0f;
    }
};
template <> struct gemm_group<FQ_Q4_1> {
    static constexpr int SUB = 1;            // ggml.c:1529-1548
    static constexpr bool HAS_MIN = true;
    __device__ static void load(const fq_wrow & r, int g, gemm_raw & w) { w.a = ld_w4(fq_at<FQ_Q4_1, 0>(r, g)); w.s0 = ld_u32(fq_at<FQ_Q4_1, 1>(r, g)); }
    __device__ static void finish(const gemm_raw & w, int, v4i & lo, v4i & hi, float & sc, float & mn) {
        const uint32_t v[4] = { w.a.x, w.a.y, w.a.z, w.a.w };
#pragma unroll
        for (int i = 0; i < 4; ++i) { lo[i] = (int)(v[i] & 0x0F0F0F0Fu); hi[i] = (int)((v[i] >> 4) & 0x0F0F0F0Fu); }
        sc = fq_h2f((uint16_t) w.s0); mn = fq_h2f((uint16_t)(w.s0 >> 16));
    }
};
template <> struct gemm_group<FQ_Q5_0> {
    static constexpr int SUB = 1;            // ggml.c:1550-1574
    static constexpr bool HAS_MIN = false;
    __device__ static void load(const fq_wrow & r, int g, gemm_raw & w) {
        w.a = ld_w4(fq_at<FQ_Q5_0, 0>(r, g)); w.s0 = ld_u32(fq_at<FQ_Q5_0, 1>(r, g)); w.s1 = ld_u16(fq_at<FQ_Q5_0, 2>(r, g));
    }
    __device__ static void finish(const gemm_raw & w, int, v4i & lo, v4i & hi, float & sc, float & mn) {
        const uint32_t qh = w.s0;
        const uint32_t v[4] = { w.a.x, w.a.y, w.a.z, w.a.w };
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            lo[i] = (int) bytes_sub((v[i] & 0x0F0F0F0Fu) | (spread4(qh >> (4 * i)) << 4), 0x10101010u);
            hi[i] = (int) bytes_sub(((v[i] >> 4) & 0x0F0F0F0Fu) | (spread4(qh >> (16 + 4 * i)) << 4), 0x10101010u);
        }
        sc = fq_h2f((uint16_t) w.s1); mn = 0.0f;
    }
};
template <> struct gemm_group<FQ_Q5_1> {
    static constexpr int SUB = 1;            // ggml.c:1576-1601
    static constexpr bool HAS_MIN = true;
    __device__ static void load(const fq_wrow & r, int g, gemm_raw & w) {
        w.a = ld_w4(fq_at<FQ_Q5_1, 0>(r, g)); w.s0 = ld_u32(fq_at<FQ_Q5_1, 1>(r, g)); w.s1 = ld_u32(fq_at<FQ_Q5_1, 2>(r, g));
    }
    __device__ static void finish(const gemm_raw & w, int, v4i & lo, v4i & hi, float & sc, float & mn) {
        const uint32_t qh = w.s0;
        const uint32_t v[4] = { w.a.x, w.a.y, w.a.z, w.a.w };
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            lo[i] = (int)((v[i] & 0x0F0F0F0Fu) | (spread4(qh >> (4 * i)) << 4));
            hi[i] = (int)(((v[i] >> 4) & 0x0F0F0F0Fu) | (spread4(qh >> (16 + 4 * i)) << 4));
        }
        sc = fq_h2f((uint16_t) w.s1); mn = fq_h2f((uint16_t)(w.s1 >> 16));
    }
};
template <> struct gemm_group<FQ_Q8_0> {
    static constexpr int SUB = 1;            // ggml.c:1603-1619
    static constexpr bool HAS_MIN = false;
    __device__ static void load(const fq_wrow & r, int g, gemm_raw & w) {
        const uint8_t * q = fq_at<FQ_Q8_0, 0>(r, g); w.a = ld_w4(q); w.b = ld_w4(q + 16); w.s0 = ld_u16(fq_at<FQ_Q8_0, 1>(r, g));
    }
    __device__ static void finish(const gemm_raw & w, int, v4i & lo, v4i & hi, float & sc, float & mn) {
        lo = v4i{ (int) w.a.x, (int) w.a.y, (int) w.a.z, (int) w.a.w }; hi = v4i{ (int) w.b.x, (int) w.b.y, (int) w.b.z, (int) w.b.w };
        sc = fq_h2f((uint16_t) w.s0); mn = 0.0f;
    }
};
// Q4_K / Q5_K: group g = sub-block j = g % 8 of super-block g / 8 (k_quants.c:607-631, 734-760); w = (d*sc)*q - dmin*m
template <int TYPE> struct gemm_group_k45 {
    static constexpr int SUB = 1;
    static constexpr bool HAS_MIN = true;
    __device__ static void load(const fq_wrow & r, int g, gemm_raw & w) {
        const int64_t sb = g >> 3; const int j = g & 7, c = j >> 1;
        const uint8_t * q = fq_at<TYPE, 0>(r, sb) + 32 * c;
        constexpr int PSC = (TYPE == FQ_Q4_K) ? 1 : 2;                  // plane of the 12 packed scale bytes; d/dmin follows
        if constexpr (TYPE == FQ_Q5_K) {
            // only the 16 bytes of each half that carry this sub-block's 5th bits are kept: bit j of every byte
            const uint8_t * qh = fq_at<TYPE, 1>(r, sb);
            const fq_u4 qa = ld_w4(qh), qb = ld_w4(qh + 16);
            const fq_u4 a = ld_w4(q), b = ld_w4(q + 16);
            const uint32_t xa[4] = { qa.x, qa.y, qa.z, qa.w }, xb[4] = { qb.x, qb.y, qb.z, qb.w };
            uint32_t ha[4], hb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { ha[i] = ((xa[i] >> j) & 0x01010101u) << 4; hb[i] = ((xb[i] >> j) & 0x01010101u) << 4; }
            const int sh = (j & 1) ? 4 : 0;
            w.a = fq_u4{ ((a.x >> sh) & 0x0F0F0F0Fu) | ha[0], ((a.y >> sh) & 0x0F0F0F0Fu) | ha[1], ((a.z >> sh) & 0x0F0F0F0Fu) | ha[2], ((a.w >> sh) & 0x0F0F0F0Fu) | ha[3] };
            w.b = fq_u4{ ((b.x >> sh) & 0x0F0F0F0Fu) | hb[0], ((b.y >> sh) & 0x0F0F0F0Fu) | hb[1], ((b.z >> sh) & 0x0F0F0F0Fu) | hb[2], ((b.w >> sh) & 0x0F0F0F0Fu) | hb[3] };
        } else {
            w.a = ld_w4(q); w.b = ld_w4(q + 16);
        }
        const uint8_t * scp = fq_at<TYPE, PSC>(r, sb);
        w.s0 = ld_u32(scp); w.s1 = ld_u32(scp + 4); w.s2 = ld_u32(scp + 8);
        w.s3 = ld_u32(fq_at<TYPE, PSC + 1>(r, sb));
    }
    __device__ static void split(const gemm_raw & w, int g, int (&isc)[2], int (&imn)[2], float & d, float & dmin) {
        int s6, m6; k4_scale_min(w.s0, w.s1, w.s2, g & 7, s6, m6);
        isc[0] = s6; imn[0] = m6; isc[1] = 0; imn[1] = 0;
        d = fq_h2f((uint16_t) w.s3); dmin = fq_h2f((uint16_t)(w.s3 >> 16));
    }
    __device__ static void finish(const gemm_raw & w, int g, v4i & lo, v4i & hi, float & sc, float & mn) {
        const int j = g & 7;
        if constexpr (TYPE == FQ_Q5_K) {
            lo = v4i{ (int) w.a.x, (int) w.a.y, (int) w.a.z, (int) w.a.w }; hi = v4i{ (int) w.b.x, (int) w.b.y, (int) w.b.z, (int) w.b.w };
        } else {
            const int sh = (j & 1) ? 4 : 0;
            lo = v4i{ (int)((w.a.x >> sh) & 0x0F0F0F0Fu), (int)((w.a.y >> sh) & 0x0F0F0F0Fu), (int)((w.a.z >> sh) & 0x0F0F0F0Fu), (int)((w.a.w >> sh) & 0x0F0F0F0Fu) };
            hi = v4i{ (int)((w.b.x >> sh) & 0x0F0F0F0Fu), (int)((w.b.y >> sh) & 0x0F0F0F0Fu), (int)((w.b.z >> sh) & 0x0F0F0F0Fu), (int)((w.b.w >> sh) & 0x0F0F0F0Fu) };
        }
        int s6, m6; k4_scale_min(w.s0, w.s1, w.s2, j, s6, m6);
        sc = fq_h2f((uint16_t) w.s3) * (float) s6;
        mn = -(fq_h2f((uint16_t)(w.s3 >> 16)) * (float) m6);
    }
};
template <> struct gemm_group<FQ_Q4_K> : gemm_group_k45<FQ_Q4_K> {};
// ---- k-quants in the INTEGER domain (gemm_kint): a super-block's 8 groups share one fp16 d (and dmin) per weight row and
// one f32 d per token, the sub-block scales / mins are small integers, so  sum_j d sc_j I_j = d (sum_j sc_j I_j)  with
// the sum exact in int32 -- what the reference's own k-quant dots do (k_quants.c:1267-1306 ...). The f32 work drops from
// 5-6 instructions per (row, token, group) to 1-2 integer multiply-adds plus one f32 flush per super-block.
//   split(w, g, isc, imn, d, dmin): the group's integer scale(s) / min(s) and the row's super-block floats
template <int TYPE> struct gemm_kint { static constexpr bool value = false; };
template <> struct gemm_group<FQ_Q5_K> : gemm_group_k45<FQ_Q5_K> {};

// ---- formats with 16-element sub-blocks. Group g of a row = elements [32 g, 32 g + 32) = sub-blocks 2g, 2g+1 of the row.
static __device__ __forceinline__ v4i u4_to_v4i(const fq_u4 & x) { return v4i{ (int) x.x, (int) x.y, (int) x.z, (int) x.w }; }
template <> struct gemm_group<FQ_Q2_K> {            // k_quants.c:344-375: w = d (sc & 15) q - dmin (sc >> 4), q = 2 bits
    static constexpr int SUB = 2;
    static constexpr bool HAS_MIN = true;
    __device__ static void load(const fq_wrow & r, int g, gemm_raw & w) {
        const int64_t sb = g >> 3; const int hf = (g >> 2) & 1, j = g & 3;
        const uint8_t * q = fq_at<FQ_Q2_K, 0>(r, sb) + 32 * hf;
        w.a = ld_w4(q); w.b = ld_w4(q + 16);
        w.s0 = ld_u16(fq_at<FQ_Q2_K, 1>(r, sb) + 8 * hf + 2 * j);      // scales[8 hf + 2 j], [.. + 1]
        w.s1 = ld_u32(fq_at<FQ_Q2_K, 2>(r, sb));
    }
    __device__ static void split(const gemm_raw & w, int, int (&isc)[2], int (&imn)[2], float & d, float & dmin) {
#pragma unroll
        for (int s = 0; s < 2; ++s) { const uint32_t b = (w.s0 >> (8 * s)) & 0xFFu; isc[s] = (int)(b & 0xFu); imn[s] = (int)(b >> 4); }
        d = fq_h2f((uint16_t) w.s1); dmin = fq_h2f((uint16_t)(w.s1 >> 16));
    }
    __device__ static void finish(const gemm_raw & w, int g, v4i & lo, v4i & hi, float (&sc)[2], float (&mn)[2]) {
        const int sh = 2 * (g & 3);
        lo = v4i{ (int)((w.a.x >> sh) & 0x03030303u), (int)((w.a.y >> sh) & 0x03030303u), (int)((w.a.z >> sh) & 0x03030303u), (int)((w.a.w >> sh) & 0x03030303u) };
        hi = v4i{ (int)((w.b.x >> sh) & 0x03030303u), (int)((w.b.y >> sh) & 0x03030303u), (int)((w.b.z >> sh) & 0x03030303u), (int)((w.b.w >> sh) & 0x03030303u) };
        const float d = fq_h2f((uint16_t) w.s1), dmin = fq_h2f((uint16_t)(w.s1 >> 16));
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const uint32_t b = (w.s0 >> (8 * s)) & 0xFFu;
            sc[s] = d * (float)(int)(b & 0xFu); mn[s] = -(dmin * (float)(int)(b >> 4));
        }
    }
};
template <> struct gemm_group<FQ_Q3_K> {            // k_quants.c:472-521: w = d (sc - 32) (q - (h ? 0 : 4))
    static constexpr int SUB = 2;
    static constexpr bool HAS_MIN = false;
    __device__ static void load(const fq_wrow & r, int g, gemm_raw & w) {
        const int64_t sb = g >> 3; const int hf = (g >> 2) & 1;
        const uint8_t * q = fq_at<FQ_Q3_K, 0>(r, sb) + 32 * hf, * hm = fq_at<FQ_Q3_K, 1>(r, sb), * scp = fq_at<FQ_Q3_K, 2>(r, sb);
        w.a = ld_w4(q); w.b = ld_w4(q + 16);
        w.c = ld_w4(hm); w.d = ld_w4(hm + 16);                          // hmask bytes 0..15 / 16..31
        w.s0 = ld_u32(scp); w.s1 = ld_u32(scp + 4); w.s2 = ld_u32(scp + 8);
        w.s3 = ld_u16(fq_at<FQ_Q3_K, 3>(r, sb));
    }
    __device__ static void split(const gemm_raw & w, int g, int (&isc)[2], int (&imn)[2], float & d, float & dmin) {
        const int hf = (g >> 2) & 1, j = g & 3;
#pragma unroll
        for (int s = 0; s < 2; ++s) { isc[s] = q3_scale(w.s0, w.s1, w.s2, 8 * hf + 2 * j + s) - 32; imn[s] = 0; }
        d = fq_h2f((uint16_t) w.s3); dmin = 0.0f;
    }
    __device__ static void finish(const gemm_raw & w, int g, v4i & lo, v4i & hi, float (&sc)[2], float (&mn)[2]) {
        const int hf = (g >> 2) & 1, j = g & 3, sh = 2 * j, hb = 4 * hf + j;
        auto val = [&](uint32_t q, uint32_t h) {                        // per byte: 2 bits | (high bit << 2), minus 4
            return (int) bytes_sub(((q >> sh) & 0x03030303u) | (((h >> hb) & 0x01010101u) << 2), 0x04040404u);
        };
        lo = v4i{ val(w.a.x, w.c.x), val(w.a.y, w.c.y), val(w.a.z, w.c.z), val(w.a.w, w.c.w) };
        hi = v4i{ val(w.b.x, w.d.x), val(w.b.y, w.d.y), val(w.b.z, w.d.z), val(w.b.w, w.d.w) };
        const float d = fq_h2f((uint16_t) w.s3);
#pragma unroll
        for (int s = 0; s < 2; ++s) { sc[s] = d * (float)(q3_scale(w.s0, w.s1, w.s2, 8 * hf + 2 * j + s) - 32); mn[s] = 0.0f; }
    }
};
template <> struct gemm_group<FQ_Q6_K> {            // k_quants.c:845-876: w = d scale (q - 32), q = 4 low | 2 high bits
    static constexpr int SUB = 2;
    static constexpr bool HAS_MIN = false;
    __device__ static void load(const fq_wrow & r, int g, gemm_raw & w) {
        const int64_t sb = g >> 3; const int h = (g >> 2) & 1, t = g & 3;
        const uint8_t * ql = fq_at<FQ_Q6_K, 0>(r, sb) + 64 * h + 32 * (t & 1), * qh = fq_at<FQ_Q6_K, 1>(r, sb) + 32 * h;
        w.a = ld_w4(ql); w.b = ld_w4(ql + 16);
        w.c = ld_w4(qh); w.d = ld_w4(qh + 16);
        w.s0 = ld_u16(fq_at<FQ_Q6_K, 2>(r, sb) + 8 * h + 2 * t);       // int8 scales[8 h + 2 t], [.. + 1]
        w.s1 = ld_u16(fq_at<FQ_Q6_K, 3>(r, sb));
    }
    __device__ static void split(const gemm_raw & w, int, int (&isc)[2], int (&imn)[2], float & d, float & dmin) {
#pragma unroll
        for (int s = 0; s < 2; ++s) { isc[s] = (int)(int8_t)(w.s0 >> (8 * s)); imn[s] = 0; }
        d = fq_h2f((uint16_t) w.s1); dmin = 0.0f;
    }
    __device__ static void finish(const gemm_raw & w, int g, v4i & lo, v4i & hi, float (&sc)[2], float (&mn)[2]) {
        const int t = g & 3, nsh = (t >> 1) ? 4 : 0, hsh = 2 * t;
        auto val = [&](uint32_t l, uint32_t h) {
            return (int) bytes_sub(((l >> nsh) & 0x0F0F0F0Fu) | (((h >> hsh) & 0x03030303u) << 4), 0x20202020u);
        };
        lo = v4i{ val(w.a.x, w.c.x), val(w.a.y, w.c.y), val(w.a.z, w.c.z), val(w.a.w, w.c.w) };
        hi = v4i{ val(w.b.x, w.d.x), val(w.b.y, w.d.y), val(w.b.z, w.d.z), val(w.b.w, w.d.w) };
        const float d = fq_h2f((uint16_t) w.s1);
#pragma unroll
        for (int s = 0; s < 2; ++s) { sc[s] = d * (float)(int)(int8_t)(w.s0 >> (8 * s)); mn[s] = 0.0f; }
    }
};

template <> struct gemm_kint<FQ_Q2_K> { static constexpr bool value = true; };
template <> struct gemm_kint<FQ_Q3_K> { static constexpr bool value = true; };
template <> struct gemm_kint<FQ_Q4_K> { static constexpr bool value = true; };
template <> struct gemm_kint<FQ_Q5_K> { static constexpr bool value = true; };
template <> struct gemm_kint<FQ_Q6_K> { static constexpr bool value = true; };

// tuning aid (ggml_hip_debug_gemm_mode): bit 1 = no MFMA / scaling (timing only: what the staging alone costs). Bit 0 (no
// global loads) went with the role-specialised pipeline, whose loads are unconditional on purpose.
__device__ int g_gemm_dbg = 0;
#ifndef GQ_STAMPS
#define GQ_STAMPS 0      // 1 (a tuning build, never the product's; scripts/gpu_gemm_stamps.py): 10 ns clock stamps of the K pipeline's phases per wave, stages 8..23 of workgroups 0 and 100
#endif                   // of a launch -> ggml_hip_debug_stamps' buffer. A stamp costs ~200 ns (s_memrealtime + a store): the launch runs at half speed, the proportions hold
#if GQ_STAMPS
#include "hip_context.h"
__device__ long long * g_gemm_stamps = nullptr;
#if GQ_STAMPS == 2       // 2: only the clock of the K loop (shader cycles against 10 ns ticks, scripts/gpu_gemm_clock.py), no per-stage stamps
#define GQ_T(slot) do { } while (0)
#else
#define GQ_T(slot) do { if (g_gemm_stamps && lane == 0 && blockIdx.y == 0 && (blockIdx.x == 0 || blockIdx.x == 100) && st >= 8 && st < 24) \
        g_gemm_stamps[((((blockIdx.x ? 1 : 0) * 16 + wid) * 8 + ((st - 8) >> 1)) * 8) + (slot)] = (long long) wall_clock64(); } while (0)
#endif
#else
#define GQ_T(slot) do { } while (0)
#endif
static int g_gemm_dbg_host = 0;
void fq_gemm_debug_mode(int m) { g_gemm_dbg_host = m; HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_dbg), &m, sizeof m)); }
int  fq_gemm_debug_get() { return g_gemm_dbg_host; }

// LDS buffer of one K stage
template <bool HAS_MIN, int TN, int SUB, int TM> struct gemm_lds {   // TN = tokens, TM = weight rows per workgroup, SUB = scale sub-groups per group
    static constexpr int XQ = 0, WQ = TN * GQ_STRIDE, DX = WQ + TM * GQ_STRIDE, SX = DX + GQ_GROUPS * TN * 4,
                         DW = SX + (HAS_MIN ? GQ_GROUPS * SUB * TN * 4 : 0), MW = DW + GQ_GROUPS * SUB * TM * 4,
                         DS = MW + (HAS_MIN ? GQ_GROUPS * SUB * TM * 4 : 0),         // k-quants: d [TM rows], dmin [TM rows]
                         BYTES = DS + 2 * TM * 4;
};

// S = waves per 32 x 32 tile (1, 2, 4), TT = 32-token tiles per workgroup (4 or 1): workgroup = TT S waves.
// The S waves of a tile split K: wave sw takes the groups g = sw (mod S) of every stage, in order, into its own partial
// sum; the partial sums are added at the end as ((P0 + P1) + P2) + P3. S = 1 is the reference's scalar order (one
// left-to-right sum over the blocks of a row); S > 1 trades that for S times the parallelism on shapes with few tiles
// (a 128-token prompt on a 4544-row matrix has 568 tiles for 1024 SIMDs) -- a fixed, documented association, the same
// kind the reference's own 8-lane AVX2 loop applies (oracle: orc_set_sum_order).
// RB = 32-row blocks per workgroup (1 or 2): with 2, a wave runs its token tile against both row blocks of every group it
// owns -- the token bytes a workgroup pulls through the CU's memory pipeline (16 KB per stage, against 2.3 KB of Q4_0
// weights per row block) are then used twice. A CU keeps ~48 KB of requests in flight whatever the kernel does, so at large
// N the 32-row form is bound by exactly that: the kernel with its math compiled out runs at 76 % of the full kernel's time.
#ifndef GQ_PAIR
#define GQ_PAIR 1
#endif
// two K stages per barrier (four LDS stage buffers instead of two): the 16-wave forms, whose workgroup has the CU to itself anyway
template <int S, int TT, int RB> struct gemm_pair { static constexpr bool value = GQ_PAIR && S * TT >= 16; };
// SEQ16 (round 6; legacy formats, the <4, 4, 1> frame): the reference's ONE left-to-right sum per row (ggml_hip_gemm_sequential, ggml.c:2591-2609 ...) at the 16-wave form's
// occupancy. A row's sum is a chain over all of K, so S = 1 leaves a 32 x 32 result tile to ONE wave -- 142 (568) dependent group steps of ~70 vector instructions each, on four
// waves per workgroup: 67 us for Wqkv at 128 tokens where the four-share form takes 32. Here the sixteen waves of the 32-row x 128-token workgroup own one 16 x 16 tile each
// (row half = wave & 1, token tile = wave >> 1; v_mfma_i32_16x16x32_i8, four results per lane) and walk EVERY group of a stage in order: the same terms, the same order, the
// same bits as S = 1 -- a quarter of the chain per wave, four waves per SIMD to fill it. Staging, stage buffers and barriers are the frame's own.
template <int TYPE, int S, int TT, int RB, bool SEQ16 = false>
__global__ void __launch_bounds__(64 * S * TT) k_gemm_q(fq_weight w, fq_act act, int64_t N, float * dst, int64_t ldd, fq_gemv_epi ep) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr int ACT = fq_act_of(TYPE);
    constexpr bool HAS_MIN = gemm_group<TYPE>::HAS_MIN, KINT = gemm_kint<TYPE>::value;
    constexpr int TN = 32 * TT, SUB = gemm_group<TYPE>::SUB, TM = GQ_TM * RB;
    static_assert(!SEQ16 || (S == 4 && TT == 4 && RB == 1 && !KINT && SUB == 1), "SEQ16: the legacy formats in the <4, 4, 1> frame");
    typedef gemm_lds<HAS_MIN, TN, SUB, TM> LB;
    constexpr int NT = 64 * S * TT, VT = (TN * 8) / NT, NR = 16;         // threads, token vectors per thread and stage, results per lane
    static_assert(NT >= TM * GQ_GROUPS && VT >= 1, "workgroup too small for the staging tasks");
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int tt = wid % TT, sw = wid / TT;                               // token tile, K share (groups sw, sw + S, ... of a stage)
#if GQ_STAMPS
    const unsigned long long clk0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    const int64_t m0 = (int64_t) blockIdx.x * TM, n0 = (int64_t) blockIdx.y * TN;
    const int64_t K = w.K, M = w.M;
    const int ngroups = (int)(K >> 5);
    const size_t img = fq_act_col_bytes(ACT, K);

    // ---- what a thread stages per K stage: weights = (row, group) tasks of threads < 128 (waves 0-1); tokens = 16-byte
    // vectors of all threads; token scales = the 4 groups' d (and s / bsums) of token tid - 128, the following wave(s)
    const int w_row = (tid >> 2) & (TM - 1), w_gg = tid & 3;
    const fq_wrow wrow = fq_row<TYPE>(w, m0 + w_row < M ? m0 + w_row : M - 1);
    // (the weight tasks sit in waves 0-1, the token-scale tasks in waves 2..: a stage's critical path is the longest
    // per-wave instruction stream up to the barrier, so the two staging roles must not land in the same wave)
    static_assert(NT >= 128 * RB + TN || (RB == 1 && NT >= 256), "staging roles need separate waves");
    const int sc_tok = (tid - 128 * RB) & (TN - 1);
    // The pipeline below exists once per staging ROLE of a wave (1 = weights: waves 0-1; 2 = token scales: the next
    // ceil(TN / 64) waves; 0 = none), selected by a wave-uniform branch: inside one copy every load is unconditional, so the
    // compiler can count them (s_waitcnt vmcnt(N) that leaves the NEXT stages' loads in flight). With the roles as
    // per-thread predicates in one copy it must assume the fewest loads and waits for the loads it has just issued.
    constexpr int SC_WAVES = (TN + 63) / 64;
    const uint8_t * sc_col = act.base + (size_t)(n0 + sc_tok < N ? n0 + sc_tok : N - 1) * img;

    struct stage_regs { gemm_raw w; v4i x[VT]; float4 d4; float2 sa, sb; uint2 ba, bb; };
    auto issue = [&](int g0, stage_regs & R, auto role) __attribute__((always_inline)) {
        constexpr int ROLE = decltype(role)::value;
        if constexpr (ROLE == 1) { const int g = g0 + w_gg; gemm_group<TYPE>::load(wrow, g < ngroups ? g : ngroups - 1, R.w); }
#pragma unroll
        for (int i = 0; i < VT; ++i) {
            const int t = tid + i * NT, tok = t >> 3, part = t & 7;
            const int64_t n = n0 + tok < N ? n0 + tok : N - 1;
            const int64_t kb = (int64_t) g0 * 32 + 16 * part;
            R.x[i] = *(const v4i *)(act.base + (size_t) n * img + (size_t)(kb < K ? kb : 0));
        }
        if constexpr (ROLE == 2) {                                         // (lanes beyond TN of the role's last wave re-read a valid token)
            const int gc = g0 + 3 < ngroups ? g0 : (ngroups >= 4 ? ngroups - 4 : 0);     // (K % 128 != 0: the tail stage re-reads valid groups)
            if constexpr (ACT == FQ_Q8_K) {
                R.d4.x = ((const float *)(sc_col + fq_act_d_off(ACT, K)))[g0 >> 3];
                const uint2 * bs = (const uint2 *)(sc_col + fq_act_aux_off(ACT, K) + 4 * (size_t) gc);       // 8 x int16
                R.ba = bs[0]; R.bb = bs[1];
            } else {
                R.d4 = *(const float4 *)(sc_col + fq_act_d_off(ACT, K) + 4 * (size_t) gc);
                if constexpr (ACT == FQ_Q8_1) {
                    const float2 * sp = (const float2 *)(sc_col + fq_act_aux_off(ACT, K) + 4 * (size_t) gc);
                    R.sa = sp[0]; R.sb = sp[1];
                }
            }
        }
    };
    auto commit = [&](int g0, const stage_regs & R, uint8_t * B, auto role) __attribute__((always_inline)) {   // registers -> LDS buffer B (zeros beyond K / N / M)
        constexpr int ROLE = decltype(role)::value;
        if constexpr (ROLE == 1) {
            const int g = g0 + w_gg;
            v4i lo = {0, 0, 0, 0}, hi = {0, 0, 0, 0}; float sc[2] = {0.0f, 0.0f}, mn[2] = {0.0f, 0.0f};
            if (g < ngroups && m0 + w_row < M) {
                if constexpr (SUB == 1) gemm_group<TYPE>::finish(R.w, g, lo, hi, sc[0], mn[0]);
                else                    gemm_group<TYPE>::finish(R.w, g, lo, hi, sc, mn);
            }
            *(v4i *)(B + LB::WQ + w_row * GQ_STRIDE + 32 * w_gg)      = lo;
            *(v4i *)(B + LB::WQ + w_row * GQ_STRIDE + 32 * w_gg + 16) = hi;
            if constexpr (KINT) {
                int isc[2] = {0, 0}, imn[2] = {0, 0}; float dd = 0.0f, dm = 0.0f;
                if (g < ngroups && m0 + w_row < M) gemm_group<TYPE>::split(R.w, g, isc, imn, dd, dm);
#pragma unroll
                for (int ss = 0; ss < SUB; ++ss) {
                    ((int *)(B + LB::DW))[(w_gg * SUB + ss) * TM + w_row] = isc[ss];
                }
                if constexpr (HAS_MIN) {        // the stage's 4 SUB mins of a row: k slots 0..7 of an int8 MFMA operand (below)
                    if constexpr (SUB == 2) ((uint16_t *)(B + LB::MW))[w_row * 4 + w_gg] = (uint16_t)(imn[0] | (imn[1] << 8));
                    else { (B + LB::MW)[w_row * 8 + w_gg] = (uint8_t) imn[0]; (B + LB::MW)[w_row * 8 + 4 + w_gg] = 0; }
                }
                if (w_gg == 0) { ((float *)(B + LB::DS))[w_row] = dd; ((float *)(B + LB::DS))[TM + w_row] = dm; }
            } else {
#pragma unroll
                for (int ss = 0; ss < SUB; ++ss) {
                    ((float *)(B + LB::DW))[(w_gg * SUB + ss) * TM + w_row] = sc[ss];
                    if constexpr (HAS_MIN) ((float *)(B + LB::MW))[(w_gg * SUB + ss) * TM + w_row] = mn[ss];
                }
            }
        }
#pragma unroll
        for (int i = 0; i < VT; ++i) {
            const int t = tid + i * NT, tok = t >> 3, part = t & 7;
            const bool ok = n0 + tok < N && (int64_t) g0 * 32 + 16 * part < K;
            *(v4i *)(B + LB::XQ + tok * GQ_STRIDE + 16 * part) = ok ? R.x[i] : v4i{0, 0, 0, 0};
        }
        if constexpr (ROLE == 2) {                                         // (aliased lanes write the same values again)
            float dx[4], sx[4] = {0.0f, 0.0f, 0.0f, 0.0f}, sx1[4] = {0.0f, 0.0f, 0.0f, 0.0f};     // sx1: second sub-group (SUB == 2)
            const bool full = g0 + 3 < ngroups;
            if constexpr (ACT == FQ_Q8_K) {
                const uint32_t bw[4] = { R.ba.x, R.ba.y, R.bb.x, R.bb.y };      // bsums of the group's two 16-element halves
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    dx[i] = R.d4.x;
                    if constexpr (KINT) continue;
                    else if constexpr (SUB == 1) sx[i] = R.d4.x * (float)((int)(int16_t) bw[i] + (int)(int16_t)(bw[i] >> 16));
                    else { sx[i] = R.d4.x * (float)(int)(int16_t) bw[i]; sx1[i] = R.d4.x * (float)(int)(int16_t)(bw[i] >> 16); }
                }
            } else {
                dx[0] = R.d4.x; dx[1] = R.d4.y; dx[2] = R.d4.z; dx[3] = R.d4.w;
                if constexpr (ACT == FQ_Q8_1) { sx[0] = R.sa.x; sx[1] = R.sa.y; sx[2] = R.sb.x; sx[3] = R.sb.y; }
            }
            if (full) {                                                     // (wave-uniform; every stage but a row's last: the four groups are where they belong -- no selects)
                const bool okt = n0 + sc_tok < N;
#pragma unroll
                for (int gg = 0; gg < 4; ++gg) {
                    ((float *)(B + LB::DX))[gg * TN + sc_tok] = okt ? dx[gg] : 0.0f;
                    if constexpr (HAS_MIN && !KINT) {
                        ((float *)(B + LB::SX))[(gg * SUB) * TN + sc_tok] = okt ? sx[gg] : 0.0f;
                        if constexpr (SUB == 2) ((float *)(B + LB::SX))[(gg * SUB + 1) * TN + sc_tok] = okt ? sx1[gg] : 0.0f;
                    }
                }
            } else {
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) {
                // a tail stage (fewer than 4 groups left) re-read the LAST four groups: shift them back into place
                const int src = gg + (g0 - (ngroups >= 4 ? ngroups - 4 : 0));
                const bool ok = n0 + sc_tok < N && g0 + gg < ngroups && src < 4;
                float dv = 0.0f, sv = 0.0f, sv1 = 0.0f;
#pragma unroll
                for (int q = 0; q < 4; ++q) if (q == src) { dv = dx[q]; sv = sx[q]; sv1 = sx1[q]; }
                ((float *)(B + LB::DX))[gg * TN + sc_tok] = ok ? dv : 0.0f;
                if constexpr (HAS_MIN && !KINT) {
                    ((float *)(B + LB::SX))[(gg * SUB) * TN + sc_tok] = ok ? sv : 0.0f;
                    if constexpr (SUB == 2) ((float *)(B + LB::SX))[(gg * SUB + 1) * TN + sc_tok] = ok ? sv1 : 0.0f;
                }
            }
            }
            if constexpr (KINT && HAS_MIN) {
                // the stage's 4 SUB block sums of a token as int8 MFMA operand bytes: bsum = 64 hi + lo, hi in [-64, 63]
                // (|bsum| <= 32 * 127), lo in [0, 63]; k slot = gg SUB + ss, hi bytes in the low 8 bytes, lo in the high 8
                const uint32_t bw[4] = { R.ba.x, R.ba.y, R.bb.x, R.bb.y };
                uint32_t H[2] = {0, 0}, Lo[2] = {0, 0};
                if (n0 + sc_tok < N) {
#pragma unroll
                    for (int gg = 0; gg < 4; ++gg) {
                        const int b0 = (int)(int16_t) bw[gg], b1 = (int)(int16_t)(bw[gg] >> 16);
                        if constexpr (SUB == 1) {
                            const int v = b0 + b1;
                            H[0] |= (uint32_t)((v >> 6) & 0xFF) << (8 * gg); Lo[0] |= (uint32_t)(v & 63) << (8 * gg);
                        } else {
                            const int sl = 2 * gg;
                            H[sl >> 2]  |= ((uint32_t)((b0 >> 6) & 0xFF) | ((uint32_t)((b1 >> 6) & 0xFF) << 8)) << (8 * (sl & 3));
                            Lo[sl >> 2] |= ((uint32_t)(b0 & 63) | ((uint32_t)(b1 & 63) << 8)) << (8 * (sl & 3));
                        }
                    }
                }
                *(v4i *)(B + LB::SX + sc_tok * 16) = v4i{ (int) H[0], (int) H[1], (int) Lo[0], (int) Lo[1] };
            }
        }
    };

    v2f acc2[RB][NR / 2];                                                // the f32 results in pairs (packed scaling epilogue)
#define ACC(rb_, i_) acc2[rb_][(i_) >> 1][(i_) & 1]
    int iacc[RB][KINT ? NR : 1];                                         // k-quants: integer sums of the current super-block
    v16i chi[RB], clo[RB];                                               // ... and of its mins term (wave sw == S - 1)
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
#pragma unroll
        for (int r = 0; r < NR; ++r) { ACC(rb, r) = 0.0f; chi[rb][r] = 0; clo[rb][r] = 0; }
#pragma unroll
        for (int r = 0; r < (KINT ? NR : 1); ++r) iacc[rb][r] = 0;
    }
    const int half = lane >> 5, l31 = lane & 31;
    constexpr int rot = 0;
    const int arow = l31;
    // the matrix instruction's C operand of the legacy formats' groups (GQ_MAGIC): sixteen registers that hold the constant for the whole kernel -- opaque to the compiler, which
    // otherwise lets the second instruction of a pair accumulate in place and re-creates the constant with eight 64-bit moves per pair
    v16i cmagic = { GQ_C0, GQ_C0, GQ_C0, GQ_C0, GQ_C0, GQ_C0, GQ_C0, GQ_C0, GQ_C0, GQ_C0, GQ_C0, GQ_C0, GQ_C0, GQ_C0, GQ_C0, GQ_C0 };
    if constexpr (GQ_MAGIC && !KINT) asm volatile("" : "+v"(cmagic));

    const int dbgm = g_gemm_dbg;
    // ---- software pipeline, prefetch distance TWO stages (a stage's math, ~0.7 us, is shorter than a load round trip):
    // while stage s is computed out of LDS buffer s & 1, stage s+1 sits in one register set and stage s+2 is in flight
    // into the other
    // k-quants: one group of the integer-domain sums. FIRST = the wave's first group of a super-block: its products START the sums (v_mul_i32_i24) instead of being added to
    // registers that the flush had to zero with 16 moves per row block
    auto kint_group = [&](const uint8_t * B, int gg, auto first_tag) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(first_tag)::value;
        const v4i a = *(const v4i *)(B + LB::XQ + (32 * tt + arow) * GQ_STRIDE + 32 * gg + 16 * half);
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const int row = 32 * rb + l31;
            const v4i b = *(const v4i *)(B + LB::WQ + row * GQ_STRIDE + 32 * gg + 16 * half);
#pragma unroll
            for (int ss = 0; ss < SUB; ++ss) {
                const v4i bm = (SUB == 1 || half == ss) ? b : v4i{0, 0, 0, 0};
                const int isc = ((const int *)(B + LB::DW))[(gg * SUB + ss) * TM + row];
                v16i c = {0};
                c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bm, c, 0, 0, 0);
#pragma unroll
                for (int i = 0; i < NR; ++i) {                             // |c| < 2^17, |isc| <= 128
                    if (FIRST && ss == 0) iacc[rb][i] = __mul24(c[i], isc);
                    else                  iacc[rb][i] = __mul24(c[i], isc) + iacc[rb][i];
                }
            }
        }
    };
    float acc16[4] = { 0.0f, 0.0f, 0.0f, 0.0f };                            // SEQ16: token 16 t16 + 4 kq + i, row 16 rh + l15
    const int l15 = lane & 15, kq = lane >> 4, rh = wid & 1, t16 = wid >> 1;
    auto compute = [&](const uint8_t * B, const bool sb_end) __attribute__((always_inline)) {
        if constexpr (SEQ16) {
            typedef int v2i_ __attribute__((ext_vector_type(2)));
            const uint8_t * xa = B + LB::XQ + (16 * t16 + l15) * GQ_STRIDE + 8 * kq;
            const uint8_t * wb = B + LB::WQ + (16 * rh + l15) * GQ_STRIDE + 8 * kq;
            const int tok = 16 * t16 + 4 * kq, row = 16 * rh + l15;
            // operands of PRE groups at a time (the formats with a min term hold twice the scales: two groups keep the 16-wave workgroup within its 128 registers)
            constexpr int PRE = HAS_MIN ? 1 : GQ_GROUPS;
#pragma unroll
            for (int g0 = 0; g0 < GQ_GROUPS; g0 += PRE) {
                v2i_ a[PRE], b[PRE]; float4 dx[PRE], sx[PRE]; float dw[PRE], mw[PRE];
#pragma unroll
                for (int q = 0; q < PRE; ++q) {
                    const int gg = g0 + q;
                    a[q] = *(const v2i_ *)(xa + 32 * gg); b[q] = *(const v2i_ *)(wb + 32 * gg);
                    dx[q] = *(const float4 *)(B + LB::DX + (gg * TN + tok) * 4);
                    dw[q] = ((const float *)(B + LB::DW))[gg * TM + row];
                    if constexpr (HAS_MIN) { sx[q] = *(const float4 *)(B + LB::SX + (gg * TN + tok) * 4); mw[q] = ((const float *)(B + LB::MW))[gg * TM + row]; }
                }
                if (dbgm & 2) continue;
#pragma unroll
                for (int q = 0; q < PRE; ++q) {
                    const v4i c = __builtin_amdgcn_mfma_i32_16x16x32_i8(__builtin_bit_cast(long, a[q]), __builtin_bit_cast(long, b[q]), v4i{0, 0, 0, 0}, 0, 0, 0);
                    const float dxv[4] = { dx[q].x, dx[q].y, dx[q].z, dx[q].w };
                    const float sxv[4] = { HAS_MIN ? sx[q].x : 0.0f, HAS_MIN ? sx[q].y : 0.0f, HAS_MIN ? sx[q].z : 0.0f, HAS_MIN ? sx[q].w : 0.0f };
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float ci = (float) c[i];
                        float t;
                        if constexpr (TYPE == FQ_Q4_0)  t = (ci * dw[q]) * dxv[i];                              // ggml.c:2606
                        else if constexpr (!HAS_MIN)    t = (dw[q] * dxv[i]) * ci;                              // ggml.c:2972, 3325
                        else                            t = (dw[q] * dxv[i]) * ci + mw[q] * sxv[i];            // ggml.c:2731, 3227
                        acc16[i] = acc16[i] + t;
                    }
                }
            }
            (void) sb_end;
            return;
        }
        if constexpr (KINT) {
            if (!(dbgm & 2)) {
                int gg = sw;
                if (!sb_end) { kint_group(B, gg, std::true_type{}); gg += S; }      // (a super-block = an even stage, then an odd one)
#pragma unroll 1
                for (; gg < GQ_GROUPS; gg += S) kint_group(B, gg, std::false_type{});
            }
        }
#pragma unroll 1                          // (unrolled, the four groups' operands are all hoisted: 212 VGPRs, one wave per SIMD)
        for (int gg = sw; gg < ((dbgm & 2) || KINT ? 0 : GQ_GROUPS); gg += S) {
            const v4i a = *(const v4i *)(B + LB::XQ + (32 * tt + arow) * GQ_STRIDE + 32 * gg + 16 * half);
            // the lane's result i  <->  token 32 tt + rot + (i & 3) + 8 (i >> 2) + 4 half: runs of 4 consecutive tokens
            float dxv[NR];
#pragma unroll
            for (int q = 0; q < NR / 4; ++q) {
                const int tok = 32 * tt + rot + 8 * q + 4 * half;
                const float4 t = *(const float4 *)(B + LB::DX + (gg * TN + tok) * 4);
                dxv[4 * q] = t.x; dxv[4 * q + 1] = t.y; dxv[4 * q + 2] = t.z; dxv[4 * q + 3] = t.w;
            }
#pragma unroll
            for (int ss = 0; ss < SUB; ++ss) {
                float sxv[NR];
                if constexpr (HAS_MIN) {
#pragma unroll
                    for (int q = 0; q < NR / 4; ++q) {
                        const int tok = 32 * tt + rot + 8 * q + 4 * half;
                        const float4 u = *(const float4 *)(B + LB::SX + ((gg * SUB + ss) * TN + tok) * 4);
                        sxv[4 * q] = u.x; sxv[4 * q + 1] = u.y; sxv[4 * q + 2] = u.z; sxv[4 * q + 3] = u.w;
                    }
                }
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) {
                    const int row = 32 * rb + l31;
                    const v4i b = *(const v4i *)(B + LB::WQ + row * GQ_STRIDE + 32 * gg + 16 * half);
                    // 16-element sub-blocks: lanes of half h hold the group's weight bytes [16 h, 16 h + 16); zeroing the other
                    // half's operand leaves exactly sub-block ss in the MFMA's sum
                    const v4i bm = (SUB == 1 || half == ss) ? b : v4i{0, 0, 0, 0};
                    const float dw = ((const float *)(B + LB::DW))[(gg * SUB + ss) * TM + row];
                    const float mw = HAS_MIN ? ((const float *)(B + LB::MW))[(gg * SUB + ss) * TM + row] : 0.0f;
                    const v16i c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bm, cmagic, 0, 0, 0);
                    // f32 epilogue: the reference's scalar per-block expression, added left to right over the groups, so
                    // that for the legacy formats a row of this GEMM is bit-identical to ggml_vec_dot_q*_q8_* (scalar branch)
#if GQ_PACKED
                    // the same IEEE operations two results at a time (v_pk_mul_f32 / v_pk_add_f32; no contraction: -ffp-contract=off)
                    const v2f dw2 = { dw, dw }, mw2 = { mw, mw };
#pragma unroll
                    for (int i = 0; i < NR; i += 2) {
                        const v2f ci = { GQ_CF(c[i]), GQ_CF(c[i + 1]) };
                        const v2f dx = { dxv[i], dxv[i + 1] };
                        v2f t;
                        if constexpr (TYPE == FQ_Q4_0)                          t = (ci * dw2) * dx;                       // ggml.c:2606
                        else if constexpr (!HAS_MIN)                            t = (dw2 * dx) * ci;                       // ggml.c:2972, 3325; Q3_K, Q6_K
                        else { const v2f sx = { sxv[i], sxv[i + 1] };           t = (dw2 * dx) * ci + mw2 * sx; }          // ggml.c:2731, 3227; k-quants
                        acc2[rb][i >> 1] = acc2[rb][i >> 1] + t;
                    }
#else
                    if constexpr (GQ_FMA && S > 1 && fq_desc(TYPE).blck == 32) {
                        // the K-split partial sums of the legacy formats (the DEFAULT order) accumulate as the reference's AVX2 build does
                        // (acc = _mm256_fmadd_ps(d, q, acc), ggml.c:2415-2438): 3 instead of 4 vector operations per result; the oracle's split orders
                        // restate it with fmaf. S == 1 (ggml_hip_gemm_sequential: the reference ORDER) keeps the scalar build's two roundings per term
#if GQ_OUTER
                        // Round 5: the 32 x 32 products d_w[row] * d_x[token] of the group are a rank-1 matrix, and v_mfma_f32_32x32x2_f32 forms it on the
                        // (idle) matrix pipe: A = the tokens' scales in k slot 0 (lanes 0-31), B = the rows' scales, C = 0 -- one fused multiply-add per
                        // k step (scripts/microbench/mb_mfma_f32.hip), fma(d_x, d_w, +0) then + 0 * 0: the single rounding of the v_mul_f32 it replaces
                        // (a zero product's sign cannot reach the sum: acc is never -0). 2 instead of 3 vector operations per result, 16 fewer LDS values.
                        const float dxl = half == 0 ? ((const float *)(B + LB::DX))[gg * TN + 32 * tt + rot + l31] : 0.0f;
                        const v16f dd = __builtin_amdgcn_mfma_f32_32x32x2f32(dxl, half == 0 ? dw : 0.0f, v16f{0}, 0, 0, 0);
#pragma unroll
                        for (int i = 0; i < NR; ++i) {
                            const float ci = GQ_CF(c[i]);
                            float a = __builtin_fmaf(dd[i], ci, ACC(rb, i));
                            if constexpr (HAS_MIN) a = __builtin_fmaf(mw, sxv[i], a);
                            ACC(rb, i) = a;
                        }
#else
#pragma unroll
                        for (int i = 0; i < NR; ++i) {
                            const float ci = GQ_CF(c[i]);
                            float a = __builtin_fmaf(dw * dxv[i], ci, ACC(rb, i));
                            if constexpr (HAS_MIN) a = __builtin_fmaf(mw, sxv[i], a);
                            ACC(rb, i) = a;
                        }
#endif
                    } else {
#pragma unroll
                    for (int i = 0; i < NR; ++i) {
                        const float ci = GQ_CF(c[i]);
                        float t;
                        if constexpr (TYPE == FQ_Q4_0)                          t = (ci * dw) * dxv[i];                    // ggml.c:2606
                        else if constexpr (TYPE == FQ_Q5_0 || TYPE == FQ_Q8_0)  t = (dw * dxv[i]) * ci;                    // ggml.c:2972, 3325
                        else if constexpr (!HAS_MIN)                            t = (dw * dxv[i]) * ci;                    // Q3_K, Q6_K
                        else                                                    t = (dw * dxv[i]) * ci + mw * sxv[i];      // ggml.c:2731, 3227; k-quants
                        ACC(rb, i) = ACC(rb, i) + t;
                    }
                    }
#endif
                }
            }
        }
        if constexpr (KINT && HAS_MIN) {
            // the mins term  sum_j m_j bsum_j  of the stage's 4 SUB sub-blocks for a whole 32 x 32 tile: a [tokens x j] by
            // [j x rows] int8 product, so two MFMAs (hi / lo bytes of the block sums) instead of 4 SUB x 16 multiply-adds;
            // one wave per tile does it (the last K share: never one of the 32-row form's staging waves)
            if (sw == S - 1 && !(dbgm & 2)) {
                v4i am = {0, 0, 0, 0};
                if (half == 0) am = *(const v4i *)(B + LB::SX + (32 * tt + arow) * 16);
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) {
                    int2 bn = {0, 0};
                    if (half == 0) bn = *(const int2 *)(B + LB::MW + (32 * rb + l31) * 8);
                    const v4i bmn = { bn.x, bn.y, 0, 0 };
                    // (a super-block = an even stage, then an odd one: the even stage STARTS the sums -- C = 0 as an inline operand -- instead of adding to registers that the
                    // flush zeroed with 32 moves per row block)
                    const v16i z = {0};
                    chi[rb] = __builtin_amdgcn_mfma_i32_32x32x32_i8(v4i{ am[0], am[1], 0, 0 }, bmn, sb_end ? chi[rb] : z, 0, 0, 0);
                    clo[rb] = __builtin_amdgcn_mfma_i32_32x32x32_i8(v4i{ am[2], am[3], 0, 0 }, bmn, sb_end ? clo[rb] : z, 0, 0, 0);
                }
            }
        }
        if constexpr (KINT) {
            // end of a super-block (every second stage): one f32 step per result, the reference's own k-quant expression
            // sumf += (d * y.d) * sum_j sc_j I_j  -  (dmin * y.d) * sum_j m_j bsum_j   (k_quants.c:1565-1583 ...)
            if (sb_end) {
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) {
                    const int row = 32 * rb + l31;
                    const float dd = ((const float *)(B + LB::DS))[row], dm = HAS_MIN ? ((const float *)(B + LB::DS))[TM + row] : 0.0f;
#pragma unroll
                    for (int q = 0; q < NR / 4; ++q) {
                        const int tok = 32 * tt + rot + 8 * q + 4 * half;
                        const float4 t = *(const float4 *)(B + LB::DX + tok * 4);      // y.d of the super-block (all four groups alike)
                        const float dxq[4] = { t.x, t.y, t.z, t.w };
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int i = 4 * q + e;
                            float t = (dd * dxq[e]) * (float) iacc[rb][i];       // (iacc is not zeroed: the next super-block's first group overwrites it, kint_group)
                            if constexpr (HAS_MIN) {       // (the oracle's / reference's expression: sumf += d isum - dmin msum)
                                if (sw == S - 1) t = t - (dm * dxq[e]) * (float)((chi[rb][i] << 6) + clo[rb][i]);
                            }
                            ACC(rb, i) = ACC(rb, i) + t;
                        }
                    }
                }
            }
        }
    };
    const int nstages = (ngroups + GQ_GROUPS - 1) / GQ_GROUPS;
    const int last_g0 = (nstages - 1) * GQ_GROUPS;
    auto g0_of = [&](int st) { return st < nstages ? st * GQ_GROUPS : last_g0; };      // beyond the end: re-read the last stage (unused)
    auto pipeline = [&](auto role) __attribute__((always_inline)) {
        stage_regs R0, R1;
        if constexpr (gemm_pair<S, TT, RB>::value) {
            // TWO stages per barrier (round 5; four LDS buffers): phase stamps of the 16-wave form at 128 tokens (scripts/gpu_gemm_stamps.py, a -DGQ_STAMPS build) showed a
            // stage as 350 ns of arithmetic, 105 ns of LDS writes, 65 ns of load issue -- and 330 ns at the barrier, ALL sixteen waves (the workgroup is the CU's only one: the
            // SIMDs idle). Same stages, same order of the sums; the registers still hold the next two stages.
            uint8_t * const b0 = smem, * const b1 = smem + LB::BYTES, * const b2 = smem + 2 * LB::BYTES, * const b3 = smem + 3 * LB::BYTES;
            issue(0, R0, role);
            issue(g0_of(1), R1, role);
            commit(0, R0, b0, role);
            issue(g0_of(2), R0, role);
            commit(g0_of(1), R1, b1, role);
            issue(g0_of(3), R1, role);
            __syncthreads();
            for (int st = 0; st < nstages; st += 4) {
                // stages st, st + 1 out of buffers 0, 1; R0 / R1 hold st + 2 / st + 3 and go into buffers 2, 3
                compute(b0, false);
                commit(g0_of(st + 2), R0, b2, role);
                issue(g0_of(st + 4), R0, role);
                if (st + 1 < nstages) compute(b1, true);
                commit(g0_of(st + 3), R1, b3, role);
                issue(g0_of(st + 5), R1, role);
                __syncthreads();
                if (st + 2 < nstages) compute(b2, false);
                commit(g0_of(st + 4), R0, b0, role);
                issue(g0_of(st + 6), R0, role);
                if (st + 3 < nstages) compute(b3, true);
                commit(g0_of(st + 5), R1, b1, role);
                issue(g0_of(st + 7), R1, role);
                __syncthreads();
            }
            return;
        }
        issue(0, R0, role);
        issue(g0_of(1), R1, role);
        commit(0, R0, smem, role);
        __syncthreads();
        for (int st = 0; st < nstages; st += 2) {
            // even stage st out of buffer 0; R1 holds st + 1; st + 2 goes into R0
            GQ_T(0);
            issue(g0_of(st + 2), R0, role);
            GQ_T(1);
            compute(smem, false);
            GQ_T(2);
            commit(g0_of(st + 1), R1, smem + LB::BYTES, role);
            GQ_T(3);
            __syncthreads();
            // odd stage st + 1 out of buffer 1; R0 holds st + 2; st + 3 goes into R1
            GQ_T(4);
            issue(g0_of(st + 3), R1, role);
            GQ_T(5);
            if (st + 1 < nstages) compute(smem + LB::BYTES, true);
            GQ_T(6);
            commit(g0_of(st + 2), R0, smem, role);
            GQ_T(7);
            __syncthreads();
        }
    };
    if (wid < 2 * RB)                 pipeline(std::integral_constant<int, 1>{});
    else if (wid < 2 * RB + SC_WAVES) pipeline(std::integral_constant<int, 2>{});
    else                         pipeline(std::integral_constant<int, 0>{});
#if GQ_STAMPS
    if (g_gemm_stamps && tid == 0 && blockIdx.y == 0 && (blockIdx.x == 0 || blockIdx.x == 50)) {      // shader cycles and 10 ns ticks of the K loop: the clock this CU ran at
        g_gemm_stamps[8000 + 2 * (blockIdx.x ? 1 : 0)] = (long long)(__builtin_amdgcn_s_memtime() - clk0);
        g_gemm_stamps[8001 + 2 * (blockIdx.x ? 1 : 0)] = (long long)(__builtin_amdgcn_s_memrealtime() - rt0);
    }
#endif
    if constexpr (SEQ16) {
        const int64_t m = m0 + 16 * rh + l15;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t n = n0 + 16 * t16 + 4 * kq + i;
            if (n < N && m < M) {
                float v = acc16[i];
                if (ep.mode == FQ_EPI_GELU)      v = h2f_bits(ep.gelu_table[f2h_bits(v)]);
                else if (ep.mode == FQ_EPI_ADD2) v = (v + ep.add1[n * ep.ld_add + m]) + ep.add2[n * ep.ld_add + m];
                dst[n * ldd + m] = v;
            }
        }
        return;
    }
    // ---- the S partial sums of a tile: ((P0 + P1) + P2) + P3, through LDS (the stage buffers are free now)
    if constexpr (S > 1) {
        float * xch = (float *) smem;                                      // [tt][rb][i][lane]
        static_assert(TT * RB * NR * 64 * 4 <= 2 * LB::BYTES, "partial sums do not fit the stage buffers");
        for (int r = 1; r < S; ++r) {
            __syncthreads();
            if (sw == r) {
#pragma unroll
                for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                    for (int i = 0; i < NR; ++i) xch[((tt * RB + rb) * NR + i) * 64 + lane] = ACC(rb, i);
            }
            __syncthreads();
            if (sw == 0) {
#pragma unroll
                for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                    for (int i = 0; i < NR; ++i) ACC(rb, i) = ACC(rb, i) + xch[((tt * RB + rb) * NR + i) * 64 + lane];
            }
        }
        if (sw != 0) return;
    }
    // ---- epilogue: token n = n0 + 32*tt + (i&3) + 8*(i>>2) + 4*half, row m = m0 + 32*rb + (lane&31)
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        const int64_t m = m0 + 32 * rb + l31;
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int64_t n = n0 + 32 * tt + rot + (i & 3) + 8 * (i >> 2) + 4 * half;
            if (n < N && m < M) {
                float v = ACC(rb, i);
                if (ep.mode == FQ_EPI_GELU)      v = h2f_bits(ep.gelu_table[f2h_bits(v)]);
                else if (ep.mode == FQ_EPI_ADD2) v = (v + ep.add1[n * ep.ld_add + m]) + ep.add2[n * ep.ld_add + m];
                dst[n * ldd + m] = v;
            }
        }
    }
}

// 1: every shape through S = 1, i.e. the reference's left-to-right sum over a row's blocks (ggml_hip_gemm_sequential)
static int g_gemm_sequential = 0;
void fq_gemm_set_sequential(int on) { g_gemm_sequential = on != 0; }

bool fq_gemm_supported(int type) {
    return type == FQ_Q4_0 || type == FQ_Q4_1 || type == FQ_Q5_0 || type == FQ_Q5_1 || type == FQ_Q8_0 ||
           type == FQ_Q2_K || type == FQ_Q3_K || type == FQ_Q4_K || type == FQ_Q5_K || type == FQ_Q6_K;
}

template <int TYPE, int S, int TT, int RB = 1, bool SEQ16 = false>
static void launch_gemm_t(const fq_weight & w, const fq_act & act, int64_t N, float * dst, int64_t ldd, const fq_gemv_epi & ep, hipStream_t st) {
    constexpr bool HAS_MIN = gemm_group<TYPE>::HAS_MIN;
    constexpr int TN = 32 * TT, TM = GQ_TM * RB;
    const size_t lds = (gemm_pair<S, TT, RB>::value ? 4 : 2) * (size_t) gemm_lds<HAS_MIN, TN, gemm_group<TYPE>::SUB, TM>::BYTES;
    if (lds > 64 * 1024) {
        static bool set = false;
        if (!set) { HIP_CHECK(hipFuncSetAttribute((const void *) k_gemm_q<TYPE, S, TT, RB, SEQ16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds)); set = true; }
    }
    const dim3 grid((unsigned)((w.M + TM - 1) / TM), (unsigned)((N + TN - 1) / TN));
#if GQ_STAMPS
    { long long * p = fq_ctx().dbg_stamps; HIP_CHECK(hipMemcpyToSymbolAsync(HIP_SYMBOL(g_gemm_stamps), &p, sizeof p, 0, hipMemcpyHostToDevice, st)); }
#endif
    hipLaunchKernelGGL((k_gemm_q<TYPE, S, TT, RB, SEQ16>), grid, dim3(64 * S * TT), lds, st, w, act, N, dst, ldd, ep);
}

template <int TYPE>
static void launch_gemm_seq16(const fq_weight & w, const fq_act & act, int64_t N, float * dst, int64_t ldd, const fq_gemv_epi & ep, hipStream_t st) {
    if constexpr (fq_desc(TYPE).blck == 32) launch_gemm_t<TYPE, 4, 4, 1, true>(w, act, N, dst, ldd, ep, st);
    else launch_gemm_t<TYPE, 1, 4>(w, act, N, dst, ldd, ep, st);
}

// the 64-row workgroups' threshold (x #CU tiles of 32 x 32; 0 = never) by prompt length
static int64_t gemm_rb_min(int64_t N, int type, int64_t M, int n_cu) {
    static const int64_t rb_env = getenv("FQ_GEMM_RB_TILES") ? atoll(getenv("FQ_GEMM_RB_TILES")) : -1;
    static const int64_t rb_long = getenv("FQ_GEMM_RB_TILES_LONG") ? atoll(getenv("FQ_GEMM_RB_TILES_LONG")) : 8;
    static const int64_t rb_mid = getenv("FQ_GEMM_RB_TILES_MID") ? atoll(getenv("FQ_GEMM_RB_TILES_MID")) : 4;
    // (256 tokens included: a prompt of 256 does not move, 12.0 ms, a lock-step pass of 256 sequences does, 14.76 -> 13.18 ms, profiles/r06zzf_lockstep_sweep.txt)
    // (the 256..511 rule for the legacy formats only: the k-quants' integer-domain GEMM loses with 64-row workgroups there -- Falcon-40B Q2_K 288-384 tokens 29-30 -> 32.5-33 ms,
    // Q4_K 23.0-23.8 -> 23.8-24.4; profiles/r06zzd_ab_rb_mid.txt)
    // (Q4_K / Q5_K below 512 tokens: 8 x #CU except at three token tile rows -- Falcon-40B shapes, 12 blocks, threshold 32 -> 8: Q4_K 128 tokens 8.91 -> 8.42 ms, 256: 16.24 -> 14.92,
    // 448: 31.06 -> 28.65, lock-step 128 / 256 per pass 9.34 -> 9.00 / 18.14 -> 17.03, but 320-384 tokens 23.4-24.0 -> 23.9-24.4; Q5_K alike; Q2_K loses up to 11 % with it and keeps 32
    // (profiles/r06zzi_kq_rb.txt, r06zzj_kq_rb8.txt))
    if (rb_env >= 0) return rb_env;
    if (N >= 512) return rb_long;
    if (type == FQ_Q4_K || type == FQ_Q5_K) return (N + 127) / 128 == 3 ? 32 : 8;
    // (one token tile row, legacy formats: only a matrix whose 64-row workgroups still fill the chip twice over -- Falcon-40B's Wup: 512 of them; 128-token prompt 8.65 -> 8.23 ms on
    // Falcon-40B Q5_1 shapes, profiles/r06zzl_*, r06zzm_*; Falcon-7B's 284 lose: 6.8 -> 7.27 ms; a full tile row only: at 40-96 tokens the same matrix loses 2-5 % as a prompt
    // and wins 12 % as a lock-step pass of 64 -- left alone)
    static const bool big_up = !(getenv("FQ_GEMM_RB_BIG_M") && atoi(getenv("FQ_GEMM_RB_BIG_M")) == 0);
    if (big_up && N > 96 && N <= 128 && fq_desc(type).blck == 32 && (M + 63) / 64 >= 2 * (int64_t) n_cu) return 1;
    return N >= 256 && fq_desc(type).blck == 32 ? rb_mid : 32;
}
// weight rows per workgroup fq_launch_gemm's tile form gives an M-row matrix at N columns (32 or 64): the callers that reason about a launch's workgroup count
int fq_gemm_wg_rows(int type, int64_t M, int64_t N, int n_cu) {
    if (g_gemm_sequential) return 32;
    const int64_t tiles = ((M + GQ_TM - 1) / GQ_TM) * ((N + 31) / 32);
    const int64_t rb = gemm_rb_min(N, type, M, n_cu);
    return (tiles >= 4 * (int64_t) n_cu && rb > 0 && tiles >= rb * (int64_t) n_cu) ? 64 : 32;
}
int fq_gemm_split_for(int64_t M, int64_t N, int n_cu) {
    if (g_gemm_sequential) return 1;
    const int64_t tiles = ((M + GQ_TM - 1) / GQ_TM) * ((N + 31) / 32);
    return tiles < 4 * (int64_t) n_cu ? 4 : 2;
}

// dst[n*ldd + m], n < N; act holds N quantized columns
void fq_launch_gemm(const fq_weight & w, const fq_act & act, int64_t N, float * dst, int64_t ldd, const fq_gemv_epi & ep, int n_cu, hipStream_t st) {
    FQ_TL(st, "gemm");
    // The unit of parallelism is a 32-token x 32-row tile (568 of them for a 4544-row matrix and 128 tokens), split S ways
    // over K. Measured on MI355X, Falcon-7B shapes, Q4_0 (scripts/gpu_gemm_prof.sh), us for qkv / wo / up / down:
    //   128 tokens   <S 1, 4 tiles>  44 / 44 /  92 / 163    <S 2, 4 tiles>  34 / 34 /  82 / 119    <S 4, 4 tiles>  31 / 31 /  95 / 106
    //   512 tokens   <S 1, 4 tiles>  94 / 93 / 265 / 340    <S 2, 4 tiles>  83 / 83 / 242 / 296    <S 4, 4 tiles>  94 / 94 / 290 / 331
    //   2048 tokens  <S 1, 4 tiles> 287 / 266 / 1043 / 1038 <S 2, 4 tiles> 259 / 238 / 969 / 929   <S 4, 4 tiles> 316 / 289 / 1157 / 1061
    // -> four partial sums per row below 4 x #CU tiles, two above; one (the reference's order) only on request
    const int64_t tiles = ((fq_form_rows(w) + GQ_TM - 1) / GQ_TM) * ((N + 31) / 32);            // (a row-split part: the whole matrix's tiles, fq_types.h)
    int cfg = g_gemm_sequential ? 0 : (tiles < 4 * (int64_t) n_cu ? 2 : 3);
    // small batches: the streaming form (kernels_gemm_skinny.hip), same K split -> same bits (FQ_GEMM_SKINNY=0: this kernel for every N)
    static const bool skinny = !(getenv("FQ_GEMM_SKINNY") && atoi(getenv("FQ_GEMM_SKINNY")) == 0);
    if (skinny && N <= 16 && !getenv("FQ_GEMM_CFG") && fq_launch_gemm_skinny(w, act, N, dst, ldd, ep, cfg == 0 ? 1 : (cfg == 2 ? 4 : 2), st)) return;
    // 17..32 columns: two passes of the streaming form (same K split as this shape's tile GEMM, so the same bits; 7.5 against 9.1 ms
    // per Falcon-7B pass of 32 lock-step sequences) -- beyond that the tiles below win. Q4_K at model widths: up to five passes (Falcon-40B, 12 blocks:
    // 48 columns 6.0 against 10.4 ms, 64: 7.9 against 10.7, 80: 9.8 against ~11; 96: 11.7 against 11.3)
    static const bool skinny2 = !(getenv("FQ_GEMM_SKINNY2") && atoi(getenv("FQ_GEMM_SKINNY2")) == 0);
    const int64_t max_cols = fq_skinny_q4k_shape(w) ? fq_skinny_kq_max_cols(w.type) : 32;      // (Q2_K, 12 blocks of Falcon-40B: 80 columns 7.6 against ~11.6 ms, 128: 12.2 against 11.8)
    if (skinny && skinny2 && N > 16 && N <= max_cols && !getenv("FQ_GEMM_CFG")) {
        const int S = cfg == 0 ? 1 : (cfg == 2 ? 4 : 2);
        for (int64_t n0 = 0; n0 < N; n0 += 16) {
            const int64_t nc = N - n0 < 16 ? N - n0 : 16;
            fq_act a1 = act; a1.ncols = nc; a1.base = act.base + n0 * fq_act_col_bytes(act.type, act.K);
            fq_gemv_epi e1 = ep;
            if (e1.add1) e1.add1 += n0 * e1.ld_add;
            if (e1.add2) e1.add2 += n0 * e1.ld_add;
            if (!fq_launch_gemm_skinny(w, a1, nc, dst + n0 * ldd, ldd, e1, S, st)) {
                if (n0 == 0) break;                                         // not this form's shape: the tiles below
                fprintf(stderr, "ggml-hip: gemm: a later small-batch pass refused\n"); exit(1);
            }
            if (n0 + 16 >= N) return;
        }
    }
    // 64-row workgroups (RB = 2) where there are tiles enough to fill the chip with them: two partial sums as <2,4> above,
    // or four (<4,4,2>: 16 waves) for the formats whose 64-row kernel stays within 128 VGPRs
    // (round 6: from 512 tokens on already at 8 x #CU tiles -- the short matrices of a long prompt (Wqkv, Wo, Wdown: 4 672 tiles at 1024 tokens) had fallen between the two
    // rules: Falcon-7B Q4_0 1024 tokens 40.7 -> 36.8 ms, 512: 22.75 -> 22.27; below 512 tokens the same threshold costs Wup 6 % at 256 and 7 % at 128. The same S: the same bits.
    // FQ_GEMM_RB_TILES=n: n x #CU for every length (A/B, profiles/r06zz_ab_rb_tiles.txt); FQ_GEMM_RB_TILES_LONG=n: the threshold from 512 tokens on)
    // (and from 257 tokens on at 4 x #CU -- every matrix of a block that takes two partial sums: three token tile rows were the other hole, 288-384 tokens 16.4-17.6 -> 13.8-14.6 ms
    // together with the residual-sum launch of falcon_hip.hip, profiles/r06zzc_*; FQ_GEMM_RB_TILES_MID=n)
    { const int64_t rbm = gemm_rb_min(N, w.type, fq_form_rows(w), n_cu); if (cfg == 3 && rbm > 0 && tiles >= rbm * (int64_t) n_cu) cfg = 6; }
    // few columns (round 6): token tiles of 32 / 64 instead of 128 -- the same four-way K split (S = 4: the same association, the same bits), a quarter /
    // half of the matrix work on padding columns gone and 4 / 8 waves per workgroup instead of 16 (FQ_GEMM_SMALL_TT=0: the 128-token tiles, as before)
    static const bool small_tt = !(getenv("FQ_GEMM_SMALL_TT") && atoi(getenv("FQ_GEMM_SMALL_TT")) == 0);
    if (small_tt && cfg == 2 && N <= 32) cfg = 1; else if (small_tt && cfg == 2 && N <= 64) cfg = 4;
    // (tuning aid, FQ_GEMM_FILL=1: a launch of 128-token workgroups that leaves CUs empty -- 142 workgroups for a 4544-row matrix and 128 tokens -- as 64-token
    // workgroups, 284 of them: measured SLOWER, 128-token prompt 9.46 against 8.30-8.34 ms A/B/A/B on one box (round 6): the 16-wave workgroup hides its own latencies better
    // than two 8-wave ones fill the chip)
    static const bool fill = getenv("FQ_GEMM_FILL") && atoi(getenv("FQ_GEMM_FILL")) != 0;
    if (fill && cfg == 2 && N > 64 && ((fq_form_rows(w) + GQ_TM - 1) / GQ_TM) * ((N + 127) / 128) < n_cu) cfg = 4;
    if (const char * e = getenv("FQ_GEMM_CFG")) cfg = atoi(e);                   // tuning override: 0 = <1,4>, 1 = <4,1>, 2 = <4,4>, 3 = <2,4>, 6 = <2,4,2>, 7 = <4,4,2>
    // the sequential sum of the legacy formats: sixteen 16 x 16 tiles per workgroup instead of four 32 x 32 ones (SEQ16, k_gemm_q's header; FQ_GEMM_SEQ16=0: the S = 1 form)
    static const bool seq16 = !(getenv("FQ_GEMM_SEQ16") && atoi(getenv("FQ_GEMM_SEQ16")) == 0);
    // one tile row of tokens only: measured on MI355X, Falcon-7B Q4_0 prompts in the fast reference order, SEQ16 against S = 1 (scripts/gpu_prompt_lengths.py, A/B/A/B):
    // 40 tokens 10.3 against 12.2 ms, 64: 10.5 / 12.6, 128: 11.7 / 13.3 -- but 256: 21.4 / 18.1, 512: 39.4 / 31.9, 2048: 194 / 171-180: with several token tile rows the S = 1
    // form has the waves to fill its chains and the 16 x 16 tiles' extra operand reads cost more than they hide
    if (cfg == 0 && seq16 && fq_desc(w.type).blck == 32 && N <= 128) cfg = 16;
#define FQ_CASE(T) case T: if (cfg == 16) launch_gemm_seq16<T>(w, act, N, dst, ldd, ep, st); else if (cfg == 0) launch_gemm_t<T, 1, 4>(w, act, N, dst, ldd, ep, st); else if (cfg == 1) launch_gemm_t<T, 4, 1>(w, act, N, dst, ldd, ep, st); \
                           else if (cfg == 2) launch_gemm_t<T, 4, 4>(w, act, N, dst, ldd, ep, st); else if (cfg == 3) launch_gemm_t<T, 2, 4>(w, act, N, dst, ldd, ep, st); \
                           else if (cfg == 4) launch_gemm_t<T, 4, 2>(w, act, N, dst, ldd, ep, st); else if (cfg == 5) launch_gemm_t<T, 2, 2>(w, act, N, dst, ldd, ep, st); \
                           else if (cfg == 6) launch_gemm_t<T, 2, 4, 2>(w, act, N, dst, ldd, ep, st); else launch_gemm_t<T, 4, 4, 2>(w, act, N, dst, ldd, ep, st); break;
    switch (w.type) {
        FQ_CASE(FQ_Q4_0) FQ_CASE(FQ_Q4_1) FQ_CASE(FQ_Q5_0) FQ_CASE(FQ_Q5_1) FQ_CASE(FQ_Q8_0)
        FQ_CASE(FQ_Q2_K) FQ_CASE(FQ_Q3_K) FQ_CASE(FQ_Q4_K) FQ_CASE(FQ_Q5_K) FQ_CASE(FQ_Q6_K)
        default: fprintf(stderr, "ggml-hip: gemm: unsupported weight type %d\n", w.type); exit(1);
    }
#undef FQ_CASE
}
