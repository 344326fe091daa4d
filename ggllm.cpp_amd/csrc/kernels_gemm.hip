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
// Tiling: workgroup = 4 waves = 128 tokens x (32*J) weight rows; wave w owns tokens [32w, 32w+32) x all 32*J rows
// (J accumulator tiles of 16 VGPRs). Per K-stage of 128 (4 groups):
//   weights  : 16-byte quant groups -> registers -> sign-corrected int8 -> LDS  Wq[row][144] (+ dW, minW per group)
//   tokens   : int8 image rows                                        -> LDS  Xq[tok][144] (+ dX, sX per group)
//   per group: A = ds_read_b128 Xq (tokens), B = ds_read_b128 Wq (rows), MFMA, 16 cvt + 16 mul + 16 fma per tile
// MFMA operand convention used: A lane l -> (row i = l & 31, 16 consecutive k bytes of half l >> 5), B likewise for
// column j; any k permutation is harmless as long as A and B agree. C: col = l & 31, row = (r&3) + 8*(r>>2) + 4*(l>>5).
// Row stride 144 B (128 + 16 pad) makes the 16-lane groups of ds_read_b128 conflict-free.
#include "fq_block_dev.h"
#include "kernels.h"

typedef int v4i  __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

#define GQ_TN      128          // tokens per workgroup
#define GQ_STRIDE  144          // LDS bytes per staged row (128 payload + 16 pad)
#define GQ_GROUPS  4            // 32-element groups per K stage

__device__ __forceinline__ uint32_t bytes_sub(uint32_t x, uint32_t c4) { return ((x | 0x80808080u) - c4) ^ 0x80808080u; }   // per-byte x - c, x in [0,127], c <= 64

// one 32-element group of a weight row -> 32 int8 (lo = elements 0..15, hi = 16..31), f32 scale, f32 min term
template <int TYPE> struct gemm_group;

template <> struct gemm_group<FQ_Q4_0> {            // ggml.c:1509-1527
    static constexpr bool HAS_MIN = false;
    __device__ static void get(const fq_wrow & r, int g, v4i & lo, v4i & hi, float & sc, float & mn) {
        const fq_u4 q = ld_u4(r.p0 + 16 * (size_t) g);
        const uint32_t v[4] = { q.x, q.y, q.z, q.w };
#pragma unroll
        for (int i = 0; i < 4; ++i) { lo[i] = (int) bytes_sub(v[i] & 0x0F0F0F0Fu, 0x08080808u); hi[i] = (int) bytes_sub((v[i] >> 4) & 0x0F0F0F0Fu, 0x08080808u); }
        sc = fq_h2f(ld_u16(r.p1 + 2 * (size_t) g)); mn = 0.0f;
    }
};
template <> struct gemm_group<FQ_Q4_1> {            // ggml.c:1529-1548
    static constexpr bool HAS_MIN = true;
    __device__ static void get(const fq_wrow & r, int g, v4i & lo, v4i & hi, float & sc, float & mn) {
        const fq_u4 q = ld_u4(r.p0 + 16 * (size_t) g);
        const uint32_t v[4] = { q.x, q.y, q.z, q.w };
#pragma unroll
        for (int i = 0; i < 4; ++i) { lo[i] = (int)(v[i] & 0x0F0F0F0Fu); hi[i] = (int)((v[i] >> 4) & 0x0F0F0F0Fu); }
        const uint32_t dm = ld_u32(r.p1 + 4 * (size_t) g);
        sc = fq_h2f((uint16_t) dm); mn = fq_h2f((uint16_t)(dm >> 16));
    }
};
template <> struct gemm_group<FQ_Q5_0> {            // ggml.c:1550-1574
    static constexpr bool HAS_MIN = false;
    __device__ static void get(const fq_wrow & r, int g, v4i & lo, v4i & hi, float & sc, float & mn) {
        const fq_u4 q = ld_u4(r.p0 + 16 * (size_t) g);
        const uint32_t qh = ld_u32(r.p1 + 4 * (size_t) g);
        const uint32_t v[4] = { q.x, q.y, q.z, q.w };
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            lo[i] = (int) bytes_sub((v[i] & 0x0F0F0F0Fu) | (spread4(qh >> (4 * i)) << 4), 0x10101010u);
            hi[i] = (int) bytes_sub(((v[i] >> 4) & 0x0F0F0F0Fu) | (spread4(qh >> (16 + 4 * i)) << 4), 0x10101010u);
        }
        sc = fq_h2f(ld_u16(r.p2 + 2 * (size_t) g)); mn = 0.0f;
    }
};
template <> struct gemm_group<FQ_Q5_1> {            // ggml.c:1576-1601
    static constexpr bool HAS_MIN = true;
    __device__ static void get(const fq_wrow & r, int g, v4i & lo, v4i & hi, float & sc, float & mn) {
        const fq_u4 q = ld_u4(r.p0 + 16 * (size_t) g);
        const uint32_t qh = ld_u32(r.p1 + 4 * (size_t) g);
        const uint32_t v[4] = { q.x, q.y, q.z, q.w };
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            lo[i] = (int)((v[i] & 0x0F0F0F0Fu) | (spread4(qh >> (4 * i)) << 4));
            hi[i] = (int)(((v[i] >> 4) & 0x0F0F0F0Fu) | (spread4(qh >> (16 + 4 * i)) << 4));
        }
        const uint32_t dm = ld_u32(r.p2 + 4 * (size_t) g);
        sc = fq_h2f((uint16_t) dm); mn = fq_h2f((uint16_t)(dm >> 16));
    }
};
template <> struct gemm_group<FQ_Q8_0> {            // ggml.c:1603-1619
    static constexpr bool HAS_MIN = false;
    __device__ static void get(const fq_wrow & r, int g, v4i & lo, v4i & hi, float & sc, float & mn) {
        const fq_u4 a = ld_u4(r.p0 + 32 * (size_t) g), b = ld_u4(r.p0 + 32 * (size_t) g + 16);
        lo = v4i{ (int) a.x, (int) a.y, (int) a.z, (int) a.w }; hi = v4i{ (int) b.x, (int) b.y, (int) b.z, (int) b.w };
        sc = fq_h2f(ld_u16(r.p1 + 2 * (size_t) g)); mn = 0.0f;
    }
};
// Q4_K / Q5_K: group g = sub-block j = g % 8 of super-block g / 8 (k_quants.c:607-631, 734-760); w = (d*sc)*q - dmin*m
template <int TYPE> struct gemm_group_k45 {
    static constexpr bool HAS_MIN = true;
    __device__ static void get(const fq_wrow & r, int g, v4i & lo, v4i & hi, float & sc, float & mn) {
        const size_t sb = (size_t)(g >> 3); const int j = g & 7, c = j >> 1, up = j & 1;
        const fq_u4 a = ld_u4(r.p0 + 128 * sb + 32 * c), b = ld_u4(r.p0 + 128 * sb + 32 * c + 16);
        const uint32_t va[4] = { a.x, a.y, a.z, a.w }, vb[4] = { b.x, b.y, b.z, b.w };
        uint32_t ha[4] = {0, 0, 0, 0}, hb[4] = {0, 0, 0, 0};
        if constexpr (TYPE == FQ_Q5_K) {
            const fq_u4 qa = ld_u4(r.p1 + 32 * sb), qb = ld_u4(r.p1 + 32 * sb + 16);
            const uint32_t xa[4] = { qa.x, qa.y, qa.z, qa.w }, xb[4] = { qb.x, qb.y, qb.z, qb.w };
#pragma unroll
            for (int i = 0; i < 4; ++i) { ha[i] = ((xa[i] >> j) & 0x01010101u) << 4; hb[i] = ((xb[i] >> j) & 0x01010101u) << 4; }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            lo[i] = (int)(((up ? (va[i] >> 4) : va[i]) & 0x0F0F0F0Fu) | ha[i]);
            hi[i] = (int)(((up ? (vb[i] >> 4) : vb[i]) & 0x0F0F0F0Fu) | hb[i]);
        }
        const uint8_t * scp = (TYPE == FQ_Q4_K ? r.p1 : r.p2) + 12 * sb;
        int s6, m6; k4_scale_min(ld_u32(scp), ld_u32(scp + 4), ld_u32(scp + 8), j, s6, m6);
        const uint32_t dm = ld_u32((TYPE == FQ_Q4_K ? r.p2 : r.p3) + 4 * sb);
        sc = fq_h2f((uint16_t) dm) * (float) s6;
        mn = -(fq_h2f((uint16_t)(dm >> 16)) * (float) m6);
    }
};
template <> struct gemm_group<FQ_Q4_K> : gemm_group_k45<FQ_Q4_K> {};
template <> struct gemm_group<FQ_Q5_K> : gemm_group_k45<FQ_Q5_K> {};

template <int TYPE, int J>
__global__ void __launch_bounds__(256) k_gemm_q(fq_weight w, fq_act act, int64_t N, float * dst, int64_t ldd, fq_gemv_epi ep) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr int ACT = fq_act_of(TYPE);
    constexpr bool HAS_MIN = gemm_group<TYPE>::HAS_MIN;
    constexpr int TM = 32 * J;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int64_t m0 = (int64_t) blockIdx.x * TM, n0 = (int64_t) blockIdx.y * GQ_TN;
    const int64_t K = w.K, M = w.M;
    const int ngroups = (int)(K >> 5);

    uint8_t * Xq = smem;                                        // [128][144]
    uint8_t * Wq = Xq + GQ_TN * GQ_STRIDE;                      // [TM][144]
    float   * dxs = (float *)(Wq + TM * GQ_STRIDE);            // [4][128]
    float   * sxs = dxs + GQ_GROUPS * GQ_TN;                    // [4][128]   (min formats)
    float   * dws = sxs + (HAS_MIN ? GQ_GROUPS * GQ_TN : 0);    // [4][TM]
    float   * mws = dws + GQ_GROUPS * TM;                       // [4][TM]    (min formats)

    const size_t img = fq_act_col_bytes(ACT, K);
    float acc[J][16];
#pragma unroll
    for (int j = 0; j < J; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;

    for (int g0 = 0; g0 < ngroups; g0 += GQ_GROUPS) {
        __syncthreads();
        // ---- stage weights: (row, group) tasks
        for (int t = tid; t < TM * GQ_GROUPS; t += 256) {
            const int row = t >> 2, gg = t & 3, g = g0 + gg;
            v4i lo = {0, 0, 0, 0}, hi = {0, 0, 0, 0}; float sc = 0.0f, mn = 0.0f;
            if (g < ngroups && m0 + row < M) gemm_group<TYPE>::get(fq_row<TYPE>(w, m0 + row), g, lo, hi, sc, mn);
            *(v4i *)(Wq + row * GQ_STRIDE + 32 * gg)      = lo;
            *(v4i *)(Wq + row * GQ_STRIDE + 32 * gg + 16) = hi;
            dws[gg * TM + row] = sc;
            if constexpr (HAS_MIN) mws[gg * TM + row] = mn;
        }
        // ---- stage tokens: 8 x 16 B per token row, then the per-group scales
        for (int t = tid; t < GQ_TN * 8; t += 256) {
            const int tok = t >> 3, part = t & 7;
            const int64_t n = n0 + tok;
            v4i v = {0, 0, 0, 0};
            if (n < N && g0 * 32 + 16 * part < K) v = *(const v4i *)(act.base + (size_t) n * img + (size_t) g0 * 32 + 16 * part);
            *(v4i *)(Xq + tok * GQ_STRIDE + 16 * part) = v;
        }
        for (int t = tid; t < GQ_TN * GQ_GROUPS; t += 256) {
            const int tok = t & (GQ_TN - 1), gg = t >> 7, g = g0 + gg;
            const int64_t n = n0 + tok;
            float dx = 0.0f, sx = 0.0f;
            if (n < N && g < ngroups) {
                const uint8_t * col = act.base + (size_t) n * img;
                if constexpr (ACT == FQ_Q8_K) {
                    dx = ((const float *)(col + fq_act_d_off(ACT, K)))[g >> 3];
                    const int16_t * bs = (const int16_t *)(col + fq_act_aux_off(ACT, K)) + 2 * g;
                    sx = dx * (float)((int) bs[0] + (int) bs[1]);
                } else {
                    dx = ((const float *)(col + fq_act_d_off(ACT, K)))[g];
                    if constexpr (ACT == FQ_Q8_1) sx = ((const float *)(col + fq_act_aux_off(ACT, K)))[g];
                }
            }
            dxs[gg * GQ_TN + tok] = dx;
            if constexpr (HAS_MIN) sxs[gg * GQ_TN + tok] = sx;
        }
        __syncthreads();
        // ---- 4 groups x J tiles
        const int half = lane >> 5, l31 = lane & 31;
#pragma unroll
        for (int gg = 0; gg < GQ_GROUPS; ++gg) {
            const v4i a = *(const v4i *)(Xq + (32 * wid + l31) * GQ_STRIDE + 32 * gg + 16 * half);
            float dxv[16], sxv[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 t = *(const float4 *)(dxs + gg * GQ_TN + 32 * wid + 8 * q + 4 * half);
                dxv[4 * q] = t.x; dxv[4 * q + 1] = t.y; dxv[4 * q + 2] = t.z; dxv[4 * q + 3] = t.w;
                if constexpr (HAS_MIN) {
                    const float4 u = *(const float4 *)(sxs + gg * GQ_TN + 32 * wid + 8 * q + 4 * half);
                    sxv[4 * q] = u.x; sxv[4 * q + 1] = u.y; sxv[4 * q + 2] = u.z; sxv[4 * q + 3] = u.w;
                }
            }
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const v4i b = *(const v4i *)(Wq + (32 * j + l31) * GQ_STRIDE + 32 * gg + 16 * half);
                const float dw = dws[gg * TM + 32 * j + l31];
                v16i c = {0};
                c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0);
                // f32 epilogue: the reference's scalar per-block expression, added left to right over the groups, so
                // that for the legacy formats a row of this GEMM is bit-identical to ggml_vec_dot_q*_q8_* (scalar branch)
                const float mw = HAS_MIN ? mws[gg * TM + 32 * j + l31] : 0.0f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float ci = (float) c[r];
                    float t;
                    if constexpr (TYPE == FQ_Q4_0)                          t = (ci * dw) * dxv[r];                    // ggml.c:2606
                    else if constexpr (TYPE == FQ_Q5_0 || TYPE == FQ_Q8_0)  t = (dw * dxv[r]) * ci;                    // ggml.c:2972, 3325
                    else                                                    t = (dw * dxv[r]) * ci + mw * sxv[r];      // ggml.c:2731, 3227; k-quants
                    acc[j][r] = acc[j][r] + t;
                }
            }
        }
    }
    // ---- epilogue: token n = n0 + 32*wid + (r&3) + 8*(r>>2) + 4*(lane>>5), row m = m0 + 32*j + (lane&31)
    const int half = lane >> 5, l31 = lane & 31;
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int64_t m = m0 + 32 * j + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t n = n0 + 32 * wid + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (n < N && m < M) {
                float v = acc[j][r];
                if (ep.mode == FQ_EPI_GELU)      v = h2f_bits(ep.gelu_table[f2h_bits(v)]);
                else if (ep.mode == FQ_EPI_ADD2) v = (v + ep.add1[n * ep.ld_add + m]) + ep.add2[n * ep.ld_add + m];
                dst[n * ldd + m] = v;
            }
        }
    }
}

bool fq_gemm_supported(int type) {
    return type == FQ_Q4_0 || type == FQ_Q4_1 || type == FQ_Q5_0 || type == FQ_Q5_1 || type == FQ_Q8_0 || type == FQ_Q4_K || type == FQ_Q5_K;
}

template <int TYPE, int J>
static void launch_gemm_t(const fq_weight & w, const fq_act & act, int64_t N, float * dst, int64_t ldd, const fq_gemv_epi & ep, hipStream_t st) {
    constexpr int TM = 32 * J;
    constexpr bool HAS_MIN = gemm_group<TYPE>::HAS_MIN;
    const size_t lds = (size_t)(GQ_TN + TM) * GQ_STRIDE + (size_t) GQ_GROUPS * (GQ_TN + TM) * 4 * (HAS_MIN ? 2 : 1);
    const dim3 grid((unsigned)((w.M + TM - 1) / TM), (unsigned)((N + GQ_TN - 1) / GQ_TN));
    hipLaunchKernelGGL((k_gemm_q<TYPE, J>), grid, dim3(256), lds, st, w, act, N, dst, ldd, ep);
}

// dst[n*ldd + m], n < N; act holds N quantized columns
void fq_launch_gemm(const fq_weight & w, const fq_act & act, int64_t N, float * dst, int64_t ldd, const fq_gemv_epi & ep, int n_cu, hipStream_t st) {
    const int64_t tiles128 = ((w.M + 127) / 128) * ((N + GQ_TN - 1) / GQ_TN);
    const bool small = tiles128 < 2 * (int64_t) n_cu;            // few workgroups: halve the row tile to fill the chip
#define FQ_CASE(T) case T: if (small) launch_gemm_t<T, 2>(w, act, N, dst, ldd, ep, st); else launch_gemm_t<T, 4>(w, act, N, dst, ldd, ep, st); break;
    switch (w.type) {
        FQ_CASE(FQ_Q4_0) FQ_CASE(FQ_Q4_1) FQ_CASE(FQ_Q5_0) FQ_CASE(FQ_Q5_1) FQ_CASE(FQ_Q8_0) FQ_CASE(FQ_Q4_K) FQ_CASE(FQ_Q5_K)
        default: fprintf(stderr, "ggml-hip: gemm: unsupported weight type %d\n", w.type); exit(1);
    }
#undef FQ_CASE
}
