// kernels_quant.hip -- weight re-tiling, dequantize_row, activation quantizers (gfx950).
//
//  * k_retile        ggml array-of-blocks -> column-interleaved planes (fq_types.h; upload time; replaces the raw cudaMemcpy of
//                    ggml_cuda_transform_tensor, reference ggml-cuda.cu:3030-3073)
//  * k_dequant_rows  dequantize_row_q* (ggml.c:1509-1619, k_quants.c:344-876): bit-exact, used by get_rows
//                    (embedding lookup, ggml.c:11975) and by the parity tests
//  * k_quantize_q8   quantize_row_q8_0/q8_1 "_reference" semantics (ggml.c:1106-1129, 1292-1325)
//  * k_quantize_q8K  quantize_row_q8_K_reference (k_quants.c:899-934)
//  * k_act_export    SoA activations -> ggml block_q8_0 / block_q8_1 / block_q8_K bytes (tests, shim)
//
// Built with -ffp-contract=off: every a*b+c below is two roundings, as in the reference C.
#include "fq_block_dev.h"
#include "kernels.h"

// ------------------------------------------------------------------------------------------------ re-tile
__global__ void k_retile(const uint8_t * __restrict__ src, fq_weight w, int type) {
    const fq_type_desc d = fq_desc(type);
    const int64_t total = w.M * w.nblk;
    for (int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t) gridDim.x * blockDim.x) {
        const uint8_t * s = src + (size_t) i * d.tsize;
        const int64_t row = i / w.nblk, blk = i - row * w.nblk;
        for (int p = 0; p < d.nplanes; ++p) {
            uint8_t * o = w.plane[0] + (size_t) row * w.row_stride + fq_il_offset(d, p, w.nblk, blk);
            for (int b = 0; b < d.plane[p].bytes; ++b) o[b] = s[d.plane[p].src_off + b];
        }
    }
}

void fq_launch_retile(const uint8_t * src_dev, const fq_weight & w, hipStream_t st) {
    FQ_TL(st, "retile");
    const int64_t total = w.M * w.nblk;
    const int blocks = (int) ((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    hipLaunchKernelGGL(k_retile, dim3(blocks), dim3(256), 0, st, src_dev, w, w.type);
}

// ------------------------------------------------------------------------------------------------ dequantize
// element e of a row, arithmetic order exactly as the reference's dequantize_row_* (cited per case)
template <int TYPE>
__device__ __forceinline__ float dequant_elem(const fq_wrow & r, int64_t e) {
    if constexpr (TYPE == FQ_Q4_0) {                                 // ggml.c:1509-1527
        const int64_t b = e >> 5; const int j = e & 15, hi = (e >> 4) & 1;
        const int qb = fq_at<FQ_Q4_0, 0>(r, b)[j];
        const int q = hi ? (qb >> 4) : (qb & 15);
        return (float)(q - 8) * fq_h2f(ld_u16(fq_at<FQ_Q4_0, 1>(r, b)));
    } else if constexpr (TYPE == FQ_Q4_1) {                          // ggml.c:1529-1548
        const int64_t b = e >> 5; const int j = e & 15, hi = (e >> 4) & 1;
        const int qb = fq_at<FQ_Q4_1, 0>(r, b)[j];
        const int q = hi ? (qb >> 4) : (qb & 15);
        const uint32_t dm = ld_u32(fq_at<FQ_Q4_1, 1>(r, b));
        return (float) q * fq_h2f((uint16_t) dm) + fq_h2f((uint16_t)(dm >> 16));
    } else if constexpr (TYPE == FQ_Q5_0) {                          // ggml.c:1550-1574
        const int64_t b = e >> 5; const int j = e & 15, hi = (e >> 4) & 1;
        const int qb = fq_at<FQ_Q5_0, 0>(r, b)[j];
        const int q = (hi ? (qb >> 4) : (qb & 15)) | (int)(((ld_u32(fq_at<FQ_Q5_0, 1>(r, b)) >> (e & 31)) & 1u) << 4);
        return (float)(q - 16) * fq_h2f(ld_u16(fq_at<FQ_Q5_0, 2>(r, b)));
    } else if constexpr (TYPE == FQ_Q5_1) {                          // ggml.c:1576-1601
        const int64_t b = e >> 5; const int j = e & 15, hi = (e >> 4) & 1;
        const int qb = fq_at<FQ_Q5_1, 0>(r, b)[j];
        const int q = (hi ? (qb >> 4) : (qb & 15)) | (int)(((ld_u32(fq_at<FQ_Q5_1, 1>(r, b)) >> (e & 31)) & 1u) << 4);
        const uint32_t dm = ld_u32(fq_at<FQ_Q5_1, 2>(r, b));
        return (float) q * fq_h2f((uint16_t) dm) + fq_h2f((uint16_t)(dm >> 16));
    } else if constexpr (TYPE == FQ_Q8_0) {                          // ggml.c:1603-1619
        return (float)(int)(int8_t) fq_at<FQ_Q8_0, 0>(r, e >> 5)[e & 31] * fq_h2f(ld_u16(fq_at<FQ_Q8_0, 1>(r, e >> 5)));
    } else if constexpr (TYPE == FQ_Q2_K) {                          // k_quants.c:344-375
        const int64_t sb = e >> 8; const int i = e & 255;
        const int is = (i >> 7) * 8 + ((i & 127) >> 5) * 2 + ((i & 31) >> 4);
        const int sc = fq_at<FQ_Q2_K, 1>(r, sb)[is];
        const int q  = (fq_at<FQ_Q2_K, 0>(r, sb)[(i >> 7) * 32 + (i & 31)] >> (2 * ((i & 127) >> 5))) & 3;
        const uint32_t dm = ld_u32(fq_at<FQ_Q2_K, 2>(r, sb));
        const float dl = fq_h2f((uint16_t) dm) * (float)(sc & 15), ml = fq_h2f((uint16_t)(dm >> 16)) * (float)(sc >> 4);
        return dl * (float) q - ml;
    } else if constexpr (TYPE == FQ_Q3_K) {                          // k_quants.c:472-521
        const int64_t sb = e >> 8; const int i = e & 255;
        const int is = (i >> 7) * 8 + ((i & 127) >> 5) * 2 + ((i & 31) >> 4);
        const int lo = (fq_at<FQ_Q3_K, 0>(r, sb)[(i >> 7) * 32 + (i & 31)] >> (2 * ((i & 127) >> 5))) & 3;
        const int hb = (fq_at<FQ_Q3_K, 1>(r, sb)[i & 31] >> (i >> 5)) & 1;
        const uint8_t * scp = fq_at<FQ_Q3_K, 2>(r, sb);
        const int sc = q3_scale(ld_u32(scp), ld_u32(scp + 4), ld_u32(scp + 8), is);
        const float dl = fq_h2f(ld_u16(fq_at<FQ_Q3_K, 3>(r, sb))) * (float)(sc - 32);
        return dl * (float)(lo - (hb ? 0 : 4));
    } else if constexpr (TYPE == FQ_Q4_K || TYPE == FQ_Q5_K) {      // k_quants.c:607-631, 734-760
        const int64_t sb = e >> 8; const int i = e & 255;
        const int c = i >> 6, hi = (i >> 5) & 1;
        constexpr int PSC = (TYPE == FQ_Q4_K) ? 1 : 2;
        const uint8_t * scp = fq_at<TYPE, PSC>(r, sb);
        int sc, mn; k4_scale_min(ld_u32(scp), ld_u32(scp + 4), ld_u32(scp + 8), 2 * c + hi, sc, mn);
        const int byte = fq_at<TYPE, 0>(r, sb)[32 * c + (i & 31)];
        int q = hi ? (byte >> 4) : (byte & 15);
        if constexpr (TYPE == FQ_Q5_K) q += ((fq_at<TYPE, 1>(r, sb)[i & 31] >> (i >> 5)) & 1) ? 16 : 0;
        const uint32_t dm = ld_u32(fq_at<TYPE, PSC + 1>(r, sb));
        const float dl = fq_h2f((uint16_t) dm) * (float) sc, ml = fq_h2f((uint16_t)(dm >> 16)) * (float) mn;
        return dl * (float) q - ml;
    } else {                                                         // Q6_K, k_quants.c:845-876
        const int64_t sb = e >> 8; const int i = e & 255;
        const int h = i >> 7, t = (i & 127) >> 5, l = i & 31;
        const int byte = fq_at<FQ_Q6_K, 0>(r, sb)[64 * h + 32 * (t & 1) + l];
        const int lo = (t < 2) ? (byte & 15) : (byte >> 4);
        const int hi = (fq_at<FQ_Q6_K, 1>(r, sb)[32 * h + l] >> (2 * t)) & 3;
        const int q = (int)(int8_t)(lo | (hi << 4)) - 32;
        const int sc = (int)(int8_t) fq_at<FQ_Q6_K, 2>(r, sb)[8 * h + 2 * t + (l >> 4)];
        return fq_h2f(ld_u16(fq_at<FQ_Q6_K, 3>(r, sb))) * (float) sc * (float) q;
    }
}

// rows[i] selects the weight row written to dst row i (rows == nullptr: identity)
template <int TYPE>
__global__ void k_dequant_rows(fq_weight w, const int32_t * __restrict__ rows, int64_t nrows, float * __restrict__ dst) {
    const int64_t total = nrows * w.K;
    for (int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t) gridDim.x * blockDim.x) {
        const int64_t ri = i / w.K, e = i - ri * w.K;
        const int64_t src = rows ? (int64_t) rows[ri] : ri;
        dst[i] = dequant_elem<TYPE>(fq_row<TYPE>(w, src), e);
    }
}

void fq_launch_dequant_rows(const fq_weight & w, const int32_t * rows_dev, int64_t nrows, float * dst, hipStream_t st) {
    FQ_TL(st, "dequant_rows");
    const int64_t total = nrows * w.K;
    const int blocks = (int) ((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256);
#define FQ_CASE(T) case T: hipLaunchKernelGGL(k_dequant_rows<T>, dim3(blocks), dim3(256), 0, st, w, rows_dev, nrows, dst); break;
    switch (w.type) {
        FQ_CASE(FQ_Q4_0) FQ_CASE(FQ_Q4_1) FQ_CASE(FQ_Q5_0) FQ_CASE(FQ_Q5_1) FQ_CASE(FQ_Q8_0)
        FQ_CASE(FQ_Q2_K) FQ_CASE(FQ_Q3_K) FQ_CASE(FQ_Q4_K) FQ_CASE(FQ_Q5_K) FQ_CASE(FQ_Q6_K)
        default: fprintf(stderr, "ggml-hip: dequantize: unsupported type %d\n", w.type); exit(1);
    }
#undef FQ_CASE
}

// column accessors of the activation image (fq_types.h)
__device__ __forceinline__ int8_t * act_qs (const fq_act & a, int64_t col) { return (int8_t *)(a.base + (size_t) col * fq_act_col_bytes(a.type, a.K)); }
__device__ __forceinline__ float  * act_d  (const fq_act & a, int64_t col) { return (float *)(a.base + (size_t) col * fq_act_col_bytes(a.type, a.K) + fq_act_d_off(a.type, a.K)); }
__device__ __forceinline__ uint8_t * act_aux(const fq_act & a, int64_t col) { return a.base + (size_t) col * fq_act_col_bytes(a.type, a.K) + fq_act_aux_off(a.type, a.K); }

// ------------------------------------------------------------------------------------------------ Q8_0 / Q8_1
// One thread = 4 consecutive elements, 8 threads = one 32-block. x is [ncols][K] with row stride ldx (floats).
template <int ACT>
__global__ void k_quantize_q8(const float * __restrict__ x, int64_t ldx, fq_act a) {
    const int64_t quads_per_col = a.K >> 2;
    const int64_t total = quads_per_col * a.ncols;
    // whole waves stay in the loop together (total is a multiple of 8 and the shuffles are within groups of 8)
    for (int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; i < ((total + 63) & ~(int64_t) 63); i += (int64_t) gridDim.x * blockDim.x) {
        const bool live = i < total;
        const int64_t ii = live ? i : total - 1;
        const int64_t col = ii / quads_per_col, q4 = ii - col * quads_per_col;
        const act_image_ptr o = { act_qs(a, col), act_d(a, col), act_aux(a, col) };
        quant_q8_quad<ACT>(*(const float4 *)(x + col * ldx + 4 * q4), q4, o, live);
    }
}

// ------------------------------------------------------------------------------------------------ Q8_K
__global__ void k_quantize_q8K(const float * __restrict__ x, int64_t ldx, fq_act a) {
    const int64_t sb_per_col = a.K >> 8;
    const int64_t total_sb = sb_per_col * a.ncols;
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = ((int64_t) blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t) gridDim.x * blockDim.x) >> 6;
    for (int64_t sbi = wave0; sbi < total_sb; sbi += nwaves) {
        const int64_t col = sbi / sb_per_col, sb = sbi - col * sb_per_col;
        const act_image_ptr o = { act_qs(a, col), act_d(a, col), act_aux(a, col) };
        quant_q8K_wave(*(const float4 *)(x + col * ldx + 256 * sb + 4 * lane), lane, sb, o);
    }
}

void fq_launch_quantize_act(const float * x, int64_t ldx, const fq_act & a, hipStream_t st) {
    FQ_TL(st, "quantize_act");
    if (a.type == FQ_Q8_K) {
        const int64_t total_sb = (a.K >> 8) * a.ncols;
        const int blocks = (int) ((total_sb + 3) / 4 > 4096 ? 4096 : (total_sb + 3) / 4);
        hipLaunchKernelGGL(k_quantize_q8K, dim3(blocks), dim3(256), 0, st, x, ldx, a);
        return;
    }
    const int64_t total = (a.K >> 2) * a.ncols;
    const int blocks = (int) ((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    if (a.type == FQ_Q8_0) hipLaunchKernelGGL(k_quantize_q8<FQ_Q8_0>, dim3(blocks), dim3(256), 0, st, x, ldx, a);
    else                   hipLaunchKernelGGL(k_quantize_q8<FQ_Q8_1>, dim3(blocks), dim3(256), 0, st, x, ldx, a);
}

// ------------------------------------------------------------------------------------------------ export (tests / shim)
__global__ void k_act_export(fq_act a, uint8_t * __restrict__ out) {
    const int64_t K = a.K;
    if (a.type == FQ_Q8_K) {
        const int64_t per_col = K >> 8, total = per_col * a.ncols;
        for (int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t) gridDim.x * blockDim.x) {
            const int64_t col = i / per_col, sb = i - col * per_col;
            uint8_t * o = out + i * 292;
            const float d = act_d(a, col)[sb];
            for (int b = 0; b < 4; ++b) o[b] = ((const uint8_t *) &d)[b];
            const int8_t * q = act_qs(a, col) + 256 * sb;
            for (int j = 0; j < 256; ++j) o[4 + j] = (uint8_t) q[j];
            const uint8_t * bs = act_aux(a, col) + 32 * sb;
            for (int j = 0; j < 32; ++j) o[260 + j] = bs[j];
        }
        return;
    }
    const int64_t per_col = K >> 5, total = per_col * a.ncols;
    for (int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t) gridDim.x * blockDim.x) {
        const int64_t col = i / per_col, b = i - col * per_col;
        const int8_t * q = act_qs(a, col) + 32 * b;
        if (a.type == FQ_Q8_0) {
            uint8_t * o = out + i * 34;
            const uint16_t h = f2h_bits(act_d(a, col)[b]);
            o[0] = (uint8_t) h; o[1] = (uint8_t)(h >> 8);
            for (int j = 0; j < 32; ++j) o[2 + j] = (uint8_t) q[j];
        } else {
            uint8_t * o = out + i * 40;
            const float d = act_d(a, col)[b], sv = ((const float *) act_aux(a, col))[b];
            for (int k = 0; k < 4; ++k) { o[k] = ((const uint8_t *) &d)[k]; o[4 + k] = ((const uint8_t *) &sv)[k]; }
            for (int j = 0; j < 32; ++j) o[8 + j] = (uint8_t) q[j];
        }
    }
}

void fq_launch_act_export(const fq_act & a, uint8_t * out, hipStream_t st) {
    FQ_TL(st, "act_export");
    const int64_t total = (a.type == FQ_Q8_K ? (a.K >> 8) : (a.K >> 5)) * a.ncols;
    const int blocks = (int) ((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL(k_act_export, dim3(blocks), dim3(256), 0, st, a, out);
}
