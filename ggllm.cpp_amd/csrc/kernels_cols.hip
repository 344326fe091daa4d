// kernels_cols.hip -- the two mat-vec launches of a decoder block for 2..4 LOCK-STEP sequences (falcon_hip_context_create_seqs):
// one pass over the block's weights serves every sequence's token.
//
//   k_gemv_cols      rows [Wqkv | Wup] (two segments, like k_gemv_ln): a workgroup of 12 waves owns 96 consecutive rows of one
//                    segment, stages the NC quantized activation columns of that segment in LDS once, and every weight
//                    unit it loads is dotted with all NC columns; epilogue store | GELU | GELU + Q8 image (per column)
//   k_gemv_out_cols  x[c][row] = (Wdown . q8(gelu(up))[c] + Wo . q8(att)[c]) + x[c][row]: a workgroup owns 2 rows per
//                    wave, both activation column sets in LDS (like k_gemv_out)
//
// Replaces, for these N, the generic k_gemv launches of the op list (one workgroup of 4 waves per 8 rows, each staging all
// columns again: 1.3-1.6 TB/s at Falcon-7B width). Arithmetic: per lane the units of a row in ascending order from 0.0f,
// then the wave reduction -- the order of k_gemv, k_gemv_ln and k_gemv_out, so every column has the bits of a single-sequence
// step (tests/test_gpu_pipeline.py). Reference work replaced: ggml_compute_forward_mul_mat_q_f32 for N = 2..4
// (ggml.c:11484-11516), dequantize_mul_mat_vec (ggml-cuda.cu:1120-1171) called once per column there.
#include "fq_device.h"
#include "fq_units.h"
#include "fq_block_dev.h"
#include "kernels.h"

namespace {

template <int TYPE> struct act_of { static constexpr int value =
    (TYPE == FQ_Q4_1 || TYPE == FQ_Q5_1) ? FQ_Q8_1 : ((TYPE == FQ_Q4_0 || TYPE == FQ_Q5_0 || TYPE == FQ_Q8_0) ? FQ_Q8_0 : FQ_Q8_K); };
// unit columns (64 units = 1 KiB of plane 0 per row) a wave keeps in flight per trip; fq_unit_regs is 5 dwords for the legacy
// formats, up to 12 for the k-quants. Rows of Falcon-7B width (142 units = 3 columns) take one trip of U_E.
template <int TYPE> struct cols_cfg {
    static constexpr bool small_regs = (TYPE == FQ_Q4_0 || TYPE == FQ_Q4_1 || TYPE == FQ_Q5_0 || TYPE == FQ_Q5_1 || TYPE == FQ_Q8_0);
    static constexpr int U_E  = small_regs ? 3 : 1;                // rows of length n_embd (k_gemv_cols, Wo)
    static constexpr int U_FF = !small_regs ? 1 : (TYPE == FQ_Q8_0 ? 2 : 4);   // rows of length n_ff (Wdown)
    static constexpr int PRE  = 1;                                 // columns requested before the activation images are staged
};
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int TYPE, int R>
__device__ __forceinline__ void rows_ptrs(const fq_weight & w, int64_t row0, fq_wrow (&rows)[R]) {
#pragma unroll
    for (int r = 0; r < R; ++r) { const int64_t row = row0 + r; rows[r] = fq_row<TYPE>(w, row < w.M ? row : w.M - 1); }
}
template <int TYPE, int R, int U>
__device__ __forceinline__ void cols_issue(const fq_wrow (&rows)[R], int units, int c0, fq_unit_regs (&regs)[U][R]) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < U; ++i) {
#pragma unroll
        for (int r = 0; r < R; ++r) regs[i][r] = fq_unit_load_col<TYPE>(rows[r], c0 + i, lane, units);
    }
}
template <int TYPE, int R, int NC, int U>
__device__ __forceinline__ void cols_consume(const fq_unit_regs (&regs)[U][R], int units, int c0, const fq_actcol (&cols)[NC], float (&acc)[R][NC]) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < U; ++i) {
        const int u = (c0 + i) * 64 + lane; const bool ok = u < units; const int uc = ok ? u : units - 1;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
#pragma unroll
            for (int r = 0; r < R; ++r) { const float v = fq_unit<TYPE>::dot(regs[i][r], cols[c], uc); acc[r][c] += ok ? v : 0.0f; }
        }
    }
}
template <int TYPE, int R, int NC, int U>
__device__ __forceinline__ void cols_trip(const fq_wrow (&rows)[R], int units, int c0, const fq_actcol (&cols)[NC], float (&acc)[R][NC]) {
    fq_unit_regs regs[U][R];
    cols_issue<TYPE, R, U>(rows, units, c0, regs);
    cols_consume<TYPE, R, NC, U>(regs, units, c0, cols, acc);
}
// unit columns [c_begin, ceil(units / 64)) in trips of U, the last trip only as wide as what is left (no clamped re-loads)
template <int TYPE, int R, int NC, int U>
__device__ __forceinline__ void cols_dot_from(const fq_wrow (&rows)[R], int units, int c_begin, const fq_actcol (&cols)[NC], float (&acc)[R][NC]) {
    const int ncol = (units + 63) >> 6;
    int c0 = c_begin;
    for (; c0 + U <= ncol; c0 += U) cols_trip<TYPE, R, NC, U>(rows, units, c0, cols, acc);
    const int rem = ncol - c0;
    if constexpr (U > 1) { if (rem == 1) cols_trip<TYPE, R, NC, 1>(rows, units, c0, cols, acc); }
    if constexpr (U > 2) { if (rem == 2) cols_trip<TYPE, R, NC, 2>(rows, units, c0, cols, acc); }
    if constexpr (U > 3) { if (rem == 3) cols_trip<TYPE, R, NC, 3>(rows, units, c0, cols, acc); }
}
// n16 16-byte vectors global -> LDS by the whole workgroup, 4 loads in flight per thread
__device__ __forceinline__ void stage16(u32x4 * __restrict__ dst, const u32x4 * __restrict__ src, int64_t n16) {
    const int tid = threadIdx.x, nt = blockDim.x;
    for (int64_t base = 0; base < n16; base += 4 * (int64_t) nt) {
        u32x4 t[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int64_t i = base + (int64_t) k * nt + tid; t[k] = src[i < n16 ? i : n16 - 1]; }
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int64_t i = base + (int64_t) k * nt + tid; if (i < n16) dst[i] = t[k]; }
    }
}
__device__ __forceinline__ fq_actcol actcol_at(const uint8_t * base, int act_type, int64_t K) {
    return { (const int8_t *) base, (const float *)(base + fq_act_d_off(act_type, K)), (const void *)(base + fq_act_aux_off(act_type, K)) };
}

// =============================================================================================== k_gemv_cols
template <int TYPE, int NC>
__global__ void __launch_bounds__(768) k_gemv_cols(fq_gemv_cols_args a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr int ACT = act_of<TYPE>::value, R = 2, U = cols_cfg<TYPE>::U_E, PRE = cols_cfg<TYPE>::PRE;
    const int bid = blockIdx.x;
    const int sidx = (a.nseg > 1 && bid >= a.seg[1].block_begin) ? 1 : 0;
    const fq_gemv_cols_seg sg = sidx ? a.seg[1] : a.seg[0];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, nw = blockDim.x >> 6;
    const int64_t K = sg.w.K, M = sg.w.M;
    const int RW = R * a.npass, WGR = RW * nw;                      // rows per wave / per workgroup (a multiple of 32)
    const int64_t row0 = (int64_t)(bid - sg.block_begin) * WGR;
    const size_t colb = fq_act_col_bytes(ACT, K);
    float * out32 = (float *)(smem + colb * NC);                    // [NC][WGR]
    const int units = (int)(K / fq_unit<TYPE>::ELEMS);

    fq_wrow rows0[R];
    rows_ptrs<TYPE, R>(sg.w, row0 + RW * wid, rows0);
    fq_unit_regs pre[PRE][R];
    cols_issue<TYPE, R, PRE>(rows0, units, 0, pre);                 // the first weight column streams while the images are staged
    stage16((u32x4 *) smem, (const u32x4 *) sg.act, (int64_t)(colb * a.ncols) >> 4);
    fq_actcol cols[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) cols[c] = actcol_at(smem + colb * (c < a.ncols ? c : 0), ACT, K);
    __syncthreads();
    // (measured on MI355X, Falcon-7B Q4_0, 4 columns: requesting trip k + 1 before trip k is consumed -- two register sets --
    // does not pay: 24.4 us against 23.4 for this form, and 30.5 against 21.6 in k_gemv_out_cols; with 4 columns per weight
    // unit the waves are busy issuing dots, not waiting for loads)
    for (int p = 0; p < a.npass; ++p) {
        float acc[R][NC];
#pragma unroll
        for (int r = 0; r < R; ++r) {
#pragma unroll
            for (int c = 0; c < NC; ++c) acc[r][c] = 0.0f;
        }
        if (p == 0) {
            cols_consume<TYPE, R, NC, PRE>(pre, units, 0, cols, acc);
            cols_dot_from<TYPE, R, NC, U>(rows0, units, PRE, cols, acc);
        } else {
            fq_wrow rowsp[R];
            rows_ptrs<TYPE, R>(sg.w, row0 + RW * wid + R * p, rowsp);
            cols_dot_from<TYPE, R, NC, U>(rowsp, units, 0, cols, acc);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
#pragma unroll
            for (int c = 0; c < NC; ++c) acc[r][c] = wave_sum(acc[r][c]);
        }
        if (lane == 0) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
#pragma unroll
                for (int c = 0; c < NC; ++c) out32[c * WGR + RW * wid + R * p + r] = acc[r][c];
            }
        }
    }
    __syncthreads();
    // epilogue: a wave finishes the 32 rows [32 g, 32 g + 32) of one column (lanes 32..63 mirror 0..31), as k_gemv_ln does
    const int ngrp = WGR / 32;
    for (int item = wid; item < ngrp * a.ncols; item += nw) {
        const int c = item / ngrp, grp = item - c * ngrp;
        const int j = lane & 31;
        const int64_t row = row0 + 32 * grp + j;
        float v = out32[c * WGR + 32 * grp + j];
        if (sg.epi == FQ_LNEPI_STORE) {
            if (lane < 32 && row < M) sg.dst[(int64_t) c * sg.ldd + row] = v;
        } else {
            v = h2f_bits(a.gelu_table[f2h_bits(v)]);                                  // ggml.c:3477-3484
            if (sg.epi == FQ_LNEPI_GELU_STORE) {
                if (lane < 32 && row < M) sg.dst[(int64_t) c * sg.ldd + row] = v;
            } else if (row0 + 32 * grp < M) {                                         // GELU -> Q8_0 / Q8_1 block of 32 (M % 32 == 0)
                const float amax = reduce32(fabsf(v), op_max());
                const float d  = amax / 127.0f;
                const float id = d ? 1.0f / d : 0.0f;
                const int q = round_half_away(v * id);
                const int s = reduce32(q, op_add());
                const act_image_ptr o = act_image_at(sg.dst_image + (size_t) c * fq_act_col_bytes(sg.next_act_type, M), sg.next_act_type, M);
                if (lane < 32) o.qs[row] = (int8_t) q;
                if (lane == 0) {
                    const int64_t b = (row0 >> 5) + grp;
                    if (sg.next_act_type == FQ_Q8_0) { o.d[b] = h2f_bits(f2h_bits(d)); ((int32_t *) o.aux)[b] = s; }
                    else                             { o.d[b] = d; ((float *) o.aux)[b] = (float) s * d; }
                }
            }
        }
    }
}

// =============================================================================================== k_gemv_out_cols
template <int TYPE, int NC>
__global__ void __launch_bounds__(768) k_gemv_out_cols(fq_gemv_out_cols_args a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr int ACT = act_of<TYPE>::value, R = 2, UD = cols_cfg<TYPE>::U_FF, UO = cols_cfg<TYPE>::U_E, PRE = cols_cfg<TYPE>::PRE;
    const int64_t E = a.w_wo.K, FF = a.w_down.K, M = a.w_wo.M;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, nw = blockDim.x >> 6;
    const size_t cb_ff = fq_act_col_bytes(ACT, FF), cb_at = fq_act_col_bytes(ACT, E);
    uint8_t * img_ff = smem, * img_at = smem + cb_ff * NC;
    const int units_d = (int)(FF / fq_unit<TYPE>::ELEMS), units_o = (int)(E / fq_unit<TYPE>::ELEMS);
    const int64_t row0 = (int64_t) blockIdx.x * (R * nw) + R * wid;

    fq_wrow rd[R], ro[R];
    rows_ptrs<TYPE, R>(a.w_down, row0, rd);
    rows_ptrs<TYPE, R>(a.w_wo, row0, ro);
    fq_unit_regs pd[PRE][R];
    cols_issue<TYPE, R, PRE>(rd, units_d, 0, pd);
    stage16((u32x4 *) img_ff, (const u32x4 *) a.act_ff_image, (int64_t)(cb_ff * a.ncols) >> 4);
    stage16((u32x4 *) img_at, (const u32x4 *) a.att_image, (int64_t)(cb_at * a.ncols) >> 4);
    fq_actcol cd[NC], co[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int cc = c < a.ncols ? c : 0;
        cd[c] = actcol_at(img_ff + cb_ff * cc, ACT, FF);
        co[c] = actcol_at(img_at + cb_at * cc, ACT, E);
    }
    __syncthreads();
    float acc_d[R][NC], acc_o[R][NC];
#pragma unroll
    for (int r = 0; r < R; ++r) {
#pragma unroll
        for (int c = 0; c < NC; ++c) { acc_d[r][c] = 0.0f; acc_o[r][c] = 0.0f; }
    }
    cols_consume<TYPE, R, NC, PRE>(pd, units_d, 0, cd, acc_d);
    cols_dot_from<TYPE, R, NC, UD>(rd, units_d, PRE, cd, acc_d);
    cols_dot_from<TYPE, R, NC, UO>(ro, units_o, 0, co, acc_o);
#pragma unroll
    for (int r = 0; r < R; ++r) {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const float vd = wave_sum(acc_d[r][c]), vo = wave_sum(acc_o[r][c]);
            if (lane == 0 && row0 + r < M && c < a.ncols) {
                const float res = a.resid[(int64_t) c * a.ld + row0 + r];
                a.dst[(int64_t) c * a.ld + row0 + r] = (vd + vo) + res;                       // libfalcon.cpp:2399-2400
            }
        }
    }
}

template <int TYPE, int NC>
void launch_cols_t(const fq_gemv_cols_args & a, unsigned blocks, size_t lds, hipStream_t st) {
    static size_t g = 0;
    if (lds > 64 * 1024 && lds > g) { HIP_CHECK(hipFuncSetAttribute((const void *) k_gemv_cols<TYPE, NC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds)); g = lds; }
    hipLaunchKernelGGL((k_gemv_cols<TYPE, NC>), dim3(blocks), dim3(768), lds, st, a);
}
template <int TYPE, int NC>
void launch_out_cols_t(const fq_gemv_out_cols_args & a, unsigned blocks, int nw, size_t lds, hipStream_t st) {
    static size_t g = 0;
    if (lds > 64 * 1024 && lds > g) { HIP_CHECK(hipFuncSetAttribute((const void *) k_gemv_out_cols<TYPE, NC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds)); g = lds; }
    hipLaunchKernelGGL((k_gemv_out_cols<TYPE, NC>), dim3(blocks), dim3(64 * nw), lds, st, a);
}

}   // namespace

// both launchers return false (nothing launched) when the shape is outside their scope; the caller keeps the op list
bool fq_launch_gemv_cols(fq_gemv_cols_args a, int n_cu, hipStream_t st) {
    FQ_TL(st, "gemv_cols");
    (void) n_cu;
    if (a.ncols < 1 || a.ncols > 4 || a.nseg < 1 || a.nseg > 2) return false;
    const int type = a.seg[0].w.type;
    const int act = fq_desc(type).act_type;
    a.npass = 4;                                                        // 12 waves x 4 passes x 2 rows = 96 rows per workgroup
    const int rows = 2 * a.npass * 12;
    unsigned blocks = 0; size_t lds = 0;
    for (int s = 0; s < a.nseg; ++s) {
        if (a.seg[s].w.type != type) return false;
        if (a.seg[s].epi == FQ_LNEPI_GELU_QUANT && a.seg[s].w.M % 32 != 0) return false;
        a.seg[s].block_begin = (int) blocks;
        blocks += (unsigned)((a.seg[s].w.M + rows - 1) / rows);
        const size_t need = fq_act_col_bytes(act, a.seg[s].w.K) * (size_t)(a.ncols > 2 ? 4 : 2) + (size_t)(a.ncols > 2 ? 4 : 2) * rows * 4;
        if (need > lds) lds = need;
    }
    if (lds > 160 * 1024) return false;
#define FQ_CASE(T) case T: if (a.ncols > 2) launch_cols_t<T, 4>(a, blocks, lds, st); else launch_cols_t<T, 2>(a, blocks, lds, st); break;
    switch (type) {
        FQ_CASE(FQ_Q4_0) FQ_CASE(FQ_Q4_1) FQ_CASE(FQ_Q5_0) FQ_CASE(FQ_Q5_1) FQ_CASE(FQ_Q8_0)
        FQ_CASE(FQ_Q2_K) FQ_CASE(FQ_Q3_K) FQ_CASE(FQ_Q4_K) FQ_CASE(FQ_Q5_K) FQ_CASE(FQ_Q6_K)
        default: return false;
    }
#undef FQ_CASE
    return true;
}

// widest column chunk (4 or 2; 0: none) whose two activation column sets fit the LDS of k_gemv_out_cols
int fq_gemv_out_cols_width(int type, int64_t K_down, int64_t K_wo) {
    const int act = fq_desc(type).act_type;
    const size_t per = fq_act_col_bytes(act, K_down) + fq_act_col_bytes(act, K_wo);
    return per * 4 + 16 <= 160 * 1024 ? 4 : (per * 2 + 16 <= 160 * 1024 ? 2 : 0);
}
bool fq_launch_gemv_out_cols(const fq_gemv_out_cols_args & a, int n_cu, hipStream_t st) {
    FQ_TL(st, "gemv_out_cols");
    if (a.ncols < 1 || a.ncols > 4 || a.w_down.type != a.w_wo.type || a.w_down.M != a.w_wo.M) return false;
    const int type = a.w_wo.type;
    const int act = fq_desc(type).act_type;
    const int nc = a.ncols > 2 ? 4 : 2;
    const size_t lds = (fq_act_col_bytes(act, a.w_down.K) + fq_act_col_bytes(act, a.w_wo.K)) * (size_t) nc + 16;
    if (lds > 160 * 1024) return false;
    int nw = 4; int64_t best_cost = INT64_MAX;                          // as fq_launch_gemv_out: fewest rounds x rows per workgroup
    for (int c = 4; c <= 12; ++c) {
        const int64_t nb = (a.w_wo.M + 2 * c - 1) / (2 * c);
        const int64_t cost = ((nb + n_cu - 1) / n_cu) * c;
        if (cost <= best_cost) { best_cost = cost; nw = c; }
    }
    const unsigned blocks = (unsigned)((a.w_wo.M + 2 * nw - 1) / (2 * nw));
#define FQ_CASE(T) case T: if (a.ncols > 2) launch_out_cols_t<T, 4>(a, blocks, nw, lds, st); else launch_out_cols_t<T, 2>(a, blocks, nw, lds, st); break;
    switch (type) {
        FQ_CASE(FQ_Q4_0) FQ_CASE(FQ_Q4_1) FQ_CASE(FQ_Q5_0) FQ_CASE(FQ_Q5_1) FQ_CASE(FQ_Q8_0)
        FQ_CASE(FQ_Q2_K) FQ_CASE(FQ_Q3_K) FQ_CASE(FQ_Q4_K) FQ_CASE(FQ_Q5_K) FQ_CASE(FQ_Q6_K)
        default: return false;
    }
#undef FQ_CASE
    return true;
}
