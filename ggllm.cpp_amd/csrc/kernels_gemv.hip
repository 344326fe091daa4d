// kernels_gemv.hip -- decode-time quantized mat-vec for all ten weight formats (gfx950, wave64).
//
// Replaces, for N <= FQ_GEMV_MAX_COLS activation columns, the reference's CPU
// ggml_compute_forward_mul_mat_q_f32 COMPUTE phase (ggml.c:11484-11516) and its CUDA twin
// dequantize_mul_mat_vec* (ggml-cuda.cu:475-845, 1120-1171) -- with the CPU's arithmetic (8-bit activations,
// exact int32 block dots, fp32 scale epilogue), not the CUDA path's dequantize-then-FMA.
//
// Shape of the kernel (HBM-bound: 18 B / 32 weights for Q4_0, the activations are tiny):
//   * activations are staged ONCE per workgroup into LDS (int8 + per-block scales, <= 40 KiB at K = 32768)
//   * a wave owns R consecutive output rows; lane l takes units l, l+64, ... of each row (unit = one 16-byte
//     quant group, fq_units.h), so every plane0 load is a fully coalesced 1 KiB `global_load_dwordx4`
//   * the loads of UNROLL x R units are issued back to back before the first dot, keeping
//     R*UNROLL KiB per wave in flight; nothing round-trips through LDS for the weights
//   * v_dot4_i32_i8 integer dots, fp32 scale per unit, 6-step wave reduction, fused epilogue
//     (store | GELU via the fp16 table | residual add)
#include "fq_device.h"
#include "fq_units.h"
#include "kernels.h"

size_t fq_gemv_lds_bytes(int act_type, int64_t K, int ncols) { return fq_act_col_bytes(act_type, K) * (size_t) ncols; }

// 16-byte vectors global -> LDS, all loads of a chunk issued before the first store (no per-iteration vmcnt(0))
__device__ __forceinline__ void stage_vec16(fq_u4 * __restrict__ dst, const fq_u4 * __restrict__ src, int64_t nvec) {
    const int tid = threadIdx.x, nt = blockDim.x;
    for (int64_t base = 0; base < nvec; base += 8 * nt) {
        fq_u4 tmp[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { const int64_t i = base + (int64_t) k * nt + tid; tmp[k] = src[i < nvec ? i : nvec - 1]; }
#pragma unroll
        for (int k = 0; k < 8; ++k) { const int64_t i = base + (int64_t) k * nt + tid; if (i < nvec) dst[i] = tmp[k]; }
    }
}

template <int TYPE, int R, int NCOLS, int UNROLL>
__global__ void __launch_bounds__(256) k_gemv(fq_weight w, fq_act act, float * dst /* may alias ep.add2: in-place residual */, int64_t ldd, fq_gemv_epi ep) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr int ACT   = (TYPE == FQ_Q4_1 || TYPE == FQ_Q5_1) ? FQ_Q8_1 : ((TYPE == FQ_Q4_0 || TYPE == FQ_Q5_0 || TYPE == FQ_Q8_0) ? FQ_Q8_0 : FQ_Q8_K);
    constexpr int ELEMS = fq_unit<TYPE>::ELEMS;
    const int64_t K = w.K, M = w.M;
    const int tid = threadIdx.x, lane = tid & 63;

    // ---- stage the quantized activation images (global/L2 -> LDS): one flat copy, same offsets in LDS
    const size_t col_stride = fq_act_col_bytes(ACT, K);
    stage_vec16((fq_u4 *) smem, (const fq_u4 *) act.base, (int64_t)(col_stride * NCOLS) >> 4);
    fq_actcol cols[NCOLS];
#pragma unroll
    for (int c = 0; c < NCOLS; ++c) {
        const uint8_t * b = smem + c * col_stride;
        cols[c] = { (const int8_t *) b, (const float *)(b + fq_act_d_off(ACT, K)), (const void *)(b + fq_act_aux_off(ACT, K)) };
    }
    __syncthreads();

    const int units  = (int)(K / ELEMS);
    const int gw     = blockIdx.x * (blockDim.x >> 6) + (tid >> 6);
    const int nw     = gridDim.x * (blockDim.x >> 6);
    const int64_t ngroups = (M + R - 1) / R;

    for (int64_t grp = gw; grp < ngroups; grp += nw) {
        float acc[R][NCOLS];
        fq_wrow rows[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int64_t row = grp * R + r;
            rows[r] = fq_row<TYPE>(w, row < M ? row : M - 1);
#pragma unroll
            for (int c = 0; c < NCOLS; ++c) acc[r][c] = 0.0f;
        }
        for (int u0 = 0; u0 < units; u0 += 64 * UNROLL) {
            fq_unit_regs regs[UNROLL][R];
#pragma unroll
            for (int i = 0; i < UNROLL; ++i) {
#pragma unroll
                for (int r = 0; r < R; ++r) regs[i][r] = fq_unit_load_col<TYPE>(rows[r], (u0 >> 6) + i, lane, units);
            }
#pragma unroll
            for (int i = 0; i < UNROLL; ++i) {
                const int u  = u0 + i * 64 + lane;
                const bool ok = u < units;
                const int uc = ok ? u : units - 1;
#pragma unroll
                for (int c = 0; c < NCOLS; ++c) {
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const float v = fq_unit<TYPE>::dot(regs[i][r], cols[c], uc);
                        acc[r][c] += ok ? v : 0.0f;
                    }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
#pragma unroll
            for (int c = 0; c < NCOLS; ++c) acc[r][c] = wave_sum(acc[r][c]);      // R*NCOLS independent chains
        }
        if (lane == 0) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int64_t row = grp * R + r;
                if (row < M) {
#pragma unroll
                    for (int c = 0; c < NCOLS; ++c) {
                        float v = acc[r][c];
                        if (ep.mode == FQ_EPI_GELU) {
                            v = h2f_bits(ep.gelu_table[f2h_bits(v)]);                  // ggml.c:3477-3484
                        } else if (ep.mode == FQ_EPI_ADD2) {
                            v = (v + ep.add1[c * ep.ld_add + row]) + ep.add2[c * ep.ld_add + row];   // libfalcon.cpp:2399-2400
                        }
                        dst[c * ldd + row] = v;
                    }
                }
            }
        }
    }
}

template <int TYPE, int NCOLS>
static void launch_gemv_t(const fq_weight & w, const fq_act & act, float * dst, int64_t ldd, const fq_gemv_epi & ep, int max_blocks, hipStream_t st) {
    // rows per wave / unroll: keep ~8 KiB of quant bytes per wave in flight without blowing the register file
    constexpr int R      = (NCOLS == 1) ? 4 : 2;
    constexpr int UNROLL = 2;
    const int64_t ngroups = (w.M + R - 1) / R;
    int64_t blocks = (ngroups + 3) / 4;
    if (blocks > max_blocks) blocks = max_blocks;
    if (blocks < 1) blocks = 1;
    const size_t lds = fq_gemv_lds_bytes(fq_desc(TYPE).act_type, w.K, NCOLS);
    if (lds > 64 * 1024) {      // 180B ffn_down: K = 59392
        static size_t granted = 0;
        if (lds > granted) { HIP_CHECK(hipFuncSetAttribute((const void *) k_gemv<TYPE, R, NCOLS, UNROLL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds)); granted = lds; }
    }
    hipLaunchKernelGGL((k_gemv<TYPE, R, NCOLS, UNROLL>), dim3((unsigned) blocks), dim3(256), lds, st, w, act, dst, ldd, ep);
}

template <int TYPE>
static void launch_gemv_n(const fq_weight & w, const fq_act & act, int ncols, float * dst, int64_t ldd, const fq_gemv_epi & ep, int max_blocks, hipStream_t st) {
    switch (ncols) {
        case 1: launch_gemv_t<TYPE, 1>(w, act, dst, ldd, ep, max_blocks, st); break;
        case 2: launch_gemv_t<TYPE, 2>(w, act, dst, ldd, ep, max_blocks, st); break;
        case 4: launch_gemv_t<TYPE, 4>(w, act, dst, ldd, ep, max_blocks, st); break;
        default: fprintf(stderr, "ggml-hip: gemv: unsupported column count %d\n", ncols); exit(1);
    }
}

// dst[c*ldd + row] for columns [0, ncols); ncols in {1,2,4} (the caller splits other counts)
void fq_launch_gemv(const fq_weight & w, const fq_act & act, int ncols, float * dst, int64_t ldd, const fq_gemv_epi & ep, int max_blocks, hipStream_t st) {
    FQ_TL(st, "gemv");
#define FQ_CASE(T) case T: launch_gemv_n<T>(w, act, ncols, dst, ldd, ep, max_blocks, st); break;
    switch (w.type) {
        FQ_CASE(FQ_Q4_0) FQ_CASE(FQ_Q4_1) FQ_CASE(FQ_Q5_0) FQ_CASE(FQ_Q5_1) FQ_CASE(FQ_Q8_0)
        FQ_CASE(FQ_Q2_K) FQ_CASE(FQ_Q3_K) FQ_CASE(FQ_Q4_K) FQ_CASE(FQ_Q5_K) FQ_CASE(FQ_Q6_K)
        default: fprintf(stderr, "ggml-hip: gemv: unsupported weight type %d\n", w.type); exit(1);
    }
#undef FQ_CASE
}
