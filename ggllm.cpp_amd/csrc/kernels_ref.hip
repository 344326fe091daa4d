// kernels_ref.hip -- quantized mat-mul in the reference's own summation order (fq_ref_dot.h), any number of columns.
// Selected by ggml_hip_reference_order(1); one thread per (weight row, activation column), no tiling, no MFMA: a parity
// instrument, not a product path. Replaces ggml_compute_forward_mul_mat_q_f32's COMPUTE phase (ggml.c:11484-11516) with
// the scalar branches of ggml_vec_dot_q*_q8_* restated per thread.
#include "fq_device.h"
#include "fq_ref_dot.h"
#include "kernels.h"

template <int TYPE>
__global__ void __launch_bounds__(256) k_mul_mat_ref(fq_weight w, fq_act act, int64_t N, float * dst, int64_t ldd, fq_gemv_epi ep) {
    constexpr int ACT = fq_act_of(TYPE);
    const int64_t idx = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= w.M * N) return;
    const int64_t row = idx % w.M, col = idx / w.M;
    const uint8_t * img = act.base + (size_t) col * fq_act_col_bytes(ACT, w.K);
    const fq_ref_act a = { (const int8_t *) img, img + fq_act_d_off(ACT, w.K), img + fq_act_aux_off(ACT, w.K) };
    float v = fq_ref_row_dot<TYPE>(w.plane[0] + (size_t) row * w.row_stride, w.nblk, a);
    if (ep.mode == FQ_EPI_GELU) {
        v = h2f_bits(ep.gelu_table[f2h_bits(v)]);                                          // ggml.c:3477-3484
    } else if (ep.mode == FQ_EPI_ADD2) {
        v = (v + ep.add1[col * ep.ld_add + row]) + ep.add2[col * ep.ld_add + row];         // libfalcon.cpp:2399-2400
    }
    dst[col * ldd + row] = v;
}

void fq_launch_mul_mat_ref(const fq_weight & w, const fq_act & act, int64_t N, float * dst, int64_t ldd, const fq_gemv_epi & ep, hipStream_t st) {
    FQ_TL(st, "mul_mat_ref");
    const int64_t total = w.M * N;
    const unsigned blocks = (unsigned)((total + 255) / 256);
#define FQ_CASE(T) case T: hipLaunchKernelGGL((k_mul_mat_ref<T>), dim3(blocks), dim3(256), 0, st, w, act, N, dst, ldd, ep); break;
    switch (w.type) {
        FQ_CASE(FQ_Q4_0) FQ_CASE(FQ_Q4_1) FQ_CASE(FQ_Q5_0) FQ_CASE(FQ_Q5_1) FQ_CASE(FQ_Q8_0)
        FQ_CASE(FQ_Q2_K) FQ_CASE(FQ_Q3_K) FQ_CASE(FQ_Q4_K) FQ_CASE(FQ_Q5_K) FQ_CASE(FQ_Q6_K)
        default: fprintf(stderr, "ggml-hip: mul_mat (reference order): unsupported weight type %d\n", w.type); exit(1);
    }
#undef FQ_CASE
}
