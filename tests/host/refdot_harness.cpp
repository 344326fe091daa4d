// Host-side check of ggllm.cpp_amd/csrc/fq_ref_dot.h (the header kernels_ref.hip compiles): re-tile one ggml row into
// the device layout, ggml activation blocks into the activation image, run the reference-order row dot.
#include "fq_ref_dot.h"
#include <vector>
#include <cstring>
#include <cstdlib>

extern "C" float refdot_row(int type, int64_t K, const uint8_t * row, const uint8_t * act_blocks) {
    const fq_type_desc d = fq_desc(type);
    const int64_t nblk = K / d.blck;
    std::vector<uint8_t> il(fq_il_row_stride(d, nblk) + 16);
    for (int p = 0; p < d.nplanes; ++p)
        for (int64_t b = 0; b < nblk; ++b)
            memcpy(il.data() + fq_il_offset(d, p, nblk, b), row + (size_t) b * d.tsize + d.plane[p].src_off, d.plane[p].bytes);
    std::vector<uint8_t> img(fq_act_col_bytes(d.act_type, K) + 16);
    int8_t * qs = (int8_t *) img.data();
    uint8_t * dd = img.data() + fq_act_d_off(d.act_type, K), * aux = img.data() + fq_act_aux_off(d.act_type, K);
    if (d.act_type == FQ_Q8_0) {
        for (int64_t b = 0; b < K / 32; ++b) {
            const uint8_t * s = act_blocks + b * 34;
            const float f = fq_ref_h2f(s); memcpy(dd + 4 * b, &f, 4); memcpy(qs + b * 32, s + 2, 32);
        }
    } else if (d.act_type == FQ_Q8_1) {
        for (int64_t b = 0; b < K / 32; ++b) {
            const uint8_t * s = act_blocks + b * 40;
            memcpy(dd + 4 * b, s, 4); memcpy(aux + 4 * b, s + 4, 4); memcpy(qs + b * 32, s + 8, 32);
        }
    } else {
        for (int64_t b = 0; b < K / 256; ++b) {
            const uint8_t * s = act_blocks + b * 292;
            memcpy(dd + 4 * b, s, 4); memcpy(qs + b * 256, s + 4, 256); memcpy(aux + 32 * b, s + 260, 32);
        }
    }
    const fq_ref_act a = { qs, dd, aux };
    switch (type) {
        case FQ_Q4_0: return fq_ref_row_dot<FQ_Q4_0>(il.data(), nblk, a); case FQ_Q4_1: return fq_ref_row_dot<FQ_Q4_1>(il.data(), nblk, a);
        case FQ_Q5_0: return fq_ref_row_dot<FQ_Q5_0>(il.data(), nblk, a); case FQ_Q5_1: return fq_ref_row_dot<FQ_Q5_1>(il.data(), nblk, a);
        case FQ_Q8_0: return fq_ref_row_dot<FQ_Q8_0>(il.data(), nblk, a); case FQ_Q2_K: return fq_ref_row_dot<FQ_Q2_K>(il.data(), nblk, a);
        case FQ_Q3_K: return fq_ref_row_dot<FQ_Q3_K>(il.data(), nblk, a); case FQ_Q4_K: return fq_ref_row_dot<FQ_Q4_K>(il.data(), nblk, a);
        case FQ_Q5_K: return fq_ref_row_dot<FQ_Q5_K>(il.data(), nblk, a); case FQ_Q6_K: return fq_ref_row_dot<FQ_Q6_K>(il.data(), nblk, a);
    }
    abort();
}
