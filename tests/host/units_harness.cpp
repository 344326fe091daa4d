// Host-side check of ggllm.cpp_amd/csrc/fq_units.h (the exact header the GEMV kernels compile): re-tile one ggml
// row into planes, re-tile ggml activation blocks into the SoA layout, and sum the per-unit dots.
#include "fq_units.h"
#include "fq_kdot.h"
#include <vector>
#include <cstring>
#include <cstdlib>

template <int TYPE>
static float row_dot(const fq_weight & w, const fq_actcol & a, int64_t K) {
    const fq_wrow r = fq_row<TYPE>(w, 0);
    const int units = (int)(K / fq_unit<TYPE>::ELEMS);
    float acc = 0.0f;
    for (int u = 0; u < units; ++u) acc += fq_unit<TYPE>::dot(fq_unit_load<TYPE>(r, u), a, u);
    return acc;
}

// fq_kdot.h (the ring consumers' restatement of the k-quant unit dots) against fq_unit<TYPE>::dot, unit by unit: number of units whose f32
// term differs in any bit (rows of whole columns only: K a multiple of the column's elements)
template <int TYPE>
static int kdot_mismatches(const fq_weight & w, const fq_actcol & a, int64_t K) {
    typedef fq_kdot<TYPE> KD;
    const fq_wrow r = fq_row<TYPE>(w, 0);
    const int units = (int)(K / fq_unit<TYPE>::ELEMS), npass = units / 64;
    constexpr int COLB = fq_lay<TYPE>::CB * fq_lay<TYPE>::TS;
    int bad = 0;
    for (int p = 0; p < npass; ++p)
        for (int ju = 0; ju < 64; ++ju) {
            const typename KD::lane_t L = KD::lane_init(ju);
            const float got = KD::dot(KD::w_load(r.p0 + (size_t) p * COLB, L), KD::act_load(a, p, L), L);
            const int u = 64 * p + ju;
            const float exp = fq_unit<TYPE>::dot(fq_unit_load<TYPE>(r, u), a, u);
            if (memcmp(&got, &exp, 4) != 0) ++bad;
        }
    return bad;
}
template <int TYPE> static float kdot_or_row(int mode, const fq_weight & w, const fq_actcol & a, int64_t K) {
    if (mode == 1) { if constexpr (fq_kdot<TYPE>::ok) return (float) kdot_mismatches<TYPE>(w, a, K); else return -1.0f; }
    return row_dot<TYPE>(w, a, K);
}

static float units_entry(int mode, int type, int64_t K, const uint8_t * row, const uint8_t * act_blocks);
extern "C" float units_row_dot(int type, int64_t K, const uint8_t * row, const uint8_t * act_blocks) { return units_entry(0, type, K, row, act_blocks); }
extern "C" int units_kdot_mismatches(int type, int64_t K, const uint8_t * row, const uint8_t * act_blocks) { return (int) units_entry(1, type, K, row, act_blocks); }
static float units_entry(int mode, int type, int64_t K, const uint8_t * row, const uint8_t * act_blocks) {
    const fq_type_desc d = fq_desc(type);
    const int64_t nblk = K / d.blck;
    fq_weight w{}; w.type = type; w.K = K; w.M = 1; w.nblk = nblk;
    std::vector<uint8_t> il;
    w.row_stride = fq_il_row_stride(d, nblk);                   // column-interleaved row (fq_types.h)
    il.resize(w.row_stride + 32);
    uint8_t * base = (uint8_t *)(((uintptr_t) il.data() + 15) & ~(uintptr_t) 15);
    for (int p = 0; p < d.nplanes; ++p)
        for (int64_t b = 0; b < nblk; ++b)
            memcpy(base + fq_il_offset(d, p, nblk, b), row + (size_t) b * d.tsize + d.plane[p].src_off, d.plane[p].bytes);
    for (int p = 0; p < d.nplanes; ++p) w.plane[p] = base;
    // activations: ggml blocks -> SoA
    std::vector<int8_t> qs((size_t) K + 64);
    int8_t * qsa = (int8_t *)(((uintptr_t) qs.data() + 15) & ~(uintptr_t) 15);
    std::vector<float> dd((size_t) K / 32 + 1);
    std::vector<uint8_t> aux((size_t) K / 16 * 4 + 16);
    if (d.act_type == FQ_Q8_0) {
        for (int64_t b = 0; b < K / 32; ++b) {
            const uint8_t * s = act_blocks + b * 34; uint16_t h; memcpy(&h, s, 2);
            dd[b] = fq_h2f(h); memcpy(qsa + b * 32, s + 2, 32);
            int t = 0; for (int j = 0; j < 32; ++j) t += (int8_t) s[2 + j];
            ((int32_t *) aux.data())[b] = t;
        }
    } else if (d.act_type == FQ_Q8_1) {
        for (int64_t b = 0; b < K / 32; ++b) {
            const uint8_t * s = act_blocks + b * 40;
            memcpy(&dd[b], s, 4); memcpy((float *) aux.data() + b, s + 4, 4); memcpy(qsa + b * 32, s + 8, 32);
        }
    } else {
        for (int64_t b = 0; b < K / 256; ++b) {
            const uint8_t * s = act_blocks + b * 292;
            memcpy(&dd[b], s, 4); memcpy(qsa + b * 256, s + 4, 256); memcpy((int16_t *) aux.data() + b * 16, s + 260, 32);
        }
    }
    fq_actcol a{ qsa, dd.data(), aux.data() };
    switch (type) {
        case FQ_Q4_0: return kdot_or_row<FQ_Q4_0>(mode, w, a, K); case FQ_Q4_1: return kdot_or_row<FQ_Q4_1>(mode, w, a, K);
        case FQ_Q5_0: return kdot_or_row<FQ_Q5_0>(mode, w, a, K); case FQ_Q5_1: return kdot_or_row<FQ_Q5_1>(mode, w, a, K);
        case FQ_Q8_0: return kdot_or_row<FQ_Q8_0>(mode, w, a, K); case FQ_Q2_K: return kdot_or_row<FQ_Q2_K>(mode, w, a, K);
        case FQ_Q3_K: return kdot_or_row<FQ_Q3_K>(mode, w, a, K); case FQ_Q4_K: return kdot_or_row<FQ_Q4_K>(mode, w, a, K);
        case FQ_Q5_K: return kdot_or_row<FQ_Q5_K>(mode, w, a, K); case FQ_Q6_K: return kdot_or_row<FQ_Q6_K>(mode, w, a, K);
    }
    abort();
}
