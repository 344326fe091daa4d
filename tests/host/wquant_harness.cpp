// Host build of ggllm.cpp_amd/csrc/fq_wquant.h (the arithmetic the device weight quantizers run), driven exactly like
// kernels_wquant.hip drives it: fit per sub-block -> header per super-block -> requantize -> pack bytes.
#include "fq_wquant.h"
#include <cstdlib>

template <int TYPE>
static void legacy_rows(const float * x, int64_t nblocks, uint8_t * out, int64_t * hist) {
    constexpr int TS = fq_desc(TYPE).tsize;
    for (int64_t b = 0; b < nblocks; ++b) {
        float v[32];
        for (int i = 0; i < 32; ++i) v[i] = x[32 * b + i];
        wq_block_legacy<TYPE>(v, out + b * TS, [&](int bin) { if (hist) hist[bin]++; });
    }
}

template <int TYPE>
static void k_rows(const float * x, int64_t nsb, uint8_t * out) {
    constexpr int N = wq_geom<TYPE>::N, NSB = wq_geom<TYPE>::NSB, TS = fq_desc(TYPE).tsize;
    for (int64_t sb = 0; sb < nsb; ++sb) {
        float v[NSB][N]; int L[NSB][N]; float scales[NSB], mins[NSB];
        for (int j = 0; j < NSB; ++j) {
            for (int i = 0; i < N; ++i) v[j][i] = x[256 * sb + N * j + i];
            wq_fit<TYPE>(v[j], L[j], scales[j], mins[j]);
        }
        uint8_t hdr[20], Lb[256];
        wq_header<TYPE>(scales, mins, hdr);
        for (int j = 0; j < NSB; ++j) {
            wq_requant<TYPE>(hdr, j, v[j], L[j]);
            for (int i = 0; i < N; ++i) Lb[N * j + i] = (uint8_t) L[j][i];
        }
        for (int i = 0; i < TS; ++i) out[sb * TS + i] = wq_pack_byte<TYPE>(hdr, Lb, i);
    }
}

extern "C" int wquant_rows(int type, const float * x, int64_t n_elems, uint8_t * out, int64_t * hist) {
    switch (type) {
        case FQ_Q4_0: legacy_rows<FQ_Q4_0>(x, n_elems / 32, out, hist); return 0;
        case FQ_Q4_1: legacy_rows<FQ_Q4_1>(x, n_elems / 32, out, hist); return 0;
        case FQ_Q5_0: legacy_rows<FQ_Q5_0>(x, n_elems / 32, out, hist); return 0;
        case FQ_Q5_1: legacy_rows<FQ_Q5_1>(x, n_elems / 32, out, hist); return 0;
        case FQ_Q8_0: legacy_rows<FQ_Q8_0>(x, n_elems / 32, out, hist); return 0;
        case FQ_Q2_K: k_rows<FQ_Q2_K>(x, n_elems / 256, out); return 0;
        case FQ_Q3_K: k_rows<FQ_Q3_K>(x, n_elems / 256, out); return 0;
        case FQ_Q4_K: k_rows<FQ_Q4_K>(x, n_elems / 256, out); return 0;
        case FQ_Q5_K: k_rows<FQ_Q5_K>(x, n_elems / 256, out); return 0;
        case FQ_Q6_K: k_rows<FQ_Q6_K>(x, n_elems / 256, out); return 0;
    }
    return -1;
}
extern "C" uint16_t wquant_f2h(float f) { return fq_f2h(f); }
