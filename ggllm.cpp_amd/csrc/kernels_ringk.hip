// kernels_ringk.hip -- ring forms of the fused decode launches beyond kernels_ring.hip's scope (gfx950, wave64):
//
//   k_ring_ln_k<T>   k_gemv_ln's launch for the K-QUANTS (Q2_K .. Q6_K; Falcon-40B / 180B widths): LayerNorm(s) of the residual row, the Q8_K
//                    image(s) (quantize_row_q8_K_reference, k_quants.c:899-934), rows [Wup | Wqkv] dotted with ggml_vec_dot_q*_K_q8_K's
//                    arithmetic (k_quants.c:1267-1306, 1684-1746, 1999-2055, 2340-2400, 2748-2789; fq_units.h), GELU through the fp16 table
//                    (ggml.c:3477-3484) stored as f32 -- its Q8_K image needs 256-element maxima, i.e. another launch's workgroups: it rides on the
//                    attention launch (k_attn_decode_seqs' rider). One workgroup per CU: an LDS-DMA loader wave streams the workgroup's rows
//                    THROUGH the LayerNorm + Q8_K prologue (~10 us with nothing streaming in the register-streaming k_gemv_ln at this width),
//                    an epilogue wave, ten consumer waves; the rows are dealt evenly over the CUs (k_gemv_ln: 438 workgroups = 1.71 rounds).
//   k_ring_out<T>    k_gemv_out's launch, x = (Wdown . q8(gelu(up)) + Wo . q8(att)) + x (libfalcon.cpp:2394-2400), for all ten formats: the two
//                    images staged (k-quants: the attention row quantized to Q8_K by the helper waves, under the stream), Wdown's rows then Wo's
//                    rows out of the ring, eleven consumer waves.
//
// Same bits as k_gemv_ln / k_gemv_out (per lane ascending units, wave butterfly, the same epilogue expressions): tests/test_gpu_ringk.py.
// The CUDA twins these replace: dequantize_mul_mat_vec_q*_k (ggml-cuda.cu:475-845). false from a launcher = outside its scope, nothing launched.
#include "fq_block_dev.h"
#include "kernels.h"
#include "hip_context.h"
#include <hip/hip_ext.h>
#include <vector>
#include "fq_ring_dev.h"

namespace {

constexpr int KNH = 11, KNC = KNH - 1, KNT = 64 * (KNH + 1), KHT = 64 * KNH;      // helper waves (epilogue wave + consumers), threads, helper threads
constexpr int KNSLOT = 6;                                                         // 96 KiB of ring: the images of a Falcon-40B block need the rest
constexpr unsigned KCTL_BYTES = eng_ctl::PTRS + 64;
#define RINGK_T(role, slot) do { if (a.dbg && lane == 0) a.dbg[((size_t)(role) * 256 + blockIdx.x) * 8 + (slot)] = (long long) wall_clock64(); } while (0)

}   // namespace

struct fq_ringk_ln_args {
    const float * x; int E, FF, nblkE; unsigned rsE;
    const uint8_t * qkv, * up; int qkv_rows;
    const float * ln_w, * ln_b, * ln2_w, * ln2_b; int two_norms;      // ln feeds Wup (and Wqkv of a one-norm block), ln2 = the attention norm
    float * qkv_dst, * up_dst;                                         // f32 outputs (up_dst: gelu(Wup . a))
    const uint16_t * gelu_tab;
    const fq_engine_sched * sched;                                     // per workgroup: rows [qg0, qg1) of Wqkv, 32-row groups [ug0, ug1) of Wup
    unsigned * epoch_word; const int * n_past_ptr; const float * rope_cs; float * rope_cur;
    unsigned * err; int debug_mode; long long * dbg;
};

// (round 6, as k_gemv_ln_ring: the first-needed arguments as leading scalars, preloaded into SGPRs by the command processor -- -amdgpu-kernarg-preload-count=12 for this file)
template <int TYPE>
__global__ void __launch_bounds__(KNT) k_ring_ln_k(const float * p_x, const fq_engine_sched * p_sched, const uint8_t * p_qkv, const uint8_t * p_up, int p_E, int p_FF, int p_nblkE, unsigned p_rsE,
                                                   fq_ringk_ln_args a) {
    a.x = p_x; a.sched = p_sched; a.qkv = p_qkv; a.up = p_up; a.E = p_E; a.FF = p_FF; a.nblkE = p_nblkE; a.rsE = p_rsE;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr int ACT = FQ_Q8_K;
    constexpr int RING = KNSLOT * ENG_SLOT;
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int E = a.E;
    fq_engine_sched sc = a.sched[blockIdx.x];                              // (wave-uniform, and made provably so: the loader's control flow and DMA operands depend on it)
    sc.qg0 = __builtin_amdgcn_readfirstlane(sc.qg0); sc.qg1 = __builtin_amdgcn_readfirstlane(sc.qg1);
    sc.ug0 = __builtin_amdgcn_readfirstlane(sc.ug0); sc.ug1 = __builtin_amdgcn_readfirstlane(sc.ug1);
    uint8_t * ring  = smem;
    float   * xrow  = (float *)(smem + RING + ENG_MIRROR);
    uint8_t * img_e = (uint8_t *) xrow + (((size_t) E * 4 + 15) & ~(size_t) 15);          // LN image feeding Wup (and Wqkv with one norm)
    uint8_t * img_e2 = img_e + fq_act_col_bytes(ACT, E);                                   // attention-norm image of a two-norm block
    uint8_t * ctlp  = img_e + (a.two_norms ? 2 : 1) * fq_act_col_bytes(ACT, E);
    const unsigned ctl = (unsigned)(uintptr_t) ctlp;
    auto ldsf_st = [&](unsigned off, float v) { lds_st(ctl + off, __builtin_bit_cast(unsigned, v)); };
    auto ldsf_ld = [&](unsigned off) { return __builtin_bit_cast(float, lds_ld(ctl + off)); };

    const unsigned rsE = a.rsE;
    const int nA1 = sc.qg1 - sc.qg0, nA2 = 32 * (sc.ug1 - sc.ug0);
    const unsigned pA1 = ring_pad1k((unsigned) nA1 * rsE), pA2 = ring_pad1k((unsigned) nA2 * rsE);

    // the hand-off tag of the launch that follows, and the rope table's row of this position (as k_gemv_ln)
    if (a.epoch_word && blockIdx.x == 0 && tid == 0) { const unsigned e = *a.epoch_word + 1u; *a.epoch_word = e ? e : 1u; }
    if (a.rope_cur && blockIdx.x == gridDim.x - 1 && tid >= 64 && tid < 128) a.rope_cur[tid - 64] = a.rope_cs[(int64_t)(*a.n_past_ptr) * 64 + (tid - 64)];
    if (tid < 32) lds_st(ctl + eng_ctl::CNT + 4 * tid, 0u);
    if (tid < 16) lds_st(ctl + eng_ctl::LOW + 4 * tid, tid < KNC ? 0u : 0xFFFFFFFFu);
    if (tid < 16) lds_st(ctl + eng_ctl::XG_DONE + 4 * tid, 0u);
    if (tid == 0) lds_st(ctl + eng_ctl::LANDED, 0u);
    if (a.dbg && tid == 0) a.dbg[(size_t) blockIdx.x * 8] = (long long) wall_clock64();
    __syncthreads();                                                       // the only workgroup barrier: before the roles split

    if (wid == 0) {
        // ================================================================================ loader
        ring_loader<KNSLOT> ld(ring, ctl, a.err, lane);
        ld.nospace = (a.debug_mode & 128) != 0;
        if (!(a.debug_mode & 1)) ld.w.until(ctl + RING_XISSUED, (unsigned) KNH, ENG_W_XG);     // the residual row's loads go first
        RINGK_T(0, 1);
        // Wup first: its rows end in the epilogue wave's GELU work, which then overlaps the Wqkv rows instead of trailing the launch
        ld.seg(a.up + (size_t) sc.ug0 * 32 * rsE, pA2);                   // (an empty segment is zero pieces)
        ld.seg(a.qkv + (size_t) sc.qg0 * rsE, pA1);
        RINGK_T(0, 2);
        ld.finish();
        RINGK_T(0, 3);
        return;
    }

    // ==================================================================================== helpers: the epilogue wave (h = 0) and the consumers (c = h - 1)
    const int h = wid - 1, c = h - 1, ht = tid - 64;
    const bool isG = h == 0;
    constexpr int NLN = 3;                                                 // float4 of the row per helper thread: n_embd <= 8448
    eng_wait w{ a.err, false, nullptr, 0 };
    const int nv = E >> 2;
    const unsigned nx = (unsigned)((E + ENG_CHUNK - 1) / ENG_CHUNK);
    const bool nodots = (a.debug_mode & 2) != 0;

    // ---- the residual row -> LDS (chunks of 1024 values, one per helper wave, each with its f64 partial sum) -> statistics -> Q8_K image(s)
    // (the one-pass LayerNorm of kernels_ring.hip: ggml.c:10577-10591 with the sums in f64)
    ln_row_regs<NLN> wr, br, w2r, b2r;
    float4 xv[NLN];                                                        // the lane's elements of the row: x - mean, later (x - mean) * scale
    {
        unsigned own = 0;
        float v[16]; int kown = -1;
        if (h < (int) nx) {                                                // (nx <= 11 for n_embd <= 11264: at most one chunk per helper)
            kown = h;
            const int base = kown * ENG_CHUNK;
#pragma unroll
            for (int j = 0; j < 16; ++j) { const int i = base + 64 * j + lane; v[j] = a.x[i < E ? i : E - 1]; }
        }
        ln_regs_issue_wb(a.ln_w, a.ln_b, E, KHT, wr, br, ht);             // (behind the row: the statistics wait for the row only)
        if (a.two_norms) ln_regs_issue_wb(a.ln2_w, a.ln2_b, E, KHT, w2r, b2r, ht); else { w2r = wr; b2r = br; }
        if (lane == 0) lds_add(ctl + RING_XISSUED, 1u);                    // the loader may start: the prologue's requests are in the queue ahead of its own
        if (kown >= 0) {
            const int base = kown * ENG_CHUNK;
            double s = 0.0;
#pragma unroll
            for (int j = 0; j < 16; ++j) { const int i = base + 64 * j + lane; if (i < E) { xrow[i] = v[j]; s += (double) v[j]; } }
            s = wave_sum(s);
            if (lane == 0) lds_st64(ctl + eng_ctl::PSUM + 8u * (unsigned) kown, (unsigned long long) __builtin_bit_cast(long long, s));
            own = 1;
        }
        if (own) {
            lds_drain();
            unsigned old = 0;
            if (lane == 0) old = lds_add_rtn(ctl + eng_ctl::XG_DONE, own);
            old = __builtin_amdgcn_readfirstlane(old);
            if (old + own == nx) {                                         // the last chunk: the row's mean, chunk sums in chunk order
                double s = 0.0;
                const unsigned long long pk = lds_ld64(ctl + eng_ctl::PSUM + 8u * (unsigned)(lane < (int) nx ? lane : 0));
                for (unsigned k = 0; k < nx; ++k) s += lane_get(__builtin_bit_cast(double, (long long) pk), (int) k);
                const float mean = (float)(s / (double) E);
                if (lane == 0) ldsf_st(eng_ctl::STAT, mean);
                lds_drain();
                if (lane == 0) lds_st(ctl + eng_ctl::LN_MEAN, 1u);
            }
        }
        w.until(ctl + eng_ctl::LN_MEAN, 1u, ENG_W_STAT);
        if (h <= 1) RINGK_T(1 + h, 1);
        const float mean = ldsf_ld(eng_ctl::STAT);
        double s2 = 0.0;
#pragma unroll
        for (int k = 0; k < NLN; ++k) {
            const int q4 = k * KHT + ht;
            float4 t = ((const float4 *) xrow)[q4 < nv ? q4 : nv - 1];
            t.x -= mean; t.y -= mean; t.z -= mean; t.w -= mean;
            xv[k] = t;
            if (q4 < nv) { s2 += (double)(t.x * t.x); s2 += (double)(t.y * t.y); s2 += (double)(t.z * t.z); s2 += (double)(t.w * t.w); }
        }
        s2 = wave_sum(s2);
        if (lane == 0) lds_st64(ctl + eng_ctl::PSQ + 8u * (unsigned) h, (unsigned long long) __builtin_bit_cast(long long, s2));
        lds_drain();
        unsigned old2 = 0;
        if (lane == 0) old2 = lds_add_rtn(ctl + eng_ctl::S2_DONE, 1u);
        old2 = __builtin_amdgcn_readfirstlane(old2);
        if (old2 + 1u == (unsigned) KNH) {                              // the last wave: partial sums in wave order -> scale
            const unsigned long long pk = lds_ld64(ctl + eng_ctl::PSQ + 8u * (unsigned)(lane < KNH ? lane : 0));
            double t = 0.0;
#pragma unroll
            for (int k = 0; k < KNH; ++k) t += lane_get(__builtin_bit_cast(double, (long long) pk), k);
            const float variance = (float)(t / (double) E);
            const float scale = 1.0f / sqrtf(variance + 1e-5f);
            if (lane == 0) ldsf_st(eng_ctl::STAT + 4, scale);
            lds_drain();
            if (lane == 0) lds_st(ctl + eng_ctl::LN_STAT, 1u);
        }
        w.until(ctl + eng_ctl::LN_STAT, 1u, ENG_W_STAT);
        const float scale = ldsf_ld(eng_ctl::STAT + 4);
        // a helper wave holds whole super-blocks: float4 number q4 = k * 704 + 64 h + lane -> super-block k * 11 + h, elements 4 lane .. 4 lane + 3.
        // Only the image the FIRST rows of the stream need (Wup's) is written here; the attention norm's image of a two-norm block -- needed once the
        // Wqkv rows arrive, the last fifth of the stream -- is written by norm2_image() below, after the consumers have started freeing the ring (the
        // ring holds 3.8 us of stream: with both images in the prologue it filled up and the stream stalled for ~2 us)
#pragma unroll
        for (int k = 0; k < NLN; ++k) { float4 t = xv[k]; t.x *= scale; t.y *= scale; t.z *= scale; t.w *= scale; xv[k] = t; }
        {
            const act_image_ptr o1 = act_image_at(img_e, ACT, E);
            float4 qv[NLN]; int64_t qsb[NLN]; act_image_ptr qo[NLN];
#pragma unroll
            for (int k = 0; k < NLN; ++k) {
                const int q4 = k * KHT + ht;
                const float4 ww = wr.t[k], bb = br.t[k];
                float4 u = xv[k];
                u.x = u.x * ww.x + bb.x; u.y = u.y * ww.y + bb.y; u.z = u.z * ww.z + bb.z; u.w = u.w * ww.w + bb.w;
                qv[k] = u; qsb[k] = (q4 >> 6) < (E >> 8) ? (q4 >> 6) : -1; qo[k] = o1;      // (wave-uniform)
            }
            quant_q8K_wave_n<NLN>(qv, lane, qsb, qo);
        }
        lds_drain();
        if (lane == 0) lds_add(ctl + eng_ctl::IMG_DONE, 1u);
        w.until(ctl + eng_ctl::IMG_DONE, (unsigned) KNH, ENG_W_IMG);
        if (h <= 1) RINGK_T(1 + h, 2);
        if (isG && a.two_norms) {                                          // the epilogue wave has nothing to do until the first 32-row groups are complete
            const act_image_ptr o2 = act_image_at(img_e2, ACT, E);
            float4 qv[NLN]; int64_t qsb[NLN]; act_image_ptr qo[NLN];
#pragma unroll
            for (int k = 0; k < NLN; ++k) {
                const int q4 = k * KHT + ht;
                const float4 w2 = w2r.t[k], b2 = b2r.t[k];
                float4 u = xv[k];
                u.x = u.x * w2.x + b2.x; u.y = u.y * w2.y + b2.y; u.z = u.z * w2.z + b2.z; u.w = u.w * w2.w + b2.w;
                qv[k] = u; qsb[k] = (q4 >> 6) < (E >> 8) ? (q4 >> 6) : -1; qo[k] = o2;
            }
            quant_q8K_wave_n<NLN>(qv, lane, qsb, qo);
            lds_drain();
            if (lane == 0) lds_add(ctl + eng_ctl::FG_DONE, 1u);
        }
    }
    // a consumer's share of the attention norm's image: written after its first trip of Wup rows (the ring's backlog from the prologue is being worked off by
    // the other consumers then), so that nobody waits for it when the Wqkv rows arrive; norm2_wait: all eleven shares are there
    auto norm2_wait = [&]() { if (a.two_norms) w.until(ctl + eng_ctl::FG_DONE, (unsigned) KNH, ENG_W_IMG); };
    auto norm2_image = [&]() {
        if (!a.two_norms) return;
        const act_image_ptr o2 = act_image_at(img_e2, ACT, E);
        float4 qv[NLN]; int64_t qsb[NLN]; act_image_ptr qo[NLN];
#pragma unroll
        for (int k = 0; k < NLN; ++k) {
            const int q4 = k * KHT + ht;
            const float4 w2 = w2r.t[k], b2 = b2r.t[k];
            float4 u = xv[k];
            u.x = u.x * w2.x + b2.x; u.y = u.y * w2.y + b2.y; u.z = u.z * w2.z + b2.z; u.w = u.w * w2.w + b2.w;
            qv[k] = u; qsb[k] = (q4 >> 6) < (E >> 8) ? (q4 >> 6) : -1; qo[k] = o2;
        }
        quant_q8K_wave_n<NLN>(qv, lane, qsb, qo);
        lds_drain();
        if (lane == 0) lds_add(ctl + eng_ctl::FG_DONE, 1u);
    };

    const fq_actcol col_e  = { (const int8_t *) img_e,  (const float *)(img_e + fq_act_d_off(ACT, E)),  (const void *)(img_e + fq_act_aux_off(ACT, E)) };
    const fq_actcol col_e2 = { (const int8_t *) img_e2, (const float *)(img_e2 + fq_act_d_off(ACT, E)), (const void *)(img_e2 + fq_act_aux_off(ACT, E)) };
    const int gA = nA2 / 32;

    if (isG) {
        // ---- the Wup epilogues of this workgroup's 32-row groups as they complete: GELU through the fp16 table, f32 store (k_gemv_ln's GELU_STORE)
        for (int gl = 0; gl < gA;) {
            for (unsigned spins = 0; (int)(lds_ld_u(ctl + eng_ctl::CNT + 4 * gl) - 32u) < 0;) { if (!w.spin(spins, ENG_W_GROUP, (unsigned) gl, 0)) break; __builtin_amdgcn_s_sleep(1); }
            const bool two = gl + 1 < gA && (int)(lds_ld_u(ctl + eng_ctl::CNT + 4 * (gl + 1)) - 32u) >= 0;
            const int j = lane & 31, half = lane >> 5;
            const int myg = gl + (two ? half : 0);
            float v = ldsf_ld(eng_ctl::OUT + 4 * (32 * myg + j));
            v = h2f_bits(a.gelu_tab[f2h_bits(v)]);                         // ggml.c:3477-3484
            if (two || half == 0) a.up_dst[32 * (sc.ug0 + myg) + j] = v;
            gl += two ? 2 : 1;
        }
        RINGK_T(1, 3);
        return;
    }

    // ---- consumers: rows out of the ring, round-robin
    const int nblkE = a.nblkE;
    auto sink_up  = [&](int i, float v) { if (lane == 0) { ldsf_st(eng_ctl::OUT + 4 * i, v); lds_add(ctl + eng_ctl::CNT + 4 * (i >> 5), 1u); } };
    auto sink_qkv = [&](int i, float v) { if (lane == 0) a.qkv_dst[sc.qg0 + i] = v; };
    const fq_actcol & col_q = a.two_norms ? col_e2 : col_e;
    // rows of KU whole columns (Falcon-40B's 8192: 4 for the 32-element units, 2 for Q2_K / Q3_K): the lane's activation slices stay in registers and the
    // unit dots are fq_kdot.h's (same f32 terms); any other width takes the generic loop
    constexpr int KU = fq_unit<TYPE>::ELEMS == 64 ? 2 : 4;
    if (nblkE == KU * fq_lay<TYPE>::CB && !(a.debug_mode & 32)) {
        const typename fq_kdot<TYPE>::lane_t L = fq_kdot<TYPE>::lane_init(lane);
        ring_kacts<TYPE, KU> A;
        ring_kacts_load<TYPE, KU>(col_e, 0, L, A);
        constexpr int KR = KU == 2 ? 2 : 1;                                 // rows per trip
        ring_rows_k<TYPE, RING, KU, KR>(ring, ctl, c, KNC, 0u, pA2, nA2, rsE, A, L, lane, w, nodots, sink_up, norm2_image);
        if (c == 0 || c == 9) RINGK_T(c == 0 ? 2 : 3, 3);
        norm2_wait();
        if (a.two_norms) ring_kacts_load<TYPE, KU>(col_e2, 0, L, A);
        ring_rows_k<TYPE, RING, KU, KR>(ring, ctl, c, KNC, pA2, pA1, nA1, rsE, A, L, lane, w, nodots, sink_qkv);
        if (c == 0 || c == 9) RINGK_T(c == 0 ? 2 : 3, 4);
        return;
    }
    const bool r2 = (unsigned)(2 * KNC) * rsE * 2u <= (unsigned) RING;                     // two rows per run where ten such runs fit half the ring
    if (r2) ring_rows<TYPE, RING, 2, 2>(ring, ctl, c, KNC, 0u, pA2, nA2, rsE, nblkE, col_e, lane, w, nodots, sink_up);
    else    ring_rows<TYPE, RING, 1, 4>(ring, ctl, c, KNC, 0u, pA2, nA2, rsE, nblkE, col_e, lane, w, nodots, sink_up);
    if (c == 0 || c == 9) RINGK_T(c == 0 ? 2 : 3, 3);
    norm2_image();
    norm2_wait();
    if (r2) ring_rows<TYPE, RING, 2, 2>(ring, ctl, c, KNC, pA2, pA1, nA1, rsE, nblkE, col_q, lane, w, nodots, sink_qkv);
    else    ring_rows<TYPE, RING, 1, 4>(ring, ctl, c, KNC, pA2, pA1, nA1, rsE, nblkE, col_q, lane, w, nodots, sink_qkv);
    if (c == 0 || c == 9) RINGK_T(c == 0 ? 2 : 3, 4);
}

// =============================================================================================== k_ring_out
struct fq_ring_out_args {
    const float * resid; float * dst;                  // the residual row in / out (may alias)
    const uint8_t * down, * wo;                        // row 0 of each matrix (device layout)
    int E, FF, nblkE, nblkF; unsigned rsE, rsF;        // Wo: K = E, Wdown: K = FF
    int rows;                                          // output rows (= n_embd), dealt evenly over the workgroups
    const uint8_t * ff_image;                          // image of gelu(up) (the format's activation type), length FF
    const float * att;                                 // f32 attention row (quantized here: the k-quants' Q8_K) ...
    const uint8_t * att_image;                         // ... or its image, written by the attention launch (Q8_0 / Q8_1)
    unsigned * err; int debug_mode; long long * dbg;
};

template <int TYPE>
__global__ void __launch_bounds__(KNT) k_ring_out(fq_ring_out_args a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr int ACT = fq_act_of(TYPE);
    constexpr int RING = KNSLOT * ENG_SLOT;
    constexpr int NC = KNH;                                                // every helper wave is a consumer
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int E = a.E, FF = a.FF;
    const int r0 = (int)((long long) blockIdx.x * a.rows / gridDim.x), r1 = (int)((long long)(blockIdx.x + 1) * a.rows / gridDim.x), nrows = r1 - r0;
    uint8_t * ring    = smem;
    uint8_t * img_ff  = smem + RING + ENG_MIRROR;
    uint8_t * img_att = img_ff + fq_act_col_bytes(ACT, FF);
    uint8_t * ctlp    = img_att + fq_act_col_bytes(ACT, E);
    const unsigned ctl = (unsigned)(uintptr_t) ctlp;
    auto ldsf_st = [&](unsigned off, float v) { lds_st(ctl + off, __builtin_bit_cast(unsigned, v)); };
    auto ldsf_ld = [&](unsigned off) { return __builtin_bit_cast(float, lds_ld(ctl + off)); };
    const unsigned rsE = a.rsE, rsF = a.rsF;
    const unsigned pD = ring_pad1k((unsigned) nrows * rsF), pO = ring_pad1k((unsigned) nrows * rsE);

    if (tid < 16) lds_st(ctl + eng_ctl::LOW + 4 * tid, tid < NC ? 0u : 0xFFFFFFFFu);
    if (tid < 16) lds_st(ctl + eng_ctl::XG_DONE + 4 * tid, 0u);
    if (tid == 0) lds_st(ctl + eng_ctl::LANDED, 0u);
    if (a.dbg && tid == 0) a.dbg[(size_t) blockIdx.x * 8] = (long long) wall_clock64();
    __syncthreads();                                                       // the only workgroup barrier: before the roles split

    if (wid == 0) {
        // ================================================================================ loader: Wdown's rows, then Wo's
        ring_loader<KNSLOT> ld(ring, ctl, a.err, lane);
        ld.nospace = (a.debug_mode & 128) != 0;
        if (!(a.debug_mode & 1)) ld.w.until(ctl + RING_XISSUED, (unsigned) KNH, ENG_W_XG);     // the images' loads go first
        RINGK_T(0, 1);
        ld.seg(a.down + (size_t) r0 * rsF, pD);                            // (an empty segment is zero pieces)
        ld.seg(a.wo + (size_t) r0 * rsE, pO);
        RINGK_T(0, 2);
        ld.finish();
        RINGK_T(0, 3);
        return;
    }

    // ==================================================================================== helpers = consumers
    const int h = wid - 1, ht = tid - 64;
    eng_wait w{ a.err, false, nullptr, 0 };
    const bool nodots = (a.debug_mode & 2) != 0;
    {
        // ---- prologue: the image of gelu(up) and the attention output's image into LDS, this workgroup's residual values beside them
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        const int nvec_ff = (int)(fq_act_col_bytes(ACT, FF) >> 4);
        const int nvec_at = a.att_image ? (int)(fq_act_col_bytes(ACT, E) >> 4) : 0;
        const u32x4 * src_ff = (const u32x4 *) a.ff_image;
        const u32x4 * src_at = (const u32x4 *)(a.att_image ? a.att_image : a.ff_image);
        constexpr int NTF = 4, NSB = 3;
        u32x4 tf[NTF], tq;
#pragma unroll
        for (int k = 0; k < NTF; ++k) { const int i = k * KHT + ht; tf[k] = src_ff[i < nvec_ff ? i : nvec_ff - 1]; }
        tq = src_at[ht < nvec_at ? ht : 0];
        float4 av[NSB];                                                    // k-quants: this wave's super-blocks h, h + 11, h + 22 of the f32 attention row
        if constexpr (ACT == FQ_Q8_K) {
#pragma unroll
            for (int k = 0; k < NSB; ++k) { const int sb = h + KNH * k; av[k] = *(const float4 *)(a.att + 256 * (sb < (E >> 8) ? sb : 0) + 4 * lane); }
        }
        float res = 0.0f;
        if (h == 0) res = a.resid[r0 + (lane < nrows ? lane : 0)];
        if (lane == 0) lds_add(ctl + RING_XISSUED, 1u);                    // the loader may start: these requests are in the queue ahead of its own
#pragma unroll
        for (int k = 0; k < NTF; ++k) { const int i = k * KHT + ht; if (i < nvec_ff) ((u32x4 *) img_ff)[i] = tf[k]; }
        for (int i = NTF * KHT + ht; i < nvec_ff; i += KHT) ((u32x4 *) img_ff)[i] = src_ff[i];
        if (ht < nvec_at) ((u32x4 *) img_att)[ht] = tq;
        for (int i = KHT + ht; i < nvec_at; i += KHT) ((u32x4 *) img_att)[i] = src_at[i];
        if constexpr (ACT == FQ_Q8_K) {
            const act_image_ptr o = act_image_at(img_att, ACT, E);
#pragma unroll
            for (int k = 0; k < NSB; ++k) { const int sb = h + KNH * k; if (sb < (E >> 8)) quant_q8K_wave(av[k], lane, sb, o); }
            for (int sb = h + KNH * NSB; sb < (E >> 8); sb += KNH) quant_q8K_wave(*(const float4 *)(a.att + 256 * sb + 4 * lane), lane, sb, o);
        }
        if (h == 0 && lane < nrows) ldsf_st(eng_ctl::XRES + 4 * lane, res);
        lds_drain();
        if (lane == 0) lds_add(ctl + eng_ctl::IMG_DONE, 1u);
        w.until(ctl + eng_ctl::IMG_DONE, (unsigned) KNH, ENG_W_IMG);
        if (h <= 1) RINGK_T(1 + h, 2);
    }
    const fq_actcol col_ff  = { (const int8_t *) img_ff,  (const float *)(img_ff + fq_act_d_off(ACT, FF)),  (const void *)(img_ff + fq_act_aux_off(ACT, FF)) };
    const fq_actcol col_att = { (const int8_t *) img_att, (const float *)(img_att + fq_act_d_off(ACT, E)),  (const void *)(img_att + fq_act_aux_off(ACT, E)) };
    // row i of both segments belongs to the same wave (i == h mod 11): Wdown's dot waits in LDS for Wo's
    auto sink_d = [&](int i, float v) { if (lane == 0) ldsf_st(eng_ctl::OUTB + 4 * i, v); };
    auto sink_o = [&](int i, float v) { if (lane == 0) a.dst[r0 + i] = (ldsf_ld(eng_ctl::OUTB + 4 * i) + v) + ldsf_ld(eng_ctl::XRES + 4 * i); };      // libfalcon.cpp:2399-2400
    if constexpr (fq_kdot<TYPE>::ok) {
        // k-quants, rows of whole columns: the fast unit dots, a row in chunks of KU passes whose ring space is handed back as soon as they sit in registers
        constexpr int KU = fq_unit<TYPE>::ELEMS == 64 ? 2 : 4;
        constexpr int CBk = fq_lay<TYPE>::CB;
        if (a.nblkE % (KU * CBk) == 0 && a.nblkF % (KU * CBk) == 0 && !(a.debug_mode & 32)) {
            const typename fq_kdot<TYPE>::lane_t L = fq_kdot<TYPE>::lane_init(lane);
            ring_rows_kc<TYPE, RING, KU>(ring, ctl, h, NC, 0u, pD, nrows, rsF, a.nblkF / CBk, col_ff, L, lane, w, nodots, sink_d);
            if (h <= 1) RINGK_T(1 + h, 3);
            ring_rows_kc<TYPE, RING, KU>(ring, ctl, h, NC, pD, pO, nrows, rsE, a.nblkE / CBk, col_att, L, lane, w, nodots, sink_o);
            if (h <= 1) RINGK_T(1 + h, 4);
            return;
        }
    }
    ring_rows<TYPE, RING, 1, 4>(ring, ctl, h, NC, 0u, pD, nrows, rsF, a.nblkF, col_ff, lane, w, nodots, sink_d);
    if (h <= 1) RINGK_T(1 + h, 3);
    ring_rows<TYPE, RING, 1, 4>(ring, ctl, h, NC, pD, pO, nrows, rsE, a.nblkE, col_att, lane, w, nodots, sink_o);
    if (h <= 1) RINGK_T(1 + h, 4);
}


// =============================================================================================== k_ring_out_sys
// k_ring_out for the k-quants at widths of whole columns, with the consumers arranged as a SYSTOLIC chain instead of one wave per row: a Wdown row of
// Falcon-40B is 18 KiB (Q4_K), eleven rows do not fit the ring, and a k-quant unit costs ~50 vector instructions -- the launch is bound by instruction
// issue as much as by HBM, so every wave must be busy all the time and must not re-read activations. The stream is a sequence of ROW UNITS
// [Wdown row i | Wo row i]; a unit is S = SD + 1 STAGES of KU passes (SD of Wdown, one of Wo), and wave (pipe, s) owns stage s of the units
// i == pipe (mod NP), NP = 11 / S pipes: its activation slices are the same for every unit and live in registers (fq_kdot.h), its integer work needs
// nothing from its neighbours, and only the f32 partial sum of the lane travels down the chain through LDS -- acc_in, + term 0, + term 1, ... in pass
// order, exactly the one-wave-per-row association (per lane ascending units, then the wave butterfly by the last Wdown stage). The Wo stage adds
// x[i] = (Wdown's dot + its own dot) + x[i] (libfalcon.cpp:2399-2400). Hand-off per (pipe, boundary): two slots of 64 floats, PROD / CONS row counters.
template <int TYPE>
__global__ void __launch_bounds__(KNT) k_ring_out_sys(fq_ring_out_args a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    typedef fq_kdot<TYPE> KD;
    constexpr int ACT = FQ_Q8_K;
    constexpr int RING = KNSLOT * ENG_SLOT;
    constexpr int KU = fq_unit<TYPE>::ELEMS == 64 ? 2 : 4;                 // passes (= columns) per stage
    constexpr unsigned COLB = (unsigned)(fq_lay<TYPE>::CB * fq_lay<TYPE>::TS);
    constexpr int D = 2;                                                   // hand-off slots per boundary
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int E = a.E, FF = a.FF;
    const int r0 = (int)((long long) blockIdx.x * a.rows / gridDim.x), r1 = (int)((long long)(blockIdx.x + 1) * a.rows / gridDim.x), nrows = r1 - r0;
    const int SD = a.nblkF / (KU * fq_lay<TYPE>::CB), S = SD + 1, NP = KNH / S;
    uint8_t * ring    = smem;
    uint8_t * img_ff  = smem + RING + ENG_MIRROR;
    uint8_t * img_att = img_ff + fq_act_col_bytes(ACT, FF);
    uint8_t * ctlp    = img_att + fq_act_col_bytes(ACT, E);
    const unsigned ctl = (unsigned)(uintptr_t) ctlp;
    const unsigned slots = ctl + KCTL_BYTES;                               // [boundary = pipe * (S - 1) + s][D][64] floats
    auto ldsf_st = [&](unsigned off, float v) { lds_st(ctl + off, __builtin_bit_cast(unsigned, v)); };
    auto ldsf_ld = [&](unsigned off) { return __builtin_bit_cast(float, lds_ld(ctl + off)); };
    const unsigned rsE = a.rsE, rsF = a.rsF;
    const unsigned pD = ring_pad1k(rsF), pO = ring_pad1k(rsE), ustride = pD + pO;

    if (tid < 32) lds_st(ctl + eng_ctl::CNT + 4 * tid, 0u);                // PROD (CNT) and CONS (CNTH) row counters of the 16 possible boundaries
    if (tid < 16) lds_st(ctl + eng_ctl::LOW + 4 * tid, tid < NP * S ? 0u : 0xFFFFFFFFu);
    if (tid < 16) lds_st(ctl + eng_ctl::XG_DONE + 4 * tid, 0u);
    if (tid == 0) lds_st(ctl + eng_ctl::LANDED, 0u);
    if (a.dbg && tid == 0) a.dbg[(size_t) blockIdx.x * 8] = (long long) wall_clock64();
    __syncthreads();                                                       // the only workgroup barrier: before the roles split

    if (wid == 0) {
        // ================================================================================ loader: row units [Wdown row | Wo row], each padded to whole KiB
        ring_loader<KNSLOT> ld(ring, ctl, a.err, lane);
        ld.nospace = (a.debug_mode & 128) != 0;
        if (!(a.debug_mode & 1)) ld.w.until(ctl + RING_XISSUED, (unsigned) KNH, ENG_W_XG);     // the images' loads go first
        RINGK_T(0, 1);
        const uint8_t * pd = a.down + (size_t) r0 * rsF, * po = a.wo + (size_t) r0 * rsE;
        for (int i = 0; i < nrows && !ld.w.dead; ++i) {
            ld.seg(pd, pD); ld.seg(po, pO);
            pd += rsF; po += rsE;
        }
        RINGK_T(0, 2);
        ld.finish();
        RINGK_T(0, 3);
        return;
    }

    // ==================================================================================== helpers
    const int h = wid - 1, ht = tid - 64;
    eng_wait w{ a.err, false, nullptr, 0 };
    const bool nodots = (a.debug_mode & 2) != 0;
    {
        // ---- prologue (k_ring_out's): the image of gelu(up), the attention row quantized to Q8_K, this workgroup's residual values
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        const int nvec_ff = (int)(fq_act_col_bytes(ACT, FF) >> 4);
        const u32x4 * src_ff = (const u32x4 *) a.ff_image;
        constexpr int NTF = 4, NSB = 3;
        u32x4 tf[NTF];
#pragma unroll
        for (int k = 0; k < NTF; ++k) { const int i = k * KHT + ht; tf[k] = src_ff[i < nvec_ff ? i : nvec_ff - 1]; }
        float4 av[NSB];
#pragma unroll
        for (int k = 0; k < NSB; ++k) { const int sb = h + KNH * k; av[k] = *(const float4 *)(a.att + 256 * (sb < (E >> 8) ? sb : 0) + 4 * lane); }
        float res = 0.0f;
        if (h == 0) res = a.resid[r0 + (lane < nrows ? lane : 0)];
        if (lane == 0) lds_add(ctl + RING_XISSUED, 1u);
#pragma unroll
        for (int k = 0; k < NTF; ++k) { const int i = k * KHT + ht; if (i < nvec_ff) ((u32x4 *) img_ff)[i] = tf[k]; }
        for (int i = NTF * KHT + ht; i < nvec_ff; i += KHT) ((u32x4 *) img_ff)[i] = src_ff[i];
        const act_image_ptr o = act_image_at(img_att, ACT, E);
#pragma unroll
        for (int k = 0; k < NSB; ++k) { const int sb = h + KNH * k; if (sb < (E >> 8)) quant_q8K_wave(av[k], lane, sb, o); }
        for (int sb = h + KNH * NSB; sb < (E >> 8); sb += KNH) quant_q8K_wave(*(const float4 *)(a.att + 256 * sb + 4 * lane), lane, sb, o);
        if (h == 0 && lane < nrows) ldsf_st(eng_ctl::XRES + 4 * lane, res);
        lds_drain();
        if (lane == 0) lds_add(ctl + eng_ctl::IMG_DONE, 1u);
        w.until(ctl + eng_ctl::IMG_DONE, (unsigned) KNH, ENG_W_IMG);
        if (h <= 1) RINGK_T(1 + h, 2);
    }
    const int pipe = h / S, s = h - pipe * S;
    if (pipe >= NP) return;                                                // (11 is not a multiple of S: the spare waves only helped with the prologue)
    const fq_actcol col_ff  = { (const int8_t *) img_ff,  (const float *)(img_ff + fq_act_d_off(ACT, FF)),  (const void *)(img_ff + fq_act_aux_off(ACT, FF)) };
    const fq_actcol col_att = { (const int8_t *) img_att, (const float *)(img_att + fq_act_d_off(ACT, E)),  (const void *)(img_att + fq_act_aux_off(ACT, E)) };
    const typename KD::lane_t L = KD::lane_init(lane);
    ring_kacts<TYPE, KU> A;
    const bool wo_stage = s == SD;
    if (wo_stage) ring_kacts_load<TYPE, KU>(col_att, 0, L, A); else ring_kacts_load<TYPE, KU>(col_ff, KU * s, L, A);
    const unsigned stage_off = wo_stage ? pD : (unsigned)(KU * s) * COLB;
    const unsigned b_in = (unsigned)(pipe * (S - 1) + s - 1), b_out = (unsigned)(pipe * (S - 1) + s);      // hand-off boundaries behind / ahead of this stage
    const unsigned prod_in = ctl + eng_ctl::CNT + 4 * b_in, cons_in = ctl + eng_ctl::CNTH + 4 * b_in;
    const unsigned prod_out = ctl + eng_ctl::CNT + 4 * b_out, cons_out = ctl + eng_ctl::CNTH + 4 * b_out;
    const unsigned slot_in = slots + b_in * (unsigned)(D * 256), slot_out = slots + b_out * (unsigned)(D * 256);
    const unsigned end_pos = (unsigned) nrows * ustride;
    if (lane == 0) lds_st(ctl + eng_ctl::LOW + 4 * h, pipe < nrows ? (unsigned) pipe * ustride + stage_off : end_pos);
    unsigned k = 0;
    for (int i = pipe; i < nrows; i += NP, ++k) {
        const unsigned upos = (unsigned) i * ustride + stage_off, need = upos + (unsigned) KU * COLB;
        for (unsigned spins = 0; (int)(lds_ld_u(ctl + eng_ctl::LANDED) - need) < 0;) { if (!w.spin(spins, ENG_W_LAND, need, 0)) break; __builtin_amdgcn_s_sleep(1); }
        float t[KU];
#pragma unroll
        for (int p = 0; p < KU; ++p) t[p] = 0.0f;
        if (!nodots) ring_kterms<TYPE, RING, KU>(ring, upos % (unsigned) RING, A, L, t);
        float acc = 0.0f;
        const bool nochain = (a.debug_mode & 64) != 0;                     // tuning aid: no hand-off (garbage results)
        if (s > 0 && !wo_stage && !nochain) {                              // the lane's partial sum so far
            for (unsigned spins = 0; (int)(lds_ld_u(prod_in) - (k + 1u)) < 0;) { if (!w.spin(spins, ENG_W_B1, k, b_in)) break; __builtin_amdgcn_s_sleep(1); }
            acc = __builtin_bit_cast(float, lds_ld(slot_in + (k & (D - 1)) * 256u + 4u * (unsigned) lane));
            if (lane == 0) lds_st(cons_in, k + 1u);
        }
#pragma unroll
        for (int p = 0; p < KU; ++p) acc += t[p];
        if (nochain) { if (wo_stage && lane == 0) a.dst[r0 + i] = acc; }
        else if (!wo_stage) {
            const float out = s == SD - 1 ? wave_sum(acc) : acc;           // the last Wdown stage hands over the row's dot product
            for (unsigned spins = 0; (int)(lds_ld_u(cons_out) - (k + 1u - (unsigned) D)) < 0 && k >= (unsigned) D;) { if (!w.spin(spins, ENG_W_B2, k, b_out)) break; __builtin_amdgcn_s_sleep(1); }
            lds_st(slot_out + (k & (D - 1)) * 256u + 4u * (unsigned) lane, __builtin_bit_cast(unsigned, out));
            lds_drain();
            if (lane == 0) lds_st(prod_out, k + 1u);
        } else {
            const float vo = wave_sum(acc);
            for (unsigned spins = 0; (int)(lds_ld_u(prod_in) - (k + 1u)) < 0;) { if (!w.spin(spins, ENG_W_B1, k, b_in)) break; __builtin_amdgcn_s_sleep(1); }
            const float vd = __builtin_bit_cast(float, lds_ld(slot_in + (k & (D - 1)) * 256u + 4u * (unsigned) lane));
            if (lane == 0) {
                lds_st(cons_in, k + 1u);
                a.dst[r0 + i] = (vd + vo) + ldsf_ld(eng_ctl::XRES + 4 * i);                         // libfalcon.cpp:2399-2400
            }
        }
        const int nx_ = i + NP;
        if (lane == 0) lds_st(ctl + eng_ctl::LOW + 4 * h, nx_ < nrows ? (unsigned) nx_ * ustride + stage_off : end_pos);
    }
    if (h <= 1) RINGK_T(1 + h, 4);
}

// ---- host side
namespace {
struct ringk_plan { int type, E, FF, qkv_rows, n_wg; fq_engine_sched * dev; };
std::vector<ringk_plan> g_kplans;

// rows [qg0, qg1) of Wqkv and 32-row groups [ug0, ug1) of Wup per workgroup (kernels_ring.hip's rule): the groups are dealt evenly, the Wqkv
// rows fill the workgroups with fewer groups up to the common row count
const fq_engine_sched * ringk_schedule(int type, int E, int FF, int qkv_rows, int n_wg, bool create) {
    for (const ringk_plan & p : g_kplans) if (p.type == type && p.E == E && p.FF == FF && p.qkv_rows == qkv_rows && p.n_wg == n_wg) return p.dev;
    if (!create) return nullptr;                                            // (a launch may be inside a stream capture: it never allocates)
    const int groups = FF / 32;
    std::vector<fq_engine_sched> s((size_t) n_wg);
    std::vector<int> ng((size_t) n_wg);
    for (int i = 0; i < n_wg; ++i) ng[(size_t) i] = groups / n_wg + (i < groups % n_wg ? 1 : 0);
    const int total = qkv_rows + FF;
    const int T = (total + n_wg - 1) / n_wg;
    std::vector<int> nq((size_t) n_wg, 0);
    int left = qkv_rows;
    for (int i = 0; i < n_wg && left > 0; ++i) { int q = T - 32 * ng[(size_t) i]; if (q < 0) q = 0; if (q > left) q = left; nq[(size_t) i] = q; left -= q; }
    for (int i = n_wg - 1; left > 0; i = (i + n_wg - 1) % n_wg) { ++nq[(size_t) i]; --left; }
    int g0 = 0, q0 = 0;
    for (int i = 0; i < n_wg; ++i) {
        s[(size_t) i] = fq_engine_sched{ q0, q0 + nq[(size_t) i], g0, g0 + ng[(size_t) i], 0, 0, 0, 0 };
        q0 += nq[(size_t) i]; g0 += ng[(size_t) i];
    }
    fq_engine_sched * dev = nullptr;
    HIP_CHECK(hipMalloc((void **) &dev, sizeof(fq_engine_sched) * (size_t) n_wg));
    HIP_CHECK(hipMemcpy(dev, s.data(), sizeof(fq_engine_sched) * (size_t) n_wg, hipMemcpyHostToDevice));
    g_kplans.push_back(ringk_plan{ type, E, FF, qkv_rows, n_wg, dev });
    return dev;
}
bool is_kquant(int t) { return t == FQ_Q2_K || t == FQ_Q3_K || t == FQ_Q4_K || t == FQ_Q5_K || t == FQ_Q6_K; }
size_t ringk_ln_fixed(int64_t E, bool two_norms) {
    return (size_t) ENG_MIRROR + (((size_t) E * 4 + 15) & ~(size_t) 15) + (two_norms ? 2 : 1) * fq_act_col_bytes(FQ_Q8_K, E) + KCTL_BYTES;
}
bool ringk_ln_shape_ok(int type, int64_t E, int64_t FF, int n_cu) {
    if (!is_kquant(type) || E % 256 || E > 8448 || FF % 32 || FF > (int64_t) 32 * 12 * n_cu) return false;
    if ((size_t) KNSLOT * ENG_SLOT + ringk_ln_fixed(E, true) > 160 * 1024) return false;
    const size_t rs = (size_t)(E / 256) * fq_desc(type).tsize;
    return 2 * rs + ENG_SLOT <= (size_t) KNSLOT * ENG_SLOT;                // a row next to the loader's restart slot
}
}   // namespace

void fq_ringk_free_plans() {
    for (ringk_plan & p : g_kplans) (void) hipFree(p.dev);
    g_kplans.clear();
}

// the schedule of a shape must exist before a stream capture (it allocates): called at context set-up
bool fq_ringk_prepare(int type, int64_t E, int64_t FF, int64_t qkv_rows, int n_cu) {
    if (!ringk_ln_shape_ok(type, E, FF, n_cu)) return false;
    ringk_schedule(type, (int) E, (int) FF, (int) qkv_rows, n_cu, true);
    return true;
}

// k_gemv_ln's launch (k-quants, GELU_STORE epilogue) through the ring form; false = outside its scope or no prepared schedule (nothing launched)
bool fq_launch_ringk_ln(const fq_gemv_ln_args & g, unsigned * err, int n_cu, hipStream_t st) {
    FQ_TL(st, "ringk_ln");
    if (g.nseg != 2 || g.seg[1].epi != FQ_LNEPI_GELU_STORE || g.seg[0].epi != FQ_LNEPI_STORE || g.argmax_val) return false;
    const fq_weight & wq = g.seg[0].w, & wu = g.seg[1].w;
    const int type = wq.type;
    if (type != wu.type || wq.K != g.E || wu.K != g.E || wq.row_stride != wu.row_stride) return false;
    if (!ringk_ln_shape_ok(type, g.E, wu.M, n_cu)) return false;
    const bool two_norms = g.seg[0].ln_w != g.seg[1].ln_w;
    fq_ringk_ln_args a{};
    a.x = g.x; a.E = (int) g.E; a.FF = (int) wu.M; a.nblkE = (int) wq.nblk; a.rsE = (unsigned) wq.row_stride;
    a.qkv = wq.plane[0]; a.up = wu.plane[0]; a.qkv_rows = (int) wq.M;
    a.ln_w = g.seg[1].ln_w; a.ln_b = g.seg[1].ln_b; a.ln2_w = two_norms ? g.seg[0].ln_w : nullptr; a.ln2_b = two_norms ? g.seg[0].ln_b : nullptr; a.two_norms = two_norms ? 1 : 0;
    a.qkv_dst = g.seg[0].dst; a.up_dst = g.seg[1].dst; a.gelu_tab = g.gelu_table;
    a.sched = ringk_schedule(type, a.E, a.FF, a.qkv_rows, n_cu, false);
    if (!a.sched) return false;
    a.epoch_word = g.epoch_word; a.n_past_ptr = g.n_past_ptr; a.rope_cs = g.rope_cs; a.rope_cur = g.rope_cur;
    a.err = err; a.dbg = g.dbg;
    static const int dbg = getenv("FQ_RING_DEBUG") ? atoi(getenv("FQ_RING_DEBUG")) : 0;
    a.debug_mode = dbg;
    const size_t lds = (size_t) KNSLOT * ENG_SLOT + ringk_ln_fixed(a.E, two_norms);
#define FQ_RK_CASE(T) case T: { \
        static bool set = false; \
        if (!set) { HIP_CHECK(hipFuncSetAttribute((const void *) k_ring_ln_k<T>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); set = true; } \
        hipEvent_t e0_ = nullptr, e1_ = nullptr; fq_prof_events(&e0_, &e1_); \
        if (e0_) hipExtLaunchKernelGGL((k_ring_ln_k<T>), dim3((unsigned) n_cu), dim3(KNT), lds, st, e0_, e1_, 0, a.x, a.sched, a.qkv, a.up, a.E, a.FF, a.nblkE, a.rsE, a); \
        else     hipLaunchKernelGGL((k_ring_ln_k<T>), dim3((unsigned) n_cu), dim3(KNT), lds, st, a.x, a.sched, a.qkv, a.up, a.E, a.FF, a.nblkE, a.rsE, a); } break;
    switch (type) {
        FQ_RK_CASE(FQ_Q2_K) FQ_RK_CASE(FQ_Q3_K) FQ_RK_CASE(FQ_Q4_K) FQ_RK_CASE(FQ_Q5_K) FQ_RK_CASE(FQ_Q6_K)
        default: return false;
    }
#undef FQ_RK_CASE
    return true;
}

// k_gemv_out's launch through the ring form (all ten formats); false = outside its scope, nothing launched
bool fq_launch_ring_out(const fq_gemv_out_args & g, unsigned * err, int n_cu, hipStream_t st) {
    FQ_TL(st, "ring_out");
    const fq_weight & wd = g.w_down, & wo = g.w_wo;
    const int type = wo.type, act = fq_desc(type).act_type;
    if (type != wd.type || wd.M != wo.M || fq_desc(type).blck == 0) return false;
    const int64_t E = wo.K, FF = wd.K, rows = wo.M;
    const int blck = fq_desc(type).blck;
    if (E % blck || FF % blck || E % 32 || FF % 32) return false;
    if (act == FQ_Q8_K ? (g.att == nullptr) : (g.att_image == nullptr)) return false;      // the k-quants quantize the f32 row here; the legacy formats get their image
    if ((rows + n_cu - 1) / n_cu > 64 || rows < 1) return false;                           // a workgroup's dots and residual values: 64 LDS floats each
    const size_t fixed = (size_t) ENG_MIRROR + fq_act_col_bytes(act, FF) + fq_act_col_bytes(act, E) + KCTL_BYTES;
    const size_t lds = (size_t) KNSLOT * ENG_SLOT + fixed;
    if (lds > 160 * 1024) return false;
    if ((size_t) wd.row_stride + ENG_SLOT > (size_t) KNSLOT * ENG_SLOT) return false;      // a Wdown row next to the loader's restart slot
    fq_ring_out_args a{};
    a.resid = g.resid; a.dst = g.dst; a.down = wd.plane[0]; a.wo = wo.plane[0];
    a.E = (int) E; a.FF = (int) FF; a.nblkE = (int) wo.nblk; a.nblkF = (int) wd.nblk; a.rsE = (unsigned) wo.row_stride; a.rsF = (unsigned) wd.row_stride;
    a.rows = (int) rows; a.ff_image = g.act_ff_image;
    a.att = act == FQ_Q8_K ? g.att : nullptr; a.att_image = act == FQ_Q8_K ? nullptr : g.att_image;
    a.err = err; a.dbg = g.dbg;
    static const int dbg = getenv("FQ_RING_DEBUG") ? atoi(getenv("FQ_RING_DEBUG")) : 0;
    a.debug_mode = dbg;
    // k-quants at widths of whole columns: the systolic form (k_ring_out_sys)
    static const bool sys_on = getenv("FQ_RING_OUT_SYS") && atoi(getenv("FQ_RING_OUT_SYS")) != 0;      // (opt-in: measured slower than the row-chunk form)
    if (sys_on && act == FQ_Q8_K) {
        const int cb = 1024 / fq_desc(type).plane[0].bytes, ku = fq_desc(type).unit_elems == 64 ? 2 : 4;
        const int sd = (int)(wd.nblk / (ku * cb)), S = sd + 1;
        const size_t ustride = ((wd.row_stride + 1023) & ~(size_t) 1023) + ((wo.row_stride + 1023) & ~(size_t) 1023);
        const size_t lds_sys = lds + (size_t) 16 * 2 * 256;               // + the hand-off slots
        if (wo.nblk == ku * cb && wd.nblk == (int64_t) sd * ku * cb && sd >= 1 && S <= KNH && lds_sys <= 160 * 1024 && ustride + ENG_SLOT <= (size_t) KNSLOT * ENG_SLOT) {
#define FQ_ROS_CASE(T) case T: { \
        static bool set = false; \
        if (!set) { HIP_CHECK(hipFuncSetAttribute((const void *) k_ring_out_sys<T>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); set = true; } \
        hipEvent_t e0_ = nullptr, e1_ = nullptr; fq_prof_events(&e0_, &e1_); \
        if (e0_) hipExtLaunchKernelGGL((k_ring_out_sys<T>), dim3((unsigned) n_cu), dim3(KNT), lds_sys, st, e0_, e1_, 0, a); \
        else     hipLaunchKernelGGL((k_ring_out_sys<T>), dim3((unsigned) n_cu), dim3(KNT), lds_sys, st, a); } return true;
            switch (type) {
                FQ_ROS_CASE(FQ_Q2_K) FQ_ROS_CASE(FQ_Q3_K) FQ_ROS_CASE(FQ_Q4_K) FQ_ROS_CASE(FQ_Q5_K) FQ_ROS_CASE(FQ_Q6_K)
                default: break;
            }
#undef FQ_ROS_CASE
        }
    }
#define FQ_RO_CASE(T) case T: { \
        static bool set = false; \
        if (!set) { HIP_CHECK(hipFuncSetAttribute((const void *) k_ring_out<T>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); set = true; } \
        hipEvent_t e0_ = nullptr, e1_ = nullptr; fq_prof_events(&e0_, &e1_); \
        if (e0_) hipExtLaunchKernelGGL((k_ring_out<T>), dim3((unsigned) n_cu), dim3(KNT), lds, st, e0_, e1_, 0, a); \
        else     hipLaunchKernelGGL((k_ring_out<T>), dim3((unsigned) n_cu), dim3(KNT), lds, st, a); } break;
    switch (type) {
        FQ_RO_CASE(FQ_Q4_0) FQ_RO_CASE(FQ_Q4_1) FQ_RO_CASE(FQ_Q5_0) FQ_RO_CASE(FQ_Q5_1) FQ_RO_CASE(FQ_Q8_0)
        FQ_RO_CASE(FQ_Q2_K) FQ_RO_CASE(FQ_Q3_K) FQ_RO_CASE(FQ_Q4_K) FQ_RO_CASE(FQ_Q5_K) FQ_RO_CASE(FQ_Q6_K)
        default: return false;
    }
#undef FQ_RO_CASE
    return true;
}
