// kernels_gemm_skinny.hip -- the quantized mat-mul for SMALL batches (5 <= N <= 16 columns) at weight-stream speed (gfx950).
//
// ggml_compute_forward_mul_mat_q_f32 (ggml.c:11318-11529; CUDA twin for every N > 1: dequantize + cublasGemmEx,
// ggml-cuda.cu:2353-2403, 2931-2951) for the case that serving produces all the time -- a handful of lock-step sequences, a short
// prompt chunk: the matrix is streamed ONCE, the int8 matrix pipe takes the N columns, the exact per-32-group f32 scaling of the CPU
// path runs on the vector ALU underneath the stream (4 results per lane and group). kernels_gemm.hip's 128-token tiles cost the same
// 8-9 ms per Falcon-7B pass whatever N is (a barrier and a register round trip per 128 of K); this form is bound by HBM.
//
//   workgroup   32 weight rows (two 16-row tiles) x all of K. Waves 0-1 are the LOADERS (16 rows each): per stage (32 blocks of K = half a column of the
//               device layout, fq_types.h) ONE global_load_lds per row moves the row's raw blocks -- 512 B of quants and the 64-128 B of
//               its scale planes, gathered by the lanes of the same instruction -- into the stage buffer: no registers, no decode, three
//               stages ahead of the arithmetic (the columns, L2 hits, one stage ahead). The 2 S consumer waves = (tile t, K share sw) stage the N activation columns of the stage the same way (qs,
//               d, aux of the Q8 images the decode path uses), transpose the token scales once per stage, and run
//               v_mfma_i32_16x16x32_i8 per group: A = 16 tokens x 32 k, B = 16 rows x 32 k (nibbles unpacked on the way out of LDS:
//               two shifts and two masks per lane), C starts at -8 isum[token] (Q4_0; -16 for Q5_0), so the int32 result is exactly
//               the CPU's sumi; then the reference's scalar per-block expression per result.
//   association the S waves of a tile split K exactly like k_gemm_q (group g -> partial sum g mod S, partial sums added as
//               ((P0 + P1) + P2) + P3; S chosen by the same rule): results are bit-identical to kernels_gemm.hip's, and the
//               oracle's split orders (orc_set_sum_order 2 / 3 / 4 / 5) apply unchanged.
// Scope of this file: the legacy formats (Q4_0, Q4_1, Q5_0, Q5_1, Q8_0). The k-quants have their own forms in kernels_gemm_skinny_k.hip
// (k_gemm_skinny_q4k: Q4_K / Q5_K; k_gemm_skinny_q2k: Q2_K / Q3_K; k_gemm_skinny_q6k), for matrices at model widths (fq_skinny_q4k_shape).
//
// (fq_skinny_dev.h: the helpers shared with the k-quant file.) Three forms, picked by fq_launch_gemm_skinny (all bit-identical to each other and to k_gemm_q):
//   k_gemm_skinny_res   the columns RESIDENT in LDS, one persistent workgroup per CU, optionally two matrices per launch -- rows up to ~4.6 k long
//   k_gemm_skinny_ks    one K share per workgroup for longer rows (+ k_skinny_sum4): a quarter of the columns resident, self-paced waves
//   k_gemm_skinny       the per-stage form described above: the fallback when neither fits LDS


#include "fq_skinny_dev.h"
#define SK_FMA FQ_SPLIT_FMA                                                 // the legacy formats' K-split sums in the fused form: ONE macro with the tile GEMM (fq_types.h)

namespace {

constexpr int SK_TOKB = 1024 + 128 + 160;  // LDS bytes per token and stage: [qs 1024 | d 128 | aux 144 (the source is read from a 16-byte boundary: LDS-DMA ignores the low address bits) | pad]
template <int TYPE> struct sk_fmt {
    static constexpr fq_type_desc D = fq_desc(TYPE);
    static constexpr int QB = D.plane[0].bytes, PB1 = D.plane[1].bytes, PB2 = D.nplanes > 2 ? D.plane[2].bytes : 0;
    static constexpr int CB = 1024 / QB, SPC = CB / SK_GS;      // blocks per column of the device layout (64; Q8_0: 32), stages per column
    static constexpr int P2PAD = PB2 ? 16 : 0;                  // plane 2 of a partial column starts on a 4-byte boundary only: read from the 16-byte boundary below
    static constexpr int ROWD = SK_GS * (QB + PB1 + PB2) + P2PAD;      // bytes of a row's stage: [quants | plane 1 | plane 2 (+ 16)]
    static constexpr int NL = ROWD / 16;                        // 16-byte lanes of the row's gather
    // LDS pitch of a row: an ODD number of 16-byte slots, so that the 16 rows of a tile fall into distinct banks (an operand read of
    // Q4_0's 576-byte rows was a 4-way bank conflict: 576 = 64 mod 256)
    static constexpr int ROWB = (NL % 2 == 0) ? ROWD + 16 : ROWD;
    static constexpr bool HAS_MIN = (TYPE == FQ_Q4_1 || TYPE == FQ_Q5_1);
    // stage buffers: the weights (HBM: long latency) run NBW - 1 stages ahead, the columns (L2 hits) NBT - 1
    static constexpr int NBW = ROWB > 800 ? 2 : 4, NBT = 2;
    static constexpr int LDS = NBW * SK_TM * ROWB + NBT * SK_TN * SK_TOKB + 2 * SK_GS * SK_TN * 4;
};

}   // namespace


template <int TYPE, int S>
__global__ void __launch_bounds__(64 * (2 + 2 * S)) k_gemm_skinny(fq_weight w, fq_act act, int N, float * dst, int64_t ldd, fq_gemv_epi ep, int dbg) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    typedef sk_fmt<TYPE> F;
    constexpr int ACT = fq_act_of(TYPE);
    constexpr int NCW = 2 * S, NBW = F::NBW, NBT = F::NBT, NLW = 2;        // consumer waves, buffers, loader waves (16 rows each)
    constexpr int WSTAGE = SK_TM * F::ROWB, TSTAGE = SK_TN * SK_TOKB;
    constexpr int WOPS = (F::NL + 63) / 64;                                // loader instructions per row and stage
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t K = w.K, M = w.M;
    const int nblk = (int) w.nblk;
    const int nstages = (nblk + SK_GS - 1) / SK_GS;
    const int64_t m0 = (int64_t) blockIdx.x * SK_TM;
    const size_t img = fq_act_col_bytes(ACT, K);
    uint8_t * dxT = smem + (size_t) NBW * WSTAGE + (size_t) NBT * TSTAGE;   // [32 groups][16 tokens] f32: the tokens' d
    uint8_t * ciT = dxT + SK_GS * SK_TN * 4;                               // [32][16]: C start values (int) or the tokens' s (f32)
    auto wbuf = [&](int s) { return smem + (size_t)(s % NBW) * WSTAGE; };
    auto tbuf = [&](int s) { return smem + (size_t) NBW * WSTAGE + (size_t)(s % NBT) * TSTAGE; };

    // ---- staging (LDS-DMA). Stage s = blocks [32 s, 32 s + 32) of every row = half hf = s & 1 of column c = s >> 1.
    auto issue_weights = [&](int s, int lw) {                                // loader wave lw: rows 16 lw .. 16 lw + 15 of the tile
        const int c = s / F::SPC, hf = s % F::SPC;
        const int rem = nblk - F::CB * c, nbc = rem < F::CB ? rem : F::CB;
        const unsigned colb = (unsigned) c * (unsigned)(F::CB * F::D.tsize);
        // the lane's 16 bytes of the LDS row -> where they are in the device layout (fq_types.h: planes back to back per column).
        // Every source address is a multiple of 16 (the DMA ignores the low bits): plane 2 is read from the boundary below it.
        auto src_of = [&](int l) {
            const int p = 16 * l;
            unsigned o;
            if (p < SK_GS * F::QB)                   o = colb + (unsigned)(SK_GS * hf * F::QB + p);
            else if (p < SK_GS * (F::QB + F::PB1))   o = colb + (unsigned)(nbc * F::QB + SK_GS * hf * F::PB1 + (p - SK_GS * F::QB));
            else                                     o = colb + (unsigned)(((nbc * (F::QB + F::PB1)) & ~15) + SK_GS * hf * F::PB2 + (p - SK_GS * (F::QB + F::PB1)));
            return o;
        };
        const unsigned v0 = src_of(lane), v1 = src_of(lane + 64 < F::NL ? lane + 64 : F::NL - 1);
        const unsigned wb = __builtin_amdgcn_readfirstlane(sk_lds(wbuf(s)));
        if (m0 + SK_TM <= M) {
            // a whole tile: the rows are row_stride apart -- per-row offsets in registers, one scalar base, 3 scalar instructions per row
            const uint8_t * base = sk_uniform(w.plane[0] + (size_t) m0 * w.row_stride);
            const unsigned rs = (unsigned) w.row_stride;
            {
                const int h = lw;
                sk_voff16 o;
#pragma unroll
                for (int i = 0; i < 16; ++i) o.v[i] = v0 + (unsigned)(16 * h + i) * rs;
                unsigned ml = wb + (unsigned)(16 * h * F::ROWB);
                if (lane < (F::NL < 64 ? F::NL : 64)) sk_dma16(base, o, ml, (unsigned) F::ROWB);
                if constexpr (WOPS > 1) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) o.v[i] = v1 + (unsigned)(16 * h + i) * rs;
                    unsigned ml2 = wb + (unsigned)(16 * h * F::ROWB + 1024);
                    if (lane + 64 < F::NL) sk_dma16(base, o, ml2, (unsigned) F::ROWB);
                }
            }
            return;
        }
        for (int r = 16 * lw; r < 16 * lw + 16; ++r) {                     // the matrix's last, partial tile: rows beyond M re-read row M - 1
            const int64_t row = m0 + r < M ? m0 + r : M - 1;
            const uint8_t * base = sk_uniform(w.plane[0] + (size_t) row * w.row_stride);
            if (lane < (F::NL < 64 ? F::NL : 64)) sk_dma(base, v0, wb + (unsigned)(r * F::ROWB));
            if constexpr (WOPS > 1) { if (lane + 64 < F::NL) sk_dma(base, v1, wb + (unsigned)(r * F::ROWB + 1024)); }
        }
    };
    auto issue_tokens = [&](int s, int cw) {                               // consumer wave cw: its share of the 16 columns
        const unsigned tb = sk_lds(tbuf(s));
        const unsigned last = (unsigned)(img - 16);
        unsigned vq = (unsigned)(SK_GS * 32 * s + 16 * lane);              // qs: 1024 bytes
        vq = vq < last ? vq : last;
        const size_t nd4 = fq_act_d_elems(ACT, K) * 4;
        // d: 128 bytes (K is a multiple of 32: aligned); aux: 144 bytes from the 16-byte boundary below it
        unsigned vs = (unsigned)(lane < 8 ? (size_t) K + 4 * SK_GS * s + 16 * lane : (((size_t) K + nd4) & ~(size_t) 15) + 4 * SK_GS * s + 16 * (lane - 8));
        vs = vs < last ? vs : last;
        for (int t = cw; t < SK_TN; t += NCW) {
            const uint8_t * base = act.base + (size_t)(t < N ? t : N - 1) * img;
            sk_dma(base, vq, tb + (unsigned)(t * SK_TOKB));
            if (lane < 17 && !(dbg & 64)) sk_dma(base, vs, tb + (unsigned)(t * SK_TOKB + 1024));
        }
    };
    constexpr int TOPS = 2 * ((SK_TN + NCW - 1) / NCW);                    // DMA instructions a consumer wave issues per stage
    constexpr int LOPS = 16 * WOPS;                                        // ... and a loader wave

    const int aux_delta = (int)(((size_t) K + fq_act_d_elems(ACT, K) * 4) & 15);     // the aux array's offset from the 16-byte boundary its DMA starts at
    const bool loader = wid < NLW;
    const int cw = wid - NLW, tile = cw & 1, sw = cw >> 1;
    const int l16 = lane & 15, kq = lane >> 4;
    float acc[4] = { 0.0f, 0.0f, 0.0f, 0.0f };

    // ---- prologue: the first NBW - 1 stages of weights, NBT - 1 of columns
    if (loader) { for (int s = 0; s < NBW - 1 && s < nstages; ++s) if (!(dbg & 8)) issue_weights(s, wid); }
    else        { for (int s = 0; s < NBT - 1 && s < nstages; ++s) if (!(dbg & 4)) issue_tokens(s, cw); }
    unsigned long long tm[5] = { 0, 0, 0, 0, 0 }, tprev = __builtin_amdgcn_s_memtime();
#define SK_TM_MARK(i) do { if (dbg & 128) { const unsigned long long tn_ = __builtin_amdgcn_s_memtime(); tm[i] += tn_ - tprev; tprev = tn_; } } while (0)
    for (int s = 0; s < nstages; ++s) {
        // stage s has landed (every issuing wave waits for its own DMA before it arrives: all but the stages issued after it), and
        // nobody reads stage s - 1 any more
        if (loader) {
            const int ahead = nstages - 1 - s < NBW - 2 ? nstages - 1 - s : NBW - 2;      // later stages already issued
            if (ahead >= 2)      asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * LOPS) : "memory");
            else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(LOPS) : "memory");
            else                 asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // (NBT = 2: the columns of stage s were this wave's newest DMA)
        SK_TM_MARK(0);
        __syncthreads();
        SK_TM_MARK(1);
        if (loader) { if (s + NBW - 1 < nstages && !(dbg & 8)) issue_weights(s + NBW - 1, wid); }
        else        { if (s + NBT - 1 < nstages && !(dbg & 4)) issue_tokens(s + NBT - 1, cw); }
        const uint8_t * W = wbuf(s), * T = tbuf(s);
        if (!loader) {
            // the tokens' scales of the stage, transposed to [group][token] (a lane needs 4 consecutive tokens of one group)
            for (int e = cw * 64 + lane; e < SK_GS * SK_TN; e += NCW * 64) {
                const int gi = e >> 4, tok = e & 15;
                const uint8_t * tp = T + tok * SK_TOKB + 1024;
                const float d = ((const float *) tp)[gi];
                const uint32_t aux = ((const uint32_t *)(tp + 128 + aux_delta))[gi];
                ((float *) dxT)[e] = d;
                uint32_t cv;
                if constexpr (TYPE == FQ_Q4_0)      cv = (uint32_t)(-8 * (int32_t) aux);         // sum (nib - 8) x = sum nib x - 8 sum x
                else if constexpr (TYPE == FQ_Q5_0) cv = (uint32_t)(-16 * (int32_t) aux);
                else if constexpr (F::HAS_MIN)      cv = aux;                                       // y.s (f32)
                else                                cv = 0u;
                ((uint32_t *) ciT)[e] = cv;
            }
        }
        SK_TM_MARK(2);
        __syncthreads();
        SK_TM_MARK(3);
        if (loader) continue;
        const int ng = nblk - SK_GS * s < SK_GS ? nblk - SK_GS * s : SK_GS;
        const int cst = s / F::SPC, remc = nblk - F::CB * cst, nbcs = remc < F::CB ? remc : F::CB;
        const int p2d = (nbcs * (F::QB + F::PB1)) & 15;                    // plane 2's offset from the boundary its DMA started at
        const uint8_t * wr = W + (16 * tile + l16) * F::ROWB;              // the lane's weight row
        const uint8_t * tq = T + l16 * SK_TOKB;                            // the lane's token (A operand)
        // ---- one group = operands out of LDS (8 k-bytes per lane, k = 8 kq + b on both sides) -> matrix instruction -> f32 epilogue.
        // The three steps of consecutive groups are software-pipelined by hand (hipcc keeps them in program order): the reads of group
        // k + 1 are in flight while group k's operand is unpacked and its matrix instruction issued, whose latency group k - 1's scaling covers.
        struct sk_ops { sk_v2i xa, raw; uint32_t s1, s2; float4 dx4, sx4; sk_v4i ci4; };
        // (lane bases once per stage; a group's operands sit at compile-time offsets from them in the unrolled loop: no address arithmetic per group)
        // (gi = sw + goff: the wave's share is folded into the bases, goff is a compile-time constant in the unrolled loop)
        const uint8_t * tqb = tq + 8 * kq + 32 * sw, * wrq = wr + (TYPE == FQ_Q8_0 ? 8 * kq + 32 * sw : 8 * (kq & 1) + 16 * sw);
        const uint8_t * dxb = dxT + 16 * kq + sw * SK_TN * 4, * cib = ciT + 16 * kq + sw * SK_TN * 4;
        const uint8_t * wr2 = wr + 2 * sw, * wr4 = wr + 4 * sw;
        auto load_ops = [&](int gi) __attribute__((always_inline)) {
            sk_ops o;
            o.xa = *(const sk_v2i *)(tqb + 32 * gi);
            o.s2 = 0u;
            if constexpr (TYPE == FQ_Q8_0) { o.raw = *(const sk_v2i *)(wrq + 32 * gi); o.s1 = *(const uint16_t *)(wr2 + SK_GS * 32 + 2 * gi); }
            else {
                o.raw = *(const sk_v2i *)(wrq + 16 * gi);
                if constexpr (TYPE == FQ_Q4_0)      o.s1 = *(const uint16_t *)(wr2 + SK_GS * 16 + 2 * gi);
                else if constexpr (TYPE == FQ_Q4_1) o.s1 = *(const uint32_t *)(wr4 + SK_GS * 16 + 4 * gi);
                else {
                    o.s1 = *(const uint32_t *)(wr4 + SK_GS * 16 + 4 * gi);                      // qh
                    if constexpr (TYPE == FQ_Q5_0) o.s2 = *(const uint16_t *)(wr2 + SK_GS * 20 + p2d + 2 * gi);
                    else                           o.s2 = *(const uint32_t *)(wr4 + SK_GS * 20 + p2d + 4 * gi);
                }
            }
            o.dx4 = *(const float4 *)(dxb + gi * SK_TN * 4);
            if constexpr (F::HAS_MIN) { o.sx4 = *(const float4 *)(cib + gi * SK_TN * 4); o.ci4 = sk_v4i{ 0, 0, 0, 0 }; }     // (two typed loads: a bit_cast of the int vector's elements is miscompiled by hipcc 7.2)
            else { o.ci4 = *(const sk_v4i *)(cib + gi * SK_TN * 4); o.sx4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f); }
            return o;
        };
        auto run_mfma = [&](const sk_ops & o, float & dw, float & mw) __attribute__((always_inline)) {
            sk_v2i wb2;
            mw = 0.0f;
            if constexpr (TYPE == FQ_Q8_0) { wb2 = o.raw; dw = fq_h2f((uint16_t) o.s1); }
            else {
                const int sh = 4 * (kq >> 1);                              // elements 0..15: low nibbles, 16..31: high nibbles (ggml.c:1509-1601)
                wb2 = sk_v2i{ (int)(((uint32_t) o.raw.x >> sh) & 0x0F0F0F0Fu), (int)(((uint32_t) o.raw.y >> sh) & 0x0F0F0F0Fu) };
                if constexpr (TYPE == FQ_Q4_0) dw = fq_h2f((uint16_t) o.s1);
                else if constexpr (TYPE == FQ_Q4_1) { dw = fq_h2f((uint16_t) o.s1); mw = fq_h2f((uint16_t)(o.s1 >> 16)); }
                else {
                    const uint32_t hb = o.s1 >> (8 * kq);                  // bit e of qh = 5th bit of element e
                    wb2.x |= (int)(spread4(hb) << 4); wb2.y |= (int)(spread4(hb >> 4) << 4);
                    dw = fq_h2f((uint16_t) o.s2);
                    if constexpr (TYPE == FQ_Q5_1) mw = fq_h2f((uint16_t)(o.s2 >> 16));
                }
            }
            sk_v4i c = o.ci4;                                              // (Q4_0 / Q5_0: -8 / -16 times the token's block sum; else 0)
            return __builtin_amdgcn_mfma_i32_16x16x32_i8(__builtin_bit_cast(long, o.xa), __builtin_bit_cast(long, wb2), c, 0, 0, 0);
        };
        // the reference's scalar per-block expression per result (as k_gemm_q)
        auto scale = [&](const sk_v4i & c, const sk_ops & o, float dw, float mw) __attribute__((always_inline)) {
            const float dxv[4] = { o.dx4.x, o.dx4.y, o.dx4.z, o.dx4.w };
            const float sxv[4] = { o.sx4.x, o.sx4.y, o.sx4.z, o.sx4.w };
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float ci = (float) c[r];
                if constexpr (SK_FMA && S > 1) {                                        // K-split partial sums: the AVX2 build's fused form, as k_gemm_q<S > 1> (kernels_gemm.hip GQ_FMA)
                    float a_ = __builtin_fmaf(dw * dxv[r], ci, acc[r]);
                    if constexpr (F::HAS_MIN) a_ = __builtin_fmaf(mw, sxv[r], a_);
                    acc[r] = a_;
                } else {
                    float t;
                    if constexpr (TYPE == FQ_Q4_0)      t = (ci * dw) * dxv[r];                                      // ggml.c:2606
                    else if constexpr (!F::HAS_MIN)     t = (dw * dxv[r]) * ci;                                      // ggml.c:2972, 3325
                    else                                t = (dw * dxv[r]) * ci + mw * sxv[r];                       // ggml.c:2731, 3227
                    acc[r] = acc[r] + t;
                }
            }
        };
        if (dbg & 16) continue;
        if (ng == SK_GS) {
            constexpr int NG = SK_GS / S;
            sk_ops o[NG]; sk_v4i c[NG]; float dwv[NG], mwv[NG];
            o[0] = load_ops(0);
#pragma unroll
            for (int k = 0; k < NG; ++k) {
                if (k + 1 < NG) o[k + 1] = load_ops((k + 1) * S);
                c[k] = run_mfma(o[k], dwv[k], mwv[k]);
                __builtin_amdgcn_sched_barrier(0);                         // (the matrix instruction first: the scaling below hides its latency)
                if (k > 0) scale(c[k - 1], o[k - 1], dwv[k - 1], mwv[k - 1]);
                __builtin_amdgcn_sched_barrier(0);
            }
            scale(c[NG - 1], o[NG - 1], dwv[NG - 1], mwv[NG - 1]);
        } else {
            for (int gi = sw; gi < ng; gi += S) {
                const sk_ops o = load_ops(gi - sw);
                float dw, mw;
                const sk_v4i c = run_mfma(o, dw, mw);
                scale(c, o, dw, mw);
            }
        }
        SK_TM_MARK(4);
    }
    if ((dbg & 128) && blockIdx.x == 1 && cw == 0 && lane == 0)
        printf("skinny wg 1 consumer 0: %d stages; cycles (100 MHz): dma wait %llu, barrier 1 %llu, issue + prep %llu, barrier 2 %llu, compute %llu\n", nstages, tm[0], tm[1], tm[2], tm[3], tm[4]);
    if (loader) return;
    // ---- the S partial sums of a tile: ((P0 + P1) + P2) + P3, through LDS (the stage buffers are free: every wave is past the loop)
    if constexpr (S > 1) {
        float * xch = (float *) smem;                                      // [tile][lane][4]
        for (int r = 1; r < S; ++r) {
            asm volatile("s_barrier" ::: "memory");                        // (consumer waves only: the loader has left; s_barrier counts the waves still alive)
            if (sw == r) *(float4 *)(xch + (tile * 64 + lane) * 4) = make_float4(acc[0], acc[1], acc[2], acc[3]);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (sw == 0) {
                const float4 p = *(const float4 *)(xch + (tile * 64 + lane) * 4);
                acc[0] = acc[0] + p.x; acc[1] = acc[1] + p.y; acc[2] = acc[2] + p.z; acc[3] = acc[3] + p.w;
            }
        }
        if (sw != 0) return;
    }
    const int64_t m = m0 + 16 * tile + l16;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int n = 4 * kq + r;
        if (n < N && m < M) {
            float v = acc[r];
            if (ep.mode == FQ_EPI_GELU)      v = h2f_bits(ep.gelu_table[f2h_bits(v)]);
            else if (ep.mode == FQ_EPI_ADD2) v = (v + ep.add1[n * ep.ld_add + m]) + ep.add2[n * ep.ld_add + m];
            dst[n * ldd + m] = v;
        }
    }
}

// =============================================================================================== resident columns, persistent workgroups
// The same arithmetic with the N activation columns RESIDENT in LDS for the whole launch (K <= ~4.6 k: the quants of 16 columns are 73 KiB,
// their scales transposed once) and one workgroup per CU that walks its 32-row pairs with the weight pipeline running across pair
// boundaries: no per-stage column DMA (the largest single cost of the form above, and it ADDS to the rest), no per-stage transposition,
// one barrier per stage, no pipeline fill per 32 rows. The S partial sums of a pair meet through LDS at the next stage's barrier.
template <int TYPE> struct sk_res {
    typedef sk_fmt<TYPE> F;
    static __host__ __device__ int tqs(int nblk) { const int q = nblk * 32; return q + ((16 - (q & 255)) & 255); }      // column stride: = 16 mod 256 -> the 16 columns' operand reads fall into distinct banks
    static __host__ __device__ size_t lds(int nblk, int S, int NBW) {       // NBW weight stages: 3, or 2 for the formats with wider rows
        return (size_t) NBW * SK_TM * F::ROWB + (size_t) SK_TN * tqs(nblk) + 2 * (size_t) nblk * SK_TN * 4 + (S > 1 ? 2 * (size_t)(S - 1) * 2 * 64 * 16 : 0);
    }
};

template <int TYPE, int S, int NBW>
__global__ void __launch_bounds__(64 * (2 + 2 * S)) k_gemm_skinny_res(fq_weight w, fq_weight w1, fq_act act, int N, float * dst, int64_t ldd, fq_gemv_epi ep, float * dst1, int64_t ldd1, fq_gemv_epi ep1, int dbg) {
    // two matrices of the same K and format in one launch (w1.M == 0: one): the 32-row pairs of w, then those of w1, share the resident columns
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    typedef sk_fmt<TYPE> F;
    constexpr int ACT = fq_act_of(TYPE);
    constexpr int NCW = 2 * S, NLW = 2;
    constexpr int WSTAGE = SK_TM * F::ROWB;
    constexpr int WOPS = (F::NL + 63) / 64;
    constexpr int LOPS = 16 * WOPS;
    static_assert(LOPS <= 63, "vmcnt range");
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t K = w.K;
    const int nblk = (int) w.nblk;
    const int nstages = (nblk + SK_GS - 1) / SK_GS;
    const int pairs0 = (int)((w.M + SK_TM - 1) / SK_TM);
    const int npairs = pairs0 + (int)((w1.M + SK_TM - 1) / SK_TM);
    const int mine = ((int) blockIdx.x < npairs) ? (npairs - 1 - (int) blockIdx.x) / (int) gridDim.x + 1 : 0;      // pairs blockIdx.x, + gridDim.x, ..
    const int nq = mine * nstages;                                         // stages this workgroup runs
    const size_t img = fq_act_col_bytes(ACT, K);
    const int TQS = sk_res<TYPE>::tqs(nblk);
    uint8_t * tqb0 = smem + (size_t) NBW * WSTAGE;                         // [16 columns][TQS]: the quants
    uint8_t * dxT  = tqb0 + (size_t) SK_TN * TQS;                          // [nblk][16] f32: the columns' d
    uint8_t * ciT  = dxT + (size_t) nblk * SK_TN * 4;                      // [nblk][16]: C start values (int) or the columns' s (f32)
    float   * xch  = (float *)(ciT + (size_t) nblk * SK_TN * 4);           // [2][S - 1][tile][lane][4]: partial sums of a finished pair
    auto wbuf = [&](int q) { return smem + (size_t)(q % NBW) * WSTAGE; };

    auto issue_weights = [&](int q, int lw) {                                // stage q = (pair q / nstages, blocks [32 s, 32 s + 32)); loader wave lw: rows 16 lw ..
        const int s = q % nstages;
        const int P = (int) blockIdx.x + (q / nstages) * (int) gridDim.x;    // pair index over both matrices
        const bool second = P >= pairs0;
        const int64_t m0 = (int64_t)(second ? P - pairs0 : P) * SK_TM, M = second ? w1.M : w.M;
        const uint8_t * plane0 = second ? w1.plane[0] : w.plane[0];
        const int64_t rstride = second ? w1.row_stride : w.row_stride;
        const int c = s / F::SPC, hf = s % F::SPC;
        const int rem = nblk - F::CB * c, nbc = rem < F::CB ? rem : F::CB;
        const unsigned colb = (unsigned) c * (unsigned)(F::CB * F::D.tsize);
        auto src_of = [&](int l) {
            const int p = 16 * l;
            unsigned o;
            if (p < SK_GS * F::QB)                   o = colb + (unsigned)(SK_GS * hf * F::QB + p);
            else if (p < SK_GS * (F::QB + F::PB1))   o = colb + (unsigned)(nbc * F::QB + SK_GS * hf * F::PB1 + (p - SK_GS * F::QB));
            else                                     o = colb + (unsigned)(((nbc * (F::QB + F::PB1)) & ~15) + SK_GS * hf * F::PB2 + (p - SK_GS * (F::QB + F::PB1)));
            return o;
        };
        const unsigned v0 = src_of(lane), v1 = src_of(lane + 64 < F::NL ? lane + 64 : F::NL - 1);
        const unsigned wb = __builtin_amdgcn_readfirstlane(sk_lds(wbuf(q)));
        if (m0 + SK_TM <= M) {
            const uint8_t * base = sk_uniform(plane0 + (size_t) m0 * rstride);
            const unsigned rs = (unsigned) rstride;
            sk_voff16 o;
#pragma unroll
            for (int i = 0; i < 16; ++i) o.v[i] = v0 + (unsigned)(16 * lw + i) * rs;
            unsigned ml = wb + (unsigned)(16 * lw * F::ROWB);
            if (lane < (F::NL < 64 ? F::NL : 64)) sk_dma16(base, o, ml, (unsigned) F::ROWB);
            if constexpr (WOPS > 1) {
#pragma unroll
                for (int i = 0; i < 16; ++i) o.v[i] = v1 + (unsigned)(16 * lw + i) * rs;
                unsigned ml2 = wb + (unsigned)(16 * lw * F::ROWB + 1024);
                if (lane + 64 < F::NL) sk_dma16(base, o, ml2, (unsigned) F::ROWB);
            }
            return;
        }
        for (int r = 16 * lw; r < 16 * lw + 16; ++r) {                     // the matrix's last, partial pair: rows beyond M re-read row M - 1
            const int64_t row = m0 + r < M ? m0 + r : M - 1;
            const uint8_t * base = sk_uniform(plane0 + (size_t) row * rstride);
            if (lane < (F::NL < 64 ? F::NL : 64)) sk_dma(base, v0, wb + (unsigned)(r * F::ROWB));
            if constexpr (WOPS > 1) { if (lane + 64 < F::NL) sk_dma(base, v1, wb + (unsigned)(r * F::ROWB + 1024)); }
        }
    };

    const bool loader = wid < NLW;
    const int cw = wid - NLW, tile = cw & 1, sw = cw >> 1;
    const int l16 = lane & 15, kq = lane >> 4;
    if (nq == 0) return;

    // ---- prologue: the loaders start the weight pipeline; the consumers bring in the columns (quants by DMA, 1 KiB per instruction) and
    // write the transposed scales
    if (loader) { for (int q = 0; q < NBW - 1 && q < nq; ++q) if (!(dbg & 8)) issue_weights(q, wid); }
    else {
        const unsigned last = (unsigned)(img - 16);
        const int kbytes = nblk * 32;
        for (int t = cw; t < SK_TN; t += NCW) {
            const uint8_t * base = act.base + (size_t)(t < N ? t : N - 1) * img;
            const unsigned tb = sk_lds(tqb0) + (unsigned)(t * TQS);
            for (int o = 0; o < kbytes; o += 1024) {
                unsigned vq = (unsigned)(o + 16 * lane);
                vq = vq < last ? vq : last;
                if (o + 16 * lane < kbytes && !(dbg & 4)) sk_dma(base, vq, tb + (unsigned) o);
            }
        }
        const size_t nd4 = fq_act_d_elems(ACT, K) * 4;
        for (int e = cw * 64 + lane; e < nblk * SK_TN; e += NCW * 64) {
            const int gi = e >> 4, tok = e & 15;
            const uint8_t * tp = act.base + (size_t)(tok < N ? tok : N - 1) * img + (size_t) K;
            const float d = ((const float *) tp)[gi];
            const uint32_t aux = ((const uint32_t *)(tp + nd4))[gi];
            ((float *) dxT)[e] = d;
            uint32_t cv;
            if constexpr (TYPE == FQ_Q4_0)      cv = (uint32_t)(-8 * (int32_t) aux);         // sum (nib - 8) x = sum nib x - 8 sum x
            else if constexpr (TYPE == FQ_Q5_0) cv = (uint32_t)(-16 * (int32_t) aux);
            else if constexpr (F::HAS_MIN)      cv = aux;                                       // y.s (f32)
            else                                cv = 0u;
            ((uint32_t *) ciT)[e] = cv;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    // this wave's column DMA has landed (the first barrier publishes it)
    }

    float acc[4] = { 0.0f, 0.0f, 0.0f, 0.0f }, fin[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
    // the pair whose last stage ran before the barrier just passed: its partial sums are in xch (and fin): sum, epilogue, store
    auto finish_pair = [&](int i) {
        if (sw != 0) return;
        if constexpr (S > 1) {
            const float * xb = xch + (size_t)(i & 1) * (S - 1) * 2 * 64 * 4;
#pragma unroll
            for (int r = 1; r < S; ++r) {                                   // ((P0 + P1) + P2) + P3
                const float4 p = *(const float4 *)(xb + ((size_t)((r - 1) * 2 + tile) * 64 + lane) * 4);
                fin[0] = fin[0] + p.x; fin[1] = fin[1] + p.y; fin[2] = fin[2] + p.z; fin[3] = fin[3] + p.w;
            }
        }
        const int P = (int) blockIdx.x + i * (int) gridDim.x;
        const bool second = P >= pairs0;
        const int64_t m = (int64_t)(second ? P - pairs0 : P) * SK_TM + 16 * tile + l16, M = second ? w1.M : w.M;
        const fq_gemv_epi & e = second ? ep1 : ep;
        float * d = second ? dst1 : dst;
        const int64_t ld = second ? ldd1 : ldd;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = 4 * kq + r;
            if (n < N && m < M) {
                float v = fin[r];
                if (e.mode == FQ_EPI_GELU)      v = h2f_bits(e.gelu_table[f2h_bits(v)]);
                else if (e.mode == FQ_EPI_ADD2) v = (v + e.add1[n * e.ld_add + m]) + e.add2[n * e.ld_add + m];
                d[n * ld + m] = v;
            }
        }
    };

    for (int q = 0; q < nq; ++q) {
        const int s = q % nstages, i = q / nstages;
        if (loader) {
            const int ahead = nq - 1 - q < NBW - 2 ? nq - 1 - q : NBW - 2;  // later stages already issued
            if (ahead >= 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(LOPS) : "memory");
            else            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();                                                   // stage q has landed; nobody reads stage q - 1 any more; the previous pair's partial sums are written
        if (loader) { if (q + NBW - 1 < nq && !(dbg & 8)) issue_weights(q + NBW - 1, wid); continue; }
        if (s == 0 && i > 0) finish_pair(i - 1);
        const uint8_t * W = wbuf(q);
        const int ng = nblk - SK_GS * s < SK_GS ? nblk - SK_GS * s : SK_GS;
        const int cst = s / F::SPC, remc = nblk - F::CB * cst, nbcs = remc < F::CB ? remc : F::CB;
        const int p2d = (nbcs * (F::QB + F::PB1)) & 15;
        const uint8_t * wr = W + (16 * tile + l16) * F::ROWB;
        struct sk_ops { sk_v2i xa, raw; uint32_t s1, s2; float4 dx4, sx4; sk_v4i ci4; };
        const int g0 = SK_GS * s;                                          // the stage's first block: offsets into the resident columns
        const uint8_t * tqb = tqb0 + (size_t) l16 * TQS + 32 * g0 + 8 * kq + 32 * sw;
        const uint8_t * wrq = wr + (TYPE == FQ_Q8_0 ? 8 * kq + 32 * sw : 8 * (kq & 1) + 16 * sw);
        const uint8_t * dxb = dxT + (size_t) g0 * SK_TN * 4 + 16 * kq + sw * SK_TN * 4, * cib = ciT + (size_t) g0 * SK_TN * 4 + 16 * kq + sw * SK_TN * 4;
        const uint8_t * wr2 = wr + 2 * sw, * wr4 = wr + 4 * sw;
        auto load_ops = [&](int gi) __attribute__((always_inline)) {
            sk_ops o;
            o.xa = *(const sk_v2i *)(tqb + 32 * gi);
            o.s2 = 0u;
            if constexpr (TYPE == FQ_Q8_0) { o.raw = *(const sk_v2i *)(wrq + 32 * gi); o.s1 = *(const uint16_t *)(wr2 + SK_GS * 32 + 2 * gi); }
            else {
                o.raw = *(const sk_v2i *)(wrq + 16 * gi);
                if constexpr (TYPE == FQ_Q4_0)      o.s1 = *(const uint16_t *)(wr2 + SK_GS * 16 + 2 * gi);
                else if constexpr (TYPE == FQ_Q4_1) o.s1 = *(const uint32_t *)(wr4 + SK_GS * 16 + 4 * gi);
                else {
                    o.s1 = *(const uint32_t *)(wr4 + SK_GS * 16 + 4 * gi);
                    if constexpr (TYPE == FQ_Q5_0) o.s2 = *(const uint16_t *)(wr2 + SK_GS * 20 + p2d + 2 * gi);
                    else                           o.s2 = *(const uint32_t *)(wr4 + SK_GS * 20 + p2d + 4 * gi);
                }
            }
            o.dx4 = *(const float4 *)(dxb + gi * SK_TN * 4);
            if constexpr (F::HAS_MIN) { o.sx4 = *(const float4 *)(cib + gi * SK_TN * 4); o.ci4 = sk_v4i{ 0, 0, 0, 0 }; }
            else { o.ci4 = *(const sk_v4i *)(cib + gi * SK_TN * 4); o.sx4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f); }
            return o;
        };
        auto run_mfma = [&](const sk_ops & o, float & dw, float & mw) __attribute__((always_inline)) {
            sk_v2i wb2;
            mw = 0.0f;
            if constexpr (TYPE == FQ_Q8_0) { wb2 = o.raw; dw = fq_h2f((uint16_t) o.s1); }
            else {
                const int sh = 4 * (kq >> 1);
                wb2 = sk_v2i{ (int)(((uint32_t) o.raw.x >> sh) & 0x0F0F0F0Fu), (int)(((uint32_t) o.raw.y >> sh) & 0x0F0F0F0Fu) };
                if constexpr (TYPE == FQ_Q4_0) dw = fq_h2f((uint16_t) o.s1);
                else if constexpr (TYPE == FQ_Q4_1) { dw = fq_h2f((uint16_t) o.s1); mw = fq_h2f((uint16_t)(o.s1 >> 16)); }
                else {
                    const uint32_t hb = o.s1 >> (8 * kq);
                    wb2.x |= (int)(spread4(hb) << 4); wb2.y |= (int)(spread4(hb >> 4) << 4);
                    dw = fq_h2f((uint16_t) o.s2);
                    if constexpr (TYPE == FQ_Q5_1) mw = fq_h2f((uint16_t)(o.s2 >> 16));
                }
            }
            sk_v4i c = o.ci4;
            return __builtin_amdgcn_mfma_i32_16x16x32_i8(__builtin_bit_cast(long, o.xa), __builtin_bit_cast(long, wb2), c, 0, 0, 0);
        };
        auto scale = [&](const sk_v4i & c, const sk_ops & o, float dw, float mw) __attribute__((always_inline)) {
            const float dxv[4] = { o.dx4.x, o.dx4.y, o.dx4.z, o.dx4.w };
            const float sxv[4] = { o.sx4.x, o.sx4.y, o.sx4.z, o.sx4.w };
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float ci = (float) c[r];
                if constexpr (SK_FMA && S > 1) {                                        // K-split partial sums: the AVX2 build's fused form, as k_gemm_q<S > 1> (kernels_gemm.hip GQ_FMA)
                    float a_ = __builtin_fmaf(dw * dxv[r], ci, acc[r]);
                    if constexpr (F::HAS_MIN) a_ = __builtin_fmaf(mw, sxv[r], a_);
                    acc[r] = a_;
                } else {
                    float t;
                    if constexpr (TYPE == FQ_Q4_0)      t = (ci * dw) * dxv[r];                                      // ggml.c:2606
                    else if constexpr (!F::HAS_MIN)     t = (dw * dxv[r]) * ci;                                      // ggml.c:2972, 3325
                    else                                t = (dw * dxv[r]) * ci + mw * sxv[r];                       // ggml.c:2731, 3227
                    acc[r] = acc[r] + t;
                }
            }
        };
        if (!(dbg & 16)) {
            if (ng == SK_GS) {
                constexpr int NG = SK_GS / S;
                sk_ops o[NG]; sk_v4i c[NG]; float dwv[NG], mwv[NG];
                o[0] = load_ops(0);
#pragma unroll
                for (int k = 0; k < NG; ++k) {
                    if (k + 1 < NG) o[k + 1] = load_ops((k + 1) * S);
                    c[k] = run_mfma(o[k], dwv[k], mwv[k]);
                    __builtin_amdgcn_sched_barrier(0);
                    if (k > 0) scale(c[k - 1], o[k - 1], dwv[k - 1], mwv[k - 1]);
                    __builtin_amdgcn_sched_barrier(0);
                }
                scale(c[NG - 1], o[NG - 1], dwv[NG - 1], mwv[NG - 1]);
            } else {
                for (int gi = sw; gi < ng; gi += S) {
                    const sk_ops o = load_ops(gi - sw);
                    float dw, mw;
                    const sk_v4i c = run_mfma(o, dw, mw);
                    scale(c, o, dw, mw);
                }
            }
        }
        if (s == nstages - 1) {                                            // the pair is complete: hand the partial sums over (read after the next barrier)
            if (sw == 0) { fin[0] = acc[0]; fin[1] = acc[1]; fin[2] = acc[2]; fin[3] = acc[3]; }
            else if constexpr (S > 1)
                *(float4 *)(xch + (size_t)(i & 1) * (S - 1) * 2 * 64 * 4 + ((size_t)((sw - 1) * 2 + tile) * 64 + lane) * 4) = make_float4(acc[0], acc[1], acc[2], acc[3]);
            acc[0] = acc[1] = acc[2] = acc[3] = 0.0f;
        }
    }
    __syncthreads();
    if (!loader) finish_pair(mine - 1);
}

// =============================================================================================== one K share per workgroup (long rows)
// Matrices with long rows and few of them (Falcon's down projection: 4544 x 18176) cannot keep 16 columns of the whole K in LDS, and a
// workgroup per 32 rows reaches only 142 CUs. Here a workgroup owns ONE of the four interleaved K shares (groups g = s, s + 4, ..: the
// association of k_gemm_q<S = 4>) of T 16-row tiles: it needs a quarter of the columns' bytes -- resident for the whole launch -- and every
// consumer wave runs one (tile, share) accumulator chain over all of its groups. The four shares of a row block run on the same XCD
// (workgroup index mod 8), so the 64-byte segments their 16-byte pieces sit in are fetched from HBM once. Partial sums go to a scratch
// ([share][column][row], f32); k_skinny_sum4 adds them as ((P0 + P1) + P2) + P3 and applies the epilogue: bit-identical to k_gemm_q.
// All five legacy formats; stage = a span of 32 blocks = 8 own groups per row: their quant pieces (stride 64 B; Q8_0: 128 B) + the span's scale planes.
// per row and stage (a span of 32 blocks, 8 of them the share's): NQ quant pieces (16 B each; Q8_0: two per group), the span's plane 1 (P1 pieces)
// and plane 2 (P2 pieces, one more than its bytes: a partial column's plane 2 is only 4-byte aligned and is read from the boundary below),
// padded to an ODD number of 16-byte slots (the 16 rows of a tile in distinct LDS banks)
template <int TYPE> struct ks_fmt {
    static constexpr fq_type_desc D = fq_desc(TYPE);
    static constexpr int QB = D.plane[0].bytes, PB1 = D.plane[1].bytes, PB2 = D.nplanes > 2 ? D.plane[2].bytes : 0;
    static constexpr int CB = 1024 / QB;                                   // blocks per column: 64 (Q8_0: 32 = one span)
    static constexpr int NQ = 8 * (QB / 16), P1 = 32 * PB1 / 16, P2 = PB2 ? 32 * PB2 / 16 + 1 : 0;
    static constexpr int NP = NQ + P1 + P2, NPP = (NP % 2) ? NP : NP + 1;
    static constexpr int ROWB = 16 * NPP, WSTAGE = 16 * ROWB, KOPS = (16 * NPP + 63) / 64;
    static constexpr int OFF1 = 16 * NQ, OFF2 = 16 * (NQ + P1);            // byte offsets of the planes inside a row's stage
    static constexpr bool HAS_MIN = (TYPE == FQ_Q4_1 || TYPE == FQ_Q5_1);
    static __host__ __device__ int nj(int nblk, int s) { return (nblk - s + 3) / 4; }                  // own groups of share s
    static __host__ __device__ int tqs(int nblk) { const int q = ((nblk + 3) / 4) * 32; return q + ((16 - (q & 255)) & 255); }
    static __host__ __device__ size_t lds(int nblk, int T, int nbw) {
        return (size_t) nbw * T * WSTAGE + (size_t) SK_TN * tqs(nblk) + 2 * (size_t)((nblk + 3) / 4) * SK_TN * 4;
    }
};

// (bid: the workgroup's index within ITS matrix's part of the grid -- k_gemm_skinny_ks2 runs two matrices in one launch)
template <int TYPE, int NBW>
__device__ __forceinline__ void ks_body(const fq_weight & w, const fq_act & act, int N, float * part, int T, int nrb, int nslots, int dbg, const int bid, uint8_t * smem) {
    typedef ks_fmt<TYPE> F;
    constexpr int ACT = fq_act_of(TYPE);
    constexpr int WSTAGE = F::WSTAGE, KOPS = F::KOPS;
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t K = w.K, M = w.M;
    const int nblk = (int) w.nblk;
    const int nsp = (nblk + 31) / 32;                                      // stages: spans of 32 blocks
    // workgroup -> (row block, share): the four shares of a row block on one XCD
    const int xcd = bid & 7, kk = bid >> 3;
    const int slot = (kk >> 2) * 8 + xcd, s = kk & 3;                     // row blocks slot, slot + nslots, .. (one round when they all fit the chip)
    if (slot >= nrb) return;
    const size_t img = fq_act_col_bytes(ACT, K);
    const int NJ = F::nj(nblk, s), NJMAX = (nblk + 3) / 4;
    const int TQS = F::tqs(nblk);
    uint8_t * tqb0 = smem + (size_t) NBW * WSTAGE * T;                     // [16 columns][TQS]: the quants of the share's groups
    uint8_t * dxT  = tqb0 + (size_t) SK_TN * TQS;                          // [own group][16] f32: the columns' d
    uint8_t * ciT  = dxT + (size_t) NJMAX * SK_TN * 4;                     // [own group][16]: C start values (int) or the columns' s (f32)
    const int NW = (int)(blockDim.x >> 6);
    const int l16 = lane & 15, kq = lane >> 4;
    constexpr int COLB = F::CB * F::D.tsize;                               // bytes of a full column of the device layout

    // ---- prologue, every wave: the columns' own groups (32 bytes each = two lanes; 32 groups per DMA instruction), the transposed scales
    {
        const unsigned last = (unsigned)(img - 16);
        for (int t = wid; t < SK_TN; t += NW) {
            const uint8_t * base = act.base + (size_t)(t < N ? t : N - 1) * img;
            const unsigned tb = sk_lds(tqb0) + (unsigned)(t * TQS);
            for (int j0 = 0; j0 < NJ; j0 += 32) {
                const int j = j0 + (lane >> 1);
                unsigned vq = (unsigned)((s + 4 * j) * 32 + 16 * (lane & 1));
                vq = vq < last ? vq : last;
                if (j < NJ && !(dbg & 4)) sk_dma(base, vq, tb + (unsigned)(j0 * 32));
            }
        }
        const size_t nd4 = fq_act_d_elems(ACT, K) * 4;
        for (int e = tid; e < NJ * SK_TN; e += (int) blockDim.x) {
            const int j = e >> 4, tok = e & 15;
            const uint8_t * tp = act.base + (size_t)(tok < N ? tok : N - 1) * img + (size_t) K;
            const int g = s + 4 * j;
            ((float *) dxT)[e] = ((const float *) tp)[g];
            const uint32_t aux = ((const uint32_t *)(tp + nd4))[g];
            uint32_t cv;
            if constexpr (TYPE == FQ_Q4_0)      cv = (uint32_t)(-8 * (int32_t) aux);         // sum (nib - 8) x = sum nib x - 8 sum x
            else if constexpr (TYPE == FQ_Q5_0) cv = (uint32_t)(-16 * (int32_t) aux);
            else if constexpr (F::HAS_MIN)      cv = aux;                                       // y.s (f32)
            else                                cv = 0u;
            ((uint32_t *) ciT)[e] = cv;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();                                                       // the only barrier: the columns are in LDS
    if (wid >= T) return;
  for (int rb = slot; rb < nrb; rb += nslots) {
    const int64_t mt = ((int64_t) rb * T + wid) * 16;                      // the wave's tile: it stages its own weights and runs its own chain
    if (mt >= M) break;

    // ---- the wave's weight pipeline: per stage (span sp) 16 rows x NPP slots = KOPS DMA instructions into its private ring, NBW - 1 stages
    // ahead, paced by vmcnt alone. Lane L = 64 k + lane of instruction k is (row L / NPP, slot L % NPP)
    uint8_t * myring = smem + (size_t) wid * NBW * WSTAGE;
    const unsigned ring_lds = __builtin_amdgcn_readfirstlane(sk_lds(myring));
    const uint8_t * wbase = sk_uniform(w.plane[0] + (size_t) mt * w.row_stride);
    unsigned rowoff[KOPS]; int piece[KOPS];
#pragma unroll
    for (int k = 0; k < KOPS; ++k) {
        const int L = 64 * k + lane, row = (L / F::NPP) & 15;                // (lanes beyond 16 NPP of the last instruction are masked off below)
        piece[k] = L % F::NPP < F::NP ? L % F::NPP : F::NP - 1;            // the padding slot re-reads the last piece
        const int64_t r = mt + row < M ? row : M - 1 - mt;                  // rows beyond M re-read row M - 1
        rowoff[k] = (unsigned)(r * (int64_t) w.row_stride);
    }
    const unsigned rs16 = (unsigned) w.row_stride - 16u;
    auto issue = [&](int sp) {
        const int c = sp / (F::CB / 32), half = sp % (F::CB / 32);
        const int rem = nblk - F::CB * c, nbc = rem < F::CB ? rem : F::CB;
        // first byte of the share's first group, of the span's plane 1 and (from the 16-byte boundary below it) of its plane 2
        const unsigned bq = (unsigned)(c * COLB + (32 * half + s) * F::QB);
        const unsigned b1 = (unsigned)(c * COLB + nbc * F::QB + 32 * half * F::PB1);
        const unsigned b2 = (unsigned)(c * COLB + ((nbc * (F::QB + F::PB1)) & ~15) + 32 * half * F::PB2);
        const unsigned dst = ring_lds + (unsigned)((sp % NBW) * WSTAGE);
#pragma unroll
        for (int k = 0; k < KOPS; ++k) {
            const int p = piece[k];
            unsigned o;
            if (p < F::NQ)              o = F::QB == 16 ? bq + 64u * (unsigned) p : bq + 128u * (unsigned)(p >> 1) + 16u * (unsigned)(p & 1);
            else if (p < F::NQ + F::P1) o = b1 + 16u * (unsigned)(p - F::NQ);
            else                        o = b2 + 16u * (unsigned)(p - F::NQ - F::P1);
            o = o < rs16 ? o : rs16;                                       // (a partial last column: pieces beyond its blocks are never used)
            if (64 * k + lane < 16 * F::NPP) sk_dma(wbase, rowoff[k] + o, dst + (unsigned)(k * 1024));
        }
    };
    if (!(dbg & 8)) { for (int q = 0; q < NBW - 1 && q < nsp; ++q) issue(q); }

    float acc[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
    const int sh = 4 * (kq >> 1);
    struct ks_ops { sk_v2i xa, raw; uint32_t s1, s2; float4 dx4, sx4; sk_v4i ci4; };
    for (int sp = 0; sp < nsp; ++sp) {
        if (sp + NBW - 1 < nsp && !(dbg & 8)) issue(sp + NBW - 1);
        // stage sp has landed: at most the stages issued after it (KOPS instructions each) are still in flight
        {
            const int later = (dbg & 8) ? 0 : (nsp - 1 - sp < NBW - 1 ? nsp - 1 - sp : NBW - 1);
            if (later >= 2)      sk_wait_vm_upto(2 * KOPS);
            else if (later == 1) sk_wait_vm_upto(KOPS);
            else                 asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (dbg & 16) continue;
        const uint8_t * wr = myring + (size_t)(sp % NBW) * WSTAGE + l16 * F::ROWB;
        const int jbase = 8 * sp;                                          // own-group index of the span's first own group
        const int njs = NJ - jbase < 8 ? NJ - jbase : 8;                   // own groups in this span
        const int cst = sp / (F::CB / 32), remc = nblk - F::CB * cst, nbcs = remc < F::CB ? remc : F::CB;
        const int p2d = (nbcs * (F::QB + F::PB1)) & 15;                    // plane 2's offset from the boundary its DMA started at
        const uint8_t * tqp = tqb0 + (size_t) l16 * TQS + 32 * jbase + 8 * kq;
        const uint8_t * dxp = dxT + (size_t) jbase * SK_TN * 4 + 16 * kq, * cip = ciT + (size_t) jbase * SK_TN * 4 + 16 * kq;
        const uint8_t * wq = wr + (TYPE == FQ_Q8_0 ? 8 * kq : 8 * (kq & 1));
        const uint8_t * w1 = wr + F::OFF1 + s * F::PB1, * w2 = wr + F::OFF2 + p2d + s * F::PB2;
        auto load_ops = [&](int jj) __attribute__((always_inline)) {
            ks_ops o;
            o.xa = *(const sk_v2i *)(tqp + 32 * jj);
            o.raw = *(const sk_v2i *)(wq + F::QB * jj);
            o.s2 = 0u;
            if constexpr (F::PB1 == 2) o.s1 = *(const uint16_t *)(w1 + 4 * F::PB1 * jj); else o.s1 = *(const uint32_t *)(w1 + 4 * F::PB1 * jj);
            if constexpr (F::PB2 == 2) o.s2 = *(const uint16_t *)(w2 + 4 * F::PB2 * jj); else if constexpr (F::PB2 == 4) o.s2 = *(const uint32_t *)(w2 + 4 * F::PB2 * jj);
            o.dx4 = *(const float4 *)(dxp + jj * SK_TN * 4);
            if constexpr (F::HAS_MIN) { o.sx4 = *(const float4 *)(cip + jj * SK_TN * 4); o.ci4 = sk_v4i{ 0, 0, 0, 0 }; }     // (two typed loads: see k_gemm_skinny)
            else { o.ci4 = *(const sk_v4i *)(cip + jj * SK_TN * 4); o.sx4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f); }
            return o;
        };
        auto run_mfma = [&](const ks_ops & o, float & dw, float & mw) __attribute__((always_inline)) {
            sk_v2i wb2;
            mw = 0.0f;
            if constexpr (TYPE == FQ_Q8_0) { wb2 = o.raw; dw = fq_h2f((uint16_t) o.s1); }
            else {
                wb2 = sk_v2i{ (int)(((uint32_t) o.raw.x >> sh) & 0x0F0F0F0Fu), (int)(((uint32_t) o.raw.y >> sh) & 0x0F0F0F0Fu) };
                if constexpr (TYPE == FQ_Q4_0) dw = fq_h2f((uint16_t) o.s1);
                else if constexpr (TYPE == FQ_Q4_1) { dw = fq_h2f((uint16_t) o.s1); mw = fq_h2f((uint16_t)(o.s1 >> 16)); }
                else {
                    const uint32_t hb = o.s1 >> (8 * kq);                  // bit e of qh = 5th bit of element e
                    wb2.x |= (int)(spread4(hb) << 4); wb2.y |= (int)(spread4(hb >> 4) << 4);
                    dw = fq_h2f((uint16_t) o.s2);
                    if constexpr (TYPE == FQ_Q5_1) mw = fq_h2f((uint16_t)(o.s2 >> 16));
                }
            }
            return __builtin_amdgcn_mfma_i32_16x16x32_i8(__builtin_bit_cast(long, o.xa), __builtin_bit_cast(long, wb2), o.ci4, 0, 0, 0);
        };
        auto scale = [&](const sk_v4i & c, const ks_ops & o, float dw, float mw) __attribute__((always_inline)) {
            const float dxv[4] = { o.dx4.x, o.dx4.y, o.dx4.z, o.dx4.w };
            const float sxv[4] = { o.sx4.x, o.sx4.y, o.sx4.z, o.sx4.w };
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float ci = (float) c[r];
                if constexpr (SK_FMA) {                                        // K-split partial sums: the AVX2 build's fused form, as k_gemm_q<S > 1> (kernels_gemm.hip GQ_FMA)
                    float a_ = __builtin_fmaf(dw * dxv[r], ci, acc[r]);
                    if constexpr (F::HAS_MIN) a_ = __builtin_fmaf(mw, sxv[r], a_);
                    acc[r] = a_;
                } else {
                    float t;
                    if constexpr (TYPE == FQ_Q4_0)      t = (ci * dw) * dxv[r];                                      // ggml.c:2606
                    else if constexpr (!F::HAS_MIN)     t = (dw * dxv[r]) * ci;                                      // ggml.c:2972, 3325
                    else                                t = (dw * dxv[r]) * ci + mw * sxv[r];                       // ggml.c:2731, 3227
                    acc[r] = acc[r] + t;
                }
            }
        };
        if (njs == 8) {
            ks_ops o[8]; sk_v4i c[8]; float dwv[8], mwv[8];
            o[0] = load_ops(0); o[1] = load_ops(1);
#pragma unroll
            for (int k = 0; k < 8; ++k) {                                   // reads two groups ahead, the matrix instruction one ahead of its scaling
                if (k + 2 < 8) o[k + 2] = load_ops(k + 2);
                c[k] = run_mfma(o[k], dwv[k], mwv[k]);
                __builtin_amdgcn_sched_barrier(0);
                if (k > 0) scale(c[k - 1], o[k - 1], dwv[k - 1], mwv[k - 1]);
                __builtin_amdgcn_sched_barrier(0);
            }
            scale(c[7], o[7], dwv[7], mwv[7]);
        } else {
            for (int jj = 0; jj < njs; ++jj) { const ks_ops o = load_ops(jj); float dw, mw; const sk_v4i c = run_mfma(o, dw, mw); scale(c, o, dw, mw); }
        }
    }
    const int64_t m = mt + l16;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int n = 4 * kq + r;
        if (n < N && m < M) part[((size_t) s * SK_TN + n) * FQ_KS_MAX_M + m] = acc[r];
    }
  }
}
template <int TYPE, int NBW>
__global__ void __launch_bounds__(64 * KS_TMAX) k_gemm_skinny_ks(fq_weight w, fq_act act, int N, float * part, int T, int nrb, int nslots, int dbg) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    ks_body<TYPE, NBW>(w, act, N, part, T, nrb, nslots, dbg, (int) blockIdx.x, smem);
}
// Two matrices of one format and row count behind their own columns in ONE launch (round 6: Wdown and Wo of a block, x = (Wdown a_ff + Wo a_att) + x): workgroups [0, g0) are
// matrix 0's grid, [g0, g0 + g1) matrix 1's (g0 a multiple of 8: the XCD of a workgroup index does not change), each with its own tiles-per-workgroup and LDS carve-up.
// The shorter matrix's workgroups follow the longer one's onto the CUs as those retire -- no launch boundary, no second grid of 142 workgroups on 256 CUs.
template <int TYPE, int NBW>
__global__ void __launch_bounds__(64 * KS_TMAX) k_gemm_skinny_ks2(fq_weight w0, fq_act act0, float * part0, int T0, int nrb0, int nslots0, int g0,
                                                                 fq_weight w1, fq_act act1, float * part1, int T1, int nrb1, int nslots1, int N, int dbg) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    if ((int) blockIdx.x < g0) ks_body<TYPE, NBW>(w0, act0, N, part0, T0, nrb0, nslots0, dbg, (int) blockIdx.x, smem);
    else                       ks_body<TYPE, NBW>(w1, act1, N, part1, T1, nrb1, nslots1, dbg, (int) blockIdx.x - g0, smem);
}

// the resident form for one matrix (w1.M == 0) or two of the same K and format sharing their columns (e.g. Wqkv and Wup behind one LayerNorm)
static bool fq_launch_gemm_skinny_res(const fq_weight & w, const fq_weight & w1, const fq_act & act, int64_t N, float * dst, int64_t ldd, const fq_gemv_epi & ep,
                                      float * dst1, int64_t ldd1, const fq_gemv_epi & ep1, int S, hipStream_t st) {
    static const bool use_res = !(getenv("FQ_SKINNY_RES") && atoi(getenv("FQ_SKINNY_RES")) == 0);
    if (!use_res) return false;
    size_t need = 0; int nbw = 0;
    for (int n : { 3, 2 }) {
        switch (w.type) {
            case FQ_Q4_0: need = sk_res<FQ_Q4_0>::lds((int) w.nblk, S, n); break; case FQ_Q4_1: need = sk_res<FQ_Q4_1>::lds((int) w.nblk, S, n); break;
            case FQ_Q5_0: need = sk_res<FQ_Q5_0>::lds((int) w.nblk, S, n); break; case FQ_Q5_1: need = sk_res<FQ_Q5_1>::lds((int) w.nblk, S, n); break;
            case FQ_Q8_0: need = sk_res<FQ_Q8_0>::lds((int) w.nblk, S, n); break;
            default: return false;
        }
        if (need <= 160 * 1024) { nbw = n; break; }
    }
    if (!nbw) return false;
    const int npairs = (int)((w.M + SK_TM - 1) / SK_TM) + (int)((w1.M + SK_TM - 1) / SK_TM);
    const unsigned g = (unsigned)(npairs < fq_ctx().n_cu ? npairs : fq_ctx().n_cu);
#define FQ_SKR_LAUNCH(T, SS, NB) { \
        static bool set = false; \
        if (!set) { HIP_CHECK(hipFuncSetAttribute((const void *) k_gemm_skinny_res<T, SS, NB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); set = true; } \
        hipLaunchKernelGGL((k_gemm_skinny_res<T, SS, NB>), dim3(g), dim3(64 * (2 + 2 * SS)), need, st, w, w1, act, (int) N, dst, ldd, ep, dst1, ldd1, ep1, fq_gemm_debug_get()); }
#define FQ_SKR_S(T, NB) if (S == 1) FQ_SKR_LAUNCH(T, 1, NB) else if (S == 2) FQ_SKR_LAUNCH(T, 2, NB) else FQ_SKR_LAUNCH(T, 4, NB)
#define FQ_SKR_CASE(T) case T: if (nbw == 3) { FQ_SKR_S(T, 3) } else { FQ_SKR_S(T, 2) } break;
    switch (w.type) {
        FQ_SKR_CASE(FQ_Q4_0) FQ_SKR_CASE(FQ_Q4_1) FQ_SKR_CASE(FQ_Q5_0) FQ_SKR_CASE(FQ_Q5_1) FQ_SKR_CASE(FQ_Q8_0)
        default: return false;
    }
#undef FQ_SKR_CASE
#undef FQ_SKR_S
#undef FQ_SKR_LAUNCH
    return true;
}

// two matrices behind the same N <= 16 activation columns in one launch (same format, same K, the same K split S); false: nothing launched
bool fq_launch_gemm_skinny_pair(const fq_weight & w0, const fq_weight & w1, const fq_act & act, int64_t N, float * dst0, int64_t ldd0, const fq_gemv_epi & ep0,
                                float * dst1, int64_t ldd1, const fq_gemv_epi & ep1, int S, hipStream_t st) {
    FQ_TL(st, "gemm_skinny_pair");
    if (N < 1 || N > SK_TN || (S != 1 && S != 2 && S != 4) || w0.type != w1.type || w0.K != w1.K || w0.nblk != w1.nblk) return false;
    return fq_launch_gemm_skinny_res(w0, w1, act, N, dst0, ldd0, ep0, dst1, ldd1, ep1, S, st);
}

// the K-share form's plan for a matrix: tiles per workgroup, row blocks, slots, LDS bytes with nbw weight stages
struct ks_plan { int T, nrb, nslots; unsigned grid; };
static size_t ks_need(int type, int nblk, int T, int nbw) {
    switch (type) {
        case FQ_Q4_0: return ks_fmt<FQ_Q4_0>::lds(nblk, T, nbw); case FQ_Q4_1: return ks_fmt<FQ_Q4_1>::lds(nblk, T, nbw);
        case FQ_Q5_0: return ks_fmt<FQ_Q5_0>::lds(nblk, T, nbw); case FQ_Q5_1: return ks_fmt<FQ_Q5_1>::lds(nblk, T, nbw);
        default:      return ks_fmt<FQ_Q8_0>::lds(nblk, T, nbw);
    }
}
static ks_plan ks_make_plan(const fq_weight & w, int n_cu) {
    const int ntiles = (int)((w.M + 15) / 16);
    int T = (ntiles * 4 + n_cu - 1) / n_cu;
    if (T < 1) T = 1;
    if (T > KS_TMAX) T = KS_TMAX;
    const int nrb = (ntiles + T - 1) / T;
    int nslots = 8 * ((nrb + 7) / 8);
    if (4 * nslots > n_cu) nslots = (n_cu / 32) * 8 > 8 ? (n_cu / 32) * 8 : 8;
    return ks_plan{ T, nrb, nslots, (unsigned)(4 * nslots) };
}

// x = (Wdown a_ff + Wo a_att) + x (libfalcon.cpp:2394-2400) for 5..16 columns of a legacy format: BOTH matrices in the K-share form in ONE launch (k_gemm_skinny_ks2) and one
// sum launch that adds each matrix's four partial sums as ((P0 + P1) + P2) + P3 and then (down + wo) + x -- the bits of the two separate mat-muls (Wo's four-share sums are
// k_gemm_q<S = 4>'s whichever form computes them), minus Wo's own launch of 142 workgroups, its result matrix and a launch boundary. false: nothing launched.
bool fq_launch_gemm_skinny_out2(const fq_weight & wo, const fq_act & a_att, const fq_weight & down, const fq_act & a_ff, int64_t N, float * x, int64_t ldx, hipStream_t st) {
    FQ_TL(st, "gemm_skinny_out2");
    static const bool on = !(getenv("FQ_SKINNY_OUT2") && atoi(getenv("FQ_SKINNY_OUT2")) == 0) && !(getenv("FQ_SKINNY_KS") && atoi(getenv("FQ_SKINNY_KS")) == 0);
    const int t = wo.type;
    if (!on || N < 1 || N > SK_TN || t != down.type || wo.M != down.M || wo.M > FQ_KS_MAX_M) return false;
    if (t != FQ_Q4_0 && t != FQ_Q4_1 && t != FQ_Q5_0 && t != FQ_Q5_1 && t != FQ_Q8_0) return false;
    if (down.nblk < 256 || wo.nblk < 32 || a_att.type != a_ff.type) return false;      // (Wdown alone would not take this form: leave the pair to the generic path)
    const int n_cu = fq_ctx().n_cu;
    const ks_plan pd = ks_make_plan(down, n_cu), pw = ks_make_plan(wo, n_cu);
    int nbw = 0; size_t need = 0;
    for (int n : { 3, 2 }) {
        const size_t nd = ks_need(t, (int) down.nblk, pd.T, n), nw = ks_need(t, (int) wo.nblk, pw.T, n);
        need = nd > nw ? nd : nw;
        if (need <= 160 * 1024) { nbw = n; break; }
    }
    if (!nbw) return false;
    float * part_d = fq_ctx().ks_scratch, * part_w = part_d + (size_t) 4 * SK_TN * FQ_KS_MAX_M;
    if ((size_t) 8 * SK_TN * FQ_KS_MAX_M > (size_t) FQ_KS_FLOATS) return false;
#define FQ_KS2_LAUNCH(TT, NB) { \
        static bool set = false; \
        if (!set) { HIP_CHECK(hipFuncSetAttribute((const void *) k_gemm_skinny_ks2<TT, NB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); set = true; } \
        hipLaunchKernelGGL((k_gemm_skinny_ks2<TT, NB>), dim3(pd.grid + pw.grid), dim3(64 * KS_TMAX), need, st, down, a_ff, part_d, pd.T, pd.nrb, pd.nslots, (int) pd.grid, \
                           wo, a_att, part_w, pw.T, pw.nrb, pw.nslots, (int) N, fq_gemm_debug_get()); }
#define FQ_KS2_CASE(TT) case TT: if (nbw == 3) FQ_KS2_LAUNCH(TT, 3) else FQ_KS2_LAUNCH(TT, 2) break;
    switch (t) {
        FQ_KS2_CASE(FQ_Q4_0) FQ_KS2_CASE(FQ_Q4_1) FQ_KS2_CASE(FQ_Q5_0) FQ_KS2_CASE(FQ_Q5_1) FQ_KS2_CASE(FQ_Q8_0)
        default: return false;
    }
#undef FQ_KS2_CASE
#undef FQ_KS2_LAUNCH
    fq_launch_skinny_sum4_out2(part_d, 1, part_w, 1, (int64_t) FQ_KS_MAX_M, N, down.M, x, ldx, st);
    return true;
}

// true (and launched) when the shape is this kernel's: a legacy format, 5 <= N <= 16; S = the K split k_gemm_q would use
bool fq_launch_gemm_skinny(const fq_weight & w, const fq_act & act, int64_t N, float * dst, int64_t ldd, const fq_gemv_epi & ep, int S, hipStream_t st) {
    FQ_TL(st, "gemm_skinny");
    if (N < 1 || N > SK_TN || (S != 1 && S != 2 && S != 4)) return false;
    if (w.type == FQ_Q4_K || w.type == FQ_Q5_K || w.type == FQ_Q2_K || w.type == FQ_Q3_K || w.type == FQ_Q6_K) return fq_launch_gemm_skinny_kq(w, act, N, dst, ldd, ep, S, st);
    if (w.type != FQ_Q4_0 && w.type != FQ_Q4_1 && w.type != FQ_Q5_0 && w.type != FQ_Q5_1 && w.type != FQ_Q8_0) return false;
    // the columns resident in LDS, one persistent workgroup per CU, when they fit (K up to ~4.6 k for Q4_0; FQ_SKINNY_RES=0: never)
    {
        fq_weight none = w; none.M = 0;
        if (fq_launch_gemm_skinny_res(w, none, act, N, dst, ldd, ep, nullptr, 0, ep, S, st)) return true;
    }
    // long rows (the columns do not fit): one K share per workgroup (Q4_0, four partial sums; FQ_SKINNY_KS=0: never)
    static const bool use_ks = !(getenv("FQ_SKINNY_KS") && atoi(getenv("FQ_SKINNY_KS")) == 0);
    if (use_ks && S == 4 && w.M <= FQ_KS_MAX_M && w.nblk >= 256) {
        const int n_cu = fq_ctx().n_cu;
        const int ntiles = (int)((w.M + 15) / 16);
        int T = (ntiles * 4 + n_cu - 1) / n_cu;                            // tiles per workgroup: all workgroups resident in one round ..
        if (T < 1) T = 1;
        if (T > KS_TMAX) T = KS_TMAX;                                      // .. or full workgroups that walk their row blocks (the columns stay resident)
        size_t need = 0; int nbw = 0;
        for (int n : { 3, 2 }) {
            switch (w.type) {
                case FQ_Q4_0: need = ks_fmt<FQ_Q4_0>::lds((int) w.nblk, T, n); break; case FQ_Q4_1: need = ks_fmt<FQ_Q4_1>::lds((int) w.nblk, T, n); break;
                case FQ_Q5_0: need = ks_fmt<FQ_Q5_0>::lds((int) w.nblk, T, n); break; case FQ_Q5_1: need = ks_fmt<FQ_Q5_1>::lds((int) w.nblk, T, n); break;
                default:      need = ks_fmt<FQ_Q8_0>::lds((int) w.nblk, T, n); break;
            }
            if (need <= 160 * 1024) { nbw = n; break; }
        }
        if (nbw) {
            const int nrb = (ntiles + T - 1) / T;
            int nslots = 8 * ((nrb + 7) / 8);
            if (4 * nslots > n_cu) nslots = (n_cu / 32) * 8 > 8 ? (n_cu / 32) * 8 : 8;      // (never an empty grid on a part with fewer than 32 CUs)
            const unsigned g = (unsigned)(4 * nslots);
#define FQ_KS_LAUNCH(TT, NB) { \
                static bool set = false; \
                if (!set) { HIP_CHECK(hipFuncSetAttribute((const void *) k_gemm_skinny_ks<TT, NB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); set = true; } \
                hipLaunchKernelGGL((k_gemm_skinny_ks<TT, NB>), dim3(g), dim3(64 * KS_TMAX), need, st, w, act, (int) N, fq_ctx().ks_scratch, T, nrb, nslots, fq_gemm_debug_get()); }
#define FQ_KS_CASE(TT) case TT: if (nbw == 3) FQ_KS_LAUNCH(TT, 3) else FQ_KS_LAUNCH(TT, 2) break;
            switch (w.type) {
                FQ_KS_CASE(FQ_Q4_0) FQ_KS_CASE(FQ_Q4_1) FQ_KS_CASE(FQ_Q5_0) FQ_KS_CASE(FQ_Q5_1) FQ_KS_CASE(FQ_Q8_0)
                default: return false;
            }
#undef FQ_KS_CASE
#undef FQ_KS_LAUNCH
            fq_launch_skinny_sum4(fq_ctx().ks_scratch, N, w.M, dst, ldd, ep, (int64_t) FQ_KS_MAX_M, 1, st);
            return true;
        }
    }
    const unsigned grid = (unsigned)((w.M + SK_TM - 1) / SK_TM);
#define FQ_SK_LAUNCH(T, SS) { \
        typedef sk_fmt<T> F; \
        const size_t lds = (size_t) F::LDS; \
        static bool set = false; \
        if (!set) { HIP_CHECK(hipFuncSetAttribute((const void *) k_gemm_skinny<T, SS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds)); set = true; } \
        hipLaunchKernelGGL((k_gemm_skinny<T, SS>), dim3(grid), dim3(64 * (2 + 2 * SS)), lds, st, w, act, (int) N, dst, ldd, ep, fq_gemm_debug_get()); }
#define FQ_SK_CASE(T) case T: if (S == 1) FQ_SK_LAUNCH(T, 1) else if (S == 2) FQ_SK_LAUNCH(T, 2) else FQ_SK_LAUNCH(T, 4) break;
    switch (w.type) {
        FQ_SK_CASE(FQ_Q4_0) FQ_SK_CASE(FQ_Q4_1) FQ_SK_CASE(FQ_Q5_0) FQ_SK_CASE(FQ_Q5_1) FQ_SK_CASE(FQ_Q8_0)
        default: return false;
    }
#undef FQ_SK_CASE
#undef FQ_SK_LAUNCH
    return true;
}
