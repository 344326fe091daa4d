// kernels_ring.hip -- the RING form of the fused decode launches (gfx950, wave64): the two launches of a block (k_gemv_ln /
// k_attn_out, kernels_decode.hip -- one N = 1 pass of ggml_compute_forward_mul_mat_q_f32, ggml.c:11318-11529, per weight matrix
// of falcon_eval_internal's block, libfalcon.cpp:2160-2400) with the persistent engine's data path inside each launch and the
// hardware's launch boundary between them:
//   * wave 0 of every workgroup is a LOADER: it streams the workgroup's rows into a ring of LDS with global_load_lds_dwordx4
//     (no registers, no decode), bounded by ring space only -- the stream runs THROUGH the LayerNorm / image prologue, which in the
//     register-streaming kernels is 3.5 us per launch with one pre-issued unit column per row in flight and nothing behind it;
//   * wave 1 runs the epilogues (GELU table, Q8 block, stores), waves 2..11 dot the rows out of the ring (fq_engine_dev.h: the
//     arithmetic and per-lane unit order of every other path -> the same bits);
//   * no workgroup barrier after the roles split, no cross-workgroup hand-off that the two-launch form does not have.
// The loader starts AFTER the helpers have issued their loads of the residual row: a CU's memory pipeline serves requests in order,
// and the row would otherwise queue behind 48 KiB of weight pieces (the engine's 8 us "x chunk in").
// Scope: legacy formats, one norm or two, grids of one workgroup per CU; anything else -> false, the caller launches k_gemv_ln.
#include "fq_block_dev.h"
#include "kernels.h"
#include "hip_context.h"
#include <hip/hip_ext.h>
#include <vector>
#include "fq_engine_dev.h"
#include "fq_ref_chain.h"

namespace {

// control words of this kernel beyond eng_ctl's (which it shares: PSUM, STAT, OUT, CNT, LANDED, LOW, XG_DONE, LN_STAT, IMG_DONE, LN_MEAN, S2_DONE, PSQ)
constexpr unsigned RING_XISSUED = eng_ctl::A_DONE;    // helper waves that have issued their loads of the residual row
constexpr unsigned RING_CTL_BYTES = eng_ctl::PTRS + 64;
constexpr unsigned RING_CTL_BYTES16 = (RING_CTL_BYTES + 15u) & ~15u;
// REF (fast reference order, fq_ref_chain.h): 32-row groups the epilogue wave has summed -- their strip slots may be written again
constexpr unsigned RING_SUMMED = eng_ctl::FG_DONE;
// waves of a workgroup: the loader, the epilogue wave, RNC consumers. (12 consumers instead of 10 were measured SLOWER, 959 against 988 tok/s: their
// working set leaves the loader less of the ring and the stream ends 1.2 us later -- the consumers' time is latency, not issue slots.)
constexpr int RNH = 11, RNC = RNH - 1, RNT = 64 * (RNH + 1), RHT = 64 * RNH;
// phase stamps (ggml_hip_debug_stamps): rows [role * 256 + workgroup][slot], role 0 = loader, 1 = epilogue wave, 2 = consumer 0, 3 = consumer 9
#define RING_T(role, slot) do { if (a.dbg && lane == 0) a.dbg[((size_t)(role) * 256 + blockIdx.x) * 8 + (slot)] = (long long) wall_clock64(); } while (0)

}   // namespace

struct fq_ring_ln_args {
    const float * x; int E, FF, nblkE; unsigned rsE;
    const uint8_t * qkv, * up;                         // row 0 of the two matrices (device layout)
    int qkv_rows;
    const float * ln_w, * ln_b, * ln2_w, * ln2_b; int two_norms;      // ln feeds Wup (and Wqkv of a one-norm block), ln2 = the attention norm
    float * qkv_dst; uint8_t * ff_image;
    const uint16_t * gelu_tab;
    const fq_engine_sched * sched;                     // per workgroup: rows [qg0, qg1) of Wqkv, 32-row groups [ug0, ug1) of Wup
    unsigned * epoch_word; const int * n_past_ptr; const float * rope_cs; float * rope_cur;
    unsigned * err; int debug_mode; long long * dbg;
    int strip_slots; unsigned strip_stride;            // REF: the term strip holds strip_slots (32 or 64) rows of strip_stride floats
};

// REF = the fast reference order (ggml_hip_reference_order(2), fq_ref_chain.h): the consumers leave every unit's f32 term -- the reference's per-block term --
// in an LDS strip [row slot][block] instead of adding them per lane and across the wave, and the epilogue wave, lane = row, adds each row's terms left to
// right (ggml.c:2591-2609): bit-identical to the reference's scalar build. The strip takes the place of the f32 residual row (dead once the image exists)
// plus the workgroup's unused LDS; the loader, the ring and the LayerNorm prologue are the default form's.
// Kernel arguments (round 6): the words the launch needs FIRST -- the residual row, the schedule entry, the two matrices' first rows, the shape -- are leading scalar
// arguments, which the command processor preloads into SGPRs before the first wave starts (-mllvm -amdgpu-kernarg-preload-count, gfx940+: the Makefile sets it for
// this file); the struct behind them is fetched by scalar loads as before, under the prologue's first requests. FQ_RING_PRELOAD=0 at compile time: the struct only.
#ifndef FQ_RING_PRELOAD
#define FQ_RING_PRELOAD 1
#endif
template <int TYPE, int NSLOT, bool TWO, bool REF>
__global__ void __launch_bounds__(RNT) k_gemv_ln_ring(
#if FQ_RING_PRELOAD
                                                      const float * p_x, const fq_engine_sched * p_sched, const uint8_t * p_qkv, const uint8_t * p_up, int p_E, int p_FF, int p_nblkE, unsigned p_rsE,
#endif
                                                      fq_ring_ln_args a) {
#if FQ_RING_PRELOAD
    a.x = p_x; a.sched = p_sched; a.qkv = p_qkv; a.up = p_up; a.E = p_E; a.FF = p_FF; a.nblkE = p_nblkE; a.rsE = p_rsE;
#endif
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr int ACT = eng_act<TYPE>::value;
    constexpr int RING = NSLOT * ENG_SLOT;
    constexpr int TS = fq_desc(TYPE).tsize;
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int E = a.E, FF = a.FF;
    const fq_engine_sched sc = a.sched[blockIdx.x];
    uint8_t * ring  = smem;
    // default: [ring][mirror][f32 row][image(s)][control]; REF: [ring][mirror][image(s)][control][f32 row, later the term strip, to the end of the LDS]
    uint8_t * img_e = REF ? smem + RING + ENG_MIRROR : smem + RING + ENG_MIRROR + (((size_t) E * 4 + 15) & ~(size_t) 15);      // LN image feeding Wup (and Wqkv with one norm)
    uint8_t * img_e2 = img_e + fq_act_col_bytes(ACT, E);                                   // attention-norm image of a two-norm block
    uint8_t * ctlp  = img_e + (a.two_norms ? 2 : 1) * fq_act_col_bytes(ACT, E);
    float   * xrow  = REF ? (float *)(ctlp + RING_CTL_BYTES16) : (float *)(smem + RING + ENG_MIRROR);
    const unsigned ctl = (unsigned)(uintptr_t) ctlp;
    auto ldsf_st = [&](unsigned off, float v) { lds_st(ctl + off, __builtin_bit_cast(unsigned, v)); };
    auto ldsf_ld = [&](unsigned off) { return __builtin_bit_cast(float, lds_ld(ctl + off)); };

    const unsigned rsE = a.rsE;
    const int nA1 = sc.qg1 - sc.qg0, nA2 = 32 * (sc.ug1 - sc.ug0);
    auto pad1k = [](unsigned v) { return (v + 1023u) & ~1023u; };
    const unsigned pA1 = pad1k((unsigned) nA1 * rsE), pA2 = pad1k((unsigned) nA2 * rsE);

    // the hand-off tag of the launch that follows, and the rope table's row of this position (as k_gemv_ln)
    if (a.epoch_word && blockIdx.x == 0 && tid == 0) { const unsigned e = *a.epoch_word + 1u; *a.epoch_word = e ? e : 1u; }
    if (a.rope_cur && blockIdx.x == gridDim.x - 1 && tid >= 64 && tid < 128) a.rope_cur[tid - 64] = a.rope_cs[(int64_t)(*a.n_past_ptr) * 64 + (tid - 64)];
    if (tid < 2) lds_st64(ctl + eng_ctl::PTRS + 8 * tid, (unsigned long long)(uintptr_t)(tid == 0 ? a.qkv + (size_t) sc.qg0 * rsE : a.up + (size_t) sc.ug0 * 32 * rsE));
    if (tid < 32) lds_st(ctl + eng_ctl::CNT + 4 * tid, 0u);
    if (tid < 16) lds_st(ctl + eng_ctl::LOW + 4 * tid, tid < RNC ? 0u : 0xFFFFFFFFu);
    if (tid < 16) lds_st(ctl + eng_ctl::XG_DONE + 4 * tid, 0u);
    if (tid == 0) lds_st(ctl + eng_ctl::LANDED, 0u);
    if (a.dbg && tid == 0) a.dbg[(size_t) blockIdx.x * 8] = (long long) wall_clock64();
    __syncthreads();                                                       // the only workgroup barrier: before the roles split

#ifndef FQ_RING_PRIO
#define FQ_RING_PRIO 3          // bit 0: the loader wave, bit 1: the epilogue wave at raised issue priority (s_setprio 3)
#endif
    if (wid == 0) {
        if (FQ_RING_PRIO & 1) __builtin_amdgcn_s_setprio(3);
        // ================================================================================ loader (kernels_engine.hip's, two segments)
        const unsigned ring_lds = (unsigned)(uintptr_t) ring;
        eng_wait w{ a.err, false, nullptr, 0 };
        constexpr unsigned NP = (unsigned)(NSLOT * 16);
        constexpr unsigned MIRP = (unsigned)(ENG_MIRROR / 1024);
        unsigned pos = 0, rp = 0, freed = 0, reported = 0;
        auto report = [&](unsigned upto) { if ((int)(upto - reported) > 0) { reported = upto; if (lane == 0) lds_st(ctl + eng_ctl::LANDED, upto); } };
        auto report_keep = [&](unsigned keep_kb) { const unsigned back = (keep_kb + MIRP) * 1024u; if (pos > back) report(pos - back); };
        auto low_water = [&]() {
            unsigned v = lds_ld(ctl + eng_ctl::LOW + 4 * (lane < 16 ? lane : 0));
            v = (unsigned) wave_reduce((int) v, [](int x, int y) { return (unsigned) x < (unsigned) y ? x : y; });
            return (unsigned) __builtin_amdgcn_readfirstlane(v);
        };
        auto wait_space = [&](unsigned bytes) {
            if (pos + bytes - freed <= (unsigned) RING) return;
            freed = low_water();
            if (pos + bytes - freed <= (unsigned) RING) return;
            for (unsigned spins = 0;;) {
                const unsigned inflight = pos - reported;
#define RING_LAND_STEP(N) if (inflight > (N + MIRP) * 1024u) { asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); report_keep(N); } else
                RING_LAND_STEP(44) RING_LAND_STEP(40) RING_LAND_STEP(36) RING_LAND_STEP(32) RING_LAND_STEP(28) RING_LAND_STEP(24) RING_LAND_STEP(20) RING_LAND_STEP(16)
                RING_LAND_STEP(12) RING_LAND_STEP(8) RING_LAND_STEP(4)
                if (inflight > 0u) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); report(pos); }
                else __builtin_amdgcn_s_sleep(1);
#undef RING_LAND_STEP
                freed = low_water();
                if (pos + bytes - freed <= (unsigned) RING) break;
                if (!w.spin(spins, ENG_W_RING, pos, freed)) break;
            }
        };
        const unsigned voff0 = (unsigned) lane * 16u;
        eng_voff vo;
#pragma unroll
        for (int p = 0; p < 16; ++p) vo.v[p] = voff0 + 1024u * (unsigned) p;
        const unsigned mstart = __builtin_amdgcn_readfirstlane(ring_lds), mend = mstart + (unsigned) RING;
        unsigned ml = mstart;
        auto after_issue = [&]() {
            if (pos - reported >= 50u * 1024u) { asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); report_keep(32u); }
        };
        auto mirror = [&](const uint8_t * src, unsigned rp0, unsigned n) {
#pragma unroll
            for (unsigned t = 0; t < MIRP; ++t) {
                unsigned k = t + NP - rp0; k = k >= NP ? k - NP : k;
                if (k < n) glds_piece(src + k * 1024u, voff0, mstart + (NP + t) * 1024u);
            }
        };
        const unsigned flag_addr = ctl + eng_ctl::THIN;                     // (sampled by the batch statement; unused here)
        auto seg = [&](const uint8_t * src, unsigned padded) {
            const unsigned nfull = padded >> 14, ntail = (padded >> 10) & 15u;
            for (unsigned k = 0; k < nfull; ++k) {
                wait_space((unsigned) ENG_SLOT);
                if (w.dead) return;
                mirror(src, rp, 16u);
                (void) glds_batch16(src, vo, ml, mstart, mend, flag_addr);
                src += ENG_SLOT; pos += (unsigned) ENG_SLOT;
                rp += 16u; rp = rp >= NP ? rp - NP : rp;
                after_issue();
            }
            if (ntail) {
                wait_space(ntail * 1024u);
                if (w.dead) return;
                mirror(src, rp, ntail);
                for (unsigned p = 0; p < ntail; ++p) {
                    glds_piece(src + p * 1024u, voff0, ml);
                    ml += 1024u; ml = ml >= mend ? mstart : ml;
                }
                pos += ntail * 1024u;
                rp += ntail; rp = rp >= NP ? rp - NP : rp;
                after_issue();
            }
        };
        auto src = [&](int i) {
            const unsigned long long v = lds_ld64(ctl + eng_ctl::PTRS + 8 * (unsigned) i);
            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned) v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
            return (const uint8_t *)(uintptr_t)(((unsigned long long) hi << 32) | lo);
        };
        // the residual row's loads go first: every helper wave has issued its share
        if (!(a.debug_mode & 1)) w.until(ctl + RING_XISSUED, (unsigned) RNH, ENG_W_XG);
        RING_T(0, 1);
        // Wup first: its rows end in the epilogue wave's GELU + Q8 work, which then overlaps the Wqkv rows instead of trailing the launch
        if (nA2 > 0) seg(src(1), pA2);
        if (nA1 > 0 && !w.dead) seg(src(0), pA1);
        RING_T(0, 2);
        // the end of the stream is reported as it lands, not in one piece: the consumers' last runs start while the final pieces are in flight
#define RING_END_STEP(N) if (pos - reported > (N + MIRP) * 1024u) { asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); report_keep(N); }
        RING_END_STEP(40) RING_END_STEP(32) RING_END_STEP(24) RING_END_STEP(16) RING_END_STEP(10) RING_END_STEP(5)
#undef RING_END_STEP
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        report(pos);
        RING_T(0, 3);
        return;
    }

    // ==================================================================================== helpers: the epilogue wave (h = 0) and the consumers (c = h - 1)
    const int h = wid - 1, c = h - 1, ht = tid - 64;
    const bool isG = h == 0;
    constexpr int NLN = 3;                                                 // float4 of the row per helper thread: n_embd <= 8448
    eng_wait w{ a.err, false, nullptr, 0 };
    const int nblkE = a.nblkE;
    const int nv = E >> 2;
    const unsigned nx = (unsigned)((E + ENG_CHUNK - 1) / ENG_CHUNK);
    const bool nodots = (a.debug_mode & 2) != 0;
    const bool early = !(a.debug_mode & 64);                               // (tuning aid, FQ_RING_DEBUG bit 64: ring space handed back after the arithmetic, as in round 3)

    // ---- the residual row -> LDS (chunks of 1024 values, one per helper wave, each with its f64 partial sum) -> statistics -> Q8 image(s)
    // (the engine's one-pass LayerNorm, kernels_engine.hip: ggml.c:10577-10591 with the sums in f64)
    ln_row_regs<NLN> wr, br;
    ln_regs_issue_wb(a.ln_w, a.ln_b, E, RHT, wr, br, ht);
    {
        unsigned own = 0;
        float v[16]; int kown = -1;
        if (h < (int) nx) {                                                // (nx <= 11 for n_embd <= 11264: at most one chunk per helper)
            kown = h;
            const int base = kown * ENG_CHUNK;
#pragma unroll
            for (int j = 0; j < 16; ++j) { const int i = base + 64 * j + lane; v[j] = a.x[i < E ? i : E - 1]; }
        }
        if (lane == 0) lds_add(ctl + RING_XISSUED, 1u);                    // the loader may start: the row's requests are in the queue ahead of its own
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
        if (h <= 1) RING_T(1 + h, 1);
        const float mean = ldsf_ld(eng_ctl::STAT);
        float4 xv[NLN];
        double s2 = 0.0;
#pragma unroll
        for (int k = 0; k < NLN; ++k) {
            const int q4 = k * RHT + ht;
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
        if (old2 + 1u == (unsigned) RNH) {                              // the last wave: partial sums in wave order -> scale
            const unsigned long long pk = lds_ld64(ctl + eng_ctl::PSQ + 8u * (unsigned)(lane < RNH ? lane : 0));
            double t = 0.0;
#pragma unroll
            for (int k = 0; k < RNH; ++k) t += lane_get(__builtin_bit_cast(double, (long long) pk), k);
            const float variance = (float)(t / (double) E);
            const float scale = 1.0f / sqrtf(variance + 1e-5f);
            if (lane == 0) ldsf_st(eng_ctl::STAT + 4, scale);
            lds_drain();
            if (lane == 0) lds_st(ctl + eng_ctl::LN_STAT, 1u);
        }
        w.until(ctl + eng_ctl::LN_STAT, 1u, ENG_W_STAT);
        const float scale = ldsf_ld(eng_ctl::STAT + 4);
        auto norm_quant = [&](const ln_row_regs<NLN> & wq, const ln_row_regs<NLN> & bq, uint8_t * img) {
            const act_image_ptr o = act_image_at(img, ACT, E);
#pragma unroll
            for (int k = 0; k < NLN; ++k) {
                const int q4 = k * RHT + ht;
                if ((q4 & ~63) < nv) {                                     // wave-uniform
                    const bool lv = q4 < nv; const int j = lv ? q4 : nv - 1;
                    float4 t = xv[k];
                    const float4 ww = wq.t[k], bb = bq.t[k];
                    t.x *= scale; t.y *= scale; t.z *= scale; t.w *= scale;
                    t.x = t.x * ww.x + bb.x; t.y = t.y * ww.y + bb.y; t.z = t.z * ww.z + bb.z; t.w = t.w * ww.w + bb.w;
                    quant_q8_quad<ACT>(t, j, o, lv);
                }
            }
        };
        norm_quant(wr, br, img_e);
        if (a.two_norms) {
            ln_row_regs<NLN> w2r, b2r;
            ln_regs_issue_wb(a.ln2_w, a.ln2_b, E, RHT, w2r, b2r, ht);
            norm_quant(w2r, b2r, img_e2);
        }
        lds_drain();
        if (lane == 0) lds_add(ctl + eng_ctl::IMG_DONE, 1u);
        w.until(ctl + eng_ctl::IMG_DONE, (unsigned) RNH, ENG_W_IMG);
        if (h <= 1) RING_T(1 + h, 2);
    }

    const fq_actcol col_e  = { (const int8_t *) img_e,  (const float *)(img_e + fq_act_d_off(ACT, E)),  (const void *)(img_e + fq_act_aux_off(ACT, E)) };
    const fq_actcol col_e2 = { (const int8_t *) img_e2, (const float *)(img_e2 + fq_act_d_off(ACT, E)), (const void *)(img_e2 + fq_act_aux_off(ACT, E)) };
    const int gA = nA2 / 32;

    if ((FQ_RING_PRIO & 2) && isG) __builtin_amdgcn_s_setprio(3);
    if constexpr (REF) if (isG) {
        // ---- REF: the row sums in the reference's order, lane = row, then the default epilogues. Groups of 32 workgroup-local rows: the Wup groups first
        // (their rows stream first), then the Wqkv rows; a group's terms are complete when its counter has reached its row count.
        const float * strip = xrow;                                        // (every helper is past IMG_DONE: the f32 row is dead)
        const int T = nA2 + nA1, nG = (T + 31) >> 5;
        const unsigned SW = a.strip_stride; const int SM = a.strip_slots - 1;
        const act_image_ptr o = act_image_at(a.ff_image, ACT, FF);
        auto full = [&](int g_) { const int left = T - 32 * g_; return (unsigned)(left < 32 ? left : 32); };
        for (int gl = 0; gl < nG;) {
            for (unsigned spins = 0; (int)(lds_ld_u(ctl + eng_ctl::CNT + 4 * gl) - full(gl)) < 0;) { if (!w.spin(spins, ENG_W_GROUP, (unsigned) gl, 0)) break; __builtin_amdgcn_s_sleep(1); }
            const bool two = gl + 1 < nG && (int)(lds_ld_u(ctl + eng_ctl::CNT + 4 * (gl + 1)) - full(gl + 1)) >= 0;
            const int j = lane & 31, half = lane >> 5;
            const int myg = gl + (two ? half : 0);
            const bool st = two || half == 0;
            const int row = 32 * myg + j, rc = row < T ? row : T - 1;
            float v = (a.debug_mode & 128) ? 0.0f : fq_ref_chain(strip + (size_t)(rc & SM) * SW, nblkE, 0.0f);      // ggml.c:2594-2609: sumf = 0; sumf += term_i, i ascending
            if (myg >= gA) {
                if (st && row < T) a.qkv_dst[sc.qg0 + (row - nA2)] = v;
            }
            if (gl < gA) {                                                 // (wave-uniform: at least one of the two groups is a Wup group)
                const int g = sc.ug0 + (myg < gA ? myg : 0);               // block index in the FF-long image
                v = h2f_bits(a.gelu_tab[f2h_bits(v)]);                     // ggml.c:3477-3484
                const float amax = reduce32(fabsf(v), op_max());
                const float d  = amax / 127.0f;
                const float id = d ? 1.0f / d : 0.0f;
                const int q = round_half_away(v * id);
                const int s = reduce32(q, op_add());
                if (st && myg < gA) o.qs[32 * g + j] = (int8_t) q;
                if (st && myg < gA && j == 0) {
                    if (ACT == FQ_Q8_0) { o.d[g] = h2f_bits(f2h_bits(d)); ((int32_t *) o.aux)[g] = s; }
                    else                { o.d[g] = d; ((float *) o.aux)[g] = (float) s * d; }
                }
            }
            gl += two ? 2 : 1;
            if (lane == 0) lds_st(ctl + RING_SUMMED, (unsigned) gl);       // (behind the chain's reads in this wave's LDS queue)
        }
        RING_T(1, 3);
        return;
    }
    if (!REF && isG) {
        // ---- the Wup epilogues of this workgroup's groups as they complete (k_gemv_ln's epilogue: GELU table, Q8 block of 32) -> the image in memory
        const act_image_ptr o = act_image_at(a.ff_image, ACT, FF);
        for (int gl = 0; gl < gA;) {
            for (unsigned spins = 0; (int)(lds_ld_u(ctl + eng_ctl::CNT + 4 * gl) - 32u) < 0;) { if (!w.spin(spins, ENG_W_GROUP, (unsigned) gl, 0)) break; __builtin_amdgcn_s_sleep(1); }
            const bool two = gl + 1 < gA && (int)(lds_ld_u(ctl + eng_ctl::CNT + 4 * (gl + 1)) - 32u) >= 0;
            const int j = lane & 31, half = lane >> 5;
            const int myg = gl + (two ? half : 0);
            const bool st = two || half == 0;
            float v = ldsf_ld(eng_ctl::OUT + 4 * (32 * myg + j));
            const int g = sc.ug0 + myg;                                    // block index in the FF-long image
            v = h2f_bits(a.gelu_tab[f2h_bits(v)]);                         // ggml.c:3477-3484
            const float amax = reduce32(fabsf(v), op_max());
            const float d  = amax / 127.0f;
            const float id = d ? 1.0f / d : 0.0f;
            const int q = round_half_away(v * id);
            const int s = reduce32(q, op_add());
            if (st) o.qs[32 * g + j] = (int8_t) q;
            if (st && j == 0) {
                if (ACT == FQ_Q8_0) { o.d[g] = h2f_bits(f2h_bits(d)); ((int32_t *) o.aux)[g] = s; }
                else                { o.d[g] = d; ((float *) o.aux)[g] = (float) s * d; }
            }
            gl += two ? 2 : 1;
        }
        RING_T(1, 3);
        return;
    }

    // ---- consumers: rows out of the ring, runs of R consecutive rows round-robin (kernels_engine.hip `rows`)
    const unsigned strip_lds = (unsigned)(uintptr_t) xrow;
    auto rows = [&](auto rtag, auto ptag, unsigned seg_pos, unsigned padded, int nrows, const fq_actcol & col, const fq_act32 (&pre)[3], auto && sink, const int jbase) {
        constexpr int R = decltype(rtag)::value;
        constexpr bool PRE = decltype(ptag)::value;                        // (compile-time: a run-time choice between the arrays would put them into scratch memory)
        const unsigned row_bytes = (unsigned)(nblkE * TS);
        if (lane == 0) lds_st(ctl + eng_ctl::LOW + 4 * c, R * c < nrows ? seg_pos + (unsigned)(R * c) * rsE : seg_pos + padded);
        const int npass = (nblkE + 63) >> 6;
        for (int i = R * c; i < nrows; i += R * RNC) {
            const int last = i + R - 1 < nrows ? i + R - 1 : nrows - 1;
            const unsigned row0 = seg_pos + (unsigned) i * rsE, rowl = seg_pos + (unsigned) last * rsE;
            unsigned pr[R]; float v[R], acc[R];
            const unsigned p0 = row0 % (unsigned) RING;
#pragma unroll
            for (int r = 0; r < R; ++r) { const unsigned q = p0 + (unsigned)(i + r <= last ? r : last - i) * rsE; pr[r] = q >= (unsigned) RING ? q - (unsigned) RING : q; acc[r] = 0.0f; }
            const int nx_ = i + R * RNC;
            const unsigned next_low = nx_ < nrows ? seg_pos + (unsigned) nx_ * rsE : seg_pos + padded;
            unsigned sa[R];                                            // REF: LDS addresses of the rows' term strips (slot = workgroup-local row mod slots)
            if constexpr (REF) {
                const int need = ((jbase + last) >> 5) - (a.strip_slots >> 5) + 1;      // groups that must have been summed before these slots are written again
                if (need > 0) w.until(ctl + RING_SUMMED, (unsigned) need, ENG_W_GROUP);
#pragma unroll
                for (int r = 0; r < R; ++r) sa[r] = strip_lds + (unsigned)((jbase + (i + r <= last ? i + r : last)) & (a.strip_slots - 1)) * a.strip_stride * 4u;
            }
            for (int ps = 0; ps < npass; ps += 3) {
                const int np = npass - ps < 3 ? npass - ps : 3;
                const unsigned upto = (unsigned)((ps + np) * 64 * TS);
                const unsigned need = rowl + (upto < row_bytes ? upto : row_bytes);
                for (unsigned spins = 0; (int)(lds_ld_u(ctl + eng_ctl::LANDED) - need) < 0;) { if (!w.spin(spins, ENG_W_LAND, need, 0)) break; __builtin_amdgcn_s_sleep(1); }
                const unsigned low_after = ps + 3 < npass ? row0 + upto : next_low;
                if (PRE && early && !nodots) {
                    // rows of <= 3 passes with the lane's activation slices resident: the run's weight units into registers, the ring space handed back
                    // BEFORE the arithmetic (the LDS serves a wave's requests in order: the reads are ahead of the LOW store) -- the loader's look-ahead is
                    // the part of the ring beyond the oldest bytes still needed, and rows held through dots + butterflies kept the stream 1 us longer
                    if (np == 3)      { eng_regs<R, 3> G; eng_pass_load<TYPE, RING, R, 3>(ring, pr, nblkE, 0, lane, G); if (lane == 0) lds_st(ctl + eng_ctl::LOW + 4 * c, low_after); eng_pass_dot_pre<TYPE, R, 3, REF>(G, nblkE, 0, pre, lane, acc, sa); }
                    else if (np == 2) { eng_regs<R, 2> G; eng_pass_load<TYPE, RING, R, 2>(ring, pr, nblkE, 0, lane, G); if (lane == 0) lds_st(ctl + eng_ctl::LOW + 4 * c, low_after); eng_pass_dot_pre<TYPE, R, 2, REF>(G, nblkE, 0, pre, lane, acc, sa); }
                    else              { eng_regs<R, 1> G; eng_pass_load<TYPE, RING, R, 1>(ring, pr, nblkE, 0, lane, G); if (lane == 0) lds_st(ctl + eng_ctl::LOW + 4 * c, low_after); eng_pass_dot_pre<TYPE, R, 1, REF>(G, nblkE, 0, pre, lane, acc, sa); }
                    continue;
                }
                if (!nodots) {
                    if (PRE) {                                  // rows of <= 3 passes: the lane's activation slices were loaded once, before the row loop
                        if (np == 3)      eng_pass_group_pre<TYPE, RING, R, 3, REF>(ring, pr, nblkE, 0, pre, lane, acc, sa);
                        else if (np == 2) eng_pass_group_pre<TYPE, RING, R, 2, REF>(ring, pr, nblkE, 0, pre, lane, acc, sa);
                        else              eng_pass_group_pre<TYPE, RING, R, 1, REF>(ring, pr, nblkE, 0, pre, lane, acc, sa);
                    }
                    else if (np == 3) eng_pass_group<TYPE, RING, R, 3, REF>(ring, pr, nblkE, 64 * ps, col, lane, acc, sa);
                    else if (np == 2) eng_pass_group<TYPE, RING, R, 2, REF>(ring, pr, nblkE, 64 * ps, col, lane, acc, sa);
                    else              eng_pass_group<TYPE, RING, R, 1, REF>(ring, pr, nblkE, 64 * ps, col, lane, acc, sa);
                }
                if (lane == 0) lds_st(ctl + eng_ctl::LOW + 4 * c, low_after);
            }
            if constexpr (REF) {
                // the rows' terms are in the strip: report them (the counter add follows the term stores in this wave's LDS queue)
#pragma unroll
                for (int r = 0; r < R; ++r) if (i + r <= last && lane == 0) lds_add(ctl + eng_ctl::CNT + 4 * ((jbase + i + r) >> 5), 1u);
                continue;
            }
#pragma unroll
            for (int r = 0; r < R; ++r) v[r] = wave_sum(acc[r]);
#pragma unroll
            for (int r = 0; r < R; ++r) if (i + r <= last) sink(i + r, v[r]);
        }
    };
    std::integral_constant<int, 1> R1; std::integral_constant<int, 2> R2;
    std::integral_constant<bool, true> PT; std::integral_constant<bool, false> PF;
    auto rows_e = [&](bool prefit_, unsigned seg_pos, unsigned padded, int nrows, const fq_actcol & col, const fq_act32 (&pre)[3], auto && sink, const int jbase) {
        const bool r2 = !(a.debug_mode & 16) && (unsigned)(2 * RNC) * rsE * 2u <= (unsigned) RING;
        if (prefit_) { if (r2) rows(R2, PT, seg_pos, padded, nrows, col, pre, sink, jbase); else rows(R1, PT, seg_pos, padded, nrows, col, pre, sink, jbase); }
        else         { if (r2) rows(R2, PF, seg_pos, padded, nrows, col, pre, sink, jbase); else rows(R1, PF, seg_pos, padded, nrows, col, pre, sink, jbase); }
    };
    // a lane's units are lane, lane + 64, lane + 128 of EVERY row: their activation slices stay in registers when a row is <= 3 passes long
    const bool prefit = ((nblkE + 63) >> 6) <= 3 && !(a.debug_mode & 32);
    fq_act32 preA[3], preB[TWO ? 3 : 1];                                 // (one norm: Wqkv reads the same image)
    if (prefit) {
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const int u = 64 * p + lane, uc = u < nblkE ? u : nblkE - 1;
            preA[p] = fq_act32_load(col_e, uc);
            if constexpr (TWO) preB[p] = fq_act32_load(col_e2, uc);
        }
    }
    rows_e(prefit, 0u, pA2, nA2, col_e, preA, [&](int i, float v) { if (lane == 0) { ldsf_st(eng_ctl::OUT + 4 * i, v); lds_add(ctl + eng_ctl::CNT + 4 * (i >> 5), 1u); } }, 0);
    if (c == 0 || c == 9) RING_T(c == 0 ? 2 : 3, 3);
    if constexpr (TWO) rows_e(prefit, pA2, pA1, nA1, col_e2, preB, [&](int i, float v) { if (lane == 0) a.qkv_dst[sc.qg0 + i] = v; }, nA2);
    else rows_e(prefit, pA2, pA1, nA1, col_e, preA, [&](int i, float v) { if (lane == 0) a.qkv_dst[sc.qg0 + i] = v; }, nA2);
    if (c == 0 || c == 9) RING_T(c == 0 ? 2 : 3, 4);
}

// ---- host side
namespace {
struct ring_plan { int type, E, FF, qkv_rows, n_wg; fq_engine_sched * dev; };
std::vector<ring_plan> g_plans;

// rows [qg0, qg1) of Wqkv and 32-row groups [ug0, ug1) of Wup per workgroup: the groups are dealt evenly, the Wqkv rows fill the
// workgroups with fewer groups up to the common row count
const fq_engine_sched * ring_schedule(int type, int E, int FF, int qkv_rows, int n_wg, bool create) {
    for (const ring_plan & p : g_plans) if (p.type == type && p.E == E && p.FF == FF && p.qkv_rows == qkv_rows && p.n_wg == n_wg) return p.dev;
    if (!create) return nullptr;                                            // (a launch may sit inside a stream capture: it never allocates; fq_ring_prepare does)
    const int groups = FF / 32;
    std::vector<fq_engine_sched> s((size_t) n_wg);
    std::vector<int> ng((size_t) n_wg);
    for (int i = 0; i < n_wg; ++i) ng[(size_t) i] = groups / n_wg + (i < groups % n_wg ? 1 : 0);
    const int total = qkv_rows + FF;
    int T = (total + n_wg - 1) / n_wg;
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
    g_plans.push_back(ring_plan{ type, E, FF, qkv_rows, n_wg, dev });
    return dev;
}
}   // namespace

void fq_ring_free_plans() {
    for (ring_plan & p : g_plans) (void) hipFree(p.dev);
    g_plans.clear();
}

// the schedule of a shape must exist before a stream capture (it allocates): called at context set-up
bool fq_ring_prepare(int type, int64_t E, int64_t FF, int64_t qkv_rows, int n_cu) {
    if (!(type == FQ_Q4_0 || type == FQ_Q4_1 || type == FQ_Q5_0 || type == FQ_Q5_1 || type == FQ_Q8_0)) return false;
    if (E % 32 || E > 8448 || FF % 32 || FF > (int64_t) 32 * 12 * n_cu) return false;
    ring_schedule(type, (int) E, (int) FF, (int) qkv_rows, n_cu, true);
    return true;
}

// k_gemv_ln's launch through the ring form; false = outside its scope (nothing launched)
bool fq_launch_gemv_ln_ring(const fq_gemv_ln_args & g, unsigned * err, int n_cu, hipStream_t st, bool ref) {
    FQ_TL(st, ref ? "gemv_ln_ring_ref" : "gemv_ln_ring");
    if (g.nseg != 2 || g.seg[1].epi != FQ_LNEPI_GELU_QUANT || g.seg[0].epi != FQ_LNEPI_STORE || g.argmax_val) return false;
    const fq_weight & wq = g.seg[0].w, & wu = g.seg[1].w;
    const int type = wq.type;
    if (type != wu.type || wq.K != g.E || wu.K != g.E || wq.row_stride != wu.row_stride) return false;
    if (!(type == FQ_Q4_0 || type == FQ_Q4_1 || type == FQ_Q5_0 || type == FQ_Q5_1 || type == FQ_Q8_0)) return false;
    if (g.E % 32 || g.E > 8448 || wu.M % 32) return false;
    const int act = fq_desc(type).act_type;
    if (g.seg[1].next_act_type != act) return false;                      // (Wdown has the format of the block: its image type is this kernel's ACT)
    const bool two_norms = g.seg[0].ln_w != g.seg[1].ln_w;
    fq_ring_ln_args a{};
    a.x = g.x; a.E = (int) g.E; a.FF = (int) wu.M; a.nblkE = (int) wq.nblk; a.rsE = (unsigned) wq.row_stride;
    a.qkv = wq.plane[0]; a.up = wu.plane[0]; a.qkv_rows = (int) wq.M;
    a.ln_w = g.seg[1].ln_w; a.ln_b = g.seg[1].ln_b; a.ln2_w = two_norms ? g.seg[0].ln_w : nullptr; a.ln2_b = two_norms ? g.seg[0].ln_b : nullptr; a.two_norms = two_norms ? 1 : 0;
    a.qkv_dst = g.seg[0].dst; a.ff_image = g.seg[1].dst_image; a.gelu_tab = g.gelu_table;
    if (a.FF > 32 * 12 * n_cu) return false;
    a.sched = ring_schedule(type, a.E, a.FF, a.qkv_rows, n_cu, false);
    if (!a.sched) return false;                                            // shape not prepared (fq_ring_prepare at context set-up): the caller launches k_gemv_ln
    a.epoch_word = g.epoch_word; a.n_past_ptr = g.n_past_ptr; a.rope_cs = g.rope_cs; a.rope_cur = g.rope_cur;
    a.err = err; a.dbg = g.dbg;
    static const int dbg = getenv("FQ_RING_DEBUG") ? atoi(getenv("FQ_RING_DEBUG")) : 0;
    a.debug_mode = dbg;
    const size_t xrow_bytes = ((size_t) a.E * 4 + 15) & ~(size_t) 15;
    const size_t fixed = (size_t) ENG_MIRROR + xrow_bytes + (two_norms ? 2 : 1) * fq_act_col_bytes(act, a.E) + (ref ? RING_CTL_BYTES16 : RING_CTL_BYTES);
    int nslot = 0;
    size_t lds = 0;
    a.strip_stride = fq_ref_strip_stride(a.nblkE); a.strip_slots = 0;
    for (int n : { 7, 6, 4 }) {
        if ((size_t) n * ENG_SLOT + fixed > 160 * 1024) continue;
        if (ref) {
            // the term strip: the f32 row's bytes (the last region of the REF layout) and whatever the workgroup leaves unused behind them; 64 row slots, or 32
            const size_t room = 160 * 1024 - ((size_t) n * ENG_SLOT + fixed - xrow_bytes), row_b = (size_t) a.strip_stride * 4;
            const int slots = room >= 64 * row_b ? 64 : (room >= 32 * row_b ? 32 : 0);
            if (!slots) continue;
            a.strip_slots = slots;
            const size_t strip_b = (size_t) slots * row_b;
            lds = (size_t) n * ENG_SLOT + fixed - xrow_bytes + (strip_b > xrow_bytes ? strip_b : xrow_bytes);
        } else lds = (size_t) n * ENG_SLOT + fixed;
        nslot = n; break;
    }
    if (!nslot) return false;
    // a row (<= 3 passes of it) must fit the ring next to the loader's restart slot
    if ((size_t) 2 * 2 * a.rsE + ENG_SLOT > (size_t) nslot * ENG_SLOT) return false;
#if FQ_RING_PRELOAD
#define FQ_RING_LEAD a.x, a.sched, a.qkv, a.up, a.E, a.FF, a.nblkE, a.rsE,
#else
#define FQ_RING_LEAD
#endif
#define FQ_RING_LAUNCH3(T, NS, TW, RF) { \
        static bool set = false; \
        if (!set) { HIP_CHECK(hipFuncSetAttribute((const void *) k_gemv_ln_ring<T, NS, TW, RF>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); set = true; } \
        hipEvent_t e0_ = nullptr, e1_ = nullptr; fq_prof_events(&e0_, &e1_);      /* (the open profile bracket's events, if any, go to the dispatch) */ \
        if (e0_) hipExtLaunchKernelGGL((k_gemv_ln_ring<T, NS, TW, RF>), dim3((unsigned) n_cu), dim3(RNT), lds, st, e0_, e1_, 0, FQ_RING_LEAD a); \
        else     hipLaunchKernelGGL((k_gemv_ln_ring<T, NS, TW, RF>), dim3((unsigned) n_cu), dim3(RNT), lds, st, FQ_RING_LEAD a); }
#define FQ_RING_LAUNCH2(T, NS, TW) { if (ref) FQ_RING_LAUNCH3(T, NS, TW, true) else FQ_RING_LAUNCH3(T, NS, TW, false) }
#define FQ_RING_LAUNCH(T, NS) { if (two_norms) FQ_RING_LAUNCH2(T, NS, true) else FQ_RING_LAUNCH2(T, NS, false) }
#define FQ_RING_CASE(T) case T: if (nslot == 7) FQ_RING_LAUNCH(T, 7) else if (nslot == 6) FQ_RING_LAUNCH(T, 6) else FQ_RING_LAUNCH(T, 4) break;
    switch (type) {
        FQ_RING_CASE(FQ_Q4_0) FQ_RING_CASE(FQ_Q4_1) FQ_RING_CASE(FQ_Q5_0) FQ_RING_CASE(FQ_Q5_1) FQ_RING_CASE(FQ_Q8_0)
        default: return false;
    }
#undef FQ_RING_CASE
#undef FQ_RING_LAUNCH
#undef FQ_RING_LAUNCH2
#undef FQ_RING_LAUNCH3
    return true;
}
