// kernels_engine.hip -- the PERSISTENT decode engine, second form: one launch per generated token (gfx950, wave64).
//
// Replaces the per-token launch list of the fused decode path (2 launches per block + lm_head: k_gemv_ln / k_attn_out,
// kernels_decode.hip) -- i.e. one N = 1 pass of falcon_eval_internal's graph (libfalcon.cpp:2115-2466) whose mat-muls are
// ggml_compute_forward_mul_mat_q_f32 (ggml.c:11318-11529) -- by ONE kernel of one workgroup per CU (12 waves) that stays
// resident for all blocks of the stage. Built to the chip's measured recipe (MI355X_MICROARCH.md, rows engine-vs-launches,
// gather-pass, polling-cost, allgather, prefetch-credit):
//
//   streaming workgroups   wave 0 = LOADER: streams the workgroup's byte ranges of [Wqkv | Wup | Wdown | Wo] of every block (then
//                          lm_head), in the order they will be used, into a RING of LDS with global_load_lds_dwordx4 (1 KiB per
//                          wave-instruction, non-temporal, no registers held), limited only by ring space: up to ~112 KiB (~4.5 us of
//                          stream) ahead of the arithmetic, across every dependency stall. The ring's first 2 KiB are mirrored behind
//                          its end, so a reader never wraps inside a 1-1.5 KiB column of a row. While the workgroup gathers a
//                          hand-off the loader thins itself to 16 pieces in flight.
//                          wave 1 = GATHERER: the Wup epilogues (GELU, Q8 block, publish) of the workgroup's 32-row groups as they
//                          complete, and its share of the gathering of the cross-workgroup hand-offs into LDS.
//                          waves 2..11 = CONSUMERS: take the landed rows round-robin and run the row dots out of LDS (the fq_units.h
//                          arithmetic, lane l = units l, l+64, ..: bit-identical to the other paths); when they run out of rows they
//                          gather their share of the next hand-off.
//                          No barrier anywhere after the roles split (the loader could never reach one): every dependency inside the
//                          workgroup is a MONOTONIC counter or flag in LDS (value = a function of the block index), polled with s_sleep.
//   attention workgroups   2 query heads each (attn_decode_group, the code of k_attn_decode / k_attn_out): the block's first key /
//                          value rows are requested BEFORE q exists, then RoPE, KV append, K.Q, soft_max, V.P, Q8 image.
//   between workgroups     8-byte {tag, value} granules written with single agent-scope stores, gathered in chunks of 1024 (16 loads
//                          in flight per lane = one memory round trip per attempt) by ONE wave per chunk and workgroup:
//                          residual row x (4 hops per block: x -> LN -> [qkv, up] -> attention / GELU image -> down, wo -> x).
//   LayerNorm              the gathering waves leave the f32 row in LDS with one f64 partial sum per chunk; the LAST of them to
//                          arrive adds the partial sums (chunk order), takes the second pass over the row alone (f64, as ggml.c:
//                          10577-10591) and publishes {mean, scale}: no barrier. All 11 waves then normalise and quantize their
//                          share (quant_q8_quad: the k_gemv_ln arithmetic).
//
// Scope: legacy formats (Q4_0 / Q4_1 / Q5_0 / Q5_1 / Q8_0 -> Q8_0 / Q8_1 activations), one format per stage; everything
// else keeps the two-launch path (fq_launch_decode_engine returns false).
#include "fq_block_dev.h"
#include "fq_attn_dev.h"
#include "fq_attn_decode_dev.h"
#include "kernels.h"
#include "hip_context.h"
#include <hip/hip_ext.h>
#include <type_traits>
#include <vector>

#include "fq_engine_dev.h"

#define ENG_NSTAMP_BLK 4
#define ENG_STAMP_AT(sbase, slot) do { if (a.dbg && lane == 0 && b >= 0 && (b < 3 || b == a.n_layers - 1)) \
        a.dbg[(sbase) + ((size_t) blockIdx.x * ENG_NSTAMP_BLK + (b < 3 ? b : 3)) * 8 + (slot)] = (long long) wall_clock64(); } while (0)

// =============================================================================================== the kernel
template <int TYPE, int NSLOT>
__global__ void __launch_bounds__(ENG_NT) k_decode_engine(fq_engine_args a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr int ACT = eng_act<TYPE>::value;
    constexpr int RING = NSLOT * ENG_SLOT;
    constexpr int TS = fq_desc(TYPE).tsize;
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);     // (wave-uniform for the compiler too: scalar address arithmetic)
    const unsigned epoch0 = *a.epoch_word;
    const int E = a.E, FF = a.FF;
    const bool nodots = (a.debug_mode & 2) != 0;

    if ((int) blockIdx.x < a.n_attn) {
        if (a.debug_mode == 1) return;
        // ================================================================================ attention workgroup
        const int grp = tid >> 8, gtid = tid & 255;
        const bool idle = grp >= a.hpw;
        int h = (int) blockIdx.x * a.hpw + grp;
        const bool live = !idle && h < a.H;
        if (h >= a.H) h = a.H - 1;
        const int hk = h / (a.H / a.HKV);
        uint8_t * gbase = smem + (size_t)(idle ? 0 : grp) * a.attn_lds_group;
        float * stage = (float *) gbase;                                   // q | k | v of the head, 64 floats each
        eng_wait w{ a.err, false, a.dbg, 0 };
        for (int b = 0; b < a.n_layers; ++b) {
            w.blk = b;
            const unsigned tag = epoch0 + (unsigned) b + 1u;
            const fq_engine_layer & L = a.layers[b];
#define ENG_ASTAMP(slot) do { if (a.dbg && tid == 0 && (b < 3 || b == a.n_layers - 1)) a.dbg[FQ_ENG_DBG_STAMPS + ((size_t) blockIdx.x * ENG_NSTAMP_BLK + (b < 3 ? b : 3)) * 8 + (slot)] = (long long) wall_clock64(); } while (0)
            ENG_ASTAMP(0);
            // the block's first key / value rows are on their way while q / k / v are still being computed elsewhere
            attn_pre P;
            if (!idle) attn_prefetch(L.kc, L.vc, a.HKV, hk, a.max_n_kv, gtid, P);
            if (!idle && gtid < 192) {
                const int part = gtid >> 6, d = gtid & 63;
                const int row = (part == 0 ? h : (part == 1 ? a.H + hk : a.H + a.HKV + hk)) * 64 + d;
                unsigned long long x = 0;
                for (unsigned spins = 0;;) {
                    x = gran_ld(a.qkvg + row);
                    if ((a.debug_mode & 8) || __all((unsigned)(x >> 32) == tag)) break;
                    if (!w.spin(spins, ENG_W_QKV, (unsigned) row, (unsigned)(x >> 32))) break;
                    __builtin_amdgcn_s_sleep(2);
                }
                stage[gtid] = __builtin_bit_cast(float, (unsigned) x);
            }
            __syncthreads();
            ENG_ASTAMP(1);
            if (idle) { attn_decode_group_idle(); continue; }
            // (qkv is not used when q / k / v sources are given; it must be a GLOBAL pointer all the same -- with an LDS-derived or
            //  null one hipcc 7.2's InstCombine crashes on the selects between the two sources)
            fq_attn_decode_args at{ a.rope_cs, a.H, a.HKV, a.n_past, a.rope_cs, L.kc, L.vc, a.exp_tab, nullptr, (uint8_t *) a.attg, ACT, a.max_n_kv, nullptr,
                                    stage, stage + 64, stage + 128 };
            // (a per-iteration opaque copy of the thread / head index: without it hipcc hoists ~50 registers of per-thread address
            //  arithmetic out of the block loop and the attention spills)
            int gt2 = gtid, h2 = h;
            asm volatile("" : "+v"(gt2), "+v"(h2));
            h2 = __builtin_amdgcn_readfirstlane(h2);
            attn_decode_group_p<true, true>(at, h2, live, gt2, gbase + 768, nullptr, fq_publish{ a.attg, tag }, P);
            ENG_ASTAMP(2);
        }
        return;
    }

    // ==================================================================================== streaming workgroup
    const int sw = (int) blockIdx.x - a.n_attn;
    const fq_engine_sched sc = a.sched[sw];
    uint8_t * ring    = smem;
    uint8_t * reg_r   = smem + RING + ENG_MIRROR;                          // f32 residual row, then the GELU image
    uint8_t * reg_s   = reg_r + eng_region_r(ACT, E, FF);                  // LayerNorm image(s), then the attention output image
    uint8_t * ctlp    = reg_s + eng_region_s(ACT, E, a.two_norms);
    float   * xrow    = (float *) reg_r;
    uint8_t * img_ff  = reg_r;
    uint8_t * img_e   = reg_s;                                             // LN image feeding Wup (and Wqkv with one norm)
    uint8_t * img_e2  = reg_s + fq_act_col_bytes(ACT, E);                  // attention-norm image of a two-norm block
    uint8_t * img_att = reg_s;
    const unsigned ctl = (unsigned)(uintptr_t) ctlp;
    // (dots, residual values and counters of the control block are touched with explicit DS instructions: in order with the
    //  counter updates that publish them)
    auto ldsf_st = [&](unsigned off, float v) { lds_st(ctl + off, __builtin_bit_cast(unsigned, v)); };
    auto ldsf_ld = [&](unsigned off) { return __builtin_bit_cast(float, lds_ld(ctl + off)); };

    // per-block stream layout of this workgroup (every segment padded to whole 1 KiB pieces)
    const unsigned rsE = a.rsE, rsF = a.rsF;
    const int nA1 = sc.qg1 - sc.qg0, nA2 = 32 * (sc.ug1 - sc.ug0), nB = sc.r1 - sc.r0;     // (qg0, qg1: ROWS of Wqkv)
    auto pad1k = [](unsigned v) { return (v + 1023u) & ~1023u; };
    const unsigned pA1 = pad1k((unsigned) nA1 * rsE), pA2 = pad1k((unsigned) nA2 * rsE), pB1 = pad1k((unsigned) nB * rsF), pB2 = pad1k((unsigned) nB * rsE);
    const unsigned p_blk = pA1 + pA2 + pB1 + pB2;
    int nH = 32 * (sc.hg1 - sc.hg0);
    if (a.lm_head && sc.hg0 * 32 + nH > a.V) nH = a.V - sc.hg0 * 32;
    if (!a.lm_head || nH < 0) nH = 0;
    const unsigned pH = pad1k((unsigned) nH * rsE);

    // the loader must not touch global memory except through its DMA (a vector load makes hipcc drain vmcnt): every block's
    // four source pointers are put into LDS up front
    for (int i = tid; i < 4 * a.n_layers + 1; i += ENG_NT) {
        const int b = i >> 2, m = i & 3;
        const uint8_t * p;
        if (i == 4 * a.n_layers) p = a.lm_head ? a.lm_head + (size_t) sc.hg0 * 32 * a.rsE : nullptr;
        else {
            const fq_engine_layer & L = a.layers[b];
            p = m == 0 ? L.qkv + (size_t) sc.qg0 * a.rsE : (m == 1 ? L.up + (size_t) sc.ug0 * 32 * a.rsE : (m == 2 ? L.down + (size_t) sc.r0 * a.rsF : L.wo + (size_t) sc.r0 * a.rsE));
        }
        lds_st64(ctl + eng_ctl::PTRS + 8 * i, (unsigned long long)(uintptr_t) p);
    }
    if (tid < 32) lds_st(ctl + eng_ctl::CNT + 4 * tid, 0u);                 // CNT and CNTH
    if (tid < 16) lds_st(ctl + eng_ctl::LOW + 4 * tid, tid < ENG_NC ? 0u : 0xFFFFFFFFu);
    if (tid < 16) lds_st(ctl + eng_ctl::XG_DONE + 4 * tid, 0u);             // all flags and counters (XG_DONE .. S2_DONE)
    if (tid == 0) lds_st(ctl + eng_ctl::LANDED, 0u);
    __syncthreads();                                                       // the only workgroup barrier: before the roles split

    if (wid == 0) {
        // ================================================================================ loader
        const unsigned ring_lds = (unsigned)(uintptr_t) ring;
        eng_wait w{ a.err, false, a.dbg, 0 };
        // pos = bytes issued, reported = bytes known to have landed (published in LANDED), freed = cached low-water mark; rp = ring
        // piece index. ALL of it is wave-uniform and kept scalar on purpose: one wave executes this loop, a dependent instruction
        // costs it ~8 cycles; pieces are issued 16 at a time, unrolled, with scalar addressing. At most 48 pieces are in flight;
        // s_waitcnt vmcnt(N) = "all but my N newest DMA operations have landed" -- up to two of those may be mirror copies, so a
        // report claims two pieces fewer than the count suggests.
        constexpr unsigned NP = (unsigned)(NSLOT * 16);
        constexpr unsigned MIRP = (unsigned)(ENG_MIRROR / 1024);
        unsigned pos = 0, rp = 0, freed = 0, reported = 0;
        unsigned long long t_blocked = 0, t_report = 0, n_blocked = 0, t_start = __builtin_amdgcn_s_memtime();
        auto report = [&](unsigned upto) { if ((int)(upto - reported) > 0) { reported = upto; if (lane == 0) lds_st(ctl + eng_ctl::LANDED, upto); } };
        auto report_keep = [&](unsigned keep_kb) { const unsigned back = (keep_kb + MIRP) * 1024u; if (pos > back) report(pos - back); };
        // make room for `bytes` (<= one slot) more: while the ring is full let the pieces in flight land step by step and report them
        auto low_water = [&]() {
            unsigned v = lds_ld(ctl + eng_ctl::LOW + 4 * (lane < 16 ? lane : 0));
            v = (unsigned) wave_reduce((int) v, [](int x, int y) { return (unsigned) x < (unsigned) y ? x : y; });
            return (unsigned) __builtin_amdgcn_readfirstlane(v);
        };
        // make room for `bytes` (<= one slot) more. While the ring is full the pieces in flight are let land and reported in SMALL steps
        // (4 pieces), with a look at the consumers' low-water mark after every step: a consumer may be waiting for exactly those pieces,
        // and space freed meanwhile must be refilled at once (waiting for everything to land first -- 1.5 us -- held the refill rate of a
        // full ring at 10 GB/s per CU).
        auto wait_space = [&](unsigned bytes) {
            if (pos + bytes - freed <= (unsigned) RING) return;
            freed = low_water();
            if (pos + bytes - freed <= (unsigned) RING) return;
            const unsigned long long tb0 = __builtin_amdgcn_s_memtime(); ++n_blocked;
            for (unsigned spins = 0;;) {
                const unsigned inflight = pos - reported;
#define ENG_LAND_STEP(N) if (inflight > (N + MIRP) * 1024u) { asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); report_keep(N); } else
                ENG_LAND_STEP(44) ENG_LAND_STEP(40) ENG_LAND_STEP(36) ENG_LAND_STEP(32) ENG_LAND_STEP(28) ENG_LAND_STEP(24) ENG_LAND_STEP(20) ENG_LAND_STEP(16)
                ENG_LAND_STEP(12) ENG_LAND_STEP(8) ENG_LAND_STEP(4)
                if (inflight > 0u) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); report(pos); }
                else __builtin_amdgcn_s_sleep(1);
#undef ENG_LAND_STEP
                freed = low_water();
                if (pos + bytes - freed <= (unsigned) RING) break;
                if (!w.spin(spins, ENG_W_RING, pos, freed)) break;
            }
            t_blocked += __builtin_amdgcn_s_memtime() - tb0;
        };
        const unsigned voff0 = (unsigned) lane * 16u;
        eng_voff vo;
#pragma unroll
        for (int p = 0; p < 16; ++p) vo.v[p] = voff0 + 1024u * (unsigned) p;
        const unsigned mstart = __builtin_amdgcn_readfirstlane(ring_lds), mend = mstart + (unsigned) RING;
        unsigned ml = mstart;                                              // LDS address of ring piece rp
        auto after_issue = [&](unsigned thin) {
            // gather-pass: while this CU gathers a hand-off its own DMA burst is what the gather's loads queue behind
            if (thin != 0u) {
                if (pos - reported > 18u * 1024u) { asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); report_keep(16u); }
            } else if (pos - reported >= 50u * 1024u) {
                const unsigned long long tr0 = __builtin_amdgcn_s_memtime();
                asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); report_keep(32u);
                t_report += __builtin_amdgcn_s_memtime() - tr0;
            }
        };
        // the ring's first MIRP pieces a second time, behind its end: k = index inside the run of n pieces just issued from src
        auto mirror = [&](const uint8_t * src, unsigned rp0, unsigned n) {
#pragma unroll
            for (unsigned t = 0; t < MIRP; ++t) {
                unsigned k = t + NP - rp0; k = k >= NP ? k - NP : k;
                if (k < n) glds_piece(src + k * 1024u, voff0, mstart + (NP + t) * 1024u);
            }
        };
        const unsigned thin_addr = ctl + eng_ctl::THIN;
        auto seg = [&](const uint8_t * src, unsigned padded) {
            const unsigned nfull = padded >> 14, ntail = (padded >> 10) & 15u;
            for (unsigned k = 0; k < nfull; ++k) {
                wait_space((unsigned) ENG_SLOT);
                if (w.dead) return;
                mirror(src, rp, 16u);                                      // (first: a mirror copy is then never newer than its piece, see report_keep)
                const unsigned thin = __builtin_amdgcn_readfirstlane(glds_batch16(src, vo, ml, mstart, mend, thin_addr));
                src += ENG_SLOT; pos += (unsigned) ENG_SLOT;
                rp += 16u; rp = rp >= NP ? rp - NP : rp;
                after_issue(a.thin_loader ? thin : 0u);
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
                after_issue(0u);
            }
        };
        auto src = [&](int i) {                                             // (wave-uniform by construction; made so for the compiler)
            const unsigned long long v = lds_ld64(ctl + eng_ctl::PTRS + 8 * (unsigned) i);
            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned) v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
            return (const uint8_t *)(uintptr_t)(((unsigned long long) hi << 32) | lo);
        };
        for (int b = 0; b < a.n_layers && !w.dead; ++b) {
            seg(src(4 * b), pA1);
            seg(src(4 * b + 1), pA2);
            seg(src(4 * b + 2), pB1);
            seg(src(4 * b + 3), pB2);
        }
        if (nH > 0 && !w.dead) seg(src(4 * a.n_layers), pH);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        report(pos);
        if (a.dbg && lane == 0) {
            long long * r = a.dbg + FQ_ENG_DBG_COUNTERS + (size_t) blockIdx.x * 16;
            r[0] = (long long)(__builtin_amdgcn_s_memtime() - t_start); r[1] = (long long) t_blocked; r[2] = (long long) t_report; r[3] = (long long) n_blocked; r[4] = pos;
        }
        return;
    }

    // ==================================================================================== helpers: the gatherer (h = 0) and the consumers (c = h - 1)
    const int h = wid - 1, c = h - 1, ht = tid - 64;
    const bool isG = h == 0;
    constexpr int NLN = 3;                                                 // float4 of the row per helper thread: n_embd <= 8448
    eng_wait w{ a.err, false, a.dbg, 0 };
    unsigned long long t_wait_land = 0, t_dot = 0, n_rows = 0, t_w_stat = 0, t_w_img = 0, t_w_ff = 0, t_w_att = 0, t_gather = 0, t_epi = 0;
    const int nblkE = E / 32, nblkF = FF / 32;
    const int nv = E >> 2;
    const int nwords_ff = (FF >> 2) + 2 * (FF >> 5), nwords_e = (E >> 2) + 2 * (E >> 5);
    const unsigned nx = (unsigned)((E + ENG_CHUNK - 1) / ENG_CHUNK), nf = (unsigned)((nwords_ff + ENG_CHUNK - 1) / ENG_CHUNK), na = (unsigned)((nwords_e + ENG_CHUNK - 1) / ENG_CHUNK);
    const size_t stamp_base = isG ? FQ_ENG_DBG_GSTAMPS : FQ_ENG_DBG_STAMPS;
    const bool stamps = isG || c == 0;
    auto timed_until = [&](unsigned addr, unsigned target, unsigned code, unsigned long long & acc) {
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        w.until(addr, target, code);
        acc += __builtin_amdgcn_s_memtime() - t0;
    };

    // rows [0, nrows) of a segment that starts at stream position seg_pos, dealt in runs of R consecutive rows: mine are
    // [R (c + NC k), R (c + NC k) + R), k = 0, 1, .. -- R rows share every activation read, and a wave only ever waits for ONE
    // contiguous run (a wave can always get its run: the ring is longer than a run plus the loader's restart slot)
    auto rows = [&](auto rtag, unsigned seg_pos, unsigned padded, int nrows, unsigned rs, int nblk, const fq_actcol & col, auto && sink) {
        constexpr int R = decltype(rtag)::value;
        const unsigned row_bytes = (unsigned)(nblk * TS);
        // nothing below my first run is needed by me (without this the loader would wait for rows of OTHER consumers to be
        // released by a wave that is itself waiting for its first rows to land)
        if (lane == 0) lds_st(ctl + eng_ctl::LOW + 4 * c, R * c < nrows ? seg_pos + (unsigned)(R * c) * rs : seg_pos + padded);
        const int npass = (nblk + 63) >> 6;                                // passes of 64 units = 64 blocks = 64 TS bytes of a row
        for (int i = R * c; i < nrows; i += R * ENG_NC) {
            const int last = i + R - 1 < nrows ? i + R - 1 : nrows - 1;
            const unsigned row0 = seg_pos + (unsigned) i * rs, rowl = seg_pos + (unsigned) last * rs;
            unsigned pr[R]; float v[R], acc[R];
            const unsigned p0 = row0 % (unsigned) RING;
#pragma unroll
            for (int r = 0; r < R; ++r) { const unsigned q = p0 + (unsigned)(i + r <= last ? r : last - i) * rs; pr[r] = q >= (unsigned) RING ? q - (unsigned) RING : q; acc[r] = 0.0f; }
            // a long row is taken three passes at a time as they land, and given back as it is consumed (ten 10 KB rows in progress would
            // otherwise BE the ring: nothing could be loaded ahead). The lane's units are added in the same ascending order.
            for (int ps = 0; ps < npass; ps += 3) {
                const int np = npass - ps < 3 ? npass - ps : 3;
                const unsigned upto = (unsigned)((ps + np) * 64 * TS);
                const unsigned need = rowl + (upto < row_bytes ? upto : row_bytes);
                const unsigned long long tw0 = __builtin_amdgcn_s_memtime();
                for (unsigned spins = 0; (int)(lds_ld_u(ctl + eng_ctl::LANDED) - need) < 0;) { if (!w.spin(spins, ENG_W_LAND, need, lds_ld_u(ctl + eng_ctl::LANDED))) break; __builtin_amdgcn_s_sleep(1); }
                const unsigned long long tw1 = __builtin_amdgcn_s_memtime();
                if (!nodots) {                                             // (tuning aid: no arithmetic, the hand-offs alone)
                    if (np == 3)      eng_pass_group<TYPE, RING, R, 3>(ring, pr, nblk, 64 * ps, col, lane, acc);
                    else if (np == 2) eng_pass_group<TYPE, RING, R, 2>(ring, pr, nblk, 64 * ps, col, lane, acc);
                    else              eng_pass_group<TYPE, RING, R, 1>(ring, pr, nblk, 64 * ps, col, lane, acc);
                }
                t_wait_land += tw1 - tw0; t_dot += __builtin_amdgcn_s_memtime() - tw1;
                if (ps + 3 < npass && lane == 0) lds_st(ctl + eng_ctl::LOW + 4 * c, row0 + upto);
            }
            n_rows += R;
#pragma unroll
            for (int r = 0; r < R; ++r) v[r] = wave_sum(acc[r]);
#pragma unroll
            for (int r = 0; r < R; ++r) if (i + r <= last) sink(i + r, v[r]);
            const int nx_ = i + R * ENG_NC;
            if (lane == 0) lds_st(ctl + eng_ctl::LOW + 4 * c, nx_ < nrows ? seg_pos + (unsigned) nx_ * rs : seg_pos + padded);
        }
    };
    std::integral_constant<int, 1> R1; std::integral_constant<int, 2> R2;
    auto rows_e = [&](unsigned seg_pos, unsigned padded, int nrows, const fq_actcol & col, auto && sink) {
        // (a round of NC runs is the consumers' working set: with runs of 4 rows it was the whole ring, the loader could never
        //  run ahead and loading and arithmetic took turns)
        if ((unsigned)(2 * ENG_NC) * rsE * 2u <= (unsigned) RING) rows(R2, seg_pos, padded, nrows, rsE, nblkE, col, sink);
        else                                                       rows(R1, seg_pos, padded, nrows, rsE, nblkE, col, sink);
    };
    auto rows_f = [&](unsigned seg_pos, unsigned padded, int nrows, const fq_actcol & col, auto && sink) {
        rows(R1, seg_pos, padded, nrows, rsF, nblkF, col, sink);
    };

    // ---- the residual row of LayerNorm number `li` (block li, or ln_f for li = n_layers) -> LDS -> Q8 image(s).
    // Every helper gathers its chunks; the last one to arrive takes the statistics; every helper normalises its share.
    // b: the block index for the phase stamps (-1: none)
    auto layer_norm_images = [&](int li, const float * w1, const float * b1, uint8_t * img1, const float * w2, const float * b2, uint8_t * img2, int b) {
        const bool from_mem = li == 0;
        const unsigned xtag = epoch0 + (unsigned) li;
        ln_row_regs<NLN> wr, br;
        ln_regs_issue_wb(w1, b1, E, ENG_HT, wr, br, ht);                   // (on their way while the row is gathered)
        // the row goes where the previous block's GELU image was: every consumer must be done with its Wdown rows
        if (li > 0) w.until(ctl + eng_ctl::B1_DONE, (unsigned) ENG_NC * (unsigned) li, ENG_W_B1);
        const unsigned long long tg0 = __builtin_amdgcn_s_memtime();
        if (isG && a.thin_loader) lds_st(ctl + eng_ctl::THIN, 1u);
        const unsigned own = from_mem ? eng_gather_mem(a.x_in, E, (unsigned *) xrow, h, lane, ctl + eng_ctl::PSUM)
                                      : eng_gather<true>(a.xg, xtag, E, (unsigned *) xrow, h, lane, w, ENG_W_XG, (a.debug_mode & 4) != 0, ctl + eng_ctl::PSUM);
        t_gather += __builtin_amdgcn_s_memtime() - tg0;
        if (stamps) ENG_STAMP_AT(stamp_base, 1);
        if (own) {
            lds_drain();                                                   // this wave's row values and partial sums are in LDS before it counts itself in
            unsigned old = 0;
            if (lane == 0) old = lds_add_rtn(ctl + eng_ctl::XG_DONE, own);
            old = __builtin_amdgcn_readfirstlane(old);
            if (old + own == nx * (unsigned)(li + 1)) {
                // ---- the last chunk has arrived: the row's mean by this wave (ggml.c:10577-10582), chunk sums in chunk order
                double s = 0.0;
                const unsigned long long pk = lds_ld64(ctl + eng_ctl::PSUM + 8u * (unsigned)(lane < (int) nx ? lane : 0));
                for (unsigned k = 0; k < nx; ++k) s += lane_get(__builtin_bit_cast(double, (long long) pk), (int) k);
                const float mean = (float)(s / (double) E);
                if (lane < nB) ldsf_st(eng_ctl::XRES + 4 * lane, xrow[sc.r0 + lane]);      // residual values of this workgroup's phase-B rows
                if (lane == 0) ldsf_st(eng_ctl::STAT, mean);
                lds_drain();
                if (lane == 0) lds_st(ctl + eng_ctl::LN_MEAN, (unsigned)(li + 1));
                if (a.thin_loader && lane == 0) lds_st(ctl + eng_ctl::THIN, 0u);
            }
        }
        timed_until(ctl + eng_ctl::LN_MEAN, (unsigned)(li + 1), ENG_W_STAT, t_w_stat);
        if (stamps) ENG_STAMP_AT(stamp_base, 2);
        // ---- second pass (ggml.c:10585-10591): every helper wave over its share of the row, which it keeps for the normalisation
        const float mean = ldsf_ld(eng_ctl::STAT);
        float4 xv[NLN];
        {
            double s2 = 0.0;
#pragma unroll
            for (int k = 0; k < NLN; ++k) {
                const int q4 = k * ENG_HT + ht;
                float4 v = ((const float4 *) xrow)[q4 < nv ? q4 : nv - 1];
                v.x -= mean; v.y -= mean; v.z -= mean; v.w -= mean;
                xv[k] = v;
                if (q4 < nv) { s2 += (double)(v.x * v.x); s2 += (double)(v.y * v.y); s2 += (double)(v.z * v.z); s2 += (double)(v.w * v.w); }
            }
            s2 = wave_sum(s2);
            if (lane == 0) lds_st64(ctl + eng_ctl::PSQ + 8u * (unsigned) h, (unsigned long long) __builtin_bit_cast(long long, s2));
            lds_drain();
            unsigned old2 = 0;
            if (lane == 0) old2 = lds_add_rtn(ctl + eng_ctl::S2_DONE, 1u);
            old2 = __builtin_amdgcn_readfirstlane(old2);
            if (old2 + 1u == (unsigned) ENG_NH * (unsigned)(li + 1)) {     // the last wave: partial sums in wave order -> scale
                const unsigned long long pk = lds_ld64(ctl + eng_ctl::PSQ + 8u * (unsigned)(lane < ENG_NH ? lane : 0));
                double t = 0.0;
#pragma unroll
                for (int k = 0; k < ENG_NH; ++k) t += lane_get(__builtin_bit_cast(double, (long long) pk), k);
                const float variance = (float)(t / (double) E);
                const float scale = 1.0f / sqrtf(variance + 1e-5f);
                if (lane == 0) ldsf_st(eng_ctl::STAT + 4, scale);
                lds_drain();
                if (lane == 0) lds_st(ctl + eng_ctl::LN_STAT, (unsigned)(li + 1));
            }
        }
        timed_until(ctl + eng_ctl::LN_STAT, (unsigned)(li + 1), ENG_W_STAT, t_w_stat);
        // the image(s) go where the previous block's attention image was: every consumer must be done with its Wo rows
        if (li > 0) w.until(ctl + eng_ctl::B2_DONE, (unsigned) ENG_NC * (unsigned) li, ENG_W_B2);
        const float scale = ldsf_ld(eng_ctl::STAT + 4);
        auto norm_quant = [&](const ln_row_regs<NLN> & wq, const ln_row_regs<NLN> & bq, uint8_t * img) {
            const act_image_ptr o = act_image_at(img, ACT, E);
#pragma unroll
            for (int k = 0; k < NLN; ++k) {
                const int q4 = k * ENG_HT + ht;
                if ((q4 & ~63) < nv) {                                     // wave-uniform
                    const bool lv = q4 < nv; const int j = lv ? q4 : nv - 1;
                    float4 v = xv[k];
                    const float4 ww = wq.t[k], bb = bq.t[k];
                    v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
                    v.x = v.x * ww.x + bb.x; v.y = v.y * ww.y + bb.y; v.z = v.z * ww.z + bb.z; v.w = v.w * ww.w + bb.w;
                    quant_q8_quad<ACT>(v, j, o, lv);
                }
            }
        };
        norm_quant(wr, br, img1);
        if (w2) {
            ln_row_regs<NLN> w2r, b2r;
            ln_regs_issue_wb(w2, b2, E, ENG_HT, w2r, b2r, ht);
            norm_quant(w2r, b2r, img2);
        }
        lds_drain();
        if (lane == 0) lds_add(ctl + eng_ctl::IMG_DONE, 1u);
        timed_until(ctl + eng_ctl::IMG_DONE, (unsigned) ENG_NH * (unsigned)(li + 1), ENG_W_IMG, t_w_img);
        if (stamps) ENG_STAMP_AT(stamp_base, 3);
    };

    const fq_actcol col_e  = { (const int8_t *) img_e,  (const float *)(img_e + fq_act_d_off(ACT, E)),  (const void *)(img_e + fq_act_aux_off(ACT, E)) };
    const fq_actcol col_e2 = { (const int8_t *) img_e2, (const float *)(img_e2 + fq_act_d_off(ACT, E)), (const void *)(img_e2 + fq_act_aux_off(ACT, E)) };
    const fq_actcol col_att = { (const int8_t *) img_att, (const float *)(img_att + fq_act_d_off(ACT, E)), (const void *)(img_att + fq_act_aux_off(ACT, E)) };
    const fq_actcol col_ff = { (const int8_t *) img_ff, (const float *)(img_ff + fq_act_d_off(ACT, FF)), (const void *)(img_ff + fq_act_aux_off(ACT, FF)) };

    if (a.debug_mode == 1) {                                               // tuning aid: the loader alone -- rows are released as soon as they land
        if (isG) return;
        auto drain = [&](unsigned seg_pos, unsigned padded, int nrows, unsigned rs, int nblk) {
            const unsigned row_bytes = (unsigned)(nblk * TS);
            for (int i = c; i < nrows; i += ENG_NC) {
                const unsigned need = seg_pos + (unsigned) i * rs + row_bytes;
                for (unsigned spins = 0; (int)(lds_ld_u(ctl + eng_ctl::LANDED) - need) < 0;) { if (!w.spin(spins, ENG_W_LAND, need, 0)) break; __builtin_amdgcn_s_sleep(1); }
                const int nx_ = i + ENG_NC;
                if (lane == 0) lds_st(ctl + eng_ctl::LOW + 4 * c, nx_ < nrows ? seg_pos + (unsigned) nx_ * rs : seg_pos + padded);
            }
            if (c >= nrows && lane == 0) lds_st(ctl + eng_ctl::LOW + 4 * c, seg_pos + padded);
        };
        for (int b = 0; b < a.n_layers; ++b) {
            const unsigned base = (unsigned) b * p_blk;
            drain(base, pA1, nA1, rsE, nblkE); drain(base + pA1, pA2, nA2, rsE, nblkE);
            drain(base + pA1 + pA2, pB1, nB, rsF, nblkF); drain(base + pA1 + pA2 + pB1, pB2, nB, rsE, nblkE);
        }
        if (a.lm_head) drain((unsigned) a.n_layers * p_blk, pH, nH, rsE, nblkE);
        return;
    }

#define ENG_STAMP(slot) do { if (stamps) ENG_STAMP_AT(stamp_base, slot); } while (0)
    for (int b = 0; b < a.n_layers; ++b) {
        const fq_engine_layer & L = a.layers[b];
        const unsigned base = (unsigned) b * p_blk;
        const unsigned tag = epoch0 + (unsigned) b + 1u;                   // tag of everything block b publishes
        const unsigned bp1 = (unsigned)(b + 1);
        w.blk = b;
        ENG_STAMP(0);
        // ---- residual row -> LayerNorm image(s)
        layer_norm_images(b, L.ln_w, L.ln_b, img_e, a.two_norms ? L.ln2_w : nullptr, L.ln2_b, img_e2, b);
        const int gA = nA2 / 32;
        const unsigned cnt_target = 32u * bp1;
        if (isG) {
            // ---- the Wup epilogues of this workgroup's groups, in the order their rows are dealt (kernels_decode.hip, k_gemv_ln epilogue)
            const unsigned long long te0 = __builtin_amdgcn_s_memtime();
            for (int gl = 0; gl < gA;) {
                for (unsigned spins = 0; (int)(lds_ld_u(ctl + eng_ctl::CNT + 4 * gl) - cnt_target) < 0;) { if (!w.spin(spins, ENG_W_GROUP, (unsigned) gl, lds_ld_u(ctl + eng_ctl::CNT + 4 * gl))) break; __builtin_amdgcn_s_sleep(1); }
                // lanes 0..31: group gl; lanes 32..63: group gl + 1 if it is complete as well (one table round trip for both), else a mirror
                const bool two = gl + 1 < gA && (int)(lds_ld_u(ctl + eng_ctl::CNT + 4 * (gl + 1)) - cnt_target) >= 0;
                const int j = lane & 31, half = lane >> 5;
                const int myg = gl + (two ? half : 0);
                const bool st = two || half == 0;
                float v = ldsf_ld(eng_ctl::OUT + 4 * (32 * myg + j));
                const int g = sc.ug0 + myg;                                // block index in the FF-long image
                v = h2f_bits(a.gelu_tab[f2h_bits(v)]);                     // ggml.c:3477-3484
                const float amax = reduce32(fabsf(v), op_max());
                const float d  = amax / 127.0f;
                const float id = d ? 1.0f / d : 0.0f;
                const int q = round_half_away(v * id);
                const int s = reduce32(q, op_add());
                unsigned wq = (unsigned) q & 0xFFu;                        // 4 lanes -> one word of qs
                wq |= ((unsigned) __shfl_down((int) wq, 1) & 0xFFu) << 8;
                wq |= ((unsigned) __shfl_down((int) wq, 2) & 0xFFFFu) << 16;
                if (st && (j & 3) == 0) gran_st(a.ffg + 8 * g + (j >> 2), tag, wq);
                if (st && j == 0) {
                    const int wd = (FF >> 2) + g, wa = wd + (FF >> 5);
                    if (ACT == FQ_Q8_0) { gran_st(a.ffg + wd, tag, __builtin_bit_cast(unsigned, h2f_bits(f2h_bits(d)))); gran_st(a.ffg + wa, tag, (unsigned) s); }
                    else                { gran_st(a.ffg + wd, tag, __builtin_bit_cast(unsigned, d)); gran_st(a.ffg + wa, tag, __builtin_bit_cast(unsigned, (float) s * d)); }
                }
                gl += two ? 2 : 1;
            }
            t_epi += __builtin_amdgcn_s_memtime() - te0;
            ENG_STAMP(4);
        } else {
            // ---- phase A: rows of Wqkv (norm image: the attention norm when the block has two) and Wup
            // Wqkv rows: q / k / v values go out one by one, before the Wup rows (the attention starts as early as it can)
            rows_e(base, pA1, nA1, a.two_norms ? col_e2 : col_e, [&](int i, float v) { if (lane == 0) gran_st(a.qkvg + sc.qg0 + i, tag, __builtin_bit_cast(unsigned, v)); });
            rows_e(base + pA1, pA2, nA2, col_e, [&](int i, float v) { if (lane == 0) { ldsf_st(eng_ctl::OUT + 4 * i, v); lds_add(ctl + eng_ctl::CNT + 4 * (i >> 5), 1u); } });
            if (lane == 0) lds_add(ctl + eng_ctl::A_DONE, 1u);
            ENG_STAMP(4);
        }
        // ---- the whole GELU image -> LDS (where the f32 row was: every helper has passed IMG_DONE), one chunk per helper
        {
            const unsigned long long tg0 = __builtin_amdgcn_s_memtime();
            if (isG && a.thin_loader && lane == 0) lds_st(ctl + eng_ctl::THIN, 1u);
            const unsigned own = eng_gather<false>(a.ffg, tag, nwords_ff, (unsigned *) img_ff, h, lane, w, ENG_W_FG, (a.debug_mode & 16) != 0, 0u);
            t_gather += __builtin_amdgcn_s_memtime() - tg0;
            if (own) { lds_drain(); if (lane == 0) lds_add(ctl + eng_ctl::FG_DONE, own); }
            timed_until(ctl + eng_ctl::FG_DONE, nf * bp1, ENG_W_FGD, t_w_ff);
            if (isG && a.thin_loader && lane == 0) lds_st(ctl + eng_ctl::THIN, 0u);
        }
        ENG_STAMP(5);
        if (!isG) {
            rows_f(base + pA1 + pA2, pB1, nB, col_ff, [&](int i, float v) { if (lane == 0) ldsf_st(eng_ctl::OUTB + 4 * i, v); });
            lds_drain();
            if (lane == 0) lds_add(ctl + eng_ctl::B1_DONE, 1u);
        }
        ENG_STAMP(6);
        // ---- the attention output image -> LDS (where the LayerNorm images were: every consumer must be done with phase A)
        {
            w.until(ctl + eng_ctl::A_DONE, (unsigned) ENG_NC * bp1, ENG_W_ADONE);
            const unsigned long long tg0 = __builtin_amdgcn_s_memtime();
            if (isG && a.thin_loader && lane == 0) lds_st(ctl + eng_ctl::THIN, 1u);
            // (the gatherer has nothing else to do while the Wdown rows are dotted: all chunks are its)
            const unsigned own = isG ? eng_gather<false>(a.attg, tag, nwords_e, (unsigned *) img_att, 0, lane, w, ENG_W_AG, (a.debug_mode & 32) != 0, 0u, 1) : 0u;
            t_gather += __builtin_amdgcn_s_memtime() - tg0;
            if (own) { lds_drain(); if (lane == 0) lds_add(ctl + eng_ctl::AG_DONE, own); }
            if (isG && a.thin_loader) { w.until(ctl + eng_ctl::AG_DONE, na * bp1, ENG_W_AGD); if (lane == 0) lds_st(ctl + eng_ctl::THIN, 0u); }
        }
        if (!isG) {
            timed_until(ctl + eng_ctl::AG_DONE, na * bp1, ENG_W_AGD, t_w_att);
            w.until(ctl + eng_ctl::B1_DONE, (unsigned) ENG_NC * bp1, ENG_W_B1);      // (a row's down-projection dot was left by another wave)
            ENG_STAMP(7);
            rows_e(base + pA1 + pA2 + pB1, pB2, nB, col_att, [&](int i, float v) {
                if (lane == 0) {
                    const float xn = (ldsf_ld(eng_ctl::OUTB + 4 * i) + v) + ldsf_ld(eng_ctl::XRES + 4 * i);     // libfalcon.cpp:2399-2400
                    const int row = sc.r0 + i;
                    gran_st(a.xg + row, tag, __builtin_bit_cast(unsigned, xn));
                    a.x[row] = xn;
                    if (a.hidden) a.hidden[(size_t)(b + 1) * E + row] = xn;
                }
            });
            lds_drain();
            if (lane == 0) lds_add(ctl + eng_ctl::B2_DONE, 1u);
        } else ENG_STAMP(7);
    }
    // ---- ln_f + lm_head (last stage): logits and the per-32-row greedy candidates
    if (a.lm_head) {
        w.blk = a.n_layers;
        layer_norm_images(a.n_layers, a.lnf_w, a.lnf_b, img_e, nullptr, nullptr, nullptr, -1);
        const unsigned hbase = (unsigned) a.n_layers * p_blk;
        const int ngroups = (nH + 31) / 32;
        if (!isG) {
            rows_e(hbase, pH, nH, col_e, [&](int i, float v) { if (lane == 0) { ldsf_st(eng_ctl::OUT + 4 * i, v); lds_add(ctl + eng_ctl::CNTH + 4 * (i >> 5), 1u); } });
        } else {
            // (a partial last group: its missing rows never arrive -- count them in)
            if (lane == 0 && (nH & 31)) lds_add(ctl + eng_ctl::CNTH + 4 * (nH >> 5), (unsigned)(32 - (nH & 31)));
            for (int gl = 0; gl < ngroups; ++gl) {
                for (unsigned spins = 0; lds_ld_u(ctl + eng_ctl::CNTH + 4 * gl) < 32u;) { if (!w.spin(spins, ENG_W_GROUP, (unsigned) gl, lds_ld_u(ctl + eng_ctl::CNTH + 4 * gl))) break; __builtin_amdgcn_s_sleep(1); }
                const int j = lane & 31;
                const int row = sc.hg0 * 32 + 32 * gl + j;
                const float v = ldsf_ld(eng_ctl::OUT + 4 * (32 * gl + j));
                if (lane < 32 && row < a.V) a.logits[row] = v;
                float bv = row < a.V ? v : -INFINITY; int bi = row;
#pragma unroll
                for (int o = 1; o < 32; o <<= 1) {
                    const float ov = __shfl_xor(bv, o); const int oi = __shfl_xor(bi, o);
                    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
                }
                if (lane == 0) { a.argmax_val[sc.hg0 + gl] = bv; a.argmax_idx[sc.hg0 + gl] = bi; }
            }
        }
    }
    if (a.dbg && lane == 0 && (isG || c == 0)) {
        long long * r = a.dbg + FQ_ENG_DBG_COUNTERS + (size_t) blockIdx.x * 16;
        if (isG) { r[13] = (long long) t_gather; r[14] = (long long) t_epi; r[15] = (long long) t_w_stat; }
        else { r[5] = (long long) t_wait_land; r[6] = (long long) t_dot; r[7] = (long long) n_rows; r[8] = (long long) t_w_stat; r[9] = (long long) t_w_img;
               r[10] = (long long) t_w_ff; r[11] = (long long) t_w_att; r[12] = (long long) t_gather; }
    }
}

// ---- the tag of a launch's hand-offs: advanced before every launch (never 0); granules carry their tag, nothing else is cleared
__global__ void k_engine_epoch(unsigned * epoch_word, unsigned step) {
    if (threadIdx.x == 0) {
        unsigned e = *epoch_word + step;
        if (e < step + 1u) e = 1u;              // wrapped
        *epoch_word = e;
    }
}

// =============================================================================================== host side
// Work split of one block over `n_stream` workgroups (host): 32-row groups of [Wqkv | Wup] and rows of [Wdown, Wo] so that every
// workgroup streams about the same number of bytes per block; the first group of as many workgroups as there are Wqkv
// groups is a Wqkv group, so that q / k / v are complete -- and the attention can start -- after the first group round.
bool fq_engine_plan(int type, int E, int FF, int qkv_rows, int V, bool with_head, int n_stream, std::vector<fq_engine_sched> & out, int * max_groups, int * max_rows) {
    const fq_type_desc d = fq_desc(type);
    if (d.blck != 32 || E % 32 || FF % 32 || qkv_rows % 32 || n_stream < 1) return false;
    const int gu = FF / 32;
    out.assign((size_t) n_stream, fq_engine_sched{});
    int q = 0, u = 0, mg = 0, mr = 0, r = 0;
    const double rows_a = (double)(FF + qkv_rows) / n_stream;             // phase-A rows per workgroup
    double acc_rows = 0.0;
    for (int s = 0; s < n_stream; ++s) {
        fq_engine_sched & o = out[(size_t) s];
        const int nu = (int)(((int64_t)(s + 1) * gu) / n_stream - ((int64_t) s * gu) / n_stream);
        // Wqkv rows: whatever brings the cumulative phase-A rows up to the even share (multiples of 4: whole runs)
        int nq = (int)(rows_a * (s + 1) - acc_rows - 32.0 * nu + 0.5);
        nq = nq < 0 ? 0 : (nq + 2) / 4 * 4;
        if (s == n_stream - 1 || q + nq > qkv_rows) nq = qkv_rows - q;
        o.qg0 = q; q += nq; o.qg1 = q;
        o.ug0 = u; u += nu; o.ug1 = u;
        acc_rows += nq + 32.0 * nu;
        const int nb = (int)(((int64_t)(s + 1) * E) / n_stream - ((int64_t) s * E) / n_stream);      // phase A is even, so phase B is dealt evenly too
        o.r0 = r; r += nb; o.r1 = r;
        if (nu > mg) mg = nu;
        if (nb > mr) mr = nb;
    }
    if (q != qkv_rows || u != gu || r != E) return false;
    if (with_head) {
        const int gh = (V + 31) / 32;
        for (int s = 0; s < n_stream; ++s) {
            out[(size_t) s].hg0 = (int)(((int64_t) s * gh) / n_stream);
            out[(size_t) s].hg1 = (int)(((int64_t)(s + 1) * gh) / n_stream);
            if (out[(size_t) s].hg1 - out[(size_t) s].hg0 > mg) mg = out[(size_t) s].hg1 - out[(size_t) s].hg0;
        }
    }
    *max_groups = mg; *max_rows = mr;
    return mg <= 12 && mr <= 64;
}

size_t fq_engine_lds_bytes(int type, int nslot, int64_t E, int64_t FF, int n_layers, int two_norms) {
    const int act = fq_desc(type).act_type;
    return (size_t) nslot * ENG_SLOT + ENG_MIRROR + eng_region_r(act, E, FF) + eng_region_s(act, E, two_norms) + eng_ctl::BYTES + 32 * (size_t) n_layers + 16;
}

int fq_engine_threads() { return ENG_NT; }

// launches the pre-kernel (tag) and the engine; false = this configuration is not supported (nothing launched)
bool fq_launch_decode_engine(const fq_engine_args & a, int nslot, size_t lds_bytes, hipStream_t st) {
    const int grid = a.n_attn + a.n_stream;
    if ((a.type != FQ_Q4_0 && a.type != FQ_Q4_1 && a.type != FQ_Q5_0 && a.type != FQ_Q5_1 && a.type != FQ_Q8_0) || (nslot != 8 && nslot != 7 && nslot != 6 && nslot != 4) ||
        a.E > 4 * 3 * ENG_HT || a.E > 16 * ENG_CHUNK) return false;
    hipLaunchKernelGGL(k_engine_epoch, dim3(1), dim3(64), 0, st, const_cast<unsigned *>(a.epoch_word), (unsigned)(a.n_layers + 2));
#define FQ_ENG_LAUNCH(T, NS) { \
        static size_t g = 0; if (lds_bytes > g) { HIP_CHECK(hipFuncSetAttribute((const void *) k_decode_engine<T, NS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds_bytes)); g = lds_bytes; } \
        hipEvent_t e0_ = nullptr, e1_ = nullptr; fq_prof_events(&e0_, &e1_); \
        if (e0_) hipExtLaunchKernelGGL((k_decode_engine<T, NS>), dim3((unsigned) grid), dim3(ENG_NT), lds_bytes, st, e0_, e1_, 0, a); \
        else     hipLaunchKernelGGL((k_decode_engine<T, NS>), dim3((unsigned) grid), dim3(ENG_NT), lds_bytes, st, a); }
#ifdef ENG_TUNE_ONLY_Q4_0
#define FQ_ENG_CASE(T) case T: if (T == FQ_Q4_0 && nslot == 7) FQ_ENG_LAUNCH(FQ_Q4_0, 7) else return false; break;
#else
#define FQ_ENG_CASE(T) case T: if (nslot == 8) FQ_ENG_LAUNCH(T, 8) else if (nslot == 7) FQ_ENG_LAUNCH(T, 7) else if (nslot == 6) FQ_ENG_LAUNCH(T, 6) else if (nslot == 4) FQ_ENG_LAUNCH(T, 4) else return false; break;
#endif
    switch (a.type) {
        FQ_ENG_CASE(FQ_Q4_0) FQ_ENG_CASE(FQ_Q4_1) FQ_ENG_CASE(FQ_Q5_0) FQ_ENG_CASE(FQ_Q5_1) FQ_ENG_CASE(FQ_Q8_0)
        default: return false;
    }
#undef FQ_ENG_CASE
#undef FQ_ENG_LAUNCH
    return true;
}
