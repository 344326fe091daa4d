// fq_ring_dev.h -- the ring form's two roles as re-usable device code (gfx950, wave64), shared by the kernels of kernels_ringk.hip:
//   ring_loader<NSLOT>   wave 0 of a workgroup: streams byte ranges of weight rows into an LDS ring with global_load_lds_dwordx4 (LDS-DMA),
//                        bounded by ring space only; the protocol is kernels_ring.hip's (LANDED = bytes that have arrived, LOW[c] = stream
//                        position below which consumer c needs nothing, a 2 KiB mirror of the ring's start behind its end so that a
//                        reader never wraps inside a column)
//   ring_rows<...>       a consumer wave: the rows r == c (mod NC) of a segment, dotted out of the ring against an activation image in LDS with
//                        fq_unit<TYPE> (fq_units.h) -- per lane the units of a row are added in ascending order, then the wave butterfly: the
//                        arithmetic and association of k_gemv / k_gemv_ln / k_gemv_out, hence the same bits, for all ten weight formats
// One N = 1 pass of ggml_compute_forward_mul_mat_q_f32 (ggml.c:11318-11529) per weight matrix; the CUDA twin is dequantize_mul_mat_vec*
// (ggml-cuda.cu:475-845, 1120-1171). Included once per translation unit (anonymous namespace).
#pragma once
#include "fq_engine_dev.h"
#include "fq_kdot.h"

namespace {

constexpr unsigned RING_XISSUED = eng_ctl::A_DONE;        // helper waves that have issued the loads of their prologue (the loader starts behind them)

template <int NSLOT>
struct ring_loader {
    static constexpr unsigned RING = (unsigned)(NSLOT * ENG_SLOT), NP = (unsigned)(NSLOT * 16), MIRP = (unsigned)(ENG_MIRROR / 1024);
    unsigned ctl, mstart, mend, ml, voff0, pos, rp, freed, reported, flag_addr;
    int lane;
    bool nospace = false;                                                 // tuning aid: never wait for ring space (results are garbage; the stream's own pace)
    eng_voff vo;
    eng_wait w;

    __device__ __forceinline__ ring_loader(uint8_t * ring, unsigned ctl_, unsigned * err, int lane_) {
        ctl = ctl_; lane = lane_; pos = 0; rp = 0; freed = 0; reported = 0;
        w = eng_wait{ err, false, nullptr, 0 };
        voff0 = (unsigned) lane * 16u;
#pragma unroll
        for (int p = 0; p < 16; ++p) vo.v[p] = voff0 + 1024u * (unsigned) p;
        mstart = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t) ring); mend = mstart + RING; ml = mstart;
        flag_addr = ctl + eng_ctl::THIN;                                  // (sampled by the batch statement; unused)
    }
    __device__ __forceinline__ void report(unsigned upto) { if ((int)(upto - reported) > 0) { reported = upto; if (lane == 0) lds_st(ctl + eng_ctl::LANDED, upto); } }
    __device__ __forceinline__ void report_keep(unsigned keep_kb) { const unsigned back = (keep_kb + MIRP) * 1024u; if (pos > back) report(pos - back); }
    __device__ __forceinline__ unsigned low_water() {
        unsigned v = lds_ld(ctl + eng_ctl::LOW + 4 * (lane < 16 ? lane : 0));
        v = (unsigned) wave_reduce((int) v, [](int x, int y) { return (unsigned) x < (unsigned) y ? x : y; });
        return (unsigned) __builtin_amdgcn_readfirstlane(v);
    }
    __device__ __forceinline__ void wait_space(unsigned bytes) {
        if (nospace || pos + bytes - freed <= RING) return;
        freed = low_water();
        if (pos + bytes - freed <= RING) return;
        for (unsigned spins = 0;;) {
            const unsigned inflight = pos - reported;
#define RING_LAND_STEP(N) if (inflight > (N + MIRP) * 1024u) { asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); report_keep(N); } else
            RING_LAND_STEP(44) RING_LAND_STEP(40) RING_LAND_STEP(36) RING_LAND_STEP(32) RING_LAND_STEP(28) RING_LAND_STEP(24) RING_LAND_STEP(20) RING_LAND_STEP(16)
            RING_LAND_STEP(12) RING_LAND_STEP(8) RING_LAND_STEP(4)
            if (inflight > 0u) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); report(pos); }
            else __builtin_amdgcn_s_sleep(1);
#undef RING_LAND_STEP
            freed = low_water();
            if (pos + bytes - freed <= RING) break;
            if (!w.spin(spins, ENG_W_RING, pos, freed)) break;
        }
    }
    __device__ __forceinline__ void after_issue() {
        if (pos - reported >= 50u * 1024u) { asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); report_keep(32u); }
    }
    // the pieces that land in the ring's first MIRP KiB are loaded a second time, behind its end
    __device__ __forceinline__ void mirror(const uint8_t * src, unsigned rp0, unsigned n) {
#pragma unroll
        for (unsigned t = 0; t < MIRP; ++t) {
            unsigned k = t + NP - rp0; k = k >= NP ? k - NP : k;
            if (k < n) glds_piece(src + k * 1024u, voff0, mstart + (NP + t) * 1024u);
        }
    }
    // `padded` bytes (a multiple of 1 KiB; the source is readable that far) from src, 16-byte aligned
    __device__ __forceinline__ void seg(const uint8_t * src_, unsigned padded_) {
        // the DMA statements take their base and LDS address in scalar registers: make the (wave-uniform) arguments provably so
        const unsigned long long sv = (unsigned long long)(uintptr_t) src_;
        const unsigned slo = __builtin_amdgcn_readfirstlane((unsigned) sv), shi = __builtin_amdgcn_readfirstlane((unsigned)(sv >> 32));
        const uint8_t * src = (const uint8_t *)(uintptr_t)(((unsigned long long) shi << 32) | slo);
        const unsigned padded = __builtin_amdgcn_readfirstlane(padded_);
        // (the state carried from an earlier segment reaches this one through the early-out's control-flow join, which hipcc treats as divergent)
        ml = __builtin_amdgcn_readfirstlane(ml); rp = __builtin_amdgcn_readfirstlane(rp); pos = __builtin_amdgcn_readfirstlane(pos);
        freed = __builtin_amdgcn_readfirstlane(freed); reported = __builtin_amdgcn_readfirstlane(reported);
        const unsigned nfull = padded >> 14, ntail = (padded >> 10) & 15u;
        if (w.dead) return;
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
    }
    // the end of the stream is reported as it lands, not in one piece: the consumers' last runs start while the final pieces are in flight
    __device__ __forceinline__ void finish() {
#define RING_END_STEP(N) if (pos - reported > (N + MIRP) * 1024u) { asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); report_keep(N); }
        RING_END_STEP(40) RING_END_STEP(32) RING_END_STEP(24) RING_END_STEP(16) RING_END_STEP(10) RING_END_STEP(5)
#undef RING_END_STEP
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        report(pos);
    }
};

__device__ __forceinline__ unsigned ring_pad1k(unsigned v) { return (v + 1023u) & ~1023u; }

// U passes (64 units each) of R rows out of the ring: all loads first, then the dots; per lane the units are added in ascending order.
// pos[r] = ring offset of row r's first byte; nblk = ggml blocks per row; p0 = index of the first pass.
template <int TYPE, int RING, int R, int U>
__device__ __forceinline__ void ring_pass_group(const uint8_t * ring, const unsigned (&pos)[R], int nblk, int nunits, int p0, const fq_actcol & col, int lane, float (&acc)[R]) {
    fq_unit_regs regs[U][R];
#pragma unroll
    for (int p = 0; p < U; ++p) {
        if constexpr (fq_lay<TYPE>::UPC == 64) {
            constexpr int CB = fq_lay<TYPE>::CB, UPS = fq_lay<TYPE>::UPS;
            constexpr unsigned COLB = (unsigned)(CB * fq_lay<TYPE>::TS);
            const int c = p0 + p;                                          // one pass = one column (scalar)
            const int rem = nblk - CB * c, nbc = rem < CB ? rem : CB;
            const int nu = nbc * UPS, ju = lane < nu ? lane : nu - 1;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                unsigned cb = pos[r] + (unsigned) c * COLB;               // scalar; < 2 RING: a row is shorter than the ring
                cb = cb >= (unsigned) RING ? cb - (unsigned) RING : cb;
                const fq_col k{ (const uint8_t *) __builtin_assume_aligned(ring + cb, 16), nbc };
                regs[p][r] = fq_unit<TYPE>::template load_at<false>(k, ju);
            }
        } else {                                                            // Q8_0: 32 blocks per column, the column base is per lane
#pragma unroll
            for (int r = 0; r < R; ++r) regs[p][r] = eng_unit_load_q8<RING>(ring, pos[r], 64 * (p0 + p), lane, nblk);
        }
    }
#pragma unroll
    for (int p = 0; p < U; ++p) {
        const int u = 64 * (p0 + p) + lane; const bool ok = u < nunits; const int uc = ok ? u : nunits - 1;
#pragma unroll
        for (int r = 0; r < R; ++r) { const float v = fq_unit<TYPE>::dot(regs[p][r], col, uc); acc[r] += ok ? v : 0.0f; }
    }
}

// Consumer c of NC: runs of R consecutive rows, round-robin, of a segment of `nrows` rows (row stride rs bytes, nblk ggml blocks each) whose first byte sits at
// stream position seg_pos (the segment occupies `padded` bytes of the stream). sink(i, v): row i's dot product, wave-uniform, called by every lane.
template <int TYPE, int RING, int R, int U, typename SINK>
__device__ __forceinline__ void ring_rows(const uint8_t * ring, unsigned ctl, int c, int NC, unsigned seg_pos, unsigned padded, int nrows, unsigned rs, int nblk,
                                          const fq_actcol & col, int lane, eng_wait & w, bool nodots, SINK && sink) {
    constexpr int TS = fq_desc(TYPE).tsize;
    constexpr unsigned PASSB = (unsigned)((64 / fq_lay<TYPE>::UPC) * fq_lay<TYPE>::CB * TS);            // bytes of one pass (64 units) of a row
    const int nunits = nblk * (fq_desc(TYPE).blck / fq_unit<TYPE>::ELEMS);
    const unsigned row_bytes = (unsigned)(nblk * TS);
    const int npass = (nunits + 63) >> 6;
    if (lane == 0) lds_st(ctl + eng_ctl::LOW + 4 * c, R * c < nrows ? seg_pos + (unsigned)(R * c) * rs : seg_pos + padded);
    for (int i = R * c; i < nrows; i += R * NC) {
        const int last = i + R - 1 < nrows ? i + R - 1 : nrows - 1;
        const unsigned row0 = seg_pos + (unsigned) i * rs, rowl = seg_pos + (unsigned) last * rs;
        unsigned pr[R]; float v[R], acc[R];
        const unsigned p0 = row0 % (unsigned) RING;
#pragma unroll
        for (int r = 0; r < R; ++r) { const unsigned q = p0 + (unsigned)(i + r <= last ? r : last - i) * rs; pr[r] = q >= (unsigned) RING ? q - (unsigned) RING : q; acc[r] = 0.0f; }
        for (int ps = 0; ps < npass; ps += U) {
            const int np = npass - ps < U ? npass - ps : U;
            const unsigned upto = (unsigned)(ps + np) * PASSB;
            const unsigned need = rowl + (upto < row_bytes ? upto : row_bytes);
            for (unsigned spins = 0; (int)(lds_ld_u(ctl + eng_ctl::LANDED) - need) < 0;) { if (!w.spin(spins, ENG_W_LAND, need, 0)) break; __builtin_amdgcn_s_sleep(1); }
            if (!nodots) {
                if (np == U) ring_pass_group<TYPE, RING, R, U>(ring, pr, nblk, nunits, ps, col, lane, acc);
                else if (U > 2 && np == 3) ring_pass_group<TYPE, RING, R, (U > 2 ? 3 : 1)>(ring, pr, nblk, nunits, ps, col, lane, acc);
                else if (U > 1 && np == 2) ring_pass_group<TYPE, RING, R, (U > 1 ? 2 : 1)>(ring, pr, nblk, nunits, ps, col, lane, acc);
                else ring_pass_group<TYPE, RING, R, 1>(ring, pr, nblk, nunits, ps, col, lane, acc);
            }
            if (ps + U < npass && lane == 0) lds_st(ctl + eng_ctl::LOW + 4 * c, row0 + upto);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) v[r] = wave_sum(acc[r]);
#pragma unroll
        for (int r = 0; r < R; ++r) if (i + r <= last) sink(i + r, v[r]);
        const int nx_ = i + R * NC;
        if (lane == 0) lds_st(ctl + eng_ctl::LOW + 4 * c, nx_ < nrows ? seg_pos + (unsigned) nx_ * rs : seg_pos + padded);
    }
}

// ---- the k-quants' fast path (fq_kdot.h): rows of whole columns, a consumer's activation slices resident in registers
template <int TYPE, int U> struct ring_kacts { typename fq_kdot<TYPE>::act_t a[U]; };

template <int TYPE, int U>
__device__ __forceinline__ void ring_kacts_load(const fq_actcol & col, int pass0, const typename fq_kdot<TYPE>::lane_t & L, ring_kacts<TYPE, U> & A) {
#pragma unroll
    for (int p = 0; p < U; ++p) A.a[p] = fq_kdot<TYPE>::act_load(col, pass0 + p, L);
}
// this lane's weight units of U consecutive passes (= columns) of one row out of the ring; column 0 starts at ring offset pos0 (< RING)
template <int TYPE, int RING, int U> struct ring_kw { typename fq_kdot<TYPE>::w_t w[U]; };
template <int TYPE, int RING, int U>
__device__ __forceinline__ void ring_kload(const uint8_t * ring, unsigned pos0, const typename fq_kdot<TYPE>::lane_t & L, ring_kw<TYPE, RING, U> & W) {
    constexpr unsigned COLB = (unsigned)(fq_lay<TYPE>::CB * fq_lay<TYPE>::TS);
#pragma unroll
    for (int p = 0; p < U; ++p) {
        unsigned cb = pos0 + (unsigned) p * COLB;                          // scalar; a column never wraps (the mirror)
        cb = cb >= (unsigned) RING ? cb - (unsigned) RING : cb;
        W.w[p] = fq_kdot<TYPE>::w_load((const uint8_t *) __builtin_assume_aligned(ring + cb, 16), L);
    }
}
// ... and their f32 terms
template <int TYPE, int RING, int U>
__device__ __forceinline__ void ring_kterms(const uint8_t * ring, unsigned pos0, const ring_kacts<TYPE, U> & A, const typename fq_kdot<TYPE>::lane_t & L, float (&t)[U]) {
    ring_kw<TYPE, RING, U> W;
    ring_kload<TYPE, RING, U>(ring, pos0, L, W);
#pragma unroll
    for (int p = 0; p < U; ++p) t[p] = fq_kdot<TYPE>::dot(W.w[p], A.a[p], L);
}
// ring_rows for rows of exactly U passes, R consecutive rows per trip (runs dealt round-robin): per row the lane's U terms added in ascending order, the
// wave butterfly -- ring_rows' bits. R = 2 for short rows (Q2_K / Q3_K: two columns): the trip's fixed costs (poll, butterfly, bookkeeping) are paid once per pair
// LONG rows (Wdown: 8 or 16 passes) by one wave per row, U passes per CHUNK: the chunk's weight units go into registers and its ring space is handed back at
// once, so a row occupies the ring only while it is landed-but-not-yet-loaded -- eleven waves on eleven consecutive 18 KiB rows then need a fraction of the
// ring instead of twice its size (the first k_ring_out held every row until its dot product was done: five consumers had data, 69 us). The activation slices
// come out of the image in LDS (lane-constant offsets, fq_kdot.h); per lane the passes are added in ascending order, then the wave butterfly: ring_rows' bits.
template <int TYPE, int RING, int U, typename SINK>
__device__ __forceinline__ void ring_rows_kc(const uint8_t * ring, unsigned ctl, int c, int NC, unsigned seg_pos, unsigned padded, int nrows, unsigned rs, int npass,
                                             const fq_actcol & col, const typename fq_kdot<TYPE>::lane_t & L, int lane, eng_wait & w, bool nodots, SINK && sink) {
    constexpr unsigned COLB = (unsigned)(fq_lay<TYPE>::CB * fq_lay<TYPE>::TS);
    if (lane == 0) lds_st(ctl + eng_ctl::LOW + 4 * c, c < nrows ? seg_pos + (unsigned) c * rs : seg_pos + padded);
    if (c >= nrows) return;
    // the wave's chunks as one sequence (row i = c, c + NC, ..; chunk ch = 0, U, .. of each): chunk q + 1 is fetched into a second register set BEFORE
    // chunk q is dotted when it has landed by then (else after), so a landed chunk waits in the ring for a free wave, not for its predecessor's arithmetic
    auto chunk_pos = [&](int i, int ch) { return seg_pos + (unsigned) i * rs + (unsigned) ch * COLB; };
    auto landed = [&](unsigned need) { return (int)(lds_ld_u(ctl + eng_ctl::LANDED) - need) >= 0; };
    auto wait_landed = [&](unsigned need) { for (unsigned spins = 0; !landed(need);) { if (!w.spin(spins, ENG_W_LAND, need, 0)) break; __builtin_amdgcn_s_sleep(1); } };
    // after chunk (i, ch) sits in registers this wave needs nothing below the chunk that follows it
    auto release_after = [&](int i, int ch) {
        unsigned nxt;
        if (ch + U < npass) nxt = chunk_pos(i, ch + U);
        else nxt = i + NC < nrows ? chunk_pos(i + NC, 0) : seg_pos + padded;
        if (lane == 0) lds_st(ctl + eng_ctl::LOW + 4 * c, nxt);            // (behind the reads: the LDS serves a wave's requests in order)
    };
    ring_kw<TYPE, RING, U> Wa, Wb;
    int i = c, ch = 0;
    wait_landed(chunk_pos(i, ch) + (unsigned) U * COLB);
    ring_kload<TYPE, RING, U>(ring, chunk_pos(i, ch) % (unsigned) RING, L, Wa);
    release_after(i, ch);
    float acc = 0.0f;
    for (;;) {
        // the chunk after (i, ch)
        int ni = i, nch = ch + U;
        if (nch >= npass) { ni = i + NC; nch = 0; }
        const bool more = ni < nrows;
        const unsigned npos = more ? chunk_pos(ni, nch) : 0u;
        bool fetched = false;
        if (more && landed(npos + (unsigned) U * COLB)) { ring_kload<TYPE, RING, U>(ring, npos % (unsigned) RING, L, Wb); release_after(ni, nch); fetched = true; }
        if (!nodots) {
#pragma unroll
            for (int p = 0; p < U; ++p) acc += fq_kdot<TYPE>::dot(Wa.w[p], fq_kdot<TYPE>::act_load(col, ch + p, L), L);
        }
        if (ch + U >= npass) { sink(i, wave_sum(acc)); acc = 0.0f; }
        if (!more) break;
        if (!fetched) { wait_landed(npos + (unsigned) U * COLB); ring_kload<TYPE, RING, U>(ring, npos % (unsigned) RING, L, Wb); release_after(ni, nch); }
        Wa = Wb; i = ni; ch = nch;
    }
}

struct ring_no_hook { __device__ __forceinline__ void operator()() const {} };
// once(): called once, after the consumer's first trip (work that must not sit in front of the first rows but has to be done well before the segment ends)
template <int TYPE, int RING, int U, int R, typename SINK, typename ONCE = ring_no_hook>
__device__ __forceinline__ void ring_rows_k(const uint8_t * ring, unsigned ctl, int c, int NC, unsigned seg_pos, unsigned padded, int nrows, unsigned rs,
                                            const ring_kacts<TYPE, U> & A, const typename fq_kdot<TYPE>::lane_t & L, int lane, eng_wait & w, bool nodots, SINK && sink,
                                            ONCE && once = ring_no_hook()) {
    bool first_trip = true;
    constexpr unsigned ROWB = (unsigned)(U * fq_lay<TYPE>::CB * fq_lay<TYPE>::TS);
    if (lane == 0) lds_st(ctl + eng_ctl::LOW + 4 * c, R * c < nrows ? seg_pos + (unsigned)(R * c) * rs : seg_pos + padded);
    for (int i = R * c; i < nrows; i += R * NC) {
        const int last = i + R - 1 < nrows ? i + R - 1 : nrows - 1;
        const unsigned row0 = seg_pos + (unsigned) i * rs, need = seg_pos + (unsigned) last * rs + ROWB;
        for (unsigned spins = 0; (int)(lds_ld_u(ctl + eng_ctl::LANDED) - need) < 0;) { if (!w.spin(spins, ENG_W_LAND, need, 0)) break; __builtin_amdgcn_s_sleep(1); }
        // the trip's weight units into registers, then the ring space is handed back BEFORE the arithmetic (the LDS serves requests in order: the reads
        // are ahead of the LOW store): the loader's look-ahead is what is left of the ring beyond the oldest row still needed -- held through dots,
        // butterfly and epilogue, ten rows kept ~46 KiB of the 96 busy and the stream ran at the latency-bound 22 KB/us per CU
        ring_kw<TYPE, RING, U> W[R];
        const unsigned p0 = row0 % (unsigned) RING;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            unsigned q = p0 + (unsigned)(i + r <= last ? r : last - i) * rs;          // (a run's second row may not exist: the first is dotted twice, the copy dropped)
            q = q >= (unsigned) RING ? q - (unsigned) RING : q;
            ring_kload<TYPE, RING, U>(ring, q, L, W[r]);
        }
        const int nx_ = i + R * NC;
        if (lane == 0) lds_st(ctl + eng_ctl::LOW + 4 * c, nx_ < nrows ? seg_pos + (unsigned) nx_ * rs : seg_pos + padded);
        float acc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = 0.0f;
        if (!nodots) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
#pragma unroll
                for (int p = 0; p < U; ++p) acc[r] += fq_kdot<TYPE>::dot(W[r].w[p], A.a[p], L);
            }
        }
        float v[R];
#pragma unroll
        for (int r = 0; r < R; ++r) v[r] = wave_sum(acc[r]);
#pragma unroll
        for (int r = 0; r < R; ++r) if (i + r <= last) sink(i + r, v[r]);
        if (first_trip) { first_trip = false; once(); }
    }
    if (first_trip) once();
}

}   // namespace
