// fq_engine_dev.h -- device pieces of the ring forms of the fused decode launches (kernels_ring.hip, kernels_ringk.hip; first written for the persistent
// decode engine of rounds 2-3, which was measured slower and removed in round 6 -- NOTEBOOK section 4): the LDS-DMA loader primitives, LDS control words, row dots out of the ring, bounded waits,
// chunked gathers. Included once per translation unit (anonymous namespace).
#pragma once
#include "fq_block_dev.h"
#include "kernels.h"
#include "fq_ref_chain.h"
#include <type_traits>

namespace {

constexpr int ENG_NH = 11;                 // helper waves: the gatherer (helper 0) + the consumers (helpers 1..10)
constexpr int ENG_NC = ENG_NH - 1;         // consumer waves
constexpr int ENG_NT = 64 * (ENG_NH + 1);  // 768 threads
constexpr int ENG_HT = 64 * ENG_NH;        // 704 helper threads
constexpr int ENG_SLOT = 16384;
constexpr int ENG_MIRROR = 2048;           // bytes of the ring's start repeated behind its end (>= the largest column: 1536 B, Q5_1)
constexpr int ENG_CHUNK = 1024;            // granules per gather chunk: 16 per lane
constexpr unsigned ENG_SPIN_MAX = 1u << 17;

template <int TYPE> struct eng_act { static constexpr int value = (TYPE == FQ_Q4_1 || TYPE == FQ_Q5_1) ? FQ_Q8_1 : FQ_Q8_0; };

// ---- LDS control words: explicit DS instructions only (a flat access would make hipcc drain vmcnt, i.e. the loader's DMA)
__device__ __forceinline__ unsigned lds_ld(unsigned addr) { unsigned v; asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(addr) : "memory"); return v; }
__device__ __forceinline__ void lds_st(unsigned addr, unsigned v) { asm volatile("ds_write_b32 %0, %1" :: "v"(addr), "v"(v) : "memory"); }
__device__ __forceinline__ void lds_add(unsigned addr, unsigned v) { asm volatile("ds_add_u32 %0, %1" :: "v"(addr), "v"(v) : "memory"); }
__device__ __forceinline__ unsigned lds_ld_u(unsigned addr) { return __builtin_amdgcn_readfirstlane(lds_ld(addr)); }
__device__ __forceinline__ unsigned lds_add_rtn(unsigned addr, unsigned v) { unsigned o; asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&v"(o) : "v"(addr), "v"(v) : "memory"); return o; }
__device__ __forceinline__ unsigned long long lds_ld64(unsigned addr) {
    unsigned long long v; asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(addr) : "memory"); return v;
}
__device__ __forceinline__ void lds_st64(unsigned addr, unsigned long long v) { asm volatile("ds_write_b64 %0, %1" :: "v"(addr), "v"(v) : "memory"); }
__device__ __forceinline__ void lds_drain() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// one unit's f32 term: into the lane's partial sum (default order) or, REF (fast reference order, fq_ref_chain.h), into the row's LDS strip at the
// block's index u (sa[r] = LDS byte address of row r's strip) -- an explicit DS store: it stays in the wave's LDS queue ahead of the counter that reports the row
template <bool REF, int R>
__device__ __forceinline__ void eng_emit(float (&acc)[R], const unsigned * sa, int r, int uc, bool ok, float v) { fq_emit_term<REF, R>(acc, sa, r, uc, ok, v); }

// ---- LDS-DMA of the loader wave. Source = scalar base (64-bit) + per-lane 32-bit offset (+ immediate), destination = M0 (wave-uniform
// LDS byte address, the lanes' 16 bytes land side by side: 1 KiB per instruction). No vector ALU work per piece.
// one piece: 1 KiB at base (+ voff = lane * 16) -> LDS at lds_dst
__device__ __forceinline__ void glds_piece(const void * base, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(base), "s"(lds_dst) : "memory");
}
// sixteen pieces: 16 KiB at base -> the ring from LDS address ml on (wrapping from mend to mstart); ml is advanced. v[p] = lane * 16 +
// p KiB (an instruction offset would be added to the LDS address as well). Also samples the LDS word at flag_addr (returned; the
// read overlaps the issue).
struct eng_voff { unsigned v[16]; };
#define ENG_DMA1(P) "s_mov_b32 m0, %[ml]\n\ts_add_u32 %[ml], %[ml], 0x400\n\tglobal_load_lds_dwordx4 %[v" #P "], %[base] nt\n\t" \
                    "s_cmp_lt_u32 %[ml], %[mend]\n\ts_cselect_b32 %[ml], %[ml], %[mstart]\n\t"
__device__ __forceinline__ unsigned glds_batch16(const void * base, const eng_voff & o, unsigned & ml, unsigned mstart, unsigned mend, unsigned flag_addr) {
    unsigned keep, flag;
    asm volatile("ds_read_b32 %[flag], %[fa]\n\ts_mov_b32 %[keep], m0\n\ts_nop 4\n\t"
                 ENG_DMA1(0) ENG_DMA1(1) ENG_DMA1(2) ENG_DMA1(3) ENG_DMA1(4) ENG_DMA1(5) ENG_DMA1(6) ENG_DMA1(7)
                 ENG_DMA1(8) ENG_DMA1(9) ENG_DMA1(10) ENG_DMA1(11) ENG_DMA1(12) ENG_DMA1(13) ENG_DMA1(14) ENG_DMA1(15)
                 "s_mov_b32 m0, %[keep]\n\ts_waitcnt lgkmcnt(0)"
                 : [keep] "=&s"(keep), [ml] "+s"(ml), [flag] "=&v"(flag)
                 : [v0] "v"(o.v[0]), [v1] "v"(o.v[1]), [v2] "v"(o.v[2]), [v3] "v"(o.v[3]), [v4] "v"(o.v[4]), [v5] "v"(o.v[5]), [v6] "v"(o.v[6]), [v7] "v"(o.v[7]),
                   [v8] "v"(o.v[8]), [v9] "v"(o.v[9]), [v10] "v"(o.v[10]), [v11] "v"(o.v[11]), [v12] "v"(o.v[12]), [v13] "v"(o.v[13]), [v14] "v"(o.v[14]), [v15] "v"(o.v[15]),
                   [base] "s"(base), [mstart] "s"(mstart), [mend] "s"(mend), [fa] "v"(flag_addr)
                 : "memory", "scc");
    return flag;
}

__device__ __forceinline__ unsigned long long gran_ld(const unsigned long long * p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void gran_st(unsigned long long * p, unsigned tag, unsigned v) { __hip_atomic_store(p, ((unsigned long long) tag << 32) | v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ bool eng_failed(const unsigned * err) { return __builtin_amdgcn_readfirstlane(__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != 0u; }
__device__ __forceinline__ void eng_fail(unsigned * err, unsigned code) { __hip_atomic_store(err, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// control block (byte offsets from its base; the base is 16-byte aligned)
struct eng_ctl {
    static constexpr unsigned PSUM = 0;           // 16 doubles: per gather chunk, the f64 sum of its values (residual row)
    static constexpr unsigned STAT = 128;         // mean, scale (f32) of the current LayerNorm
    static constexpr unsigned OUT = 144;          // 384 floats: dots of up to twelve 32-row groups (phase A / lm_head)
    static constexpr unsigned CNT = OUT + 1536;   // 16 words: rows finished per group (all blocks: a group's count grows by 32 per block)
    static constexpr unsigned CNTH = CNT + 64;    // 16 words: the same for the lm_head groups
    static constexpr unsigned OUTB = CNTH + 64;   // 64 floats: down-projection dots of the workgroup's rows
    static constexpr unsigned XRES = OUTB + 256;  // 64 floats: residual values of the workgroup's rows
    static constexpr unsigned LANDED = XRES + 256;
    static constexpr unsigned LOW = LANDED + 4;   // 16 words: per consumer, stream position below which it needs nothing
    // monotonic flags / counters (targets are functions of the block index)
    static constexpr unsigned XG_DONE = LOW + 64; // residual-row chunks gathered
    static constexpr unsigned LN_STAT = XG_DONE + 4;   // = index of the LayerNorm whose {mean, scale} are in STAT, + 1
    static constexpr unsigned IMG_DONE = XG_DONE + 8;  // helper waves that wrote their share of the LayerNorm image(s)
    static constexpr unsigned A_DONE = XG_DONE + 12;   // consumers that finished their phase-A rows
    static constexpr unsigned FG_DONE = XG_DONE + 16;  // GELU-image chunks gathered
    static constexpr unsigned B1_DONE = XG_DONE + 20;  // consumers that finished their Wdown rows
    static constexpr unsigned AG_DONE = XG_DONE + 24;  // attention-image chunks gathered
    static constexpr unsigned B2_DONE = XG_DONE + 28;  // consumers that finished their Wo rows
    static constexpr unsigned THIN = XG_DONE + 32;     // != 0: the loader keeps at most 16 pieces in flight
    static constexpr unsigned LN_MEAN = XG_DONE + 36;  // = index of the LayerNorm whose mean is in STAT, + 1
    static constexpr unsigned S2_DONE = XG_DONE + 40;  // helper waves that left their partial sum of squares
    static constexpr unsigned PSQ = XG_DONE + 64;      // 16 doubles: per helper wave, the f64 sum of its squared deviations
    static constexpr unsigned PTRS = PSQ + 128;        // per block 4 x 8 bytes: this workgroup's first byte of Wqkv, Wup, Wdown, Wo; then lm_head's
    static constexpr unsigned BYTES = PTRS;            // + 32 * n_layers + 8
};

// one unit (32 weights) of a row out of the ring. cb = ring offset of the unit's column (a column never wraps: the mirror),
// nbc = blocks in the column; both wave-uniform for the formats with 64 blocks per column, so only the lane's own offset is
// vector work.
template <int TYPE>
__device__ __forceinline__ fq_unit_regs eng_unit_load_col(const uint8_t * ring, unsigned cb, int nbc, int lane) {
    constexpr fq_type_desc D = fq_desc(TYPE);
    static_assert(fq_lay<TYPE>::CB == 64, "64 blocks per column");
    fq_unit_regs r{};
    const int j = lane < nbc ? lane : nbc - 1;
    const uint8_t * c0 = (const uint8_t *) __builtin_assume_aligned(ring + cb, 16);      // rows and columns start on 16-byte boundaries
    typedef unsigned int u32x4_ld __attribute__((ext_vector_type(4)));
    const u32x4_ld q = *(const u32x4_ld *)(c0 + j * 16);                                   // one ds_read_b128
    r.q = fq_u4{ q.x, q.y, q.z, q.w };
    const uint8_t * p1 = c0 + nbc * 16 + j * D.plane[1].bytes;
    if constexpr (TYPE == FQ_Q4_0)      r.dm = *(const uint16_t *) p1;
    else if constexpr (TYPE == FQ_Q4_1) r.dm = *(const uint32_t *) p1;
    else {                                                               // Q5_0 / Q5_1: plane 1 = qh, plane 2 = d (,m)
        r.s0 = *(const uint32_t *) p1;
        const uint8_t * p2 = c0 + nbc * (16 + D.plane[1].bytes) + j * D.plane[2].bytes;
        if constexpr (TYPE == FQ_Q5_0) r.dm = *(const uint16_t *) p2; else r.dm = *(const uint32_t *) p2;
    }
    return r;
}
// Q8_0: 32 blocks of 32 + 2 bytes per column, the 64 lanes of a pass sit in two columns: the column base is per lane
template <int RING>
__device__ __forceinline__ fq_unit_regs eng_unit_load_q8(const uint8_t * ring, unsigned pos, int u0, int lane, int nblk) {
    constexpr int CB = fq_lay<FQ_Q8_0>::CB, TS = fq_lay<FQ_Q8_0>::TS;
    fq_unit_regs r{};
    const int u = u0 + lane, uc = u < nblk ? u : nblk - 1;
    const int c = uc / CB, j = uc - c * CB;
    const int rem = nblk - c * CB, nbc = rem < CB ? rem : CB;
    unsigned cb = pos + (unsigned)(c * CB * TS);
    cb = cb >= (unsigned) RING ? cb - (unsigned) RING : cb;             // a row is shorter than the ring
    r.q  = *(const fq_u4 *)(ring + cb + (unsigned)(j * 32));
    r.q2 = *(const fq_u4 *)(ring + cb + (unsigned)(j * 32 + 16));
    r.dm = *(const uint16_t *)(ring + cb + (unsigned)(nbc * 32 + j * 2));
    return r;
}
// U passes (of 64 units) of R rows: all loads first, then the dots -- per lane the units are still added in ascending order
template <int TYPE, int RING, int R, int U, bool REF = false>
__device__ __forceinline__ void eng_pass_group(const uint8_t * ring, const unsigned (&pos)[R], int nblk, int u0, const fq_actcol & col, int lane, float (&acc)[R], const unsigned * sa = nullptr) {
    fq_unit_regs regs[U][R];
#pragma unroll
    for (int p = 0; p < U; ++p) {
        if constexpr (fq_lay<TYPE>::CB == 64) {
            constexpr unsigned COLB = 64u * (unsigned) fq_lay<TYPE>::TS;
            const int c = (u0 >> 6) + p;                                   // scalar
            const int rem = nblk - 64 * c, nbc = rem < 64 ? rem : 64;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                unsigned cb = pos[r] + (unsigned) c * COLB;               // scalar; < 2 RING: a row is shorter than the ring
                cb = cb >= (unsigned) RING ? cb - (unsigned) RING : cb;
                regs[p][r] = eng_unit_load_col<TYPE>(ring, cb, nbc, lane);
            }
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r) regs[p][r] = eng_unit_load_q8<RING>(ring, pos[r], u0 + 64 * p, lane, nblk);
        }
    }
    // the activation slices of all U passes as well, before any arithmetic: one LDS round trip per pass group instead of three per unit
    fq_act32 act[U];
#pragma unroll
    for (int p = 0; p < U; ++p) { const int u = u0 + 64 * p + lane; act[p] = fq_act32_load(col, u < nblk ? u : nblk - 1); }
#pragma unroll
    for (int p = 0; p < U; ++p) {
        const bool ok = u0 + 64 * p + lane < nblk;
#pragma unroll
        for (int r = 0; r < R; ++r) { const float v = fq_unit<TYPE>::dot_x(regs[p][r], act[p]); eng_emit<REF, R>(acc, sa, r, ok ? u0 + 64 * p + lane : nblk - 1, ok, v); }
    }
}
// the same with the lane's activation slices already in registers (a lane's units are the same in every row: hoisted out of the row loop)
template <int TYPE, int RING, int R, int U, bool REF = false>
__device__ __forceinline__ void eng_pass_group_pre(const uint8_t * ring, const unsigned (&pos)[R], int nblk, int u0, const fq_act32 * act, int lane, float (&acc)[R], const unsigned * sa = nullptr) {
    fq_unit_regs regs[U][R];
#pragma unroll
    for (int p = 0; p < U; ++p) {
        if constexpr (fq_lay<TYPE>::CB == 64) {
            constexpr unsigned COLB = 64u * (unsigned) fq_lay<TYPE>::TS;
            const int c = (u0 >> 6) + p;
            const int rem = nblk - 64 * c, nbc = rem < 64 ? rem : 64;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                unsigned cb = pos[r] + (unsigned) c * COLB;
                cb = cb >= (unsigned) RING ? cb - (unsigned) RING : cb;
                regs[p][r] = eng_unit_load_col<TYPE>(ring, cb, nbc, lane);
            }
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r) regs[p][r] = eng_unit_load_q8<RING>(ring, pos[r], u0 + 64 * p, lane, nblk);
        }
    }
#pragma unroll
    for (int p = 0; p < U; ++p) {
        const bool ok = u0 + 64 * p + lane < nblk;
#pragma unroll
        for (int r = 0; r < R; ++r) { const float v = fq_unit<TYPE>::dot_x(regs[p][r], act[p]); eng_emit<REF, R>(acc, sa, r, ok ? u0 + 64 * p + lane : nblk - 1, ok, v); }
    }
}
// eng_pass_group_pre in two steps, so that the caller can hand the ring space back between them (kernels_ring.hip): the weight units of U passes of R rows
// into registers, then their dots with the lane's resident activation slices -- the same loads, the same arithmetic in the same order
template <int R, int U> struct eng_regs { fq_unit_regs r[U][R]; };
template <int TYPE, int RING, int R, int U>
__device__ __forceinline__ void eng_pass_load(const uint8_t * ring, const unsigned (&pos)[R], int nblk, int u0, int lane, eng_regs<R, U> & G) {
#pragma unroll
    for (int p = 0; p < U; ++p) {
        if constexpr (fq_lay<TYPE>::CB == 64) {
            constexpr unsigned COLB = 64u * (unsigned) fq_lay<TYPE>::TS;
            const int c = (u0 >> 6) + p;
            const int rem = nblk - 64 * c, nbc = rem < 64 ? rem : 64;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                unsigned cb = pos[r] + (unsigned) c * COLB;
                cb = cb >= (unsigned) RING ? cb - (unsigned) RING : cb;
                G.r[p][r] = eng_unit_load_col<TYPE>(ring, cb, nbc, lane);
            }
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r) G.r[p][r] = eng_unit_load_q8<RING>(ring, pos[r], u0 + 64 * p, lane, nblk);
        }
    }
}
template <int TYPE, int R, int U, bool REF = false>
__device__ __forceinline__ void eng_pass_dot_pre(const eng_regs<R, U> & G, int nblk, int u0, const fq_act32 * act, int lane, float (&acc)[R], const unsigned * sa = nullptr) {
#pragma unroll
    for (int p = 0; p < U; ++p) {
        const bool ok = u0 + 64 * p + lane < nblk;
#pragma unroll
        for (int r = 0; r < R; ++r) { const float v = fq_unit<TYPE>::dot_x(G.r[p][r], act[p]); eng_emit<REF, R>(acc, sa, r, ok ? u0 + 64 * p + lane : nblk - 1, ok, v); }
    }
}
struct eng_wait {                 // per-wave state of the bounded waits
    unsigned * err; bool dead; long long * dbg; int blk;
    __device__ __forceinline__ bool spin(unsigned & spins, unsigned code, unsigned x0 = 0, unsigned x1 = 0) {      // true = keep waiting
        if (dead) return false;
        ++spins;
        if ((spins & 255u) == 0u && eng_failed(err)) { dead = true; return false; }
        if (spins > ENG_SPIN_MAX) {
            if ((threadIdx.x & 63) == 0) {
                eng_fail(err, code);
                if (dbg) {                                                 // failure record: who gave up, where, on what
                    const unsigned long long k = atomicAdd((unsigned long long *) dbg, 1ull);
                    if (k < 500) { long long * r = dbg + 16 + 8 * k; r[0] = code; r[1] = blockIdx.x; r[2] = threadIdx.x >> 6; r[3] = blk; r[4] = x0; r[5] = x1; }
                }
            }
            dead = true; return false;
        }
        return true;
    }
    // wait until the LDS word at `addr` has reached `target` (monotonic counters and flags)
    __device__ __forceinline__ void until(unsigned addr, unsigned target, unsigned code) {
        for (unsigned spins = 0; (int)(lds_ld_u(addr) - target) < 0;) { if (!spin(spins, code, addr, target)) break; __builtin_amdgcn_s_sleep(1); }
    }
};

// failure / wait codes
enum { ENG_W_RING = 1, ENG_W_LAND = 2, ENG_W_QKV = 4, ENG_W_XG = 5, ENG_W_GROUP = 6, ENG_W_FG = 7, ENG_W_AG = 8,
       ENG_W_STAT = 12, ENG_W_IMG = 13, ENG_W_ADONE = 14, ENG_W_B1 = 15, ENG_W_B2 = 16, ENG_W_FGD = 17, ENG_W_AGD = 18 };

}   // namespace
