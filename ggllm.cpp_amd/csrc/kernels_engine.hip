// kernels_engine.hip -- the PERSISTENT decode engine: one launch per generated token (gfx950, wave64).
//
// Replaces the per-token launch list of the fused decode path (2 launches per block + lm_head: k_gemv_ln / k_attn_out,
// kernels_decode.hip) -- i.e. one N = 1 pass of falcon_eval_internal's graph (libfalcon.cpp:2115-2466) whose mat-muls are
// ggml_compute_forward_mul_mat_q_f32 (ggml.c:11318-11529) -- by ONE kernel of one workgroup per CU (12 waves) that stays
// resident for all blocks of the stage. What a launch boundary costs this workload is not the ~1.7 us gap itself but the
// empty memory pipeline on either side of it (4.8 us per launch in which nothing streams, 65 launches
// per token); here the weight stream never stops:
//
//   streaming workgroups   wave 0 = LOADER: streams the workgroup's byte ranges of [Wqkv | Wup | Wdown | Wo] of every block
//                          (then lm_head), in the order they will be used, into a RING of LDS with global_load_lds_dwordx4
//                          (1 KiB per wave-instruction, non-temporal, no registers held), limited only by ring space: it
//                          runs up to ~128 KiB ahead of the arithmetic and keeps streaming across every dependency stall.
//                          waves 1..11 = CONSUMERS: take the landed rows round-robin, run the row dots out of LDS (the
//                          fq_units.h arithmetic, lane l = units l, l+64, ..: bit-identical to the other paths), LayerNorm
//                          + Q8 images, GELU + Q8 epilogues, residual update. They synchronise among themselves with
//                          counters in LDS (the loader never reaches a barrier, so s_barrier cannot be used).
//   attention workgroups   2 query heads each (attn_decode_group, the code of k_attn_decode / k_attn_out): RoPE, KV append,
//                          K.Q, soft_max, V.P, Q8 image of the output.
//   between workgroups     8-byte {tag, value} granules written with single agent-scope stores and swept by the readers
//                          (the hand-off k_attn_out already uses; MI355X_MICROARCH.md "handoff" / "allgather" rows):
//                          residual row x (4 hops per block: x -> LN -> [qkv, up] -> attention / GELU image -> down, wo -> x).
//
// scripts/microbench/mb_engine.hip measures the engine's core alone (no dependencies): 6.9 TB/s with 1 loader + 11 consumer
// waves per CU against 5.4 TB/s inside a k_gemv_ln launch.
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

#ifndef ENG_OPT_NLN2
#define ENG_OPT_NLN2 0
#endif
#ifndef ENG_OPT_PREFETCH_ATT
#define ENG_OPT_PREFETCH_ATT 0
#endif

namespace {

constexpr int ENG_NC = 11;                 // consumer waves
constexpr int ENG_NT = 64 * (ENG_NC + 1);  // 768 threads
constexpr int ENG_SLOT = 16384;
constexpr unsigned ENG_SPIN_MAX = 1u << 17;

template <int TYPE> struct eng_act { static constexpr int value = (TYPE == FQ_Q4_1 || TYPE == FQ_Q5_1) ? FQ_Q8_1 : FQ_Q8_0; };

// ---- LDS control words: explicit DS instructions only (a flat access would make hipcc drain vmcnt, i.e. the loader's DMA)
__device__ __forceinline__ unsigned lds_ld(unsigned addr) { unsigned v; asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(addr) : "memory"); return v; }
__device__ __forceinline__ void lds_st(unsigned addr, unsigned v) { asm volatile("ds_write_b32 %0, %1" :: "v"(addr), "v"(v) : "memory"); }
__device__ __forceinline__ void lds_add(unsigned addr, unsigned v) { asm volatile("ds_add_u32 %0, %1" :: "v"(addr), "v"(v) : "memory"); }
__device__ __forceinline__ unsigned lds_ld_u(unsigned addr) { return __builtin_amdgcn_readfirstlane(lds_ld(addr)); }
__device__ __forceinline__ unsigned lds_add_rtn(unsigned addr, unsigned v) { unsigned o; asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&v"(o) : "v"(addr), "v"(v) : "memory"); return o; }
// arrival counters of the four hand-offs (device words 128 bytes apart, zeroed before every launch). They only say WHEN a sweep
// is worth starting -- thousands of waves re-reading 40 KB of granules while they wait take a large share of the L2 bandwidth the
// weight stream needs; the tagged granules remain what makes a hand-off correct.
enum { ENG_CNT_X = 0, ENG_CNT_QKV = 32, ENG_CNT_FF = 64, ENG_CNT_ATT = 96 };
__device__ __forceinline__ void cnt_add(unsigned * cnt, unsigned v) { __hip_atomic_fetch_add(cnt, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned cnt_ld(const unsigned * cnt) { return __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ void glds16_nt(const void * gsrc, unsigned lds_dst) {       // 64 lanes x 16 B -> 1 KiB of LDS at lds_dst
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

__device__ __forceinline__ unsigned long long gran_ld(const unsigned long long * p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void gran_st(unsigned long long * p, unsigned tag, unsigned v) { __hip_atomic_store(p, ((unsigned long long) tag << 32) | v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ bool eng_failed(const unsigned * err) { return __builtin_amdgcn_readfirstlane(__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != 0u; }
__device__ __forceinline__ void eng_fail(unsigned * err, unsigned code) { __hip_atomic_store(err, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// control block (byte offsets from its base; the base is 16-byte aligned)
struct eng_ctl {
    static constexpr unsigned RED = 0;            // 32 doubles: LayerNorm partial sums
    static constexpr unsigned OUT = 256;          // 384 floats: dots of up to twelve 32-row groups (phase A / lm_head)
    static constexpr unsigned CNT = OUT + 1536;   // 16 words: rows finished per group (all blocks: a group's count grows by 32 per block)
    static constexpr unsigned CNTH = CNT + 64;    // 16 words: the same for the lm_head groups
    static constexpr unsigned OUTB = CNTH + 64;   // 64 floats: down-projection dots of the workgroup's rows
    static constexpr unsigned XRES = OUTB + 256;  // 64 floats: residual values of the workgroup's rows
    static constexpr unsigned LANDED = XRES + 256;
    static constexpr unsigned LOW = LANDED + 4;   // ENG_NC words: per consumer, stream position below which it needs nothing
    static constexpr unsigned CBAR = LOW + 4 * 16;
    static constexpr unsigned EPI = CBAR + 4;     // Wup epilogues finished (all blocks)
    static constexpr unsigned BDONE = CBAR + 8;   // consumers that finished their Wo rows (all blocks)
    static constexpr unsigned QDONE = CBAR + 12;  // consumers that finished their Wqkv rows (all blocks)
    static constexpr unsigned GO = CBAR + 16;     // 4 words: the last target each hand-off counter was seen to reach
    static constexpr unsigned PTRS = GO + 16;     // per block 4 x 8 bytes: this workgroup's first byte of Wqkv, Wup, Wdown, Wo; then lm_head's
    static constexpr unsigned BYTES = PTRS;       // + 32 * n_layers + 8
};
__device__ __forceinline__ unsigned long long lds_ld64(unsigned addr) {
    unsigned long long v; asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(addr) : "memory"); return v;
}
__device__ __forceinline__ void lds_st64(unsigned addr, unsigned long long v) { asm volatile("ds_write_b64 %0, %1" :: "v"(addr), "v"(v) : "memory"); }

// one unit (32 weights) of a row out of the ring. u0 = first unit of the pass (a multiple of 64), the lane takes unit u0 + lane
// clamped to the row's last. For the formats with 64 blocks per column the column -- its ring offset, its block count -- is
// wave-uniform: scalar arithmetic, only the lane's own offset and the wrap are vector work.
template <int TYPE, int RING>
__device__ __forceinline__ fq_unit_regs eng_unit_load(const uint8_t * ring, unsigned pos, int u0, int lane, int nblk) {
    constexpr int CB = fq_lay<TYPE>::CB, TS = fq_lay<TYPE>::TS;
    constexpr fq_type_desc D = fq_desc(TYPE);
    auto wrap = [](unsigned o) { const unsigned m = o - (unsigned) RING; return m < o ? m : o; };      // o in [0, 2 RING): one v_min_u32 after the subtract
    fq_unit_regs r{};
    if constexpr (CB == 64) {
        const int c = u0 >> 6;                                             // scalar
        const int rem = nblk - 64 * c, nbc = rem < 64 ? rem : 64;
        const int j = lane < nbc ? lane : nbc - 1;
        const unsigned cb = pos + (unsigned)(c * 64 * TS);                 // < 2 RING: a row is shorter than the ring
        r.q = *(const fq_u4 *)(ring + wrap(cb + (unsigned)(j * 16)));
        const unsigned p1 = wrap(cb + (unsigned)(nbc * 16) + (unsigned)(j * D.plane[1].bytes));
        if constexpr (TYPE == FQ_Q4_0)      r.dm = *(const uint16_t *)(ring + p1);
        else if constexpr (TYPE == FQ_Q4_1) r.dm = *(const uint32_t *)(ring + p1);
        else {                                                               // Q5_0 / Q5_1: plane 1 = qh, plane 2 = d (,m)
            r.s0 = *(const uint32_t *)(ring + p1);
            const unsigned p2 = wrap(cb + (unsigned)(nbc * (16 + D.plane[1].bytes)) + (unsigned)(j * D.plane[2].bytes));
            if constexpr (TYPE == FQ_Q5_0) r.dm = *(const uint16_t *)(ring + p2); else r.dm = *(const uint32_t *)(ring + p2);
        }
    } else {                                                                 // Q8_0: 32 blocks of 32 + 2 bytes per column
        const int u = u0 + lane, uc = u < nblk ? u : nblk - 1;
        const int c = uc / CB, j = uc - c * CB;
        const int rem = nblk - c * CB, nbc = rem < CB ? rem : CB;
        const unsigned cb = pos + (unsigned)(c * CB * TS);
        r.q  = *(const fq_u4 *)(ring + wrap(cb + (unsigned)(j * 32)));
        r.q2 = *(const fq_u4 *)(ring + wrap(cb + (unsigned)(j * 32 + 16)));
        r.dm = *(const uint16_t *)(ring + wrap(cb + (unsigned)(nbc * 32 + j * 2)));
    }
    return r;
}
// U passes (of 64 units) of R rows: all loads first, then the dots -- per lane the units are still added in ascending order
template <int TYPE, int RING, int R, int U>
__device__ __forceinline__ void eng_pass_group(const uint8_t * ring, const unsigned (&pos)[R], int nblk, int u0, const fq_actcol & col, int lane, float (&acc)[R]) {
    fq_unit_regs regs[U][R];
#pragma unroll
    for (int p = 0; p < U; ++p) {
#pragma unroll
        for (int r = 0; r < R; ++r) regs[p][r] = eng_unit_load<TYPE, RING>(ring, pos[r], u0 + 64 * p, lane, nblk);
    }
#pragma unroll
    for (int p = 0; p < U; ++p) {
        const int u = u0 + 64 * p + lane; const bool ok = u < nblk; const int uc = ok ? u : nblk - 1;
#pragma unroll
        for (int r = 0; r < R; ++r) { const float v = fq_unit<TYPE>::dot(regs[p][r], col, uc); acc[r] += ok ? v : 0.0f; }
    }
}
// the dots of R rows out of the ring. pos[r] = the row's stream position reduced modulo RING (wave-uniform)
template <int TYPE, int RING, int R>
__device__ __forceinline__ void eng_rows_dot(const uint8_t * ring, const unsigned (&pos)[R], int nblk, const fq_actcol & col, int lane, float (&out)[R]) {
    float acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0.0f;
    const int npass = (nblk + 63) >> 6;
    int p0 = 0;
    for (; p0 + 3 <= npass; p0 += 3) eng_pass_group<TYPE, RING, R, 3>(ring, pos, nblk, 64 * p0, col, lane, acc);
    if (npass - p0 == 2)      eng_pass_group<TYPE, RING, R, 2>(ring, pos, nblk, 64 * p0, col, lane, acc);
    else if (npass - p0 == 1) eng_pass_group<TYPE, RING, R, 1>(ring, pos, nblk, 64 * p0, col, lane, acc);
#pragma unroll
    for (int r = 0; r < R; ++r) out[r] = wave_sum(acc[r]);
}

struct eng_wait {                 // per-wave state of the bounded waits
    unsigned * err; bool dead; long long * dbg; int blk;
    __device__ __forceinline__ bool spin(unsigned & spins, unsigned code, unsigned x0 = 0, unsigned x1 = 0) {      // true = keep waiting
        if (dead) return false;
        ++spins;
        if ((spins & 255u) == 0u && eng_failed(err)) { dead = true; return false; }
        // (the consumer barrier gets 16 x the patience of the waits it may be waiting behind, so that the culprit reports first)
        if (spins > (code == 3u ? 16u * ENG_SPIN_MAX : ENG_SPIN_MAX)) {
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
};
#define ENG_STAMP(slot) do { if (a.dbg && c == 0 && lane == 0 && (b < 3 || b == a.n_layers - 1)) \
        a.dbg[4096 + ((size_t) blockIdx.x * 4 + (b < 3 ? b : 3)) * 8 + (slot)] = (long long) wall_clock64(); } while (0)

// barrier among the consumer waves (generation counter in LDS)
__device__ __forceinline__ void eng_cbar(unsigned ctl, unsigned & gen, eng_wait & w) {
    ++gen;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                      // this wave's LDS stores are done before it arrives
    if ((threadIdx.x & 63) == 0) lds_add(ctl + eng_ctl::CBAR, 1u);
    const unsigned target = gen * (unsigned) ENG_NC;
    for (unsigned spins = 0; lds_ld_u(ctl + eng_ctl::CBAR) < target;) { if (!w.spin(spins, 3u)) break; __builtin_amdgcn_s_sleep(1); }
}

// the consumers' share of a granule buffer: words [0, nwords) -> LDS (dst) once every granule carries `tag`. All NG loads of a
// thread are in flight together: an attempt costs ONE memory round trip (~2 us while the chip streams).
template <int NG>
__device__ __forceinline__ void eng_sweep(const unsigned long long * gran, unsigned tag, int nwords, unsigned * dst, int ctid, eng_wait & w, unsigned code) {
    constexpr int CT = 64 * ENG_NC;
    for (int base = 0; base < nwords; base += NG * CT) {
        unsigned v[NG];
        for (unsigned spins = 0;;) {
            unsigned long long x[NG];
#pragma unroll
            for (int k = 0; k < NG; ++k) { const int i = base + k * CT + ctid; x[k] = gran_ld(gran + (i < nwords ? i : nwords - 1)); }
            bool ok = true;
#pragma unroll
            for (int k = 0; k < NG; ++k) { v[k] = (unsigned) x[k]; ok = ok && (unsigned)(x[k] >> 32) == tag; }
            if (__all(ok)) break;
            if (!w.spin(spins, code, (unsigned) base, tag)) break;
            __builtin_amdgcn_s_sleep(4);
        }
#pragma unroll
        for (int k = 0; k < NG; ++k) { const int i = base + k * CT + ctid; if (i < nwords) dst[i] = v[k]; }
    }
}

// LDS of a streaming workgroup: [ring][R][ATT][control block + pointer table]. R holds the LayerNorm image(s) during phase A and
// the GELU image during phase B1 (a consumer barrier separates the two uses), ATT the attention output image.
__host__ __device__ inline size_t eng_region_r(int act, int64_t E, int64_t FF) {
    const size_t ff = fq_act_col_bytes(act, FF), e2 = 2 * fq_act_col_bytes(act, E);
    return ff > e2 ? ff : e2;
}

}   // namespace

// =============================================================================================== the kernel
template <int TYPE, int NSLOT>
__global__ void __launch_bounds__(ENG_NT) k_decode_engine(fq_engine_args a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr int ACT = eng_act<TYPE>::value;
    constexpr int RING = NSLOT * ENG_SLOT;
    constexpr int TS = fq_desc(TYPE).tsize;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const unsigned epoch0 = *a.epoch_word;
    const int E = a.E, FF = a.FF;

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
#define ENG_ASTAMP(slot) do { if (a.dbg && tid == 0 && (b < 3 || b == a.n_layers - 1)) a.dbg[4096 + ((size_t) blockIdx.x * 4 + (b < 3 ? b : 3)) * 8 + (slot)] = (long long) wall_clock64(); } while (0)
            ENG_ASTAMP(0);
            if (!idle && gtid < 192) {
                for (unsigned spins = 0; a.use_counters && cnt_ld(a.cnt + ENG_CNT_QKV) < (unsigned)((a.H + 2 * a.HKV) * 64) * (unsigned)(b + 1);) { if (!w.spin(spins, 9u)) break; __builtin_amdgcn_s_sleep(8); }
                const int part = gtid >> 6, d = gtid & 63;
                const int row = (part == 0 ? h : (part == 1 ? a.H + hk : a.H + a.HKV + hk)) * 64 + d;
                unsigned long long x = 0;
                for (unsigned spins = 0;;) {
                    x = gran_ld(a.qkvg + row);
                    if (__all((unsigned)(x >> 32) == tag)) break;
                    if (!w.spin(spins, 4u, (unsigned) row, (unsigned)(x >> 32))) break;
                    __builtin_amdgcn_s_sleep(2);
                }
                stage[gtid] = __builtin_bit_cast(float, (unsigned) x);
            }
            __syncthreads();
            ENG_ASTAMP(1);
            if (idle) { attn_decode_group_idle(); continue; }
            const fq_engine_layer & L = a.layers[b];
            // (qkv is not used when q / k / v sources are given; it must be a GLOBAL pointer all the same -- with an LDS-derived or
            //  null one hipcc 7.2's InstCombine crashes on the selects between the two sources)
            fq_attn_decode_args at{ a.rope_cs, a.H, a.HKV, a.n_past, a.rope_cs, L.kc, L.vc, a.exp_tab, nullptr, (uint8_t *) a.attg, ACT, a.max_n_kv, nullptr,
                                    stage, stage + 64, stage + 128 };
            // (a per-iteration opaque copy of the thread / head index: without it hipcc hoists ~50 registers of per-thread address
            //  arithmetic out of the block loop and the attention spills)
            int gt2 = gtid, h2 = h;
            asm volatile("" : "+v"(gt2), "+v"(h2));
            h2 = __builtin_amdgcn_readfirstlane(h2);
            attn_decode_group<true>(at, h2, live, gt2, gbase + 768, nullptr, fq_publish{ a.attg, tag });
            ENG_ASTAMP(2);
            if (gtid < 64) {                                               // (the group's first wave stored the image)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (gtid == 0 && live && a.use_counters) cnt_add(a.cnt + ENG_CNT_ATT, 1u);
            }
        }
        return;
    }

    // ==================================================================================== streaming workgroup
    const int sw = (int) blockIdx.x - a.n_attn;
    const fq_engine_sched sc = a.sched[sw];
    uint8_t * ring    = smem;
    uint8_t * img_ff  = smem + RING;                                       // region R: GELU image (phase B1) ...
    uint8_t * img_e   = img_ff;                                            // ... LN image feeding Wup (and Wqkv with one norm) (phase A)
    uint8_t * img_e2  = img_e + fq_act_col_bytes(ACT, E);                  // ... attention-norm image of a two-norm block (phase A)
    uint8_t * img_att = img_ff + eng_region_r(ACT, E, FF);                 // attention output image (phase B2)
    uint8_t * ctlp = img_att + fq_act_col_bytes(ACT, E);
    const unsigned ctl = (unsigned)(uintptr_t) ctlp;
    double * red  = (double *)(ctlp + eng_ctl::RED);
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
    if (tid == 0) { lds_st(ctl + eng_ctl::LANDED, 0u); lds_st(ctl + eng_ctl::CBAR, 0u); lds_st(ctl + eng_ctl::EPI, 0u); lds_st(ctl + eng_ctl::BDONE, 0u); lds_st(ctl + eng_ctl::QDONE, 0u); }
    if (tid < 4) lds_st(ctl + eng_ctl::GO + 4 * tid, 0u);
    __syncthreads();                                                       // the only workgroup barrier: before the roles split

    if (wid == 0) {
        // ================================================================================ loader
        const unsigned ring_lds = (unsigned)(uintptr_t) ring;
        eng_wait w{ a.err, false, a.dbg, 0 };
        // pos = bytes issued, reported = bytes known to have landed (published in LANDED), freed = cached low-water mark; rp = ring
        // piece index. ALL of it is wave-uniform and kept scalar on purpose: one wave executes this loop, a dependent instruction
        // costs it ~8 cycles, and a per-piece loop of ~45 vector instructions was measured at 350 cycles per 1 KiB piece (6.6 GB/s per
        // CU); pieces are therefore issued 16 at a time, unrolled, with scalar addressing (~90 cycles per piece as in mb_engine.hip).
        // At most 48 pieces are in flight; s_waitcnt vmcnt(N) = "all but my N newest pieces have landed".
        constexpr unsigned NP = (unsigned)(NSLOT * 16);
        unsigned pos = 0, rp = 0, freed = 0, reported = 0;
        unsigned long long t_blocked = 0, t_report = 0, n_blocked = 0, t_start = __builtin_amdgcn_s_memtime();
        auto report = [&](unsigned upto) { if (upto > reported) { reported = upto; if (lane == 0) lds_st(ctl + eng_ctl::LANDED, upto); } };
        // make room for `bytes` (<= one slot) more: while the ring is full let the pieces in flight land step by step and report them
        auto wait_space = [&](unsigned bytes) {
            if (pos + bytes - freed <= (unsigned) RING) return;
            const unsigned long long tb0 = __builtin_amdgcn_s_memtime(); ++n_blocked;
            for (unsigned spins = 0;;) {
                const unsigned inflight = pos - reported;
                if (inflight > 32u * 1024u)      { asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); report(pos - 32u * 1024u); }
                else if (inflight > 16u * 1024u) { asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); report(pos - 16u * 1024u); }
                else if (inflight > 0u)          { asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  report(pos); }
                unsigned v = lds_ld(ctl + eng_ctl::LOW + 4 * (lane < 16 ? lane : 0));
                v = (unsigned) wave_reduce((int) v, [](int x, int y) { return (unsigned) x < (unsigned) y ? x : y; });
                freed = __builtin_amdgcn_readfirstlane(v);
                if (pos + (unsigned) ENG_SLOT - freed <= (unsigned) RING) break;      // resume with a whole slot of space
                if (!w.spin(spins, 1u, pos, freed)) break;
                __builtin_amdgcn_s_sleep(1);
            }
            freed = __builtin_amdgcn_readfirstlane(freed);
            t_blocked += __builtin_amdgcn_s_memtime() - tb0;
        };
        auto after_issue = [&]() {
            if (pos - reported >= 48u * 1024u) {
                const unsigned long long tr0 = __builtin_amdgcn_s_memtime();
                asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); report(pos - 32u * 1024u);
                t_report += __builtin_amdgcn_s_memtime() - tr0;
            }
        };
        auto seg = [&](const uint8_t * src, unsigned padded) {
            const uint8_t * sl = src + lane * 16;                          // this lane's 16 bytes of every piece
            const unsigned nfull = padded >> 14, ntail = (padded >> 10) & 15u;
            for (unsigned k = 0; k < nfull; ++k) {
                wait_space((unsigned) ENG_SLOT);
                if (w.dead) return;
#pragma unroll
                for (unsigned p = 0; p < 16; ++p) {
                    unsigned q = rp + p; q = q >= NP ? q - NP : q;
                    glds16_nt(sl + p * 1024u, __builtin_amdgcn_readfirstlane(ring_lds + q * 1024u));
                }
                sl += ENG_SLOT; pos += (unsigned) ENG_SLOT;
                rp += 16u; rp = rp >= NP ? rp - NP : rp;
                after_issue();
            }
            for (unsigned p = 0; p < ntail; ++p) {
                wait_space(1024u);
                if (w.dead) return;
                glds16_nt(sl, __builtin_amdgcn_readfirstlane(ring_lds + rp * 1024u));
                sl += 1024; pos += 1024u;
                if (++rp == NP) rp = 0;
                after_issue();
            }
        };
        auto src = [&](int i) { return (const uint8_t *)(uintptr_t) lds_ld64(ctl + eng_ctl::PTRS + 8 * (unsigned) i); };
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
            long long * r = a.dbg + 4096 + 256 * 4 * 8 + (size_t) blockIdx.x * 8;
            r[0] = (long long)(__builtin_amdgcn_s_memtime() - t_start); r[1] = (long long) t_blocked; r[2] = (long long) t_report; r[3] = (long long) n_blocked; r[4] = pos;
        }
        return;
    }

    // ==================================================================================== consumers
    const int c = wid - 1, ctid = tid - 64;
    constexpr int CT = 64 * ENG_NC;
    constexpr int NLN = 3;                                                 // float4 of the row per consumer thread: n_embd <= 8448
    eng_wait w{ a.err, false, a.dbg, 0 };
    unsigned long long t_wait_land = 0, t_dot = 0, n_rows = 0;
    unsigned gen = 0;                                                      // consumer barrier generation
    const int nblkE = E / 32, nblkF = FF / 32;
    const int64_t nv = E >> 2;

    // rows [0, nrows) of a segment that starts at stream position seg_pos, dealt in runs of R consecutive rows: mine are
    // [R (c + NC k), R (c + NC k) + R), k = 0, 1, .. -- R rows share every activation read, and a wave only ever waits for ONE
    // contiguous run (a wave can always get its run: the ring is longer than a run plus the loader's restart slot)
    auto rows = [&](auto rtag, unsigned seg_pos, unsigned padded, int nrows, unsigned rs, int nblk, const fq_actcol & col, auto && sink) {
        constexpr int R = decltype(rtag)::value;
        const unsigned row_bytes = (unsigned)(nblk * TS);
        // nothing below my first run is needed by me (without this the loader would wait for rows of OTHER consumers to be
        // released by a wave that is itself waiting for its first rows to land)
        if (lane == 0) lds_st(ctl + eng_ctl::LOW + 4 * c, R * c < nrows ? seg_pos + (unsigned)(R * c) * rs : seg_pos + padded);
        for (int i = R * c; i < nrows; i += R * ENG_NC) {
            const int last = i + R - 1 < nrows ? i + R - 1 : nrows - 1;
            const unsigned need = seg_pos + (unsigned) last * rs + row_bytes;
            const unsigned long long tw0 = __builtin_amdgcn_s_memtime();
            for (unsigned spins = 0; lds_ld_u(ctl + eng_ctl::LANDED) < need;) { if (!w.spin(spins, 2u, need, lds_ld_u(ctl + eng_ctl::LANDED))) break; __builtin_amdgcn_s_sleep(1); }
            const unsigned long long tw1 = __builtin_amdgcn_s_memtime();
            unsigned pr[R]; float v[R];
            const unsigned p0 = (seg_pos + (unsigned) i * rs) % (unsigned) RING;
#pragma unroll
            for (int r = 0; r < R; ++r) { const unsigned q = p0 + (unsigned)(i + r <= last ? r : last - i) * rs; pr[r] = q >= (unsigned) RING ? q - (unsigned) RING : q; }
            eng_rows_dot<TYPE, RING, R>(ring, pr, nblk, col, lane, v);
            t_wait_land += tw1 - tw0; t_dot += __builtin_amdgcn_s_memtime() - tw1; n_rows += R;
#pragma unroll
            for (int r = 0; r < R; ++r) if (i + r <= last) sink(i + r, v[r]);
            const int nx = i + R * ENG_NC;
            if (lane == 0) lds_st(ctl + eng_ctl::LOW + 4 * c, nx < nrows ? seg_pos + (unsigned) nx * rs : seg_pos + padded);
        }
    };
    // wait until hand-off counter e has reached target: ONE consumer wave polls the device word, the others the word it leaves in LDS
    auto edge_wait = [&](int e, unsigned target) {
        if (!a.use_counters) return;
        if (c == 0) {
            for (unsigned spins = 0; cnt_ld(a.cnt + e) < target;) { if (!w.spin(spins, 10u, (unsigned) e, target)) break; __builtin_amdgcn_s_sleep(8); }
            if (lane == 0) lds_st(ctl + eng_ctl::GO + (unsigned) e / 8u, target);
        } else {
            for (unsigned spins = 0; lds_ld_u(ctl + eng_ctl::GO + (unsigned) e / 8u) < target;) { if (!w.spin(spins, 11u, (unsigned) e, target)) break; __builtin_amdgcn_s_sleep(2); }
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
    // LayerNorm (+ the second norm of a two-norm block) of the residual row -> Q8 images; the row comes from memory (first
    // block of the stage) or from the granules the previous block's owners published. Also keeps x[r0 .. r1) for phase B.
    auto layer_norm_images = [&](bool from_mem, unsigned tag, unsigned x_target, const float * w1, const float * b1, uint8_t * img1,
                                 const float * w2, const float * b2, uint8_t * img2) {
        ln_row_regs<NLN> xr, wr, br;
        ln_regs_issue_wb(w1, b1, E, CT, wr, br, ctid);
        if (!from_mem) edge_wait(ENG_CNT_X, x_target);
        if (from_mem) {
#pragma unroll
            for (int k = 0; k < NLN; ++k) { const int64_t i = (int64_t) k * CT + ctid, j = i < nv ? i : nv - 1; xr.t[k] = ((const float4 *) a.x_in)[j]; }
        } else {
            for (unsigned spins = 0;;) {                                   // all 4 NLN granules of the thread in flight together
                unsigned long long g[NLN][4];
#pragma unroll
                for (int k = 0; k < NLN; ++k) {
                    const int64_t i = (int64_t) k * CT + ctid, j = i < nv ? i : nv - 1;
#pragma unroll
                    for (int q = 0; q < 4; ++q) g[k][q] = gran_ld(a.xg + 4 * j + q);
                }
                bool ok = true;
#pragma unroll
                for (int k = 0; k < NLN; ++k) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) ok = ok && (unsigned)(g[k][q] >> 32) == tag;
                    xr.t[k] = make_float4(__builtin_bit_cast(float, (unsigned) g[k][0]), __builtin_bit_cast(float, (unsigned) g[k][1]),
                                          __builtin_bit_cast(float, (unsigned) g[k][2]), __builtin_bit_cast(float, (unsigned) g[k][3]));
                }
                if (__all(ok)) break;
                if (!w.spin(spins, 5u, 0u, tag)) break;
                __builtin_amdgcn_s_sleep(4);
            }
        }
#pragma unroll
        for (int k = 0; k < NLN; ++k) {                                    // residual values of this workgroup's phase-B rows
            const int64_t i = (int64_t) k * CT + ctid;
            if (i < nv) {
                const int e0 = (int)(4 * i);
                const float t4[4] = { xr.t[k].x, xr.t[k].y, xr.t[k].z, xr.t[k].w };
#pragma unroll
                for (int q = 0; q < 4; ++q) { const int e = e0 + q; if (e >= sc.r0 && e < sc.r1) ldsf_st(eng_ctl::XRES + 4 * (e - sc.r0), t4[q]); }
            }
        }
        ln_regs_stage1(xr, E, CT, red, ctid);
        eng_cbar(ctl, gen, w);
        ln_regs_stage2(xr, E, CT, red, ctid);
        eng_cbar(ctl, gen, w);
        ln_regs_stage3<ACT>(xr, wr, br, E, CT, act_image_at(img1, ACT, E), red, ctid);
        if (w2) {
            ln_row_regs<NLN> w2r, b2r;
            ln_regs_issue_wb(w2, b2, E, CT, w2r, b2r, ctid);
            ln_regs_stage3<ACT>(xr, w2r, b2r, E, CT, act_image_at(img2, ACT, E), red, ctid);
        }
        eng_cbar(ctl, gen, w);
    };

    // the 32 rows of a finished group, by one wave (lanes 32..63 mirror 0..31): waits until all of them are in outA
    auto group_ready = [&](unsigned cnt, int gl, unsigned target) {
        for (unsigned spins = 0; lds_ld_u(ctl + cnt + 4 * gl) < target;) { if (!w.spin(spins, 6u, (unsigned) gl, lds_ld_u(ctl + cnt + 4 * gl))) break; __builtin_amdgcn_s_sleep(1); }
    };
    const fq_actcol col_e  = { (const int8_t *) img_e,  (const float *)(img_e + fq_act_d_off(ACT, E)),  (const void *)(img_e + fq_act_aux_off(ACT, E)) };
    const fq_actcol col_e2 = { (const int8_t *) img_e2, (const float *)(img_e2 + fq_act_d_off(ACT, E)), (const void *)(img_e2 + fq_act_aux_off(ACT, E)) };
    const fq_actcol col_att = { (const int8_t *) img_att, (const float *)(img_att + fq_act_d_off(ACT, E)), (const void *)(img_att + fq_act_aux_off(ACT, E)) };
    const fq_actcol col_ff = { (const int8_t *) img_ff, (const float *)(img_ff + fq_act_d_off(ACT, FF)), (const void *)(img_ff + fq_act_aux_off(ACT, FF)) };
    const int nwords_ff = (FF >> 2) + 2 * (FF >> 5), nwords_e = (E >> 2) + 2 * (E >> 5);

    if (a.debug_mode == 1) {                                               // tuning aid: the loader alone -- rows are released as soon as they land
        auto drain = [&](unsigned seg_pos, unsigned padded, int nrows, unsigned rs, int nblk) {
            const unsigned row_bytes = (unsigned)(nblk * TS);
            for (int i = c; i < nrows; i += ENG_NC) {
                const unsigned need = seg_pos + (unsigned) i * rs + row_bytes;
                for (unsigned spins = 0; lds_ld_u(ctl + eng_ctl::LANDED) < need;) { if (!w.spin(spins, 2u, need, 0)) break; __builtin_amdgcn_s_sleep(1); }
                const int nx = i + ENG_NC;
                if (lane == 0) lds_st(ctl + eng_ctl::LOW + 4 * c, nx < nrows ? seg_pos + (unsigned) nx * rs : seg_pos + padded);
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
    for (int b = 0; b < a.n_layers; ++b) {
        const fq_engine_layer & L = a.layers[b];
        const unsigned base = (unsigned) b * p_blk;
        const unsigned tag = epoch0 + (unsigned) b + 1u;                   // tag of everything block b publishes
        w.blk = b;
        ENG_STAMP(0);
        // ---- residual row -> LayerNorm image(s)
        layer_norm_images(b == 0, epoch0 + (unsigned) b, (unsigned) E * (unsigned) b, L.ln_w, L.ln_b, img_e, a.two_norms ? L.ln2_w : nullptr, L.ln2_b, img_e2);
        ENG_STAMP(1);
        // ---- phase A: rows of Wqkv (norm image: the attention norm when the block has two) and Wup
        const unsigned cnt_target = 32u * (unsigned)(b + 1);
        const int gA = nA2 / 32;
        auto epilogue = [&](int gl) {                                      // a finished Wup group: GELU, Q8 block of 32 (kernels_decode.hip, k_gemv_ln epilogue)
            group_ready(eng_ctl::CNT, gl, cnt_target);
            const int j = lane & 31;
            float v = ldsf_ld(eng_ctl::OUT + 4 * (32 * gl + j));
            const int g = sc.ug0 + gl;                                     // block index in the FF-long image
            v = h2f_bits(a.gelu_tab[f2h_bits(v)]);                         // ggml.c:3477-3484
            const float amax = reduce32(fabsf(v), op_max());
            const float d  = amax / 127.0f;
            const float id = d ? 1.0f / d : 0.0f;
            const int q = round_half_away(v * id);
            const int s = reduce32(q, op_add());
            unsigned wq = (unsigned) q & 0xFFu;                            // 4 lanes -> one word of qs
            wq |= ((unsigned) __shfl_down((int) wq, 1) & 0xFFu) << 8;
            wq |= ((unsigned) __shfl_down((int) wq, 2) & 0xFFFFu) << 16;
            if (lane < 32 && (lane & 3) == 0) gran_st(a.ffg + 8 * g + (lane >> 2), tag, wq);
            if (lane == 0) {
                const int wd = (FF >> 2) + g, wa = wd + (FF >> 5);
                if (ACT == FQ_Q8_0) { gran_st(a.ffg + wd, tag, __builtin_bit_cast(unsigned, h2f_bits(f2h_bits(d)))); gran_st(a.ffg + wa, tag, (unsigned) s); }
                else                { gran_st(a.ffg + wd, tag, __builtin_bit_cast(unsigned, d)); gran_st(a.ffg + wa, tag, __builtin_bit_cast(unsigned, (float) s * d)); }
            }
            if (a.use_counters) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0 && lds_add_rtn(ctl + eng_ctl::EPI, 1u) + 1u == (unsigned) gA * (unsigned)(b + 1))
                    cnt_add(a.cnt + ENG_CNT_FF, (unsigned) gA);            // the workgroup's last Wup epilogue: one device atomic per workgroup
            }
        };
        // Wqkv rows: q / k / v values go out one by one, before the Wup rows (the attention starts as early as it can)
        rows_e(base, pA1, nA1, a.two_norms ? col_e2 : col_e, [&](int i, float v) { if (lane == 0) gran_st(a.qkvg + sc.qg0 + i, tag, __builtin_bit_cast(unsigned, v)); });
        if (a.use_counters) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0 && lds_add_rtn(ctl + eng_ctl::QDONE, 1u) + 1u == (unsigned) ENG_NC * (unsigned)(b + 1) && nA1 > 0) cnt_add(a.cnt + ENG_CNT_QKV, (unsigned) nA1);
        }
        rows_e(base + pA1, pA2, nA2, col_e, [&](int i, float v) { if (lane == 0) { ldsf_st(eng_ctl::OUT + 4 * i, v); lds_add(ctl + eng_ctl::CNT + 4 * (i >> 5), 1u); } });
        for (int gl = c; gl < gA; gl += ENG_NC) epilogue(gl);
        ENG_STAMP(2);
        // ---- the whole GELU image -> LDS (over the LayerNorm images: every consumer must be done with phase A), then Wdown
        eng_cbar(ctl, gen, w);
        ENG_STAMP(3);
        edge_wait(ENG_CNT_FF, (unsigned)(FF / 32) * (unsigned)(b + 1));
        eng_sweep<9>(a.ffg, tag, nwords_ff, (unsigned *) img_ff, ctid, w, 7u);
        eng_cbar(ctl, gen, w);
        ENG_STAMP(4);
        rows_f(base + pA1 + pA2, pB1, nB, col_ff, [&](int i, float v) { if (lane == 0) ldsf_st(eng_ctl::OUTB + 4 * i, v); });
        ENG_STAMP(5);
        // ---- the attention output image -> LDS, then the rows of Wo
        edge_wait(ENG_CNT_ATT, (unsigned) a.H * (unsigned)(b + 1));
        eng_sweep<3>(a.attg, tag, nwords_e, (unsigned *) img_att, ctid, w, 8u);
        eng_cbar(ctl, gen, w);
        ENG_STAMP(6);
        rows_e(base + pA1 + pA2 + pB1, pB2, nB, col_att, [&](int i, float v) {
            if (lane == 0) {
                const float xn = (ldsf_ld(eng_ctl::OUTB + 4 * i) + v) + ldsf_ld(eng_ctl::XRES + 4 * i);     // libfalcon.cpp:2399-2400
                const int row = sc.r0 + i;
                gran_st(a.xg + row, tag, __builtin_bit_cast(unsigned, xn));
                a.x[row] = xn;
                if (a.hidden) a.hidden[(size_t)(b + 1) * E + row] = xn;
            }
        });
        if (a.use_counters) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // this wave's rows of x are out: count the workgroup in once
            if (lane == 0 && lds_add_rtn(ctl + eng_ctl::BDONE, 1u) + 1u == (unsigned) ENG_NC * (unsigned)(b + 1) && nB > 0) cnt_add(a.cnt + ENG_CNT_X, (unsigned) nB);
        }
        ENG_STAMP(7);
    }
    if (a.dbg && c == 0 && lane == 0) {
        long long * r = a.dbg + 4096 + 256 * 4 * 8 + (size_t) blockIdx.x * 8;
        r[5] = (long long) t_wait_land; r[6] = (long long) t_dot; r[7] = (long long) n_rows;
    }
    // ---- ln_f + lm_head (last stage): logits and the per-32-row greedy candidates
    if (a.lm_head) {
        layer_norm_images(a.n_layers == 0, epoch0 + (unsigned) a.n_layers, (unsigned) E * (unsigned) a.n_layers, a.lnf_w, a.lnf_b, img_e, nullptr, nullptr, nullptr);
        const unsigned hbase = (unsigned) a.n_layers * p_blk;
        const int ngroups = (nH + 31) / 32;
        rows_e(hbase, pH, nH, col_e, [&](int i, float v) { if (lane == 0) { ldsf_st(eng_ctl::OUT + 4 * i, v); lds_add(ctl + eng_ctl::CNTH + 4 * (i >> 5), 1u); } });
        // (a partial last group: its missing rows never arrive -- count them in)
        if (c == 0 && lane == 0 && (nH & 31)) lds_add(ctl + eng_ctl::CNTH + 4 * (nH >> 5), (unsigned)(32 - (nH & 31)));
        for (int gl = c; gl < ngroups; gl += ENG_NC) {
            group_ready(eng_ctl::CNTH, gl, 32u);
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

// ---- the tag of a launch's hand-offs: advanced before every launch (never 0); also clears nothing else -- granules carry their tag
__global__ void k_engine_epoch(unsigned * epoch_word, unsigned step, unsigned * cnt) {
    if (threadIdx.x < 4) cnt[32 * threadIdx.x] = 0u;
    if (threadIdx.x == 0) {
        unsigned e = *epoch_word + step;
        if (e < step + 1u) e = 1u;              // wrapped
        *epoch_word = e;
    }
}

// =============================================================================================== host side
struct fq_engine_plan_impl {
    std::vector<fq_engine_sched> sched;
};

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

size_t fq_engine_lds_bytes(int type, int nslot, int64_t E, int64_t FF, int n_layers) {
    const int act = fq_desc(type).act_type;
    return (size_t) nslot * ENG_SLOT + eng_region_r(act, E, FF) + fq_act_col_bytes(act, E) + eng_ctl::BYTES + 32 * (size_t) n_layers + 16;
}

int fq_engine_threads() { return ENG_NT; }

// launches the pre-kernel (tag) and the engine; false = this configuration is not supported (nothing launched)
bool fq_launch_decode_engine(const fq_engine_args & a, int nslot, size_t lds_bytes, hipStream_t st) {
    const int grid = a.n_attn + a.n_stream;
    if ((a.type != FQ_Q4_0 && a.type != FQ_Q4_1 && a.type != FQ_Q5_0 && a.type != FQ_Q5_1 && a.type != FQ_Q8_0) || (nslot != 8 && nslot != 6 && nslot != 4) ||
        a.E > 4 * 3 * 64 * ENG_NC) return false;
    hipLaunchKernelGGL(k_engine_epoch, dim3(1), dim3(64), 0, st, const_cast<unsigned *>(a.epoch_word), (unsigned)(a.n_layers + 2), a.cnt);
#define FQ_ENG_LAUNCH(T, NS) { \
        static size_t g = 0; if (lds_bytes > g) { HIP_CHECK(hipFuncSetAttribute((const void *) k_decode_engine<T, NS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds_bytes)); g = lds_bytes; } \
        hipEvent_t e0_ = nullptr, e1_ = nullptr; fq_prof_events(&e0_, &e1_); \
        if (e0_) hipExtLaunchKernelGGL((k_decode_engine<T, NS>), dim3((unsigned) grid), dim3(ENG_NT), lds_bytes, st, e0_, e1_, 0, a); \
        else     hipLaunchKernelGGL((k_decode_engine<T, NS>), dim3((unsigned) grid), dim3(ENG_NT), lds_bytes, st, a); }
#ifdef ENG_TUNE_ONLY_Q4_0
#define FQ_ENG_CASE(T) case T: if (T == FQ_Q4_0 && nslot == 8) FQ_ENG_LAUNCH(FQ_Q4_0, 8) else return false; break;
#else
#define FQ_ENG_CASE(T) case T: if (nslot == 8) FQ_ENG_LAUNCH(T, 8) else if (nslot == 6) FQ_ENG_LAUNCH(T, 6) else if (nslot == 4) FQ_ENG_LAUNCH(T, 4) else return false; break;
#endif
    switch (a.type) {
        FQ_ENG_CASE(FQ_Q4_0) FQ_ENG_CASE(FQ_Q4_1) FQ_ENG_CASE(FQ_Q5_0) FQ_ENG_CASE(FQ_Q5_1) FQ_ENG_CASE(FQ_Q8_0)
        default: return false;
    }
#undef FQ_ENG_CASE
#undef FQ_ENG_LAUNCH
    return true;
}
