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
#include <vector>

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

__device__ __forceinline__ void glds16_nt(const void * gsrc, unsigned lds_dst) {       // 64 lanes x 16 B -> 1 KiB of LDS at lds_dst
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

__device__ __forceinline__ unsigned long long gran_ld(const unsigned long long * p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void gran_st(unsigned long long * p, unsigned tag, unsigned v) { __hip_atomic_store(p, ((unsigned long long) tag << 32) | v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ bool eng_failed(const unsigned * err) { return __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u; }
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
    static constexpr unsigned PTRS = CBAR + 16;   // per block 4 x 8 bytes: this workgroup's first byte of Wqkv, Wup, Wdown, Wo; then lm_head's
    static constexpr unsigned BYTES = PTRS;       // + 32 * n_layers + 8
};
__device__ __forceinline__ unsigned long long lds_ld64(unsigned addr) {
    unsigned long long v; asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(addr) : "memory"); return v;
}
__device__ __forceinline__ void lds_st64(unsigned addr, unsigned long long v) { asm volatile("ds_write_b64 %0, %1" :: "v"(addr), "v"(v) : "memory"); }

// one row's dot out of the ring. pos = the row's stream position reduced modulo RING (wave-uniform); units = 32-element blocks
template <int TYPE, int RING>
__device__ __forceinline__ float eng_row_dot(const uint8_t * ring, unsigned pos, int nblk, const fq_actcol & col, int lane) {
    constexpr int CB = fq_lay<TYPE>::CB, TS = fq_lay<TYPE>::TS;
    constexpr fq_type_desc D = fq_desc(TYPE);
    float acc = 0.0f;
    auto wrap = [](unsigned o) { return o >= (unsigned) RING ? o - (unsigned) RING : o; };
    for (int u0 = 0; u0 < nblk; u0 += 64) {
        const int u = u0 + lane;
        const bool ok = u < nblk;
        const int uc = ok ? u : nblk - 1;
        const int c = uc / CB, j = uc - c * CB;
        const int rem = nblk - c * CB, nbc = rem < CB ? rem : CB;
        const unsigned cb = pos + (unsigned)(c * CB * TS);                 // < 2 RING: a row is shorter than the ring
        fq_unit_regs r{};
        r.q = *(const fq_u4 *)(ring + wrap(cb + (unsigned)(j * D.plane[0].bytes)));
        if constexpr (TYPE == FQ_Q8_0) r.q2 = *(const fq_u4 *)(ring + wrap(cb + (unsigned)(j * 32 + 16)));
        const unsigned p1 = wrap(cb + (unsigned)(nbc * D.plane[0].bytes + j * D.plane[1].bytes));
        if constexpr (TYPE == FQ_Q4_0 || TYPE == FQ_Q8_0) r.dm = *(const uint16_t *)(ring + p1);
        else if constexpr (TYPE == FQ_Q4_1)               r.dm = *(const uint32_t *)(ring + p1);
        else {                                                               // Q5_0 / Q5_1: plane 1 = qh, plane 2 = d (,m)
            r.s0 = *(const uint32_t *)(ring + p1);
            const unsigned p2 = wrap(cb + (unsigned)(nbc * (D.plane[0].bytes + D.plane[1].bytes) + j * D.plane[2].bytes));
            if constexpr (TYPE == FQ_Q5_0) r.dm = *(const uint16_t *)(ring + p2); else r.dm = *(const uint32_t *)(ring + p2);
        }
        const float v = fq_unit<TYPE>::dot(r, col, uc);
        acc += ok ? v : 0.0f;
    }
    return wave_sum(acc);
}

struct eng_wait {                 // per-wave state of the bounded waits
    unsigned * err; bool dead;
    __device__ __forceinline__ bool spin(unsigned & spins, unsigned code) {      // true = keep waiting
        if (dead) return false;
        ++spins;
        if ((spins & 255u) == 0u && eng_failed(err)) { dead = true; return false; }
        if (spins > ENG_SPIN_MAX) { if ((threadIdx.x & 63) == 0) eng_fail(err, code); dead = true; return false; }
        return true;
    }
};

// barrier among the consumer waves (generation counter in LDS)
__device__ __forceinline__ void eng_cbar(unsigned ctl, unsigned & gen, eng_wait & w) {
    ++gen;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                      // this wave's LDS stores are done before it arrives
    if ((threadIdx.x & 63) == 0) lds_add(ctl + eng_ctl::CBAR, 1u);
    const unsigned target = gen * (unsigned) ENG_NC;
    for (unsigned spins = 0; lds_ld_u(ctl + eng_ctl::CBAR) < target;) { if (!w.spin(spins, 3u)) break; __builtin_amdgcn_s_sleep(1); }
}

// the consumers' share of a granule buffer: words [0, nwords) -> LDS (dst) once every granule carries `tag`
__device__ __forceinline__ void eng_sweep(const unsigned long long * gran, unsigned tag, int nwords, unsigned * dst, int ctid, eng_wait & w, unsigned code) {
    constexpr int NG = 3, CT = 64 * ENG_NC;
    for (int base = 0; base < nwords; base += NG * CT) {
        unsigned v[NG];
        for (unsigned spins = 0;;) {
            bool ok = true;
#pragma unroll
            for (int k = 0; k < NG; ++k) {
                const int i = base + k * CT + ctid;
                const unsigned long long x = gran_ld(gran + (i < nwords ? i : nwords - 1));
                v[k] = (unsigned) x; ok = ok && (unsigned)(x >> 32) == tag;
            }
            if (__all(ok)) break;
            if (!w.spin(spins, code)) break;
            __builtin_amdgcn_s_sleep(2);
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
        // ================================================================================ attention workgroup
        const int grp = tid >> 8, gtid = tid & 255;
        const bool idle = grp >= a.hpw;
        int h = (int) blockIdx.x * a.hpw + grp;
        const bool live = !idle && h < a.H;
        if (h >= a.H) h = a.H - 1;
        const int hk = h / (a.H / a.HKV);
        uint8_t * gbase = smem + (size_t)(idle ? 0 : grp) * a.attn_lds_group;
        float * stage = (float *) gbase;                                   // q | k | v of the head, 64 floats each
        eng_wait w{ a.err, false };
        for (int b = 0; b < a.n_layers; ++b) {
            const unsigned tag = epoch0 + (unsigned) b + 1u;
            if (!idle && gtid < 192) {
                const int part = gtid >> 6, d = gtid & 63;
                const int row = (part == 0 ? h : (part == 1 ? a.H + hk : a.H + a.HKV + hk)) * 64 + d;
                unsigned long long x = 0;
                for (unsigned spins = 0;;) {
                    x = gran_ld(a.qkvg + row);
                    if (__all((unsigned)(x >> 32) == tag)) break;
                    if (!w.spin(spins, 4u)) break;
                    __builtin_amdgcn_s_sleep(2);
                }
                stage[gtid] = __builtin_bit_cast(float, (unsigned) x);
            }
            __syncthreads();
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
    const int nA1 = 32 * (sc.qg1 - sc.qg0), nA2 = 32 * (sc.ug1 - sc.ug0), nB = sc.r1 - sc.r0;
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
            p = m == 0 ? L.qkv + (size_t) sc.qg0 * 32 * a.rsE : (m == 1 ? L.up + (size_t) sc.ug0 * 32 * a.rsE : (m == 2 ? L.down + (size_t) sc.r0 * a.rsF : L.wo + (size_t) sc.r0 * a.rsE));
        }
        lds_st64(ctl + eng_ctl::PTRS + 8 * i, (unsigned long long)(uintptr_t) p);
    }
    if (tid < 32) lds_st(ctl + eng_ctl::CNT + 4 * tid, 0u);                 // CNT and CNTH
    if (tid < 16) lds_st(ctl + eng_ctl::LOW + 4 * tid, tid < ENG_NC ? 0u : 0xFFFFFFFFu);
    if (tid == 0) { lds_st(ctl + eng_ctl::LANDED, 0u); lds_st(ctl + eng_ctl::CBAR, 0u); }
    __syncthreads();                                                       // the only workgroup barrier: before the roles split

    if (wid == 0) {
        // ================================================================================ loader
        const unsigned ring_lds = (unsigned)(uintptr_t) ring;
        eng_wait w{ a.err, false };
        unsigned pos = 0, rp = 0, freed = 0, issued = 0;                   // stream position (bytes), ring piece index, cached low-water mark
        auto seg = [&](const uint8_t * src, unsigned padded) {
            for (unsigned off = 0; off < padded; off += 1024u) {
                if (pos + 1024u - freed > (unsigned) RING) {               // ring full: let everything in flight land and report it
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (lane == 0) lds_st(ctl + eng_ctl::LANDED, pos);
                    for (unsigned spins = 0;;) {
                        unsigned v = lds_ld(ctl + eng_ctl::LOW + 4 * (lane < 16 ? lane : 0));
                        v = (unsigned) wave_reduce((int) v, [](int x, int y) { return (unsigned) x < (unsigned) y ? x : y; });
                        freed = __builtin_amdgcn_readfirstlane(v);
                        if (pos + 1024u - freed <= (unsigned) RING) break;
                        if (!w.spin(spins, 1u)) break;
                        __builtin_amdgcn_s_sleep(1);
                    }
                    if (w.dead) return;
                }
                glds16_nt(src + off + lane * 16, __builtin_amdgcn_readfirstlane(ring_lds + rp * 1024u));
                pos += 1024u; ++issued;
                if (++rp == (unsigned)(NSLOT * 16)) rp = 0;
                if ((issued & 15u) == 0u) {                                // at most 32 + 16 pieces in flight
                    asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
                    if (lane == 0 && pos >= 32u * 1024u) lds_st(ctl + eng_ctl::LANDED, pos - 32u * 1024u);
                }
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
        if (lane == 0) lds_st(ctl + eng_ctl::LANDED, pos);
        return;
    }

    // ==================================================================================== consumers
    const int c = wid - 1, ctid = tid - 64;
    constexpr int CT = 64 * ENG_NC;
    constexpr int NLN = 3;                                                 // float4 of the row per consumer thread: n_embd <= 8448
    eng_wait w{ a.err, false };
    unsigned gen = 0;                                                      // consumer barrier generation
    const int nblkE = E / 32, nblkF = FF / 32;
    const int64_t nv = E >> 2;

    // rows [0, nrows) of a segment that starts at stream position seg_pos: mine are c, c + NC, ...
    auto rows = [&](unsigned seg_pos, unsigned padded, int nrows, unsigned rs, int nblk, const fq_actcol & col, auto && sink) {
        const unsigned row_bytes = (unsigned)(nblk * TS);
        for (int i = c; i < nrows; i += ENG_NC) {
            const unsigned p = seg_pos + (unsigned) i * rs;
            const unsigned need = p + row_bytes;
            for (unsigned spins = 0; lds_ld_u(ctl + eng_ctl::LANDED) < need;) { if (!w.spin(spins, 2u)) break; __builtin_amdgcn_s_sleep(1); }
            const float v = eng_row_dot<TYPE, RING>(ring, p % (unsigned) RING, nblk, col, lane);
            sink(i, v);
            const int nx = i + ENG_NC;
            if (lane == 0) lds_st(ctl + eng_ctl::LOW + 4 * c, nx < nrows ? seg_pos + (unsigned) nx * rs : seg_pos + padded);
        }
        if (c >= nrows && lane == 0) lds_st(ctl + eng_ctl::LOW + 4 * c, seg_pos + padded);     // nothing of this segment was mine
    };
    // LayerNorm (+ the second norm of a two-norm block) of the residual row -> Q8 images; the row comes from memory (first
    // block of the stage) or from the granules the previous block's owners published. Also keeps x[r0 .. r1) for phase B.
    auto layer_norm_images = [&](bool from_mem, unsigned tag, const float * w1, const float * b1, uint8_t * img1,
                                 const float * w2, const float * b2, uint8_t * img2) {
        ln_row_regs<NLN> xr, wr, br;
        ln_regs_issue_wb(w1, b1, E, CT, wr, br, ctid);
#pragma unroll
        for (int k = 0; k < NLN; ++k) {
            const int64_t i = (int64_t) k * CT + ctid, j = i < nv ? i : nv - 1;
            if (from_mem) { xr.t[k] = ((const float4 *) a.x_in)[j]; continue; }
            const unsigned long long * g = a.xg + 4 * j;
            unsigned v0 = 0, v1 = 0, v2 = 0, v3 = 0;
            for (unsigned spins = 0;;) {
                const unsigned long long a0 = gran_ld(g), a1 = gran_ld(g + 1), a2 = gran_ld(g + 2), a3 = gran_ld(g + 3);
                v0 = (unsigned) a0; v1 = (unsigned) a1; v2 = (unsigned) a2; v3 = (unsigned) a3;
                const bool ok = (unsigned)(a0 >> 32) == tag && (unsigned)(a1 >> 32) == tag && (unsigned)(a2 >> 32) == tag && (unsigned)(a3 >> 32) == tag;
                if (__all(ok)) break;
                if (!w.spin(spins, 5u)) break;
                __builtin_amdgcn_s_sleep(2);
            }
            xr.t[k] = make_float4(__builtin_bit_cast(float, v0), __builtin_bit_cast(float, v1), __builtin_bit_cast(float, v2), __builtin_bit_cast(float, v3));
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
        for (unsigned spins = 0; lds_ld_u(ctl + cnt + 4 * gl) < target;) { if (!w.spin(spins, 6u)) break; __builtin_amdgcn_s_sleep(1); }
    };
    const fq_actcol col_e  = { (const int8_t *) img_e,  (const float *)(img_e + fq_act_d_off(ACT, E)),  (const void *)(img_e + fq_act_aux_off(ACT, E)) };
    const fq_actcol col_e2 = { (const int8_t *) img_e2, (const float *)(img_e2 + fq_act_d_off(ACT, E)), (const void *)(img_e2 + fq_act_aux_off(ACT, E)) };
    const fq_actcol col_att = { (const int8_t *) img_att, (const float *)(img_att + fq_act_d_off(ACT, E)), (const void *)(img_att + fq_act_aux_off(ACT, E)) };
    const fq_actcol col_ff = { (const int8_t *) img_ff, (const float *)(img_ff + fq_act_d_off(ACT, FF)), (const void *)(img_ff + fq_act_aux_off(ACT, FF)) };
    const int nwords_ff = (FF >> 2) + 2 * (FF >> 5), nwords_e = (E >> 2) + 2 * (E >> 5);

    for (int b = 0; b < a.n_layers; ++b) {
        const fq_engine_layer & L = a.layers[b];
        const unsigned base = (unsigned) b * p_blk;
        const unsigned tag = epoch0 + (unsigned) b + 1u;                   // tag of everything block b publishes
        // ---- residual row -> LayerNorm image(s)
        layer_norm_images(b == 0, epoch0 + (unsigned) b, L.ln_w, L.ln_b, img_e, a.two_norms ? L.ln2_w : nullptr, L.ln2_b, img_e2);
        // ---- phase A: rows of Wqkv (norm image: the attention norm when the block has two) and Wup
        const unsigned cnt_target = 32u * (unsigned)(b + 1);
        rows(base, pA1, nA1, rsE, nblkE, a.two_norms ? col_e2 : col_e, [&](int i, float v) { if (lane == 0) { ldsf_st(eng_ctl::OUT + 4 * i, v); lds_add(ctl + eng_ctl::CNT + 4 * (i >> 5), 1u); } });
        rows(base + pA1, pA2, nA2, rsE, nblkE, col_e, [&](int i, float v) { if (lane == 0) { ldsf_st(eng_ctl::OUT + 4 * (nA1 + i), v); lds_add(ctl + eng_ctl::CNT + 4 * ((nA1 + i) >> 5), 1u); } });
        for (int gl = c; gl < (nA1 + nA2) / 32; gl += ENG_NC) {
            group_ready(eng_ctl::CNT, gl, cnt_target);
            const int j = lane & 31;
            float v = ldsf_ld(eng_ctl::OUT + 4 * (32 * gl + j));
            if (32 * gl < nA1) {                                           // a Wqkv group: 32 f32 values for the attention workgroups
                const int row = sc.qg0 * 32 + 32 * gl + j;
                if (lane < 32) gran_st(a.qkvg + row, tag, __builtin_bit_cast(unsigned, v));
            } else {                                                       // a Wup group: GELU, Q8 block of 32 (kernels_decode.hip, k_gemv_ln epilogue)
                const int g = sc.ug0 + (32 * gl - nA1) / 32;               // block index in the FF-long image
                v = h2f_bits(a.gelu_tab[f2h_bits(v)]);                     // ggml.c:3477-3484
                const float amax = reduce32(fabsf(v), op_max());
                const float d  = amax / 127.0f;
                const float id = d ? 1.0f / d : 0.0f;
                const int q = round_half_away(v * id);
                const int s = reduce32(q, op_add());
                unsigned wq = (unsigned) q & 0xFFu;                        // 4 lanes -> one word of qs
                wq |= ((unsigned) __shfl_down((int) wq, 1) & 0xFFu) << 8;
                wq |= ((unsigned) __shfl_down((int) wq, 2) & 0xFFFFu) << 16;
                if (lane < 32 && (lane & 3) == 0) gran_st(a.ffg + 8 * g + (lane >> 2), tag, wq);
                if (lane == 0) {
                    const int wd = (FF >> 2) + g, wa = wd + (FF >> 5);
                    if (ACT == FQ_Q8_0) { gran_st(a.ffg + wd, tag, __builtin_bit_cast(unsigned, h2f_bits(f2h_bits(d)))); gran_st(a.ffg + wa, tag, (unsigned) s); }
                    else                { gran_st(a.ffg + wd, tag, __builtin_bit_cast(unsigned, d)); gran_st(a.ffg + wa, tag, __builtin_bit_cast(unsigned, (float) s * d)); }
                }
            }
        }
        // ---- the whole GELU image -> LDS (over the LayerNorm images: every consumer must be done with phase A), then Wdown
        eng_cbar(ctl, gen, w);
        eng_sweep(a.ffg, tag, nwords_ff, (unsigned *) img_ff, ctid, w, 7u);
        eng_cbar(ctl, gen, w);
        rows(base + pA1 + pA2, pB1, nB, rsF, nblkF, col_ff, [&](int i, float v) { if (lane == 0) ldsf_st(eng_ctl::OUTB + 4 * i, v); });
        // ---- the attention output image -> LDS, then the rows of Wo
        eng_sweep(a.attg, tag, nwords_e, (unsigned *) img_att, ctid, w, 8u);
        eng_cbar(ctl, gen, w);
        rows(base + pA1 + pA2 + pB1, pB2, nB, rsE, nblkE, col_att, [&](int i, float v) {
            if (lane == 0) {
                const float xn = (ldsf_ld(eng_ctl::OUTB + 4 * i) + v) + ldsf_ld(eng_ctl::XRES + 4 * i);     // libfalcon.cpp:2399-2400
                const int row = sc.r0 + i;
                gran_st(a.xg + row, tag, __builtin_bit_cast(unsigned, xn));
                a.x[row] = xn;
                if (a.hidden) a.hidden[(size_t)(b + 1) * E + row] = xn;
            }
        });
    }
    // ---- ln_f + lm_head (last stage): logits and the per-32-row greedy candidates
    if (a.lm_head) {
        layer_norm_images(a.n_layers == 0, epoch0 + (unsigned) a.n_layers, a.lnf_w, a.lnf_b, img_e, nullptr, nullptr, nullptr);
        const unsigned hbase = (unsigned) a.n_layers * p_blk;
        const int ngroups = (nH + 31) / 32;
        rows(hbase, pH, nH, rsE, nblkE, col_e, [&](int i, float v) { if (lane == 0) { ldsf_st(eng_ctl::OUT + 4 * i, v); lds_add(ctl + eng_ctl::CNTH + 4 * (i >> 5), 1u); } });
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
__global__ void k_engine_epoch(unsigned * epoch_word, unsigned step) {
    unsigned e = *epoch_word + step;
    if (e < step + 1u) e = 1u;                  // wrapped
    *epoch_word = e;
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
    const int64_t rsE = (int64_t) fq_il_row_stride(d, E / 32), rsF = (int64_t) fq_il_row_stride(d, FF / 32);
    const int gq = qkv_rows / 32, gu = FF / 32, ga = gq + gu;
    out.assign((size_t) n_stream, fq_engine_sched{});
    // groups per workgroup: as even as 32-row groups allow
    std::vector<int> nA((size_t) n_stream);
    for (int s = 0; s < n_stream; ++s) nA[(size_t) s] = (int)(((int64_t)(s + 1) * ga) / n_stream - ((int64_t) s * ga) / n_stream);
    // Wqkv groups: one each to the workgroups that have room, round after round
    std::vector<int> nq((size_t) n_stream, 0);
    for (int left = gq; left > 0;) {
        int given = 0;
        for (int s = 0; s < n_stream && left > 0; ++s) if (nq[(size_t) s] < nA[(size_t) s]) { ++nq[(size_t) s]; --left; ++given; }
        if (!given) return false;
    }
    int q = 0, u = 0, mg = 0, mr = 0;
    const double total = (double) ga * 32 * rsE + (double) E * (rsF + rsE);
    double acc_bytes = 0.0; int r = 0;
    for (int s = 0; s < n_stream; ++s) {
        fq_engine_sched & o = out[(size_t) s];
        o.qg0 = q; q += nq[(size_t) s]; o.qg1 = q;
        o.ug0 = u; u += nA[(size_t) s] - nq[(size_t) s]; o.ug1 = u;
        // rows of phase B: up to the cumulative byte target
        acc_bytes += (double) nA[(size_t) s] * 32 * rsE;
        const double target = total * (s + 1) / n_stream;
        int nb = (int)((target - acc_bytes) / (double)(rsF + rsE) + 0.5);
        if (nb < 0) nb = 0;
        if (s == n_stream - 1 || r + nb > E) nb = E - r;
        o.r0 = r; r += nb; o.r1 = r;
        acc_bytes += (double) nb * (rsF + rsE);
        if (nA[(size_t) s] > mg) mg = nA[(size_t) s];
        if (nb > mr) mr = nb;
    }
    if (q != gq || u != gu || r != E) return false;
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
    hipLaunchKernelGGL(k_engine_epoch, dim3(1), dim3(1), 0, st, const_cast<unsigned *>(a.epoch_word), (unsigned)(a.n_layers + 2));
#define FQ_ENG_LAUNCH(T, NS) { \
        static size_t g = 0; if (lds_bytes > g) { HIP_CHECK(hipFuncSetAttribute((const void *) k_decode_engine<T, NS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds_bytes)); g = lds_bytes; } \
        hipEvent_t e0_ = nullptr, e1_ = nullptr; fq_prof_events(&e0_, &e1_); \
        if (e0_) hipExtLaunchKernelGGL((k_decode_engine<T, NS>), dim3((unsigned) grid), dim3(ENG_NT), lds_bytes, st, e0_, e1_, 0, a); \
        else     hipLaunchKernelGGL((k_decode_engine<T, NS>), dim3((unsigned) grid), dim3(ENG_NT), lds_bytes, st, a); }
#define FQ_ENG_CASE(T) case T: if (nslot == 8) FQ_ENG_LAUNCH(T, 8) else if (nslot == 6) FQ_ENG_LAUNCH(T, 6) else if (nslot == 4) FQ_ENG_LAUNCH(T, 4) else return false; break;
    switch (a.type) {
        FQ_ENG_CASE(FQ_Q4_0) FQ_ENG_CASE(FQ_Q4_1) FQ_ENG_CASE(FQ_Q5_0) FQ_ENG_CASE(FQ_Q5_1) FQ_ENG_CASE(FQ_Q8_0)
        default: return false;
    }
#undef FQ_ENG_CASE
#undef FQ_ENG_LAUNCH
    return true;
}
