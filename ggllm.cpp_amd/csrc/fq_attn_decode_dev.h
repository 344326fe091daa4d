// fq_attn_decode_dev.h -- one query head of a single-token (N = 1) attention by a lockstep group of 256 threads, shared by
// k_attn_decode, k_attn_out (kernels_decode.hip) and the persistent decode engine (kernels_engine.hip): identical code =>
// identical bits on every fused path.
#pragma once
#include "fq_block_dev.h"
#include "fq_attn_dev.h"

#ifndef FQ_STAMP
#define FQ_STAMP(dbg, slot) do { if ((dbg) && threadIdx.x == 0) (dbg)[(size_t) blockIdx.x * 8 + (slot)] = (long long) wall_clock64(); } while (0)
#endif

struct fq_attn_decode_args {
    const float * qkv; int H, HKV; const int * n_past_ptr; const float * cs; float * kc, * vc; const uint16_t * exp_tab;
    float * att; uint8_t * att_image; int att_act_type;
    int cache_rows;                 // key/value rows [0, cache_rows) are allocated (>= n_past + 1): prefetch bound before n_past is known
    const float * cs_cur;           // optional: the rope table's row for n_past, prepared by the preceding k_gemv_ln
    // optional (persistent engine): the head's q row and its kv head's k / v rows (64 floats each, LDS) instead of rows of qkv
    const float * q_src, * k_src, * v_src;
};
// PUBLISH: the 32-bit word `word` of the image / f32 row goes out as ONE 8-byte granule {tag = epoch, value}, a single
// agent-scope (write-through) store: the data is its own flag (the consumer re-reads a granule until its tag is this
// launch's epoch), so no drain, no barrier and no counter are needed on the producing side.
struct fq_publish { unsigned long long * gran; unsigned epoch; };
template <bool PUBLISH, typename T> __device__ __forceinline__ void out_store(T * base, int64_t word, T v, const fq_publish & pub) {
    static_assert(sizeof(T) == 4, "32-bit words");
    if constexpr (PUBLISH) __hip_atomic_store(pub.gran + word, ((unsigned long long) pub.epoch << 32) | __builtin_bit_cast(unsigned, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else base[word] = v;
}
// PRE: the first key / value rows were requested by the caller (attn_prefetch with the same arguments) before q was known and sit in P
template <bool PUBLISH, bool PRE>
__device__ __forceinline__ void attn_decode_group_p(const fq_attn_decode_args & a, int h, bool live, int tid, uint8_t * smem, long long * dbg,
                                                    const fq_publish pub, attn_pre & P) {
    constexpr int D = 64, HALF = 32;
    const int H = a.H, HKV = a.HKV;
    const int group = H / HKV, hk = h / group;
    float * qr = (float *) smem;                    // rotated q [64]
    float * kr = qr + D;                            // rotated new k [64]
    float * vn = kr + D;                            // new v [64]
    const attn_lds L = attn_lds_carve(smem + 3 * D * 4);
    const float * qh = a.q_src ? a.q_src : a.qkv + (int64_t) h * D;
    const float * kh = a.k_src ? a.k_src : a.qkv + (int64_t)(H + hk) * D;
    const float * vh = a.v_src ? a.v_src : a.qkv + (int64_t)(H + HKV + hk) * D;
    // requests in the order they are needed: n_past, the rope's inputs (every thread asks, 128 use them), then the first
    // 256 key / 128 value rows (whatever n_past is: rows beyond it are never used)
    const int np = *a.n_past_ptr;
    const int k = tid & (HALF - 1);
    const float * src = tid < HALF ? qh : kh;
    const float * csr = a.cs_cur ? a.cs_cur : a.cs + (int64_t) np * HALF * 2;
    const float x0 = src[k], x1 = src[k + HALF];
    const float c = csr[2 * k], s = csr[2 * k + 1];
    const float vnew = vh[tid & (D - 1)];
    if constexpr (!PRE) attn_prefetch(a.kc, a.vc, HKV, hk, a.cache_rows, tid, P);
    FQ_STAMP(dbg, 1);
    const bool append = live && h % group == 0;
    if (tid < 2 * HALF) {
        const float r0 = x0 * c - x1 * s, r1 = x0 * s + x1 * c;                      // ggml.c:12974-12975
        float * dst = tid < HALF ? qr : kr;
        dst[k] = r0; dst[k + HALF] = r1;
        if (tid >= HALF && append) { float * o = a.kc + ((int64_t) np * HKV + hk) * D; o[k] = r0; o[k + HALF] = r1; }
    } else if (tid < 2 * HALF + D) {
        const int d = tid - 2 * HALF;
        vn[d] = vnew;
        if (append) a.vc[((int64_t) np * HKV + hk) * D + d] = vnew;
    }
    __syncthreads();
    FQ_STAMP(dbg, 2);
    const float o = attn_head_block(qr, a.kc, a.vc, HKV, hk, np, kr, vn, a.exp_tab, L, tid, P, dbg);
    FQ_STAMP(dbg, 6);
    if (tid < 64) {
        if (a.att && live) out_store<PUBLISH>(a.att, (int64_t) h * D + tid, o, pub);
        if (a.att_image) {                                                           // lanes 0-31 / 32-63 = the head's two 32-blocks
            const float amax = reduce32(fabsf(o), op_max());
            const float d  = amax / 127.0f;
            const float id = d ? 1.0f / d : 0.0f;
            const int q = round_half_away(o * id);
            const int s = reduce32(q, op_add());
            const int64_t E = (int64_t) H * D;
            // the image as 32-bit words: [qs E/4 | d E/32 | aux E/32]
            unsigned w = (unsigned) q & 0xFFu;                                       // 4 lanes -> one word of qs
            w |= ((unsigned) __shfl_down((int) w, 1) & 0xFFu) << 8;
            w |= ((unsigned) __shfl_down((int) w, 2) & 0xFFFFu) << 16;
            if (live && (tid & 3) == 0) out_store<PUBLISH>((unsigned *) a.att_image, ((int64_t) h * D + tid) >> 2, w, pub);
            if (live && (tid & 31) == 0) {
                const int64_t wd = (E >> 2) + 2 * (int64_t) h + (tid >> 5), wa = wd + (E >> 5);      // words of d and aux
                if (a.att_act_type == FQ_Q8_0) {
                    out_store<PUBLISH>((float *) a.att_image, wd, h2f_bits(f2h_bits(d)), pub);
                    out_store<PUBLISH>((int32_t *) a.att_image, wa, (int32_t) s, pub);
                } else {
                    out_store<PUBLISH>((float *) a.att_image, wd, d, pub);
                    out_store<PUBLISH>((float *) a.att_image, wa, (float) s * d, pub);
                }
            }
        }
    }
}

template <bool PUBLISH>
__device__ __forceinline__ void attn_decode_group(const fq_attn_decode_args & a, int h, bool live, int tid, uint8_t * smem, long long * dbg = nullptr,
                                                  const fq_publish pub = fq_publish{ nullptr, 0u }) {
    attn_pre P;
    attn_decode_group_p<PUBLISH, false>(a, h, live, tid, smem, dbg, pub, P);
}

// the barriers of attn_decode_group (1 after the rope + attn_head_block's), for waves of the same workgroup that sit a group out
__device__ __forceinline__ void attn_decode_group_idle() {
#pragma unroll
    for (int i = 0; i < 1 + FQ_ATTN_HEAD_BARRIERS; ++i) __syncthreads();
}

static inline size_t attn_decode_lds(int max_n_kv) { return 3 * 64 * 4 + 16 * 4 + 16 * 64 * 8 + (((size_t) max_n_kv * 4 + 15) & ~(size_t) 15); }
