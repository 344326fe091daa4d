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
// F64: the two dot products accumulated in f64 (fq_attn_dev.h; the reference's portable ggml_vec_dot_f32, ggml.c:2296-2300): the fast reference order
template <bool PUBLISH, bool PRE, bool F64 = false>
__device__ __forceinline__ void attn_decode_group_p(const fq_attn_decode_args & a, int h, bool live, int tid, uint8_t * smem, long long * dbg,
                                                    const fq_publish pub, attn_pre & P) {
    constexpr int D = 64, HALF = 32;
    const int H = a.H, HKV = a.HKV;
    const int group = H / HKV, hk = h / group;
    const attn_lds L = attn_lds_carve(smem + 3 * D * 4);
    const float * qh = a.q_src ? a.q_src : a.qkv + (int64_t) h * D;
    const float * kh = a.k_src ? a.k_src : a.qkv + (int64_t)(H + hk) * D;
    const float * vh = a.v_src ? a.v_src : a.qkv + (int64_t)(H + HKV + hk) * D;
    // Every thread rotates ITS OWN dims of q and of the new key (dims 4 s8 .. and 32 + 4 s8 ..: what its score lanes multiply) and fetches its own 4 dims of the
    // new value: no trip of the three rows through LDS and no workgroup barrier in front of the scores (round 4; the rotation is ggml.c:12957-12978's, the same
    // two products and one sum / difference per element). Requests in the order they are needed: n_past, the rope's inputs, then the first 256 key / 128 value rows
    // (whatever n_past is: rows beyond it are never used)
    // (a caller that runs head after head in one launch -- the persistent engine -- has its own workgroup barrier between two heads: the hand-over of q / k / v)
    const int np = *a.n_past_ptr;
    const int s8 = tid & 7, sub = tid & 15;
    const float * csr = a.cs_cur ? a.cs_cur : a.cs + (int64_t) np * HALF * 2;
    const f32x4 x0q = *(const f32x4 *)(qh + 4 * s8), x1q = *(const f32x4 *)(qh + HALF + 4 * s8);
    const f32x4 x0k = *(const f32x4 *)(kh + 4 * s8), x1k = *(const f32x4 *)(kh + HALF + 4 * s8);
    const f32x4 cs0 = *(const f32x4 *)(csr + 8 * s8), cs1 = *(const f32x4 *)(csr + 8 * s8 + 4);      // (cos, sin) of dims 4 s8 .. 4 s8 + 3
    attn_new nw;
    nw.v4 = *(const f32x4 *)(vh + 4 * sub);
    // the key / value row this head's kv group appends (first head of the group): threads 32..63 rotate one pair of key dims each, 64..127 copy one value dim
    const bool append = live && h % group == 0;
    const int k = tid & (HALF - 1);
    float ax0 = 0.0f, ax1 = 0.0f, ac = 0.0f, as = 0.0f, av = 0.0f;
    if (append && tid >= HALF && tid < 2 * HALF) { ax0 = kh[k]; ax1 = kh[k + HALF]; ac = csr[2 * k]; as = csr[2 * k + 1]; }
    if (append && tid >= 2 * HALF && tid < 2 * HALF + D) av = vh[tid - 2 * HALF];
    if constexpr (!PRE) attn_prefetch(a.kc, a.vc, HKV, hk, a.cache_rows, tid, P);
    FQ_STAMP(dbg, 1);
    {
        const f32x4 c = { cs0.x, cs0.z, cs1.x, cs1.z }, sn = { cs0.y, cs0.w, cs1.y, cs1.w };
        nw.qa = f32x4{ x0q.x * c.x - x1q.x * sn.x, x0q.y * c.y - x1q.y * sn.y, x0q.z * c.z - x1q.z * sn.z, x0q.w * c.w - x1q.w * sn.w };      // ggml.c:12974
        nw.qb = f32x4{ x0q.x * sn.x + x1q.x * c.x, x0q.y * sn.y + x1q.y * c.y, x0q.z * sn.z + x1q.z * c.z, x0q.w * sn.w + x1q.w * c.w };      // ggml.c:12975
        nw.ka = f32x4{ x0k.x * c.x - x1k.x * sn.x, x0k.y * c.y - x1k.y * sn.y, x0k.z * c.z - x1k.z * sn.z, x0k.w * c.w - x1k.w * sn.w };
        nw.kb = f32x4{ x0k.x * sn.x + x1k.x * c.x, x0k.y * sn.y + x1k.y * c.y, x0k.z * sn.z + x1k.z * c.z, x0k.w * sn.w + x1k.w * c.w };
    }
    if (append && tid >= HALF && tid < 2 * HALF) {
        const float r0 = ax0 * ac - ax1 * as, r1 = ax0 * as + ax1 * ac;
        float * o = a.kc + ((int64_t) np * HKV + hk) * D; o[k] = r0; o[k + HALF] = r1;
    }
    if (append && tid >= 2 * HALF && tid < 2 * HALF + D) a.vc[((int64_t) np * HKV + hk) * D + (tid - 2 * HALF)] = av;
    FQ_STAMP(dbg, 2);
    const float o = attn_head_block<F64, true>(nullptr, a.kc, a.vc, HKV, hk, np, nullptr, nullptr, a.exp_tab, L, tid, P, dbg, &nw);
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

template <bool PUBLISH, bool F64 = false>
__device__ __forceinline__ void attn_decode_group(const fq_attn_decode_args & a, int h, bool live, int tid, uint8_t * smem, long long * dbg = nullptr,
                                                  const fq_publish pub = fq_publish{ nullptr, 0u }) {
    attn_pre P;
    attn_decode_group_p<PUBLISH, false, F64>(a, h, live, tid, smem, dbg, pub, P);
}

// the barriers of attn_decode_group (attn_head_block's; the rope in front of it needs none since round 4), for waves of the same workgroup that sit a group out
__device__ __forceinline__ void attn_decode_group_idle() {
#pragma unroll
    for (int i = 0; i < FQ_ATTN_HEAD_BARRIERS; ++i) __syncthreads();
}

static inline size_t attn_decode_lds(int max_n_kv) { return 3 * 64 * 4 + 16 * 4 + 16 * 64 * 8 + (((size_t) max_n_kv * 4 + 15) & ~(size_t) 15); }
