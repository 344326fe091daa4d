/*
 * ref_harness.c -- TEST INFRASTRUCTURE ONLY.
 *
 * A thin C harness that is compiled TOGETHER WITH the reference's own ggml.c / k_quants.c
 * (where they lie under /root/reference, see oracle/Makefile) into oracle/_ref/libggml_ref.so.
 * It contains no reference source: it only calls the reference's public API (ggml.h) so that
 * Python (ctypes) can drive the real reference arithmetic with plain pointers:
 *
 *   - quantize / dequantize / vec_dot through ggml_internal_get_quantize_fn   (ggml.c:1721)
 *   - a one-node GGML_OP_MUL_MAT graph through ggml_graph_compute            (ggml.c:17307)
 *     -> ggml_compute_forward_mul_mat_q_f32                                   (ggml.c:11318)
 *   - norm / rope(neox, dynamic NTK) / scale+mask+soft_max / gelu graphs     (ggml.c:10540, 12819, 12389, 3477)
 *   - a whole Falcon decoder stack written with the same op sequence as
 *     falcon_eval_internal (libfalcon.cpp:2115-2466), FALCON_NO_KV_UPGRADE flavour of the
 *     V cache (libfalcon.cpp:2246-2253, 2331-2343) which is arithmetically identical.
 *
 * It is used (a) to generate tests/golden/ fixtures (oracle/gen_golden.py), (b) to validate
 * oracle/oracle_*.c in this container, (c) optionally as bench.py's cpu_baseline ("reference").
 * Nothing in the product path may load it.
 */
#include "ggml.h"

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

static struct ggml_context * make_ctx(size_t bytes) {
    struct ggml_init_params ip;
    ip.mem_size   = bytes;
    ip.mem_buffer = NULL;
    ip.no_alloc   = false;
    return ggml_init(ip);
}

/* Must be called once: ggml_init() builds the fp16 / GELU / EXP tables on first use. */
void ref_init(void) {
    struct ggml_context * c = make_ctx(1 << 20);
    ggml_free(c);
}

size_t ref_type_size(int type)  { return ggml_type_size((enum ggml_type) type); }
int    ref_blck_size(int type)  { return ggml_blck_size((enum ggml_type) type); }
int    ref_vec_dot_type(int type) { return (int) ggml_internal_get_quantize_fn(type).vec_dot_type; }

float    ref_fp16_to_fp32(uint16_t h) { ggml_fp16_t x; memcpy(&x, &h, 2); return ggml_fp16_to_fp32(x); }
uint16_t ref_fp32_to_fp16(float f)    { ggml_fp16_t x = ggml_fp32_to_fp16(f); uint16_t h; memcpy(&h, &x, 2); return h; }

/* weight quantizer, "_reference" flavour (deterministic model-file creation) */
void ref_quantize_reference(int type, const float * x, void * out, int k) {
    ggml_internal_get_quantize_fn(type).quantize_row_q_reference(x, out, k);
}
/* the model-file quantizer entry (falcon_model_quantize_internal calls it per chunk, libfalcon.cpp:3669-3705): blocks of
 * src[start .. start+n) into dst, histogram added to hist[16] */
size_t ref_quantize_chunk(int type, const float * src, void * dst, int start, int n, int64_t * hist) {
    return ggml_quantize_chunk((enum ggml_type) type, src, dst, start, n, hist);
}
/* weight quantizer, build-native flavour (SIMD where the build has it) */
void ref_quantize_native(int type, const float * x, void * out, int k) {
    ggml_internal_get_quantize_fn(type).quantize_row_q(x, out, k);
}
/* activation quantizer actually used by mul_mat for weights of `wtype` (build-native flavour) */
void ref_quantize_dot(int wtype, const float * x, void * out, int k) {
    ggml_internal_get_quantize_fn(wtype).quantize_row_q_dot(x, out, k);
}
void ref_dequantize(int type, const void * in, float * y, int k) {
    ggml_internal_get_quantize_fn(type).dequantize_row_q(in, y, k);
}
float ref_vec_dot(int wtype, int n, const void * w, const void * a) {
    float s = 0.0f;
    ggml_internal_get_quantize_fn(wtype).vec_dot_q(n, &s, w, a);
    return s;
}

/* dst[M x N] = W_q[K x M]^T x[K x N] through the real graph executor. */
void ref_mul_mat(int wtype, const void * w, int64_t K, int64_t M,
                 const float * x, int64_t N, float * dst, int n_threads) {
    const size_t wbytes = (size_t)(K / ggml_blck_size(wtype)) * ggml_type_size(wtype) * M;
    struct ggml_context * c = make_ctx(wbytes + (size_t)(K * N + M * N) * 4 + (64u << 20));
    struct ggml_tensor * W = ggml_new_tensor_2d(c, (enum ggml_type) wtype, K, M);
    struct ggml_tensor * X = ggml_new_tensor_2d(c, GGML_TYPE_F32, K, N);
    memcpy(W->data, w, wbytes);
    memcpy(X->data, x, (size_t) K * N * 4);
    struct ggml_tensor * Y = ggml_mul_mat(c, W, X);
    struct ggml_cgraph g = ggml_build_forward(Y);
    g.n_threads = n_threads;
    ggml_graph_compute(c, &g);
    memcpy(dst, Y->data, (size_t) M * N * 4);
    ggml_free(c);
}

/* y = norm(x) rows of length n (eps 1e-5, ggml.c:10540) */
void ref_norm(const float * x, int64_t n, int64_t rows, float * y) {
    struct ggml_context * c = make_ctx((size_t) n * rows * 8 + (16u << 20));
    struct ggml_tensor * X = ggml_new_tensor_2d(c, GGML_TYPE_F32, n, rows);
    memcpy(X->data, x, (size_t) n * rows * 4);
    struct ggml_tensor * Y = ggml_norm(c, X);
    struct ggml_cgraph g = ggml_build_forward(Y);
    g.n_threads = 1;
    ggml_graph_compute(c, &g);
    memcpy(y, Y->data, (size_t) n * rows * 4);
    ggml_free(c);
}

void ref_gelu(const float * x, int64_t n, float * y) {
    struct ggml_context * c = make_ctx((size_t) n * 8 + (16u << 20));
    struct ggml_tensor * X = ggml_new_tensor_1d(c, GGML_TYPE_F32, n);
    memcpy(X->data, x, (size_t) n * 4);
    struct ggml_tensor * Y = ggml_gelu_inplace(c, X);
    struct ggml_cgraph g = ggml_build_forward(Y);
    g.n_threads = 1;
    ggml_graph_compute(c, &g);
    memcpy(y, Y->data, (size_t) n * 4);
    ggml_free(c);
}

/* Falcon's rope call (libfalcon.cpp:2229-2234): x is [head_dim, n_head, N] contiguous, mode 2,
 * dynamic NTK with alpha parameter 2. In place semantic -> result copied to y. */
void ref_rope_falcon(const float * x, int head_dim, int n_head, int N, int n_past, int n_ctx, float * y) {
    const size_t ne = (size_t) head_dim * n_head * N;
    struct ggml_context * c = make_ctx(ne * 8 + (16u << 20));
    struct ggml_tensor * X = ggml_new_tensor_3d(c, GGML_TYPE_F32, head_dim, n_head, N);
    memcpy(X->data, x, ne * 4);
    struct ggml_tensor * Y = ggml_rope_inplace(c, X, n_past, head_dim, 2, n_ctx);
    Y->meta.i_custom[GGML_CUSTOM_I_ROPE_DYNAMIC_MODE] = 1;
    Y->meta.f_custom[GGML_CUSTOM_F_ROPE_NTK_ALPHA]    = 2;
    struct ggml_cgraph g = ggml_build_forward(Y);
    g.n_threads = 1;
    ggml_graph_compute(c, &g);
    memcpy(y, Y->data, ne * 4);
    ggml_free(c);
}

/* rows of KQ [n_kv, N, n_head] -> scale, causal mask (n_past), soft_max; in place chain as
 * libfalcon.cpp:2312-2326 */
void ref_scale_mask_softmax(const float * kq, int n_kv, int N, int n_head, int n_past, float scale, float * out) {
    const size_t ne = (size_t) n_kv * N * n_head;
    struct ggml_context * c = make_ctx(ne * 8 + (16u << 20));
    struct ggml_tensor * X = ggml_new_tensor_3d(c, GGML_TYPE_F32, n_kv, N, n_head);
    memcpy(X->data, kq, ne * 4);
    struct ggml_tensor * S = ggml_scale_inplace(c, X, ggml_new_f32(c, scale));
    struct ggml_tensor * M = ggml_diag_mask_inf_inplace(c, S, n_past);
    struct ggml_tensor * P = ggml_soft_max_inplace(c, M);
    struct ggml_cgraph g = ggml_build_forward(P);
    g.n_threads = 1;
    ggml_graph_compute(c, &g);
    memcpy(out, P->data, ne * 4);
    ggml_free(c);
}

/* ------------------------------------------------------------------------------------------
 * Whole-model harness. The caller hands over a flat description of a (tiny or full size)
 * Falcon model; weights are ggml block bytes exactly as they would sit in a model file.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    int32_t n_vocab, n_embd, n_head, n_head_kv, n_layer, n_ff, n_ctx;
    int32_t wtype;          /* ggml_type of every 2-D weight (tok_embeddings included)            */
    int32_t two_norms;      /* 1 = Falcon-40B style (ln_attn + ln_mlp), 0 = 7B (shared)             */
    int32_t rope_n_ctx;     /* n_ctx argument handed to ggml_rope (n_max_real_ctx or n_ctx)         */
} ref_hparams;

typedef struct {
    const void  * qkv, * wo, * up, * down;      /* quantized                                      */
    const float * ln_w, * ln_b;                 /* input_layernorm (feeds the MLP; also QKV on 7B) */
    const float * ln2_w, * ln2_b;               /* attention_norm (40B only)                       */
} ref_layer;

typedef struct {
    ref_hparams hp;
    const void  * tok_emb;                      /* [n_embd x n_vocab] quantized                    */
    const float * out_norm_w, * out_norm_b;
    const void  * lm_head;                      /* [n_embd x n_vocab] quantized                    */
    const ref_layer * layers;
    float * k_cache;                            /* [n_layer][n_ctx][n_head_kv][64] f32, caller-owned */
    float * v_cache;                            /* same layout                                     */
} ref_model;

static struct ggml_tensor * wrap2d(struct ggml_context * c, int type, int64_t ne0, int64_t ne1, const void * data) {
    struct ggml_tensor * t = ggml_new_tensor_2d(c, (enum ggml_type) type, ne0, ne1);
    t->data = (void *) data;     /* no_alloc context: we only borrow the caller's bytes */
    return t;
}
static struct ggml_tensor * wrap1d(struct ggml_context * c, int64_t ne0, const float * data) {
    struct ggml_tensor * t = ggml_new_tensor_1d(c, GGML_TYPE_F32, ne0);
    t->data = (void *) data;
    return t;
}

static struct ggml_tensor * layer_norm(struct ggml_context * c, struct ggml_tensor * x,
                                       struct ggml_tensor * w, struct ggml_tensor * b) {
    struct ggml_tensor * n = ggml_norm(c, x);
    struct ggml_tensor * s = ggml_mul(c, n, ggml_repeat(c, w, n));
    return ggml_add(c, s, ggml_repeat(c, b, n));
}

/* Evaluate N tokens at position n_past. logits_out receives N * n_vocab floats (all rows).
 * hidden_out (optional) receives the residual stream entering each layer + final: (n_layer+1)*N*n_embd. */
void ref_falcon_eval(const ref_model * m, const int32_t * tokens, int N, int n_past, int n_threads,
                     float * logits_out, float * hidden_out) {
    const ref_hparams * hp = &m->hp;
    const int64_t E = hp->n_embd, H = hp->n_head, HKV = hp->n_head_kv, D = E / H, L = hp->n_layer;
    const int64_t n_ctx = hp->n_ctx;
    const size_t act = (size_t) N * (E * 24 + hp->n_ff * 3 + hp->n_vocab) * 4
                     + (size_t) H * N * (n_past + N) * 8 + (size_t)(n_past + N) * HKV * D * 8 * H;
    struct ggml_context * wc;   /* weights: borrowed pointers */
    {
        struct ggml_init_params ip = { (size_t)(64u << 20), NULL, true };
        wc = ggml_init(ip);
    }
    struct ggml_context * c = make_ctx(act * (size_t)(L + 2) / 2 + act + (256u << 20));

    struct ggml_tensor * kc = ggml_new_tensor_1d(wc, GGML_TYPE_F32, L * n_ctx * HKV * D); kc->data = m->k_cache;
    struct ggml_tensor * vc = ggml_new_tensor_1d(wc, GGML_TYPE_F32, L * n_ctx * HKV * D); vc->data = m->v_cache;

    struct ggml_cgraph gf; memset(&gf, 0, sizeof(gf));
    gf.n_threads = n_threads;

    struct ggml_tensor * embd = ggml_new_tensor_1d(c, GGML_TYPE_I32, N);
    memcpy(embd->data, tokens, (size_t) N * 4);
    struct ggml_tensor * inpL = ggml_get_rows(c, wrap2d(wc, hp->wtype, E, hp->n_vocab, m->tok_emb), embd);

    struct ggml_tensor ** taps = (struct ggml_tensor **) calloc((size_t) L + 1, sizeof(void *));

    for (int il = 0; il < L; ++il) {
        const ref_layer * ly = &m->layers[il];
        taps[il] = inpL;
        struct ggml_tensor * ln_out = layer_norm(c, inpL, wrap1d(wc, E, ly->ln_w), wrap1d(wc, E, ly->ln_b));
        struct ggml_tensor * cur = ln_out;
        if (hp->two_norms) {
            cur = layer_norm(c, inpL, wrap1d(wc, E, ly->ln2_w), wrap1d(wc, E, ly->ln2_b));
        }
        cur = ggml_mul_mat(c, wrap2d(wc, hp->wtype, E, (H + 2 * HKV) * D, ly->qkv), cur);

        const size_t rs = (size_t) D * (H + 2 * HKV) * 4;
        struct ggml_tensor * Qcur = ggml_view_3d(c, cur, D, H,   N, D * 4, rs, 0);
        struct ggml_tensor * Kcur = ggml_view_3d(c, cur, D, HKV, N, D * 4, rs, (size_t) D * H * 4);
        struct ggml_tensor * Vcur = ggml_view_3d(c, cur, D, HKV, N, D * 4, rs, (size_t) D * (H + HKV) * 4);
        Qcur = ggml_rope_inplace(c, Qcur, n_past, D, 2, hp->rope_n_ctx);
        Kcur = ggml_rope_inplace(c, Kcur, n_past, D, 2, hp->rope_n_ctx);
        Qcur->meta.i_custom[GGML_CUSTOM_I_ROPE_DYNAMIC_MODE] = 1;
        Kcur->meta.i_custom[GGML_CUSTOM_I_ROPE_DYNAMIC_MODE] = 1;
        Qcur->meta.f_custom[GGML_CUSTOM_F_ROPE_NTK_ALPHA] = 2;
        Kcur->meta.f_custom[GGML_CUSTOM_F_ROPE_NTK_ALPHA] = 2;

        struct ggml_tensor * kdst = ggml_view_1d(c, kc, N * HKV * D, (size_t) 4 * HKV * D * (il * n_ctx + n_past));
        struct ggml_tensor * vdst = ggml_view_1d(c, vc, N * HKV * D, (size_t) 4 * HKV * D * (il * n_ctx + n_past));
        ggml_build_forward_expand(&gf, ggml_cpy(c, Kcur, kdst));
        ggml_build_forward_expand(&gf, ggml_cpy(c, Vcur, vdst));

        struct ggml_tensor * K = ggml_permute(c,
            ggml_view_3d(c, kc, D, HKV, n_past + N, D * 4, D * HKV * 4, (size_t) il * n_ctx * 4 * HKV * D),
            0, 2, 1, 3);
        struct ggml_tensor * Q  = ggml_permute(c, Qcur, 0, 2, 1, 3);
        struct ggml_tensor * KQ = ggml_mul_mat(c, K, Q);            /* broadcast i02 = i12/(H/HKV), ggml.c:11074 */
        struct ggml_tensor * KQs = ggml_scale_inplace(c, KQ, ggml_new_f32(c, 1.0f / sqrtf((float) D)));
        struct ggml_tensor * KQm = ggml_diag_mask_inf_inplace(c, KQs, n_past);
        struct ggml_tensor * KQp = ggml_soft_max_inplace(c, KQm);
        struct ggml_tensor * V = ggml_cont(c, ggml_permute(c,
            ggml_view_3d(c, vc, D, HKV, n_past + N, D * 4, D * HKV * 4, (size_t) il * n_ctx * 4 * HKV * D),
            1, 2, 0, 3));
        struct ggml_tensor * KQV = ggml_mul_mat(c, V, KQp);
        struct ggml_tensor * merged = ggml_permute(c, KQV, 0, 2, 1, 3);
        cur = ggml_cpy(c, merged, ggml_new_tensor_2d(c, GGML_TYPE_F32, E, N));
        cur = ggml_mul_mat(c, wrap2d(wc, hp->wtype, E, E, ly->wo), cur);
        struct ggml_tensor * attn_out = ggml_cpy(c, cur, ggml_new_tensor_2d(c, GGML_TYPE_F32, E, N));

        cur = ggml_mul_mat(c, wrap2d(wc, hp->wtype, E, hp->n_ff, ly->up), ln_out);
        cur = ggml_gelu_inplace(c, cur);
        cur = ggml_mul_mat(c, wrap2d(wc, hp->wtype, hp->n_ff, E, ly->down), cur);
        cur = ggml_add(c, cur, attn_out);
        cur = ggml_add(c, cur, inpL);
        inpL = cur;
    }
    taps[L] = inpL;
    struct ggml_tensor * cur = layer_norm(c, inpL, wrap1d(wc, E, m->out_norm_w), wrap1d(wc, E, m->out_norm_b));
    cur = ggml_mul_mat(c, wrap2d(wc, hp->wtype, E, hp->n_vocab, m->lm_head), cur);
    ggml_build_forward_expand(&gf, cur);
    ggml_graph_compute(c, &gf);

    memcpy(logits_out, cur->data, (size_t) N * hp->n_vocab * 4);
    if (hidden_out) {
        for (int il = 0; il <= L; ++il) {
            memcpy(hidden_out + (size_t) il * N * E, taps[il]->data, (size_t) N * E * 4);
        }
    }
    free(taps);
    ggml_free(c);
    ggml_free(wc);
}
