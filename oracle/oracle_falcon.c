/*
 * oracle_falcon.c -- TEST INFRASTRUCTURE (see oracle.h). CPU restatement of the non-mat-mul ops of a
 * Falcon decoder block and of falcon_eval_internal's op sequence (libfalcon.cpp:2115-2466).
 */
#include "oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ fp16 lookup tables (ggml.c:4276-4290) */
static uint16_t g_gelu_tab[1 << 16];
static uint16_t g_exp_tab[1 << 16];
static int      g_tab_ready = 0;

static float gelu_exact(float x) {                 /* ggml.c:3465-3467 */
    const float a = 0.044715f, s2pi = 0.79788456080286535587989211986876f;
    return 0.5f * x * (1.0f + tanhf(s2pi * x * (1.0f + a * x * x)));
}

void orc_tables_init(void) {
    if (g_tab_ready) return;
    for (uint32_t i = 0; i < (1u << 16); ++i) {
        const float f = orc_fp16_to_fp32((uint16_t) i);
        g_gelu_tab[i] = orc_fp32_to_fp16(gelu_exact(f));
        g_exp_tab[i]  = orc_fp32_to_fp16(expf(f));
    }
    g_tab_ready = 1;
}

const uint16_t * orc_gelu_table(void) { orc_tables_init(); return g_gelu_tab; }
const uint16_t * orc_exp_table(void)  { orc_tables_init(); return g_exp_tab; }

float orc_gelu(float x)    { orc_tables_init(); return orc_fp16_to_fp32(g_gelu_tab[orc_fp32_to_fp16(x)]); }   /* ggml.c:3477-3484 */
float orc_exp_f16(float x) { orc_tables_init(); return orc_fp16_to_fp32(g_exp_tab[orc_fp32_to_fp16(x)]); }    /* ggml.c:12436-12442 */

int orc_exact_f64(void);          /* the f64 yardstick (oracle_quants.c, orc_set_sum_order(6)): products formed in f64 as well */
/* ------------------------------------------------------------------ norm (ggml.c:10540-10594) */
void orc_norm(const float * x, int64_t n, int64_t rows, float * y) {
    for (int64_t r = 0; r < rows; ++r, x += n, y += n) {
        double sum = 0.0;
        for (int64_t i = 0; i < n; ++i) sum += (double) x[i];
        const float mean = (float)(sum / (double) n);
        double sum2 = 0.0;
        for (int64_t i = 0; i < n; ++i) {
            const float v = x[i] - mean;
            y[i] = v;
            sum2 += orc_exact_f64() ? (double) v * (double) v : (double)(v * v);
        }
        const float variance = (float)(sum2 / (double) n);
        const float scale = 1.0f / sqrtf(variance + 1e-5f);
        for (int64_t i = 0; i < n; ++i) y[i] *= scale;
    }
}

/* norm -> * weight -> + bias  (libfalcon.cpp:2166-2188) */
void orc_layer_norm(const float * x, int64_t n, int64_t rows, const float * w, const float * b, float * y) {
    orc_norm(x, n, rows, y);
    for (int64_t r = 0; r < rows; ++r)
        for (int64_t i = 0; i < n; ++i) y[r * n + i] = y[r * n + i] * w[i] + b[i];
}

/* ------------------------------------------------------------------ RoPE, NeoX pairing, dynamic NTK */
/* ggml.c:12875-12898 with DYNAMIC_MODE = 1 and NTK_ALPHA = 2 (libfalcon.cpp:2231-2234). n_ctx/2048 is an
 * INTEGER division in the reference. */
float orc_rope_theta_scale(int n_dims, int n_ctx) {
    float alpha = 1.0f;
    if (n_ctx >= 2048) alpha = powf((float)(((n_ctx / 2048) - 1) * 2.0f + 1), (float)(n_dims / (n_dims - 2.0)));
    return powf(alpha * 10000.0f, -2.0f / (float) n_dims);
}

/* x: [head_dim][n_head][N] contiguous, rotated in place; position of token t is n_past + t.
 * theta is advanced by repeated f32 multiplication exactly as ggml.c:12962-12966. */
void orc_rope_neox(float * x, int head_dim, int n_head, int N, int n_past, int n_ctx) {
    const float ts = orc_rope_theta_scale(head_dim, n_ctx);
    const int half = head_dim / 2;
    for (int t = 0; t < N; ++t) {
        for (int h = 0; h < n_head; ++h) {
            float * v = x + ((size_t) t * n_head + h) * head_dim;
            float theta = (float)(n_past + t);
            for (int k = 0; k < half; ++k) {
                const float c = cosf(theta), s = sinf(theta);
                theta *= ts;
                const float x0 = v[k], x1 = v[k + half];
                v[k]        = x0 * c - x1 * s;
                v[k + half] = x0 * s + x1 * c;
            }
        }
    }
}

/* cos/sin table the product precomputes on the host: entry [p][k] = {cosf(theta_pk), sinf(theta_pk)} */
void orc_rope_table(float * cs, int head_dim, int n_pos, int n_ctx) {
    const float ts = orc_rope_theta_scale(head_dim, n_ctx);
    const int half = head_dim / 2;
    for (int p = 0; p < n_pos; ++p) {
        float theta = (float) p;
        for (int k = 0; k < half; ++k) {
            cs[((size_t) p * half + k) * 2 + 0] = cosf(theta);
            cs[((size_t) p * half + k) * 2 + 1] = sinf(theta);
            theta *= ts;
        }
    }
}

/* ------------------------------------------------------------------ soft_max rows (ggml.c:12389-12456) */
void orc_softmax_rows(float * x, int64_t nc, int64_t nr) {
    orc_tables_init();
    for (int64_t r = 0; r < nr; ++r, x += nc) {
        float mx = -INFINITY;
        for (int64_t i = 0; i < nc; ++i) if (x[i] > mx) mx = x[i];
        double sum = 0.0;
        for (int64_t i = 0; i < nc; ++i) {
            if (x[i] == -INFINITY) { x[i] = 0.0f; continue; }
            const float v = orc_fp16_to_fp32(g_exp_tab[orc_fp32_to_fp16(x[i] - mx)]);
            sum += (double) v;
            x[i] = v;
        }
        const float inv = (float)(1.0 / sum);           /* ggml_vec_scale_f32(nc, dp, sum) takes a float */
        for (int64_t i = 0; i < nc; ++i) x[i] *= inv;
    }
}

/* ------------------------------------------------------------------ whole model */
/* ggml_vec_dot_f32, portable branch (ggml.c:2296-2300): f32 products accumulated in double, left to right.
 * (The SIMD branches, ggml.c:2270-2294, keep 32 f32 partial sums instead; only the association differs.) */
static float dot_f32(const float * a, const float * b, int64_t n, int64_t stride_a) {
    double s = 0.0;
    if (orc_exact_f64()) { for (int64_t i = 0; i < n; ++i) s += (double) a[i * stride_a] * (double) b[i]; return (float) s; }
    for (int64_t i = 0; i < n; ++i) s += (double)(a[i * stride_a] * b[i]);
    return (float) s;
}

/* The HIP backend's default attention arithmetic (ggllm.cpp_amd/csrc/fq_attn_dev.h, F64 = false): f32 fused multiply-add
 * chains like the reference's SIMD builds. K.Q: lane s (0..7) owns dims 4s..4s+3 and 32+4s..32+4s+3 in that order, the
 * eight partial sums are added as ((p0+p1)+(p2+p3)) + ((p4+p5)+(p6+p7)). V.P: one chain per class j mod 16 over
 * increasing j, the 16 classes added in order. Selected by orc_set_sum_order(2) ("as the backend"). */
int orc_attn_backend_order(void);
int orc_backend_batch(void);      /* > 0: the rows at hand stand for a batch of this many tokens (oracle_quants.c) */
static float dot_qk_backend(const float * k, const float * q) {
    float part[8];
    for (int s = 0; s < 8; ++s) {
        const float * a = k + 4 * s, * b = q + 4 * s;
        float v = a[0] * b[0];
        v = fmaf(a[1], b[1], v); v = fmaf(a[2], b[2], v); v = fmaf(a[3], b[3], v);
        v = fmaf(a[32], b[32], v); v = fmaf(a[33], b[33], v); v = fmaf(a[34], b[34], v); v = fmaf(a[35], b[35], v);
        part[s] = v;
    }
    return ((part[0] + part[1]) + (part[2] + part[3])) + ((part[4] + part[5]) + (part[6] + part[7]));
}
static float dot_pv_backend(const float * v, const float * p, int64_t n, int64_t stride_v) {
    float part[16];
    for (int r = 0; r < 16; ++r) part[r] = 0.0f;
    for (int64_t j = 0; j < n; ++j) part[j & 15] = fmaf(v[j * stride_v], p[j], part[j & 15]);
    float o = part[0];
    for (int r = 1; r < 16; ++r) o = o + part[r];
    return o;
}

/* The backend's prefill attention for N >= 32 tokens runs on the f32 matrix pipe (k_attention_mfma, kernels_block.hip):
 * v_mfma_f32_32x32x2_f32 is one fused multiply-add per k step and output element, k ascending (measured bit for bit,
 * scripts/microbench/mb_mfma_f32.hip), so the association is sequential:
 *   K.Q  one chain over s = 0..31 of dims s, 32 + s
 *   V.P  two chains, over the even and the odd 32-key tiles (keys 32 T .. 32 T + 31, inside a tile s = 0..15 of keys s, 16 + s),
 *        added at the end; a masked key has weight 0 and leaves its chain unchanged (the chain never holds -0) */
#define ORC_ATTN_MFMA_MIN_N 32
static float dot_qk_mfma(const float * k, const float * q) {
    float c = 0.0f;
    for (int s = 0; s < 32; ++s) { c = fmaf(q[s], k[s], c); c = fmaf(q[32 + s], k[32 + s], c); }
    return c;
}
static float dot_pv_mfma(const float * v, const float * p, int64_t n, int64_t stride_v) {
    float c[2] = { 0.0f, 0.0f };
    for (int64_t t0 = 0; t0 < n; t0 += 32) {
        float * a = &c[(t0 >> 5) & 1];
        for (int s = 0; s < 16; ++s) {
            if (t0 + s < n)      *a = fmaf(p[t0 + s], v[(t0 + s) * stride_v], *a);
            if (t0 + 16 + s < n) *a = fmaf(p[t0 + 16 + s], v[(t0 + 16 + s) * stride_v], *a);
        }
    }
    return c[0] + c[1];
}

void orc_falcon_eval(const orc_model * m, const int32_t * tokens, int N, int n_past, int n_threads,
                     int flavour, float * logits_out, float * hidden_out) {
    orc_tables_init();
    const orc_hparams * hp = &m->hp;
    const int64_t E = hp->n_embd, H = hp->n_head, HKV = hp->n_head_kv, D = E / H, L = hp->n_layer, FF = hp->n_ff;
    const int64_t QKV = (H + 2 * HKV) * D;
    const int64_t n_kv = n_past + N;
    const int group = (int)(H / HKV);                      /* i02 = i12 / (ne12/ne02), ggml.c:11074 */
    const size_t erow = orc_row_bytes(hp->wtype, E);

    float * inp   = (float *) malloc(sizeof(float) * N * E);
    float * ln    = (float *) malloc(sizeof(float) * N * E);
    float * ln2   = (float *) malloc(sizeof(float) * N * E);
    float * qkv   = (float *) malloc(sizeof(float) * N * QKV);
    float * qrot  = (float *) malloc(sizeof(float) * N * H * D);
    float * krot  = (float *) malloc(sizeof(float) * N * HKV * D);
    float * att   = (float *) malloc(sizeof(float) * N * E);
    float * wo    = (float *) malloc(sizeof(float) * N * E);
    float * up    = (float *) malloc(sizeof(float) * N * FF);
    float * down  = (float *) malloc(sizeof(float) * N * E);
    float * p     = (float *) malloc(sizeof(float) * n_kv);

    /* embedding lookup = dequantize_row of the token's row (ggml_compute_forward_get_rows_q, ggml.c:11975) */
    for (int t = 0; t < N; ++t)
        orc_dequantize_row(hp->wtype, (const uint8_t *) m->tok_emb + (size_t) tokens[t] * erow, inp + (size_t) t * E, E);

    for (int il = 0; il < L; ++il) {
        const orc_layer * ly = &m->layers[il];
        if (hidden_out) memcpy(hidden_out + (size_t) il * N * E, inp, sizeof(float) * N * E);

        orc_layer_norm(inp, E, N, ly->ln_w, ly->ln_b, ln);
        const float * attn_in = ln;
        if (hp->two_norms) { orc_layer_norm(inp, E, N, ly->ln2_w, ly->ln2_b, ln2); attn_in = ln2; }

        orc_mul_mat_q(hp->wtype, ly->qkv, E, QKV, attn_in, N, qkv, n_threads, flavour);

        /* split fused QKV row: [n_head Q heads | n_head_kv K heads | n_head_kv V heads] (libfalcon.cpp:2205-2227) */
        float * kc = m->k_cache + (size_t) il * hp->n_ctx * HKV * D;
        float * vc = m->v_cache + (size_t) il * hp->n_ctx * HKV * D;
        for (int t = 0; t < N; ++t) {
            memcpy(qrot + (size_t) t * H * D,   qkv + (size_t) t * QKV,               sizeof(float) * H * D);
            memcpy(krot + (size_t) t * HKV * D, qkv + (size_t) t * QKV + H * D,       sizeof(float) * HKV * D);
            memcpy(vc + (size_t)(n_past + t) * HKV * D, qkv + (size_t) t * QKV + (H + HKV) * D, sizeof(float) * HKV * D);
        }
        orc_rope_neox(qrot, (int) D, (int) H,   N, n_past, hp->rope_n_ctx);
        orc_rope_neox(krot, (int) D, (int) HKV, N, n_past, hp->rope_n_ctx);
        memcpy(kc + (size_t) n_past * HKV * D, krot, sizeof(float) * N * HKV * D);      /* libfalcon.cpp:2238-2244 */

        const float kq_scale = 1.0f / sqrtf((float) D);
        const int backend_attn = orc_attn_backend_order() && D == 64;
        const int mfma_attn = backend_attn && N >= ORC_ATTN_MFMA_MIN_N;
        for (int t = 0; t < N; ++t) {
            for (int h = 0; h < H; ++h) {
                const int hk = h / group;
                const float * q = qrot + ((size_t) t * H + h) * D;
                for (int64_t s = 0; s < n_kv; ++s) {
                    float v = (mfma_attn ? dot_qk_mfma(kc + ((size_t) s * HKV + hk) * D, q)
                               : backend_attn ? dot_qk_backend(kc + ((size_t) s * HKV + hk) * D, q)
                                            : dot_f32(kc + ((size_t) s * HKV + hk) * D, q, D, 1)) * kq_scale;  /* K.Q then scale */
                    if (s > n_past + t) v = -INFINITY;                                    /* ggml.c:12341-12347 */
                    p[s] = v;
                }
                orc_softmax_rows(p, n_kv, 1);
                float * o = att + (size_t) t * E + (size_t) h * D;                        /* merged [n_embd, N] */
                for (int64_t d = 0; d < D; ++d) {
                    /* (masked keys carry p = 0: the chains of the matrix-pipe form run over the visible keys only) */
                    o[d] = mfma_attn ? dot_pv_mfma(vc + (size_t) hk * D + d, p, n_past + t + 1, HKV * D)
                         : backend_attn ? dot_pv_backend(vc + (size_t) hk * D + d, p, n_kv, HKV * D)
                                        : dot_f32(vc + (size_t) hk * D + d, p, n_kv, HKV * D);    /* V^T row . P row */
                }
            }
        }
        orc_mul_mat_q(hp->wtype, ly->wo, E, E, att, N, wo, n_threads, flavour);

        orc_mul_mat_q(hp->wtype, ly->up, E, FF, ln, N, up, n_threads, flavour);
        for (int64_t i = 0; i < (int64_t) N * FF; ++i) up[i] = orc_gelu(up[i]);
        orc_mul_mat_q(hp->wtype, ly->down, FF, E, up, N, down, n_threads, flavour);

        /* cur = (mlp + attn) + inpL  (libfalcon.cpp:2399-2400) */
        for (int64_t i = 0; i < (int64_t) N * E; ++i) inp[i] = (down[i] + wo[i]) + inp[i];
    }
    if (hidden_out) memcpy(hidden_out + (size_t) L * N * E, inp, sizeof(float) * N * E);

    orc_layer_norm(inp, E, N, m->out_norm_w, m->out_norm_b, ln);
    orc_mul_mat_q(hp->wtype, m->lm_head, E, hp->n_vocab, ln, N, logits_out, n_threads, flavour);

    free(inp); free(ln); free(ln2); free(qkv); free(qrot); free(krot); free(att); free(wo); free(up); free(down); free(p);
}

/* ------------------------------------------------------------------ one block, SAMPLED tokens (full-size parity tests)
 * The block `il` of libfalcon.cpp:2160-2400 for the tokens sample[0..ns) of a batch whose N block-input rows X (positions
 * pos0 .. pos0 + N - 1, the context before them empty unless the K / V of earlier positions are passed in kv_prev) are known:
 * K / V of ALL N tokens (the mat-mul restricted to the K and V rows of Wqkv), everything else for the sampled tokens only.
 * Exactly the arithmetic of orc_falcon_eval for those tokens -- the masked keys of the full evaluation contribute
 * soft_max weight 0 and are left out. out: ns rows of n_embd floats (the block's output = next block's input).
 * A 2048-token prompt through a Falcon-40B-sized block is ~1.4e12 multiply-adds; this is ~2e10. */
typedef struct { const orc_model * m; int il, N, pos0, ns; const int32_t * sample; const float * q, * kc, * vc; float * att; int ith, nth; } att_job;
static void * att_worker(void * arg) {
    const att_job * j = (const att_job *) arg;
    const orc_hparams * hp = &j->m->hp;
    const int64_t E = hp->n_embd, H = hp->n_head, HKV = hp->n_head_kv, D = E / H;
    const int group = (int)(H / HKV);
    const float kq_scale = 1.0f / sqrtf((float) D);
    const int backend_attn = orc_attn_backend_order() && D == 64;
    const int mfma_attn = backend_attn && (orc_backend_batch() > 0 ? orc_backend_batch() : j->N) >= ORC_ATTN_MFMA_MIN_N;
    float * p = (float *) malloc(sizeof(float) * (size_t)(j->pos0 + j->N));
    for (int64_t w = j->ith; w < (int64_t) j->ns * H; w += j->nth) {
        const int si = (int)(w / H), h = (int)(w % H), hk = h / group;
        const int64_t n_kv = (int64_t) j->pos0 + j->sample[si] + 1;
        const float * q = j->q + ((size_t) si * H + h) * D;
        for (int64_t s = 0; s < n_kv; ++s)
            p[s] = (mfma_attn ? dot_qk_mfma(j->kc + ((size_t) s * HKV + hk) * D, q)
                    : backend_attn ? dot_qk_backend(j->kc + ((size_t) s * HKV + hk) * D, q)
                                 : dot_f32(j->kc + ((size_t) s * HKV + hk) * D, q, D, 1)) * kq_scale;
        orc_softmax_rows(p, n_kv, 1);
        float * o = j->att + (size_t) si * E + (size_t) h * D;
        for (int64_t d = 0; d < D; ++d)
            o[d] = mfma_attn ? dot_pv_mfma(j->vc + (size_t) hk * D + d, p, n_kv, HKV * D)
                 : backend_attn ? dot_pv_backend(j->vc + (size_t) hk * D + d, p, n_kv, HKV * D)
                                : dot_f32(j->vc + (size_t) hk * D + d, p, n_kv, HKV * D);
    }
    free(p);
    return NULL;
}

void orc_falcon_block_sampled(const orc_model * m, int il, const float * X, int N, int pos0, const float * k_prev, const float * v_prev,
                              const int32_t * sample, int ns, int n_threads, int flavour, float * out, float * k_out, float * v_out) {
    orc_tables_init();
    const orc_hparams * hp = &m->hp;
    const orc_layer * ly = &m->layers[il];
    const int64_t E = hp->n_embd, H = hp->n_head, HKV = hp->n_head_kv, D = E / H, FF = hp->n_ff;
    const size_t wrow = orc_row_bytes(hp->wtype, E);
    const int64_t KV = HKV * D;

    float * ln_all = (float *) malloc(sizeof(float) * (size_t) N * E);                 /* the norm that feeds Wqkv, all tokens */
    orc_layer_norm(X, E, N, hp->two_norms ? ly->ln2_w : ly->ln_w, hp->two_norms ? ly->ln2_b : ly->ln_b, ln_all);
    /* K and V rows of the fused matrix: rows [H*D, (H + 2 HKV)*D) (libfalcon.cpp:2205-2227) */
    float * kvrows = (float *) malloc(sizeof(float) * (size_t) N * 2 * KV);
    orc_mul_mat_q(hp->wtype, (const uint8_t *) ly->qkv + (size_t)(H * D) * wrow, E, 2 * KV, ln_all, N, kvrows, n_threads, flavour);
    float * kc = (float *) malloc(sizeof(float) * (size_t)(pos0 + N) * KV);
    float * vc = (float *) malloc(sizeof(float) * (size_t)(pos0 + N) * KV);
    if (pos0 > 0) { memcpy(kc, k_prev, sizeof(float) * (size_t) pos0 * KV); memcpy(vc, v_prev, sizeof(float) * (size_t) pos0 * KV); }
    for (int t = 0; t < N; ++t) {
        memcpy(kc + (size_t)(pos0 + t) * KV, kvrows + (size_t) t * 2 * KV,      sizeof(float) * KV);
        memcpy(vc + (size_t)(pos0 + t) * KV, kvrows + (size_t) t * 2 * KV + KV, sizeof(float) * KV);
    }
    orc_rope_neox(kc + (size_t) pos0 * KV, (int) D, (int) HKV, N, pos0, hp->rope_n_ctx);
    if (k_out) memcpy(k_out, kc + (size_t) pos0 * KV, sizeof(float) * (size_t) N * KV);
    if (v_out) memcpy(v_out, vc + (size_t) pos0 * KV, sizeof(float) * (size_t) N * KV);

    /* sampled tokens */
    float * xs   = (float *) malloc(sizeof(float) * (size_t) ns * E);
    float * lnq  = (float *) malloc(sizeof(float) * (size_t) ns * E);
    float * lnm  = (float *) malloc(sizeof(float) * (size_t) ns * E);
    float * q    = (float *) malloc(sizeof(float) * (size_t) ns * H * D);
    float * att  = (float *) malloc(sizeof(float) * (size_t) ns * E);
    float * wo   = (float *) malloc(sizeof(float) * (size_t) ns * E);
    float * up   = (float *) malloc(sizeof(float) * (size_t) ns * FF);
    float * down = (float *) malloc(sizeof(float) * (size_t) ns * E);
    for (int i = 0; i < ns; ++i) {
        memcpy(xs  + (size_t) i * E, X + (size_t) sample[i] * E,      sizeof(float) * E);
        memcpy(lnq + (size_t) i * E, ln_all + (size_t) sample[i] * E, sizeof(float) * E);
    }
    orc_layer_norm(xs, E, ns, ly->ln_w, ly->ln_b, lnm);                                  /* the norm that feeds the MLP */
    orc_mul_mat_q(hp->wtype, ly->qkv, E, H * D, lnq, ns, q, n_threads, flavour);         /* Q rows */
    for (int i = 0; i < ns; ++i) orc_rope_neox(q + (size_t) i * H * D, (int) D, (int) H, 1, pos0 + sample[i], hp->rope_n_ctx);

    if (n_threads < 1) n_threads = 1;
    att_job   * jobs = (att_job *) malloc(sizeof(att_job) * (size_t) n_threads);
    pthread_t * th   = (pthread_t *) malloc(sizeof(pthread_t) * (size_t) n_threads);
    for (int t = 0; t < n_threads; ++t) {
        jobs[t] = (att_job){ m, il, N, pos0, ns, sample, q, kc, vc, att, t, n_threads };
        if (t > 0) pthread_create(&th[t], NULL, att_worker, &jobs[t]);
    }
    att_worker(&jobs[0]);
    for (int t = 1; t < n_threads; ++t) pthread_join(th[t], NULL);
    free(th); free(jobs);

    orc_mul_mat_q(hp->wtype, ly->wo, E, E, att, ns, wo, n_threads, flavour);
    orc_mul_mat_q(hp->wtype, ly->up, E, FF, lnm, ns, up, n_threads, flavour);
    for (int64_t i = 0; i < (int64_t) ns * FF; ++i) up[i] = orc_gelu(up[i]);
    orc_mul_mat_q(hp->wtype, ly->down, FF, E, up, ns, down, n_threads, flavour);
    for (int64_t i = 0; i < (int64_t) ns * E; ++i) out[i] = (down[i] + wo[i]) + xs[i];

    free(ln_all); free(kvrows); free(kc); free(vc); free(xs); free(lnq); free(lnm); free(q); free(att); free(wo); free(up); free(down);
}

/* ln_f + lm_head for given residual rows (libfalcon.cpp:2421-2440) */
void orc_falcon_head_rows(const orc_model * m, const float * X, int ns, int n_threads, int flavour, float * logits) {
    const orc_hparams * hp = &m->hp;
    float * ln = (float *) malloc(sizeof(float) * (size_t) ns * hp->n_embd);
    orc_layer_norm(X, hp->n_embd, ns, m->out_norm_w, m->out_norm_b, ln);
    orc_mul_mat_q(hp->wtype, m->lm_head, hp->n_embd, hp->n_vocab, ln, ns, logits, n_threads, flavour);
    free(ln);
}
