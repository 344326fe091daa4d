/*
 * oracle.h -- TEST INFRASTRUCTURE. CPU restatement of the reference's quantized mat-mul + Falcon
 * attention path (cmp-nct/ggllm.cpp: ggml.c, k_quants.c, libfalcon.cpp). Every function cites the
 * reference file:line whose arithmetic it restates. Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load this; the product (ggllm.cpp_amd/) never does.
 *
 * Parity pin: checked in tests/test_oracle_vs_golden.py against golden vectors produced by the
 * REAL reference compiled in the build container (oracle/_ref, oracle/gen_golden.py).
 */
#ifndef ORACLE_H
#define ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ggml_type numbering (ggml.h:247-268) */
enum {
    ORC_F32 = 0, ORC_F16 = 1, ORC_Q4_0 = 2, ORC_Q4_1 = 3, ORC_Q5_0 = 6, ORC_Q5_1 = 7, ORC_Q8_0 = 8, ORC_Q8_1 = 9,
    ORC_Q2_K = 10, ORC_Q3_K = 11, ORC_Q4_K = 12, ORC_Q5_K = 13, ORC_Q6_K = 14, ORC_Q8_K = 15,
};

/* rounding flavour of the Q8_0 / Q8_1 activation quantizers:
 *   0 = "_reference" semantics (ggml.c:1106-1129, 1292-1325): id = 1/d, roundf (half away from zero)
 *   1 = what the reference's AVX2 build executes (ggml.c:1202-1224): id = 127/amax, round-half-even   */
enum { ORC_ROUND_REFERENCE = 0, ORC_ROUND_AVX = 1 };

float    orc_fp16_to_fp32(uint16_t h);
uint16_t orc_fp32_to_fp16(float f);

int    orc_blck_size(int type);           /* elements per block  (ggml.c GGML_BLCK_SIZE)  */
size_t orc_type_size(int type);           /* bytes per block     (ggml.c GGML_TYPE_SIZE)  */
int    orc_vec_dot_type(int wtype);       /* ggml.c:1627-1718                              */
size_t orc_row_bytes(int type, int64_t k);

/* weight quantizers, legacy formats only (k-quant quantizers are a "next" row) */
void orc_quantize_row(int type, const float * x, void * out, int64_t k);
/* activation quantizers Q8_0 / Q8_1 / Q8_K */
void orc_quantize_act(int act_type, const float * x, void * out, int64_t k, int flavour);
void orc_dequantize_row(int type, const void * in, float * y, int64_t k);
float orc_vec_dot(int wtype, int64_t n, const void * w, const void * act);

/* dst[M x N] (column n contiguous, M floats) = W^T x ; ggml.c:11318-11529 */
void orc_mul_mat_q(int wtype, const void * w, int64_t K, int64_t M, const float * x, int64_t N,
                   float * dst, int n_threads, int flavour);

/* ---- Falcon block pieces ---- */
void  orc_tables_init(void);                                   /* ggml.c:4276-4290 */
float orc_gelu(float x);                                       /* ggml.c:3461-3484 */
float orc_exp_f16(float x);                                    /* ggml.c:12436-12442 */
void  orc_norm(const float * x, int64_t n, int64_t rows, float * y);             /* ggml.c:10540-10594 */
void  orc_layer_norm(const float * x, int64_t n, int64_t rows, const float * w, const float * b, float * y);
float orc_rope_theta_scale(int n_dims, int n_ctx);             /* ggml.c:12875-12898 with Falcon's flags */
void  orc_rope_neox(float * x, int head_dim, int n_head, int N, int n_past, int n_ctx);  /* ggml.c:12957-12978 */
void  orc_rope_table(float * cos_sin, int head_dim, int n_pos, int n_ctx);       /* [n_pos][head_dim/2][2] */
void  orc_softmax_rows(float * x, int64_t nc, int64_t nr);                        /* ggml.c:12389-12456 */

typedef struct {
    int32_t n_vocab, n_embd, n_head, n_head_kv, n_layer, n_ff, n_ctx;
    int32_t wtype;
    int32_t two_norms;
    int32_t rope_n_ctx;
} orc_hparams;

typedef struct {
    const void  * qkv, * wo, * up, * down;
    const float * ln_w, * ln_b, * ln2_w, * ln2_b;
} orc_layer;

typedef struct {
    orc_hparams hp;
    const void  * tok_emb;
    const float * out_norm_w, * out_norm_b;
    const void  * lm_head;
    const orc_layer * layers;
    float * k_cache;    /* [n_layer][n_ctx][n_head_kv][head_dim] */
    float * v_cache;
} orc_model;

/* libfalcon.cpp:2011-2588 restated; logits_out: N*n_vocab, hidden_out optional (n_layer+1)*N*n_embd */
void orc_falcon_eval(const orc_model * m, const int32_t * tokens, int N, int n_past, int n_threads,
                     int flavour, float * logits_out, float * hidden_out);

/* block `il` for the tokens sample[0..ns) of a batch of N block-input rows X at positions pos0.. (K / V of the positions
 * before pos0 in k_prev / v_prev, [pos0][n_head_kv][64]); k_out / v_out (optional) receive the batch's N K / V rows */
void orc_falcon_block_sampled(const orc_model * m, int il, const float * X, int N, int pos0, const float * k_prev, const float * v_prev,
                              const int32_t * sample, int ns, int n_threads, int flavour, float * out, float * k_out, float * v_out);
/* ln_f + lm_head of ns residual rows */
void orc_falcon_head_rows(const orc_model * m, const float * X, int ns, int n_threads, int flavour, float * logits);

#ifdef __cplusplus
}
#endif
#endif
