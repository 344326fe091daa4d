// kernels.h -- host-callable launchers of the HIP kernels (internal to libggml_hip.so)
#pragma once
#include <hip/hip_runtime.h>
#include "fq_types.h"
#include <stdio.h>

enum { FQ_EPI_STORE = 0, FQ_EPI_GELU = 1, FQ_EPI_ADD2 = 2 };

struct fq_gemv_epi {
    int              mode;
    const uint16_t * gelu_table;   // 65536 fp16 entries (FQ_EPI_GELU)
    const float    * add1;         // FQ_EPI_ADD2: dst = (v + add1) + add2
    const float    * add2;
    int64_t          ld_add;       // column stride of add1/add2
};

// kernels_quant.hip
void   fq_launch_retile(const uint8_t * src_dev, const fq_weight & w, hipStream_t st);
bool   fq_launch_wquant(int type, const float * x, int64_t n_elems, uint8_t * out, unsigned long long * hist, hipStream_t st);
void   fq_launch_f32_to_f16(const float * src, uint16_t * dst, int64_t n, hipStream_t st);
void   fq_launch_f16_to_f32(const uint16_t * src, float * dst, int64_t n, hipStream_t st);
void   fq_launch_dequant_rows(const fq_weight & w, const int32_t * rows_dev, int64_t nrows, float * dst, hipStream_t st);
void   fq_launch_quantize_act(const float * x, int64_t ldx, const fq_act & a, hipStream_t st);
void   fq_launch_act_export(const fq_act & a, uint8_t * out, hipStream_t st);

// kernels_gemv.hip
#define FQ_GEMV_MAX_COLS 4
size_t fq_gemv_lds_bytes(int act_type, int64_t K, int ncols);
void   fq_launch_gemv(const fq_weight & w, const fq_act & act, int ncols, float * dst, int64_t ldd,
                      const fq_gemv_epi & ep, int max_blocks, hipStream_t st);

// kernels_ref.hip -- any N, the reference's scalar summation order (ggml_hip_reference_order)
void   fq_launch_mul_mat_ref(const fq_weight & w, const fq_act & act, int64_t N, float * dst, int64_t ldd, const fq_gemv_epi & ep, hipStream_t st);

// kernels_kqref.hip -- the k-quants' mat-vec (per column) in the reference's scalar association at wave speed (ggml_hip_reference_order(2)); false: outside its scope
bool   fq_gemv_kq_ref_supported(const fq_weight & w);
bool   fq_launch_gemv_kq_ref(const fq_weight & w, const fq_act & act, int64_t N, float * dst, int64_t ldd, const fq_gemv_epi & ep, hipStream_t st);

// kernels_gemm.hip -- int8 MFMA mat-mul for N > 4 columns
bool   fq_gemm_supported(int type);
void   fq_gemm_debug_mode(int m);       // tuning aid. bit 1: no MFMA / scaling; kernels_gemm_skinny.hip: 4 = no token DMA, 8 = no weight DMA, 16 = no arithmetic
int    fq_gemm_debug_get();
void   fq_gemm_set_sequential(int on);
void   fq_attn_set_f64(int on);
void   fq_attn_set_form(int form);        // prefill attention on the matrix pipe: 0 default, 16 = 16 rows with f32 scores in LDS, 17 = 16 rows with fp16 probabilities in LDS (two workgroups per CU)
int    fq_attn_f64();
void   fq_launch_gemm(const fq_weight & w, const fq_act & act, int64_t N, float * dst, int64_t ldd,
                      const fq_gemv_epi & ep, int n_cu, hipStream_t st);
// kernels_gemm_skinny.hip -- N <= 16 columns of a legacy format at weight-stream speed; S = K split (1, 2, 4: k_gemm_q's association);
// false = outside its scope, nothing launched
bool   fq_launch_gemm_skinny(const fq_weight & w, const fq_act & act, int64_t N, float * dst, int64_t ldd, const fq_gemv_epi & ep, int S, hipStream_t st);
// two matrices of one format and K behind the same columns in ONE launch of the resident form (Wqkv and Wup of a one-norm block); false: nothing launched
bool   fq_launch_gemm_skinny_pair(const fq_weight & w0, const fq_weight & w1, const fq_act & act, int64_t N, float * dst0, int64_t ldd0, const fq_gemv_epi & ep0,
                                  float * dst1, int64_t ldd1, const fq_gemv_epi & ep1, int S, hipStream_t st);
bool   fq_launch_gemm_skinny_q4k_gelu_q8k(const fq_weight & w, const fq_act & act, int64_t N, float * dst, int64_t ldd, const uint16_t * gelu_table, const fq_act & out, hipStream_t st);
bool   fq_launch_gemm_skinny_q4k_out2(const fq_weight & wo, const fq_act & a_att, const fq_weight & down, const fq_act & a_ff, int64_t N, float * x, int64_t ldx, hipStream_t st);
#define FQ_SKINNY_Q4K_MAX_COLS 112                        // .. in passes of 16, at most (fq_skinny_kq_max_cols(type): 80 for Q4_K / Q5_K, 112 for Q2_K; beyond: the tile GEMM)
static inline int fq_skinny_kq_max_cols(int type) { return (type == FQ_Q2_K || type == FQ_Q3_K) ? 112 : 80; }
bool   fq_launch_gemm_skinny_out2(const fq_weight & wo, const fq_act & a_att, const fq_weight & down, const fq_act & a_ff, int64_t N, float * x, int64_t ldx, hipStream_t st);      // legacy formats, N <= 16: Wdown and Wo in one K-share launch + one sum launch
bool   fq_skinny_q4k_shape(const fq_weight & w);          // Q4_K shapes the small-batch form takes (N <= 16: always four partial sums)
int    fq_gemm_wg_rows(int type, int64_t M, int64_t N, int n_cu);        // weight rows per workgroup of the tile form (32 / 64)
int    fq_gemm_split_for(int64_t M, int64_t N, int n_cu);      // the K split (1, 2, 4) fq_launch_gemm gives an M x N result

// kernels_block.hip
void   fq_launch_layer_norm(const float * x, int64_t n, int64_t rows, const float * w, const float * b, float * y, hipStream_t st);
// the LayerNorm(s) + activation image(s) that FOLLOW a block's residual sum (the next block's ln_mlp [+ ln_attn], or the output norm), for fq_launch_add2_ln: one workgroup per
// token sums its row, keeps it in LDS and runs k_layer_norm_quant's device code on it. a0 / a1: images, column 0 first; w1 == nullptr: one norm; a0.type == a1.type
struct fq_next_norm { const float * w0, * b0; fq_act a0; const float * w1, * b1; fq_act a1; };
bool   fq_add2_ln_ok(const fq_next_norm & nn, int64_t E);
void   fq_launch_add2_ln(float * x, const float * a, const float * b, int64_t E, int64_t rows, const fq_next_norm & nn, hipStream_t st);      // x = (a + b) + x, then the norm(s) of x
void   fq_launch_add2_inplace(float * x, const float * a, const float * b, int64_t n, hipStream_t st);      // x = (a + b) + x, n % 4 == 0 (the residual sum of a block whose two branches ran on two streams)
void   fq_launch_layer_norm_quant(const float * x, int64_t n, int64_t rows, const float * w, const float * b, float * y, const fq_act & a, hipStream_t st);
bool   fq_launch_layer_norm_quant2(const float * x, int64_t n, int64_t rows, const float * w0, const float * b0, const fq_act & a0,
                                   const float * w1, const float * b1, const fq_act & a1, hipStream_t st);      // two norms of the same rows, one launch (same image type; else false)
void   fq_launch_gelu(const float * x, float * y, int64_t n, const uint16_t * gelu_table, hipStream_t st);
void   fq_launch_add3(const float * a, const float * b, const float * c, float * y, int64_t n, hipStream_t st);
// qkv: [N][(H+2HKV)*D] fused rows; rotates Q (in place) and K, appends K/V at positions n_past.. of the layer's cache.
// seq_stride > 0 (both launchers): the N rows are N independent sequences, all at position n_past, row t with its own
// cache at k_cache / v_cache + t * seq_stride floats
void   fq_launch_rope_kv(float * qkv, int N, int H, int HKV, int D, const int * n_past_dev, const float * rope_cs,
                         float * k_cache, float * v_cache, hipStream_t st, int64_t seq_stride = 0);
// att[N][H*D] = softmax(mask(K.Q * scale)) V, one workgroup per (head, token)
// own_scratch: the caller's score-row buffer for the long-prompt forms (a model context sizes it once with
// fq_attention_scratch_need(n_batch, H, n_ctx); never grown inside a launch); nullptr: a process-wide buffer grown on demand
// (op-level API only: it synchronizes and reallocates).
struct fq_att_scratch { float * p; size_t bytes; };
size_t fq_attention_scratch_need(int N, int H, int max_n_kv, int HKV = 1);          // 0: no launch of that size uses a scratch
void   fq_launch_attention(const float * qkv, int N, int H, int HKV, int D, const int * n_past_dev, int max_n_kv, const float * k_cache,
                           const float * v_cache, const uint16_t * exp_table, float * att, hipStream_t st, int64_t seq_stride = 0,
                           fq_att_scratch * own_scratch = nullptr);

int    fq_selftest_reduce(hipStream_t st);
int    fq_exp_boundary(const uint16_t * exp_table, unsigned * out_host, int cap, hipStream_t st);   // diagnostic: the inputs the f32 fast path of exp_f16_formula leaves undecided
int    fq_verify_exp_formula(const uint16_t * exp_table, hipStream_t st);   // number of non-NaN inputs where the formula != table   // 0 = DPP wave reductions agree with the __shfl_xor butterfly

// kernels_decode.hip -- fused N = 1 decode kernels
enum { FQ_LNEPI_STORE = 0, FQ_LNEPI_GELU_QUANT = 1, FQ_LNEPI_GELU_STORE = 2 };
struct fq_gemv_ln_seg {
    fq_weight     w;               // K = n_embd rows of this segment
    const float * ln_w, * ln_b;    // LayerNorm feeding the segment
    int           epi;             // FQ_LNEPI_*
    float *       dst;             // f32 output (STORE / GELU_STORE)
    uint8_t *     dst_image;       // Q8_0 / Q8_1 activation image of length w.M (GELU_QUANT)
    int           next_act_type;
    int           block_begin;     // first workgroup of the segment (set by the launcher)
};
struct fq_gemv_ln_args {
    const float * x; int64_t E; int nseg; fq_gemv_ln_seg seg[2]; const uint16_t * gelu_table; long long * dbg;
    float * argmax_val; int * argmax_idx;      // optional (lm_head): per-workgroup best logit and its row, for greedy sampling
    int npass;                                 // 4-row passes per wave (set by the launcher)
    int n_blocks;                              // workgroups of this launch (set by the launcher)
    unsigned * epoch_word;                     // optional: the hand-off tag of the k_attn_out that follows; this launch increments it (never 0)
    // optional: copy the rope table's row of the current position (cos/sin pairs, 64 floats) to rope_cur, so that the
    // attention that follows does not have to wait for n_past before it can ask for them
    const int * n_past_ptr; const float * rope_cs; float * rope_cur;
};
struct fq_gemv_out_args {
    fq_weight w_down, w_wo;
    const uint8_t * act_ff_image;  // quantized gelu(up), image of length w_down.K
    const float *   att;           // f32 attention output, quantized in the prologue (used when att_image == nullptr)
    const uint8_t * att_image;     // attention output already quantized by k_attn_decode (Q8_0 / Q8_1), or nullptr
    const float *   resid;         // residual stream (may alias dst)
    float *         dst;
    long long *     dbg;           // optional phase stamps (wall_clock64), 8 per workgroup
};
size_t fq_gemv_ln_lds(int type, int64_t E);
size_t fq_attn_decode_lds_bytes(int max_n_kv);          // LDS of one decode-attention head group whose score row holds max_n_kv keys (<= 160 KiB to launch)
// ref (all the fused decode launchers): the fast reference order (fq_ref_chain.h, legacy formats): each row's block terms added left to right, the decode
// attention's dots accumulated in f64 -- results bit-identical to the reference's scalar build
void   fq_launch_gemv_ln(fq_gemv_ln_args a, int n_cu, hipStream_t st, bool ref = false);       // fills seg[].block_begin
void   fq_launch_gemv_out(const fq_gemv_out_args & a, int n_cu, hipStream_t st, bool ref = false);
// attention + output mat-vec in one launch (k_attn_out); returns false (nothing launched) when the grid would not be
// resident at once -- the caller then uses fq_launch_attn_decode + fq_launch_gemv_out. gran: >= n_embd granules (8 bytes
// each), zero-filled once; epoch_word: incremented by the k_gemv_ln launch (or phase) before it. With ln != nullptr the launch
// is k_attn_out_ln: *ln (the next block's k_gemv_ln, or ln_f + lm_head) runs as a second phase of the same launch and takes
// the residual row from xgran (>= n_embd granules, zero-filled once) instead of memory.
bool   fq_attn_out_fits(const fq_gemv_out_args & g, int H, int max_n_kv, int n_cu);      // would fq_launch_attn_out (ln == nullptr) launch?
bool   fq_launch_attn_out(const fq_gemv_out_args & g, const float * qkv, int H, int HKV, const int * n_past_dev, int max_n_kv,
                          const float * rope_cs, const float * rope_cur, float * k_cache, float * v_cache, const uint16_t * exp_table,
                          int att_act_type, unsigned long long * gran, const unsigned * epoch_word, unsigned * err, int n_cu, hipStream_t st,
                          const fq_gemv_ln_args * ln = nullptr, unsigned long long * xgran = nullptr);
void   fq_launch_attn_decode(const float * qkv, int H, int HKV, const int * n_past_dev, int max_n_kv, const float * rope_cs,
                             float * k_cache, float * v_cache, const uint16_t * exp_table, float * att, uint8_t * att_image,
                             int att_act_type, hipStream_t st, bool f64 = false);
// k_attn_out in the fast reference order (legacy formats; f64 attention dots, rows summed left to right); false: nothing launched
bool   fq_launch_attn_out_ref(const fq_gemv_out_args & g, const float * qkv, int H, int HKV, const int * n_past_dev, int max_n_kv,
                              const float * rope_cs, const float * rope_cur, float * k_cache, float * v_cache, const uint16_t * exp_table,
                              int att_act_type, unsigned long long * gran, const unsigned * epoch_word, unsigned * err, int n_cu, hipStream_t st);
// B lock-step sequences: row t of qkv / att, KV cache t (seq_stride floats apart), image column t (image_stride bytes apart)
void   fq_launch_attn_decode_seqs(const float * qkv, int n_seq, int H, int HKV, const int * n_past_dev, int max_n_kv, const float * rope_cs,
                                  float * k_cache, float * v_cache, int64_t seq_stride, const uint16_t * exp_table, float * att, uint8_t * att_image,
                                  int att_act_type, int64_t image_stride, hipStream_t st, const float * qx = nullptr, int64_t q_ldx = 0, const fq_act * qa = nullptr);
// (qx / qa: optional rider -- the f32 matrix qx [n_seq][q_ldx] is quantized into the image qa by extra workgroups of the same launch)

// kernels_ring.hip -- the ring form of k_gemv_ln's launch (LDS-DMA loader wave + consumers out of an LDS ring, one workgroup per CU);
// false = outside its scope, nothing launched. fq_ring_prepare: builds the shape's schedule (allocates: not inside a stream capture)
bool   fq_ring_prepare(int type, int64_t E, int64_t FF, int64_t qkv_rows, int n_cu);
// ref: the fast reference order (fq_ref_chain.h): every row's block terms added left to right as the reference's scalar build adds them
bool   fq_launch_gemv_ln_ring(const fq_gemv_ln_args & g, unsigned * err, int n_cu, hipStream_t st, bool ref = false);

// kernels_ringk.hip -- ring forms beyond kernels_ring.hip's scope: k_gemv_ln's launch for the k-quants (GELU_STORE epilogue: the Q8_K image of gelu(up)
// rides on the attention launch), and k_gemv_out's launch for all ten formats. false = outside the form's scope (or no prepared schedule): nothing launched.
bool   fq_ringk_prepare(int type, int64_t E, int64_t FF, int64_t qkv_rows, int n_cu);      // builds the shape's schedule (allocates: not inside a stream capture)
bool   fq_launch_ringk_ln(const fq_gemv_ln_args & g, unsigned * err, int n_cu, hipStream_t st);
bool   fq_launch_ring_out(const fq_gemv_out_args & g, unsigned * err, int n_cu, hipStream_t st);
void   fq_ringk_free_plans();
void   fq_ring_free_plans();

// kernels_cols.hip -- the two mat-vec launches of a block for 2..4 lock-step sequences (one weight pass serves all columns)
struct fq_gemv_cols_seg {
    fq_weight       w;
    const uint8_t * act;           // ncols quantized activation columns of length w.K (column stride fq_act_col_bytes)
    int             epi;           // FQ_LNEPI_*
    float *         dst; int64_t ldd;      // f32 output, column c at dst + c * ldd (STORE / GELU_STORE)
    uint8_t *       dst_image;     // Q8_0 / Q8_1 image columns of length w.M (GELU_QUANT)
    int             next_act_type;
    int             block_begin;   // set by the launcher
};
struct fq_gemv_cols_args { fq_gemv_cols_seg seg[2]; int nseg, ncols; const uint16_t * gelu_table; int npass; };
struct fq_gemv_out_cols_args {
    fq_weight w_down, w_wo;
    const uint8_t * act_ff_image, * att_image;     // ncols columns each
    const float * resid; float * dst; int64_t ld;  // residual rows [ncols][ld] (dst may alias resid)
    int ncols;
};
bool   fq_launch_gemv_cols(fq_gemv_cols_args a, int n_cu, hipStream_t st);                 // false: outside its scope, nothing launched
bool   fq_launch_gemv_out_cols(const fq_gemv_out_cols_args & a, int n_cu, hipStream_t st);
int    fq_gemv_out_cols_width(int type, int64_t K_down, int64_t K_wo);      // columns per launch that fit its LDS: 4, 2 or 0


// ---- per-launch timing table (round 5; takes the place of the reference's --debug-timings node table, libfalcon.cpp:2506-2520 / ggml.c:18266-18360, on the
// resident path, where there is no ggml graph whose nodes could be listed). Every fq_launch_* opens a scope; while a table is being collected (fq_tl_begin ..
// fq_tl_end) the outermost scope of a launch site brackets its launches with two events on its stream. Inactive: one load and a branch per launch site.
struct fq_tl_scope {
    hipStream_t st; int slot;
    fq_tl_scope(hipStream_t st_, const char * name);
    ~fq_tl_scope();
};
#define FQ_TL(st_, name_) fq_tl_scope fq_tl_scope_((st_), (name_))
void fq_tl_begin();                                   // start collecting (plain launches only: the caller must not replay a captured graph meanwhile)
int  fq_tl_end(FILE * out, const char * title);       // synchronises, prints one line per launch site kind (calls, total / average microseconds, share), returns the number of brackets
bool fq_tl_collecting();

// one streaming workgroup of the ring forms (kernels_ring.hip / kernels_ringk.hip): rows [qg0, qg1) of Wqkv, 32-row groups [ug0, ug1) of Wup (r*, hg*: rows of the output form)
#include <vector>
struct fq_engine_sched { int qg0, qg1, ug0, ug1, r0, r1, hg0, hg1; };
