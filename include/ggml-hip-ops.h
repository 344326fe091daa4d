/*
 * ggml-hip-ops.h -- C ABI of libggml_hip.so, operator level. Plain pointers and sizes only.
 *
 * Each entry point names the reference operator it replaces (file:line in cmp-nct/ggllm.cpp). Pointers called
 * *_dev are device (HBM) addresses obtained from ggml_hip_malloc; everything runs on the library's stream
 * (ggml_hip_stream) unless stated. Errors are fatal (message + exit(1)), the reference backend's convention
 * (CUDA_CHECK, ggml-cuda.cu:22-51). There is NO CPU fallback: without a HIP device every call aborts.
 */
#ifndef GGML_HIP_OPS_H
#define GGML_HIP_OPS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ggml_hip_weight ggml_hip_weight;   /* a quantized weight matrix resident in HBM (re-tiled)  */
typedef struct ggml_hip_acts   ggml_hip_acts;     /* quantized activations (Q8_0 / Q8_1 / Q8_K) in HBM      */

/* ---- device / memory plumbing (reference: ggml_init_cublas ggml-cuda.cu:1982-2041, pool 1738-1853) ---- */
int     ggml_hip_init(int device);                /* idempotent; returns number of visible HIP devices        */
int     ggml_hip_device_count(void);
/* tuning aid: wall_clock64 phase stamps (8 per workgroup) of the last k_gemv_ln [0,4096) and k_gemv_out [4096,8192) */
void    ggml_hip_debug_stamps(int enable, long long * out_host);
void    ggml_hip_debug_force_gemv(int on);        /* tests: N > 4 through column-chunked mat-vec instead of the MFMA GEMM */
void    ggml_hip_debug_attention_form(int form);  /* tests / tuning: the prefill attention kernel for N >= 32 tokens -- 0 default (k_attention_flash: 32-token tiles, K.Q twice,
                                                     fp16 probabilities in LDS, while they fit; else 32), 1 the same forced, 32 the scratch form (scores through HBM), 16 / 17:
                                                     16-token tiles with the score rows (f32) / the probabilities (fp16, exact) in LDS; all bit-identical (ggml.c:10911-11102, 12389-12456) */
int     ggml_hip_debug_exp_boundary(unsigned * out_host, int cap);   /* diagnostic: (input bits << 16 | table entry) of the fp16 inputs whose exp() the in-kernel formula's f32 fast
                                                     path leaves undecided (they are looked up in csrc/fq_exp_fix.h, generated from this list); returns their number */
/* Prefill GEMM (N > 4): by default each row's sum over its 32-element blocks is split into S interleaved partial sums,
 * P_s = blocks s, s + S, ... added left to right, result ((P0 + P1) + P2) + P3: S = 4 for matrices with fewer than 4 x #CU
 * 32x32 tiles, S = 2 above (S times the K-parallelism; 1.1-1.5 x faster). on = 1: S = 1 always, the reference's single
 * left-to-right sum (ggml_vec_dot_q*_q8_*, scalar branch) -- legacy-format results are then bit-identical with the
 * reference's scalar build.                                                                                              */
void    ggml_hip_gemm_sequential(int on);
/* on = 1: the arithmetic ORDER of the reference's portable (SIMD-less) build everywhere: every quantized mat-mul, any N,
 * through a per-thread restatement of the scalar branches of ggml_vec_dot_q*_q8_* (csrc/fq_ref_dot.h: left-to-right block
 * sums for the legacy formats, the eight float lanes of k_quants.c:1684-1746 ... for Q3_K .. Q6_K), f64 accumulation of the
 * attention's two dot products (ggml.c:2296-2300; the default is f32 fused multiply-add chains like the reference's SIMD
 * builds), N = 1 steps through the op-by-op launch list. Prefill AND decode logits of all ten formats are then
 * bit-identical with that reference build's (tests/test_gpu_falcon.py). A parity instrument: 10-100 x slower.
 * on = 2 (round 6): the SAME association -- the same bits as on = 1 and as the reference's scalar build -- on the fast kernels wherever they have it, i.e. the legacy
 * formats (Q4_0, Q4_1, Q5_0, Q5_1, Q8_0): N = 1 steps through the fused decode launches, whose lanes leave every unit's f32 term (the reference's per-block term,
 * ggml.c:2591-2609 ...) in an LDS strip [row][block] that a wave with lane = row adds left to right (csrc/fq_ref_chain.h); batches of N > 4 through the int8-MFMA GEMM
 * with one left-to-right sum per row (ggml_hip_gemm_sequential); attention dots in f64. The k-quants: every mat-mul column by column through k_gemv_kq_ref
 * (csrc/kernels_kqref.hip: the scalar branches' eight float lanes per super-block at wave speed: ~0.5-0.7 x the default order's decode). Legacy N = 2..4 runs
 * mode 1's kernel. Falcon-7B Q4_0
 * decode: 925-930 tok/s (default order 1 005-1 020, mode 1: 40). tests/test_gpu_ref_fast.py.                                                                      */
void    ggml_hip_reference_order(int on);
int     ggml_hip_get_reference_order(void);
/* ---- row-split tensor parallelism, one process per GPU (csrc/split_tp.hip): the reference's `-ts` / GGML_BACKEND_GPU_SPLIT.
 * ggml_hip_tensor_split_rows: the row range of every device for a matrix of nrows rows, with the reference's arithmetic
 *   (ggml_cuda_set_tensor_split ggml-cuda.cu:2050-2077, ggml_cuda_transform_tensor :3044-3052); host only.
 * ggml_hip_weight_upload_rows: rows [row_low, row_high) of the ggml block bytes onto this process's device (NULL: empty range).
 * ggml_hip_mul_mat_q_split: every rank multiplies its rows, then the ranks exchange them by RCCL send/recv (in place of the
 *   host-side gather on the main device, :2779-2788); dst [N][M] is complete on every rank and bit-identical to the unsplit
 *   ggml_hip_mul_mat_q. unique_id: 128 bytes of falcon_hip_pipeline_unique_id (world == 1: NULL, no RCCL).
 * ggml_hip_mul_mat_q_split_local: all ranks' parts in one process on one device (tests).                                   */
typedef struct ggml_hip_split_comm ggml_hip_split_comm;
/* joins this process to a row-split job at the ggml-cuda.h boundary: from now on ggml_cuda_transform_tensor uploads this
 * rank's rows of GGML_BACKEND_GPU_SPLIT tensors (proportions: ggml_cuda_set_tensor_split) and their mat-muls inside
 * ggml_cuda_compute_forward exchange rows by RCCL. world <= 1 leaves / restores the default (whole matrices). Returns 0; 1 (nothing changed)
 * while split tensors uploaded under the current job are still alive. */
int     ggml_hip_split_configure(int rank, int world, const void * unique_id);
void    ggml_hip_tensor_split_rows(const float * tensor_split, int n_devices, int64_t nrows, int64_t * row_low, int64_t * row_high);
ggml_hip_weight * ggml_hip_weight_upload_rows(int type, const void * host_blocks, int64_t K, int64_t nrows, int64_t row_low, int64_t row_high);
ggml_hip_split_comm * ggml_hip_split_comm_create(int rank, int world, const void * unique_id);
void    ggml_hip_split_comm_free(ggml_hip_split_comm * c);
/* collective: 0 when every rank passed the same n bytes (the shim checks the -ts proportions with it once per job), 1 otherwise */
int     ggml_hip_split_comm_agree(ggml_hip_split_comm * c, const void * bytes, size_t n);
int     ggml_hip_mul_mat_q_split(ggml_hip_split_comm * c, const ggml_hip_weight * w_rows, const float * x_dev, int64_t K, int64_t N,
                                 float * dst_dev, int64_t M, const int64_t * row_low, const int64_t * row_high);
int     ggml_hip_mul_mat_q_split_local(ggml_hip_weight * const * parts, int world, const float * x_dev, int64_t K, int64_t N,
                                       float * dst_dev, int64_t M, const int64_t * row_low, const int64_t * row_high);
/* the local form with the EXCHANGE of ggml_hip_mul_mat_q_split run for real: a communicator of one RCCL rank (this process and GPU)
 * standing for `world` virtual ranks; every part's rows are computed into a private buffer, packed, sent with a grouped ncclSend /
 * ncclRecv addressed to rank 0 itself, and unpacked into dst -- pack, RCCL calls and unpack of a real job on a single-GPU box.
 * ggml_hip_split_comm_create_loopback: NULL when RCCL is missing or refuses; ggml_hip_split_comm_rccl_ranks: ncclCommCount (0: no RCCL communicator). */
ggml_hip_split_comm * ggml_hip_split_comm_create_loopback(int virtual_world);
int     ggml_hip_split_comm_rccl_ranks(ggml_hip_split_comm * c);
int     ggml_hip_mul_mat_q_split_loopback(ggml_hip_split_comm * c, ggml_hip_weight * const * parts, const float * x_dev, int64_t K, int64_t N,
                                          float * dst_dev, int64_t M, const int64_t * row_low, const int64_t * row_high);
/* staging-buffer pool of the ggml-cuda.h boundary (ggml_cuda_compute_forward's src1 / dst device copies; the reference's
 * ggml_cuda_pool_malloc, ggml-cuda.cu:1738-1816): buffers ever allocated, hand-outs served by reuse, buffers free now */
void    ggml_hip_shim_pool_stats(size_t * n_alloc, size_t * n_reuse, size_t * n_free);
int     ggml_hip_selftest(void);                  /* device self-checks (wave reductions); 0 = pass                */
/* soft_max's fp16 EXP table entries are recomputed in the attention kernels instead of gathered when -- checked at init, for
 * every non-NaN fp16 input -- the recomputation equals the host-built table. Returns the number of mismatching inputs (0 = in
 * use), -1 when GGML_HIP_EXP_TABLE=1 forces the gather. */
int     ggml_hip_exp_formula_mismatches(void);
void *  ggml_hip_stream(void);                    /* hipStream_t used for every launch of this library        */
void *  ggml_hip_malloc(size_t bytes);
void    ggml_hip_free(void * dev);
void    ggml_hip_memcpy_h2d(void * dst_dev, const void * src_host, size_t bytes);   /* stream-ordered, then waits */
void    ggml_hip_memcpy_d2h(void * dst_host, const void * src_dev, size_t bytes);   /* stream-ordered, then waits */
void    ggml_hip_memcpy_d2d(void * dst_dev, const void * src_dev, size_t bytes);    /* stream-ordered, async      */
void    ggml_hip_memset(void * dst_dev, int value, size_t bytes);
void    ggml_hip_synchronize(void);
/* timing on the library's stream (hipEvent); elapsed in milliseconds */
void *  ggml_hip_event_create(void);
void    ggml_hip_event_record(void * ev);
float   ggml_hip_event_elapsed_ms(void * ev_start, void * ev_stop);   /* synchronises on ev_stop */
void    ggml_hip_event_destroy(void * ev);
/* per-launch timing of the quantized mat-vec kernels (hipEvents on the launch stream) for the roofline report:
 * between begin and end every GEMV launch is bracketed; end returns #launches, summed microseconds and summed
 * algorithmic bytes (the weight matrix of each launch, read once)                                            */
void    ggml_hip_profile_begin(void);
void    ggml_hip_profile_end(int64_t * n_launches, double * total_us, double * total_bytes);
double  ggml_hip_profile_bracket_overhead_us(void);   /* duration an EMPTY event bracket reports on this stream */
const uint16_t * ggml_hip_gelu_table_dev(void);   /* 65536 fp16 entries, built like ggml.c:4276-4290 */
const uint16_t * ggml_hip_exp_table_dev(void);

/* ---- weights: ggml_cuda_transform_tensor (ggml-cuda.cu:3030-3073) ---------------------------------------- */
/* host_blocks: M rows of K/blck ggml blocks exactly as in a model file (type = enum ggml_type value).         */
ggml_hip_weight * ggml_hip_weight_upload(int type, const void * host_blocks, int64_t K, int64_t M);
void    ggml_hip_weight_free(ggml_hip_weight * w);                 /* ggml_cuda_free_data, ggml-cuda.cu:3075-3092 */
size_t  ggml_hip_weight_nbytes(const ggml_hip_weight * w);         /* == ggml_nbytes of the tensor               */

/* dequantize_row_q* (ggml.c:1509-1619, k_quants.c:344-876) / ggml_compute_forward_get_rows_q (ggml.c:11975):
 * dst_dev[i][0..K) = dequantized weight row rows_dev[i] (rows_dev == NULL: rows 0..nrows-1)                  */
void    ggml_hip_dequantize_rows(const ggml_hip_weight * w, const int32_t * rows_dev, int64_t nrows, float * dst_dev);

/* ---- weight quantizers: ggml_quantize_chunk (ggml.c:19479-19560) over quantize_row_q*_reference
 * (ggml.c:927-1129; k_quants.c:275-343, 396-471, 542-606, 652-733, 781-844) ------------------------------- */
/* nrows rows of K f32 -> nrows * K/blck ggml blocks, byte-identical with the reference's model files.
 * hist_dev: 16 int64 counters that are ADDED to (the reference's quantize histogram; legacy formats only, the
 * k-quants never count), or NULL. Returns 0, or -1 (message on stderr) for a bad type / row length.             */
int     ggml_hip_quantize_rows(int type, const float * x_dev, int64_t K, int64_t nrows, void * blocks_dev, int64_t * hist_dev);
/* same, straight into a resident weight matrix (quantize + re-tile on the device)                              */
ggml_hip_weight * ggml_hip_weight_quantize(int type, const float * x_dev, int64_t K, int64_t M);
void    ggml_hip_fp16_to_fp32_row(const uint16_t * src_dev, float * dst_dev, int64_t n);   /* ggml.c:370-374 */

/* ---- activations: INIT phase of mul_mat_q (ggml.c:11462-11476) ------------------------------------------- */
/* act_type: 8 = Q8_0 (ggml.c:1106-1129), 9 = Q8_1 (ggml.c:1292-1325), 15 = Q8_K (k_quants.c:899-934)          */
ggml_hip_acts * ggml_hip_acts_alloc(int act_type, int64_t K, int64_t max_cols);
void    ggml_hip_acts_free(ggml_hip_acts * a);
void    ggml_hip_quantize_acts(ggml_hip_acts * a, const float * x_dev, int64_t ldx, int64_t ncols);
/* writes ncols * K/blck ggml blocks (block_q8_0 / block_q8_1 / block_q8_K bytes) for bit-exact comparison     */
void    ggml_hip_acts_export(const ggml_hip_acts * a, int64_t ncols, void * out_dev);

/* ---- quantized mat-mul: ggml_compute_forward_mul_mat_q_f32 (ggml.c:11318-11529) --------------------------- */
/* dst_dev[n*ldd + m] = sum_k W[m][k] * x[n][k] for n < N, m < M with the reference CPU arithmetic
 * (activations quantized to the weight type's vec_dot_type, exact integer block dots, f32 epilogue).          */
void    ggml_hip_mul_mat_q(const ggml_hip_weight * w, const float * x_dev, int64_t ldx, int64_t N,
                           float * dst_dev, int64_t ldd);
/* same, activations already quantized; epilogue: 0 = store, 1 = GELU (ggml.c:3477-3484),
 * 2 = dst = (v + add1) + add2 (libfalcon.cpp:2399-2400; add1/add2 have column stride ldd)                      */
void    ggml_hip_mul_mat_q_acts(const ggml_hip_weight * w, const ggml_hip_acts * a, int64_t N, float * dst_dev,
                                int64_t ldd, int epilogue, const float * add1_dev, const float * add2_dev);

/* ---- the other ops of a Falcon block ------------------------------------------------------------------------ */
/* ggml_norm (ggml.c:10540-10594) followed by * w + b (libfalcon.cpp:2166-2188); w_dev == NULL: plain norm       */
void    ggml_hip_layer_norm(const float * x_dev, int64_t n, int64_t rows, const float * w_dev, const float * b_dev, float * y_dev);
void    ggml_hip_gelu(const float * x_dev, float * y_dev, int64_t n);                 /* ggml.c:3477-3484 */
void    ggml_hip_add3(const float * a, const float * b, const float * c, float * y, int64_t n);   /* (a+b)+c */
/* cos/sin table for ggml_rope mode 2 with Falcon's dynamic-NTK flags (ggml.c:12875-12898, libfalcon.cpp:2229-2234):
 * returns a device table [n_pos][head_dim/2][2]                                                                */
float * ggml_hip_rope_table_create(int head_dim, int n_pos, int rope_n_ctx);
/* rotate Q in place, rotate K into k_cache, copy V into v_cache at positions n_past..n_past+N-1
 * (ggml.c:12957-12978, libfalcon.cpp:2238-2280). qkv_dev rows are [H q-heads | HKV k-heads | HKV v-heads] x D. */
void    ggml_hip_rope_kv_store(float * qkv_dev, int N, int H, int HKV, int D, int n_past, const float * rope_table_dev,
                               float * k_cache_dev, float * v_cache_dev);
/* K.Q, scale 1/sqrt(D), causal mask, soft_max, V.P, merge heads (libfalcon.cpp:2285-2366):
 * att_dev[N][H*D]; caches are [n_ctx][HKV][D] f32 for ONE layer                                                */
void    ggml_hip_attention(const float * qkv_dev, int N, int H, int HKV, int D, int n_past,
                           const float * k_cache_dev, const float * v_cache_dev, float * att_dev);

#ifdef __cplusplus
}
#endif
#endif
