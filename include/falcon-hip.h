/*
 * falcon-hip.h -- C ABI of libggml_hip.so, model level: a device-resident Falcon decoder stack.
 *
 * Mirrors the part of libfalcon.h that sits on the hot path (cmp-nct/ggllm.cpp):
 *   falcon_hip_model_*    <- falcon_model / falcon_model_load_internal   (libfalcon.cpp:1552-1959)
 *   falcon_hip_context_*  <- falcon_context_prepare / kv_cache_init      (libfalcon.cpp:3755, 1335-1385)
 *   falcon_hip_eval       <- falcon_eval -> falcon_eval_internal         (libfalcon.cpp:4566, 2011-2588)
 *   falcon_hip_get_logits <- falcon_get_logits                           (libfalcon.cpp:4678)
 * Unlike the reference (which rebuilds a ggml graph per call, runs attention/LN/GELU on the CPU and round-trips
 * every mat-mul over PCIe), all weights, the KV cache and every activation stay in HBM; only token ids go in and
 * logits come out. A context may own a contiguous range of layers (a pipeline STAGE): the first
 * stage embeds tokens, inner stages take/give the residual stream as device pointers, the last stage applies
 * ln_f + lm_head.
 */
#ifndef FALCON_HIP_H
#define FALCON_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct falcon_hip_model   falcon_hip_model;
typedef struct falcon_hip_context falcon_hip_context;

typedef struct falcon_hip_hparams {      /* falcon_hparams, libfalcon.cpp:146-160 */
    int32_t n_vocab, n_embd, n_head, n_head_kv, n_layer, n_ff;
    int32_t two_norms;                   /* 1: ln_attn + ln_mlp per block (40B / 180B), 0: shared input_layernorm (7B) */
    int32_t layer_begin, layer_end;      /* this process holds blocks [layer_begin, layer_end)                         */
} falcon_hip_hparams;

falcon_hip_model * falcon_hip_model_create(const falcon_hip_hparams * hp);
void               falcon_hip_model_free(falcon_hip_model * m);
/* Upload one tensor by its model-file name (libfalcon.cpp:1764-1861), e.g.
 *   transformer.word_embeddings.weight, transformer.h.<i>.self_attention.query_key_value.weight,
 *   transformer.h.<i>.self_attention.dense.weight, transformer.h.<i>.mlp.dense_h_to_4h.weight,
 *   transformer.h.<i>.mlp.dense_4h_to_h.weight, transformer.h.<i>.input_layernorm.{weight,bias} (7B),
 *   transformer.h.<i>.ln_mlp.* / ln_attn.* (40B), transformer.ln_f.{weight,bias}, lm_head.weight
 * type: enum ggml_type value (0 = f32 for the norms). data: host bytes in ggml layout. Returns 0, or -1 if the
 * name is unknown / outside this stage's layer range (tensor ignored).                                         */
int  falcon_hip_model_set_tensor(falcon_hip_model * m, const char * name, int type, const void * data, int64_t ne0, int64_t ne1);
size_t falcon_hip_model_weight_bytes(const falcon_hip_model * m);   /* quantized bytes read per decoded token */

/* The reference's model file format, GGCC v10 (what falcon_quantize writes and falcon_main loads; libfalcon.cpp:770-973).
 * falcon_hip_model_load_ggcc maps the file and uploads blocks [layer_begin, layer_end) (layer_end <= 0: all; a pipeline
 * stage passes its own range) plus the embedding (first stage) and ln_f + lm_head (last stage); the vocabulary and BPE
 * merges are skipped (the tokenizer is not on this path). Returns NULL on error (message on stderr).
 * falcon_hip_ggcc_scan is the host-only part (no device needed): header + tensor directory (one text line per tensor:
 * "name ggml_type ne0 ne1 file_offset bytes"); returns the number of tensors or -1.                                  */
falcon_hip_model * falcon_hip_model_load_ggcc(const char * path, int layer_begin, int layer_end, falcon_hip_hparams * hp_out);
int  falcon_hip_ggcc_scan(const char * path, falcon_hip_hparams * hp_out, int * ftype_out, char * dir_out, size_t dir_cap);

/* Stage plan of a layer pipeline for a GGCC file (host-only): n_stages contiguous block ranges that minimise the weight
 * bytes the slowest stage streams per token (the last stage also streams ln_f + lm_head) -- what the reference's VRAM
 * planner decides per device (libfalcon.cpp:1660-1900) -- and the device bytes each stage needs (weights + embedding on
 * the first stage + n_streams x (KV cache of n_ctx positions + activations of n_batch tokens)). layer_begin / layer_end /
 * stage_bytes: n_stages entries. Returns 0, 1 if a stage needs more than vram_per_gpu (0 = no limit), -1 on error.       */
int  falcon_hip_plan_stages(const char * path, int n_stages, int n_ctx, int n_batch, int n_streams, size_t vram_per_gpu,
                            int * layer_begin, int * layer_end, size_t * stage_bytes);

/* falcon_model_quantize (libfalcon.h:176-179, libfalcon.cpp:3533-3743) for GGCC v10 files: writes path_out with every
 * 2-D "...weight" tensor converted to the tensor type of ftype (enum llama_ftype value: 0 f32, 1 f16, 2 Q4_0, 3 Q4_1,
 * 7 Q8_0, 8 Q5_0, 9 Q5_1, 10 Q2_K, 11-13 Q3_K, 14-15 Q4_K, 16-17 Q5_K, 18 Q6_K), lm_head.weight only when
 * quantize_output_tensor, already-quantized sources only when allow_requantize; the quantizers run on the device and the
 * file is byte-identical with the reference's. hist_out: NULL or 16 counters (the reference's printed histogram, summed
 * over tensors). Returns 0, or 1 after a message on stderr (the reference's convention); a failed run leaves no file.   */
int  falcon_hip_model_quantize(const char * path_in, const char * path_out, int ftype, int quantize_output_tensor,
                               int allow_requantize, int64_t * hist_out);

void falcon_hip_model_get_hparams(const falcon_hip_model * m, falcon_hip_hparams * hp_out);

/* n_ctx: KV capacity; n_batch: largest N of one eval; rope_n_ctx: the n_ctx handed to ggml_rope
 * (n_max_real_ctx or n_ctx, libfalcon.cpp:2229-2230)                                                           */
falcon_hip_context * falcon_hip_context_create(falcon_hip_model * m, int n_ctx, int n_batch, int rope_n_ctx);
void                 falcon_hip_context_free(falcon_hip_context * c);
/* A context of n_seq (1..256) independent sequences that advance in LOCK STEP: every falcon_hip_eval_stage / falcon_hip_stage_step
 * evaluates n_seq rows = one token of each sequence, all at position n_past, row t attending to its own KV cache. One pass
 * over the weights serves n_seq tokens. n_seq <= 4 (k-quant models at model widths: <= 2): the column mat-vec kernels, the same
 * bits per sequence as a context of its own; more: the small-batch mat-muls (5..16 columns per pass, larger contexts in
 * passes of 16 or through the tile GEMM) -- a sequence's logits then equal a context of its own within the association
 * spread of the f32 sums (a row of those mat-muls does not depend on the other rows).
 * token_dev / next_token_dev of falcon_hip_stage_step then hold n_seq ids, the hidden rows are [n_seq][n_embd].          */
falcon_hip_context * falcon_hip_context_create_seqs(falcon_hip_model * m, int n_ctx, int n_seq, int rope_n_ctx);
int                  falcon_hip_context_n_seq(const falcon_hip_context * c);

/* Evaluate n_tokens at position n_past (falcon_eval). Whole model in this process: tokens are host ids.
 * logits_all = 0 keeps the last row only. Returns 0; 1 for an empty / oversized batch or a position past n_ctx (nothing is
 * evaluated), 2 for a token id outside the vocabulary, 3 if an in-launch hand-off timed out (results invalid).          */
int falcon_hip_eval(falcon_hip_context * c, const int32_t * tokens, int n_tokens, int n_past, int logits_all);
/* Pipeline-stage form: hidden_in_dev / hidden_out_dev are device [n_tokens][n_embd] f32 (NULL where the stage
 * embeds tokens / produces logits).                                                                             */
int falcon_hip_eval_stage(falcon_hip_context * c, const int32_t * tokens, const float * hidden_in_dev, int n_tokens,
                          int n_past, int logits_all, float * hidden_out_dev);
/* One decode step (N = 1) of a pipeline stage with device-resident inputs/outputs, asynchronous on the library stream:
 * token_dev (stage 0) or hidden_in_dev [n_embd] (other stages) -> hidden_out_dev [n_embd] (not the last stage) or the
 * greedy-sampled next_token_dev (last stage). The caller exchanges them with RCCL send/recv (bench_pipeline.py).      */
int falcon_hip_stage_step(falcon_hip_context * c, const int32_t * token_dev, const float * hidden_in_dev, int n_past,
                          float * hidden_out_dev, int32_t * next_token_dev);
/* Greedy decode loop entirely stream-ordered: evaluates `first_token` at n_past, then n_steps-1 more argmax-sampled
 * tokens (falcon_main --temp 0, falcon_main.cpp:958-960); out_tokens receives n_steps ids. No host sync per step. */
int falcon_hip_decode_greedy(falcon_hip_context * c, int32_t first_token, int n_past, int n_steps, int32_t * out_tokens);

/* falcon_eval with n_tokens = 1 (libfalcon.cpp:4566) without a host round trip: the fused decode launches are replayed from a
 * hipGraph and the logits row is copied into page-locked host memory behind them; falcon_hip_get_logits waits for that copy
 * (libfalcon.h:256, 263). Returns as falcon_hip_eval; an in-launch hand-off that timed out in such an asynchronous step is
 * noticed at the next falcon_hip_get_logits and is STICKY: falcon_hip_context_last_error reports 3 and every later eval of the
 * context returns 3 (its KV cache holds invalid rows). */
int falcon_hip_eval_token(falcon_hip_context * c, int32_t token, int n_past);
int falcon_hip_context_last_error(const falcon_hip_context * c);      /* 0 or 3; host state only (no device wait) */
/* the n_ctx ggml_rope is given may change per call (falcon_evaluation_config::n_max_real_ctx, libfalcon.cpp:2229-2230);
 * the table is rebuilt when the dynamic-NTK bucket n_ctx / 2048 changes (<= 0: the context's n_ctx)                       */
void falcon_hip_context_set_rope_n_ctx(falcon_hip_context * c, int rope_n_ctx);
/* falcon_hip_eval, then one line per launch site on stderr (calls, total / average microseconds, share): what the reference's --debug-timings node table
 * (libfalcon.cpp:2506-2520) becomes on the resident path, where a block is a handful of launches instead of a ggml graph; falcon_wrap.cpp calls it where the
 * reference would print its table (falcon_evaluation_config::debug_timings, same first / last / every-token rule) */
int   falcon_hip_eval_debug_timings(falcon_hip_context * c, const int32_t * tokens, int n_tokens, int n_past, int logits_all);
const float * falcon_hip_get_logits(falcon_hip_context * c);        /* host, n_vocab (or n_tokens*n_vocab) floats */
/* The reference's perplexity loop (falcon_perplexity.cpp:28-124) over a token stream: chunks of n_ctx tokens evaluated
 * from an empty context in batches of n_batch, NLL of the second half of every chunk (host soft_max as in :12-27).
 * Returns the number of scored tokens and their summed NLL (perplexity = exp(nll / count)).                            */
int   falcon_hip_perplexity(falcon_hip_context * c, const int32_t * tokens, int64_t n_tokens, int n_ctx, int n_batch, double * nll_out);
/* test hook: residual stream entering each local layer (+ leaving the last), device->host, after an eval with
 * falcon_hip_context_keep_hidden(c, 1): (n_local_layers + 1) * n_tokens * n_embd floats                          */
void  falcon_hip_context_keep_hidden(falcon_hip_context * c, int keep);
void  falcon_hip_get_hidden(falcon_hip_context * c, float * dst_host);
void  falcon_hip_context_use_graph(falcon_hip_context * c, int enable);   /* capture decode steps into a hipGraph */
/* N == 1 evals: 2 (default) = fused decode kernels, two launches per block (LayerNorm mat-vec in the ring form, csrc/kernels_ring.hip, where the format has it |
 * attention + output mat-vec, when the grid fits the chip), 3 = one launch per block (the next block's LayerNorm mat-vec as a second phase of the same
 * launch; measured slower, kept for A/B), 1 = three launches, 0 = one launch per graph op (tests), 5 = mode 2 with the ring form forced. All produce the
 * same bits. (4 was the persistent decode engine of rounds 2-3 -- one launch per token, measured 36 % slower, removed in round 6: it now selects mode 2.)
 * With ggml_hip_reference_order(2) (ggml-hip-ops.h) modes 1, 2 and 5 run the same launches in the reference's own association (csrc/fq_ref_chain.h). */
void  falcon_hip_context_set_fused(falcon_hip_context * c, int mode);
/* 1 if an in-launch wait of the 2-launch form ever timed out (results invalid; never expected). Synchronises the device. */
int   falcon_hip_context_sync_error(falcon_hip_context * c);

/* ---- layer pipeline over several GPUs, one process per GPU (csrc/falcon_pipeline.hip). Rank r holds the blocks
   [layer_begin, layer_end) of its falcon_hip_model (falcon_hip_plan_stages picks the ranges); n_groups groups of `batch`
   lock-step sequences are in flight (n_groups >= world; >= 2 x world: transfers overlap stage steps); the residual rows and
   the sampled tokens travel by RCCL ncclSend / ncclRecv (xGMI). What the reference does instead: every mat-mul split over the
   devices with peer copies and a gather on the main device (ggml-cuda.cu:2586-2608, 2713-2732).
     rank 0:  falcon_hip_pipeline_unique_id(id)  -> hand the 128 bytes to the other ranks (launcher's store, file, MPI ...)
     all:     p = falcon_hip_pipeline_create(model, rank, world, id, n_groups, batch, n_ctx)
     rank 0:  falcon_hip_pipeline_set_tokens(p, first_tokens)            n_groups * batch ids (sequence i = group * batch + b)
     all:     falcon_hip_pipeline_run(p, rounds, n_past0)                 asynchronous; every sequence advances `rounds` tokens
     last:    falcon_hip_pipeline_get_history(p, out, first_round, n)     the sampled tokens [round][sequence]; waits
   world == 1 needs neither RCCL nor an id (NULL). */
#define FALCON_HIP_PIPELINE_ID_BYTES 128
typedef struct falcon_hip_pipeline falcon_hip_pipeline;
int   falcon_hip_pipeline_unique_id(void * id_out);                       /* 0, or -1 when RCCL is not available */
falcon_hip_pipeline * falcon_hip_pipeline_create(falcon_hip_model * m, int rank, int world, const void * unique_id,
                                                 int n_groups, int batch, int n_ctx);
void  falcon_hip_pipeline_free(falcon_hip_pipeline * p);
int   falcon_hip_pipeline_rccl_ranks(falcon_hip_pipeline * p);           /* ncclCommCount of its communicator (1: one rank, 0: local transport) */
/* how this rank's hand-offs travel: 0 one stage (none), 1 RCCL ncclSend / ncclRecv, 2 device copies inside one process (local), 3 the local job over
 * a one-rank RCCL communicator, 4 host shared memory between the processes of one node -- selected for falcon_hip_pipeline_create by the environment,
 * FALCON_PIPE_TRANSPORT=shm: the multi-process job on a node whose ranks share a GPU (RCCL refuses two ranks on one device); same ranks, same unique-id
 * hand-out, same slot schedule and stage steps, the exchange blocking on the host */
int   falcon_hip_pipeline_transport(falcon_hip_pipeline * p);
/* one small message around the ring of ranks over a fresh RCCL communicator (rank r -> r + 1), checked and torn down: 0 = RCCL's send / recv works between
 * these ranks. bench_pipeline.py runs it in a child process per rank under a time-out before the job and falls back to FALCON_PIPE_TRANSPORT=shm otherwise. */
int   falcon_hip_rccl_selftest(int rank, int world, const void * unique_id, int device);
int   falcon_hip_pipeline_set_tokens(falcon_hip_pipeline * p, const int32_t * tokens);
int   falcon_hip_pipeline_run(falcon_hip_pipeline * p, int rounds, int n_past0);
int   falcon_hip_pipeline_get_history(falcon_hip_pipeline * p, int32_t * out, int first_round, int n_rounds);
/* the same job with every rank inside ONE process on one device (hand-off by device copies instead of RCCL): the identical
   schedule and stage code, for tests and for sizing a pipeline before the GPUs are there */
falcon_hip_pipeline * falcon_hip_pipeline_create_local(falcon_hip_model * m, int rank, int world, int n_groups, int batch, int n_ctx);
int   falcon_hip_pipeline_run_local(falcon_hip_pipeline ** ranks, int world, int rounds, int n_past0);
/* switches a local job's hand-offs from device copies to RCCL itself: one communicator of ONE rank (this process and GPU), every
   message a grouped ncclSend / ncclRecv addressed to rank 0 on the pipeline's second stream, ordered by the events of the overlapped
   schedule -- the library's RCCL binding and call sequence exercised on a box with a single GPU. 0, or -1 (RCCL missing / refused);
   afterwards falcon_hip_pipeline_rccl_ranks reports the communicator's count (1) for every rank of the job */
int   falcon_hip_pipeline_local_attach_rccl(falcon_hip_pipeline ** ranks, int world);
/* host only: the slot schedule (see falcon_pipeline.hip); out: 9 ints; returns the number of slots of the run */
int   falcon_hip_pipeline_schedule(int rank, int world, int n_groups, int rounds, int slot, int * out);

/* ---- tokenizer (host only): falcon_tokenize / falcon_token_to_str (libfalcon.h:236-247, libfalcon.cpp:2594-3035, 4623-4641)
   on the vocabulary and BPE merges stored in a GGCC v10 file. Same ids as the reference for any text. */
typedef struct falcon_hip_vocab falcon_hip_vocab;
falcon_hip_vocab * falcon_hip_vocab_load_ggcc(const char * path);          /* never NULL; check falcon_hip_vocab_error */
const char * falcon_hip_vocab_error(const falcon_hip_vocab * v);           /* NULL = loaded */
void  falcon_hip_vocab_free(falcon_hip_vocab * v);
int   falcon_hip_vocab_size(const falcon_hip_vocab * v);
int   falcon_hip_vocab_merges(const falcon_hip_vocab * v);
/* ids of `text` (bos = 11 first when add_bos and the text is not empty); returns their number, or minus that number when
   n_max is too small (nothing written), as falcon_tokenize does */
int   falcon_hip_tokenize(const falcon_hip_vocab * v, const char * text, int32_t * tokens, int n_max, int add_bos);
/* the token's bytes (not NUL-terminated: byte tokens may be 0); returns their number or -1 */
int   falcon_hip_token_to_bytes(const falcon_hip_vocab * v, int32_t id, const char ** bytes);
int32_t falcon_hip_token_bos(void);
int32_t falcon_hip_token_eos(void);

#ifdef __cplusplus
}
#endif
#endif
