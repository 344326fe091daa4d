// falcon_hip.hip -- device-resident Falcon decoder stack (include/falcon-hip.h).
//
// Op sequence = falcon_eval_internal (libfalcon.cpp:2115-2466), one launch list per block:
//   LN(x) [+ LN2(x) on 40B]  -> Q8 activations        (ggml_norm/mul/add 2166-2188; mul_mat INIT ggml.c:11462)
//   QKV = Wqkv . a                                     (2192)
//   RoPE(Q,K) neox + KV append                         (2229-2280)
//   att = softmax(mask(K.Q/8)) V, heads merged         (2285-2366)
//   wo  = Wo . q8(att)                                 (2370)
//   up  = gelu(Wup . a)                                (2389-2392)
//   x   = (Wdown . q8(up) + wo) + x                    (2394-2400)
// then ln_f + lm_head (2421-2440). Everything runs on the library stream; a decode step (N = 1) reads its token id
// and n_past from device memory, so one captured hipGraph is replayed for every generated token.
#include "fq_device.h"
#include "kernels.h"
#include "hip_context.h"
#include "../../include/falcon-hip.h"
#include <cmath>
#include <cstring>

#include <string>
#include <string.h>
#include <vector>
#include <atomic>

struct layer_weights {
    fq_weight qkv{}, wo{}, up{}, down{};
    float * ln_w = nullptr, * ln_b = nullptr, * ln2_w = nullptr, * ln2_b = nullptr;
};

struct falcon_hip_model {
    falcon_hip_hparams hp{};
    fq_weight tok_emb{}, lm_head{};
    float * out_norm_w = nullptr, * out_norm_b = nullptr;
    std::vector<layer_weights> layers;        // local layers only
    std::vector<void *> allocs;
    // Weight ARENA: every matrix of the stage is carved out of a few very large allocations instead of one hipMalloc per
    // tensor. Measured on MI355X (Falcon-7B Q4_0, the persistent engine's loader alone, round 3): 131
    // separate allocations 2680 us per token (1.45 TB/s), one arena 695 us (5.6 TB/s) -- the driver maps a large allocation
    // with large page fragments, and a wave that streams 1 KiB pieces lives or dies by the translation reach.
    uint8_t * arena_cur = nullptr; size_t arena_left = 0;
    size_t weight_bytes = 0;                   // quantized bytes streamed per decoded token (local layers [+ lm_head])
    bool first_stage() const { return hp.layer_begin == 0; }
    bool last_stage()  const { return hp.layer_end == hp.n_layer; }
};

struct falcon_hip_context {
    falcon_hip_model * m = nullptr;
    int n_ctx = 0, n_batch = 0, rope_n_ctx = 0;
    // n_seq > 0: a context of n_seq independent sequences that advance in lock step (falcon_hip_context_create_seqs): every
    // step evaluates one token of each, row t of the activations belongs to sequence t and attends to its own KV cache
    // ([layer][seq][n_ctx][HKV][D]) -- one pass over the stage's weights serves n_seq tokens
    int n_seq = 0;
    float * x = nullptr, * ln = nullptr, * ln2 = nullptr, * qkv = nullptr, * att = nullptr, * wo_out = nullptr, * up = nullptr;
    fq_att_scratch att_scratch{nullptr, 0};
    float * logits_dev = nullptr;
    // quantized-activation images. Each buffer holds the LARGEST image of its length (Q8_1: 1.25 bytes per element) and is
    // typed per use from the weight that consumes it (ggml.c:1627-1718 vec_dot_type), so that a file which mixes weight
    // formats between tensors or blocks never writes an image into a buffer sized for another family.
    uint8_t * buf_e = nullptr, * buf_e2 = nullptr, * buf_att = nullptr, * buf_ff = nullptr;
    float * k_cache = nullptr, * v_cache = nullptr;
    float * rope_cs = nullptr;
    int * n_past_dev = nullptr;
    int32_t * tokens_dev = nullptr, * out_tokens_dev = nullptr;
    float * hidden_dev = nullptr;
    float * argmax_val = nullptr;              // per-workgroup greedy candidates written by the lm_head kernel
    int   * argmax_idx = nullptr;
    bool keep_hidden = false;
    int  hidden_tokens = 0;
    std::vector<float> logits_host;
    float * logits_pinned = nullptr;           // one row of page-locked host memory: falcon_hip_eval_token's copy is enqueued behind the graph replay
    const float * logits_last = nullptr;       // what falcon_hip_get_logits returns: logits_host.data() or logits_pinned
    bool logits_pending = false;               // falcon_hip_eval_token: the row's copy is in flight (falcon_hip_get_logits waits for it)
    bool sync_err_sticky = false;              // an in-launch hand-off timed out in an asynchronous step: every later eval of this context fails (3)
    hipGraphExec_t token_graph = nullptr;      // falcon_hip_eval_token's captured step
    int token_sig = -1;
    std::vector<void *> allocs;
    bool use_graph = false;
    // fused decode: MLP-up GEMV on a side stream, concurrent with QKV GEMV + attention. Measured on MI355X (Falcon-7B
    // Q4_0, hipGraph): 555 tok/s with the fork/join vs 674 tok/s in stream order -- the cross-stream dependencies cost
    // more than the ~8 us of attention they hide, so it is OFF by default (FALCON_HIP_DUAL=1 turns it on).
    bool dual_stream = false;
    hipStream_t side = nullptr;
    std::vector<hipEvent_t> ev_fork, ev_join, ev_attn;  // per local layer (dual-stream decode; two-branch prefill: fork, MLP branch done, attention branch done)
    int  par2_max_n = 1 << 30;                  // batches of 33 .. par2_max_n tokens run a block's attention and MLP branches on two streams (FALCON_HIP_PAR2_MAX_N; 0: never). Up to 32 columns the streaming mat-mul fills every CU by itself (16 tokens: 4.14 ms in stream order against 4.63 forked)
    bool fused_decode = true;                  // N == 1: k_gemv_ln / k_attn_decode / k_gemv_out instead of the op-by-op list
    bool merged_attn_out = true;               // ... with attention and the output mat-vec in one launch (k_attn_out) when the grid fits the chip
    // ... and the next block's k_gemv_ln as a second phase of that launch (k_attn_out_ln): one launch per block. Measured on
    // MI355X (Falcon-7B Q4_0): 867 tok/s against 934 with two launches -- the second phase starts 4.3 us after the last
    // residual value is published (every workgroup sweeps the 36 KB row out of the hand-off buffer, the early finishers
    // keep polling it) and the launch boundary it removes costs less than that. OFF by default; set_fused(3) /
    // FALCON_HIP_TWO_PHASE=1 turn it on.
    bool two_phase = false;
    unsigned long long * x_gran = nullptr;     // hand-off granules of the residual row between the two phases
    unsigned * sync_words = nullptr;           // [0] hand-off epoch of k_attn_out, [1] its time-out flag, [16..80) rope row of the position
    unsigned long long * att_gran = nullptr;   // hand-off granules of k_attn_out: one per 32-bit word of the attention image / row
    hipGraphExec_t decode_graph = nullptr;
    // one captured pipeline-stage step (falcon_hip_stage_step): valid for these device pointers; n_past lives on the device
    hipGraphExec_t step_graph = nullptr;
    const void * sg_in[2] = { nullptr, nullptr }; void * sg_out[2] = { nullptr, nullptr };
    int step_next_n_past = -1;
    bool stage_graph = true;                   // FALCON_HIP_STAGE_GRAPH=0: plain launches
    bool ring_ln = false;                      // k_gemv_ln's launches in the ring form (kernels_ring.hip; k-quants: kernels_ringk.hip); FALCON_HIP_RING=0: never
    bool ring_out = false;                     // the unmerged k_gemv_out launches in the ring form (kernels_ringk.hip): FALCON_HIP_RING_OUT=1 always, 0 never, unset: where it measured faster (ring_out_auto)
    // batched evaluation replayed from hipGraphs (FALCON_HIP_PREFILL_GRAPH=1; default: plain launches): the ~320 launches and 96 cross-stream joins of a
    // prompt batch are then scheduled by the graph instead of the host (the joins cost ~12 us of idle device each as stream events: 11 % of a
    // 128-token Falcon-7B prompt). Keyed by (tokens, keys, mode signature, keep_hidden); the position and the token ids are read from device memory.
    struct batch_graph { int N, max_kv, sig; hipGraphExec_t exec; };
    std::vector<batch_graph> batch_graphs;
    bool prefill_graph = false;                // (off by default: a replay saves 1-2 % of a batch, capturing a new (tokens, keys) shape costs ~5 ms once)
    int  graph_base = -1;                      // n_past the captured graph was built for
    int  decode_sig = -1, step_sig = -1;       // graph_signature() at capture time
    unsigned sync_err_host = 0;                // copy of sync_words[1], fetched wherever the host synchronises anyway
};

// k_attn_out hands the attention output from workgroup to workgroup with a BOUNDED spin; a time-out sets sync_words[1] and
// the launch carries on with whatever the granules hold. Wherever the host waits for the stream anyway the word comes
// along, and a set word fails the call (callers of the purely stream-ordered falcon_hip_stage_step poll
// falcon_hip_context_sync_error themselves).
static void fetch_sync_error(falcon_hip_context * c, hipStream_t st) {
    HIP_CHECK(hipMemcpyAsync(&c->sync_err_host, c->sync_words + 1, 4, hipMemcpyDeviceToHost, st));
}
static int report_sync_error(falcon_hip_context * c, const char * where) {
    if (!c->sync_err_host) return 0;
    c->sync_err_sticky = true;
    fprintf(stderr, "falcon-hip: %s: an in-launch hand-off timed out (sync word %u) -- the results of this call are invalid\n", where, c->sync_err_host);
    return 3;
}

static void * dev_alloc(std::vector<void *> & keep, size_t bytes) {
    void * p = nullptr;
    HIP_CHECK(hipMalloc(&p, bytes ? bytes : 16));
    keep.push_back(p);
    return p;
}

extern "C" falcon_hip_model * falcon_hip_model_create(const falcon_hip_hparams * hp) {
    fq_ctx();
    falcon_hip_model * m = new falcon_hip_model();
    m->hp = *hp;
    if (m->hp.layer_end <= 0 || m->hp.layer_end > hp->n_layer) m->hp.layer_end = hp->n_layer;
    if (m->hp.layer_begin < 0) m->hp.layer_begin = 0;
    if (hp->n_embd % hp->n_head || hp->n_embd / hp->n_head != 64 || hp->n_head % hp->n_head_kv) {
        fprintf(stderr, "falcon-hip: unsupported head geometry (n_embd %d, n_head %d, n_head_kv %d)\n", hp->n_embd, hp->n_head, hp->n_head_kv);
        exit(1);
    }
    m->layers.resize((size_t)(m->hp.layer_end - m->hp.layer_begin));
    return m;
}

extern "C" void falcon_hip_model_free(falcon_hip_model * m) {
    if (!m) return;
    for (void * p : m->allocs) HIP_CHECK(hipFree(p));
    delete m;
}

static uint8_t * arena_alloc(falcon_hip_model * m, size_t bytes) {
    bytes = (bytes + 2048 + 65535) & ~(size_t) 65535;            // slack for clamped tail loads / the engine's last DMA piece; 64 KiB granules
    if (bytes > m->arena_left) {
        size_t chunk = (size_t) 1 << 30;
        if (const char * e = getenv("FALCON_HIP_ARENA_MB")) chunk = (size_t) atoll(e) << 20;
        if (chunk < bytes) chunk = bytes;
        void * p = nullptr;
        HIP_CHECK(hipMalloc(&p, chunk));
        m->allocs.push_back(p);
        m->arena_cur = (uint8_t *) p; m->arena_left = chunk;
    }
    uint8_t * r = m->arena_cur;
    m->arena_cur += bytes; m->arena_left -= bytes;
    return r;
}

static fq_weight upload_weight(falcon_hip_model * m, int type, const void * data, int64_t K, int64_t M) {
    hip_context & c = fq_ctx();
    const fq_type_desc d = fq_desc(type);
    if (d.blck == 0 || K % d.blck != 0) { fprintf(stderr, "falcon-hip: weight type %d with K=%lld unsupported\n", type, (long long) K); exit(1); }
    fq_weight w{};
    w.type = type; w.K = K; w.M = M; w.nblk = K / d.blck;
    w.bytes = (size_t) M * w.nblk * d.tsize;
    w.row_stride = fq_il_row_stride(d, w.nblk);
    uint8_t * slab = arena_alloc(m, (size_t) M * w.row_stride);
    for (int p = 0; p < d.nplanes; ++p) w.plane[p] = slab;
    const size_t row_bytes = (size_t) w.nblk * d.tsize;
    int64_t rows_per_chunk = (int64_t)((256u << 20) / row_bytes);
    if (rows_per_chunk < 1) rows_per_chunk = 1;
    if (rows_per_chunk > M) rows_per_chunk = M;
    uint8_t * stage = nullptr;
    HIP_CHECK(hipMalloc((void **) &stage, (size_t) rows_per_chunk * row_bytes));
    for (int64_t r0 = 0; r0 < M; r0 += rows_per_chunk) {
        const int64_t nr = (M - r0 < rows_per_chunk) ? M - r0 : rows_per_chunk;
        HIP_CHECK(hipMemcpyAsync(stage, (const uint8_t *) data + (size_t) r0 * row_bytes, (size_t) nr * row_bytes, hipMemcpyHostToDevice, c.stream));
        fq_weight sub = w;
        sub.M = nr;
        for (int p = 0; p < d.nplanes; ++p)
            sub.plane[p] = w.plane[0] + (size_t) r0 * w.row_stride;
        fq_launch_retile(stage, sub, c.stream);
        HIP_CHECK(hipStreamSynchronize(c.stream));
    }
    HIP_CHECK(hipFree(stage));
    return w;
}

static float * upload_f32(falcon_hip_model * m, const void * data, int64_t n) {
    float * p = (float *) dev_alloc(m->allocs, (size_t) n * 4);
    HIP_CHECK(hipMemcpy(p, data, (size_t) n * 4, hipMemcpyHostToDevice));
    return p;
}

extern "C" int falcon_hip_model_set_tensor(falcon_hip_model * m, const char * name_c, int type, const void * data, int64_t ne0, int64_t ne1) {
    const std::string name(name_c);
    const falcon_hip_hparams & hp = m->hp;
    auto want2d = [&](int64_t k, int64_t rows) {
        if (ne0 != k || ne1 != rows) { fprintf(stderr, "falcon-hip: %s has shape [%lld,%lld], expected [%lld,%lld]\n", name_c, (long long) ne0, (long long) ne1, (long long) k, (long long) rows); exit(1); }
    };
    auto want_f32 = [&]() { if (type != FQ_F32 || ne0 != hp.n_embd) { fprintf(stderr, "falcon-hip: %s must be f32[%d]\n", name_c, hp.n_embd); exit(1); } };
    const int64_t E = hp.n_embd, D = 64, QKV = (int64_t)(hp.n_head + 2 * hp.n_head_kv) * D;
    if (name == "transformer.word_embeddings.weight") {
        if (!m->first_stage()) return -1;
        want2d(E, hp.n_vocab); m->tok_emb = upload_weight(m, type, data, E, hp.n_vocab); return 0;
    }
    if (name == "lm_head.weight") {
        if (!m->last_stage()) return -1;
        want2d(E, hp.n_vocab); m->lm_head = upload_weight(m, type, data, E, hp.n_vocab); m->weight_bytes += m->lm_head.bytes; return 0;
    }
    if (name == "transformer.ln_f.weight") { if (!m->last_stage()) return -1; want_f32(); m->out_norm_w = upload_f32(m, data, E); return 0; }
    if (name == "transformer.ln_f.bias")   { if (!m->last_stage()) return -1; want_f32(); m->out_norm_b = upload_f32(m, data, E); return 0; }
    const std::string pre = "transformer.h.";
    if (name.compare(0, pre.size(), pre) != 0) return -1;
    const size_t dot = name.find('.', pre.size());
    if (dot == std::string::npos) return -1;
    const int il = atoi(name.substr(pre.size(), dot - pre.size()).c_str());
    if (il < hp.layer_begin || il >= hp.layer_end) return -1;
    layer_weights & L = m->layers[(size_t)(il - hp.layer_begin)];
    const std::string leaf = name.substr(dot + 1);
    if (leaf == "self_attention.query_key_value.weight") { want2d(E, QKV);     L.qkv  = upload_weight(m, type, data, E, QKV);      m->weight_bytes += L.qkv.bytes;  return 0; }
    if (leaf == "self_attention.dense.weight")           { want2d(E, E);       L.wo   = upload_weight(m, type, data, E, E);        m->weight_bytes += L.wo.bytes;   return 0; }
    if (leaf == "mlp.dense_h_to_4h.weight")              { want2d(E, hp.n_ff); L.up   = upload_weight(m, type, data, E, hp.n_ff);  m->weight_bytes += L.up.bytes;   return 0; }
    if (leaf == "mlp.dense_4h_to_h.weight")              { want2d(hp.n_ff, E); L.down = upload_weight(m, type, data, hp.n_ff, E);  m->weight_bytes += L.down.bytes; return 0; }
    // norms: 7B has input_layernorm (shared); 40B/180B have ln_mlp (-> MLP input) and ln_attn (-> QKV input), libfalcon.cpp:1845-1855
    if (leaf == "input_layernorm.weight" || leaf == "ln_mlp.weight") { want_f32(); L.ln_w  = upload_f32(m, data, E); return 0; }
    if (leaf == "input_layernorm.bias"   || leaf == "ln_mlp.bias")   { want_f32(); L.ln_b  = upload_f32(m, data, E); return 0; }
    if (leaf == "ln_attn.weight") { want_f32(); L.ln2_w = upload_f32(m, data, E); return 0; }
    if (leaf == "ln_attn.bias")   { want_f32(); L.ln2_b = upload_f32(m, data, E); return 0; }
    return -1;
}

extern "C" void falcon_hip_model_get_hparams(const falcon_hip_model * m, falcon_hip_hparams * hp_out) { *hp_out = m->hp; }
extern "C" size_t falcon_hip_model_weight_bytes(const falcon_hip_model * m) { return m->weight_bytes; }

static size_t act_col_bytes_max(int64_t K) {
    size_t b = fq_act_col_bytes(FQ_Q8_1, K), b0 = fq_act_col_bytes(FQ_Q8_0, K), bk = fq_act_col_bytes(FQ_Q8_K, K);
    if (b0 > b) b = b0;
    if (bk > b) b = bk;
    return b;
}
// the image view of `buf` for the activations that weight `w` consumes
static fq_act act_for(uint8_t * buf, const fq_weight & w, int64_t cols) {
    fq_act a{}; a.type = fq_desc(w.type).act_type; a.K = w.K; a.ncols = cols; a.base = buf; return a;
}

static std::atomic<int> g_live_contexts{0};       // the ring forms' schedule tables are shared by every context of the process and released with the last one
// the ring forms' per-shape schedules (device tables) must exist before any launch is captured into a graph: legacy formats kernels_ring.hip, k-quants kernels_ringk.hip
// (one schedule per distinct (format, shape) of the stage's blocks -- not only block 0's: launchers look plans up without creating them, and a block of another format
// would silently take the register-streaming kernel)
static bool ring_prepare_any(const falcon_hip_model * m) {
    bool any = false; int seen[32]; int n_seen = 0;
    for (const layer_weights & L : m->layers) {
        const fq_weight & q = L.qkv;
        bool dup = false;
        for (int i = 0; i < n_seen; ++i) dup = dup || seen[i] == q.type;
        if (dup) continue;
        if (n_seen < 32) seen[n_seen++] = q.type;
        const bool ok = fq_ring_prepare(q.type, m->hp.n_embd, m->hp.n_ff, q.M, fq_ctx().n_cu) || fq_ringk_prepare(q.type, m->hp.n_embd, m->hp.n_ff, q.M, fq_ctx().n_cu);
        any = any || ok;
    }
    return any;
}

// the ring form of the unmerged output mat-vec launch (k_ring_out, one wave per row in chunks): measured on Falcon-40B against k_gemv_out with the fast k-quant
// dots -- Q2_K 234 -> 246 tok/s, Q3_K 188.5 -> 192, but Q4_K 198 -> 181, Q5_K 157 -> 146, Q6_K 153 -> 142: on by default for the two formats
// with 64-element units only
static bool ring_out_auto(const falcon_hip_model * m) {
    if (const char * e = getenv("FALCON_HIP_RING_OUT")) return atoi(e) != 0;
    if (m->layers.empty()) return false;
    const int t = m->layers[0].down.type;
    return t == FQ_Q2_K || t == FQ_Q3_K;
}

static falcon_hip_context * context_create(falcon_hip_model * m, int n_ctx, int n_batch, int rope_n_ctx, int n_seq) {
    const falcon_hip_hparams & hp = m->hp;
    for (size_t i = 0; i < m->layers.size(); ++i) {
        const layer_weights & L = m->layers[i];
        if (!L.qkv.plane[0] || !L.wo.plane[0] || !L.up.plane[0] || !L.down.plane[0] || !L.ln_w || !L.ln_b || (hp.two_norms && (!L.ln2_w || !L.ln2_b))) {
            fprintf(stderr, "falcon-hip: layer %d is missing tensors\n", hp.layer_begin + (int) i); exit(1);
        }
    }
    if (m->first_stage() && !m->tok_emb.plane[0]) { fprintf(stderr, "falcon-hip: missing word_embeddings\n"); exit(1); }
    if (m->last_stage() && (!m->lm_head.plane[0] || !m->out_norm_w || !m->out_norm_b)) { fprintf(stderr, "falcon-hip: missing ln_f / lm_head\n"); exit(1); }

    falcon_hip_context * c = new falcon_hip_context();
    c->m = m; c->n_ctx = n_ctx; c->n_batch = n_batch; c->rope_n_ctx = rope_n_ctx > 0 ? rope_n_ctx : n_ctx; c->n_seq = n_seq;
    const int64_t E = hp.n_embd, D = 64, QKV = (int64_t)(hp.n_head + 2 * hp.n_head_kv) * D, FF = hp.n_ff, B = n_batch;
    const int64_t nl = (int64_t) m->layers.size();
    c->x      = (float *) dev_alloc(c->allocs, (size_t) B * E * 4);
    c->ln     = (float *) dev_alloc(c->allocs, (size_t) B * E * 4);
    c->ln2    = (float *) dev_alloc(c->allocs, (size_t) B * E * 4);
    c->qkv    = (float *) dev_alloc(c->allocs, (size_t) B * QKV * 4);
    c->att    = (float *) dev_alloc(c->allocs, (size_t) B * E * 4);
    // score rows of the long-prompt attention forms, owned here and sized once (kernels.h): no launch of this context allocates
    c->att_scratch.bytes = n_seq > 0 ? 0 : fq_attention_scratch_need(n_batch, hp.n_head, n_ctx, hp.n_head_kv);
    c->att_scratch.p     = c->att_scratch.bytes ? (float *) dev_alloc(c->allocs, c->att_scratch.bytes) : nullptr;
    c->wo_out = (float *) dev_alloc(c->allocs, (size_t) B * E * 4);
    c->up     = (float *) dev_alloc(c->allocs, (size_t) B * FF * 4);
    if (m->last_stage()) c->logits_dev = (float *) dev_alloc(c->allocs, (size_t) B * hp.n_vocab * 4);
    c->buf_e   = (uint8_t *) dev_alloc(c->allocs, act_col_bytes_max(E) * (size_t) B + 256);
    c->buf_e2  = (uint8_t *) dev_alloc(c->allocs, act_col_bytes_max(E) * (size_t) B + 256);
    c->buf_att = (uint8_t *) dev_alloc(c->allocs, act_col_bytes_max(E) * (size_t) B + 256);
    c->buf_ff  = (uint8_t *) dev_alloc(c->allocs, act_col_bytes_max(FF) * (size_t) B + 256);
    const size_t kvb = (size_t) nl * (n_seq > 0 ? n_seq : 1) * n_ctx * hp.n_head_kv * D * 4;
    c->k_cache = (float *) dev_alloc(c->allocs, kvb);
    c->v_cache = (float *) dev_alloc(c->allocs, kvb);
    HIP_CHECK(hipMemset(c->k_cache, 0, kvb ? kvb : 16));
    HIP_CHECK(hipMemset(c->v_cache, 0, kvb ? kvb : 16));
    {
        std::vector<float> cs = fq_rope_table_host((int) D, n_ctx, c->rope_n_ctx);
        c->rope_cs = (float *) dev_alloc(c->allocs, cs.size() * 4);
        HIP_CHECK(hipMemcpy(c->rope_cs, cs.data(), cs.size() * 4, hipMemcpyHostToDevice));
    }
    c->n_past_dev     = (int *) dev_alloc(c->allocs, 256);
    c->tokens_dev     = (int32_t *) dev_alloc(c->allocs, (size_t) B * 4 + 256);
    c->out_tokens_dev = (int32_t *) dev_alloc(c->allocs, (size_t) n_ctx * 4 + 256);
    HIP_CHECK(hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
    c->ev_fork.resize((size_t) nl); c->ev_join.resize((size_t) nl); c->ev_attn.resize((size_t) nl);
    for (int64_t i = 0; i < nl; ++i) { HIP_CHECK(hipEventCreateWithFlags(&c->ev_fork[i], hipEventDisableTiming)); HIP_CHECK(hipEventCreateWithFlags(&c->ev_join[i], hipEventDisableTiming)); HIP_CHECK(hipEventCreateWithFlags(&c->ev_attn[i], hipEventDisableTiming)); }
    if (const char * e = getenv("FALCON_HIP_PAR2_MAX_N")) c->par2_max_n = atoi(e);
    if (const char * e = getenv("FALCON_HIP_DUAL")) c->dual_stream = atoi(e) != 0;
    c->argmax_val     = (float *) dev_alloc(c->allocs, (size_t)(hp.n_vocab / 32 + 8) * 4);
    c->argmax_idx     = (int *) dev_alloc(c->allocs, (size_t)(hp.n_vocab / 32 + 8) * 4);
    c->sync_words     = (unsigned *) dev_alloc(c->allocs, 64 + 256);        // + the rope table row of the current position
    HIP_CHECK(hipMemset(c->sync_words, 0, 64 + 256));
    c->att_gran       = (unsigned long long *) dev_alloc(c->allocs, (size_t) hp.n_embd * 8 + 64);
    HIP_CHECK(hipMemset(c->att_gran, 0, (size_t) hp.n_embd * 8 + 64));
    c->x_gran         = (unsigned long long *) dev_alloc(c->allocs, (size_t) hp.n_embd * 8 + 64);
    HIP_CHECK(hipMemset(c->x_gran, 0, (size_t) hp.n_embd * 8 + 64));
    if (const char * e = getenv("FALCON_HIP_TWO_PHASE")) c->two_phase = atoi(e) != 0;
    if (const char * e = getenv("FALCON_HIP_STAGE_GRAPH")) c->stage_graph = atoi(e) != 0;
    c->ring_ln = !(getenv("FALCON_HIP_RING") && atoi(getenv("FALCON_HIP_RING")) == 0);      // (default on: a context starts in mode 2)
    if (const char * e = getenv("FALCON_HIP_PREFILL_GRAPH")) c->prefill_graph = atoi(e) != 0;
    ++g_live_contexts;
    if (c->ring_ln && nl > 0) c->ring_ln = ring_prepare_any(m);
    c->ring_out = ring_out_auto(m);
    if (const char * e = getenv("FALCON_HIP_MERGED")) c->merged_attn_out = atoi(e) != 0;
    return c;
}

extern "C" falcon_hip_context * falcon_hip_context_create(falcon_hip_model * m, int n_ctx, int n_batch, int rope_n_ctx) {
    return context_create(m, n_ctx, n_batch, rope_n_ctx, 0);
}
extern "C" falcon_hip_context * falcon_hip_context_create_seqs(falcon_hip_model * m, int n_ctx, int n_seq, int rope_n_ctx) {
    if (n_seq < 1 || n_seq > 256) { fprintf(stderr, "falcon-hip: a lock-step context holds 1..256 sequences, not %d\n", n_seq); return nullptr; }
    return context_create(m, n_ctx, n_seq, rope_n_ctx, n_seq > 1 ? n_seq : 0);
}
extern "C" int falcon_hip_context_n_seq(const falcon_hip_context * c) { return c->n_seq > 0 ? c->n_seq : 1; }

extern "C" void falcon_hip_context_free(falcon_hip_context * c) {
    if (!c) return;
    if (--g_live_contexts == 0) { HIP_CHECK(hipStreamSynchronize(fq_ctx().stream)); fq_ring_free_plans(); fq_ringk_free_plans(); }
    if (c->decode_graph) HIP_CHECK(hipGraphExecDestroy(c->decode_graph));
    if (c->step_graph) HIP_CHECK(hipGraphExecDestroy(c->step_graph));
    if (c->token_graph) HIP_CHECK(hipGraphExecDestroy(c->token_graph));
    for (auto & bg : c->batch_graphs) HIP_CHECK(hipGraphExecDestroy(bg.exec));
    for (hipEvent_t e : c->ev_fork) HIP_CHECK(hipEventDestroy(e));
    for (hipEvent_t e : c->ev_join) HIP_CHECK(hipEventDestroy(e));
    for (hipEvent_t e : c->ev_attn) HIP_CHECK(hipEventDestroy(e));
    if (c->side) HIP_CHECK(hipStreamDestroy(c->side));
    for (void * p : c->allocs) HIP_CHECK(hipFree(p));
    if (c->logits_pinned) HIP_CHECK(hipHostFree(c->logits_pinned));
    delete c;
}

extern "C" void falcon_hip_context_keep_hidden(falcon_hip_context * c, int keep) {
    c->keep_hidden = keep != 0;
    if (keep && !c->hidden_dev)
        c->hidden_dev = (float *) dev_alloc(c->allocs, (size_t)(c->m->layers.size() + 1) * c->n_batch * c->m->hp.n_embd * 4);
}
extern "C" void falcon_hip_get_hidden(falcon_hip_context * c, float * dst) {
    HIP_CHECK(hipStreamSynchronize(fq_ctx().stream));
    HIP_CHECK(hipMemcpy(dst, c->hidden_dev, (size_t)(c->m->layers.size() + 1) * c->hidden_tokens * c->m->hp.n_embd * 4, hipMemcpyDeviceToHost));
}
extern "C" void falcon_hip_context_use_graph(falcon_hip_context * c, int enable) { c->use_graph = enable != 0; }
extern "C" int falcon_hip_context_sync_error(falcon_hip_context * c) {      // 1 if a k_attn_out poll ever timed out (results invalid)
    unsigned w[2] = {0, 0};
    HIP_CHECK(hipDeviceSynchronize());
    HIP_CHECK(hipMemcpy(w, c->sync_words, sizeof w, hipMemcpyDeviceToHost));
    return (int) w[1];
}
extern "C" void falcon_hip_context_set_fused(falcon_hip_context * c, int mode) {      // 0 op list, 1 three launches per block, 2 two (default), 3 one (two-phase), 5 two with the ring form forced; 4 was the persistent engine (removed in round 6, NOTEBOOK section 4): now the default form
    if (c->decode_graph) { HIP_CHECK(hipGraphExecDestroy(c->decode_graph)); c->decode_graph = nullptr; }
    if (c->step_graph) { HIP_CHECK(hipGraphExecDestroy(c->step_graph)); c->step_graph = nullptr; }
    if (c->token_graph) { HIP_CHECK(hipGraphExecDestroy(c->token_graph)); c->token_graph = nullptr; }
    for (auto & bg : c->batch_graphs) HIP_CHECK(hipGraphExecDestroy(bg.exec));
    c->batch_graphs.clear();
    c->fused_decode = mode != 0;
    c->merged_attn_out = mode >= 2;
    if (mode == 4) mode = 2;
    c->two_phase = mode == 3;
    // the LayerNorm mat-vec launch in the ring form (kernels_ring.hip): mode 5 always, mode 2 unless FALCON_HIP_RING=0 (measured +1.6 % on
    // Falcon-7B Q4_0; legacy formats only, other models keep k_gemv_ln)
    static const bool ring_default = !(getenv("FALCON_HIP_RING") && atoi(getenv("FALCON_HIP_RING")) == 0);
    c->ring_ln = (mode == 5 || (mode == 2 && ring_default)) && !c->m->layers.empty() && ring_prepare_any(c->m);
    c->ring_out = mode == 5 || (mode == 2 && ring_out_auto(c->m));
}

static bool stage_uniform(const falcon_hip_model * m) {
    for (const layer_weights & L : m->layers) if (L.qkv.type != L.up.type || L.down.type != L.wo.type) return false;
    return true;
}
// THE predicate for "an N = 1 step of this context runs the fused decode kernels" -- launch_stage, the greedy sampler and
// the pipeline step all ask here, so the sampler reads the per-workgroup argmax candidates exactly when the fused lm_head
// wrote them. The fused kernels are instantiated per weight format (a model that mixes formats inside a block takes the
// op list) and implement the default summation order only (ggml_hip_reference_order -> op list).
// Round 6: the FAST reference order (ggml_hip_reference_order(2)) has the fused launches too, for stages of one legacy format (fq_ref_chain.h).
static bool stage_legacy(const falcon_hip_model * m) {
    auto leg = [](int t) { return t == FQ_Q4_0 || t == FQ_Q4_1 || t == FQ_Q5_0 || t == FQ_Q5_1 || t == FQ_Q8_0; };
    if (m->layers.empty() || m->hp.n_embd % 32 || m->hp.n_ff % 32) return false;
    const int t = m->layers[0].qkv.type;
    for (const layer_weights & L : m->layers) if (L.qkv.type != t || L.up.type != t || L.down.type != t || L.wo.type != t) return false;
    if (m->last_stage() && !leg(m->lm_head.type)) return false;
    return leg(t);
}
static bool stage_fused_ref(const falcon_hip_context * c) {        // N = 1 steps run the fused launches in the reference's association
    return c->fused_decode && fq_reference_fast() && stage_legacy(c->m) && !c->two_phase && !c->dual_stream;
}
static bool stage_fused(const falcon_hip_context * c) {
    if (stage_fused_ref(c)) return true;
    return c->fused_decode && stage_uniform(c->m) && !fq_reference_order() && !fq_attn_f64();
}
// everything a captured graph bakes in besides its pointers: a change invalidates decode_graph / step_graph
static int graph_signature(const falcon_hip_context * c) {
    return (stage_fused(c) ? 1 : 0) | (fq_reference_order() ? 2 : 0) | (fq_attn_f64() ? 4 : 0) | (c->merged_attn_out ? 8 : 0) |
           (c->two_phase ? 16 : 0) | (c->dual_stream ? 32 : 0) | (stage_fused_ref(c) ? 128 : 0);
}

// ------------------------------------------------------------------------------------------------ one eval
// Launches every kernel of this stage for N tokens. Inputs already in place: tokens_dev (first stage) or x, and
// n_past_dev. max_n_kv bounds n_past + N for LDS sizing.
// lock-step contexts of up to this many sequences run the column mat-vec kernels in chunks of 4 (FALCON_HIP_COLS_MAX_N overrides): 4 for the
// legacy formats, whose streaming small-batch mat-mul serves 5..16 columns in less time than two chunks (Falcon-7B Q4_0: 2.9 ms per pass
// against 3.6 at 8 sequences), 12 for k-quant models whose matrices are not at model widths (they have only the tile GEMM beyond)
static int fq_cols_max_n(int wtype) {
    static const int v = getenv("FALCON_HIP_COLS_MAX_N") ? atoi(getenv("FALCON_HIP_COLS_MAX_N")) : 0;
    if (v > 0) return v;
    return (wtype == FQ_Q4_0 || wtype == FQ_Q4_1 || wtype == FQ_Q5_0 || wtype == FQ_Q5_1 || wtype == FQ_Q8_0) ? 4 : 12;
}
// k-quant models at model widths (fq_skinny_q4k_shape) have their own streaming forms (kernels_gemm_skinny_k.hip) and use them from 3 sequences up: a
// pass of 16 columns costs less there than the column kernels' pass of 4; contexts of 2 keep the column kernels and their bit-identity with a single stream
#define FQ_COLS_MAX_N (m->layers.empty() ? 4 : (fq_skinny_q4k_shape(m->layers[0].qkv) && !getenv("FALCON_HIP_COLS_MAX_N") ? 2 : fq_cols_max_n(m->layers[0].qkv.type)))
static void launch_stage(falcon_hip_context * c, int N, int max_n_kv, hipStream_t st) {
    falcon_hip_model * m = c->m;
    const falcon_hip_hparams & hp = m->hp;
    hip_context & hc = fq_ctx();
    const int64_t E = hp.n_embd, D = 64, H = hp.n_head, HKV = hp.n_head_kv, QKV = (H + 2 * HKV) * D, FF = hp.n_ff;
    const fq_gemv_epi store{ FQ_EPI_STORE, hc.gelu_table, nullptr, nullptr, 0 };

    if (m->first_stage()) fq_launch_dequant_rows(m->tok_emb, c->tokens_dev, N, c->x, st);    // ggml_get_rows, libfalcon.cpp:2120

    // the fused kernels are instantiated per weight format: a model that mixes formats inside a block (e.g. the reference's
    // Q4_K_M for Falcon-7B: only the 18176-wide Wdown can hold 256-element super-blocks) takes the op list
    const int64_t seq_stride = c->n_seq > 0 ? (int64_t) c->n_ctx * HKV * D : 0;             // lock-step sequences: one KV cache per row
    const int64_t n_caches = c->n_seq > 0 ? c->n_seq : 1;
    if (c->n_seq > 0 && N != c->n_seq) { fprintf(stderr, "falcon-hip: a context of %d lock-step sequences evaluates %d rows per step, not %d\n", c->n_seq, c->n_seq, N); exit(1); }
    if (N == 1 && stage_fused(c)) {
        // ---- fused single-token path (kernels_decode.hip), bit-identical to the op list below. Per block, by mode:
        //   3 launches  k_gemv_ln | k_attn_decode | k_gemv_out
        //   2 launches  k_gemv_ln | k_attn_out                        (attention inside the output mat-vec launch)
        //   1 launch    k_attn_out_ln = k_attn_out of block l + k_gemv_ln of block l+1 (or ln_f + lm_head) as a second phase
        const bool prof = fq_prof_active();
        const bool dual = c->dual_stream && !prof;
        const bool ref = stage_fused_ref(c);                    // the reference's association on the same launches (fq_ref_chain.h)
        auto ln_args = [&](size_t li) {
            const layer_weights & L = m->layers[li];
            const int ff_act = fq_desc(L.down.type).act_type;
            const bool quant_epi = (ff_act == FQ_Q8_0 || ff_act == FQ_Q8_1);
            fq_gemv_ln_args ga{};
            ga.x = c->x; ga.E = E; ga.nseg = 2; ga.gelu_table = hc.gelu_table; ga.dbg = hc.dbg_stamps;
            ga.epoch_word = c->merged_attn_out ? c->sync_words : nullptr;
            if (c->merged_attn_out) { ga.n_past_ptr = c->n_past_dev; ga.rope_cs = c->rope_cs; ga.rope_cur = (float *)(c->sync_words + 16); }
            ga.seg[0] = { L.qkv, hp.two_norms ? L.ln2_w : L.ln_w, hp.two_norms ? L.ln2_b : L.ln_b, FQ_LNEPI_STORE, c->qkv, nullptr, 0, 0 };
            ga.seg[1] = { L.up, L.ln_w, L.ln_b, quant_epi ? FQ_LNEPI_GELU_QUANT : FQ_LNEPI_GELU_STORE, c->up, c->buf_ff, ff_act, 0 };
            return ga;
        };
        auto head_args = [&]() {
            fq_gemv_ln_args ga{};
            ga.x = c->x; ga.E = E; ga.nseg = 1; ga.gelu_table = hc.gelu_table;
            ga.epoch_word = c->merged_attn_out ? c->sync_words : nullptr;
            ga.seg[0] = { m->lm_head, m->out_norm_w, m->out_norm_b, FQ_LNEPI_STORE, c->logits_dev, nullptr, 0, 0 };
            ga.argmax_val = c->argmax_val; ga.argmax_idx = c->argmax_idx;
            return ga;
        };
        bool ln_done = false;                                   // this block's k_gemv_ln already ran as the previous launch's second phase
        bool head_done = false;
        for (size_t li = 0; li < m->layers.size(); ++li) {
            const layer_weights & L = m->layers[li];
            if (c->keep_hidden) HIP_CHECK(hipMemcpyAsync(c->hidden_dev + li * (size_t) E, c->x, (size_t) E * 4, hipMemcpyDeviceToDevice, st));
            const int ff_act = fq_desc(L.down.type).act_type;
            const bool quant_epi = (ff_act == FQ_Q8_0 || ff_act == FQ_Q8_1);
            const fq_gemv_ln_args ga = ln_args(li);
            if (dual) {
                // the MLP-up mat-vec only needs x: run it on the side stream while QKV + attention run on the main one
                fq_gemv_ln_args gu = ga; gu.nseg = 1; gu.seg[0] = ga.seg[1]; gu.seg[0].block_begin = 0;
                fq_gemv_ln_args gq = ga; gq.nseg = 1;
                HIP_CHECK(hipEventRecord(c->ev_fork[li], st));
                HIP_CHECK(hipStreamWaitEvent(c->side, c->ev_fork[li], 0));
                fq_launch_gemv_ln(gu, hc.n_cu, c->side);
                if (!quant_epi) fq_launch_quantize_act(c->up, FF, act_for(c->buf_ff, L.down, 1), c->side);
                HIP_CHECK(hipEventRecord(c->ev_join[li], c->side));
                fq_launch_gemv_ln(gq, hc.n_cu, st);
            } else if (!ln_done) {
                if (prof) fq_prof_open(st);
                bool ring = false;
                if (c->ring_ln && quant_epi) ring = fq_launch_gemv_ln_ring(ga, c->sync_words + 1, hc.n_cu, st, ref);
                else if (c->ring_ln && !ref) ring = fq_launch_ringk_ln(ga, c->sync_words + 1, hc.n_cu, st);      // k-quants: GELU stored as f32
                if (!ring) fq_launch_gemv_ln(ga, hc.n_cu, st, ref);
                if (prof) fq_prof_close(st, (double)(L.qkv.bytes + L.up.bytes));
            }
            ln_done = false;
            float * kc = c->k_cache + li * (size_t) n_caches * c->n_ctx * HKV * D;
            float * vc = c->v_cache + li * (size_t) n_caches * c->n_ctx * HKV * D;
            const int att_act = fq_desc(L.wo.type).act_type;
            const bool att_q = (att_act == FQ_Q8_0 || att_act == FQ_Q8_1);      // the head's 64 outputs = two 32-blocks
            fq_gemv_out_args go{ L.down, L.wo, c->buf_ff, c->att, att_q ? c->buf_att : nullptr, c->x, c->x,
                                 hc.dbg_stamps ? hc.dbg_stamps + 4096 * 8 : nullptr };
            if (dual) HIP_CHECK(hipStreamWaitEvent(st, c->ev_join[li], 0));
            // gelu(up) still f32 (k-quant consumers: a Q8_K block spans 256 rows, i.e. several workgroups of the launch above): quantized by its own launch
            // ahead of the merged form, or by extra workgroups of the attention launch (the rider) in the three-launch form
            bool ff_pending = !quant_epi && !dual;
            bool merged = false;
            if (ref && c->merged_attn_out) {
                // the merged launch in the reference's association; where its grid does not fit the chip: attention (f64 dots) + output launch below
                if (prof) fq_prof_open(st);
                merged = fq_launch_attn_out_ref(go, c->qkv, (int) H, (int) HKV, c->n_past_dev, max_n_kv, c->rope_cs, (const float *)(c->sync_words + 16),
                                                kc, vc, hc.exp_table_attn, att_act, c->att_gran, c->sync_words, c->sync_words + 1, hc.n_cu, st);
                if (prof) { if (merged) fq_prof_close(st, (double)(L.down.bytes + L.wo.bytes)); else fq_prof_cancel(); }
            }
            const bool will_merge = !ref && c->merged_attn_out && !dual && fq_attn_out_fits(go, (int) H, max_n_kv, hc.n_cu);
            if (ff_pending && will_merge) { fq_launch_quantize_act(c->up, FF, act_for(c->buf_ff, L.down, 1), st); ff_pending = false; }
            if (will_merge) {
                // second phase: the next block's k_gemv_ln (its GELU output must be quantized in its epilogue: a separate
                // quantizer launch cannot sit between the phases), or ln_f + lm_head after the last block
                fq_gemv_ln_args next{}; bool have_next = false; double next_bytes = 0.0;
                if (c->two_phase && !c->keep_hidden) {
                    if (li + 1 < m->layers.size()) {
                        const layer_weights & Ln = m->layers[li + 1];
                        const int a2 = fq_desc(Ln.down.type).act_type;
                        if (a2 == FQ_Q8_0 || a2 == FQ_Q8_1) { next = ln_args(li + 1); have_next = true; next_bytes = (double)(Ln.qkv.bytes + Ln.up.bytes); }
                    } else if (m->last_stage()) { next = head_args(); have_next = true; next_bytes = (double) m->lm_head.bytes; }
                }
                if (prof) fq_prof_open(st);
                if (have_next) {
                    merged = fq_launch_attn_out(go, c->qkv, (int) H, (int) HKV, c->n_past_dev, max_n_kv, c->rope_cs, (const float *)(c->sync_words + 16),
                                                kc, vc, hc.exp_table_attn, att_act, c->att_gran, c->sync_words, c->sync_words + 1, hc.n_cu, st, &next, c->x_gran);
                    if (merged) {
                        if (li + 1 < m->layers.size()) ln_done = true; else head_done = true;
                        if (prof) { fq_prof_close(st, (double)(L.down.bytes + L.wo.bytes) + next_bytes); }
                    }
                }
                if (!merged) {
                    merged = fq_launch_attn_out(go, c->qkv, (int) H, (int) HKV, c->n_past_dev, max_n_kv, c->rope_cs, (const float *)(c->sync_words + 16),
                                                kc, vc, hc.exp_table_attn, att_act, c->att_gran, c->sync_words, c->sync_words + 1, hc.n_cu, st);
                    if (merged && prof) fq_prof_close(st, (double)(L.down.bytes + L.wo.bytes));
                }
                if (!merged && prof) fq_prof_cancel();
                if (!merged) { fprintf(stderr, "falcon-hip: the merged attention + output launch refused a shape fq_attn_out_fits accepted\n"); exit(1); }
            }
            if (!merged) {
                const fq_act a_ff1 = act_for(c->buf_ff, L.down, 1);
                static const bool ride_on = !(getenv("FALCON_HIP_QUANT_RIDER") && atoi(getenv("FALCON_HIP_QUANT_RIDER")) == 0);
                if (ff_pending && ride_on && a_ff1.type == FQ_Q8_K && FF % 256 == 0) {
                    // attention of the one sequence + the rider: k_attn_decode's code (k_attn_decode_seqs at one sequence), gelu(up)'s Q8_K image by extra workgroups
                    fq_launch_attn_decode_seqs(c->qkv, 1, (int) H, (int) HKV, c->n_past_dev, max_n_kv, c->rope_cs, kc, vc, 0, hc.exp_table_attn,
                                               att_q ? nullptr : c->att, att_q ? c->buf_att : nullptr, att_act, (int64_t) fq_act_col_bytes(att_act, E), st, c->up, FF, &a_ff1);
                    ff_pending = false;
                } else {
                    if (ff_pending) { fq_launch_quantize_act(c->up, FF, a_ff1, st); ff_pending = false; }
                    fq_launch_attn_decode(c->qkv, (int) H, (int) HKV, c->n_past_dev, max_n_kv, c->rope_cs, kc, vc, hc.exp_table_attn,
                                          att_q ? nullptr : c->att, att_q ? c->buf_att : nullptr, att_act, st, ref);
                }
                if (prof) fq_prof_open(st);
                if (ref) fq_launch_gemv_out(go, hc.n_cu, st, true);
                else if (!(c->ring_out && fq_launch_ring_out(go, c->sync_words + 1, hc.n_cu, st))) fq_launch_gemv_out(go, hc.n_cu, st);
                if (prof) fq_prof_close(st, (double)(L.down.bytes + L.wo.bytes));
            }
        }
        if (c->keep_hidden) {
            HIP_CHECK(hipMemcpyAsync(c->hidden_dev + m->layers.size() * (size_t) E, c->x, (size_t) E * 4, hipMemcpyDeviceToDevice, st));
            c->hidden_tokens = 1;
        }
        if (m->last_stage() && !head_done) {
            const fq_gemv_ln_args ga = head_args();
            if (prof) fq_prof_open(st);
            fq_launch_gemv_ln(ga, hc.n_cu, st, ref);
            if (prof) fq_prof_close(st, (double) m->lm_head.bytes);
        }
        return;
    }
    // The LayerNorm(s) in front of a block (and the output norm) come out of the PREVIOUS block's residual-sum launch where that is a launch of its own (the short prompts'
    // two-stream form below): norm_of(li) = what block li (li == #blocks: the head) needs, ln_done = the images are there
    auto norm_of = [&](size_t li, fq_next_norm & nn) -> bool {
        nn = fq_next_norm{};
        if (li == m->layers.size()) {
            if (!m->last_stage()) return false;
            nn.w0 = m->out_norm_w; nn.b0 = m->out_norm_b; nn.a0 = act_for(c->buf_e, m->lm_head, N);
            return fq_add2_ln_ok(nn, E);
        }
        const layer_weights & Ln = m->layers[li];
        nn.w0 = Ln.ln_w; nn.b0 = Ln.ln_b; nn.a0 = act_for(c->buf_e, Ln.up, N);
        if (hp.two_norms) { nn.w1 = Ln.ln2_w; nn.b1 = Ln.ln2_b; nn.a1 = act_for(c->buf_e2, Ln.qkv, N); return fq_add2_ln_ok(nn, E); }
        return fq_desc(Ln.qkv.type).act_type == nn.a0.type && fq_add2_ln_ok(nn, E);      // (one norm, two activation families: two images of it -- the launches below)
    };
    bool ln_done = false;
    for (size_t li = 0; li < m->layers.size(); ++li) {
        const layer_weights & L = m->layers[li];
        if (c->keep_hidden) HIP_CHECK(hipMemcpyAsync(c->hidden_dev + li * (size_t) N * E, c->x, (size_t) N * E * 4, hipMemcpyDeviceToDevice, st));
        const fq_act a_up = act_for(c->buf_e, L.up, N);
        fq_act a_qkv = a_up;
        if (hp.two_norms) {                                                              // ln_mlp and ln_attn: one launch when both images are of one type
            a_qkv = act_for(c->buf_e2, L.qkv, N);
            if (ln_done) {
            } else if (!fq_launch_layer_norm_quant2(c->x, E, N, L.ln_w, L.ln_b, a_up, L.ln2_w, L.ln2_b, a_qkv, st)) {
                fq_launch_layer_norm_quant(c->x, E, N, L.ln_w, L.ln_b, nullptr, a_up, st);
                fq_launch_layer_norm_quant(c->x, E, N, L.ln2_w, L.ln2_b, nullptr, a_qkv, st);
            }
        } else if (!ln_done) fq_launch_layer_norm_quant(c->x, E, N, L.ln_w, L.ln_b, nullptr, a_up, st);      // (the f32 row is not needed)
        ln_done = false;
        if (hp.two_norms) {
        } else if (fq_desc(L.qkv.type).act_type != a_up.type) {                          // same norm, the other activation family
            a_qkv = act_for(c->buf_e2, L.qkv, N);
            fq_launch_layer_norm_quant(c->x, E, N, L.ln_w, L.ln_b, nullptr, a_qkv, st);
        }
        float * kc = c->k_cache + li * (size_t) n_caches * c->n_ctx * HKV * D;
        float * vc = c->v_cache + li * (size_t) n_caches * c->n_ctx * HKV * D;
        const fq_act a_att = act_for(c->buf_att, L.wo, N), a_ff = act_for(c->buf_ff, L.down, N);
        const bool att_q = (a_att.type == FQ_Q8_0 || a_att.type == FQ_Q8_1);
        // 2..4 lock-step sequences: the block's weights in TWO launches that serve every column (kernels_cols.hip), the
        // decode attention of all sequences in one launch between them -- same bits as the generic launches below
        // up to FQ_COLS_MAX_N (8) sequences in chunks of 4 columns; beyond that one pass of the streaming small-batch mat-mul
        // (kernels_gemm_skinny.hip, N <= 16) is cheaper: measured on Falcon-7B Q4_0 3.6 ms per pass of 8 in two chunks, 4.0 ms for 12
        // and 4.1 ms for 16 in one pass (the 128-token tile GEMM beyond 16: 8.3-9.2 ms for any N up to 64)
        const int cw_out = (L.down.type == L.wo.type) ? fq_gemv_out_cols_width(L.wo.type, FF, E) : 0;      // Falcon-40B width: 2 columns per output launch
        const bool cols_path = seq_stride && N >= 2 && N <= FQ_COLS_MAX_N && c->fused_decode && !fq_reference_order() && !fq_attn_f64() &&
                               L.qkv.type == L.up.type && cw_out > 0;
        bool up_done = false, ff_quantized = false;
        if (cols_path) {
            const bool quant_epi = (a_ff.type == FQ_Q8_0 || a_ff.type == FQ_Q8_1) && FF % 32 == 0;
            up_done = true;
            for (int c0 = 0; c0 < N && up_done; c0 += 4) {
                fq_gemv_cols_args ga{};
                ga.nseg = 2; ga.ncols = N - c0 < 4 ? N - c0 : 4; ga.gelu_table = hc.gelu_table;
                ga.seg[0] = { L.qkv, a_qkv.base + (size_t) c0 * fq_act_col_bytes(a_qkv.type, E), FQ_LNEPI_STORE, c->qkv + (size_t) c0 * QKV, QKV, nullptr, 0, 0 };
                ga.seg[1] = { L.up, a_up.base + (size_t) c0 * fq_act_col_bytes(a_up.type, E), quant_epi ? FQ_LNEPI_GELU_QUANT : FQ_LNEPI_GELU_STORE,
                              c->up + (size_t) c0 * FF, FF, c->buf_ff + (size_t) c0 * fq_act_col_bytes(a_ff.type, FF), a_ff.type, 0 };
                up_done = fq_launch_gemv_cols(ga, hc.n_cu, st) || (c0 > 0 && (fprintf(stderr, "falcon-hip: column mat-vec refused a later chunk\n"), exit(1), false));
            }
            ff_quantized = up_done && quant_epi;
        }
        // Batched evaluation (N > 4): a block's two branches -- {Wqkv, RoPE, attention, Wo} and {Wup + GELU, Wdown} -- only meet in
        // the residual sum, and three of the four mat-muls of a short prompt launch fewer workgroups than the chip has CUs (142
        // for a 4544-row matrix and 128 tokens). They run on two streams: the MLP branch on the side stream, its Wdown (whose
        // epilogue adds Wo's result and the residual) after the attention branch. Same kernels, same bits. Measured in one
        // process on the same resident Falcon-7B Q4_0 (scripts/gpu_par2_ab.py), one stream -> two branches: 16 tokens 8.66 -> 7.83
        // ms, 32: 9.17 -> 8.13, 128: 9.41 -> 8.82, 512: 26.75 -> 24.59, 1024: 49.3 -> 47.3, 2048: 104.9 -> 101.7.
        // (not where the mat-muls are passes of the Q4_K small-batch form: they fill the chip and share one partial-sum scratch)
        auto passes_of = [&](const fq_weight & w) { return N <= fq_skinny_kq_max_cols(w.type) && fq_skinny_q4k_shape(w); };      // per matrix: a block may mix k-quant formats with different column limits
        const bool q4k_passes = passes_of(L.qkv) || passes_of(L.up) || passes_of(L.wo) || passes_of(L.down);
        // (round 6: lock-step passes of more than 32 sequences as well -- the same launches of ~50-100 us each, the decode attention of all sequences in the attention branch;
        // FALCON_HIP_PAR2_SEQS=0: those in stream order, as before)
        // (from 17 tokens for prompts -- 32 tokens 4.86 -> 4.70 ms, 20: 4.70 -> 4.62 --, from 33 sequences for lock-step passes, whose launches sit in the step's hipGraph:
        // 32 per pass 4.78 -> 5.0 ms with it; profiles/r06zu_ab_par2_min_n.txt. FALCON_HIP_PAR2_MIN_N=n: both from n)
        static const int par2_min_env = getenv("FALCON_HIP_PAR2_MIN_N") ? atoi(getenv("FALCON_HIP_PAR2_MIN_N")) : 0;
        const int par2_min_n = par2_min_env ? par2_min_env : (seq_stride ? 33 : 17);
        static const bool par2_seqs = !(getenv("FALCON_HIP_PAR2_SEQS") && atoi(getenv("FALCON_HIP_PAR2_SEQS")) == 0);
        const bool par2 = !cols_path && (!seq_stride || par2_seqs) && N >= par2_min_n && N <= c->par2_max_n && !q4k_passes && !fq_prof_active() && !fq_ctx().dbg_stamps;
        if (par2) {
            HIP_CHECK(hipEventRecord(c->ev_fork[li], st));
            HIP_CHECK(hipStreamWaitEvent(c->side, c->ev_fork[li], 0));
            const fq_gemv_epi gelu{ FQ_EPI_GELU, hc.gelu_table, nullptr, nullptr, 0 };
            fq_mul_mat_q_acts(L.up, a_up, N, c->up, FF, gelu, c->side);
            fq_launch_quantize_act(c->up, FF, a_ff, c->side);
            fq_mul_mat_q_acts(L.qkv, a_qkv, N, c->qkv, QKV, store, st);
            if (seq_stride && !fq_reference_order() && !fq_attn_f64()) {
                fq_launch_attn_decode_seqs(c->qkv, N, (int) H, (int) HKV, c->n_past_dev, max_n_kv, c->rope_cs, kc, vc, seq_stride, hc.exp_table_attn,
                                           att_q ? nullptr : c->att, att_q ? c->buf_att : nullptr, a_att.type, (int64_t) fq_act_col_bytes(a_att.type, E), st, nullptr, FF, nullptr);
                if (!att_q) fq_launch_quantize_act(c->att, E, a_att, st);
            } else {
                fq_launch_rope_kv(c->qkv, N, (int) H, (int) HKV, (int) D, c->n_past_dev, c->rope_cs, kc, vc, st, seq_stride);
                fq_launch_attention(c->qkv, N, (int) H, (int) HKV, (int) D, c->n_past_dev, max_n_kv, kc, vc, hc.exp_table_attn, c->att, st, seq_stride, &c->att_scratch);
                fq_launch_quantize_act(c->att, E, a_att, st);
            }
            // (round 6) Short prompts: neither branch waits for the other to carry the residual sum in an epilogue. Wdown -- the long launch of a short prompt (K = 4 n_embd,
            // 142 workgroups on 256 CUs at 128 tokens) -- follows Wup on the side stream and leaves its result in the (free again) f32 Wup matrix, Wo leaves its own, and
            // x = (down + wo) + x (libfalcon.cpp:2399-2400, the same additions in the same order) is a small launch behind the join.
            // Measured on MI355X, Falcon-7B Q4_0, epilogue form -> this (scripts/gpu_prompt_lengths.py, A/B/A/B on one box, profiles/r06zm_*): 40 tokens 6.80 -> 6.25 ms,
            // 64: 6.88 -> 6.40, 128: 8.17 -> 7.03 (18.2 k tok/s), 256: 14.3 -> 12.2, 512: 23.08 -> 23.00, 1024: 42.1 -> 41.6 -- but 384: 17.5 -> 18.2: Wdown's 426 workgroups
            // are 1.7 rounds of the chip there and the attention branch beside them only stretches the second. Hence: Wdown's launch at most 1.25 rounds or at least 2.
            // FALCON_HIP_PAR2_DOWN_FIRST=0: never (the epilogue form below), 2: always
            static const int down_first = getenv("FALCON_HIP_PAR2_DOWN_FIRST") ? atoi(getenv("FALCON_HIP_PAR2_DOWN_FIRST")) : 1;
            const int64_t down_rows = fq_gemm_wg_rows(L.down.type, fq_form_rows(L.down), N, hc.n_cu);
            const int64_t down_wgs = ((L.down.M + down_rows - 1) / down_rows) * ((N + 127) / 128);
            if ((E & 3) == 0 && (down_first == 2 || (down_first == 1 && (4 * down_wgs <= 5 * (int64_t) hc.n_cu || down_wgs >= 2 * (int64_t) hc.n_cu)))) {
                fq_mul_mat_q_acts(L.down, a_ff, N, c->up, E, store, c->side);
                HIP_CHECK(hipEventRecord(c->ev_join[li], c->side));
                fq_mul_mat_q_acts(L.wo, a_att, N, c->wo_out, E, store, st);
                HIP_CHECK(hipStreamWaitEvent(st, c->ev_join[li], 0));
                fq_next_norm nn;
                if (norm_of(li + 1, nn)) { fq_launch_add2_ln(c->x, c->up, c->wo_out, E, N, nn, st); ln_done = true; }
                else fq_launch_add2_inplace(c->x, c->up, c->wo_out, (int64_t) N * E, st);
                continue;
            }
            fq_mul_mat_q_acts(L.wo, a_att, N, c->wo_out, E, store, st);
            HIP_CHECK(hipEventRecord(c->ev_attn[li], st));
            HIP_CHECK(hipStreamWaitEvent(c->side, c->ev_attn[li], 0));
            const fq_gemv_epi resid{ FQ_EPI_ADD2, hc.gelu_table, c->wo_out, c->x, E };          // x = (down + wo) + x, in place
            fq_mul_mat_q_acts(L.down, a_ff, N, c->x, E, resid, c->side);
            HIP_CHECK(hipEventRecord(c->ev_join[li], c->side));
            HIP_CHECK(hipStreamWaitEvent(st, c->ev_join[li], 0));
            continue;
        }
        // Wqkv and Wup behind the same LayerNorm image (one-norm blocks): one launch of the small-batch mat-mul for both (5..16 columns)
        bool pair_done = false;
        if (!up_done && a_qkv.base == a_up.base && a_qkv.type == a_up.type) {
            const fq_gemv_epi gelu{ FQ_EPI_GELU, hc.gelu_table, nullptr, nullptr, 0 };
            pair_done = fq_mul_mat_q_acts_pair(L.qkv, L.up, a_up, N, c->qkv, QKV, store, c->up, FF, gelu, st);
        }
        const int min_cols = seq_stride ? 3 : 5;                             // (lock-step contexts: the k-quant small-batch forms from 3 sequences up)
        if (!up_done && !pair_done) { if (seq_stride) fq_mul_mat_q_acts_from3(L.qkv, a_qkv, N, c->qkv, QKV, store, st); else fq_mul_mat_q_acts(L.qkv, a_qkv, N, c->qkv, QKV, store, st); }
        if (seq_stride && !fq_reference_order() && !fq_attn_f64()) {
            // lock-step sequences: RoPE, KV append, attention and (Q8_0 / Q8_1 consumers) the activation image of all N tokens in
            // one launch of the decode attention (k_attn_decode's code: the same bits as the three launches below)
            // (Wup's GELU output exists already when both mat-muls went in one launch: its quantizer rides on this launch)
            static const bool ride_on = !(getenv("FALCON_HIP_QUANT_RIDER") && atoi(getenv("FALCON_HIP_QUANT_RIDER")) == 0);
            const bool ride = ride_on && pair_done && !ff_quantized && (a_ff.type == FQ_Q8_0 || a_ff.type == FQ_Q8_1);
            fq_launch_attn_decode_seqs(c->qkv, N, (int) H, (int) HKV, c->n_past_dev, max_n_kv, c->rope_cs, kc, vc, seq_stride, hc.exp_table_attn,
                                       att_q ? nullptr : c->att, att_q ? c->buf_att : nullptr, a_att.type, (int64_t) fq_act_col_bytes(a_att.type, E), st,
                                       ride ? c->up : nullptr, FF, ride ? &a_ff : nullptr);
            if (ride) ff_quantized = true;
            if (!att_q) fq_launch_quantize_act(c->att, E, a_att, st);
        } else {
            fq_launch_rope_kv(c->qkv, N, (int) H, (int) HKV, (int) D, c->n_past_dev, c->rope_cs, kc, vc, st, seq_stride);
            fq_launch_attention(c->qkv, N, (int) H, (int) HKV, (int) D, c->n_past_dev, max_n_kv, kc, vc, hc.exp_table_attn, c->att, st, seq_stride, &c->att_scratch);
            fq_launch_quantize_act(c->att, E, a_att, st);
        }
        if (!up_done && !pair_done) {
            // (Q4_K, 5..16 columns: GELU and Wdown's Q8_K image come out of the small-batch form's sum launch)
            if (fq_mul_mat_q_acts_gelu_q8k(L.up, a_up, N, c->up, FF, a_ff, st, min_cols)) ff_quantized = true;
            else {
                const fq_gemv_epi gelu{ FQ_EPI_GELU, hc.gelu_table, nullptr, nullptr, 0 };
                if (seq_stride) fq_mul_mat_q_acts_from3(L.up, a_up, N, c->up, FF, gelu, st); else fq_mul_mat_q_acts(L.up, a_up, N, c->up, FF, gelu, st);
            }
        }
        if (!ff_quantized) fq_launch_quantize_act(c->up, FF, a_ff, st);
        bool out_done = false;
        if (cols_path) {
            out_done = true;
            for (int c0 = 0; c0 < N && out_done; c0 += cw_out) {
                const fq_gemv_out_cols_args go{ L.down, L.wo, c->buf_ff + (size_t) c0 * fq_act_col_bytes(a_ff.type, FF), c->buf_att + (size_t) c0 * fq_act_col_bytes(a_att.type, E),
                                                c->x + (size_t) c0 * E, c->x + (size_t) c0 * E, E, N - c0 < cw_out ? N - c0 : cw_out };
                out_done = fq_launch_gemv_out_cols(go, hc.n_cu, st) || (c0 > 0 && (fprintf(stderr, "falcon-hip: column mat-vec refused a later chunk\n"), exit(1), false));
            }
        }
        if (!out_done && fq_mul_mat_q_acts_out2(L.wo, a_att, L.down, a_ff, N, c->x, E, st, min_cols)) out_done = true;      // (Q4_K, 5..16 columns: one sum launch for both)
        if (!out_done) {
            const fq_gemv_epi resid{ FQ_EPI_ADD2, hc.gelu_table, c->wo_out, c->x, E };          // x = (down + wo) + x, in place
            if (seq_stride) { fq_mul_mat_q_acts_from3(L.wo, a_att, N, c->wo_out, E, store, st); fq_mul_mat_q_acts_from3(L.down, a_ff, N, c->x, E, resid, st); }
            else            { fq_mul_mat_q_acts(L.wo, a_att, N, c->wo_out, E, store, st); fq_mul_mat_q_acts(L.down, a_ff, N, c->x, E, resid, st); }
        }
    }
    if (c->keep_hidden) {
        HIP_CHECK(hipMemcpyAsync(c->hidden_dev + m->layers.size() * (size_t) N * E, c->x, (size_t) N * E * 4, hipMemcpyDeviceToDevice, st));
        c->hidden_tokens = N;
    }
    if (m->last_stage()) {
        const fq_act a_head = act_for(c->buf_e, m->lm_head, N);
        if (!ln_done) fq_launch_layer_norm_quant(c->x, E, N, m->out_norm_w, m->out_norm_b, nullptr, a_head, st);
        bool head_done = false;
        if (seq_stride && N >= 2 && N <= FQ_COLS_MAX_N && c->fused_decode && !fq_reference_order()) {
            head_done = true;
            for (int c0 = 0; c0 < N && head_done; c0 += 4) {
                fq_gemv_cols_args ga{};
                ga.nseg = 1; ga.ncols = N - c0 < 4 ? N - c0 : 4; ga.gelu_table = hc.gelu_table;
                ga.seg[0] = { m->lm_head, a_head.base + (size_t) c0 * fq_act_col_bytes(a_head.type, E), FQ_LNEPI_STORE, c->logits_dev + (size_t) c0 * hp.n_vocab, hp.n_vocab, nullptr, 0, 0 };
                head_done = fq_launch_gemv_cols(ga, hc.n_cu, st) || (c0 > 0 && (fprintf(stderr, "falcon-hip: column mat-vec refused a later chunk\n"), exit(1), false));
            }
        }
        if (!head_done) { if (seq_stride) fq_mul_mat_q_acts_from3(m->lm_head, a_head, N, c->logits_dev, hp.n_vocab, store, st); else fq_mul_mat_q_acts(m->lm_head, a_head, N, c->logits_dev, hp.n_vocab, store, st); }   // all N rows, libfalcon.cpp:2440
    }
}

extern "C" int falcon_hip_eval_stage(falcon_hip_context * c, const int32_t * tokens, const float * hidden_in_dev, int N,
                                     int n_past, int logits_all, float * hidden_out_dev) {
    hip_context & hc = fq_ctx();
    falcon_hip_model * m = c->m;
    const int adv = c->n_seq > 0 ? 1 : N;                            // lock-step sequences: N rows = one position of each of N sequences
    if (N < 1 || N > c->n_batch || n_past < 0 || n_past + adv > c->n_ctx || (m->first_stage() && !tokens) || (!m->first_stage() && !hidden_in_dev)) {
        fprintf(stderr, "falcon-hip: eval of %d tokens at n_past %d: needs 1 <= n_tokens <= n_batch (%d), n_past + n_tokens <= n_ctx (%d) and its input\n", N, n_past, c->n_batch, c->n_ctx);
        return 1;                                                    // (falcon_eval's convention: non-zero = failed to eval, libfalcon.cpp:4588-4591)
    }
    if (m->first_stage()) {
        for (int i = 0; i < N; ++i) if (tokens[i] < 0 || tokens[i] >= m->hp.n_vocab) {
            fprintf(stderr, "falcon-hip: token id %d at position %d is outside [0, %d)\n", tokens[i], i, m->hp.n_vocab);
            return 2;
        }
    }
    if (c->sync_err_sticky) { fprintf(stderr, "falcon-hip: eval: an earlier step of this context lost an in-launch hand-off -- its KV cache is invalid, the context must be recreated\n"); return 3; }
    hipStream_t st = hc.stream;
    c->logits_pending = false;
    c->logits_last = nullptr;
    HIP_CHECK(hipMemcpyAsync(c->n_past_dev, &n_past, 4, hipMemcpyHostToDevice, st));
    if (m->first_stage()) HIP_CHECK(hipMemcpyAsync(c->tokens_dev, tokens, (size_t) N * 4, hipMemcpyHostToDevice, st));
    else                  HIP_CHECK(hipMemcpyAsync(c->x, hidden_in_dev, (size_t) N * m->hp.n_embd * 4, hipMemcpyDeviceToDevice, st));
    HIP_CHECK(hipStreamSynchronize(st));        // n_past / tokens may live on the caller's stack
    // batches: one hipGraph replay per (size, keys) instead of ~10 launches and 3 cross-stream joins per block from the host
    if (c->prefill_graph && N > 4 && c->n_seq == 0 && !fq_prof_active() && !fq_tl_collecting() && !hc.dbg_stamps && (!fq_reference_order() || fq_reference_fast())) {
        const int sig = (graph_signature(c) | (c->keep_hidden ? 256 : 0)) + 512 * fq_config_epoch();
        hipGraphExec_t exec = nullptr;
        for (const auto & bg : c->batch_graphs) if (bg.N == N && bg.max_kv == n_past + adv && bg.sig == sig) { exec = bg.exec; break; }
        if (!exec) {
            hipGraph_t gr;
            HIP_CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            launch_stage(c, N, n_past + adv, st);
            HIP_CHECK(hipStreamEndCapture(st, &gr));
            HIP_CHECK(hipGraphInstantiate(&exec, gr, nullptr, nullptr, 0));
            HIP_CHECK(hipGraphDestroy(gr));
            if (c->batch_graphs.size() >= 8) { HIP_CHECK(hipGraphExecDestroy(c->batch_graphs.front().exec)); c->batch_graphs.erase(c->batch_graphs.begin()); }
            c->batch_graphs.push_back({ N, n_past + adv, sig, exec });
        }
        HIP_CHECK(hipGraphLaunch(exec, st));
        if (c->keep_hidden) c->hidden_tokens = N;                  // (host-side state that launch_stage sets when it runs)
    } else {
        launch_stage(c, N, n_past + adv, st);
    }
    if (m->last_stage()) {
        const int64_t V = m->hp.n_vocab;
        if (logits_all) {
            c->logits_host.resize((size_t) N * V);
            HIP_CHECK(hipMemcpyAsync(c->logits_host.data(), c->logits_dev, (size_t) N * V * 4, hipMemcpyDeviceToHost, st));
        } else {                                                                         // libfalcon.cpp:2545-2547
            c->logits_host.resize((size_t) V);
            HIP_CHECK(hipMemcpyAsync(c->logits_host.data(), c->logits_dev + (size_t)(N - 1) * V, (size_t) V * 4, hipMemcpyDeviceToHost, st));
        }
        fetch_sync_error(c, st);
        HIP_CHECK(hipStreamSynchronize(st));
        return report_sync_error(c, "eval");
    } else if (hidden_out_dev) {
        HIP_CHECK(hipMemcpyAsync(hidden_out_dev, c->x, (size_t) N * m->hp.n_embd * 4, hipMemcpyDeviceToDevice, st));
    }
    return 0;
}

extern "C" int falcon_hip_eval(falcon_hip_context * c, const int32_t * tokens, int n_tokens, int n_past, int logits_all) {
    if (!c->m->first_stage() || !c->m->last_stage()) { fprintf(stderr, "falcon-hip: falcon_hip_eval needs the whole model in one process\n"); exit(1); }
    return falcon_hip_eval_stage(c, tokens, nullptr, n_tokens, n_past, logits_all, nullptr);
}

// falcon_hip_eval with the per-launch timing table printed on stderr afterwards (kernels.h fq_tl_*): the resident path's counterpart of the reference's
// --debug-timings node table (libfalcon.cpp:2506-2520 -> ggml_graph_print_impl, ggml.c:18266-18360). Plain launches (no graph replay), one synchronisation at the end.
extern "C" int falcon_hip_eval_debug_timings(falcon_hip_context * c, const int32_t * tokens, int n_tokens, int n_past, int logits_all) {
    fq_tl_begin();
    const int rc = falcon_hip_eval(c, tokens, n_tokens, n_past, logits_all);
    char title[160];
    snprintf(title, sizeof(title), "falcon-hip: launches of this eval (%d token%s at n_past %d; device-resident path, no ggml graph: one line per launch site)", n_tokens, n_tokens == 1 ? "" : "s", n_past);
    fq_tl_end(stderr, title);
    return rc;
}

extern "C" const float * falcon_hip_get_logits(falcon_hip_context * c) {
    if (c->logits_pending) {                                         // the last falcon_hip_eval_token's row copy (page-locked memory) is in flight behind its launches
        // (polled, not slept on: the caller samples the moment the row is there, and a blocking wait's wake-up costs tens of microseconds of a ~1 ms step)
        hipStream_t st = fq_ctx().stream;
        // bounded: ~20 ms of polling (a step is ~1 ms), then a blocking wait -- a stream that hangs must not burn a host core for ever
        bool done = false;
        for (int spin = 0; spin < 200000 && !done; ++spin) {
            const hipError_t e = hipStreamQuery(st);
            if (e == hipSuccess) done = true; else if (e != hipErrorNotReady) HIP_CHECK(e); else __builtin_ia32_pause();
        }
        if (!done) HIP_CHECK(hipStreamSynchronize(st));
        c->sync_err_host = *(const unsigned *)(c->logits_pinned + c->m->hp.n_vocab);
        c->logits_pending = false;
        (void) report_sync_error(c, "eval");                        // sticky: falcon_hip_context_last_error / the next eval report it (this call cannot)
    }
    return c->logits_last ? c->logits_last : c->logits_host.data();
}
// 0, or 3 when a step of this context lost an in-launch hand-off (every result since is invalid). Host state only: callers of
// falcon_hip_get_logits, which cannot fail, ask here after it.
extern "C" int falcon_hip_context_last_error(const falcon_hip_context * c) { return c->sync_err_sticky ? 3 : 0; }

__global__ void k_set_i32(int * p, int v);
__global__ void k_set2_i32(int * p, int v, int * q, int w);
// the decode attention keeps a score row of max_n_kv floats in LDS; a captured step is sized for the whole context (its position is read
// from device memory), so contexts longer than that buffer allows run their single-token steps as plain launches sized for n_past + 1
static bool fused_graph_fits(const falcon_hip_context * c) { return fq_attn_decode_lds_bytes(c->n_ctx) <= 160 * 1024; }
// One token at n_past (falcon_eval with n_tokens = 1, libfalcon.cpp:4566), asynchronous: the fused decode launches are replayed
// from a hipGraph (captured on first use, position read from device memory) and the logits row's copy into page-locked host memory is
// enqueued right behind them; the host does not wait -- falcon_hip_get_logits does, when (and only when) the caller asks for the row.
// Returns 0, 1 / 2 as falcon_hip_eval, or 3 once an earlier asynchronous step of this context has lost a hand-off.
extern "C" int falcon_hip_eval_token(falcon_hip_context * c, int32_t token, int n_past) {
    hip_context & hc = fq_ctx();
    falcon_hip_model * m = c->m;
    if (!m->first_stage() || !m->last_stage() || c->n_seq > 0) { fprintf(stderr, "falcon-hip: falcon_hip_eval_token needs the whole model in one process and a single sequence\n"); exit(1); }
    if (c->sync_err_sticky) { fprintf(stderr, "falcon-hip: eval: an earlier step of this context lost an in-launch hand-off -- its KV cache is invalid, the context must be recreated\n"); return 3; }
    if (n_past < 0 || n_past + 1 > c->n_ctx) { fprintf(stderr, "falcon-hip: eval of one token at n_past %d exceeds n_ctx %d\n", n_past, c->n_ctx); return 1; }
    if (token < 0 || token >= m->hp.n_vocab) { fprintf(stderr, "falcon-hip: token id %d is outside [0, %d)\n", token, m->hp.n_vocab); return 2; }
    hipStream_t st = hc.stream;
    if (c->logits_pending) {                                         // a row nobody asked for: its copy must land before the next one is enqueued into the same buffer
        HIP_CHECK(hipStreamSynchronize(st));
        c->sync_err_host = *(const unsigned *)(c->logits_pinned + m->hp.n_vocab);
        c->logits_pending = false;
        if (report_sync_error(c, "eval")) return 3;
    }
    if (!c->logits_pinned) HIP_CHECK(hipHostMalloc((void **) &c->logits_pinned, (size_t) m->hp.n_vocab * 4 + 64, hipHostMallocDefault));      // (+ the hand-off error word: a copy into pageable memory would make this call wait for the whole step)
    hipLaunchKernelGGL(k_set2_i32, dim3(1), dim3(1), 0, st, c->n_past_dev, n_past, (int *) c->tokens_dev, (int) token);
    const bool was_keep = c->keep_hidden;
    c->keep_hidden = false;
    if (stage_fused(c) && fused_graph_fits(c) && !fq_prof_active() && !hc.dbg_stamps && (!fq_reference_order() || fq_reference_fast())) {
        if (!c->token_graph || c->token_sig != graph_signature(c)) {
            if (c->token_graph) { HIP_CHECK(hipGraphExecDestroy(c->token_graph)); c->token_graph = nullptr; }
            hipGraph_t g;
            HIP_CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            launch_stage(c, 1, c->n_ctx, st);
            HIP_CHECK(hipStreamEndCapture(st, &g));
            HIP_CHECK(hipGraphInstantiate(&c->token_graph, g, nullptr, nullptr, 0));
            HIP_CHECK(hipGraphDestroy(g));
            c->token_sig = graph_signature(c);
        }
        HIP_CHECK(hipGraphLaunch(c->token_graph, st));
    } else {
        launch_stage(c, 1, n_past + 1, st);
    }
    c->keep_hidden = was_keep;
    HIP_CHECK(hipMemcpyAsync(c->logits_pinned, c->logits_dev, (size_t) m->hp.n_vocab * 4, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipMemcpyAsync(c->logits_pinned + m->hp.n_vocab, c->sync_words + 1, 4, hipMemcpyDeviceToHost, st));
    c->logits_last = c->logits_pinned;
    c->logits_pending = true;
    return 0;
}

// The n_ctx handed to ggml_rope may change from call to call (falcon_evaluation_config::n_max_real_ctx, libfalcon.cpp:2229-2230):
// the NTK factor only depends on n_ctx / 2048 (ggml.c:12880-12887), so the table is rebuilt -- in place, after the stream has
// drained -- only when that bucket changes.
extern "C" void falcon_hip_context_set_rope_n_ctx(falcon_hip_context * c, int rope_n_ctx) {
    if (rope_n_ctx <= 0) rope_n_ctx = c->n_ctx;
    const int old_b = c->rope_n_ctx >= 2048 ? c->rope_n_ctx / 2048 : 0, new_b = rope_n_ctx >= 2048 ? rope_n_ctx / 2048 : 0;
    c->rope_n_ctx = rope_n_ctx;
    if (old_b == new_b || !c->rope_cs) return;
    HIP_CHECK(hipStreamSynchronize(fq_ctx().stream));
    const std::vector<float> cs = fq_rope_table_host(64, c->n_ctx, rope_n_ctx);
    HIP_CHECK(hipMemcpy(c->rope_cs, cs.data(), cs.size() * 4, hipMemcpyHostToDevice));
}

// ------------------------------------------------------------------------------------------------ perplexity
// The reference's perplexity loop (examples/falcon_perplexity/falcon_perplexity.cpp:28-124), same chunking and the same
// host arithmetic: the token stream is cut into chunks of n_ctx tokens, each chunk is evaluated from an empty context in
// batches of n_batch (logits of every position), and the negative log-likelihood of token j+1 is added for
// j in [min(512, n_ctx/2), n_ctx-1) with soft_max as in :12-27 (float expf of logit - max, double sum, float division).
// Returns the number of scored tokens; *nll_out = their summed NLL (perplexity = exp(nll / count)).
extern "C" int falcon_hip_perplexity(falcon_hip_context * c, const int32_t * tokens, int64_t n_tokens, int n_ctx, int n_batch, double * nll_out) {
    if (n_ctx < 2 || n_ctx > c->n_ctx || n_batch < 1 || n_batch > c->n_batch) {
        fprintf(stderr, "falcon-hip: perplexity: n_ctx %d / n_batch %d exceed the context's (%d / %d)\n", n_ctx, n_batch, c->n_ctx, c->n_batch); exit(1);
    }
    const int64_t n_chunk = n_tokens / n_ctx;
    const int V = c->m->hp.n_vocab;
    for (int64_t i = 0; i < n_chunk * n_ctx; ++i) if (tokens[i] < 0 || tokens[i] >= V) {
        fprintf(stderr, "falcon-hip: perplexity: token id %d at position %lld is outside [0, %d)\n", tokens[i], (long long) i, V);
        return -1;
    }
    double nll = 0.0; int count = 0;
    std::vector<float> logits((size_t) n_ctx * V), probs((size_t) V);
    for (int64_t i = 0; i < n_chunk; ++i) {
        const int64_t start = i * n_ctx, end = start + n_ctx;
        const int num_batches = (n_ctx + n_batch - 1) / n_batch;
        for (int j = 0; j < num_batches; ++j) {
            const int64_t batch_start = start + (int64_t) j * n_batch;
            const int batch_size = (int)(end - batch_start < n_batch ? end - batch_start : n_batch);
            if (falcon_hip_eval(c, tokens + batch_start, batch_size, j * n_batch, 1)) return -1;
            memcpy(logits.data() + (size_t) j * n_batch * V, falcon_hip_get_logits(c), (size_t) batch_size * V * sizeof(float));
        }
        for (int j = (512 < n_ctx / 2 ? 512 : n_ctx / 2); j < n_ctx - 1; ++j) {
            const float * l = logits.data() + (size_t) j * V;
            float max_logit = l[0];
            for (int v = 0; v < V; ++v) max_logit = l[v] > max_logit ? l[v] : max_logit;
            double sum_exp = 0.0;
            for (int v = 0; v < V; ++v) { const float e = expf(l[v] - max_logit); sum_exp += e; probs[v] = e; }
            const float prob = (float)(probs[tokens[start + j + 1]] / sum_exp);
            nll += -std::log(prob);
            ++count;
        }
    }
    if (nll_out) *nll_out = nll;
    return count;
}

// ------------------------------------------------------------------------------------------------ greedy decode
// argmax with first-maximum tie-break (std::max_element in llama_sample_token_greedy, libfalcon.cpp:3440-3450);
// also advances the device-side loop state: next token id, n_past + 1, output slot. `vals`/`idxs` are either the
// logits themselves (idxs == nullptr: candidate i has index i) or the per-workgroup candidates written by the
// lm_head kernel (fused decode path) -- the single-workgroup scan then covers 2032 instead of 65024 entries.
__global__ void __launch_bounds__(1024) k_argmax_advance(const float * __restrict__ vals, const int * __restrict__ idxs, int n,
                                                         int32_t * __restrict__ token, int * __restrict__ n_past, int32_t * __restrict__ out, int n_past0) {
    __shared__ float bv[16];
    __shared__ int   bi[16];
    float best = -INFINITY; int idx = 0x7FFFFFFF;
    for (int i0 = threadIdx.x; i0 < n; i0 += 4 * blockDim.x) {            // 4 independent loads per step
        float v[4]; int id[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int i = i0 + k * blockDim.x; const int ic = i < n ? i : n - 1; v[k] = vals[ic]; id[k] = idxs ? idxs[ic] : ic; }
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int i = i0 + k * blockDim.x; if (i < n && (v[k] > best || (v[k] == best && id[k] < idx))) { best = v[k]; idx = id[k]; } }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o); const int oi = __shfl_xor(idx, o);
        if (ov > best || (ov == best && oi < idx)) { best = ov; idx = oi; }
    }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) { bv[wid] = best; bi[wid] = idx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) if (bv[w] > best || (bv[w] == best && bi[w] < idx)) { best = bv[w]; idx = bi[w]; }
        const int np = *n_past;
        out[n_past0 < 0 ? 0 : np - n_past0] = idx;              // (n_past0 < 0: a pipeline stage's captured step always uses slot 0)
        token[0] = idx;
        *n_past = np + 1;
    }
}

// ------------------------------------------------------------------------------------------------ pipeline step
__global__ void k_set_i32(int * p, int v) { *p = v; }
__global__ void k_set2_i32(int * p, int v, int * q, int w) { *p = v; *q = w; }
__global__ void k_copy_i32(int32_t * dst, const int32_t * src, int n = 1) { for (int i = threadIdx.x; i < n; i += blockDim.x) dst[i] = src[i]; }
// greedy sample of every row of a lock-step step: token[row] = argmax(logits[row]), lowest index on ties (as k_argmax_advance)
__global__ void __launch_bounds__(1024) k_argmax_rows(const float * __restrict__ logits, int n, int32_t * __restrict__ token) {
    __shared__ float bv[16];
    __shared__ int   bi[16];
    const float * vals = logits + (int64_t) blockIdx.x * n;
    float best = -INFINITY; int idx = 0x7FFFFFFF;
    if ((n & 3) == 0) {
        // four rows of 16 bytes in flight per thread (the scalar loop below was a chain of 64 dependent round trips: 25 us per step for 65024 logits)
        const float4 * v4 = (const float4 *) vals;
        const int n4 = n >> 2;
#pragma unroll 4
        for (int i = threadIdx.x; i < n4; i += blockDim.x) {
            const float4 v = v4[i];
            const float e[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
            for (int k = 0; k < 4; ++k) if (e[k] > best || (e[k] == best && 4 * i + k < idx)) { best = e[k]; idx = 4 * i + k; }
        }
    } else {
        for (int i = threadIdx.x; i < n; i += blockDim.x) { const float v = vals[i]; if (v > best || (v == best && i < idx)) { best = v; idx = i; } }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o); const int oi = __shfl_xor(idx, o);
        if (ov > best || (ov == best && oi < idx)) { best = ov; idx = oi; }
    }
    if ((threadIdx.x & 63) == 0) { bv[threadIdx.x >> 6] = best; bi[threadIdx.x >> 6] = idx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) if (bv[w] > best || (bv[w] == best && bi[w] < idx)) { best = bv[w]; idx = bi[w]; }
        token[blockIdx.x] = idx;
    }
}
__global__ void k_inc_i32(int * p) { *p = *p + 1; }

// One decode step of one pipeline stage, fully stream-ordered (no host synchronisation, no host memory): the token id
// (first stage) / residual row (other stages) are read from device memory, the residual row (inner stages) / the
// greedy-sampled next token (last stage) are written to device memory. The caller moves them between ranks (RCCL).
extern "C" int falcon_hip_stage_step(falcon_hip_context * c, const int32_t * token_dev, const float * hidden_in_dev, int n_past,
                                     float * hidden_out_dev, int32_t * next_token_dev) {
    hip_context & hc = fq_ctx();
    falcon_hip_model * m = c->m;
    hipStream_t st = hc.stream;
    if (n_past < 0 || n_past + 1 > c->n_ctx) { fprintf(stderr, "falcon-hip: stage step at n_past %d exceeds n_ctx %d\n", n_past, c->n_ctx); exit(1); }
    const bool was_keep = c->keep_hidden;
    c->keep_hidden = false;
    // body of one step; n_past_base >= 0: position baked into the launch arguments (plain launches), < 0: the position is
    // whatever n_past_dev holds and the step leaves n_past_dev + 1 behind (captured form, replayable)
    const int B = c->n_seq > 0 ? c->n_seq : 1;                       // lock-step sequences: token_dev / next_token_dev hold B ids, the hidden rows are [B][n_embd]
    auto body = [&](int n_past_base, int max_kv) {
        if (m->first_stage()) hipLaunchKernelGGL(k_copy_i32, dim3(1), dim3(64), 0, st, c->tokens_dev, token_dev, B);
        else HIP_CHECK(hipMemcpyAsync(c->x, hidden_in_dev, (size_t) B * m->hp.n_embd * 4, hipMemcpyDeviceToDevice, st));
        launch_stage(c, B, max_kv, st);
        bool advanced = false;
        if (m->last_stage()) {
            if (next_token_dev) {
                // greedy sample; the loop-state outputs of k_argmax_advance go to scratch slots of this context
                if (B > 1)
                    hipLaunchKernelGGL(k_argmax_rows, dim3((unsigned) B), dim3(1024), 0, st, c->logits_dev, m->hp.n_vocab, next_token_dev);
                else if (stage_fused(c))
                    hipLaunchKernelGGL(k_argmax_advance, dim3(1), dim3(256), 0, st, c->argmax_val, c->argmax_idx, (m->hp.n_vocab + 31) / 32, next_token_dev, c->n_past_dev, c->out_tokens_dev, n_past_base);
                else
                    hipLaunchKernelGGL(k_argmax_advance, dim3(1), dim3(1024), 0, st, c->logits_dev, (const int *) nullptr, m->hp.n_vocab, next_token_dev, c->n_past_dev, c->out_tokens_dev, n_past_base);
                advanced = B == 1;
            }
        } else if (hidden_out_dev) {
            HIP_CHECK(hipMemcpyAsync(hidden_out_dev, c->x, (size_t) B * m->hp.n_embd * 4, hipMemcpyDeviceToDevice, st));
        }
        if (n_past_base < 0 && !advanced) hipLaunchKernelGGL(k_inc_i32, dim3(1), dim3(1), 0, st, c->n_past_dev);
    };
    if (c->stage_graph && !fq_prof_active() && !fq_ctx().dbg_stamps && fused_graph_fits(c)) {
        // one hipGraph replay per step instead of ~70 launches from the host: a stage of a deep pipeline holds few blocks,
        // and the host side of a step would otherwise cost as much as its device side
        const bool same = c->step_graph && c->step_sig == graph_signature(c) && c->sg_in[0] == token_dev && c->sg_in[1] == hidden_in_dev && c->sg_out[0] == hidden_out_dev && c->sg_out[1] == next_token_dev;
        if (!same) {
            if (c->step_graph) { HIP_CHECK(hipGraphExecDestroy(c->step_graph)); c->step_graph = nullptr; }
            hipGraph_t g;
            HIP_CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            body(-1, c->n_ctx);
            HIP_CHECK(hipStreamEndCapture(st, &g));
            HIP_CHECK(hipGraphInstantiate(&c->step_graph, g, nullptr, nullptr, 0));
            HIP_CHECK(hipGraphDestroy(g));
            c->sg_in[0] = token_dev; c->sg_in[1] = hidden_in_dev; c->sg_out[0] = hidden_out_dev; c->sg_out[1] = next_token_dev;
            c->step_next_n_past = -1; c->step_sig = graph_signature(c);
        }
        if (n_past != c->step_next_n_past) hipLaunchKernelGGL(k_set_i32, dim3(1), dim3(1), 0, st, c->n_past_dev, n_past);
        HIP_CHECK(hipGraphLaunch(c->step_graph, st));
        c->step_next_n_past = n_past + 1;
    } else {
        hipLaunchKernelGGL(k_set_i32, dim3(1), dim3(1), 0, st, c->n_past_dev, n_past);
        body(n_past, n_past + 1);
        c->step_next_n_past = -1;
    }
    c->keep_hidden = was_keep;
    return 0;
}

extern "C" int falcon_hip_decode_greedy(falcon_hip_context * c, int32_t first_token, int n_past, int n_steps, int32_t * out_tokens) {
    hip_context & hc = fq_ctx();
    falcon_hip_model * m = c->m;
    if (!m->first_stage() || !m->last_stage()) { fprintf(stderr, "falcon-hip: greedy decode needs the whole model in one process\n"); exit(1); }
    if (n_past + n_steps > c->n_ctx) { fprintf(stderr, "falcon-hip: decode past n_ctx\n"); exit(1); }
    hipStream_t st = hc.stream;
    HIP_CHECK(hipMemcpyAsync(c->n_past_dev, &n_past, 4, hipMemcpyHostToDevice, st));
    HIP_CHECK(hipMemcpyAsync(c->tokens_dev, &first_token, 4, hipMemcpyHostToDevice, st));
    HIP_CHECK(hipStreamSynchronize(st));
    const bool was_keep = c->keep_hidden;
    c->keep_hidden = false;
    auto one_step = [&](hipStream_t s, int max_kv) {
        launch_stage(c, 1, max_kv, s);
        if (stage_fused(c))
            hipLaunchKernelGGL(k_argmax_advance, dim3(1), dim3(256), 0, s, c->argmax_val, c->argmax_idx, (m->hp.n_vocab + 31) / 32, c->tokens_dev, c->n_past_dev, c->out_tokens_dev, n_past);
        else
            hipLaunchKernelGGL(k_argmax_advance, dim3(1), dim3(1024), 0, s, c->logits_dev, (const int *) nullptr, m->hp.n_vocab, c->tokens_dev, c->n_past_dev, c->out_tokens_dev, n_past);
    };
    if (c->use_graph && fused_graph_fits(c)) {
        // the graph bakes n_past0 into k_argmax_advance's arguments: re-capture when the base position changes
        if (!c->decode_graph || c->graph_base != n_past || c->decode_sig != graph_signature(c)) {
            if (c->decode_graph) { HIP_CHECK(hipGraphExecDestroy(c->decode_graph)); c->decode_graph = nullptr; }
            hipGraph_t g;
            HIP_CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            one_step(st, c->n_ctx);
            HIP_CHECK(hipStreamEndCapture(st, &g));
            HIP_CHECK(hipGraphInstantiate(&c->decode_graph, g, nullptr, nullptr, 0));
            HIP_CHECK(hipGraphDestroy(g));
            c->graph_base = n_past; c->decode_sig = graph_signature(c);
        }
        for (int s = 0; s < n_steps; ++s) HIP_CHECK(hipGraphLaunch(c->decode_graph, st));
    } else {
        for (int s = 0; s < n_steps; ++s) one_step(st, n_past + n_steps);
    }
    HIP_CHECK(hipMemcpyAsync(out_tokens, c->out_tokens_dev, (size_t) n_steps * 4, hipMemcpyDeviceToHost, st));
    fetch_sync_error(c, st);
    HIP_CHECK(hipStreamSynchronize(st));
    c->keep_hidden = was_keep;
    return report_sync_error(c, "greedy decode");
}
