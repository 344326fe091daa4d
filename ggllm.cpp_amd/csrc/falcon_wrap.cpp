// falcon_wrap.cpp -- the device-resident fast path BEHIND the reference's own entry points, with the reference's sources
// unchanged. Host C++ only; compiled by the integrator TOGETHER with the reference's libfalcon.cpp / ggml.c / the CLIs
// (it includes the reference's libfalcon.h), linked with
//
//   -Wl,--wrap=falcon_init_from_file,--wrap=falcon_context_prepare,--wrap=falcon_eval,--wrap=falcon_get_logits,--wrap=falcon_print_timings,--wrap=llama_free
//   -Wl,--wrap=llama_load_session_file,--wrap=llama_save_session_file,--wrap=falcon_copy_state_data,--wrap=falcon_set_state_data,--wrap=falcon_get_embeddings
//   -Wl,--wrap=llama_apply_lora_from_file
//   -L<repo>/ggllm.cpp_amd -lggml_hip
//
// The GNU linker then sends every call that falcon_main.cpp / falcon_perplexity.cpp / falcon_common.cpp make to these six
// functions of libfalcon.h (:165-168, :220-223, :256, :263, :172, :321) to the __wrap_ versions below, which keep the
// reference's own objects alive (tokenizer, samplers, sessions, timings structure: everything the CLIs touch besides the
// evaluation still is the reference's code) and run the evaluation itself on the device:
//
//   falcon_init_from_file   n_gpu_layers > 0 (the CLIs' default is 200, examples/falcon_common.h:32): the reference loads the file as
//                           usual but with n_gpu_layers = 0 (its per-op CUDA offload is not used: nothing is uploaded twice), then
//                           falcon_hip_model_load_ggcc puts the same file's weights into HBM and a falcon_hip_context gets the KV
//                           cache (libfalcon.cpp:1552-1959, 3755). ANY positive value keeps the whole model resident (288 GB of
//                           HBM: the reference's partial offload, libfalcon.cpp:1813-1883, exists for cards the model does not fit).
//                           n_gpu_layers == 0 (`-ngl 0`, BASELINE config 1's command) means what it means in the reference: no
//                           device side at all, every call below falls through to the reference's own CPU path.
//   falcon_context_prepare  a further context over the same model (falcon_main's system-prompt context, falcon_main.cpp:169)
//   falcon_eval             falcon_hip_eval: the whole falcon_eval_internal graph (libfalcon.cpp:2011-2588) on the device
//   falcon_get_logits       the logits falcon_hip_eval brought back (last row, or all rows with logits_all)
//   falcon_print_timings    the reference's report (libfalcon.cpp:4700-4714) over this path's own clocks
//   llama_free              releases the device side, then the reference's context
//
//   llama_apply_lora_from_file   would patch the reference's HOST tensors (libfalcon.h:187-191) while the device copy stays as loaded:
//                           fails loudly (returns 1) for a context with a device side
//   llama_load_session_file, llama_save_session_file, falcon_copy_state_data, falcon_set_state_data, falcon_get_embeddings
//                           address the reference context's host KV cache / embedding buffer, which this path does not fill: for a
//                           context with a device side they FAIL LOUDLY (false / 0 / NULL after a message) instead of silently
//                           restoring or saving an empty cache (libfalcon.h:205-214, 267)
#include "libfalcon.h"
#include "../../include/falcon-hip.h"
#include "../../include/ggml-hip-ops.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <vector>
#include <string>

extern "C" {
struct falcon_context * __real_falcon_init_from_file(const char * path_model, struct falcon_context_params params);
struct falcon_context * __real_falcon_context_prepare(falcon_context_params params, falcon_model * model, std::string context_name, bool verbose);
int     __real_falcon_eval(struct falcon_context * ctx, const falcon_token * tokens, falcon_evaluation_config & configuration);
float * __real_falcon_get_logits(struct falcon_context * ctx);
bool    __real_llama_load_session_file(struct falcon_context * ctx, const char * path_session, falcon_token * tokens_out, size_t n_token_capacity, size_t * n_token_count_out);
bool    __real_llama_save_session_file(struct falcon_context * ctx, const char * path_session, const falcon_token * tokens, size_t n_token_count);
size_t  __real_falcon_copy_state_data(struct falcon_context * ctx, uint8_t * dst);
size_t  __real_falcon_set_state_data(struct falcon_context * ctx, uint8_t * src);
float * __real_falcon_get_embeddings(struct falcon_context * ctx);
int     __real_llama_apply_lora_from_file(struct falcon_context * ctx, const char * path_lora, const char * path_base_model, int n_threads);
void    __real_falcon_print_timings(struct falcon_context * ctx);
void    __real_llama_free(struct falcon_context * ctx);
}

namespace {

struct hip_model_ref { falcon_hip_model * m; int users; };
struct hip_side {
    falcon_hip_context * c = nullptr;
    falcon_model * ref_model = nullptr;                       // key of the shared device model
    bool logits_all = false;
    bool pending_one = false;                                 // the last eval was a single token whose logits are still on the device
    int  n_ctx = 0, n_batch = 0;
    int64_t t_eval_us = 0, t_p_eval_us = 0, t_start_us = 0; int n_eval = 0, n_p_eval = 0;
};

std::mutex g_mu;
std::map<falcon_context *, hip_side> g_ctx;                   // reference context -> device side
std::map<falcon_model *, hip_model_ref> g_model;              // reference model -> device model (shared by its contexts)

int64_t now_us() { return std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

bool disabled() { const char * e = getenv("FALCON_HIP_WRAP"); return e && atoi(e) == 0; }      // FALCON_HIP_WRAP=0: the reference's own path

void attach(falcon_context * ctx, falcon_hip_model * hm, const falcon_context_params & p) {
    hip_side s;
    s.ref_model = falcon_get_falcon_model(ctx);
    s.logits_all = p.logits_all; s.n_ctx = p.n_ctx; s.n_batch = p.n_batch > 0 ? p.n_batch : 1;
    s.c = falcon_hip_context_create(hm, s.n_ctx, s.n_batch, s.n_ctx);
    // First-use set-up -- code objects of every kernel of the path, the decode step's hipGraph, ring schedules, pinned logits row -- belongs to context
    // preparation, not to the caller's first falcon_eval (round 5: the reference's CLI booked it as batch-eval time: 128 tokens in 19 ms instead of 9).
    // One batch of n_batch placeholder tokens and one single-token step run here; they leave nothing behind that a later eval reads (every position's KV
    // rows are written by the eval that first makes them visible). FALCON_HIP_WRAP_WARM=0 skips it.
    const char * warm = getenv("FALCON_HIP_WRAP_WARM");
    if (s.c && !(warm && atoi(warm) == 0)) {
        const int nb = s.n_batch < s.n_ctx ? s.n_batch : s.n_ctx - 1;
        std::vector<int32_t> zeros((size_t)(nb > 0 ? nb : 1), 0);
        if (nb > 1) falcon_hip_eval(s.c, zeros.data(), nb, 0, 0);
        else if (s.n_ctx > 2) falcon_hip_eval(s.c, zeros.data(), 1, 0, 0);      // (n_batch 1: position 0's KV rows exist before the step at position 1 attends over them)
        if (s.n_ctx > 2) { falcon_hip_eval_token(s.c, 0, 1); (void) falcon_hip_get_logits(s.c); falcon_hip_eval_token(s.c, 0, 2); (void) falcon_hip_get_logits(s.c); }
    }
    s.t_start_us = now_us();
    g_model[s.ref_model].m = hm; ++g_model[s.ref_model].users;
    g_ctx[ctx] = s;
}

}   // namespace

extern "C" {

struct falcon_context * __wrap_falcon_init_from_file(const char * path_model, struct falcon_context_params params) {
    if (disabled() || params.vocab_only) return __real_falcon_init_from_file(path_model, params);
    if (params.n_gpu_layers <= 0) {                            // -ngl 0: the reference's own CPU path, as in the reference (libfalcon.cpp:1813-1826)
        fprintf(stderr, "falcon-hip: n_gpu_layers = 0: %s stays on the host (the reference's ggml.c path); no device side\n", path_model);
        return __real_falcon_init_from_file(path_model, params);
    }
    falcon_context_params host = params;
    host.n_gpu_layers = 0;                                    // the reference keeps its mmap; the weights go to HBM once, below
    falcon_context * ctx = __real_falcon_init_from_file(path_model, host);
    if (!ctx) return nullptr;
    ggml_hip_init(params.main_gpu);
    if (const char * e = getenv("GGML_HIP_REFERENCE_ORDER")) ggml_hip_reference_order(atoi(e));
    falcon_hip_hparams hp;
    falcon_hip_model * hm = falcon_hip_model_load_ggcc(path_model, 0, 0, &hp);
    if (!hm) { fprintf(stderr, "falcon-hip: %s could not be loaded onto the device -- the reference's own path stays in place\n", path_model); return ctx; }
    std::lock_guard<std::mutex> lk(g_mu);
    attach(ctx, hm, params);
    fprintf(stderr, "falcon-hip: %s resident on the device (%.2f GB of weights per token); falcon_eval runs there\n", path_model,
            falcon_hip_model_weight_bytes(hm) / 1e9);
    return ctx;
}

struct falcon_context * __wrap_falcon_context_prepare(falcon_context_params params, falcon_model * model, std::string context_name, bool verbose) {
    falcon_context * ctx = __real_falcon_context_prepare(params, model, context_name, verbose);
    if (!ctx) return ctx;
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_model.find(model);
    if (it != g_model.end() && it->second.m) attach(ctx, it->second.m, params);
    return ctx;
}

int __wrap_falcon_eval(struct falcon_context * ctx, const falcon_token * tokens, falcon_evaluation_config & configuration) {
    hip_side * s = nullptr;
    { std::lock_guard<std::mutex> lk(g_mu); auto it = g_ctx.find(ctx); if (it != g_ctx.end()) s = &it->second; }
    if (!s) return __real_falcon_eval(ctx, tokens, configuration);
    const int N = configuration.n_tokens, n_past = configuration.n_past;
    if (N < 1 || N > s->n_batch || n_past + N > s->n_ctx) {
        fprintf(stderr, "falcon-hip: falcon_eval of %d tokens at n_past %d exceeds n_batch %d / n_ctx %d\n", N, n_past, s->n_batch, s->n_ctx);
        return 1;
    }
    // the n_ctx the reference hands to ggml_rope: n_max_real_ctx when the caller has set it (libfalcon.cpp:2229-2230; falcon_main
    // does for non-interactive runs, falcon_main.cpp:836), else the context's -- it selects the dynamic-NTK factor
    falcon_hip_context_set_rope_n_ctx(s->c, configuration.n_max_real_ctx ? configuration.n_max_real_ctx : s->n_ctx);
    const int64_t t0 = now_us();
    int rc;
    // --debug-timings (falcon_evaluation_config::debug_timings; the reference's rule, libfalcon.cpp:2506-2516: 1 the first eval, 2 the first eval past the prompt's
    // first batch -- falcon_main turns it into 3 for its last token --, 3 every eval): the reference prints its ggml graph's nodes; here the eval's launches
    bool timings = false;
    if (configuration.debug_timings && (n_past > 0 || configuration.debug_timings != 2)) {
        static bool first = true;
        if ((first && configuration.debug_timings <= 2) || configuration.debug_timings > 2) { first = false; timings = true; }
    }
    if (timings) {
        rc = falcon_hip_eval_debug_timings(s->c, (const int32_t *) tokens, N, n_past, s->logits_all ? 1 : 0);
        s->pending_one = false;
    } else if (N == 1 && !s->logits_all) {
        // one token: the captured graph, no copy, no wait -- falcon_get_logits fetches the row if (and when) the caller samples from it;
        // the time until then is booked there
        rc = falcon_hip_eval_token(s->c, (int32_t) tokens[0], n_past);
        s->pending_one = rc == 0;
    } else {
        rc = falcon_hip_eval(s->c, (const int32_t *) tokens, N, n_past, s->logits_all ? 1 : 0);
        s->pending_one = false;
    }
    const int64_t dt = now_us() - t0;
    if (N == 1) { s->t_eval_us += dt; ++s->n_eval; } else { s->t_p_eval_us += dt; s->n_p_eval += N; }     // libfalcon.cpp:2578-2585
    return rc;
}

float * __wrap_falcon_get_logits(struct falcon_context * ctx) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_ctx.find(ctx);
    if (it == g_ctx.end()) return __real_falcon_get_logits(ctx);
    const int64_t t0 = now_us();
    float * lg = const_cast<float *>(falcon_hip_get_logits(it->second.c));
    if (falcon_hip_context_last_error(it->second.c)) {
        // falcon_get_logits cannot report failure and the caller is about to sample from this row: errors of the device path are fatal,
        // as in the reference's backend (CUDA_CHECK -> exit, ggml-cuda.cu:22-51)
        fprintf(stderr, "falcon-hip: falcon_get_logits: the step that produced these logits lost an in-launch hand-off -- aborting instead of sampling from invalid logits\n");
        exit(3);
    }
    if (it->second.pending_one) { it->second.t_eval_us += now_us() - t0; it->second.pending_one = false; }     // (the step's device time ends here)
    return lg;
}

void __wrap_falcon_print_timings(struct falcon_context * ctx) {
    hip_side s;
    { std::lock_guard<std::mutex> lk(g_mu); auto it = g_ctx.find(ctx); if (it == g_ctx.end()) { __real_falcon_print_timings(ctx); return; } s = it->second; }
    const int n_eval = s.n_eval > 0 ? s.n_eval : 1, n_p_eval = s.n_p_eval > 0 ? s.n_p_eval : 1;
    fprintf(stderr, "\n");
    fprintf(stderr, "falcon_print_timings: (device-resident path, libggml_hip.so)\n");
    fprintf(stderr, "falcon_print_timings: batch eval time = %8.2f ms / %5d tokens (%8.2f ms per token, %8.2f tokens per second)\n",
            1e-3 * s.t_p_eval_us, s.n_p_eval, 1e-3 * s.t_p_eval_us / n_p_eval, 1e6 / (s.t_p_eval_us > 0 ? (double) s.t_p_eval_us / n_p_eval : 1e18));
    fprintf(stderr, "falcon_print_timings:       eval time = %8.2f ms / %5d runs   (%8.2f ms per token, %8.2f tokens per second)\n",
            1e-3 * s.t_eval_us, s.n_eval, 1e-3 * s.t_eval_us / n_eval, 1e6 / (s.t_eval_us > 0 ? (double) s.t_eval_us / n_eval : 1e18));
    fprintf(stderr, "falcon_print_timings:      total time = %8.2f ms\n", 1e-3 * (now_us() - s.t_start_us));
}

// ---- state that lives in the reference context's host buffers, which the device path neither fills nor reads
static bool has_device_side(falcon_context * ctx, const char * what) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_ctx.find(ctx) == g_ctx.end()) return false;
    fprintf(stderr, "falcon-hip: %s is not supported for a context evaluated on the device (its weights, KV cache and embeddings are not in the reference's host buffers); "
                    "run with FALCON_HIP_WRAP=0 to use it\n", what);
    return true;
}
int __wrap_llama_apply_lora_from_file(struct falcon_context * ctx, const char * path_lora, const char * path_base_model, int n_threads) {
    if (has_device_side(ctx, "llama_apply_lora_from_file")) return 1;      // (non-zero = failure, libfalcon.h:186)
    return __real_llama_apply_lora_from_file(ctx, path_lora, path_base_model, n_threads);
}
bool __wrap_llama_load_session_file(struct falcon_context * ctx, const char * path_session, falcon_token * tokens_out, size_t n_token_capacity, size_t * n_token_count_out) {
    if (has_device_side(ctx, "llama_load_session_file")) { if (n_token_count_out) *n_token_count_out = 0; return false; }
    return __real_llama_load_session_file(ctx, path_session, tokens_out, n_token_capacity, n_token_count_out);
}
bool __wrap_llama_save_session_file(struct falcon_context * ctx, const char * path_session, const falcon_token * tokens, size_t n_token_count) {
    if (has_device_side(ctx, "llama_save_session_file")) return false;
    return __real_llama_save_session_file(ctx, path_session, tokens, n_token_count);
}
size_t __wrap_falcon_copy_state_data(struct falcon_context * ctx, uint8_t * dst) {
    if (has_device_side(ctx, "falcon_copy_state_data")) return 0;
    return __real_falcon_copy_state_data(ctx, dst);
}
size_t __wrap_falcon_set_state_data(struct falcon_context * ctx, uint8_t * src) {
    if (has_device_side(ctx, "falcon_set_state_data")) return 0;
    return __real_falcon_set_state_data(ctx, src);
}
float * __wrap_falcon_get_embeddings(struct falcon_context * ctx) {
    if (has_device_side(ctx, "falcon_get_embeddings")) return nullptr;
    return __real_falcon_get_embeddings(ctx);
}

void __wrap_llama_free(struct falcon_context * ctx) {
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_ctx.find(ctx);
        if (it != g_ctx.end()) {
            falcon_hip_context_free(it->second.c);
            auto mi = g_model.find(it->second.ref_model);
            if (mi != g_model.end() && --mi->second.users == 0) { falcon_hip_model_free(mi->second.m); g_model.erase(mi); }
            g_ctx.erase(it);
        }
    }
    __real_llama_free(ctx);
}

}
