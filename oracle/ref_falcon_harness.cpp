// ref_falcon_harness.cpp -- TEST INFRASTRUCTURE. A few C entry points over the REAL reference's model path
// (libfalcon.cpp: falcon_init_from_file -> GGCC loader -> falcon_eval -> falcon_get_logits), compiled by oracle/Makefile
// together with the reference's own sources where they lie under /root/reference into oracle/_ref/libfalcon_ref.so.
// Used by oracle/gen_golden.py in the build container to capture logits of synthetic GGCC files written by
// tests/ggcc_writer.py (which pins the writer, our loader and our graph restatement against the reference itself).
#include "libfalcon.h"
#include "cmpnct_unicode.h"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <cstdio>
#include <vector>

extern "C" {

// falcon_init_backend once: ggml_init fills the fp16 tables this SIMD-less build converts through (the reference's tools
// call it first thing, examples/falcon_quantize/quantize.cpp; without it the k-quant quantizers read d = 0)
static void init_once() {
    static bool once = false;
    if (!once) { falcon_init_backend(); once = true; }
}

void * reff_load(const char * path, int n_ctx, int n_batch) {
    init_once();
    falcon_context_params p = falcon_context_default_params();
    p.n_ctx = n_ctx; p.n_batch = n_batch; p.n_gpu_layers = 0; p.logits_all = true; p.f16_kv = false; p.use_mmap = true; p.seed = 1;
    return (void *) falcon_init_from_file(path, p);
}

// the same with n_gpu_layers (the -ngl of the CLIs): only meaningful in the -DGGML_USE_CUBLAS builds of `make ref_falcon_hip`
void * reff_load_ngl(const char * path, int n_ctx, int n_batch, int n_gpu_layers) {
    init_once();
    falcon_context_params p = falcon_context_default_params();
    p.n_ctx = n_ctx; p.n_batch = n_batch; p.n_gpu_layers = n_gpu_layers; p.logits_all = true; p.f16_kv = false; p.use_mmap = true; p.seed = 1;
    return (void *) falcon_init_from_file(path, p);
}

// llama_apply_lora_from_file (libfalcon.h:187-191); returns its result (0 = applied)
int reff_apply_lora(void * ctx, const char * path_lora, int n_threads) {
    return llama_apply_lora_from_file((falcon_context *) ctx, path_lora, nullptr, n_threads);
}

// falcon_model_quantize (the falcon_quantize tool's work, libfalcon.cpp:3914-3925) with one thread (deterministic)
int reff_quantize(const char * path_in, const char * path_out, int ftype, int quantize_output_tensor, int allow_requantize) {
    init_once();
    llama_model_quantize_params p = llama_model_quantize_default_params();
    p.nthread = 1; p.ftype = (enum llama_ftype) ftype; p.quantize_output_tensor = quantize_output_tensor != 0; p.allow_requantize = allow_requantize != 0;
    return falcon_model_quantize(path_in, path_out, &p);
}

// logits_out: n * n_vocab floats (logits_all). Returns 0 on success.
int reff_eval(void * ctx, const int * tokens, int n, int n_past, int n_threads, float * logits_out) {
    falcon_evaluation_config cfg;
    cfg.n_tokens = n; cfg.n_past = n_past; cfg.n_threads = n_threads; cfg.n_max_real_ctx = falcon_n_ctx((falcon_context *) ctx);
    const int rc = falcon_eval((falcon_context *) ctx, (const falcon_token *) tokens, cfg);
    if (rc) return rc;
    std::memcpy(logits_out, falcon_get_logits((falcon_context *) ctx), sizeof(float) * (size_t) n * falcon_n_vocab((falcon_context *) ctx));
    return 0;
}

// NLL of `target` under the logits of one position, as the reference's perplexity example computes it
// (examples/falcon_perplexity/falcon_perplexity.cpp:12-27 soft_max + :115-117), with this build's libm
double reff_token_nll(const float * logits, int n_vocab, int target) {
    float max_logit = logits[0];
    for (int i = 0; i < n_vocab; ++i) max_logit = std::max(max_logit, logits[i]);
    double sum_exp = 0.0;
    std::vector<float> probs((size_t) n_vocab);
    for (int i = 0; i < n_vocab; ++i) { const float e = expf(logits[i] - max_logit); sum_exp += e; probs[i] = e; }
    for (int i = 0; i < n_vocab; ++i) probs[i] /= sum_exp;
    const float prob = probs[target];
    return (double) -std::log(prob);
}

// falcon_tokenize (libfalcon.cpp:4623-4641) on the loaded model's vocabulary
int reff_tokenize(void * ctx, const char * text, int * tokens, int n_max, int add_bos) {
    return falcon_tokenize((falcon_context *) ctx, text, (falcon_token *) tokens, n_max, add_bos != 0);
}

// the pre-tokenizer's class of a code point (cmpnct_unicode.cpp:98-115): 0 digit, 1 letter, 2 whitespace, 3.. others
int reff_code_type(int c) { return (int) CNCTUnicode::get_code_type(c); }

int reff_n_vocab(void * ctx) { return falcon_n_vocab((falcon_context *) ctx); }
void reff_free(void * ctx) { llama_free((falcon_context *) ctx); }

}
