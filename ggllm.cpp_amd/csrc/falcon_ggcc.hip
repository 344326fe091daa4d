// falcon_ggcc.hip -- the reference's model file format, GGCC v10 (libfalcon.cpp:770-973: falcon_file_loader::read_magic /
// read_hparams / read_vocab / read_tensor_metadata), read straight into a device-resident falcon_hip_model.
//
//   u32 magic 0x67676363 'ggcc', u32 version 10
//   u32 n_vocab, n_embd, n_head, n_head_kv, n_layer, n_falcon_type (7 | 40), ftype, n_bpe_merges
//   n_vocab x { u32 len, bytes, f32 score }                 (skipped: the tokenizer is not on this path)
//   u32 n_merges, n_merges x { u32 len, bytes, u32 len, bytes }   (skipped)
//   until EOF: u32 n_dims (1 | 2), u32 name_len, u32 ggml_type, u32 ne[n_dims], name, pad to 32 bytes, data
// n_ff is not stored: 4 * n_embd (libfalcon.cpp:1598). Two LayerNorms per block <=> Falcon-40B layout: n_layer 60 / 80,
// else n_falcon_type == 40 (libfalcon.cpp:1578-1592). The file is mmap'ed; tensors outside the requested block range
// are never touched (a pipeline stage reads only its own blocks). Host-only code except for the uploads it triggers.
#include <hip/hip_runtime.h>
#include "../../include/falcon-hip.h"
#include "fq_types.h"
#include "hip_context.h"
#include "../../include/ggml-hip-ops.h"
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {
struct ggcc_tensor { std::string name; int type; int n_dims; int64_t ne[2]; size_t offset, size; };   // offset of the data
struct ggcc_file {
    int fd = -1; const uint8_t * base = nullptr; size_t size = 0;
    falcon_hip_hparams hp{}; int ftype = 0, falcon_type = 0, n_merges = 0;
    size_t vocab_begin = 0, vocab_end = 0;                       // vocabulary + merges, byte range in the file
    std::vector<ggcc_tensor> tensors;
    std::string error;
    ~ggcc_file() { if (base) munmap((void *) base, size); if (fd >= 0) close(fd); }
};

size_t tensor_bytes(int type, int64_t ne0, int64_t ne1) {
    if (ne0 <= 0 || ne1 <= 0 || ne0 > ((int64_t) 1 << 31) || ne1 > ((int64_t) 1 << 31) || ne0 > ((int64_t) 1 << 60) / ne1) return 0;   // (sizes of a hostile file must not wrap)
    if (type == FQ_F32) return (size_t) ne0 * ne1 * 4;
    if (type == 1 /* GGML_TYPE_F16 */) return (size_t) ne0 * ne1 * 2;
    const fq_type_desc d = fq_desc(type);
    if (!d.blck || ne0 % d.blck) return 0;
    return (size_t)(ne0 / d.blck) * d.tsize * (size_t) ne1;
}

bool ggcc_open(const char * path, ggcc_file & f) {
    f.fd = open(path, O_RDONLY);
    if (f.fd < 0) { f.error = std::string("cannot open ") + path; return false; }
    struct stat st;
    if (fstat(f.fd, &st) != 0 || st.st_size < 40) { f.error = "file too small"; return false; }
    f.size = (size_t) st.st_size;
    void * p = mmap(nullptr, f.size, PROT_READ, MAP_PRIVATE, f.fd, 0);
    if (p == MAP_FAILED) { f.error = "mmap failed"; return false; }
    f.base = (const uint8_t *) p;
    size_t pos = 0;
    auto u32 = [&](uint32_t & v) { if (pos + 4 > f.size) return false; memcpy(&v, f.base + pos, 4); pos += 4; return true; };
    auto skip = [&](size_t n) { if (pos + n > f.size) return false; pos += n; return true; };
    uint32_t magic = 0, version = 0;
    if (!u32(magic) || !u32(version)) { f.error = "truncated header"; return false; }
    if (magic != 0x67676363u || version != 10u) {
        char b[128]; snprintf(b, sizeof b, "not a GGCC v10 file (magic %08x, version %u): re-convert with falcon_quantize", magic, version);
        f.error = b; return false;
    }
    uint32_t h[8];
    for (uint32_t & v : h) if (!u32(v)) { f.error = "truncated hparams"; return false; }
    f.hp.n_vocab = (int32_t) h[0]; f.hp.n_embd = (int32_t) h[1]; f.hp.n_head = (int32_t) h[2]; f.hp.n_head_kv = (int32_t) h[3];
    f.hp.n_layer = (int32_t) h[4]; f.falcon_type = (int) h[5]; f.ftype = (int) h[6];
    f.hp.n_ff = 4 * f.hp.n_embd;
    f.hp.two_norms = (f.hp.n_layer == 60 || f.hp.n_layer == 80) ? 1 : (f.hp.n_layer == 32 ? 0 : (f.falcon_type == 40 ? 1 : 0));
    f.hp.layer_begin = 0; f.hp.layer_end = f.hp.n_layer;
    if (f.hp.n_embd <= 0 || f.hp.n_head <= 0 || f.hp.n_embd != 64 * f.hp.n_head || f.hp.n_head_kv <= 0 || f.hp.n_head % f.hp.n_head_kv) {
        f.error = "implausible hparams (head_dim must be 64)"; return false;
    }
    if (f.hp.n_vocab <= 0 || f.hp.n_vocab > (1 << 24) || f.hp.n_layer <= 0 || f.hp.n_layer > 4096 || f.hp.n_embd > (1 << 20)) {
        f.error = "implausible hparams (n_vocab " + std::to_string(f.hp.n_vocab) + ", n_layer " + std::to_string(f.hp.n_layer) + ", n_embd " + std::to_string(f.hp.n_embd) + ")";
        return false;
    }
    f.vocab_begin = pos;
    for (int i = 0; i < f.hp.n_vocab; ++i) {                     // vocabulary: len, bytes, f32 score
        uint32_t len = 0;
        if (!u32(len) || !skip((size_t) len + 4)) { f.error = "truncated vocabulary"; return false; }
    }
    uint32_t nm = 0;
    if (!u32(nm)) { f.error = "truncated merges"; return false; }
    f.n_merges = (int) nm;
    for (uint32_t i = 0; i < nm; ++i) {
        uint32_t l1 = 0, l2 = 0;
        if (!u32(l1) || !skip(l1) || !u32(l2) || !skip(l2)) { f.error = "truncated merges"; return false; }
    }
    f.vocab_end = pos;
    while (pos < f.size) {
        uint32_t n_dims = 0, name_len = 0, type = 0, ne[2] = {1, 1};
        if (!u32(n_dims) || !u32(name_len) || !u32(type)) { f.error = "truncated tensor record"; return false; }
        if (n_dims < 1 || n_dims > 2) { f.error = "tensor with " + std::to_string(n_dims) + " dimensions"; return false; }
        for (uint32_t d = 0; d < n_dims; ++d) if (!u32(ne[d])) { f.error = "truncated tensor record"; return false; }
        if (pos + name_len > f.size) { f.error = "truncated tensor name"; return false; }
        ggcc_tensor t;
        t.name.assign((const char *) f.base + pos, name_len); pos += name_len;
        pos += (size_t)(-(int64_t) pos & 31);                    // data starts at the next multiple of 32 (libfalcon.cpp:949-952)
        t.type = (int) type; t.n_dims = (int) n_dims; t.ne[0] = ne[0]; t.ne[1] = ne[1];
        t.size = tensor_bytes(t.type, t.ne[0], t.ne[1]);
        if (!t.size) { f.error = "tensor " + t.name + ": unsupported type " + std::to_string(type) + " / row length"; return false; }
        t.offset = pos;
        if (pos + t.size > f.size) { f.error = "tensor " + t.name + " runs past the end of the file"; return false; }
        pos += t.size;
        f.tensors.push_back(std::move(t));
    }
    return true;
}
}   // namespace

// Host-only: header + tensor directory (no device needed). Returns the number of tensors, or -1 (message on stderr).
// If dir_out != NULL, up to dir_cap bytes of a text directory are written: one line per tensor "name type ne0 ne1 offset size".
extern "C" int falcon_hip_ggcc_scan(const char * path, falcon_hip_hparams * hp_out, int * ftype_out, char * dir_out, size_t dir_cap) {
    ggcc_file f;
    if (!ggcc_open(path, f)) { fprintf(stderr, "falcon-hip: %s: %s\n", path, f.error.c_str()); return -1; }
    if (hp_out) *hp_out = f.hp;
    if (ftype_out) *ftype_out = f.ftype;
    if (dir_out && dir_cap) {
        std::string s;
        for (const ggcc_tensor & t : f.tensors) {
            char b[512];
            snprintf(b, sizeof b, "%s %d %lld %lld %zu %zu\n", t.name.c_str(), t.type, (long long) t.ne[0], (long long) t.ne[1], t.offset, t.size);
            s += b;
        }
        snprintf(dir_out, dir_cap, "%s", s.c_str());
    }
    return (int) f.tensors.size();
}

// Loads blocks [layer_begin, layer_end) (layer_end <= 0: all) of a GGCC v10 file into a new device-resident model.
// hp_out (optional) receives the hyper-parameters. NULL on error (message on stderr).
extern "C" falcon_hip_model * falcon_hip_model_load_ggcc(const char * path, int layer_begin, int layer_end, falcon_hip_hparams * hp_out) {
    ggcc_file f;
    if (!ggcc_open(path, f)) { fprintf(stderr, "falcon-hip: %s: %s\n", path, f.error.c_str()); return nullptr; }
    falcon_hip_hparams hp = f.hp;
    hp.layer_begin = layer_begin > 0 ? layer_begin : 0;
    hp.layer_end = (layer_end > 0 && layer_end <= hp.n_layer) ? layer_end : hp.n_layer;
    if (hp.layer_begin >= hp.layer_end) { fprintf(stderr, "falcon-hip: %s: empty block range [%d, %d)\n", path, hp.layer_begin, hp.layer_end); return nullptr; }
    falcon_hip_model * m = falcon_hip_model_create(&hp);
    int used = 0;
    for (const ggcc_tensor & t : f.tensors) {
        if (t.type == 1) {                                       // f16 tensors (unquantized output / embedding matrices) are not on this path
            const bool mine = (t.name == "lm_head.weight" && hp.layer_end == hp.n_layer) || (t.name == "transformer.word_embeddings.weight" && hp.layer_begin == 0);
            if (mine) { fprintf(stderr, "falcon-hip: %s: %s is f16; quantize it (falcon_quantize quantizes the output tensor by default)\n", path, t.name.c_str()); falcon_hip_model_free(m); return nullptr; }
            continue;
        }
        {   // shapes are checked HERE (a bad file is an error return, not the exit(1) of the in-memory upload path)
            const int64_t E = hp.n_embd, QKV = (int64_t)(hp.n_head + 2 * hp.n_head_kv) * 64, FF = hp.n_ff, V = hp.n_vocab;
            const std::string & n = t.name;
            auto ends = [&](const char * suf) { const size_t l = strlen(suf); return n.size() >= l && n.compare(n.size() - l, l, suf) == 0; };
            int64_t e0 = -1, e1 = -1;
            if (n == "transformer.word_embeddings.weight" || n == "lm_head.weight") { e0 = E; e1 = V; }
            else if (ends("self_attention.query_key_value.weight")) { e0 = E; e1 = QKV; }
            else if (ends("self_attention.dense.weight")) { e0 = E; e1 = E; }
            else if (ends("mlp.dense_h_to_4h.weight")) { e0 = E; e1 = FF; }
            else if (ends("mlp.dense_4h_to_h.weight")) { e0 = FF; e1 = E; }
            else if (t.n_dims == 1) { e0 = E; e1 = 1; }                       // the norms
            const int64_t n1 = t.n_dims > 1 ? t.ne[1] : 1;
            if (e0 >= 0 && (t.ne[0] != e0 || n1 != e1 || (e1 > 1 && t.type == 0))) {
                fprintf(stderr, "falcon-hip: %s: tensor %s is [%lld x %lld] of type %d, expected [%lld x %lld]%s\n", path, n.c_str(), (long long) t.ne[0], (long long) n1, t.type,
                        (long long) e0, (long long) e1, (e1 > 1 && t.type == 0) ? " quantized (f32 matrices are not on this path: run falcon_quantize)" : "");
                falcon_hip_model_free(m);
                return nullptr;
            }
        }
        if (falcon_hip_model_set_tensor(m, t.name.c_str(), t.type, f.base + t.offset, t.ne[0], t.n_dims > 1 ? t.ne[1] : 1) == 0) ++used;
    }
    const int per_block = hp.two_norms ? 8 : 6;
    const int want = per_block * (hp.layer_end - hp.layer_begin) + (hp.layer_begin == 0 ? 1 : 0) + (hp.layer_end == hp.n_layer ? 3 : 0);
    if (used != want) {
        fprintf(stderr, "falcon-hip: %s: %d of the %d tensors of blocks [%d, %d) found\n", path, used, want, hp.layer_begin, hp.layer_end);
        falcon_hip_model_free(m);
        return nullptr;
    }
    if (hp_out) *hp_out = hp;
    return m;
}

// ------------------------------------------------------------------------------------------------ stage planner
// The job of the reference's VRAM planner (falcon_model_load_internal, libfalcon.cpp:1660-1900: how many blocks fit the
// device) for a fully resident layer pipeline: split the blocks of a GGCC file over n_stages contiguous ranges so that the
// slowest stage streams as few weight bytes per token as possible (the first stage also owns the embedding, which it only
// gathers from; the last one ln_f + lm_head, which it streams), and report what each stage needs in device memory:
// weights, n_streams KV caches of n_ctx positions, activations of n_batch tokens. Host-only (no device needed).
// Returns 0, 1 if a stage exceeds vram_per_gpu (> 0), -1 on error.
extern "C" int falcon_hip_plan_stages(const char * path, int n_stages, int n_ctx, int n_batch, int n_streams, size_t vram_per_gpu,
                                      int * layer_begin, int * layer_end, size_t * stage_bytes) {
    ggcc_file f;
    if (!ggcc_open(path, f)) { fprintf(stderr, "falcon-hip: %s: %s\n", path, f.error.c_str()); return -1; }
    const int L = f.hp.n_layer;
    if (n_stages < 1 || n_stages > L || n_ctx < 1 || n_batch < 1 || n_streams < 1) { fprintf(stderr, "falcon-hip: plan: %d stages for %d blocks\n", n_stages, L); return -1; }
    std::vector<size_t> blk((size_t) L, 0);
    size_t emb = 0, head = 0;
    for (const ggcc_tensor & t : f.tensors) {
        int il = -1;
        if (sscanf(t.name.c_str(), "transformer.h.%d.", &il) == 1 && il >= 0 && il < L) blk[(size_t) il] += t.size;
        else if (t.name == "transformer.word_embeddings.weight") emb = t.size;
        else head += t.size;                                               // ln_f, lm_head
    }
    // contiguous partition minimising the largest streamed-bytes load: cost[s][i] = best max over the first s stages covering
    // blocks [0, i); the last stage's load includes the head
    std::vector<size_t> pre((size_t) L + 1, 0);
    for (int i = 0; i < L; ++i) pre[(size_t) i + 1] = pre[(size_t) i] + blk[(size_t) i];
    const size_t INF = ~(size_t) 0;
    std::vector<std::vector<size_t>> best((size_t) n_stages + 1, std::vector<size_t>((size_t) L + 1, INF));
    std::vector<std::vector<int>> cut((size_t) n_stages + 1, std::vector<int>((size_t) L + 1, 0));
    best[0][0] = 0;
    for (int s = 1; s <= n_stages; ++s)
        for (int i = s; i <= L - (n_stages - s); ++i)
            for (int j = s - 1; j < i; ++j) {
                if (best[(size_t) s - 1][(size_t) j] == INF) continue;
                size_t load = pre[(size_t) i] - pre[(size_t) j];
                if (s == n_stages) { if (i != L) continue; load += head; }
                const size_t c = load > best[(size_t) s - 1][(size_t) j] ? load : best[(size_t) s - 1][(size_t) j];
                if (c < best[(size_t) s][(size_t) i]) { best[(size_t) s][(size_t) i] = c; cut[(size_t) s][(size_t) i] = j; }
            }
    int end = L;
    for (int s = n_stages; s >= 1; --s) { const int b = cut[(size_t) s][(size_t) end]; layer_begin[s - 1] = b; layer_end[s - 1] = end; end = b; }
    const size_t E = (size_t) f.hp.n_embd, QKV = (size_t)(f.hp.n_head + 2 * f.hp.n_head_kv) * 64, FF = (size_t) f.hp.n_ff, V = (size_t) f.hp.n_vocab;
    int over = 0;
    for (int s = 0; s < n_stages; ++s) {
        const size_t nb = (size_t)(layer_end[s] - layer_begin[s]);
        size_t bytes = pre[(size_t) layer_end[s]] - pre[(size_t) layer_begin[s]];
        if (s == 0) bytes += emb;
        if (s == n_stages - 1) bytes += head;
        const size_t kv = 2 * nb * (size_t) n_ctx * (size_t) f.hp.n_head_kv * 64 * 4;                  // f32 K and V per context
        size_t act = (size_t) n_batch * (5 * E + QKV + FF) * 4 + (size_t) n_batch * (3 * E + FF) * 5 / 4;   // f32 rows + Q8 images
        if (s == n_stages - 1) act += (size_t) n_batch * V * 4;
        bytes += (size_t) n_streams * (kv + act);
        if (stage_bytes) stage_bytes[s] = bytes;
        if (vram_per_gpu && bytes > vram_per_gpu) over = 1;
    }
    return over;
}

// ------------------------------------------------------------------------------------------------ model quantization
// falcon_model_quantize (libfalcon.cpp:3533-3743, 3914-3925) for GGCC v10 input: header and vocabulary are copied with the
// new ftype, every 2-D tensor whose name ends in "weight" is converted to the ftype's tensor type (lm_head.weight only
// when quantize_output_tensor), everything else is copied. The conversion itself runs on the device
// (kernels_wquant.hip), a tensor at a time: the f32 source of the largest Falcon tensor is 2.1 GB. Output files are
// byte-identical with the reference's (see fq_wquant.h for the one exception its uninitialised array allows).
namespace {
int ftype_tensor_type(int ftype) {                                // libfalcon.cpp:3538-3562
    switch (ftype) {
        case 0: return FQ_F32;  case 1: return 1 /* F16 */;
        case 2: return FQ_Q4_0; case 3: return FQ_Q4_1; case 7: return FQ_Q8_0; case 8: return FQ_Q5_0; case 9: return FQ_Q5_1;
        case 10: return FQ_Q2_K; case 11: case 12: case 13: return FQ_Q3_K; case 14: case 15: return FQ_Q4_K;
        case 16: case 17: return FQ_Q5_K; case 18: return FQ_Q6_K;
    }
    return -1;
}
bool ends_with(const std::string & s, const char * tail) { const size_t n = strlen(tail); return s.size() >= n && s.compare(s.size() - n, n, tail) == 0; }
struct dev_buf {
    void * p = nullptr;
    explicit dev_buf(size_t n) { p = ggml_hip_malloc(n ? n : 16); }
    ~dev_buf() { ggml_hip_free(p); }
};
}   // namespace

extern "C" int falcon_hip_model_quantize(const char * path_in, const char * path_out, int ftype, int quantize_output_tensor,
                                         int allow_requantize, int64_t * hist_out) {
    const int qtype = ftype_tensor_type(ftype);
    if (qtype < 0) { fprintf(stderr, "falcon_hip_model_quantize: failed to quantize: invalid output file type %d\n", ftype); return 1; }
    ggcc_file f;
    if (!ggcc_open(path_in, f)) { fprintf(stderr, "falcon_hip_model_quantize: failed to quantize: %s: %s\n", path_in, f.error.c_str()); return 1; }
    if (ggml_hip_init(-1) <= 0) { fprintf(stderr, "falcon_hip_model_quantize: no HIP device\n"); return 1; }
    FILE * out = fopen(path_out, "wb");
    if (!out) { fprintf(stderr, "falcon_hip_model_quantize: failed to quantize: cannot open %s\n", path_out); return 1; }
    bool ok = true;
    auto put = [&](const void * p, size_t n) { if (n && fwrite(p, 1, n, out) != n) ok = false; };
    auto put_u32 = [&](uint32_t v) { put(&v, 4); };
    // magic, version, hparams with the new ftype, vocabulary + merges verbatim (llama_file_saver, libfalcon.cpp:975-1025)
    put(f.base, 8 + 6 * 4);
    put_u32((uint32_t) ftype);
    put(f.base + 8 + 7 * 4, 4);
    put(f.base + f.vocab_begin, f.vocab_end - f.vocab_begin);
    int64_t hist_all[16] = {0};
    dev_buf hist_dev(16 * sizeof(int64_t));
    const bool q_is_k = qtype >= FQ_Q2_K && qtype <= FQ_Q6_K;
    for (const ggcc_tensor & t : f.tensors) {
        bool quantize = ends_with(t.name, "weight") && t.n_dims == 2;                 // libfalcon.cpp:3609-3616
        quantize = quantize && (quantize_output_tensor || t.name != "lm_head.weight");
        quantize = quantize && qtype != t.type;
        int new_type = t.type;
        std::vector<uint8_t> converted;
        const uint8_t * data = f.base + t.offset;
        size_t new_size = t.size;
        if (quantize) {
            new_type = qtype;
            if (q_is_k && t.ne[0] % 256 != 0) {                                        // libfalcon.cpp:3634-3643
                fprintf(stderr, "falcon_hip_model_quantize: failed to quantize: tensor %s: row length %lld is not divisible by 256 - use a legacy format\n",
                        t.name.c_str(), (long long) t.ne[0]);
                ok = false; break;
            }
            const int64_t n = t.ne[0] * t.ne[1];
            dev_buf f32((size_t) n * 4);
            if (t.type == FQ_F32) ggml_hip_memcpy_h2d(f32.p, data, (size_t) n * 4);
            else if (t.type == 1) {
                dev_buf h((size_t) n * 2);
                ggml_hip_memcpy_h2d(h.p, data, (size_t) n * 2);
                ggml_hip_fp16_to_fp32_row((const uint16_t *) h.p, (float *) f32.p, n);
                ggml_hip_synchronize();
            } else {
                if (!allow_requantize) {
                    fprintf(stderr, "falcon_hip_model_quantize: failed to quantize: requantizing from type %d is disabled\n", t.type);
                    ok = false; break;
                }
                ggml_hip_weight * w = ggml_hip_weight_upload(t.type, data, t.ne[0], t.ne[1]);
                ggml_hip_dequantize_rows(w, nullptr, t.ne[1], (float *) f32.p);
                ggml_hip_synchronize();
                ggml_hip_weight_free(w);
            }
            new_size = tensor_bytes(new_type, t.ne[0], t.ne[1]);
            if (!new_size) { fprintf(stderr, "falcon_hip_model_quantize: failed to quantize: tensor %s: row length %lld does not fit type %d\n", t.name.c_str(), (long long) t.ne[0], new_type); ok = false; break; }
            converted.resize(new_size);
            if (new_type == FQ_F32) ggml_hip_memcpy_d2h(converted.data(), f32.p, new_size);
            else {
                dev_buf q(new_size);
                if (new_type == 1) fq_launch_f32_to_f16((const float *) f32.p, (uint16_t *) q.p, n, (hipStream_t) ggml_hip_stream());
                else {
                    ggml_hip_memset(hist_dev.p, 0, 16 * sizeof(int64_t));
                    if (ggml_hip_quantize_rows(new_type, (const float *) f32.p, t.ne[0], t.ne[1], q.p, (int64_t *) hist_dev.p) != 0) { ok = false; break; }
                    int64_t h[16];
                    ggml_hip_memcpy_d2h(h, hist_dev.p, sizeof h);
                    for (int i = 0; i < 16; ++i) hist_all[i] += h[i];
                }
                ggml_hip_memcpy_d2h(converted.data(), q.p, new_size);
            }
            data = converted.data();
        }
        // tensor record (llama_file_saver::write_tensor, libfalcon.cpp:1026-1051)
        put_u32((uint32_t) t.n_dims); put_u32((uint32_t) t.name.size()); put_u32((uint32_t) new_type);
        for (int d = 0; d < t.n_dims; ++d) put_u32((uint32_t) t.ne[d]);
        put(t.name.data(), t.name.size());
        static const uint8_t zeros[32] = {0};
        put(zeros, (size_t)(-(int64_t) ftell(out) & 31));
        put(data, new_size);
        if (!ok) break;
    }
    if (fclose(out) != 0) ok = false;
    if (!ok) { fprintf(stderr, "falcon_hip_model_quantize: failed to quantize: write to %s incomplete\n", path_out); remove(path_out); return 1; }
    if (hist_out) memcpy(hist_out, hist_all, sizeof hist_all);
    return 0;
}
