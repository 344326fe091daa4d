// ggml_hip_ops.hip -- the operator-level C ABI (include/ggml-hip-ops.h) on top of the kernels.
#include "fq_device.h"
#include "kernels.h"
#include "hip_context.h"
#include "../../include/ggml-hip-ops.h"

#include <math.h>
#include <mutex>
#include <string.h>
#include <vector>

struct ggml_hip_weight { fq_weight w; void * slab; int64_t valid_rows = 0; };      // valid_rows > 0: a row-split part padded with zero rows (fq_weight_upload_part)
struct ggml_hip_acts   { fq_act a; int64_t max_cols; void * slab; };

static hip_context g_ctx;
static std::once_flag g_once;

hip_context & fq_ctx() {
    if (!g_ctx.ready) { fprintf(stderr, "ggml-hip: not initialised (call ggml_hip_init / ggml_init_cublas first)\n"); exit(1); }
    return g_ctx;
}

// GELU / EXP fp16 tables, computed on the host with the host libm exactly as ggml_init does (ggml.c:4276-4290)
static void build_tables(hip_context & c) {
    std::vector<uint16_t> gelu(1 << 16), ex(1 << 16);
    for (uint32_t i = 0; i < (1u << 16); ++i) {
        const uint16_t hb = (uint16_t) i;
        const float f = (float) __builtin_bit_cast(_Float16, hb);
        const float g = 0.5f * f * (1.0f + tanhf(0.79788456080286535587989211986876f * f * (1.0f + 0.044715f * f * f)));
        gelu[i] = __builtin_bit_cast(uint16_t, (_Float16) g);
        ex[i]   = __builtin_bit_cast(uint16_t, (_Float16) expf(f));
    }
    HIP_CHECK(hipMalloc((void **) &c.gelu_table, 1 << 17));
    HIP_CHECK(hipMalloc((void **) &c.exp_table, 1 << 17));
    HIP_CHECK(hipMemcpy(c.gelu_table, gelu.data(), 1 << 17, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(c.exp_table, ex.data(), 1 << 17, hipMemcpyHostToDevice));
    HIP_CHECK(hipMalloc((void **) &c.ks_scratch, FQ_KS_FLOATS * 4));
}

extern "C" int ggml_hip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int ggml_hip_init(int device) {
    std::call_once(g_once, [&]() {
        int n = 0;
        hipError_t e = hipGetDeviceCount(&n);
        if (e != hipSuccess || n == 0) {
            fprintf(stderr, "ggml-hip: no HIP device visible (%s) -- this backend has no CPU fallback\n", hipGetErrorString(e));
            exit(1);
        }
        if (device < 0 || device >= n) device = 0;
        HIP_CHECK(hipSetDevice(device));
        g_ctx.device = device;
        g_ctx.n_devices = n;
        HIP_CHECK(hipStreamCreateWithFlags(&g_ctx.stream, hipStreamNonBlocking));
        hipDeviceProp_t prop;
        HIP_CHECK(hipGetDeviceProperties(&prop, device));
        g_ctx.n_cu = prop.multiProcessorCount;
        snprintf(g_ctx.name, sizeof(g_ctx.name), "%s", prop.name);
        build_tables(g_ctx);
        // the attention kernels gather exp() from the fp16 table (round 5, measured A/B/A/B on one MI355X: Falcon-7B Q4_0 decode 1 022 tok/s against 1 004 with the
        // in-kernel recomputation -- rounds 1-4 had it the other way round by ~0.5 %). GGML_HIP_EXP_FORMULA=1 selects the recomputation, and then only if it
        // reproduces the host-built table for every input (tests/test_gpu_block_ops.py::test_exp_formula_reproduces_table checks that on every run)
        g_ctx.exp_table_attn = g_ctx.exp_table;
        if (getenv("GGML_HIP_EXP_FORMULA") && atoi(getenv("GGML_HIP_EXP_FORMULA")) && !getenv("GGML_HIP_EXP_TABLE") && fq_verify_exp_formula(g_ctx.exp_table, g_ctx.stream) == 0) g_ctx.exp_table_attn = nullptr;
        HIP_CHECK(hipMalloc((void **) &g_ctx.scalar_i32, 256));
        g_ctx.ready = true;
        if (const char * e = getenv("GGML_HIP_REFERENCE_ORDER")) ggml_hip_reference_order(atoi(e));      // (callers that only know ggml-cuda.h)
    });
    // one process drives one GPU: a later call naming another device cannot re-bind the library
    if (device >= 0 && device < g_ctx.n_devices && device != g_ctx.device)
        fprintf(stderr, "ggml-hip: already initialised on device %d; the request for device %d is ignored (one process per GPU)\n", g_ctx.device, device);
    return g_ctx.n_devices;
}

// phase stamps of the fused decode kernels: enable -> allocate/zero the buffer; read -> copy 2*4096*8 int64 to the host
extern "C" void ggml_hip_debug_stamps(int enable, long long * out_host) {
    hip_context & c = fq_ctx();
    const size_t n = 2 * 4096 * 8 * sizeof(long long);
    if (enable && !c.dbg_stamps) { HIP_CHECK(hipMalloc((void **) &c.dbg_stamps, n)); HIP_CHECK(hipMemset(c.dbg_stamps, 0, n)); }
    if (out_host && c.dbg_stamps) { HIP_CHECK(hipStreamSynchronize(c.stream)); HIP_CHECK(hipMemcpy(out_host, c.dbg_stamps, n, hipMemcpyDeviceToHost)); }
    if (!enable && c.dbg_stamps) { HIP_CHECK(hipFree(c.dbg_stamps)); c.dbg_stamps = nullptr; }
}

// every switch that changes which kernels a launch list contains bumps this: captured launch lists (hipGraphs) are keyed by it
static int g_config_epoch = 0;
int fq_config_epoch() { return g_config_epoch; }
extern "C" void ggml_hip_debug_gemm_mode(int m) { ++g_config_epoch; fq_gemm_debug_mode(m); }
extern "C" int ggml_hip_selftest(void) { return fq_selftest_reduce(fq_ctx().stream); }
// 0 = the fp16 EXP table is recomputed in-kernel (verified identical at init), else the number of mismatching inputs / -1 forced gather
extern "C" int ggml_hip_exp_formula_mismatches(void) {
    if (getenv("GGML_HIP_EXP_TABLE")) return -1;
    return fq_verify_exp_formula(fq_ctx().exp_table, fq_ctx().stream);
}

extern "C" void * ggml_hip_stream(void) { return (void *) fq_ctx().stream; }

extern "C" void * ggml_hip_malloc(size_t bytes) {
    fq_ctx();
    void * p = nullptr;
    HIP_CHECK(hipMalloc(&p, bytes ? bytes : 16));
    return p;
}
extern "C" void ggml_hip_free(void * dev) { if (dev) HIP_CHECK(hipFree(dev)); }
extern "C" void ggml_hip_memcpy_h2d(void * d, const void * s, size_t n) {
    HIP_CHECK(hipMemcpyAsync(d, s, n, hipMemcpyHostToDevice, fq_ctx().stream));
    HIP_CHECK(hipStreamSynchronize(fq_ctx().stream));
}
extern "C" void ggml_hip_memcpy_d2h(void * d, const void * s, size_t n) {
    HIP_CHECK(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToHost, fq_ctx().stream));
    HIP_CHECK(hipStreamSynchronize(fq_ctx().stream));
}
extern "C" void ggml_hip_memcpy_d2d(void * d, const void * s, size_t n) {
    HIP_CHECK(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToDevice, fq_ctx().stream));
}
extern "C" void ggml_hip_memset(void * d, int v, size_t n) { HIP_CHECK(hipMemsetAsync(d, v, n, fq_ctx().stream)); }
extern "C" void ggml_hip_synchronize(void) { HIP_CHECK(hipStreamSynchronize(fq_ctx().stream)); }

extern "C" void * ggml_hip_event_create(void) { hipEvent_t e; fq_ctx(); HIP_CHECK(hipEventCreate(&e)); return (void *) e; }
extern "C" void   ggml_hip_event_record(void * ev) { HIP_CHECK(hipEventRecord((hipEvent_t) ev, fq_ctx().stream)); }
extern "C" float  ggml_hip_event_elapsed_ms(void * a, void * b) {
    HIP_CHECK(hipEventSynchronize((hipEvent_t) b));
    float ms = 0.0f; HIP_CHECK(hipEventElapsedTime(&ms, (hipEvent_t) a, (hipEvent_t) b)); return ms;
}
extern "C" void   ggml_hip_event_destroy(void * ev) { HIP_CHECK(hipEventDestroy((hipEvent_t) ev)); }
extern "C" const uint16_t * ggml_hip_gelu_table_dev(void) { return fq_ctx().gelu_table; }
extern "C" const uint16_t * ggml_hip_exp_table_dev(void)  { return fq_ctx().exp_table; }

// ------------------------------------------------------------------------------------------------ weights
fq_weight fq_weight_alloc(int type, int64_t K, int64_t M, void ** slab_out) {
    const fq_type_desc d = fq_desc(type);
    if (d.blck == 0 || K % d.blck != 0) { fprintf(stderr, "ggml-hip: weight type %d with K=%lld unsupported\n", type, (long long) K); exit(1); }
    fq_weight w{};
    w.type = type; w.K = K; w.M = M; w.nblk = K / d.blck;
    w.bytes = (size_t) M * w.nblk * d.tsize;
    w.row_stride = fq_il_row_stride(d, w.nblk);               // one slab, rows of row_stride bytes (fq_types.h)
    const size_t total = (size_t) M * w.row_stride;
    uint8_t * slab = nullptr;
    HIP_CHECK(hipMalloc((void **) &slab, total + 2048));       // slack: clamped tail loads / the engine's last 1 KiB DMA piece never leave the allocation
    for (int p = 0; p < d.nplanes; ++p) w.plane[p] = slab;
    *slab_out = slab;
    return w;
}

// rows [0, rows) from host_blocks; alloc_rows >= rows: the rest are zero rows (all-zero blocks: every weight 0.0)
static ggml_hip_weight * weight_upload_rows(int type, const void * host_blocks, int64_t K, int64_t rows, int64_t alloc_rows) {
    hip_context & c = fq_ctx();
    ggml_hip_weight * hw = new ggml_hip_weight();
    hw->w = fq_weight_alloc(type, K, alloc_rows, &hw->slab);
    if (alloc_rows > rows) HIP_CHECK(hipMemsetAsync(hw->w.plane[0] + (size_t) rows * hw->w.row_stride, 0, (size_t)(alloc_rows - rows) * hw->w.row_stride, c.stream));
    // stage the ggml bytes in HBM in bounded chunks of rows, re-tile on the device
    const fq_type_desc d = fq_desc(type);
    const size_t row_bytes = (size_t) hw->w.nblk * d.tsize;
    const int64_t M = rows;
    int64_t rows_per_chunk = (int64_t) ((256u << 20) / (row_bytes ? row_bytes : 1));
    if (rows_per_chunk < 1) rows_per_chunk = 1;
    if (rows_per_chunk > M) rows_per_chunk = M;
    uint8_t * stage = nullptr;
    HIP_CHECK(hipMalloc((void **) &stage, (size_t) rows_per_chunk * row_bytes));
    for (int64_t r0 = 0; r0 < M; r0 += rows_per_chunk) {
        const int64_t nr = (M - r0 < rows_per_chunk) ? M - r0 : rows_per_chunk;
        HIP_CHECK(hipMemcpyAsync(stage, (const uint8_t *) host_blocks + (size_t) r0 * row_bytes, (size_t) nr * row_bytes, hipMemcpyHostToDevice, c.stream));
        fq_weight sub = hw->w;
        sub.M = nr;
        for (int p = 0; p < d.nplanes; ++p)
            sub.plane[p] = hw->w.plane[0] + (size_t) r0 * hw->w.row_stride;
        fq_launch_retile(stage, sub, c.stream);
        HIP_CHECK(hipStreamSynchronize(c.stream));
    }
    HIP_CHECK(hipFree(stage));
    return hw;
}
extern "C" ggml_hip_weight * ggml_hip_weight_upload(int type, const void * host_blocks, int64_t K, int64_t M) { return weight_upload_rows(type, host_blocks, K, M, M); }
// A row range of a matrix of `whole_rows` rows (row-split tensor parallelism, split_tp.hip): it must sum every row in the association the UNSPLIT matrix would
// use, whatever its own shape. form_M carries the whole matrix's rows to the choices that fix the association (fq_types.h); the k-quants' small-batch form works
// in whole 16-row tiles and the reference's row ranges are not rounded (ggml-cuda.cu:3046-3047), so a k-quant part is padded with zero rows to a multiple of 16
// and ggml_hip_mul_mat_q keeps the padding's results out of dst.
ggml_hip_weight * fq_weight_upload_part(int type, const void * host_blocks, int64_t K, int64_t rows, int64_t whole_rows) {
    const bool kq = type == FQ_Q2_K || type == FQ_Q3_K || type == FQ_Q4_K || type == FQ_Q5_K || type == FQ_Q6_K;
    const int64_t alloc_rows = kq && whole_rows % 16 == 0 ? (rows + 15) & ~(int64_t) 15 : rows;
    ggml_hip_weight * hw = weight_upload_rows(type, host_blocks, K, rows, alloc_rows);
    hw->w.form_M = whole_rows;
    if (alloc_rows > rows) hw->valid_rows = rows;
    return hw;
}
extern "C" void ggml_hip_weight_free(ggml_hip_weight * w) { if (!w) return; HIP_CHECK(hipFree(w->slab)); delete w; }
// (a row-split part of a k-quant matrix is padded with zero rows to whole 16-row tiles on the device: report the bytes of its real rows)
extern "C" size_t ggml_hip_weight_nbytes(const ggml_hip_weight * w) {
    if (w->valid_rows > 0 && w->valid_rows < w->w.M) return (size_t)((w->w.bytes / (size_t) w->w.M) * (size_t) w->valid_rows);
    return w->w.bytes;
}

// ---- weight quantizers (kernels_wquant.hip): what ggml_quantize_chunk (ggml.c:19479-19560) writes into a model file
extern "C" int ggml_hip_quantize_rows(int type, const float * x_dev, int64_t K, int64_t nrows, void * blocks_dev, int64_t * hist_dev) {
    const fq_type_desc d = fq_desc(type);
    if (d.blck == 0 || K <= 0 || K % d.blck != 0 || nrows < 0) {
        fprintf(stderr, "ggml-hip: cannot quantize rows of %lld weights to type %d (row length must be a multiple of %d)\n", (long long) K, type, d.blck);
        return -1;
    }
    if (!fq_launch_wquant(type, x_dev, K * nrows, (uint8_t *) blocks_dev, (unsigned long long *) hist_dev, fq_ctx().stream)) return -1;
    return 0;
}
extern "C" void ggml_hip_fp16_to_fp32_row(const uint16_t * src_dev, float * dst_dev, int64_t n) {
    fq_launch_f16_to_f32(src_dev, dst_dev, n, fq_ctx().stream);
}
extern "C" ggml_hip_weight * ggml_hip_weight_quantize(int type, const float * x_dev, int64_t K, int64_t M) {
    hip_context & c = fq_ctx();
    const fq_type_desc d = fq_desc(type);
    if (d.blck == 0 || K % d.blck != 0) { fprintf(stderr, "ggml-hip: weight type %d with K=%lld unsupported\n", type, (long long) K); return nullptr; }
    ggml_hip_weight * hw = new ggml_hip_weight();
    hw->w = fq_weight_alloc(type, K, M, &hw->slab);
    uint8_t * blocks = nullptr;
    HIP_CHECK(hipMalloc((void **) &blocks, hw->w.bytes + 16));
    fq_launch_wquant(type, x_dev, K * M, blocks, nullptr, c.stream);
    fq_launch_retile(blocks, hw->w, c.stream);
    HIP_CHECK(hipStreamSynchronize(c.stream));
    HIP_CHECK(hipFree(blocks));
    return hw;
}

extern "C" void ggml_hip_dequantize_rows(const ggml_hip_weight * w, const int32_t * rows_dev, int64_t nrows, float * dst_dev) {
    fq_launch_dequant_rows(w->w, rows_dev, nrows, dst_dev, fq_ctx().stream);
}

// ------------------------------------------------------------------------------------------------ activations
fq_act fq_act_alloc(int act_type, int64_t K, int64_t max_cols, void ** slab_out) {
    if (act_type != FQ_Q8_0 && act_type != FQ_Q8_1 && act_type != FQ_Q8_K) { fprintf(stderr, "ggml-hip: bad activation type %d\n", act_type); exit(1); }
    if (K % (act_type == FQ_Q8_K ? 256 : 32)) { fprintf(stderr, "ggml-hip: K=%lld not a multiple of the activation block\n", (long long) K); exit(1); }
    fq_act a{};
    a.type = act_type; a.K = K; a.ncols = max_cols;
    uint8_t * slab = nullptr;
    HIP_CHECK(hipMalloc((void **) &slab, fq_act_col_bytes(act_type, K) * (size_t) max_cols + 256));
    a.base = slab;
    *slab_out = slab;
    return a;
}
extern "C" ggml_hip_acts * ggml_hip_acts_alloc(int act_type, int64_t K, int64_t max_cols) {
    fq_ctx();
    ggml_hip_acts * h = new ggml_hip_acts();
    h->a = fq_act_alloc(act_type, K, max_cols, &h->slab);
    h->max_cols = max_cols;
    return h;
}
extern "C" void ggml_hip_acts_free(ggml_hip_acts * a) { if (!a) return; HIP_CHECK(hipFree(a->slab)); delete a; }
extern "C" void ggml_hip_quantize_acts(ggml_hip_acts * a, const float * x_dev, int64_t ldx, int64_t ncols) {
    if (ncols > a->max_cols) { fprintf(stderr, "ggml-hip: quantize_acts: %lld columns > capacity %lld\n", (long long) ncols, (long long) a->max_cols); exit(1); }
    fq_act v = a->a; v.ncols = ncols;
    fq_launch_quantize_act(x_dev, ldx, v, fq_ctx().stream);
}
extern "C" void ggml_hip_acts_export(const ggml_hip_acts * a, int64_t ncols, void * out_dev) {
    fq_act v = a->a; v.ncols = ncols;
    fq_launch_act_export(v, (uint8_t *) out_dev, fq_ctx().stream);
}

// ------------------------------------------------------------------------------------------------ mat-mul
// column c of an activation set (SoA arrays are [ncols][...] so a column range is a pointer offset)
static fq_act act_cols(const fq_act & a, int64_t c0, int64_t n) {
    fq_act v = a;
    v.base = a.base + (size_t) c0 * fq_act_col_bytes(a.type, a.K);
    v.ncols = n;
    return v;
}

static bool g_force_gemv = false;      // tests: run N > 4 through the mat-vec kernel (column chunks) instead of the MFMA GEMM
extern "C" void ggml_hip_debug_force_gemv(int on) { ++g_config_epoch; g_force_gemv = on != 0; }
extern "C" void ggml_hip_debug_attention_form(int form) { ++g_config_epoch; fq_attn_set_form(form); }
// diagnostic (scripts/gpu_exp_boundary.py generates csrc/fq_exp_fix.h from it): fp16 inputs whose exp() the f32 fast path of the in-kernel formula cannot decide
extern "C" int ggml_hip_debug_exp_boundary(unsigned * out_host, int cap) { return fq_exp_boundary(fq_ctx().exp_table, out_host, cap, fq_ctx().stream); }
extern "C" void ggml_hip_gemm_sequential(int on) { ++g_config_epoch; fq_gemm_set_sequential(on); }
// reference order: every mat-mul through the per-thread scalar restatement (kernels_ref.hip: the reference's own block /
// lane order for all ten formats and any N), attention with f64 accumulation (the portable ggml_vec_dot_f32)
// mode 2 (round 6) = the FAST reference order: the same association -- results bit-identical to mode 1 and to the reference's scalar build -- on the fast kernels
// where they have it: legacy formats, N = 1 through the fused decode launches (fq_ref_chain.h: per-block terms into an LDS strip, lane = row adds them left to
// right; f64 attention dots), N > 4 through the int8-MFMA GEMM with one left-to-right sum per row (ggml_hip_gemm_sequential). Everything else as mode 1.
static int g_reference_order = 0;
bool fq_reference_order() { return g_reference_order != 0; }
bool fq_reference_fast() { return g_reference_order == 2; }
extern "C" void ggml_hip_reference_order(int on) { ++g_config_epoch; g_reference_order = on == 2 ? 2 : (on != 0 ? 1 : 0); fq_gemm_set_sequential(on != 0); fq_attn_set_f64(on != 0); }
extern "C" int  ggml_hip_get_reference_order(void) { return g_reference_order; }
static bool legacy_type(int t) { return t == FQ_Q4_0 || t == FQ_Q4_1 || t == FQ_Q5_0 || t == FQ_Q5_1 || t == FQ_Q8_0; }

// ---- optional per-launch timing of the GEMV kernels (bench.py roofline leg): hipEvents on the launch stream
static bool g_prof_on = false;
static std::vector<std::pair<hipEvent_t, hipEvent_t>> g_prof_ev;
static double g_prof_bytes = 0.0;

extern "C" void ggml_hip_profile_begin(void) {
    for (auto & p : g_prof_ev) { (void) hipEventDestroy(p.first); (void) hipEventDestroy(p.second); }
    g_prof_ev.clear(); g_prof_bytes = 0.0; g_prof_on = true;
}
extern "C" void ggml_hip_profile_end(int64_t * n_launches, double * total_us, double * total_bytes) {
    g_prof_on = false;
    HIP_CHECK(hipStreamSynchronize(fq_ctx().stream));
    double us = 0.0;
    for (auto & p : g_prof_ev) { float ms = 0.0f; HIP_CHECK(hipEventElapsedTime(&ms, p.first, p.second)); us += 1e3 * (double) ms; }
    *n_launches = (int64_t) g_prof_ev.size(); *total_us = us; *total_bytes = g_prof_bytes;
    for (auto & p : g_prof_ev) { (void) hipEventDestroy(p.first); (void) hipEventDestroy(p.second); }
    g_prof_ev.clear();
}

// cost of an empty hipEventRecord pair on the launch stream (information only: the profile brackets do not record markers)
extern "C" double ggml_hip_profile_bracket_overhead_us(void) {
    hip_context & c = fq_ctx();
    const int n = 64;
    std::vector<hipEvent_t> ev(2 * n);
    for (auto & e : ev) HIP_CHECK(hipEventCreate(&e));
    HIP_CHECK(hipStreamSynchronize(c.stream));
    for (int i = 0; i < n; ++i) { HIP_CHECK(hipEventRecord(ev[2 * i], c.stream)); HIP_CHECK(hipEventRecord(ev[2 * i + 1], c.stream)); }
    HIP_CHECK(hipStreamSynchronize(c.stream));
    double us = 0.0;
    for (int i = 0; i < n; ++i) { float ms = 0.0f; HIP_CHECK(hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1])); us += 1e3 * (double) ms; }
    for (auto & e : ev) HIP_CHECK(hipEventDestroy(e));
    return us / n;
}

// bracket one launch (used by the fused decode path in falcon_hip.hip)
bool fq_prof_active() { return g_prof_on; }
// A profile bracket = the pair of events the NEXT fused mat-vec launch hands to hipExtLaunchKernelGGL: the runtime stamps
// them with the dispatch's own begin / end times (the clock rocprofv3's kernel trace reads), so the elapsed time is the
// kernel's duration without the cost of two extra marker packets (~4.6 us for an empty hipEventRecord pair).
static hipEvent_t g_prof_pending[2] = { nullptr, nullptr };
void fq_prof_open(hipStream_t) {
    hipEvent_t e0, e1;
    HIP_CHECK(hipEventCreate(&e0)); HIP_CHECK(hipEventCreate(&e1));
    g_prof_ev.emplace_back(e0, e1);
    g_prof_pending[0] = e0; g_prof_pending[1] = e1;
}
void fq_prof_cancel() {                                             // the bracket just opened enclosed no launch after all
    if (g_prof_ev.empty()) return;
    (void) hipEventDestroy(g_prof_ev.back().first); (void) hipEventDestroy(g_prof_ev.back().second);
    g_prof_ev.pop_back();
    g_prof_pending[0] = g_prof_pending[1] = nullptr;
}
void fq_prof_events(hipEvent_t * start, hipEvent_t * stop) { *start = g_prof_pending[0]; *stop = g_prof_pending[1]; }
void fq_prof_close(hipStream_t, double bytes) {
    g_prof_pending[0] = g_prof_pending[1] = nullptr;
    g_prof_bytes += bytes;
}

// two mat-muls behind the same activation columns (Wqkv and Wup of a one-norm block) as ONE launch of the small-batch form, where that
// applies (5..16 columns, default order, same format / K / K split); false: nothing launched, the caller runs them one by one
bool fq_mul_mat_q_acts_pair(const fq_weight & w0, const fq_weight & w1, const fq_act & a, int64_t N, float * dst0, int64_t ldd0, const fq_gemv_epi & ep0,
                            float * dst1, int64_t ldd1, const fq_gemv_epi & ep1, hipStream_t st) {
    static const bool skinny = !(getenv("FQ_GEMM_SKINNY") && atoi(getenv("FQ_GEMM_SKINNY")) == 0) && !getenv("FQ_GEMM_CFG") && !(getenv("FQ_SKINNY_PAIR") && atoi(getenv("FQ_SKINNY_PAIR")) == 0);
    // (round 6) 17..32 columns: two passes of the pair launch, as fq_launch_gemm runs a single matrix of that width as two passes of the streaming form -- the same K split
    // (one token tile row: fq_gemm_split_for does not change between 16 and 32 columns), the same bits, half the launches and the fuller grid of the pair (FQ_SKINNY_PAIR2=0: off)
    static const bool pair2 = !(getenv("FQ_SKINNY_PAIR2") && atoi(getenv("FQ_SKINNY_PAIR2")) == 0) && !(getenv("FQ_GEMM_SKINNY2") && atoi(getenv("FQ_GEMM_SKINNY2")) == 0);
    if (!skinny || g_reference_order || g_force_gemv || N <= FQ_GEMV_MAX_COLS || N > (pair2 ? 32 : 16)) return false;
    if (fq_desc(w0.type).act_type != a.type || a.K != w0.K || w1.K != w0.K || w1.type != w0.type) return false;
    const int n_cu = fq_ctx().n_cu;
    const int S0 = fq_gemm_split_for(w0.M, N, n_cu), S1 = fq_gemm_split_for(w1.M, N, n_cu);
    if (S0 != S1) return false;
    if (N <= 16) return fq_launch_gemm_skinny_pair(w0, w1, a, N, dst0, ldd0, ep0, dst1, ldd1, ep1, S0, st);
    if (fq_skinny_q4k_shape(w0) || fq_skinny_q4k_shape(w1)) return false;                 // (the k-quants' small-batch forms have their own passes)
    for (int64_t n0 = 0; n0 < N; n0 += 16) {
        const int64_t nc = N - n0 < 16 ? N - n0 : 16;
        fq_act a1 = a; a1.ncols = nc; a1.base = a.base + n0 * fq_act_col_bytes(a.type, a.K);
        fq_gemv_epi e0 = ep0, e1 = ep1;
        if (e0.add1) e0.add1 += n0 * e0.ld_add;
        if (e0.add2) e0.add2 += n0 * e0.ld_add;
        if (e1.add1) e1.add1 += n0 * e1.ld_add;
        if (e1.add2) e1.add2 += n0 * e1.ld_add;
        if (!fq_launch_gemm_skinny_pair(w0, w1, a1, nc, dst0 + n0 * ldd0, ldd0, e0, dst1 + n0 * ldd1, ldd1, e1, S0, st)) {
            if (n0 == 0) return false;
            fprintf(stderr, "ggml-hip: pair mat-mul: the second pass refused\n"); exit(1);
        }
    }
    return true;
}

// Q4_K blocks at 5..16 columns, default order: the fused sum launches of the small-batch form. false: nothing launched, the caller runs the generic calls
static bool q4k_fused_ok(int64_t N, int64_t min_cols = FQ_GEMV_MAX_COLS + 1) {
    static const bool on = !(getenv("FQ_GEMM_SKINNY") && atoi(getenv("FQ_GEMM_SKINNY")) == 0) && !getenv("FQ_GEMM_CFG") && !(getenv("FQ_SKINNY_Q4K_FUSED") && atoi(getenv("FQ_SKINNY_Q4K_FUSED")) == 0);
    return on && !g_reference_order && !g_force_gemv && fq_gemm_split_for(16, N, fq_ctx().n_cu) != 1 && N >= min_cols && N <= 16;
}
// (round 6) 17 .. fq_skinny_kq_max_cols columns -- the widths fq_launch_gemm runs as passes of 16 of the same form -- keep the fused sum launches: one pass of 16 columns at a time
static bool q4k_fused_passes(const fq_weight & w, int64_t N) {
    static const bool on = !(getenv("FQ_GEMM_SKINNY2") && atoi(getenv("FQ_GEMM_SKINNY2")) == 0) && !(getenv("FQ_SKINNY_Q4K_FUSED2") && atoi(getenv("FQ_SKINNY_Q4K_FUSED2")) == 0);
    return on && N > 16 && fq_skinny_q4k_shape(w) && N <= fq_skinny_kq_max_cols(w.type) && q4k_fused_ok(16);
}
bool fq_mul_mat_q_acts_gelu_q8k(const fq_weight & w, const fq_act & a, int64_t N, float * dst, int64_t ldd, const fq_act & out, hipStream_t st, int min_cols) {
    if (fq_desc(w.type).act_type != a.type || a.K != w.K) return false;
    if (q4k_fused_passes(w, N)) {
        for (int64_t n0 = 0; n0 < N; n0 += 16) {
            const int64_t nc = N - n0 < 16 ? N - n0 : 16;
            fq_act a1 = a, o1 = out;
            a1.ncols = nc; a1.base = a.base + n0 * fq_act_col_bytes(a.type, a.K);
            o1.ncols = nc; o1.base = out.base + n0 * fq_act_col_bytes(out.type, out.K);
            if (!fq_launch_gemm_skinny_q4k_gelu_q8k(w, a1, nc, dst + n0 * ldd, ldd, fq_ctx().gelu_table, o1, st)) {
                if (n0 == 0) return false;
                fprintf(stderr, "ggml-hip: gelu + Q8_K mat-mul: a later pass refused\n"); exit(1);
            }
        }
        return true;
    }
    if (!q4k_fused_ok(N, min_cols)) return false;
    return fq_launch_gemm_skinny_q4k_gelu_q8k(w, a, N, dst, ldd, fq_ctx().gelu_table, out, st);
}
bool fq_mul_mat_q_acts_out2(const fq_weight & wo, const fq_act & a_att, const fq_weight & down, const fq_act & a_ff, int64_t N, float * x, int64_t ldx, hipStream_t st, int min_cols) {
    if (a_att.K != wo.K || a_ff.K != down.K) return false;
    if (q4k_fused_ok(N, min_cols) && fq_launch_gemm_skinny_q4k_out2(wo, a_att, down, a_ff, N, x, ldx, st)) return true;
    if (q4k_fused_passes(wo, N) && q4k_fused_passes(down, N)) {
        for (int64_t n0 = 0; n0 < N; n0 += 16) {
            const int64_t nc = N - n0 < 16 ? N - n0 : 16;
            fq_act a1 = a_att, a2 = a_ff;
            a1.ncols = nc; a1.base = a_att.base + n0 * fq_act_col_bytes(a_att.type, a_att.K);
            a2.ncols = nc; a2.base = a_ff.base + n0 * fq_act_col_bytes(a_ff.type, a_ff.K);
            if (!fq_launch_gemm_skinny_q4k_out2(wo, a1, down, a2, nc, x + n0 * ldx, ldx, st)) {
                if (n0 == 0) break;
                fprintf(stderr, "ggml-hip: output pair (k-quants): a later pass refused\n"); exit(1);
            }
            if (n0 + 16 >= N) return true;
        }
    }
    // legacy formats (round 6): both matrices in the K-share form in one launch; 17..32 columns as two passes of it (the streaming forms' own rule for that width)
    static const bool skinny = !(getenv("FQ_GEMM_SKINNY") && atoi(getenv("FQ_GEMM_SKINNY")) == 0) && !getenv("FQ_GEMM_CFG");
    static const bool skinny2 = !(getenv("FQ_GEMM_SKINNY2") && atoi(getenv("FQ_GEMM_SKINNY2")) == 0);
    if (!skinny || g_reference_order || g_force_gemv || N <= FQ_GEMV_MAX_COLS || N > (skinny2 ? 32 : 16)) return false;
    if (fq_desc(wo.type).act_type != a_att.type || fq_desc(down.type).act_type != a_ff.type) return false;
    const int n_cu = fq_ctx().n_cu;
    if (fq_gemm_split_for(wo.M, N, n_cu) != 4 || fq_gemm_split_for(down.M, N, n_cu) != 4) return false;
    for (int64_t n0 = 0; n0 < N; n0 += 16) {
        const int64_t nc = N - n0 < 16 ? N - n0 : 16;
        fq_act a1 = a_att, a2 = a_ff;
        a1.ncols = nc; a1.base = a_att.base + n0 * fq_act_col_bytes(a_att.type, a_att.K);
        a2.ncols = nc; a2.base = a_ff.base + n0 * fq_act_col_bytes(a_ff.type, a_ff.K);
        if (!fq_launch_gemm_skinny_out2(wo, a1, down, a2, nc, x + n0 * ldx, ldx, st)) {
            if (n0 == 0) return false;
            fprintf(stderr, "ggml-hip: output pair: the second pass refused\n"); exit(1);
        }
    }
    return true;
}

// lock-step contexts of 3 and 4 sequences on the k-quant formats with a small-batch form at this shape: that form (a pass of 16 columns costs less than
// the column mat-vec kernels' pass of 4 there: Falcon-40B Q2_K 8.2 against 18.5 ms, Q4_K 9.9 against 15.2); everything else: fq_mul_mat_q_acts
void fq_mul_mat_q_acts_from3(const fq_weight & w, const fq_act & a, int64_t N, float * dst, int64_t ldd, const fq_gemv_epi & ep, hipStream_t st) {
    if (N >= 3 && N <= FQ_GEMV_MAX_COLS && q4k_fused_ok(N, 3) && fq_skinny_q4k_shape(w) && fq_desc(w.type).act_type == a.type && a.K == w.K) {
        fq_launch_gemm(w, a, N, dst, ldd, ep, fq_ctx().n_cu, st);
        return;
    }
    fq_mul_mat_q_acts(w, a, N, dst, ldd, ep, st);
}

void fq_mul_mat_q_acts(const fq_weight & w, const fq_act & a, int64_t N, float * dst, int64_t ldd, const fq_gemv_epi & ep0, hipStream_t st) {
    hip_context & c = fq_ctx();
    if (fq_desc(w.type).act_type != a.type || a.K != w.K) { fprintf(stderr, "ggml-hip: mul_mat: activation format/length mismatch\n"); exit(1); }
    // reference order: one thread per output (mode 1), or -- mode 2, legacy formats, batches -- the GEMM below with S = 1 (fq_gemm_set_sequential: one
    // left-to-right sum per row, the scalar build's two roundings per term: == the reference, tests/test_gpu_mul_mat.py)
    // (mode 2, legacy formats: from TWO columns on -- a lock-step context of 2-4 sequences is a 128-token tile of padding, still far ahead of one thread per output)
    if (g_reference_order && !(g_reference_order == 2 && legacy_type(w.type) && N >= 2 && fq_gemm_supported(w.type) && !g_force_gemv)) {
        // mode 2, the k-quants: the wave-speed mat-vec in the reference's association (kernels_kqref.hip), column by column. A prompt re-reads the matrix per
        // token (out of L2 / the Infinity Cache) and is still 4.5 x faster than mode 1's one thread per output (Falcon-40B Q4_K, 128 tokens, 16 blocks: 268 against
        // 1 217 ms). FQ_KQREF_MAX_N=n: mode 1's kernel beyond n columns (A/B)
        static const int64_t kq_max_n = getenv("FQ_KQREF_MAX_N") ? atoll(getenv("FQ_KQREF_MAX_N")) : INT64_MAX;
        // (and the legacy formats' single columns: the op-level API and the ggml-cuda.h shim; batches took the GEMM above)
        if (g_reference_order == 2 && N <= kq_max_n && fq_launch_gemv_kq_ref(w, a, N, dst, ldd, ep0, st)) return;
        fq_launch_mul_mat_ref(w, a, N, dst, ldd, ep0, st); return;
    }
    if ((N > FQ_GEMV_MAX_COLS || (g_reference_order == 2 && N >= 2)) && fq_gemm_supported(w.type) && !g_force_gemv) {      // prefill: int8 MFMA GEMM
        fq_launch_gemm(w, a, N, dst, ldd, ep0, c.n_cu, st);
        return;
    }
    const int max_blocks = c.n_cu * 4;
    int64_t n0 = 0;
    while (n0 < N) {
        const int64_t left = N - n0;
        const int nc = left >= 4 ? 4 : (left >= 2 ? 2 : 1);
        fq_gemv_epi ep = ep0;
        if (ep.add1) ep.add1 += n0 * ep.ld_add;
        if (ep.add2) ep.add2 += n0 * ep.ld_add;
        if (g_prof_on) {
            hipEvent_t e0, e1;
            HIP_CHECK(hipEventCreate(&e0)); HIP_CHECK(hipEventCreate(&e1));
            HIP_CHECK(hipEventRecord(e0, st));
            fq_launch_gemv(w, act_cols(a, n0, nc), nc, dst + n0 * ldd, ldd, ep, max_blocks, st);
            HIP_CHECK(hipEventRecord(e1, st));
            g_prof_ev.emplace_back(e0, e1);
            g_prof_bytes += (double) w.bytes;         // algorithmic bytes of the launch = the weight matrix, read once
        } else {
            fq_launch_gemv(w, act_cols(a, n0, nc), nc, dst + n0 * ldd, ldd, ep, max_blocks, st);
        }
        n0 += nc;
    }
}

extern "C" void ggml_hip_mul_mat_q_acts(const ggml_hip_weight * w, const ggml_hip_acts * a, int64_t N, float * dst_dev,
                                        int64_t ldd, int epilogue, const float * add1_dev, const float * add2_dev) {
    if (w->valid_rows > 0 && w->valid_rows < w->w.M) {      // a padded row-split part would write its zero rows over the next rank's rows of dst
        fprintf(stderr, "ggml-hip: ggml_hip_mul_mat_q_acts on a padded row-split part (%lld of %lld device rows are real): use ggml_hip_mul_mat_q / _split\n",
                (long long) w->valid_rows, (long long) w->w.M);
        exit(1);
    }
    fq_gemv_epi ep{ epilogue, fq_ctx().gelu_table, add1_dev, add2_dev, ldd };
    fq_mul_mat_q_acts(w->w, a->a, N, dst_dev, ldd, ep, fq_ctx().stream);
}

extern "C" void ggml_hip_mul_mat_q(const ggml_hip_weight * w, const float * x_dev, int64_t ldx, int64_t N, float * dst_dev, int64_t ldd) {
    hip_context & c = fq_ctx();
    void * slab = nullptr;
    fq_act a = fq_act_alloc(fq_desc(w->w.type).act_type, w->w.K, N, &slab);
    fq_launch_quantize_act(x_dev, ldx, a, c.stream);
    // a padded row-split part (fq_weight_upload_part): all its rows into a private matrix, the real ones from there into dst
    const bool padded = w->valid_rows > 0 && w->valid_rows < w->w.M;
    float * out = dst_dev; int64_t ldo = ldd;
    static float * priv = nullptr; static size_t priv_bytes = 0;      // grow-only private matrix of the padded parts (the call synchronises before it returns)
    if (padded) {
        ldo = w->w.M;
        const size_t need = (size_t) N * (size_t) ldo * 4;
        if (need > priv_bytes) { if (priv) HIP_CHECK(hipFree(priv)); HIP_CHECK(hipMalloc((void **) &priv, need)); priv_bytes = need; }
        out = priv;
    }
    fq_gemv_epi ep{ FQ_EPI_STORE, c.gelu_table, nullptr, nullptr, ldo };
    fq_mul_mat_q_acts(w->w, a, N, out, ldo, ep, c.stream);
    if (padded) HIP_CHECK(hipMemcpy2DAsync(dst_dev, (size_t) ldd * 4, out, (size_t) ldo * 4, (size_t) w->valid_rows * 4, (size_t) N, hipMemcpyDeviceToDevice, c.stream));
    HIP_CHECK(hipStreamSynchronize(c.stream));
    HIP_CHECK(hipFree(slab));
}


// ------------------------------------------------------------------------------------------------ per-launch timing table (kernels.h)
namespace {
struct tl_rec { const char * name; hipEvent_t e0, e1; };
struct tl_state { bool on = false; int depth = 0; std::vector<tl_rec> recs; std::vector<hipEvent_t> pool; size_t next = 0; } g_tl;
hipEvent_t tl_event() {
    if (g_tl.next == g_tl.pool.size()) { hipEvent_t e; HIP_CHECK(hipEventCreate(&e)); g_tl.pool.push_back(e); }
    return g_tl.pool[g_tl.next++];
}
}
bool fq_tl_collecting() { return g_tl.on; }
fq_tl_scope::fq_tl_scope(hipStream_t st_, const char * name) : st(st_), slot(-1) {
    if (!g_tl.on) return;
    if (g_tl.depth++ > 0) return;                                  // (a launch site inside another: the outer one carries the bracket)
    slot = (int) g_tl.recs.size();
    g_tl.recs.push_back(tl_rec{ name, tl_event(), tl_event() });
    HIP_CHECK(hipEventRecord(g_tl.recs[(size_t) slot].e0, st));
}
fq_tl_scope::~fq_tl_scope() {
    if (!g_tl.on) return;
    --g_tl.depth;
    if (slot >= 0) HIP_CHECK(hipEventRecord(g_tl.recs[(size_t) slot].e1, st));
}
void fq_tl_begin() { g_tl.recs.clear(); g_tl.next = 0; g_tl.depth = 0; g_tl.on = true; }
int fq_tl_end(FILE * out, const char * title) {
    g_tl.on = false;
    HIP_CHECK(hipDeviceSynchronize());
    struct agg { const char * name; int n; double us; };
    std::vector<agg> a;
    double total = 0.0;
    for (const tl_rec & r : g_tl.recs) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) { (void) hipGetLastError(); continue; }
        size_t i = 0;
        for (; i < a.size(); ++i) if (!strcmp(a[i].name, r.name)) break;
        if (i == a.size()) a.push_back(agg{ r.name, 0, 0.0 });
        a[i].n += 1; a[i].us += 1e3 * (double) ms; total += 1e3 * (double) ms;
    }
    if (out) {
        fprintf(out, "%s\n", title ? title : "falcon-hip: launches");
        fprintf(out, "  %-28s %8s %12s %10s %7s\n", "launch site", "calls", "total us", "avg us", "share");
        for (const agg & g : a) fprintf(out, "  %-28s %8d %12.1f %10.2f %6.1f%%\n", g.name, g.n, g.us, g.us / g.n, total > 0 ? 100.0 * g.us / total : 0.0);
        fprintf(out, "  %-28s %8zu %12.1f   (event brackets around each launch site on its stream; sites on the second stream overlap the first)\n", "all", g_tl.recs.size(), total);
    }
    return (int) g_tl.recs.size();
}

// ------------------------------------------------------------------------------------------------ block ops
extern "C" void ggml_hip_layer_norm(const float * x, int64_t n, int64_t rows, const float * w, const float * b, float * y) {
    fq_launch_layer_norm(x, n, rows, w, b, y, fq_ctx().stream);
}
extern "C" void ggml_hip_gelu(const float * x, float * y, int64_t n) { fq_launch_gelu(x, y, n, fq_ctx().gelu_table, fq_ctx().stream); }
extern "C" void ggml_hip_add3(const float * a, const float * b, const float * c, float * y, int64_t n) { fq_launch_add3(a, b, c, y, n, fq_ctx().stream); }

// host-side table: theta advanced by repeated f32 multiplication, cosf/sinf of the host libm (ggml.c:12962-12966)
std::vector<float> fq_rope_table_host(int head_dim, int n_pos, int rope_n_ctx) {
    float alpha = 1.0f;                                                      // ggml.c:12880-12887, DYNAMIC_MODE=1, NTK_ALPHA=2
    if (rope_n_ctx >= 2048) alpha = powf((float)(((rope_n_ctx / 2048) - 1) * 2.0f + 1), (float)(head_dim / (head_dim - 2.0)));
    const float theta_scale = powf(alpha * 10000.0f, -2.0f / (float) head_dim);    // ggml.c:12898
    const int half = head_dim / 2;
    std::vector<float> cs((size_t) n_pos * half * 2);
    for (int p = 0; p < n_pos; ++p) {
        float theta = (float) p;
        for (int k = 0; k < half; ++k) {
            cs[((size_t) p * half + k) * 2 + 0] = cosf(theta);
            cs[((size_t) p * half + k) * 2 + 1] = sinf(theta);
            theta *= theta_scale;
        }
    }
    return cs;
}
extern "C" float * ggml_hip_rope_table_create(int head_dim, int n_pos, int rope_n_ctx) {
    std::vector<float> cs = fq_rope_table_host(head_dim, n_pos, rope_n_ctx);
    float * dev = (float *) ggml_hip_malloc(cs.size() * 4);
    ggml_hip_memcpy_h2d(dev, cs.data(), cs.size() * 4);
    return dev;
}
// the kernels read n_past from device memory (so a captured graph can be replayed for every token)
static const int * upload_n_past(int n_past) {
    hip_context & c = fq_ctx();
    HIP_CHECK(hipMemcpyAsync(c.scalar_i32, &n_past, sizeof(int), hipMemcpyHostToDevice, c.stream));
    HIP_CHECK(hipStreamSynchronize(c.stream));       // n_past lives on the caller's stack
    return c.scalar_i32;
}
extern "C" void ggml_hip_rope_kv_store(float * qkv, int N, int H, int HKV, int D, int n_past, const float * rope_table, float * kc, float * vc) {
    fq_launch_rope_kv(qkv, N, H, HKV, D, upload_n_past(n_past), rope_table, kc, vc, fq_ctx().stream);
}
extern "C" void ggml_hip_attention(const float * qkv, int N, int H, int HKV, int D, int n_past, const float * kc, const float * vc, float * att) {
    fq_launch_attention(qkv, N, H, HKV, D, upload_n_past(n_past), n_past + N, kc, vc, fq_ctx().exp_table_attn, att, fq_ctx().stream);
}
