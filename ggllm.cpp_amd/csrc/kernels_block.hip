// kernels_block.hip -- the non-mat-mul ops of a Falcon decoder block, device resident (gfx950).
//
// In the reference these always run on the CPU, even in its CUDA build (libfalcon.cpp:2309, 2356 set
// cuda_op_directive = 0; ggml-cuda.cu:2433 supports rope mode 0 only). Arithmetic follows the CPU ops so that
// logits match the CPU reference:
//   layer norm  ggml.c:10540-10594  (f64 sums, eps 1e-5) then * weight + bias (libfalcon.cpp:2166-2188)
//   gelu        ggml.c:3477-3484    fp16 table lookup (table built on the host exactly like ggml.c:4276-4290)
//   rope        ggml.c:12957-12978  NeoX pairing (i, i+32); cos/sin come from a host-built table whose theta is
//                                   advanced by repeated f32 multiplication like the reference loop
//   attention   K.Q (ggml.c:11049-11088, GQA broadcast i02 = i12/(H/HKV)), scale, causal mask (ggml.c:12300-12351),
//               soft_max with the fp16 exp table and an f64 sum (ggml.c:12389-12456), V.P
// Built with -ffp-contract=off.
#include "fq_block_dev.h"
#include "fq_attn_dev.h"
#include "kernels.h"
#include "hip_context.h"
#include <vector>

// ------------------------------------------------------------------------------------------------ layer norm
// one 256-thread workgroup per row; the row lives in LDS between the passes
__global__ void __launch_bounds__(256) k_layer_norm(const float * __restrict__ x, int64_t n, const float * __restrict__ w,
                                                    const float * __restrict__ b, float * __restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    float  * row = (float *) smem;
    double * red = (double *)(smem + ((n * 4 + 15) & ~(int64_t) 15));
    layer_norm_row_block(x + (int64_t) blockIdx.x * n, n, w, b, row, red);
    float * yr = y + (int64_t) blockIdx.x * n;
    for (int64_t i = threadIdx.x; i < (n >> 2); i += blockDim.x) ((float4 *) yr)[i] = ((const float4 *) row)[i];
}

// layer norm + the mat-mul's activation quantizer in one launch (prefill): the normalised row goes from LDS straight into
// the Q8 image; y (optional) also receives the f32 row. Same device functions as k_layer_norm + k_quantize_q8*.
template <int ACT>
__global__ void __launch_bounds__(256) k_layer_norm_quant(const float * __restrict__ x, int64_t n, const float * __restrict__ w,
                                                          const float * __restrict__ b, float * __restrict__ y, fq_act a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    float  * row = (float *) smem;
    double * red = (double *)(smem + ((n * 4 + 15) & ~(int64_t) 15));
    layer_norm_row_block(x + (int64_t) blockIdx.x * n, n, w, b, row, red);
    if (y) {
        float * yr = y + (int64_t) blockIdx.x * n;
        for (int64_t i = threadIdx.x; i < (n >> 2); i += blockDim.x) ((float4 *) yr)[i] = ((const float4 *) row)[i];
    }
    __syncthreads();
    uint8_t * col = a.base + (size_t) blockIdx.x * fq_act_col_bytes(ACT, n);
    quantize_row_block<ACT>(row, n, act_image_at(col, ACT, n));
}
void fq_launch_layer_norm_quant(const float * x, int64_t n, int64_t rows, const float * w, const float * b, float * y, const fq_act & a, hipStream_t st) {
    FQ_TL(st, "layer_norm_quant");
    const size_t lds = ((n * 4 + 15) & ~(size_t) 15) + 64;
    if (a.type == FQ_Q8_0)      hipLaunchKernelGGL(k_layer_norm_quant<FQ_Q8_0>, dim3((unsigned) rows), dim3(256), lds, st, x, n, w, b, y, a);
    else if (a.type == FQ_Q8_1) hipLaunchKernelGGL(k_layer_norm_quant<FQ_Q8_1>, dim3((unsigned) rows), dim3(256), lds, st, x, n, w, b, y, a);
    else                        hipLaunchKernelGGL(k_layer_norm_quant<FQ_Q8_K>, dim3((unsigned) rows), dim3(256), lds, st, x, n, w, b, y, a);
}

// x = (a + b) + x: the residual sum of a block (libfalcon.cpp:2399-2400: inpL = (ffn_out + attn_out) + inpL) as a launch of its own -- the short prompts' two-stream form,
// where neither mat-mul should wait for the other just to carry the sum in its epilogue (falcon_hip.hip)
__global__ void __launch_bounds__(256) k_add2_inplace(float * __restrict__ x, const float * __restrict__ a, const float * __restrict__ b, int64_t nv) {
    for (int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (int64_t) gridDim.x * blockDim.x) {
        const float4 p = ((const float4 *) a)[i], q = ((const float4 *) b)[i]; float4 v = ((float4 *) x)[i];
        v.x = (p.x + q.x) + v.x; v.y = (p.y + q.y) + v.y; v.z = (p.z + q.z) + v.z; v.w = (p.w + q.w) + v.w;
        ((float4 *) x)[i] = v;
    }
}
void fq_launch_add2_inplace(float * x, const float * a, const float * b, int64_t n, hipStream_t st) {
    FQ_TL(st, "add2");
    const int64_t nv = n >> 2;
    const unsigned grid = (unsigned)((nv + 255) / 256 < 2048 ? (nv + 255) / 256 : 2048);
    hipLaunchKernelGGL(k_add2_inplace, dim3(grid ? grid : 1), dim3(256), 0, st, x, a, b, nv);
}

// ... and, one workgroup per token, with the LayerNorm(s) + activation image(s) that follow (fq_next_norm): the sum, then k_layer_norm_quant's own device code on the row it left
// in LDS (per thread the same elements in the same order: the same f64 sums, the same bits); a second norm of a two-norm block repeats only * w + b
template <int ACT>
__global__ void __launch_bounds__(256) k_add2_ln(float * __restrict__ x, const float * __restrict__ a, const float * __restrict__ b, int64_t n, fq_next_norm nn) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const size_t rowb = ((size_t) n * 4 + 15) & ~(size_t) 15;
    float  * row = (float *) smem;
    float  * out = nn.w1 ? (float *)(smem + rowb) : row;
    double * red = (double *)(smem + (nn.w1 ? 2 : 1) * rowb);
    const int tid = threadIdx.x;
    const int64_t r0 = (int64_t) blockIdx.x * n, nv = n >> 2;
    double s = 0.0;
    for (int64_t i = tid; i < nv; i += 256) {
        const float4 p = ((const float4 *)(a + r0))[i], q = ((const float4 *)(b + r0))[i]; float4 v = ((float4 *)(x + r0))[i];
        v.x = (p.x + q.x) + v.x; v.y = (p.y + q.y) + v.y; v.z = (p.z + q.z) + v.z; v.w = (p.w + q.w) + v.w;
        ((float4 *)(x + r0))[i] = v; ((float4 *) row)[i] = v;
        s += (double) v.x; s += (double) v.y; s += (double) v.z; s += (double) v.w;
    }
    layer_norm_from_row(s, n, nn.w0, nn.b0, row, out, red);
    quantize_row_block<ACT>(out, n, act_image_at(nn.a0.base + (size_t) blockIdx.x * fq_act_col_bytes(ACT, n), ACT, n));
    if (nn.w1) {
        __syncthreads();
        layer_norm_apply(n, nn.w1, nn.b1, row, out);
        quantize_row_block<ACT>(out, n, act_image_at(nn.a1.base + (size_t) blockIdx.x * fq_act_col_bytes(ACT, n), ACT, n));
    }
}
bool fq_add2_ln_ok(const fq_next_norm & nn, int64_t E) {
    static const bool on = !(getenv("FALCON_HIP_ADD2_LN") && atoi(getenv("FALCON_HIP_ADD2_LN")) == 0);
    if (!on || !nn.w0 || E % 4 || nn.a0.K != E || (nn.w1 && (nn.a1.type != nn.a0.type || nn.a1.K != E))) return false;
    if (nn.a0.type != FQ_Q8_0 && nn.a0.type != FQ_Q8_1 && nn.a0.type != FQ_Q8_K) return false;
    if (nn.a0.type == FQ_Q8_K ? E % 256 : E % 32) return false;
    return (nn.w1 ? 2 : 1) * (((size_t) E * 4 + 15) & ~(size_t) 15) + 64 <= 160 * 1024;
}
void fq_launch_add2_ln(float * x, const float * a, const float * b, int64_t E, int64_t rows, const fq_next_norm & nn, hipStream_t st) {
    FQ_TL(st, "add2_ln");
    const size_t lds = (nn.w1 ? 2 : 1) * (((size_t) E * 4 + 15) & ~(size_t) 15) + 64;
#define FQ_A2LN(A) { static size_t gmax = 64 * 1024; if (lds > gmax) { HIP_CHECK(hipFuncSetAttribute((const void *) k_add2_ln<A>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds)); gmax = lds; } \
        hipLaunchKernelGGL(k_add2_ln<A>, dim3((unsigned) rows), dim3(256), lds, st, x, a, b, E, nn); }
    if (nn.a0.type == FQ_Q8_0) FQ_A2LN(FQ_Q8_0) else if (nn.a0.type == FQ_Q8_1) FQ_A2LN(FQ_Q8_1) else FQ_A2LN(FQ_Q8_K)
#undef FQ_A2LN
}

// two norms of the same rows in one launch (Falcon-40B's ln_mlp and ln_attn: blockIdx.y picks the weights and the image)
template <int ACT>
__global__ void __launch_bounds__(256) k_layer_norm_quant2(const float * __restrict__ x, int64_t n, const float * __restrict__ w0, const float * __restrict__ b0, fq_act a0,
                                                           const float * __restrict__ w1, const float * __restrict__ b1, fq_act a1) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    float  * row = (float *) smem;
    double * red = (double *)(smem + ((n * 4 + 15) & ~(int64_t) 15));
    const bool second = blockIdx.y != 0;
    layer_norm_row_block(x + (int64_t) blockIdx.x * n, n, second ? w1 : w0, second ? b1 : b0, row, red);
    __syncthreads();
    uint8_t * col = (second ? a1.base : a0.base) + (size_t) blockIdx.x * fq_act_col_bytes(ACT, n);
    quantize_row_block<ACT>(row, n, act_image_at(col, ACT, n));
}
bool fq_launch_layer_norm_quant2(const float * x, int64_t n, int64_t rows, const float * w0, const float * b0, const fq_act & a0,
                                 const float * w1, const float * b1, const fq_act & a1, hipStream_t st) {
    FQ_TL(st, "layer_norm_quant2");
    if (a0.type != a1.type) return false;
    const size_t lds = ((n * 4 + 15) & ~(size_t) 15) + 64;
    const dim3 grid((unsigned) rows, 2);
    if (a0.type == FQ_Q8_0)      hipLaunchKernelGGL(k_layer_norm_quant2<FQ_Q8_0>, grid, dim3(256), lds, st, x, n, w0, b0, a0, w1, b1, a1);
    else if (a0.type == FQ_Q8_1) hipLaunchKernelGGL(k_layer_norm_quant2<FQ_Q8_1>, grid, dim3(256), lds, st, x, n, w0, b0, a0, w1, b1, a1);
    else                         hipLaunchKernelGGL(k_layer_norm_quant2<FQ_Q8_K>, grid, dim3(256), lds, st, x, n, w0, b0, a0, w1, b1, a1);
    return true;
}

void fq_launch_layer_norm(const float * x, int64_t n, int64_t rows, const float * w, const float * b, float * y, hipStream_t st) {
    FQ_TL(st, "layer_norm");
    const size_t lds = ((n * 4 + 15) & ~(size_t) 15) + 64;
    hipLaunchKernelGGL(k_layer_norm, dim3((unsigned) rows), dim3(256), lds, st, x, n, w, b, y);
}

// ------------------------------------------------------------------------------------------------ gelu / add
__global__ void k_gelu(const float * __restrict__ x, float * __restrict__ y, int64_t n, const uint16_t * __restrict__ tab) {
    for (int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t) gridDim.x * blockDim.x)
        y[i] = h2f_bits(tab[f2h_bits(x[i])]);
}
void fq_launch_gelu(const float * x, float * y, int64_t n, const uint16_t * gelu_table, hipStream_t st) {
    FQ_TL(st, "gelu");
    const int blocks = (int) ((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    hipLaunchKernelGGL(k_gelu, dim3(blocks), dim3(256), 0, st, x, y, n, gelu_table);
}
__global__ void k_add3(const float * __restrict__ a, const float * __restrict__ b, const float * __restrict__ c, float * __restrict__ y, int64_t n) {
    for (int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t) gridDim.x * blockDim.x)
        y[i] = (a[i] + b[i]) + c[i];
}
void fq_launch_add3(const float * a, const float * b, const float * c, float * y, int64_t n, hipStream_t st) {
    FQ_TL(st, "add3");
    const int blocks = (int) ((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    hipLaunchKernelGGL(k_add3, dim3(blocks), dim3(256), 0, st, a, b, c, y, n);
}

// ------------------------------------------------------------------------------------------------ rope + KV append
// thread = one rotation pair (or one V element pair). Heads 0..H-1 are Q (rotated in place), H..H+HKV-1 are K
// (rotated into the cache), H+HKV.. are V (copied into the cache). Caches: [n_ctx][HKV][D] f32 for this layer.
// seq_stride > 0: the N rows are N independent sequences at the SAME position n_past, row t owning the cache at
// kc / vc + t * seq_stride (lock-step decode streams of one pipeline stage step)
__global__ void k_rope_kv(float * __restrict__ qkv, int N, int H, int HKV, int D, const int * __restrict__ n_past_ptr, const float * __restrict__ cs,
                          float * __restrict__ kc, float * __restrict__ vc, int64_t seq_stride) {
    const int half = D >> 1;
    const int heads = H + 2 * HKV;
    const int n_past = *n_past_ptr;          // device scalar: one captured hipGraph serves every decode step
    const int64_t total = (int64_t) N * heads * half;
    for (int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t) gridDim.x * blockDim.x) {
        const int k = (int)(i % half);
        const int h = (int)((i / half) % heads);
        const int t = (int)(i / ((int64_t) half * heads));
        float * v = qkv + ((int64_t) t * heads + h) * D;
        const int pos = seq_stride ? n_past : n_past + t;
        float * kcs = kc + (int64_t) t * seq_stride, * vcs = vc + (int64_t) t * seq_stride;
        if (h < H + HKV) {
            const float c = cs[((int64_t) pos * half + k) * 2], s = cs[((int64_t) pos * half + k) * 2 + 1];
            const float x0 = v[k], x1 = v[k + half];
            const float r0 = x0 * c - x1 * s, r1 = x0 * s + x1 * c;     // ggml.c:12974-12975
            if (h < H) { v[k] = r0; v[k + half] = r1; }
            else { float * o = kcs + ((int64_t) pos * HKV + (h - H)) * D; o[k] = r0; o[k + half] = r1; }
        } else {
            float * o = vcs + ((int64_t) pos * HKV + (h - H - HKV)) * D;
            o[k] = v[k]; o[k + half] = v[k + half];
        }
    }
}
void fq_launch_rope_kv(float * qkv, int N, int H, int HKV, int D, const int * n_past_dev, const float * rope_cs, float * k_cache, float * v_cache, hipStream_t st,
                       int64_t seq_stride) {
    FQ_TL(st, "rope_kv");
    const int64_t total = (int64_t) N * (H + 2 * HKV) * (D / 2);
    const int blocks = (int) ((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    hipLaunchKernelGGL(k_rope_kv, dim3(blocks), dim3(256), 0, st, qkv, N, H, HKV, D, n_past_dev, rope_cs, k_cache, v_cache, seq_stride);
}

// ------------------------------------------------------------------------------------------------ attention
// One 256-thread workgroup per (head, token); n_kv = n_past + t + 1 keys are visible, all read from the KV cache
// (k_rope_kv has already appended this launch's keys). The arithmetic lives in fq_attn_dev.h.
template <bool F64>
__global__ void __launch_bounds__(256) k_attention(const float * __restrict__ qkv, int H, int HKV, const int * __restrict__ n_past_ptr,
                                                   const float * __restrict__ kc, const float * __restrict__ vc,
                                                   const uint16_t * __restrict__ exp_tab, float * __restrict__ att, int64_t seq_stride) {
    constexpr int D = 64;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int h = blockIdx.x, t = blockIdx.y;
    const int n_kv = *n_past_ptr + (seq_stride ? 0 : t) + 1;
    kc += (int64_t) t * seq_stride; vc += (int64_t) t * seq_stride;
    const int heads = H + 2 * HKV;
    const int hk = h / (H / HKV);
    const attn_lds L = attn_lds_carve(smem);
    const float o = attn_head_block<F64>(qkv + ((int64_t) t * heads + h) * D, kc, vc, HKV, hk, n_kv, nullptr, nullptr, exp_tab, L);
    if (threadIdx.x < 64) att[(int64_t) t * H * D + (int64_t) h * D + threadIdx.x] = o;
}

// R consecutive tokens of a head per workgroup (prefill): key / value tiles are loaded once per R tokens (fq_attn_dev.h)
template <int R, bool F64>
__global__ void __launch_bounds__(256) k_attention_rows(const float * __restrict__ qkv, int N, int H, int HKV, const int * __restrict__ n_past_ptr,
                                                        const float * __restrict__ kc, const float * __restrict__ vc,
                                                        const uint16_t * __restrict__ exp_tab, float * __restrict__ att, int p_stride,
                                                        float * __restrict__ p_scratch) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int h = blockIdx.x, t0 = blockIdx.y * R;
    const int nrows = N - t0 < R ? N - t0 : R;
    float * redf = (float *) smem;
    double * red = (double *)(smem + 64);
    // the R score rows: LDS, or (long contexts) this workgroup's own slice of a global scratch buffer -- written and read
    // back by the same workgroup within microseconds (L2 / Infinity Cache), and the LDS they would take no longer limits
    // the number of workgroups per CU
    float * p = p_scratch ? p_scratch + ((size_t) blockIdx.y * gridDim.x + blockIdx.x) * (size_t) R * (size_t) p_stride
                          : (float *)(smem + 64 + 16 * 64 * 8);
    attn_rows_block<R, F64>(qkv, H + 2 * HKV, h, t0, nrows, *n_past_ptr, kc, vc, HKV, h / (H / HKV), exp_tab, redf, red, p, p_stride, att, H);
}
template <int R, bool F64>
static void launch_attention_rows_t(const float * qkv, int N, int H, int HKV, const int * n_past_dev, int p_stride, size_t lds, const float * k_cache,
                                    const float * v_cache, const uint16_t * exp_table, float * att, float * p_scratch, hipStream_t st) {
    if (lds > 64 * 1024) { static size_t g = 0; if (lds > g) { HIP_CHECK(hipFuncSetAttribute((const void *) k_attention_rows<R, F64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds)); g = lds; } }
    hipLaunchKernelGGL((k_attention_rows<R, F64>), dim3((unsigned) H, (unsigned)((N + R - 1) / R)), dim3(256), lds, st, qkv, N, H, HKV, n_past_dev, k_cache, v_cache, exp_table, att, p_stride, p_scratch);
}
// ---- prefill attention on the f32 matrix pipe (N >= 32) ----------------------------------------------------------------------
// v_mfma_f32_32x32x2_f32 performs, per output element, ONE fused multiply-add per k step in ascending k
// (scripts/microbench/mb_mfma_f32.hip: 0 of 1024 outputs differ from the fmaf chain over 64 steps) -- the arithmetic of the
// reference's SIMD builds (f32 FMA accumulation, ggml.c:2270-2294) with a SEQUENTIAL association the oracle restates in two
// lines (oracle_falcon.c dot_qk_mfma / dot_pv_mfma):
//   score(i, j) = chain over s = 0..31 of  q[s] k[s],  q[32 + s] k[32 + s]            (lanes 0-31 feed dims 0..31, lanes 32-63 dims 32..63)
//   out(i, d)   = E + O,  E / O = chains over the even / odd 32-key tiles, in a tile over s = 0..15 of  p[s] v[s],  p[16 + s] v[16 + s]
// One workgroup = one head x 32 query tokens, 4 waves: the waves split the key tiles of K.Q (one 32 x 32 score tile = 32
// MFMAs), the soft_max rows (max, fp16-table exp, f64 sum -- exact in any order: <= 2^13 fp16-valued terms --, scaling), and
// V.P as (dim half) x (tile parity). Scores / probabilities live in the workgroup's slice of the global scratch (L2).
typedef float v16f __attribute__((ext_vector_type(16)));
__global__ void __launch_bounds__(256) k_attention_mfma(const float * __restrict__ qkv, int N, int H, int HKV, const int * __restrict__ n_past_ptr,
                                                        const float * __restrict__ kc, const float * __restrict__ vc,
                                                        const uint16_t * __restrict__ exp_tab, float * __restrict__ att, int p_stride,
                                                        float * __restrict__ p_scratch) {
    __shared__ float xch[2][16][64];
    __shared__ float rmax[4][32], rinv[32];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, li = lane & 31, hf = lane >> 5;
    const int h = blockIdx.x, i0 = blockIdx.y * 32, hk = h / (H / HKV), heads = H + 2 * HKV;
    const int n_past = *n_past_ptr;
    const int nrows = N - i0 < 32 ? N - i0 : 32;
    const int n_kv_max = n_past + i0 + nrows;                       // keys the tile's last token sees
    const int n_rows_cache = n_past + N;                            // key / value rows that exist
    const int ntile = (n_kv_max + 31) >> 5;
    float * p = p_scratch + ((size_t) blockIdx.y * gridDim.x + blockIdx.x) * (size_t) 32 * (size_t) p_stride;
    // The score matrix makes FOUR trips through the scratch (written here, read + rewritten as exp() by the soft_max, read by V.P):
    // the row maxima are taken from the MFMA results in registers, and the 1 / sum scaling is applied to the operand V.P loads
    // (p = e * inv, the soft_max's own rounding) -- the scratch of a 2048-token prompt is 1.2 GB and does not stay in the caches.
    // ---- scores
    {
        f32x4 q8[8];
        const float * qrow = qkv + ((int64_t)(i0 + (li < nrows ? li : nrows - 1)) * heads + h) * 64 + 32 * hf;
#pragma unroll
        for (int v = 0; v < 8; ++v) q8[v] = ((const f32x4 *) qrow)[v];
        auto load_k = [&](int T, f32x4 (&k8)[8]) {
            const int j = 32 * T + li;
            const float * krow = kc + ((int64_t)(j < n_rows_cache ? j : n_rows_cache - 1) * HKV + hk) * 64 + 32 * hf;
#pragma unroll
            for (int v = 0; v < 8; ++v) k8[v] = ((const f32x4 *) krow)[v];
        };
        f32x4 ka[8], kb[8];
        float mx[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) mx[r] = -INFINITY;
        if (wid < ntile) load_k(wid, ka);
        for (int T = wid; T < ntile; T += 4) {
            if (T + 4 < ntile) load_k(T + 4, kb);
            v16f c = {0};
#pragma unroll
            for (int v = 0; v < 8; ++v) {
                c = __builtin_amdgcn_mfma_f32_32x32x2f32(q8[v].x, ka[v].x, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x2f32(q8[v].y, ka[v].y, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x2f32(q8[v].z, ka[v].z, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x2f32(q8[v].w, ka[v].w, c, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ir = (r & 3) + 8 * (r >> 2) + 4 * hf;
                const float sc = c[r] * 0.125f;
                p[(size_t) ir * p_stride + 32 * T + li] = sc;
                mx[r] = fq_max_f32(mx[r], (ir < nrows && 32 * T + li <= n_past + i0 + ir) ? sc : -INFINITY);
            }
#pragma unroll
            for (int v = 0; v < 8; ++v) ka[v] = kb[v];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float m = reduce32(mx[r], op_max());
            if (li == 0) rmax[wid][(r & 3) + 8 * (r >> 2) + 4 * hf] = m;
        }
    }
    __syncthreads();
    // ---- soft_max: wave w takes rows w, w + 4, ...: exp() of the visible keys, zeros up to the last tile, 1 / sum kept aside
    for (int ir = wid; ir < 32; ir += 4) {
        float * pr = p + (size_t) ir * p_stride;
        const int n_kv = ir < nrows ? n_past + i0 + ir + 1 : 0;
        const float m = fq_max_f32(fq_max_f32(rmax[0][ir], rmax[1][ir]), fq_max_f32(rmax[2][ir], rmax[3][ir]));
        double lsum = 0.0;
#pragma unroll 4
        for (int j = 4 * lane; j < 32 * ntile; j += 256) {
            const f32x4 x = *(const f32x4 *)(pr + j);
            f32x4 e;
            e.x = j     < n_kv ? soft_max_exp(exp_tab, x.x - m) : 0.0f;
            e.y = j + 1 < n_kv ? soft_max_exp(exp_tab, x.y - m) : 0.0f;
            e.z = j + 2 < n_kv ? soft_max_exp(exp_tab, x.z - m) : 0.0f;
            e.w = j + 3 < n_kv ? soft_max_exp(exp_tab, x.w - m) : 0.0f;
            *(f32x4 *)(pr + j) = e;
            lsum += (double) e.x; lsum += (double) e.y; lsum += (double) e.z; lsum += (double) e.w;     // (exact in any order)
        }
        lsum = wave_sum(lsum);
        if (lane == 0) rinv[ir] = (float)(1.0 / lsum);
    }
    __syncthreads();
    // ---- V.P: wave = (dim half, tile parity)
    {
        const int dh = wid & 1, par = wid >> 1;
        const float * prow = p + (size_t) li * p_stride + 16 * hf;
        const float inv = rinv[li];
        auto load_pv = [&](int T, f32x4 (&p4)[4], float (&v16)[16]) {
#pragma unroll
            for (int v = 0; v < 4; ++v) p4[v] = ((const f32x4 *)(prow + 32 * T))[v];
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int j = 32 * T + 16 * hf + s;
                v16[s] = vc[((int64_t)(j < n_rows_cache ? j : n_rows_cache - 1) * HKV + hk) * 64 + 32 * dh + li];
            }
        };
        f32x4 pa[4], pb[4]; float va[16], vb[16];
        if (par < ntile) load_pv(par, pa, va);
        v16f c = {0};
        for (int T = par; T < ntile; T += 2) {
            if (T + 2 < ntile) load_pv(T + 2, pb, vb);
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                c = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[v].x * inv, va[4 * v + 0], c, 0, 0, 0);      // p = e * inv, the rounding of the soft_max's own scaling step (ggml_vec_scale_f32)
                c = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[v].y * inv, va[4 * v + 1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[v].z * inv, va[4 * v + 2], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[v].w * inv, va[4 * v + 3], c, 0, 0, 0);
            }
#pragma unroll
            for (int v = 0; v < 4; ++v) pa[v] = pb[v];
#pragma unroll
            for (int s = 0; s < 16; ++s) va[s] = vb[s];
        }
        if (par == 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) xch[dh][r][lane] = c[r];
        }
        __syncthreads();
        if (par == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ir = (r & 3) + 8 * (r >> 2) + 4 * hf;
                if (ir < nrows) att[(int64_t)(i0 + ir) * H * 64 + (int64_t) h * 64 + 32 * dh + li] = c[r] + xch[dh][r][lane];
            }
        }
    }
}

// ---- the same arithmetic with the score rows in LDS: one head x 16 query tokens per workgroup, v_mfma_f32_16x16x4_f32 ------------
// v_mfma_f32_16x16x4_f32 is the same fused chain, four k steps per instruction in ascending k (scripts/microbench/mb_mfma_f32_16.hip:
// 0 of 2048 outputs differ), so the chains above are kept term for term by feeding k = 0..3 of instruction m with
//   K.Q : dims  2m, 32 + 2m, 2m + 1, 33 + 2m     (m = 0..15)        V.P : keys  2m, 16 + 2m, 2m + 1, 17 + 2m  of a 32-key tile (m = 0..7)
// i.e. lane group kq = lane >> 4 holds the even (kq < 2) or odd elements of the low (kq even) or high (kq odd) half. Results are
// bit-identical to k_attention_mfma (and to the oracle's dot_qk_mfma / dot_pv_mfma). 16 rows x n_kv floats fit the LDS up to ~2400
// keys (128 KiB at 2048): the score matrix never leaves the CU -- the 32-row form moves 2.5 GB of HBM traffic per 2048-token
// launch for it. 8 waves: K.Q by 16-key sub-tiles, soft_max two rows per wave, V.P as (16-dim tile) x (32-key tile parity).
typedef float v4f_ __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(512) k_attention_mfma16(const float * __restrict__ qkv, int N, int H, int HKV, const int * __restrict__ n_past_ptr,
                                                          const float * __restrict__ kc, const float * __restrict__ vc,
                                                          const uint16_t * __restrict__ exp_tab, float * __restrict__ att, int ps) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    float * p = (float *) smem;                                     // [16][ps], ps = 32 ntile_max + 4
    float * rmax = p + 16 * ps;                                     // [8 waves][16 rows]
    float * rinv = rmax + 8 * 16;                                   // [16]
    float * xch  = rinv + 16;                                       // [4 dim tiles][4][64]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l16 = lane & 15, kq = lane >> 4;
    const int h = blockIdx.x, i0 = blockIdx.y * 16, hk = h / (H / HKV), heads = H + 2 * HKV;
    const int n_past = *n_past_ptr;
    const int nrows = N - i0 < 16 ? N - i0 : 16;
    const int n_kv_max = n_past + i0 + nrows;                       // keys the tile's last token sees
    const int n_rows_cache = n_past + N;                            // key / value rows that exist
    const int ntile = (n_kv_max + 31) >> 5, nsub = 2 * ntile;
    const int eo = kq >> 1, hi = kq & 1;                            // the lane group's elements: odd (1) / even (0) ones of the high (1) / low (0) half
    // ---- scores
    {
        float q16[16];
        {
            const float * qrow = qkv + ((int64_t)(i0 + (l16 < nrows ? l16 : nrows - 1)) * heads + h) * 64 + 32 * hi;
#pragma unroll
            for (int v = 0; v < 8; ++v) { const f32x4 t = ((const f32x4 *) qrow)[v]; q16[2 * v] = eo ? t.y : t.x; q16[2 * v + 1] = eo ? t.w : t.z; }
        }
        auto load_k = [&](int U, float (&k16)[16]) {
            const int j = 16 * U + l16;
            const float * krow = kc + ((int64_t)(j < n_rows_cache ? j : n_rows_cache - 1) * HKV + hk) * 64 + 32 * hi;
#pragma unroll
            for (int v = 0; v < 8; ++v) { const f32x4 t = ((const f32x4 *) krow)[v]; k16[2 * v] = eo ? t.y : t.x; k16[2 * v + 1] = eo ? t.w : t.z; }
        };
        float ka[16], kb[16];
        float mx[4] = { -INFINITY, -INFINITY, -INFINITY, -INFINITY };
        if (wid < nsub) load_k(wid, ka);
        for (int U = wid; U < nsub; U += 8) {
            if (U + 8 < nsub) load_k(U + 8, kb);
            v4f_ c = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int m = 0; m < 16; ++m) c = __builtin_amdgcn_mfma_f32_16x16x4f32(q16[m], ka[m], c, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ir = 4 * kq + r, j = 16 * U + l16;
                const float sc = c[r] * 0.125f;
                p[ir * ps + j] = sc;
                mx[r] = fq_max_f32(mx[r], (ir < nrows && j <= n_past + i0 + ir) ? sc : -INFINITY);
            }
#pragma unroll
            for (int v = 0; v < 16; ++v) ka[v] = kb[v];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float m = reduce16(mx[r], op_max());
            if (l16 == 0) rmax[wid * 16 + 4 * kq + r] = m;
        }
    }
    __syncthreads();
    // ---- soft_max: wave w takes rows w and w + 8: exp() of the visible keys, zeros up to the last tile, 1 / sum kept aside
    for (int ir = wid; ir < 16; ir += 8) {
        float * pr = p + ir * ps;
        const int n_kv = ir < nrows ? n_past + i0 + ir + 1 : 0;
        float m = rmax[ir];
#pragma unroll
        for (int w2 = 1; w2 < 8; ++w2) m = fq_max_f32(m, rmax[w2 * 16 + ir]);
        double lsum = 0.0;
        for (int j = 4 * lane; j < 32 * ntile; j += 256) {
            const f32x4 x = *(const f32x4 *)(pr + j);
            f32x4 e;
            e.x = j     < n_kv ? soft_max_exp(exp_tab, x.x - m) : 0.0f;
            e.y = j + 1 < n_kv ? soft_max_exp(exp_tab, x.y - m) : 0.0f;
            e.z = j + 2 < n_kv ? soft_max_exp(exp_tab, x.z - m) : 0.0f;
            e.w = j + 3 < n_kv ? soft_max_exp(exp_tab, x.w - m) : 0.0f;
            *(f32x4 *)(pr + j) = e;
            lsum += (double) e.x; lsum += (double) e.y; lsum += (double) e.z; lsum += (double) e.w;     // (exact in any order)
        }
        lsum = wave_sum(lsum);
        if (lane == 0) rinv[ir] = (float)(1.0 / lsum);
    }
    __syncthreads();
    // ---- V.P: wave = (16-dim tile dt, tile parity par)
    {
        const int dt = wid & 3, par = wid >> 2;
        const float * prow = p + l16 * ps + 16 * hi;                // keys 16 hi .. 16 hi + 15 of a tile: the even or the odd ones
        const float inv = rinv[l16];
        auto load_pv = [&](int T, float (&p8)[8], float (&v8)[8]) {
#pragma unroll
            for (int v = 0; v < 4; ++v) { const f32x4 t = ((const f32x4 *)(prow + 32 * T))[v]; p8[2 * v] = eo ? t.y : t.x; p8[2 * v + 1] = eo ? t.w : t.z; }
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const int j = 32 * T + 16 * hi + 2 * m + eo;
                v8[m] = vc[((int64_t)(j < n_rows_cache ? j : n_rows_cache - 1) * HKV + hk) * 64 + 16 * dt + l16];
            }
        };
        float pa[8], pb[8], va[8], vb[8];
        if (par < ntile) load_pv(par, pa, va);
        v4f_ c = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int T = par; T < ntile; T += 2) {
            if (T + 2 < ntile) load_pv(T + 2, pb, vb);
#pragma unroll
            for (int m = 0; m < 8; ++m) c = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[m] * inv, va[m], c, 0, 0, 0);      // p = e * inv, the rounding of the soft_max's own scaling step (ggml_vec_scale_f32)
#pragma unroll
            for (int m = 0; m < 8; ++m) { pa[m] = pb[m]; va[m] = vb[m]; }
        }
        if (par == 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) xch[(dt * 4 + r) * 64 + lane] = c[r];
        }
        __syncthreads();
        if (par == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ir = 4 * kq + r;
                if (ir < nrows) att[(int64_t)(i0 + ir) * H * 64 + (int64_t) h * 64 + 16 * dt + l16] = c[r] + xch[(dt * 4 + r) * 64 + lane];
            }
        }
    }
}

// ---- the 16-row form with TWO workgroups per CU: the probabilities in LDS as the fp16 values they are -------------------------------
// exp() of the soft_max comes out of a table of fp16 values (ggml.c:10911-10960: table_exp_f16), so an (unscaled) probability is
// stored exactly in 2 bytes: 16 rows x 2048 keys = 64 KiB and two 8-wave workgroups share a CU (k_attention_mfma16 keeps f32 scores:
// 128 KiB, one workgroup). The price: a score must be known in f32 until its row's maximum is, so K.Q runs twice -- pass A keeps only
// the row maxima, pass B repeats the same MFMA chains (same operands, same order: same bits) and stores table[f16(s - max)]; the
// f64 sum of those values is exact in any order (<= 2^13 fp16-valued terms), V.P reads p = (float) e16 * inv as before. Results
// bit-identical to k_attention_mfma / k_attention_mfma16 and to the oracle's dot_qk_mfma / dot_pv_mfma. Measured (Falcon-7B block, 71 heads, one launch;
// profiles/r04_attn_forms.txt): 2048 tokens 1.364 ms against 0.978 (k_attention_mfma) and 1.292 (k_attention_mfma16); 512 tokens 0.137 / 0.114 / 0.103 --
// two workgroups per CU do not pay for the second K.Q: a 16-token tile fetches every key / value row twice as often per query as a 32-token one, and that
// traffic (L2), not the score matrix's trips through HBM, is what the 16-row forms wait for. Opt-in (FQ_ATTN_MFMA16H=1 / ggml_hip_debug_attention_form(17)),
// kept with its test. ps: uint16 elements per row (32 ntile_max + 8).
__global__ void __launch_bounds__(512, 4) k_attention_mfma16h(const float * __restrict__ qkv, int N, int H, int HKV, const int * __restrict__ n_past_ptr,
                                                              const float * __restrict__ kc, const float * __restrict__ vc,
                                                              const uint16_t * __restrict__ exp_tab, float * __restrict__ att, int ps) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    float  * rmax = (float *) smem;                                 // [8 waves][16 rows]
    double * rsum = (double *)(rmax + 8 * 16);                      // [8 waves][16 rows]
    float  * xch  = (float *)(rsum + 8 * 16);                       // [4 dim tiles][4][64]
    uint16_t * p  = (uint16_t *)(xch + 4 * 4 * 64);                 // [16][ps]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l16 = lane & 15, kq = lane >> 4;
    const int h = blockIdx.x, i0 = blockIdx.y * 16, hk = h / (H / HKV), heads = H + 2 * HKV;
    const int n_past = *n_past_ptr;
    const int nrows = N - i0 < 16 ? N - i0 : 16;
    const int n_kv_max = n_past + i0 + nrows;                       // keys the tile's last token sees
    const int n_rows_cache = n_past + N;                            // key / value rows that exist
    const int ntile = (n_kv_max + 31) >> 5, nsub = 2 * ntile;
    const int eo = kq >> 1, hi = kq & 1;                            // the lane group's elements: odd (1) / even (0) ones of the high (1) / low (0) half
    // ---- scores, twice
    {
        float q16[16];
        {
            const float * qrow = qkv + ((int64_t)(i0 + (l16 < nrows ? l16 : nrows - 1)) * heads + h) * 64 + 32 * hi;
#pragma unroll
            for (int v = 0; v < 8; ++v) { const f32x4 t = ((const f32x4 *) qrow)[v]; q16[2 * v] = eo ? t.y : t.x; q16[2 * v + 1] = eo ? t.w : t.z; }
        }
        auto load_k = [&](int U, float (&k16)[16]) {
            const int j = 16 * U + l16;
            const float * krow = kc + ((int64_t)(j < n_rows_cache ? j : n_rows_cache - 1) * HKV + hk) * 64 + 32 * hi;
#pragma unroll
            for (int v = 0; v < 8; ++v) { const f32x4 t = ((const f32x4 *) krow)[v]; k16[2 * v] = eo ? t.y : t.x; k16[2 * v + 1] = eo ? t.w : t.z; }
        };
        float ka[16], kb[16];
        float mx[4] = { -INFINITY, -INFINITY, -INFINITY, -INFINITY };
        // pass A: the row maxima
        if (wid < nsub) load_k(wid, ka);
        for (int U = wid; U < nsub; U += 8) {
            if (U + 8 < nsub) load_k(U + 8, kb);
            v4f_ c = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int m = 0; m < 16; ++m) c = __builtin_amdgcn_mfma_f32_16x16x4f32(q16[m], ka[m], c, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ir = 4 * kq + r, j = 16 * U + l16;
                mx[r] = fq_max_f32(mx[r], (ir < nrows && j <= n_past + i0 + ir) ? c[r] * 0.125f : -INFINITY);
            }
#pragma unroll
            for (int v = 0; v < 16; ++v) ka[v] = kb[v];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float m = reduce16(mx[r], op_max());
            if (l16 == 0) rmax[wid * 16 + 4 * kq + r] = m;
        }
        if (wid < nsub) load_k(wid, ka);                            // (pass B's first keys: requested before the barrier)
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float m = rmax[4 * kq + r];
#pragma unroll
            for (int w2 = 1; w2 < 8; ++w2) m = fq_max_f32(m, rmax[w2 * 16 + 4 * kq + r]);
            mx[r] = m;
        }
        // pass B: the same chains again; exp() of the visible keys as fp16 bits, zeros up to the last tile, the sums aside
        double ls[4] = { 0.0, 0.0, 0.0, 0.0 };
        for (int U = wid; U < nsub; U += 8) {
            if (U + 8 < nsub) load_k(U + 8, kb);
            v4f_ c = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int m = 0; m < 16; ++m) c = __builtin_amdgcn_mfma_f32_16x16x4f32(q16[m], ka[m], c, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ir = 4 * kq + r, j = 16 * U + l16;
                uint16_t e = 0;
                if (ir < nrows && j <= n_past + i0 + ir) { const uint16_t hb = f2h_bits(c[r] * 0.125f - mx[r]); e = exp_tab ? exp_tab[hb] : exp_f16_formula(hb); }
                p[ir * ps + j] = e;
                ls[r] += (double) h2f_bits(e);                                                           // (exact in any order)
            }
#pragma unroll
            for (int v = 0; v < 16; ++v) ka[v] = kb[v];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double t = ls[r];
            t += __shfl_xor(t, 1); t += __shfl_xor(t, 2); t += __shfl_xor(t, 4); t += __shfl_xor(t, 8);
            if (l16 == 0) rsum[wid * 16 + 4 * kq + r] = t;
        }
    }
    __syncthreads();
    // ---- V.P: wave = (16-dim tile dt, tile parity par)
    {
        const int dt = wid & 3, par = wid >> 2;
        const uint16_t * prow = p + l16 * ps + 16 * hi;             // keys 16 hi .. 16 hi + 15 of a tile: the even or the odd ones
        double sum = rsum[l16];
#pragma unroll
        for (int w2 = 1; w2 < 8; ++w2) sum += rsum[w2 * 16 + l16];
        const float inv = (float)(1.0 / sum);
        auto load_pv = [&](int T, fq_u4 (&p2)[2], float (&v8)[8]) {
            p2[0] = *(const fq_u4 *)(prow + 32 * T); p2[1] = *(const fq_u4 *)(prow + 32 * T + 8);
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const int j = 32 * T + 16 * hi + 2 * m + eo;
                v8[m] = vc[((int64_t)(j < n_rows_cache ? j : n_rows_cache - 1) * HKV + hk) * 64 + 16 * dt + l16];
            }
        };
        // element 2 m + eo of the 16 halves: the low (eo = 0) or high half of word m
        auto pe = [&](const fq_u4 (&p2)[2], int m) {
            const fq_u4 & q = p2[m >> 2];
            const uint32_t w = (m & 3) == 0 ? q.x : ((m & 3) == 1 ? q.y : ((m & 3) == 2 ? q.z : q.w));
            return h2f_bits((uint16_t)(eo ? w >> 16 : w & 0xFFFFu));
        };
        fq_u4 pa[2], pb[2]; float va[8], vb[8];
        if (par < ntile) load_pv(par, pa, va);
        v4f_ c = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int T = par; T < ntile; T += 2) {
            if (T + 2 < ntile) load_pv(T + 2, pb, vb);
#pragma unroll
            for (int m = 0; m < 8; ++m) c = __builtin_amdgcn_mfma_f32_16x16x4f32(pe(pa, m) * inv, va[m], c, 0, 0, 0);      // p = e * inv, the rounding of the soft_max's own scaling step (ggml_vec_scale_f32)
            pa[0] = pb[0]; pa[1] = pb[1];
#pragma unroll
            for (int m = 0; m < 8; ++m) va[m] = vb[m];
        }
        if (par == 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) xch[(dt * 4 + r) * 64 + lane] = c[r];
        }
        __syncthreads();
        if (par == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ir = 4 * kq + r;
                if (ir < nrows) att[(int64_t)(i0 + ir) * H * 64 + (int64_t) h * 64 + 16 * dt + l16] = c[r] + xch[(dt * 4 + r) * 64 + lane];
            }
        }
    }
}

// 1: the two dot products of the attention accumulate f32 products in f64 like the reference's portable build (fq_attn_dev.h)

// ---- prefill attention with the probabilities resident in LDS, 32 query tokens per workgroup ("flash" form, round 5: the default while the rows fit) -------
// The arithmetic of k_attention_mfma, operand for operand (v_mfma_f32_32x32x2_f32 chains: score(i, j) over s = 0..31 of dims s, 32 + s; out = E + O, the chains
// over the even / odd 32-key tiles, in a tile over s = 0..15 of keys s, 16 + s; soft_max: max, table[f16(s - max)], exact f64 sum, p = e * (float)(1 / sum)) --
// bit-identical to it and to the oracle's dot_qk_mfma / dot_pv_mfma -- but no score ever leaves the CU: the scratch form moves the N x n_kv score matrix through
// HBM four times (1.25 GB per Falcon-7B launch at 2048 tokens, 17 x the q / K / V / output bytes). A score must be known in f32 until its row's maximum is, and
// exp() comes out of a table of fp16 values (ggml.c:10911-10960), so:
//   pass A   K.Q on the matrix pipe; a wave keeps the f32 scores of ITS tiles in registers (KEEP, <= 2048 keys: 8 tiles x 16 registers; beyond that only the row
//            maxima are kept and pass B runs the same chains again -- same operands, same order: the same bits); row maxima through 1 KiB of LDS
//   pass B   e = table[f16(s - max)] stored as the 2-byte value it is: 32 rows x n_kv x 2 B of LDS (128 KiB at 2048 keys); row sums as exact integers, f64 across waves
//   pass C   V.P with p = (float) e16 * inv, formed one tile ahead of the matrix instructions that take it
// 8 waves: passes A and B deal the 32-key tiles round-robin (two waves per SIMD; key tiles go from L2 straight to registers one tile ahead, a tile is used by one
// wave only); pass C is four chains (dim half x tile parity: the association fixes that) on waves 0-3, one per SIMD, value tiles three ahead. Workgroups are
// PERSISTENT from 4 items per CU on (one per CU, items = (head, query tile) handed out by a counter, heavy query tiles first). Two units of f32 matrix work at the
// pipe's measured rate (one v_mfma_f32_32x32x2_f32 per 30.5 ns and SIMD, and vector instructions do NOT overlap it: scripts/microbench/mb_mfma_f32_*.hip): 0.28 ms
// per Falcon-7B block at 2048 tokens; measured 0.75 (NOTEBOOK 9.1: value loads and p-forming in front of pass C's one wave per SIMD, item turnover).
// pitch_h: halfwords per LDS row = 32 ntile_max + 8 (row pitch 16 bytes off a multiple of 64: conflict-free b128 reads down a column of rows).
// Loads: compiler-managed (FQ_FLASH_PLAIN). The first versions requested the tiles by inline assembly with hand-counted waits tied to the destination registers ("+v")
// because hipcc sinks a plain prefetch load down to its first use -- measured: the same speed (the passes are not bound by load latency), so the plain form is the default.
#ifndef FQ_FLASH_PLAIN
#define FQ_FLASH_PLAIN 1            // 1 (default): compiler-managed loads. 0: the tiles requested by inline-asm loads one / two tiles ahead with hand-counted waits -- measured the SAME speed on MI355X (0.926 against 0.922 ms per 2048-token Falcon-7B launch: the passes are not bound by load latency) and, on one box of the pool, intermittently wrong in tiles whose waves do not all hold keys (scripts/gpu_attn_bisect.py): off
#endif
#if FQ_FLASH_PLAIN
template <int OFF> __device__ __forceinline__ void fl_gload4(f32x4 & d, const float * p) { d = *(const f32x4 *)((const char *) p + OFF); }
__device__ __forceinline__ void fl_gload1(float & d, const float * p) { d = *p; }
#define FL_WAIT8(n, a)  do { } while (0)
#define FL_WAIT16(n, a) do { } while (0)
#else
template <int OFF> __device__ __forceinline__ void fl_gload4(f32x4 & d, const float * p) { asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(d) : "v"(p), "n"(OFF)); }
__device__ __forceinline__ void fl_gload1(float & d, const float * p) { asm volatile("global_load_dword %0, %1, off" : "=v"(d) : "v"(p)); }
#define FL_WAIT8(n, a)  asm volatile("s_waitcnt vmcnt(" #n ")" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]))
#define FL_WAIT16(n, a) asm volatile("s_waitcnt vmcnt(" #n ")" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), \
                                                                "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]))
#endif
#define FL_QK(c, k8) _Pragma("unroll") for (int v_ = 0; v_ < 8; ++v_) { \
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(q8[v_].x, k8[v_].x, c, 0, 0, 0); c = __builtin_amdgcn_mfma_f32_32x32x2f32(q8[v_].y, k8[v_].y, c, 0, 0, 0); \
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(q8[v_].z, k8[v_].z, c, 0, 0, 0); c = __builtin_amdgcn_mfma_f32_32x32x2f32(q8[v_].w, k8[v_].w, c, 0, 0, 0); }
// k_attn_pack_k: the keys [0, 32 nt) of every KV head re-laid in the matrix instruction's operand order -- tile T, register v, lane (hf, li): the four floats
// K[32 T + li][32 hf + 4 v ..] -- so that a wave's request for a tile is eight instructions of 1 KiB of CONSECUTIVE bytes each. Read from the cache as it
// lies, lane (hf, li) of a request touches its own 128-byte line: 64 lines per instruction, 512 tag look-ups per tile, and eight waves of them keep a CU's
// vector cache busy for as long as the matrix pipe needs for the tile -- the first versions of k_attention_flash ran at the scratch form's speed whatever
// their inside looked like. 512 KiB per Falcon-7B launch at 2048 keys; rows beyond the cache re-read its last row.
#ifndef FQ_ATTN_PACK_V
#define FQ_ATTN_PACK_V 1
#endif
__global__ void __launch_bounds__(256) k_attn_pack_k(const float * __restrict__ kc, const float * __restrict__ vc, int N, int HKV, const int * __restrict__ n_past_ptr, float * __restrict__ kt, int nt_total) {
    const int T = blockIdx.x, hk = blockIdx.y, n_rows_cache = *n_past_ptr + N;
    f32x4 * const dst = (f32x4 *) kt + ((size_t) hk * nt_total + T) * 512;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int o = threadIdx.x + 256 * u;                          // slot of the packed tile: (v, hf, li)
        const int v = o >> 6, hf = (o >> 5) & 1, li = o & 31;
        const int j = 32 * T + li;
        dst[o] = *(const f32x4 *)(kc + ((int64_t)(j < n_rows_cache ? j : n_rows_cache - 1) * HKV + hk) * 64 + 32 * hf + 4 * v);
    }
#if FQ_ATTN_PACK_V
    // ... and the values in pass C's operand order behind all the keys: tile T, dim half dh, register quad q, lane (hf, li): V[32 T + 16 hf + 4 q + e][32 dh + li], e = 0..3 --
    // four requests of 1 KiB of consecutive bytes per tile and wave instead of sixteen that touch two 128-byte lines each
    f32x4 * const vdst = (f32x4 *)(kt + (size_t) HKV * nt_total * 2048) + ((size_t) hk * nt_total + T) * 512;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int o = threadIdx.x + 256 * u;                          // (dh, q, lane)
        const int dh = o >> 8, q = (o >> 6) & 3, hf = (o >> 5) & 1, li = o & 31;
        f32x4 t;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int j = 32 * T + 16 * hf + 4 * q + e;
            t[e] = vc[((int64_t)(j < n_rows_cache ? j : n_rows_cache - 1) * HKV + hk) * 64 + 32 * dh + li];
        }
        vdst[o] = t;
    }
#endif
}
// KEEP (round 5, the default up to 64 key tiles): a wave keeps the scores of ITS tiles -- T = wid, wid + 8, ...: NT / 8 tiles x 16 registers -- from pass A to pass B instead
// of running the chains twice: two units of matrix work per launch instead of three (the f32 matrix instruction and the vector ALU do not overlap on a SIMD of this
// chip, scripts/microbench/mb_mfma_f32_mix.hip: a second run of the chains is 0.18 ms of a 0.91 ms launch at 2048 tokens). The scores are the same chains' results, kept
// as c * 0.125f with the invisible ones set to -inf (table[f16(-inf)] = 0, the reference's own rule for a masked score: ggml.c:10925); pass B is then vector work only.
// LONG (round 6): contexts whose 32 rows of fp16 probabilities do NOT fit the LDS (beyond 74 key tiles = 2368 keys; the reference's 8k / 16k contexts, BASELINE
// config 5). The LDS rows hold a CHUNK of NT key tiles; the row maxima (pass A) and the row sums (pass B without stores) come from two runs of the K.Q chains over
// all tiles, then chunk after chunk: K.Q a third time -> the chunk's exp() values into LDS -> V.P of the chunk's tiles into accumulators that live across the
// chunks. Four units of f32 matrix work instead of two -- the same chains on the same operands in the same order, so the same bits as every other form
// (tests/test_gpu_block_ops.py::test_prefill_attention_forms_are_bit_identical at 4096 / 8192 / 16384 keys) -- and still no score in HBM: the scratch form
// moves N x n_kv x H x 16 bytes per launch (4.8 GB for a 512-token batch at 8192 keys) and needs a 1.2 GB scratch allocation.
template <bool TAB, int NT, bool PACKED, bool KEEP, bool LONG = false>
__global__ void __launch_bounds__(512) k_attention_flash(const float * __restrict__ qkv, int N, int H, int HKV, const int * __restrict__ n_past_ptr,
                                                         const float * __restrict__ kc, const float * __restrict__ vc,
                                                         const uint16_t * __restrict__ exp_tab, float * __restrict__ att, const float * __restrict__ kt, int nt_total, int dbg, int * __restrict__ next_item) {
    constexpr int PH = 32 * NT + 8;                                               // halfwords per LDS row (a compile-time pitch: the 16 rows of a lane are immediate offsets)
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint16_t * const eh   = (uint16_t *) smem;                                    // [32][PH] fp16 bits of exp()
    float    * const rmax = (float *)(smem + (size_t) 32 * PH * 2);               // [8 waves][32 rows]
    double   * const rsum = (double *)(rmax + 8 * 32);                            // [8 waves][32 rows]
    float    * const xch  = (float *)(rsum + 8 * 32);                             // [2 dim halves][16][64]
    const int tid = threadIdx.x, lane0 = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    // PERSISTENT workgroups (round 5): the grid is one workgroup per CU (a workgroup takes a whole CU: 8 waves x ~210 VGPRs), and each walks the (head, query tile)
    // items blockIdx.x, blockIdx.x + gridDim.x, ... -- heavy tiles (the last tokens of the prompt) first, so every workgroup gets the same mix. Per-pass timing had
    // shown 19 % of a launch to be workgroup turnover (dispatch of a 512-thread, 160 KiB workgroup onto a drained CU, kernel arguments, position word).
    // Items are handed out by a counter (next_item, zeroed by the launcher; nullptr: one item per workgroup): a fixed round-robin gave workgroups 0..70 the heaviest
    // tile of every round (512 tokens: 0.128 against 0.121 ms). The next index is fetched by one lane during the item's pass A and read by everybody behind barrier 3.
    const int n_past = *n_past_ptr;
    const int ntq = (N + 31) >> 5, n_items = ntq * H, heads = H + 2 * HKV;
    int * const nxt = (int *)(xch + 2 * 16 * 64);                     // one LDS word behind xch
  for (int item = (int) blockIdx.x; item < n_items;) {
    // (the lane's coordinates opaque per item: what the compiler derives from them -- row offsets of the output, of the value rows, LDS addresses -- is then formed in
    // the item instead of once per kernel and kept in spill slots: the registers of pass A are the kernel's peak, 224 of 256 for the scores, the queries and two key tiles)
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    const int li = lane & 31, hf = lane >> 5;
    if (next_item && tid == 64) *nxt = (int) gridDim.x + atomicAdd(next_item, 1);      // (a lane of wave 1: published by the time barrier 3 has been passed)
    const int h = item % H, i0 = (ntq - 1 - item / H) * 32, hk = h / (H / HKV);
    const int nrows = N - i0 < 32 ? N - i0 : 32;
    const int n_kv_max = n_past + i0 + nrows;                       // keys the tile's last token sees
    const int n_rows_cache = n_past + N;                            // key / value rows that exist
    const int ntile_real = (n_kv_max + 31) >> 5;
    // (tuning aid, FQ_ATTN_DBG: bits 1 / 2 / 4 run pass A / B / C over no tiles -- results are garbage, the time is what the other passes cost)
    const int ntileA = (dbg & 1) ? 0 : ntile_real, ntileB = (dbg & 2) ? 0 : ntile_real, ntileC = (dbg & 4) ? 0 : ntile_real;
    // (pass C's accumulators and roles, declared here: the LONG form runs pass C inside the scope of passes A and B, chunk by chunk)
    const int dh = wid & 1, par = (wid >> 1) & 1;
    v16f c = {0};
    // ---- passes A and B: K.Q, tiles wid, wid + 8, ...; the next tile is requested before a tile's matrix instructions start (rows beyond the cache
    // re-read its last row: every request is unconditional, the counts in the waits are exact)
    {
        f32x4 q8[8];
        const float * qrow = qkv + ((int64_t)(i0 + (li < nrows ? li : nrows - 1)) * heads + h) * 64 + 32 * hf;
#pragma unroll
        for (int v = 0; v < 8; ++v) q8[v] = ((const f32x4 *) qrow)[v];
        const float * const kbase = PACKED ? kt + (int64_t) hk * nt_total * 2048 + 4 * lane : kc + (int64_t) hk * 64 + 32 * hf;
        auto load_k = [&](int T, f32x4 (&k8)[8]) {
            if constexpr (PACKED) {                                   // (k_attn_pack_k's layout: register v of the tile is 1 KiB of consecutive bytes over the wave)
                const float * kp = kbase + (int64_t)(T < nt_total ? T : nt_total - 1) * 2048;
                const float * kq = kp + 1024;                          // (the instruction's offset field is 13 bits, signed)
                fl_gload4<0>(k8[0], kp); fl_gload4<1024>(k8[1], kp); fl_gload4<2048>(k8[2], kp); fl_gload4<3072>(k8[3], kp);
                fl_gload4<0>(k8[4], kq); fl_gload4<1024>(k8[5], kq); fl_gload4<2048>(k8[6], kq); fl_gload4<3072>(k8[7], kq);
            } else {
                const int j = 32 * T + li;
                const float * krow = kbase + (int64_t)(j < n_rows_cache ? j : n_rows_cache - 1) * HKV * 64;
                fl_gload4<0>(k8[0], krow);  fl_gload4<16>(k8[1], krow); fl_gload4<32>(k8[2], krow); fl_gload4<48>(k8[3], krow);
                fl_gload4<64>(k8[4], krow); fl_gload4<80>(k8[5], krow); fl_gload4<96>(k8[6], krow); fl_gload4<112>(k8[7], krow);
            }
        };
        f32x4 ka[8], kb[8];
        load_k(wid, ka);
        if constexpr (KEEP) {
            static_assert(FQ_FLASH_PLAIN && NT % 8 == 0, "the KEEP form uses compiler-managed loads");
            constexpr int KW = NT / 8;                                 // tiles of a wave
            v16f S[KW];
            const int tfull = (n_past + i0 + 1) >> 5;                  // tiles [0, tfull) are visible to every row of the item
#pragma unroll
            for (int k = 0; k < KW; ++k) {
                const int T = wid + 8 * k;
                if (T < ntile_real) {                                  // (wave-uniform)
                    if (k + 1 < KW) load_k(T + 8, kb);
                    v16f c = {0};
                    FL_QK(c, ka);
                    if (T < tfull) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) c[r] = c[r] * 0.125f;
                    } else {
                        const int lim = n_past + i0 - 32 * T - li;     // key 32 T + li is visible to row ir iff ir + lim >= 0
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int ir = (r & 3) + 8 * (r >> 2) + 4 * hf;
                            c[r] = (ir + lim >= 0) ? c[r] * 0.125f : -INFINITY;
                        }
                    }
                    S[k] = c;
                    if (k + 1 < KW) {
#pragma unroll
                        for (int v = 0; v < 8; ++v) ka[v] = kb[v];
                    }
                }
            }
            float mx[16];                                              // (the maxima only now: the query and key registers are free again)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx[r] = -INFINITY;
#pragma unroll
            for (int k = 0; k < KW; ++k) {
                if (wid + 8 * k < ntile_real) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) mx[r] = fq_max_f32(mx[r], S[k][r]);
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float m = reduce32(mx[r], op_max());
                if (li == 0) rmax[wid * 32 + (r & 3) + 8 * (r >> 2) + 4 * hf] = m;
            }
            __syncthreads();
            unsigned lsum[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ir = (r & 3) + 8 * (r >> 2) + 4 * hf;
                float m = rmax[ir];
#pragma unroll
                for (int w = 1; w < 8; ++w) m = fq_max_f32(m, rmax[w * 32 + ir]);
                mx[r] = m; lsum[r] = 0u;
            }
            uint16_t * const ecol = eh + (size_t)(4 * hf) * PH + li;
#pragma unroll
            for (int k = 0; k < KW; ++k) {
                const int T = wid + 8 * k;
                if (T < ntile_real) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int irl = (r & 3) + 8 * (r >> 2);
                        const uint16_t hb = f2h_bits(S[k][r] - mx[r]);
                        uint16_t eb;
                        if constexpr (TAB) eb = exp_tab[hb]; else eb = exp_f16_formula(hb);
                        ecol[irl * PH + 32 * T] = eb;
                        lsum[r] += (unsigned)(h2f_bits(eb) * 16777216.0f);
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const double sm = reduce32((double) lsum[r] * (1.0 / 16777216.0), op_add());
                if (li == 0) rsum[wid * 32 + (r & 3) + 8 * (r >> 2) + 4 * hf] = sm;
            }
        } else {
        {   // pass A: row maxima
            float mx[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) mx[r] = -INFINITY;
            for (int T = wid; T < ntileA; T += 8) {
                load_k(T + 8, kb);
                FL_WAIT8(8, ka);
                __builtin_amdgcn_sched_barrier(0);                    // (the scheduler moves an asm statement wherever its operands allow: the wait for the NEXT tile went in front of this tile's matrix instructions)
                v16f c = {0};
                FL_QK(c, ka);
                const int lim = n_past + i0 - 32 * T - li;           // key 32 T + li is visible to row ir iff ir + lim >= 0
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ir = (r & 3) + 8 * (r >> 2) + 4 * hf;
                    const float sc = c[r] * 0.125f;
                    mx[r] = fq_max_f32(mx[r], (ir < nrows && ir + lim >= 0) ? sc : -INFINITY);
                }
                __builtin_amdgcn_sched_barrier(0);
                FL_WAIT8(0, kb);                                      // (requested a whole tile of matrix instructions ago)
#pragma unroll
                for (int v = 0; v < 8; ++v) ka[v] = kb[v];
            }
            FL_WAIT8(0, ka);                                          // (a wave without tiles: its one request has landed before the registers are used for anything else)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float m = reduce32(mx[r], op_max());
                if (li == 0) rmax[wid * 32 + (r & 3) + 8 * (r >> 2) + 4 * hf] = m;
            }
        }
        load_k(wid, ka);                                              // (pass B's first tile is in flight across the barrier)
        __syncthreads();
        float mrow[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ir = (r & 3) + 8 * (r >> 2) + 4 * hf;
            float m = rmax[ir];
#pragma unroll
            for (int w = 1; w < 8; ++w) m = fq_max_f32(m, rmax[w * 32 + ir]);
            mrow[r] = m;
        }
        if constexpr (LONG) {
            static_assert(!KEEP && FQ_FLASH_PLAIN && NT % 2 == 0, "the LONG form: compiler-managed loads, whole tile pairs per chunk");
            unsigned lsum[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) lsum[r] = 0u;
            v16f cp = {0}; int Tp = 0, toff = 0;
            uint16_t * const ecol = eh + (size_t)(4 * hf) * PH + li;
            // one score of the previous tile: STORE = its exp() value into the chunk's LDS rows (tile Tp - toff of the chunk), else into the row's integer sum
            auto epiL = [&](int r, auto store_tag) {
                constexpr bool STORE = decltype(store_tag)::value;
                const int irl = (r & 3) + 8 * (r >> 2), ir = irl + 4 * hf;
                const float sc = cp[r] * 0.125f;
                const bool vis = ir < nrows && 32 * Tp + li <= n_past + i0 + ir;
                const uint16_t hb = f2h_bits(sc - mrow[r]);
                uint16_t eb;
                if constexpr (TAB) eb = exp_tab[hb]; else eb = exp_f16_formula(hb);
                eb = vis ? eb : (uint16_t) 0;
                if constexpr (STORE) ecol[irl * PH + 32 * (Tp - toff)] = eb;
                else lsum[r] += (unsigned)(h2f_bits(eb) * 16777216.0f);
            };
            // the K.Q chains of tiles Tb + wid, Tb + wid + 8, ... < Te (ka holds the first of them), pass B's software pipeline: the exp() work of a tile between the
            // matrix instructions of the next
            auto pass_b = [&](int Tb, int Te, auto store_tag) {
                if (Tb + wid < Te) {
                    load_k(Tb + wid + 8, kb);
                    v16f c0_ = {0};
                    FL_QK(c0_, ka);
                    cp = c0_; Tp = Tb + wid;
#pragma unroll
                    for (int v = 0; v < 8; ++v) ka[v] = kb[v];
                    for (int T = Tb + wid + 8; T < Te; T += 8) {
                        load_k(T + 8, kb);
                        v16f c1_ = {0};
#pragma unroll
                        for (int v = 0; v < 8; ++v) {
                            c1_ = __builtin_amdgcn_mfma_f32_32x32x2f32(q8[v].x, ka[v].x, c1_, 0, 0, 0);
                            c1_ = __builtin_amdgcn_mfma_f32_32x32x2f32(q8[v].y, ka[v].y, c1_, 0, 0, 0);
                            epiL(2 * v, store_tag);
                            __builtin_amdgcn_sched_barrier(0);
                            c1_ = __builtin_amdgcn_mfma_f32_32x32x2f32(q8[v].z, ka[v].z, c1_, 0, 0, 0);
                            c1_ = __builtin_amdgcn_mfma_f32_32x32x2f32(q8[v].w, ka[v].w, c1_, 0, 0, 0);
                            epiL(2 * v + 1, store_tag);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        cp = c1_; Tp = T;
#pragma unroll
                        for (int v = 0; v < 8; ++v) ka[v] = kb[v];
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) epiL(r, store_tag);
                }
            };
            std::integral_constant<bool, true> STORE_T; std::integral_constant<bool, false> SUM_T;
            // ---- the row sums: every tile once more, nothing stored (a lane adds <= 64 values <= 2^24 per row at 16384 keys: exact in 32 bits)
            pass_b(0, ntileB, SUM_T);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const double sm = reduce32((double) lsum[r] * (1.0 / 16777216.0), op_add());
                if (li == 0) rsum[wid * 32 + (r & 3) + 8 * (r >> 2) + 4 * hf] = sm;
            }
            __syncthreads();
            float inv = 0.0f;
            if (wid < 4) {
                double sm = rsum[li];
#pragma unroll
                for (int w = 1; w < 8; ++w) sm += rsum[w * 32 + li];
                inv = (float)(1.0 / sm);
            }
            // ---- pass C's pieces (k_attention_flash's own, with the LDS column relative to the chunk)
            const uint16_t * erow = eh + (size_t) li * PH + 16 * hf;
            const float * const vbase = vc + (int64_t) hk * 64 + 32 * dh + li;
            const f32x4 * const vpk = (const f32x4 *)(kt + (int64_t) HKV * nt_total * 2048) + ((int64_t) hk * nt_total * 2 + dh) * 256 + lane;
            auto load_v = [&](int T, float (&v16)[16]) {
                if constexpr (PACKED && FQ_ATTN_PACK_V) {
                    const f32x4 * vp = vpk + (int64_t)(T < nt_total ? T : nt_total - 1) * 512;
#pragma unroll
                    for (int q = 0; q < 4; ++q) { const f32x4 t = vp[64 * q]; v16[4 * q] = t.x; v16[4 * q + 1] = t.y; v16[4 * q + 2] = t.z; v16[4 * q + 3] = t.w; }
                } else {
#pragma unroll
                    for (int s_ = 0; s_ < 16; ++s_) {
                        const int j = 32 * T + 16 * hf + s_;
                        fl_gload1(v16[s_], vbase + (int64_t)(j < n_rows_cache ? j : n_rows_cache - 1) * HKV * 64);
                    }
                }
            };
            auto form_p = [&](int Tl, float (&p16)[16]) {                 // Tl: the tile's index inside the chunk (one beyond the chunk reads words behind the row: never used)
                const uint4 e0 = *(const uint4 *)(erow + 32 * Tl), e1 = *(const uint4 *)(erow + 32 * Tl + 8);
                const unsigned w[8] = { e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w };
#pragma unroll
                for (int u = 0; u < 8; ++u) { p16[2 * u] = h2f_bits((uint16_t)(w[u] & 0xFFFFu)) * inv; p16[2 * u + 1] = h2f_bits((uint16_t)(w[u] >> 16)) * inv; }
            };
            auto pv_tile = [&](const float (&p16)[16], const float (&v16)[16]) {
#pragma unroll
                for (int u = 0; u < 16; ++u) c = __builtin_amdgcn_mfma_f32_32x32x2f32(p16[u], v16[u], c, 0, 0, 0);
            };
            // ---- chunk after chunk: exp() values of NT tiles into LDS, their V.P into the accumulators (tile order of a chain = ascending: the association of every form)
            for (int c0 = 0; c0 < ntile_real; c0 += NT) {
                const int c1 = c0 + NT < ntile_real ? c0 + NT : ntile_real;
                load_k(c0 + wid, ka);
                if (c0) __syncthreads();                              // (the previous chunk's pass C has read its rows)
                toff = c0;
                if (!(dbg & 2)) pass_b(c0, c1, STORE_T);
                __syncthreads();
                if (wid < 4 && !(dbg & 4)) {
                    constexpr int NB = 4;
                    float vv[NB][16], pp[2][16];
#pragma unroll
                    for (int b = 0; b < NB - 1; ++b) load_v(c0 + par + 2 * b, vv[b]);
                    int T = c0 + par;
                    form_p(T - c0, pp[0]);
                    for (; T + 2 * (NB - 1) < c1; T += 2 * NB) {
#pragma unroll
                        for (int b = 0; b < NB; ++b) {
                            load_v(T + 2 * b + 2 * (NB - 1), vv[(b + NB - 1) % NB]);
                            form_p(T - c0 + 2 * b + 2, pp[(b + 1) & 1]);
                            pv_tile(pp[b & 1], vv[b]);
                        }
                    }
#pragma unroll
                    for (int b = 0; b < NB - 1; ++b) {
                        if (T + 2 * b < c1) {
                            if (b + 1 < NB - 1) form_p(T - c0 + 2 * b + 2, pp[(b + 1) & 1]);
                            pv_tile(pp[b & 1], vv[b]);
                        }
                    }
                }
            }
            if (wid < 4 && par == 1) {
#pragma unroll
                for (int r = 0; r < 16; ++r) xch[(dh * 16 + r) * 64 + lane] = c[r];
            }
        } else
        {   // pass B: exp() values into LDS, row sums. Software-pipelined by hand: the exp() arithmetic of the wave's previous tile (~25 VALU instructions per
            // score, 16 scores per lane) is issued BETWEEN the matrix instructions of the current one -- a 64-cycle v_mfma_f32_32x32x2_f32 leaves ~14 issue
            // slots, and a wave that ran its 32 dependent matrix instructions back to back and its 400 VALU instructions after them would keep the pipe idle
            // for half of the pass. Row sums as integers: a probability's exp() is an fp16 value <= 1, i.e. k x 2^-24 with k <= 2^24, and a lane adds at most
            // ten of them per row -- exact in 32 bits (and in f64 across lanes and waves afterwards).
            unsigned lsum[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) lsum[r] = 0u;
            v16f cp = {0}; int Tp = 0;
            uint16_t * const ecol = eh + (size_t)(4 * hf) * PH + li;
            auto epi = [&](int r) {                                   // one score of the previous tile (r is a compile-time constant after unrolling)
                const int irl = (r & 3) + 8 * (r >> 2), ir = irl + 4 * hf;
                const float sc = cp[r] * 0.125f;
                const bool vis = ir < nrows && 32 * Tp + li <= n_past + i0 + ir;
                const uint16_t hb = f2h_bits(sc - mrow[r]);
                uint16_t eb;
                if constexpr (TAB) eb = exp_tab[hb]; else eb = exp_f16_formula(hb);
                eb = vis ? eb : (uint16_t) 0;
                ecol[irl * PH + 32 * Tp] = eb;
                lsum[r] += (unsigned)(h2f_bits(eb) * 16777216.0f);
            };
            if (wid < ntileB) {
                load_k(wid + 8, kb);
                FL_WAIT8(8, ka);
                __builtin_amdgcn_sched_barrier(0);
                FL_QK(cp, ka);
                Tp = wid;
                __builtin_amdgcn_sched_barrier(0);
                FL_WAIT8(0, kb);
#pragma unroll
                for (int v = 0; v < 8; ++v) ka[v] = kb[v];
                for (int T = wid + 8; T < ntileB; T += 8) {
                    load_k(T + 8, kb);
                    FL_WAIT8(8, ka);
                    __builtin_amdgcn_sched_barrier(0);
                    v16f c = {0};
#pragma unroll
                    for (int v = 0; v < 8; ++v) {
                        c = __builtin_amdgcn_mfma_f32_32x32x2f32(q8[v].x, ka[v].x, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x2f32(q8[v].y, ka[v].y, c, 0, 0, 0);
                        epi(2 * v);
                        __builtin_amdgcn_sched_barrier(0);
                        c = __builtin_amdgcn_mfma_f32_32x32x2f32(q8[v].z, ka[v].z, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x2f32(q8[v].w, ka[v].w, c, 0, 0, 0);
                        epi(2 * v + 1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    cp = c; Tp = T;
                    FL_WAIT8(0, kb);
#pragma unroll
                    for (int v = 0; v < 8; ++v) ka[v] = kb[v];
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) epi(r);
            }
            FL_WAIT8(0, ka);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const double sm = reduce32((double) lsum[r] * (1.0 / 16777216.0), op_add());
                if (li == 0) rsum[wid * 32 + (r & 3) + 8 * (r >> 2) + 4 * hf] = sm;
            }
        }
        }   // !KEEP
    }
    __syncthreads();
    // ---- pass C: V.P, wave = (dim half, tile parity), waves 0-3 (one per SIMD): four chains is what the association allows. Values two tiles ahead.
    if (!LONG && wid < 4) {
        double sm = rsum[li];
#pragma unroll
        for (int w = 1; w < 8; ++w) sm += rsum[w * 32 + li];
        const float inv = (float)(1.0 / sm);
        const uint16_t * erow = eh + (size_t) li * PH + 16 * hf;
        const float * const vbase = vc + (int64_t) hk * 64 + 32 * dh + li;
        const f32x4 * const vpk = (const f32x4 *)(kt + (int64_t) HKV * nt_total * 2048) + ((int64_t) hk * nt_total * 2 + dh) * 256 + lane;      // (PACKED: k_attn_pack_k's value part)
        auto load_v = [&](int T, float (&v16)[16]) {
            if constexpr (PACKED && FQ_ATTN_PACK_V) {
                const f32x4 * vp = vpk + (int64_t)(T < nt_total ? T : nt_total - 1) * 512;
#pragma unroll
                for (int q = 0; q < 4; ++q) { const f32x4 t = vp[64 * q]; v16[4 * q] = t.x; v16[4 * q + 1] = t.y; v16[4 * q + 2] = t.z; v16[4 * q + 3] = t.w; }
            } else {
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int j = 32 * T + 16 * hf + s;
                fl_gload1(v16[s], vbase + (int64_t)(j < n_rows_cache ? j : n_rows_cache - 1) * HKV * 64);
            }
            }
        };
        // p = e * inv, the rounding of the soft_max's own scaling step (ggml_vec_scale_f32), formed ONE TILE AHEAD of the matrix instructions that take it: the chain
        // LDS read -> convert -> multiply -> matrix instruction was exposed twice per tile (ds_read_b128, s_waitcnt lgkmcnt(0), 8 matrix instructions) -- a fifth of the
        // pass; with the next tile's sixteen values formed between this tile's matrix instructions only their issue slots are left
        auto form_p = [&](int T, float (&p16)[16]) {
            const uint4 e0 = *(const uint4 *)(erow + 32 * T), e1 = *(const uint4 *)(erow + 32 * T + 8);
            const unsigned w[8] = { e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w };
#pragma unroll
            for (int u = 0; u < 8; ++u) { p16[2 * u] = h2f_bits((uint16_t)(w[u] & 0xFFFFu)) * inv; p16[2 * u + 1] = h2f_bits((uint16_t)(w[u] >> 16)) * inv; }
        };
        auto pv_tile = [&](const float (&p16)[16], const float (&v16)[16]) {
#pragma unroll
            for (int u = 0; u < 16; ++u) c = __builtin_amdgcn_mfma_f32_32x32x2f32(p16[u], v16[u], c, 0, 0, 0);
        };
        constexpr int NB = 4;                                           // value tiles T, T + 2, T + 4, T + 6 of this parity in flight (even: the p buffers alternate)
        float vv[NB][16], pp[2][16];
#pragma unroll
        for (int b = 0; b < NB - 1; ++b) load_v(par + 2 * b, vv[b]);
        int T = par;
        form_p(T, pp[0]);                                               // (a tile index beyond the item's last reads LDS words behind the row: never used)
        for (; T + 2 * (NB - 1) < ntileC; T += 2 * NB) {
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                load_v(T + 2 * b + 2 * (NB - 1), vv[(b + NB - 1) % NB]);
                form_p(T + 2 * b + 2, pp[(b + 1) & 1]);
                pv_tile(pp[b & 1], vv[b]);
            }
        }
#pragma unroll
        for (int b = 0; b < NB - 1; ++b) {
            if (T + 2 * b < ntileC) {
                if (b + 1 < NB - 1) form_p(T + 2 * b + 2, pp[(b + 1) & 1]);
                pv_tile(pp[b & 1], vv[b]);
            }
        }
        if (par == 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) xch[(dh * 16 + r) * 64 + lane] = c[r];
        }
    }
    __syncthreads();
    if (wid < 2) {                                                      // par == 0: E + O
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ir = (r & 3) + 8 * (r >> 2) + 4 * hf;
            if (ir < nrows) att[(int64_t)(i0 + ir) * H * 64 + (int64_t) h * 64 + 32 * dh + li] = c[r] + xch[(dh * 16 + r) * 64 + lane];
        }
    }
    // (no barrier before the next item: its first LDS writes are rmax after its pass A -- a region nobody reads now --, eh / rsum / xch only behind its own barriers;
    // the next index word is rewritten by wave 1 only after wave 1 itself has read it, and every other wave reads it before its next barrier 1)
    item = next_item ? __builtin_amdgcn_readfirstlane(*(volatile int *) nxt) : n_items;      // (wave-uniform, and known to the compiler as such: the item's tile counts and row bases stay scalar)
    if (next_item) __syncthreads();                                   // (the word is rewritten at the top of the next item: everybody must have read it)
  }
}
// LDS rows for 16, 32 or 74 key tiles (512, 1024, 2368 keys): the instantiations of the compile-time pitch
// (64: the KEEP form's largest -- 8 tiles x 16 score registers per wave; 65-74 tiles run the three-pass form)
static bool attn_flash_keep() { static const int v = getenv("FQ_ATTN_KEEP") ? atoi(getenv("FQ_ATTN_KEEP")) : 1; return v != 0; }
// (beyond 74 tiles: the LONG form, chunks of 64 tiles -- returned as 1064; FQ_ATTN_FLASH_LONG=0: the scratch form there, as before round 6)
static bool attn_flash_long_on() { static const int v = getenv("FQ_ATTN_FLASH_LONG") ? atoi(getenv("FQ_ATTN_FLASH_LONG")) : 1; return v != 0; }
static int attn_flash_nt(int max_n_kv) { const int nt = (max_n_kv + 31) >> 5; return nt <= 16 ? 16 : (nt <= 32 ? 32 : (nt <= 64 && attn_flash_keep() ? 64 : (nt <= 74 ? 74 : (attn_flash_long_on() ? 1064 : 0)))); }
static size_t attn_flash_lds(int nt) { return (size_t) 32 * (size_t)(32 * nt + 8) * 2 + 8 * 32 * 4 + 8 * 32 * 8 + 2 * 16 * 64 * 4 + 16; }
static bool attn_flash_fits(int max_n_kv) { return attn_flash_nt(max_n_kv) != 0; }
template <bool TAB, int NT, bool PACKED, bool KEEP, bool LONG = false>
static void launch_attention_flash_t(const float * qkv, int N, int H, int HKV, const int * n_past_dev, const float * k_cache, const float * v_cache,
                                     const uint16_t * exp_table, float * att, const float * kt, int nt_total, hipStream_t st) {
    const size_t lds = attn_flash_lds(NT);
    static bool attr = false;
    if (!attr && lds > 64 * 1024) { HIP_CHECK(hipFuncSetAttribute((const void *) k_attention_flash<TAB, NT, PACKED, KEEP, LONG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds)); attr = true; }
    static const int dbg = getenv("FQ_ATTN_DBG") ? atoi(getenv("FQ_ATTN_DBG")) : 0;
    static const int persist = getenv("FQ_ATTN_PERSIST") ? atoi(getenv("FQ_ATTN_PERSIST")) : 1;      // 0: one workgroup per item (A/B)
    const int n_items = H * ((N + 31) / 32);
    const bool per = persist && n_items >= 4 * fq_ctx().n_cu;           // persistent workgroups where there are rounds enough to amortise over
    int * counter = per ? fq_ctx().scalar_i32 + 32 : nullptr;           // (a word of the library's scalar scratch; zeroed on the launch's own stream)
    if (per) HIP_CHECK(hipMemsetAsync(counter, 0, 4, st));
    hipLaunchKernelGGL((k_attention_flash<TAB, NT, PACKED, KEEP, LONG>), dim3((unsigned)(per ? fq_ctx().n_cu : n_items)), dim3(512), lds, st, qkv, N, H, HKV, n_past_dev, k_cache, v_cache, exp_table, att, kt, nt_total, dbg, counter);
}
template <bool TAB, bool PACKED>
static void launch_attention_flash(int nt, const float * qkv, int N, int H, int HKV, const int * n_past_dev, const float * k_cache, const float * v_cache,
                                   const uint16_t * exp_table, float * att, const float * kt, int nt_total, hipStream_t st) {
    const bool keep = attn_flash_keep();
    if (nt == 16)      { if (keep) launch_attention_flash_t<TAB, 16, PACKED, true>(qkv, N, H, HKV, n_past_dev, k_cache, v_cache, exp_table, att, kt, nt_total, st);
                         else      launch_attention_flash_t<TAB, 16, PACKED, false>(qkv, N, H, HKV, n_past_dev, k_cache, v_cache, exp_table, att, kt, nt_total, st); }
    else if (nt == 32) { if (keep) launch_attention_flash_t<TAB, 32, PACKED, true>(qkv, N, H, HKV, n_past_dev, k_cache, v_cache, exp_table, att, kt, nt_total, st);
                         else      launch_attention_flash_t<TAB, 32, PACKED, false>(qkv, N, H, HKV, n_past_dev, k_cache, v_cache, exp_table, att, kt, nt_total, st); }
    else if (nt == 64) launch_attention_flash_t<TAB, 64, PACKED, true>(qkv, N, H, HKV, n_past_dev, k_cache, v_cache, exp_table, att, kt, nt_total, st);
    else if (nt == 1064) launch_attention_flash_t<TAB, 64, PACKED, false, true>(qkv, N, H, HKV, n_past_dev, k_cache, v_cache, exp_table, att, kt, nt_total, st);
    else               launch_attention_flash_t<TAB, 74, PACKED, false>(qkv, N, H, HKV, n_past_dev, k_cache, v_cache, exp_table, att, kt, nt_total, st);
}
// bytes of the packed keys a launch of this size wants (k_attn_pack_k): prompts of FQ_ATTN_PACK_MIN_N (default 256) tokens and more
static int attn_pack_min_n() { static const int v = getenv("FQ_ATTN_PACK_MIN_N") ? atoi(getenv("FQ_ATTN_PACK_MIN_N")) : 256; return v; }
static size_t attn_pack_bytes(int N, int HKV, int max_n_kv) { return N >= attn_pack_min_n() ? (size_t) HKV * (size_t)((max_n_kv + 31) >> 5) * 8192 * (FQ_ATTN_PACK_V ? 2 : 1) : 0; }

static int g_attn_f64 = 0;
static int g_attn_form = 0;      // prefill attention on the matrix pipe: 0 = default (k_attention_flash while 32 rows of fp16 probabilities fit LDS, else the scratch form), 32 = k_attention_mfma (scores in the global scratch), 16 = k_attention_mfma16, 17 = k_attention_mfma16h
void fq_attn_set_form(int form) { g_attn_form = form; }
void fq_attn_set_f64(int on) { g_attn_f64 = on != 0; }
int  fq_attn_f64() { return g_attn_f64; }
template <int R>
static void launch_attention_rows(const float * qkv, int N, int H, int HKV, const int * n_past_dev, int p_stride, size_t lds, const float * k_cache,
                                  const float * v_cache, const uint16_t * exp_table, float * att, hipStream_t st, float * p_scratch = nullptr) {
    if (g_attn_f64) launch_attention_rows_t<R, true>(qkv, N, H, HKV, n_past_dev, p_stride, lds, k_cache, v_cache, exp_table, att, p_scratch, st);
    else            launch_attention_rows_t<R, false>(qkv, N, H, HKV, n_past_dev, p_stride, lds, k_cache, v_cache, exp_table, att, p_scratch, st);
}
// score rows of long-context launches (one slice per workgroup). A model context owns its own buffer, sized once for its
// (n_batch, n_ctx) at context_create (fq_attention_scratch_need) and passed in; the process-wide one below serves only the
// op-level API (ggml_hip_attention: synchronous, one stream) and is grown on demand up to FQ_ATTN_SCRATCH_GB (default 4) GiB.
static bool attn_flash_default() { static const int v = getenv("FQ_ATTN_FLASH") ? atoi(getenv("FQ_ATTN_FLASH")) : 1; return v != 0; }
static size_t att_scratch_cap() {
    static const size_t cap = (size_t)(getenv("FQ_ATTN_SCRATCH_GB") ? atoi(getenv("FQ_ATTN_SCRATCH_GB")) : 4) << 30;
    return cap;
}
size_t fq_attention_scratch_need(int N, int H, int max_n_kv, int HKV) {
    if (N < 4) return 0;
    if (N >= 32 && attn_flash_default() && attn_flash_fits(max_n_kv)) return attn_pack_bytes(N, HKV, max_n_kv);      // k_attention_flash: only the packed keys
    const size_t ps = (size_t)((max_n_kv + 31) & ~31);
    const size_t a = (N >= 32 && !(attn_flash_default() && attn_flash_fits(max_n_kv))) ? (size_t)((N + 31) / 32) * (size_t) H * 32 * ps * 4 : 0;   // k_attention_mfma (32 tokens per workgroup): only contexts too long for k_attention_flash's LDS rows
    const size_t row = (size_t)((max_n_kv + 3) & ~3) * 4;
    const size_t b = (16 * 4 + 16 * 64 * 8) + 4 * row > 56 * 1024 ? (size_t)((N + 3) / 4) * (size_t) H * 4 * row : 0;   // 4 tokens per workgroup, once their rows leave LDS
    const size_t need = a > b ? a : b;
    // beyond the cap the context still gets cap bytes: att_scratch() checks every launch's own byte count, so batches at low n_past (whose score rows are
    // short) keep the MFMA / 4-row forms in a context whose (n_batch, n_ctx) corner would not fit
    return need > att_scratch_cap() ? att_scratch_cap() : need;
}
static float * g_att_scratch = nullptr;
static size_t  g_att_scratch_bytes = 0;
static float * att_scratch(fq_att_scratch * own, size_t bytes, hipStream_t st) {
    if (own) return bytes <= own->bytes ? own->p : nullptr;          // never grown here: launches of a context may be captured
    if (bytes > att_scratch_cap()) return nullptr;
    if (bytes > g_att_scratch_bytes) {
        HIP_CHECK(hipStreamSynchronize(st));
        if (g_att_scratch) HIP_CHECK(hipFree(g_att_scratch));
        g_att_scratch = nullptr; g_att_scratch_bytes = 0;
        if (hipMalloc((void **) &g_att_scratch, bytes) != hipSuccess) { (void) hipGetLastError(); g_att_scratch = nullptr; return nullptr; }
        g_att_scratch_bytes = bytes;
    }
    return g_att_scratch;
}

void fq_launch_attention(const float * qkv, int N, int H, int HKV, int D, const int * n_past_dev, int max_n_kv, const float * k_cache,
                         const float * v_cache, const uint16_t * exp_table, float * att, hipStream_t st, int64_t seq_stride, fq_att_scratch * own_scratch) {
    FQ_TL(st, "attention");
    if (D != 64) { fprintf(stderr, "ggml-hip: attention: head_dim %d != 64\n", D); exit(1); }
    const int p_stride = (max_n_kv + 3) & ~3;
    const size_t fixed = 16 * 4 + 16 * 64 * 8, row = (size_t) p_stride * 4, budget = 150 * 1024;
    if (fixed + row > 160 * 1024) { fprintf(stderr, "ggml-hip: attention: %d keys do not fit the score buffer in LDS\n", max_n_kv); exit(1); }
    // tokens per workgroup: more tokens re-use a key tile more often, but their score rows (n_kv floats each) sit in LDS and
    // leave fewer workgroups per CU. Measured on Falcon-7B, ms per block for 4 / 2 / 1 tokens: 2048-token prompt 2.08 / 2.21 /
    // (f64: 3.7); 4096 tokens 8.65 / 8.36; 8192 tokens 52.9 / 36.1 / 41.9 -> 4 while that keeps >= 3 workgroups per CU
    // (n_kv <= ~3000), else 2
    static const int force = getenv("FQ_ATTN_ROWS") ? atoi(getenv("FQ_ATTN_ROWS")) : -1;      // tuning override: 0 = one token per workgroup
    const bool fit4 = fixed + 4 * row <= budget, fit2 = fixed + 2 * row <= budget;
    int R = (N >= 4 && fixed + 4 * row <= 56 * 1024) ? 4 : ((N >= 2 && fit2) ? 2 : 1);
    if (seq_stride) R = 1;                                       // independent sequences share no key tile
    if (force == 0 || seq_stride) R = 1; else if (force == 8 && N >= 4 && fixed + 8 * row <= budget) R = 8; else if (force == 4 && N >= 4 && fit4) R = 4; else if (force == 2 && N >= 2 && fit2) R = 2;
    // beyond ~3000 keys: 4 tokens per workgroup with the score rows in the global scratch instead of 2 with them in LDS
    // (8192-token prompt: 32 ms per block against 36-40; 8 tokens per workgroup: 38). The scratch is N x H x n_kv floats --
    // 1.2 GB for a 512-token batch at 8192 keys -- and is not used beyond FQ_ATTN_SCRATCH_GB (default 4) GiB
    static const int use_mfma = getenv("FQ_ATTN_MFMA") ? atoi(getenv("FQ_ATTN_MFMA")) : 1;
    if (use_mfma && N >= 32 && !g_attn_f64 && !seq_stride && force < 0) {
        // the score rows of 16 query tokens in LDS while they fit (~2400 keys); beyond that 32 tokens per workgroup and the global scratch
        static const int env_form = (getenv("FQ_ATTN_MFMA16H") && atoi(getenv("FQ_ATTN_MFMA16H"))) ? 17 : ((getenv("FQ_ATTN_MFMA16") && atoi(getenv("FQ_ATTN_MFMA16"))) ? 16 : 0);
        const int form = g_attn_form ? g_attn_form : env_form;      // (16 and 17: measured slower than the scratch form at 2048 tokens, see k_attention_mfma16h)
        if ((form == 1 || (form == 0 && attn_flash_default())) && attn_flash_fits(max_n_kv)) {
            const int nt = attn_flash_nt(max_n_kv), nt_total = (max_n_kv + 31) >> 5;
            // pass B reads exp() from the fp16 table even when the in-kernel formula is verified: ~25 VALU instructions per score make the pass VALU-bound (measured in
            // the model: 2048-token prompt 110.6 ms with the formula, the table gather rides on the vector cache under the matrix instructions); FQ_ATTN_FLASH_TAB=0: formula
            static const int use_tab = getenv("FQ_ATTN_FLASH_TAB") ? atoi(getenv("FQ_ATTN_FLASH_TAB")) : 1;
            if (!exp_table && use_tab) exp_table = fq_ctx().exp_table;
            const size_t pk = attn_pack_bytes(N, HKV, max_n_kv);
            float * kt = pk ? att_scratch(own_scratch, pk, st) : nullptr;
            if (kt) {
                hipLaunchKernelGGL(k_attn_pack_k, dim3((unsigned) nt_total, (unsigned) HKV), dim3(256), 0, st, k_cache, v_cache, N, HKV, n_past_dev, kt, nt_total);
                if (exp_table) launch_attention_flash<true, true>(nt, qkv, N, H, HKV, n_past_dev, k_cache, v_cache, exp_table, att, kt, nt_total, st);
                else           launch_attention_flash<false, true>(nt, qkv, N, H, HKV, n_past_dev, k_cache, v_cache, exp_table, att, kt, nt_total, st);
            } else {
                if (exp_table) launch_attention_flash<true, false>(nt, qkv, N, H, HKV, n_past_dev, k_cache, v_cache, exp_table, att, nullptr, 0, st);
                else           launch_attention_flash<false, false>(nt, qkv, N, H, HKV, n_past_dev, k_cache, v_cache, exp_table, att, nullptr, 0, st);
            }
            return;
        }
        static const int max_kv16 = getenv("FQ_ATTN_MFMA16_MAXKV") ? atoi(getenv("FQ_ATTN_MFMA16_MAXKV")) : 4096;
        const int ps16 = ((max_n_kv + 31) & ~31) + 4;
        const size_t lds16 = ((size_t) 16 * ps16 + 8 * 16 + 16 + 4 * 4 * 64) * 4;
        if (form == 16 && lds16 <= 158 * 1024 && max_n_kv <= max_kv16) {
            static size_t g16 = 0;
            if (lds16 > 64 * 1024 && lds16 > g16) { HIP_CHECK(hipFuncSetAttribute((const void *) k_attention_mfma16, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds16)); g16 = lds16; }
            hipLaunchKernelGGL(k_attention_mfma16, dim3((unsigned) H, (unsigned)((N + 15) / 16)), dim3(512), lds16, st, qkv, N, H, HKV, n_past_dev, k_cache, v_cache, exp_table, att, ps16);
            return;
        }
        const int ps16h = ((max_n_kv + 31) & ~31) + 8;                                   // uint16 elements per row
        const size_t lds16h = (8 * 16) * 4 + (8 * 16) * 8 + (4 * 4 * 64) * 4 + (size_t) 16 * ps16h * 2;
        if (form == 17 && lds16h <= 158 * 1024) {
            static size_t g16h = 0;
            if (lds16h > 64 * 1024 && lds16h > g16h) { HIP_CHECK(hipFuncSetAttribute((const void *) k_attention_mfma16h, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds16h)); g16h = lds16h; }
            hipLaunchKernelGGL(k_attention_mfma16h, dim3((unsigned) H, (unsigned)((N + 15) / 16)), dim3(512), lds16h, st, qkv, N, H, HKV, n_past_dev, k_cache, v_cache, exp_table, att, ps16h);
            return;
        }
        const int ps = (max_n_kv + 31) & ~31;
        float * scr = att_scratch(own_scratch, (size_t)((N + 31) / 32) * (size_t) H * 32 * (size_t) ps * 4, st);
        if (scr) {
            // (a 128-token-per-workgroup form with the key / value tiles shared through LDS was built and measured: 124 ms against
            // 112 ms for a 2048-token Falcon-7B prompt -- a barrier per tile and idle waves at the causal edge cost more than the
            // L2 traffic it saves)
            hipLaunchKernelGGL(k_attention_mfma, dim3((unsigned) H, (unsigned)((N + 31) / 32)), dim3(256), 0, st, qkv, N, H, HKV, n_past_dev, k_cache, v_cache, exp_table, att, ps, scr);
            return;
        }
    }
    static const int use_scratch = getenv("FQ_ATTN_SCRATCH") ? atoi(getenv("FQ_ATTN_SCRATCH")) : 1;
    if (use_scratch && force < 0 && R == 2 && N >= 4 && !seq_stride) {
        const size_t need = (size_t)((N + 3) / 4) * (size_t) H * 4 * row;
        float * scr = att_scratch(own_scratch, need, st);
        if (scr) { launch_attention_rows<4>(qkv, N, H, HKV, n_past_dev, p_stride, fixed, k_cache, v_cache, exp_table, att, st, scr); return; }
    }
    if (R == 8)      launch_attention_rows<8>(qkv, N, H, HKV, n_past_dev, p_stride, fixed + 8 * row, k_cache, v_cache, exp_table, att, st);
    else if (R == 4) launch_attention_rows<4>(qkv, N, H, HKV, n_past_dev, p_stride, fixed + 4 * row, k_cache, v_cache, exp_table, att, st);
    else if (R == 2) launch_attention_rows<2>(qkv, N, H, HKV, n_past_dev, p_stride, fixed + 2 * row, k_cache, v_cache, exp_table, att, st);
    else {
        const size_t lds = fixed + row;
        if (lds > 64 * 1024) {
            static size_t g = 0;
            if (lds > g) {
                HIP_CHECK(hipFuncSetAttribute((const void *) k_attention<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
                HIP_CHECK(hipFuncSetAttribute((const void *) k_attention<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
                g = lds;
            }
        }
        if (g_attn_f64) hipLaunchKernelGGL(k_attention<true>, dim3((unsigned) H, (unsigned) N), dim3(256), lds, st, qkv, H, HKV, n_past_dev, k_cache, v_cache, exp_table, att, seq_stride);
        else            hipLaunchKernelGGL(k_attention<false>, dim3((unsigned) H, (unsigned) N), dim3(256), lds, st, qkv, H, HKV, n_past_dev, k_cache, v_cache, exp_table, att, seq_stride);
    }
}

// ------------------------------------------------------------------------------------------------ self-test
// DPP/readlane wave reductions against the plain __shfl_xor butterfly (same pairing order): must agree bit for bit.
__global__ void k_selftest_reduce(int * __restrict__ mismatches, unsigned seed) {
    const int lane = threadIdx.x & 63;
    unsigned h = seed ^ (blockIdx.x * 2654435761u) ^ (threadIdx.x * 40503u);
    for (int it = 0; it < 64; ++it) {
        h = h * 1664525u + 1013904223u;
        const float  f = __builtin_bit_cast(float, (h & 0x007FFFFFu) | 0x3F800000u) * ((h >> 24) & 1 ? -3.7f : 1.3f) * (float)(1 + (h >> 28));
        const double d = (double) f * 1.000000119 + 1e-9 * (double)(h & 1023);
        const int    i = (int)(h >> 7) - (1 << 23);
        const float  fa = wave_reduce(f, op_add()), fb = wave_reduce_shfl(f, op_add());
        const double da = wave_reduce(d, op_add()), db = wave_reduce_shfl(d, op_add());
        const int    ia = wave_reduce(i, op_add()), ib = wave_reduce_shfl(i, op_add());
        const float  ma = wave_reduce(f, op_max()), mb = wave_reduce_shfl(f, op_max());
        int bad = 0;
        bad += __builtin_bit_cast(int, fa) != __builtin_bit_cast(int, fb);
        bad += __builtin_bit_cast(long long, da) != __builtin_bit_cast(long long, db);
        bad += ia != ib;
        bad += __builtin_bit_cast(int, ma) != __builtin_bit_cast(int, mb);
        if (bad) atomicAdd(mismatches, bad);
        (void) lane;
    }
}
int fq_selftest_reduce(hipStream_t st) {
    int * dev = nullptr; int host = -1;
    HIP_CHECK(hipMalloc((void **) &dev, 4));
    HIP_CHECK(hipMemsetAsync(dev, 0, 4, st));
    hipLaunchKernelGGL(k_selftest_reduce, dim3(64), dim3(256), 0, st, dev, 12345u);
    HIP_CHECK(hipMemcpyAsync(&host, dev, 4, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    HIP_CHECK(hipFree(dev));
    return host;
}

// ---- exp_f16_formula (fq_device.h) against the host-built table, every non-NaN fp16 input
__global__ void k_verify_exp_formula(const uint16_t * __restrict__ table, int * mismatches) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 65536u) return;
    if ((i & 0x7C00u) == 0x7C00u && (i & 0x03FFu)) return;      // NaN in: NaN out either way, payloads are not compared
    if (exp_f16_formula((uint16_t) i) != table[i]) atomicAdd(mismatches, 1);
}
// diagnostic: the inputs exp_f16_fast leaves undecided, with the table's entries: (bits << 16 | entry), ascending; returns their number
__global__ void k_exp_boundary(const uint16_t * __restrict__ table, unsigned * __restrict__ flags) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 65536u) return;
    uint16_t h;
    const bool nan = (i & 0x7C00u) == 0x7C00u && (i & 0x03FFu);
    flags[i] = (!nan && !exp_f16_fast((uint16_t) i, h)) ? ((i << 16) | table[i]) : 0xFFFFFFFFu;
}
int fq_exp_boundary(const uint16_t * exp_table, unsigned * out_host, int cap, hipStream_t st) {
    unsigned * d = nullptr;
    HIP_CHECK(hipMalloc((void **) &d, 65536 * 4));
    hipLaunchKernelGGL(k_exp_boundary, dim3(256), dim3(256), 0, st, exp_table, d);
    std::vector<unsigned> f(65536);
    HIP_CHECK(hipMemcpyAsync(f.data(), d, 65536 * 4, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    HIP_CHECK(hipFree(d));
    int n = 0;
    for (unsigned v : f) if (v != 0xFFFFFFFFu) { if (n < cap) out_host[n] = v; ++n; }
    return n;
}
int fq_verify_exp_formula(const uint16_t * exp_table, hipStream_t st) {
    int * d = nullptr; int h = -1;
    HIP_CHECK(hipMalloc((void **) &d, 4));
    HIP_CHECK(hipMemsetAsync(d, 0, 4, st));
    hipLaunchKernelGGL(k_verify_exp_formula, dim3(256), dim3(256), 0, st, exp_table, d);
    HIP_CHECK(hipMemcpyAsync(&h, d, 4, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    HIP_CHECK(hipFree(d));
    return h;
}
