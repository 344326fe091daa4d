// hip_context.h -- per-process device state of libggml_hip.so (one process drives one GPU; multi-GPU = one
// process per GPU, see pipeline.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>
#include "fq_types.h"
#include "kernels.h"

struct hip_context {
    bool        ready = false;
    int         device = 0;
    int         n_devices = 0;
    int         n_cu = 256;
    char        name[128] = {0};
    hipStream_t stream = nullptr;
    uint16_t *  gelu_table = nullptr;     // device, 65536 x fp16
    uint16_t *  exp_table = nullptr;      // device, 65536 x fp16
    const uint16_t * exp_table_attn = nullptr;   // what the attention kernels get: nullptr once exp_f16_formula is verified == exp_table
    int *       scalar_i32 = nullptr;     // device scratch scalar for the op-level API
    long long * dbg_stamps = nullptr;     // optional phase-stamp buffer (ggml_hip_debug_stamps), 2 x 4096 x 8 entries
    float *     ks_scratch = nullptr;     // partial sums of the K-share small-batch mat-mul: 4 x 16 columns x FQ_KS_MAX_M rows (allocated at init: launches may be captured)
};

#define FQ_KS_MAX_M 32768
#define FQ_KS_FLOATS ((size_t) 16 << 20)      // the scratch: 64 MiB = [segment][share][16 columns][rows] of the Q4_K form (4 x 16 x FQ_KS_MAX_M of the legacy K-share form)
hip_context & fq_ctx();
fq_weight fq_weight_alloc(int type, int64_t K, int64_t M, void ** slab_out);
fq_act    fq_act_alloc(int act_type, int64_t K, int64_t max_cols, void ** slab_out);
void      fq_mul_mat_q_acts(const fq_weight & w, const fq_act & a, int64_t N, float * dst, int64_t ldd,
                            const fq_gemv_epi & ep, hipStream_t st);
// Q4_K blocks at 5..16 columns (the small-batch form's fused sum launches); false: nothing launched, run the generic calls
// (min_cols: 5 = the mat-mul's own threshold; 3 for lock-step contexts, see fq_mul_mat_q_acts_from3)
bool      fq_mul_mat_q_acts_gelu_q8k(const fq_weight & w, const fq_act & a, int64_t N, float * dst, int64_t ldd, const fq_act & out, hipStream_t st, int min_cols = 5);      // dst = gelu(w a) + its Q8_K image
bool      fq_mul_mat_q_acts_out2(const fq_weight & wo, const fq_act & a_att, const fq_weight & down, const fq_act & a_ff, int64_t N, float * x, int64_t ldx, hipStream_t st, int min_cols = 5);   // x = (down + wo) + x
void      fq_mul_mat_q_acts_from3(const fq_weight & w, const fq_act & a, int64_t N, float * dst, int64_t ldd, const fq_gemv_epi & ep, hipStream_t st);
bool      fq_mul_mat_q_acts_pair(const fq_weight & w0, const fq_weight & w1, const fq_act & a, int64_t N, float * dst0, int64_t ldd0, const fq_gemv_epi & ep0,
                                 float * dst1, int64_t ldd1, const fq_gemv_epi & ep1, hipStream_t st);
struct ggml_hip_weight;
ggml_hip_weight * fq_weight_upload_part(int type, const void * host_blocks, int64_t K, int64_t rows, int64_t whole_rows);      // a row range of a larger matrix (split_tp.hip)
std::vector<float> fq_rope_table_host(int head_dim, int n_pos, int rope_n_ctx);
bool      fq_reference_order();
bool      fq_reference_fast();          // ggml_hip_reference_order(2): the reference's association on the fast kernels (legacy formats)
int       fq_config_epoch();        // bumped by every global switch that changes a launch list (reference order, forced mat-vec, sequential GEMM, debug modes)
bool      fq_prof_active();
void      fq_prof_open(hipStream_t st);
void      fq_prof_close(hipStream_t st, double bytes);
void      fq_prof_cancel();
void      fq_prof_events(hipEvent_t * start, hipEvent_t * stop);   // the open bracket's events (nullptr, nullptr when none)
