// mb_mfma_i8_mix.hip -- k_gemm_q's stage in miniature on MI355X: one v_mfma_i32_32x32x32_i8 per group and 16 results to scale. What do the scaling instructions
// cost per kind -- alone and next to the matrix instruction -- and do the packed f32 forms (two results per instruction) issue at full rate?
//   hipcc --offload-arch=gfx950 -O3 -o mb_mfma_i8_mix mb_mfma_i8_mix.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int   v16i __attribute__((ext_vector_type(16)));
typedef int   v4i  __attribute__((ext_vector_type(4)));
typedef float v2f  __attribute__((ext_vector_type(2)));
// KIND: 0 nothing, 1 = 16 v_fma_f32, 2 = 8 v_pk_fma_f32, 3 = 16 v_cvt_f32_i32, 4 = 8 v_pk_mul_f32, 5 = 8 v_pk_add_f32, 6 = 16 v_mul_f32
template <int KIND, bool MFMA>
__global__ void __launch_bounds__(1024) k(float * out, int iters) {
    v16i c = {0};
    const v4i a = { (int) threadIdx.x, 3, 5, 7 }, b = { 1, (int) threadIdx.x, 2, 4 };
    float f[16]; v2f p[8]; int n[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { f[i] = threadIdx.x * 1e-3f + i; n[i] = threadIdx.x + i; }
#pragma unroll
    for (int i = 0; i < 8; ++i) p[i] = v2f{ f[2 * i], f[2 * i + 1] };
    const v2f m2 = { 1.0001f, 0.9999f }, a2 = { 0.5f, 0.25f };
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MFMA) c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0);
            if (KIND == 1) {
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(m2.x), "v"(a2.x));
            }
            if (KIND == 2) {
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(m2), "v"(a2));
            }
            if (KIND == 3) {
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(f[i]) : "v"(n[i]));
            }
            if (KIND == 4) {
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(m2));
            }
            if (KIND == 5) {
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(a2));
            }
            if (KIND == 6) {
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[i]) : "v"(m2.x));
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += (float) c[r] + f[r];
#pragma unroll
    for (int r = 0; r < 8; ++r) s += p[r].x + p[r].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int KIND, bool MFMA> static void run(const char * what, float * out, int wg_threads) {
    const int iters = 2000;
    hipLaunchKernelGGL((k<KIND, MFMA>), dim3(256), dim3(wg_threads), 0, 0, out, iters); hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND, MFMA>), dim3(256), dim3(wg_threads), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const int wps = wg_threads / 256;
    printf("%-34s %s  %d waves per SIMD: %6.1f ns per group and SIMD\n", what, MFMA ? "with the i8 matrix instruction" : "alone                         ", wps, ms * 1e6 / (iters * 8.0) / wps);
}
template <int KIND> static void both(const char * what, float * out) {
    for (int t = 512; t <= 1024; t *= 2) { run<KIND, false>(what, out, t); run<KIND, true>(what, out, t); }
}
int main() {
    float * out; hipMalloc(&out, 256 * 1024 * 4);
    both<0>("nothing", out);
    both<1>("16 v_fma_f32", out);
    both<2>("8 v_pk_fma_f32 (16 results)", out);
    both<3>("16 v_cvt_f32_i32", out);
    both<6>("16 v_mul_f32", out);
    both<4>("8 v_pk_mul_f32 (16 results)", out);
    both<5>("8 v_pk_add_f32 (16 results)", out);
    return 0;
}
