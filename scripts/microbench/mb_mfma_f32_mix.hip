// mb_mfma_f32_mix.hip -- what does ONE other instruction cost between v_mfma_f32_32x32x2_f32 on MI355X? Per kind: 8 independent vector ALU instructions (not feeding the
// matrix instruction), 8 LDS writes, 4 global loads (L2 hits), 8 scalar ALU instructions; 2 waves per SIMD (the occupancy of k_attention_flash2).
//   hipcc --offload-arch=gfx950 -O3 -o mb_mfma_f32_mix mb_mfma_f32_mix.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v16f __attribute__((ext_vector_type(16)));
template <int KIND>
__global__ void __launch_bounds__(512) k(float * out, const float * in, int iters) {
    __shared__ float lds[512 * 9];
    v16f c = {0};
    const float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    float f0 = a, f1 = b, f2 = a + 1, f3 = b + 1, g = 0.0f;
    unsigned s0 = blockIdx.x;
    const float * p = in + threadIdx.x;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
            if (KIND == 1) { f0 = __builtin_fmaf(f0, 1.0001f, 0.5f); f1 = __builtin_fmaf(f1, 1.0001f, 0.5f); f2 = __builtin_fmaf(f2, 1.0001f, 0.5f); f3 = __builtin_fmaf(f3, 1.0001f, 0.5f);
                             f0 = __builtin_fmaf(f0, 0.9999f, 0.25f); f1 = __builtin_fmaf(f1, 0.9999f, 0.25f); f2 = __builtin_fmaf(f2, 0.9999f, 0.25f); f3 = __builtin_fmaf(f3, 0.9999f, 0.25f); }
            if (KIND == 2) {
#pragma unroll
                for (int q = 0; q < 8; ++q) lds[threadIdx.x + 512 * q] = f0;
            }
            if (KIND == 3) {
#pragma unroll
                for (int q = 0; q < 4; ++q) g += __builtin_nontemporal_load(p + 1024 * ((u * 4 + q) & 31));
            }
            if (KIND == 4) {
#pragma unroll
                for (int q = 0; q < 8; ++q) asm volatile("s_add_u32 %0, %0, 7" : "+s"(s0));
            }
        }
    }
    float s = 0; for (int r = 0; r < 16; ++r) s += c[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + f0 + f1 + f2 + f3 + g + lds[threadIdx.x] + (float) s0;
}
template <int KIND> static void run(const char * what, float * out, const float * in) {
    const int iters = 1000;
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(512), 0, 0, out, in, iters); hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(512), 0, 0, out, in, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %.1f ns per matrix instruction and SIMD (2 waves per SIMD; 64 cycles at 2.4 GHz = 26.7 ns)\n", what, ms * 1e6 / (iters * 16.0) / 2.0);
}
int main() {
    float * out, * in; hipMalloc(&out, 256 * 512 * 4); hipMalloc(&in, 64 * 1024 * 4); hipMemset(in, 0, 64 * 1024 * 4);
    run<0>("nothing between", out, in);
    run<1>("8 independent v_fma_f32 between", out, in);
    run<2>("8 ds_write_b32 between", out, in);
    run<3>("4 global_load_dword (L2) between", out, in);
    run<4>("8 s_add_u32 between", out, in);
    return 0;
}
