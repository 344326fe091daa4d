// mb_mfma_f32_rate.hip -- how fast does v_mfma_f32_32x32x2_f32 issue on MI355X: one dependent chain per wave, W waves per SIMD, with and without
// VALU work between the matrix instructions (the shape of k_attention_flash's passes).   hipcc --offload-arch=gfx950 -O3 -o mb_mfma_f32_rate mb_mfma_f32_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v16f __attribute__((ext_vector_type(16)));
template <int FILL>
__global__ void __launch_bounds__(1024) k(float * out, int iters, long long * cyc) {
    v16f c = {0};
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f, f = a;
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
            if (FILL) {
#pragma unroll
                for (int q = 0; q < FILL; ++q) f = __builtin_fmaf(f, 1.0001f, 0.5f);
                a = f * 1e-9f;                      // (the next instruction's operand depends on VALU work, as p = e * inv does)
            }
        }
    }
    const long long t1 = clock64();
    float s = 0; for (int r = 0; r < 16; ++r) s += c[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + f;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main() {
    float * out; long long * cyc; hipMalloc(&out, 1024 * 1024 * 4); hipMalloc(&cyc, 8);
    const int iters = 2000;
    for (int fill : {0, 8, 24}) for (int waves : {4, 8, 16}) {
        long long h = 0;
        for (int rep = 0; rep < 2; ++rep) {
            if (fill == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(64 * waves), 0, 0, out, iters, cyc);
            else if (fill == 8) hipLaunchKernelGGL(k<8>, dim3(256), dim3(64 * waves), 0, 0, out, iters, cyc);
            else hipLaunchKernelGGL(k<24>, dim3(256), dim3(64 * waves), 0, 0, out, iters, cyc);
            hipDeviceSynchronize();
        }
        hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        // clock64 = s_memtime at 100 MHz on this part: report per-instruction time in ns too
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        if (fill == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(64 * waves), 0, 0, out, iters, cyc);
        else if (fill == 8) hipLaunchKernelGGL(k<8>, dim3(256), dim3(64 * waves), 0, 0, out, iters, cyc);
        else hipLaunchKernelGGL(k<24>, dim3(256), dim3(64 * waves), 0, 0, out, iters, cyc);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        const double per_wave_ns = ms * 1e6 / (iters * 16.0);          // one wave's interval between its matrix instructions
        const double per_simd_ns = per_wave_ns / (waves / 4.0);         // the SIMD's interval between matrix instructions
        printf("fill %2d VALU between, %2d waves per CU (%d per SIMD): %.1f ns between a wave's instructions, %.1f ns per instruction and SIMD (64 cycles at 2.4 GHz = 26.7 ns)\n",
               fill, waves, waves / 4, per_wave_ns, per_simd_ns);
    }
    return 0;
}
