// mb_dot4.hip -- issue rate of v_dot4_i32_i8 against v_fma_f32 on one SIMD (are the integer dots of the mat-vec kernels
// full rate?). hipcc --offload-arch=gfx950 -O3 -o mb_dot4 mb_dot4.hip && ./mb_dot4
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int MODE>
__global__ void __launch_bounds__(1024) k(int * out, int n) {
    int a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = 5, a5 = 6, a6 = 7, a7 = 8;
    float f0 = a0, f1 = a1, f2 = a2, f3 = a3, f4 = 4, f5 = 5, f6 = 6, f7 = 7;
    const int x = out[0], y = out[1];
    const float fx = (float) x, fy = (float) y;
    for (int i = 0; i < n; ++i) {
        if (MODE == 0) {
            a0 = __builtin_amdgcn_sdot4(x, y, a0, false); a1 = __builtin_amdgcn_sdot4(x, y, a1, false);
            a2 = __builtin_amdgcn_sdot4(x, y, a2, false); a3 = __builtin_amdgcn_sdot4(x, y, a3, false);
            a4 = __builtin_amdgcn_sdot4(x, y, a4, false); a5 = __builtin_amdgcn_sdot4(x, y, a5, false);
            a6 = __builtin_amdgcn_sdot4(x, y, a6, false); a7 = __builtin_amdgcn_sdot4(x, y, a7, false);
        } else {
            f0 = __builtin_fmaf(fx, fy, f0); f1 = __builtin_fmaf(fx, fy, f1); f2 = __builtin_fmaf(fx, fy, f2); f3 = __builtin_fmaf(fx, fy, f3);
            f4 = __builtin_fmaf(fx, fy, f4); f5 = __builtin_fmaf(fx, fy, f5); f6 = __builtin_fmaf(fx, fy, f6); f7 = __builtin_fmaf(fx, fy, f7);
        }
    }
    out[2 + threadIdx.x + blockIdx.x * blockDim.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (int)(f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7);
}
int main() {
    int * d; hipMalloc(&d, (2 + 256 * 1024) * 4); hipMemset(d, 0, (2 + 256 * 1024) * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int n = 100000;
    for (int mode = 0; mode < 2; ++mode) for (int waves = 1; waves <= 4; waves *= 2) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(256 * waves > 1024 ? 1024 : 256 * waves), 0, 0, d, n);
            else           hipLaunchKernelGGL(k<1>, dim3(256), dim3(256 * waves > 1024 ? 1024 : 256 * waves), 0, 0, d, n);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("%s  %d wave(s) per SIMD: %.2f cycles per instruction per wave at 2.4 GHz (%.3f ms)\n", mode ? "v_fma_f32 " : "v_dot4_i32", waves, ms * 1e-3 * 2.4e9 / (8.0 * n) / waves, ms);
        }
    }
    return 0;
}
