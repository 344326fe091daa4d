// mb_clock_vs_cus.hip -- does the shader clock of an MI355X depend on how many CUs run vector + matrix work? Every workgroup (16 waves, one per CU) runs the GEMM's instruction
// mix (one v_mfma_i32_32x32x32_i8 + 48 f32 vector instructions per group) for a fixed number of iterations and reads both clocks before and after: s_memtime (shader clock
// cycles) and s_memrealtime (100 MHz). cycles / time = the clock the CU actually ran at; iterations / time = what the chip delivered.
//   hipcc --offload-arch=gfx950 -O3 -o mb_clock_vs_cus mb_clock_vs_cus.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
typedef int v16i __attribute__((ext_vector_type(16)));
typedef int v4i  __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(1024) k(float * out, unsigned long long * stamps, int iters) {
    v16i c = {0};
    const v4i a = { (int) threadIdx.x, 3, 5, 7 }, b = { 1, (int) threadIdx.x, 2, 4 };
    float f[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) f[i] = threadIdx.x * 1e-3f + i;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(f[i]) : "v"(c[i]));
            asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[i]) : "v"(1.0001f));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(0.9999f), "v"(0.5f));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += (float) c[r] + f[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}
int main() {
    float * out; unsigned long long * st;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&st, 256 * 2 * 8);
    const int iters = 20000;
    for (int n : { 16, 32, 64, 71, 128, 142, 192, 213, 256 }) {
        for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k, dim3(n), dim3(1024), 0, 0, out, st, iters); hipDeviceSynchronize(); }
        std::vector<unsigned long long> h(2 * n);
        hipMemcpy(h.data(), st, 2 * n * 8, hipMemcpyDeviceToHost);
        std::vector<double> mhz, us;
        for (int i = 0; i < n; ++i) { mhz.push_back((double) h[2 * i] / ((double) h[2 * i + 1] / 100.0)); us.push_back((double) h[2 * i + 1] / 100.0); }
        std::sort(mhz.begin(), mhz.end()); std::sort(us.begin(), us.end());
        printf("%3d workgroups of 16 waves: shader clock %6.0f MHz (min %6.0f, max %6.0f)   %8.1f us per workgroup (max %8.1f)   %7.1f group-steps per us for the chip\n",
               n, mhz[n / 2], mhz[0], mhz[n - 1], us[n / 2], us[n - 1], (double) n * 16 * iters / us[n - 1]);
    }
    return 0;
}
