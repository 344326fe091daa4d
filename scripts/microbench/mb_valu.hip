// micro-benchmark (tuning aid): issue cadence of dependent / independent VALU chains for 1..3 waves per SIMD, f32 / f64 / DPP,
// in wall-clock ns per instruction (s_memrealtime, 100 MHz) -- what "N instructions" costs in a latency-bound prologue.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE>
__global__ void __launch_bounds__(1024) k(float * out, long long * t, int n) {
    float a = threadIdx.x * 1e-3f, b = 1.0001f, c = 0.5f, d = 0.25f;
    double x = threadIdx.x * 1e-3, y = 1.0001;
    __syncthreads();
    const long long t0 = wall_clock64();
    const long long c0 = clock64();
    for (int i = 0; i < n; ++i) {
        if (MODE == 0) { a = a * b + c; a = a * b + c; a = a * b + c; a = a * b + c; }                        // 4 dependent f32 fma
        if (MODE == 1) { a = a * b + c; c = c * b + d; d = d * b + a; b = b * 1.0f + 1e-9f; }                  // ~independent mix
        if (MODE == 2) { x = x * y + 0.5; x = x * y + 0.5; x = x * y + 0.5; x = x * y + 0.5; }                // 4 dependent f64 fma
        if (MODE == 3) { x += (double) a; a = a * b + c; x += (double) a; a = a * b + c; }                   // cvt + f64 add chain
        if (MODE == 4) {                                                                                       // dependent DPP + add (f32)
            a += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), 0xB1, 0xF, 0xF, false));
            a += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), 0x4E, 0xF, 0xF, false));
        }
    }
    const long long c1 = clock64();
    const long long t1 = wall_clock64();
    if (threadIdx.x == 0) { t[blockIdx.x * 2] = t1 - t0; t[blockIdx.x * 2 + 1] = c1 - c0; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + (float) x + b + c + d;
}

int main() {
    float * out; long long * t; CK(hipMalloc(&out, 256 * 1024 * 4)); CK(hipMalloc(&t, 256 * 16));
    long long h[512];
    const int n = 2000;
    const char * names[5] = { "4 dependent f32 fma", "4 mostly independent f32 fma", "4 dependent f64 fma", "2 x (cvt + f64 add) + 2 f32 fma", "2 x (dpp mov + f32 add), dependent" };
    const int per_iter[5] = { 4, 4, 4, 6, 4 };
    for (int threads = 256; threads <= 768; threads += 256) {
        printf("--- %d threads per workgroup (%d wave(s) per SIMD), 32 workgroups\n", threads, threads / 256);
#define RUN(M) { hipLaunchKernelGGL((k<M>), dim3(32), dim3(threads), 0, 0, out, t, n); hipLaunchKernelGGL((k<M>), dim3(32), dim3(threads), 0, 0, out, t, n); CK(hipDeviceSynchronize()); \
        CK(hipMemcpy(h, t, 64 * 8, hipMemcpyDeviceToHost)); \
        printf("%-40s %6.2f ns / instr / wave   (%.2f shader-clock ticks per instr; ticks per us %.0f)\n", names[M], h[0] * 10.0 / (n * per_iter[M]), (double) h[1] / (n * per_iter[M]), h[1] / (h[0] * 0.01)); }
        RUN(0) RUN(1) RUN(2) RUN(3) RUN(4)
    }
    return 0;
}
