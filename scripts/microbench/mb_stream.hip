// micro-benchmark (tuning aid, not part of the library): what does a pure weight-stream launch of Falcon-7B size cost on
// this chip, back to back, for different grid shapes / loads in flight?   hipcc --offload-arch=gfx950 -O3 mb_stream.hip -o mb_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// linear: thread i reads vectors i, i+T, i+2T ... (T = total threads), U loads in flight
template <int U>
__global__ void __launch_bounds__(1024) k_linear(const u32x4 * __restrict__ p, long nvec, unsigned * out) {
    const long T = (long) gridDim.x * blockDim.x;
    long i = (long) blockIdx.x * blockDim.x + threadIdx.x;
    unsigned acc = 0;
    for (; i + (U - 1) * T < nvec; i += U * T) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(p + i + u * T);
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    for (; i < nvec; i += T) { u32x4 v = p[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) out[0] = acc;
}

// contiguous chunk per block: block b owns [b*per, (b+1)*per), waves own contiguous sub-chunks (the gemv shape: rows)
template <int U>
__global__ void __launch_bounds__(1024) k_chunk(const u32x4 * __restrict__ p, long nvec, unsigned * out) {
    const int nw = blockDim.x >> 6, wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long per_wave = (nvec / ((long) gridDim.x * nw)) & ~63L;
    const u32x4 * q = p + ((long) blockIdx.x * nw + wid) * per_wave + lane;
    unsigned acc = 0;
    long i = 0;
    for (; i + (U - 1) * 64 < per_wave; i += U * 64) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(q + i + u * 64);
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

// the decode GEMV's access shape: per wave and step 1 KiB of quants (16 B per lane) + 128 B of fp16 scales (2 B per lane),
// either from two planes (quants [n][16 B], scales [n][2 B]) or interleaved per 64 units ([1024 B quants | 128 B scales])
template <bool INTERLEAVED>
__global__ void __launch_bounds__(768) k_units(const unsigned char * __restrict__ q, const unsigned char * __restrict__ d, long nunits, unsigned * out) {
    const long nwaves = (long) gridDim.x * (blockDim.x >> 6);
    const long wave = (long) blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const long per_wave = (nunits / 64 / nwaves) * 64;                 // units per wave, contiguous
    unsigned acc = 0;
    for (long u0 = wave * per_wave; u0 + 256 <= (wave + 1) * per_wave; u0 += 256) {
        u32x4 v[4]; unsigned short s[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const long u = u0 + 64 * k + lane;
            if (INTERLEAVED) { const unsigned char * b = q + (u >> 6) * 1152; v[k] = __builtin_nontemporal_load((const u32x4 *)(b + (u & 63) * 16)); s[k] = *(const unsigned short *)(b + 1024 + (u & 63) * 2); }
            else             { v[k] = __builtin_nontemporal_load((const u32x4 *)(q + u * 16)); s[k] = *(const unsigned short *)(d + u * 2); }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) acc += (v[k].x ^ v[k].y ^ v[k].z ^ v[k].w) + s[k];
    }
    if (acc == 0x12345678u) out[0] = acc;
}

__global__ void k_empty(unsigned * out) { if (out == (unsigned *) 1) out[0] = 0; }

int main(int argc, char ** argv) {
    const long bytes = argc > 1 ? atol(argv[1]) : 58400000L;
    const int nbuf = 24, reps = 20;
    const long nvec = bytes / 16;
    std::vector<u32x4 *> bufs(nbuf);
    for (auto & b : bufs) { CK(hipMalloc(&b, nvec * 16)); CK(hipMemset(b, 1, nvec * 16)); }
    unsigned * out; CK(hipMalloc(&out, 64));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char * name, auto launch) {
        for (int i = 0; i < nbuf; ++i) launch(bufs[i]);
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int r = 0; r < reps; ++r) for (int i = 0; i < nbuf; ++i) launch(bufs[i]);
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / (reps * nbuf);
        printf("%-34s %8.2f us/launch  %7.1f GB/s\n", name, us, bytes / us / 1e3);
    };
    printf("bytes per launch %ld, %d distinct buffers (%.1f GB cycled)\n", bytes, nbuf, nbuf * bytes / 1e9);
    timeit("empty kernel", [&](u32x4 *) { hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, st, out); });
#define LIN(U, G, B) { char n[64]; snprintf(n, 64, "linear U=%d grid=%d x %d", U, G, B); timeit(n, [&](u32x4 * b) { hipLaunchKernelGGL((k_linear<U>), dim3(G), dim3(B), 0, st, b, nvec, out); }); }
#define CHK(U, G, B) { char n[64]; snprintf(n, 64, "chunk  U=%d grid=%d x %d", U, G, B); timeit(n, [&](u32x4 * b) { hipLaunchKernelGGL((k_chunk<U>), dim3(G), dim3(B), 0, st, b, nvec, out); }); }
    if (argc > 2 && argv[2][0] == 'u') {   // two planes vs interleaved scales
        const long nunits = bytes / 18;
        std::vector<unsigned char *> dp(nbuf);
        for (auto & b : dp) { CK(hipMalloc(&b, nunits * 2 + 4096)); CK(hipMemset(b, 1, nunits * 2 + 4096)); }
        int bi = 0;
        for (int g : {239, 256}) {
            char n[96];
            snprintf(n, 96, "two planes   grid=%d x 768", g);
            timeit(n, [&](u32x4 * b) { hipLaunchKernelGGL((k_units<false>), dim3(g), dim3(768), 0, st, (const unsigned char *) b, dp[bi++ % nbuf], nunits, out); });
            snprintf(n, 96, "interleaved  grid=%d x 768", g);
            timeit(n, [&](u32x4 * b) { hipLaunchKernelGGL((k_units<true>), dim3(g), dim3(768), 0, st, (const unsigned char *) b, (const unsigned char *) nullptr, nunits * 16 / 18, out); });
        }
        return 0;
    }
    if (argc > 2 && argv[2][0] == 'm') {   // Infinity-Cache (MALL) residency: the SAME 58 MB again right after a read of it, with
                                           // 0 / 58 / 117 / 234 MB of other traffic in between
        auto pair = [&](const char * name, int between) {
            float tot = 0; const int R = 20;
            for (int r = 0; r < R; ++r) {
                for (int i = 0; i < 8; ++i) hipLaunchKernelGGL((k_linear<8>), dim3(256), dim3(768), 0, st, bufs[8 + i], nvec, out);   // flush
                hipLaunchKernelGGL((k_linear<8>), dim3(256), dim3(768), 0, st, bufs[0], nvec, out);
                for (int i = 0; i < between; ++i) hipLaunchKernelGGL((k_linear<8>), dim3(256), dim3(768), 0, st, bufs[1 + i], nvec, out);
                CK(hipEventRecord(e0, st));
                hipLaunchKernelGGL((k_linear<8>), dim3(256), dim3(768), 0, st, bufs[0], nvec, out);
                CK(hipEventRecord(e1, st));
                CK(hipStreamSynchronize(st));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); tot += ms;
            }
            printf("%-40s %8.2f us (event-bracketed single launch)\n", name, tot * 1e3 / R);
        };
        pair("re-read, nothing in between", 0); pair("re-read, 58 MB in between", 1); pair("re-read, 117 MB in between", 2);
        pair("re-read, 234 MB in between", 4); pair("re-read, 350 MB in between", 6);
        return 0;
    }
    if (argc > 2) {     // CU-count sweep: how many streaming CUs does the chip need?
        LIN(8, 128, 768) LIN(8, 160, 768) LIN(8, 190, 768) LIN(8, 207, 704) LIN(8, 228, 640) LIN(8, 240, 640) LIN(8, 256, 640) LIN(8, 256, 768)
        LIN(4, 190, 768) LIN(4, 228, 640) LIN(16, 190, 768) LIN(12, 228, 640)
        return 0;
    }
    LIN(4, 256, 1024) LIN(8, 256, 1024) LIN(4, 512, 512) LIN(8, 512, 512) LIN(4, 1024, 256) LIN(8, 1024, 256) LIN(4, 2048, 256) LIN(8, 2048, 256) LIN(2, 4096, 256) LIN(4, 4096, 256)
    LIN(8, 256, 768) LIN(16, 256, 768) LIN(8, 256, 512) LIN(16, 256, 512) LIN(16, 256, 256) LIN(1, 8192, 256) LIN(1, 16384, 256)
    CHK(4, 256, 768) CHK(8, 256, 768) CHK(16, 256, 768) CHK(8, 256, 1024) CHK(8, 512, 512) CHK(8, 1024, 256) CHK(4, 2048, 256) CHK(8, 714, 256) CHK(8, 239, 768)
    return 0;
}
