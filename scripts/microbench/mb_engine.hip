// micro-benchmark (tuning aid, not part of the library): the weight-streaming ENGINE a persistent decode kernel would be
// built on (MI355X_MICROARCH.md "ldsdma-fill" / "nt-weights" rows). One workgroup per CU:
//   NL loader waves   stream the workgroup's contiguous byte range of a Q4_0 matrix (device layout of fq_types.h, 2560-byte
//                     rows of K = 4544) into a ring of 16 KiB LDS slots with global_load_lds_dwordx4 (1 KiB per
//                     wave-instruction, no VGPRs held), run ahead of the consumers by up to the ring's depth
//   NC consumer waves take the rows round-robin, wait until a row's last byte has landed, run the REAL Q4_0 x Q8_0 row dot
//                     (fq_units.h arithmetic, lane l = units l, l+64, ..; DPP butterfly) out of LDS, release ring space
// against the same rows through the register-streaming form the current kernels use (mode 1). Both write every row's dot;
// the two result arrays must be bit-identical, and a few rows are checked against a host computation.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I ../../ggllm.cpp_amd/csrc mb_engine.hip -o mb_engine
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "fq_device.h"
#include "fq_units.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int K = 4544, NBLK = K / 32, ROW = 2560;          // Falcon-7B row: 142 blocks, 2556 bytes padded to 2560
constexpr int SLOT = 16384, PIECES = 16;
constexpr int IMG = K + NBLK * 4 * 2;                        // Q8_0 image: qs | d f32 | isum i32

// ---- synthetic weights in the device layout: random quants, fp16 scales in [2^-10, 2^-9)
__device__ __forceinline__ unsigned hash32(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
__global__ void k_fill(unsigned * w, long nwords) {
    for (long i = (long) blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += (long) gridDim.x * blockDim.x) {
        const int o = (int)((i * 4) % ROW);
        const bool scale = (o >= 1024 && o < 1152) || (o >= 2176 && o < 2304) || (o >= 2528);
        const unsigned h = hash32((unsigned) i * 2654435761u + 12345u);
        w[i] = scale ? (0x14001400u | (h & 0x03ff03ffu)) : h;
    }
}

template <bool NT>
__device__ __forceinline__ void glds16(const void * gsrc, unsigned lds_dst) {
    unsigned keep;
    if (NT) asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
    else    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// control words live in LDS and are touched with explicit DS instructions only: a flat (generic-pointer) access would make
// hipcc wait for vmcnt(0) -- i.e. drain the loader's in-flight DMA -- at every poll
__device__ __forceinline__ unsigned lds_ld(unsigned addr) { unsigned v; asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(addr) : "memory"); return v; }
__device__ __forceinline__ void lds_st(unsigned addr, unsigned v) { asm volatile("ds_write_b32 %0, %1" :: "v"(addr), "v"(v) : "memory"); }

// one row's dot out of a ring in LDS; `off` = the row's byte offset in the stream, ring addressed modulo RING
template <int RING>
__device__ __forceinline__ float row_dot_lds(const uint8_t * ring, long off, const fq_actcol & col, int lane) {
    float acc = 0.0f;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        const int u = p * 64 + lane;
        const bool ok = u < NBLK;
        const int uc = ok ? u : NBLK - 1;
        const int c = uc >> 6, j = uc & 63, nbc = c < 2 ? 64 : NBLK - 128;
        const unsigned cb = (unsigned)(off + c * 1152);
        fq_unit_regs r{};
        r.q  = *(const fq_u4 *)(ring + ((cb + j * 16) & (RING - 1)));
        r.dm = *(const uint16_t *)(ring + ((cb + nbc * 16 + j * 2) & (RING - 1)));
        const float v = fq_unit<FQ_Q4_0>::dot(r, col, uc);
        acc += ok ? v : 0.0f;
    }
    return wave_sum(acc);
}

struct eng_args {
    const uint8_t * w; long chunk_bytes; int rows_per_wg; int repeat; const uint8_t * image; float * out; unsigned * err; long matrix_stride;   // > 0: repeat r reads chunk w of matrix r
};

// LDS: [ring RING][image IMG, 16-aligned][ctl: landed[4], done_row[32], abort]
template <int NC, int NL, int NSLOT, bool NT>
__global__ void __launch_bounds__(64 * (NC + NL)) k_engine(eng_args a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr int RING = NSLOT * SLOT;
    uint8_t * ring = smem;
    uint8_t * img = smem + RING;
    const unsigned landed = (unsigned)(uintptr_t) smem + RING + ((IMG + 15) & ~15);       // LDS byte addresses of the control words
    const unsigned done_row = landed + 16, abortw = done_row + 128;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    for (int i = tid; i < IMG / 4; i += blockDim.x) ((unsigned *) img)[i] = ((const unsigned *) a.image)[i];
    if (tid < 4) lds_st(landed + 4 * tid, 0);
    if (tid < 32) lds_st(done_row + 4 * tid, tid < NC ? (unsigned) tid : 0xFFFFFFFFu);
    if (tid == 0) lds_st(abortw, 0);
    __syncthreads();
    const long total_rows = (long) a.rows_per_wg * a.repeat;
    const long total_slots = (total_rows * ROW + SLOT - 1) / SLOT;
    const uint8_t * chunk = a.w + (long) blockIdx.x * a.chunk_bytes;
    if (wid < NL) {
        // ------------------------------------------------------------------ loader
        const unsigned ring_lds = (unsigned)(uintptr_t) ring;
        long k = 0;                                                       // own slots issued
        long pos = (long) wid * SLOT;                                     // source offset inside the chunk (a multiple of SLOT long)
        long mat = 0;
        for (long s = wid; s < total_slots; s += NL, ++k, pos += (long) NL * SLOT) {
            if (pos >= a.chunk_bytes) { pos -= a.chunk_bytes; mat += a.matrix_stride; }
            if (s >= NSLOT) {                                             // slot s - NSLOT must have been consumed
                const long need = (s - NSLOT + 1) * (long) SLOT;
                bool stop = false;
                for (unsigned spins = 0;; ++spins) {
                    unsigned v = lds_ld(done_row + 4 * (lane < NC ? lane : 0));
                    v = __builtin_amdgcn_readfirstlane(wave_reduce((int) v, [](int x, int y) { return (unsigned) x < (unsigned) y ? x : y; }));
                    if ((long) v * ROW >= need) break;
                    if (spins == 0) {                                     // ring full: nothing more can be issued, so let everything
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // in flight land and report it (also rules out waiting on
                        if (lane == 0) lds_st(landed + 4 * wid, (unsigned) k);   // a slot whose report needs a later issue)
                    }
                    if (__builtin_amdgcn_readfirstlane(lds_ld(abortw))) { stop = true; break; }
                    if (spins > (1u << 22)) { if (lane == 0) { *a.err = 1; lds_st(abortw, 1); } stop = true; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
                if (stop) break;
            }
            const unsigned dst = ring_lds + (unsigned)((s & (NSLOT - 1)) * SLOT);
#pragma unroll
            for (int p = 0; p < PIECES; ++p) glds16<NT>(chunk + mat + pos + p * 1024 + lane * 16, __builtin_amdgcn_readfirstlane(dst + p * 1024));
            if (k >= 2) {                                                 // own slots 0 .. k-2 have landed
                asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
                if (lane == 0) lds_st(landed + 4 * wid, (unsigned)(k - 1));
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) lds_st(landed + 4 * wid, (unsigned) k);
        return;
    }
    // ---------------------------------------------------------------------- consumer
    const int c = wid - NL;
    const fq_actcol col = { (const int8_t *) img, (const float *)(img + K), (const void *)(img + K + NBLK * 4) };
    int rin = c % a.rows_per_wg;                                          // row index inside the chunk
    for (long row = c; row < total_rows; row += NC, rin += NC) {
        if (rin >= a.rows_per_wg) rin -= a.rows_per_wg;
        const long off = row * ROW;
        const unsigned slot = (unsigned)((off + ROW - 1) >> 14);          // the slot holding the row's last byte
        const unsigned need = slot / NL + 1;
        const int lj = (int)(slot % NL);
        bool stop = false;
        for (unsigned spins = 0;; ++spins) {
            if (__builtin_amdgcn_readfirstlane(lds_ld(landed + 4 * lj)) >= need) break;
            if (__builtin_amdgcn_readfirstlane(lds_ld(abortw))) { stop = true; break; }
            if (spins > (1u << 22)) { if (lane == 0) { *a.err = 2; lds_st(abortw, 1); } stop = true; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        if (stop) break;
        const float v = row_dot_lds<RING>(ring, off, col, lane);
        if (lane == 0) { a.out[(long) blockIdx.x * a.rows_per_wg + rin] = v; lds_st(done_row + 4 * c, (unsigned)(row + NC)); }
    }
}

// the register-streaming form (what k_gemv_ln's later passes do): NW waves, R rows per wave at a time, all of their unit
// columns requested with non-temporal global loads before the first dot
template <int NW, int R>
__global__ void __launch_bounds__(64 * NW) k_regs(eng_args a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t * img = smem;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    for (int i = tid; i < IMG / 4; i += blockDim.x) ((unsigned *) img)[i] = ((const unsigned *) a.image)[i];
    __syncthreads();
    const fq_actcol col = { (const int8_t *) img, (const float *)(img + K), (const void *)(img + K + NBLK * 4) };
    const long total_rows = (long) a.rows_per_wg * a.repeat;
    const uint8_t * chunk = a.w + (long) blockIdx.x * a.chunk_bytes;
    int rin0 = (wid * R) % a.rows_per_wg;                                 // rows_per_wg is a multiple of R: a group never wraps
    for (long row0 = (long) wid * R; row0 < total_rows; row0 += (long) NW * R, rin0 += NW * R) {
        if (rin0 >= a.rows_per_wg) rin0 -= a.rows_per_wg;
        fq_unit_regs regs[3][R];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const fq_wrow wr{ chunk + (long)(rin0 + r) * ROW, NBLK };
                regs[p][r] = fq_unit_load_col<FQ_Q4_0>(wr, p, lane, NBLK);
            }
        }
        float acc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = 0.0f;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const int u = p * 64 + lane; const bool ok = u < NBLK; const int uc = ok ? u : NBLK - 1;
#pragma unroll
            for (int r = 0; r < R; ++r) { const float v = fq_unit<FQ_Q4_0>::dot(regs[p][r], col, uc); acc[r] += ok ? v : 0.0f; }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float v = wave_sum(acc[r]);
            if (lane == 0) a.out[(long) blockIdx.x * a.rows_per_wg + rin0 + r] = v;
        }
    }
}

static float h2f_host(uint16_t h) { return fq_h2f_ref(h); }

int main(int argc, char ** argv) {
    int ncu = 256;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    ncu = prop.multiProcessorCount;
    const int rows_per_wg = argc > 1 ? atoi(argv[1]) : 1632;            // multiple of 32 -> chunk is whole slots
    const int repeat = argc > 2 ? atoi(argv[2]) : 4;
    const long chunk = (long) rows_per_wg * ROW;
    const bool matrices = argc > 3 && atoi(argv[3]) != 0;           // engine-like: `repeat` matrices of ncu chunks each
    const long total = chunk * ncu * (matrices ? repeat : 1);
    printf("%s: %d CUs; %d rows (%.2f MB) per workgroup x %d repeats; %.2f GB unique, %.2f GB streamed per launch\n", prop.name, ncu, rows_per_wg,
           chunk / 1e6, repeat, total / 1e9, total * (double) repeat / 1e9);
    uint8_t * w; CK(hipMalloc(&w, total + 65536));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (unsigned *) w, (total + 65536) / 4);
    // activation image
    std::vector<uint8_t> img(IMG);
    srand(7);
    for (int b = 0; b < NBLK; ++b) {
        int s = 0;
        for (int j = 0; j < 32; ++j) { const int q = rand() % 255 - 127; img[b * 32 + j] = (uint8_t)(int8_t) q; s += q; }
        const float d = 0.01f + 0.0001f * (rand() % 100);
        memcpy(&img[K + 4 * b], &d, 4); memcpy(&img[K + NBLK * 4 + 4 * b], &s, 4);
    }
    uint8_t * dimg; CK(hipMalloc(&dimg, IMG + 64)); CK(hipMemcpy(dimg, img.data(), IMG, hipMemcpyHostToDevice));
    float * out0, * out1; const long nout = (long) rows_per_wg * ncu;
    CK(hipMalloc(&out0, nout * 4)); CK(hipMalloc(&out1, nout * 4));
    unsigned * err; CK(hipMalloc(&err, 64)); CK(hipMemset(err, 0, 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipDeviceSynchronize());

    std::vector<float> ref(nout), got(nout);
    auto run = [&](const char * name, auto launch, float * out, int grid) {
        CK(hipMemset(out, 0xff, nout * 4));
        launch(out);                                                   // warm
        CK(hipDeviceSynchronize());
        float best = 1e30f, sum = 0;
        const int reps = 5;
        for (int r = 0; r < reps; ++r) {
            CK(hipEventRecord(e0, 0)); launch(out); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); sum += ms; if (ms < best) best = ms;
        }
        unsigned eh = 0; CK(hipMemcpy(&eh, err, 4, hipMemcpyDeviceToHost));
        const double bytes = (double) chunk * grid * repeat * (2556.0 / 2560.0);
        printf("%-44s grid %3d  best %8.1f us  avg %8.1f us  %7.1f GB/s (best)  err %u\n", name, grid, best * 1e3, sum / reps * 1e3, bytes / (best * 1e-3) / 1e9, eh);
        fflush(stdout);
        if (eh) CK(hipMemset(err, 0, 64));
    };
    eng_args a{ w, chunk, rows_per_wg, repeat, dimg, nullptr, err, matrices ? chunk * ncu : 0 };

    // ---- reference: register streaming
#define REGS(NW, R, G) { char n[96]; snprintf(n, 96, "registers  %2d waves x %d rows", NW, R); \
        run(n, [&](float * o) { eng_args b = a; b.out = o; hipLaunchKernelGGL((k_regs<NW, R>), dim3(G), dim3(64 * NW), IMG + 64, 0, b); }, out0, G); }
    REGS(12, 4, ncu) REGS(12, 2, ncu) REGS(16, 2, ncu) REGS(16, 4, ncu) REGS(8, 4, ncu)
    REGS(12, 4, ncu)
    CK(hipMemcpy(ref.data(), out0, nout * 4, hipMemcpyDeviceToHost));
    // host check of a few rows
    {
        int bad = 0;
        std::vector<uint8_t> rowb(ROW);
        for (long r : { 0L, 1L, (long) rows_per_wg - 1, (long) rows_per_wg * 17 + 5, nout - 1 }) {
            CK(hipMemcpy(rowb.data(), w + r * ROW, ROW, hipMemcpyDeviceToHost));
            float lanes[64] = {0};
            for (int u = 0; u < NBLK; ++u) {
                const int c = u >> 6, j = u & 63, nbc = c < 2 ? 64 : NBLK - 128;
                const uint8_t * q = &rowb[c * 1152 + j * 16]; uint16_t dh; memcpy(&dh, &rowb[c * 1152 + nbc * 16 + j * 2], 2);
                int s = 0;
                for (int i = 0; i < 16; ++i) s += (q[i] & 15) * (int8_t) img[u * 32 + i] + (q[i] >> 4) * (int8_t) img[u * 32 + 16 + i];
                int isum; memcpy(&isum, &img[K + NBLK * 4 + 4 * u], 4); float d; memcpy(&d, &img[K + 4 * u], 4);
                s -= 8 * isum;
                lanes[u & 63] += ((float) s * h2f_host(dh)) * d;
            }
            for (int o = 1; o < 64; o <<= 1) { float t[64]; for (int l = 0; l < 64; ++l) t[l] = lanes[l] + lanes[l ^ o]; memcpy(lanes, t, sizeof t); }
            if (memcmp(&lanes[0], &ref[r], 4)) { ++bad; printf("  host check row %ld: host %.9g device %.9g\n", r, lanes[0], ref[r]); }
        }
        printf("host check of 5 rows (register form): %s\n", bad ? "MISMATCH" : "ok");
    }

    // ---- the engine
    auto engine_cmp = [&](const char * n) {
        CK(hipMemcpy(got.data(), out1, nout * 4, hipMemcpyDeviceToHost));
        long bad = 0; for (long i = 0; i < nout; ++i) bad += memcmp(&got[i], &ref[i], 4) != 0;
        if (bad) printf("  !! %s: %ld of %ld rows differ from the register form\n", n, bad, nout);
    };
#define ENG(NC, NL, NSLOT, NT, G) { char n[96]; snprintf(n, 96, "engine %2d consumers %d loaders %3d KiB ring %s", NC, NL, NSLOT * 16, NT ? "nt" : "  "); \
        const size_t lds = (size_t) NSLOT * SLOT + ((IMG + 15) & ~15) + 256; \
        CK(hipFuncSetAttribute((const void *) k_engine<NC, NL, NSLOT, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds)); \
        run(n, [&](float * o) { eng_args b = a; b.out = o; hipLaunchKernelGGL((k_engine<NC, NL, NSLOT, NT>), dim3(G), dim3(64 * (NC + NL)), lds, 0, b); }, out1, G); \
        if (G == ncu && !matrices) engine_cmp(n); }
    ENG(11, 1, 8, true, ncu) ENG(11, 1, 8, false, ncu) ENG(11, 1, 4, true, ncu) ENG(7, 1, 8, true, ncu) ENG(15, 1, 8, true, ncu) ENG(3, 1, 8, true, ncu)
    ENG(10, 2, 8, true, ncu) ENG(14, 2, 8, true, ncu) ENG(6, 2, 8, true, ncu) ENG(10, 2, 4, true, ncu)
    ENG(11, 1, 8, true, 224) ENG(11, 1, 8, true, 192) ENG(10, 2, 8, true, 224) ENG(10, 2, 8, true, 192) ENG(10, 2, 8, true, 128)
    return 0;
}
