// fq_skinny_dev.h -- device helpers shared by the small-batch mat-mul files (kernels_gemm_skinny.hip: the legacy formats; kernels_gemm_skinny_k.hip:
// the k-quants): LDS-DMA of 16-byte pieces, vmcnt pacing, the tile constants.
#pragma once
#include "fq_block_dev.h"
#include "kernels.h"
#include "hip_context.h"

typedef int  sk_v4i __attribute__((ext_vector_type(4)));
typedef int  sk_v2i __attribute__((ext_vector_type(2)));

namespace {


constexpr int SK_TM = 32;                  // weight rows per workgroup (two 16-row tiles)
constexpr int SK_TN = 16;                  // columns
constexpr int SK_GS = 32;                  // blocks (32-element groups) per stage


__device__ __forceinline__ void sk_dma(const void * base, unsigned voff, unsigned lds_dst) {      // active lanes: 16 B at base + voff -> LDS lds_dst + 16 lane
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(base), "s"(lds_dst) : "memory");
}
// sixteen rows in one statement: row i's active lanes read 16 B at base + v[i], LDS destination ml + i * rowb (+ 16 lane); ml is advanced
struct sk_voff16 { unsigned v[16]; };
#define SK_DMA1(P) "s_mov_b32 m0, %[ml]\n\ts_add_u32 %[ml], %[ml], %[rowb]\n\tglobal_load_lds_dwordx4 %[v" #P "], %[base] nt\n\t"
__device__ __forceinline__ void sk_dma16(const void * base, const sk_voff16 & o, unsigned & ml, unsigned rowb) {
    unsigned keep;
    asm volatile("s_mov_b32 %[keep], m0\n\ts_nop 4\n\t"
                 SK_DMA1(0) SK_DMA1(1) SK_DMA1(2) SK_DMA1(3) SK_DMA1(4) SK_DMA1(5) SK_DMA1(6) SK_DMA1(7)
                 SK_DMA1(8) SK_DMA1(9) SK_DMA1(10) SK_DMA1(11) SK_DMA1(12) SK_DMA1(13) SK_DMA1(14) SK_DMA1(15)
                 "s_mov_b32 m0, %[keep]"
                 : [keep] "=&s"(keep), [ml] "+s"(ml)
                 : [v0] "v"(o.v[0]), [v1] "v"(o.v[1]), [v2] "v"(o.v[2]), [v3] "v"(o.v[3]), [v4] "v"(o.v[4]), [v5] "v"(o.v[5]), [v6] "v"(o.v[6]), [v7] "v"(o.v[7]),
                   [v8] "v"(o.v[8]), [v9] "v"(o.v[9]), [v10] "v"(o.v[10]), [v11] "v"(o.v[11]), [v12] "v"(o.v[12]), [v13] "v"(o.v[13]), [v14] "v"(o.v[14]), [v15] "v"(o.v[15]),
                   [base] "s"(base), [rowb] "s"(rowb)
                 : "memory", "scc");
}
__device__ __forceinline__ unsigned sk_lds(const void * p) { return (unsigned)(uintptr_t) p; }
// s_waitcnt vmcnt(n) for a wave-uniform run-time n <= 15 (the immediate must be a constant)
__device__ __forceinline__ void sk_wait_vm_upto(int n) {
    switch (n) {
#define SK_WV(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
        SK_WV(1) SK_WV(2) SK_WV(3) SK_WV(4) SK_WV(5) SK_WV(6) SK_WV(7) SK_WV(8) SK_WV(9) SK_WV(10) SK_WV(11) SK_WV(12) SK_WV(13) SK_WV(14) SK_WV(15)
#undef SK_WV
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}
__device__ __forceinline__ const uint8_t * sk_uniform(const uint8_t * p) {
    const unsigned long long v = (unsigned long long)(uintptr_t) p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned) v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (const uint8_t *)(uintptr_t)(((unsigned long long) hi << 32) | lo);
}

constexpr int KS_TMAX = 8;                 // tiles (consumer waves) per workgroup of the self-paced forms

}   // namespace

// the partial sums of the self-paced forms: [segment][share][16 columns][mstride rows] -> dst = the segments' ((P0 + P1) + P2) + P3 added left to right, then the epilogue
// (kernels_gemm_skinny_k.hip)
void fq_launch_skinny_sum4(const float * part, int64_t N, int64_t M, float * dst, int64_t ldd, const fq_gemv_epi & ep, int64_t mstride, int nseg, hipStream_t st);
// x = (sum of part_d's segments + sum of part_w's) + x, each segment ((P0 + P1) + P2) + P3 (kernels_gemm_skinny_k.hip)
void fq_launch_skinny_sum4_out2(const float * part_d, int nseg_d, const float * part_w, int nseg_w, int64_t mstride, int64_t N, int64_t M, float * x, int64_t ldx, hipStream_t st);
// the k-quant forms (Q4_K / Q5_K, Q2_K / Q3_K, Q6_K at model widths): true = launched (kernels_gemm_skinny_k.hip)
bool fq_launch_gemm_skinny_kq(const fq_weight & w, const fq_act & act, int64_t N, float * dst, int64_t ldd, const fq_gemv_epi & ep, int S, hipStream_t st);
