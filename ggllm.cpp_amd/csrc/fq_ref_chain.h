// fq_ref_chain.h -- the reference's LEFT-TO-RIGHT f32 sum over a row's per-block terms, at the fast decode kernels' speed (round 6).
//
// The scalar build of the reference adds a row's block terms one after the other into one float (`sumf += sumi*d_w*d_x`, ggml.c:2591-2609 for Q4_0;
// 2719-2735 Q4_1; 2951-2972 Q5_0; 3207-3228 Q5_1; 3317-3329 Q8_0; the N = 1 caller is ggml.c:11484-11516). The fast mat-vec kernels produce exactly those
// terms -- one per lane-unit, fq_units.h -- but add them per lane and then across the wave. In the FAST REFERENCE ORDER (ggml_hip_reference_order(2)) the
// lanes drop their terms into an LDS strip [row][block] instead, and a wave whose LANES ARE ROWS adds each row's strip left to right: the association of
// the reference, 64 rows per dependent chain. Only the last step of the arithmetic is serial; the weight stream, the integer dots and the per-block f32
// expressions are the fast kernels'.
//
// Strip geometry: a row's terms are `stride` floats apart, stride = 4 * odd >= nblk, so that the 16-byte reads of consecutive lanes (rows) fall into
// different LDS banks and every row starts 16-byte aligned.
#pragma once
#include "fq_device.h"

__host__ __device__ inline unsigned fq_ref_strip_stride(int nblk) { unsigned q = ((unsigned) nblk + 3u) >> 2; q |= 1u; return 4u * q; }

#if defined(__HIPCC__)
// one unit's f32 term: into the lane's partial sum (default order) or, REF, into the row's strip at the block's index (sa[r] = LDS byte address of row r's
// strip). A DS store: it stays in the wave's LDS queue ahead of the counter add that reports the row to the summing wave.
// uc = the unit's index CLAMPED to the row's last unit: a lane beyond the row's end (ok == false) holds the clamped re-read of the last unit and has computed
// exactly that unit's term, so in REF it stores unconditionally (the same value to the same word: no branch around every store); default order: adds 0
template <bool REF, int R>
__device__ __forceinline__ void fq_emit_term(float (&acc)[R], const unsigned * sa, int r, int uc, bool ok, float v) {
    // (an LDS-address-space store: a ds_write the compiler may schedule among the dots, never a flat store -- a flat access would leave the LDS queue's order)
    if constexpr (REF) *(__attribute__((address_space(3))) float *)(uintptr_t)(sa[r] + 4u * (unsigned) uc) = v;
    else acc[r] += ok ? v : 0.0f;
}
// s + row[0] + row[1] + ... + row[n - 1], strictly in this order (-ffp-contract=off, no reassociation). row: LDS, 16-byte aligned. The next 16 terms are
// requested before the current 16 are added: the chain runs at the latency of a dependent v_add_f32, not at that of the LDS.
// FQ_REF_CHAIN_PRIO (default 1): the chain's wave raises its issue priority for the duration (s_setprio 3) -- a dependent add should wait for nobody but its predecessor, and the
// other waves of its SIMD are in the middle of their dots. A/B/A/B on one box: 940.9 / 941.4 against 930.5 / 932.0 tok/s in the fast reference order (profiles/r06t_ab_chain_prio.txt)
#ifndef FQ_REF_CHAIN_PRIO
#define FQ_REF_CHAIN_PRIO 1
#endif
__device__ __forceinline__ float fq_ref_chain_(const float * __restrict__ row, int n, float s);
__device__ __forceinline__ float fq_ref_chain(const float * __restrict__ row, int n, float s) {
#if FQ_REF_CHAIN_PRIO
    __builtin_amdgcn_s_setprio(3);
    const float r = fq_ref_chain_(row, n, s);
    __builtin_amdgcn_s_setprio(0);
    return r;
#else
    return fq_ref_chain_(row, n, s);
#endif
}
__device__ __forceinline__ float fq_ref_chain_(const float * __restrict__ row, int n, float s) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4 * r4 = (const f4 *) __builtin_assume_aligned(row, 16);
    int b = 0;
    if (n >= 16) {
        f4 a0 = r4[0], a1 = r4[1], a2 = r4[2], a3 = r4[3];
        for (b = 16; b + 16 <= n; b += 16) {
            const f4 n0 = r4[(b >> 2) + 0], n1 = r4[(b >> 2) + 1], n2 = r4[(b >> 2) + 2], n3 = r4[(b >> 2) + 3];
            s += a0.x; s += a0.y; s += a0.z; s += a0.w; s += a1.x; s += a1.y; s += a1.z; s += a1.w;
            s += a2.x; s += a2.y; s += a2.z; s += a2.w; s += a3.x; s += a3.y; s += a3.z; s += a3.w;
            a0 = n0; a1 = n1; a2 = n2; a3 = n3;
        }
        s += a0.x; s += a0.y; s += a0.z; s += a0.w; s += a1.x; s += a1.y; s += a1.z; s += a1.w;
        s += a2.x; s += a2.y; s += a2.z; s += a2.w; s += a3.x; s += a3.y; s += a3.z; s += a3.w;
    }
    for (; b + 4 <= n; b += 4) { const f4 a = r4[b >> 2]; s += a.x; s += a.y; s += a.z; s += a.w; }
    for (; b < n; ++b) s += row[b];
    return s;
}
#endif
