// kernels_decode.hip -- fused single-token (N = 1) decode kernels for a Falcon block (gfx950, wave64).
//
// A decode step streams ~117 MB of Q4_0 weights per block in ~19 us at HBM speed, so every extra launch (~1.7 us of
// boundary, ~1.2 us before its first load can be issued, a latency-bound prologue) costs as much as megabytes of weights.
// The stand-alone kernels (layer norm, activation quantizer, rope, four GEMVs, attention, residual add) are folded into:
//
//   k_gemv_ln   rows [Wqkv | Wup]: one workgroup of 12 waves per CU; each re-derives the LayerNorm of the 18 KB residual
//               row and its Q8 image (registers -> LDS; cheaper than a launch), streams 96 weight rows, and finishes with
//               a plain store (QKV) or GELU + Q8 quantization straight into the next mat-vec's activation image.
//               The same kernel is the lm_head launch (ln_f + 65 024 rows + per-32-row argmax candidates).
//   k_attn_out  36 attention workgroups (RoPE of q and the new k, KV append, K.Q, soft_max, V.P, Q8 image; 2 heads each)
//               next to 190 mat-vec workgroups, x = (Wdown . q8(gelu(up)) + Wo . q8(att)) + x: Wdown streams while the
//               attention runs, its output crosses workgroups through tagged granules, the residual is added at the end.
//   k_attn_decode, k_gemv_out   the same two roles as separate launches (models whose grid does not fit the chip at once).
//   k_attn_out_ln               k_attn_out + the next block's k_gemv_ln as a second phase of one launch (measured slower).
//
// Arithmetic is the stand-alone kernels' arithmetic (same device functions, same per-lane unit order, same reductions):
// logits are bit-identical to the unfused path, which the tests check.
#include "fq_block_dev.h"
#include "fq_kdot.h"
#include "fq_attn_dev.h"
#include "fq_attn_decode_dev.h"
#include "fq_ref_chain.h"
#include "kernels.h"
#include "hip_context.h"
#include <hip/hip_ext.h>

// launch, handing the open profile bracket's events (if any) to the dispatch
#define FQ_LAUNCH_PROF(kern, grid, block, lds, st, ...) do { \
        hipEvent_t e0_ = nullptr, e1_ = nullptr; fq_prof_events(&e0_, &e1_); \
        if (e0_) hipExtLaunchKernelGGL(kern, grid, block, lds, st, e0_, e1_, 0, __VA_ARGS__); \
        else     hipLaunchKernelGGL(kern, grid, block, lds, st, __VA_ARGS__); } while (0)

#define FQ_STAMP(dbg, slot) do { if ((dbg) && threadIdx.x == 0) (dbg)[(size_t) blockIdx.x * 8 + (slot)] = (long long) wall_clock64(); } while (0)

template <int TYPE> struct act_of { static constexpr int value =
    (TYPE == FQ_Q4_1 || TYPE == FQ_Q5_1) ? FQ_Q8_1 : ((TYPE == FQ_Q4_0 || TYPE == FQ_Q5_0 || TYPE == FQ_Q8_0) ? FQ_Q8_0 : FQ_Q8_K); };

// ---- weight streaming helpers. Lane l owns units l, l+64, ... of each of R rows (fq_units.h). The first NPRE
// unit-columns are ISSUED before the workgroup's prologue (so HBM streams while the LayerNorm / quantizer runs),
// consumed after it; longer rows continue with a chunked loop. Per-lane accumulation order is u ascending on every
// path, so the result does not depend on NPRE / UNROLL.
template <int TYPE, int R>
__device__ __forceinline__ void rows_ptrs(const fq_weight & w, int64_t row0, fq_wrow (&rows)[R]) {
#pragma unroll
    for (int r = 0; r < R; ++r) { const int64_t row = row0 + r; rows[r] = fq_row<TYPE>(w, row < w.M ? row : w.M - 1); }
}
template <int TYPE, int R, int NPRE, int C0 = 0, int C1 = NPRE>            // unit columns [C0, C1) of the NPRE pre-issued ones
__device__ __forceinline__ void rows_issue(const fq_wrow (&rows)[R], int units, fq_unit_regs (&regs)[NPRE][R]) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int i = C0; i < C1; ++i) {
#pragma unroll
        for (int r = 0; r < R; ++r) regs[i][r] = fq_unit_load_col<TYPE>(rows[r], i, lane, units);
    }
}
// k-quants, rows of whole columns (units % 64 == 0: Falcon-40B / 180B's matrices behind K = 8192 / 32768): the lane sits in the same slot ju = lane of every
// column, so the unit dot is fq_kdot.h's (lane-constant index math hoisted, both sub-block scales decoded at once; the SAME f32 term, tests/test_units_host.py)
template <int TYPE> __device__ __forceinline__ bool kq_fast_units(int units) {
    if constexpr (fq_kdot<TYPE>::ok) return (units & 63) == 0; else return false;
}
// REF (both functions; legacy formats): every unit's f32 term goes to the strip of its row, sa[r], at the unit's = block's index (fq_ref_chain.h)
// col0: regs[i] holds unit column col0 + i of the rows (a caller that consumes the pre-issued columns one by one)
template <int TYPE, int R, int NPRE, bool REF = false>
__device__ __forceinline__ void rows_consume(const fq_unit_regs (&regs)[NPRE][R], int units, const fq_actcol & col, float (&acc)[R], const unsigned * sa = nullptr, const int col0 = 0) {
    const int lane = threadIdx.x & 63;
    static_assert(!REF || !fq_kdot<TYPE>::ok, "the fast reference order covers the legacy formats");
    if constexpr (fq_kdot<TYPE>::ok) {
        if (kq_fast_units<TYPE>(units)) {
            typedef fq_kdot<TYPE> KD;
            const typename KD::lane_t L = KD::lane_init(lane);
#pragma unroll
            for (int i = 0; i < NPRE; ++i) {
                const bool ok = 64 * i < units;                              // (wave-uniform; a pre-issued column beyond the row's end was clamped to its last one)
                const typename KD::act_t y = KD::act_load(col, ok ? i : (units >> 6) - 1, L);
#pragma unroll
                for (int r = 0; r < R; ++r) { const float v = KD::dot(KD::from_regs(regs[i][r]), y, L); acc[r] += ok ? v : 0.0f; }
            }
            return;
        }
    }
#pragma unroll
    for (int i = 0; i < NPRE; ++i) {
        const int u = (col0 + i) * 64 + lane; const bool ok = u < units; const int uc = ok ? u : units - 1;
#pragma unroll
        for (int r = 0; r < R; ++r) { const float v = fq_unit<TYPE>::dot(regs[i][r], col, uc); fq_emit_term<REF, R>(acc, sa, r, uc, ok, v); }
    }
}
template <int TYPE, int R, int UNROLL, bool REF = false>
__device__ __forceinline__ void rows_dot_from(const fq_wrow (&rows)[R], int units, int u_begin, const fq_actcol & col, float (&acc)[R], const unsigned * sa = nullptr) {
    const int lane = threadIdx.x & 63;
    static_assert(!REF || !fq_kdot<TYPE>::ok, "the fast reference order covers the legacy formats");
    if constexpr (fq_kdot<TYPE>::ok) {
        if (kq_fast_units<TYPE>(units)) {
            typedef fq_kdot<TYPE> KD;
            const typename KD::lane_t L = KD::lane_init(lane);
            const int ncol = units >> 6;
            for (int c0 = u_begin >> 6; c0 < ncol; c0 += UNROLL) {
                fq_unit_regs regs[UNROLL][R];
#pragma unroll
                for (int i = 0; i < UNROLL; ++i) {
#pragma unroll
                    for (int r = 0; r < R; ++r) regs[i][r] = fq_unit_load_col<TYPE>(rows[r], c0 + i, lane, units);
                }
#pragma unroll
                for (int i = 0; i < UNROLL; ++i) {
                    const bool ok = c0 + i < ncol;
                    const typename KD::act_t y = KD::act_load(col, ok ? c0 + i : ncol - 1, L);
#pragma unroll
                    for (int r = 0; r < R; ++r) { const float v = KD::dot(KD::from_regs(regs[i][r]), y, L); acc[r] += ok ? v : 0.0f; }
                }
            }
            return;
        }
    }
    for (int u0 = u_begin; u0 < units; u0 += 64 * UNROLL) {
        fq_unit_regs regs[UNROLL][R];
#pragma unroll
        for (int i = 0; i < UNROLL; ++i) {
#pragma unroll
            for (int r = 0; r < R; ++r) regs[i][r] = fq_unit_load_col<TYPE>(rows[r], (u0 >> 6) + i, lane, units);
        }
#pragma unroll
        for (int i = 0; i < UNROLL; ++i) {
            const int u = u0 + i * 64 + lane; const bool ok = u < units; const int uc = ok ? u : units - 1;
#pragma unroll
            for (int r = 0; r < R; ++r) { const float v = fq_unit<TYPE>::dot(regs[i][r], col, uc); fq_emit_term<REF, R>(acc, sa, r, uc, ok, v); }
        }
    }
}

// per-format shape of the fused GEMVs: R rows per pass and NPRE pre-issued unit columns, sized so that a 12-wave
// workgroup stays within 168 VGPRs (3 waves per SIMD; fq_unit_regs is 5 dwords for Q4_0 ... 12 for Q5_K)
template <int TYPE> struct decode_cfg {
    static constexpr bool four_bit = (TYPE == FQ_Q4_0 || TYPE == FQ_Q4_1 || TYPE == FQ_Q5_0 || TYPE == FQ_Q5_1);
    static constexpr int LN_R     = 4;                                  // rows per pass in k_gemv_ln
    static constexpr int LN_NPRE  = four_bit ? 3 : 2;                   // 4-wave workgroups (small models)
    static constexpr int LN_NPRE_BIG = 1;                               // 12-wave workgroups, see k_gemv_ln
    static constexpr int OUT_NPRE_D = four_bit ? 4 : 2;                 // k_gemv_out, down projection (R = 2)
    static constexpr int OUT_NPRE_O = four_bit ? 3 : 1;                 // k_gemv_out, attention projection
};

__device__ __forceinline__ fq_actcol actcol_at(const uint8_t * base, int act_type, int64_t K) {
    return { (const int8_t *) base, (const float *)(base + fq_act_d_off(act_type, K)), (const void *)(base + fq_act_aux_off(act_type, K)) };
}

// =============================================================================================== k_gemv_ln
// one workgroup = NW waves = RW*NW consecutive rows of one segment (NW = blockDim/64 and RW = 4 * a.npass rows per wave,
// chosen by the launcher so that the grid is about one workgroup per CU: the LayerNorm + Q8 prologue is then computed
// ~n_cu times per launch instead of once per 32 rows); wave w owns rows RW*w .. RW*w + RW - 1 (npass passes of 4)
// xs.gran != nullptr: the residual row arrives through a hand-off buffer of the same launch (k_attn_out_ln), see fq_block_dev.h
struct fq_xsrc { const unsigned long long * gran; unsigned epoch; unsigned * err; };

// REF = the fast reference order (fq_ref_chain.h; legacy formats): a wave leaves the unit terms of two passes (8 rows) in its own LDS strip and adds them with
// lanes 0..7 = rows, left to right as the reference's scalar build does (ggml.c:2591-2609) -- no other wave is involved, no barrier is added
template <int TYPE, int MAXT, bool REF = false>
__device__ __forceinline__ void gemv_ln_body(const fq_gemv_ln_args & a, const int bid, uint8_t * smem, const fq_xsrc xs) {
    constexpr int ACT = act_of<TYPE>::value;
    const int64_t E = a.E;
    const int sidx = (a.nseg > 1 && bid >= a.seg[1].block_begin) ? 1 : 0;
    const fq_gemv_ln_seg sg = sidx ? a.seg[1] : a.seg[0];       // whole-struct select: no runtime-indexed kernarg array
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, nw = blockDim.x >> 6;
    const int RW = 4 * a.npass;
    const int64_t row0 = (int64_t)(bid - sg.block_begin) * (RW * nw);

    // LDS: f32 row [E] | image | out rows (<= 384) | reduction scratch
    float   * rowf  = (float *) smem;
    uint8_t * image = smem + (((size_t) E * 4 + 15) & ~(size_t) 15);
    float   * out32 = (float *)(image + fq_act_col_bytes(ACT, E));
    double  * red   = (double *)(out32 + 384);

    constexpr int R = 4, NPRE = MAXT > 256 ? decode_cfg<TYPE>::LN_NPRE_BIG : decode_cfg<TYPE>::LN_NPRE;              // 8 rows per wave = two passes of 4
    const int units = (int)(E / fq_unit<TYPE>::ELEMS);
    // REF: this wave's strip, 8 row slots (slot = 4 (pass & 1) + row of the pass) of SW floats, behind the reduction scratch
    const unsigned SW = REF ? fq_ref_strip_stride(units) : 0u;
    float * const strip_w = (float *)(red + 32) + (size_t) wid * 8 * SW;
    unsigned sa[R];
    auto strip_pass = [&](int p) {
#pragma unroll
        for (int r = 0; r < R; ++r) sa[r] = (unsigned)(uintptr_t)(strip_w + (size_t)(4 * (p & 1) + r) * SW);
    };
    // the rows of passes p - 1 and p, summed in the reference's order by lanes 0..7 (the stores above are ahead of these reads in the wave's LDS queue)
    auto strip_sum = [&](int p) {
        const int sl = lane & 7;
        const float v = fq_ref_chain(strip_w + (size_t) sl * SW, units, 0.0f);
        if (lane < 8) out32[RW * wid + R * (p - 1 + (sl >> 2)) + (sl & 3)] = v;
    };
    // 1. the residual row's loads, 2. pass-0 weight loads, 3. LayerNorm + Q8 image while those stream, 4. dots of pass 0,
    //    5. pass 1 (its loads overlap other workgroups' dots)
    // the hand-off tag of the launch that follows (in the two-phase kernel: after the row has been read with the current one)
    if (!xs.gran && a.epoch_word && bid == 0 && tid == 0) { const unsigned e = *a.epoch_word + 1u; *a.epoch_word = e ? e : 1u; }
    if (a.rope_cur && bid == a.n_blocks - 1 && tid < 64) a.rope_cur[tid] = a.rope_cs[(int64_t)(*a.n_past_ptr) * 64 + tid];
    FQ_STAMP(a.dbg, 0);
    // LayerNorm + Q8 image in registers (NLN float4 of the row per thread) when the row fits, through LDS otherwise
    constexpr int NLN = MAXT > 256 ? 3 : 5;
    const bool in_regs = (E >> 2) <= (int64_t) NLN * blockDim.x;        // Falcon-7B (1136 float4) and 40B (2048) with 12 waves: yes
    fq_wrow rows0[R];
    fq_unit_regs pre0[NPRE][R];
    if (in_regs) {
        ln_row_regs<NLN> xr, wr, br;
        if (xs.gran) ln_regs_issue_wb(sg.ln_w, sg.ln_b, E, blockDim.x, wr, br);
        else         ln_regs_issue(a.x, sg.ln_w, sg.ln_b, E, blockDim.x, xr, wr, br);
        rows_ptrs<TYPE, R>(sg.w, row0 + RW * wid, rows0);
        if (xs.gran) {
            // two-phase kernel: the first unit column is requested BEFORE the row exists (it streams while the producers
            // finish), then the row is swept out of the hand-off buffer
            rows_issue<TYPE, R, NPRE>(rows0, units, pre0);
            ln_regs_sweep_x(xs.gran, xs.epoch, E, blockDim.x, xr, xs.err);
            if (a.epoch_word && bid == 0 && tid == 0) { const unsigned e = xs.epoch + 1u; *a.epoch_word = e ? e : 1u; }
        }
        ln_regs_stage1(xr, E, blockDim.x, red);                          // waits for the row only
        // only NPRE (12 waves: ONE) unit column per row is requested ahead of the LayerNorm (48 KB per CU): a CU keeps
        // only so many requests in flight, more makes the later waves' loads block at issue, and a wave that cannot
        // issue cannot reach the LN's barriers either (measured: +3 us with 3 columns; requesting the other columns
        // right after the last statistics barrier, ahead of the quantizer, moves the image 1.4 us later and the end of
        // the kernel nowhere)
        if (!xs.gran) rows_issue<TYPE, R, NPRE>(rows0, units, pre0);
        FQ_STAMP(a.dbg, 1);
        __syncthreads();
        ln_regs_stage2(xr, E, blockDim.x, red);
        __syncthreads();
        ln_regs_stage3<ACT>(xr, wr, br, E, blockDim.x, act_image_at(image, ACT, E), red);
        FQ_STAMP(a.dbg, 2);
        __syncthreads();
    } else {
        ln_row_regs<8> xr;
        layer_norm_issue(a.x, E, xr);
        rows_ptrs<TYPE, R>(sg.w, row0 + RW * wid, rows0);
        rows_issue<TYPE, R, NPRE>(rows0, units, pre0);
        FQ_STAMP(a.dbg, 1);
        layer_norm_finish(xr, a.x, E, sg.ln_w, sg.ln_b, rowf, red);      // identical to k_layer_norm
        FQ_STAMP(a.dbg, 2);
        quantize_row_block<ACT>(rowf, E, act_image_at(image, ACT, E));    // identical to k_quantize_q8 / q8K
        __syncthreads();
    }
    FQ_STAMP(a.dbg, 3);
    const fq_actcol col = actcol_at(image, ACT, E);
    {
        float acc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = 0.0f;
        if constexpr (REF) strip_pass(0);
        rows_consume<TYPE, R, NPRE, REF>(pre0, units, col, acc, sa);
        rows_dot_from<TYPE, R, 2, REF>(rows0, units, 64 * NPRE, col, acc, sa);
        if constexpr (!REF) {
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r] = wave_sum(acc[r]);
            if (lane == 0) {
#pragma unroll
                for (int r = 0; r < R; ++r) out32[RW * wid + r] = acc[r];
            }
        }
        FQ_STAMP(a.dbg, 4);
    }
    for (int p = 1; p < a.npass; ++p) {                // later passes: all unit columns of the 4 rows in flight at once
        fq_wrow rowsp[R];
        rows_ptrs<TYPE, R>(sg.w, row0 + RW * wid + R * p, rowsp);
        float acc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = 0.0f;
        // (rows of <= 192 units -- Falcon-7B: 142 -- in one trip of LN_NPRE columns; longer rows two columns at a time, so
        // that the last trip does not re-request clamped columns)
        if constexpr (REF) strip_pass(p);
        if (units <= 64 * decode_cfg<TYPE>::LN_NPRE) rows_dot_from<TYPE, R, decode_cfg<TYPE>::LN_NPRE, REF>(rowsp, units, 0, col, acc, sa);
        else                                         rows_dot_from<TYPE, R, 2, REF>(rowsp, units, 0, col, acc, sa);
        if constexpr (REF) { if (p & 1) strip_sum(p); }
        else {
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r] = wave_sum(acc[r]);
            if (lane == 0) {
#pragma unroll
                for (int r = 0; r < R; ++r) out32[RW * wid + R * p + r] = acc[r];
            }
        }
    }
    FQ_STAMP(a.dbg, 5);
    if (a.dbg && lane == 0) atomicMax((unsigned long long *)(a.dbg + (size_t) blockIdx.x * 8 + 6), (unsigned long long) wall_clock64());   // slowest wave
    __syncthreads();
    for (int grp = wid; grp < (RW * nw) / 32; grp += nw) {     // a wave finishes rows [32*grp, 32*grp+32) (lanes 32..63 mirror 0..31)
        const int j = lane & 31;
        const int64_t row = row0 + 32 * grp + j;
        float v = out32[32 * grp + j];
        if (sg.epi == FQ_LNEPI_STORE) {
            if (lane < 32 && row < sg.w.M) sg.dst[row] = v;
            if (a.argmax_val) {                       // first stage of the greedy sampler: best (value, row) of these 32 rows
                float bv = row < sg.w.M ? v : -INFINITY; int bi = (int) row;
#pragma unroll
                for (int o = 1; o < 32; o <<= 1) {
                    const float ov = __shfl_xor(bv, o); const int oi = __shfl_xor(bi, o);
                    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
                }
                if (lane == 0 && row0 + 32 * grp < sg.w.M) { a.argmax_val[(row0 >> 5) + grp] = bv; a.argmax_idx[(row0 >> 5) + grp] = bi; }
            }
        } else {
            v = h2f_bits(a.gelu_table[f2h_bits(v)]);                                  // ggml.c:3477-3484
            if (sg.epi == FQ_LNEPI_GELU_STORE) {
                if (lane < 32 && row < sg.w.M) sg.dst[row] = v;
            } else if (row0 + 32 * grp < sg.w.M) {                                    // GELU -> Q8_0 / Q8_1 block of 32 (M % 32 == 0)
                const float amax = reduce32(fabsf(v), op_max());
                const float d  = amax / 127.0f;
                const float id = d ? 1.0f / d : 0.0f;
                const int q = round_half_away(v * id);
                const int s = reduce32(q, op_add());
                const act_image_ptr o = act_image_at(sg.dst_image, sg.next_act_type, sg.w.M);
                if (lane < 32) o.qs[row] = (int8_t) q;
                if (lane == 0) {
                    const int64_t b = (row0 >> 5) + grp;
                    if (sg.next_act_type == FQ_Q8_0) { o.d[b] = h2f_bits(f2h_bits(d)); ((int32_t *) o.aux)[b] = s; }
                    else                             { o.d[b] = d; ((float *) o.aux)[b] = (float) s * d; }
                }
            }
        }
    }
    FQ_STAMP(a.dbg, 7);
}

template <int TYPE, int MAXT>
__global__ void __launch_bounds__(MAXT) k_gemv_ln(fq_gemv_ln_args a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    gemv_ln_body<TYPE, MAXT>(a, (int) blockIdx.x, smem, fq_xsrc{ nullptr, 0u, nullptr });
}
template <int TYPE, int MAXT>
__global__ void __launch_bounds__(MAXT) k_gemv_ln_ref(fq_gemv_ln_args a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    gemv_ln_body<TYPE, MAXT, true>(a, (int) blockIdx.x, smem, fq_xsrc{ nullptr, 0u, nullptr });
}

size_t fq_gemv_ln_lds(int type, int64_t E) {
    const int act = fq_desc(type).act_type;
    return (((size_t) E * 4 + 15) & ~(size_t) 15) + fq_act_col_bytes(act, E) + 384 * 4 + 32 * 8;
}
static bool fq_legacy_type(int t) { return t == FQ_Q4_0 || t == FQ_Q4_1 || t == FQ_Q5_0 || t == FQ_Q5_1 || t == FQ_Q8_0; }

// workgroup shape: nw waves (4, 8 or 12) x 4*npass rows per wave (npass even, so that a workgroup is whole 32-row groups;
// at most 8 passes = 384 rows).
//   nw    : with 8 rows per wave, the count that needs the fewest rounds x rows-per-workgroup over the chip's CUs, ties
//           to the larger workgroup (fewer LN + Q8 prologues): 12 for Falcon-7B (239 workgroups) and 40B (438, two rounds)
//   npass : 2, except for the single-segment lm_head launch, which takes as many passes as it needs to fit the chip in
//           one round (12 x 6 rows, 226 workgroups: one prologue per CU, measured -8 us on Falcon-7B). A block's
//           two-segment launch does NOT gain from that at Falcon-40B width (measured 8 % slower than two rounds of
//           96-row workgroups, whose second-round prologues fall into the first round's stream).
static void gemv_ln_shape(const fq_gemv_ln_args & a, int n_cu, int & nw_out, int & npass_out) {
    int64_t best_cost = INT64_MAX;
    for (int nw = 4; nw <= 12; nw += 4) {
        int64_t blocks = 0;
        for (int s = 0; s < a.nseg; ++s) blocks += (a.seg[s].w.M + 8 * nw - 1) / (8 * nw);
        const int64_t cost = ((blocks + n_cu - 1) / n_cu) * nw;
        if (cost <= best_cost) { best_cost = cost; nw_out = nw; }
    }
    npass_out = 2;
    // (only where the prologue is a large share of a 96-row workgroup: < 384 KB of weights per workgroup, i.e. Falcon-7B width)
    if (a.nseg == 1 && a.seg[0].w.bytes / (size_t)(a.seg[0].w.M ? a.seg[0].w.M : 1) * 8 * nw_out < 384 * 1024)
        while (npass_out < 8 && (a.seg[0].w.M + 4 * npass_out * nw_out - 1) / (4 * npass_out * nw_out) > n_cu) npass_out += 2;
}

void fq_launch_gemv_ln(fq_gemv_ln_args a, int n_cu, hipStream_t st, bool ref) {
    FQ_TL(st, ref ? "gemv_ln_ref" : "gemv_ln");
    int nw = 4, npass = 2;
    gemv_ln_shape(a, n_cu, nw, npass);
    a.npass = npass;
    const int rows = 4 * npass * nw;
    int blocks = 0;
    for (int s = 0; s < a.nseg; ++s) { a.seg[s].block_begin = blocks; blocks += (int)((a.seg[s].w.M + rows - 1) / rows); }
    a.n_blocks = blocks;
    const int type = a.seg[0].w.type;
    size_t lds = fq_gemv_ln_lds(type, a.E);
    if (ref) {
        // the fast reference order: + 8 strip rows per wave (fq_ref_chain.h); legacy formats, one format per launch
        for (int s = 0; s < a.nseg; ++s) if (!fq_legacy_type(a.seg[s].w.type) || a.seg[s].w.type != type) { fprintf(stderr, "ggml-hip: gemv_ln: the fast reference order covers launches of one legacy format\n"); exit(1); }
        lds += (size_t) nw * 8 * fq_ref_strip_stride((int)(a.E / 32)) * 4;
        if (lds > 160 * 1024) { fprintf(stderr, "ggml-hip: gemv_ln: the term strips of a %lld-wide row do not fit the LDS\n", (long long) a.E); exit(1); }
    }
    // a grid that fits the chip gets one workgroup per CU: claim more than half of the 160 KiB LDS so that the dispatcher
    // cannot co-locate two of them while other CUs stay empty
    if (blocks <= n_cu && lds < 84 * 1024) lds = 84 * 1024;
#define FQ_LAUNCH_R(T, MAXT) { \
        static size_t g = 0; if (lds > 64 * 1024 && lds > g) { HIP_CHECK(hipFuncSetAttribute((const void *) k_gemv_ln_ref<T, MAXT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds)); g = lds; } \
        FQ_LAUNCH_PROF((k_gemv_ln_ref<T, MAXT>), dim3((unsigned) blocks), dim3(64 * nw), lds, st, a); }
#define FQ_CASE_R(T) case T: if (nw <= 4) FQ_LAUNCH_R(T, 256) else FQ_LAUNCH_R(T, 768) break;
    if (ref) {
        switch (type) {
            FQ_CASE_R(FQ_Q4_0) FQ_CASE_R(FQ_Q4_1) FQ_CASE_R(FQ_Q5_0) FQ_CASE_R(FQ_Q5_1) FQ_CASE_R(FQ_Q8_0)
            default: break;
        }
        return;
    }
#undef FQ_CASE_R
#undef FQ_LAUNCH_R
#define FQ_LAUNCH(T, MAXT) { \
        static size_t g = 0; if (lds > 64 * 1024 && lds > g) { HIP_CHECK(hipFuncSetAttribute((const void *) k_gemv_ln<T, MAXT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds)); g = lds; } \
        FQ_LAUNCH_PROF((k_gemv_ln<T, MAXT>), dim3((unsigned) blocks), dim3(64 * nw), lds, st, a); }
#define FQ_CASE(T) case T: if (nw <= 4) FQ_LAUNCH(T, 256) else FQ_LAUNCH(T, 768) break;
    switch (type) {
        FQ_CASE(FQ_Q4_0) FQ_CASE(FQ_Q4_1) FQ_CASE(FQ_Q5_0) FQ_CASE(FQ_Q5_1) FQ_CASE(FQ_Q8_0)
        FQ_CASE(FQ_Q2_K) FQ_CASE(FQ_Q3_K) FQ_CASE(FQ_Q4_K) FQ_CASE(FQ_Q5_K) FQ_CASE(FQ_Q6_K)
        default: fprintf(stderr, "ggml-hip: gemv_ln: unsupported weight type %d\n", type); exit(1);
    }
#undef FQ_CASE
#undef FQ_LAUNCH
}

// =============================================================================================== k_gemv_out
// one workgroup = NW waves = 2*NW consecutive output rows (wave w: rows 2w, 2w+1), NW chosen by the launcher for about
// one workgroup per CU; x[row] = (down + wo) + x[row]
// (round 6: the first-needed arguments as leading scalars, preloaded into SGPRs -- see k_attn_out)
template <int TYPE, int MAXT>
__global__ void __launch_bounds__(MAXT) k_gemv_out(const uint8_t * p_ff, const uint8_t * p_att_image, uint8_t * p_down, uint8_t * p_wo, const float * p_resid, float * p_dst, fq_gemv_out_args a) {
    a.act_ff_image = p_ff; a.att_image = p_att_image; a.w_down.plane[0] = p_down; a.w_wo.plane[0] = p_wo; a.resid = p_resid; a.dst = p_dst;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr int ACT = act_of<TYPE>::value;
    const int64_t E = a.w_wo.K, FF = a.w_down.K;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    uint8_t * img_ff  = smem;                                           // image of gelu(up), already quantized
    uint8_t * img_att = smem + fq_act_col_bytes(ACT, FF);

    constexpr int NPD = decode_cfg<TYPE>::OUT_NPRE_D, NPO = decode_cfg<TYPE>::OUT_NPRE_O;
    const int units_d = (int)(FF / fq_unit<TYPE>::ELEMS), units_o = (int)(E / fq_unit<TYPE>::ELEMS);
    const int nt = blockDim.x;
    const int64_t row0 = (int64_t) blockIdx.x * (2 * (nt >> 6)) + 2 * wid;
    (void) lane;
    FQ_STAMP(a.dbg, 0);
    // 1. prologue loads: the quantized gelu(up) image and either the already-quantized attention image (flat 16-byte
    //    vectors, img_att directly follows img_ff in LDS) or the f32 attention row (Q8_K: quantized here)
    const int64_t nvec_ff = (int64_t)(fq_act_col_bytes(ACT, FF) >> 4);
    const int64_t nvec_at = a.att_image ? (int64_t)(fq_act_col_bytes(ACT, E) >> 4) : 0;
    const int64_t nq = a.att_image ? 0 : (E >> 2);
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 * src_ff = (const u32x4 *) a.act_ff_image;
    const u32x4 * src_at = (const u32x4 *) (a.att_image ? a.att_image : a.act_ff_image);
    constexpr int NTF = MAXT > 256 ? 4 : 8, NTQ = MAXT > 256 ? 1 : 3;       // prologue vectors held in registers per thread
    u32x4 tf[NTF], tq[NTQ];
#pragma unroll
    for (int k = 0; k < NTF; ++k) { const int64_t i = (int64_t) k * nt + tid; tf[k] = src_ff[i < nvec_ff ? i : nvec_ff - 1]; }
#pragma unroll
    for (int k = 0; k < NTQ; ++k) { const int64_t i = (int64_t) k * nt + tid; tq[k] = src_at[i < nvec_at ? i : 0]; }
    __builtin_amdgcn_sched_barrier(0);
    // 2. weight loads for both sources
    const float res0 = a.resid[row0 < a.w_wo.M ? row0 : 0], res1 = a.resid[row0 + 1 < a.w_wo.M ? row0 + 1 : 0];
    fq_wrow rd[2], ro[2];
    rows_ptrs<TYPE, 2>(a.w_down, row0, rd);
    rows_ptrs<TYPE, 2>(a.w_wo, row0, ro);
    fq_unit_regs pd[NPD][2], po[NPO][2];
    rows_issue<TYPE, 2, NPD>(rd, units_d, pd);
    rows_issue<TYPE, 2, NPO>(ro, units_o, po);
    FQ_STAMP(a.dbg, 1);
    // 3. finish the prologue while the weights stream
#pragma unroll
    for (int k = 0; k < NTF; ++k) { const int64_t i = (int64_t) k * nt + tid; if (i < nvec_ff) ((u32x4 *) img_ff)[i] = tf[k]; }
    for (int64_t i = (int64_t) NTF * nt + tid; i < nvec_ff; i += nt) ((u32x4 *) img_ff)[i] = src_ff[i];
#pragma unroll
    for (int k = 0; k < NTQ; ++k) { const int64_t i = (int64_t) k * nt + tid; if (i < nvec_at) ((u32x4 *) img_att)[i] = tq[k]; }
    for (int64_t i = (int64_t) NTQ * nt + tid; i < nvec_at; i += nt) ((u32x4 *) img_att)[i] = src_at[i];
    if (nq) {
        float * att_f = (float *)(img_att + fq_act_col_bytes(ACT, E));      // f32 copy of the attention row
        for (int64_t i = tid; i < nq; i += nt) ((float4 *) att_f)[i] = ((const float4 *) a.att)[i];
        __syncthreads();
        FQ_STAMP(a.dbg, 2);
        quantize_row_block<ACT>(att_f, E, act_image_at(img_att, ACT, E));
    }
    __syncthreads();
    FQ_STAMP(a.dbg, 3);

    float acc_d[2] = {0.0f, 0.0f}, acc_o[2] = {0.0f, 0.0f};
    const fq_actcol col_d = actcol_at(img_ff, ACT, FF), col_o = actcol_at(img_att, ACT, E);
    rows_consume<TYPE, 2, NPD>(pd, units_d, col_d, acc_d);
    FQ_STAMP(a.dbg, 4);
    // (k-quants: 3, 4 or 6 unit columns per trip instead of 2 measured within 1.5 % on Falcon-40B Q2_K / Q4_K / Q6_K -- the trips are not what bounds the launch)
    rows_dot_from<TYPE, 2, (decode_cfg<TYPE>::four_bit ? 5 : 2)>(rd, units_d, 64 * NPD, col_d, acc_d);      // 7B: the remaining 5 columns in one round trip
    FQ_STAMP(a.dbg, 5);
    rows_consume<TYPE, 2, NPO>(po, units_o, col_o, acc_o);
    rows_dot_from<TYPE, 2, 2>(ro, units_o, 64 * NPO, col_o, acc_o);
#pragma unroll
    for (int r = 0; r < 2; ++r) { acc_d[r] = wave_sum(acc_d[r]); acc_o[r] = wave_sum(acc_o[r]); }
    if (lane == 0) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int64_t row = row0 + r;
            if (row < a.w_wo.M) a.dst[row] = (acc_d[r] + acc_o[r]) + (r ? res1 : res0);            // libfalcon.cpp:2399-2400
        }
    }
    FQ_STAMP(a.dbg, 7);
}

static void gemv_out_ref_launch(const fq_gemv_out_args & a, int n_cu, hipStream_t st);
void fq_launch_gemv_out(const fq_gemv_out_args & a, int n_cu, hipStream_t st, bool ref) {
    FQ_TL(st, ref ? "gemv_out_ref" : "gemv_out");
    if (ref) { gemv_out_ref_launch(a, n_cu, st); return; }
    const int type = a.w_wo.type;
    const int act = fq_desc(type).act_type;
    size_t lds = fq_act_col_bytes(act, a.w_down.K) + fq_act_col_bytes(act, a.w_wo.K) + (a.att_image ? 0 : (size_t) a.w_wo.K * 4) + 16;
    // 2 rows per wave; 4..12 waves per workgroup, the count with the fewest rounds x rows-per-workgroup over the CUs
    int nw = 4; int64_t best_cost = INT64_MAX;
    for (int c = 4; c <= 12; ++c) {
        const int64_t nb = (a.w_wo.M + 2 * c - 1) / (2 * c);
        const int64_t cost = ((nb + n_cu - 1) / n_cu) * c;
        if (cost <= best_cost) { best_cost = cost; nw = c; }
    }
    const unsigned blocks = (unsigned)((a.w_wo.M + 2 * nw - 1) / (2 * nw));
    if ((int) blocks <= n_cu && lds < 84 * 1024) lds = 84 * 1024;       // one workgroup per CU (see fq_launch_gemv_ln)
#define FQ_LAUNCH(T, MAXT) { \
        static size_t g = 0; if (lds > 64 * 1024 && lds > g) { HIP_CHECK(hipFuncSetAttribute((const void *) k_gemv_out<T, MAXT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds)); g = lds; } \
        FQ_LAUNCH_PROF((k_gemv_out<T, MAXT>), dim3(blocks), dim3(64 * nw), lds, st, a.act_ff_image, a.att_image, a.w_down.plane[0], a.w_wo.plane[0], a.resid, a.dst, a); }
#define FQ_CASE(T) case T: if (nw <= 4) FQ_LAUNCH(T, 256) else FQ_LAUNCH(T, 768) break;
    switch (type) {
        FQ_CASE(FQ_Q4_0) FQ_CASE(FQ_Q4_1) FQ_CASE(FQ_Q5_0) FQ_CASE(FQ_Q5_1) FQ_CASE(FQ_Q8_0)
        FQ_CASE(FQ_Q2_K) FQ_CASE(FQ_Q3_K) FQ_CASE(FQ_Q4_K) FQ_CASE(FQ_Q5_K) FQ_CASE(FQ_Q6_K)
        default: fprintf(stderr, "ggml-hip: gemv_out: unsupported weight type %d\n", type); exit(1);
    }
#undef FQ_CASE
#undef FQ_LAUNCH
}

// =============================================================================================== k_attn_decode
// one query head, N = 1, by a group of 256 threads (tid = index in the group; barriers are workgroup barriers, every
// group of a workgroup runs this in lockstep). Rotates q and the new k itself (ggml.c:12957-12978), appends k/v to the
// cache (first head of each kv group), runs fq_attn_dev.h with the newest key/value taken from LDS, and -- when the
// output projection's activation format is Q8_0 / Q8_1 -- quantizes its 64 outputs (two 32-blocks) straight into the
// activation image the output mat-vec stages, so no separate quantizer pass or f32 round trip is needed.
// PUBLISH: the results are consumed by other workgroups of the SAME launch (k_attn_out): agent-scope write-through
// stores. live = false: a padding group (no head left) computes head H-1 again and stores nothing.
__global__ void __launch_bounds__(256) k_attn_decode(fq_attn_decode_args a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    attn_decode_group<false>(a, (int) blockIdx.x, true, (int) threadIdx.x, smem);
}
__global__ void __launch_bounds__(256) k_attn_decode_f64(fq_attn_decode_args a) {      // the fast reference order: dots accumulated in f64
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    attn_decode_group<false, true>(a, (int) blockIdx.x, true, (int) threadIdx.x, smem);
}


size_t fq_attn_decode_lds_bytes(int max_n_kv) { return attn_decode_lds(max_n_kv); }

void fq_launch_attn_decode(const float * qkv, int H, int HKV, const int * n_past_dev, int max_n_kv, const float * rope_cs,
                           float * k_cache, float * v_cache, const uint16_t * exp_table, float * att, uint8_t * att_image, int att_act_type, hipStream_t st, bool f64) {
    FQ_TL(st, f64 ? "attn_decode_f64" : "attn_decode");
    const size_t lds = attn_decode_lds(max_n_kv);
    if (lds > 160 * 1024) { fprintf(stderr, "ggml-hip: attention: %d keys do not fit the score buffer in LDS\n", max_n_kv); exit(1); }
    const fq_attn_decode_args a{ qkv, H, HKV, n_past_dev, rope_cs, k_cache, v_cache, exp_table, att, att_image, att_act_type, max_n_kv, nullptr, nullptr, nullptr, nullptr };
    if (f64) {
        if (lds > 64 * 1024) { static size_t g = 0; if (lds > g) { HIP_CHECK(hipFuncSetAttribute((const void *) k_attn_decode_f64, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds)); g = lds; } }
        hipLaunchKernelGGL(k_attn_decode_f64, dim3((unsigned) H), dim3(256), lds, st, a);
        return;
    }
    if (lds > 64 * 1024) { static size_t g = 0; if (lds > g) { HIP_CHECK(hipFuncSetAttribute((const void *) k_attn_decode, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds)); g = lds; } }
    hipLaunchKernelGGL(k_attn_decode, dim3((unsigned) H), dim3(256), lds, st, a);
}

// the same for B lock-step sequences (falcon_hip_context_create_seqs): blockIdx.y = sequence, with its own qkv row, KV cache
// and output column -- RoPE, KV append, attention and the Q8 image of B tokens in one launch
// Optional rider (qx != nullptr): workgroups blockIdx.x >= H quantize column blockIdx.y of the f32 matrix qx into the image qa (k_quantize_q8's
// code) -- the GELU output of the block's other branch, which is ready at the same time as q / k / v and otherwise costs a launch of its own.
__global__ void __launch_bounds__(256) k_attn_decode_seqs(fq_attn_decode_args a, int64_t qkv_stride, int64_t seq_stride, int64_t att_stride, int64_t image_stride,
                                                          int H, const float * __restrict__ qx, int64_t q_ldx, fq_act qa) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int64_t t = blockIdx.y;
    if ((int) blockIdx.x >= H && qa.type == FQ_Q8_K) {                     // k-quant consumers: one wave per 256-element super-block (k_quantize_q8K's code)
        const int lane = threadIdx.x & 63;
        const act_image_ptr o = act_image_at(qa.base + (size_t) t * fq_act_col_bytes(qa.type, qa.K), qa.type, qa.K);
        for (int64_t sb = (int64_t)((int) blockIdx.x - H) * 4 + (threadIdx.x >> 6); sb < (qa.K >> 8); sb += (int64_t)((int) gridDim.x - H) * 4)
            quant_q8K_wave(*(const float4 *)(qx + t * q_ldx + 256 * sb + 4 * lane), lane, sb, o);
        return;
    }
    if ((int) blockIdx.x >= H) {
        const int64_t quads = qa.K >> 2;                                   // (a multiple of 8: whole 32-blocks)
        const int nb = (int) gridDim.x - H;
        const act_image_ptr o = act_image_at(qa.base + (size_t) t * fq_act_col_bytes(qa.type, qa.K), qa.type, qa.K);
        for (int64_t q4 = (int64_t)((int) blockIdx.x - H) * 256 + threadIdx.x; q4 < ((quads + 63) & ~(int64_t) 63); q4 += (int64_t) nb * 256) {
            const bool live = q4 < quads;
            const int64_t qq = live ? q4 : quads - 1;
            const float4 v = *(const float4 *)(qx + t * q_ldx + 4 * qq);
            if (qa.type == FQ_Q8_0) quant_q8_quad<FQ_Q8_0>(v, qq, o, live); else quant_q8_quad<FQ_Q8_1>(v, qq, o, live);
        }
        return;
    }
    a.qkv += t * qkv_stride; a.kc += t * seq_stride; a.vc += t * seq_stride;
    if (a.att) a.att += t * att_stride;
    if (a.att_image) a.att_image += t * image_stride;
    attn_decode_group<false>(a, (int) blockIdx.x, true, (int) threadIdx.x, smem);
}
void fq_launch_attn_decode_seqs(const float * qkv, int n_seq, int H, int HKV, const int * n_past_dev, int max_n_kv, const float * rope_cs,
                                float * k_cache, float * v_cache, int64_t seq_stride, const uint16_t * exp_table, float * att, uint8_t * att_image,
                                int att_act_type, int64_t image_stride, hipStream_t st, const float * qx, int64_t q_ldx, const fq_act * qa) {
    FQ_TL(st, "attn_decode_seqs");
    const size_t lds = attn_decode_lds(max_n_kv);
    if (lds > 160 * 1024) { fprintf(stderr, "ggml-hip: attention: %d keys do not fit the score buffer in LDS\n", max_n_kv); exit(1); }
    if (lds > 64 * 1024) { static size_t g = 0; if (lds > g) { HIP_CHECK(hipFuncSetAttribute((const void *) k_attn_decode_seqs, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds)); g = lds; } }
    const fq_attn_decode_args a{ qkv, H, HKV, n_past_dev, rope_cs, k_cache, v_cache, exp_table, att, att_image, att_act_type, max_n_kv, nullptr, nullptr, nullptr, nullptr };
    const bool ride = qx && qa && (qa->type == FQ_Q8_0 || qa->type == FQ_Q8_1 || (qa->type == FQ_Q8_K && qa->K % 256 == 0)) && qa->ncols >= n_seq;
    const int extra = !ride ? 0 : (qa->type == FQ_Q8_K ? (int)(((qa->K >> 8) + 3) / 4) : (int)(((qa->K >> 2) + 255) / 256));      // quantizer workgroups per column
    hipLaunchKernelGGL(k_attn_decode_seqs, dim3((unsigned)(H + extra), (unsigned) n_seq), dim3(256), lds, st, a, (int64_t)(H + 2 * HKV) * 64, seq_stride, (int64_t) H * 64,
                       image_stride, H, ride ? qx : nullptr, q_ldx, ride ? *qa : fq_act{});
}

// =============================================================================================== k_attn_out
// Attention and the output mat-vec of one block in ONE launch, one workgroup of 12 waves per CU:
//   workgroups [0, n_attn)      3 query heads each (three lockstep 256-thread groups running attn_decode_group), results
//                               PUBLISHED to the others: agent-scope write-through stores, every storing wave drains
//                               (s_waitcnt vmcnt(0)), barrier, one agent-scope counter increment per workgroup
//   workgroups [n_attn, grid)   24 rows of x = (Wdown . q8(gelu(up)) + Wo . q8(att)) + x each: the Wdown part (80 % of
//                               the bytes) does not need the attention output and streams while the attention runs;
//                               then ONE lane polls the counter (relaxed, s_sleep), barrier, the attention image is read
//                               with agent-scope loads (the producers stored write-through: no acquire fence needed),
//                               and the pre-fetched Wo rows are finished.
// ~190 streaming CUs already saturate HBM (scripts/microbench/mb_stream.hip), so lending 24 CUs to the latency-bound
// attention costs the stream nothing and removes a launch boundary plus the attention's 8 us from every block.
// Requires every workgroup to be resident at once (grid <= CUs; the launcher checks) -- or at least the attention
// workgroups to start first, which in-order dispatch gives; the poll is bounded and reports through a.err.
// The arithmetic is k_attn_decode's and k_gemv_out's: bit-identical results.
struct fq_attn_out_args {
    fq_gemv_out_args g;             // g.att / g.att_image are what the attention role writes
    fq_attn_decode_args at;
    unsigned long long * gran;      // hand-off buffer: one granule per 32-bit word of the attention image (or f32 row)
    const unsigned * epoch_word;    // this launch's tag (incremented by the preceding k_gemv_ln launch, never 0)
    unsigned * err;                 // set to 1 if a sweep gave up
    int n_attn, n_mv, heads_per_wg, attn_lds_group;
    int ref_debug;                  // tuning aid of the reference-order form (FQ_REF_DBG; results are then garbage): 2 / 4 = no Wdown / Wo chain
};

// xpub.gran != nullptr: the new residual values are ALSO published as granules for the LayerNorm phase of the same launch
template <int TYPE>
__device__ __forceinline__ void attn_out_body(const fq_attn_out_args & a, uint8_t * smem, const unsigned epoch, const fq_publish xpub) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, nt = blockDim.x;
    if ((int) blockIdx.x >= a.n_attn + a.n_mv) return;             // (two-phase kernel: workgroups that only have a second phase)
    if ((int) blockIdx.x < a.n_attn) {
        // ------------------------------------------------------------------------------------ attention role
        const int grp = tid >> 8, gtid = tid & 255;
        if (grp >= a.heads_per_wg) { if (xpub.gran) attn_decode_group_idle(); return; }   // two-phase kernel: these waves come back for phase 2
        int h = (int) blockIdx.x * a.heads_per_wg + grp;
        const bool live = h < a.at.H;
        if (!live) h = a.at.H - 1;
        long long * dbg = a.g.dbg ? a.g.dbg + 2048 * 8 : nullptr;
        FQ_STAMP(dbg, 0);
        attn_decode_group<true>(a.at, h, live, gtid, smem + (size_t) grp * a.attn_lds_group, dbg, fq_publish{ a.gran, epoch });
        FQ_STAMP(dbg, 7);
        return;
    }
    // ---------------------------------------------------------------------------------------- mat-vec role
    constexpr int ACT = act_of<TYPE>::value;
    const fq_gemv_out_args & g = a.g;
    const int64_t E = g.w_wo.K, FF = g.w_down.K;
    uint8_t * img_ff  = smem;
    uint8_t * img_att = smem + fq_act_col_bytes(ACT, FF);
    constexpr int NPD = decode_cfg<TYPE>::OUT_NPRE_D, NPO = decode_cfg<TYPE>::OUT_NPRE_O;
    const int units_d = (int)(FF / fq_unit<TYPE>::ELEMS), units_o = (int)(E / fq_unit<TYPE>::ELEMS);
    const int64_t row0 = (int64_t)((int) blockIdx.x - a.n_attn) * (2 * (nt >> 6)) + 2 * wid;
    long long * dbg = g.dbg ? g.dbg - (size_t) a.n_attn * 8 : nullptr;         // stamps indexed by mat-vec workgroup
    FQ_STAMP(dbg, 0);
    const int64_t nvec_ff = (int64_t)(fq_act_col_bytes(ACT, FF) >> 4);
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 * src_ff = (const u32x4 *) g.act_ff_image;
    constexpr int NTF = 4;
    u32x4 tf[NTF];
#pragma unroll
    for (int k = 0; k < NTF; ++k) { const int64_t i = (int64_t) k * nt + tid; tf[k] = src_ff[i < nvec_ff ? i : nvec_ff - 1]; }
    __builtin_amdgcn_sched_barrier(0);
    // (the residual is asked for now: as a dependent load at the very end it would add a memory round trip to the tail)
    const float res0 = g.resid[row0 < g.w_wo.M ? row0 : 0], res1 = g.resid[row0 + 1 < g.w_wo.M ? row0 + 1 : 0];
    fq_wrow rd[2], ro[2];
    rows_ptrs<TYPE, 2>(g.w_down, row0, rd);
    rows_ptrs<TYPE, 2>(g.w_wo, row0, ro);
    fq_unit_regs pd[NPD][2], po[NPO][2];
    rows_issue<TYPE, 2, NPD>(rd, units_d, pd);
    rows_issue<TYPE, 2, NPO>(ro, units_o, po);
    FQ_STAMP(dbg, 1);
#pragma unroll
    for (int k = 0; k < NTF; ++k) { const int64_t i = (int64_t) k * nt + tid; if (i < nvec_ff) ((u32x4 *) img_ff)[i] = tf[k]; }
    for (int64_t i = (int64_t) NTF * nt + tid; i < nvec_ff; i += nt) ((u32x4 *) img_ff)[i] = src_ff[i];
    __syncthreads();
    FQ_STAMP(dbg, 3);
    float acc_d[2] = {0.0f, 0.0f}, acc_o[2] = {0.0f, 0.0f};
    const fq_actcol col_d = actcol_at(img_ff, ACT, FF), col_o = actcol_at(img_att, ACT, E);
    rows_consume<TYPE, 2, NPD>(pd, units_d, col_d, acc_d);
    FQ_STAMP(dbg, 4);
    rows_dot_from<TYPE, 2, (decode_cfg<TYPE>::four_bit ? 5 : 2)>(rd, units_d, 64 * NPD, col_d, acc_d);
    FQ_STAMP(dbg, 5);
    // ---- the attention output: every wave re-reads its share of the granules (agent-scope loads, past the caches) until
    // all of them carry this launch's tag, and drops the values into the image in LDS
    {
        const int64_t nwords = g.att_image ? (E >> 2) + 2 * (E >> 5) : E;      // published words: [qs | d | aux] of Q8_0 / Q8_1, or the f32 row
        unsigned * dstw = g.att_image ? (unsigned *) img_att : (unsigned *)(img_att + fq_act_col_bytes(ACT, E));
        constexpr int NG = 3;                                        // granules per thread and sweep (768 threads: 2304 words)
        for (int64_t base = 0; base < nwords; base += (int64_t) NG * nt) {
            unsigned v[NG];
            for (unsigned spins = 0;; ++spins) {
                bool ok = true;
#pragma unroll
                for (int k = 0; k < NG; ++k) {
                    const int64_t i = base + (int64_t) k * nt + tid;
                    const unsigned long long x = __hip_atomic_load(a.gran + (i < nwords ? i : nwords - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    v[k] = (unsigned) x; ok = ok && (unsigned)(x >> 32) == epoch;
                }
                if (__all(ok)) break;
                if (spins > (1u << 20)) { if (lane == 0) __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                __builtin_amdgcn_s_sleep(4);
            }
#pragma unroll
            for (int k = 0; k < NG; ++k) { const int64_t i = base + (int64_t) k * nt + tid; if (i < nwords) dstw[i] = v[k]; }
        }
    }
    FQ_STAMP(dbg, 6);
    if (!g.att_image) {
        __syncthreads();
        quantize_row_block<ACT>((const float *)(img_att + fq_act_col_bytes(ACT, E)), E, act_image_at(img_att, ACT, E));
    }
    __syncthreads();
    rows_consume<TYPE, 2, NPO>(po, units_o, col_o, acc_o);
    rows_dot_from<TYPE, 2, 2>(ro, units_o, 64 * NPO, col_o, acc_o);
#pragma unroll
    for (int r = 0; r < 2; ++r) { acc_d[r] = wave_sum(acc_d[r]); acc_o[r] = wave_sum(acc_o[r]); }
    if (lane == 0) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int64_t row = row0 + r;
            if (row < g.w_wo.M) {
                const float v = (acc_d[r] + acc_o[r]) + (r ? res1 : res0);                          // libfalcon.cpp:2399-2400
                g.dst[row] = v;
                if (xpub.gran) __hip_atomic_store(xpub.gran + row, ((unsigned long long) xpub.epoch << 32) | __builtin_bit_cast(unsigned, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    FQ_STAMP(dbg, 7);
}

// Kernel arguments (round 6, as k_gemv_ln_ring): the words both roles need first are leading scalar arguments, preloaded into SGPRs by the command processor
// (-mllvm -amdgpu-kernarg-preload-count=14 for this file); the ~300-byte struct behind them is fetched by scalar loads as before. FQ_ATTN_OUT_PRELOAD=0: struct only.
#ifndef FQ_ATTN_OUT_PRELOAD
#define FQ_ATTN_OUT_PRELOAD 1
#endif
#if FQ_ATTN_OUT_PRELOAD
#define FQ_AO_PARAMS const uint8_t * p_ff, uint8_t * p_down, uint8_t * p_wo, const float * p_qkv, const int * p_np, const unsigned * p_epoch, int p_n_attn, int p_n_mv,
#define FQ_AO_TAKE   a.g.act_ff_image = p_ff; a.g.w_down.plane[0] = p_down; a.g.w_wo.plane[0] = p_wo; a.at.qkv = p_qkv; a.at.n_past_ptr = p_np; a.epoch_word = p_epoch; a.n_attn = p_n_attn; a.n_mv = p_n_mv;
#define FQ_AO_LEAD   a.g.act_ff_image, a.g.w_down.plane[0], a.g.w_wo.plane[0], a.at.qkv, a.at.n_past_ptr, a.epoch_word, a.n_attn, a.n_mv,
#else
#define FQ_AO_PARAMS
#define FQ_AO_TAKE
#define FQ_AO_LEAD
#endif
template <int TYPE>
__global__ void __launch_bounds__(768) k_attn_out(FQ_AO_PARAMS fq_attn_out_args a) {
    FQ_AO_TAKE
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    attn_out_body<TYPE>(a, smem, *a.epoch_word, fq_publish{ nullptr, 0u });
}

// =============================================================================================== k_attn_out_ln
// TWO phases in one launch (256 workgroups of 12 waves, one per CU): phase 1 = k_attn_out of block l, phase 2 = k_gemv_ln
// of block l+1 (or ln_f + lm_head after the last block). The new residual row crosses from the 190 mat-vec workgroups to
// all phase-2 workgroups through tagged granules, like the attention output inside phase 1: no launch boundary, no kernel
// start between the two, and a workgroup that finished phase 1 already has its LayerNorm weights and its first weight
// column in flight while it waits for the row.
template <int TYPE>
__global__ void __launch_bounds__(768) k_attn_out_ln(fq_attn_out_args a, fq_gemv_ln_args b, unsigned long long * xgran) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const unsigned epoch = *a.epoch_word;
    attn_out_body<TYPE>(a, smem, epoch, fq_publish{ xgran, epoch });
    __syncthreads();
    if ((int) blockIdx.x < b.n_blocks) gemv_ln_body<TYPE, 768>(b, (int) blockIdx.x, smem, fq_xsrc{ xgran, epoch, a.err });
}

// =============================================================================================== the fast reference order of the output launches
// k_gemv_out_ref / k_attn_out_ref (ggml_hip_reference_order(2), fq_ref_chain.h; legacy formats): the same weight stream, integer dots and per-block f32
// terms as k_gemv_out / k_attn_out, but each row's terms are added LEFT TO RIGHT, as the reference's scalar build adds them (ggml.c:2591-2609 and the other
// legacy vec_dots), and the decode attention accumulates its dots in f64 (ggml.c:2296-2300). A workgroup of nw waves = nw - 1 consumer waves with two rows
// each + ONE summing wave whose lanes are the workgroup's rows: the consumers leave the unit terms in an LDS strip [row][Wdown blocks | Wo blocks], the
// summing wave follows them -- the pre-issued Wdown columns while the rest of Wdown streams, the rest while the attention output is awaited, Wo's at the end:
// only Wo's chain (n_embd / 32 dependent adds) trails the stream. x[row] = (down + wo) + x[row] as libfalcon.cpp:2399-2400.
struct fq_out_ref_geom { unsigned swd, swt; };           // floats: offset of the Wo terms inside a row's strip, stride of the rows
__host__ __device__ inline fq_out_ref_geom fq_out_ref_strip(int units_d, int units_o) {
    fq_out_ref_geom g; g.swd = ((unsigned) units_d + 3u) & ~3u; g.swt = fq_ref_strip_stride((int) g.swd + units_o); return g;
}
static size_t fq_out_ref_lds(int act, int64_t FF, int64_t E, int nw) {
    const fq_out_ref_geom sg = fq_out_ref_strip((int)(FF / 32), (int)(E / 32));
    return fq_act_col_bytes(act, FF) + fq_act_col_bytes(act, E) + 32 + (size_t) 2 * (nw - 1) * sg.swt * 4;
}
__device__ __forceinline__ void ref_wait(unsigned addr, unsigned target, unsigned * err) {      // LDS counter (monotonic) >= target; bounded
    for (unsigned spins = 0;; ++spins) {
        unsigned v; asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(addr) : "memory");
        if ((int)(__builtin_amdgcn_readfirstlane(v) - target) >= 0) break;
        if (spins > (1u << 22)) { if (err && (threadIdx.x & 63) == 0) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        __builtin_amdgcn_s_sleep(1);
    }
}
__device__ __forceinline__ void ref_count(unsigned addr) { if ((threadIdx.x & 63) == 0) asm volatile("ds_add_u32 %0, %1" :: "v"(addr), "v"(1u) : "memory"); }

// wg = index among the mat-vec workgroups; gran != nullptr: the attention image arrives through the granules of the same launch (k_attn_out_ref),
// else it is in memory (k_gemv_out_ref)
template <int TYPE>
__device__ __forceinline__ void gemv_out_ref_body(const fq_gemv_out_args & g, uint8_t * smem, const int wg, const unsigned long long * gran, const unsigned epoch, unsigned * err, const int dbgm = 0) {
    constexpr int ACT = act_of<TYPE>::value;
    static_assert(ACT == FQ_Q8_0 || ACT == FQ_Q8_1, "legacy formats");
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, nt = blockDim.x, ncw = (nt >> 6) - 1;      // ncw consumer waves + the summing wave
    const int64_t E = g.w_wo.K, FF = g.w_down.K;
    uint8_t * img_ff  = smem;
    uint8_t * img_att = smem + fq_act_col_bytes(ACT, FF);
    uint8_t * ctlp    = img_att + fq_act_col_bytes(ACT, E);
    float   * strip   = (float *)(ctlp + 32);
    const unsigned ctl = (unsigned)(uintptr_t) ctlp, DCNT_A = ctl, DCNT_B = ctl + 4, OCNT = ctl + 8, SWEPT = ctl + 12, OCOL = ctl + 16;      // consumer waves done with: the pre-issued Wdown columns, all of Wdown, Wo, their share of the attention image
    constexpr int NPD = decode_cfg<TYPE>::OUT_NPRE_D, NPO = decode_cfg<TYPE>::OUT_NPRE_O;
    const int units_d = (int)(FF / 32), units_o = (int)(E / 32);
    const fq_out_ref_geom sg = fq_out_ref_strip(units_d, units_o);
    const int64_t wg_row0 = (int64_t) wg * (2 * ncw);
    const bool summing = wid == ncw;
    const int64_t row0 = wg_row0 + 2 * (summing ? 0 : wid);
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const int64_t nvec_ff = (int64_t)(fq_act_col_bytes(ACT, FF) >> 4);
    const int64_t nvec_at = gran ? 0 : (int64_t)(fq_act_col_bytes(ACT, E) >> 4);       // (img_att directly follows img_ff in LDS: one flat copy of 16-byte vectors)
    const u32x4 * src_ff = (const u32x4 *) g.act_ff_image;
    const u32x4 * src_at = (const u32x4 *) g.att_image;
    constexpr int NTF = 4;
    u32x4 tf[NTF];
#pragma unroll
    for (int k = 0; k < NTF; ++k) { const int64_t i = (int64_t) k * nt + tid; tf[k] = src_ff[i < nvec_ff ? i : nvec_ff - 1]; }
    if (tid < 8) asm volatile("ds_write_b32 %0, %1" :: "v"(ctl + 4u * (unsigned) tid), "v"(0u) : "memory");
    __builtin_amdgcn_sched_barrier(0);
    // the summing wave: lane = row of the workgroup; its residual values now (a dependent load at the very end would add a memory round trip to the tail)
    const int64_t srow = wg_row0 + lane;
    const bool slive = summing && lane < 2 * ncw && srow < g.w_wo.M;
    const float sres = summing ? g.resid[slive ? srow : 0] : 0.0f;
    fq_wrow rd[2], ro[2];
    rows_ptrs<TYPE, 2>(g.w_down, row0, rd);
    rows_ptrs<TYPE, 2>(g.w_wo, row0, ro);
    fq_unit_regs pd[NPD][2], po[NPO][2];
    if (!summing) {
        rows_issue<TYPE, 2, NPD>(rd, units_d, pd);
        rows_issue<TYPE, 2, NPO>(ro, units_o, po);
    }
#pragma unroll
    for (int k = 0; k < NTF; ++k) { const int64_t i = (int64_t) k * nt + tid; if (i < nvec_ff) ((u32x4 *) img_ff)[i] = tf[k]; }
    for (int64_t i = (int64_t) NTF * nt + tid; i < nvec_ff; i += nt) ((u32x4 *) img_ff)[i] = src_ff[i];
    for (int64_t i = tid; i < nvec_at; i += nt) ((u32x4 *) img_att)[i] = src_at[i];
    __syncthreads();
    const fq_actcol col_d = actcol_at(img_ff, ACT, FF), col_o = actcol_at(img_att, ACT, E);
    float acc[2] = { 0.0f, 0.0f };
    unsigned sa_d[2], sa_o[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) { sa_d[r] = (unsigned)(uintptr_t)(strip + (size_t)(2 * (summing ? 0 : wid) + r) * sg.swt); sa_o[r] = sa_d[r] + 4u * sg.swd; }
    const float * srow_strip = strip + (size_t)(lane < 2 * ncw ? lane : 0) * sg.swt;
    const int n_pre = 64 * NPD < units_d ? 64 * NPD : units_d;
    float sd = 0.0f;
    if (!summing) {
        rows_consume<TYPE, 2, NPD, true>(pd, units_d, col_d, acc, sa_d);
        ref_count(DCNT_A);
        rows_dot_from<TYPE, 2, (decode_cfg<TYPE>::four_bit ? 5 : 2), true>(rd, units_d, 64 * NPD, col_d, acc, sa_d);
        ref_count(DCNT_B);
    } else {
        ref_wait(DCNT_A, (unsigned) ncw, err);
        if (!(dbgm & 2)) sd = fq_ref_chain(srow_strip, n_pre, sd);
        ref_wait(DCNT_B, (unsigned) ncw, err);
        if (!(dbgm & 2)) sd = fq_ref_chain(srow_strip + n_pre, units_d - n_pre, sd);
    }
    if (gran && !summing) {
        // the attention output (k_attn_out's sweep) by the consumer waves: each re-reads its share of the granules until all of them carry this launch's tag.
        // No workgroup barrier behind it: the summing wave is still adding Wdown's terms and must not hold the consumers up -- they count themselves in
        // (the counter add follows the image stores in the wave's LDS queue) and go on when all have
        const int64_t nwords = (E >> 2) + 2 * (E >> 5);
        unsigned * dstw = (unsigned *) img_att;
        constexpr int NG = 3;
        const int ntc = nt - 64;                                         // consumer threads: tid < ntc
        for (int64_t base = 0; base < nwords; base += (int64_t) NG * ntc) {
            unsigned v[NG];
            for (unsigned spins = 0;; ++spins) {
                bool ok = true;
#pragma unroll
                for (int k = 0; k < NG; ++k) {
                    const int64_t i = base + (int64_t) k * ntc + tid;
                    const unsigned long long x = __hip_atomic_load(gran + (i < nwords ? i : nwords - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    v[k] = (unsigned) x; ok = ok && (unsigned)(x >> 32) == epoch;
                }
                if (__all(ok)) break;
                if (spins > (1u << 20)) { if (lane == 0) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                __builtin_amdgcn_s_sleep(4);
            }
#pragma unroll
            for (int k = 0; k < NG; ++k) {
                const int64_t i = base + (int64_t) k * ntc + tid;
                if (i < nwords) asm volatile("ds_write_b32 %0, %1" :: "v"((unsigned)(uintptr_t)(dstw + i)), "v"(v[k]) : "memory");
            }
        }
        ref_count(SWEPT);
        ref_wait(SWEPT, (unsigned) ncw, err);
    }
    // Wo: one chain behind the workgroup's last dot. (-DFQ_REF_WO_PIPE=1: the pre-issued columns consumed and reported one by one, the summing wave adding a
    // column's 64 terms while the consumers are in the dots of the next -- built and measured SLOWER, 912 against 925 tok/s A/B/A/B on one box
    // (profiles/r06i_ab_wo_pipe.txt): three more polled waits and three chain start-ups cost more than the ~100 adds they take off the tail)
#ifndef FQ_REF_WO_PIPE
#define FQ_REF_WO_PIPE 0
#endif
    constexpr int NOC = !FQ_REF_WO_PIPE ? 0 : (NPO < 3 ? NPO : 3);   // columns reported singly (counters OCOL + 4 i); 0: one chain behind the last dot (A/B)
    if (!summing) {
#pragma unroll
        for (int i = 0; i < NPO; ++i) {
            const fq_unit_regs (&one)[1][2] = *(const fq_unit_regs (*)[1][2]) &po[i];
            rows_consume<TYPE, 2, 1, true>(one, units_o, col_o, acc, sa_o, i);
            if (i < NOC) ref_count(OCOL + 4u * (unsigned) i);
        }
        rows_dot_from<TYPE, 2, 2, true>(ro, units_o, 64 * NPO, col_o, acc, sa_o);
        ref_count(OCNT);
    } else {
        float so = 0.0f;
        int done = 0;
        if (!(dbgm & 4)) {
#pragma unroll
            for (int i = 0; i < NOC; ++i) {
                const int n = 64 * (i + 1) < units_o ? 64 : (units_o - 64 * i > 0 ? units_o - 64 * i : 0);
                ref_wait(OCOL + 4u * (unsigned) i, (unsigned) ncw, err);
                so = fq_ref_chain(srow_strip + sg.swd + 64 * i, n, so);
                done += n;
            }
        }
        ref_wait(OCNT, (unsigned) ncw, err);
        if (!(dbgm & 4)) so = fq_ref_chain(srow_strip + sg.swd + done, units_o - done, so);
        if (slive) g.dst[srow] = (sd + so) + sres;                                               // libfalcon.cpp:2399-2400
    }
}

template <int TYPE>
__global__ void __launch_bounds__(768) k_gemv_out_ref(fq_gemv_out_args a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    gemv_out_ref_body<TYPE>(a, smem, (int) blockIdx.x, nullptr, 0u, nullptr);
}
template <int TYPE>
__global__ void __launch_bounds__(768) k_attn_out_ref(FQ_AO_PARAMS fq_attn_out_args a) {
    FQ_AO_TAKE
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const unsigned epoch = *a.epoch_word;
    if ((int) blockIdx.x < a.n_attn) {
        const int tid = threadIdx.x, grp = tid >> 8, gtid = tid & 255;
        if (grp >= a.heads_per_wg) return;
        int h = (int) blockIdx.x * a.heads_per_wg + grp;
        const bool live = h < a.at.H;
        if (!live) h = a.at.H - 1;
        attn_decode_group<true, true>(a.at, h, live, gtid, smem + (size_t) grp * a.attn_lds_group, nullptr, fq_publish{ a.gran, epoch });
        return;
    }
    gemv_out_ref_body<TYPE>(a.g, smem, (int) blockIdx.x - a.n_attn, a.gran, epoch, a.err, a.ref_debug);
}
// waves per workgroup of the reference-order output launches: the most (<= 12) whose strip fits the LDS next to the two images
static int fq_out_ref_waves(int act, int64_t FF, int64_t E) {
    for (int nw = 12; nw >= 3; --nw) if (fq_out_ref_lds(act, FF, E, nw) <= 160 * 1024) return nw;
    return 0;
}
static void gemv_out_ref_launch(const fq_gemv_out_args & a, int n_cu, hipStream_t st) {
    const int type = a.w_wo.type, act = fq_desc(type).act_type;
    if (!fq_legacy_type(type) || a.w_down.type != type || !a.att_image || a.w_wo.K % 32 || a.w_down.K % 32) { fprintf(stderr, "ggml-hip: gemv_out: the fast reference order covers blocks of one legacy format\n"); exit(1); }
    const int nw = fq_out_ref_waves(act, a.w_down.K, a.w_wo.K);
    if (!nw) { fprintf(stderr, "ggml-hip: gemv_out: the term strips do not fit the LDS\n"); exit(1); }
    size_t lds = fq_out_ref_lds(act, a.w_down.K, a.w_wo.K, nw);
    const unsigned blocks = (unsigned)((a.w_wo.M + 2 * (nw - 1) - 1) / (2 * (nw - 1)));
    if ((int) blocks <= n_cu && lds < 84 * 1024) lds = 84 * 1024;
#define FQ_CASE(T) case T: { static size_t g = 0; if (lds > 64 * 1024 && lds > g) { HIP_CHECK(hipFuncSetAttribute((const void *) k_gemv_out_ref<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds)); g = lds; } \
        FQ_LAUNCH_PROF((k_gemv_out_ref<T>), dim3(blocks), dim3(64 * nw), lds, st, a); } break;
    switch (type) { FQ_CASE(FQ_Q4_0) FQ_CASE(FQ_Q4_1) FQ_CASE(FQ_Q5_0) FQ_CASE(FQ_Q5_1) FQ_CASE(FQ_Q8_0) default: break; }
#undef FQ_CASE
}
// the merged attention + output launch in the fast reference order; false: the grid would not be resident at once (nothing launched)
bool fq_launch_attn_out_ref(const fq_gemv_out_args & g, const float * qkv, int H, int HKV, const int * n_past_dev, int max_n_kv,
                            const float * rope_cs, const float * rope_cur, float * k_cache, float * v_cache, const uint16_t * exp_table,
                            int att_act_type, unsigned long long * gran, const unsigned * epoch_word, unsigned * err, int n_cu, hipStream_t st) {
    FQ_TL(st, "attn_out_ref");
    const int type = g.w_wo.type, act = fq_desc(type).act_type;
    if (!fq_legacy_type(type) || g.w_down.type != type || !g.att_image || g.w_wo.K % 32 || g.w_down.K % 32) return false;
    const int nw = 12, hpw = 2;
    if (fq_out_ref_lds(act, g.w_down.K, g.w_wo.K, nw) > 160 * 1024) return false;
    const int n_attn = (H + hpw - 1) / hpw;
    const int n_mv = (int)((g.w_wo.M + 2 * (nw - 1) - 1) / (2 * (nw - 1)));
    if (n_attn + n_mv > n_cu) return false;
    const size_t lds_group = (attn_decode_lds(max_n_kv) + 15) & ~(size_t) 15;
    const size_t lds_mv = fq_out_ref_lds(act, g.w_down.K, g.w_wo.K, nw);
    size_t lds = lds_group * hpw > lds_mv ? lds_group * hpw : lds_mv;
    if (lds < 84 * 1024) lds = 84 * 1024;
    if (lds > 160 * 1024) return false;
    fq_attn_out_args a{};
    a.g = g;
    a.at = fq_attn_decode_args{ qkv, H, HKV, n_past_dev, rope_cs, k_cache, v_cache, exp_table, nullptr, const_cast<uint8_t *>(g.att_image), att_act_type, max_n_kv, rope_cur, nullptr, nullptr, nullptr };
    a.gran = gran; a.epoch_word = epoch_word; a.err = err; a.n_attn = n_attn; a.n_mv = n_mv; a.heads_per_wg = hpw; a.attn_lds_group = (int) lds_group;
    static const int dbgm = getenv("FQ_REF_DBG") ? atoi(getenv("FQ_REF_DBG")) : 0;
    a.ref_debug = dbgm;
    const int grid = n_attn + n_mv;
#define FQ_CASE(T) case T: { static size_t gmax = 0; if (lds > gmax) { HIP_CHECK(hipFuncSetAttribute((const void *) k_attn_out_ref<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds)); gmax = lds; } \
        FQ_LAUNCH_PROF((k_attn_out_ref<T>), dim3((unsigned) grid), dim3(64 * nw), lds, st, FQ_AO_LEAD a); } break;
    switch (type) { FQ_CASE(FQ_Q4_0) FQ_CASE(FQ_Q4_1) FQ_CASE(FQ_Q5_0) FQ_CASE(FQ_Q5_1) FQ_CASE(FQ_Q8_0) default: return false; }
#undef FQ_CASE
    return true;
}

// the merged form's own conditions (fq_launch_attn_out with ln == nullptr launches exactly when this is true)
bool fq_attn_out_fits(const fq_gemv_out_args & g, int H, int max_n_kv, int n_cu) {
    const int act = fq_desc(g.w_wo.type).act_type;
    const int nw = 12, hpw = 2;
    const int n_attn = (H + hpw - 1) / hpw;
    const int n_mv = (int)((g.w_wo.M + 2 * nw - 1) / (2 * nw));
    const size_t lds_group = (attn_decode_lds(max_n_kv) + 15) & ~(size_t) 15;
    const size_t lds_mv = fq_act_col_bytes(act, g.w_down.K) + fq_act_col_bytes(act, g.w_wo.K) + (g.att_image ? 0 : (size_t) g.w_wo.K * 4) + 16;
    size_t lds = lds_group * hpw > lds_mv ? lds_group * hpw : lds_mv;
    static const int rounds = getenv("FQ_ATTN_OUT_ROUNDS") ? atoi(getenv("FQ_ATTN_OUT_ROUNDS")) : 1;
    if (lds > 160 * 1024) return false;
    if (n_attn + n_mv > n_cu && !(rounds >= 2 && n_attn <= n_cu / 2 && n_attn + n_mv <= 2 * n_cu)) return false;
    if (lds < 84 * 1024) lds = 84 * 1024;
    return lds <= 160 * 1024;
}

// true (and launched) when the merged form applies: every workgroup resident at once
// ln == nullptr: k_attn_out. ln != nullptr: k_attn_out_ln with *ln as the second phase (xgran: >= n_embd granules).
bool fq_launch_attn_out(const fq_gemv_out_args & g, const float * qkv, int H, int HKV, const int * n_past_dev, int max_n_kv,
                        const float * rope_cs, const float * rope_cur, float * k_cache, float * v_cache, const uint16_t * exp_table,
                        int att_act_type, unsigned long long * gran, const unsigned * epoch_word, unsigned * err, int n_cu, hipStream_t st,
                        const fq_gemv_ln_args * ln, unsigned long long * xgran) {
    FQ_TL(st, "attn_out");
    const int type = g.w_wo.type, act = fq_desc(type).act_type;
    const int nw = 12, hpw = 2;             // 2 heads per attention workgroup: the attention is instruction-issue bound per SIMD
    const int n_attn = (H + hpw - 1) / hpw;
    const int n_mv = (int)((g.w_wo.M + 2 * nw - 1) / (2 * nw));
    const size_t lds_group = (attn_decode_lds(max_n_kv) + 15) & ~(size_t) 15;
    const size_t lds_mv = fq_act_col_bytes(act, g.w_down.K) + fq_act_col_bytes(act, g.w_wo.K) + (g.att_image ? 0 : (size_t) g.w_wo.K * 4) + 16;
    size_t lds = lds_group * hpw > lds_mv ? lds_group * hpw : lds_mv;
    // more workgroups than CUs (Falcon-40B width: 64 + 342): only with FQ_ATTN_OUT_ROUNDS=2 -- the attention workgroups are the
    // PREFIX of the grid, the dispatcher places workgroups in index order, so every producer is resident before any consumer
    // spins on it and the second round of mat-vec workgroups follows as the first retires (measured against the three-launch
    // form, DESIGN "Falcon-40B width")
    static const int rounds = getenv("FQ_ATTN_OUT_ROUNDS") ? atoi(getenv("FQ_ATTN_OUT_ROUNDS")) : 1;
    if (lds > 160 * 1024) return false;
    if (n_attn + n_mv > n_cu && !(rounds >= 2 && !ln && n_attn <= n_cu / 2 && n_attn + n_mv <= 2 * n_cu)) return false;
    fq_attn_out_args a{};
    a.g = g;
    a.at = fq_attn_decode_args{ qkv, H, HKV, n_past_dev, rope_cs, k_cache, v_cache, exp_table, const_cast<float *>(g.att_image ? nullptr : g.att),
                                const_cast<uint8_t *>(g.att_image), att_act_type, max_n_kv, rope_cur, nullptr, nullptr, nullptr };
    a.gran = gran; a.epoch_word = epoch_word; a.err = err; a.n_attn = n_attn; a.n_mv = n_mv; a.heads_per_wg = hpw; a.attn_lds_group = (int) lds_group;
    int grid = n_attn + n_mv;
    fq_gemv_ln_args b{};
    if (ln) {
        // second phase: same format, 12-wave workgroups, the row in registers, everything resident at once
        b = *ln;
        for (int s = 0; s < b.nseg; ++s) if (b.seg[s].w.type != type) return false;
        int lnw = 4, npass = 2;
        gemv_ln_shape(b, n_cu, lnw, npass);
        if (lnw != nw || (b.E >> 2) > 3 * 64 * nw) return false;
        b.npass = npass;
        const int rows = 4 * npass * nw;
        int blocks = 0;
        for (int s = 0; s < b.nseg; ++s) { b.seg[s].block_begin = blocks; blocks += (int)((b.seg[s].w.M + rows - 1) / rows); }
        b.n_blocks = blocks;
        if (blocks > n_cu) return false;
        const size_t lds_ln = fq_gemv_ln_lds(type, b.E);
        if (lds_ln > lds) lds = lds_ln;
        if (blocks > grid) grid = blocks;
    }
    if (lds < 84 * 1024) lds = 84 * 1024;                                 // one workgroup per CU
    if (lds > 160 * 1024) return false;
#define FQ_CASE(T) case T: if (ln) { \
        static size_t gmax2 = 0; if (lds > gmax2) { HIP_CHECK(hipFuncSetAttribute((const void *) k_attn_out_ln<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds)); gmax2 = lds; } \
        FQ_LAUNCH_PROF((k_attn_out_ln<T>), dim3((unsigned) grid), dim3(64 * nw), lds, st, a, b, xgran); \
    } else { \
        static size_t gmax = 0; if (lds > gmax) { HIP_CHECK(hipFuncSetAttribute((const void *) k_attn_out<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds)); gmax = lds; } \
        FQ_LAUNCH_PROF((k_attn_out<T>), dim3((unsigned) grid), dim3(64 * nw), lds, st, FQ_AO_LEAD a); } break;
    switch (type) {
        FQ_CASE(FQ_Q4_0) FQ_CASE(FQ_Q4_1) FQ_CASE(FQ_Q5_0) FQ_CASE(FQ_Q5_1) FQ_CASE(FQ_Q8_0)
        FQ_CASE(FQ_Q2_K) FQ_CASE(FQ_Q3_K) FQ_CASE(FQ_Q4_K) FQ_CASE(FQ_Q5_K) FQ_CASE(FQ_Q6_K)
        default: fprintf(stderr, "ggml-hip: attn_out: unsupported weight type %d\n", type); exit(1);
    }
#undef FQ_CASE
    return true;
}
