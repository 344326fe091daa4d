// kernels_gemm_skinny_k.hip -- the small-batch mat-mul (5 <= N <= 16 columns per pass) for the k-quants at model widths (gfx950), and the sum launches
// of the self-paced forms. ggml_vec_dot_q{2,3,4,5,6}_K_q8_K (k_quants.c:1005-2789) once per row and column, the matrix streamed once by LDS-DMA, the columns
// resident in LDS, v_mfma_i32_16x16x32_i8 per 32-element group; each form has a fixed association that the oracle restates (oracle_quants.c: g_kseg).
//   k_gemm_skinny_q4k   Q4_K / Q5_K: a workgroup owns a PAIR of K shares (the bytes of two of a super-block's four 32-byte chunks), segments of 32 super-blocks
//   k_gemm_skinny_q2k   Q2_K / Q3_K: a wave owns all four shares of a tile, the sub-block scales go INTO the matrix operand (v_perm_b32 table look-up), segments of 16
//   k_gemm_skinny_q6k   Q6_K: the same frame, two half-zeroed operands per group, scales by v_mad_i32_i24
//   k_skinny_sum4*      partial sums -> result (+ epilogue; + GELU and the next mat-mul's Q8_K image; Wo + Wdown + residual in one launch)
#include "fq_skinny_dev.h"

// =============================================================================================== Q4_K: a pair of K shares per workgroup
// The k-quants' mat-mul for 5..16 columns (ggml_vec_dot_q4_K_q8_K, k_quants.c:1751-2055, once per row and column). Association: K is cut into SEGMENTS
// of 32 super-blocks; inside a segment k_gemm_q<S = 4>'s order -- the super-block's eight 32-element groups are dealt to four partial sums (group g ->
// g mod 4), each adds (d dy) * (its two groups' integer sum) per super-block, the last one (d dy) isum - (dmin dy) msum, segment value ((P0 + P1) + P2) + P3
// -- and the segment values are added left to right (k_skinny_sum4). The oracle restates it (orc_set_sum_order: split 4 with segments); rows of up to
// 32 super-blocks (Falcon-40B's K = 8192) are exactly the tile GEMM's four-sum order.
// Q4_K packs groups 2c (low nibbles) and 2c + 1 (high nibbles) into the 32 bytes of chunk c, so shares {0, 1} live in chunks {0, 2} and shares
// {2, 3} in chunks {1, 3}: a workgroup owns one PAIR q of shares of one segment = half of the rows' quant bytes there and half of the columns' bytes,
// which stay resident in LDS. Every wave runs the two accumulator chains of one 16-row tile per row block and stages its own weights by LDS-DMA into a
// private ring, paced by vmcnt alone, the pipeline running on across the workgroup's row blocks. Both pairs of a (row block, segment) sit on one XCD (the
// 64-byte segments their 32-byte chunks share come from HBM once). Per stage (4 super-blocks) and row: 16 quant pieces | 3 of packed scales | 2 of d, dmin.
constexpr int KQ_SEG = 32;                 // super-blocks per segment
// Q5_K is the same kernel plus the plane of fifth bits (32 bytes per super-block: bit g of byte l belongs to element l of group g; both pairs need all
// of it): 8 more pieces per row and stage, the bit of the lane's 8 elements or-ed into the nibbles on the way to the matrix instruction.
template <int TYPE> struct kq_fmt {
    static constexpr bool Q5 = TYPE == FQ_Q5_K;
    static constexpr int ROWP = Q5 ? 29 : 21;                              // 16-byte slots per row and stage (odd: the 16 rows of a tile in distinct banks)
    static constexpr int ROWB = 16 * ROWP, WSTAGE = 16 * ROWB, KOPS = (16 * ROWP + 63) / 64;
    static constexpr int QHOFF = 256, SOFF = Q5 ? 384 : 256, DOFF = SOFF + 48;      // LDS offsets inside a row's stage: quants | (fifth bits) | scales | d, dmin
    static constexpr int COLB = Q5 ? 1408 : 1152;                          // bytes of a full column (8 super-blocks) of the device layout
    static constexpr int PRE_QH = 128, PRE_SC = Q5 ? 160 : 128, PRE_DM = PRE_SC + 12;   // bytes per super-block in front of each plane
};
struct kq_plan { int tqs; size_t rings, cols, dy, gs, total; };
template <int TYPE> static __host__ __device__ inline kq_plan kq_lds(int seg_sb, int T, int nbw) {
    constexpr int KQ_WSTAGE = kq_fmt<TYPE>::WSTAGE;
    kq_plan p;
    const int qb = seg_sb * 128;
    p.tqs = qb + ((16 - (qb & 255)) & 255);                                 // column pitch = 16 mod 256 bytes: the 16 tokens of an operand read in distinct banks
    p.rings = (size_t) nbw * T * KQ_WSTAGE; p.cols = (size_t) SK_TN * p.tqs; p.dy = (size_t) seg_sb * SK_TN * 4; p.gs = (size_t) seg_sb * SK_TN * 16;
    p.total = p.rings + p.cols + p.dy + p.gs + 16;                          // (+ 16 zero bytes: the mins' matrix operands of the lanes that carry none)
    return p;
}

// part: [segment][share][16 columns][mstride rows]
template <int TYPE, int NBW>
__global__ void __launch_bounds__(64 * KS_TMAX) k_gemm_skinny_q4k(fq_weight w, fq_act act, int N, float * part, int64_t mstride, int T, int nrb, int nslots, int seg_sb, int dbg) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    typedef kq_fmt<TYPE> F;
    constexpr int KQ_ROWP = F::ROWP, KQ_ROWB = F::ROWB, KQ_WSTAGE = F::WSTAGE, KQ_KOPS = F::KOPS;
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t K = w.K, M = w.M;
    const int nsb = (int) w.nblk;
    const int xcd = (int) blockIdx.x & 7, kk = (int) blockIdx.x >> 3;
    const int slot = (kk >> 1) * 8 + xcd, q = kk & 1;                      // row blocks slot, slot + nslots, ..; pair q = shares 2 q, 2 q + 1
    if (slot >= nrb) return;
    const int seg = (int) blockIdx.y;
    const int sb0 = seg * seg_sb, nsbs = nsb - sb0 < seg_sb ? nsb - sb0 : seg_sb;
    const size_t img = fq_act_col_bytes(FQ_Q8_K, K);
    const kq_plan P = kq_lds<TYPE>(seg_sb, T, NBW);
    const int TQS = P.tqs;
    uint8_t * cols = smem + P.rings;                                       // [16 columns][TQS]: per super-block 4 groups x 32 B: 2 q, 2 q + 1, 2 q + 4, 2 q + 5
    float   * dyT  = (float *)(cols + P.cols);                             // [super-block][16]: the columns' d
    uint8_t * gsT  = (uint8_t *) dyT + P.dy;                               // [super-block][16][16 B]: sub-block sums gs_j = 64 hi_j + lo_j as bytes hi_0..7 | lo_0..7
    const int l16 = lane & 15, kq = lane >> 4;
    int nmine = 0;                                                         // the wave's tiles: one per row block of the workgroup
    for (int rb = slot; rb < nrb; rb += nslots) if (((int64_t) rb * T + wid) * 16 < M) ++nmine;
    // lane L = 64 k + lane of a stage's DMA instruction k is (row L / 21, piece L % 21) -- LDS rows are exactly 21 slots
    unsigned poff[KQ_KOPS], rowb[KQ_KOPS]; int pkind[KQ_KOPS];
#pragma unroll
    for (int k = 0; k < KQ_KOPS; ++k) {
        const int L = 64 * k + lane, row = (L / KQ_ROWP) & 15, p = L % KQ_ROWP;
        constexpr int NQH = F::Q5 ? 8 : 0;                                  // pieces of fifth bits (4 super-blocks x 32 B)
        if (p < 16)            { poff[k] = (unsigned)((p >> 2) * 128 + ((p >> 1) & 1) * 64 + (p & 1) * 16); pkind[k] = 0; }
        else if (p < 16 + NQH) { poff[k] = (unsigned)(16 * (p - 16)); pkind[k] = 3; }
        else if (p < 19 + NQH) { poff[k] = (unsigned)(16 * (p - 16 - NQH)); pkind[k] = 1; }
        else                   { poff[k] = (unsigned)(16 * (p - 19 - NQH)); pkind[k] = 2; }
        rowb[k] = (unsigned) row * (unsigned) w.row_stride;
    }
    const unsigned rs16 = (unsigned) w.row_stride - 16u;
    uint8_t * myring = smem + (size_t) wid * NBW * KQ_WSTAGE;
    const unsigned ring_lds = __builtin_amdgcn_readfirstlane(sk_lds(myring));
    const int nst = (nsbs + 3) / 4;                                        // stages per row block
    const int total = nmine * nst;
    auto issue = [&](int u) {
        const int rbi = u / nst, sp = u - rbi * nst;
        const int64_t mt = ((int64_t)(slot + rbi * nslots) * T + wid) * 16;
        const uint8_t * wbase = sk_uniform(w.plane[0] + (size_t) mt * w.row_stride);
        const int gsb = sb0 + 4 * sp, c = gsb >> 3, in = gsb & 7;
        const int nbc = nsb - 8 * c < 8 ? nsb - 8 * c : 8;
        const unsigned b0 = (unsigned)(c * F::COLB + in * 128 + q * 32);
        const unsigned bh = (unsigned)(c * F::COLB + nbc * F::PRE_QH + in * 32);
        const unsigned b1 = (unsigned)(c * F::COLB + nbc * F::PRE_SC + in * 12);
        const unsigned b2 = (unsigned)(c * F::COLB + ((nbc * F::PRE_DM) & ~15) + in * 4);
        const unsigned dst = ring_lds + (unsigned)((u % NBW) * KQ_WSTAGE);
#pragma unroll
        for (int k = 0; k < KQ_KOPS; ++k) {
            unsigned o = poff[k] + (pkind[k] == 0 ? b0 : (pkind[k] == 1 ? b1 : (pkind[k] == 2 ? b2 : bh)));
            o = o < rs16 ? o : rs16;                                       // (a partial last column: pieces beyond its blocks are never used)
            if (64 * k + lane < 16 * KQ_ROWP) sk_dma(wbase, rowb[k] + o, dst + (unsigned)(k * 1024));
        }
    };
    // ---- the weights' first stages are on their way while the columns are staged
    if (!(dbg & 8)) { for (int u = 0; u < NBW - 1 && u < total; ++u) issue(u); }
    // ---- the segment's columns: the pair's groups (8 pieces per super-block and column, 8 super-blocks per DMA instruction), d, sub-block sums
    {
        const unsigned last = (unsigned)(K - 16);
        for (int t = wid; t < SK_TN; t += T) {
            const uint8_t * base = sk_uniform(act.base + (size_t)(t < N ? t : N - 1) * img);
            const unsigned tb = sk_lds(cols) + (unsigned)(t * TQS);
            for (int j0 = 0; j0 < nsbs; j0 += 8) {
                const int sb = j0 + (lane >> 3), p = lane & 7;
                unsigned vq = (unsigned)((sb0 + sb) * 256 + 64 * q + 128 * (p >> 2) + 16 * (p & 3));
                vq = vq < last ? vq : last;
                if (sb < nsbs && !(dbg & 4)) sk_dma(base, vq, tb + (unsigned)(j0 * 128));
            }
        }
        const size_t aux = fq_act_aux_off(FQ_Q8_K, K);
        if (tid < 4) ((uint32_t *)(gsT + P.gs))[tid] = 0u;
        for (int e = tid; e < nsbs * SK_TN; e += (int) blockDim.x) {
            const int sbl = e >> 4, tok = e & 15;
            const uint8_t * tp = act.base + (size_t)(tok < N ? tok : N - 1) * img;
            dyT[e] = ((const float *)(tp + K))[sb0 + sbl];
            const uint32_t * bs = (const uint32_t *)(tp + aux) + (size_t)(sb0 + sbl) * 8;      // 16 x int16 bsums: gs_j = bsums[2 j] + bsums[2 j + 1]
            uint32_t hi[2] = { 0u, 0u }, lo[2] = { 0u, 0u };
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint32_t v = bs[j];
                const int gs = (int)(int16_t)(v & 0xFFFFu) + (int)(int16_t)(v >> 16);
                hi[j >> 2] |= ((uint32_t)(gs >> 6) & 0xFFu) << (8 * (j & 3));
                lo[j >> 2] |= ((uint32_t) gs & 63u) << (8 * (j & 3));
            }
            *(uint4 *)(gsT + (size_t) e * 16) = make_uint4(hi[0], hi[1], lo[0], lo[1]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();                                                       // the only barrier: the columns are in LDS

    auto run = [&](auto pair_tag) __attribute__((always_inline)) {
        constexpr int Q = decltype(pair_tag)::value;
        float * part0 = part + ((size_t)(4 * seg + 2 * Q) * SK_TN) * (size_t) mstride, * part1 = part0 + (size_t) SK_TN * (size_t) mstride;
        int u = 0;
        for (int rbi = 0; rbi < nmine; ++rbi) {
            const int64_t m = ((int64_t)(slot + rbi * nslots) * T + wid) * 16 + l16;
            float acc0[4] = { 0.0f, 0.0f, 0.0f, 0.0f }, acc1[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
            for (int sp = 0; sp < nst; ++sp, ++u) {
                if (u + NBW - 1 < total && !(dbg & 8)) issue(u + NBW - 1);
                {
                    const int later = (dbg & 8) ? 0 : (total - 1 - u < NBW - 1 ? total - 1 - u : NBW - 1);
                    if (later >= 2)      sk_wait_vm_upto(2 * KQ_KOPS);
                    else if (later == 1) sk_wait_vm_upto(KQ_KOPS);
                    else                 asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                if (dbg & 16) continue;
                const uint8_t * wr = myring + (size_t)(u % NBW) * KQ_WSTAGE + l16 * KQ_ROWB;
                const int gsb = sb0 + 4 * sp, c = gsb >> 3;
                const int nbc = nsb - 8 * c < 8 ? nsb - 8 * c : 8;
                const int p2d = (nbc * F::PRE_DM) & 15;                    // d, dmin: offset from the boundary the DMA started at
                const int ns = nsbs - 4 * sp < 4 ? nsbs - 4 * sp : 4;      // super-blocks in this stage
                const uint8_t * tqp = cols + (size_t) l16 * TQS + (size_t)(4 * sp) * 128 + 8 * kq;
                const uint8_t * dyp = (const uint8_t *) dyT + (size_t)(4 * sp) * 64 + 16 * kq;
                // the mins' sum_j m_j gs_j as two more matrix instructions (hi and lo bytes of gs): k slots 0..7 = j, carried by the lanes kq = 0 -- token
                // l16's bytes as the A operand, row l16's mins as B; the other lanes read zeros
                const uint8_t * gsp = kq == 0 ? gsT + (size_t)(4 * sp) * 256 + 16 * l16 : gsT + P.gs;
                const int gstep = kq == 0 ? 256 : 0;
                const uint32_t kq0 = kq == 0 ? 0xFFFFFFFFu : 0u;
                struct kq_ops { sk_v2i xa[4], raw0, raw1, qh; uint32_t u0, u1, u2, dm; float4 dy; uint4 gs; };
                auto load_ops = [&](int i) __attribute__((always_inline)) {
                    kq_ops o;
#pragma unroll
                    for (int gi = 0; gi < 4; ++gi) o.xa[gi] = *(const sk_v2i *)(tqp + 128 * i + 32 * gi);
                    o.raw0 = *(const sk_v2i *)(wr + 64 * i + 8 * kq);
                    o.raw1 = *(const sk_v2i *)(wr + 64 * i + 32 + 8 * kq);
                    if constexpr (F::Q5) o.qh = *(const sk_v2i *)(wr + F::QHOFF + 32 * i + 8 * kq); else o.qh = sk_v2i{ 0, 0 };
                    o.u0 = *(const uint32_t *)(wr + F::SOFF + 12 * i); o.u1 = *(const uint32_t *)(wr + F::SOFF + 4 + 12 * i); o.u2 = *(const uint32_t *)(wr + F::SOFF + 8 + 12 * i);
                    o.dm = *(const uint32_t *)(wr + F::DOFF + p2d + 4 * i);
                    o.dy = *(const float4 *)(dyp + 64 * i);
                    if constexpr (Q == 1) o.gs = *(const uint4 *)(gsp + gstep * i);
                    return o;
                };
                struct kq_c { sk_v4i c[6]; };
                auto run_mfma = [&](const kq_ops & o) __attribute__((always_inline)) {
                    kq_c r;
                    const sk_v4i z = { 0, 0, 0, 0 };
                    sk_v2i l0 = { (int)((uint32_t) o.raw0.x & 0x0F0F0F0Fu), (int)((uint32_t) o.raw0.y & 0x0F0F0F0Fu) };
                    sk_v2i h0 = { (int)(((uint32_t) o.raw0.x >> 4) & 0x0F0F0F0Fu), (int)(((uint32_t) o.raw0.y >> 4) & 0x0F0F0F0Fu) };
                    sk_v2i l1 = { (int)((uint32_t) o.raw1.x & 0x0F0F0F0Fu), (int)((uint32_t) o.raw1.y & 0x0F0F0F0Fu) };
                    sk_v2i h1 = { (int)(((uint32_t) o.raw1.x >> 4) & 0x0F0F0F0Fu), (int)(((uint32_t) o.raw1.y >> 4) & 0x0F0F0F0Fu) };
                    if constexpr (F::Q5) {                                  // k_quants.c:2358-2364: bit g of qh[l] adds 16 to element l of group g
                        const uint32_t qx = (uint32_t) o.qh.x, qy = (uint32_t) o.qh.y;
                        auto fifth = [&](sk_v2i & v, int g) __attribute__((always_inline)) {
                            v.x |= (int)(((qx >> g) & 0x01010101u) << 4); v.y |= (int)(((qy >> g) & 0x01010101u) << 4);
                        };
                        fifth(l0, 2 * Q); fifth(h0, 2 * Q + 1); fifth(l1, 2 * Q + 4); fifth(h1, 2 * Q + 5);
                    }
                    r.c[0] = __builtin_amdgcn_mfma_i32_16x16x32_i8(__builtin_bit_cast(long, o.xa[0]), __builtin_bit_cast(long, l0), z, 0, 0, 0);     // group 2 Q
                    r.c[1] = __builtin_amdgcn_mfma_i32_16x16x32_i8(__builtin_bit_cast(long, o.xa[1]), __builtin_bit_cast(long, h0), z, 0, 0, 0);     // 2 Q + 1
                    r.c[2] = __builtin_amdgcn_mfma_i32_16x16x32_i8(__builtin_bit_cast(long, o.xa[2]), __builtin_bit_cast(long, l1), z, 0, 0, 0);     // 2 Q + 4
                    r.c[3] = __builtin_amdgcn_mfma_i32_16x16x32_i8(__builtin_bit_cast(long, o.xa[3]), __builtin_bit_cast(long, h1), z, 0, 0, 0);     // 2 Q + 5
                    if constexpr (Q == 1) {                                 // get_scale_min_k4's mins, four to a register
                        const sk_v2i mn = { (int)((o.u1 & 0x3F3F3F3Fu) & kq0), (int)((((o.u2 >> 4) & 0x0F0F0F0Fu) | ((o.u1 >> 2) & 0x30303030u)) & kq0) };
                        r.c[4] = __builtin_amdgcn_mfma_i32_16x16x32_i8(__builtin_bit_cast(long, sk_v2i{ (int) o.gs.x, (int) o.gs.y }), __builtin_bit_cast(long, mn), z, 0, 0, 0);
                        r.c[5] = __builtin_amdgcn_mfma_i32_16x16x32_i8(__builtin_bit_cast(long, sk_v2i{ (int) o.gs.z, (int) o.gs.w }), __builtin_bit_cast(long, mn), z, 0, 0, 0);
                    }
                    return r;
                };
                auto scale = [&](const kq_c & cc, const kq_ops & o) __attribute__((always_inline)) {
                    // the 6-bit scales / mins of the row's super-block, four to a register (get_scale_min_k4, k_quants.c:264-272)
                    const uint32_t scLo = o.u0 & 0x3F3F3F3Fu, scHi = (o.u2 & 0x0F0F0F0Fu) | ((o.u0 >> 2) & 0x30303030u);
                    const int sc0 = (int)((scLo >> (16 * Q)) & 0xFFu), sc1 = (int)((scLo >> (16 * Q + 8)) & 0xFFu);
                    const int sc2 = (int)((scHi >> (16 * Q)) & 0xFFu), sc3 = (int)((scHi >> (16 * Q + 8)) & 0xFFu);
                    const float d = fq_h2f((uint16_t) o.dm);
                    const float dyv[4] = { o.dy.x, o.dy.y, o.dy.z, o.dy.w };
                    float dmin = 0.0f;
                    if constexpr (Q == 1) dmin = fq_h2f((uint16_t)(o.dm >> 16));
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int is0 = __mul24(cc.c[0][r], sc0) + __mul24(cc.c[2][r], sc2);      // share 2 Q: groups 2 Q, 2 Q + 4 (|c| < 2^16, sc < 64)
                        const int is1 = __mul24(cc.c[1][r], sc1) + __mul24(cc.c[3][r], sc3);      // share 2 Q + 1
                        const float dd = d * dyv[r];                                        // k_quants.c:2027 (d = y.d * fp16(x.d))
                        acc0[r] = acc0[r] + dd * (float) is0;
                        float t1 = dd * (float) is1;
                        if constexpr (Q == 1) t1 = t1 - (dmin * dyv[r]) * (float)((cc.c[4][r] << 6) + cc.c[5][r]);      // the mins with the last share: - (dmin dy) sum_j m_j gs_j
                        acc1[r] = acc1[r] + t1;
                    }
                };
                if (ns == 4) {
                    kq_ops o[4]; kq_c c4[4];
                    o[0] = load_ops(0); o[1] = load_ops(1);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {                           // reads two super-blocks ahead, the matrix instructions one ahead of their scaling
                        if (k + 2 < 4) o[k + 2] = load_ops(k + 2);
                        c4[k] = run_mfma(o[k]);
                        __builtin_amdgcn_sched_barrier(0);
                        if (k > 0) scale(c4[k - 1], o[k - 1]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    scale(c4[3], o[3]);
                } else {
                    for (int i = 0; i < ns; ++i) { const kq_ops o = load_ops(i); const kq_c c1 = run_mfma(o); scale(c1, o); }
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = 4 * kq + r;
                if (n < N) { part0[(size_t) n * mstride + m] = acc0[r]; part1[(size_t) n * mstride + m] = acc1[r]; }
            }
        }
    };
    if (q) run(std::integral_constant<int, 1>{}); else run(std::integral_constant<int, 0>{});
}

// =============================================================================================== Q2_K: all four K shares per wave
// ggml_vec_dot_q2_K_q8_K (k_quants.c:1005-1306) for 5..16 columns. Q2_K keeps the four 32-element groups of a 128-element half in the SAME 32 bytes
// (bits 2t of byte l = element l of group t): every K share g mod 4 = t reads all of a row's quant bytes, so a wave owns a whole 16-row tile -- four
// accumulator chains -- and a workgroup keeps the full 16 columns of one SEGMENT of 16 super-blocks resident (64 KiB). Association: the segmented four-sum
// order of k_gemm_skinny_q4k with 16-super-block segments (the oracle restates it). The 16-element sub-blocks carry 4-bit scales: a lane's 8 elements of a
// group sit in ONE sub-block (lanes kq < 2: the first, kq >= 2: the second), so the scale goes INTO the matrix operand -- v_perm_b32 looks the 2-bit
// quants up in the lane's table {0, sc, 2 sc, 3 sc} (<= 45: int8) -- and ONE v_mfma_i32_16x16x32_i8 per group returns sum_b sc_b I_b; the two groups of
// a share chain through the accumulator operand: no integer multiply-adds at all. The mins: sum_b m_b bsum_b (16 sub-blocks) as two more matrix
// instructions on the hi / lo bytes of the Q8_K block sums (bsum = 64 hi + lo), k slots 0..15 carried by the lanes kq < 2.
// Per stage (4 super-blocks) and row: 16 quant pieces | 4 of scales | 1 of d, dmin = 21 slots, as in the Q4_K form.
constexpr int K2_SEG = 16;
// Q3_K (ggml_vec_dot_q3_K_q8_K, k_quants.c:1310-1746) is the same kernel with the plane of high bits (32 bytes per super-block, bit g of byte l = element
// l of group g; clear = subtract 4) and sixteen 6-bit scales - 32: the quant q_lo - 4 hbar times the scale s in [-32, 31] reaches 128, one past int8, so
// a group is TWO matrix instructions -- the table look-up of q_lo s in [-96, 93] and hbar (4 s) in [-128, 124], whose result is subtracted -- and no mins.
template <int TYPE> struct k2_fmt {
    static constexpr bool Q3 = TYPE == FQ_Q3_K;
    static constexpr int ROWP = Q3 ? 29 : 21, ROWB = 16 * ROWP, WSTAGE = 16 * ROWB, KOPS = (16 * ROWP + 63) / 64;
    static constexpr int HMOFF = 256, SOFF = Q3 ? 384 : 256, DOFF = Q3 ? 432 : 320;      // LDS offsets inside a row's stage: quants | (high bits) | scales | d (, dmin)
    static constexpr int COLB = Q3 ? 1760 : 1344;                          // bytes of a full column (16 super-blocks) of the device layout
    static constexpr int PRE_HM = 64, PRE_SC = Q3 ? 96 : 64, PRE_D = Q3 ? 108 : 80, SCB = Q3 ? 12 : 16, DB = Q3 ? 2 : 4;
};
struct k2_plan { int tqs; size_t rings, cols, dy, bs, total; };
template <int TYPE> static __host__ __device__ inline k2_plan k2_lds(int seg_sb, int T, int nbw) {
    constexpr int K2_WSTAGE = k2_fmt<TYPE>::WSTAGE;
    k2_plan p;
    const int qb = seg_sb * 256;
    p.tqs = qb + ((16 - (qb & 255)) & 255);
    p.rings = (size_t) nbw * T * K2_WSTAGE; p.cols = (size_t) SK_TN * p.tqs; p.dy = (size_t) seg_sb * SK_TN * 4; p.bs = k2_fmt<TYPE>::Q3 ? 0 : (size_t) seg_sb * SK_TN * 32;
    p.total = p.rings + p.cols + p.dy + p.bs + 32;                          // (+ 32 zero bytes: the mins' operands of the lanes that carry none)
    return p;
}

template <int TYPE, int NBW>
__global__ void __launch_bounds__(64 * KS_TMAX) k_gemm_skinny_q2k(fq_weight w, fq_act act, int N, float * part, int64_t mstride, int T, int nrb, int nslots, int seg_sb, int dbg) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    typedef k2_fmt<TYPE> F;
    constexpr bool Q3 = F::Q3;
    constexpr int K2_ROWP = F::ROWP, K2_ROWB = F::ROWB, K2_WSTAGE = F::WSTAGE, K2_KOPS = F::KOPS;
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t K = w.K, M = w.M;
    const int nsb = (int) w.nblk;
    const int slot = (int) blockIdx.x, seg = (int) blockIdx.y;
    if (slot >= nrb) return;
    const int sb0 = seg * seg_sb, nsbs = nsb - sb0 < seg_sb ? nsb - sb0 : seg_sb;
    const size_t img = fq_act_col_bytes(FQ_Q8_K, K);
    const k2_plan P = k2_lds<TYPE>(seg_sb, T, NBW);
    const int TQS = P.tqs;
    uint8_t * cols = smem + P.rings;                                       // [16 columns][TQS]: the segment's quants as they are
    float   * dyT  = (float *)(cols + P.cols);                             // [super-block][16]: the columns' d
    uint8_t * bsT  = (uint8_t *) dyT + P.dy;                               // [super-block][16][32 B]: block sums bsum_b = 64 hi_b + lo_b as bytes hi_0..15 | lo_0..15
    const int l16 = lane & 15, kq = lane >> 4;
    int nmine = 0;
    for (int rb = slot; rb < nrb; rb += nslots) if (((int64_t) rb * T + wid) * 16 < M) ++nmine;
    unsigned poff[K2_KOPS], rowb[K2_KOPS]; int pkind[K2_KOPS];
#pragma unroll
    for (int k = 0; k < K2_KOPS; ++k) {
        const int L = 64 * k + lane, row = (L / K2_ROWP) & 15, p = L % K2_ROWP;
        constexpr int NHM = Q3 ? 8 : 0, NSC = Q3 ? 3 : 4;                   // pieces of high bits (4 x 32 B), of scales (4 x 12 / 16 B)
        if (p < 16)                  { poff[k] = (unsigned)(16 * p); pkind[k] = 0; }
        else if (p < 16 + NHM)       { poff[k] = (unsigned)(16 * (p - 16)); pkind[k] = 3; }
        else if (p < 16 + NHM + NSC) { poff[k] = (unsigned)(16 * (p - 16 - NHM)); pkind[k] = 1; }
        else                         { poff[k] = (unsigned)(16 * (p - 16 - NHM - NSC)); pkind[k] = 2; }
        rowb[k] = (unsigned) row * (unsigned) w.row_stride;
    }
    const unsigned rs16 = (unsigned) w.row_stride - 16u;
    uint8_t * myring = smem + (size_t) wid * NBW * K2_WSTAGE;
    const unsigned ring_lds = __builtin_amdgcn_readfirstlane(sk_lds(myring));
    const int nst = (nsbs + 3) / 4;
    const int total = nmine * nst;
    auto issue = [&](int u) {
        const int rbi = u / nst, sp = u - rbi * nst;
        const int64_t mt = ((int64_t)(slot + rbi * nslots) * T + wid) * 16;
        const uint8_t * wbase = sk_uniform(w.plane[0] + (size_t) mt * w.row_stride);
        const int gsb = sb0 + 4 * sp, c = gsb >> 4, in = gsb & 15;
        const int nbc = nsb - 16 * c < 16 ? nsb - 16 * c : 16;
        const unsigned b0 = (unsigned)(c * F::COLB + in * 64);
        const unsigned bh = (unsigned)(c * F::COLB + nbc * F::PRE_HM + in * 32);
        const unsigned b1 = (unsigned)(c * F::COLB + nbc * F::PRE_SC + in * F::SCB);
        const unsigned b2 = (unsigned)(c * F::COLB + ((nbc * F::PRE_D + in * F::DB) & ~15));      // (Q2_K: aligned as it is; Q3_K: read from the boundary below)
        const unsigned dst = ring_lds + (unsigned)((u % NBW) * K2_WSTAGE);
#pragma unroll
        for (int k = 0; k < K2_KOPS; ++k) {
            unsigned o = poff[k] + (pkind[k] == 0 ? b0 : (pkind[k] == 1 ? b1 : (pkind[k] == 2 ? b2 : bh)));
            o = o < rs16 ? o : rs16;
            if (64 * k + lane < 16 * K2_ROWP) sk_dma(wbase, rowb[k] + o, dst + (unsigned)(k * 1024));
        }
    };
    if (!(dbg & 8)) { for (int u = 0; u < NBW - 1 && u < total; ++u) issue(u); }
    {
        const unsigned last = (unsigned)(K - 16);
        const int NW = (int)(blockDim.x >> 6);
        for (int t = wid; t < SK_TN; t += NW) {
            const uint8_t * base = sk_uniform(act.base + (size_t)(t < N ? t : N - 1) * img);
            const unsigned tb = sk_lds(cols) + (unsigned)(t * TQS);
            for (int j0 = 0; j0 < nsbs; j0 += 4) {
                unsigned vq = (unsigned)((sb0 + j0) * 256 + 16 * lane);
                vq = vq < last ? vq : last;
                if (j0 + (lane >> 4) < nsbs && !(dbg & 4)) sk_dma(base, vq, tb + (unsigned)(j0 * 256));
            }
        }
        const size_t aux = fq_act_aux_off(FQ_Q8_K, K);
        if (tid < 8) ((uint32_t *)(bsT + P.bs))[tid] = 0u;
        for (int e = tid; e < nsbs * SK_TN; e += (int) blockDim.x) {
            const int sbl = e >> 4, tok = e & 15;
            const uint8_t * tp = act.base + (size_t)(tok < N ? tok : N - 1) * img;
            dyT[e] = ((const float *)(tp + K))[sb0 + sbl];
            if constexpr (Q3) continue;
            const uint32_t * bs = (const uint32_t *)(tp + aux) + (size_t)(sb0 + sbl) * 8;
            uint32_t hi[4] = { 0u, 0u, 0u, 0u }, lo[4] = { 0u, 0u, 0u, 0u };
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint32_t v = bs[j];
                const int b0 = (int)(int16_t)(v & 0xFFFFu), b1 = (int)(int16_t)(v >> 16);
                hi[j >> 1] |= (((uint32_t)(b0 >> 6) & 0xFFu) << (16 * (j & 1))) | (((uint32_t)(b1 >> 6) & 0xFFu) << (16 * (j & 1) + 8));
                lo[j >> 1] |= (((uint32_t) b0 & 63u) << (16 * (j & 1))) | (((uint32_t) b1 & 63u) << (16 * (j & 1) + 8));
            }
            *(uint4 *)(bsT + (size_t) e * 32)      = make_uint4(hi[0], hi[1], hi[2], hi[3]);
            *(uint4 *)(bsT + (size_t) e * 32 + 16) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();                                                       // the only barrier: the columns are in LDS

    float * pbase = part + ((size_t)(4 * seg) * SK_TN) * (size_t) mstride;
    const size_t sstride = (size_t) SK_TN * (size_t) mstride;
    const int s01 = kq >> 1;                                               // which 16-element sub-block of a group the lane's elements sit in
    const bool mlane = kq < 2;                                             // lanes that carry the mins' k slots (sub-blocks 8 kq .. 8 kq + 7)
    int u = 0;
    for (int rbi = 0; rbi < nmine; ++rbi) {
        const int64_t m = ((int64_t)(slot + rbi * nslots) * T + wid) * 16 + l16;
        float acc[4][4];
#pragma unroll
        for (int t = 0; t < 4; ++t) { acc[t][0] = 0.0f; acc[t][1] = 0.0f; acc[t][2] = 0.0f; acc[t][3] = 0.0f; }
        for (int sp = 0; sp < nst; ++sp, ++u) {
            if (u + NBW - 1 < total && !(dbg & 8)) issue(u + NBW - 1);
            {
                const int later = (dbg & 8) ? 0 : (total - 1 - u < NBW - 1 ? total - 1 - u : NBW - 1);
                if (later >= 2)      sk_wait_vm_upto(2 * K2_KOPS);
                else if (later == 1) sk_wait_vm_upto(K2_KOPS);
                else                 asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            if (dbg & 16) continue;
            const uint8_t * wr = myring + (size_t)(u % NBW) * K2_WSTAGE + l16 * K2_ROWB;
            const int ns = nsbs - 4 * sp < 4 ? nsbs - 4 * sp : 4;
            const int gsb_ = sb0 + 4 * sp, cc_ = gsb_ >> 4, nbc_ = nsb - 16 * cc_ < 16 ? nsb - 16 * cc_ : 16;
            const int p3d = (nbc_ * F::PRE_D + (gsb_ & 15) * F::DB) & 15;      // d (, dmin): offset from the boundary the DMA started at (Q2_K: 0)
            const uint8_t * tqp = cols + (size_t) l16 * TQS + (size_t)(4 * sp) * 256 + 8 * kq;
            const uint8_t * dyp = (const uint8_t *) dyT + (size_t)(4 * sp) * 64 + 16 * kq;
            const uint8_t * bsp = mlane ? bsT + (size_t)(4 * sp) * 512 + 32 * l16 + 8 * kq : bsT + P.bs;
            const int bstep = mlane ? 512 : 0;
            struct k2_ops { sk_v2i xa[8], raw[2], bh, bl, hm; uint4 sc; uint32_t dm; float4 dy; };
            auto load_ops = [&](int i) __attribute__((always_inline)) {
                k2_ops o;
#pragma unroll
                for (int g = 0; g < 8; ++g) o.xa[g] = *(const sk_v2i *)(tqp + 256 * i + 32 * g);
                o.raw[0] = *(const sk_v2i *)(wr + 64 * i + 8 * kq);
                o.raw[1] = *(const sk_v2i *)(wr + 64 * i + 32 + 8 * kq);
                if constexpr (Q3) {
                    const uint32_t s0 = *(const uint32_t *)(wr + F::SOFF + 12 * i), s1 = *(const uint32_t *)(wr + F::SOFF + 4 + 12 * i), s2 = *(const uint32_t *)(wr + F::SOFF + 8 + 12 * i);
                    // the sixteen 6-bit scales (still + 32) as bytes of four words (k_quants.c:491-496)
                    o.sc = make_uint4((s0 & 0x0F0F0F0Fu) | ((s2 & 0x03030303u) << 4), (s1 & 0x0F0F0F0Fu) | (((s2 >> 2) & 0x03030303u) << 4),
                                      ((s0 >> 4) & 0x0F0F0F0Fu) | (((s2 >> 4) & 0x03030303u) << 4), ((s1 >> 4) & 0x0F0F0F0Fu) | (((s2 >> 6) & 0x03030303u) << 4));
                    o.dm = *(const uint16_t *)(wr + F::DOFF + p3d + 2 * i);
                    o.hm = *(const sk_v2i *)(wr + F::HMOFF + 32 * i + 8 * kq);
                    o.bh = sk_v2i{ 0, 0 }; o.bl = sk_v2i{ 0, 0 };
                } else {
                    o.sc = *(const uint4 *)(wr + F::SOFF + 16 * i);
                    o.dm = *(const uint32_t *)(wr + F::DOFF + 4 * i);
                    o.bh = *(const sk_v2i *)(bsp + bstep * i);
                    o.bl = *(const sk_v2i *)(bsp + bstep * i + 16);
                    o.hm = sk_v2i{ 0, 0 };
                }
                o.dy = *(const float4 *)(dyp + 64 * i);
                return o;
            };
            struct k2_c { sk_v4i c[6]; };
            auto run_mfma = [&](const k2_ops & o) __attribute__((always_inline)) {
                k2_c r;
                const sk_v4i z = { 0, 0, 0, 0 };
                // the lane's sub-block scale of group (h, t): byte 8 h + 2 t + s01 of the 16 -> after the shift by s01 bytes: dword 2 h + (t >> 1), bits 16 (t & 1)
                const uint32_t S[4] = { o.sc.x >> (8 * s01), o.sc.y >> (8 * s01), o.sc.z >> (8 * s01), o.sc.w >> (8 * s01) };
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    sk_v4i c = z, c2 = z;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const uint32_t qx = ((uint32_t) o.raw[h].x >> (2 * t)) & 0x03030303u, qy = ((uint32_t) o.raw[h].y >> (2 * t)) & 0x03030303u;
                        uint32_t lut;
                        if constexpr (Q3) {
                            const int sv = (int)((S[2 * h + (t >> 1)] >> (16 * (t & 1))) & 63u) - 32;            // the lane's sub-block scale
                            lut = (((uint32_t) sv & 0xFFu) << 8) | (((uint32_t)(2 * sv) & 0xFFu) << 16) | (((uint32_t)(3 * sv) & 0xFFu) << 24);   // bytes {0, s, 2 s, 3 s} (int8)
                            // elements whose high bit is clear: - 4 s each = minus the product with the byte 4 s in [-128, 124]
                            const uint32_t nx = (~(uint32_t) o.hm.x >> (4 * h + t)) & 0x01010101u, ny = (~(uint32_t) o.hm.y >> (4 * h + t)) & 0x01010101u;
                            const uint32_t s4 = __builtin_amdgcn_perm(0u, (uint32_t)(4 * sv) & 0xFFu, 0u);                  // the byte 4 s in every byte
                            const sk_v2i b2 = { (int)(((nx << 8) - nx) & s4), (int)(((ny << 8) - ny) & s4) };
                            c2 = __builtin_amdgcn_mfma_i32_16x16x32_i8(__builtin_bit_cast(long, o.xa[4 * h + t]), __builtin_bit_cast(long, b2), c2, 0, 0, 0);
                        } else {
                            const uint32_t sc = (S[2 * h + (t >> 1)] >> (16 * (t & 1))) & 15u;
                            lut = __umul24(sc, 0x030201u) << 8;                            // bytes {0, sc, 2 sc, 3 sc}
                        }
                        const sk_v2i b = { (int) __builtin_amdgcn_perm(0u, lut, qx), (int) __builtin_amdgcn_perm(0u, lut, qy) };
                        c = __builtin_amdgcn_mfma_i32_16x16x32_i8(__builtin_bit_cast(long, o.xa[4 * h + t]), __builtin_bit_cast(long, b), c, 0, 0, 0);
                    }
                    if constexpr (Q3) r.c[t] = sk_v4i{ c[0] - c2[0], c[1] - c2[1], c[2] - c2[2], c[3] - c2[3] }; else r.c[t] = c;
                }
                if constexpr (Q3) { r.c[4] = z; r.c[5] = z; return r; }
                // the mins (high nibbles of the 16 scale bytes), sub-blocks 8 kq .. 8 kq + 7 in the lanes kq < 2
                const uint32_t m0 = kq == 0 ? o.sc.x : o.sc.z, m1 = kq == 0 ? o.sc.y : o.sc.w;
                const sk_v2i mn = { mlane ? (int)((m0 >> 4) & 0x0F0F0F0Fu) : 0, mlane ? (int)((m1 >> 4) & 0x0F0F0F0Fu) : 0 };
                r.c[4] = __builtin_amdgcn_mfma_i32_16x16x32_i8(__builtin_bit_cast(long, o.bh), __builtin_bit_cast(long, mn), z, 0, 0, 0);
                r.c[5] = __builtin_amdgcn_mfma_i32_16x16x32_i8(__builtin_bit_cast(long, o.bl), __builtin_bit_cast(long, mn), z, 0, 0, 0);
                return r;
            };
            auto scale = [&](const k2_c & cc, const k2_ops & o) __attribute__((always_inline)) {
                const float d = fq_h2f((uint16_t) o.dm), dmin = Q3 ? 0.0f : fq_h2f((uint16_t)(o.dm >> 16));
                const float dyv[4] = { o.dy.x, o.dy.y, o.dy.z, o.dy.w };
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float dd = Q3 ? d * dyv[r] : dyv[r] * d, dmn = dyv[r] * dmin;    // k_quants.c:1282-1283: dall = y.d * d, dmin = y.d * dmin (Q3_K: d = y.d * fp16(x.d), k_quants.c:1737)
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        float a = dd * (float) cc.c[t][r];
                        if (!Q3 && t == 3) a = a - dmn * (float)((cc.c[4][r] << 6) + cc.c[5][r]);
                        acc[t][r] = acc[t][r] + a;
                    }
                }
            };
            if (ns == 4) {
                k2_ops o[4]; k2_c c4[4];
                o[0] = load_ops(0); o[1] = load_ops(1);
#pragma unroll
                for (int k = 0; k < 4; ++k) {                               // reads two super-blocks ahead, the matrix instructions one ahead of their scaling
                    if (k + 2 < 4) o[k + 2] = load_ops(k + 2);
                    c4[k] = run_mfma(o[k]);
                    __builtin_amdgcn_sched_barrier(0);
                    if (k > 0) scale(c4[k - 1], o[k - 1]);
                    __builtin_amdgcn_sched_barrier(0);
                }
                scale(c4[3], o[3]);
            } else {
                for (int i = 0; i < ns; ++i) { const k2_ops o = load_ops(i); const k2_c c1 = run_mfma(o); scale(c1, o); }
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = 4 * kq + r;
            if (n < N) {
#pragma unroll
                for (int t = 0; t < 4; ++t) pbase[(size_t) t * sstride + (size_t) n * mstride + m] = acc[t][r];
            }
        }
    }
}

// =============================================================================================== Q6_K: all four K shares per wave, two half operands per group
// ggml_vec_dot_q6_K_q8_K (k_quants.c:2406-2789) in the frame of k_gemm_skinny_q2k (a wave = a 16-row tile with four accumulator chains, the full 16
// columns of a 16-super-block segment resident, the same segmented four-sum order). Q6_K's 6-bit quants - 32 are int8 operands as they are, but its
// sub-block scales are int8 too and cannot go into the operand: a group is two matrix instructions -- the lanes kq < 2 carry the group's first 16
// elements, kq >= 2 the second, each launch with the other half zeroed -- and the scales are applied by v_mad_i32_i24 per result. 210 bytes per
// super-block: a stage is TWO super-blocks (16 pieces of low nibbles | 8 of high bits | 2 of scales | 3 of d, read from the boundary below = 29 slots).
constexpr int K6_ROWP = 29, K6_ROWB = 16 * K6_ROWP, K6_WSTAGE = 16 * K6_ROWB, K6_KOPS = (16 * K6_ROWP + 63) / 64;
static __host__ __device__ inline k2_plan k6_lds(int seg_sb, int T, int nbw) {
    k2_plan p;
    const int qb = seg_sb * 256;
    p.tqs = qb + ((16 - (qb & 255)) & 255);
    p.rings = (size_t) nbw * T * K6_WSTAGE; p.cols = (size_t) SK_TN * p.tqs; p.dy = (size_t) seg_sb * SK_TN * 4; p.bs = 0;
    p.total = p.rings + p.cols + p.dy;
    return p;
}

template <int NBW>
__global__ void __launch_bounds__(64 * KS_TMAX) k_gemm_skinny_q6k(fq_weight w, fq_act act, int N, float * part, int64_t mstride, int T, int nrb, int nslots, int seg_sb, int dbg) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t K = w.K, M = w.M;
    const int nsb = (int) w.nblk;
    const int slot = (int) blockIdx.x, seg = (int) blockIdx.y;
    if (slot >= nrb) return;
    const int sb0 = seg * seg_sb, nsbs = nsb - sb0 < seg_sb ? nsb - sb0 : seg_sb;
    const size_t img = fq_act_col_bytes(FQ_Q8_K, K);
    const k2_plan P = k6_lds(seg_sb, T, NBW);
    const int TQS = P.tqs;
    uint8_t * cols = smem + P.rings;
    float   * dyT  = (float *)(cols + P.cols);
    const int l16 = lane & 15, kq = lane >> 4;
    int nmine = 0;
    for (int rb = slot; rb < nrb; rb += nslots) if (((int64_t) rb * T + wid) * 16 < M) ++nmine;
    unsigned poff[K6_KOPS], rowb[K6_KOPS]; int pkind[K6_KOPS];
#pragma unroll
    for (int k = 0; k < K6_KOPS; ++k) {
        const int L = 64 * k + lane, row = (L / K6_ROWP) & 15, p = L % K6_ROWP;
        if (p < 16)      { poff[k] = (unsigned)(16 * p); pkind[k] = 0; }
        else if (p < 24) { poff[k] = (unsigned)(16 * (p - 16)); pkind[k] = 3; }
        else if (p < 26) { poff[k] = (unsigned)(16 * (p - 24)); pkind[k] = 1; }
        else             { poff[k] = (unsigned)(16 * (p - 26)); pkind[k] = 2; }
        rowb[k] = (unsigned) row * (unsigned) w.row_stride;
    }
    const unsigned rs16 = (unsigned) w.row_stride - 16u;
    uint8_t * myring = smem + (size_t) wid * NBW * K6_WSTAGE;
    const unsigned ring_lds = __builtin_amdgcn_readfirstlane(sk_lds(myring));
    const int nst = (nsbs + 1) / 2;                                        // stages of two super-blocks
    const int total = nmine * nst;
    auto issue = [&](int u) {
        const int rbi = u / nst, sp = u - rbi * nst;
        const int64_t mt = ((int64_t)(slot + rbi * nslots) * T + wid) * 16;
        const uint8_t * wbase = sk_uniform(w.plane[0] + (size_t) mt * w.row_stride);
        const int gsb = sb0 + 2 * sp, c = gsb >> 3, in = gsb & 7;             // a column of the device layout = 8 super-blocks
        const int nbc = nsb - 8 * c < 8 ? nsb - 8 * c : 8;
        const unsigned b0 = (unsigned)(c * 1680 + in * 128);
        const unsigned bh = (unsigned)(c * 1680 + nbc * 128 + in * 64);
        const unsigned b1 = (unsigned)(c * 1680 + nbc * 192 + in * 16);
        const unsigned b2 = (unsigned)(c * 1680 + ((nbc * 208 + in * 2) & ~15));
        const unsigned dst = ring_lds + (unsigned)((u % NBW) * K6_WSTAGE);
#pragma unroll
        for (int k = 0; k < K6_KOPS; ++k) {
            unsigned o = poff[k] + (pkind[k] == 0 ? b0 : (pkind[k] == 1 ? b1 : (pkind[k] == 2 ? b2 : bh)));
            o = o < rs16 ? o : rs16;
            if (64 * k + lane < 16 * K6_ROWP) sk_dma(wbase, rowb[k] + o, dst + (unsigned)(k * 1024));
        }
    };
    if (!(dbg & 8)) { for (int u = 0; u < NBW - 1 && u < total; ++u) issue(u); }
    {
        const unsigned last = (unsigned)(K - 16);
        const int NW = (int)(blockDim.x >> 6);
        for (int t = wid; t < SK_TN; t += NW) {
            const uint8_t * base = sk_uniform(act.base + (size_t)(t < N ? t : N - 1) * img);
            const unsigned tb = sk_lds(cols) + (unsigned)(t * TQS);
            for (int j0 = 0; j0 < nsbs; j0 += 4) {
                unsigned vq = (unsigned)((sb0 + j0) * 256 + 16 * lane);
                vq = vq < last ? vq : last;
                if (j0 + (lane >> 4) < nsbs && !(dbg & 4)) sk_dma(base, vq, tb + (unsigned)(j0 * 256));
            }
        }
        for (int e = tid; e < nsbs * SK_TN; e += (int) blockDim.x) {
            const int sbl = e >> 4, tok = e & 15;
            const uint8_t * tp = act.base + (size_t)(tok < N ? tok : N - 1) * img;
            dyT[e] = ((const float *)(tp + K))[sb0 + sbl];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();                                                       // the only barrier: the columns are in LDS

    float * pbase = part + ((size_t)(4 * seg) * SK_TN) * (size_t) mstride;
    const size_t sstride = (size_t) SK_TN * (size_t) mstride;
    const uint32_t ma = kq < 2 ? 0xFFFFFFFFu : 0u;                         // the lane's elements are the group's first / second 16
    int u = 0;
    for (int rbi = 0; rbi < nmine; ++rbi) {
        const int64_t m = ((int64_t)(slot + rbi * nslots) * T + wid) * 16 + l16;
        float acc[4][4];
#pragma unroll
        for (int t = 0; t < 4; ++t) { acc[t][0] = 0.0f; acc[t][1] = 0.0f; acc[t][2] = 0.0f; acc[t][3] = 0.0f; }
        for (int sp = 0; sp < nst; ++sp, ++u) {
            if (u + NBW - 1 < total && !(dbg & 8)) issue(u + NBW - 1);
            {
                const int later = (dbg & 8) ? 0 : (total - 1 - u < NBW - 1 ? total - 1 - u : NBW - 1);
                if (later >= 1) sk_wait_vm_upto(K6_KOPS); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (NBW = 2: one stage ahead)
            }
            if (dbg & 16) continue;
            const uint8_t * wr = myring + (size_t)(u % NBW) * K6_WSTAGE + l16 * K6_ROWB;
            const int ns = nsbs - 2 * sp < 2 ? nsbs - 2 * sp : 2;
            const int gsb_ = sb0 + 2 * sp, cc_ = gsb_ >> 3, nbc_ = nsb - 8 * cc_ < 8 ? nsb - 8 * cc_ : 8;
            const int p3d = (nbc_ * 208 + (gsb_ & 7) * 2) & 15;
            const uint8_t * tqp = cols + (size_t) l16 * TQS + (size_t)(2 * sp) * 256 + 8 * kq;
            const uint8_t * dyp = (const uint8_t *) dyT + (size_t)(2 * sp) * 64 + 16 * kq;
            struct k6_ops { sk_v2i xa[8], ql[4], qh[2]; uint4 sc; uint32_t dm; float4 dy; };
            auto load_ops = [&](int i) __attribute__((always_inline)) {
                k6_ops o;
#pragma unroll
                for (int g = 0; g < 8; ++g) o.xa[g] = *(const sk_v2i *)(tqp + 256 * i + 32 * g);
#pragma unroll
                for (int j = 0; j < 4; ++j) o.ql[j] = *(const sk_v2i *)(wr + 128 * i + 32 * j + 8 * kq);      // j = 2 h + (t & 1): bytes 64 h + 32 (t & 1) + l
                o.qh[0] = *(const sk_v2i *)(wr + 256 + 64 * i + 8 * kq);
                o.qh[1] = *(const sk_v2i *)(wr + 256 + 64 * i + 32 + 8 * kq);
                o.sc = *(const uint4 *)(wr + 384 + 16 * i);
                o.dm = *(const uint16_t *)(wr + 416 + p3d + 2 * i);
                o.dy = *(const float4 *)(dyp + 64 * i);
                return o;
            };
            struct k6_c { sk_v4i c[4]; };
            auto run_mfma = [&](const k6_ops & o) __attribute__((always_inline)) {
                k6_c r;
                const sk_v4i z = { 0, 0, 0, 0 };
                const uint32_t S[4] = { o.sc.x, o.sc.y, o.sc.z, o.sc.w };
                sk_v4i ia[8], ib[8];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const sk_v2i lo = o.ql[2 * h + (t & 1)], hi = o.qh[h];
                        // (low nibble | 2 high bits << 4) - 32 per byte (k_quants.c:2759-2766): int8 as it is
                        uint32_t vx = (((uint32_t) lo.x >> (4 * (t >> 1))) & 0x0F0F0F0Fu) | ((((uint32_t) hi.x >> (2 * t)) & 0x03030303u) << 4);
                        uint32_t vy = (((uint32_t) lo.y >> (4 * (t >> 1))) & 0x0F0F0F0Fu) | ((((uint32_t) hi.y >> (2 * t)) & 0x03030303u) << 4);
                        vx = ((vx | 0x80808080u) - 0x20202020u) ^ 0x80808080u; vy = ((vy | 0x80808080u) - 0x20202020u) ^ 0x80808080u;
                        const sk_v2i ba = { (int)(vx & ma), (int)(vy & ma) }, bb = { (int)(vx & ~ma), (int)(vy & ~ma) };
                        ia[4 * h + t] = __builtin_amdgcn_mfma_i32_16x16x32_i8(__builtin_bit_cast(long, o.xa[4 * h + t]), __builtin_bit_cast(long, ba), z, 0, 0, 0);
                        ib[4 * h + t] = __builtin_amdgcn_mfma_i32_16x16x32_i8(__builtin_bit_cast(long, o.xa[4 * h + t]), __builtin_bit_cast(long, bb), z, 0, 0, 0);
                    }
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    sk_v4i c = z;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const uint32_t sw = S[2 * h + (t >> 1)] >> (16 * (t & 1));            // int8 scales of sub-blocks 8 h + 2 t, + 1
                        const int sa = (int)(int8_t)(sw & 0xFFu), sb = (int)(int8_t)((sw >> 8) & 0xFFu);
#pragma unroll
                        for (int q = 0; q < 4; ++q) c[q] = __mul24(ib[4 * h + t][q], sb) + (__mul24(ia[4 * h + t][q], sa) + c[q]);
                    }
                    r.c[t] = c;
                }
                return r;
            };
            auto scale = [&](const k6_c & cc, const k6_ops & o) __attribute__((always_inline)) {
                const float d = fq_h2f((uint16_t) o.dm);
                const float dyv[4] = { o.dy.x, o.dy.y, o.dy.z, o.dy.w };
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float dd = d * dyv[r];                                    // k_quants.c:2779 (d = x.d * y.d)
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[t][r] = acc[t][r] + dd * (float) cc.c[t][r];
                }
            };
            if (ns == 2) {
                const k6_ops o0 = load_ops(0), o1 = load_ops(1);
                const k6_c c0 = run_mfma(o0);
                __builtin_amdgcn_sched_barrier(0);
                const k6_c c1 = run_mfma(o1);
                __builtin_amdgcn_sched_barrier(0);
                scale(c0, o0);
                scale(c1, o1);
            } else {
                for (int i = 0; i < ns; ++i) { const k6_ops o = load_ops(i); const k6_c c1 = run_mfma(o); scale(c1, o); }
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = 4 * kq + r;
            if (n < N) {
#pragma unroll
                for (int t = 0; t < 4; ++t) pbase[(size_t) t * sstride + (size_t) n * mstride + m] = acc[t][r];
            }
        }
    }
}

// part: [segment][share][16][mstride]; dst = the segments' ((P0 + P1) + P2) + P3 added left to right, then the epilogue
__global__ void k_skinny_sum4(const float * __restrict__ part, int N, int64_t M, float * __restrict__ dst, int64_t ldd, fq_gemv_epi ep, int64_t mstride, int nseg) {
    const int64_t m = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const int n = blockIdx.y;
    if (m >= M || n >= N) return;
    float v = 0.0f;
    for (int sg = 0; sg < nseg; ++sg) {
        const float * p = part + ((size_t)(4 * sg) * SK_TN + n) * (size_t) mstride + m;
        const size_t st = (size_t) SK_TN * (size_t) mstride;
        const float one = ((p[0] + p[st]) + p[2 * st]) + p[3 * st];
        v = sg ? v + one : one;
    }
    if (ep.mode == FQ_EPI_GELU)      v = h2f_bits(ep.gelu_table[f2h_bits(v)]);
    else if (ep.mode == FQ_EPI_ADD2) v = (v + ep.add1[n * ep.ld_add + m]) + ep.add2[n * ep.ld_add + m];
    dst[n * ldd + m] = v;
}

// a segment list's value at (column n, row m): the segments' ((P0 + P1) + P2) + P3 left to right
__device__ __forceinline__ float sk_sum_segments(const float * __restrict__ part, int64_t mstride, int nseg, int n, int64_t m) {
    const size_t st = (size_t) SK_TN * (size_t) mstride;
    float v = 0.0f;
    for (int sg = 0; sg < nseg; ++sg) {
        const float * p = part + ((size_t)(4 * sg) * SK_TN + n) * (size_t) mstride + m;
        const float one = ((p[0] + p[st]) + p[2 * st]) + p[3 * st];
        v = sg ? v + one : one;
    }
    return v;
}
// x = (down + wo) + x: FQ_EPI_ADD2 with Wo's result summed here instead of read from a matrix
__global__ void k_skinny_sum4_out2(const float * __restrict__ part_d, int nseg_d, const float * __restrict__ part_w, int nseg_w, int64_t mstride, int N, int64_t M,
                                   float * __restrict__ x, int64_t ldx) {
    const int64_t m = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const int n = blockIdx.y;
    if (m >= M || n >= N) return;
    const float v = sk_sum_segments(part_d, mstride, nseg_d, n, m), wv = sk_sum_segments(part_w, mstride, nseg_w, n, m);
    x[n * ldx + m] = (v + wv) + x[n * ldx + m];
}
// GELU + the next mat-mul's Q8_K image: one wave per (column, 256 rows), a lane = 4 consecutive rows (quant_q8K_wave: k_quantize_q8K's arithmetic)
__global__ void __launch_bounds__(256) k_skinny_sum4_gelu_q8k(const float * __restrict__ part, int N, int64_t M, float * __restrict__ dst, int64_t ldd,
                                                              const uint16_t * __restrict__ gelu_table, fq_act out, int64_t mstride, int nseg) {
    const int lane = threadIdx.x & 63;
    const int64_t wv = (int64_t) blockIdx.x * 4 + (threadIdx.x >> 6), per_col = M >> 8;
    if (wv >= (int64_t) N * per_col) return;                               // (wave-uniform)
    const int n = (int)(wv / per_col);
    const int64_t sb = wv - (int64_t) n * per_col, m = 256 * sb + 4 * lane;
    const size_t st = (size_t) SK_TN * (size_t) mstride;
    float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    for (int sg = 0; sg < nseg; ++sg) {
        const float * p = part + ((size_t)(4 * sg) * SK_TN + n) * (size_t) mstride + m;
        const float4 p0 = *(const float4 *) p, p1 = *(const float4 *)(p + st), p2 = *(const float4 *)(p + 2 * st), p3 = *(const float4 *)(p + 3 * st);
        const float4 one = make_float4(((p0.x + p1.x) + p2.x) + p3.x, ((p0.y + p1.y) + p2.y) + p3.y, ((p0.z + p1.z) + p2.z) + p3.z, ((p0.w + p1.w) + p2.w) + p3.w);
        v = sg ? make_float4(v.x + one.x, v.y + one.y, v.z + one.z, v.w + one.w) : one;
    }
    v.x = h2f_bits(gelu_table[f2h_bits(v.x)]); v.y = h2f_bits(gelu_table[f2h_bits(v.y)]); v.z = h2f_bits(gelu_table[f2h_bits(v.z)]); v.w = h2f_bits(gelu_table[f2h_bits(v.w)]);
    if (dst) *(float4 *)(dst + n * ldd + m) = v;
    quant_q8K_wave(v, lane, sb, act_image_at(out.base + (size_t) n * fq_act_col_bytes(FQ_Q8_K, out.K), FQ_Q8_K, out.K));
}

// Q4_K / Q5_K / Q2_K, 5..16 columns (fq_skinny_q4k_shape: the shapes it takes; the oracle's mode 2 follows the same rule)
bool fq_skinny_q4k_shape(const fq_weight & w) {
    static const bool on = !(getenv("FQ_SKINNY_Q4K") && atoi(getenv("FQ_SKINNY_Q4K")) == 0);
    const int seg = (w.type == FQ_Q2_K || w.type == FQ_Q3_K || w.type == FQ_Q6_K) ? K2_SEG : KQ_SEG;
    // the decision from the rows of the whole matrix (a row-split part follows the matrix it is a part of; ggml_hip_weight_upload_rows pads a part to whole
    // 16-row tiles so that it can), the part's own rows only have to fit the form
    const int64_t Mf = fq_form_rows(w);
    const int64_t nseg = (w.nblk + seg - 1) / seg, mstride = (Mf + 63) & ~(int64_t) 63;
    return on && (w.type == FQ_Q4_K || w.type == FQ_Q5_K || w.type == FQ_Q2_K || w.type == FQ_Q3_K || w.type == FQ_Q6_K) && Mf % 16 == 0 && w.M % 16 == 0 && w.M <= Mf &&
           nseg * 4 * SK_TN * mstride <= (int64_t) FQ_KS_FLOATS && w.nblk >= 8 && w.K < ((int64_t) 1 << 24) && w.row_stride * 16 < ((size_t) 1 << 31);
}
static bool q6k_main(const fq_weight & w, const fq_act & act, int64_t N, float * part, int64_t & mstride, int & nseg, hipStream_t st) {
    if (!fq_skinny_q4k_shape(w) || w.type != FQ_Q6_K || act.type != FQ_Q8_K || N < 1 || N > SK_TN) return false;
    const int n_cu = fq_ctx().n_cu;
    const int ntiles = (int)(w.M / 16);
    const int nsb = (int) w.nblk;
    static const int env_t = getenv("FQ_KQ_T") ? atoi(getenv("FQ_KQ_T")) : 0;
    nseg = (nsb + K2_SEG - 1) / K2_SEG;
    const int seg_sb = nseg > 1 ? K2_SEG : ((nsb + 3) & ~3);
    int T = env_t > 0 ? env_t : (ntiles * nseg + n_cu - 1) / n_cu;
    if (T < 1) T = 1;
    if (T > KS_TMAX) T = KS_TMAX;
    while (T > 1 && k6_lds(seg_sb, T, 2).total > 160 * 1024) --T;
    if (k6_lds(seg_sb, T, 2).total > 160 * 1024) return false;
    const size_t need = k6_lds(seg_sb, T, 2).total;
    const int nrb = (ntiles + T - 1) / T;
    int nslots = nrb;
    if (nrb * nseg > n_cu) { const int cap = n_cu / nseg > 0 ? n_cu / nseg : 1; if (nrb % cap == 0 || nrb > 2 * cap) nslots = cap; }
    mstride = (w.M + 63) & ~(int64_t) 63;
    static bool set = false;
    if (!set) { HIP_CHECK(hipFuncSetAttribute((const void *) k_gemm_skinny_q6k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); set = true; }
    hipLaunchKernelGGL((k_gemm_skinny_q6k<2>), dim3((unsigned) nslots, (unsigned) nseg), dim3(64 * T), need, st, w, act, (int) N, part, mstride, T, nrb, nslots, seg_sb, fq_gemm_debug_get());
    return true;
}
static bool q2k_main(const fq_weight & w, const fq_act & act, int64_t N, float * part, int64_t & mstride, int & nseg, hipStream_t st) {
    if (!fq_skinny_q4k_shape(w) || (w.type != FQ_Q2_K && w.type != FQ_Q3_K) || act.type != FQ_Q8_K || N < 1 || N > SK_TN) return false;
    const bool q3 = w.type == FQ_Q3_K;
    auto lds_of = [&](int sg, int t, int n) { return q3 ? k2_lds<FQ_Q3_K>(sg, t, n).total : k2_lds<FQ_Q2_K>(sg, t, n).total; };
    const int n_cu = fq_ctx().n_cu;
    const int ntiles = (int)(w.M / 16);
    const int nsb = (int) w.nblk;
    static const int env_t = getenv("FQ_KQ_T") ? atoi(getenv("FQ_KQ_T")) : 0, env_nbw = getenv("FQ_KQ_NBW") ? atoi(getenv("FQ_KQ_NBW")) : 0;
    nseg = (nsb + K2_SEG - 1) / K2_SEG;
    const int seg_sb = nseg > 1 ? K2_SEG : ((nsb + 3) & ~3);
    int T = env_t > 0 ? env_t : (ntiles * nseg + n_cu - 1) / n_cu;         // tiles (waves) per workgroup: one round when they fit
    if (T < 1) T = 1;
    if (T > KS_TMAX) T = KS_TMAX;
    int nbw = 0;
    for (;;) {
        for (int n : { 3, 2 }) { if (env_nbw && n != env_nbw) continue; if (q3 && n == 3) continue; if (lds_of(seg_sb, T, n) <= 160 * 1024) { nbw = n; break; } }   // (Q3_K: 8 DMA instructions per stage)
        if (nbw || T == 1) break;
        --T;
    }
    if (!nbw) return false;
    const size_t need = lds_of(seg_sb, T, nbw);
    const int nrb = (ntiles + T - 1) / T;
    // more (row block, segment) pairs than CUs: persistent workgroups in full rounds, unless the last round would be nearly empty
    int nslots = nrb;
    if (nrb * nseg > n_cu) { const int cap = n_cu / nseg > 0 ? n_cu / nseg : 1; if (nrb % cap == 0 || nrb > 2 * cap) nslots = cap; }
    mstride = (w.M + 63) & ~(int64_t) 63;
#define FQ_K2_LAUNCH(TT, NB) { \
        static bool set = false; \
        if (!set) { HIP_CHECK(hipFuncSetAttribute((const void *) k_gemm_skinny_q2k<TT, NB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); set = true; } \
        hipLaunchKernelGGL((k_gemm_skinny_q2k<TT, NB>), dim3((unsigned) nslots, (unsigned) nseg), dim3(64 * T), need, st, w, act, (int) N, part, mstride, T, nrb, nslots, seg_sb, fq_gemm_debug_get()); }
    if (q3) FQ_K2_LAUNCH(FQ_Q3_K, 2) else if (nbw == 3) FQ_K2_LAUNCH(FQ_Q2_K, 3) else FQ_K2_LAUNCH(FQ_Q2_K, 2)
#undef FQ_K2_LAUNCH
    return true;
}
// the main launch: partial sums of w x act into scratch region `part` ([segment][share][16][mstride]); false: not this form's shape
static bool q4k_main(const fq_weight & w, const fq_act & act, int64_t N, float * part, int64_t & mstride, int & nseg, hipStream_t st) {
    if (w.type == FQ_Q2_K || w.type == FQ_Q3_K) return q2k_main(w, act, N, part, mstride, nseg, st);
    if (w.type == FQ_Q6_K) return q6k_main(w, act, N, part, mstride, nseg, st);
    if (!fq_skinny_q4k_shape(w) || act.type != FQ_Q8_K || N < 1 || N > SK_TN) return false;
    const int n_cu = fq_ctx().n_cu;
    const int ntiles = (int)(w.M / 16);
    const int nsb = (int) w.nblk;
    static const int env_t = getenv("FQ_KQ_T") ? atoi(getenv("FQ_KQ_T")) : 0, env_nbw = getenv("FQ_KQ_NBW") ? atoi(getenv("FQ_KQ_NBW")) : 0;
    nseg = (nsb + KQ_SEG - 1) / KQ_SEG;
    const int seg_sb = nseg > 1 ? KQ_SEG : ((nsb + 7) & ~7);               // (LDS is sized by it; stages of 4 super-blocks never straddle a column of the device layout)
    int T = env_t > 0 ? env_t : (ntiles * 2 * nseg + n_cu - 1) / n_cu;     // tiles per workgroup: all workgroups resident in one round when they fit
    if (T < 1) T = 1;
    if (T > KS_TMAX) T = KS_TMAX;
    const bool q5 = w.type == FQ_Q5_K;
    auto lds_of = [&](int t, int n) { return q5 ? kq_lds<FQ_Q5_K>(seg_sb, t, n).total : kq_lds<FQ_Q4_K>(seg_sb, t, n).total; };
    int nbw = 0;
    for (;;) {
        for (int n : { 3, 2 }) { if (env_nbw && n != env_nbw) continue; if (q5 && n == 3) continue; if (lds_of(T, n) <= 160 * 1024) { nbw = n; break; } }   // (Q5_K: 8 DMA instructions per stage: two stages ahead would pass vmcnt's 15)
        if (nbw || T == 1) break;
        --T;
    }
    if (!nbw) return false;
    const size_t need = lds_of(T, nbw);
    const int nrb = (ntiles + T - 1) / T;
    int nslots = 8 * ((nrb + 7) / 8);
    const int cap = ((n_cu / (2 * nseg)) / 8) * 8;                         // row-block slots per segment when the launch is larger than the chip
    if (2 * nslots * nseg > n_cu) nslots = cap < 8 ? 8 : (nslots < cap ? nslots : cap);
    mstride = (w.M + 63) & ~(int64_t) 63;
#define FQ_KQ_LAUNCH(TT, NB) { \
        static bool set = false; \
        if (!set) { HIP_CHECK(hipFuncSetAttribute((const void *) k_gemm_skinny_q4k<TT, NB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); set = true; } \
        hipLaunchKernelGGL((k_gemm_skinny_q4k<TT, NB>), dim3((unsigned)(2 * nslots), (unsigned) nseg), dim3(64 * T), need, st, w, act, (int) N, part, mstride, T, nrb, nslots, seg_sb, fq_gemm_debug_get()); }
    if (q5) FQ_KQ_LAUNCH(FQ_Q5_K, 2) else if (nbw == 3) FQ_KQ_LAUNCH(FQ_Q4_K, 3) else FQ_KQ_LAUNCH(FQ_Q4_K, 2)
#undef FQ_KQ_LAUNCH
    return true;
}
void fq_launch_skinny_sum4(const float * part, int64_t N, int64_t M, float * dst, int64_t ldd, const fq_gemv_epi & ep, int64_t mstride, int nseg, hipStream_t st) {
    FQ_TL(st, "skinny_sum4");
    hipLaunchKernelGGL(k_skinny_sum4, dim3((unsigned)((M + 255) / 256), (unsigned) N), dim3(256), 0, st, part, (int) N, M, dst, ldd, ep, mstride, nseg);
}
bool fq_launch_gemm_skinny_kq(const fq_weight & w, const fq_act & act, int64_t N, float * dst, int64_t ldd, const fq_gemv_epi & ep, int S, hipStream_t st) {
    FQ_TL(st, "gemm_skinny_kq");
    int64_t mstride; int nseg;
    if (S == 1 || !q4k_main(w, act, N, fq_ctx().ks_scratch, mstride, nseg, st)) return false;
    hipLaunchKernelGGL(k_skinny_sum4, dim3((unsigned)((w.M + 255) / 256), (unsigned) N), dim3(256), 0, st, fq_ctx().ks_scratch, (int) N, w.M, dst, ldd, ep, mstride, nseg);
    return true;
}
// Wup of a block whose Wdown takes Q8_K columns: the sum launch applies GELU and writes the Q8_K image of the result (k_quantize_q8K's code) next to
// the f32 matrix. false: nothing launched (not this form's shape)
bool fq_launch_gemm_skinny_q4k_gelu_q8k(const fq_weight & w, const fq_act & act, int64_t N, float * dst, int64_t ldd, const uint16_t * gelu_table, const fq_act & out, hipStream_t st) {
    FQ_TL(st, "gemm_skinny_q4k_gelu_q8k");
    if (out.type != FQ_Q8_K || out.K != w.M || w.M % 256 || out.ncols < N) return false;
    int64_t mstride; int nseg;
    if (!q4k_main(w, act, N, fq_ctx().ks_scratch, mstride, nseg, st)) return false;
    const int64_t waves = N * (w.M / 256);
    hipLaunchKernelGGL(k_skinny_sum4_gelu_q8k, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, fq_ctx().ks_scratch, (int) N, w.M, dst, ldd, gelu_table, out, mstride, nseg);
    return true;
}
void fq_launch_skinny_sum4_out2(const float * part_d, int nseg_d, const float * part_w, int nseg_w, int64_t mstride, int64_t N, int64_t M, float * x, int64_t ldx, hipStream_t st) {
    hipLaunchKernelGGL(k_skinny_sum4_out2, dim3((unsigned)((M + 255) / 256), (unsigned) N), dim3(256), 0, st, part_d, nseg_d, part_w, nseg_w, mstride, (int) N, M, x, ldx);
}
// x = (Wdown a_ff + Wo a_att) + x for both matrices in this form: two main launches, ONE sum launch (Wo's result never exists as a matrix)
bool fq_launch_gemm_skinny_q4k_out2(const fq_weight & wo, const fq_act & a_att, const fq_weight & down, const fq_act & a_ff, int64_t N, float * x, int64_t ldx, hipStream_t st) {
    FQ_TL(st, "gemm_skinny_q4k_out2");
    if (wo.M != down.M || !fq_skinny_q4k_shape(wo) || !fq_skinny_q4k_shape(down) || a_att.type != FQ_Q8_K || a_ff.type != FQ_Q8_K || N < 1 || N > SK_TN) return false;
    const int64_t ms = (down.M + 63) & ~(int64_t) 63;
    const int sgd = (down.type == FQ_Q4_K || down.type == FQ_Q5_K) ? KQ_SEG : K2_SEG, sgw = (wo.type == FQ_Q4_K || wo.type == FQ_Q5_K) ? KQ_SEG : K2_SEG;
    const int64_t nsd = (down.nblk + sgd - 1) / sgd, nsw = (wo.nblk + sgw - 1) / sgw;
    if ((nsd + nsw) * 4 * SK_TN * ms > (int64_t) FQ_KS_FLOATS) return false;
    float * part_d = fq_ctx().ks_scratch, * part_w = part_d + (size_t) nsd * 4 * SK_TN * ms;
    int64_t m1, m2; int s1, s2;
    if (!q4k_main(wo, a_att, N, part_w, m2, s2, st)) return false;
    if (!q4k_main(down, a_ff, N, part_d, m1, s1, st)) { fprintf(stderr, "ggml-hip: gemm: the output pair's second launch refused\n"); exit(1); }
    hipLaunchKernelGGL(k_skinny_sum4_out2, dim3((unsigned)((down.M + 255) / 256), (unsigned) N), dim3(256), 0, st, part_d, s1, part_w, s2, m1, (int) N, down.M, x, ldx);
    return true;
}

