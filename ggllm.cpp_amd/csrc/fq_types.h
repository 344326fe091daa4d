// fq_types.h -- quant formats, device-side plane layout and activation layout shared by host and device code.
//
// Weight tensors are uploaded once (ggml_cuda_transform_tensor, reference ggml-cuda.cu:3030-3073). The
// backend owns the device copy, so it is re-tiled from ggml's array-of-blocks into PLANES (structure of
// arrays). Total bytes are unchanged (18 B / 32 weights for Q4_0 ...); what changes is that the 16-byte quant
// groups become 16-byte aligned so that one lane = one `global_load_dwordx4`, and a wave reads 1 KiB of
// consecutive HBM per instruction.
//   The planes are INTERLEAVED per "column" inside a row: a column = the blocks whose plane-0 bytes add up to 1 KiB
//   (64 blocks of Q4_0 .. Q5_1, 32 of Q8_0, 16 super-blocks of Q2_K / Q3_K, 8 of Q4_K / Q5_K / Q6_K), stored as
//   [plane 0 of all its blocks | plane 1 of all its blocks | ...], e.g. Q4_0: [64 x 16 B quants | 64 x 2 B scales].
//   The side data a wave needs then follows its 1 KiB quant read in the same DRAM page instead of opening another page
//   in a separate plane (a pure-read kernel of this access shape is 12 % faster with the scales interleaved,
//   scripts/microbench/mb_stream.hip). Rows are padded to a multiple of 16 bytes; a last, partial column packs its
//   planes the same way.
//
//   type   plane0 (quants)   plane1          plane2         plane3      ggml block (bytes @offset)
//   Q4_0   qs 16             d f16 2                                     d@0 qs@2            (ggml.c:879-883)
//   Q4_1   qs 16             d,m 2xf16 4                                 d@0 m@2 qs@4        (ggml.c:886-891)
//   Q5_0   qs 16             qh u32 4        d f16 2                     d@0 qh@2 qs@6       (ggml.c:894-899)
//   Q5_1   qs 16             qh u32 4        d,m 4                       d@0 m@2 qh@4 qs@8   (ggml.c:902-908)
//   Q8_0   qs 32             d f16 2                                     d@0 qs@2            (ggml.c:911-915)
//   Q2_K   qs 64             scales 16       d,dmin 4                    sc@0 qs@16 d@80 dmin@82   (k_quants.h:20-25)
//   Q3_K   qs 64             hmask 32        scales 12      d 2          hm@0 qs@32 sc@96 d@108    (k_quants.h:32-37)
//   Q4_K   qs 128            scales 12       d,dmin 4                    d@0 dmin@2 sc@4 qs@16     (k_quants.h:44-49)
//   Q5_K   qs 128            qh 32           scales 12      d,dmin 4     d@0 dmin@2 sc@4 qh@16 qs@48 (k_quants.h:56-62)
//   Q6_K   ql 128            qh 64           scales 16 i8   d 2          ql@0 qh@128 sc@192 d@208  (k_quants.h:69-74)
#pragma once
// ONE switch for the fused form of the legacy formats' K-split partial sums, acc = fma(d_w * d_x, (float) isum, acc), in the tile GEMM (kernels_gemm.hip) and
// the streaming small-batch forms (kernels_gemm_skinny.hip): both must agree ("same K split -> same bits"), and the oracle's split orders restate the same
// form with fmaf (oracle_quants.c orc_legacy_block_parts). FMA form, oracle-pinned; the single sequential sum (reference order) keeps two roundings per term.
#ifndef FQ_SPLIT_FMA
#define FQ_SPLIT_FMA 1
#endif
#include <stdint.h>
#include <stddef.h>

#if defined(__HIPCC__)
#define FQ_HD  __host__ __device__ __forceinline__
#define FQ_HDM __host__ __device__ __forceinline__      // member functions
#else
#define FQ_HD  static inline
#define FQ_HDM inline
#endif

enum fq_type {            // numbering = enum ggml_type (ggml.h:247-268)
    FQ_F32 = 0, FQ_F16 = 1, FQ_Q4_0 = 2, FQ_Q4_1 = 3, FQ_Q5_0 = 6, FQ_Q5_1 = 7, FQ_Q8_0 = 8, FQ_Q8_1 = 9,
    FQ_Q2_K = 10, FQ_Q3_K = 11, FQ_Q4_K = 12, FQ_Q5_K = 13, FQ_Q6_K = 14, FQ_Q8_K = 15,
};

#define FQ_MAX_PLANES 4

struct fq_plane_desc { int src_off; int bytes; };     // where a plane's bytes sit inside the ggml block

struct fq_type_desc {
    int blck;                         // elements per ggml block (32 / 256)
    int tsize;                        // bytes per ggml block
    int act_type;                     // vec_dot_type (ggml.c:1627-1718)
    int nplanes;
    fq_plane_desc plane[FQ_MAX_PLANES];
    int unit_elems;                   // weights covered by one "unit" = 16-byte quant group of plane0 (32 B for Q8_0)
};

constexpr FQ_HD fq_type_desc fq_desc(int type) {
    switch (type) {
        case FQ_Q4_0: return { 32,  18, FQ_Q8_0, 2, {{2, 16}, {0, 2}, {0, 0}, {0, 0}}, 32 };
        case FQ_Q4_1: return { 32,  20, FQ_Q8_1, 2, {{4, 16}, {0, 4}, {0, 0}, {0, 0}}, 32 };
        case FQ_Q5_0: return { 32,  22, FQ_Q8_0, 3, {{6, 16}, {2, 4}, {0, 2}, {0, 0}}, 32 };
        case FQ_Q5_1: return { 32,  24, FQ_Q8_1, 3, {{8, 16}, {4, 4}, {0, 4}, {0, 0}}, 32 };
        case FQ_Q8_0: return { 32,  34, FQ_Q8_0, 2, {{2, 32}, {0, 2}, {0, 0}, {0, 0}}, 32 };
        case FQ_Q2_K: return { 256, 84, FQ_Q8_K, 3, {{16, 64}, {0, 16}, {80, 4}, {0, 0}}, 64 };
        case FQ_Q3_K: return { 256, 110, FQ_Q8_K, 4, {{32, 64}, {0, 32}, {96, 12}, {108, 2}}, 64 };
        case FQ_Q4_K: return { 256, 144, FQ_Q8_K, 3, {{16, 128}, {4, 12}, {0, 4}, {0, 0}}, 32 };
        case FQ_Q5_K: return { 256, 176, FQ_Q8_K, 4, {{48, 128}, {16, 32}, {4, 12}, {0, 4}}, 32 };
        case FQ_Q6_K: return { 256, 210, FQ_Q8_K, 4, {{0, 128}, {128, 64}, {192, 16}, {208, 2}}, 32 };
        default:      return { 0, 0, 0, 0, {{0, 0}, {0, 0}, {0, 0}, {0, 0}}, 0 };
    }
}

// vec_dot_type of a weight format as a compile-time constant (ggml.c:1627-1718)
constexpr int fq_act_of(int type) {
    return (type == FQ_Q4_1 || type == FQ_Q5_1) ? FQ_Q8_1 : ((type == FQ_Q4_0 || type == FQ_Q5_0 || type == FQ_Q8_0) ? FQ_Q8_0 : FQ_Q8_K);
}

// A weight matrix on the device: K inputs (ne00), M output rows (ne01).
struct fq_weight {
    int     type;
    int64_t K, M;
    int64_t nblk;                     // ggml blocks per row = K / blck
    uint8_t * plane[FQ_MAX_PLANES];   // plane[0] = device pointer of row 0 (the other entries repeat it)
    size_t  row_stride;               // bytes from one row to the next
    size_t  bytes;                    // total device bytes = M * nblk * tsize  (== ggml_nbytes)
    int64_t form_M;                   // a row range of a larger matrix (row-split tensor parallelism): the rows of the WHOLE matrix, 0 otherwise. Every choice that
                                      // fixes the association of a row's sum (the GEMM's K split, the k-quants' small-batch form) is made from fq_form_rows(), so
                                      // that a part sums its rows exactly as the unsplit matrix would
};
static inline int64_t fq_form_rows(const fq_weight & w) { return w.form_M > 0 ? w.form_M : w.M; }

// Quantized activations on the device. Every column (token) is ONE contiguous, 16-byte aligned image
//      [ qs int8 x K | d f32 x nd | aux ]          image stride = fq_act_col_bytes(type, K)
// so that a GEMV workgroup stages a column into LDS with a flat 16-byte copy and uses the same offsets there.
//   Q8_0: d = fp16-rounded delta (as f32), nd = K/32 ; aux = isum i32 x K/32 (sum of the block's qs)
//   Q8_1: d = delta f32, nd = K/32               ; aux = s f32 x K/32 (= d * sum qs)           (ggml.c:918-923)
//   Q8_K: d = delta f32, nd = K/256              ; aux = bsums i16 x K/16                       (k_quants.h:78-82)
struct fq_act {
    int       type;
    int64_t   K;
    int64_t   ncols;
    uint8_t * base;                   // column c starts at base + c * fq_act_col_bytes(type, K)
};

FQ_HD size_t fq_act_aux_elems(int act_type, int64_t K) { return act_type == FQ_Q8_K ? (size_t)(K / 16) : (size_t)(K / 32); }
FQ_HD size_t fq_act_aux_esize(int act_type)            { return act_type == FQ_Q8_K ? 2 : 4; }
FQ_HD size_t fq_act_d_elems(int act_type, int64_t K)   { return act_type == FQ_Q8_K ? (size_t)(K / 256) : (size_t)(K / 32); }
FQ_HD size_t fq_act_d_off(int act_type, int64_t K)     { (void) act_type; return (size_t) K; }
FQ_HD size_t fq_act_aux_off(int act_type, int64_t K)   { return (size_t) K + fq_act_d_elems(act_type, K) * 4; }
FQ_HD size_t fq_act_col_bytes(int act_type, int64_t K) {
    return (fq_act_aux_off(act_type, K) + fq_act_aux_elems(act_type, K) * fq_act_aux_esize(act_type) + 15) & ~(size_t) 15;
}

// ---- interleaved layout (header comment)
constexpr FQ_HD int fq_plane_pre(const fq_type_desc & d, int p) { int pre = 0; for (int q = 0; q < p; ++q) pre += d.plane[q].bytes; return pre; }
FQ_HD size_t fq_il_row_stride(const fq_type_desc & d, int64_t nblk) { return ((size_t) nblk * (size_t) d.tsize + 15) & ~(size_t) 15; }
// byte offset, inside its row, of plane p's chunk of block b. cb = blocks per column (1024 / plane0 bytes), pre = bytes of
// the planes before p in a block, pb = bytes of plane p in a block
FQ_HD size_t fq_il_offset(int cb, int tsize, int pre, int pb, int64_t nblk, int64_t b) {
    const int64_t col = b / cb, j = b - col * cb;
    const int64_t rem = nblk - col * cb;
    const int64_t nbc = rem < cb ? rem : cb;                  // blocks in this column
    return (size_t) col * (size_t)(cb * tsize) + (size_t) nbc * (size_t) pre + (size_t) j * (size_t) pb;
}
FQ_HD size_t fq_il_offset(const fq_type_desc & d, int p, int64_t nblk, int64_t b) {
    return fq_il_offset(1024 / d.plane[0].bytes, d.tsize, fq_plane_pre(d, p), d.plane[p].bytes, nblk, b);
}

