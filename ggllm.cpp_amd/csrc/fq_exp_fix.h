// fq_exp_fix.h -- (input bits << 16 | table entry) of the fp16 inputs whose exp() the f32 fast path of exp_f16_formula (fq_device.h) cannot decide.
// GENERATED on an MI355X by scripts/gpu_exp_boundary.py from ggml_hip_debug_exp_boundary (48 of 63488 inputs); an input missing here keeps the fast path's value -- which is why the formula is used only on request (GGML_HIP_EXP_FORMULA=1; the table gather is the default since round 5) and only after ggml_hip_init has compared formula and table for EVERY input (fq_verify_exp_formula == 0): on any mismatch (another chip's v_exp_f32) the kernels keep the table.
#pragma once
#define FQ_EXP_FIX_N 48
static __device__ const unsigned fq_exp_fix[48] = {
    0x0FFE3C00u, 0x0FFF3C00u, 0x10003C01u, 0x15FF3C02u, 0x1AFD3C04u, 0x1F793C08u, 0x23333C0Fu, 0x25CF3C18u,
    0x264C3C19u, 0x32373CDCu, 0x3F0345C6u, 0x3F85468Eu, 0x45435A07u, 0x4758660Bu, 0x497C7B16u, 0x49807B4Fu,
    0x49817B5Eu, 0x49827B6Du, 0x49837B7Bu, 0x49847B8Au, 0x49857B9Au, 0x49867BA9u, 0x49877BB8u, 0x49887BC8u,
    0x49897BD7u, 0x498A7BE7u, 0x498B7BF7u, 0x8BFD3C00u, 0x8BFE3C00u, 0x8BFF3C00u, 0x8C003C00u, 0x8C013BFFu,
    0x92003BFFu, 0x92013BFEu, 0x95013BFDu, 0x99823BFAu, 0x9E453BF3u, 0xA0E63BECu, 0xA1A83BE9u, 0xA51D3BD7u,
    0xA57F3BD4u, 0xA9223BAFu, 0xA9543BACu, 0xAA0C3BA1u, 0xAC953B73u, 0xC13B2CAFu, 0xC1EF2A97u, 0xC64E177Du,
};
