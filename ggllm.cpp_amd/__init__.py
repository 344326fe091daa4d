"""ggllm.cpp_amd -- thin ctypes view of libggml_hip.so (the MI355X-native backend for ggllm.cpp's hot path).

The product is the C-ABI shared library built from csrc/ (HIP kernels for gfx950 + the C entry points declared in
include/*.h). This module only loads it and wraps pointers for tests / bench.py; there is no Python compute path and
no CPU fallback: importing works anywhere, but every call needs the library AND a HIP device, otherwise it raises.
"""
import ctypes as C
import os
import subprocess

import numpy as np

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG_DIR)
LIB_PATH = os.environ.get("GGLLM_HIP_LIB") or os.path.join(PKG_DIR, "libggml_hip.so")      # (GGLLM_HIP_LIB: A/B builds of the same library)

# enum ggml_type values (ggml.h:247-268)
F32, F16, Q4_0, Q4_1, Q5_0, Q5_1, Q8_0, Q8_1 = 0, 1, 2, 3, 6, 7, 8, 9
Q2_K, Q3_K, Q4_K, Q5_K, Q6_K, Q8_K = 10, 11, 12, 13, 14, 15
BLCK = {Q4_0: 32, Q4_1: 32, Q5_0: 32, Q5_1: 32, Q8_0: 32, Q8_1: 32, Q2_K: 256, Q3_K: 256, Q4_K: 256, Q5_K: 256, Q6_K: 256, Q8_K: 256}
TSIZE = {Q4_0: 18, Q4_1: 20, Q5_0: 22, Q5_1: 24, Q8_0: 34, Q8_1: 40, Q2_K: 84, Q3_K: 110, Q4_K: 144, Q5_K: 176, Q6_K: 210, Q8_K: 292}
LEGACY = (Q4_0, Q4_1, Q5_0, Q5_1, Q8_0)
KQUANTS = (Q2_K, Q3_K, Q4_K, Q5_K, Q6_K)
TYPE_NAME = {Q4_0: "q4_0", Q4_1: "q4_1", Q5_0: "q5_0", Q5_1: "q5_1", Q8_0: "q8_0", Q8_1: "q8_1",
             Q2_K: "q2_K", Q3_K: "q3_K", Q4_K: "q4_K", Q5_K: "q5_K", Q6_K: "q6_K", Q8_K: "q8_K"}
VEC_DOT = {Q4_0: Q8_0, Q5_0: Q8_0, Q8_0: Q8_0, Q4_1: Q8_1, Q5_1: Q8_1, Q2_K: Q8_K, Q3_K: Q8_K, Q4_K: Q8_K, Q5_K: Q8_K, Q6_K: Q8_K}

EXPORTS_OPS = """ggml_hip_init ggml_hip_split_configure ggml_hip_tensor_split_rows ggml_hip_weight_upload_rows ggml_hip_split_comm_create ggml_hip_split_comm_free ggml_hip_split_comm_agree ggml_hip_mul_mat_q_split ggml_hip_mul_mat_q_split_local ggml_hip_split_comm_create_loopback ggml_hip_split_comm_rccl_ranks ggml_hip_mul_mat_q_split_loopback ggml_hip_shim_pool_stats ggml_hip_get_reference_order ggml_hip_debug_force_gemv ggml_hip_debug_attention_form ggml_hip_debug_exp_boundary ggml_hip_gemm_sequential ggml_hip_reference_order ggml_hip_debug_stamps ggml_hip_selftest ggml_hip_exp_formula_mismatches ggml_hip_device_count ggml_hip_stream ggml_hip_malloc ggml_hip_free ggml_hip_memcpy_h2d
ggml_hip_memcpy_d2h ggml_hip_memcpy_d2d ggml_hip_memset ggml_hip_synchronize ggml_hip_event_create ggml_hip_event_record
ggml_hip_event_elapsed_ms ggml_hip_event_destroy ggml_hip_profile_begin ggml_hip_profile_end ggml_hip_profile_bracket_overhead_us ggml_hip_gelu_table_dev ggml_hip_exp_table_dev ggml_hip_weight_upload
ggml_hip_weight_free ggml_hip_weight_nbytes ggml_hip_dequantize_rows ggml_hip_quantize_rows ggml_hip_weight_quantize ggml_hip_fp16_to_fp32_row ggml_hip_acts_alloc ggml_hip_acts_free
ggml_hip_quantize_acts ggml_hip_acts_export ggml_hip_mul_mat_q ggml_hip_mul_mat_q_acts ggml_hip_layer_norm ggml_hip_gelu
ggml_hip_add3 ggml_hip_rope_table_create ggml_hip_rope_kv_store ggml_hip_attention""".split()
EXPORTS_FALCON = """falcon_hip_model_create falcon_hip_model_free falcon_hip_model_set_tensor falcon_hip_model_weight_bytes
falcon_hip_context_create falcon_hip_context_free falcon_hip_eval falcon_hip_eval_stage falcon_hip_stage_step falcon_hip_decode_greedy falcon_hip_eval_token falcon_hip_context_last_error falcon_hip_context_set_rope_n_ctx
falcon_hip_get_logits falcon_hip_context_keep_hidden falcon_hip_get_hidden falcon_hip_context_use_graph
falcon_hip_eval_debug_timings falcon_hip_context_set_fused falcon_hip_context_sync_error falcon_hip_model_load_ggcc falcon_hip_ggcc_scan falcon_hip_plan_stages falcon_hip_model_quantize falcon_hip_perplexity
falcon_hip_vocab_load_ggcc falcon_hip_vocab_error falcon_hip_vocab_free falcon_hip_vocab_size falcon_hip_vocab_merges falcon_hip_tokenize
falcon_hip_token_to_bytes falcon_hip_token_bos falcon_hip_token_eos
falcon_hip_model_get_hparams falcon_hip_context_create_seqs falcon_hip_context_n_seq
falcon_hip_pipeline_unique_id falcon_hip_pipeline_create falcon_hip_pipeline_create_local falcon_hip_pipeline_free falcon_hip_pipeline_rccl_ranks falcon_hip_pipeline_transport falcon_hip_rccl_selftest falcon_hip_pipeline_set_tokens
falcon_hip_pipeline_run falcon_hip_pipeline_run_local falcon_hip_pipeline_local_attach_rccl falcon_hip_pipeline_get_history falcon_hip_pipeline_schedule""".split()


def build(verbose=False):
    """hipcc every HIP source for gfx950 into ggllm.cpp_amd/libggml_hip.so (cross-compiles without a GPU)."""
    cmd = ["make", "-C", os.path.join(PKG_DIR, "csrc"), "-j", str(min(16, os.cpu_count() or 4))]
    if not verbose:
        cmd.insert(1, "-s")
    subprocess.check_call(cmd + ["all"])
    return LIB_PATH


class HParams(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("n_vocab", "n_embd", "n_head", "n_head_kv", "n_layer", "n_ff", "two_norms", "layer_begin", "layer_end")]


_lib = None


def load():
    """dlopen the library and declare the signatures. Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: run __graft_entry__.build() (hipcc) first; there is no fallback path")
    L = C.CDLL(LIB_PATH)
    vp, i64, i32, sz = C.c_void_p, C.c_int64, C.c_int32, C.c_size_t
    sig = {
        "ggml_hip_init": (C.c_int, [C.c_int]), "ggml_hip_selftest": (C.c_int, []), "ggml_hip_exp_formula_mismatches": (C.c_int, []), "ggml_hip_debug_force_gemv": (None, [C.c_int]), "ggml_hip_debug_attention_form": (None, [C.c_int]), "ggml_hip_debug_exp_boundary": (C.c_int, [vp, C.c_int]), "ggml_hip_debug_stamps": (None, [C.c_int, vp]), "ggml_hip_device_count": (C.c_int, []), "ggml_hip_stream": (vp, []),
        "ggml_hip_malloc": (vp, [sz]), "ggml_hip_free": (None, [vp]),
        "ggml_hip_memcpy_h2d": (None, [vp, vp, sz]), "ggml_hip_memcpy_d2h": (None, [vp, vp, sz]), "ggml_hip_memcpy_d2d": (None, [vp, vp, sz]),
        "ggml_hip_memset": (None, [vp, C.c_int, sz]), "ggml_hip_synchronize": (None, []),
        "ggml_hip_event_create": (vp, []), "ggml_hip_event_record": (None, [vp]), "ggml_hip_event_elapsed_ms": (C.c_float, [vp, vp]),
        "ggml_hip_profile_begin": (None, []), "ggml_hip_profile_bracket_overhead_us": (C.c_double, []), "ggml_hip_profile_end": (None, [vp, vp, vp]),
        "ggml_hip_event_destroy": (None, [vp]), "ggml_hip_gelu_table_dev": (vp, []), "ggml_hip_exp_table_dev": (vp, []),
        "ggml_hip_weight_upload": (vp, [C.c_int, vp, i64, i64]), "ggml_hip_weight_free": (None, [vp]), "ggml_hip_weight_nbytes": (sz, [vp]),
        "ggml_hip_dequantize_rows": (None, [vp, vp, i64, vp]), "ggml_hip_gemm_sequential": (None, [C.c_int]), "ggml_hip_reference_order": (None, [C.c_int]), "ggml_hip_get_reference_order": (C.c_int, []), "ggml_hip_shim_pool_stats": (None, [vp, vp, vp]),
        "ggml_hip_tensor_split_rows": (None, [vp, C.c_int, i64, vp, vp]), "ggml_hip_weight_upload_rows": (vp, [C.c_int, vp, i64, i64, i64, i64]),
        "ggml_hip_split_comm_create": (vp, [C.c_int, C.c_int, vp]), "ggml_hip_split_configure": (C.c_int, [C.c_int, C.c_int, vp]), "ggml_hip_split_comm_free": (None, [vp]), "ggml_hip_split_comm_agree": (C.c_int, [vp, vp, C.c_size_t]),
        "ggml_hip_mul_mat_q_split": (C.c_int, [vp, vp, vp, i64, i64, vp, i64, vp, vp]),
        "ggml_hip_mul_mat_q_split_local": (C.c_int, [vp, C.c_int, vp, i64, i64, vp, i64, vp, vp]),
        "ggml_hip_split_comm_create_loopback": (vp, [C.c_int]), "ggml_hip_split_comm_rccl_ranks": (C.c_int, [vp]),
        "ggml_hip_mul_mat_q_split_loopback": (C.c_int, [vp, vp, vp, i64, i64, vp, i64, vp, vp]),
        "ggml_hip_quantize_rows": (C.c_int, [C.c_int, vp, i64, i64, vp, vp]), "ggml_hip_weight_quantize": (vp, [C.c_int, vp, i64, i64]),
        "ggml_hip_fp16_to_fp32_row": (None, [vp, vp, i64]),
        "ggml_hip_acts_alloc": (vp, [C.c_int, i64, i64]), "ggml_hip_acts_free": (None, [vp]),
        "ggml_hip_quantize_acts": (None, [vp, vp, i64, i64]), "ggml_hip_acts_export": (None, [vp, i64, vp]),
        "ggml_hip_mul_mat_q": (None, [vp, vp, i64, i64, vp, i64]),
        "ggml_hip_mul_mat_q_acts": (None, [vp, vp, i64, vp, i64, C.c_int, vp, vp]),
        "ggml_hip_layer_norm": (None, [vp, i64, i64, vp, vp, vp]), "ggml_hip_gelu": (None, [vp, vp, i64]),
        "ggml_hip_add3": (None, [vp, vp, vp, vp, i64]), "ggml_hip_rope_table_create": (vp, [C.c_int, C.c_int, C.c_int]),
        "ggml_hip_rope_kv_store": (None, [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp]),
        "ggml_hip_attention": (None, [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp]),
        "falcon_hip_model_create": (vp, [C.POINTER(HParams)]), "falcon_hip_model_free": (None, [vp]),
        "falcon_hip_model_set_tensor": (C.c_int, [vp, C.c_char_p, C.c_int, vp, i64, i64]),
        "falcon_hip_model_weight_bytes": (sz, [vp]),
        "falcon_hip_model_get_hparams": (None, [vp, vp]),
        "falcon_hip_context_create_seqs": (vp, [vp, C.c_int, C.c_int, C.c_int]), "falcon_hip_context_n_seq": (C.c_int, [vp]),
        "falcon_hip_pipeline_unique_id": (C.c_int, [vp]), "falcon_hip_pipeline_create": (vp, [vp, C.c_int, C.c_int, vp, C.c_int, C.c_int, C.c_int]),
        "falcon_hip_pipeline_create_local": (vp, [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]), "falcon_hip_pipeline_free": (None, [vp]), "falcon_hip_pipeline_rccl_ranks": (C.c_int, [vp]), "falcon_hip_pipeline_transport": (C.c_int, [vp]), "falcon_hip_rccl_selftest": (C.c_int, [C.c_int, C.c_int, C.c_char_p, C.c_int]),
        "falcon_hip_pipeline_set_tokens": (C.c_int, [vp, vp]), "falcon_hip_pipeline_run": (C.c_int, [vp, C.c_int, C.c_int]),
        "falcon_hip_pipeline_run_local": (C.c_int, [vp, C.c_int, C.c_int, C.c_int]), "falcon_hip_pipeline_local_attach_rccl": (C.c_int, [vp, C.c_int]),
        "falcon_hip_pipeline_get_history": (C.c_int, [vp, vp, C.c_int, C.c_int]),
        "falcon_hip_pipeline_schedule": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
        "falcon_hip_context_create": (vp, [vp, C.c_int, C.c_int, C.c_int]), "falcon_hip_context_free": (None, [vp]),
        "falcon_hip_eval": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int]),
        "falcon_hip_eval_stage": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, vp]),
        "falcon_hip_decode_greedy": (C.c_int, [vp, i32, C.c_int, C.c_int, vp]),
        "falcon_hip_eval_token": (C.c_int, [vp, i32, C.c_int]), "falcon_hip_context_last_error": (C.c_int, [vp]), "falcon_hip_context_set_rope_n_ctx": (None, [vp, C.c_int]),
        "falcon_hip_stage_step": (C.c_int, [vp, vp, vp, C.c_int, vp, vp]),
        "falcon_hip_get_logits": (C.POINTER(C.c_float), [vp]),
        "falcon_hip_context_keep_hidden": (None, [vp, C.c_int]), "falcon_hip_get_hidden": (None, [vp, vp]),
        "falcon_hip_vocab_load_ggcc": (vp, [C.c_char_p]), "falcon_hip_vocab_error": (C.c_char_p, [vp]), "falcon_hip_vocab_free": (None, [vp]),
        "falcon_hip_vocab_size": (C.c_int, [vp]), "falcon_hip_vocab_merges": (C.c_int, [vp]),
        "falcon_hip_tokenize": (C.c_int, [vp, C.c_char_p, vp, C.c_int, C.c_int]),
        "falcon_hip_token_to_bytes": (C.c_int, [vp, C.c_int32, C.POINTER(C.c_char_p)]),
        "falcon_hip_token_bos": (C.c_int32, []), "falcon_hip_token_eos": (C.c_int32, []),
        "falcon_hip_context_use_graph": (None, [vp, C.c_int]), "falcon_hip_context_set_fused": (None, [vp, C.c_int]), "falcon_hip_eval_debug_timings": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int]), "falcon_hip_context_sync_error": (C.c_int, [vp]),
        "falcon_hip_model_load_ggcc": (vp, [C.c_char_p, C.c_int, C.c_int, vp]), "falcon_hip_ggcc_scan": (C.c_int, [C.c_char_p, vp, vp, C.c_char_p, C.c_size_t]),
        "falcon_hip_model_quantize": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, vp]),
        "falcon_hip_plan_stages": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_size_t, vp, vp, vp]),
        "falcon_hip_perplexity": (C.c_int, [vp, vp, C.c_int64, C.c_int, C.c_int, vp]),
    }
    for name, (res, args) in sig.items():
        f = getattr(L, name)          # AttributeError here = an include/*.h symbol is not exported
        f.restype, f.argtypes = res, args
    _lib = L
    return L


def init(device=0):
    """Initialise the HIP backend on `device`. Aborts (exit 1) when no GPU is visible -- by design."""
    return load().ggml_hip_init(device)


# ------------------------------------------------------------------------------------------- small host helpers
class DevBuf:
    """device allocation + numpy transfer (tests / bench plumbing)"""

    def __init__(self, nbytes=None, host=None):
        L = load()
        if host is not None:
            host = np.ascontiguousarray(host)
            nbytes = host.nbytes
        self.nbytes = int(nbytes)
        self.ptr = L.ggml_hip_malloc(max(self.nbytes, 16))
        if host is not None and self.nbytes:
            L.ggml_hip_memcpy_h2d(self.ptr, host.ctypes.data, self.nbytes)

    def to_host(self, dtype, shape):
        out = np.empty(shape, dtype)
        assert out.nbytes <= self.nbytes
        load().ggml_hip_memcpy_d2h(out.ctypes.data, self.ptr, out.nbytes)
        return out

    def free(self):
        if self.ptr:
            load().ggml_hip_free(self.ptr)
            self.ptr = None


def quantize_rows(wtype, x, hist=False):
    """x: [nrows, K] f32 -> ggml blocks (uint8 [nrows, K/blck*tsize]) through ggml_hip_quantize_rows; hist=True also
    returns the 16-bin histogram the reference's quantizer prints"""
    L = load()
    x = np.ascontiguousarray(x, np.float32)
    nrows, K = x.shape
    xb = DevBuf(host=x)
    ob = DevBuf(nrows * (K // BLCK[wtype]) * TSIZE[wtype])
    hb = DevBuf(host=np.zeros(16, np.int64)) if hist else None
    rc = L.ggml_hip_quantize_rows(wtype, xb.ptr, K, nrows, ob.ptr, hb.ptr if hb else None)
    if rc != 0:
        raise ValueError("ggml_hip_quantize_rows failed")
    out = ob.to_host(np.uint8, (nrows, (K // BLCK[wtype]) * TSIZE[wtype]))
    h = hb.to_host(np.int64, (16,)) if hb else None
    for b in (xb, ob, hb):
        if b:
            b.free()
    return (out, h) if hist else out


class Weight:
    @classmethod
    def quantize(cls, wtype, x):
        """x: [M, K] f32, quantized and re-tiled on the device (ggml_hip_weight_quantize)"""
        x = np.ascontiguousarray(x, np.float32)
        self = cls.__new__(cls)
        self.type, self.M, self.K = wtype, x.shape[0], x.shape[1]
        xb = DevBuf(host=x)
        self.h = load().ggml_hip_weight_quantize(wtype, xb.ptr, self.K, self.M)
        xb.free()
        if not self.h:
            raise ValueError("ggml_hip_weight_quantize failed")
        return self

    def __init__(self, wtype, blocks, K, M):
        blocks = np.ascontiguousarray(blocks, np.uint8)
        assert blocks.size == M * (K // BLCK[wtype]) * TSIZE[wtype]
        self.type, self.K, self.M = wtype, K, M
        self.h = load().ggml_hip_weight_upload(wtype, blocks.ctypes.data, K, M)

    def dequantize(self, rows=None):
        L = load()
        n = self.M if rows is None else len(rows)
        out = DevBuf(n * self.K * 4)
        rb = DevBuf(host=np.asarray(rows, np.int32)) if rows is not None else None
        L.ggml_hip_dequantize_rows(self.h, rb.ptr if rb else None, n, out.ptr)
        y = out.to_host(np.float32, (n, self.K))
        out.free()
        if rb:
            rb.free()
        return y

    def mul_mat(self, x):
        """x: [N, K] f32 -> [N, M] f32 through ggml_hip_mul_mat_q"""
        L = load()
        x = np.ascontiguousarray(x, np.float32).reshape(-1, self.K)
        N = x.shape[0]
        xb, yb = DevBuf(host=x), DevBuf(N * self.M * 4)
        L.ggml_hip_mul_mat_q(self.h, xb.ptr, self.K, N, yb.ptr, self.M)
        y = yb.to_host(np.float32, (N, self.M))
        xb.free()
        yb.free()
        return y

    def free(self):
        if self.h:
            load().ggml_hip_weight_free(self.h)
            self.h = None


def ggcc_scan(path):
    """host-only parse of a GGCC v10 file: (hparams dict, ftype, [(name, ggml type, ne0, ne1, offset, bytes)])"""
    L = load()
    hp, ft = HParams(), C.c_int(0)
    buf = C.create_string_buffer(1 << 20)
    n = L.falcon_hip_ggcc_scan(os.fsencode(path), C.byref(hp), C.byref(ft), buf, len(buf))
    if n < 0:
        raise RuntimeError("not a readable GGCC v10 file: %s" % path)
    rows = []
    for line in buf.value.decode().splitlines():
        name, t, ne0, ne1, off, sz = line.rsplit(" ", 5)
        rows.append((name, int(t), int(ne0), int(ne1), int(off), int(sz)))
    assert len(rows) == n
    d = dict(n_vocab=hp.n_vocab, n_embd=hp.n_embd, n_head=hp.n_head, n_head_kv=hp.n_head_kv, n_layer=hp.n_layer, n_ff=hp.n_ff, two_norms=bool(hp.two_norms))
    return d, ft.value, rows


def plan_stages(path, n_stages, n_ctx=2048, n_batch=1, n_streams=1, vram_per_gpu=0):
    """host-only: ([(layer_begin, layer_end)], [device bytes per stage], fits) for a layer pipeline over a GGCC file"""
    lb, le = (C.c_int * n_stages)(), (C.c_int * n_stages)()
    sb = (C.c_size_t * n_stages)()
    rc = load().falcon_hip_plan_stages(os.fsencode(path), n_stages, n_ctx, n_batch, n_streams, vram_per_gpu, lb, le, sb)
    if rc < 0:
        raise RuntimeError("falcon_hip_plan_stages(%s, %d) failed" % (path, n_stages))
    return [(lb[i], le[i]) for i in range(n_stages)], [int(sb[i]) for i in range(n_stages)], rc == 0


def quantize_model(path_in, path_out, ftype, quantize_output_tensor=True, allow_requantize=False):
    """falcon_model_quantize on the device (falcon_hip_model_quantize); returns the 16-bin histogram"""
    hist = np.zeros(16, np.int64)
    rc = load().falcon_hip_model_quantize(os.fsencode(path_in), os.fsencode(path_out), int(ftype), int(bool(quantize_output_tensor)),
                                          int(bool(allow_requantize)), hist.ctypes.data)
    if rc != 0:
        raise RuntimeError("falcon_hip_model_quantize(%s, ftype %d) failed" % (path_in, ftype))
    return hist


def quantize_acts(act_type, x):
    """x: [N, K] f32 -> ggml block bytes [N, K/blck*tsize] produced on the device"""
    L = load()
    x = np.ascontiguousarray(x, np.float32)
    N, K = x.shape
    a = L.ggml_hip_acts_alloc(act_type, K, N)
    xb = DevBuf(host=x)
    L.ggml_hip_quantize_acts(a, xb.ptr, K, N)
    nbytes = N * (K // BLCK[act_type]) * TSIZE[act_type]
    ob = DevBuf(nbytes)
    L.ggml_hip_acts_export(a, N, ob.ptr)
    out = ob.to_host(np.uint8, (N, nbytes // N))
    for b in (xb, ob):
        b.free()
    L.ggml_hip_acts_free(a)
    return out


TENSOR_NAMES_7B = dict(ln_w="input_layernorm.weight", ln_b="input_layernorm.bias")
TENSOR_NAMES_40B = dict(ln_w="ln_mlp.weight", ln_b="ln_mlp.bias", ln2_w="ln_attn.weight", ln2_b="ln_attn.bias")


class FalconModel:
    """weights dict (tests/synth.py::make_model layout) -> device-resident model + context"""

    def __init__(self, weights, n_ctx, n_batch, rope_n_ctx=0, layer_begin=0, layer_end=0):
        L = load()
        hp = weights["hparams"]
        self.hp = hp
        wt = weights["wtype"]
        E, H, HKV, FF, V = hp["n_embd"], hp["n_head"], hp["n_head_kv"], hp["n_ff"], hp["n_vocab"]
        self.c_hp = HParams(V, E, H, HKV, hp["n_layer"], FF, 1 if hp.get("two_norms") else 0, layer_begin, layer_end or hp["n_layer"])
        self.m = L.falcon_hip_model_create(C.byref(self.c_hp))

        def put(name, t, arr, ne0, ne1):
            arr = np.ascontiguousarray(arr)
            L.falcon_hip_model_set_tensor(self.m, name.encode(), t, arr.ctypes.data, ne0, ne1)

        put("transformer.word_embeddings.weight", wt, weights["tok_emb"], E, V)
        put("lm_head.weight", wt, weights["lm_head"], E, V)
        put("transformer.ln_f.weight", F32, weights["out_norm_w"], E, 1)
        put("transformer.ln_f.bias", F32, weights["out_norm_b"], E, 1)
        names = TENSOR_NAMES_40B if hp.get("two_norms") else TENSOR_NAMES_7B
        for i, lw in enumerate(weights["layers"]):
            if lw is None:
                continue
            p = f"transformer.h.{i}."
            put(p + "self_attention.query_key_value.weight", wt, lw["qkv"], E, (H + 2 * HKV) * 64)
            put(p + "self_attention.dense.weight", wt, lw["wo"], E, E)
            put(p + "mlp.dense_h_to_4h.weight", wt, lw["up"], E, FF)
            put(p + "mlp.dense_4h_to_h.weight", wt, lw["down"], FF, E)
            for k, leaf in names.items():
                put(p + leaf, F32, lw[k], E, 1)
        self.ctx = L.falcon_hip_context_create(self.m, n_ctx, n_batch, rope_n_ctx)
        self.n_local = (layer_end or hp["n_layer"]) - layer_begin

    @classmethod
    def from_ggcc(cls, path, n_ctx, n_batch, rope_n_ctx=0, layer_begin=0, layer_end=0):
        """load a GGCC v10 model file (the reference's format) through falcon_hip_model_load_ggcc"""
        L = load()
        self = cls.__new__(cls)
        self.c_hp = HParams()
        self.m = L.falcon_hip_model_load_ggcc(os.fsencode(path), layer_begin, layer_end, C.byref(self.c_hp))
        if not self.m:
            raise RuntimeError("cannot load %s" % path)
        self.hp = dict(n_vocab=self.c_hp.n_vocab, n_embd=self.c_hp.n_embd, n_head=self.c_hp.n_head, n_head_kv=self.c_hp.n_head_kv,
                       n_layer=self.c_hp.n_layer, n_ff=self.c_hp.n_ff, two_norms=bool(self.c_hp.two_norms))
        self.ctx = L.falcon_hip_context_create(self.m, n_ctx, n_batch, rope_n_ctx)
        self.n_local = self.c_hp.layer_end - self.c_hp.layer_begin
        return self

    def new_context(self, n_ctx, n_batch=1, rope_n_ctx=0):
        """another context (own KV cache / scratch) over the same device weights: one per concurrent decode stream"""
        return load().falcon_hip_context_create(self.m, n_ctx, n_batch, rope_n_ctx)

    def eval(self, tokens, n_past, logits_all=True, want_hidden=False):
        L = load()
        tok = np.ascontiguousarray(tokens, np.int32)
        N = tok.size
        if want_hidden:
            L.falcon_hip_context_keep_hidden(self.ctx, 1)
        rc = L.falcon_hip_eval(self.ctx, tok.ctypes.data, N, n_past, 1 if logits_all else 0)
        if rc != 0:
            raise RuntimeError("falcon_hip_eval failed (%d)" % rc)
        rows = N if logits_all else 1
        lg = np.ctypeslib.as_array(L.falcon_hip_get_logits(self.ctx), (rows, self.hp["n_vocab"])).copy()
        if want_hidden:
            hid = np.empty((self.n_local + 1, N, self.hp["n_embd"]), np.float32)
            L.falcon_hip_get_hidden(self.ctx, hid.ctypes.data)
            L.falcon_hip_context_keep_hidden(self.ctx, 0)
            return lg, hid
        return lg

    def eval_token(self, token, n_past):
        """one token through the captured graph, no host round trip; logits() fetches the row"""
        rc = load().falcon_hip_eval_token(self.ctx, int(token), int(n_past))
        if rc != 0:
            raise RuntimeError("falcon_hip_eval_token failed (%d)" % rc)

    def logits(self):
        n = self.hp["n_vocab"]
        return np.ctypeslib.as_array(load().falcon_hip_get_logits(self.ctx), shape=(n,)).copy()

    def set_rope_n_ctx(self, n):
        load().falcon_hip_context_set_rope_n_ctx(self.ctx, int(n))

    def decode_greedy(self, first_token, n_past, n_steps, use_graph=False):
        L = load()
        L.falcon_hip_context_use_graph(self.ctx, 1 if use_graph else 0)
        out = np.zeros(n_steps, np.int32)
        rc = L.falcon_hip_decode_greedy(self.ctx, int(first_token), n_past, n_steps, out.ctypes.data)
        if rc != 0:
            raise RuntimeError("falcon_hip_decode_greedy failed (%d)" % rc)
        return out

    def perplexity(self, tokens, n_ctx, n_batch):
        """(summed NLL, scored tokens) of the reference's perplexity loop over a token stream"""
        tok = np.ascontiguousarray(tokens, np.int32)
        nll = C.c_double(0.0)
        n = load().falcon_hip_perplexity(self.ctx, tok.ctypes.data, tok.size, n_ctx, n_batch, C.byref(nll))
        return nll.value, n

    def set_fused(self, mode):
        """0 = op list, 1 = three launches per block, 2 (True) = two (default), 3 = one launch per block, 5 = two launches, ring form forced (4, the removed persistent engine, selects 2)"""
        load().falcon_hip_context_set_fused(self.ctx, 2 if mode is True else int(mode))

    def sync_error(self):
        return load().falcon_hip_context_sync_error(self.ctx)

    def weight_bytes(self):
        return load().falcon_hip_model_weight_bytes(self.m)

    def free(self):
        L = load()
        L.falcon_hip_context_free(self.ctx)
        L.falcon_hip_model_free(self.m)


class SeqContext:
    """n_seq independent sequences of one model advancing in lock step (falcon_hip_context_create_seqs): eval takes one
    token per sequence and returns one logits row per sequence"""

    def __init__(self, model, n_ctx, n_seq):
        self.model, self.n_seq = model, n_seq
        self.ctx = load().falcon_hip_context_create_seqs(model.m, n_ctx, n_seq, 0)
        if not self.ctx:
            raise RuntimeError("falcon_hip_context_create_seqs failed")

    def eval(self, tokens, n_past):
        L = load()
        tok = np.ascontiguousarray(tokens, np.int32)
        assert tok.size == self.n_seq
        rc = L.falcon_hip_eval_stage(self.ctx, tok.ctypes.data, None, tok.size, n_past, 1, None)
        if rc != 0:
            raise RuntimeError("falcon_hip_eval_stage failed (%d)" % rc)
        return np.ctypeslib.as_array(L.falcon_hip_get_logits(self.ctx), (self.n_seq, self.model.hp["n_vocab"])).copy()

    def free(self):
        load().falcon_hip_context_free(self.ctx)


class Pipeline:
    """falcon_hip_pipeline_* (csrc/falcon_pipeline.hip): rank `rank` of a `world`-stage layer pipeline over `model` (a
    FalconModel holding that rank's blocks). unique_id: 128 bytes from Pipeline.unique_id() on rank 0 (RCCL transport), or
    local=True for the in-process transport (all ranks in this process, see run_local)."""

    def __init__(self, model, rank, world, n_groups, batch, n_ctx, unique_id=None, local=False):
        L = load()
        self.model, self.rank, self.world, self.G, self.B = model, rank, world, n_groups, batch
        if local:
            self.p = L.falcon_hip_pipeline_create_local(model.m, rank, world, n_groups, batch, n_ctx)
        else:
            self.p = L.falcon_hip_pipeline_create(model.m, rank, world, unique_id, n_groups, batch, n_ctx)
        if not self.p:
            raise RuntimeError("falcon_hip_pipeline_create failed")

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(128)
        if load().falcon_hip_pipeline_unique_id(buf) != 0:
            raise RuntimeError("RCCL is not available")
        return buf.raw

    def set_tokens(self, tokens):
        tok = np.ascontiguousarray(tokens, np.int32)
        assert tok.size == self.G * self.B
        if load().falcon_hip_pipeline_set_tokens(self.p, tok.ctypes.data) != 0:
            raise RuntimeError("falcon_hip_pipeline_set_tokens failed")

    def run(self, rounds, n_past0):
        if load().falcon_hip_pipeline_run(self.p, rounds, n_past0) != 0:
            raise RuntimeError("falcon_hip_pipeline_run failed")

    @staticmethod
    def run_local(ranks, rounds, n_past0):
        arr = (C.c_void_p * len(ranks))(*[r.p for r in ranks])
        if load().falcon_hip_pipeline_run_local(arr, len(ranks), rounds, n_past0) != 0:
            raise RuntimeError("falcon_hip_pipeline_run_local failed")

    @staticmethod
    def attach_rccl(ranks):
        """the local job's hand-offs through a one-rank RCCL communicator (self send / recv) instead of device copies"""
        arr = (C.c_void_p * len(ranks))(*[r.p for r in ranks])
        if load().falcon_hip_pipeline_local_attach_rccl(arr, len(ranks)) != 0:
            raise RuntimeError("falcon_hip_pipeline_local_attach_rccl failed (RCCL missing or refused)")

    def rccl_ranks(self):
        return load().falcon_hip_pipeline_rccl_ranks(self.p)

    TRANSPORTS = {0: "none (one stage)", 1: "rccl", 2: "local (device copies in one process)", 3: "local over a one-rank RCCL communicator", 4: "shm (host shared memory between processes)",
                  5: "ipc (device-to-device copies into the peer's IPC-exported mailboxes)"}

    def transport(self):
        return self.TRANSPORTS.get(load().falcon_hip_pipeline_transport(self.p), "?")

    def history(self, first_round, n_rounds):
        """[n_rounds][n_groups * batch] sampled tokens (last rank; waits for the device); None on other ranks"""
        out = np.zeros((n_rounds, self.G * self.B), np.int32)
        rc = load().falcon_hip_pipeline_get_history(self.p, out.ctypes.data, first_round, n_rounds)
        if rc == -1:
            return None
        if rc != 0:
            raise RuntimeError("falcon_hip_pipeline_get_history failed (%d)" % rc)
        return out

    def free(self):
        load().falcon_hip_pipeline_free(self.p)


def pipeline_schedule(rank, world, n_groups, rounds):
    """host only: the slots of one rank as [(exchange ops [(kind, group, peer)], computed group or None, round)], kinds
    'send_hidden' 'send_token' 'recv_hidden' 'recv_token'"""
    L = load()
    names = ("send_hidden", "send_token", "recv_hidden", "recv_token")
    out = (C.c_int * 9)()
    T = L.falcon_hip_pipeline_schedule(rank, world, n_groups, rounds, -1, out)
    slots = []
    for t in range(T):
        L.falcon_hip_pipeline_schedule(rank, world, n_groups, rounds, t, out)
        ops = [(names[out[1 + 3 * i]], out[2 + 3 * i], out[3 + 3 * i]) for i in range(out[0])]
        slots.append((ops, None if out[7] < 0 else out[7], out[8]))
    return slots


class Vocab:
    """the tokenizer of a GGCC v10 file (falcon_tokenize / falcon_token_to_str of the reference); host only"""

    def __init__(self, path):
        L = load()
        self.h = L.falcon_hip_vocab_load_ggcc(os.fsencode(path))
        err = L.falcon_hip_vocab_error(self.h)
        if err:
            L.falcon_hip_vocab_free(self.h)
            self.h = None
            raise ValueError("%s: %s" % (path, err.decode()))
        self.n_vocab, self.n_merges = L.falcon_hip_vocab_size(self.h), L.falcon_hip_vocab_merges(self.h)

    def tokenize(self, text, add_bos=False, n_max=None):
        """token ids (np.int32); with n_max: the C return value when the buffer is too small (minus the count)"""
        L = load()
        raw = text if isinstance(text, bytes) else text.encode("utf-8")
        cap = n_max if n_max is not None else 4 * len(raw) + 8
        buf = np.zeros(max(cap, 1), np.int32)
        n = L.falcon_hip_tokenize(self.h, raw, buf.ctypes.data, cap, 1 if add_bos else 0)
        if n < 0:
            return n
        return buf[:n].copy()

    def token_bytes(self, tid):
        p = C.c_char_p()
        n = load().falcon_hip_token_to_bytes(self.h, int(tid), C.byref(p))
        if n < 0:
            raise IndexError(tid)
        return C.string_at(p, n)

    def detokenize(self, ids):
        return b"".join(self.token_bytes(i) for i in ids)

    def free(self):
        if self.h:
            load().falcon_hip_vocab_free(self.h)
            self.h = None
