"""ctypes bindings for the CHECKER libraries (test infrastructure only).

  Oracle  -> oracle/liboracle.so          our CPU restatement (oracle_quants.c, oracle_falcon.c)
  Ref     -> oracle/_ref/libggml_ref*.so  the real reference compiled by oracle/Makefile (only where it
             was built: the dev container; the .so travels to the GPU box, the sources do not)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

F32, F16, Q4_0, Q4_1, Q5_0, Q5_1, Q8_0, Q8_1 = 0, 1, 2, 3, 6, 7, 8, 9
Q2_K, Q3_K, Q4_K, Q5_K, Q6_K, Q8_K = 10, 11, 12, 13, 14, 15
LEGACY = [Q4_0, Q4_1, Q5_0, Q5_1, Q8_0]
KQUANTS = [Q2_K, Q3_K, Q4_K, Q5_K, Q6_K]
WEIGHT_TYPES = LEGACY + KQUANTS
TYPE_NAME = {Q4_0: "q4_0", Q4_1: "q4_1", Q5_0: "q5_0", Q5_1: "q5_1", Q8_0: "q8_0", Q8_1: "q8_1",
             Q2_K: "q2_K", Q3_K: "q3_K", Q4_K: "q4_K", Q5_K: "q5_K", Q6_K: "q6_K", Q8_K: "q8_K"}
BLCK = {Q4_0: 32, Q4_1: 32, Q5_0: 32, Q5_1: 32, Q8_0: 32, Q8_1: 32,
        Q2_K: 256, Q3_K: 256, Q4_K: 256, Q5_K: 256, Q6_K: 256, Q8_K: 256}
TSIZE = {Q4_0: 18, Q4_1: 20, Q5_0: 22, Q5_1: 24, Q8_0: 34, Q8_1: 40,
         Q2_K: 84, Q3_K: 110, Q4_K: 144, Q5_K: 176, Q6_K: 210, Q8_K: 292}
VEC_DOT = {Q4_0: Q8_0, Q5_0: Q8_0, Q8_0: Q8_0, Q4_1: Q8_1, Q5_1: Q8_1,
           Q2_K: Q8_K, Q3_K: Q8_K, Q4_K: Q8_K, Q5_K: Q8_K, Q6_K: Q8_K}
ROUND_REFERENCE, ROUND_AVX = 0, 1


def row_bytes(t, k):
    return k // BLCK[t] * TSIZE[t]


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def build_oracle():
    """gcc the restatement (and, when /root/reference exists, the reference itself). Idempotent."""
    subprocess.check_call(["make", "-s", "-C", HERE, "oracle"])
    if os.path.isdir("/root/reference"):
        subprocess.check_call(["make", "-s", "-C", HERE, "ref", "ref_scalar"])


class HParams(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("n_vocab", "n_embd", "n_head", "n_head_kv", "n_layer", "n_ff", "n_ctx", "wtype", "two_norms", "rope_n_ctx")]


class Layer(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("qkv", "wo", "up", "down", "ln_w", "ln_b", "ln2_w", "ln2_b")]


class Model(C.Structure):
    _fields_ = [("hp", HParams), ("tok_emb", C.c_void_p), ("out_norm_w", C.c_void_p), ("out_norm_b", C.c_void_p),
                ("lm_head", C.c_void_p), ("layers", C.POINTER(Layer)), ("k_cache", C.c_void_p), ("v_cache", C.c_void_p)]


class _Base:
    """Shared model-eval plumbing: `weights` is the dict produced by tests/synth.py::make_model."""

    def _pack_model(self, w, n_ctx, rope_n_ctx=None):
        hp = w["hparams"]
        L = hp["n_layer"]
        layers = (Layer * L)()
        keep = []
        for i in range(L):
            lw = w["layers"][i]
            for name in ("qkv", "wo", "up", "down", "ln_w", "ln_b", "ln2_w", "ln2_b"):
                a = lw.get(name)
                if a is None:
                    setattr(layers[i], name, None)
                else:
                    a = np.ascontiguousarray(a)
                    keep.append(a)
                    setattr(layers[i], name, a.ctypes.data)
        D = hp["n_embd"] // hp["n_head"]
        kc = np.zeros(L * n_ctx * hp["n_head_kv"] * D, np.float32)
        vc = np.zeros_like(kc)
        m = Model()
        m.hp = HParams(hp["n_vocab"], hp["n_embd"], hp["n_head"], hp["n_head_kv"], L, hp["n_ff"], n_ctx,
                       w["wtype"], 1 if hp.get("two_norms") else 0, rope_n_ctx or n_ctx)
        for name in ("tok_emb", "out_norm_w", "out_norm_b", "lm_head"):
            a = np.ascontiguousarray(w[name])
            keep.append(a)
            setattr(m, name, a.ctypes.data)
        m.layers = layers
        m.k_cache = kc.ctypes.data
        m.v_cache = vc.ctypes.data
        keep += [layers, kc, vc]
        return m, keep


class Oracle(_Base):
    def __init__(self, path=None):
        path = path or os.path.join(HERE, "liboracle.so")
        if not os.path.exists(path):
            build_oracle()
        L = self.lib = C.CDLL(path)
        L.orc_fp16_to_fp32.restype = C.c_float
        L.orc_fp16_to_fp32.argtypes = [C.c_uint16]
        L.orc_fp32_to_fp16.restype = C.c_uint16
        L.orc_fp32_to_fp16.argtypes = [C.c_float]
        L.orc_quantize_row.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int64]
        L.orc_quantize_act.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int]
        L.orc_dequantize_row.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int64]
        L.orc_vec_dot.restype = C.c_float
        L.orc_vec_dot.argtypes = [C.c_int, C.c_int64, C.c_void_p, C.c_void_p]
        L.orc_mul_mat_q.argtypes = [C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_int]
        L.orc_norm.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]
        L.orc_layer_norm.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_gelu.restype = C.c_float
        L.orc_gelu.argtypes = [C.c_float]
        L.orc_rope_neox.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_rope_table.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.orc_rope_theta_scale.restype = C.c_float
        L.orc_rope_theta_scale.argtypes = [C.c_int, C.c_int]
        L.orc_softmax_rows.argtypes = [C.c_void_p, C.c_int64, C.c_int64]
        L.orc_gelu_table.restype = C.POINTER(C.c_uint16)
        L.orc_exp_table.restype = C.POINTER(C.c_uint16)
        L.orc_falcon_eval.argtypes = [C.POINTER(Model), C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_set_sum_order.argtypes = [C.c_int]
        L.orc_set_backend_batch.argtypes = [C.c_int]
        L.orc_set_kq_min_cols.argtypes = [C.c_int]
        L.orc_falcon_block_sampled.argtypes = [C.POINTER(Model), C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_falcon_head_rows.argtypes = [C.POINTER(Model), C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.orc_tables_init()

    def fp16_to_fp32(self, h):
        return self.lib.orc_fp16_to_fp32(int(h))

    def fp32_to_fp16(self, f):
        return self.lib.orc_fp32_to_fp16(float(f))

    def quantize(self, t, x):
        x = _f32(x).ravel()
        out = np.zeros(row_bytes(t, x.size), np.uint8)
        self.lib.orc_quantize_row(t, _ptr(x), _ptr(out), x.size)
        return out

    def quantize_act(self, act_t, x, flavour=ROUND_REFERENCE):
        x = _f32(x).ravel()
        out = np.zeros(row_bytes(act_t, x.size), np.uint8)
        self.lib.orc_quantize_act(act_t, _ptr(x), _ptr(out), x.size, flavour)
        return out

    def dequantize(self, t, blob, k):
        blob = np.ascontiguousarray(blob, np.uint8)
        y = np.zeros(k, np.float32)
        self.lib.orc_dequantize_row(t, _ptr(blob), _ptr(y), k)
        return y

    def vec_dot(self, wt, n, w, a):
        w = np.ascontiguousarray(w, np.uint8)
        a = np.ascontiguousarray(a, np.uint8)
        return self.lib.orc_vec_dot(wt, n, _ptr(w), _ptr(a))

    def mul_mat(self, wt, w, K, M, x, n_threads=4, flavour=ROUND_REFERENCE):
        w = np.ascontiguousarray(w, np.uint8)
        x = _f32(x).reshape(-1, K)
        N = x.shape[0]
        dst = np.zeros((N, M), np.float32)
        self.lib.orc_mul_mat_q(wt, _ptr(w), K, M, _ptr(x), N, _ptr(dst), n_threads, flavour)
        return dst

    def norm(self, x):
        x = _f32(x)
        y = np.zeros_like(x)
        self.lib.orc_norm(_ptr(x), x.shape[-1], x.size // x.shape[-1], _ptr(y))
        return y

    def layer_norm(self, x, w, b):
        x, w, b = _f32(x), _f32(w), _f32(b)
        y = np.zeros_like(x)
        self.lib.orc_layer_norm(_ptr(x), x.shape[-1], x.size // x.shape[-1], _ptr(w), _ptr(b), _ptr(y))
        return y

    def gelu(self, x):
        x = _f32(x)
        return np.array([self.lib.orc_gelu(float(v)) for v in x.ravel()], np.float32).reshape(x.shape)

    def gelu_table(self):
        return np.ctypeslib.as_array(self.lib.orc_gelu_table(), (1 << 16,)).copy()

    def exp_table(self):
        return np.ctypeslib.as_array(self.lib.orc_exp_table(), (1 << 16,)).copy()

    def rope(self, x, head_dim, n_head, N, n_past, n_ctx):
        y = _f32(x).copy()
        self.lib.orc_rope_neox(_ptr(y), head_dim, n_head, N, n_past, n_ctx)
        return y

    def rope_table(self, head_dim, n_pos, n_ctx):
        cs = np.zeros((n_pos, head_dim // 2, 2), np.float32)
        self.lib.orc_rope_table(_ptr(cs), head_dim, n_pos, n_ctx)
        return cs

    def softmax_rows(self, x):
        y = _f32(x).copy()
        self.lib.orc_softmax_rows(_ptr(y), y.shape[-1], y.size // y.shape[-1])
        return y

    def model(self, weights, n_ctx, rope_n_ctx=None):
        return _ModelRunner(self.lib.orc_falcon_eval, *self._pack_model(weights, n_ctx, rope_n_ctx), weights, True)


class Ref(_Base):
    """The real reference (only where oracle/_ref/*.so exists)."""

    def __init__(self, scalar=False):
        name = "libggml_ref_scalar.so" if scalar else "libggml_ref.so"
        path = os.path.join(HERE, "_ref", name)
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        L = self.lib = C.CDLL(path)
        L.ref_fp16_to_fp32.restype = C.c_float
        L.ref_fp16_to_fp32.argtypes = [C.c_uint16]
        L.ref_fp32_to_fp16.restype = C.c_uint16
        L.ref_fp32_to_fp16.argtypes = [C.c_float]
        L.ref_type_size.restype = C.c_size_t
        for f in ("ref_quantize_reference", "ref_quantize_native", "ref_quantize_dot"):
            getattr(L, f).argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.ref_dequantize.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        if hasattr(L, "ref_quantize_chunk"):
            L.ref_quantize_chunk.restype = C.c_size_t
            L.ref_quantize_chunk.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.ref_vec_dot.restype = C.c_float
        L.ref_vec_dot.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.ref_mul_mat.argtypes = [C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int]
        L.ref_norm.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]
        L.ref_gelu.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
        L.ref_rope_falcon.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.ref_scale_mask_softmax.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p]
        L.ref_falcon_eval.argtypes = [C.POINTER(Model), C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.ref_init()

    @staticmethod
    def available(scalar=False):
        return os.path.exists(os.path.join(HERE, "_ref", "libggml_ref_scalar.so" if scalar else "libggml_ref.so"))

    def fp16_to_fp32(self, h):
        return self.lib.ref_fp16_to_fp32(int(h))

    def fp32_to_fp16(self, f):
        return self.lib.ref_fp32_to_fp16(float(f))

    def quantize(self, t, x, native=False):
        x = _f32(x).ravel()
        out = np.zeros(row_bytes(t, x.size), np.uint8)
        (self.lib.ref_quantize_native if native else self.lib.ref_quantize_reference)(t, _ptr(x), _ptr(out), x.size)
        return out

    def quantize_chunk(self, t, x):
        """ggml_quantize_chunk over the whole array: (blocks, hist[16])"""
        x = _f32(x).ravel()
        out = np.zeros(row_bytes(t, x.size), np.uint8)
        hist = np.zeros(16, np.int64)
        n = self.lib.ref_quantize_chunk(t, _ptr(x), _ptr(out), 0, x.size, _ptr(hist))
        assert n == out.size
        return out, hist

    def quantize_dot(self, wt, x):
        x = _f32(x).ravel()
        out = np.zeros(row_bytes(VEC_DOT[wt], x.size), np.uint8)
        self.lib.ref_quantize_dot(wt, _ptr(x), _ptr(out), x.size)
        return out

    def dequantize(self, t, blob, k):
        blob = np.ascontiguousarray(blob, np.uint8)
        y = np.zeros(k, np.float32)
        self.lib.ref_dequantize(t, _ptr(blob), _ptr(y), k)
        return y

    def vec_dot(self, wt, n, w, a):
        w = np.ascontiguousarray(w, np.uint8)
        a = np.ascontiguousarray(a, np.uint8)
        return self.lib.ref_vec_dot(wt, n, _ptr(w), _ptr(a))

    def mul_mat(self, wt, w, K, M, x, n_threads=4):
        w = np.ascontiguousarray(w, np.uint8)
        x = _f32(x).reshape(-1, K)
        N = x.shape[0]
        dst = np.zeros((N, M), np.float32)
        self.lib.ref_mul_mat(wt, _ptr(w), K, M, _ptr(x), N, _ptr(dst), n_threads)
        return dst

    def norm(self, x):
        x = _f32(x)
        y = np.zeros_like(x)
        self.lib.ref_norm(_ptr(x), x.shape[-1], x.size // x.shape[-1], _ptr(y))
        return y

    def gelu(self, x):
        x = _f32(x)
        y = np.zeros_like(x)
        self.lib.ref_gelu(_ptr(x), x.size, _ptr(y))
        return y

    def rope(self, x, head_dim, n_head, N, n_past, n_ctx):
        x = _f32(x)
        y = np.zeros_like(x)
        self.lib.ref_rope_falcon(_ptr(x), head_dim, n_head, N, n_past, n_ctx, _ptr(y))
        return y

    def scale_mask_softmax(self, kq, n_past, scale):
        kq = _f32(kq)  # [n_head, N, n_kv]
        out = np.zeros_like(kq)
        n_head, N, n_kv = kq.shape
        self.lib.ref_scale_mask_softmax(_ptr(kq), n_kv, N, n_head, n_past, scale, _ptr(out))
        return out

    def model(self, weights, n_ctx, rope_n_ctx=None):
        return _ModelRunner(self.lib.ref_falcon_eval, *self._pack_model(weights, n_ctx, rope_n_ctx), weights, False)


class _ModelRunner:
    def __init__(self, fn, m, keep, weights, has_flavour):
        self.fn, self.m, self.keep, self.w, self.has_flavour = fn, m, keep, weights, has_flavour

    def block_sampled(self, lib, il, X, sample, pos0=0, k_prev=None, v_prev=None, n_threads=8, flavour=ROUND_REFERENCE, want_kv=False):
        """block `il` for the sampled tokens of a batch whose block inputs X[N, n_embd] are known (orc_falcon_block_sampled)"""
        hp = self.w["hparams"]
        X = _f32(X)
        N = X.shape[0]
        sample = np.ascontiguousarray(sample, np.int32)
        out = np.zeros((sample.size, hp["n_embd"]), np.float32)
        KV = hp["n_head_kv"] * 64
        ko = np.zeros((N, KV), np.float32) if want_kv else None
        vo = np.zeros((N, KV), np.float32) if want_kv else None
        kp = _f32(k_prev) if pos0 else None
        vp = _f32(v_prev) if pos0 else None
        lib.orc_falcon_block_sampled(C.byref(self.m), il, _ptr(X), N, pos0, _ptr(kp) if pos0 else None, _ptr(vp) if pos0 else None,
                                     _ptr(sample), sample.size, n_threads, flavour, _ptr(out), _ptr(ko) if want_kv else None, _ptr(vo) if want_kv else None)
        return (out, ko, vo) if want_kv else out

    def head_rows(self, lib, X, n_threads=8, flavour=ROUND_REFERENCE):
        hp = self.w["hparams"]
        X = _f32(X)
        lg = np.zeros((X.shape[0], hp["n_vocab"]), np.float32)
        lib.orc_falcon_head_rows(C.byref(self.m), _ptr(X), X.shape[0], n_threads, flavour, _ptr(lg))
        return lg

    def eval(self, tokens, n_past, n_threads=4, flavour=ROUND_REFERENCE, want_hidden=False):
        hp = self.w["hparams"]
        tok = np.ascontiguousarray(tokens, np.int32)
        N = tok.size
        logits = np.zeros((N, hp["n_vocab"]), np.float32)
        hidden = np.zeros((hp["n_layer"] + 1, N, hp["n_embd"]), np.float32) if want_hidden else None
        hptr = _ptr(hidden) if want_hidden else None
        if self.has_flavour:
            self.fn(C.byref(self.m), _ptr(tok), N, n_past, n_threads, flavour, _ptr(logits), hptr)
        else:
            self.fn(C.byref(self.m), _ptr(tok), N, n_past, n_threads, _ptr(logits), hptr)
        return (logits, hidden) if want_hidden else logits
