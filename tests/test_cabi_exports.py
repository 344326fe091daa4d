"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/*.h declares (no compute)."""
import ctypes
import os
import re

import ggllm_cpp_amd as g

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b((?:ggml_hip|falcon_hip|ggml_cuda|ggml_init)_?\w*)\s*\(", src)))


def test_library_builds_and_exports_every_declared_symbol():
    g.build()
    lib = ctypes.CDLL(g.LIB_PATH)
    for header in ("ggml-hip-ops.h", "falcon-hip.h", os.path.join("dropin", "ggml-cuda.h")):
        names = _declared(header)
        assert len(names) > 10
        for n in names:
            assert hasattr(lib, n), f"{n} declared in include/{header} but not exported"


def test_python_signature_table_matches_headers():
    declared = set(_declared("ggml-hip-ops.h")) | set(_declared("falcon-hip.h"))
    assert set(g.EXPORTS_OPS) | set(g.EXPORTS_FALCON) == declared


def test_load_declares_all_signatures():
    g.build()
    g.load()
