"""generate ggllm.cpp_amd/csrc/fq_exp_fix.h ON AN MI355X: the fp16 inputs whose exp() the f32 fast path of exp_f16_formula (csrc/fq_device.h) cannot decide
(within 4 f32-ulps of an fp16 rounding boundary), with the entries of the host-built table (ggml.c:4276-4290 table_exp_f16). Prints the header to stdout."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ggllm_cpp_amd as g
g.init(0); L = g.load()
buf = np.zeros(8192, np.uint32)
n = L.ggml_hip_debug_exp_boundary(buf.ctypes.data, buf.size)
assert 0 <= n <= buf.size, n
v = buf[:n]
print("// fq_exp_fix.h -- (input bits << 16 | table entry) of the fp16 inputs whose exp() the f32 fast path of exp_f16_formula (fq_device.h) cannot decide.")
print("// GENERATED on an MI355X by scripts/gpu_exp_boundary.py from ggml_hip_debug_exp_boundary (%d of 63488 inputs); an input missing here takes the f64 path." % n)
print("#pragma once")
print("#define FQ_EXP_FIX_N %d" % n)
print("static __device__ const unsigned fq_exp_fix[%d] = {" % max(n, 1))
for i in range(0, n, 8):
    print("    " + ", ".join("0x%08Xu" % x for x in v[i:i + 8]) + ",")
if n == 0:
    print("    0u")
print("};")
