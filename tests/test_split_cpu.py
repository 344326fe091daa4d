"""CPU: the row ranges of the reference's -ts tensor split (ggml_hip_tensor_split_rows, host code of libggml_hip.so) against a
restatement of ggml_cuda_set_tensor_split (ggml-cuda.cu:2050-2077) + the row_low / row_high lines of
ggml_cuda_transform_tensor (:3044-3052) in float32 arithmetic."""
import ctypes as C

import numpy as np
import pytest

import ggllm_cpp_amd as g


def _ref_rows(ts, nrows):
    n = len(ts)
    f = np.float32
    start = [f(0)] * (n + 1)
    if all(t == 0 for t in ts):
        for i in range(n):
            start[i] = f(i) / f(n)
    else:
        s = f(0)
        for i in range(n):
            start[i] = s
            s = f(s + f(ts[i]))
        for i in range(n):
            start[i] = f(1) if f(ts[i]) / s == 0 else f(start[i] / s)
    out = []
    for i in range(n):
        lo = 0 if i == 0 else int(f(nrows) * start[i])
        hi = nrows if i == n - 1 else int(f(nrows) * start[i + 1])
        out.append((lo, max(hi, lo)))
    return out


@pytest.mark.parametrize("ts", [[1, 1], [3, 1], [0, 0, 0, 0], [1, 2, 3, 4], [0.6, 0.4], [1, 0, 1], [2, 2, 0], [1] * 8, [0.13, 0.29, 0.58], [5, 1, 1, 1, 1, 1, 1, 1]])
@pytest.mark.parametrize("nrows", [4544, 4672, 18176, 9216, 65024, 7, 1])
def test_row_ranges_follow_the_reference(ts, nrows):
    L = g.load()
    n = len(ts)
    arr = (C.c_float * n)(*ts)
    lo, hi = (C.c_int64 * n)(), (C.c_int64 * n)()
    L.ggml_hip_tensor_split_rows(arr, n, nrows, lo, hi)
    got = list(zip(lo, hi))
    assert got == _ref_rows(ts, nrows)
    assert got[0][0] == 0 and got[-1][1] == nrows
    cover = sorted(r for r in got if r[1] > r[0])
    assert all(a[1] == b[0] for a, b in zip(cover, cover[1:])) and sum(b - a for a, b in cover) == nrows or any(t == 0 for t in ts)
