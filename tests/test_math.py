"""include/tgpu_math.h (the arithmetic contract): accuracy vs float64 libm on the ranges the path uses."""
import ctypes as C

import numpy as np
import pytest


def ulp_err(got, ref64):
    ref32 = ref64.astype(np.float32)
    ulp = np.spacing(np.abs(ref32)).astype(np.float64)
    return np.abs(got.astype(np.float64) - ref64)/np.maximum(ulp, 1e-45)


@pytest.fixture(scope="module")
def ev(oracle_py):
    L = oracle_py.lib()
    L.oracle_eval_math.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]

    def f(kind, a, b=None):
        a = np.ascontiguousarray(a, np.float32)
        b = np.ascontiguousarray(b if b is not None else np.zeros_like(a), np.float32)
        out = np.zeros_like(a)
        L.oracle_eval_math(kind, a.ctypes.data, b.ctypes.data, a.size, out.ctypes.data)
        return out
    return f


def test_accuracy(ev):
    rng = np.random.default_rng(7)
    x = rng.uniform(-2*np.pi, 2*np.pi, 400000).astype(np.float32)
    assert ulp_err(ev(0, x), np.sin(x.astype(np.float64))).max() < 2.0
    assert ulp_err(ev(1, x), np.cos(x.astype(np.float64))).max() < 2.0
    e = rng.uniform(-85, 20, 400000).astype(np.float32)
    assert ulp_err(ev(2, e), np.exp(e.astype(np.float64))).max() < 2.0
    l = np.concatenate([rng.uniform(1e-8, 4, 200000), rng.uniform(0, 1, 200000)**8 + 1e-30]).astype(np.float32)
    assert ulp_err(ev(3, l), np.log(l.astype(np.float64))).max() < 2.0
    y, xx = rng.uniform(-1, 1, 400000).astype(np.float32), rng.uniform(-1, 1, 400000).astype(np.float32)
    assert ulp_err(ev(4, y, xx), np.arctan2(y.astype(np.float64), xx.astype(np.float64))).max() < 4.0
    a = rng.uniform(-1, 1, 400000).astype(np.float32)
    assert ulp_err(ev(5, a), np.arccos(a.astype(np.float64))).max() < 2.0


def test_special_values(ev):
    assert ev(2, [0.0])[0] == 1.0 and ev(2, [-100.0])[0] == 0.0 and np.isinf(ev(2, [100.0])[0])
    assert ev(3, [1.0])[0] == 0.0 and np.isneginf(ev(3, [0.0])[0]) and np.isnan(ev(3, [-1.0])[0])
    assert ev(5, [1.0])[0] == 0.0 and abs(ev(5, [-1.0])[0] - np.pi) < 1e-6
    assert ev(0, [0.0])[0] == 0.0 and ev(1, [0.0])[0] == 1.0
    assert ev(4, [0.0], [0.0])[0] == 0.0 and abs(ev(4, [0.0], [-1.0])[0] - np.pi) < 1e-6
