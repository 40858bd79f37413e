"""cameras/OutputBuffer.hpp of the reference (compiled from its own sources by oracle/Makefile.ref, driver oracle/ref_driver/ref_kat3.cpp)
against the oracle's accumulation: running means, the two half buffers, the variance sums and the counts, bit for bit."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN

KAT = json.load(open(os.path.join(GOLDEN, "ref_kat3.json")))
COUNTS = (0, 1, 2, 7, 24, 40)


class Gen:
    """The sample generator of ref_kat3.cpp (64-bit LCG, top bits)."""

    def __init__(self, seed):
        self.s = seed

    def next(self):
        self.s = (self.s*6364136223846793005 + 1442695040888963407) & 0xFFFFFFFFFFFFFFFF
        return self.s >> 33

    def value(self):
        r = self.next()
        kind = r % 23
        if kind == 0:
            return np.float32(np.nan)
        if kind == 1:
            return np.float32(np.inf)
        if kind == 2:
            return np.float32(-np.inf)
        mag = np.float32(self.next() % 100000)/np.float32(1000.0)
        if kind < 8:
            mag = np.float32(mag*np.float32(1e-3))
        return np.float32(-mag) if (r & 0x10000) else np.float32(mag)


@pytest.mark.parametrize("case", KAT["output_buffer"], ids=lambda c: f"{c['type']}-two{int(c['two_buffer_variance'])}-var{int(c['sample_variance'])}")
def test_output_buffer_accumulation(oracle_py, case):
    ch = case["channels"]
    g = Gen(case["seed"])
    pixel, values = [], []
    for p, n in enumerate(COUNTS):
        for _ in range(n):
            pixel.append(p)
            values.append([g.value() for _ in range(ch)])
    pixel = np.array(pixel, np.uint32)
    values = np.array(values, np.float32).reshape(-1, ch)
    n_pix = len(COUNTS)
    a = np.zeros((n_pix, ch), np.float32)
    b = np.zeros((n_pix, ch), np.float32) if case["two_buffer_variance"] else None
    var = np.zeros((n_pix, ch), np.float32) if case["sample_variance"] else None
    cnt = np.zeros(n_pix, np.uint32)
    oracle_py.lib().oracle_kat_output_buffer(ch, len(pixel), pixel.ctypes.data, values.ctypes.data, a.ctypes.data,
                                             b.ctypes.data if b is not None else None, var.ctypes.data if var is not None else None, cnt.ctypes.data)
    assert cnt.tolist() == case["count"]
    assert a.view(np.uint32).ravel().tolist() == case["a"]
    if b is not None:
        assert b.view(np.uint32).ravel().tolist() == case["b"]
    if var is not None:
        assert var.view(np.uint32).ravel().tolist() == case["variance"]
    # OutputBuffer::operator[] (the value the image writers and Camera::getLinear read)
    if b is not None:
        n = cnt.astype(np.float32)[:, None]
        ca = ((cnt + 1)//2).astype(np.float32)[:, None]
        cb = (cnt//2).astype(np.float32)[:, None]
        mean = (a*ca + b*cb)/np.maximum(n, np.float32(1))
    else:
        mean = a
    assert mean.astype(np.float32).view(np.uint32).ravel().tolist() == case["mean"]
