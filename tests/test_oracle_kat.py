"""Known-answer tests of the oracle's leaf functions against vectors produced by the reference's own
code (oracle/ref_driver/ref_kat.cpp, compiled from /root/reference sources -> tests/golden/ref_kat.json)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, scene_path


def f32(bits):
    return np.uint32(bits).view(np.float32)


def bits(x):
    return int(np.float32(x).view(np.uint32))


@pytest.fixture(scope="module")
def kat():
    return json.load(open(os.path.join(GOLDEN, "ref_kat.json")))


@pytest.fixture(scope="module")
def L(oracle_py):
    L = oracle_py.lib()
    L.oracle_normalized_uint.restype = C.c_float
    L.oracle_normalized_uint.argtypes = [C.c_uint32]
    L.oracle_cosine_hemisphere.argtypes = [C.c_int, C.c_float, C.c_float, C.c_void_p]
    L.oracle_microfacet.argtypes = [C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
    L.oracle_conductor_reflectance.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]
    L.oracle_dielectric_reflectance.argtypes = [C.c_float, C.c_float, C.c_void_p]
    L.oracle_tangent_frame.argtypes = [C.c_void_p, C.c_void_p]
    return L


def test_hash32(kat, L):
    for x, h in kat["hash32"]:
        assert L.oracle_hash32(x) == h


def test_pcg32(kat, oracle_py):
    out = oracle_py.pcg_stream(kat["pcg"]["seed"], 0, len(kat["pcg"]["out"]))
    assert out.tolist() == kat["pcg"]["out"]


def test_normalized_uint(kat, L):
    for v, b in kat["normalized_uint"]:
        assert bits(L.oracle_normalized_uint(v)) == b


def test_sobol(kat, ta, oracle_py):
    hs = ta.HostScene(scene_path("materialtest/c2.json"), resolution=(16, 16))
    o = oracle_py.Oracle(hs)
    for idx, dim, scr, want in kat["sobol"]:
        assert o.sobol(idx, dim, scr) == want


def test_cosine_hemisphere(kat, L):
    out = np.zeros(3, np.float32)
    for xx, xy, wx, wy, wz in kat["cosine_hemisphere"]:
        L.oracle_cosine_hemisphere(0, f32(xx), f32(xy), out.ctypes.data)
        assert [bits(v) for v in out] == [wx, wy, wz]


def test_microfacet(kat, L):
    out = np.zeros(6, np.float32)
    for dist, alpha, xx, xy, mx, my, mz, D, pdf, vx, vy, vz, G1 in kat["microfacet"]:
        v = np.array([f32(vx), f32(vy), f32(vz)], np.float32)
        L.oracle_microfacet(0, 0 if dist == 0 else 2, f32(alpha), f32(xx), f32(xy), v.ctypes.data, out.ctypes.data)
        assert [bits(t) for t in out] == [mx, my, mz, D, pdf, G1]


def test_conductor(kat, L):
    eta = np.array([0.2004376970, 0.9240334304, 1.1022119527], np.float32)
    k = np.array([3.9129485033, 2.4528477015, 2.1421879552], np.float32)
    out = np.zeros(3, np.float32)
    for c, fx, fy, fz in kat["conductor"]:
        L.oracle_conductor_reflectance(eta.ctypes.data, k.ctypes.data, f32(c), out.ctypes.data)
        assert [bits(t) for t in out] == [fx, fy, fz]


def test_dielectric(kat, L):
    out = np.zeros(2, np.float32)
    for eta, c, F, ct in kat["dielectric"]:
        L.oracle_dielectric_reflectance(f32(eta), f32(c), out.ctypes.data)
        assert [bits(t) for t in out] == [F, ct]


def test_tangent_frame(kat, L):
    out = np.zeros(6, np.float32)
    for row in kat["tangent_frame"]:
        n = np.array([f32(b) for b in row[:3]], np.float32)
        L.oracle_tangent_frame(n.ctypes.data, out.ctypes.data)
        assert [bits(t) for t in out] == row[3:]


def test_fmath_exp(kat, ta, oracle_py):
    """ExponentialTransmittance evaluates fmath::exp_ps (1024-entry table + first-order term): the oracle's table
    and 96 x 3 results, incl. the clamped range, against the reference's own."""
    hs = ta.HostScene(scene_path("materialtest/c5.json"), resolution=(16, 9))
    o = oracle_py.Oracle(hs, **oracle_py.REF)
    lib = oracle_py.lib()
    tbl = np.zeros(1024, np.uint32)
    lib.oracle_exp_table(C.c_void_p(o._h), tbl.ctypes.data_as(C.c_void_p))
    assert np.array_equal(tbl, np.array(kat["fmath_exp_table"], np.uint32))
    for x, ex, ey, ez in kat["fmath_exp"]:
        xf = f32(x)
        inp = np.array([xf, xf*np.float32(0.5), -xf*np.float32(0.25)], np.float32)
        out = np.zeros(3, np.float32)
        lib.oracle_fast_exp(C.c_void_p(o._h), inp.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
        assert [int(v) for v in out.view(np.uint32)] == [ex, ey, ez]
