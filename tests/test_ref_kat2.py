"""Known-answer tests against tests/golden/ref_kat2.json, written by oracle/_ref/ref_kat2 -- a driver compiled by the committed recipe
(oracle/Makefile.ref) from the reference's own sources where they lie: bvh/BvhBuilder.cpp + bvh/BinaryBvh.hpp, cameras/
ReconstructionFilter.cpp, math/Mat4f.cpp, sampling/Distribution2D.hpp and thirdparty/stbi decoding the vendored environment map.
They pin, above leaf level: the instance BVH2 (build + visit order, which decides which instance a ray hits: Instance.cpp:290-311),
the host loader's pixel-filter table, Radiance .hdr decoder, environment importance tables and rotation matrices.  CPU only."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, scene_path


@pytest.fixture(scope="module")
def kat():
    return json.load(open(os.path.join(GOLDEN, "ref_kat2.json")))


def f32(bits):
    return np.array(bits, dtype=np.uint32).view(np.float32)


def fnv1a_words(a):
    """FNV-1a over 32-bit words (the driver's digest)."""
    h = 1469598103934665603
    for w in np.ascontiguousarray(a).view(np.uint32).ravel().tolist():
        h = ((h ^ w)*1099511628211) & 0xFFFFFFFFFFFFFFFF
    return f"{h:016x}"


def test_instance_bvh_visit_order_matches_the_reference(kat, oracle_py):
    L = oracle_py.lib()
    L.oracle_kat_binary_bvh.restype = C.c_uint32
    L.oracle_kat_binary_bvh.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
    total = 0
    for s in kat["binary_bvh"]:
        boxes = f32(s["boxes"])
        n = len(boxes)//6
        assert n in (50, 300)
        for ray in s["rays"]:
            o, d = f32(ray["o"]), f32(ray["d"])
            out = np.zeros(2*1024, np.uint32)
            k = L.oracle_kat_binary_bvh(boxes.ctypes.data, n, o.ctypes.data, d.ctypes.data, out.ctypes.data, 1024)
            got = out[:2*k].reshape(-1, 2).tolist()
            assert got == ray["visits"], f"{n} boxes: visit sequence differs"
            total += k
    assert total > 300


def test_host_pixel_filter_table_matches_the_reference(kat, ta):
    """ReconstructionFilter::sample (ReconstructionFilter.hpp:86-103) replayed on the host loader's table in float32."""
    hs = ta.HostScene(scene_path("materialtest/c2.json"), resolution=(32, 32))
    cam = hs.desc.contents.camera
    cdf = np.array(list(cam.filter_cdf), np.float32)
    bin_size = np.float32(cam.filter_bin_size)
    assert f32([kat["filter"]["width"]])[0] == np.float32(1.0)

    def sample1(xi):
        xi = np.float32(xi)
        negative = xi < np.float32(0.5)
        xi = xi*np.float32(2) if negative else (xi - np.float32(0.5))*np.float32(2)
        idx = 30
        for i in range(30):
            if xi < cdf[i]:
                idx = i
                break
        pdf = cdf[idx] - cdf[idx - 1]
        u = bin_size*(np.float32(idx) + (xi - cdf[idx - 1])/pdf)
        return -u if negative else u

    for row in kat["filter"]["samples"]:
        xi = f32(row[:2])
        want = f32(row[2:4])
        got = np.array([sample1(xi[0]), sample1(xi[1])], np.float32)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (xi, got, want)


def test_host_hdr_decoder_and_environment_tables_match_the_reference(kat, ta):
    hs = ta.HostScene(scene_path("materialtest/c2.json"), resolution=(32, 32))
    d = hs.desc.contents
    env = d.env
    w, h = kat["envmap"]["w"], kat["envmap"]["h"]
    assert (env.dist_w, env.dist_h) == (w, h)
    bm = d.bitmaps[d.textures[env.emission_tex].bitmap]
    texels = np.ctypeslib.as_array(bm.texels, shape=(w*h*3,))
    assert fnv1a_words(texels) == kat["envmap"]["texels"], "Radiance .hdr decode differs from stb_image's"
    pdf = np.ctypeslib.as_array(env.pdf, shape=(h, w))
    cdf = np.ctypeslib.as_array(env.cdf, shape=(h, w + 1))
    mpdf = np.ctypeslib.as_array(env.marginal_pdf, shape=(h,))
    mcdf = np.ctypeslib.as_array(env.marginal_cdf, shape=(h + 1,))
    cell = (pdf*mpdf[:, None]).astype(np.float32)          # Distribution2D::pdf(row, column)
    assert fnv1a_words(cell) == kat["envmap"]["cell_pdf"]
    one, zero = np.float32(1), np.float32(0)
    for row in kat["envmap"]["warp"]:                        # Distribution2D::warp (Distribution2D.hpp:66-76)
        ux, uy = f32(row[:2])
        r = int(np.searchsorted(mcdf, uy, side="right")) - 1
        vy = min(max((uy - mcdf[r])/mpdf[r], zero), one)
        c = int(np.searchsorted(cdf[r], ux, side="right")) - 1
        vx = min(max((ux - cdf[r, c])/pdf[r, c], zero), one)
        assert (r, c) == (row[2], row[3])
        got = np.array([vx, vy, cell[min(max(r, 0), h - 1), min(max(c, 0), w - 1)]], np.float32)
        assert np.array_equal(got.view(np.uint32), np.array(row[4:7], np.uint32)), row


def test_host_rotation_matrices_match_the_reference(kat, ta):
    L = ta.host_lib()
    L.tgh_rot_yxz.argtypes = [C.c_void_p, C.c_void_p]
    for case in kat["rot_yxz"]:
        rot = f32(case["rot"])
        out = np.zeros(9, np.float32)
        L.tgh_rot_yxz(rot.ctypes.data, out.ctypes.data)
        assert out.view(np.uint32).tolist() == case["m"]
