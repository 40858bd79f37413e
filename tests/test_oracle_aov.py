"""Auxiliary output buffers of the oracle (PathTracer.cpp:78-96,133-140 through cameras/OutputBuffer.hpp): properties the values must
have whatever the scene, on the CPU.  The accumulation itself is pinned to the reference by tests/test_ref_kat3.py; the GPU parity test
compares all of these buffers with the HIP core bit for bit."""
import numpy as np
import pytest

from conftest import scene_path

ALL = [("depth", True, True), ("normal", False, True), ("albedo", True, False), ("visibility", False, False)]


def render(ta, oracle_py, scene, res, spp, step, outputs, **mode):
    hs = ta.HostScene(scene_path(scene), resolution=res, output_buffers=outputs)
    shard = ta.Shard(hs.res_x, hs.res_y)
    o = oracle_py.Oracle(hs, **(mode or oracle_py.GPU))
    bufs = o.bind_output_buffers(shard.n_pixels)
    rgb, cnt = o.render_shard(shard, spp, spp_step=step)
    return hs, shard, rgb, cnt, bufs


def test_colour_buffer_is_unchanged_by_tracking(ta, oracle_py):
    hs = ta.HostScene(scene_path("cornell-box/scene_sobol.json"), resolution=(48, 27))
    shard = ta.Shard(hs.res_x, hs.res_y)
    plain, _ = oracle_py.Oracle(hs, **oracle_py.GPU).render_shard(shard, 8, spp_step=4)
    _, _, rgb, cnt, bufs = render(ta, oracle_py, "cornell-box/scene_sobol.json", (48, 27), 8, 4, ALL)
    assert np.array_equal(plain.view(np.uint32), rgb.view(np.uint32))
    assert np.array_equal(bufs["color"]["a"].view(np.uint32), rgb.view(np.uint32)) and np.array_equal(bufs["color"]["count"], cnt)


def test_first_hit_values_in_a_closed_diffuse_box(ta, oracle_py):
    """Cornell box (no environment): a sample whose primary ray hits the box records at that diffuse vertex -- depth is the hit distance,
    the normal a unit vector, the albedo the surface's (plus emission on the light); one that escapes adds depth 0 and -direction."""
    hs, shard, rgb, cnt, bufs = render(ta, oracle_py, "cornell-box/scene_sobol.json", (48, 27), 4, 4, ALL)
    for name in ("depth", "normal"):
        assert (bufs[name]["count"] == 4).all(), name
    inside = bufs["albedo"]["count"] == 4            # all four samples hit the box
    outside = bufs["albedo"]["count"] == 0           # all four escaped: nothing to show (no infinite light)
    assert inside.mean() > 0.4 and outside.any()
    d = bufs["depth"]
    mean = (d["a"]*((d["count"] + 1)//2)[:, None] + d["b"]*(d["count"]//2)[:, None])/np.maximum(d["count"], 1)[:, None]
    assert (mean[inside] > 0.5).all() and (mean[inside] < 20).all() and (mean[outside] == 0).all()
    assert (d["variance"] >= 0).all() and (d["variance"][outside] == 0).all()
    n = bufs["normal"]["a"]
    assert (np.linalg.norm(n, axis=1) <= 1.0 + 1e-5).all()  # a mean of unit vectors
    assert np.median(np.linalg.norm(n[inside], axis=1)) > 0.99  # mostly one face per pixel
    a = bufs["albedo"]["a"]
    assert (a >= 0).all() and (a[inside].max(axis=1) > 0.05).all()
    # light sampling is on and every vertex is diffuse: a recorded vertex saw its light sample's shadow test unless the sample failed before it
    v = bufs["visibility"]
    assert (v["count"] <= bufs["albedo"]["count"]).all() and (v["count"][inside] > 0).mean() > 0.7
    assert (v["a"] >= 0).all() and (v["a"] <= 1).all()


def test_escaped_and_specular_paths(ta, oracle_py):
    """Material test scene under an environment map: primary rays that escape add depth 0, -direction as the normal and the
    environment's radiance as the albedo; paths through the glass record at their first non-specular vertex, further away than the
    first hit."""
    hs, shard, rgb, cnt, bufs = render(ta, oracle_py, "next/aov.json", (64, 36), 4, 4, None)
    plain = ta.HostScene(scene_path("next/aov.json"), resolution=(64, 36))
    o = oracle_py.Oracle(plain, **oracle_py.GPU)
    W = hs.res_x
    pix = np.asarray(shard.pixel_index)
    # primary hit distances at the pixel centres, through the ray-level entry
    rays = np.zeros((shard.n_pixels, 8), np.float32)
    import ctypes as C
    for k, p in enumerate(pix):
        od = np.zeros(6, np.float32)
        oracle_py.lib().oracle_camera_ray(o._h, int(p % W), int(p//W), C.c_float(0.5), C.c_float(0.5), od.ctypes.data)
        rays[k, 0:3], rays[k, 3], rays[k, 4:7], rays[k, 7] = od[0:3], 1e-4, od[3:6], np.inf
    hits = o.trace_closest(rays)
    miss = hits[:, 3].view(np.uint32) == 0xFFFFFFFF
    assert miss.any() and (~miss).any()
    interior = np.zeros(shard.n_pixels, bool)
    # pixels whose whole 3x3 neighbourhood misses: all 4 samples escape at bounce 0
    m2 = np.zeros((hs.res_y, hs.res_x), bool)
    m2.ravel()[pix] = miss
    pad = np.pad(m2, 1, constant_values=False)
    all_miss = np.ones_like(m2)
    for dy in range(3):
        for dx in range(3):
            all_miss &= pad[dy:dy + hs.res_y, dx:dx + hs.res_x]
    interior = all_miss.ravel()[pix]
    assert interior.any()
    assert (bufs["depth"]["a"][interior] == 0).all() and (bufs["depth"]["count"][interior] == 4).all()
    assert np.allclose(np.linalg.norm(bufs["normal"]["a"][interior], axis=1), 1.0, atol=1e-2)
    assert (bufs["albedo"]["count"][interior] == 4).all() and (bufs["albedo"]["a"][interior] >= 0).all()
    assert (bufs["visibility"]["count"][interior] == 0).all()
    # somewhere a path crossed glass before recording: mean depth beyond the primary hit distance
    hit = ~miss & (bufs["depth"]["count"] > 0)
    assert (bufs["depth"]["a"][hit, 0] > hits[hit, 0]*1.05).any()


def test_two_half_buffers_hold_even_and_odd_samples(ta, oracle_py):
    hs, shard, rgb, cnt, one = render(ta, oracle_py, "cornell-box/scene_sobol.json", (32, 18), 6, 3, ["normal"])
    _, _, _, _, two = render(ta, oracle_py, "cornell-box/scene_sobol.json", (32, 18), 6, 3, [("normal", True, False)])
    assert np.array_equal(one["normal"]["count"], two["normal"]["count"])
    n = two["normal"]["count"].astype(np.float32)[:, None]
    combined = (two["normal"]["a"]*((two["normal"]["count"] + 1)//2)[:, None] + two["normal"]["b"]*(two["normal"]["count"]//2)[:, None])/np.maximum(n, 1)
    assert np.allclose(combined, one["normal"]["a"], atol=1e-5)


def test_reference_mode_records_the_same_first_vertices(ta, oracle_py):
    """The tile-sequential reference mode and the path-local GPU mode differ in their random streams only: depth and normal of pixels
    whose samples all record at a diffuse first hit agree to within the pixel filter's jitter."""
    _, _, _, _, g = render(ta, oracle_py, "cornell-box/scene_sobol.json", (32, 18), 16, 16, ["depth"])
    hs = ta.HostScene(scene_path("cornell-box/scene_sobol.json"), resolution=(32, 18), output_buffers=["depth"])
    o = oracle_py.Oracle(hs, **oracle_py.REF)
    bufs = o.bind_output_buffers(hs.res_x*hs.res_y)
    o.render_image(16, spp_step=16)
    shard = ta.Shard(hs.res_x, hs.res_y)
    ref_depth = bufs["depth"]["a"][np.asarray(shard.pixel_index), 0]
    hit = (ref_depth > 0) & (g["depth"]["a"][:, 0] > 0)
    assert hit.mean() > 0.4
    assert np.median(np.abs(ref_depth[hit] - g["depth"]["a"][hit, 0])/ref_depth[hit]) < 0.02
