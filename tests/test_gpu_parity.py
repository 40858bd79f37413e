"""Parity tests proper: the HIP path (through the C ABI) against the CPU oracle on identical inputs.
Bar: BIT-EXACT framebuffers (the oracle in GPU mode evaluates the same arithmetic contract:
include/tgpu_math.h, path-local PCG, IEEE division).  Run with `pytest -m gpu` on an MI355X."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import GOLDEN, edge_and_vertex_rays, scene_path

pytestmark = pytest.mark.gpu


def render_gpu(ta, ctx, scene, shard, spp, spp_step, max_paths=0):
    ctx.upload(scene, shard, max_paths)
    ctx.render(spp, spp_step=spp_step)
    return ctx.download()


def assert_bit_exact(got, ref, what):
    (g_rgb, g_cnt), (r_rgb, r_cnt) = got, ref
    assert np.array_equal(g_cnt, r_cnt), f"{what}: sample counts differ"
    bad = (g_rgb != r_rgb).any(axis=1)
    assert not bad.any(), (f"{what}: {int(bad.sum())} of {len(bad)} pixels differ, max abs diff "
                           f"{np.abs(g_rgb - r_rgb).max():.3e}")


@pytest.mark.parametrize("scene,res,spp,step", [
    ("materialtest/c2.json", (128, 72), 16, 16),          # config 2: Lambert + RoughConductor + env NEE/MIS
    ("materialtest/c2.json", (100, 57), 12, 4),           # ragged tiles, 3 passes
    ("materialtest/c1_noadapt.json", (128, 72), 16, 8),   # stock scene: SmoothCoat (supplemental-stream lobe picks)
    ("cornell-box/scene_sobol.json", (96, 54), 16, 16),   # quads + cubes + quad light + Null BSDF
    ("materialtest/c2.json", (1, 1), 4, 4),               # minimal image
    # "instances" primitives (a25), reference semantics: BVH2 order, last hit wins, double-transformed hit point
    ("instances/inst_cubes.json", (128, 72), 8, 8),       # cube + quad masters, 16 instances, env light
    ("instances/inst_many.json", (96, 54), 4, 4),         # 294 instances: binned-SAH levels
    ("instances/inst_mesh.json", (96, 54), 8, 8),         # mesh masters (own BVH4 each) + quad light (finite end cap)
    # config 5 (a26): rough_dielectric + homogeneous media (volume list, fmath exp, volume NEE)
    ("materialtest/c5.json", (128, 72), 16, 8),
    ("materialtest/c5_absorb.json", (128, 72), 8, 8),
    ("cornell-box/scene_fog.json", (96, 54), 16, 16),      # camera medium + quad light
    # the fork's example scene: non-exponential media behind ForwardBsdf walls, 'power' lights, uniform sampler
    ("non-exponential/scene.json", (128, 128), 16, 16),
    # specular BSDFs (shared generic shade kernel)
    ("next/mirror.json", (128, 72), 8, 8),
    ("next/dielectric.json", (128, 72), 8, 8),
    ("next/conductor.json", (128, 72), 8, 8),
    ("next/plastic.json", (128, 72), 8, 8),
    ("next/rough_plastic.json", (128, 72), 8, 8),
    ("next/dielectric_hg_medium.json", (128, 72), 8, 8),   # Henyey-Greenstein phase function
    ("next/sphere_disk.json", (128, 72), 8, 8),            # spheres, a disk, a sphere light next to the env light
    ("next/oren_nayar.json", (128, 72), 8, 8),
    ("next/phong.json", (128, 72), 8, 8),
    # nested BSDFs and the general forward lobe (pass-through surfaces that tint radiance and shadow rays)
    ("next/mixed.json", (128, 72), 8, 8),
    ("next/rough_coat.json", (128, 72), 8, 8),
    ("next/transparency.json", (128, 72), 8, 8),
    ("next/thinsheet.json", (128, 72), 8, 8),
    ("next/ldr_texture.json", (128, 72), 8, 8),  # 8-bit PNG bitmaps (gamma table, palette, scalar average)
    ("next/ldr_texture2.json", (128, 72), 8, 8), # the same through .bmp / .tga files
    ("next/ldr_texture3.json", (128, 72), 8, 8), # and through .jpg files (4:2:0, 4:2:2 + restart markers, 4:4:4, grey)
    ("next/fog_nested.json", (96, 54), 16, 8),   # a mixed wall and a rough_coat box inside a camera medium: k_volume leaves those surface steps to k_volume_nested
    ("next/ldr_texture4.json", (128, 72), 8, 8), # progressive .jpg textures
    # the fork's remaining transmittance models in the Cornell box's camera medium
    ("next/trans_erlang.json", (96, 54), 8, 8),
    ("next/trans_davis.json", (96, 54), 8, 8),
    ("next/trans_davis_weinstein.json", (96, 54), 8, 8),
    ("next/trans_interpolated.json", (96, 54), 8, 8),
    ("next/trans_interpolated2.json", (96, 54), 8, 8),
])
def test_framebuffer_bit_exact(ta, oracle_py, gpu_ctx, scene, res, spp, step):
    hs = ta.HostScene(scene_path(scene), resolution=res)
    shard = ta.Shard(hs.res_x, hs.res_y)
    got = render_gpu(ta, gpu_ctx, hs, shard, spp, step)
    ref = oracle_py.Oracle(hs, **oracle_py.GPU).render_shard(shard, spp, spp_step=step)
    assert_bit_exact(got, ref, f"{scene} {res} {spp}spp")


def test_instance_kernel_forms_agree(ta, oracle_py, gpu_ctx, monkeypatch):
    """Scenes with instances have two forms of each traversal kernel: nested loops (k_extend / k_shadow, one ray per lane) and the
    flat state machine behind persistent lanes (k_extend_iq + k_classify / k_shadow_iq).  All four combinations give the oracle's bits."""
    hs = ta.HostScene(scene_path("instances/inst_mesh.json"), resolution=(96, 54))
    shard = ta.Shard(hs.res_x, hs.res_y)
    ref = oracle_py.Oracle(hs, **oracle_py.GPU).render_shard(shard, 8, spp_step=4)
    for ext_q, sh_q in (("1", "4"), ("1", "0"), ("0", "4"), ("0", "0")):
        monkeypatch.setenv("TGPU_INST_EXTEND_Q", ext_q)
        monkeypatch.setenv("TGPU_SHADOW_Q", sh_q)
        monkeypatch.setenv("TGPU_TAIL_MAX", "0")
        got = render_gpu(ta, gpu_ctx, hs, shard, 8, 4)
        assert_bit_exact(got, ref, f"inst_mesh, extend queue {ext_q}, shadow queue {sh_q}")
    for k in ("TGPU_INST_EXTEND_Q", "TGPU_SHADOW_Q", "TGPU_TAIL_MAX"):
        monkeypatch.delenv(k, raising=False)
    gpu_ctx.upload(hs, shard)


def test_instance_rays_bit_exact(ta, oracle_py, gpu_ctx):
    """Ray level: closest hits (t, u, v; instance hits report kind 3 << 29 | instance) and any-hit against the oracle."""
    hs = ta.HostScene(scene_path("instances/inst_mesh.json"), resolution=(64, 36))
    gpu_ctx.upload(hs, ta.Shard(hs.res_x, hs.res_y))
    o = oracle_py.Oracle(hs, **oracle_py.GPU)
    rng = np.random.default_rng(3)
    n = 20000
    org = rng.uniform(-4, 4, (n, 3)).astype(np.float32)
    org[:, 1] = np.abs(org[:, 1]) + 0.2
    tgt = rng.uniform(-2, 2, (n, 3)).astype(np.float32)
    d = tgt - org
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.concatenate([org, np.full((n, 1), 1e-4, np.float32), d.astype(np.float32), np.full((n, 1), np.inf, np.float32)], axis=1)
    got, ref = gpu_ctx.trace_closest(rays), o.trace_closest(rays)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    kinds = got[:, 3].view(np.uint32) >> 29
    assert (kinds == 3).sum() > 1000 and (got[:, 3].view(np.uint32) == 0xFFFFFFFF).sum() > 100
    rays[:, 7] = rng.uniform(0.5, 6.0, n).astype(np.float32)
    assert np.array_equal(gpu_ctx.trace_occluded(rays), o.trace_occluded(rays))


def test_settings_variants(ta, oracle_py, gpu_ctx):
    for kw in (dict(light_sampling=False), dict(max_bounces=2), dict(max_bounces=1), dict(sobol=False)):
        hs = ta.HostScene(scene_path("materialtest/c2.json"), resolution=(64, 36), **kw)
        shard = ta.Shard(hs.res_x, hs.res_y)
        got = render_gpu(ta, gpu_ctx, hs, shard, 8, 8)
        ref = oracle_py.Oracle(hs, **oracle_py.GPU).render_shard(shard, 8, spp_step=8)
        assert_bit_exact(got, ref, str(kw))


def test_sub_batching_and_pass_splitting_do_not_change_the_image(ta, gpu_ctx):
    hs = ta.HostScene(scene_path("materialtest/c2.json"), resolution=(96, 54))
    shard = ta.Shard(hs.res_x, hs.res_y)
    a = render_gpu(ta, gpu_ctx, hs, shard, 16, 16)
    b = render_gpu(ta, gpu_ctx, hs, shard, 16, 16, max_paths=4096)   # 20+ pixel batches per pass
    c = render_gpu(ta, gpu_ctx, hs, shard, 16, 4)                    # 4 passes of 4 spp
    assert_bit_exact(b, a, "sub-batched")
    assert_bit_exact(c, a, "4x4spp vs 1x16spp")


def test_tile_shards_reassemble_bit_identically(ta, gpu_ctx):
    """Multi-GPU determinism by construction: the union of world=3 shards equals the world=1 image."""
    hs = ta.HostScene(scene_path("materialtest/c2.json"), resolution=(112, 64))
    full = ta.Shard(hs.res_x, hs.res_y)
    ref_rgb, _ = render_gpu(ta, gpu_ctx, hs, full, 8, 8)
    ref = full.scatter(ref_rgb)
    img = np.zeros_like(ref)
    for r in range(3):
        sh = ta.Shard(hs.res_x, hs.res_y, 0xBA5EBA11, r, 3)
        rgb, _ = render_gpu(ta, gpu_ctx, hs, sh, 8, 8)
        img = sh.scatter(rgb, img)
    assert np.array_equal(img, ref)


def _random_rays(n, seed, lo=-2.0, hi=3.5):
    rng = np.random.default_rng(seed)
    o = rng.uniform(lo, hi, (n, 3)).astype(np.float32)
    target = rng.uniform(-0.6, 0.9, (n, 3)).astype(np.float32)
    d = target - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.zeros((n, 8), np.float32)
    rays[:, 0:3] = o
    rays[:, 3] = 1e-4
    rays[:, 4:7] = d.astype(np.float32)
    rays[:, 7] = np.where(rng.random(n) < 0.3, rng.uniform(0.5, 4.0, n), np.inf).astype(np.float32)
    return rays


@pytest.mark.parametrize("scene", ["materialtest/c2.json", "cornell-box/scene_sobol.json"])
def test_ray_level_parity(ta, oracle_py, gpu_ctx, scene):
    """TraceableScene::intersect / occluded on a dumped ray batch: (t,u,v,prim) bit-exact."""
    hs = ta.HostScene(scene_path(scene), resolution=(32, 32))
    gpu_ctx.upload(hs, ta.Shard(32, 32))
    orc = oracle_py.Oracle(hs, **oracle_py.GPU)
    rays = _random_rays(200000, 3)
    if "cornell" in scene:
        rays[:, 0:3] = rays[:, 0:3]*0.3 + np.array([0, 1, 0], np.float32)
    # axis-parallel and degenerate directions exercise the slab test's 1/0 handling
    rays[:64, 4:7] = np.eye(3, dtype=np.float32)[np.arange(64) % 3]
    g, o = gpu_ctx.trace_closest(rays), orc.trace_closest(rays)
    assert np.array_equal(g.view(np.uint32), o.view(np.uint32))
    hit = o[:, 3].view(np.uint32) != 0xFFFFFFFF
    assert 0.2 < hit.mean() < 1.0
    assert np.array_equal(gpu_ctx.trace_occluded(rays), orc.trace_occluded(rays))


def test_edge_and_vertex_rays_match_the_definition(ta, oracle_py, gpu_ctx):
    """The nearest hit is the lexicographic minimum of (t, triangle id) over ALL triangles whose test passes
    (include/tgpu_bounds.h); the oracle in brute-force mode evaluates exactly that.  Rays aimed at shared edges and vertices
    are where a walk that prunes with unpadded boxes loses a triangle (the three round-1 full-size mismatches)."""
    hs = ta.HostScene(scene_path("materialtest/c2.json"), resolution=(32, 32))
    gpu_ctx.upload(hs, ta.Shard(32, 32))
    orc = oracle_py.Oracle(hs, **oracle_py.GPU)
    orc.set_brute_force(True)
    rays = edge_and_vertex_rays(hs, 24000, 5)
    g, o = gpu_ctx.trace_closest(rays), orc.trace_closest(rays)
    bad = (g.view(np.uint32) != o.view(np.uint32)).any(axis=1)
    assert not bad.any(), f"{int(bad.sum())} of {len(rays)} edge/vertex rays differ from the brute-force definition"
    assert (o[:, 3].view(np.uint32) != 0xFFFFFFFF).mean() > 0.9
    # any-hit with the far end just before / just behind the targeted edge
    rays[:, 7] = o[:, 0]*np.where(np.arange(len(rays)) % 2 == 0, np.float32(1.0 - 1e-6), np.float32(1.0 + 1e-6))
    rays[~np.isfinite(rays[:, 7]), 7] = 1.0
    assert np.array_equal(gpu_ctx.trace_occluded(rays), orc.trace_occluded(rays))
    # the mesh masters of an "instances" scene: local rays, their own BVHs
    hs = ta.HostScene(scene_path("instances/inst_mesh.json"), resolution=(32, 32))
    gpu_ctx.upload(hs, ta.Shard(32, 32))
    orc = oracle_py.Oracle(hs, **oracle_py.GPU)
    orc.set_brute_force(True)
    rng = np.random.default_rng(9)
    n = 20000
    org = rng.uniform(-4, 4, (n, 3)).astype(np.float32)
    org[:, 1] = np.abs(org[:, 1]) + 0.2
    d = rng.uniform(-2, 2, (n, 3)).astype(np.float32) - org
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.concatenate([org, np.full((n, 1), 1e-4, np.float32), d.astype(np.float32), np.full((n, 1), np.inf, np.float32)], axis=1)
    assert np.array_equal(gpu_ctx.trace_closest(rays).view(np.uint32), orc.trace_closest(rays).view(np.uint32))


def test_device_side_bvh_build(ta, oracle_py, monkeypatch, tmp_path):
    """tungsten_amd/csrc/bvh_gpu.h: Morton sort + Karras hierarchy + bottom-up boxes + breadth-first collapse to the quantised BVH4, all on the device.
    The nearest-hit contract (min (t, id) over every triangle whose test passes; conservative boxes) makes results independent of the tree, so
      1. the benchmark scene forced through the device builder gives the oracle's image and the brute-force definition's hits on rays aimed at
         edges and vertices, and
      2. a flat soup above the default threshold (1.25 M triangles: 125 baked spheres of config 4) takes the device path by itself and renders the
         image the host-built tree renders, bit for bit."""
    import subprocess
    import sys
    monkeypatch.setenv("TGPU_GPU_BVH", "1")
    hs = ta.HostScene(scene_path("materialtest/c2.json"), resolution=(96, 54))
    shard = ta.Shard(hs.res_x, hs.res_y)
    ctx = ta.Context(0)
    try:
        ctx.upload(hs, shard)
        st = ctx.stats()
        assert st["bvh_on_device"] == 1 and st["bvh_build_ms"] > 0 and st["upload_ms"] >= st["bvh_build_ms"]
        ctx.render(16, spp_step=8)
        ref = oracle_py.Oracle(hs, **oracle_py.GPU).render_shard(shard, 16, spp_step=8)
        assert_bit_exact(ctx.download(), ref, "c2 through the device-built tree")
        orc = oracle_py.Oracle(hs, **oracle_py.GPU)
        orc.set_brute_force(True)
        rays = edge_and_vertex_rays(hs, 12000, 5)
        g, o = ctx.trace_closest(rays), orc.trace_closest(rays)
        assert np.array_equal(g.view(np.uint32), o.view(np.uint32))
        rays[:, 7] = o[:, 0]*np.where(np.arange(len(rays)) % 2 == 0, np.float32(1.0 - 1e-6), np.float32(1.0 + 1e-6))
        rays[~np.isfinite(rays[:, 7]), 7] = 1.0
        g_occ, o_occ = ctx.trace_occluded(rays), orc.trace_occluded(rays)
        assert np.array_equal(g_occ, o_occ), f"any-hit: {int((g_occ & ~o_occ & 1).sum())} rays blocked only on the GPU, {int((~g_occ & o_occ & 1).sum())} only in the definition"
        # the same through the wavefront kernels only (per-stage traversal kernels instead of the whole-path tail kernel)
        monkeypatch.setenv("TGPU_TAIL_MAX", "0")
        ctx.upload(hs, shard, 4096)
        ctx.render(16, spp_step=8)
        assert_bit_exact(ctx.download(), ref, "c2 through the device-built tree, wavefront kernels")
        monkeypatch.delenv("TGPU_TAIL_MAX")
        # a soup above the threshold: no environment variable needed
        monkeypatch.delenv("TGPU_GPU_BVH")
        subprocess.run([sys.executable, scene_path("instances/make_c4.py"), str(tmp_path), "5", "160", "90", "--flat"], check=True, stdout=subprocess.DEVNULL)
        flat = ta.HostScene(str(tmp_path / "c4_flat.json"))
        assert flat.desc.contents.n_tris == 1250000
        fshard = ta.Shard(flat.res_x, flat.res_y)
        ctx.upload(flat, fshard)
        st = ctx.stats()
        assert st["bvh_on_device"] == 1
        device_ms = st["bvh_build_ms"]
        ctx.render(8, spp_step=8)
        device_image = ctx.download()
        monkeypatch.setenv("TGPU_GPU_BVH", "0")
        ctx.upload(flat, fshard)
        st = ctx.stats()
        assert st["bvh_on_device"] == 0
        ctx.render(8, spp_step=8)
        assert_bit_exact(device_image, ctx.download(), "1.25 M-triangle soup: device-built against host-built tree")
        host_ms = st["bvh_build_ms"]
        assert device_ms < host_ms, f"device build {device_ms:.0f} ms, host build {host_ms:.0f} ms"
        print(f"BVH build of 1.25 M triangles: device {device_ms:.1f} ms, host {host_ms:.0f} ms")
    finally:
        ctx.close()


def test_portable_math_is_bit_identical_on_device(ta, oracle_py, gpu_ctx):
    L = oracle_py.lib()
    L.oracle_eval_math.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    rng = np.random.default_rng(11)
    n = 1 << 20
    inputs = {
        0: (rng.uniform(-7, 7, n), None), 1: (rng.uniform(-7, 7, n), None), 2: (rng.uniform(-90, 90, n), None),
        3: (np.abs(rng.normal(0, 1, n))**4 + 1e-38, None), 4: (rng.uniform(-1, 1, n), rng.uniform(-1, 1, n)),
        5: (rng.uniform(-1.01, 1.01, n), None), 6: (rng.normal(0, 10, n), rng.normal(0, 1e-3, n)),
        7: (np.abs(rng.normal(0, 100, n)), None),
    }
    for kind, (a, b) in inputs.items():
        a = a.astype(np.float32)
        b = (b if b is not None else np.zeros(n)).astype(np.float32)
        host = np.zeros(n, np.float32)
        L.oracle_eval_math(kind, a.ctypes.data, b.ctypes.data, n, host.ctypes.data)
        dev = gpu_ctx.eval_math(kind, a, b)
        assert np.array_equal(host.view(np.uint32), dev.view(np.uint32)), f"math kind {kind} differs between host and gfx950"


def test_full_size_pass_properties(ta, oracle_py, gpu_ctx):
    """BASELINE config 2 at full size (1280x720, one 16-spp pass): size-independent properties plus a
    bit-exact check of a band of tiles against the oracle."""
    hs = ta.HostScene(scene_path("materialtest/c2.json"))
    assert (hs.res_x, hs.res_y) == (1280, 720)
    shard = ta.Shard(hs.res_x, hs.res_y)
    gpu_ctx.upload(hs, shard)
    gpu_ctx.reset_stats()
    gpu_ctx.render(16, spp_step=16)
    rgb, cnt = gpu_ctx.download()
    st = gpu_ctx.stats()
    assert st["samples"] == 1280*720*16
    assert (cnt == 16).all() and np.isfinite(rgb).all() and (rgb >= 0).all()
    assert 4.0 < (st["extend_rays"] + st["shadow_rays"])/st["samples"] < 5.0   # 4.35 casts/sample
    assert 1.2 < st["surface_events"]/st["samples"] < 1.5                       # 1.32 surface events/sample
    # idempotence: a fresh upload renders the same bits
    gpu_ctx.upload(hs, shard)
    gpu_ctx.render(16, spp_step=16)
    rgb2, _ = gpu_ctx.download()
    assert np.array_equal(rgb, rgb2)
    # a horizontal band of tiles (rows 320..351) against the oracle
    img = shard.scatter(rgb)
    band_px = np.arange(320*1280, 352*1280, dtype=np.uint32)
    sel = np.isin(shard.pixel_index, band_px)
    idx = np.nonzero(sel)[0]
    sub = ta.Shard(hs.res_x, hs.res_y)  # same seeds; render only the band on the CPU through a masked shard
    d = sub.desc.contents
    import ctypes
    from tungsten_amd import _abi
    pix = np.ascontiguousarray(shard.pixel_index[idx])
    seeds = np.ascontiguousarray(np.ctypeslib.as_array(d.tile_seed, shape=(sub.n_pixels,))[idx])
    masked = _abi.ShardDesc(len(pix), pix.ctypes.data_as(_abi.pu32), seeds.ctypes.data_as(_abi.pu32))

    class _S:
        n_pixels = len(pix)
        desc = ctypes.pointer(masked)
    ref_rgb, _ = oracle_py.Oracle(hs, **oracle_py.GPU).render_shard(_S, 16, spp_step=16)
    assert np.array_equal(ref_rgb, rgb[idx])
    assert img.shape == (720, 1280, 3)


def test_scheduling_choices_do_not_change_the_image(ta, gpu_ctx, monkeypatch):
    """How rays are scheduled is a performance choice only: one ray per lane, a fixed queue of 4 / 8 per lane or persistent lanes (k_extend /
    k_extend_q / k_extend_p), the BVH's memory layout, NEE rays traced and resolved per vertex (k_shadow) or as a ray queue with the resolve in the retire kernel
    (k_shadow_q), the top BVH levels staged in LDS or not, the BVH leaf limit, the pool size and the LDS share of the traversal stack must all give the bits of
    the default configuration (640x360, 8 spp of config 2)."""
    hs = ta.HostScene(scene_path("materialtest/c2.json"), resolution=(640, 360))
    shard = ta.Shard(hs.res_x, hs.res_y)

    def render(env, max_paths=0):
        for k in ("TGPU_EXTEND_P", "TGPU_EXTEND_Q", "TGPU_SHADOW_Q", "TGPU_BVH_LEAF", "TGPU_STACK_LDS", "TGPU_TAIL_MAX", "TGPU_LDS_NODELETS", "TGPU_BVH_BFS", "TGPU_BVH_ALIGN"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        gpu_ctx.upload(hs, shard, max_paths)
        gpu_ctx.render(8, spp_step=4)
        return gpu_ctx.download()[0]

    base = render({})
    for env, pool in [({"TGPU_EXTEND_Q": "0"}, 0), ({"TGPU_EXTEND_P": "0"}, 0), ({"TGPU_EXTEND_P": "0", "TGPU_EXTEND_Q": "8"}, 0), ({"TGPU_BVH_LEAF": "4"}, 0),
                      ({"TGPU_BVH_BFS": "1", "TGPU_BVH_ALIGN": "1"}, 0), ({"TGPU_TAIL_MAX": "0"}, 3000),
                      ({"TGPU_SHADOW_Q": "0"}, 0), ({"TGPU_SHADOW_Q": "8", "TGPU_TAIL_MAX": "0"}, 30000),
                      ({"TGPU_LDS_NODELETS": "3"}, 0), ({"TGPU_LDS_NODELETS": "2", "TGPU_STACK_LDS": "12", "TGPU_TAIL_MAX": "0"}, 20000),
                      ({"TGPU_BVH_LEAF": "1", "TGPU_EXTEND_Q": "8"}, 50000), ({"TGPU_STACK_LDS": "3", "TGPU_TAIL_MAX": "0"}, 4096)]:
        assert np.array_equal(render(env, pool), base), f"{env} pool {pool} changed the image"
    render({})  # leave the shared context in its default configuration


def test_full_size_image_bit_exact(ta, oracle_py, gpu_ctx):
    """BASELINE config 2 at its full size, every pixel: 1280x720, 64 spp in 4 passes (the 256-spp render of the benchmark
    was checked the same way once: 0 of 921600 pixels differ, GPU 0.29 s, oracle 50 s -- DESIGN.md section 3)."""
    hs = ta.HostScene(scene_path("materialtest/c2.json"))
    shard = ta.Shard(hs.res_x, hs.res_y)
    gpu_ctx.upload(hs, shard)
    gpu_ctx.render(64, spp_step=16)
    got = gpu_ctx.download()
    ref = oracle_py.Oracle(hs, **oracle_py.GPU).render_shard(shard, 64, spp_step=16)
    assert_bit_exact(got, ref, "config 2, 1280x720x64spp")


@pytest.mark.parametrize("config", ["c5", "c4"])
def test_full_size_other_configs_bit_exact(ta, oracle_py, gpu_ctx, tmp_path, config):
    """Configs 5 (medium inside rough glass, 32 spp) and 4 (10 M instanced triangles, 16 spp) at 1280x720, every pixel."""
    import subprocess
    import sys
    if config == "c5":
        path, spp = scene_path("materialtest/c5.json"), 32
    else:
        subprocess.run([sys.executable, scene_path("instances/make_c4.py"), str(tmp_path)], check=True, stdout=subprocess.DEVNULL)
        path, spp = str(tmp_path / "c4.json"), 16
    hs = ta.HostScene(path)
    shard = ta.Shard(hs.res_x, hs.res_y)
    gpu_ctx.upload(hs, shard)
    gpu_ctx.render(spp, spp_step=16)
    got = gpu_ctx.download()
    ref = oracle_py.Oracle(hs, **oracle_py.GPU).render_shard(shard, spp, spp_step=16)
    assert_bit_exact(got, ref, f"{config} at full size")


def test_full_size_config_3_bit_exact(ta, oracle_py, gpu_ctx):
    """BASELINE config 3 (the Cornell box with the Sobol' sampler, NEE towards its quad light) at its full 1000x563, every pixel, 32 spp."""
    hs = ta.HostScene(scene_path("cornell-box/scene_sobol.json"))
    assert (hs.res_x, hs.res_y) == (1000, 563)
    shard = ta.Shard(hs.res_x, hs.res_y)
    gpu_ctx.upload(hs, shard)
    gpu_ctx.render(32, spp_step=16)
    got = gpu_ctx.download()
    ref = oracle_py.Oracle(hs, **oracle_py.GPU).render_shard(shard, 32, spp_step=16)
    assert_bit_exact(got, ref, "config 3, 1000x563x32spp")


@pytest.mark.parametrize("scene,spp", [("materialtest/c2.json", 256), ("cornell-box/scene_sobol.json", 1024)])
def test_baseline_configs_at_their_full_size_and_sample_count(ta, oracle_py, gpu_ctx, scene, spp):
    """BASELINE.json configs[1] and configs[2] exactly as they are quoted -- 1280x720 at 256 spp, 1000x563 at 1024 spp, 16 spp per pass, the pool kept
    full by regeneration as in bench.py -- against the oracle in every pixel: 236 M and 576 M samples, sample indices up to 1023."""
    hs = ta.HostScene(scene_path(scene))
    shard = ta.Shard(hs.res_x, hs.res_y)
    gpu_ctx.upload(hs, shard)
    gpu_ctx.render(spp, spp_step=16)
    got = gpu_ctx.download()
    ref = oracle_py.Oracle(hs, **oracle_py.GPU).render_shard(shard, spp, spp_step=16)
    assert_bit_exact(got, ref, f"{scene} at full size, {spp} spp")


@pytest.mark.parametrize("scene,res,begin,spp,step", [
    ("materialtest/c2.json", (128, 72), 256, 32, 16),            # configs are quoted at 256..1024 spp: the second Sobol' block and beyond
    ("materialtest/c2.json", (64, 36), 240, 48, 16),             # a render that crosses index 256 between two passes
    ("materialtest/c2.json", (64, 36), 250, 12, 12),             # ... and inside one pass
    ("cornell-box/scene_sobol.json", (96, 54), 1008, 16, 16),    # config 3's last pass
    ("materialtest/c5.json", (64, 36), 1000, 24, 8),             # config 5's last passes (media, supplemental PCG per sample index)
    ("instances/inst_mesh.json", (64, 36), 496, 16, 16),         # config 4's last pass
    ("non-exponential/scene.json", (64, 64), 65536, 8, 8),       # far beyond any block
])
def test_sample_indices_past_256(ta, oracle_py, gpu_ctx, scene, res, begin, spp, step):
    """SobolPathSampler::startPath (SobolPathSampler.hpp:47-52) splits the sample index: idx = (index & ~0xFF) | ((index + scramble) & 0xFF).
    Every other GPU test stays inside the first block of 256; these start a render at a later index (what a resumed render does) and compare with
    the oracle started at the same index -- itself pinned past 256 by c2_32x18_320spp.pfm."""
    hs = ta.HostScene(scene_path(scene), resolution=res)
    shard = ta.Shard(hs.res_x, hs.res_y)
    gpu_ctx.upload(hs, shard)
    gpu_ctx.render(spp, spp_step=step, spp_begin=begin)
    got = gpu_ctx.download()
    ref = oracle_py.Oracle(hs, **oracle_py.GPU).render_shard(shard, spp, spp_step=step, spp_begin=begin)
    assert_bit_exact(got, ref, f"{scene} samples [{begin}, {begin + spp})")


def test_a_320_spp_render_matches_the_oracle(ta, oracle_py, gpu_ctx):
    """One whole render across the block boundary, pool kept full by regeneration the way the benchmark runs: 64x36, 320 spp in 20 passes."""
    hs = ta.HostScene(scene_path("materialtest/c2.json"), resolution=(64, 36))
    shard = ta.Shard(hs.res_x, hs.res_y)
    gpu_ctx.upload(hs, shard)
    gpu_ctx.render(320, spp_step=16)
    got = gpu_ctx.download()
    ref = oracle_py.Oracle(hs, **oracle_py.GPU).render_shard(shard, 320, spp_step=16)
    assert_bit_exact(got, ref, "c2 64x36x320spp")


def test_upload_knobs_apply_to_one_upload_only(ta, gpu_ctx, monkeypatch):
    """The TGPU_* A/B knobs are read by tgpu_upload_scene and hold for that upload.  They used to stick to the context: one upload with TGPU_TAIL_MAX=0
    (several tests above do that on the shared context) left every later render of the session without its tail kernel."""
    hs = ta.HostScene(scene_path("materialtest/c2.json"), resolution=(64, 36))
    shard = ta.Shard(hs.res_x, hs.res_y)
    monkeypatch.setenv("TGPU_TAIL_MAX", "0")
    gpu_ctx.upload(hs, shard)
    gpu_ctx.reset_stats()
    gpu_ctx.render(8, spp_step=8)
    without = gpu_ctx.download()[0]
    assert gpu_ctx.stats()["tail_launches"] == 0
    monkeypatch.delenv("TGPU_TAIL_MAX")
    gpu_ctx.upload(hs, shard)
    gpu_ctx.reset_stats()
    gpu_ctx.render(8, spp_step=8)
    assert gpu_ctx.stats()["tail_launches"] >= 1
    assert np.array_equal(gpu_ctx.download()[0].view(np.uint32), without.view(np.uint32))


def test_more_passes_in_flight_than_the_result_ring_has_slices(ta, oracle_py, gpu_ctx):
    """80 passes of 2 spp enqueued at once on a small image: more jobs than the result ring has slices (32).  The pool runs out of admitted samples while
    the ring is still full; the core must then finish the jobs in flight (tail kernel) and go on -- it used to drain the pool one bounce per iteration
    until the slowest path of the oldest job had ended (116 instead of 52 ms for a 20-pass request on an 8-rank shard when the ring had 16 slices).
    The image is the oracle's either way."""
    hs = ta.HostScene(scene_path("materialtest/c2.json"), resolution=(64, 36))
    shard = ta.Shard(hs.res_x, hs.res_y)
    gpu_ctx.upload(hs, shard)
    gpu_ctx.reset_stats()
    gpu_ctx.render(160, spp_step=2)
    st = gpu_ctx.stats()
    got = gpu_ctx.download()
    ref = oracle_py.Oracle(hs, **oracle_py.GPU).render_shard(shard, 160, spp_step=2)
    assert_bit_exact(got, ref, "c2 64x36, 80 passes of 2 spp")
    assert st["tail_launches"] >= 2, st          # the ring filled up at least once before the end
    assert st["extend_launches"] < 80, st        # ... and nobody waited a path's length in iterations per ring of jobs


def test_output_buffers_and_adaptive_passes_past_256(ta, oracle_py, gpu_ctx):
    """The auxiliary buffers and an adaptive pass at sample indices >= 256 (per-tile sample offsets past the first Sobol' block)."""
    hs = ta.HostScene(scene_path("materialtest/materialtest.json"), resolution=(64, 36), output_buffers=["depth", ("normal", True, False), "albedo", ("visibility", False, True)])
    assert hs.adaptive
    shard = ta.Shard(hs.res_x, hs.res_y)
    var_w, var_h = (hs.res_x + 3)//4, (hs.res_y + 3)//4
    idx = np.full(var_w*var_h, 250, np.uint32) + (np.arange(var_w*var_h) % 3).astype(np.uint32)*7   # tiles at 250, 257, 264
    counts = (2 + (np.arange(var_w*var_h) % 9)).astype(np.uint32)                                  # 2..10 samples: some tiles cross 256
    gpu_ctx.upload(hs, shard)
    gpu_ctx.render_pass_adaptive(idx, counts)
    gpu_ctx.render_pass_adaptive(idx + counts, counts[::-1].copy())
    gpu_ctx.wait()
    got = gpu_ctx.download()
    o = oracle_py.Oracle(hs, **oracle_py.GPU)
    ref_bufs = o.bind_output_buffers(shard.n_pixels)
    o_rgb = np.zeros((shard.n_pixels, 3), np.float32)
    o_cnt = np.zeros(shard.n_pixels, np.uint32)
    o_rec = np.zeros(var_w*var_h, ta.RECORD_DTYPE)
    o.render_shard_adaptive(shard, idx, counts, o_rgb, o_cnt, o_rec)
    o.render_shard_adaptive(shard, idx + counts, counts[::-1].copy(), o_rgb, o_cnt, o_rec)
    assert_bit_exact(got, (o_rgb, o_cnt), "adaptive colour past 256")
    _assert_outputs_bit_exact(gpu_ctx, ref_bufs, "adaptive past 256")
    assert np.array_equal(gpu_ctx.download_records().view(np.uint32), o_rec.view(np.uint32))


def test_image_matches_reference_golden_within_tolerance(ta, gpu_ctx):
    """GPU image vs the compiled reference's golden PFM: identical Sobol sequences, but the supplemental
    stream is path-local and t,u,v use IEEE division -> agreement is statistical, not bitwise."""
    hs = ta.HostScene(scene_path("materialtest/c2.json"), resolution=(128, 72))
    shard = ta.Shard(hs.res_x, hs.res_y)
    rgb, _ = render_gpu(ta, gpu_ctx, hs, shard, 48, 16)
    img = shard.scatter(rgb)
    ref = ta.load_pfm(os.path.join(GOLDEN, "c2_128x72_48spp.pfm"))
    assert abs(img.mean()/ref.mean() - 1.0) < 0.02
    pool = lambda a: a.reshape(18, 4, 32, 4, 3).mean(axis=(1, 3))
    assert np.abs(pool(img) - pool(ref)).mean()/pool(ref).mean() < 0.05


def test_errors_are_reported_not_swallowed(ta, gpu_ctx):
    hs = ta.HostScene(scene_path("materialtest/c2.json"), resolution=(16, 16))
    fresh = ta.Context(0)
    with pytest.raises(ta.CoreError, match="before tgpu_upload_scene"):
        fresh.render_pass(0, 4)
    empty = ta.Shard(16, 16, 0xBA5EBA11, 1, 2)  # second rank of a one-tile image: no pixels
    assert empty.n_pixels == 0
    with pytest.raises(ta.CoreError, match="empty shard"):
        fresh.upload(hs, empty)
    fresh.close()


def test_cpp_host_integrator_cli(ta, oracle_py, tmp_path):
    """The reference-shaped C++ host (Integrator::prepareForRender/startRender/waitForCompletion/saveOutputs
    driven by the StandaloneRenderer loop) produces the same image as the ctypes harness path."""
    import json
    import subprocess
    from conftest import REPO, ASSETS
    d = json.load(open(scene_path("materialtest/c2.json")))
    d["camera"]["resolution"] = [96, 54]
    d["renderer"].update({"spp": 20, "spp_step": 8, "output_file": "cli.png", "hdr_output_file": "cli.pfm"})
    for p in d["primitives"]:
        for key in ("file", "emission"):
            if isinstance(p.get(key), str):
                p[key] = os.path.join(ASSETS, "materialtest", p[key])
    scene_file = tmp_path / "cli_scene.json"
    scene_file.write_text(json.dumps(d))
    exe = os.path.join(REPO, "tungsten_amd", "tungsten_amd_cli")
    r = subprocess.run([exe, "-d", str(tmp_path), str(scene_file)], capture_output=True, text=True,
                       env=dict(os.environ, TUNGSTEN_AMD_ASSETS=ASSETS))
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Completed 20/20 spp" in r.stdout
    img = ta.load_pfm(tmp_path / "cli.pfm")
    hs = ta.HostScene(str(scene_file))
    shard = ta.Shard(hs.res_x, hs.res_y)
    ref, _ = oracle_py.Oracle(hs, **oracle_py.GPU).render_shard(shard, 20, spp_step=8)
    assert np.array_equal(img, shard.scatter(ref))
    assert os.path.getsize(tmp_path / "cli.png") > 1000


def test_cli_resume_file_continues_a_render_bit_for_bit(ta, oracle_py, tmp_path):
    """Integrator::saveRenderResumeData / resumeRender on the reference's file layout (Integrator.cpp:108-162): 16 spp, then a second
    run that resumes to 32 spp, equals a straight 32-spp render in every bit; -r (restart) ignores the file; --timeout stops after a
    pass; -i resolves resources; a state file with another scene hash is refused (another scene's file in the same directory; the reference's own
    file tests/golden/next/RenderState_*.dat) unless --accept-foreign-resume-file asks for it."""
    import json
    import shutil
    import struct
    import subprocess
    from conftest import REPO, ASSETS
    d = json.load(open(scene_path("materialtest/c2.json")))
    d["camera"]["resolution"] = [64, 36]
    d["renderer"].update({"spp": 16, "spp_step": 16, "output_file": "r.png", "hdr_output_file": "r.pfm", "enable_resume_render": True,
                          "resume_render_file": "RenderState.dat"})
    exe = os.path.join(REPO, "tungsten_amd", "tungsten_amd_cli")
    env = dict(os.environ, TUNGSTEN_AMD_ASSETS=ASSETS)
    indir = os.path.join(ASSETS, "materialtest")  # relative "file" entries resolve against -i

    def run(spp, *flags, outdir):
        d["renderer"]["spp"] = spp
        scene_file = tmp_path / "resume_scene.json"
        scene_file.write_text(json.dumps(d))
        os.makedirs(outdir, exist_ok=True)
        r = subprocess.run([exe, "-i", indir, "-d", str(outdir), *flags, str(scene_file)], capture_output=True, text=True, env=env)
        assert r.returncode == 0, r.stdout + r.stderr
        return r.stdout, scene_file

    a = tmp_path / "a"
    out, _ = run(16, outdir=a)
    assert "Resume unsuccessful" in out and "Completed 16/16 spp" in out
    state = (a / "RenderState.dat").read_bytes()
    header = state[:state.index(b"\0")]
    assert json.loads(header) == {"current_spp": 16, "adaptive_sampling": False, "stratified_sampler": True}
    assert header == b'{\n    "current_spp": 16,\n    "adaptive_sampling": false,\n    "stratified_sampler": true\n}'   # rapidjson PrettyWriter's layout
    n_px, n_vt, n_tiles = 64*36, 16*9, 4*3
    assert len(state) == len(header) + 1 + 8 + n_px*16 + n_vt*24 + n_tiles*12                                           # the reference's sizes
    out, _ = run(32, outdir=a)                                  # second run: resumes at 16, renders one more pass
    assert "Resume successful" in out and "Completed 32/32 spp" in out and "Completed 16/32 spp" not in out
    b = tmp_path / "b"
    out, _ = run(32, "-r", outdir=b)                            # straight render
    assert "Trying to resume" not in out
    resumed, straight = ta.load_pfm(a / "r.pfm"), ta.load_pfm(b / "r.pfm")
    assert np.array_equal(resumed, straight)
    hs = ta.HostScene(scene_path("materialtest/c2.json"), resolution=(64, 36))   # the same scene, resources next to the file
    shard = ta.Shard(hs.res_x, hs.res_y)
    ref, _ = oracle_py.Oracle(hs, **oracle_py.GPU).render_shard(shard, 32, spp_step=16)
    assert np.array_equal(straight, shard.scatter(ref))
    # --timeout: the pass loop stops after the first pass that ends past the limit (Shared.hpp:295-296)
    c = tmp_path / "c"
    out, _ = run(64, "-r", "--timeout", "1ms", outdir=c)
    assert "Completed 16/64 spp" in out and "Completed 32/64 spp" not in out
    # scene B must not continue from scene A's state file (same directory, same default file name, same resolution and sampler flags):
    # Integrator.cpp:146-148 refuses the hash, and so does this host -- B renders from 0 spp and its image is B's
    d["bsdfs"][0]["albedo"] = 0.25 if d["bsdfs"][0].get("albedo") != 0.25 else 0.5
    out, _ = run(32, outdir=a)
    assert "Resume unsuccessful" in out and "Completed 16/32 spp" in out and "Completed 32/32 spp" in out
    assert not np.array_equal(ta.load_pfm(a / "r.pfm"), straight)
    d = json.load(open(scene_path("materialtest/c2.json")))
    d["camera"]["resolution"] = [64, 36]
    d["renderer"].update({"spp": 16, "spp_step": 16, "output_file": "r.png", "hdr_output_file": "r.pfm", "enable_resume_render": True,
                          "resume_render_file": "RenderState.dat"})
    # the reference's own file (c2 at 64x36, 32 spp): same layout, but its scene hash is the reference's re-serialisation of the scene
    e = tmp_path / "e"
    os.makedirs(e)
    shutil.copy(os.path.join(GOLDEN, "next", "RenderState_c2_64x36_32spp_step16.dat"), e / "RenderState.dat")
    os.makedirs(tmp_path / "e_refused")
    shutil.copy(os.path.join(GOLDEN, "next", "RenderState_c2_64x36_32spp_step16.dat"), tmp_path / "e_refused" / "RenderState.dat")
    out, _ = run(32, outdir=tmp_path / "e_refused")
    assert "Resume unsuccessful" in out                          # refused by default
    out, _ = run(32, "--accept-foreign-resume-file", outdir=e)
    assert "Resume successful" in out and "Warning: the resume file carries another scene hash" in out
    assert "Completed" not in out                                # 32 of 32 spp are in the file: nothing left to render
    got = ta.load_pfm(e / "r.pfm")
    want = ta.load_pfm(os.path.join(GOLDEN, "next", "RenderState_c2_64x36_32spp_step16.pfm"))
    assert np.array_equal(got, want)                             # the colour buffer of the file, written back out unchanged


def test_abort_mid_render_then_render_again(ta, oracle_py, gpu_ctx):
    """Integrator::abortRender (PathTraceIntegrator.cpp:249-256) from a second thread while passes are queued: tgpu_wait reports
    TGPU_ERR_ABORTED, and the same context then renders correctly again."""
    import threading
    import time
    hs = ta.HostScene(scene_path("materialtest/c2.json"), resolution=(640, 360))
    shard = ta.Shard(hs.res_x, hs.res_y)
    gpu_ctx.upload(hs, shard)
    for i in range(64):                                          # 1024 spp queued: several hundred ms of work
        gpu_ctx.render_pass(16*i, 16)
    killer = threading.Timer(0.03, lambda: ta.core_lib().tgpu_abort(gpu_ctx._h))
    killer.start()
    t0 = time.perf_counter()
    with pytest.raises(ta.CoreError, match="aborted"):
        gpu_ctx.wait()
    killer.join()
    assert time.perf_counter() - t0 < 5.0
    gpu_ctx.upload(hs, shard)                                    # fresh framebuffer on the same context
    gpu_ctx.render(8, spp_step=4)
    ref = oracle_py.Oracle(hs, **oracle_py.GPU).render_shard(shard, 8, spp_step=4)
    assert_bit_exact(gpu_ctx.download(), ref, "render after an abort")


def _records_image(rec, hs):
    """RECORD_DTYPE table -> (varH, varW, 3) float32 like Oracle.render_reference's third result."""
    var_w, var_h = (hs.res_x + 3)//4, (hs.res_y + 3)//4
    return np.stack([rec["sample_count"].astype(np.float32), rec["mean"], rec["running_variance"]], axis=1).reshape(var_h, var_w, 3)


def test_adaptive_sampling_bit_exact(ta, oracle_py, gpu_ctx):
    """settings.adaptive_sampling: SampleRecord statistics accumulated on the device in renderTile's order,
    the host's generateWork in between, per-variance-tile sample counts in the next pass -- against the
    oracle's restatement of the whole reference loop (itself pinned by c1_adaptive_128x72_64spp.pfm)."""
    from tungsten_amd.passes import render_passes
    hs = ta.HostScene(scene_path("materialtest/materialtest.json"), resolution=(128, 72))
    assert hs.adaptive
    shard = ta.Shard(hs.res_x, hs.res_y)
    gpu_ctx.upload(hs, shard)
    render_passes(gpu_ctx, hs.res_x, hs.res_y, 64, 16, adaptive=True)
    rgb, cnt = gpu_ctx.download()
    rec = gpu_ctx.download_records()
    ref_img, ref_cnt, ref_rec = oracle_py.Oracle(hs, **oracle_py.GPU).render_reference(64, spp_step=16, adaptive=True)
    assert ref_cnt.min() < 64 < ref_cnt.max()
    assert np.array_equal(shard.scatter(cnt.reshape(-1, 1).repeat(3, 1).astype(np.float32))[..., 0], ref_cnt.astype(np.float32))
    assert np.array_equal(_records_image(rec, hs).view(np.uint32), ref_rec.view(np.uint32))
    assert np.array_equal(shard.scatter(rgb), ref_img)


def test_adaptive_passes_larger_than_a_result_slice_are_split(ta, oracle_py, gpu_ctx):
    """A pass whose samples exceed one slice of the result ring (spp_step 32 > 16 spp per slice; a shard handed three times its
    share) runs as several jobs cut along 16x16 tiles: same bits as the oracle's single loop (round 1 refused such passes)."""
    from tungsten_amd.passes import render_passes
    hs = ta.HostScene(scene_path("materialtest/materialtest.json"), resolution=(128, 72))
    shard = ta.Shard(hs.res_x, hs.res_y)
    gpu_ctx.upload(hs, shard)
    render_passes(gpu_ctx, hs.res_x, hs.res_y, 96, 32, adaptive=True)          # uniform 32-spp pass, then two adaptive ones
    rgb, cnt = gpu_ctx.download()
    rec = gpu_ctx.download_records()
    ref_img, ref_cnt, ref_rec = oracle_py.Oracle(hs, **oracle_py.GPU).render_reference(96, spp_step=32, adaptive=True)
    assert np.array_equal(_records_image(rec, hs).view(np.uint32), ref_rec.view(np.uint32))
    assert np.array_equal(shard.scatter(rgb), ref_img)
    # a synthetic, skewed table: a band of variance tiles gets 60 samples per pixel, the rest 1 -> 3x the slice of an even pass
    var_w, var_h = (hs.res_x + 3)//4, (hs.res_y + 3)//4
    idx = np.zeros(var_w*var_h, np.uint32)
    skew = np.ones((var_h, var_w), np.uint32)
    skew[4:14, :] = 60
    gpu_ctx.upload(hs, shard)
    gpu_ctx.render_pass_adaptive(idx, skew.ravel())
    gpu_ctx.wait()
    rgb, cnt = gpu_ctx.download()
    o_rgb = np.zeros((shard.n_pixels, 3), np.float32)
    o_cnt = np.zeros(shard.n_pixels, np.uint32)
    o_rec = np.zeros(var_w*var_h, ta.RECORD_DTYPE)
    oracle_py.Oracle(hs, **oracle_py.GPU).render_shard_adaptive(shard, idx, skew.ravel(), o_rgb, o_cnt, o_rec)
    assert int(cnt.sum()) == int(o_cnt.sum()) > shard.n_pixels*24
    assert_bit_exact((rgb, cnt), (o_rgb, o_cnt), "skewed adaptive pass")
    got_rec = gpu_ctx.download_records()
    assert np.array_equal(got_rec.view(np.uint32), o_rec.view(np.uint32))


def test_adaptive_sampling_ragged_image_and_two_shards(ta, oracle_py, gpu_ctx):
    """100x57: ragged 16x16 tiles AND ragged 4x4 variance tiles; two shards on two contexts with the record
    tables merged on the host (what dist.exchange_records does across processes)."""
    hs = ta.HostScene(scene_path("materialtest/materialtest.json"), resolution=(100, 57))
    shards = [ta.Shard(hs.res_x, hs.res_y, 0xBA5EBA11, r, 2) for r in range(2)]
    ctxs = [gpu_ctx, ta.Context(0)]
    for c, s in zip(ctxs, shards):
        c.upload(hs, s)
    work = ta.AdaptiveScheduler(hs.res_x, hs.res_y)
    current = 0
    while current < 40:
        nxt = min(current + 8, 40)
        if work.needs_statistics(True, current):
            table = ctxs[0].download_records()
            other = ctxs[1].download_records()
            assert not ((table["sample_count"] > 0) & (other["sample_count"] > 0)).any()  # disjoint tiles
            table["sample_count"] += other["sample_count"]
            table["mean"] += other["mean"]
            table["running_variance"] += other["running_variance"]
            work.set_statistics(table)
        job = work.generate(current, nxt, True)
        assert job is not None
        for c in ctxs:
            c.render_pass_adaptive(*job)
        for c in ctxs:
            c.wait()
        current = nxt
    img = None
    for c, s in zip(ctxs, shards):
        img = s.scatter(c.download()[0], img)
    ctxs[1].close()
    ref_img, ref_cnt, _ = oracle_py.Oracle(hs, **oracle_py.GPU).render_reference(40, spp_step=8, adaptive=True)
    assert ref_cnt.min() < 40 < ref_cnt.max()
    assert np.array_equal(img, ref_img)


def test_adaptive_records_upload_and_errors(ta, gpu_ctx):
    hs = ta.HostScene(scene_path("materialtest/materialtest.json"), resolution=(64, 36))
    shard = ta.Shard(hs.res_x, hs.res_y)
    gpu_ctx.upload(hs, shard)
    n = 16*9
    table = np.zeros(n, ta.RECORD_DTYPE)
    table["sample_count"] = np.arange(n)
    table["mean"] = np.linspace(0, 1, n, dtype=np.float32)
    table["running_variance"] = np.linspace(2, 3, n, dtype=np.float32)
    gpu_ctx.upload_records(table)                      # resume: records come back from a checkpoint
    assert np.array_equal(gpu_ctx.download_records(), table)
    with pytest.raises(ValueError):
        gpu_ctx.render_pass_adaptive(np.zeros(3, np.uint32), np.zeros(3, np.uint32))
    huge = np.full(n, 1000, np.uint32)                 # more samples than a result-ring slice holds
    with pytest.raises(ta.CoreError, match="exceeds the result-ring slice"):
        gpu_ctx.render_pass_adaptive(np.zeros(n, np.uint32), huge)
    plain = ta.HostScene(scene_path("materialtest/c2.json"), resolution=(64, 36))
    gpu_ctx.upload(plain, shard)
    with pytest.raises(ta.CoreError, match="adaptive_sampling"):
        gpu_ctx.render_pass_adaptive(np.zeros(n, np.uint32), np.ones(n, np.uint32))
    with pytest.raises(ta.CoreError, match="no sample records"):
        gpu_ctx.download_records()


def test_cpp_host_integrator_cli_adaptive(ta, oracle_py, tmp_path):
    """The stock scene file as it ships (adaptive_sampling on) through the C++ Integrator mirror."""
    import json
    import subprocess
    from conftest import REPO, ASSETS
    d = json.load(open(scene_path("materialtest/materialtest.json")))
    assert d["renderer"]["adaptive_sampling"]
    d["camera"]["resolution"] = [96, 54]
    d["renderer"].update({"spp": 48, "spp_step": 16, "output_file": "cli.png", "hdr_output_file": "cli.pfm"})
    for p in d["primitives"]:
        for key in ("file", "emission"):
            if isinstance(p.get(key), str):
                p[key] = os.path.join(ASSETS, "materialtest", p[key])
    scene_file = tmp_path / "cli_scene.json"
    scene_file.write_text(json.dumps(d))
    exe = os.path.join(REPO, "tungsten_amd", "tungsten_amd_cli")
    r = subprocess.run([exe, "-d", str(tmp_path), str(scene_file)], capture_output=True, text=True,
                       env=dict(os.environ, TUNGSTEN_AMD_ASSETS=ASSETS))
    assert r.returncode == 0, r.stdout + r.stderr
    img = ta.load_pfm(tmp_path / "cli.pfm")
    hs = ta.HostScene(str(scene_file))
    ref, cnt, _ = oracle_py.Oracle(hs, **oracle_py.GPU).render_reference(48, spp_step=16, adaptive=True)
    assert cnt.min() < 48 < cnt.max()
    assert np.array_equal(img, ref)


@pytest.mark.parametrize("tail_max", ["0", None])
@pytest.mark.parametrize("scene", ["materialtest/c2.json", "instances/inst_mesh.json"])
def test_traversal_stack_spill_path(ta, oracle_py, scene, tail_max, monkeypatch):
    """Only the first levels of the traversal stack live in LDS; TGPU_STACK_LDS=2 pushes nearly every entry through
    the global spill area (wavefront kernels and the tail kernel), and the image must not change."""
    monkeypatch.setenv("TGPU_STACK_LDS", "2")
    if tail_max is not None:
        monkeypatch.setenv("TGPU_TAIL_MAX", tail_max)
    hs = ta.HostScene(scene_path(scene), resolution=(96, 54))
    shard = ta.Shard(hs.res_x, hs.res_y)
    ctx = ta.Context(0)
    try:
        ctx.upload(hs, shard, 2048)
        ctx.render(4, spp_step=4)
        got = ctx.download()
    finally:
        ctx.close()
    ref = oracle_py.Oracle(hs, **oracle_py.GPU).render_shard(shard, 4, spp_step=4)
    assert_bit_exact(got, ref, f"{scene} with the stack spilled")


@pytest.mark.parametrize("scene,res,spp", [
    ("materialtest/c5.json", (128, 72), 8),            # volume list + k_volume + class-specialised shade kernels
    ("cornell-box/scene_fog.json", (96, 54), 8),
    ("instances/inst_mesh.json", (96, 54), 8),         # INST variants of extend / shadow
    ("materialtest/c1_noadapt.json", (128, 72), 8),
    ("non-exponential/scene.json", (96, 96), 8),       # FWD variant of the shadow kernel (multi-segment walk)
])
def test_wavefront_kernels_without_the_tail_kernel(ta, oracle_py, scene, res, spp, monkeypatch):
    """Small renders finish inside the tail megakernel; TGPU_TAIL_MAX=0 forces every bounce through the per-stage
    wavefront kernels (extend -> miss | shade<class> | volume -> shadow -> retire/regenerate) instead."""
    monkeypatch.setenv("TGPU_TAIL_MAX", "0")
    hs = ta.HostScene(scene_path(scene), resolution=res)
    shard = ta.Shard(hs.res_x, hs.res_y)
    ctx = ta.Context(0)
    try:
        ctx.upload(hs, shard, 2048)  # a pool much smaller than the job: slots are retired and regenerated many times
        ctx.render(spp, spp_step=spp)
        got = ctx.download()
        st = ctx.stats()
        assert st["tail_launches"] == 0 and st["extend_launches"] > 10
    finally:
        ctx.close()
    ref = oracle_py.Oracle(hs, **oracle_py.GPU).render_shard(shard, spp, spp_step=spp)
    assert_bit_exact(got, ref, f"{scene} wavefront-only")


def test_cli_renders_the_unmodified_non_exponential_scene(ta, oracle_py, tmp_path):
    """The reference's own data/example-scenes/non-exponential/scene.json, byte for byte as it ships (512x512, uniform
    sampler, adaptive sampling on, max_bounces 3), through the C++ Integrator mirror; only --spp is lowered."""
    import subprocess
    from conftest import REPO, ASSETS
    scene_file = scene_path("non-exponential/scene.json")
    exe = os.path.join(REPO, "tungsten_amd", "tungsten_amd_cli")
    r = subprocess.run([exe, "-d", str(tmp_path), "--spp", "64", "-e", "out.pfm", scene_file], capture_output=True, text=True,
                       env=dict(os.environ, TUNGSTEN_AMD_ASSETS=ASSETS))
    assert r.returncode == 0, r.stdout + r.stderr
    img = ta.load_pfm(tmp_path / "out.pfm")
    assert img.shape == (512, 512, 3)
    hs = ta.HostScene(scene_file)
    assert hs.adaptive and not hs.use_sobol
    ref, cnt, _ = oracle_py.Oracle(hs, **oracle_py.GPU).render_reference(64, spp_step=16, adaptive=True)
    assert cnt.min() < 64 < cnt.max()
    assert np.array_equal(img, ref)


# ---- renderer "output_buffers": depth / normal / albedo / visibility (PathTracer.cpp:78-96,133-140) and the colour buffer's optional parts ----
ALL_OUTPUTS = [("color", True, True), ("depth", True, True), ("normal", False, True), ("albedo", True, False), "visibility"]


def _assert_outputs_bit_exact(ctx, ref_bufs, what):
    for name, parts in ref_bufs.items():
        for part, ref in parts.items():
            got = ctx.download_output_buffer(name, part)
            same = got.view(np.uint32) == ref.view(np.uint32)
            assert same.all(), f"{what}: {name}.{part} differs in {int((~same).sum())} of {same.size} values"


@pytest.mark.parametrize("scene,res,spp,step,outputs,tail_max", [
    ("next/aov.json", (96, 54), 8, 4, None, None),                           # the scene's own list: glass (late records), escaping primary rays
    ("next/aov.json", (96, 54), 8, 4, None, "0"),                            # ... through the wavefront kernels (lazy NEE resolve, retire epilogue)
    ("cornell-box/scene_sobol.json", (64, 36), 9, 4, ALL_OUTPUTS, "0"),      # every part of every buffer, odd sample count, quad light
    ("materialtest/c5.json", (64, 36), 8, 8, ALL_OUTPUTS, "0"),              # media: visibility is the light sample's transmittance
    ("non-exponential/scene.json", (64, 64), 8, 8, ALL_OUTPUTS, "0"),        # forward-scattering walls: the shadow walk's transmittance
    ("next/transparency.json", (64, 36), 8, 8, ALL_OUTPUTS, None),           # TransparencyBsdf shows its base's albedo
    ("instances/inst_mesh.json", (64, 36), 8, 4, ["depth", "normal", "albedo", "visibility"], "0"),
])
def test_output_buffers_bit_exact(ta, oracle_py, scene, res, spp, step, outputs, tail_max, monkeypatch):
    if tail_max is not None:
        monkeypatch.setenv("TGPU_TAIL_MAX", tail_max)
    hs = ta.HostScene(scene_path(scene), resolution=res, output_buffers=outputs)
    shard = ta.Shard(hs.res_x, hs.res_y)
    o = oracle_py.Oracle(hs, **oracle_py.GPU)
    ref_bufs = o.bind_output_buffers(shard.n_pixels)
    ref = o.render_shard(shard, spp, spp_step=step)
    assert len(ref_bufs) >= 4 and ref_bufs["depth"]["count"].max() > 0
    ctx = ta.Context(0)
    try:
        got = render_gpu(ta, ctx, hs, shard, spp, step, 4096 if tail_max == "0" else 0)
        # tgpu_download returns _bufferA of the colour buffer: the mean of the even samples when the buffer is split in two
        assert_bit_exact(got, (ref_bufs["color"]["a"], ref_bufs["color"]["count"]), f"{scene} colour with output buffers")
        if "b" not in ref_bufs["color"]:
            assert_bit_exact(got, ref, f"{scene} colour")
        _assert_outputs_bit_exact(ctx, ref_bufs, scene)
        with pytest.raises(RuntimeError):
            ctx.download_output_buffer("albedo", "variance" if "variance" not in ref_bufs["albedo"] else "b")  # a part nobody asked for
    finally:
        ctx.close()


def test_output_buffers_immediate_and_queued_shadow_kernels_agree(ta, oracle_py, monkeypatch):
    """The visibility buffer is completed where the shadow ray's result becomes known: in the shadow kernel itself (TGPU_SHADOW_Q=0) or
    at the lazy resolve of the queued form."""
    hs = ta.HostScene(scene_path("materialtest/c2.json"), resolution=(96, 54), output_buffers=["depth", "normal", "albedo", ("visibility", True, True)])
    shard = ta.Shard(hs.res_x, hs.res_y)
    o = oracle_py.Oracle(hs, **oracle_py.GPU)
    ref_bufs = o.bind_output_buffers(shard.n_pixels)
    ref = o.render_shard(shard, 8, spp_step=8)
    for q in ("0", "4"):
        monkeypatch.setenv("TGPU_SHADOW_Q", q)
        monkeypatch.setenv("TGPU_TAIL_MAX", "0")
        ctx = ta.Context(0)
        try:
            got = render_gpu(ta, ctx, hs, shard, 8, 8, 8192)
            assert_bit_exact(got, ref, f"c2 colour, shadow queue {q}")
            _assert_outputs_bit_exact(ctx, ref_bufs, f"c2, shadow queue {q}")
        finally:
            ctx.close()


def test_output_buffers_with_adaptive_sampling(ta, oracle_py, gpu_ctx):
    """Adaptive passes hold a pixel's samples contiguously (k_accumulate_records): same buffers as the oracle's adaptive shard pass."""
    hs = ta.HostScene(scene_path("materialtest/materialtest.json"), resolution=(64, 36), output_buffers=["depth", ("normal", True, False), "albedo", "visibility"])
    assert hs.adaptive
    shard = ta.Shard(hs.res_x, hs.res_y)
    var_w, var_h = (hs.res_x + 3)//4, (hs.res_y + 3)//4
    idx = np.zeros(var_w*var_h, np.uint32)
    counts = (1 + (np.arange(var_w*var_h) % 5)).astype(np.uint32)
    gpu_ctx.upload(hs, shard)
    gpu_ctx.render_pass(0, 4)                      # a uniform pass first (records routed through the same kernel)
    gpu_ctx.render_pass_adaptive(idx + 4, counts)
    gpu_ctx.wait()
    got = gpu_ctx.download()
    o = oracle_py.Oracle(hs, **oracle_py.GPU)
    ref_bufs = o.bind_output_buffers(shard.n_pixels)
    o_rgb = np.zeros((shard.n_pixels, 3), np.float32)
    o_cnt = np.zeros(shard.n_pixels, np.uint32)
    o_rec = np.zeros(var_w*var_h, ta.RECORD_DTYPE)
    o.render_shard_adaptive(shard, idx, np.full(var_w*var_h, 4, np.uint32), o_rgb, o_cnt, o_rec)
    o.render_shard_adaptive(shard, idx + 4, counts, o_rgb, o_cnt, o_rec)
    assert_bit_exact(got, (o_rgb, o_cnt), "adaptive colour with output buffers")
    _assert_outputs_bit_exact(gpu_ctx, ref_bufs, "adaptive")


def test_cli_writes_output_buffers_and_resumes_them(ta, oracle_py, tmp_path):
    """The C++ host: `output_buffers` files next to the image (Camera::saveOutputBuffers) and all buffers inside the resume file."""
    import json
    import subprocess
    scene = json.load(open(scene_path("next/aov.json")))
    for prim in scene["primitives"]:
        for key in ("file", "emission"):
            if isinstance(prim.get(key), str) and prim[key].startswith("../"):
                prim[key] = os.path.abspath(os.path.join(os.path.dirname(scene_path("next/aov.json")), prim[key]))
    scene["camera"]["resolution"] = [64, 36]
    scene["renderer"].update(spp=8, spp_step=4, hdr_output_file="img.pfm", output_file="img.png", enable_resume_render=True,
                             resume_render_file="state.dat", checkpoint_interval="0")
    path = tmp_path/"aov.json"
    path.write_text(json.dumps(scene))
    cli = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tungsten_amd", "tungsten_amd_cli")
    r = subprocess.run([cli, "-d", str(tmp_path), str(path)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    hs = ta.HostScene(str(path))
    shard = ta.Shard(hs.res_x, hs.res_y)
    o = oracle_py.Oracle(hs, **oracle_py.GPU)
    bufs = o.bind_output_buffers(shard.n_pixels)
    o.render_shard(shard, 8, spp_step=4)

    def pfm(name, channels):
        with open(tmp_path/name, "rb") as f:
            magic = f.readline().strip()
            w, h = map(int, f.readline().split())
            f.readline()
            assert magic == (b"Pf" if channels == 1 else b"PF")
            return np.frombuffer(f.read(), np.float32).reshape(h, w, channels)[::-1]

    def image_order(values, channels):
        img = np.zeros((hs.res_y*hs.res_x, channels), np.float32)
        img[np.asarray(shard.pixel_index)] = values
        return img.reshape(hs.res_y, hs.res_x, channels)

    assert np.array_equal(pfm("aov_depth.pfm", 1).view(np.uint32), image_order(bufs["depth"]["a"], 1).view(np.uint32))
    assert np.array_equal(pfm("aov_albedo.pfm", 3).view(np.uint32), image_order(bufs["albedo"]["a"], 3).view(np.uint32))
    assert np.array_equal(pfm("aov_visibility.pfm", 1).view(np.uint32), image_order(bufs["visibility"]["a"], 1).view(np.uint32))
    assert np.array_equal(pfm("aov_normalA.pfm", 3).view(np.uint32), image_order(bufs["normal"]["a"], 3).view(np.uint32))
    assert np.array_equal(pfm("aov_normalB.pfm", 3).view(np.uint32), image_order(bufs["normal"]["b"], 3).view(np.uint32))
    var = bufs["depth"]["variance"][:, 0]/(bufs["depth"]["count"]*np.maximum(1, bufs["depth"]["count"].astype(np.int64) - 1)).astype(np.float32)
    with np.errstate(invalid="ignore", divide="ignore"):
        assert np.array_equal(pfm("aov_depthVariance.pfm", 1), image_order(var.astype(np.float32)[:, None], 1), equal_nan=True)
    for name in ("aov_depth.png", "aov_normal.png", "aov_normalA.png", "aov_depthVariance.png"):
        assert (tmp_path/name).stat().st_size > 100
    # the resume file holds every part of every buffer in the reference's order; a resumed render of 8 more spp continues all of them
    size = (tmp_path/"state.dat").stat().st_size
    n = hs.res_x*hs.res_y
    assert size > n*4*(3 + 1 + (1 + 1 + 1) + (3 + 3 + 1) + (3 + 1) + (1 + 1))
    scene["renderer"]["spp"] = 16
    path.write_text(json.dumps(scene))
    r = subprocess.run([cli, "-d", str(tmp_path), str(path)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "Resume successful" in r.stdout, r.stdout + r.stderr
    o.render_shard(shard, 8, spp_step=4, spp_begin=8)
    assert np.array_equal(pfm("aov_depth.pfm", 1).view(np.uint32), image_order(bufs["depth"]["a"], 1).view(np.uint32))
    assert np.array_equal(pfm("aov_normalB.pfm", 3).view(np.uint32), image_order(bufs["normal"]["b"], 3).view(np.uint32))
