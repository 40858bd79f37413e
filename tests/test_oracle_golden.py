"""The oracle (CPU restatement) against float framebuffers rendered by the compiled, unmodified
reference (tests/golden/*.pfm, provenance in tests/golden/make_golden.py).  These pin the oracle."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, scene_path

# name, scene, resolution, spp
CASES = [
    ("c2_128x72_1spp", "materialtest/c2.json", (128, 72), 1),
    ("c2_128x72_16spp", "materialtest/c2.json", (128, 72), 16),
    ("c2_128x72_48spp", "materialtest/c2.json", (128, 72), 48),      # 3 passes: tile PCG state carries over
    ("c2_100x57_4spp", "materialtest/c2.json", (100, 57), 4),        # ragged tiles
    ("c1_128x72_4spp", "materialtest/c1_noadapt.json", (128, 72), 4),  # stock scene: SmoothCoat over Cu
    ("c1_128x72_32spp", "materialtest/c1_noadapt.json", (128, 72), 32),
    ("cornell_96x54_8spp", "cornell-box/scene_sobol.json", (96, 54), 8),  # quads, cubes, quad light
    # "instances" primitives: BVH2 build + ordered traversal + last-hit-wins + the double-transformed hit point
    ("inst_cubes_128x72_8spp", "instances/inst_cubes.json", (128, 72), 8),   # 16 instances (full-sweep SAH only)
    ("inst_many_96x54_4spp", "instances/inst_many.json", (96, 54), 4),       # 294 instances (binned SAH levels)
    # config 5 (a26): rough_dielectric boundary, homogeneous medium inside (distance sampling, volume NEE, fmath exp)
    ("c5_128x72_16spp", "materialtest/c5.json", (128, 72), 16),
    ("c5_absorb_128x72_8spp", "materialtest/c5_absorb.json", (128, 72), 8),  # absorption only + beckmann glass
    ("cornell_fog_96x54_8spp", "cornell-box/scene_fog.json", (96, 54), 8),   # camera medium, quad light: finite transmittance
    ("mirror_128x72_4spp", "next/mirror.json", (128, 72), 4),
    ("dielectric_128x72_4spp", "next/dielectric.json", (128, 72), 4),
    ("conductor_128x72_4spp", "next/conductor.json", (128, 72), 4),
    ("plastic_128x72_4spp", "next/plastic.json", (128, 72), 4),
    ("rough_plastic_128x72_4spp", "next/rough_plastic.json", (128, 72), 4),
    ("dielectric_hg_medium_128x72_4spp", "next/dielectric_hg_medium.json", (128, 72), 4),
    ("sphere_disk_128x72_4spp", "next/sphere_disk.json", (128, 72), 4),  # spheres, disk, sphere light + env light selection
    ("oren_nayar_128x72_4spp", "next/oren_nayar.json", (128, 72), 4),
    ("phong_128x72_4spp", "next/phong.json", (128, 72), 4),
    ("mixed_128x72_4spp", "next/mixed.json", (128, 72), 4),                # nested: lambert + rough_conductor
    ("rough_coat_128x72_4spp", "next/rough_coat.json", (128, 72), 4),      # nested: GGX coat over a substrate
    ("transparency_128x72_4spp", "next/transparency.json", (128, 72), 4),  # forward lobe with an opacity: pass-through + shadow attenuation
    ("thinsheet_128x72_4spp", "next/thinsheet.json", (128, 72), 4),        # forward lobe with an angle-dependent transmittance
    # the fork's remaining transmittance models, in the Cornell box's camera medium (distance sampling + volume NEE towards the quad light)
    ("trans_erlang_96x54_8spp", "next/trans_erlang.json", (96, 54), 8),
    ("trans_davis_96x54_8spp", "next/trans_davis.json", (96, 54), 8),
    ("trans_davis_weinstein_96x54_8spp", "next/trans_davis_weinstein.json", (96, 54), 8),
    ("trans_interpolated_96x54_8spp", "next/trans_interpolated.json", (96, 54), 8),    # linear (Dirac mediumMedium) + erlang
    ("trans_interpolated2_96x54_8spp", "next/trans_interpolated2.json", (96, 54), 8),  # exponential + davis
    # 8-bit PNG bitmaps: gamma-table RGB albedo (interpolated, wrapped), palette albedo (nearest, clamped, scaled), scalar-average roughness
    ("ldr_texture_128x72_4spp", "next/ldr_texture.json", (128, 72), 4),
    # .bmp (24-bit bottom-up, 8-bit palette top-down) and .tga (32-bit raw top-down, 8-bit grey run-length encoded) bitmaps: stb_image in the reference
    ("ldr_texture2_128x72_4spp", "next/ldr_texture2.json", (128, 72), 4),
    # baseline JPEG bitmaps: 4:2:0, 4:2:2 with optimised tables and restart intervals, 4:4:4 with partial MCUs, one-component grey
    ("ldr_texture3_128x72_4spp", "next/ldr_texture3.json", (128, 72), 4),
    ("fog_nested_96x54_16spp", "next/fog_nested.json", (96, 54), 16),      # nested BSDFs hit by paths that run inside a medium (k_volume_nested's case)
    ("ldr_texture4_128x72_4spp", "next/ldr_texture4.json", (128, 72), 4),  # the same through progressive JPEG files
    # round 3: long enough to contain rare events -- box culls and exact ties that only Embree's own traversal order decides (oracle/embree_bvh4.h)
    ("cornell_sobol_96x54_64spp", "cornell-box/scene_sobol.json", (96, 54), 64),     # pixel (67,7) sample 41: ceiling quad culled by its flat box
    ("cornell_sobol_64x36_128spp", "cornell-box/scene_sobol.json", (64, 36), 128),   # pixel (47,33) sample 15: floor / wall tie, last visited wins
    ("cornell_stock_96x54_48spp", "cornell-box/scene.json", (96, 54), 48),           # BASELINE config 3's scene as shipped (uniform sampler)
    ("c2_320x180_32spp", "materialtest/c2.json", (320, 180), 32),                    # first-found triangle on an exact tie (2 pixels)
    ("c2_32x18_320spp", "materialtest/c2.json", (32, 18), 320),                      # sample indices >= 256
    ("c5_128x72_64spp", "materialtest/c5.json", (128, 72), 64),
    ("inst_many_96x54_48spp", "instances/inst_many.json", (96, 54), 48),
    ("cornell_fog_96x54_64spp", "cornell-box/scene_fog.json", (96, 54), 64),
]


# Fixtures that contain an event only Embree's own traversal decides (a primitive culled by its leaf box, two primitives at one binary32 distance:
# DESIGN.md section 2a, oracle/embree_bvh4.h, oracle_ctx::refIntersect), all through its RCPPS-based reciprocal.  Where RCPPS matches the fixture
# host they must be exact in EVERY pixel; elsewhere one flipped decision desynchronises a tile's sequential PCG stream, so only the image mean is
# asserted there (and the test says so by its id).
EMBREE_DECIDED = {"trans_erlang_96x54_8spp", "trans_davis_weinstein_96x54_8spp", "trans_interpolated_96x54_8spp",
                  "cornell_sobol_96x54_64spp", "cornell_sobol_64x36_128spp", "cornell_stock_96x54_48spp", "c2_320x180_32spp", "cornell_fog_96x54_64spp"}


def intel_rcp(oracle_py):
    """RCPPS is implementation defined; the fixtures were rendered on an Intel Xeon.  True when this
    host's rcp approximation produces the same bits for a probe set (then bit-exactness is expected)."""
    import ctypes as C
    L = oracle_py.lib()
    L.oracle_embree_rcp.restype = C.c_float
    L.oracle_embree_rcp.argtypes = [C.c_float]
    probes = {3.0: 0x3EAAAAAA, 7.0: 0x3E124924, 0.3: 0x40555555, 1.7: 0x3F169697, 123.456: 0x3C04B615}
    return all(np.float32(L.oracle_embree_rcp(C.c_float(k))).view(np.uint32) == v for k, v in probes.items())


@pytest.mark.parametrize("name,scene,res,spp", CASES)
def test_oracle_matches_reference(ta, oracle_py, name, scene, res, spp):
    ref = ta.load_pfm(os.path.join(GOLDEN, name + ".pfm"))
    hs = ta.HostScene(scene_path(scene), resolution=res)
    # reference behaviour: tile-sequential PCG, glibc libm, Embree's rcp()
    img, cnt = oracle_py.Oracle(hs, **oracle_py.REF_RCP).render_image(spp)
    assert img.shape == ref.shape
    assert (cnt == spp).all()
    rmse = float(np.sqrt(((img - ref)**2).mean()))
    differing = int((img != ref).any(axis=2).sum())
    if intel_rcp(oracle_py):
        # RCPPS as on the fixture host: the reference's image, every pixel, every bit
        assert differing == 0, f"{name}: {differing} pixels differ from the reference (rmse {rmse:.3e})"
        return
    # Another reciprocal approximation: isolated decisions may flip (and, in EMBREE_DECIDED fixtures, desynchronise a tile's sequential PCG
    # stream), so only statistical agreement can be asserted -- and the report shows that this weaker mode ran.
    if name not in EMBREE_DECIDED:
        assert rmse < 1e-4, f"{name}: rmse {rmse}"
    assert abs(float(img.mean())/float(ref.mean()) - 1.0) < 5e-3
    pytest.skip(f"RCPPS differs from the fixture host: {name} checked statistically only ({differing} pixels differ, rmse {rmse:.2e})")


def test_oracle_adaptive_sampling_matches_reference(ta, oracle_py):
    """generateWork (error percentile, dilation, stochastic distribution of the budget with the integrator's
    own PCG) + renderTile with per-variance-tile sample counts, against the reference run with
    adaptive_sampling on: 1 uniform pass and 3 adaptive ones."""
    ref = ta.load_pfm(os.path.join(GOLDEN, "c1_adaptive_128x72_64spp.pfm"))
    hs = ta.HostScene(scene_path("materialtest/materialtest.json"), resolution=(128, 72))
    assert hs.adaptive
    img, cnt, rec = oracle_py.Oracle(hs, **oracle_py.REF_RCP).render_reference(64, spp_step=16, adaptive=True)
    assert cnt.min() < 64 < cnt.max()            # samples really moved between regions
    assert abs(float(cnt.mean()) - 64.0) < 0.5   # the budget is conserved (up to the stochastic rounding)
    # a variance tile's record saw exactly the samples of its pixels
    tiles = cnt.reshape(18, 4, 32, 4).sum(axis=(1, 3))
    assert np.array_equal(tiles, rec[..., 0].astype(np.uint32))
    rmse = float(np.sqrt(((img - ref)**2).mean()))
    assert rmse < 1e-3, rmse
    if intel_rcp(oracle_py):
        assert (img == ref).all()


@pytest.mark.parametrize("name,res,spp,kw,adaptive", [
    ("nonexp_stock_128_16spp", (128, 128), 16, {}, False),                                   # as shipped: uniform sampler
    ("nonexp_stock_sobol_128_16spp", (128, 128), 16, {"sobol": True}, False),
    ("nonexp_stock_deep_96_8spp", (96, 96), 8, {"sobol": True, "max_bounces": 16}, False),
    ("nonexp_stock_adaptive_128_48spp", (128, 128), 48, {}, True),                           # as shipped: adaptive on
])
def test_oracle_matches_reference_on_the_non_exponential_scene(ta, oracle_py, name, res, spp, kw, adaptive):
    """data/example-scenes/non-exponential/scene.json, the scene this fork of Tungsten exists for: linear, quadratic,
    double-exponential and pulse transmittances (surface/medium start and end forms), ForwardBsdf walls that switch the
    medium, multi-segment shadow rays through them, 'power' quad lights, max_bounces 3."""
    ref = ta.load_pfm(os.path.join(GOLDEN, name + ".pfm"))
    hs = ta.HostScene(scene_path("non-exponential/scene.json"), resolution=res, **kw)
    o = oracle_py.Oracle(hs, **oracle_py.REF_RCP)
    img = o.render_reference(spp, spp_step=16, adaptive=True)[0] if adaptive else o.render_image(spp)[0]
    assert float(np.sqrt(((img - ref)**2).mean())) < 1e-4
    if intel_rcp(oracle_py):
        assert (img == ref).all()


def test_ieee_division_mode_stays_within_tolerance(ta, oracle_py):
    """What the GPU computes (IEEE division for t,u,v instead of RCPPS) differs from the reference
    only through chaotic path divergence: quantify it on a fixture."""
    ref = ta.load_pfm(os.path.join(GOLDEN, "c2_128x72_48spp.pfm"))
    hs = ta.HostScene(scene_path("materialtest/c2.json"), resolution=(128, 72))
    img, _ = oracle_py.Oracle(hs, **oracle_py.REF).render_image(48)
    rel = np.abs(img - ref).mean()/ref.mean()
    assert rel < 2e-2  # a few diverged samples per pixel at 48 spp; converges with spp


def test_path_local_stream_is_unbiased(ta, oracle_py):
    """GPU mode (path-local PCG + portable libm) against the reference image: same estimator, different
    random decisions -> the difference is Monte-Carlo noise that shrinks with spp."""
    hs = ta.HostScene(scene_path("materialtest/c2.json"), resolution=(128, 72))
    ref = ta.load_pfm(os.path.join(GOLDEN, "c2_128x72_48spp.pfm"))
    img, _ = oracle_py.Oracle(hs, **oracle_py.GPU).render_image(48)
    # compare on a 4x4 box-filtered version to average noise down
    def pool(a):
        return a[:72, :128].reshape(18, 4, 32, 4, 3).mean(axis=(1, 3))
    rel = np.abs(pool(img) - pool(ref)).mean()/pool(ref).mean()
    assert rel < 0.05
    assert abs(img.mean()/ref.mean() - 1.0) < 0.02


def test_golden_fixtures_match_their_manifest():
    """Every reference-rendered fixture is listed in tests/golden/MANIFEST.sha256 with its digest: the PFMs came from a build of
    the unmodified reference that this repo's recipe cannot redo, so their bytes are pinned here."""
    import hashlib
    manifest = {}
    for line in open(os.path.join(GOLDEN, "MANIFEST.sha256")):
        digest, name = line.split()
        manifest[name] = digest
    on_disk = set()
    for root, _, files in os.walk(GOLDEN):
        for f in files:
            if f.endswith((".pfm", ".dat", ".png")) or f in ("ref_kat.json", "ref_kat2.json", "ref_kat3.json", "ref_kat4.json", "ref_kat5.npz", "ref_kat6.json"):
                on_disk.add(os.path.relpath(os.path.join(root, f), GOLDEN))
    assert on_disk == set(manifest), f"fixtures without manifest entry or vice versa: {sorted(on_disk ^ set(manifest))}"
    for name, digest in manifest.items():
        assert hashlib.sha256(open(os.path.join(GOLDEN, name), "rb").read()).hexdigest() == digest, name


def _load_pfm_any(path):
    """PF (3 channels) or Pf (1 channel) -> (h, w, channels) float32, top row first."""
    with open(path, "rb") as f:
        magic = f.readline().strip()
        w, h = map(int, f.readline().split())
        f.readline()
        ch = 1 if magic == b"Pf" else 3
        return np.frombuffer(f.read(), np.float32).reshape(h, w, ch)[::-1]


@pytest.mark.parametrize("name,scene,res,spp", [
    ("aov", "next/aov.json", (128, 72), 4),                       # late records behind glass, escaping primary rays, environment albedo
    ("cornell", "cornell-box/scene_sobol.json", (96, 54), 5),     # quad light: emissive albedo, finite-light visibility, odd sample count
])
def test_oracle_output_buffers_match_reference(ta, oracle_py, name, scene, res, spp):
    """renderer "output_buffers" as the compiled reference wrote them (tests/golden/aov/, make_aov_golden.py): depth with
    sample_variance, normal with two_buffer_variance, albedo, visibility with both -- the values of PathTracer.cpp:78-96,133-140
    accumulated by cameras/OutputBuffer.hpp and saved by OutputBuffer::save (mean, A / B halves, scaled variance)."""
    outputs = [("depth", False, True), ("normal", True, False), ("albedo", False, False), ("visibility", True, True)]
    hs = ta.HostScene(scene_path(scene), resolution=res, output_buffers=outputs)
    o = oracle_py.Oracle(hs, **oracle_py.REF_RCP)
    bufs = o.bind_output_buffers(hs.res_x*hs.res_y)
    img, cnt = o.render_image(spp, spp_step=spp)
    exact_host = intel_rcp(oracle_py)
    checked = 0

    def compare(file, values, channels):
        nonlocal checked
        ref = _load_pfm_any(os.path.join(GOLDEN, "aov", file))
        got = np.asarray(values, np.float32).reshape(hs.res_y, hs.res_x, channels)
        assert ref.shape == got.shape, file
        both_nan = np.isnan(ref) & np.isnan(got)
        same = (ref == got) | both_nan
        finite = np.isfinite(ref) & np.isfinite(got)
        if exact_host: # RCPPS as on the fixture host: every value of every file, NaN ("no sample") positions included
            assert np.array_equal(np.isfinite(ref), np.isfinite(got)), file
            assert same.all(), f"{file}: {int((~same.all(axis=2)).sum())} pixels differ from the reference"
        else:          # another reciprocal approximation flips isolated decisions: statistical agreement only
            assert (np.isfinite(ref) == np.isfinite(got)).mean() > 0.999, file
            assert float(np.sqrt(((ref - got)[finite]**2).mean())) < 1e-3*max(1.0, float(np.abs(ref[finite]).max())), file
        checked += 1

    def mean_of(b, channels):
        if "b" not in b:
            return b["a"]
        n = b["count"].astype(np.float32)[:, None]
        return (b["a"]*((b["count"] + 1)//2).astype(np.float32)[:, None] + b["b"]*(b["count"]//2).astype(np.float32)[:, None])/np.maximum(n, np.float32(1))

    def scaled_variance(b):
        n = b["count"]
        with np.errstate(invalid="ignore", divide="ignore"):
            return b["variance"]/(n*np.maximum(1, n.astype(np.int64) - 1)).astype(np.float32)[:, None]

    compare(f"{name}_color.pfm", img, 3)
    compare(f"{name}_depth.pfm", bufs["depth"]["a"], 1)
    compare(f"{name}_depthVariance.pfm", scaled_variance(bufs["depth"]), 1)
    compare(f"{name}_normal.pfm", mean_of(bufs["normal"], 3), 3)
    compare(f"{name}_normalA.pfm", bufs["normal"]["a"], 3)
    compare(f"{name}_normalB.pfm", bufs["normal"]["b"], 3)
    compare(f"{name}_albedo.pfm", bufs["albedo"]["a"], 3)
    compare(f"{name}_visibility.pfm", mean_of(bufs["visibility"], 1), 1)
    compare(f"{name}_visibilityA.pfm", bufs["visibility"]["a"], 1)
    compare(f"{name}_visibilityB.pfm", bufs["visibility"]["b"], 1)
    compare(f"{name}_visibilityVariance.pfm", scaled_variance(bufs["visibility"]), 1)
    assert checked == 11


def test_coplanar_tie_follows_the_reference(ta, oracle_py):
    """One recorded event of the compiled reference (tests/tools/ref_path_dump.cpp on assets/next/trans_erlang.json, pixel (12, 21), sample 3,
    vertex 13): a ray from below the Cornell box reaches the floor quad and the short box's bottom face at the same binary32 distance 0x3eec7e10;
    the reference reports the CUBE (Embree skips the quad's leaf once the cube's hit has shortened the ray).  The oracle's reference mode must
    agree; the GPU contract (primitive-level tests only) is the quad at the same distance."""
    hs = ta.HostScene(scene_path("next/trans_erlang.json"), resolution=(96, 54))
    bits = lambda *w: np.array(w, np.uint32).view(np.float32)
    ray = np.concatenate([bits(0x3e1cee6c, 0xbecae726, 0x3d0e44d0), [0.0], bits(0x3eb29f54, 0x3f5ba3c9, 0x3ec10ef0), [np.inf]]).astype(np.float32)
    # the vertex before it on the same path: same direction, no tie -> the floor quad in both modes
    ray2 = np.concatenate([bits(0xbdda858e, 0xbeecb858, 0xbeb0d9ba), [0.0], bits(0x3eb29f54, 0x3f5ba3c9, 0x3ec10ef0), [np.inf]]).astype(np.float32)
    gpu = oracle_py.Oracle(hs, **oracle_py.GPU).trace_closest(np.stack([ray, ray2]))
    assert gpu[:, 0].view(np.uint32).tolist() == [0x3eec7e10, 0x3f09f435]
    assert (gpu[:, 3].view(np.uint32) >> 29).tolist() == [1, 1]          # quad, quad
    if intel_rcp(oracle_py):
        ref = oracle_py.Oracle(hs, **oracle_py.REF_RCP).trace_closest(np.stack([ray, ray2]))
        assert ref[:, 0].view(np.uint32).tolist() == [0x3eec7e10, 0x3f09f435]
        assert (ref[:, 3].view(np.uint32) >> 29).tolist() == [2, 1]      # cube, quad


def test_embree_decided_events_of_the_cornell_box(ta, oracle_py):
    """Two recorded events of the compiled reference on BASELINE config 3's scene (cornell-box/scene_sobol.json, camera rays, found with
    tests/tools/ref_path_dump.cpp beside ORACLE_TRACE_AT):
      1. 96x54, pixel (67,7), sample 41: the ray leaves through the edge between the ceiling and the right wall.  The ceiling quad's own test passes
         (t = 0x40fcbc5a, u = 3e-8) but the ray misses the ceiling's flat Embree leaf box by an ulp, so the reference reports the WALL at 0x40fcbc5d.
      2. 64x36, pixel (47,33), sample 15: floor and right wall at the SAME distance 0x40d5e339; Quad::intersect keeps t <= farT, so the leaf Embree
         visits last wins -- the floor.
    The oracle's reference mode (Embree's BVH4 and traversal restated) must agree; the GPU contract is the primitives' own tests in index order:
    the ceiling, and the wall (the later quad keeps the tie)."""
    hs = ta.HostScene(scene_path("cornell-box/scene_sobol.json"), resolution=(96, 54))
    bits = lambda *w: np.array(w, np.uint32).view(np.float32)
    org = bits(0x00000000, 0x3f800000, 0x40d9999a)
    r1 = np.concatenate([org, [1e-4], bits(0x3e01a738, 0x3e01a737, 0xbf7bdcd1), [np.inf]]).astype(np.float32)
    r2 = np.concatenate([org, [1e-4], bits(0x3e1933b4, 0xbe1933b4, 0xbf7a3447), [np.inf]]).astype(np.float32)
    gpu = oracle_py.Oracle(hs, **oracle_py.GPU).trace_closest(np.stack([r1, r2]))
    assert gpu[:, 0].view(np.uint32).tolist() == [0x40fcbc5a, 0x40d5e339]
    assert gpu[:, 3].view(np.uint32).tolist() == [(1 << 29) | 1, (1 << 29) | 3]      # ceiling, right wall
    if intel_rcp(oracle_py):
        ref = oracle_py.Oracle(hs, **oracle_py.REF_RCP).trace_closest(np.stack([r1, r2]))
        assert ref[:, 0].view(np.uint32).tolist() == [0x40fcbc5d, 0x40d5e339]
        assert ref[:, 3].view(np.uint32).tolist() == [(1 << 29) | 3, (1 << 29) | 0]  # right wall, floor
