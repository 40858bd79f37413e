"""Host logic: scene loading, shards, ABI layout, library exports (no GPU)."""
import ctypes as C
import json
import os
import re
import subprocess
import tempfile

import numpy as np
import pytest

from conftest import REPO, scene_path


def test_load_materialtest(ta):
    s = ta.HostScene(scene_path("materialtest/materialtest.json"))
    assert (s.res_x, s.res_y) == (1000, 563)
    assert s.spp == 32 and s.spp_step == 16 and s.use_sobol and s.adaptive
    d = s.desc.contents
    assert d.n_tris == 80768  # Ring 17536 + T-Ball 49920 + Stand 13312
    assert d.n_quads == 1 and d.n_lights == 1 and d.env.present == 1 and d.env.samplable == 1
    assert (d.env.dist_w, d.env.dist_h) == (1024, 512)
    mc = np.ctypeslib.as_array(d.env.marginal_cdf, shape=(513,))
    assert mc[0] == 0.0 and mc[-1] == 1.0 and (np.diff(mc) >= 0).all()
    assert abs(d.camera.plane_dist - 1.0/np.tan(np.radians(35.0)/2)) < 1e-5
    cdf = np.array(d.camera.filter_cdf[:32])
    assert cdf[0] == 0 and cdf[31] == 1 and (np.diff(cdf) >= 0).all()


def test_resolution_override_and_errors(ta):
    s = ta.HostScene(scene_path("materialtest/c2.json"), resolution=(64, 32))
    assert (s.res_x, s.res_y) == (64, 32) and not s.adaptive
    with pytest.raises(RuntimeError):
        ta.HostScene(scene_path("materialtest/does_not_exist.json"))
    with tempfile.TemporaryDirectory() as td:
        bad = os.path.join(td, "bad.json")
        open(bad, "w").write('{"bsdfs": [{"name": "x", "type": "hair"}]}')
        with pytest.raises(RuntimeError, match="outside the hot-path scope"):
            ta.HostScene(bad)
        open(bad, "w").write('{"primitives": [')
        with pytest.raises(RuntimeError, match="JSON"):
            ta.HostScene(bad)
        open(bad, "w").write('{"integrator": {"type": "bidirectional_path_tracer"}}')
        with pytest.raises(RuntimeError, match="path_tracer only"):
            ta.HostScene(bad)


def test_empty_scene_gets_default_light(ta):
    with tempfile.TemporaryDirectory() as td:
        p = os.path.join(td, "empty.json")
        open(p, "w").write('{"camera": {"resolution": [8, 8]}}')
        s = ta.HostScene(p)
        d = s.desc.contents
        assert d.n_tris == 0 and d.n_lights == 1 and d.env.present == 1  # TraceableScene.hpp:97-102


@pytest.mark.parametrize("res", [(128, 72), (100, 57), (16, 16), (17, 1), (1, 1)])
@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_shards_partition_the_image(ta, oracle_py, res, world):
    w, h = res
    seeds = oracle_py.tile_seeds(w, h)
    seen = np.zeros(w*h, np.int32)
    for r in range(world):
        sh = ta.Shard(w, h, 0xBA5EBA11, r, world)
        px = sh.pixel_index
        seen[px] += 1
        d = sh.desc.contents
        ts = np.ctypeslib.as_array(d.tile_seed, shape=(sh.n_pixels,)) if sh.n_pixels else np.zeros(0, np.uint32)
        tile = (px % w)//16 + (px//w)//16*((w + 15)//16)
        assert (tile % world == r).all()          # tile t -> rank t % world
        assert (ts == seeds[tile]).all()          # seeds as PathTraceIntegrator::diceTiles draws them
    assert (seen == 1).all()                      # every pixel exactly once


def _header_functions():
    text = open(os.path.join(REPO, "include", "tungsten_amd.h")).read()
    return sorted(set(re.findall(r"\b(tgpu_[a-z_]+)\s*\(", text)))


def test_core_library_exports_every_declared_symbol(ta):
    lib = ta.core_lib()  # loads without a GPU; compute entry points are not called here
    names = _header_functions()
    assert len(names) >= 16
    for n in names:
        assert hasattr(lib, n), f"{n} is declared in include/tungsten_amd.h but not exported"


def test_abi_struct_sizes_match_the_c_header(ta):
    from tungsten_amd import _abi
    src = r'''
#include "tungsten_amd.h"
#include <stdio.h>
int main(void){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(tgpu_texture), sizeof(tgpu_bitmap),
 sizeof(tgpu_bsdf), sizeof(tgpu_quad), sizeof(tgpu_cube), sizeof(tgpu_env), sizeof(tgpu_camera), sizeof(tgpu_settings),
 sizeof(tgpu_light), sizeof(tgpu_scene_desc), sizeof(tgpu_shard_desc), sizeof(tgpu_stats), sizeof(tgpu_medium),
 sizeof(tgpu_master), sizeof(tgpu_instances), sizeof(tgpu_sample_record), sizeof(tgpu_sphere), sizeof(tgpu_disk));return 0;}'''
    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, "s.c")
        open(c, "w").write(src)
        exe = os.path.join(td, "s")
        subprocess.run(["gcc", "-std=c99", "-I", os.path.join(REPO, "include"), c, "-o", exe], check=True)  # header is plain C
        sizes = [int(t) for t in subprocess.run([exe], capture_output=True, text=True, check=True).stdout.split()]
    mirrors = [_abi.Texture, _abi.Bitmap, _abi.Bsdf, _abi.Quad, _abi.Cube, _abi.Env, _abi.Camera, _abi.Settings,
               _abi.Light, _abi.SceneDesc, _abi.ShardDesc, _abi.Stats, _abi.Medium, _abi.Master, _abi.Instances,
               _abi.SampleRecord, _abi.Sphere, _abi.Disk]
    assert sizes == [C.sizeof(m) for m in mirrors]


def test_no_gpu_means_loud_failure(ta):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(ta.CoreError, match="no CPU fallback"):
        ta.Context(0)


def test_product_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under tungsten_amd/ may reference it."""
    for root, _, files in os.walk(os.path.join(REPO, "tungsten_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".h", ".hip")):
                text = open(os.path.join(root, f), errors="ignore").read()
                for needle in ("oracle_", "liboracle", "import oracle", "from oracle", "oracle/"):
                    assert needle not in text, f"{f} links, imports or calls the oracle ({needle})"


def test_pfm_roundtrip(ta, tmp_path):
    img = np.random.default_rng(0).random((5, 7, 3), dtype=np.float32)
    p = tmp_path / "x.pfm"
    ta.save_pfm(p, img)
    assert np.array_equal(ta.load_pfm(p), img)


def test_reference_side_integrator_compiles_against_the_reference_headers():
    """integration/Mi355xPathTraceIntegrator.{hpp,cpp} -- the Integrator subclass of INTEGRATION.md section 2 -- type-checks against
    the reference's own headers (only where /root/reference exists: the GPU box has no reference tree)."""
    import subprocess
    if not os.path.isdir("/root/reference/src/core"):
        pytest.skip("no reference tree here")
    r = subprocess.run(["make", "-C", os.path.join(REPO, "oracle"), "-f", "Makefile.ref", "integration-check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def _png_decode(path):
    """Independent PNG decode (zlib + the five scanline filters) -> (h, w, 3) uint8 RGB and the colour type."""
    import struct
    import zlib
    s = open(path, "rb").read()
    o, idat, pal = 8, b"", None
    while o < len(s):
        n, = struct.unpack(">I", s[o:o + 4])
        tag, data = s[o + 4:o + 8], s[o + 8:o + 8 + n]
        o += 12 + n
        if tag == b"IHDR":
            w, h, depth, ct = struct.unpack(">IIBB", data[:10])
        elif tag == b"PLTE":
            pal = data
        elif tag == b"IDAT":
            idat += data
    raw = zlib.decompress(idat)
    bpp = {2: 3, 3: 1, 6: 4}[ct]
    st = w*bpp
    img = np.zeros((h, st), np.int32)
    prev = [0]*st
    for y in range(h):
        f, row, cur = raw[y*(st + 1)], raw[y*(st + 1) + 1:(y + 1)*(st + 1)], [0]*st
        for x in range(st):
            a, b, c = (cur[x - bpp] if x >= bpp else 0), prev[x], (prev[x - bpp] if x >= bpp else 0)
            q = a + b - c
            pa, pb, pc = abs(q - a), abs(q - b), abs(q - c)
            pred = [0, a, b, (a + b) >> 1, a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)][f]
            cur[x] = (row[x] + pred) & 255
        img[y] = prev = cur
    img = img.reshape(h, w, bpp)
    if ct == 3:
        img = np.frombuffer(pal, np.uint8).reshape(-1, 3)[img[:, :, 0]].astype(np.int32)
    return img[:, :, :3]


def test_png_bitmap_textures_expand_to_the_texels_the_reference_reads(ta):
    """8-bit PNG bitmaps (io/ImageIO.cpp:386-407,496-526; textures/BitmapTexture.cpp:15-23,140-156,211-219): an RGB request goes through the
    integer gamma table int(255*(i/255)^2.2) unless gamma_correct is off; a scalar (REQUEST_AVERAGE) request is the integer (r+g+b)/3 of the
    uncorrected bytes; a texel is float(byte)*(1/255).  The files use colour types 2, 6 and 3, all five scanline filters, dynamic Huffman
    blocks and split IDAT chunks; the decode is checked against zlib.  The rendered image is pinned by tests/test_oracle_golden.py."""
    hs = ta.HostScene(scene_path("next/ldr_texture.json"), resolution=(32, 18))
    d = hs.desc.contents
    tex = os.path.join(REPO, "assets", "next", "textures")
    gamma = np.array([int(255*(i/255)**2.2) for i in range(256)])
    k = np.float32(1.0/255.0)
    rough = _png_decode(os.path.join(tex, "rough.png"))
    s = ((rough[:, :, 0] + rough[:, :, 1] + rough[:, :, 2])//3).astype(np.float32)*k
    expect = {
        (64, 48): (gamma[_png_decode(os.path.join(tex, "albedo.png"))].astype(np.float32)*k, 0, 1, 1.0),
        (16, 16): (_png_decode(os.path.join(tex, "tiles.png")).astype(np.float32)*k, 1, 0, 0.8),
        (32, 32): (np.stack([s, s, s], -1), 0, 1, 1.0),
    }
    seen = 0
    for i in range(d.n_bitmaps):
        b = d.bitmaps[i]
        if (b.w, b.h) not in expect:
            continue
        texels, clamp, linear, scale = expect[(b.w, b.h)]
        got = np.ctypeslib.as_array(b.texels, (b.h, b.w, 3))
        assert (got == texels).all(), (b.w, b.h)
        assert (b.clamp, b.linear) == (clamp, linear) and b.scale == np.float32(scale)
        avg = np.zeros(3, np.float32)  # BitmapTexture::init: _avg += texel/float(w*h), row-major
        for t in texels.reshape(-1, 3):
            avg += t/np.float32(b.w*b.h)
        assert (np.array(list(b.average), np.float32) == avg).all()
        seen += 1
    assert seen == 3


def test_bmp_and_tga_bitmap_textures_decode_like_stb_image(ta, tmp_path):
    """.bmp and .tga bitmaps (the reference hands them to stb_image, io/ImageIO.cpp:489-490): 24-bit bottom-up and 8-bit-palette top-down BMP,
    32-bit raw top-down and 8-bit grey run-length-encoded TGA.  Texels against an independent decode (pillow) pushed through the same
    byte -> float rule as the PNGs; the rendered image is pinned by a reference-rendered golden (ldr_texture2_128x72_4spp).  Forms outside the
    covered set are refused by name."""
    import struct
    from PIL import Image
    hs = ta.HostScene(scene_path("next/ldr_texture2.json"), resolution=(32, 18))
    d = hs.desc.contents
    tex = os.path.join(REPO, "assets", "next", "textures")
    gamma = np.array([int(255*(i/255)**2.2) for i in range(256)])
    k = np.float32(1.0/255.0)
    rgb = lambda f: np.asarray(Image.open(os.path.join(tex, f)).convert("RGB")).astype(np.int64)
    grey = rgb("rough.tga")
    s = ((grey[:, :, 0] + grey[:, :, 1] + grey[:, :, 2])//3).astype(np.float32)*k
    expect = [gamma[rgb("albedo.tga")].astype(np.float32)*k, gamma[rgb("albedo.bmp")].astype(np.float32)*k, rgb("tiles.bmp").astype(np.float32)*k, np.stack([s, s, s], -1)]
    matched = 0
    for i in range(d.n_bitmaps):
        b = d.bitmaps[i]
        got = np.ctypeslib.as_array(b.texels, (b.h, b.w, 3))
        matched += any(e.shape == got.shape and (e == got).all() for e in expect)
    assert matched >= 4
    # refused, with the reason in the message: 16-bit BMP, run-length-encoded BMP, colour-mapped TGA, truncated files
    scene = json.load(open(scene_path("next/ldr_texture2.json")))
    for p in scene["primitives"]:
        if p.get("file"):
            p["file"] = os.path.join(REPO, "assets", "materialtest", p["file"])
        if p.get("emission") and isinstance(p["emission"], str):
            p["emission"] = os.path.join(REPO, "assets", "materialtest", p["emission"])
    for b in scene["bsdfs"]:   # the other textures keep pointing at the vendored files
        for key in ("albedo", "roughness"):
            v = b.get(key)
            if isinstance(v, str) and v.startswith("textures/"):
                b[key] = os.path.join(REPO, "assets", "next", v)
            elif isinstance(v, dict) and str(v.get("file", "")).startswith("textures/"):
                v["file"] = os.path.join(REPO, "assets", "next", v["file"])
    good_bmp = open(os.path.join(tex, "albedo.bmp"), "rb").read()
    good_tga = open(os.path.join(tex, "albedo.tga"), "rb").read()
    cases = {
        "bits16.bmp": (good_bmp[:28] + struct.pack("<H", 16) + good_bmp[30:], "16 bits per pixel"),
        "rle.bmp": (good_bmp[:30] + struct.pack("<I", 1) + good_bmp[34:], "run-length"),
        "short.bmp": (good_bmp[:2000], "truncated"),
        "mapped.tga": (good_tga[:1] + b"\x01" + good_tga[2:], "colour-mapped"),
        "short.tga": (good_tga[:500], "truncated"),
    }
    for name, (data, reason) in cases.items():
        (tmp_path / name).write_bytes(data)
        scene["bsdfs"][1]["albedo"] = str(tmp_path / name)
        f = tmp_path / "s.json"
        f.write_text(json.dumps(scene))
        with pytest.raises(RuntimeError, match=reason):
            ta.HostScene(str(f))


def test_unsupported_bitmap_files_fail_loudly(ta, tmp_path):
    d = json.load(open(scene_path("next/ldr_texture.json")))
    bad = tmp_path/"broken.png"
    bad.write_bytes(b"\x89PNG\r\n\x1a\n" + b"\0"*32)
    d["bsdfs"][0]["albedo"] = str(bad)
    for p in d["primitives"]:
        if p.get("file"):
            p["file"] = os.path.join(REPO, "assets", "next", p["file"])
        if isinstance(p.get("emission"), str):
            p["emission"] = os.path.join(REPO, "assets", "next", p["emission"])
    for b in d["bsdfs"][1:]:
        for key in ("albedo", "roughness"):
            if isinstance(b.get(key), str):
                b[key] = os.path.join(REPO, "assets", "next", b[key])
            elif isinstance(b.get(key), dict) and "file" in b[key]:
                b[key]["file"] = os.path.join(REPO, "assets", "next", b[key]["file"])
    path = tmp_path/"scene.json"
    path.write_text(json.dumps(d))
    with pytest.raises(RuntimeError, match="Unable to load texture"):
        ta.HostScene(str(path), resolution=(32, 18))


@pytest.mark.parametrize("tonemap", ["linear", "gamma", "reinhard", "filmic", "pbrt"])
def test_ldr_output_matches_the_reference_png(ta, tonemap, tmp_path):
    """The LDR half of Integrator::writeBuffers (Integrator.cpp:56-80): Camera::tonemap (cameras/Tonemap.hpp:26-62: clamp at 0, the operator),
    *255, truncate, clamp, PNG.  The compiled reference wrote tests/golden/tonemap_<op>_48x27_4spp.png NEXT TO the .pfm of the same render, so
    this host's PFM -> PNG path, fed the reference's float image, has to reproduce the reference's PNG pixels byte for byte."""
    from PIL import Image
    from conftest import GOLDEN
    name = f"tonemap_{tonemap}_48x27_4spp"
    hdr = ta.load_pfm(os.path.join(GOLDEN, name + ".pfm"))
    want = np.asarray(Image.open(os.path.join(GOLDEN, name + ".png")).convert("RGB"))
    assert want.shape == hdr.shape == (27, 48, 3)
    out = str(tmp_path / "out.png")
    rgb = np.ascontiguousarray(hdr, np.float32)
    rc = ta.host_lib().tgh_save_png_tonemapped(out.encode(), tonemap.encode(), rgb.ctypes.data_as(C.POINTER(C.c_float)), 48, 27)
    assert rc == 0
    got = np.asarray(Image.open(out).convert("RGB"))
    assert np.array_equal(got, want), f"{int((got != want).any(axis=2).sum())} of {48*27} pixels differ from the reference's PNG"
    assert len(np.unique(want)) > 50   # a real image, not a constant


def test_no_device_function_outgrows_the_branch_range():
    """A leaf device function above 128 KB is miscompiled by this toolchain (far branches through the return-address register:
    profiles/r03_nested_notes.txt) -- the cause of round 2's GPU fault.  Every non-kernel function of the built objects has to stay below."""
    import glob
    import sys
    sys.path.insert(0, os.path.join(REPO, "profiles"))
    import check_leaf_functions as chk
    objects = sorted(glob.glob(os.path.join(REPO, "build", "core", "*.o"))) or [os.path.join(REPO, "tungsten_amd", "libtungsten_amd.so")]
    seen = 0
    for obj in objects:
        for name, size, kernel in chk.device_functions(obj):
            if not kernel:
                seen += 1
                assert size < chk.LIMIT, f"{os.path.basename(obj)}: device function {name} is {size} bytes"
    assert seen >= 3   # the out-of-line children of the nested BSDFs exist


def _damaged_copies(tmp_path):
    """Small scenes whose one resource file is a damaged copy of a vendored asset: truncated at several lengths, bytes flipped, counts blown up.
    Yields (scene path, description)."""
    import struct
    rng = np.random.default_rng(7)
    mt = os.path.join(REPO, "assets", "materialtest")
    camera = {"type": "pinhole", "resolution": [32, 18], "fov": 40, "transform": {"position": [0, 1, 5], "look_at": [0, 0.5, 0], "up": [0, 1, 0]}}
    common = {"camera": camera, "integrator": {"type": "path_tracer", "max_bounces": 4}, "renderer": {"spp": 1, "scene_bvh": True}}
    floor = {"type": "quad", "bsdf": "grey", "transform": {"scale": [4, 1, 4]}}
    case = 0

    def scene_with(files, bsdfs, primitives):
        nonlocal case
        d = tmp_path / f"case{case}"
        case += 1
        os.makedirs(d)
        for name, data in files.items():
            (d / name).write_bytes(data)
        (d / "scene.json").write_text(json.dumps(dict(common, bsdfs=bsdfs, primitives=primitives)))
        return str(d / "scene.json")

    grey = [{"name": "grey", "type": "lambert", "albedo": 0.5}]
    mesh = lambda data: scene_with({"m.wo3": data}, grey, [floor, {"type": "mesh", "file": "m.wo3", "bsdf": "grey", "smooth": True}])
    env = lambda data: scene_with({"e.hdr": data}, grey, [floor, {"type": "infinite_sphere", "emission": "e.hdr", "sample": True}])
    tex = lambda data: scene_with({"t.png": data}, [{"name": "grey", "type": "lambert", "albedo": "t.png"}], [floor])

    wo3 = open(os.path.join(mt, "Stand.wo3"), "rb").read()
    hdr = open(os.path.join(mt, "envmap.hdr"), "rb").read()
    yield mesh(wo3), "intact wo3 (control)"
    for cut in (0, 4, 8, 9, 100, len(wo3)//2, len(wo3) - 1):
        yield mesh(wo3[:cut]), f"wo3 truncated at {cut}"
    nv = struct.unpack_from("<Q", wo3, 0)[0]
    yield mesh(struct.pack("<Q", 1 << 60) + wo3[8:]), "wo3 vertex count 2^60"
    yield mesh(struct.pack("<Q", (1 << 32) - 1) + wo3[8:]), "wo3 vertex count 2^32-1"
    yield mesh(wo3[:8 + nv*32] + struct.pack("<Q", 1 << 59) + wo3[16 + nv*32:]), "wo3 triangle count 2^59"
    yield mesh(wo3[:8 + nv*32] + struct.pack("<Q", (1 << 32) - 1) + wo3[16 + nv*32:]), "wo3 triangle count 2^32-1"
    bad = bytearray(wo3)
    off = 16 + nv*32
    bad[off:off + 4] = struct.pack("<I", 0xFFFFFFF0)      # first triangle points outside the vertex array
    yield mesh(bytes(bad)), "wo3 vertex index out of range"
    yield env(hdr), "intact hdr (control)"
    for cut in (0, 10, 60, 200, len(hdr)//3, len(hdr) - 5):
        yield env(hdr[:cut]), f"hdr truncated at {cut}"
    for k in range(6):
        b = bytearray(hdr)
        for pos in rng.integers(0, len(b), 40):
            b[pos] = int(rng.integers(0, 256))
        yield env(bytes(b)), f"hdr with 40 random bytes (#{k})"
    assert b"-Y 512 +X 1024" in hdr
    yield env(hdr.replace(b"-Y 512 +X 1024", b"-Y 99999999 +X 99999999")), "hdr with an absurd resolution"
    yield env(hdr.replace(b"-Y 512 +X 1024", b"-Y 2000000000 +X 7")), "hdr with an absurd flat resolution"
    # PNG textures (assets/next/textures): truncated, corrupted deflate stream, lying IHDR
    tex_dir = os.path.join(REPO, "assets", "next", "textures")
    png_name = sorted(f for f in os.listdir(tex_dir) if f.endswith(".png"))[0]
    png = open(os.path.join(tex_dir, png_name), "rb").read()
    yield tex(png), "intact png (control)"
    for cut in (0, 8, 20, 33, 60, len(png)//2, len(png) - 13):
        yield tex(png[:cut]), f"png truncated at {cut}"
    for k in range(8):
        b = bytearray(png)
        for pos in rng.integers(33, len(b), 6):
            b[pos] = int(rng.integers(0, 256))
        yield tex(bytes(b)), f"png with 6 random bytes (#{k})"
    b = bytearray(png)
    b[16:24] = struct.pack(">II", 60000, 60000)               # IHDR width / height
    yield tex(bytes(b)), "png claiming 60000 x 60000"
    # .bmp / .tga (stb_image's formats): truncated, random bytes, absurd sizes
    bmp = open(os.path.join(tex_dir, "albedo.bmp"), "rb").read()
    tga = open(os.path.join(tex_dir, "rough.tga"), "rb").read()
    texb = lambda data: scene_with({"t.bmp": data}, [{"name": "grey", "type": "lambert", "albedo": "t.bmp"}], [floor])
    text = lambda data: scene_with({"t.tga": data}, [{"name": "grey", "type": "lambert", "albedo": "t.tga"}], [floor])
    yield texb(bmp), "intact bmp (control)"
    yield text(tga), "intact tga (control)"
    for cut in (0, 10, 30, 54, 200, len(bmp) - 7):
        yield texb(bmp[:cut]), f"bmp truncated at {cut}"
    for cut in (0, 5, 18, 100, len(tga) - 3):
        yield text(tga[:cut]), f"tga truncated at {cut}"
    yield texb(bmp[:18] + struct.pack("<ii", 1 << 30, 1 << 30) + bmp[26:]), "bmp claiming 2^30 x 2^30"
    yield texb(bmp[:10] + struct.pack("<I", 0xFFFFFFF0) + bmp[14:]), "bmp with a pixel offset beyond the file"
    yield text(tga[:12] + struct.pack("<HH", 65535, 65535) + tga[16:]), "tga claiming 65535 x 65535"
    for k in range(6):
        b = bytearray(bmp if k % 2 else tga)
        for pos in rng.integers(0, 60 if k % 2 else 18, 4):
            b[pos] = int(rng.integers(0, 256))
        yield (texb if k % 2 else text)(bytes(b)), f"{'bmp' if k % 2 else 'tga'} header with random bytes (#{k})"
    # .jpg, baseline and progressive: truncated (stb_image -- and so this loader -- feeds zero bits once the entropy-coded data ends, so a cut
    # inside the scan data still yields an image; a cut inside the headers does not), random bytes anywhere, a frame header claiming 65535 x 65535
    texj = lambda data: scene_with({"t.jpg": data}, [{"name": "grey", "type": "lambert", "albedo": "t.jpg"}], [floor])
    for jname in ("albedo422.jpg", "albedo_prog.jpg", "rough_prog_grey.jpg"):
        jpg = open(os.path.join(tex_dir, jname), "rb").read()
        yield texj(jpg), f"intact {jname} (control)"
        for cut in (0, 2, 30, 150, 400, len(jpg)//2, len(jpg) - 5):
            yield texj(jpg[:cut]), f"jpg {jname} truncated at {cut}"
        for k in range(12):
            b = bytearray(jpg)
            for pos in rng.integers(2, len(b), 5):
                b[pos] = int(rng.integers(0, 256))
            yield texj(bytes(b)), f"jpg {jname} with 5 random bytes (#{k})"
        sof = max(jpg.find(b"\xff\xc0"), jpg.find(b"\xff\xc2"))
        assert sof > 0
        yield texj(jpg[:sof + 5] + struct.pack(">HH", 65535, 65535) + jpg[sof + 9:]), f"{jname} claiming 65535 x 65535"
    # instance file (Instance.cpp:207-232): truncated, count blown up
    blob = struct.pack("<II6f", 3, 0, -1, -1, -1, 1, 1, 1) + struct.pack("<18f", *([0.1]*18)) + bytes([0, 0, 0])
    inst = lambda data: scene_with({"i.bin": data}, grey, [floor, {"type": "instances", "masters": [{"type": "cube", "bsdf": "grey", "transform": {"scale": 0.2}}],
                                                                  "instances": "i.bin"}])
    yield inst(blob), "intact instance file (control)"
    for data, what in ((blob[:7], "truncated header"), (blob[:40], "truncated body"), (struct.pack("<II", 1 << 30, 0) + blob[8:], "count 2^30"),
                       (struct.pack("<II", 0xFFFFFFFF, 1) + blob[8:], "count 2^32-1, lossy"), (blob[:-1], "missing id")):
        yield inst(data), f"instance file: {what}"


def test_damaged_inputs_are_refused_with_a_message(ta, tmp_path):
    """Truncated / corrupt .wo3, .hdr, .png and instance files must end in an error message, never in a crash or in memory the parsers do not
    own.  Every case runs through the ordinary host library here and, in one batch, through the ASan + UBSan build of the loader
    (`make asan`: build/asan/host_check), where an out-of-bounds read or an overflowing size computation aborts the process."""
    cases = list(_damaged_copies(tmp_path))
    assert len(cases) > 50
    refused = 0
    for path, what in cases:
        try:
            ta.HostScene(path)
            assert "control" in what or "random bytes" in what or what.endswith("- 3") or "tga truncated at 1064" in what or what.startswith("jpg "), f"{what}: loaded without complaint"   # a few byte flips leave a decodable file
        except RuntimeError as e:
            assert str(e) and "control" not in what, f"{what}: {e}"
            refused += 1
    assert refused >= len(cases) - 6 - 22 - 3*(1 + 7 + 12)
    r = subprocess.run(["make", "-s", "asan"], cwd=REPO, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([os.path.join(REPO, "build", "asan", "host_check")] + [p for p, _ in cases], capture_output=True, text=True, env=env)
    assert r.returncode == 0, "the sanitizer build died on a damaged input:\n" + r.stdout[-3000:] + r.stderr[-6000:]
    assert f"of {len(cases)} refused" in r.stdout
