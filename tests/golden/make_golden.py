#!/usr/bin/env python3
"""Regenerates the golden PFMs in this directory by running the COMPILED, UNMODIFIED REFERENCE.

Provenance (recorded in DESIGN.md "Oracle"): the renderer binary used here is the survey-phase
build of /root/reference (SURVEY.md section 8c recipe: cmake + ninja in /tmp, Embree 2.11 vendored,
g++ 11.4 -O3, no FMA) that was left at /tmp/tungsten-build/tungsten.  This repository does not and
cannot rebuild it (its Embree needs cmake-generated headers); the fixtures are data: float
framebuffers written by the reference's own ImageIO::saveHdr (.pfm).

Each fixture = (scene under assets/, resolution, spp).  Scenes are written next to the assets with
the resolution/spp/output names patched in, rendered with `tungsten -t 8`, and removed again.
The default seed 0xBA5EBA11 applies (Shared.hpp:246-248).
"""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF_BIN = os.environ.get("TUNGSTEN_REF_BIN", "/tmp/tungsten-build/tungsten")

FIXTURES = [
    # name, scene (relative to assets/), resolution, spp
    ("c2_128x72_1spp", "materialtest/c2.json", (128, 72), 1),
    ("c2_128x72_16spp", "materialtest/c2.json", (128, 72), 16),
    ("c2_128x72_48spp", "materialtest/c2.json", (128, 72), 48),
    ("c2_100x57_4spp", "materialtest/c2.json", (100, 57), 4),
    ("c1_128x72_4spp", "materialtest/c1_noadapt.json", (128, 72), 4),
    ("c1_128x72_32spp", "materialtest/c1_noadapt.json", (128, 72), 32),
    ("cornell_96x54_8spp", "cornell-box/scene_sobol.json", (96, 54), 8),
    # the stock scene WITH its adaptive sampling (4 passes of 16 spp: 1 uniform + 3 redistributed)
    # "instances" primitive: cube + quad masters, overlapping rotated instances (Instance.cpp's last-hit-wins traversal)
    ("inst_cubes_128x72_8spp", "instances/inst_cubes.json", (128, 72), 8),
    ("inst_many_96x54_4spp", "instances/inst_many.json", (96, 54), 4),    # 294 instances: binned-SAH levels of the BVH2
    # config 5: rough_dielectric boundary + homogeneous medium inside (volume scattering / absorption only)
    ("c5_128x72_16spp", "materialtest/c5.json", (128, 72), 16),
    ("c5_absorb_128x72_8spp", "materialtest/c5_absorb.json", (128, 72), 8),
    ("cornell_fog_96x54_8spp", "cornell-box/scene_fog.json", (96, 54), 8),   # camera medium + quad light: finite-segment transmittance
    # the fork's own example scene: four Cornell boxes behind forward-scattering walls, filled with linear / quadratic /
    # double-exponential / pulse media, lit by 'power' quads (data/example-scenes/non-exponential/scene.json, unmodified)
    ("nonexp_stock_128_16spp", "non-exponential/scene.json", (128, 128), 16, False, {}),                          # stock: uniform sampler
    ("nonexp_stock_sobol_128_16spp", "non-exponential/scene.json", (128, 128), 16, False, {"stratified_sampler": True}),
    ("nonexp_stock_deep_96_8spp", "non-exponential/scene.json", (96, 96), 8, False, {"stratified_sampler": True, "max_bounces": 16}),
    ("nonexp_stock_adaptive_128_48spp", "non-exponential/scene.json", (128, 128), 48, True, {}),
    ("c1_adaptive_128x72_64spp", "materialtest/materialtest.json", (128, 72), 64, True),
    # round 3: fixtures LONG enough to see rare events (one flipped decision desynchronises a tile's sequential PCG stream for good).
    # The first two contain the events that showed the oracle's reference mode needed Embree's traversal restated (oracle/embree_bvh4.h):
    # a ceiling quad culled by its own flat box (pixel (67,7) sample 41) and a floor / wall tie decided by visit order (pixel (47,33) sample 15)
    ("cornell_sobol_96x54_64spp", "cornell-box/scene_sobol.json", (96, 54), 64),
    ("cornell_sobol_64x36_128spp", "cornell-box/scene_sobol.json", (64, 36), 128),
    ("cornell_stock_96x54_48spp", "cornell-box/scene.json", (96, 54), 48),           # BASELINE config 3's scene as shipped (uniform sampler)
    ("c2_320x180_32spp", "materialtest/c2.json", (320, 180), 32),                    # two pixels of this one need Embree's first-found triangle
    ("c2_32x18_320spp", "materialtest/c2.json", (32, 18), 320),                      # sample indices >= 256: the second Sobol' block (SobolPathSampler.hpp:47-52)
    ("c5_128x72_64spp", "materialtest/c5.json", (128, 72), 64),
    ("inst_many_96x54_48spp", "instances/inst_many.json", (96, 54), 48),
    ("cornell_fog_96x54_64spp", "cornell-box/scene_fog.json", (96, 54), 64),
    # the LDR output path (Integrator::writeBuffers: Camera::tonemap, *255, clamp, PNG; Integrator.cpp:56-80, cameras/Tonemap.hpp:26-62): the PNG the
    # reference writes NEXT TO the PFM is kept for these, one per tonemap operator, so that PFM -> PNG can be checked byte for byte on the CPU
    ("ldr_texture2_128x72_4spp", "next/ldr_texture2.json", (128, 72), 4),   # .bmp / .tga bitmaps: the reference's stb_image route
    ("ldr_texture3_128x72_4spp", "next/ldr_texture3.json", (128, 72), 4),   # baseline .jpg bitmaps (4:2:0, 4:2:2 + restart intervals, 4:4:4, grey): stb_image again
    ("fog_nested_96x54_16spp", "next/fog_nested.json", (96, 54), 16),   # BSDFs made of BSDFs (mixed wall, rough_coat box) reached from inside a camera medium
    ("ldr_texture4_128x72_4spp", "next/ldr_texture4.json", (128, 72), 4),   # progressive .jpg bitmaps (4:2:0, 4:4:4 with partial blocks, grey with restart markers)
    ("tonemap_linear_48x27_4spp", "materialtest/c2.json", (48, 27), 4, False, {"tonemap": "linear", "keep_png": True}),
    ("tonemap_gamma_48x27_4spp", "materialtest/c2.json", (48, 27), 4, False, {"tonemap": "gamma", "keep_png": True}),
    ("tonemap_reinhard_48x27_4spp", "materialtest/c2.json", (48, 27), 4, False, {"tonemap": "reinhard", "keep_png": True}),
    ("tonemap_filmic_48x27_4spp", "materialtest/c2.json", (48, 27), 4, False, {"tonemap": "filmic", "keep_png": True}),
    ("tonemap_pbrt_48x27_4spp", "materialtest/c2.json", (48, 27), 4, False, {"tonemap": "pbrt", "keep_png": True}),
]


def main():
    if not os.path.exists(REF_BIN):
        sys.exit(f"reference binary {REF_BIN} not found; fixtures cannot be regenerated here")
    only = set(sys.argv[1:])
    for name, scene, res, spp, *rest in FIXTURES:
        adaptive = rest[:1]
        overrides = rest[1] if len(rest) > 1 else {}
        if only and name not in only:
            continue
        src = os.path.join(REPO, "assets", scene)
        if not os.path.exists(src):
            print("skip (scene missing):", name)
            continue
        d = json.load(open(src))
        d["camera"]["resolution"] = list(res)
        d["renderer"].update({"spp": spp, "output_file": name + ".png", "hdr_output_file": name + ".pfm",
                              "overwrite_output_files": True, "adaptive_sampling": bool(adaptive and adaptive[0])})
        if "max_bounces" in overrides:
            d["integrator"]["max_bounces"] = overrides["max_bounces"]
        if "stratified_sampler" in overrides:
            d["renderer"]["stratified_sampler"] = overrides["stratified_sampler"]
        if "tonemap" in overrides:
            d["camera"]["tonemap"] = overrides["tonemap"]
        tmp = os.path.join(os.path.dirname(src), f"_tmp_{name}.json")
        json.dump(d, open(tmp, "w"))
        try:
            r = subprocess.run([REF_BIN, "-t", "8", "-d", HERE, tmp], capture_output=True, text=True)
        finally:
            os.remove(tmp)
        if r.returncode != 0:
            sys.exit(r.stdout + r.stderr)
        png = os.path.join(HERE, name + ".png")
        if os.path.exists(png) and not overrides.get("keep_png"):
            os.remove(png)
        print("wrote", name + ".pfm")


if __name__ == "__main__":
    main()
