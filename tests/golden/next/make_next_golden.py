#!/usr/bin/env python3
"""Golden images for rows of SURVEY.md section 8f that are NOT built yet (other stock BSDFs, Sphere/Disk, Henyey-
Greenstein): rendered by the compiled,
unmodified reference while it is available (same provenance as ../make_golden.py), so that the oracle of a later round
can be pinned even if the reference binary is gone.  Data only: float framebuffers + the scene files under assets/.
"""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
REF_BIN = os.environ.get("TUNGSTEN_REF_BIN", "/tmp/tungsten-build/tungsten")
OUT = os.path.dirname(HERE)  # tests/golden/: the rows these fixtures belong to are built by now

FIXTURES = [  # name, scene, resolution, spp, renderer overrides, integrator overrides
] + [(f"{n}_128x72_4spp", f"next/{n}.json", (128, 72), 4, {"adaptive_sampling": False}, {}) for n in (
    "dielectric", "mirror", "conductor", "plastic", "rough_plastic", "rough_coat", "thinsheet", "oren_nayar", "phong",
    "transparency", "mixed", "dielectric_hg_medium", "sphere_disk")] + [
    # the remaining transmittance models of the fork (transmittances/TransmittanceFactory.cpp:14-24) in the Cornell box's camera medium
    (f"{n}_96x54_8spp", f"next/{n}.json", (96, 54), 8, {"adaptive_sampling": False}, {}) for n in (
    "trans_erlang", "trans_davis", "trans_davis_weinstein", "trans_interpolated", "trans_interpolated2")] + [
    # 8-bit PNG bitmaps (assets/next/textures/): gamma-corrected RGB albedo, palette albedo without interpolation (clamped, scaled), scalar roughness
    ("ldr_texture_128x72_4spp", "next/ldr_texture.json", (128, 72), 4, {"adaptive_sampling": False}, {})]


def main():
    if not os.path.exists(REF_BIN):
        sys.exit(f"reference binary {REF_BIN} not found")
    only = set(sys.argv[1:])
    for name, scene, res, spp, rover, iover in FIXTURES:
        if only and name not in only:
            continue
        src = os.path.join(REPO, "assets", scene)
        d = json.load(open(src))
        d["camera"]["resolution"] = list(res)
        d["renderer"].update({"spp": spp, "output_file": name + ".png", "hdr_output_file": name + ".pfm", "overwrite_output_files": True})
        d["renderer"].update(rover)
        d["integrator"].update(iover)
        tmp = os.path.join(os.path.dirname(src), f"_tmp_{name}.json")
        json.dump(d, open(tmp, "w"))
        try:
            r = subprocess.run([REF_BIN, "-t", "8", "-d", OUT, tmp], capture_output=True, text=True)
        finally:
            os.remove(tmp)
        png = os.path.join(OUT, name + ".png")
        if os.path.exists(png):
            os.remove(png)
        ok = r.returncode == 0 and os.path.exists(os.path.join(OUT, name + ".pfm"))
        print(("wrote " if ok else "FAILED ") + name, "" if ok else (r.stdout + r.stderr)[-300:])


if __name__ == "__main__":
    main()

# RenderState_c2_64x36_32spp_step16.{dat,pfm}: the reference's resume file (Integrator::saveRenderResumeData,
# integrators/Integrator.cpp:108-129) and final image of assets/materialtest/c2.json at 64x36, spp 32, spp_step 16,
# "enable_resume_render": true, "checkpoint_interval": "0" -- fixture for the resume-file row.
