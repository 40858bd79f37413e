#!/usr/bin/env python3
"""Golden auxiliary output buffers (renderer "output_buffers": depth / normal / albedo / visibility, with two_buffer_variance and
sample_variance parts) rendered by the compiled, unmodified reference -- same provenance as make_golden.py: the survey-phase build at
/tmp/tungsten-build/tungsten, which this repo's recipe cannot redo.  Data only: float buffers + the scene under
assets/.  Scenes: next/aov.json (glass over a checker floor under an environment map: late records behind specular vertices, escaping
primary rays, emission-free albedo) and cornell-box/scene_sobol.json (quad light: emissive albedo, finite-light visibility)."""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF_BIN = os.environ.get("TUNGSTEN_REF_BIN", "/tmp/tungsten-build/tungsten")
OUT = os.path.join(HERE, "aov")

BUFFERS = [  # type, two_buffer_variance, sample_variance
    ("depth", False, True), ("normal", True, False), ("albedo", False, False), ("visibility", True, True)]
FIXTURES = [("aov", "next/aov.json", (128, 72), 4), ("cornell", "cornell-box/scene_sobol.json", (96, 54), 5)]


def main():
    if not os.path.exists(REF_BIN):
        sys.exit(f"reference binary {REF_BIN} not found")
    os.makedirs(OUT, exist_ok=True)
    for name, scene, res, spp in FIXTURES:
        src = os.path.join(REPO, "assets", scene)
        d = json.load(open(src))
        d["camera"]["resolution"] = list(res)
        d["renderer"].update({"spp": spp, "spp_step": spp, "output_file": name + ".png", "hdr_output_file": name + "_color.pfm", "overwrite_output_files": True,
                              "adaptive_sampling": False, "enable_resume_render": False, "checkpoint_interval": "0"})
        d["renderer"]["output_buffers"] = [{"type": t, "hdr_output_file": f"{name}_{t}.pfm", "two_buffer_variance": two, "sample_variance": var}
                                           for t, two, var in BUFFERS]
        tmp = os.path.join(os.path.dirname(src), f"_tmp_{name}.json")
        json.dump(d, open(tmp, "w"))
        try:
            r = subprocess.run([REF_BIN, "-t", "8", "-d", OUT, tmp], capture_output=True, text=True)
        finally:
            os.remove(tmp)
        png = os.path.join(OUT, name + ".png")
        if os.path.exists(png):
            os.remove(png)
        ok = r.returncode == 0 and os.path.exists(os.path.join(OUT, f"{name}_depth.pfm"))
        print(("wrote " if ok else "FAILED ") + name, "" if ok else (r.stdout + r.stderr)[-400:])


if __name__ == "__main__":
    main()
