#!/usr/bin/env python3
"""Upload and BVH build time of a flat 10 M-triangle soup (config 4's 1000 spheres baked into one mesh, assets/instances/make_c4.py --flat) through the
device-side builder (tungsten_amd/csrc/bvh_gpu.h), and what the Morton-order tree costs at render time.  Run on a GPU box from the repo root:
    python profiles/bvh_build_time.py [n_per_axis=10] [--host]        (--host also times the host builder on the same soup: tens of seconds)
Prints one JSON line."""
import json
import os
import subprocess
import sys
import tempfile
import time

sys.path.insert(0, os.getcwd())
import numpy as np
import tungsten_amd as ta

n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 10
with tempfile.TemporaryDirectory() as tmp:
    subprocess.run([sys.executable, "assets/instances/make_c4.py", tmp, str(n), "1280", "720", "--flat"], check=True, stdout=subprocess.DEVNULL)
    t0 = time.time()
    hs = ta.HostScene(os.path.join(tmp, "c4_flat.json"))
    load_s = time.time() - t0
    shard = ta.Shard(hs.res_x, hs.res_y)
    ctx = ta.Context(0)
    out = {"triangles": int(hs.desc.contents.n_tris), "host_load_and_flatten_s": round(load_s, 2)}

    def measure(label):
        ctx.upload(hs, shard)
        st = ctx.stats()
        ctx.set_profiling(3)
        ctx.render(16, spp_step=16)     # warm-up pass (first launches)
        ctx.reset_stats()
        t = time.time()
        ctx.render(32, spp_step=16, spp_begin=16)
        dt = time.time() - t
        s = ctx.stats()
        out[label] = {"bvh_on_device": int(st["bvh_on_device"]), "upload_ms": round(st["upload_ms"], 1), "bvh_build_ms": round(st["bvh_build_ms"], 1),
                      "msamples_per_s": round(hs.res_x*hs.res_y*32/dt/1e6, 1),
                      "extend_node_visits_per_ray": round(s["extend_node_visits"]/max(1, s["extend_rays"]), 2),
                      "extend_tri_tests_per_ray": round(s["extend_tri_tests"]/max(1, s["extend_rays"]), 2)}
        return ctx.download()

    dev = measure("device_builder")
    if "--host" in sys.argv:
        os.environ["TGPU_GPU_BVH"] = "0"
        host = measure("host_builder")
        out["images_identical"] = bool(np.array_equal(dev[0], host[0]) and np.array_equal(dev[1], host[1]))
    ctx.close()
print(json.dumps(out))
