"""Ray-level probe of config 4 (10 M instanced triangles): random closest-hit / any-hit rays, GPU vs oracle.
Round 1: 6.3 M rays, 0 mismatches.  Run on a GPU box from the repo root: python tests/tools/ray_probe_c4.py"""
import os, sys, subprocess, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "oracle")
import numpy as np
import tungsten_amd as ta
import oracle_py
os.makedirs("gpurun_out/c4", exist_ok=True)
subprocess.run([sys.executable, "assets/instances/make_c4.py", "gpurun_out/c4"], check=True, stdout=subprocess.DEVNULL)
hs = ta.HostScene("gpurun_out/c4/c4.json", resolution=(64, 36))
ctx = ta.Context(0)
ctx.upload(hs, ta.Shard(hs.res_x, hs.res_y), 1 << 20)
o = oracle_py.Oracle(hs, **oracle_py.GPU)
rng = np.random.default_rng(11)
total = 0; badc = 0; bado = 0
for rep in range(6):
    n = 1 << 20
    # rays between points in and around the lattice (extent about +-14): many graze spheres and box edges
    org = rng.uniform(-16, 16, (n, 3)).astype(np.float32)
    tgt = rng.uniform(-14, 14, (n, 3)).astype(np.float32)
    d = tgt - org
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.concatenate([org, np.full((n, 1), 1e-4, np.float32), d.astype(np.float32), np.full((n, 1), np.inf, np.float32)], axis=1)
    got, ref = ctx.trace_closest(rays), o.trace_closest(rays)
    bad = np.nonzero((got.view(np.uint32) != ref.view(np.uint32)).any(axis=1))[0]
    total += n; badc += len(bad)
    for i in bad[:2]:
        print("closest mismatch ray", rays[i].tolist(), "gpu", got[i].tolist(), hex(int(got[i, 3:4].view(np.uint32)[0])), "oracle", ref[i].tolist(), hex(int(ref[i, 3:4].view(np.uint32)[0])), flush=True)
    rays[:, 7] = rng.uniform(0.5, 30.0, n).astype(np.float32)
    go, ro = ctx.trace_occluded(rays), o.trace_occluded(rays)
    b2 = np.nonzero(go != ro)[0]
    bado += len(b2)
    for i in b2[:2]:
        print("occluded mismatch ray", rays[i].tolist(), "gpu", int(go[i]), "oracle", int(ro[i]), flush=True)
    print(f"rep {rep}: {total} rays, closest mismatches {badc}, occluded mismatches {bado}", flush=True)
ctx.close()
