#!/usr/bin/env python3
"""Generates config 4 of BASELINE.json / SURVEY.md section 8d: a procedural 10 M-triangle instanced scene.

  master  = closed UV sphere of exactly 10 000 triangles (100 slices x 51 stacks), unit radius, smooth normals
  lattice = 10 x 10 x 10 instances, spacing 2.5, jitter +-0.5, random rotations, all from PCG32(seed 1234)
  two masters (same mesh): Lambert 0.5 and rough_conductor Cu / ggx / 0.2, alternating
  checker floor quad, constant white infinite_sphere (emission 1), pinhole fov 35 looking at the lattice centre

Written in the reference's own `"type": "instances"` JSON (Instance.cpp:60-93).  NOTE: the reference's command
line renderer cannot render file-backed mesh masters (Instance::fromJson instantiates them privately and nobody
calls loadResources() on them: the master stays empty and TriangleMesh::intersect dereferences a null Embree
scene -- DESIGN.md section 7b), so for this scene the oracle, not a reference image, is the checker.

usage: make_c4.py OUTDIR [n_per_axis=10] [res_x res_y] [--flat]

--flat additionally writes c4_flat.json: the same spheres baked into ONE mesh (n^3 x 10 000 triangles in c4_flat.wo3, two materials through the
triangles' material index) -- the flat soup the device-side BVH builder (tungsten_amd/csrc/bvh_gpu.h) exists for.  Needs numpy.
"""
import json
import math
import os
import struct
import sys


class Pcg32:  # sampling/UniformSampler.hpp:40-52
    def __init__(self, seed, seq=0):
        self.state, self.inc = seed & (2**64 - 1), (seq | 1) & (2**64 - 1)

    def next_u32(self):
        old = self.state
        self.state = (old*6364136223846793005 + self.inc) & (2**64 - 1)
        xs = (((old >> 18) ^ old) >> 27) & 0xFFFFFFFF
        rot = old >> 59
        return ((xs >> rot) | (xs << ((-rot) & 31))) & 0xFFFFFFFF

    def next_float(self):
        return (self.next_u32() >> 8)/float(1 << 24)


def write_sphere(path, slices=100, stacks=51):
    verts, tris = [], []
    verts.append((0.0, 1.0, 0.0, 0.0, 1.0, 0.0, 0.5, 0.0))
    for st in range(1, stacks):
        theta = math.pi*st/stacks
        for sl in range(slices):
            phi = 2*math.pi*sl/slices
            x, y, z = math.sin(theta)*math.cos(phi), math.cos(theta), math.sin(theta)*math.sin(phi)
            verts.append((x, y, z, x, y, z, sl/slices, st/stacks))
    verts.append((0.0, -1.0, 0.0, 0.0, -1.0, 0.0, 0.5, 1.0))
    south = len(verts) - 1
    ring = lambda st, sl: 1 + (st - 1)*slices + (sl % slices)
    for sl in range(slices):
        tris.append((0, ring(1, sl + 1), ring(1, sl), 0))
        tris.append((south, ring(stacks - 1, sl), ring(stacks - 1, sl + 1), 0))
    for st in range(1, stacks - 1):
        for sl in range(slices):
            a, b, c, d = ring(st, sl), ring(st, sl + 1), ring(st + 1, sl), ring(st + 1, sl + 1)
            tris.append((a, b, c, 0))
            tris.append((b, d, c, 0))
    with open(path, "wb") as f:  # .wo3 (io/MeshIO.cpp:12-42): u64 count + 32 B vertices, u64 count + 16 B triangles
        f.write(struct.pack("<Q", len(verts)))
        for v in verts:
            f.write(struct.pack("<8f", *v))
        f.write(struct.pack("<Q", len(tris)))
        for t in tris:
            f.write(struct.pack("<3Ii", *t))
    return len(verts), len(tris)


def write_flat(out, instances, scene):
    import numpy as np
    raw = open(os.path.join(out, "sphere10k.wo3"), "rb").read()
    nv = struct.unpack_from("<Q", raw, 0)[0]
    verts = np.frombuffer(raw, np.float32, nv*8, 8).reshape(nv, 8)
    nt = struct.unpack_from("<Q", raw, 8 + nv*32)[0]
    tris = np.frombuffer(raw, np.uint32, nt*4, 16 + nv*32).reshape(nt, 4)
    all_v = np.empty((len(instances), nv, 8), np.float32)
    all_t = np.empty((len(instances), nt, 4), np.uint32)
    for k, inst in enumerate(instances):
        rx, ry, rz = (math.radians(a) for a in inst["transform"]["rotation"])
        cx, sx, cy, sy, cz, sz = math.cos(rx), math.sin(rx), math.cos(ry), math.sin(ry), math.cos(rz), math.sin(rz)
        rot = (np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]]) @ np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]]) @ np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])).astype(np.float32)
        all_v[k, :, 0:3] = verts[:, 0:3] @ rot.T + np.array(inst["transform"]["position"], np.float32)
        all_v[k, :, 3:6] = verts[:, 3:6] @ rot.T
        all_v[k, :, 6:8] = verts[:, 6:8]
        all_t[k, :, 0:3] = tris[:, 0:3] + k*nv
        all_t[k, :, 3] = inst["id"]
    with open(os.path.join(out, "c4_flat.wo3"), "wb") as f:
        f.write(struct.pack("<Q", len(instances)*nv)); all_v.tofile(f)
        f.write(struct.pack("<Q", len(instances)*nt)); all_t.tofile(f)
    flat = json.loads(json.dumps(scene))
    flat["primitives"][0] = {"type": "mesh", "file": "c4_flat.wo3", "smooth": True, "bsdf": ["diffuse", "metal"], "transform": {}}
    flat["renderer"].update({"output_file": "c4_flat.png", "hdr_output_file": "c4_flat.pfm"})
    with open(os.path.join(out, "c4_flat.json"), "w") as f:
        json.dump(flat, f)
    print(f"wrote {out}/c4_flat.json: one mesh of {len(instances)*nt/1e6:.2f} M triangles")


def main():
    flat = "--flat" in sys.argv
    sys.argv = [a for a in sys.argv if a != "--flat"]
    out = sys.argv[1]
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    res = [int(sys.argv[3]), int(sys.argv[4])] if len(sys.argv) > 4 else [1280, 720]
    os.makedirs(out, exist_ok=True)
    nv, nt = write_sphere(os.path.join(out, "sphere10k.wo3"))
    assert nt == 10000, nt
    rng = Pcg32(1234)
    instances = []
    half = (n - 1)*2.5/2
    for ix in range(n):
        for iy in range(n):
            for iz in range(n):
                p = [ix*2.5 - half + (rng.next_float() - 0.5), iy*2.5 + 1.5 + (rng.next_float() - 0.5), iz*2.5 - half + (rng.next_float() - 0.5)]
                r = [rng.next_float()*360 - 180, rng.next_float()*360, rng.next_float()*360 - 180]
                instances.append({"id": (ix + iy + iz) & 1, "transform": {"position": [round(v, 4) for v in p], "rotation": [round(v, 2) for v in r]}})
    centre = [0.0, (n - 1)*2.5/2 + 1.5, 0.0]
    dist = 3.6*half + 14
    scene = {
        "media": [],
        "bsdfs": [
            {"name": "diffuse", "albedo": 0.5, "type": "lambert"},
            {"name": "metal", "albedo": 1.0, "type": "rough_conductor", "material": "Cu", "distribution": "ggx", "roughness": 0.2},
            {"name": "floor", "type": "lambert", "albedo": {"type": "checker", "on_color": [0.725, 0.71, 0.68], "off_color": [0.325, 0.31, 0.25], "res_u": 40, "res_v": 40}}],
        "primitives": [
            {"type": "instances", "transform": {},
             "masters": [{"type": "mesh", "file": "sphere10k.wo3", "smooth": True, "bsdf": "diffuse", "transform": {}},
                         {"type": "mesh", "file": "sphere10k.wo3", "smooth": True, "bsdf": "metal", "transform": {}}],
             "instances": instances},
            {"type": "quad", "bsdf": "floor", "transform": {"position": [0, 0, 0], "scale": [8*half + 40, 1, 8*half + 40]}},
            {"type": "infinite_sphere", "emission": 1.0, "sample": True, "transform": {}}],
        "camera": {"tonemap": "filmic", "resolution": res, "reconstruction_filter": "tent", "type": "pinhole", "fov": 35,
                   "transform": {"position": [dist*0.62, centre[1] + dist*0.45, dist*0.64], "look_at": centre, "up": [0, 1, 0]}},
        "integrator": {"type": "path_tracer", "min_bounces": 0, "max_bounces": 64, "enable_consistency_checks": False,
                       "enable_two_sided_shading": True, "enable_light_sampling": True, "enable_volume_light_sampling": True},
        "renderer": {"output_file": "c4.png", "hdr_output_file": "c4.pfm", "overwrite_output_files": True, "adaptive_sampling": False,
                     "enable_resume_render": False, "stratified_sampler": True, "scene_bvh": True, "spp": 512, "spp_step": 16}}
    with open(os.path.join(out, "c4.json"), "w") as f:
        json.dump(scene, f)
    print(f"wrote {out}/c4.json: {len(instances)} instances x {nt} triangles = {len(instances)*nt/1e6:.1f} M instanced triangles")
    if flat:
        write_flat(out, instances, scene)


if __name__ == "__main__":
    main()
