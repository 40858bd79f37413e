#!/usr/bin/env python3
"""Regenerates tests/golden/ref_kat5.npz: what the COMPILED Embree 2.11 of the reference build does with boxes registered as one user
geometry (the way TraceableScene registers the scene's finites) and with triangle meshes (the way TriangleMesh registers itself).

Provenance: like the golden PFMs, the outputs come from the survey-phase build of the unmodified reference (its Embree static libraries in
/tmp/tungsten-build), through tests/tools/ref_embree_probe.cpp (build line in its header; pass the binary as argv[1] or
TUNGSTEN_EMBREE_PROBE).  Inputs are made here with a fixed seed and stored next to the outputs, so the test needs nothing but the file.

user cases  : per case boxes (n,6) and rays (r, 8 + n) = org, dir, tnear, tfar, scripted hit distance per box (negative = miss);
              recorded: the order in which Embree asks the primitives, the primitive it ends on and the final tfar.
tri cases   : verts, tris, rays (r, 8); recorded: primID, t, u, v.
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
PROBE = sys.argv[1] if len(sys.argv) > 1 else os.environ.get("TUNGSTEN_EMBREE_PROBE", "/tmp/r3/ref_embree_probe")
f32 = np.float32


def unit(v):
    v = np.asarray(v, np.float64)
    return (v/np.linalg.norm(v, axis=-1, keepdims=True)).astype(f32)


def box_entry_exit(boxes, org, d):
    """float32 slab distances of every ray against every box: (r, n) entry and exit (entry > exit = miss)."""
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = (f32(1.0)/d)[:, None, :]
        a = (boxes[None, :, 0:3] - org[:, None, :])*inv
        b = (boxes[None, :, 3:6] - org[:, None, :])*inv
    lo = np.fmax.reduce(np.minimum(a, b), axis=2)
    hi = np.fmin.reduce(np.maximum(a, b), axis=2)
    return lo.astype(f32), hi.astype(f32)


def scripted_from_boxes(boxes, org, d, rng, miss_fraction=0.2):
    """A primitive 'is hit' where its box is entered (flat boxes: exactly the plane distance, as for a Quad) unless drawn as a miss."""
    lo, hi = box_entry_exit(boxes, org, d)
    t = np.where((lo <= hi) & (hi > 0), np.maximum(lo, 0), -1).astype(f32)
    t[rng.random(t.shape) < miss_fraction] = -1
    return t


def cornell_boxes():
    """The eight finites of data/example-scenes/cornell-box (5 walls, 2 boxes, light) as axis-aligned boxes; the two cubes by their rotated corners."""
    def quad(center, ex, ez):
        c, ex, ez = np.array(center, f32), np.array(ex, f32), np.array(ez, f32)
        pts = np.stack([c + sx*ex*f32(0.5) + sz*ez*f32(0.5) for sx in (-1, 1) for sz in (-1, 1)])
        return np.concatenate([pts.min(0), pts.max(0)])

    def cube(pos, half, deg):
        a = np.deg2rad(deg)
        rot = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
        pts = np.array([[sx*half[0], sy*half[1], sz*half[2]] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)]) @ rot.T + np.array(pos)
        pts = pts.astype(f32)
        return np.concatenate([pts.min(0), pts.max(0)])

    return np.stack([
        quad([0, 0, 0], [2, 0, 0], [0, 0, 2]), quad([0, 2, 0], [2, 0, 0], [0, 0, 2]), quad([0, 1, -1], [2, 0, 0], [0, 2, 0]),
        quad([1, 1, 0], [0, 2, 0], [0, 0, 2]), quad([-1, 1, 0], [0, 2, 0], [0, 0, 2]),
        cube([0.328631, 0.3, 0.374592], [0.3, 0.3, 0.3], -16.64), cube([-0.335439, 0.6, -0.291415], [0.3, 0.6, 0.3], 19.188),
        quad([-0.005, 1.98, -0.03], [0.47, 0, 0], [0, 0, 0.38]),
    ]).astype(f32)


def edge_rays(boxes, rng, n):
    """Rays from in front of the room to points on the room's edges and corners (+- a few ulps): two flat boxes reached at (nearly) one distance."""
    org = np.tile(np.array([[0, 1, 6.8]], f32), (n, 1)) + (rng.random((n, 3))*f32(0.2) - f32(0.1)).astype(f32)*np.array([1, 1, 0], f32)
    tgt = np.empty((n, 3), f32)
    for i in range(n):
        kind = rng.integers(0, 6)
        s = f32(rng.random()*2 - 1)
        if kind == 0: tgt[i] = (1, 2, s)           # ceiling / right wall
        elif kind == 1: tgt[i] = (-1, 2, s)        # ceiling / left wall
        elif kind == 2: tgt[i] = (1, 0, s)         # floor / right wall
        elif kind == 3: tgt[i] = (-1, 0, s)        # floor / left wall
        elif kind == 4: tgt[i] = (s, 2, -1)        # ceiling / back wall
        else: tgt[i] = (s, 0, -1)                  # floor / back wall
    d = unit(tgt - org)
    bump = rng.integers(-2, 3, size=(n, 3)).astype(np.int32)
    d = (d.view(np.int32) + bump).view(f32)  # a few ulps either way
    return org, d


def user_case(boxes, org, d, scripted, tnear=f32(1e-4)):
    r = len(org)
    rays = np.empty((r, 8 + len(boxes)), f32)
    rays[:, 0:3] = org; rays[:, 3:6] = d; rays[:, 6] = tnear; rays[:, 7] = np.inf
    rays[:, 8:] = scripted
    return boxes.astype(f32), rays


def run_user(boxes, rays):
    with tempfile.TemporaryDirectory() as tmp:
        fin, fout = os.path.join(tmp, "in.bin"), os.path.join(tmp, "out.bin")
        with open(fin, "wb") as f:
            f.write(np.array([len(boxes), len(rays)], np.uint32).tobytes()); f.write(boxes.tobytes()); f.write(rays.tobytes())
        subprocess.run([PROBE, "user", fin, fout], check=True)
        raw = np.fromfile(fout, np.uint32)
    visits, offsets, prim, tfar = [], [0], [], []
    p = 0
    for _ in range(len(rays)):
        nv = int(raw[p]); p += 1
        visits.extend(raw[p:p + nv].tolist()); p += nv
        offsets.append(len(visits))
        prim.append(raw[p]); tfar.append(raw[p + 1]); p += 2
    return (np.array(visits, np.uint32), np.array(offsets, np.uint32), np.array(prim, np.uint32), np.array(tfar, np.uint32).view(f32))


def run_tris(verts, tris, rays):
    with tempfile.TemporaryDirectory() as tmp:
        fin, fout = os.path.join(tmp, "in.bin"), os.path.join(tmp, "out.bin")
        with open(fin, "wb") as f:
            f.write(np.array([len(verts), len(tris), len(rays)], np.uint32).tobytes())
            f.write(verts.tobytes()); f.write(tris.tobytes()); f.write(rays.tobytes())
        subprocess.run([PROBE, "tris", fin, fout], check=True)
        raw = np.fromfile(fout, np.uint32).reshape(len(rays), 4)
    return raw[:, 0].copy(), raw[:, 1:4].copy().view(f32)


def uv_sphere(nu, nv, radius=1.0):
    verts = [(0.0, radius, 0.0)]
    for j in range(1, nv):
        th = np.pi*j/nv
        for i in range(nu):
            ph = 2*np.pi*i/nu
            verts.append((radius*np.sin(th)*np.cos(ph), radius*np.cos(th), radius*np.sin(th)*np.sin(ph)))
    verts.append((0.0, -radius, 0.0))
    tris = []
    for i in range(nu):
        tris.append((0, 1 + i, 1 + (i + 1) % nu))
    for j in range(nv - 2):
        for i in range(nu):
            a, b = 1 + j*nu + i, 1 + j*nu + (i + 1) % nu
            c, d = a + nu, b + nu
            tris.append((a, c, b)); tris.append((b, c, d))
    last = len(verts) - 1
    base = 1 + (nv - 2)*nu
    for i in range(nu):
        tris.append((last, base + (i + 1) % nu, base + i))
    return np.array(verts, f32), np.array(tris, np.uint32)


def read_wo3(path):
    raw = open(path, "rb").read()
    nv = int(np.frombuffer(raw, np.uint64, 1, 0)[0])
    v = np.frombuffer(raw, f32, nv*8, 8).reshape(nv, 8)[:, 0:3].copy()
    off = 8 + nv*32
    nt = int(np.frombuffer(raw, np.uint64, 1, off)[0])
    t = np.frombuffer(raw, np.uint32, nt*4, off + 8).reshape(nt, 4)[:, 0:3].copy()
    return v, t


def mesh_rays(verts, tris, rng, n_random, n_feature):
    c = verts.mean(0)
    ext = float(np.abs(verts - c).max())
    org = (c + unit(rng.normal(size=(n_random, 3)))*f32(3*ext)).astype(f32)
    tgt = (c + (rng.random((n_random, 3))*2 - 1).astype(f32)*f32(0.8*ext)).astype(f32)
    rays = [np.concatenate([org, unit(tgt - org)], 1)]
    # rays aimed exactly at vertices and edge midpoints: several triangles reached at one distance
    pick = rng.integers(0, len(tris), n_feature)
    origins = (c + unit(rng.normal(size=(8, 3)))*f32(3*ext)).astype(f32)
    o = origins[rng.integers(0, 8, n_feature)]
    t = tris[pick]
    which = rng.integers(0, 3, n_feature)
    v0 = verts[t[np.arange(n_feature), which]]
    v1 = verts[t[np.arange(n_feature), (which + 1) % 3]]
    mid = ((v0 + v1)*f32(0.5)).astype(f32)
    tg = np.where((rng.random(n_feature) < 0.5)[:, None], v0, mid)
    rays.append(np.concatenate([o, unit(tg - o)], 1))
    rays = np.concatenate(rays).astype(f32)
    out = np.empty((len(rays), 8), f32)
    out[:, 0:6] = rays; out[:, 6] = f32(1e-4); out[:, 7] = np.inf
    return out


def main():
    if not os.path.exists(PROBE):
        sys.exit(f"{PROBE} not found: build tests/tools/ref_embree_probe.cpp (see its header)")
    rng = np.random.default_rng(20260305)
    store = {}
    # ---- user geometry ----
    cases = []
    cb = cornell_boxes()
    org, d = edge_rays(cb, rng, 1500)
    scripted = scripted_from_boxes(cb, org, d, rng, 0.0)
    cases.append(user_case(cb, org, d, scripted))                                  # room edges: near-ties from the slab arithmetic itself
    forced = scripted.copy()                                                       # exact ties: every entered flat box reports the SAME distance
    flat = (cb[:, 0:3] == cb[:, 3:6]).any(1)
    for i in range(len(forced)):
        hit = np.nonzero((forced[i] >= 0) & flat)[0]
        if len(hit) >= 2: forced[i, hit] = forced[i, hit].max()
    cases.append(user_case(cb, org, d, forced))
    org = (rng.random((600, 3))*2 - 1).astype(f32)*np.array([0.9, 0.9, 0.9], f32) + np.array([0, 1, 0], f32)   # rays starting inside the room
    d = unit(rng.normal(size=(600, 3)))
    cases.append(user_case(cb, org, d, scripted_from_boxes(cb, org, d, rng, 0.3), tnear=f32(5e-4)))
    for n, r in ((1, 50), (2, 200), (3, 200), (5, 300), (9, 300), (17, 300), (40, 300), (150, 200), (300, 150)):
        lo = (rng.random((n, 3))*(6 if n < 50 else 9)).astype(f32)
        size = (rng.random((n, 3))*(4.0 if n < 50 else 2.0)).astype(f32)
        size[rng.random((n, 3)) < 0.15] = 0                                           # flat boxes
        boxes = np.concatenate([lo, lo + size], 1).astype(f32)
        org = (rng.random((r, 3))*30 - 10).astype(f32)
        tgt = (rng.random((r, 3))*10).astype(f32)
        d = unit(tgt - org)
        cases.append(user_case(boxes, org, d, scripted_from_boxes(boxes, org, d, rng, 0.25)))
        if n > 1: cases.append(user_case(boxes, org, d, np.full((r, n), -1, f32)))    # nothing hits: the whole front-to-back order of the entered leaves
    conc = np.array([[-s, -s, -s, s, s, s] for s in (1, 2, 3, 4, 5, 6, 7)], f32)     # one centroid: the median fall-back of the builder
    org = (unit(rng.normal(size=(200, 3)))*f32(12)).astype(f32)
    d = unit(-org + rng.normal(size=(200, 3)))
    cases.append(user_case(conc, org, d, scripted_from_boxes(conc, org, d, rng, 0.3)))
    store["n_user"] = np.array(len(cases))
    for i, (boxes, rays) in enumerate(cases):
        visits, offsets, prim, tfar = run_user(boxes, rays)
        store[f"user{i}_boxes"] = boxes; store[f"user{i}_rays"] = rays
        store[f"user{i}_visits"] = visits; store[f"user{i}_offsets"] = offsets; store[f"user{i}_prim"] = prim; store[f"user{i}_tfar"] = tfar
        print(f"user case {i}: {len(boxes)} boxes, {len(rays)} rays, {len(visits)} visits, {(prim != 0xFFFFFFFF).sum()} hits")
    # ---- triangle meshes ----
    meshes = [uv_sphere(64, 40), uv_sphere(7, 5), read_wo3(os.path.join(HERE, "..", "..", "assets", "materialtest", "Stand.wo3"))]
    grid = np.array([(x, 0, z) for z in range(9) for x in range(9)], f32)             # a flat 8x8 grid: every interior vertex shared by six coplanar triangles
    gt = []
    for z in range(8):
        for x in range(8):
            a = z*9 + x
            gt += [(a, a + 9, a + 1), (a + 1, a + 9, a + 10)]
    meshes.append((grid, np.array(gt, np.uint32)))
    store["n_tri"] = np.array(len(meshes))
    for i, (verts, tris) in enumerate(meshes):
        rays = mesh_rays(verts, tris, rng, 2500 if len(tris) > 500 else 600, 2500 if len(tris) > 500 else 900)
        prim, tuv = run_tris(verts, tris, rays)
        store[f"tri{i}_verts"] = verts; store[f"tri{i}_tris"] = tris; store[f"tri{i}_rays"] = rays
        store[f"tri{i}_prim"] = prim; store[f"tri{i}_tuv"] = tuv
        print(f"tri case {i}: {len(tris)} triangles, {len(rays)} rays, {(prim != 0xFFFFFFFF).sum()} hits")
    np.savez_compressed(os.path.join(HERE, "ref_kat5.npz"), **store)
    print("wrote ref_kat5.npz", os.path.getsize(os.path.join(HERE, "ref_kat5.npz")), "bytes")


if __name__ == "__main__":
    main()
