"""The oracle's restatement of Embree 2.11's BVH4 builder and single-ray traversal (oracle/embree_bvh4.h: the part of the reference that decides
which primitive TraceableScene::intersect reports where binary32 leaves room) against the COMPILED library of the reference build:
tests/golden/ref_kat5.npz holds, for box sets registered as one user geometry, the order in which Embree asks the primitives and what it ends on
(scripted hit distances, ties included), and for triangle meshes (primID, t, u, v) of rays aimed at faces, edges and vertices.  Recorded by
tests/tools/ref_embree_probe.cpp through tests/golden/make_ref_kat5.py.  Everything must agree bit for bit."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import GOLDEN

KAT = np.load(os.path.join(GOLDEN, "ref_kat5.npz"))
needs_rcpps = pytest.mark.skipif(not hasattr(os, "uname") or os.uname().machine != "x86_64", reason="Embree's rcp() is RCPPS + Newton: x86 only")


@needs_rcpps
@pytest.mark.parametrize("case", range(int(KAT["n_user"])))
def test_user_geometry_visit_order(oracle_py, case):
    boxes, rays = KAT[f"user{case}_boxes"], KAT[f"user{case}_rays"]
    want_visits, offsets = KAT[f"user{case}_visits"], KAT[f"user{case}_offsets"]
    n, r = len(boxes), len(rays)
    cap = max(int(np.diff(offsets.astype(np.int64)).max()), 1) + 4
    visits = np.full((r, cap), 0xFFFFFFFF, np.uint32)
    n_visits = np.zeros(r, np.uint32)
    prim = np.zeros(r, np.uint32)
    tfar = np.zeros(r, np.float32)
    oracle_py.lib().oracle_kat_embree_user(boxes.ctypes.data, n, rays.ctypes.data, r, visits.ctypes.data, cap, n_visits.ctypes.data, prim.ctypes.data,
                                           tfar.ctypes.data)
    assert np.array_equal(n_visits, np.diff(offsets.astype(np.int64)).astype(np.uint32)), "number of primitives asked differs"
    got = np.concatenate([visits[i, :n_visits[i]] for i in range(r)]) if r else np.zeros(0, np.uint32)
    assert np.array_equal(got, want_visits), "visit order differs"
    assert np.array_equal(prim, KAT[f"user{case}_prim"])
    assert np.array_equal(tfar.view(np.uint32), KAT[f"user{case}_tfar"].view(np.uint32))


@needs_rcpps
@pytest.mark.parametrize("case", range(int(KAT["n_tri"])))
def test_triangle_mesh_hits(oracle_py, case):
    verts, tris, rays = KAT[f"tri{case}_verts"], KAT[f"tri{case}_tris"], KAT[f"tri{case}_rays"]
    prim = np.zeros(len(rays), np.uint32)
    tuv = np.zeros((len(rays), 3), np.float32)
    oracle_py.lib().oracle_kat_embree_tris(verts.ctypes.data, len(verts), tris.ctypes.data, len(tris), rays.ctypes.data, len(rays), prim.ctypes.data,
                                           tuv.ctypes.data)
    want_prim, want_tuv = KAT[f"tri{case}_prim"], KAT[f"tri{case}_tuv"]
    assert (want_prim != 0xFFFFFFFF).sum() > len(rays)//2
    assert np.array_equal(prim, want_prim), f"{(prim != want_prim).sum()} of {len(rays)} rays end on another triangle"
    hit = want_prim != 0xFFFFFFFF
    assert np.array_equal(tuv[hit].view(np.uint32), want_tuv[hit].view(np.uint32))
