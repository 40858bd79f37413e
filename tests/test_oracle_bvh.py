"""The oracle's own BVH walk against its brute-force mode (every triangle tested): the nearest hit is DEFINED as the lexicographic
minimum of (t, triangle id) over all triangles whose binary32 test passes (include/tgpu_bounds.h); a BVH may only prune.
CPU only; the GPU walk is checked against the same definition in tests/test_gpu_parity.py."""
import numpy as np

from conftest import edge_and_vertex_rays, scene_path


def test_oracle_bvh_walk_equals_brute_force_on_edge_and_vertex_rays(ta, oracle_py):
    hs = ta.HostScene(scene_path("materialtest/c2.json"), resolution=(32, 32))
    walk = oracle_py.Oracle(hs, **oracle_py.GPU)
    brute = oracle_py.Oracle(hs, **oracle_py.GPU)
    brute.set_brute_force(True)
    rays = edge_and_vertex_rays(hs, 6000, 5)
    a, b = walk.trace_closest(rays), brute.trace_closest(rays)
    bad = (a.view(np.uint32) != b.view(np.uint32)).any(axis=1)
    assert not bad.any(), f"{int(bad.sum())} of {len(rays)} rays: the BVH walk lost a triangle whose test passes"
    hit = b[:, 3].view(np.uint32) != 0xFFFFFFFF
    assert hit.mean() > 0.9
    # the rays really are borderline: a large share of them ends on an edge (u, v or 1-u-v within 1e-4 of zero)
    u, v = b[hit, 1], b[hit, 2]
    on_edge = (np.minimum(np.minimum(u, v), 1.0 - u - v) < 1e-4)
    assert on_edge.mean() > 0.1
    rays[:, 7] = np.where(hit, b[:, 0]*np.where(np.arange(len(rays)) % 2 == 0, np.float32(1.0 - 1e-6), np.float32(1.0 + 1e-6)), 1.0)
    assert np.array_equal(walk.trace_occluded(rays), brute.trace_occluded(rays))
