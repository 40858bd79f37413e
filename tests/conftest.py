import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")
ASSETS = os.path.join(REPO, "assets")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the libraries are build artefacts (git-ignored); build them on first use in a fresh checkout
    need = [os.path.join(REPO, "tungsten_amd", "libtungsten_host.so"), os.path.join(REPO, "oracle", "liboracle.so"),
            os.path.join(REPO, "tungsten_amd", "libtungsten_amd.so")]
    if not all(os.path.exists(p) for p in need):
        subprocess.run(["make", "-j", "8", "host", "core", "oracle"], cwd=REPO, check=True)
        subprocess.run(["make", "cli"], cwd=REPO, check=True)


@pytest.fixture(scope="session")
def ta():
    import tungsten_amd
    return tungsten_amd


@pytest.fixture(scope="session")
def oracle_py():
    from oracle import oracle_py as m
    return m


def scene_path(rel):
    return os.path.join(ASSETS, rel)


@pytest.fixture(scope="session")
def gpu_ctx(ta):
    """One HIP context for the whole GPU session (fails loudly without a GPU: no fallback)."""
    ctx = ta.Context(0)
    yield ctx
    ctx.close()


def edge_and_vertex_rays(host_scene, n, seed, origin_scale=6.0, first_tri=0, n_tris=None):
    """Rays aimed EXACTLY (in binary32) at points on triangle edges and at triangle vertices of the scene's soup: the inputs on
    which a BVH walk that prunes with exact boxes loses triangles whose binary32 test passes (round 1: 3 samples of 45 M at
    full size).  Half of the rays come from random origins, half graze the surface at the target (direction within ~1e-3 rad
    of the triangle's plane), where the target is the first point of its leaf box the ray meets and the computed t is least
    accurate.  Returns (n, 8) float32 rays: origin, tnear, direction, tfar."""
    import numpy as np
    d = host_scene.desc.contents
    n_tris = d.n_tris if n_tris is None else n_tris
    pos = np.ctypeslib.as_array(d.tri_pos, shape=((d.n_tris + d.n_master_tris)*9,)).reshape(-1, 3, 3)[first_tri:first_tri + n_tris]
    rng = np.random.default_rng(seed)
    tri = rng.integers(0, len(pos), n)
    a = rng.integers(0, 3, n)
    p0, p1, p2 = pos[tri, a], pos[tri, (a + 1) % 3], pos[tri, (a + 2) % 3]
    s = rng.random(n).astype(np.float32)
    s[rng.random(n) < 0.3] = 0.0                       # a vertex: up to ~6 triangles share the hit
    target = (p0 + (p1 - p0)*s[:, None]).astype(np.float32)
    centre = pos.reshape(-1, 3).mean(axis=0)
    org = (centre + rng.normal(0, 1, (n, 3))*origin_scale*rng.random((n, 1))).astype(np.float32)
    # grazing half: a random direction in the triangle's plane, tilted out of it by up to 1e-3, from 0.05 .. 5 units away
    nrm = np.cross(p1 - p0, p2 - p0).astype(np.float64)
    nrm /= np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-30)
    e = (p1 - p0).astype(np.float64)
    e /= np.maximum(np.linalg.norm(e, axis=1, keepdims=True), 1e-30)
    f = np.cross(nrm, e)
    phi = rng.uniform(0, 2*np.pi, (n, 1))
    g = np.cos(phi)*e + np.sin(phi)*f + nrm*rng.uniform(-1e-3, 1e-3, (n, 1))
    graze = rng.random(n) < 0.5
    dist = 10.0**rng.uniform(-1.3, 0.7, (n, 1))
    org[graze] = (target - g*dist).astype(np.float32)[graze]
    dirs = target - org
    dirs = (dirs/np.maximum(np.linalg.norm(dirs, axis=1, keepdims=True), 1e-30)).astype(np.float32)
    rays = np.zeros((n, 8), np.float32)
    rays[:, 0:3] = org
    rays[:, 3] = 1e-4
    rays[:, 4:7] = dirs
    rays[:, 7] = np.inf
    return rays
