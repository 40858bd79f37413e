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
