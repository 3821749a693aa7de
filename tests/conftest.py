import json
import os
import sys

import numpy as np
import pytest

# torch first: it ships its own copy of the HIP runtime, and a process that has already loaded the system's copy
# (through librt_mi355x.so) and only then imports torch ends up with a torch that sees no GPU
import torch  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

GOLDEN = os.path.join(ROOT, "tests", "golden")
REF_ASSETS = "/root/reference/assets"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def frames():
    return np.load(os.path.join(GOLDEN, "frames.npz"))


@pytest.fixture(scope="session")
def checksums():
    with open(os.path.join(GOLDEN, "checksums.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def intersect_vectors():
    return np.load(os.path.join(GOLDEN, "intersect.npz"))


@pytest.fixture(scope="session")
def ref_assets():
    if not os.path.isdir(REF_ASSETS):
        pytest.skip("reference tree not present (GPU box)")
    return REF_ASSETS


@pytest.fixture(scope="session")
def rt():
    """One device context for the whole GPU session.  No fallback: a missing library or GPU fails."""
    from raytracing_amd import RayTracer
    r = RayTracer(0)
    yield r
    r.close()


_scene_cache = {}


@pytest.fixture(scope="session")
def scene_of():
    from raytracing_amd import host

    def get(name, params=None):
        key = (name, tuple(params or ()))
        if key not in _scene_cache:
            _scene_cache[key] = host.make_scene(name, params=params)
        return _scene_cache[key]
    return get
