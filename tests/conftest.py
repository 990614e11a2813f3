import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
REFERENCE_ASSETS = "/root/reference/assets"  # absent on the GPU box


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    config.addinivalue_line("markers", "slow: takes more than a few seconds on CPU")


@pytest.fixture(scope="session")
def rt():
    import ray_tracer_2_amd
    return ray_tracer_2_amd


@pytest.fixture(scope="session")
def oracle():
    from ray_tracer_2_amd.build import build_oracle
    build_oracle()
    from oracle import oracle as o
    o.load()
    return o


@pytest.fixture(scope="session")
def cornell(rt):
    """CornellBox-Original scene arrays from the committed fixture."""
    return rt.SceneArrays.load(os.path.join(GOLDEN, "cornell_scene.npz"))


@pytest.fixture(scope="session")
def have_reference_assets():
    return os.path.isdir(REFERENCE_ASSETS)


@pytest.fixture(scope="session")
def tracer(rt):
    t = rt.RayTracer(device=0, max_width=1920, max_height=1080)
    yield t
    t.close()


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def experiments_build():
    """True when the loaded library was built with -DRT_EXPERIMENTS=1 (tools/build_variant.sh exp -DRT_EXPERIMENTS=1, then
    RT2_LIB=ray_tracer_2_amd/librt2_mi355x_exp.so pytest ...): the measured-slower features -- options lds_top, lds_tlas,
    hybrid, wavefront -- only exist there; the product library rejects the options."""
    import ray_tracer_2_amd
    return b"+experiments" in ray_tracer_2_amd.load().rt_version()
