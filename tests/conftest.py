import os
import sys

import numpy as np
import pytest
import torch  # noqa: F401  -- first: its bundled HIP runtime must be the one librtk.so binds to (one runtime per process)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLD, name), allow_pickle=False)


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.fixture(scope="session")
def stock_scene_golden():
    z = load_golden("scene_stock.npz")
    return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def oracle_stock(stock_scene_golden):
    from oracle import pyoracle as po
    o = po.Oracle()
    o.set_scene(stock_scene_golden)
    return o


@pytest.fixture(scope="session")
def gpu_ctx(stock_scene_golden):
    """One rtk_context on cuda:0 holding the stock scene; fails loudly when librtk or the GPU is missing."""
    import raytracinginoneweekend_amd as rt
    ctx = rt.Context(0)
    ctx.set_scene(stock_scene_golden)
    yield ctx
    ctx.close()
