import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def ulp_distance(a, b):
    """Units in the last place between two float32 arrays (monotone integer mapping of the bit patterns)."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    ia = a.view(np.int32).astype(np.int64)
    ib = b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7FFFFFFF), ia)
    ib = np.where(ib < 0, -(ib & 0x7FFFFFFF), ib)
    return np.abs(ia - ib)


def scaled_ulp(a, b, scale):
    """|a-b| in units of the float32 spacing at magnitude `scale` (per element or scalar)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    sp = np.spacing(np.maximum(np.abs(np.asarray(scale, dtype=np.float32)), np.float32(1e-30)).astype(np.float32)).astype(np.float64)
    return np.abs(a - b) / sp


@pytest.fixture(scope="session")
def orc():
    from oracle import oracle

    oracle.lib()
    return oracle


@pytest.fixture(scope="session")
def ca():
    import craytracer_amd

    return craytracer_amd


@pytest.fixture(scope="session")
def gpu(ca):
    """One libcaster_hip context on cuda:0.  Fails (not skips) when the library or the device is missing."""
    ctx = ca.CasterHip(0)
    yield ctx
    ctx.close()


def random_rays(rng, n, origin_scale=15.0):
    o = rng.uniform(-origin_scale, origin_scale, size=(n, 3)).astype(np.float32)
    o[:, 1] = np.abs(o[:, 1]) * 0.3 + 0.05
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d *= rng.uniform(0.05, 12.0, size=(n, 1)).astype(np.float32)
    return np.concatenate([o, d], axis=1).astype(np.float32)
