import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (oracle/fr3d_oracle.c), built on demand with gcc. Test infrastructure only."""
    from oracle import oracle as o
    o.build()
    o.lib()
    return o


@pytest.fixture(scope="session")
def hip():
    """The HIP engine, initialised on GPU 0.  Fails loudly if the library or the GPU is missing."""
    import flowreg3d_amd
    from flowreg3d_amd import _lib
    _lib.init(0)
    return flowreg3d_amd


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def params_of(g):
    p = g["params"]
    return dict(alpha=tuple(float(x) for x in p[:3]), update_lag=int(p[3]), iterations=int(p[4]),
                min_level=int(p[5]), levels=int(p[6]), eta=float(p[7]), a_smooth=float(p[8]),
                a_data=float(p[9]))
