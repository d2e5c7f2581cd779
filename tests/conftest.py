import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
TESTS = os.path.dirname(os.path.abspath(__file__))
if TESTS not in sys.path:
    sys.path.insert(0, TESTS)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def vo():
    import __graft_entry__ as g
    pkg = g.load_package()
    if not os.path.exists(pkg.LIB_PATH):     # fresh checkout: compile the library first (hipcc cross-compiles without a GPU)
        g.build()
    return pkg


@pytest.fixture(scope="session")
def o32():
    from oracle.oracle import Oracle
    return Oracle(32)


@pytest.fixture(scope="session")
def o64():
    from oracle.oracle import Oracle
    return Oracle(64)


@pytest.fixture(scope="session")
def ctx(vo):
    """GPU context; GPU tests must fail (not skip) when the HIP path is unusable."""
    return vo.Context(0)


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))
