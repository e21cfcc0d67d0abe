import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950); run with `-m gpu` on the GPU box")


@pytest.fixture(scope="session")
def built():
    """Build libc3d.so + the oracle once per session (cross-compiles without a GPU)."""
    import __graft_entry__ as g
    g.build()
    return True


@pytest.fixture(scope="session")
def solver(built):
    from chromosome3d_amd import Solver
    s = Solver(0)
    yield s
    s.close()
