import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built_library():
    """Build (if stale) and load libmra_hip.so; hipcc cross-compiles without a GPU."""
    import __graft_entry__ as g
    g.build()
    from pymra_amd import plan
    return plan.load_library()
