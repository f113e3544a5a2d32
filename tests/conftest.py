import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an MI355X (gfx950); run with -m gpu on the GPU box")


@pytest.fixture(scope="session")
def engine():
    """The real libgpx engine on GPU 0.  Fails loudly (no fallback) when unavailable."""
    from gpax_amd._lib import Engine

    return Engine(0)
