import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure).  Built on demand with oracle/Makefile."""
    from oracle import orc as _orc
    _orc.build()
    _orc.lib()
    return _orc


@pytest.fixture(scope="session")
def gi():
    """The product library on a real GPU; fails loudly if the HIP extension is missing."""
    from gatling_amd import capi
    capi.initialize(0)
    return capi
