import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """the CPU oracle binding (checker only)"""
    from oracle import oracle as orc

    orc.lib()
    return orc


@pytest.fixture(scope="session")
def ctx():
    """one HIP context for the whole GPU session; fails loudly when the extension is missing"""
    from nerf_prv_amd import api

    c = api.Context(0)
    yield c
    c.close()
