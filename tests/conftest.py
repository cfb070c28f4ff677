import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import pyoracle
    pyoracle.build()
    return pyoracle


@pytest.fixture(scope="session")
def hip_lib():
    """the HIP C-ABI library; GPU tests must run the native path, never a fallback"""
    import lvt_amd
    if not os.path.exists(lvt_amd.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lvt_amd.load_library()
    return lvt_amd
