import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def oracle_lib(oracle_mod):
    """The CPU oracle exporting the vdb_* ABI — injected explicitly, tests only."""
    return oracle_mod.lib()


@pytest.fixture(scope="session")
def hip_lib():
    """The product library.  GPU tests must run on it; a missing library is a failure, not a skip."""
    from vectordb_amd import _capi
    return _capi.load_library()
