import os
import sys

import pytest

# the oracle's OpenMP threads must not spin when the box gives the process fewer cores than threads
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def backend_cls():
    """The HIP backend class; GPU tests fail (not skip) when the library or the device is missing."""
    from sadvio_amd import capi
    capi.load_library()
    return capi.Backend
