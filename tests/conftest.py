import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

REFERENCE = "/root/reference"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu through gpurun)")


@pytest.fixture(scope="session")
def have_reference():
    return os.path.isdir(os.path.join(REFERENCE, "data"))


@pytest.fixture(scope="session", autouse=True)
def _built_library():
    """-m 'not gpu' tests need libls_amd.so to exist (host logic + exported symbols); build it once
    if a toolchain is here and it is missing."""
    from distributed_matvec_amd import _lib

    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g

        g.build()
    yield
