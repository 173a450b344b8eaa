import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import itm_testlib
    return itm_testlib.oracle_backend()


@pytest.fixture(scope="session")
def reference():
    import itm_testlib
    be = itm_testlib.reference_backend()
    if be is None:
        pytest.skip("reference build (oracle/_ref) not available on this machine")
    return be


@pytest.fixture(scope="session")
def hip():
    import itm_testlib
    return itm_testlib.hip_backend()


@pytest.fixture(scope="session")
def hip_host():
    """The product library for its host-only entry points (file formats); loads without a GPU."""
    import itm_testlib
    return itm_testlib.hip_backend()
