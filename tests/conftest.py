import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

import hisatgenotype_amd  # noqa: E402,F401  (registers the package alias)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    import orclib
    return orclib.load()


@pytest.fixture(autouse=True)
def _clear_test_switches():
    """Path-forcing switches (engine.test_switch) never outlive the test that set them."""
    yield
    from hisatgenotype_amd import capi
    if capi._lib is not None:
        capi._lib.hgx_test_switch_set(None, None)
