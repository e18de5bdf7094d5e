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
    """The CPU oracle (test infrastructure): built on demand with gcc."""
    import oracle_lib
    oracle_lib.lib()
    return oracle_lib


@pytest.fixture(scope="session")
def halart():
    """The product package.  GPU tests call through the C ABI of libhalart.so; it must already be built
    (__graft_entry__.build()) — a missing library is an error, never a skip or a fallback."""
    import hala_renderer_amd as H
    if not os.path.exists(H.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    H.load_library()
    return H


GOLDEN = os.path.join(ROOT, "tests", "golden")
