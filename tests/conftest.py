import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def nmx():
    """The product library on a GPU box; GPU tests must fail (not skip) when it cannot initialise."""
    import nova_amd
    from nova_amd import _lib
    L = _lib.lib()
    rc = L.nmx_init(0)
    assert rc == 0, f"nmx_init failed: {L.nmx_last_error().decode()}"
    return nova_amd
