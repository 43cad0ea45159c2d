import os
import sys

import pytest

os.environ.setdefault("DSG_TESTING", "1")   # dsg_set_tuning (kernel-selection switches for A/B parity tests) is a test hook

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def lib_built():
    """Build libdsg.so once per session (hipcc cross-compiles on CPU-only hosts)."""
    from drivescenegen_amd.csrc.build import build
    return build(verbose=False)
