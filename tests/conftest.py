import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """CPU oracle and the HIP extension are built once per session (hipcc cross-compiles without a GPU)."""
    from oracle import cpu_oracle
    cpu_oracle.build()
    from emd_amd.build import build_native
    build_native()
