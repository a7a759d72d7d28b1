import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (runs through the HIP C-ABI)")


@pytest.fixture(scope="session", autouse=True)
def _build_everything():
    import __graft_entry__ as g

    g.build()
