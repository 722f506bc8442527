import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU restatement (test infrastructure): builds oracle/libqb3oracle.so on first use."""
    from oracle import pyoracle
    return pyoracle


@pytest.fixture(scope="session")
def qb3():
    """The product library through its C ABI."""
    import qb3_amd
    return qb3_amd
