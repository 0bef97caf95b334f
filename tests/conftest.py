import os
import sys

import pytest

try:  # load torch's HIP runtime BEFORE libpynnd_amd.so initialises its own: one runtime per process (GPU tests)
    import torch  # noqa: F401
except ImportError:  # pragma: no cover
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: longer CPU test")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def oracle_strict():
    from oracle import oracle as O

    O.build()
    return O.load("strict")
