import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def built_lib():
    """Make sure libfpca.so and the oracle exist (both are built in-tree; the GPU box gets them prebuilt)."""
    import flashpca_amd

    if not os.path.exists(flashpca_amd.LIB_PATH):
        flashpca_amd.build()
    return flashpca_amd.LIB_PATH
