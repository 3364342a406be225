import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "full: emulator runs of opt-in / non-default kernel variants; skipped unless STGCN_FULL_TESTS=1 "
                                       "(keeps the default CPU suite at a few minutes)")


def pytest_collection_modifyitems(config, items):
    if os.environ.get("STGCN_FULL_TESTS") == "1":
        return
    skip = pytest.mark.skip(reason="opt-in kernel variant: set STGCN_FULL_TESTS=1")
    for item in items:
        if "full" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
