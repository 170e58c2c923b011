import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
REFERENCE = "/root/reference"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs /root/reference (build container only)")


def pytest_collection_modifyitems(config, items):
    have_ref = os.path.isdir(os.path.join(REFERENCE, "src"))
    skip_ref = pytest.mark.skip(reason="/root/reference not present (GPU box)")
    for item in items:
        if "reference" in item.keywords and not have_ref:
            item.add_marker(skip_ref)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
