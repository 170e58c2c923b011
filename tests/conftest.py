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


# ---- two-rank job of tests/test_shard_gpu.py: started HERE, at the end of collection, because this process has not made a GPU
# call yet (an exec from a process that has initialised the GPU is refused on the pool); the test only waits for its result.
_SHARD_JOB = {"job": None}


def pytest_collection_finish(session):
    if not any(item.name == "test_two_ranks_on_device_tensors" for item in session.items):
        return
    if not os.path.exists("/dev/kfd"):                          # no GPU driver here (build container): the test skips
        return
    if getattr(session.config.option, "collectonly", False):     # nothing will run
        return
    import subprocess
    import tempfile
    out = tempfile.mkdtemp(prefix="rtod_shard_gpu_")
    log = open(os.path.join(out, "job.log"), "w")
    proc = subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "shard_gpu_job.py"), out], stdout=log, stderr=subprocess.STDOUT)
    _SHARD_JOB["job"] = {"out": out, "proc": proc}


@pytest.fixture(scope="session")
def shard_gpu_job():
    return _SHARD_JOB["job"]
