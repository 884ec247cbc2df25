import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN_DIR = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz")))

    return load


def pytest_addoption(parser):
    parser.addoption("--update-emu-manifest", action="store_true", default=False,
                     help="after the run, write the list of CPU-emulation libraries the tests asked for to tests/emu/prebuild_manifest.json (what a later run compiles in parallel up front)")


def pytest_collection_finish(session):
    """A (nearly) full run of the emulation tests compiles its ~60 libraries on several cores before the first test instead of one by one inside the tests."""
    emu_files = {"test_generated_emulation.py", "test_idsva_so_oracle.py", "test_capi_boundary.py"}
    n_emu = sum(1 for it in session.items if os.path.basename(str(it.fspath)) in emu_files)
    if n_emu >= 40 and not session.config.getoption("--update-emu-manifest"):
        sys.path.insert(0, os.path.join(REPO, "tests"))
        from emu_harness import prebuild_from_manifest

        cpus = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 2)
        prebuild_from_manifest(max(1, min(6, cpus - 1)))


def pytest_sessionfinish(session, exitstatus):
    if session.config.getoption("--update-emu-manifest") and exitstatus == 0:
        from emu_harness import write_manifest

        write_manifest()
