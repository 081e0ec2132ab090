import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
TESTS = os.path.join(ROOT, "tests")
if TESTS not in sys.path:
    sys.path.insert(0, TESTS)  # tests/cases.py


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "perf: wall-clock assertion, not a parity test: only runs when selected "
                                       "by name (-m perf); a noisy box must not turn the parity run red")


def pytest_collection_modifyitems(config, items):
    # `-m gpu` selects every GPU test, perf-marked ones included (they carry both markers through the
    # module's pytestmark): keep wall-clock assertions out unless the run asks for them explicitly
    if "perf" in (config.getoption("-m") or ""):
        return
    skip = pytest.mark.skip(reason="wall-clock assertion: run with -m perf")
    for it in items:
        if "perf" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
