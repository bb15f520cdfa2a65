import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "timing: wall-clock comparisons on a real MI355X (tools/form_selection.sh runs them with "
                                       "-m timing; never part of -m gpu or of the CPU run: a shared pool box must not be able to "
                                       "turn a correctness run red)")


def pytest_collection_modifyitems(config, items):
    """`timing` tests run only when asked for by name (-m timing): they are deselected from every other run."""
    if "timing" in (config.getoption("-m") or ""):
        return
    keep, drop = [], []
    for it in items:
        (drop if it.get_closest_marker("timing") else keep).append(it)
    if drop:
        config.hook.pytest_deselected(items=drop)
        items[:] = keep


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
