import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "pin: CPU-only test that pins the oracle / the host code to the reference or its "
                                       "golden vectors; ALSO selected by `-m gpu` so that the GPU box's record shows "
                                       "oracle == reference next to HIP == oracle")


# The pins (oracle == the real reference built under oracle/_ref, oracle == golden vectors, host message layer ==
# reference text) need no GPU and run in the CPU suite.  They are a few seconds, so a `-m gpu` run takes them along:
# every test of these modules that is not a GPU test itself.
PIN_MODULES = ("test_oracle_vs_ref.py", "test_oracle_golden.py", "test_nmea.py", "test_range.py", "test_vessels.py",
               "test_sinks.py")


@pytest.hookimpl(tryfirst=True)
def pytest_itemcollected(item):
    if os.path.basename(str(item.fspath)) in PIN_MODULES and "gpu" not in item.keywords:
        item.add_marker(pytest.mark.pin)
        if (item.config.getoption("-m") or "").strip() == "gpu":
            item.add_marker(pytest.mark.gpu)


def _gpu_available() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _gpu_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords and "pin" not in item.keywords:
            item.add_marker(skip)
