import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


def pytest_sessionfinish(session, exitstatus):
    """GPU runs leave the measured parity margins in gpurun_out/parity_margins.json (tests/margins.py)."""
    try:
        from tests import margins

        path = margins.flush()
        if path:
            print(f"\nparity margins written to {path}")
    except Exception as e:  # never turn a finished run red from here
        print(f"\nparity margins not written: {e!r}")
