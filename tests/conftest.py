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


# ---- the sanitizer run (ASan + UBSan build of the kernel + engine sources on the lane emulator): minutes of CPU, so it is
# started in the background when a whole CPU session starts and joined by tests/test_emu_parity.py::test_emu_sanitized
def pytest_sessionstart(session):
    from tests import san_runner
    m = session.config.getoption("-m") or ""
    k = session.config.getoption("-k") or ""
    whole = all(os.path.isdir(a.split("::")[0]) for a in session.config.args)  # (the suite, not a file or a test picked out)
    if "not gpu" in m and not k and whole:
        try:
            san_runner.start()
        except Exception:  # the test itself reports what went wrong
            pass


def pytest_sessionfinish(session, exitstatus):
    from tests import san_runner
    san_runner.stop()
