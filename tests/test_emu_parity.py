"""CPU run of the parity cases through the lane emulator: the UNCHANGED kernel + engine sources
(3bz_amd/csrc) compiled for the host (tests/emu/), checked against the oracle.  This is how the
device code is debugged and sanitized here (no GPU in the build container); the `-m gpu` module
runs the same cases on the real library.  The emulator is test infrastructure — the product never
loads it."""
import importlib
import os
import subprocess

import pytest

from tests import parity_cases as P

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_DIR = os.path.join(ROOT, "tests", "emu")


@pytest.fixture(scope="module")
def eng():
    subprocess.check_call(["make", "-C", EMU_DIR, "libtbz_emu.so"], stdout=subprocess.DEVNULL)
    T = importlib.import_module("3bz_amd")
    e = T.Engine(0, lib_path=os.path.join(EMU_DIR, "libtbz_emu.so"))
    yield e
    e.close()


@pytest.mark.parametrize("case", P.ALL_CASES, ids=lambda c: c.__name__)
def test_emu_case(eng, case):
    case(eng)
