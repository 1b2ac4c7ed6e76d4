"""TEST INFRASTRUCTURE: starts the sanitizer run (tests/sanitizer_scenarios.py under ASan + UBSan on the lane emulator) in
background processes — once per test session, whoever asks first: tests/conftest.py when a whole CPU session starts, or
tests/test_emu_parity.py::test_emu_sanitized when it is run on its own — and hands the processes to the test that joins them."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GROUPS = [
    ["k0b_k6_lds_ring_slices", "sessions_resume_inside_blocks"],
    ["gzip_members_walk", "k0c_inside_items_and_fixed_chains", "wide_items_beside_a_narrow_launch"],
    ["gangs_of_32_parked_lists_and_dense_tokens", "vectors_and_false_markers"],
    ["deep_codes_small_pools", "staged_host_copies"],
]
_STATE = {"procs": None}


def start():
    """build libtbz_emu_asan.so and start the scenario groups (idempotent); returns the list of (names, Popen)"""
    if _STATE["procs"] is not None:
        return _STATE["procs"]
    emu = os.path.join(ROOT, "tests", "emu")
    r = subprocess.run(["make", "-C", emu, "libtbz_emu_asan.so"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:detect_stack_use_after_return=0")
    for k in [k for k in env if k.startswith("TBZ_")]:
        env.pop(k)
    procs = []
    for names in GROUPS:
        p = subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "sanitizer_scenarios.py"),
                              os.path.join(emu, "libtbz_emu_asan.so")] + names,
                             stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
        procs.append((names, p))
    _STATE["procs"] = procs
    return procs


def stop():
    for _, p in (_STATE["procs"] or []):
        if p.poll() is None:
            p.kill()
