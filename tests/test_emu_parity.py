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


@pytest.fixture(scope="module", params=["auto", "findalways", "hostlayout", "k2single", "k2ring2", "gang8", "lane"])
def eng(request):
    """the three K1 flavours (TBZ_K1_MODE is read when a context is created)"""
    subprocess.check_call(["make", "-C", EMU_DIR, "libtbz_emu.so"], stdout=subprocess.DEVNULL)
    T = importlib.import_module("3bz_amd")
    if request.param == "hostlayout":  # chain walk + layout on the host even when the device could do it (K3)
        os.environ["TBZ_HOST_LAYOUT"] = "1"
    elif request.param == "findalways":  # K0b block-start finder on every stream, however small (default: large items only)
        os.environ["TBZ_FIND"] = "always"
    elif request.param == "k2single":  # one wave per group in K2 (default: front end and resolve on two waves)
        os.environ["TBZ_K2_MODE"] = "single"
    elif request.param == "k2ring2":  # the ring kernel on two waves (default: three — front end, far sources, resolve)
        os.environ["TBZ_K2_RING"] = "2"
    elif request.param != "auto":
        os.environ["TBZ_K1_MODE"] = request.param
    e = T.Engine(0, lib_path=os.path.join(EMU_DIR, "libtbz_emu.so"))
    os.environ.pop("TBZ_K1_MODE", None)
    os.environ.pop("TBZ_HOST_LAYOUT", None)
    os.environ.pop("TBZ_K2_MODE", None)
    os.environ.pop("TBZ_K2_RING", None)
    os.environ.pop("TBZ_FIND", None)
    yield e
    e.close()


@pytest.mark.parametrize("case", P.ALL_CASES, ids=lambda c: c.__name__)
def test_emu_case(eng, case, request):
    flavour = request.node.callspec.params["eng"]
    if flavour in P.FLAVOUR_CASES:
        if case.__name__ not in P.FLAVOUR_CASES[flavour]:
            pytest.skip("this flavour runs the cases that can tell it from the default")
    elif flavour != "auto" and case not in P.K1_CASES:
        pytest.skip("does not depend on the K1 flavour")
    case(eng)


def test_emu_large_items_handed_to_wide_gangs():
    """a gang narrower than 64 lanes declines items far larger than the launch's mean (SEG_WIDE) and the host decodes
    them with gangs of 64: forced here with gangs of 8 and a 20 Kbit threshold (every TBZ_* switch is read when the
    context is created)"""
    T = importlib.import_module("3bz_amd")
    os.environ["TBZ_K1_MODE"] = "gang8"
    os.environ["TBZ_WIDE_BITS"] = "20000"
    e = T.Engine(0, lib_path=os.path.join(EMU_DIR, "libtbz_emu.so"))
    os.environ.pop("TBZ_K1_MODE", None)
    os.environ.pop("TBZ_WIDE_BITS", None)
    try:
        import zlib
        from tools import corpus as K
        p = K.enwik_like(300_000, 5)
        out = bytearray(len(p))
        r = e.inflate(zlib.compress(p, 6), T.FORMATS["zlib"], out)
        t = e.timings()
        assert r.status == 0 and bytes(out) == p
        assert t.k1_gang == 8 and t.huff_launches >= 2, (t.k1_gang, t.huff_launches)  # the second launch: gangs of 64
        P.case_flush_streams(e)
        P.case_overflow_and_underrun(e)
    finally:
        e.close()


def test_emu_batch_over_two_contexts():
    """tbz_inflate_batch_multi with two (emulated) contexts: LPT assignment in C, a host thread per context"""
    subprocess.check_call(["make", "-C", EMU_DIR, "libtbz_emu.so"], stdout=subprocess.DEVNULL)
    T = importlib.import_module("3bz_amd")
    engines = [T.Engine(0, lib_path=os.path.join(EMU_DIR, "libtbz_emu.so")) for _ in range(2)]
    try:
        P.multi_context_batch(engines)
    finally:
        for e in engines:
            e.close()


def _factory(env):
    T = importlib.import_module("3bz_amd")
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        return T.Engine(0, lib_path=os.path.join(EMU_DIR, "libtbz_emu.so"))
    finally:
        for k in env:
            os.environ.pop(k, None)


def test_emu_small_fused():
    """the one-launch path for small streams against the general path and the oracle"""
    P.small_fused(_factory)


def test_emu_host_pipeline():
    """the part-by-part host-to-host path of tbz_inflate, forced at 1 MiB"""
    T = importlib.import_module("3bz_amd")

    def factory(env):
        os.environ.update({k: str(v) for k, v in env.items()})
        try:
            return T.Engine(0, lib_path=os.path.join(EMU_DIR, "libtbz_emu.so"))
        finally:
            for k in env:
                os.environ.pop(k, None)
    P.host_pipeline(factory)


def test_emu_k0b_two_tiles_per_wave():
    """large launches of the K0b validation put two tiles on a wave (32 lanes each): forced here at a small size"""
    T = importlib.import_module("3bz_amd")
    os.environ["TBZ_K0B_PAIR"] = "1"
    e = T.Engine(0, lib_path=os.path.join(EMU_DIR, "libtbz_emu.so"))
    try:
        P.case_block_starts_found(e, n_blocks=10)
        os.environ["TBZ_FIND"] = "always"
        e2 = T.Engine(0, lib_path=os.path.join(EMU_DIR, "libtbz_emu.so"))
        os.environ.pop("TBZ_FIND", None)
        try:
            P.case_known_answer_vectors(e2)
        finally:
            e2.close()
    finally:
        os.environ.pop("TBZ_K0B_PAIR", None)
        os.environ.pop("TBZ_FIND", None)
        e.close()


def test_emu_k6_resolve_with_the_window_in_lds():
    """K6's parallel resolve has a flavour for long ranges (mean >= 24 KiB) that stages the 32 KiB its pointers refer to
    in LDS: forced here for every range, on streams whose groups reach into their predecessors"""
    T = importlib.import_module("3bz_amd")
    os.environ["TBZ_K6_LDS_MIN"] = "0"
    os.environ["TBZ_FIND"] = "always"
    try:
        e = T.Engine(0, lib_path=os.path.join(EMU_DIR, "libtbz_emu.so"))
    finally:
        os.environ.pop("TBZ_K6_LDS_MIN", None)
        os.environ.pop("TBZ_FIND", None)
    try:
        P.case_noflush_streams(e)
        P.case_history_across_groups(e)
    finally:
        e.close()


def test_emu_sanitized():
    """ASan/UBSan are CPU-only on this pool: the kernel + engine sources, compiled with both, run
    tests/sanitizer_scenarios.py — scenarios sized for the sanitizer that reach every kernel family, the ones rounds 3
    and 4 added included (K0g and the member walk, K6's LDS resolve, the ring kernel's far read-back, tbz_k3_slice, K0c
    inside items, ITEM_RESUME, the second stream, gangs of 32 with parked lists, staged host copies).  The run is
    started in the background when the CPU session starts (tests/conftest.py) and joined here."""
    from tests import san_runner
    procs = san_runner.start()
    for names, p in procs:
        try:
            out, err = p.communicate(timeout=1500)
        except Exception:
            p.kill()
            raise
        assert p.returncode == 0 and "sanitized ok" in out, (names, out[-800:], err[-3000:])
        for n in names:
            assert (n + " ok") in out, (n, out[-800:])
