"""TEST INFRASTRUCTURE — run under LD_PRELOAD=libasan.so against tests/emu/libtbz_emu_asan.so (AddressSanitizer + UBSan
build of the UNCHANGED kernel + engine sources; GPU sanitizers are not available on this pool).  Started in the
background when the CPU test session starts (tests/conftest.py) and joined by tests/test_emu_parity.py::test_emu_sanitized.

Scenarios are sized for the sanitizer (a decode costs seconds here) and chosen so that every kernel family is reached —
in particular the ones round 3 added: tbz_k0g_scan and the gzip-member walk with second chances,
tbz_k6_resolve_lds (TBZ_K6_LDS_MIN=0), the 11 KB ring with far read-back, tbz_k3_slice (TBZ_SLICE forced), K0c inside
flush-delimited items, the ITEM_RESUME path of sessions, the gangs of 64 beside a narrow launch on the second stream —
and round 4's: gangs of 32 with their canonical lists parked in memory and twelve-word windows, staged host copies
through the copy pool.  Each scenario compares with the oracle as the parity cases do."""
import importlib
import os
import random
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import parity_cases as P  # noqa: E402
from tools import corpus as K  # noqa: E402

T = importlib.import_module("3bz_amd")
LIB = sys.argv[1]


def engine(**env):
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        return T.Engine(0, lib_path=LIB)
    finally:
        for k in env:
            os.environ.pop(k, None)


def noflush_small(eng, n):
    p = K.enwik_like(n, 0x3B7)
    s = zlib.compress(p, 6)
    w = P.assert_same(eng, s, "zlib", n, what="no-flush zlib")
    assert w["flag"] == "finished" and w["bytes"] == p
    t = eng.timings()
    assert t.n_hgroups >= 1 and t.n_candidates >= 2, (t.n_hgroups, t.n_candidates)
    P.assert_same(eng, s, "zlib", n // 2 + 7, what="no-flush zlib, short buffer")
    P.assert_same(eng, s[:len(s) // 2], "zlib", n, what="no-flush zlib, cut")
    page = K.xorshift64star_bytes(20_000, 77)
    P.assert_same(eng, zlib.compress(page * 3, 6), "zlib", 60_000, what="repeated page")


SCENARIOS = []


def scenario(f):
    SCENARIOS.append(f)
    return f


@scenario
def vectors_and_false_markers():
    e = engine()
    P.case_known_answer_vectors(e)
    P.case_false_markers(e)
    e.close()


@scenario
def k0b_k6_lds_ring_slices():
    # block-start finder on a small stream, K6's resolve with the window in LDS for every range, segments cut into
    # 4 KiB slices (tbz_k3_slice), the ring kernel with far read-back (matches beyond its 8 KiB of history)
    e = engine(TBZ_FIND="always", TBZ_K6_LDS_MIN=0, TBZ_SLICE=4096)
    noflush_small(e, 70_000)
    P.case_history_across_groups(e)
    e.close()


@scenario
def gzip_members_walk():
    e = engine()
    P.case_gzip_members(e, n_members=3, max_len=2500, n_false=24)
    e.close()


@scenario
def sessions_resume_inside_blocks():
    e = engine()
    plain = P._mixed_plain(9000, 5)
    rng = random.Random(4)
    for fmt, blob in (("zlib", zlib.compress(plain, 6)), ("deflate", zlib.compress(plain, 1)[2:-4])):
        steps = [rng.randrange(300, 1500) for _ in range(8)]
        assert P._chunked_lockstep(e, blob, fmt, steps, [len(plain) + 10], "san: input chunks") == plain
        sizes = [rng.randrange(500, 4000) for _ in range(8)]
        assert P._chunked_lockstep(e, blob, fmt, [len(blob)], sizes, "san: output buffers") == plain
    e.close()


@scenario
def k0c_inside_items_and_fixed_chains():
    e = engine()
    s, p = K.adversarial_stream(total=96 << 10, full_flush_every=32 << 10)
    w = P.assert_same(e, s, "zlib", len(p), what="config 5 with flush points")
    assert w["bytes"] == p
    s, p = P._fixed_chain(3, 120, 30)
    w = P.assert_same(e, s, "deflate", len(p) + 10, what="fixed chain")
    assert w["bytes"] == p
    e.close()


@scenario
def wide_items_beside_a_narrow_launch():
    e = engine(TBZ_K1_MODE="gang8", TBZ_WIDE_BITS=20000)
    p = K.enwik_like(120_000, 5)
    out = bytearray(len(p))
    r = e.inflate(zlib.compress(p, 6), T.FORMATS["zlib"], out)
    assert r.status == 0 and bytes(out) == p and e.timings().huff_launches >= 2
    e.close()


@scenario
def gangs_of_32_parked_lists_and_dense_tokens():
    e = engine(TBZ_K1_MODE="gang32")
    s, p, a = K.zlib_flush_stream(64 << 10)
    P.assert_same(e, s, "zlib", len(p), what="full flush, gangs of 32")
    z = zlib.compress(bytes(40 << 10) + K.enwik_like(12_000, 3), 6)   # zeros: declined, decoded again in a region of their own
    P.assert_same(e, z, "zlib", (40 << 10) + 12_000, what="zeros then text, gangs of 32")
    e.close()


@scenario
def deep_codes_small_pools():
    e = engine(TBZ_K1_MODE="gang32")   # (288 / 64 second-level entries: codes that need more take the exact step)
    P.case_deep_codes(e)
    e.close()


@scenario
def staged_host_copies():
    e = engine(TBZ_STAGE_CHUNK_KIB=256, TBZ_COPY_THREADS=3)
    rng = random.Random(5)
    plains = [K.xorshift64star_bytes(rng.randrange(1, 160_000), seed=i) for i in range(12)] + [b""]
    ins = [zlib.compress(q, 0) for q in plains]   # (stored blocks: the octets are what this scenario is about)
    outs = [bytearray(len(q) + rng.randrange(0, 50)) for q in plains]
    res = e.inflate_batch(ins, T.FORMATS["zlib"], outs)
    for r, q, o in zip(res, plains, outs):
        assert r.status == 0 and r.out_len == len(q) and bytes(o[:len(q)]) == q
    p = K.xorshift64star_bytes(1300 << 10, 9)
    out = bytearray(len(p))
    r = e.inflate(zlib.compress(p, 0), T.FORMATS["zlib"], out)
    assert r.status == 0 and bytes(out) == p
    r, buf = e.inflate_alloc(zlib.compress(p, 0), T.FORMATS["zlib"])
    assert r.status == 0 and bytes(buf) == p
    e.close()


if __name__ == "__main__":
    names = sys.argv[2:]
    for f in SCENARIOS:
        if names and f.__name__ not in names:
            continue
        t = time.time()
        f()
        print("%s ok %.0fs" % (f.__name__, time.time() - t), flush=True)
    print("sanitized ok", flush=True)
