"""ONE flush-delimited stream across ranks (3bz_amd/multi.py): every rank's part is decoded
here by the same engine in turn (lane-emulator build — no GPU in this container; the `-m gpu` module runs the same
on the card, tests/test_multirank_gloo.py runs it with two real processes over gloo), and the verdict function —
which every rank evaluates on the same gathered records — must accept exactly the streams whose seams are proven."""
import importlib
import os
import subprocess
import zlib

import pytest

from tools import corpus as K

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_DIR = os.path.join(ROOT, "tests", "emu")
M = importlib.import_module("3bz_amd.multi")


@pytest.fixture(scope="module")
def eng():
    subprocess.check_call(["make", "-C", EMU_DIR, "libtbz_emu.so"], stdout=subprocess.DEVNULL)
    T = importlib.import_module("3bz_amd")
    e = T.Engine(0, lib_path=os.path.join(EMU_DIR, "libtbz_emu.so"))
    yield e
    e.close()


def run_all_ranks(eng, data, fmt, world):
    cuts = M.shard_plan(data, world, lib=eng.lib)
    recs, parts = [], []
    for r in range(world):
        rec, d, n = M.shard_decode(eng, data, fmt, cuts, r)
        b = bytearray(n)
        if d is not None:
            if n:
                eng.d2h(b, d, n)
            eng.free(d)
        recs.append(rec)
        parts.append(bytes(b))
    return cuts, recs, M.shard_verdict(recs, data, fmt, cuts, lib=eng.lib), b"".join(parts)


def flushed(plain, wbits, every, mode=zlib.Z_FULL_FLUSH, level=6):
    c = zlib.compressobj(level, zlib.DEFLATED, wbits)
    out = b""
    for i in range(0, len(plain), every):
        out += c.compress(plain[i:i + every]) + c.flush(mode)
    return out + c.flush()


def test_checksum_combines():
    a, b = K.enwik_like(5000, seed=1), K.enwik_like(70001, seed=2)
    assert M.adler32_combine(zlib.adler32(a), zlib.adler32(b), len(b)) == zlib.adler32(a + b)
    assert M.crc32_combine(zlib.crc32(a), zlib.crc32(b), len(b)) == zlib.crc32(a + b)
    assert M.crc32_combine(zlib.crc32(a), zlib.crc32(b""), 0) == zlib.crc32(a)
    assert M.adler32_combine(zlib.adler32(b""), zlib.adler32(b), len(b)) == zlib.adler32(b)
    assert M.crc32_combine(zlib.crc32(a), zlib.crc32(b[:1]), 1) == zlib.crc32(a + b[:1])


def shard_cases(eng, n=200_000):
    """(also run on the card by tests/test_gpu_parity.py)"""
    plain = K.enwik_like(n, seed=0x3B2)
    every = max(4096, n // 48)
    # the three containers, several rank counts: all seams clean, parts concatenate to the plaintext
    for fmt, wbits, ck in ((M.FMT_ZLIB, 15, zlib.adler32(plain)), (M.FMT_GZIP, 31, zlib.crc32(plain)),
                           (M.FMT_DEFLATE, -15, None)):
        data = flushed(plain, wbits, every)
        for world in (1, 2, 3, 4, 8):
            cuts, recs, v, out = run_all_ranks(eng, data, fmt, world)
            assert v["ok"], (fmt, world, v)
            assert out == plain and v["total"] == len(plain) and v["check"] == ck
            assert v["in_consumed"] == len(data)
            assert cuts == sorted(cuts) and all(data[c - 4:c] == M.MARK for c in cuts[1:-1])
    z = flushed(plain, 15, every)
    # does not shard -> verdict says so on every rank (then rank 0 decodes the whole stream the ordinary way)
    sync = flushed(plain, 15, every, zlib.Z_SYNC_FLUSH)               # matches reach across the cuts
    assert not run_all_ranks(eng, sync, M.FMT_ZLIB, 4)[2]["ok"]
    assert not run_all_ranks(eng, z[:-1] + bytes([z[-1] ^ 1]), M.FMT_ZLIB, 4)[2]["ok"]   # adler mismatch
    assert not run_all_ranks(eng, z[:-3], M.FMT_ZLIB, 4)[2]["ok"]     # trailer cut short
    assert not run_all_ranks(eng, z[:len(z) // 2], M.FMT_ZLIB, 4)[2]["ok"]   # truncated
    b = bytearray(z)
    b[len(z) // 3] ^= 0x10                                            # damage inside some rank's part
    v = run_all_ranks(eng, bytes(b), M.FMT_ZLIB, 4)[2]
    assert not v["ok"]
    # false markers: stored blocks whose payload is nothing but 00 00 FF FF — every cut falls inside a block
    pay = M.MARK * (n // 8)
    cuts, recs, v, out = run_all_ranks(eng, zlib.compress(pay, 0), M.FMT_ZLIB, 4)
    assert not v["ok"] and "seam" in v["why"]
    # no markers at all: one part, the other ranks idle
    cuts, recs, v, out = run_all_ranks(eng, zlib.compress(plain, 6), M.FMT_ZLIB, 4)
    assert v["ok"] and out == plain and cuts[1:] == [cuts[-1]] * 4


def test_sharded_stream_all_ranks_in_turn(eng):
    shard_cases(eng, n=120_000)


def test_boundary_report(eng):
    """tbz_result.in_consumed for a stream that is not finished: the last octet-aligned block boundary the
    decoder is sure of (include/tbz_amd.h) — what the seam proof rests on."""
    plain = K.enwik_like(64 << 10, seed=5)
    z = flushed(plain, 15, 4096)
    ends, p = [], z.find(M.MARK)
    while p >= 0:
        ends.append(p + 4)
        p = z.find(M.MARK, p + 4)
    out = bytearray(len(plain))
    for cut, want, octets in ((2, 2, 0), (ends[0], ends[0], 4096), (ends[3], ends[3], 4 * 4096),
                              (ends[3] - 2, ends[2], 4 * 4096), (ends[3] + 100, ends[3], None),
                              (ends[-1], ends[-1], len(plain))):
        r = eng.inflate(z[:cut], M.FMT_ZLIB, out)
        assert r.status == 1 and r.in_consumed == want, (cut, r.status, r.in_consumed, want)
        if octets is not None:
            assert r.out_len == octets
        assert bytes(out[:r.out_len]) == plain[:r.out_len]
    # entered at a boundary as raw deflate, ending on one: a whole number of blocks
    r = eng.inflate(z[ends[3]:ends[9]], M.FMT_DEFLATE, out)
    assert r.status == 1 and r.in_consumed == ends[9] - ends[3] and bytes(out[:r.out_len]) == plain[4 * 4096:10 * 4096]
