"""CPU tests: pin the oracle (oracle/tbz_oracle.c) to the reference's own fixtures
and to an independent implementation (system zlib), and re-create the reference's
chunk-fuzz patterns (test-chunked-input.lisp:56-75, test-chunked-output.lisp:70-89).
"""
import gzip as pygzip
import hashlib
import json
import os
import random
import zlib

import pytest

from oracle import oracle as O
from tools import corpus as K


def _vectors(golden_dir):
    return json.load(open(os.path.join(golden_dir, "deflate_vectors.json")))["vectors"]


def _run_oneshot(data, fmt=O.DEFLATE, cap=1024):
    out = bytearray(cap)
    st = O.State(fmt, out)
    ctx = O.make_octet_vector_context(data)
    try:
        n = O.decompress(ctx, st)
    except O.OracleError as e:
        return "error", e.code, b""
    flag = ("finished" if O.finished(st) else "underrun" if O.input_underrun(st)
            else "overflow" if O.output_overflow(st) else "none")
    return flag, n, bytes(out[:n])


def test_37_known_answer_vectors(golden_dir):
    """deflate-test.lisp:69-302.  OK rows pin bytes; eof rows pin input-underrun; for
    format rows the reference harness asserts nothing (deflate-test.lisp:66), so only
    'an error, or not finished' is pinned."""
    vs = _vectors(golden_dir)
    assert len(vs) == 37
    for v in vs:
        flag, n, out = _run_oneshot(bytes.fromhex(v["input_hex"]))
        if v["class"] == "ok":
            assert flag == "finished" and out.hex() == v["expected_hex"], v
        elif v["class"] == "eof":
            assert flag == "underrun", (v, flag)
        else:
            assert flag in ("error", "underrun"), (v, flag)


def test_partial_output_before_underrun(golden_dir):
    """rows 100/142/145 of the table: bytes produced before the underrun"""
    vs = {v["line"]: v for v in _vectors(golden_dir)}
    assert _run_oneshot(bytes.fromhex(vs[100]["input_hex"]))[2] == bytes.fromhex("55ee")
    assert _run_oneshot(bytes.fromhex(vs[142]["input_hex"]))[2] == b"\x00"
    # 38 bits pack to 5 octets: the 2 pad bits complete dist #8's 3 extra bits, so one more 3-byte match lands
    assert _run_oneshot(bytes.fromhex(vs[145]["input_hex"]))[2] == b"\x00" * 262


def test_test_deflated_fixture(golden_dir):
    raw = open(os.path.join(golden_dir, "test_deflated.bin"), "rb").read()
    meta = json.load(open(os.path.join(golden_dir, "test_deflated.json")))
    n = int.from_bytes(raw[:8], "little")
    assert n == meta["plain_len"]
    out, cnt = O.decompress_vector(raw[8:], "deflate")
    assert cnt == n
    assert hashlib.sha256(out).hexdigest() == meta["sha256"]
    s1, s2 = O.adler32(out)
    assert "%08x" % (s1 | (s2 << 16)) == meta["adler32"]
    assert "%08x" % O.crc32(out) == meta["crc32"]


def test_chunked_input_pattern(golden_dir):
    """test-chunked-input.lisp:27-75: 3-byte chunks, then random chunk sizes < 1234;
    every call ends finished or input-underrun; output equals the one-shot output."""
    raw = open(os.path.join(golden_dir, "test_deflated.bin"), "rb").read()[8:]
    want, n = O.decompress_vector(raw, "deflate")
    rng = random.Random(1234)
    for trial in range(60):
        out = bytearray(n)
        st = O.make_deflate_state(out)
        pos = 0
        calls = 0
        while not O.finished(st):
            step = 3 if trial == 0 else rng.randrange(1, 1234)
            ctx = O.make_octet_vector_context(raw, start=pos, end=min(len(raw), pos + step))
            O.decompress(ctx, st)
            assert O.finished(st) or O.input_underrun(st)
            pos = min(len(raw), pos + step)
            calls += 1
            assert calls < 10000
        assert bytes(out[:st.output_offset]) == want


def test_chunked_output_pattern(golden_dir):
    """test-chunked-output.lisp:27-89: output buffers of 3 bytes, then random sizes 1…12345;
    exercises eoo / window carry / :continue-copy-history / :out-byte."""
    raw = open(os.path.join(golden_dir, "test_deflated.bin"), "rb").read()[8:]
    want, n = O.decompress_vector(raw, "deflate")
    rng = random.Random(4321)
    for trial in range(40):
        st = O.make_deflate_state()
        ctx = O.make_octet_vector_context(raw)
        got = bytearray()
        while not O.finished(st):
            size = 3 if trial == 0 else rng.randrange(1, 12346)
            buf = bytearray(size)
            O.replace_output_buffer(st, buf)
            c = O.decompress(ctx, st)
            assert O.finished(st) or O.output_overflow(st)
            got += buf[:c]
        assert bytes(got) == want


def test_chunked_both_zlib_gzip():
    plain = K.enwik_like(300_000, seed=7)
    rng = random.Random(99)
    for fmt, blob in (("zlib", zlib.compress(plain, 6)), ("gzip", pygzip.compress(plain, 6, mtime=0))):
        for trial in range(6):
            st = O.State(O.FORMATS[fmt])
            got = bytearray()
            pos = 0
            buf = bytearray(rng.randrange(1, 5000))
            O.replace_output_buffer(st, buf)
            guard = 0
            while not O.finished(st):
                guard += 1
                assert guard < 100000
                step = rng.randrange(1, 3000)
                ctx = O.make_octet_vector_context(blob, start=pos, end=min(len(blob), pos + step))
                while True:
                    c = O.decompress(ctx, st)
                    if O.output_overflow(st):
                        got += buf[:c]
                        buf = bytearray(rng.randrange(1, 5000))
                        O.replace_output_buffer(st, buf)
                        continue
                    break
                pos = min(len(blob), pos + step)
            got += buf[:st.output_offset]
            assert bytes(got) == plain, (fmt, trial)


@pytest.mark.parametrize("level", [0, 1, 6, 9])
def test_cross_check_system_zlib(level):
    """independent implementation cross-check on generated corpora (all three containers)"""
    plain = K.enwik_like(200_000, seed=level + 1) + K.xorshift64star_bytes(20_000, 5) + b"\x00" * 70_000
    z = zlib.compress(plain, level)
    assert O.decompress_vector(z, "zlib")[0] == plain
    c = zlib.compressobj(level, zlib.DEFLATED, -15)
    raw = c.compress(plain) + c.flush()
    assert O.decompress_vector(raw, "deflate")[0] == plain
    g = pygzip.compress(plain, level, mtime=0)
    assert O.decompress_vector(g, "gzip")[0] == plain
    out = bytearray(len(plain))
    _, n = O.decompress_vector(z, "zlib", output=out)
    assert n == len(plain) and bytes(out) == plain


def test_checksums_against_zlib():
    rng = random.Random(3)
    for n in (0, 1, 31, 32, 33, 5552, 5553, 65521, 100_000, 1_000_003):
        b = bytes(rng.getrandbits(8) for _ in range(min(n, 4096))) * (n // 4096 + 1)
        b = b[:n]
        s1, s2 = O.adler32(b)
        assert (s1 | (s2 << 16)) == zlib.adler32(b)
        assert O.crc32(b) == zlib.crc32(b)
    # chaining (zlib.lisp:97-102 / gzip.lisp:80-81 pass the running value back in)
    a, b = os.urandom(1000), os.urandom(777)
    s1, s2 = O.adler32(a)
    s1, s2 = O.adler32(b, s1, s2)
    assert (s1 | (s2 << 16)) == zlib.adler32(a + b)
    assert O.crc32(b, O.crc32(a)) == zlib.crc32(a + b)
    # 0xFF * many: exercises the deferred modulo
    ff = b"\xff" * 3_000_000
    s1, s2 = O.adler32(ff)
    assert (s1 | (s2 << 16)) == zlib.adler32(ff)


def test_all_five_configs_small():
    # config 1 (both readings of "64 KiB")
    for two in (False, True):
        s, p = K.config1_stream(two)
        assert O.decompress_vector(s, "deflate")[0] == p
    # config 2 shape
    s, p, a = K.zlib_flush_stream(1 << 20)
    assert s.count(b"\x00\x00\xff\xff") >= 64
    assert O.decompress_vector(s, "zlib")[0] == p
    # config 2b (sync flush: history crosses segments)
    s, p, a = K.zlib_flush_stream(1 << 18, flush=zlib.Z_SYNC_FLUSH)
    assert O.decompress_vector(s, "zlib")[0] == p
    # config 3: per-member parity (3bz decodes exactly one member, gzip.lisp:277-286)
    blob, offs, plains = K.gzip_members(5, 32 << 10)
    for o, p in zip(offs, plains):
        assert O.decompress_vector(blob, "gzip", start=o)[0] == p
    # config 5
    s, p = K.adversarial_stream(total=1 << 20)
    assert zlib.decompress(s) == p
    assert O.decompress_vector(s, "zlib")[0] == p


def test_container_errors_and_flags():
    plain = b"hello hello hello hello"
    z = zlib.compress(plain)
    # bad adler
    bad = z[:-1] + bytes([z[-1] ^ 1])
    with pytest.raises(O.OracleError) as e:
        O.decompress_vector(bad, "zlib")
    assert e.value.code == -11
    # truncated trailer -> input-underrun; zlib returns output-offset (zlib.lisp:83-86)
    out = bytearray(100)
    st = O.make_zlib_state(out)
    n = O.decompress(O.make_octet_vector_context(z[:-2]), st)
    assert O.input_underrun(st) and not O.finished(st) and n == len(plain)
    # gzip truncated trailer returns 0 (gzip.lisp:83-86)
    g = pygzip.compress(plain, mtime=0)
    st = O.make_gzip_state(bytearray(100))
    n = O.decompress(O.make_octet_vector_context(g[:-6]), st)
    assert O.input_underrun(st) and n == 0
    # header errors
    for blob, code in ((b"\x78\x9d" + z[2:], -9), (b"\x79\x9c", -9), (b"\x78\xbb" + z[2:], -10)):
        with pytest.raises(O.OracleError):
            O.decompress_vector(blob, "zlib")
    with pytest.raises(O.OracleError):
        O.decompress_vector(b"\x1f\x8c" + g[2:], "gzip")
    # gzip with every optional field incl. header crc
    import struct
    hdr = bytearray(b"\x1f\x8b\x08\x1e\x00\x00\x00\x00\x00\x03")
    hdr += struct.pack("<H", 5) + b"extra" + b"name\x00" + b"comment\x00"
    hdr += struct.pack("<H", zlib.crc32(bytes(hdr)) & 0xFFFF)
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    body = c.compress(plain) + c.flush()
    blob = bytes(hdr) + body + struct.pack("<II", zlib.crc32(plain), len(plain))
    assert O.decompress_vector(blob, "gzip")[0] == plain
    assert pygzip.decompress(blob) == plain
    # output too small with :output -> error (api.lisp:45-46)
    with pytest.raises(O.OracleError) as e:
        O.decompress_vector(z, "zlib", output=bytearray(5))
    assert e.value.code == -21
    with pytest.raises(O.OracleError) as e:
        O.decompress_vector(z[:8], "zlib", output=bytearray(100))
    assert e.value.code == -20
    # distance before start of output with no window (deflate.lisp:345)
    w = K.FixedHuffmanWriter()
    w.begin_block(True)
    w.literal(65)
    w.match(3, 2)
    w.end_block()
    w.align()
    with pytest.raises(O.OracleError) as e:
        O.decompress_vector(w.getvalue(), "deflate")
    assert e.value.code == -8
    # overflow leaves the buffer full of the correct prefix
    big = zlib.compress(K.enwik_like(100_000, 3))
    st = O.make_zlib_state(bytearray(1000))
    n = O.decompress(O.make_octet_vector_context(big), st)
    assert O.output_overflow(st) and n == 1000
    assert bytes(st.output_buffer) == K.enwik_like(1000, 3)


def test_sanitizer_build_runs_vectors(golden_dir):
    """ASan/UBSan are CPU-only on this pool: run the vectors through a sanitized build."""
    import subprocess
    import sys
    here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")
    r = subprocess.run(["make", "-C", here, "libtbz_oracle_asan.so"], capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("sanitizer build unavailable: " + r.stderr[-200:])
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    code = (
        "import ctypes,json,sys\n"
        "L=ctypes.CDLL(sys.argv[1])\n"
        "L.tbzo_decompress_vector_into.restype=ctypes.c_int64\n"
        "L.tbzo_decompress_vector_into.argtypes=[ctypes.c_char_p,ctypes.c_size_t,ctypes.c_size_t,ctypes.c_int,ctypes.c_char_p,ctypes.c_size_t]\n"
        "vs=json.load(open(sys.argv[2]))['vectors']\n"
        "for v in vs:\n"
        "    d=bytes.fromhex(v['input_hex']); o=ctypes.create_string_buffer(1024)\n"
        "    r=L.tbzo_decompress_vector_into(d,0,len(d),0,o,1024)\n"
        "    if v['class']=='ok': assert r>=0 and o.raw[:r].hex()==v['expected_hex'],v\n"
        "    else: assert r<0,v\n"
        "print('SAN_OK')\n")
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0")
    r = subprocess.run([sys.executable, "-c", code, os.path.join(here, "libtbz_oracle_asan.so"),
                        os.path.join(golden_dir, "deflate_vectors.json")], capture_output=True, text=True, env=env)
    assert "SAN_OK" in r.stdout, r.stderr[-2000:]
