"""Parity cases shared by the CPU (lane-emulator) and GPU (lib3bz_amd.so) test modules.

Every case drives the product through the 3bz-shaped API (3bz_amd.api) and checks it against the
oracle (oracle/tbz_oracle.c, itself pinned to the reference's vectors) on the same inputs:
bit-exact octets, identical status flag, identical count, identical error class.
"""
import gzip as pygzip
import hashlib
import importlib
import json
import os
import random
import struct
import zlib

from oracle import oracle as O
from tools import corpus as K

T = importlib.import_module("3bz_amd")
A = T.api
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FMT = {"deflate": 0, "zlib": 1, "gzip": 2}


# ---------------------------------------------------------------------------------- helpers
def oracle_oneshot(data, fmt, cap, start=0, end=None):
    """what 3bz's (decompress ctx state) does on a fresh state with a `cap`-octet buffer"""
    out = bytearray(cap)
    st = O.State(FMT[fmt], out)
    ctx = O.make_octet_vector_context(data, start=start, end=end)
    try:
        n = O.decompress(ctx, st)
    except O.OracleError as e:
        return {"flag": "error", "code": e.code, "ret": None, "bytes": b"", "offset": None}
    flag = ("finished" if O.finished(st) else "underrun" if O.input_underrun(st)
            else "overflow" if O.output_overflow(st) else "none")
    return {"flag": flag, "code": 0, "ret": n, "bytes": bytes(out[:st.output_offset]), "offset": st.output_offset}


def engine_oneshot(eng, data, fmt, cap, start=0, end=None):
    out = bytearray(cap)
    mk = {"deflate": A.make_deflate_state, "zlib": A.make_zlib_state, "gzip": A.make_gzip_state}[fmt]
    st = mk(out)
    ctx = A.make_octet_vector_context(data, start=start, end=end)
    try:
        n = A.decompress(ctx, st, engine=eng)
    except A.ThreeBzError as e:
        return {"flag": "error", "code": e.code, "ret": None, "bytes": b"", "offset": None}
    flag = ("finished" if A.finished(st) else "underrun" if A.input_underrun(st)
            else "overflow" if A.output_overflow(st) else "none")
    return {"flag": flag, "code": 0, "ret": n, "bytes": bytes(out[:st.output_offset]), "offset": st.output_offset}


def assert_same(eng, data, fmt, cap, start=0, end=None, what=""):
    want = oracle_oneshot(data, fmt, cap, start, end)
    got = engine_oneshot(eng, data, fmt, cap, start, end)
    assert got["flag"] == want["flag"], (what, got["flag"], want["flag"], got["code"], want["code"])
    if want["flag"] == "error":
        assert got["code"] == want["code"], (what, got["code"], want["code"])
        return want
    assert got["offset"] == want["offset"], (what, got["offset"], want["offset"])
    assert got["ret"] == want["ret"], (what, got["ret"], want["ret"])
    assert got["bytes"] == want["bytes"], (what, "octets differ",
                                           next((i for i, (a, b) in enumerate(zip(got["bytes"], want["bytes"]))
                                                 if a != b), None))
    return want


def same_or_stale_tables(eng, data, fmt, cap, what=""):
    """assert_same, except for the one documented deviation: a block whose code-length, literal/length
    or distance alphabet is ALL ZERO leaves the previous block's table in place in the reference (huffman-tree.lisp:
    156-157) — decoder state that crosses block and flush boundaries, in streams no encoder emits — where the device
    path behaves like a fresh reference state (every entry invalid).  Such a stream must then agree with the oracle
    run with exactly that switch.  Returns (oracle result, deviated?)."""
    try:
        return assert_same(eng, data, fmt, cap, what=what), False
    except AssertionError:
        O.set_fresh_tables(True)
        try:
            return assert_same(eng, data, fmt, cap, what=what + " [fresh tables]"), True
        finally:
            O.set_fresh_tables(False)


# ---------------------------------------------------------------------------------- cases
def case_known_answer_vectors(eng):
    """deflate-test.lisp:69-302 through the state API; the device path must behave exactly like the
    oracle on all 37 (which is stronger than the reference harness, deflate-test.lisp:66)."""
    vs = json.load(open(os.path.join(GOLDEN, "deflate_vectors.json")))["vectors"]
    assert len(vs) == 37
    for v in vs:
        data = bytes.fromhex(v["input_hex"])
        w = assert_same(eng, data, "deflate", 1024, what="vector@%d" % v["line"])
        if v["class"] == "ok":
            assert w["flag"] == "finished" and w["bytes"].hex() == v["expected_hex"]
        elif v["class"] == "eof":
            assert w["flag"] == "underrun"
        else:
            assert w["flag"] in ("error", "underrun")


def case_test_deflated(eng):
    raw = open(os.path.join(GOLDEN, "test_deflated.bin"), "rb").read()
    meta = json.load(open(os.path.join(GOLDEN, "test_deflated.json")))
    buf, n = A.decompress_vector(raw, format="deflate", start=8, engine=eng)
    t = eng.timings()   # without :output the stream is decoded ONCE (tbz_inflate_alloc): one input copy, one Huffman pass
    assert t.huff_launches == 1 and t.h2d_copies == 1, (t.huff_launches, t.h2d_copies)
    assert n == meta["plain_len"] == int.from_bytes(raw[:8], "little")
    assert hashlib.sha256(bytes(buf[:n])).hexdigest() == meta["sha256"]
    out = bytearray(n)
    _, n2 = A.decompress_vector(raw[8:], format="deflate", output=out, engine=eng)
    assert n2 == n and hashlib.sha256(bytes(out)).hexdigest() == meta["sha256"]


def case_reference_chunk_patterns(eng, in_step=3, out_step=3, n_random=3, max_calls=250):
    """the reference's own chunking tests on its own fixture (test-chunked-input.lisp:27-75, test-chunked-output.lisp:
    27-89): `test.deflated` fed in 3-octet chunks, then in random chunks < 1234; decoded into 3-octet buffers, then
    into random buffers <= 12345 — every call compared with the oracle.  (The reference runs 30 000 random rounds
    against libz; tools/fuzz_chunked.py is the open-ended version.)  `max_calls` bounds the fixed-step runs on the
    CPU emulator, where a call costs milliseconds: the GPU module runs them to the end."""
    raw = open(os.path.join(GOLDEN, "test_deflated.bin"), "rb").read()[8:]
    meta = json.load(open(os.path.join(GOLDEN, "test_deflated.json")))
    n = meta["plain_len"]
    rng = random.Random(0x3B2)
    cut = raw if max_calls is None else raw[: in_step * max_calls]
    got = _chunked_lockstep(eng, cut, "deflate", [in_step], [n + 8], "test.deflated in %d-octet chunks" % in_step,
                            max_calls=None if max_calls is None else max_calls + 10)
    if max_calls is None:
        assert hashlib.sha256(got).hexdigest() == meta["sha256"]
    for k in range(n_random):
        steps = [rng.randrange(1, 1234) for _ in range(64)]
        got = _chunked_lockstep(eng, raw, "deflate", steps, [n + 8], "test.deflated random chunks %d" % k)
        assert hashlib.sha256(got).hexdigest() == meta["sha256"]
    small = raw if max_calls is None else raw[:160]   # (3-octet buffers: one call per 3 octets of OUTPUT)
    _chunked_lockstep(eng, small, "deflate", [len(small)], [out_step], "test.deflated into %d-octet buffers" % out_step,
                      max_calls=None if max_calls is None else 100_000)
    for k in range(n_random):
        sizes = [rng.randrange(1, 12346) for _ in range(16)]
        got = _chunked_lockstep(eng, raw, "deflate", [len(raw)], sizes, "test.deflated random buffers %d" % k)
        assert hashlib.sha256(got).hexdigest() == meta["sha256"]


def _mixed_plain(n, seed):
    rng = random.Random(seed)
    parts = [K.enwik_like(n // 2, seed=seed), K.xorshift64star_bytes(n // 8, seed + 1), b"\x00" * (n // 8),
             bytes(rng.choice(b"abc") for _ in range(n // 8)), K.enwik_like(n // 8, seed=seed + 2)]
    return b"".join(parts)


def case_containers_and_levels(eng, n=48_000):
    for level in (0, 1, 6, 9):
        plain = _mixed_plain(n, level + 10)
        z = zlib.compress(plain, level)
        c = zlib.compressobj(level, zlib.DEFLATED, -15)
        raw = c.compress(plain) + c.flush()
        g = pygzip.compress(plain, level, mtime=0)
        for fmt, blob in (("zlib", z), ("deflate", raw), ("gzip", g)):
            w = assert_same(eng, blob, fmt, len(plain), what="%s L%d" % (fmt, level))
            assert w["flag"] == "finished" and w["bytes"] == plain
            buf, cnt = A.decompress_vector(blob, format=fmt, engine=eng)
            assert cnt == len(plain) and bytes(buf) == plain
    # zlib strategies that change the block mix: fixed-Huffman only, Huffman-only, RLE
    plain = _mixed_plain(n // 2, 77)
    for strat in (zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE):
        c = zlib.compressobj(6, zlib.DEFLATED, 15, 8, strat)
        z = c.compress(plain) + c.flush()
        w = assert_same(eng, z, "zlib", len(plain), what="strategy %d" % strat)
        assert w["bytes"] == plain
    # gzip with every optional header field incl. header crc16 (gzip.lisp:180-255)
    hdr = bytearray(b"\x1f\x8b\x08\x1e\x00\x00\x00\x00\x00\x03")
    hdr += struct.pack("<H", 5) + b"extra" + b"name\x00" + b"comment\x00"
    hdr += struct.pack("<H", zlib.crc32(bytes(hdr)) & 0xFFFF)
    small = plain[:5000]
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    blob = bytes(hdr) + c.compress(small) + c.flush() + struct.pack("<II", zlib.crc32(small), len(small))
    w = assert_same(eng, blob, "gzip", len(small), what="gzip optional fields")
    assert w["bytes"] == small
    bad = bytearray(blob)
    bad[len(hdr) - 1] ^= 0x40
    assert_same(eng, bytes(bad), "gzip", len(small), what="gzip bad header crc")
    # :start / :end (api.lisp:23)
    pad = b"\xAA" * 7 + z + b"\xBB" * 9
    buf, cnt = A.decompress_vector(pad, format="zlib", start=7, end=7 + len(z), engine=eng)
    assert bytes(buf[:cnt]) == plain


def case_flush_streams(eng, n=96 << 10):
    """config 2 shape: Z_FULL_FLUSH every 16 KiB -> independent segments found by the marker scan"""
    s, p, a = K.zlib_flush_stream(n)
    out = bytearray(n)
    res = eng.inflate(s, 1, out)
    assert res.status == 0 and res.out_len == n and bytes(out) == p
    assert res.segments == n // 16384, res.segments
    assert res.adler32 == a == res.trailer_check and (res.flags & 1)
    assert res.in_consumed == len(s)
    assert_same(eng, s, "zlib", n, what="full-flush stream")
    # config 2b: Z_SYNC_FLUSH — history crosses segments, so they must share one window
    s, p, a = K.zlib_flush_stream(n // 2, flush=zlib.Z_SYNC_FLUSH)
    out = bytearray(len(p))
    res = eng.inflate(s, 1, out)
    assert res.status == 0 and bytes(out) == p
    t = eng.timings()   # the segments reach into each other: shared windows, or symbolic history + K6
    assert t.n_hgroups >= 1 or t.n_groups < t.n_segments
    # small blocks: a flush every 1000 octets
    s, p, a = K.zlib_flush_stream(40_000, block=1000)
    assert_same(eng, s, "zlib", len(p), what="1000-octet flush blocks")


def case_noflush_streams(eng, n=160_000):
    """ordinary zlib / gzip / deflate streams — no flush markers anywhere.  The block-start finder
    (K0b) splits them at dynamic-block headers, the LZ77 groups run against symbolic history and K6 resolves the
    references across them; results are the oracle's, for every capacity and cut."""
    p = K.enwik_like(n, 0x3B7)
    for level in (1, 6, 9):
        c = zlib.compressobj(level, zlib.DEFLATED, -15)
        raw = c.compress(p) + c.flush()
        assert_same(eng, raw, "deflate", n, what="no-flush deflate level %d" % level)
        assert_same(eng, zlib.compress(p, level), "zlib", n, what="no-flush zlib level %d" % level)
    s = zlib.compress(p, 6)
    out = bytearray(n)
    res = eng.inflate(s, 1, out)
    t = eng.timings()
    assert res.status == 0 and bytes(out) == p and res.adler32 == zlib.adler32(p)
    assert t.n_groups >= 2 and t.n_hgroups >= 1, (t.n_candidates, t.n_groups, t.n_hgroups)
    if len(s) >= 128 << 10 or os.environ.get("TBZ_FIND") == "always":   # (K0b searches streams of at least 48 KiB whose items are large)
        assert t.n_candidates >= 2, t.n_candidates
    assert_same(eng, pygzip.compress(p, 6, mtime=0), "gzip", n, what="no-flush gzip")
    # capacities that end inside an H-group, inside its last 32 KiB, at a group seam ...
    for cap in (n - 1, n // 2, 100_000, 70_001, 33_000, 1):
        assert_same(eng, s, "zlib", cap, what="no-flush zlib, capacity %d" % cap)
    for cut in (len(s) - 1, len(s) - 4, len(s) // 2, len(s) // 3, 3000):
        assert_same(eng, s[:cut], "zlib", n, what="no-flush zlib cut at %d" % cut)
    # Z_SYNC_FLUSH every 4 KiB: small segments that all reach back -> merged into H-groups
    s2, p2, _ = K.zlib_flush_stream(n // 2, block=4096, flush=zlib.Z_SYNC_FLUSH)
    assert_same(eng, s2, "zlib", len(p2), what="sync flush every 4 KiB")
    assert_same(eng, s2, "zlib", len(p2) // 3, what="sync flush every 4 KiB, small buffer")
    # long-lived references: a 20 KiB page repeated (every copy reaches back ~20 KiB, pointers of pointers)
    page = K.xorshift64star_bytes(20_000, 77)
    rep = page * (n // 20_000)
    assert_same(eng, zlib.compress(rep, 6), "zlib", len(rep), what="repeated page")
    assert_same(eng, zlib.compress(bytes(n), 6), "zlib", n, what="zeros (distance-1 runs across every group)")


def _fixed_chain(seed, nblk, maxtok, final_at=None):
    """raw deflate of `nblk` fixed-Huffman blocks of 1..maxtok random tokens each; BFINAL on block `final_at`
    (default: an empty final block at the end).  Returns (stream, plain up to and including the final block)."""
    rng = random.Random(seed)
    w = K.FixedHuffmanWriter()
    out = bytearray()
    fin = None
    for b in range(nblk):
        w.begin_block(final_at == b)
        for _ in range(rng.randrange(1, maxtok + 1)):
            if len(out) > 300 and rng.random() < 0.3:
                ln, d = rng.randrange(3, 259), rng.randrange(1, min(len(out), 32768) + 1)
                w.match(ln, d)
                K._lz_apply(out, ln, d)
            else:
                c = rng.randrange(256)
                w.literal(c)
                out.append(c)
        w.end_block()
        if final_at == b:
            fin = len(out)
    if final_at is None:
        w.begin_block(True)
        w.end_block()
        fin = len(out)
    w.align()
    return w.getvalue(), bytes(out[:fin])


def case_block_starts_found(eng, n_blocks=18, chunk=40_000):
    """K0b finds EVERY dynamic-block header, wherever in a 32-bit word of memory it starts: a stream of `n_blocks`
    blocks whose count is known by construction (deflate's Z_BLOCK flush ends a block and adds nothing — no marker for
    K0), at all four octet alignments of the input, one K1 item per block.  (A first version of the scan read too few
    bits for headers starting in the last three bits of a word: 9 % of the blocks were missed, their items spanned two
    blocks and K1 was as slow as its stragglers.)  The final block (BFINAL = 1) is a candidate too."""
    text = K.enwik_like(n_blocks * chunk, 0x3B8)
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    parts = []
    for k in range(n_blocks):
        parts.append(c.compress(text[k * chunk:(k + 1) * chunk]))
        parts.append(c.flush(zlib.Z_BLOCK if k + 1 < n_blocks else zlib.Z_FINISH))
    s = b"".join(parts)
    assert_same(eng, s, "deflate", len(text), what="%d Z_BLOCK-delimited blocks" % n_blocks)
    searched = len(s) >= 128 << 10 or os.environ.get("TBZ_FIND") == "always"
    d_in, d_out = eng.malloc(len(s) + 80), eng.malloc(len(text) + 64)
    counts = []
    try:
        for shift in range(4):
            eng.h2d(d_in + shift, s)
            res = eng.inflate_device(d_in + shift, len(s), d_out, len(text), T.FORMATS["deflate"])
            t = eng.timings()
            out = bytearray(len(text))
            eng.d2h(out, d_out)
            assert res.status == 0 and bytes(out) == text, shift
            if searched:  # every block but the first (the head item's) is a candidate
                assert t.n_candidates >= n_blocks - 1 and t.n_segments >= n_blocks, (shift, t.n_candidates, t.n_segments)
            counts.append(t.n_candidates)
        # the number of blocks does not depend on where the stream lies in memory (zlib may have split a chunk: the
        # exact count is not known here, its independence of the alignment is)
        assert len(set(counts)) == 1, counts
    finally:
        eng.free(d_in)
        eng.free(d_out)


def case_close_block_starts(eng):
    """Item starts closer than one run-table slot (2^RUN_SHIFT bits): a Z_BLOCK-flushed fixed block of one or two
    literals is 18-27 bits long, so the dynamic block behind it is a K0b candidate right after a flush marker, the
    stream's head or another candidate.  Such candidates are dropped (tbz_k0b_space) and the item before decodes
    through them; before that rule two items shared a run-table slot and valid streams decoded to wrong octets (since
    round 3 an item's first run lives in its result record, so even flush MARKERS a few octets apart cannot collide:
    case_token_density).
    Every position of the marker in a 32-bit word, raw deflate and zlib, plus a mixed-flush fuzz."""
    always = os.environ.get("TBZ_FIND") == "always"
    big = 24_000 if always else 160 << 10   # (the default finder searches streams of at least 128 KiB with large items)
    text = K.enwik_like(big, 7)

    def stream(parts, wbits=-15):
        c = zlib.compressobj(6, zlib.DEFLATED, wbits)
        out = b"".join(c.compress(d) + (c.flush(fl) if fl is not None else b"") for d, fl in parts)
        return out + c.flush()

    for lits in (b"a", b"ab"):
        s = stream([(lits, zlib.Z_BLOCK), (text, None)])
        assert_same(eng, s, "deflate", len(lits) + len(text), what="head + %d-literal fixed block + dynamic block" % len(lits))
    n_var = 12 if always else 2
    for pad in range(n_var):
        t2 = K.enwik_like(big + pad * 7, 11 + pad)
        lits = b"a" if pad & 1 else b"ab"
        for fmt, wb in (("deflate", -15), ("zlib", 15)):
            s = stream([(t2, zlib.Z_SYNC_FLUSH), (lits, zlib.Z_BLOCK), (text, None)], wb)
            assert_same(eng, s, fmt, len(t2) + len(lits) + len(text), what="marker + short fixed block + dynamic block, variant %d %s" % (pad, fmt))
    # two candidates a few bits apart: dynamic block, one-literal fixed blocks, dynamic block
    s = stream([(text, zlib.Z_BLOCK), (b"x", zlib.Z_BLOCK), (b"y", zlib.Z_BLOCK), (text, zlib.Z_BLOCK), (b"z", zlib.Z_PARTIAL_FLUSH), (text, None)])
    assert_same(eng, s, "deflate", 3 * len(text) + 3, what="candidates a few bits apart")
    rng = random.Random(0xB10C)
    flushes = [zlib.Z_BLOCK, zlib.Z_PARTIAL_FLUSH, zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH, None]
    for k in range(5 if always else 2):
        parts, plain = [], b""
        for _ in range(rng.randrange(3, 9)):
            n = rng.choice((1, 2, 3, 5, 40, big // 3, big))
            d = K.enwik_like(n, rng.randrange(1 << 20))
            parts.append((d, rng.choice(flushes)))
            plain += d
        fmt, wb = rng.choice((("deflate", -15), ("zlib", 15)))
        assert_same(eng, stream(parts, wb), fmt, len(plain), what="mixed-flush fuzz %d" % k)


def case_fixed_block_chains(eng):
    """Consecutive fixed-Huffman blocks are decoded THROUGH by the gang kernel (end-of-block + header consumed like a
    token; a lane that starts inside a block assumes it is not the final one): blocks of one to three tokens, blocks
    of hundreds, a final block in the middle with more 'blocks' behind it, cuts and small buffers everywhere."""
    for seed, nblk, maxtok in ((1, 300, 3), (2, 800, 1), (3, 400, 40), (4, 40, 700)):
        s, p = _fixed_chain(seed, nblk, maxtok)
        want = assert_same(eng, s, "deflate", len(p) + 10, what="fixed chain %d" % seed)
        assert want["flag"] == "finished" and want["bytes"] == p
        for cut in (len(s) - 1, len(s) - 2, len(s) // 2, len(s) // 3 + 1, 40, 7):
            assert_same(eng, s[:cut], "deflate", len(p) + 10, what="fixed chain %d cut %d" % (seed, cut))
        for cap in (len(p) - 1, len(p) // 2, 33, 1, 0):
            assert_same(eng, s, "deflate", cap, what="fixed chain %d cap %d" % (seed, cap))
    # BFINAL in the middle: what follows looks like more fixed blocks, but the stream ended
    for seed, nblk, fin in ((5, 400, 200), (6, 400, 17), (7, 900, 899), (8, 200, 0)):
        s, p = _fixed_chain(seed, nblk, 4, final_at=fin)
        want = assert_same(eng, s, "deflate", 200_000, what="final block %d of %d" % (fin, nblk))
        assert want["flag"] == "finished" and want["bytes"] == p
        assert_same(eng, zlib_wrap(s, p), "zlib", 200_000, what="final block %d of %d, zlib" % (fin, nblk))
    # a dynamic block between runs of fixed ones, and a stored one
    s1, p1 = _fixed_chain(9, 200, 5)
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    mid = _mixed_plain(30_000, 3)
    dyn = c.compress(mid) + c.flush(zlib.Z_SYNC_FLUSH)  # ends on a stored block, not final
    # (s1 ends with an empty FINAL fixed block: rebuild without it)
    w = K.FixedHuffmanWriter()
    for b in (b"abc", b"de", b"f" * 5):
        w.begin_block(False)
        for ch in b:
            w.literal(ch)
        w.end_block()
    w.bits(0, 3)  # an empty stored block aligns the stream to an octet
    w.align()
    w.bits(0, 16)
    w.bits(0xFFFF, 16)
    head = w.getvalue()
    tail, ptail = _fixed_chain(10, 300, 3)
    s = head + dyn + tail
    p = b"abcdefffff" + mid + ptail
    want = assert_same(eng, s, "deflate", len(p) + 5, what="fixed / dynamic / stored / fixed")
    assert want["flag"] == "finished" and want["bytes"] == p


def zlib_wrap(raw, plain):
    return b"\x78\x9c" + raw + struct.pack(">I", zlib.adler32(plain))


def case_token_density(eng):
    """The gang kernels' token pool holds one word per TWO input bits (scratch): data whose tokens are denser
    than that — long runs of one octet cost two or three bits per match, stored blocks of an octet or two — is declined
    and decoded again into regions of its own; run 0 of an item lives in its result record, so that items a few octets
    apart (a writer that flushes after every octet) never share a run-table slot.  Same results either way, and the
    same as with pools of one word per bit (TBZ_TOK_FULL=1)."""
    text = _mixed_plain(90_000, 77)
    zeros = bytes(256 << 10)
    cases = []
    cases.append(("zlib", zlib.compress(zeros, 6), len(zeros), "zeros"))
    cases.append(("zlib", zlib.compress(text[:40_000] + bytes(100_000) + text[40_000:] + b"\x55" * 60_000, 6),
                  len(text) + 160_000, "text, zeros, text, a run"))
    cases.append(("deflate", K.stored_stream(text[:3000], max_block=1), 3000, "stored blocks of one octet"))
    cases.append(("deflate", K.stored_stream(text[:5000], max_block=2), 5000, "stored blocks of two octets"))
    for flush in (zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH):
        c = zlib.compressobj(6, zlib.DEFLATED, 15)
        z = bytearray()
        for i in range(300):       # an octet, a flush: token-bearing items 56 bits apart
            z += c.compress(text[i:i + 1]) + c.flush(flush)
        for i in range(40):        # flushes with nothing between them: empty stored blocks back to back
            z += c.flush(flush) if i else b""
        z += c.compress(text[300:]) + c.flush()
        cases.append(("zlib", bytes(z), len(text), "a flush after every octet (%d)" % flush))
    os.environ["TBZ_TOK_FULL"] = "1"
    try:
        e2 = T.Engine(eng.device, lib_path=eng.lib._name)
    finally:
        os.environ.pop("TBZ_TOK_FULL", None)
    os.environ["TBZ_K1_MODE"] = "32"  # a forced flavour hands what it declines straight to the one-lane kernel
    try:
        e3 = T.Engine(eng.device, lib_path=eng.lib._name)
    finally:
        os.environ.pop("TBZ_K1_MODE", None)
    try:
        for k, (fmt, data, n, what) in enumerate(cases):
            w = assert_same(eng, data, fmt, n, what=what)
            assert w["flag"] == "finished", what
            w2 = assert_same(e2, data, fmt, n, what=what + " [one word per bit]")
            assert w2["bytes"] == w["bytes"]
            if k in (0, 2, 4):
                assert_same(e3, data, fmt, n, what=what + " [gangs of 32]")
            assert_same(eng, data, fmt, n // 2, what=what + ", short buffer")
            assert_same(eng, data, fmt, n, end=len(data) * 2 // 3, what=what + ", cut")
        # the zeros went through the second launch (regions of their own), not through a pool of one word per bit
        out = bytearray(len(zeros))
        r = eng.inflate(cases[0][1], FMT["zlib"], out)
        assert r.status == 0 and bytes(out) == zeros and eng.timings().huff_launches >= 2, eng.timings().huff_launches
    finally:
        e2.close()
        e3.close()


def case_scratch_bounds(eng):
    """Device scratch is bounded by what a pass decodes: (1) streams far apart in one buffer pay for their own extent
    only (token pool and run tables are addressed relative to the call's first stream octet, not to the base
    pointer); (2) a batch whose scratch would exceed the pool cap is decoded in several passes over consecutive
    streams, results unchanged; (3) tbz_ctx_trim gives the scratch back."""
    plains = [_mixed_plain(30_000 + 977 * i, 40 + i) for i in range(10)]
    streams = [zlib.compress(p, 6) for p in plains]
    os.environ["TBZ_POOL_CAP_MIB"] = "1"
    try:
        e2 = T.Engine(eng.device, lib_path=eng.lib._name)
    finally:
        os.environ.pop("TBZ_POOL_CAP_MIB", None)
    try:
        outs = [bytearray(len(p)) for p in plains]
        res = e2.inflate_batch(streams, FMT["zlib"], outs)
        t = e2.timings()
        assert t.passes >= 2, t.passes                      # ~150 KB of streams x 20 > 1 MiB
        for r, o, p in zip(res, outs, plains):
            assert r.status == 0 and r.out_len == len(p) and bytes(o) == p and (r.flags & 1)
        # (1) two streams 48 MiB apart in one device buffer: scratch follows the streams, not the gap
        gap = 48 << 20
        d_in = e2.malloc(gap + len(streams[1]) + 64)
        d_out = e2.malloc(len(plains[0]) + len(plains[1]) + 64)
        try:
            e2.h2d(d_in + gap, streams[1])
            r1 = e2.inflate_batch_device(d_in, [gap], [len(streams[1])], d_out, [0], [len(plains[1])], FMT["zlib"])
            assert r1[0].status == 0 and r1[0].out_len == len(plains[1])
            assert e2.timings().scratch_bytes < 64 * len(streams[1]) + (8 << 20), e2.timings().scratch_bytes
            got = bytearray(len(plains[1]))
            e2.d2h(got, d_out)
            assert bytes(got) == plains[1]
        finally:
            e2.free(d_in)
            e2.free(d_out)
        # (3)
        e2.trim()
        out = bytearray(len(plains[2]))
        r = e2.inflate(streams[2], FMT["zlib"], out)          # works again after a trim
        assert r.status == 0 and bytes(out) == plains[2]
        small = e2.timings().scratch_bytes
        e2.trim()
        r = e2.inflate(streams[2], FMT["zlib"], out)
        assert r.status == 0 and e2.timings().scratch_bytes == small
    finally:
        e2.close()


def case_gzip_metadata(eng):
    """the gzip header's metadata through the state's slots (gzip.lisp:17-25, filled by gzip.lisp:123-241:
    method, flags, mtime, level, operating system, extra / name / comment), one-shot and as the header arrives
    three octets at a time — against the oracle's restatement of the same slots"""
    plain = _mixed_plain(20_000, 12)

    def member(name=None, comment=None, extra=None, mtime=0, xfl=0, os_=3, hcrc=False, text=False):
        flg = ((1 if text else 0) | (2 if hcrc else 0) | (4 if extra is not None else 0) | (8 if name is not None else 0) |
               (16 if comment is not None else 0))
        h = bytes([0x1f, 0x8b, 8, flg]) + struct.pack("<I", mtime) + bytes([xfl, os_])
        if extra is not None:
            h += struct.pack("<H", len(extra)) + extra
        if name is not None:
            h += name + b"\0"
        if comment is not None:
            h += comment + b"\0"
        if hcrc:
            h += struct.pack("<H", zlib.crc32(h) & 0xffff)
        c = zlib.compressobj(6, zlib.DEFLATED, -15)
        return h + c.compress(plain) + c.flush() + struct.pack("<II", zlib.crc32(plain), len(plain)), len(h)

    variants = [dict(), dict(name=b"file.txt", mtime=1_700_000_000, xfl=2),
                dict(comment="caf\xe9 au lait".encode("latin-1"), name="na\u00efve".encode("utf-8"), os_=11, xfl=4),
                dict(extra=b"\x01\x02abc" * 9, hcrc=True, text=True, os_=200, xfl=7, comment=b"c"),
                # an extra field long enough for the gang to take its CRC in parallel (K1g), name + comment behind it
                dict(extra=bytes((i * 7 + 3) & 0xff for i in range(3001)), hcrc=True, name=b"big-extra", comment=b"x" * 300),
                dict(extra=bytes(range(256)) * 255, hcrc=True),
                dict(name=b"", comment=b"", extra=b"")]
    for kw in variants:
        g, hlen = member(**kw)
        so, se = O.State(FMT["gzip"], bytearray(len(plain))), A.make_gzip_state(bytearray(len(plain)))
        O.decompress(O.make_octet_vector_context(g), so)
        A.decompress(A.make_octet_vector_context(g), se, engine=eng)
        assert A.finished(se) and bytes(se.output_buffer) == plain
        assert se.gzip_meta() == so.gzip_meta(), (kw, se.gzip_meta(), so.gzip_meta())
        m = se.gzip_meta()
        assert m["compression_method"] == "deflate" and (m["name"] is None) == ("name" not in kw)
        if kw.get("os_") == 11:  # not utf-8: substituted, NOT re-read as iso-8859-1 (gzip.lisp:236-239 with :errorp nil)
            assert m["comment"] == "caf\ufffd au lait" and m["name"] == "na\u00efve", (m["comment"], m["name"])
        if hlen > 1000:  # (a damaged octet inside the long extra field: the header CRC says so, as the reference's does)
            bad = bytearray(g)
            bad[hlen // 2] ^= 0x40
            assert_same(eng, bytes(bad), "gzip", len(plain), what="header crc over a long extra field")
            assert_same(eng, g[:hlen // 2], "gzip", len(plain), what="input ends inside a long extra field")
            continue
        # the header in 3-octet chunks, then the rest
        so, se = O.State(FMT["gzip"], bytearray(len(plain))), A.make_gzip_state(bytearray(len(plain)))
        pos = 0
        while pos < len(g):
            end = min(len(g), pos + 3) if pos < hlen + 6 else len(g)
            ro = O.decompress(O.make_octet_vector_context(g, start=pos, end=end), so)
            re_ = A.decompress(A.make_octet_vector_context(g, start=pos, end=end), se, engine=eng)
            assert ro == re_ and (O.finished(so), O.input_underrun(so)) == (A.finished(se), A.input_underrun(se))
            assert se.gzip_meta() == so.gzip_meta(), (kw, pos, se.gzip_meta(), so.gzip_meta())
            pos = end
        assert A.finished(se) and bytes(se.output_buffer) == plain


def case_history_across_groups(eng):
    """Z_SYNC_FLUSH stream whose middle segment copies nothing from before itself (incompressible octets: it opens a
    LZ77 group of its own) while the segment after it copies from the FIRST one, i.e. from before its predecessor's
    group: the groups must be merged back until the history is covered.  (Found by the corruption fuzzer on the
    device as a repaired block reaching across independent segments; this is the valid-stream form.)"""
    rng = random.Random(77)
    a = K.enwik_like(20_000, seed=31)
    for mid_len in (1, 40, 300, 5000):
        mid = bytes(rng.randrange(256) for _ in range(mid_len))
        c = zlib.compressobj(9, zlib.DEFLATED, 15)
        blob = c.compress(a) + c.flush(zlib.Z_SYNC_FLUSH) + c.compress(mid) + c.flush(zlib.Z_SYNC_FLUSH)
        blob += c.compress(a[3000:15000]) + c.flush(zlib.Z_SYNC_FLUSH) + c.compress(mid + a[:4000]) + c.flush()
        plain = a + mid + a[3000:15000] + mid + a[:4000]
        w = assert_same(eng, blob, "zlib", len(plain) + 10, what="history across groups, middle %d" % mid_len)
        assert w["flag"] == "finished" and w["bytes"] == plain
        assert_same(eng, blob, "zlib", len(a) + mid_len + 5000, what="... with overflow")


def case_configs_1_3_5(eng, adv_total=160 << 10):
    for two in (False, True):  # config 1: one / two stored blocks
        s, p = K.config1_stream(two)
        w = assert_same(eng, s, "deflate", len(p), what="config1")
        assert w["bytes"] == p
    # config 3: per-member parity through the batch entry point
    blob, offs, plains = K.gzip_members(5, 24 << 10)
    ends = offs[1:] + [len(blob)]
    outs = [bytearray(len(p)) for p in plains]
    res = eng.inflate_batch([blob[o:e] for o, e in zip(offs, ends)], 2, outs)
    for r, o, p in zip(res, outs, plains):
        assert r.status == 0 and bytes(o) == p and r.crc32 == zlib.crc32(p) and r.trailer_isize == len(p)
    # 3bz decodes exactly ONE member and ignores what follows (gzip.lisp:277-286)
    w = assert_same(eng, blob, "gzip", len(plains[0]), what="multi-member: first member only")
    assert w["bytes"] == plains[0]
    w = assert_same(eng, blob, "gzip", len(plains[2]), start=offs[2], what="member 2 via :start")
    assert w["bytes"] == plains[2]
    # config 5: adversarial LZ77, one sequential segment and with flush points
    for ff in (0, 64 << 10):
        s, p = K.adversarial_stream(total=adv_total, full_flush_every=ff)
        w = assert_same(eng, s, "zlib", len(p), what="config5 ff=%d" % ff)
        assert w["bytes"] == p


def case_overflow_and_underrun(eng):
    plain = K.enwik_like(60_000, seed=5)
    s, p, a = K.zlib_flush_stream(48 << 10)
    blobs = [("zlib", zlib.compress(plain, 6), plain), ("gzip", pygzip.compress(plain, 6, mtime=0), plain),
             ("zlib", s, p), ("deflate", K.config1_stream(True)[0], K.config1_stream(True)[1])]
    rng = random.Random(8)
    for fmt, blob, pl in blobs:
        # output too small: flag, count == cap, buffer holds the correct prefix (deflate.lisp:254-269,:693-697)
        for cap in (0, 1, 2, 257, 258, 259, 4095, 16384, 16385, len(pl) - 1, len(pl)):
            w = assert_same(eng, blob, fmt, cap, what="%s cap %d" % (fmt, cap))
            assert w["flag"] == ("finished" if cap >= len(pl) else "overflow")
        # truncated input at interesting cut points: header, mid-block, trailer
        cuts = {0, 1, 2, 3, 5, 9, 10, 11, 17, len(blob) - 9, len(blob) - 8, len(blob) - 5, len(blob) - 4,
                len(blob) - 3, len(blob) - 1}
        cuts |= {rng.randrange(len(blob)) for _ in range(6)}
        for c in sorted(x for x in cuts if 0 <= x < len(blob)):
            w = assert_same(eng, blob, fmt, len(pl) + 10, end=c, what="%s cut %d" % (fmt, c))
            assert w["flag"] in ("underrun", "error"), (fmt, c, w["flag"])
    # both at once: inside a stored block the reference asks for output space BEFORE it asks for input
    # (copy-byte-or-fail, deflate.lisp:538-573), so a payload cut off exactly where the buffer is full is
    # output-overflow; everywhere else a token needs its input first (input-underrun)
    z0 = zlib.compress(plain[:9000], 0)          # header, stored block (5-octet block header), adler32
    for cut in (7, 8, 100, 4000, len(z0) - 5):
        have = cut - 7                            # payload octets present
        for cap in (have - 1, have, have + 1):
            if cap >= 0:
                w = assert_same(eng, z0, "zlib", cap, end=cut, what="stored cut %d cap %d" % (cut, cap))
                assert w["flag"] == ("underrun" if cap > have else "overflow"), (cut, cap, w["flag"])
    zc = zlib.compress(plain[:9000], 6)
    for cut in (400, 1000):
        n_out = oracle_oneshot(zc, "zlib", 20000, end=cut)["offset"]
        assert n_out > 0
        for cap in (n_out - 1, n_out, n_out + 1):
            w = assert_same(eng, zc, "zlib", cap, end=cut, what="huffman cut %d cap %d" % (cut, cap))
            assert w["flag"] == ("overflow" if cap < n_out else "underrun"), (cut, cap, w["flag"])
    # decompress-vector's own errors (api.lisp:41-47,:55)
    z = zlib.compress(plain)
    for kw, code in (({"output": bytearray(10)}, -21), ({"end": len(z) - 3, "output": bytearray(len(plain))}, -20),
                     ({"end": len(z) - 3}, -20)):
        try:
            A.decompress_vector(z, format="zlib", engine=eng, **kw)
            raise AssertionError("expected an error")
        except A.ThreeBzError as e:
            assert e.code == code, (e.code, code)


def case_errors(eng):
    plain = K.enwik_like(20_000, seed=3)
    z = zlib.compress(plain, 6)
    g = pygzip.compress(plain, 6, mtime=0)
    mut = []
    mut.append(("zlib", z[:-1] + bytes([z[-1] ^ 1])))          # adler mismatch (zlib.lisp:95)
    mut.append(("zlib", b"\x78\x9d" + z[2:]))                  # header check (zlib.lisp:20-24)
    mut.append(("zlib", b"\x79\x9c" + z[2:]))                  # CM != 8
    mut.append(("zlib", b"\x88\x1c" + z[2:]))                  # CINFO > 7
    mut.append(("zlib", b"\x78\xbb" + z[2:]))                  # FDICT
    mut.append(("gzip", b"\x1f\x8c" + g[2:]))                  # magic
    mut.append(("gzip", g[:2] + b"\x09" + g[3:]))              # CM
    mut.append(("gzip", g[:3] + b"\x20" + g[4:]))              # reserved flag
    mut.append(("gzip", g[:-8] + bytes([g[-8] ^ 1]) + g[-7:]))  # crc mismatch (gzip.lisp:93)
    mut.append(("gzip", g[:-1] + bytes([g[-1] ^ 1])))          # ISIZE is NOT checked (gzip.lisp:278)
    mut.append(("deflate", b"\x07"))                           # BTYPE 3
    mut.append(("deflate", bytes.fromhex("0104089fac")))       # LEN/NLEN
    # distance before start of output, no window (deflate.lisp:345)
    w = K.FixedHuffmanWriter()
    w.begin_block(True)
    w.literal(65)
    w.match(3, 2)
    w.end_block()
    w.align()
    mut.append(("deflate", w.getvalue()))
    # ... and the same error against output-overflow: 3bz meets whichever comes first in the octet order, i.e. the
    # error iff the offending match STARTS at or before the end of the buffer (copy-history checks the source
    # before it copies, deflate.lisp:343-345).  One segment, and the match in the second of two segments.
    for lead in (0, 10):
        w = K.FixedHuffmanWriter()
        if lead:
            w.begin_block(False)
            for i in range(lead):
                w.literal(97 + i)
            w.end_block()
            w.bits(0, 3)  # empty stored block = sync-flush marker: the second segment starts after it
            w.align()
            w.buf += b"\x00\x00\xff\xff"
        w.begin_block(True)
        for i in range(20):
            w.literal(65 + i)
        w.match(5, 4)              # fine
        w.match(5, lead + 26)      # starts at octet lead+25 and reaches one octet before the stream's first
        for i in range(20):
            w.literal(48 + i)
        w.end_block()
        w.align()
        blob = w.getvalue()
        for cap in (0, 1, lead, lead + 19, lead + 24, lead + 25, lead + 26, lead + 29, lead + 30, lead + 31, lead + 60):
            assert_same(eng, blob, "deflate", cap, what="distance error vs overflow, lead %d cap %d" % (lead, cap))
        assert assert_same(eng, blob, "deflate", lead + 25)["flag"] == "error"
        assert assert_same(eng, blob, "deflate", lead + 24)["flag"] == "overflow"
    # bit flips inside compressed data: whatever 3bz does (error, underrun, garbage+adler error), we do
    rng = random.Random(11)
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    raw = c.compress(plain) + c.flush()
    for _ in range(12):
        b = bytearray(raw)
        i = rng.randrange(len(b))
        b[i] ^= 1 << rng.randrange(8)
        mut.append(("deflate", bytes(b)))
    for fmt, blob in mut:
        assert_same(eng, blob, fmt, len(plain) + 300, what="mutation %s %s" % (fmt, blob[:6].hex()))


def case_false_markers(eng):
    """00 00 FF FF inside payload bytes: every hit is a speculative marker the chain must reject"""
    pat = b"\x00\x00\xff\xff"
    payload = (pat * 50 + b"ABCD" + pat + b"x" * 100 + b"\x00" + pat + b"\xff\xff" + pat * 3) * 20
    s = K.stored_stream(payload, max_block=1000)  # stored blocks: the pattern survives verbatim
    w = assert_same(eng, s, "deflate", len(payload), what="false markers in stored blocks")
    assert w["bytes"] == payload
    res = eng.inflate(s, 0, bytearray(len(payload)))
    assert res.status == 0
    # (one small stream alone is the one-launch path's: a chain of stored blocks is copied before anything looks for
    # markers.  Two of them in a batch go the general way, where every hit is an item start the chain has to reject)
    o2 = [bytearray(len(payload)), bytearray(len(payload))]
    r2 = eng.inflate_batch([s, s], 0, o2)
    assert all(r.status == 0 and r.out_len == len(payload) for r in r2) and bytes(o2[0]) == bytes(o2[1]) == payload
    assert eng.timings().fixup_rounds >= 1
    # a real flush stream whose plaintext is full of the pattern, level 0 (stored) and level 6
    plain = (pat * 10 + K.enwik_like(3000, 9)) * 12
    for level in (0, 6):
        c = zlib.compressobj(level, zlib.DEFLATED, 15)
        blob = b""
        for i in range(0, len(plain), 5000):
            blob += c.compress(plain[i:i + 5000]) + c.flush(zlib.Z_FULL_FLUSH)
        blob += c.flush()
        w = assert_same(eng, blob, "zlib", len(plain), what="pattern-rich flush stream L%d" % level)
        assert w["bytes"] == plain
    # pattern straddling the end of input / right at the end
    for tail in (pat, pat[:3], b"\x00" + pat):
        blob = K.stored_stream(b"hello" + tail)
        assert_same(eng, blob, "deflate", 64, what="marker at the very end")


def case_device_buffers(eng, n=64 << 10):
    """device-resident entry points with unaligned bases and a multi-stream batch"""
    s, p, a = K.zlib_flush_stream(n)
    for mis_in, mis_out in ((0, 0), (1, 3), (3, 1), (2, 15), (5, 16)):
        d_in = eng.malloc(len(s) + 64)
        d_out = eng.malloc(n + 64)
        eng.h2d(d_in + mis_in, s)
        res = eng.inflate_device(d_in + mis_in, len(s), d_out + mis_out, n, 1)
        got = bytearray(n)
        eng.d2h(got, d_out + mis_out)
        eng.free(d_in)
        eng.free(d_out)
        assert res.status == 0 and res.out_len == n and bytes(got) == p, (mis_in, mis_out, res.status)
        assert res.adler32 == a
    # batch: three streams of different formats' worth of data at odd offsets in one buffer
    plains = [K.enwik_like(20_000, 21), K.enwik_like(7_777, 22), b"", K.enwik_like(33_333, 23)]
    blobs = [zlib.compress(x, 6) for x in plains]
    in_offs, pos = [], 3
    for b in blobs:
        in_offs.append(pos)
        pos += len(b) + 5
    out_offs, opos = [], 1
    for x in plains:
        out_offs.append(opos)
        opos += len(x) + 7
    d_in = eng.malloc(pos + 64)
    d_out = eng.malloc(opos + 64)
    for o, b in zip(in_offs, blobs):
        eng.h2d(d_in + o, b)
    res = eng.inflate_batch_device(d_in, in_offs, [len(b) for b in blobs], d_out, out_offs,
                                   [len(x) for x in plains], 1)
    whole = bytearray(opos)
    eng.d2h(whole, d_out)
    eng.free(d_in)
    eng.free(d_out)
    for r, o, x in zip(res, out_offs, plains):
        assert r.status == 0 and r.out_len == len(x) and bytes(whole[o:o + len(x)]) == x
        assert r.adler32 == zlib.adler32(x)


def case_checksum_kernels(eng):
    """K4/K5 against the oracle's adler32/ub64 and crc32/table incl. the chaining convention"""
    rng = random.Random(5)
    base = (bytes(rng.getrandbits(8) for _ in range(70_000)) + b"\xff" * 70_000) * 5  # 700 000: several 256 KiB chunks
    for n in (0, 1, 3, 4, 255, 256, 257, 1023, 65535, 65536, 65537, 131072 + 5, 262144, 262145, 524288 + 77, len(base) - 8):
        for mis in (0, 1, 7):
            if n + mis > len(base):
                continue
            d = eng.malloc(n + 64)
            buf = base[mis:mis + n]
            eng.h2d(d + mis, buf)
            s1, s2 = eng.adler32_device(d + mis, n)
            assert (s1, s2) == O.adler32(buf), (n, mis)
            assert eng.crc32_device(d + mis, n) == O.crc32(buf), (n, mis)
            # chaining (zlib.lisp:97-102, gzip.lisp:80-81)
            c1, c2 = eng.adler32_device(d + mis, n, 12345, 54321)
            assert (c1, c2) == O.adler32(buf, 12345, 54321)
            assert eng.crc32_device(d + mis, n, 0xDEADBEEF) == O.crc32(buf, 0xDEADBEEF)
            eng.free(d)


def case_deep_codes(eng, n_tokens=12000):
    """hand-built dynamic blocks with 15-bit codes on most tokens (second-level table lookups), and a
    distance code whose long prefixes need more second-level entries than the engine's pool holds (its
    long codes take the exact step); whole, truncated inside long codes, and with a short output buffer"""
    for ov in (False, True):
        s, p = K.deep_code_stream(n_tokens=n_tokens, seed=11 + ov, dist_overflow=ov)
        assert zlib.decompressobj(-15).decompress(s) == p
        w = assert_same(eng, s, "deflate", len(p), what="deep codes ov=%d" % ov)
        assert w["flag"] == "finished" and w["bytes"] == p
        assert_same(eng, s, "deflate", len(p) // 3, what="deep codes, short buffer ov=%d" % ov)
        for cut in (len(s) - 1, len(s) // 2, len(s) // 2 + 1, 300, 115, 114, 113):
            assert_same(eng, s, "deflate", len(p), end=cut, what="deep codes cut at %d ov=%d" % (cut, ov))
        z = zlib.compress(p[:50000], 6)  # same octets through zlib's own (shallow) codes, as a container stream
        assert_same(eng, z, "zlib", 50000, what="deep plain via zlib")
    # tokens as dense as bits (1-bit literal codes): the gang kernel hands such an item to the one-lane kernel
    s, p = K.dense_literal_stream()
    assert zlib.decompressobj(-15).decompress(s) == p
    assert_same(eng, s, "deflate", len(p), what="dense literals")
    assert_same(eng, s, "deflate", len(p), end=len(s) // 2, what="dense literals, cut")
    s, p = K.dense_literal_stream(n_lits=40 * 11, blocks=40)  # short dense blocks: runs that end ragged
    assert zlib.decompressobj(-15).decompress(s) == p
    assert_same(eng, s, "deflate", len(p), what="dense literals, short blocks")


def _chunked_lockstep(eng, blob, fmt, in_steps, out_sizes, what, max_calls=2000):
    """feed `blob` to the oracle and to the engine in the same input chunks / output buffers and compare every call"""
    mk_e = {"deflate": A.make_deflate_state, "zlib": A.make_zlib_state, "gzip": A.make_gzip_state}[fmt]
    so, se = O.State(FMT[fmt]), mk_e()
    got_o, got_e = bytearray(), bytearray()
    bo, be = bytearray(out_sizes[0]), bytearray(out_sizes[0])
    O.replace_output_buffer(so, bo)
    A.replace_output_buffer(se, be)
    oi, pos, step_i, guard = 1, 0, 0, 0
    while not O.finished(so):
        guard += 1
        assert max_calls is None or guard < max_calls, what
        step = in_steps[step_i % len(in_steps)]
        step_i += 1
        end = min(len(blob), pos + step)
        co = O.make_octet_vector_context(blob, start=pos, end=end)
        ce = A.make_octet_vector_context(blob, start=pos, end=end)
        while True:
            eo = ee = None
            try:
                ro = O.decompress(co, so)
            except O.OracleError as e:
                eo = e.code
            try:
                re_ = A.decompress(ce, se, engine=eng)
            except A.ThreeBzError as e:
                ee = e.code
            if eo is not None or ee is not None:   # a damaged stream: the same error in the same call
                assert eo == ee, (what, guard, "error", ee, eo)
                _chunked_lockstep.last_state = se
                _chunked_lockstep.last_error = eo
                return bytes(got_o)
            flags_o = (O.finished(so), O.input_underrun(so), O.output_overflow(so))
            flags_e = (A.finished(se), A.input_underrun(se), A.output_overflow(se))
            assert flags_e == flags_o and re_ == ro, (what, guard, flags_e, flags_o, re_, ro)
            assert bytes(be[:se.output_offset]) == bytes(bo[:so.output_offset]), (what, guard, "buffer differs")
            if O.output_overflow(so):
                got_o += bo[:ro]
                got_e += be[:re_]
                size = out_sizes[oi % len(out_sizes)]
                oi += 1
                bo, be = bytearray(size), bytearray(size)
                O.replace_output_buffer(so, bo)
                A.replace_output_buffer(se, be)
                continue
            break
        pos = end
        if pos >= len(blob) and not O.finished(so):
            break  # truncated stream: both sit in input-underrun
    got_o += bo[:so.output_offset]
    got_e += be[:se.output_offset]
    assert bytes(got_e) == bytes(got_o), what
    _chunked_lockstep.last_state = se
    _chunked_lockstep.last_error = None
    return bytes(got_o)


def case_chunked_resume(eng, n=60_000):
    """the chunked protocol (deflate.lisp:114-137): more input after input-underrun, a new buffer after
    output-overflow — every call compared with the oracle (flags, return value, buffer contents)"""
    plain = _mixed_plain(n, 5)
    rng = random.Random(2024)
    fs, fp, _ = K.zlib_flush_stream(n, block=8192)
    blobs = [("zlib", zlib.compress(plain, 6), plain), ("gzip", pygzip.compress(plain, 6, mtime=0), plain),
             ("deflate", zlib.compress(plain, 1)[2:-4], plain), ("zlib", fs, fp)]
    for fmt, blob, plain in blobs:
        big = len(plain) + 10
        # input chunks only
        steps = [rng.randrange(2000, 9000) for _ in range(8)]
        assert _chunked_lockstep(eng, blob, fmt, steps, [big], "%s input chunks" % fmt) == plain
        # output buffers only
        sizes = [rng.randrange(3000, 30000) for _ in range(8)]
        assert _chunked_lockstep(eng, blob, fmt, [len(blob)], sizes, "%s output buffers" % fmt) == plain
        # both, plus a first chunk that ends inside the container header
        steps = [1] + [rng.randrange(3000, 12000) for _ in range(6)]
        sizes = [rng.randrange(5000, 40000) for _ in range(6)]
        assert _chunked_lockstep(eng, blob, fmt, steps, sizes, "%s both" % fmt) == plain
    # truncated stream fed in chunks: ends in input-underrun on both sides
    _chunked_lockstep(eng, blobs[0][1][: len(blobs[0][1]) // 2], "zlib", [5000], [n + 10], "truncated zlib in chunks")
    # stored payload cut off exactly where an output buffer is full (4093 - 7 = 3 x 1362): the reference reports
    # output-overflow there (it asks for space before input, deflate.lisp:538-573), then input-underrun
    z0 = zlib.compress(fp[:9000], 0)
    assert _chunked_lockstep(eng, z0, "zlib", [4093, 3000], [1362], "stored cut at a full buffer") == fp[:9000]
    # where the session resumes (tbz_session_*, include/tbz_amd.h): at the start of the block in which the input ran
    # out — raw blocks from there, checksum continued, container trailer compared by the session — in all three
    # containers; a Z_SYNC_FLUSH stream (blocks copy from before the resume point) resumes just the same: the 32 KiB
    # before the resume point are the session's window
    def flushed(wbits, mode):
        c = zlib.compressobj(6, zlib.DEFLATED, wbits)
        return b"".join(c.compress(fp[i:i + 8192]) + c.flush(mode) for i in range(0, len(fp), 8192)) + c.flush()
    for fmt, wbits in (("zlib", 15), ("gzip", 31), ("deflate", -15)):
        blob = flushed(wbits, zlib.Z_FULL_FLUSH)
        for steps, sizes in (([7000], [len(fp) + 10]), ([5000, 11000, 3000], [20000, 9000]), ([len(blob) - 3, 1], [len(fp) + 10])):
            assert _chunked_lockstep(eng, blob, fmt, steps, sizes, "%s full-flush resume" % fmt) == fp
            st = _chunked_lockstep.last_state   # (boundary_out: octets of output before the session's resume point)
            assert st.result.boundary_out > 0 or len(steps) == 2, (fmt, steps, st.result.boundary_out)  # (trailer-only chunks: nothing to move)
        bad = blob[:-1] + bytes([blob[-1] ^ 1]) if fmt != "deflate" else None
        if fmt == "zlib":   # a checksum mismatch: same error, same call as the reference
            so, se = O.State(FMT[fmt], bytearray(len(fp) + 10)), A.make_zlib_state(bytearray(len(fp) + 10))
            for lo in range(0, len(bad), 9000):
                hi = min(len(bad), lo + 9000)
                eo = ee = None
                try:
                    O.decompress(O.make_octet_vector_context(bad, start=lo, end=hi), so)
                except O.OracleError as e:
                    eo = e.code
                try:
                    A.decompress(A.make_octet_vector_context(bad, start=lo, end=hi), se, engine=eng)
                except A.ThreeBzError as e:
                    ee = e.code
                assert eo == ee, (lo, eo, ee)
            assert ee == -11
    blob = flushed(15, zlib.Z_SYNC_FLUSH)
    assert _chunked_lockstep(eng, blob, "zlib", [7000], [len(fp) + 10], "sync-flush resume") == fp
    assert _chunked_lockstep.last_state.result.boundary_out > 0


def case_fuzz(eng, seed=7, n=30):
    """random corruptions (bit flips, truncation, inserted 00 00 FF FF markers, overwritten octets) of four stream
    shapes: whatever the reference makes of them — finished with some octets, underrun, one of its errors — the engine
    must make the same.  Seed 7 / case 26 is the stream that exposed the one parity bug of the round (a repair item
    starts mid-octet; its run offsets were taken from the unaligned start)."""
    rng = random.Random(seed)
    plain = _mixed_plain(60000, 9)
    bases = [("zlib", zlib.compress(plain, 6)), ("zlib", K.zlib_flush_stream(50000, block=4096)[0]),
             ("deflate", zlib.compress(plain, 1)[2:-4]),
             ("zlib", K.zlib_flush_stream(30000, block=1000, flush=zlib.Z_SYNC_FLUSH)[0])]
    for k in range(n):
        fmt, b = rng.choice(bases)
        b = bytearray(b)
        for _ in range(rng.randrange(1, 4)):
            mode = rng.randrange(4)
            if mode == 0:
                b[rng.randrange(len(b))] ^= 1 << rng.randrange(8)
            elif mode == 1:
                b = b[: rng.randrange(1, max(2, len(b)))]
            elif mode == 2:
                i = rng.randrange(len(b))
                b[i:i] = bytes([0, 0, 255, 255])
            else:
                b[rng.randrange(len(b))] = rng.randrange(256)
        assert_same(eng, bytes(b), fmt, 70000, what="fuzz seed %d case %d" % (seed, k))


def case_container_headers(eng):
    """zlib / gzip framing octet by octet (zlib.lisp:14-37, gzip.lisp:113-266): every truncation of the header and of
    the trailer, and damage at every header position (also cut right after the damaged octet: the reference reads
    ID1+ID2 and CM+FLG as PAIRS, so one octet of a pair is input-underrun whatever it holds) — with and without the
    optional gzip fields (FEXTRA, FNAME, FCOMMENT, FHCRC)."""
    plain = b"hello world, hello gzip! " * 20
    raw = zlib.compress(plain, 6)[2:-4]
    blobs = []
    for flg in (0x00, 0x1e, 0x04, 0x0a):
        hdr = bytes([0x1f, 0x8b, 8, flg, 1, 2, 3, 4, 0, 3])
        if flg & 4:
            hdr += struct.pack("<H", 5) + b"extra"
        if flg & 8:
            hdr += b"name.txt\0"
        if flg & 16:
            hdr += b"a comment\0"
        if flg & 2:
            hdr += struct.pack("<H", zlib.crc32(hdr) & 0xffff)
        blobs.append(("gzip", len(hdr), hdr + raw + struct.pack("<II", zlib.crc32(plain), len(plain))))
    blobs.append(("zlib", 2, zlib.compress(plain, 6)))
    for fmt, nh, g in blobs:
        w = assert_same(eng, g, fmt, 1000)
        assert w["flag"] == "finished" and w["bytes"] == plain
        for cut in list(range(0, nh + 4)) + list(range(len(g) - 9, len(g))):
            assert_same(eng, g[:cut], fmt, 1000, what="%s cut %d" % (fmt, cut))
        for pos in range(nh):
            for val in (0, 0xff, g[pos] ^ 1, g[pos] ^ 0x20):
                b = bytearray(g)
                b[pos] = val
                assert_same(eng, bytes(b), fmt, 1000, what="%s octet %d = %d" % (fmt, pos, val))
                assert_same(eng, bytes(b[:pos + 1]), fmt, 1000, what="%s octet %d = %d, cut after it" % (fmt, pos, val))


def case_gzip_members(eng, n_members=8, max_len=12_000, n_false=300):
    """many-member gzip file (SURVEY §8d config 3 as a public call, §8f-4): member starts are found speculatively
    (every 1f 8b 08 is a candidate, a candidate range is a member iff it finishes having consumed exactly its
    octets).  Parity is per member: member i must be what the reference yields for
    (decompress-vector v :format :gzip :start offset_i) — checked against the oracle at the true offsets."""
    rng = random.Random(0x3B3)
    plains = [K.enwik_like(rng.randrange(1, max_len), seed=100 + i) for i in range(n_members)] + [b""]
    plains.insert(3, A.GZIP_MAGIC * n_false)   # a stored member made of magics: false candidates inside it
    plains.insert(7, bytes(rng.randrange(256) for _ in range(3000)))
    parts = [pygzip.compress(p, 0 if p[:3] == A.GZIP_MAGIC else rng.choice([1, 6, 9]), mtime=0) for p in plains]
    blob = b"".join(parts)
    got = A.decompress_gzip_members(blob, engine=eng)
    assert len(got) == len(plains)
    off = 0
    for g, p, part in zip(got, plains, parts):
        w = oracle_oneshot(blob, "gzip", len(p) + 8, start=off)
        assert w["flag"] == "finished" and w["bytes"] == p == bytes(g), ("member at", off)
        off += len(part)
    # the same file resident in device memory: tbz_inflate_gzip_members_device — offsets found, never given
    room = sum(len(p) for p in plains) + 16 * len(plains) + max(len(p) for p in plains) * 2 + 4096
    d_in, d_out = eng.malloc(len(blob) + 64), eng.malloc(room + 64)
    try:
        eng.h2d(d_in, blob)
        res, ioffs, ooffs = eng.inflate_gzip_members_device(d_in, len(blob), d_out, room, 64)
        assert len(res) == len(plains), (len(res), len(plains))
        host = bytearray(room)
        eng.d2h(host, d_out)
        off = 0
        for r, io_, oo_, p, part in zip(res, ioffs, ooffs, plains, parts):
            assert r.status == 0 and io_ == off and r.out_len == len(p) and r.in_consumed == len(part), (off, r.status, r.out_len)
            assert r.crc32 == zlib.crc32(p) and bytes(host[oo_:oo_ + len(p)]) == p, ("member at", off)
            off += len(part)
        # max_members cuts the walk; a buffer with no room for the member that must be decoded on its own says so
        res2, _, _ = eng.inflate_gzip_members_device(d_in, len(blob), d_out, room, 2)
        assert len(res2) == 2 and all(r.status == 0 for r in res2)
        res3, _, _ = eng.inflate_gzip_members_device(d_in, len(blob), d_out, 64, 64)
        assert any(r.status == 2 for r in res3), [r.status for r in res3]
    finally:
        eng.free(d_in)
        eng.free(d_out)
    assert [bytes(m) for m in A.decompress_gzip_members(blob + b"\x00trailing", engine=eng)] == plains
    assert [bytes(m) for m in A.decompress_gzip_members(b"xx" + blob, start=2, engine=eng)] == plains
    # a member whose ISIZE understates (3bz does not check ISIZE, gzip.lisp:277-286; members over 4 GiB do it by
    # construction), followed by good members: its range overflows the buffer sized by the lie, and the second chance
    # must not be taken in the NEXT member's buffer — that member has decoded correctly there (it was
    # delivered with status 0 and a verified crc over octets the failed merge had overwritten)
    lp = [K.enwik_like(9000, seed=301), K.enwik_like(7000, seed=302), K.enwik_like(5000, seed=303)]
    lparts = [pygzip.compress(q, 6, mtime=0) for q in lp]
    lparts[0] = lparts[0][:-4] + struct.pack("<I", 1500)
    lblob = b"".join(lparts)
    got = A.decompress_gzip_members(lblob, engine=eng)
    assert [bytes(m) for m in got] == lp, "lying ISIZE: host variant"
    room = sum(len(q) for q in lp) * 2 + 4096
    d_in, d_out = eng.malloc(len(lblob) + 64), eng.malloc(room + 64)
    try:
        eng.h2d(d_in, lblob)
        res, ioffs, ooffs = eng.inflate_gzip_members_device(d_in, len(lblob), d_out, room, 64)
        host = bytearray(room)
        eng.d2h(host, d_out)
        assert len(res) == 3
        for r, oo_, q in zip(res, ooffs, lp):
            assert r.status == 0 and r.out_len == len(q) and r.crc32 == zlib.crc32(q), (r.status, r.out_len)
            assert bytes(host[oo_:oo_ + len(q)]) == q, "lying ISIZE: device variant"
        # no room above the ranges for the member that must be decoded on its own: it says so, and the members already
        # delivered keep their octets (place() handed out space inside the ranges)
        tight = 1500 + 16 + len(lp[1]) + 16 + len(lp[2]) + 16 + 2000
        res, ioffs, ooffs = eng.inflate_gzip_members_device(d_in, len(lblob), d_out, tight, 64)
        assert res[0].status == 2 and len(res) == 1, [r.status for r in res]
    finally:
        eng.free(d_in)
        eng.free(d_out)
    # a damaged / truncated member raises what the one-member call at its offset raises
    for bad, start in ((blob[:-3], len(blob) - len(parts[-1])), (blob[:len(parts[0]) - 8] + b"\x00" + blob[len(parts[0]) - 7:], 0),
                       (b"\x1f\x8c" + blob[2:], 0)):
        w = oracle_oneshot(bad, "gzip", max(len(q) for q in plains) + 8, start=start)
        try:
            A.decompress_gzip_members(bad, engine=eng)
            raise AssertionError("damaged member accepted")
        except A.ThreeBzError as e:
            if w["flag"] == "error":
                assert e.code == w["code"], (e.code, w["code"])
            else:
                assert w["flag"] == "underrun" and e.code == -20


def case_stream_contexts(eng, n=50_000):
    """make-octet-stream-context / %resync-file-stream (io-common.lisp:47-63): a stream that sits in the middle of a file —
    octets before it, another stream behind it — decoded through a file stream; results as with the vector context over
    the same octets (i.e. the oracle's), the context's offset and the FILE POSITION left just behind the stream's trailer
    so that the caller reads on from there (what the resync is for), chunked reading through :end, and the asserts."""
    import io
    import tempfile
    p1, p2 = _mixed_plain(n, 21), K.enwik_like(n // 3, seed=22)
    s1, s2 = zlib.compress(p1, 6), pygzip.compress(p2, 6, mtime=0)
    blob = b"HEAD" + s1 + s2 + b"TAIL"
    with tempfile.TemporaryFile() as f:
        f.write(blob)
        f.flush()
        f.seek(0)
        want = oracle_oneshot(blob, "zlib", n + 10, start=4)
        st = A.make_zlib_state(bytearray(n + 10))
        ctx = A.make_octet_stream_context(f, offset=4)
        assert A.valid_octet_stream(f) and ctx.end == len(blob)
        ret = A.decompress(ctx, st, engine=eng)
        assert A.finished(st) and ret == want["ret"] == len(p1) and bytes(st.output_buffer[:ret]) == p1
        assert ctx.offset == 4 + len(s1) and f.tell() == ctx.offset       # just behind the adler32
        # the caller goes on with the next stream from where the file stands
        st2 = A.make_gzip_state(bytearray(len(p2)))
        ctx2 = A.make_octet_stream_context(f, offset=f.tell())
        ret = A.decompress(ctx2, st2, engine=eng)
        assert A.finished(st2) and bytes(st2.output_buffer[:ret]) == p2 and f.read() == b"TAIL"
        # a context stored for later: the stream has moved on meanwhile, the resync puts it back
        ctx3 = A.make_octet_stream_context(f, offset=4, end=4 + 1000)
        f.seek(0, 2)
        A.resync_file_stream(ctx3)
        assert f.tell() == 4
        A.resync_file_stream(A.make_octet_vector_context(blob))           # (the default method: nothing)
        # chunked: :end moved forward call by call, compared with the oracle fed the same chunks
        so, se = O.State(FMT["zlib"], bytearray(n + 10)), A.make_zlib_state(bytearray(n + 10))
        pos = 4
        while not O.finished(so):
            end = min(len(blob), pos + 7001)
            ro = O.decompress(O.make_octet_vector_context(blob, start=pos, end=end), so)
            re_ = A.decompress(A.make_octet_stream_context(f, offset=pos, end=end), se, engine=eng)
            assert re_ == ro and (A.finished(se), A.input_underrun(se)) == (O.finished(so), O.input_underrun(so))
            pos = end
        assert bytes(se.output_buffer[:se.output_offset]) == p1
    for bad in (io.StringIO("text"), None):
        try:
            A.make_octet_stream_context(bad)
            raise AssertionError("accepted something that is no binary input stream")
        except A.ThreeBzError:
            pass
    closed = io.BytesIO(s1)
    c4 = A.make_octet_stream_context(closed)
    closed.close()
    try:
        A.decompress(c4, A.make_zlib_state(bytearray(10)), engine=eng)
        raise AssertionError("closed stream accepted")
    except A.ThreeBzError:
        pass


def case_long_stored_runs(eng, sizes=(14_337, 65_535, 100_001)):
    """Stored blocks long enough to leave the ring kernels' window behind (their head goes straight from the input to
    the output: k2_body), FOLLOWED by compressed data that copies out of them — near the run's end, far back in it, and
    across its seams — in the three containers, with buffers that end inside the run, right behind it and inside the
    data after it.  deflate.lisp:532-573 (:uncompressed-block / copy-block), :343-352 (copy-history)."""
    rng = random.Random(0x570)
    for n in sizes:
        raw = bytes(rng.getrandbits(8) for _ in range(n))
        tail = raw[-3000:] * 3 + b"abcabcabc" * 300 + raw[-30_000:-100] + raw[:2000]
        plain = raw + tail
        for fmt, wbits in (("deflate", -15), ("zlib", 15), ("gzip", 31)):
            c = zlib.compressobj(0, zlib.DEFLATED, wbits)
            head = c.compress(raw) + c.flush(zlib.Z_SYNC_FLUSH if n & 1 else zlib.Z_FULL_FLUSH)
            body = zlib.compressobj(6, zlib.DEFLATED, -15, zdict=raw[-32768:])
            s = head + body.compress(tail) + body.flush()
            if fmt == "zlib":
                s += struct.pack(">I", zlib.adler32(plain))
            elif fmt == "gzip":
                s += struct.pack("<II", zlib.crc32(plain), len(plain) & 0xffffffff)
            for cap in (len(plain) + 8, len(plain), len(plain) - 1, n + 5, n, n - 7, 1000):
                r = assert_same(eng, s, fmt, cap, what="stored run of %d, %s, cap %d" % (n, fmt, cap))
                if cap >= len(plain):
                    assert r["flag"] == "finished" and r["bytes"] == plain, (n, fmt, cap, r["flag"])
        # a stored run that ENDS its stream (nothing is kept in the ring), and one cut short by the input's end
        for fmt, wbits in (("deflate", -15), ("zlib", 15)):
            c = zlib.compressobj(0, zlib.DEFLATED, wbits)
            s = c.compress(raw) + c.flush()
            assert assert_same(eng, s, fmt, n, what="stored only %d %s" % (n, fmt))["bytes"] == raw
            assert_same(eng, s, fmt, n, end=len(s) - 2000, what="stored only, cut %d %s" % (n, fmt))
            assert_same(eng, s, fmt, n - 4000, what="stored only, short buffer %d %s" % (n, fmt))


def case_pointer_contexts(eng, n=60_000):
    """with-octet-pointer / make-octet-pointer-context (io-mmap.lisp:26-54): the same calls over foreign memory —
    a host pointer (tbz_inflate reads it in place) and a device pointer (tbz_inflate_device, no staging) — must give
    what the vector context gives, i.e. what the oracle gives; resumed calls included."""
    mixed = _mixed_plain(n, 8)
    fs, fp, _ = K.zlib_flush_stream(n, block=4096)
    for fmt, blob, plain in (("zlib", zlib.compress(mixed, 6), mixed), ("gzip", pygzip.compress(mixed, 6, mtime=0), mixed),
                             ("deflate", zlib.compress(mixed, 9)[2:-4], mixed), ("zlib", fs, fp)):
        mk = {"deflate": A.make_deflate_state, "zlib": A.make_zlib_state, "gzip": A.make_gzip_state}[fmt]
        d_in = eng.malloc(len(blob) + 64)
        eng.h2d(d_in, blob)
        try:
            for cap in (n + 100, n // 3, 0):
                want = oracle_oneshot(blob, fmt, cap)
                for device in (False, True):
                    with A.with_octet_pointer(d_in if device else blob, len(blob), device=device) as op:
                        assert A.valid_octet_pointer(op)
                        st = mk(bytearray(cap))
                        ctx = A.make_octet_pointer_context(op)
                        ret = A.decompress(ctx, st, engine=eng)
                        flag = ("finished" if st.finished else "underrun" if st.input_underrun else "overflow")
                        assert (flag, ret) == (want["flag"], want["ret"]), (fmt, cap, device, flag, ret)
                        assert bytes(st.output_buffer[:st.output_offset]) == want["bytes"]
                        # a second buffer after overflow: the state resumes from its own copy of the input
                        if st.output_overflow:
                            A.replace_output_buffer(st, bytearray(n + 100))
                            A.decompress(ctx, st, engine=eng)
                            assert st.finished and want["bytes"] + bytes(st.output_buffer[:st.output_offset]) == plain
                    assert not A.valid_octet_pointer(op)
                    try:   # (assert (valid-octet-pointer …)) io-mmap.lisp:66
                        A.decompress(A.make_octet_pointer_context(op), mk(bytearray(8)), engine=eng)
                        raise AssertionError("dead pointer accepted")
                    except A.ThreeBzError:
                        pass
            # input in two pieces through pointer contexts (:end, then :offset)
            for device in (False, True):
                with A.with_octet_pointer(d_in if device else blob, len(blob), device=device) as op:
                    st = mk(bytearray(n + 100))
                    cut = len(blob) // 2
                    A.decompress(A.make_octet_pointer_context(op, end=cut), st, engine=eng)
                    assert st.input_underrun
                    w = oracle_oneshot(blob, fmt, n + 100, end=cut)
                    assert bytes(st.output_buffer[:st.output_offset]) == w["bytes"]
                    A.decompress(A.make_octet_pointer_context(op, offset=cut), st, engine=eng)
                    assert st.finished and bytes(st.output_buffer[:st.output_offset]) == plain
        finally:
            eng.free(d_in)


def small_fused(eng_factory):
    """tbz_small_fused: one launch for one small stream without flush points (the call floor).  The kernel only decides
    the clean case; everything else falls back to the general path.  Here: the one-shot entry (tbz_inflate) with the
    fused path on against the same engine with it off (TBZ_SMALL_FUSED=0) — every field of the result and the octets —
    and against the oracle, over the reference's vectors, the three containers, stored / fixed / dynamic blocks, empty
    output, truncations, damage, buffers that are too small, flush points (not its case), and config 1."""
    e, e0 = eng_factory({}), eng_factory({"TBZ_SMALL_FUSED": 0})
    fields = ("status", "out_len", "out_total", "in_consumed", "adler32", "crc32", "trailer_check", "trailer_isize", "boundary_out")
    took = 0
    try:
        def check(blob, fmt, cap, what):
            nonlocal took
            out, out0 = bytearray(cap), bytearray(cap)
            r, r0 = e.inflate(blob, FMT[fmt], out), e0.inflate(blob, FMT[fmt], out0)
            t = e.timings()
            took += t.k1_gang == 64 and t.huff_launches == 1 and t.scan_ms == 0.0
            for f in fields:
                assert getattr(r, f) == getattr(r0, f), (what, f, getattr(r, f), getattr(r0, f))
            assert (r.flags & 7) == (r0.flags & 7), (what, r.flags, r0.flags)
            assert bytes(out[:r.out_len]) == bytes(out0[:r0.out_len]), what
            w = oracle_oneshot(blob, fmt, cap)
            if w["flag"] == "finished":
                assert r.status == 0 and bytes(out[:r.out_len]) == w["bytes"], what
        vs = json.load(open(os.path.join(GOLDEN, "deflate_vectors.json")))["vectors"]
        for v in vs:
            check(bytes.fromhex(v["input_hex"]), "deflate", 1024, "vector@%d" % v["line"])
        s1, p1 = K.config1_stream()
        check(s1, "deflate", len(p1), "config 1")
        check(s1, "deflate", len(p1) - 1, "config 1, short buffer")
        s2, p2 = K.config1_stream(True)
        check(s2, "deflate", len(p2), "config 1, two stored blocks")
        for n in (1, 700, 9000, 40_000):
            plain = _mixed_plain(n, n)
            for level in (0, 1, 6, 9):
                z = zlib.compress(plain, level)
                g = pygzip.compress(plain, level, mtime=0)
                for fmt, blob in (("zlib", z), ("gzip", g), ("deflate", z[2:-4])):
                    check(blob, fmt, len(plain), "%s L%d n%d" % (fmt, level, n))
                    check(blob, fmt, len(plain) + 77, "%s L%d n%d roomy" % (fmt, level, n))
                    check(blob, fmt, len(plain) // 2, "%s L%d n%d short buffer" % (fmt, level, n))
                    check(blob[:-1], fmt, len(plain), "%s L%d n%d cut by one" % (fmt, level, n))
                    check(blob[:len(blob) // 2], fmt, len(plain), "%s L%d n%d cut in half" % (fmt, level, n))
                    bad = bytearray(blob)
                    bad[len(bad) // 2] ^= 0x20
                    check(bytes(bad), fmt, len(plain), "%s L%d n%d damaged" % (fmt, level, n))
                    check(blob[:-1] + bytes([blob[-1] ^ 1]), fmt, len(plain), "%s L%d n%d bad trailer" % (fmt, level, n))
        for fmt in ("zlib", "gzip", "deflate"):
            empty = {"zlib": zlib.compress(b"", 6), "gzip": pygzip.compress(b"", 6, mtime=0), "deflate": zlib.compress(b"", 6)[2:-4]}[fmt]
            check(empty, fmt, 0, "empty " + fmt)
            check(empty, fmt, 10, "empty, roomy " + fmt)
            e0s = {"zlib": zlib.compress(b"", 0), "gzip": pygzip.compress(b"", 0, mtime=0), "deflate": zlib.compress(b"", 0)[2:-4]}[fmt]
            check(e0s, fmt, 0, "empty, one final stored block " + fmt)
            check(e0s + b"trailing", fmt, 4, "empty stored block, octets after the stream " + fmt)
        c = zlib.compressobj(6, zlib.DEFLATED, 15, 8, zlib.Z_FIXED)
        check(c.compress(p1[:3000]) + c.flush(), "zlib", 3000, "fixed-Huffman block")
        check(zlib.compress(bytes(200_000), 6), "zlib", 200_000, "zeros (dense tokens)")
        check(zlib.compress(bytes(600_000), 9), "zlib", 600_000, "output beyond the fused path's bound")
        fs, fp, _ = K.zlib_flush_stream(48 << 10)
        check(fs, "zlib", len(fp), "flush points: the general path's stream")
        s3, p3 = _fixed_chain(3, 60, 20)
        check(s3, "deflate", len(p3) + 10, "chain of fixed blocks")
        pm = b"ab" * 100 + b"\x00\x00\xff\xff" + b"cd" * 100   # (a stored chain is copied before anything looks for markers)
        for fmt, blob in (("zlib", zlib.compress(pm, 0)), ("deflate", zlib.compress(pm, 0)[2:-4])):
            check(blob, fmt, len(pm), "stored data that holds a marker's octets, " + fmt)
        c = zlib.compressobj(0)
        sf = c.compress(pm) + c.flush(zlib.Z_SYNC_FLUSH) + c.compress(pm) + c.flush()
        check(sf, "zlib", 2 * len(pm), "stored blocks around a flush point")
        assert took > 100, took   # (the clean cases did go through the one launch)
    finally:
        e.close()
        e0.close()


def host_pipeline(eng_factory, n=800 << 10):
    """tbz_inflate on a LARGE host stream decodes it part by part — input of part k+1, decode of part k, output of part
    k-1 at the same time — where it is a clean chain of flush-delimited parts (tbz_inflate_sharded_plan / _verdict), and
    by the ordinary path everywhere else: results are the ordinary path's, to the octet and the flag.  `eng_factory(env)`
    makes an engine with the thresholds in `env` (the CPU suite forces the path at 1 MiB)."""
    env = {"TBZ_PIPE_MIN_KIB": 256, "TBZ_PIPE_PART_KIB": 96, "TBZ_STAGE_CHUNK_KIB": 128, "TBZ_COPY_THREADS": 3} if n < (32 << 20) else {}
    e, plain_eng = eng_factory(env), eng_factory({"TBZ_PIPE_MIN_KIB": 0})
    try:
        s, p, ad = K.zlib_flush_stream(n)
        d = s[2:-4]                                                                # the same blocks in the other two containers
        g = b"\x1f\x8b\x08\x00\x00\x00\x00\x00\x00\x03" + d + struct.pack("<II", zlib.crc32(p), len(p) & 0xffffffff)
        for fmt, blob in (("zlib", s), ("gzip", g), ("deflate", d)):
            out, out0 = bytearray(len(p)), bytearray(len(p))
            r, r0 = e.inflate(blob, FMT[fmt], out), plain_eng.inflate(blob, FMT[fmt], out0)
            assert e.timings().passes >= 2, "the pipelined path did not take this stream"
            assert bytes(out) == p == bytes(out0)
            for f in ("status", "out_len", "out_total", "in_consumed", "adler32", "crc32", "trailer_check", "trailer_isize",
                      "segments", "boundary_out"):
                assert getattr(r, f) == getattr(r0, f), (fmt, f, getattr(r, f), getattr(r0, f))
            assert (r.flags & 3) == (r0.flags & 3), (r.flags, r0.flags)
        # everything that is not a clean chain of parts: the ordinary path's answer
        s2, p2, _ = K.zlib_flush_stream(n // 2, flush=zlib.Z_SYNC_FLUSH)          # history across the cuts
        pz = K.enwik_like(n // 2, 3)
        bad = bytearray(s)
        bad[len(bad) * 2 // 3] ^= 4                                               # damage in a late part
        for what, blob, cap in (("sync flush", s2, len(p2)), ("no flush", zlib.compress(pz, 6), len(pz)), ("damaged", bytes(bad), len(p)),
                                ("short buffer", s, len(p) // 2), ("truncated", s[:len(s) * 3 // 4], len(p)), ("bad adler", s[:-1] + bytes([s[-1] ^ 1]), len(p))):
            out, out0 = bytearray(cap), bytearray(cap)
            r, r0 = e.inflate(blob, FMT["zlib"], out), plain_eng.inflate(blob, FMT["zlib"], out0)
            assert (r.status, r.out_len, r.in_consumed, r.adler32) == (r0.status, r0.out_len, r0.in_consumed, r0.adler32), (what, r.status, r0.status)
            assert bytes(out[:r.out_len]) == bytes(out0[:r0.out_len]), what
        # a large BATCH goes sub-batch by sub-batch through the same three movers (no seams to prove: streams are
        # independent): results and octets as the one-batch call's, failures and short buffers among them
        rng = random.Random(0x3B8)
        # (full size: 36 streams of up to 20 MiB, six distinct ones — a quarter of a GiB of input without minutes of zlib)
        m, kinds = (48, 48) if n < (32 << 20) else (36, 6)
        base = [K.enwik_like(rng.randrange(1, n // 12) if kinds == m else n // 8 - i * (n // 64), seed=500 + i) for i in range(kinds)]
        base_z = [zlib.compress(q, rng.choice([1, 6])) for q in base]
        plains = [base[i % kinds] for i in range(m)] + [b""]
        streams = [base_z[i % kinds] for i in range(m)] + [zlib.compress(b"", 6)]
        streams[5] = streams[5][:len(streams[5]) // 2]
        streams[9] = streams[9][:30] + b"\xff" + streams[9][31:]
        caps = [len(q) for q in plains]
        caps[3] //= 2
        o1, o0 = [bytearray(c) for c in caps], [bytearray(c) for c in caps]
        r1, r0s = e.inflate_batch(streams, FMT["zlib"], o1), plain_eng.inflate_batch(streams, FMT["zlib"], o0)
        assert e.timings().passes >= 2, "the batch was not pipelined"
        for i, (a, b) in enumerate(zip(r1, r0s)):
            assert (a.status, a.out_len, a.out_total, a.adler32, a.in_consumed, a.flags & 7) == (b.status, b.out_len, b.out_total, b.adler32, b.in_consumed, b.flags & 7), i
            assert bytes(o1[i][:a.out_len]) == bytes(o0[i][:b.out_len]), i
            if a.status == 0:
                assert bytes(o1[i]) == plains[i], i
    finally:
        e.close()
        plain_eng.close()


def multi_context_batch(engines):
    """tbz_inflate_batch_multi: n streams over several contexts, one host thread each, results in stream order; the
    assignment is multi.assign_streams's (longest compressed first to the least loaded).  Statuses, counts, checksums
    and octets are what one context reports for the same streams."""
    M = importlib.import_module("3bz_amd.multi")
    rng = random.Random(0x3B6)
    plains = [K.enwik_like(rng.randrange(1, 200_000), seed=300 + i) for i in range(11)] + [b"", bytes(70_000)]
    streams = [zlib.compress(p, rng.choice([1, 6, 9])) for p in plains]
    streams[4] = streams[4][:len(streams[4]) // 2]            # input-underrun
    streams[7] = streams[7][:40] + b"\xff" + streams[7][41:]  # (most likely) an error
    caps = [len(p) for p in plains]
    caps[2] = caps[2] // 3                                     # output-overflow
    eng = engines[0]
    sizes = [len(s) for s in streams]
    for parts in (1, 2, 3, 8):
        assert eng.assign_streams(sizes, parts) == M.assign_streams(sizes, parts), parts
    outs1 = [bytearray(c) for c in caps]
    want = eng.inflate_batch(streams, FMT["zlib"], outs1)
    outs2 = [bytearray(c) for c in caps]
    got = T.Engine.inflate_batch_multi(engines, streams, FMT["zlib"], outs2)
    for i, (w, g) in enumerate(zip(want, got)):
        assert (w.status, w.out_len, w.out_total, w.adler32, w.in_consumed) == (g.status, g.out_len, g.out_total, g.adler32, g.in_consumed), i
        assert bytes(outs1[i][:w.out_len]) == bytes(outs2[i][:g.out_len]), i
        o = oracle_oneshot(streams[i], "zlib", caps[i])
        if o["flag"] != "error":
            assert bytes(outs2[i][:g.out_len]) == o["bytes"], i
    # the same streams ALREADY RESIDENT on their contexts' devices (tbz_inflate_batch_multi_device): nothing is staged
    owner = eng.assign_streams(sizes, len(engines))
    parts, bufs = [], []
    for k, e in enumerate(engines):
        mine = [i for i in range(len(streams)) if owner[i] == k]
        io_, oo_, ip, op = [], [], 0, 0
        for i in mine:
            io_.append(ip)
            ip += (len(streams[i]) + 15) & ~15
            oo_.append(op)
            op += (caps[i] + 15) & ~15
        d_in, d_out = e.malloc(ip + 64), e.malloc(op + 64)
        for i, o in zip(mine, io_):
            if streams[i]:
                e.h2d(d_in + o, streams[i])
        parts.append((d_in, io_, [len(streams[i]) for i in mine], d_out, oo_, [caps[i] for i in mine]))
        bufs.append((mine, d_in, d_out, oo_))
    try:
        got_d = T.Engine.inflate_batch_multi_device(engines, parts, FMT["zlib"])
        for (mine, d_in, d_out, oo_), rs, e in zip(bufs, got_d, engines):
            for i, o, g in zip(mine, oo_, rs):
                w = want[i]
                assert (w.status, w.out_len, w.out_total, w.adler32, w.in_consumed) == (g.status, g.out_len, g.out_total, g.adler32, g.in_consumed), i
                back = bytearray(int(g.out_len))
                if g.out_len:
                    e.d2h(back, d_out + o)
                assert bytes(back) == bytes(outs1[i][:w.out_len]), i
    finally:
        for (mine, d_in, d_out, oo_), e in zip(bufs, engines):
            e.free(d_in)
            e.free(d_out)
    # one context is one in-flight call: the same context twice is refused, by both entries
    if len(engines) >= 2:
        for call in (lambda: T.Engine.inflate_batch_multi([eng, eng], streams[:2], FMT["zlib"], [bytearray(caps[0]), bytearray(caps[1])]),
                     lambda: T.Engine.inflate_batch_multi_device([eng, eng], [parts[0], parts[0]], FMT["zlib"])):
            try:
                call()
                raise AssertionError("duplicate contexts accepted")
            except T.EngineError:
                pass
    # ONE flush-delimited stream over the contexts in one call (tbz_inflate_sharded_multi): sharded where the seams are
    # clean, the first context alone where they are not — results as one context's either way
    if len(engines) >= 2:
        fs, fp, fa = K.zlib_flush_stream(400 << 10)
        ss, sp_, _ = K.zlib_flush_stream(200 << 10, flush=zlib.Z_SYNC_FLUSH)
        for blob, plain, cap, want_sharded in ((fs, fp, len(fp), True), (ss, sp_, len(sp_), False), (fs, fp, len(fp) // 2, False),
                                               (fs[:-1] + bytes([fs[-1] ^ 1]), fp, len(fp), False)):
            o1, o2 = bytearray(cap), bytearray(cap)
            w = eng.inflate(blob, FMT["zlib"], o1)
            g, sh = T.Engine.inflate_sharded_multi(engines, blob, FMT["zlib"], o2)
            assert sh == want_sharded, (sh, want_sharded)
            assert (w.status, w.out_len, w.out_total, w.adler32, w.in_consumed, w.segments) == (g.status, g.out_len, g.out_total, g.adler32, g.in_consumed, g.segments)
            assert bytes(o1[:w.out_len]) == bytes(o2[:g.out_len])
    # ... and a single decode straight into device memory (tbz_inflate_to_device)
    res, d = eng.inflate_to_device(streams[0], FMT["zlib"])
    try:
        back = bytearray(len(plains[0]))
        eng.d2h(back, d)
        assert res.status == 0 and res.out_len == len(plains[0]) and bytes(back) == plains[0]
    finally:
        eng.free(d)


ALL_CASES = [case_known_answer_vectors, case_test_deflated, case_reference_chunk_patterns, case_containers_and_levels, case_flush_streams,
             case_noflush_streams, case_block_starts_found, case_close_block_starts, case_fixed_block_chains, case_history_across_groups,
             case_configs_1_3_5, case_overflow_and_underrun, case_errors, case_false_markers, case_device_buffers,
             case_checksum_kernels, case_deep_codes, case_chunked_resume, case_gzip_members,
             case_long_stored_runs, case_pointer_contexts, case_stream_contexts, case_container_headers, case_gzip_metadata, case_scratch_bounds, case_token_density, case_fuzz]
# what each engine flavour of the test modules runs.  "auto" runs everything; the others run the cases that can
# tell them apart (the CPU suite has to stay within minutes: a case costs seconds on the lane emulator)
FLAVOUR_CASES = {
    # K0b on every stream, however small: candidates, chains through false ones, symbolic history everywhere
    "findalways": ["case_known_answer_vectors", "case_containers_and_levels", "case_noflush_streams",
                   "case_close_block_starts", "case_fixed_block_chains", "case_overflow_and_underrun", "case_false_markers", "case_errors", "case_fuzz"],
    # chain walk + layout on the host even where the device could (K3)
    "hostlayout": ["case_known_answer_vectors", "case_flush_streams", "case_configs_1_3_5",
                   "case_false_markers", "case_device_buffers", "case_errors", "case_fuzz"],
    # one wave per group in K2
    "k2single": ["case_flush_streams", "case_history_across_groups", "case_configs_1_3_5", "case_deep_codes",
                 "case_overflow_and_underrun"],
    # the ring kernel on two waves instead of three (large groups, H-groups)
    "k2ring2": ["case_noflush_streams", "case_history_across_groups", "case_configs_1_3_5", "case_containers_and_levels",
                "case_long_stored_runs"],
}
# the cases whose behaviour depends on the K1 flavour (forced-flavour runs skip the rest: checksums, device
# buffers and the replay protocol go through the same engine calls whatever decodes the Huffman codes)
K1_CASES = [case_known_answer_vectors, case_test_deflated, case_containers_and_levels, case_fixed_block_chains,
            case_overflow_and_underrun, case_errors, case_deep_codes, case_fuzz]
