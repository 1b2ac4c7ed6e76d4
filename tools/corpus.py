"""Deterministic synthetic corpora for the five BASELINE.json configs.

3bz has no compressor, so system zlib is used as the COMPRESSOR only (never as
the thing under test); the adversarial config is hand-assembled with BitWriter
because zlib never emits distance 32768.

Everything is seeded and chunk-size independent: the same (seed, size) always
yields the same bytes, here and on the GPU box.
"""
import struct
import zlib
from concurrent.futures import ProcessPoolExecutor

import numpy as np

# --------------------------------------------------------------------------- text


_VOCAB_CACHE = {}


def _vocab(seed):
    """50k pseudo-word vocabulary: English letter frequencies, Zipf(1.05) ranks,
    a handful of punctuation / wiki-markup 'words'."""
    if seed in _VOCAB_CACHE:
        return _VOCAB_CACHE[seed]
    rng = np.random.default_rng([seed, 0x766F63])
    V = 50000
    letters = np.frombuffer(b"etaoinshrdlcumwfgypbvkjxqz", dtype=np.uint8)
    freq = np.array([12.7, 9.06, 8.17, 7.51, 6.97, 6.75, 6.33, 6.09, 5.99, 4.25, 4.03, 2.78, 2.76, 2.41,
                     2.36, 2.23, 2.02, 1.97, 1.93, 1.49, 0.98, 0.77, 0.15, 0.15, 0.095, 0.074])
    freq = freq / freq.sum()
    # frequent words are short: length grows slowly with rank
    ranks = np.arange(1, V + 1)
    wl = np.clip((1.5 + 1.1 * np.log2(ranks + 1) * 0.55 + rng.normal(0, 1.2, V)).astype(np.int64), 1, 14)
    total = int(wl.sum())
    chars = letters[rng.choice(26, size=total, p=freq)]
    starts = np.concatenate([[0], np.cumsum(wl)[:-1]])
    words = [bytes(chars[s:s + l]) + b" " for s, l in zip(starts, wl)]
    specials = [b". ", b", ", b".\n\n", b"[[", b"]] ", b"<ref>", b"</ref> ", b"''", b"== ", b" ==\n", b"* ",
                b"{{", b"}} ", b"&quot;", b"1", b"19", b"20", b"0 ", b"| ", b"; "]
    # splice specials in at fairly frequent ranks
    for k, sp in enumerate(specials):
        words[3 + 7 * k] = sp
    lens = np.array([len(w) for w in words], dtype=np.int64)
    pool = np.frombuffer(b"".join(words), dtype=np.uint8)
    pstart = np.concatenate([[0], np.cumsum(lens)[:-1]])
    p = 1.0 / ranks ** 1.05
    cdf = np.cumsum(p / p.sum())
    _VOCAB_CACHE[seed] = (pool, pstart, lens, cdf)
    return _VOCAB_CACHE[seed]


_UNIT = 1 << 20


def _text_unit(seed, unit):
    """exactly 1 MiB of text for (seed, unit index)"""
    pool, pstart, lens, cdf = _vocab(seed)
    rng = np.random.default_rng([seed, 0x747874, unit])
    m = _UNIT // 4
    ids = np.searchsorted(cdf, rng.random(m), side="right").clip(0, len(lens) - 1)
    wl = lens[ids]
    ends = np.cumsum(wl)
    k = int(np.searchsorted(ends, _UNIT, side="left")) + 1
    assert k <= m
    ids, wl, ends = ids[:k], wl[:k], ends[:k]
    total = int(ends[-1])
    out_start = ends - wl
    idx = np.arange(total, dtype=np.int64) - np.repeat(out_start, wl) + np.repeat(pstart[ids], wl)
    return pool[idx][:_UNIT]


def enwik_like(nbytes, seed=0x3B2, offset=0):
    """nbytes of deterministic enwik-style text starting at byte `offset` of the
    infinite (seed) text."""
    first, last = offset // _UNIT, (offset + nbytes + _UNIT - 1) // _UNIT
    parts = [_text_unit(seed, u) for u in range(first, last)]
    buf = np.concatenate(parts) if parts else np.zeros(0, np.uint8)
    lo = offset - first * _UNIT
    return buf[lo:lo + nbytes].tobytes()


def xorshift64star_bytes(n, seed):
    """uniform bytes from xorshift64* (config 1 / config 5 payloads)"""
    out = bytearray()
    x = seed & 0xFFFFFFFFFFFFFFFF or 1
    while len(out) < n:
        x ^= x >> 12
        x ^= (x << 25) & 0xFFFFFFFFFFFFFFFF
        x ^= x >> 27
        out += struct.pack("<Q", (x * 0x2545F4914F6CDD1D) & 0xFFFFFFFFFFFFFFFF)
    return bytes(out[:n])


# --------------------------------------------------------------------------- zlib-built streams


def _deflate_pieces(args):
    """raw deflate of plain text [offset, offset+n) with a flush every `block` bytes.
    Returns (raw_deflate_bytes_ending_on_a_flush_marker, adler32_of_slice, n)."""
    seed, offset, n, block, level, flush = args
    plain = enwik_like(n, seed, offset)
    c = zlib.compressobj(level, zlib.DEFLATED, -15)
    out = []
    for i in range(0, n, block):
        out.append(c.compress(plain[i:i + block]))
        out.append(c.flush(flush))
    return b"".join(out), plain


def _adler_combine(a1, a2, len2):
    """adler32 of A||B from adler32(A), adler32(B), len(B) (zlib's adler32_combine)"""
    BASE = 65521
    rem = len2 % BASE
    sum1 = a1 & 0xffff
    sum2 = (rem * sum1) % BASE
    sum1 += (a2 & 0xffff) + BASE - 1
    sum2 += ((a1 >> 16) & 0xffff) + ((a2 >> 16) & 0xffff) + BASE - rem
    if sum1 >= BASE:
        sum1 -= BASE
    if sum1 >= BASE:
        sum1 -= BASE
    if sum2 >= (BASE << 1):
        sum2 -= (BASE << 1)
    if sum2 >= BASE:
        sum2 -= BASE
    return sum1 | (sum2 << 16)


def zlib_flush_stream(nbytes, seed=0x3B2, block=16384, level=6, flush=zlib.Z_FULL_FLUSH, workers=1,
                      slice_bytes=16 << 20, want_plain=True):
    """Config 2 (and 4, 2b): :zlib stream of `nbytes` of text, a flush every `block`
    input bytes.  With Z_FULL_FLUSH every segment is history-independent, so slices
    can be compressed in parallel and concatenated (each ends on 00 00 FF FF).
    With Z_SYNC_FLUSH (variant 2b) history crosses flush points, so workers must be 1.
    Returns (stream_bytes, plain_bytes_or_None, adler32)."""
    assert nbytes % 1 == 0
    if flush != zlib.Z_FULL_FLUSH:
        workers, slice_bytes = 1, nbytes
    jobs = [(seed, o, min(slice_bytes, nbytes - o), block, level, flush) for o in range(0, nbytes, slice_bytes)]
    if workers > 1 and len(jobs) > 1:
        with ProcessPoolExecutor(max_workers=workers) as ex:
            res = list(ex.map(_deflate_pieces, jobs))
    else:
        res = [_deflate_pieces(j) for j in jobs]
    adler = 1
    for _, plain in res:
        adler = zlib.adler32(plain, adler)
    stream = b"".join([b"\x78\x9c"] + [r[0] for r in res] + [b"\x03\x00", struct.pack(">I", adler)])
    plain = b"".join(r[1] for r in res) if want_plain else None
    return stream, plain, adler


def gzip_member(plain, level=6):
    """Config 3 member: 10-byte header (no optional fields, mtime 0) + deflate + CRC32 + ISIZE"""
    c = zlib.compressobj(level, zlib.DEFLATED, -15)
    body = c.compress(plain) + c.flush()
    return (b"\x1f\x8b\x08\x00\x00\x00\x00\x00\x00\x03" + body +
            struct.pack("<II", zlib.crc32(plain), len(plain) & 0xFFFFFFFF))


def _gzip_member_job(args):
    seed, offset, n, level = args
    plain = enwik_like(n, seed, offset)
    return gzip_member(plain, level), plain


def gzip_members(n_members, member_bytes=256 << 10, seed=0x3B2, level=6, workers=1):
    """Config 3: concatenated multi-member .gz.  Returns (blob, offsets, plains)."""
    jobs = [(seed, i * member_bytes, member_bytes, level) for i in range(n_members)]
    if workers > 1 and n_members > 1:
        with ProcessPoolExecutor(max_workers=workers) as ex:
            res = list(ex.map(_gzip_member_job, jobs, chunksize=max(1, n_members // (workers * 4))))
    else:
        res = [_gzip_member_job(j) for j in jobs]
    offsets, pos = [], 0
    for m, _ in res:
        offsets.append(pos)
        pos += len(m)
    return b"".join(m for m, _ in res), offsets, [p for _, p in res]


def stored_stream(payload, max_block=65535):
    """Config 1: raw :deflate of stored (type-0) blocks; LEN is 16-bit so 65536 bytes need two."""
    out = bytearray()
    n = len(payload)
    pos = 0
    if n == 0:
        return b"\x01\x00\x00\xff\xff"
    while pos < n:
        k = min(max_block, n - pos)
        final = 1 if pos + k == n else 0
        out += bytes([final]) + struct.pack("<HH", k, k ^ 0xFFFF) + payload[pos:pos + k]
        pos += k
    return bytes(out)


# --------------------------------------------------------------------------- hand bit-writer


class BitWriter:
    """LSB-first bit packer (RFC 1951 §3.1.1): Huffman codes go in MSB-first, extra bits LSB-first."""

    def __init__(self):
        self.buf = bytearray()
        self.acc = 0
        self.n = 0

    def bits(self, value, nbits):
        self.acc |= (value & ((1 << nbits) - 1)) << self.n
        self.n += nbits
        while self.n >= 8:
            self.buf.append(self.acc & 0xFF)
            self.acc >>= 8
            self.n -= 8

    def code(self, code, nbits):
        r = 0
        for i in range(nbits):
            r |= ((code >> i) & 1) << (nbits - 1 - i)
        self.bits(r, nbits)

    def align(self):
        if self.n:
            self.bits(0, 8 - self.n)

    def getvalue(self):
        assert self.n == 0
        return bytes(self.buf)


_LEN_BASE = [3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163,
             195, 227, 258]
_LEN_EXTRA = [0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0]
_DIST_BASE = [1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049,
              3073, 4097, 6145, 8193, 12289, 16385, 24577]
_DIST_EXTRA = [0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13]


class FixedHuffmanWriter(BitWriter):
    """emit BTYPE=1 blocks token by token (RFC 1951 §3.2.6 fixed codes)"""

    def begin_block(self, final=False):
        self.bits(1 if final else 0, 1)
        self.bits(1, 2)

    def sym(self, s):
        if s <= 143:
            self.code(0x30 + s, 8)
        elif s <= 255:
            self.code(0x190 + (s - 144), 9)
        elif s <= 279:
            self.code(s - 256, 7)
        else:
            self.code(0xC0 + (s - 280), 8)

    def literal(self, b):
        self.sym(b)

    def match(self, length, dist):
        li = max(i for i in range(29) if _LEN_BASE[i] <= length) if length < 258 else 28
        self.sym(257 + li)
        if _LEN_EXTRA[li]:
            self.bits(length - _LEN_BASE[li], _LEN_EXTRA[li])
        di = max(i for i in range(30) if _DIST_BASE[i] <= dist)
        self.code(di, 5)
        if _DIST_EXTRA[di]:
            self.bits(dist - _DIST_BASE[di], _DIST_EXTRA[di])

    def end_block(self):
        self.sym(256)


def canonical_codes(lens):
    """RFC 1951 §3.2.2 canonical code values for a list of code lengths (0 = unused)"""
    cnt = [0] * 16
    for l in lens:
        cnt[l] += 1
    cnt[0] = 0
    nxt, code = [0] * 16, 0
    for L in range(1, 16):
        code = (code + cnt[L - 1]) << 1
        nxt[L] = code
    out = []
    for l in lens:
        out.append(nxt[l] if l else 0)
        if l:
            nxt[l] += 1
    return out


class DynamicHuffmanWriter(BitWriter):
    """emit BTYPE=2 blocks from EXPLICIT code lengths (test streams whose codes a compressor would never
    choose: 15-bit codes everywhere, distance codes that overflow a decoder's second-level table ...).
    The code-length code is fixed: symbols 0..12 get 4 bits, 13..18 get 5 bits (complete), no repeat codes."""

    def __init__(self, lit_lens, dist_lens):
        super().__init__()
        assert 257 <= len(lit_lens) <= 286 and 1 <= len(dist_lens) <= 30
        self.lit_lens, self.dist_lens = list(lit_lens), list(dist_lens)
        self.lit_codes, self.dist_codes = canonical_codes(self.lit_lens), canonical_codes(self.dist_lens)
        self.cl_lens = [4] * 13 + [5] * 6
        self.cl_codes = canonical_codes(self.cl_lens)

    def begin_block(self, final=False):
        self.bits(1 if final else 0, 1)
        self.bits(2, 2)
        self.bits(len(self.lit_lens) - 257, 5)
        self.bits(len(self.dist_lens) - 1, 5)
        self.bits(19 - 4, 4)
        for sym in (16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15):
            self.bits(self.cl_lens[sym], 3)
        for l in self.lit_lens + self.dist_lens:
            self.code(self.cl_codes[l], self.cl_lens[l])

    def sym(self, s):
        assert self.lit_lens[s], s
        self.code(self.lit_codes[s], self.lit_lens[s])

    def literal(self, b):
        self.sym(b)

    def match(self, length, dist):
        li = max(i for i in range(29) if _LEN_BASE[i] <= length) if length < 258 else 28
        self.sym(257 + li)
        if _LEN_EXTRA[li]:
            self.bits(length - _LEN_BASE[li], _LEN_EXTRA[li])
        di = max(i for i in range(30) if _DIST_BASE[i] <= dist)
        assert self.dist_lens[di], di
        self.code(self.dist_codes[di], self.dist_lens[di])
        if _DIST_EXTRA[di]:
            self.bits(dist - _DIST_BASE[di], _DIST_EXTRA[di])

    def end_block(self):
        self.sym(256)


def deep_code_stream(n_tokens=40000, seed=7, dist_overflow=False, blocks=3):
    """Raw deflate stream of dynamic blocks whose codes are as deep as RFC 1951 allows.
    lit/len: 16 symbols coded with lengths 1,2,...,14,15,15 (literals, end-of-block and five length symbols),
    used UNIFORMLY, so most tokens carry codes longer than any first-level lookup table.
    distance: the same shape over 16 symbols, or (dist_overflow) lengths 1..7 plus two 8-bit prefixes that
    each run down to 15 bits - 256 second-level entries, more than the engine's pool holds.
    Returns (stream, plain)."""
    rng = np.random.default_rng([seed, 0xDEE9])
    lit_syms = [ord(c) for c in "etaoinshrd"] + [256, 257, 260, 266, 275, 285]
    order = rng.permutation(16)
    chain = list(range(1, 15)) + [15, 15]
    lit_lens = [0] * 286
    for k, i in enumerate(order):
        lit_lens[lit_syms[i]] = chain[k]
    if dist_overflow:
        dl = list(range(1, 8)) + [9, 10, 11, 12, 13, 14, 15, 15] * 2  # 23 symbols, complete
    else:
        dl = chain[:]
    dperm = rng.permutation(len(dl))
    dist_lens = [0] * len(dl)
    for k, i in enumerate(dperm):
        dist_lens[i] = dl[k]
    w = DynamicHuffmanWriter(lit_lens, dist_lens)
    out = bytearray()
    lits = [s for s in lit_syms if s < 256]
    lens_ = {257: (3, 3), 260: (6, 6), 266: (13, 14), 275: (51, 58), 285: (258, 258)}
    for b in range(blocks):
        w.begin_block(final=(b == blocks - 1))
        for t in range(n_tokens // blocks):
            s = lit_syms[int(rng.integers(0, 16))]
            if s == 256:
                continue
            if s < 256 or len(out) < 4:
                if s >= 256:
                    s = lits[t % len(lits)]
                w.literal(s)
                out.append(s)
                continue
            lo, hi = lens_[s]
            length = int(rng.integers(lo, hi + 1))
            for _ in range(64):  # a distance symbol that is coded and reaches no further than the history
                di = int(rng.integers(0, len(dist_lens)))
                if dist_lens[di] and _DIST_BASE[di] <= min(len(out), 32768):
                    break
            else:
                di = next(i for i in range(len(dist_lens)) if dist_lens[i] and _DIST_BASE[i] <= len(out))
            hi_d = min(_DIST_BASE[di] + (1 << _DIST_EXTRA[di]) - 1, len(out), 32768)
            dist = int(rng.integers(_DIST_BASE[di], hi_d + 1))
            w.match(length, dist)
            _lz_apply(out, length, dist)
        w.end_block()
    w.align()
    return w.getvalue(), bytes(out)


def dense_literal_stream(n_lits=60000, blocks=2):
    """Raw deflate stream whose tokens are as dense as its bits: a dynamic block with a two-symbol lit/len code
    ('a' and end-of-block, one bit each) and a single one-bit distance code, filled with literals.  A decoder
    that parks tokens by bit position has no slack here.  Returns (stream, plain)."""
    lit_lens = [0] * 257
    lit_lens[ord("a")] = 1
    lit_lens[256] = 1
    w = DynamicHuffmanWriter(lit_lens, [1])
    out = bytearray()
    for b in range(blocks):
        w.begin_block(final=(b == blocks - 1))
        for _ in range(n_lits // blocks):
            w.literal(ord("a"))
            out.append(ord("a"))
        w.end_block()
    w.align()
    return w.getvalue(), bytes(out)


def _lz_apply(out, length, dist):
    """reference LZ77 semantics for building the expected plaintext (pure Python, byte-exact)"""
    start = len(out) - dist
    if dist >= length:
        out += out[start:start + length]
    else:
        pat = bytes(out[start:])
        reps = length // dist + 1
        out += (pat * reps)[:length]


def adversarial_stream(total=256 << 20, seed=0x3B5A0005, full_flush_every=0, sync_flush_every=0):
    """Config 5: :zlib, fixed-Huffman blocks.
      phase A (first half): per block one literal then (len 258, dist 1) runs;
      phase B: a 32 KiB random page, then (len 3..258, dist 32768) references
               interleaved with overlapping short-period references
               (dist in {2,3,5,7,257}, len 258).
    Blocks hold <= 64 KiB of output.  full_flush_every=0 (default): no flush markers —
    the stream is ONE sequential segment (stated in DESIGN.md).  Otherwise an empty stored
    block + history restart is inserted every `full_flush_every` output bytes (phase B then
    re-emits its page so distances stay legal).  sync_flush_every: the empty stored block alone, history
    kept (what Z_SYNC_FLUSH does): matches keep reaching back across the markers.
    Returns (stream, plain)."""
    rng = np.random.default_rng([seed, 5])
    w = FixedHuffmanWriter()
    w.bits(0x78, 8)
    w.bits(0x9C, 8)
    out = bytearray()
    half = total // 2
    since_flush = 0
    hist = 0  # bytes of legal history since the last flush

    def flush_marker():
        nonlocal since_flush, hist
        w.bits(0, 3)
        w.align()
        w.bits(0x0000, 16)
        w.bits(0xFFFF, 16)
        since_flush = 0
        hist = 0

    def maybe_flush():
        nonlocal since_flush
        if full_flush_every and since_flush >= full_flush_every:
            flush_marker()
        elif sync_flush_every and since_flush >= sync_flush_every:
            w.bits(0, 3)
            w.align()
            w.bits(0x0000, 16)
            w.bits(0xFFFF, 16)
            since_flush = 0

    # ---- phase A
    blk = 0
    while len(out) < half:
        maybe_flush()
        w.begin_block(False)
        lit = (blk * 37 + 11) & 0xFF
        w.literal(lit)
        out.append(lit)
        produced = 1
        # 254 max-length RLE matches => 13*254+18 bits == 0 mod 8: blocks are byte aligned
        for _ in range(254):
            if len(out) + 258 > half:
                break
            w.match(258, 1)
            out += bytes([lit]) * 258
            produced += 258
        w.end_block()
        since_flush += produced
        hist += produced
        blk += 1
    # ---- phase B
    lens_cycle = rng.integers(3, 259, size=4096)
    k = 0
    page_needed = True
    while len(out) < total:
        maybe_flush()
        w.begin_block(False)
        produced = 0
        if page_needed or hist < 32768:
            page = rng.integers(0, 256, size=32768, dtype=np.uint8).tobytes()
            for b in page:
                w.literal(b)
            out += page
            produced += 32768
            hist += 32768
            page_needed = False
        while produced < 65536 - 258 and len(out) < total:
            if k % 3 == 2:
                d = (2, 3, 5, 7, 257)[(k // 3) % 5]
                ln = 258
            else:
                d = 32768
                ln = int(lens_cycle[k % 4096])
            ln = min(ln, total - len(out))
            if ln < 3:
                for _ in range(ln):
                    w.literal(0x5A)
                    out.append(0x5A)
                produced += ln
                break
            w.match(ln, d)
            _lz_apply(out, ln, d)
            produced += ln
            k += 1
        w.end_block()
        since_flush += produced
        hist += produced
    # final empty fixed block + adler
    w.begin_block(True)
    w.end_block()
    w.align()
    plain = bytes(out)
    body = w.getvalue()
    return body + struct.pack(">I", zlib.adler32(plain)), plain


def config1_stream(two_blocks=False):
    """Config 1: `:deflate`, one stored block of 65535 uniform bytes (xorshift64* seed 0x3B5A0001);
    two_blocks=True is the literal '64 KiB' = 65536 reading (65535 + 1)."""
    n = 65536 if two_blocks else 65535
    payload = xorshift64star_bytes(n, 0x3B5A0001)
    return stored_stream(payload), payload


# --------------------------------------------------------------------------- full-size corpora, generated in ONE pool
# (for the -m gpu property tests: every fork happens before the process touches the GPU)


def _full_job(job):
    kind, n, seed = job
    if kind == "2b":
        return zlib_flush_stream(n, seed=seed, flush=zlib.Z_SYNC_FLUSH)
    if kind == "nf":
        p = enwik_like(n, seed)
        return zlib.compress(p, 6), p, zlib.adler32(p)
    if kind == "2":
        return zlib_flush_stream(n, seed=seed)
    if kind in ("5", "5f"):
        s, p = adversarial_stream(n, sync_flush_every=(1 << 20) if kind == "5f" else 0)
        return s, p, zlib.adler32(p)
    if kind == "3":  # one slice of the member list: (first member, count)
        first, count = seed
        return [_gzip_member_job((0x3B2, (first + i) * n, n, 6)) for i in range(count)]
    raise ValueError(kind)


def full_size_corpora(jobs, workers=16):
    """jobs: {name: (kind, n, seed)} -> {name: result}; every job runs in a worker process of one pool"""
    names = list(jobs)
    with ProcessPoolExecutor(max_workers=workers) as ex:
        res = list(ex.map(_full_job, [jobs[k] for k in names]))
    return dict(zip(names, res))
