"""Multi-GPU: independent streams shard across ranks, one process per GPU.

A stream's decode never needs a collective (each deflate-state is self-contained,
deflate.lisp:4-62); the only exchange is an all_gather of the fixed 64-byte result records
(struct tbz_result) so every rank knows every stream's status / length / checksum — 8 x 64 B,
latency-bound, not bandwidth-bound.  `torch.distributed` is plumbing here (backend "nccl" is RCCL on
ROCm; the CPU tests use "gloo").
"""
import ctypes as C

from . import _lib


def assign_streams(compressed_sizes, world):
    """longest-processing-time-first on compressed size; round-robin when equal.
    Returns owner[i] = rank that decodes stream i."""
    order = sorted(range(len(compressed_sizes)), key=lambda i: (-compressed_sizes[i], i))
    load = [0] * world
    owner = [0] * len(compressed_sizes)
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        owner[i] = r
        load[r] += compressed_sizes[i]
    return owner


def results_to_tensor(results, torch, device="cpu"):
    """pack a list of tbz_result into a [n, 64] uint8 tensor"""
    buf = bytearray()
    for r in results:
        buf += bytes(r)
    t = torch.frombuffer(buf if buf else bytearray(64), dtype=torch.uint8).clone()
    return t.reshape(-1, 64)[: len(results)].to(device)


def tensor_to_results(t):
    out = []
    raw = bytes(t.cpu().contiguous().numpy().tobytes())
    for i in range(len(raw) // 64):
        out.append(_lib.Result.from_buffer_copy(raw[i * 64:(i + 1) * 64]))
    return out


def exchange_results(local_results, owner, rank, world, dist, torch, device="cpu"):
    """X1: all_gather the result records.  `local_results` are this rank's streams in stream order;
    returns the records of ALL streams in stream order on every rank."""
    n = len(owner)
    per_rank = [[i for i in range(n) if owner[i] == r] for r in range(world)]
    width = max(1, max(len(p) for p in per_rank))
    mine = torch.zeros((width, 64), dtype=torch.uint8, device=device)
    if local_results:
        mine[: len(local_results)] = results_to_tensor(local_results, torch, device)
    gathered = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)
    out = [None] * n
    for r in range(world):
        recs = tensor_to_results(gathered[r])
        for k, i in enumerate(per_rank[r]):
            out[i] = recs[k]
    return out


# ------------------------------------------------------------------------------------------------
# ONE flush-delimited stream across ranks: contiguous ranges of its segments per
# rank, one exchange of fixed-size records, seams proven by the ranks' own results.
#
# A deflate stream may be entered at any block boundary (deflate.lisp:518-528 reads BFINAL/BTYPE with no
# other state), and the octet after a flush marker 00 00 FF FF is one — if the marker is real.  So rank r
# decodes [cut_r, cut_r+1) where cut_r is the end of the first marker at or after r*C/W: rank 0 in the
# stream's own format (container header), the others as raw deflate.  Nobody trusts a cut: a rank that
# is not the last reports `input-underrun` with in_consumed == its range length exactly when its block
# chain — started at cut_r — ran out of input just where a block starts (tbz_amd.h).  By induction from
# rank 0's true start, all seams clean  =>  every cut is a true block boundary and the concatenation of
# the parts is what a front-to-back decoder produces; a match that reaches before a cut (sync-flush
# history) is `distance before start` on that rank, which is not clean.  Anything not clean — false
# marker at a cut, history across a cut, any error, trailer or checksum mismatch — and rank 0 decodes
# the whole stream by the ordinary path, so statuses and errors are exactly the single-GPU ones.
# Exchange: one all_gather of 8 x int64 per rank (status, lengths, checksum partial); the checksum of
# the whole is the ordered combine of the partials (adler32: zlib.lisp:97-102 / crc32: gzip.lisp:80-81
# chain per call in the reference; here they combine algebraically).
# ------------------------------------------------------------------------------------------------
MARK = b"\x00\x00\xff\xff"
FMT_DEFLATE, FMT_ZLIB, FMT_GZIP = 0, 1, 2
_BASE = 65521


def adler32_combine(a1, a2, len2):
    """adler32 of A||B from adler32(A), adler32(B), len(B)   (s1 | s2 << 16, checksums.lisp:18-62)"""
    s1a, s2a, s1b, s2b = a1 & 0xFFFF, a1 >> 16, a2 & 0xFFFF, a2 >> 16
    s1 = (s1a + s1b - 1) % _BASE
    s2 = (s2a + s2b + (len2 % _BASE) * (s1a - 1)) % _BASE
    return s1 | (s2 << 16)


def _gf2_times(mat, vec):
    s, i = 0, 0
    while vec:
        if vec & 1:
            s ^= mat[i]
        vec >>= 1
        i += 1
    return s


def crc32_combine(c1, c2, len2):
    """crc32 of A||B from the finalised crc32(A), crc32(B), len(B): multiply c1 by x^(8 len2) mod the
    reflected polynomial #xedb88320 (checksums.lisp:177-193) by repeated squaring of the shift operator"""
    if len2 <= 0:
        return c1
    odd = [0xEDB88320] + [1 << i for i in range(31)]           # shift by one zero bit
    even = [_gf2_times(odd, odd[i]) for i in range(32)]        # two
    odd = [_gf2_times(even, even[i]) for i in range(32)]       # four
    while True:
        even = [_gf2_times(odd, odd[i]) for i in range(32)]    # first pass: one zero OCTET
        if len2 & 1:
            c1 = _gf2_times(even, c1)
        len2 >>= 1
        if not len2:
            break
        odd = [_gf2_times(even, even[i]) for i in range(32)]
        if len2 & 1:
            c1 = _gf2_times(odd, c1)
        len2 >>= 1
        if not len2:
            break
    return c1 ^ c2


def shard_plan(data, world, start=0, end=None, lib=None):
    """world+1 cut points in `data[start:end]`: cut_r = end of the first flush marker at or after the r-th
    equal share of the compressed octets (a range may be empty when markers are scarce).  tbz_inflate_sharded_plan
    (include/tbz_amd.h): the same arithmetic a Lisp host calls through the shim."""
    end = len(data) if end is None else end
    L = lib or _lib.load()
    view = bytes(memoryview(data)[start:end])
    cuts = (C.c_uint64 * (world + 1))()
    rc = L.tbz_inflate_sharded_plan(C.cast(C.c_char_p(view), C.c_void_p), len(view), world, cuts)
    if rc:
        raise ValueError("tbz_inflate_sharded_plan: %d" % rc)
    return [start + int(c) for c in cuts]


def shard_decode(eng, data, fmt, cuts, r):
    """rank r's part: decode data[cuts[r]:cuts[r+1]] into a device buffer of exactly its size.
    Returns (record, d_out, n): record = 8 ints for the exchange; the caller owns d_out (eng.free)."""
    lo, hi = cuts[r], cuts[r + 1]
    f = fmt if r == 0 else FMT_DEFLATE
    if hi == lo and r > 0:
        return [1, 0, 0, 0, 0, 0, 0, 0], None, 0   # nothing to do: trivially clean
    # ONE decode: the output buffer is allocated on the device once K1 has sized it (tbz_inflate_to_device)
    res, d_out = eng.inflate_to_device(data, f, lo, hi)
    got = int(res.out_len) if res.status >= 0 else 0
    ck = 0
    if res.status in (0, 1) and fmt != FMT_DEFLATE:
        if fmt == FMT_ZLIB:
            s1, s2 = eng.adler32_device(d_out, got, 1, 0)
            ck = s1 | (s2 << 16)
        else:
            ck = eng.crc32_device(d_out, got, 0)
    rec = [int(res.status), got, int(res.in_consumed), hi - lo, ck, int(res.flags), 0, 0]
    return rec, d_out, got


_WHY = {-1: "last part did not finish", -2: "trailer incomplete", -3: "checksum mismatch"}


def shard_verdict(recs, data, fmt, cuts, lib=None):
    """the same decision on every rank from the gathered records: {"ok": bool, ...}.  ok => "offsets"
    (where each rank's part starts in the output), "total", "check", "in_consumed".  The rules live in C
    (tbz_inflate_sharded_verdict, include/tbz_amd.h) so that every host language applies the same ones."""
    world = len(recs)
    L = lib or _lib.load()
    base = cuts[0]
    view = bytes(memoryview(data)[base:cuts[-1]])
    rs = (_lib.Result * world)()
    cks = (C.c_uint32 * world)()
    for r in range(world):
        rs[r].status, rs[r].out_len, rs[r].in_consumed = recs[r][0], recs[r][1], recs[r][2]
        if fmt == FMT_ZLIB:
            rs[r].adler32 = recs[r][4]
        elif fmt == FMT_GZIP:
            rs[r].crc32 = recs[r][4]
        cks[r] = recs[r][4] & 0xFFFFFFFF
    cc = (C.c_uint64 * (world + 1))(*[c - base for c in cuts])
    offs = (C.c_uint64 * world)()
    total, consumed, check, why = C.c_uint64(), C.c_uint64(), C.c_uint32(), C.c_int()
    rc = L.tbz_inflate_sharded_verdict(fmt, C.cast(C.c_char_p(view), C.c_void_p), len(view), world, cc, rs, cks, offs,
                                       C.byref(total), C.byref(check), C.byref(consumed), C.byref(why))
    if rc < 0:
        raise ValueError("tbz_inflate_sharded_verdict: %d" % rc)
    if rc:
        w = why.value
        return {"ok": False, "why": _WHY.get(w, "seam after rank %d not clean (status %d)" % (w - 1, recs[w - 1][0] if w > 0 else 0))}
    return {"ok": True, "offsets": [int(o) for o in offs], "total": int(total.value),
            "check": None if fmt == FMT_DEFLATE else int(check.value), "in_consumed": int(consumed.value)}


def inflate_sharded(eng, data, fmt, rank, world, dist, torch, device="cpu"):
    """decode ONE stream with all ranks.  Returns a dict: "sharded" (bool), "status", "total", "check",
    and this rank's part: "d_out" (device pointer or None), "offset", "len".  When the stream does not
    shard (see above) rank 0 holds everything and "status" is the ordinary single-GPU status."""
    cuts = shard_plan(data, world, lib=eng.lib)
    rec, d_out, n = shard_decode(eng, data, fmt, cuts, rank)
    mine = torch.tensor(rec, dtype=torch.int64, device=device)
    gathered = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)
    recs = [[int(x) for x in g.cpu().tolist()] for g in gathered]
    v = shard_verdict(recs, data, fmt, cuts, lib=eng.lib)
    if v["ok"]:
        return {"sharded": True, "status": 0, "total": v["total"], "check": v["check"],
                "in_consumed": v["in_consumed"], "d_out": d_out, "offset": v["offsets"][rank], "len": n}
    if d_out is not None:
        eng.free(d_out)
    out = {"sharded": False, "why": v["why"], "d_out": None, "offset": 0, "len": 0}
    rec = [0] * 8
    if rank == 0:   # the ordinary path, on one GPU: its result IS the answer
        res, d_o = eng.inflate_to_device(data, fmt)
        got = int(res.out_len) if res.status >= 0 else 0
        rec = [int(res.status), got, int(res.in_consumed), len(data), int(res.adler32 if fmt == FMT_ZLIB else res.crc32), 0, 0, 0]
        out.update(d_out=d_o, len=got)
    t = torch.tensor(rec, dtype=torch.int64, device=device)
    dist.broadcast(t, src=0)
    rec = [int(x) for x in t.cpu().tolist()]
    out.update(status=rec[0], total=rec[1], check=rec[4], in_consumed=rec[2])
    return out
