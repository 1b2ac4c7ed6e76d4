"""Host-side mirror of 3bz's exported API (package.lisp:13-27) over the C ABI.

The reference's host language is Common Lisp; no Lisp implementation exists in this image, so the
shim that a Lisp user would load is lisp/3bz-amd.lisp (CFFI, mechanical, untestable here) and THIS
module is the same surface in Python — same names, argument meaning, return values and error
behaviour — so the parity tests read like the reference's REPL tests:

    (decompress-vector v :format :zlib :output out)  ->  decompress_vector(v, format="zlib", output=out)
    (make-zlib-state :output-buffer b)               ->  make_zlib_state(output_buffer=b)
    (decompress ctx state)                           ->  decompress(ctx, state)
    (finished s) (input-underrun s) (output-overflow s)

Everything runs on the MI355X through lib3bz_amd.so; nothing here decodes on the CPU.
"""
import ctypes as C

from . import _lib

FORMATS = _lib.FORMATS


class ThreeBzError(Exception):
    """a Lisp `error` / `assert` / `ecase` failure of the reference (SURVEY §8a contract list)"""

    def __init__(self, code, message):
        super().__init__(message)
        self.code = code


class EngineError(RuntimeError):
    """the engine itself failed (HIP error, bad argument)"""

    def __init__(self, code, message):
        super().__init__("tbz engine error %d: %s" % (code, message))
        self.code = code


def _addr(buf):
    if buf is None:
        return None
    if isinstance(buf, bytes):
        return C.cast(C.c_char_p(buf), C.c_void_p).value
    if isinstance(buf, bytearray):
        return C.addressof((C.c_char * len(buf)).from_buffer(buf)) if len(buf) else None
    if isinstance(buf, memoryview):
        return C.addressof((C.c_char * len(buf)).from_buffer(buf)) if len(buf) else None
    return buf.ctypes.data  # numpy uint8 array


class Engine:
    """one tbz_ctx: a HIP stream + scratch pools on one device"""

    def __init__(self, device=0, lib_path=None):
        self.lib = _lib.load(lib_path)
        p = C.c_void_p()
        r = self.lib.tbz_ctx_create(device, C.byref(p))
        if r != 0:
            raise EngineError(r, self.lib.tbz_strerror(r).decode())
        self._ctx = p
        self.device = device

    def close(self):
        if getattr(self, "_ctx", None):
            self.lib.tbz_ctx_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, r):
        if r != 0:
            raise EngineError(r, self.lib.tbz_strerror(r).decode() + ": " +
                              self.lib.tbz_last_error(self._ctx).decode())

    # ---- host buffers
    def inflate(self, data, fmt, out, start=0, end=None):
        end = len(data) if end is None else end
        res = _lib.Result()
        base = _addr(data)
        self._check(self.lib.tbz_inflate(self._ctx, fmt, (base or 0) + start if base else None, end - start,
                                         _addr(out), len(out) if out is not None else 0, C.byref(res)))
        return res

    def inflate_size(self, data, fmt, start=0, end=None):
        end = len(data) if end is None else end
        res = _lib.Result()
        base = _addr(data)
        self._check(self.lib.tbz_inflate_size(self._ctx, fmt, (base or 0) + start if base else None, end - start,
                                              C.byref(res)))
        return res

    def inflate_batch(self, datas, fmt, outs):
        n = len(datas)
        ins = (C.c_void_p * n)(*[_addr(d) for d in datas])
        il = (C.c_size_t * n)(*[len(d) for d in datas])
        os_ = (C.c_void_p * n)(*[_addr(o) for o in outs])
        ol = (C.c_size_t * n)(*[len(o) for o in outs])
        res = (_lib.Result * n)()
        self._check(self.lib.tbz_inflate_batch(self._ctx, fmt, n, ins, il, os_, ol, res))
        return list(res)

    # ---- device buffers (raw device pointers as ints, e.g. torch.Tensor.data_ptr())
    def inflate_device(self, d_in, in_len, d_out, out_cap, fmt):
        res = _lib.Result()
        self._check(self.lib.tbz_inflate_device(self._ctx, fmt, d_in, in_len, d_out, out_cap, C.byref(res)))
        return res

    def inflate_batch_device(self, d_in, in_offs, in_lens, d_out, out_offs, out_caps, fmt):
        n = len(in_offs)
        a = lambda v: (C.c_uint64 * n)(*v)
        res = (_lib.Result * n)()
        self._check(self.lib.tbz_inflate_batch_device(self._ctx, fmt, n, d_in, a(in_offs), a(in_lens), d_out,
                                                      a(out_offs), a(out_caps), res))
        return list(res)

    def adler32_device(self, d_buf, n, s1=1, s2=0):
        o1, o2 = C.c_uint32(), C.c_uint32()
        self._check(self.lib.tbz_adler32_device(self._ctx, d_buf, n, s1, s2, C.byref(o1), C.byref(o2)))
        return o1.value, o2.value

    def crc32_device(self, d_buf, n, crc=0):
        o = C.c_uint32()
        self._check(self.lib.tbz_crc32_device(self._ctx, d_buf, n, crc, C.byref(o)))
        return o.value

    def malloc(self, n):
        p = C.c_void_p()
        self._check(self.lib.tbz_device_malloc(self._ctx, n, C.byref(p)))
        return p.value

    def free(self, p):
        self._check(self.lib.tbz_device_free(self._ctx, p))

    def h2d(self, d_dst, data):
        self._check(self.lib.tbz_memcpy_h2d(self._ctx, d_dst, _addr(data), len(data)))

    def d2h(self, out, d_src, n=None):
        self._check(self.lib.tbz_memcpy_d2h(self._ctx, _addr(out), d_src, len(out) if n is None else n))

    def trim(self):
        """release the context's device scratch (it only grows otherwise)"""
        self._check(self.lib.tbz_ctx_trim(self._ctx))

    def timings(self):
        t = _lib.Timings()
        self._check(self.lib.tbz_last_timings(self._ctx, C.byref(t)))
        return t

    def strerror(self, code):
        return self.lib.tbz_strerror(code).decode()


_default = None


def default_engine():
    global _default
    if _default is None:
        _default = Engine(0)
    return _default


def set_default_engine(e):
    global _default
    _default = e


# ------------------------------------------------------------------------------------------------
# 3bz API surface
# ------------------------------------------------------------------------------------------------
class OctetVectorContext:
    """make-octet-vector-context (io-common.lisp:40-45): vector + context-boxes (start, end, offset)"""

    def __init__(self, vector, start=0, offset=None, end=None):
        self.octet_vector = vector
        self.start = start
        self.end = len(vector) if end is None else end
        self.offset = start if offset is None else offset


def make_octet_vector_context(vector, start=0, offset=None, end=None):
    return OctetVectorContext(vector, start, offset, end)


class OctetPointer:
    """octet-pointer (io-mmap.lisp:21-24): foreign memory [base, base+size) valid inside a scope.  `device=True`:
    the memory is in HBM (a raw device pointer, e.g. torch.Tensor.data_ptr()) — the engine reads it in place."""

    def __init__(self, base, size, device=False):
        self.base, self.size, self.device = int(base or 0), int(size), bool(device)
        self.scope = True


def valid_octet_pointer(op):
    """io-mmap.lisp:37-40"""
    return bool(op.scope and op.base and op.size > 0)


class with_octet_pointer:
    """(with-octet-pointer (var pointer size) …) — io-mmap.lisp:26-35: the pointer object dies with the scope.
    `pointer` may also be a Python buffer (bytes, bytearray, numpy uint8): its address is taken, as
    cffi:with-pointer-to-vector-data does for the reference's callers (bench.lisp:61)."""

    def __init__(self, pointer, size=None, device=False):
        self._keep = None
        if not isinstance(pointer, int):
            self._keep = pointer
            size = len(pointer) if size is None else size
            pointer = _addr(pointer)
        self.op = OctetPointer(pointer, size, device)

    def __enter__(self):
        return self.op

    def __exit__(self, *exc):
        self.op.scope = False
        return False


class OctetPointerContext:
    """make-octet-pointer-context (io-mmap.lisp:47-54)"""

    def __init__(self, octet_pointer, start=0, offset=0, end=None):
        self.op = octet_pointer
        self.pointer = octet_pointer.base
        self.start = start
        self.offset = offset
        self.end = octet_pointer.size if end is None else end


def make_octet_pointer_context(octet_pointer, start=0, offset=0, end=None):
    return OctetPointerContext(octet_pointer, start, offset, end)


def _context_octets(eng, context):
    """the octets [offset, end) of a context as bytes (a copy: the states keep their input for resuming)"""
    if isinstance(context, OctetPointerContext):
        if not valid_octet_pointer(context.op):   # (assert (valid-octet-pointer (op context))) io-mmap.lisp:66
            raise ThreeBzError(-22, "octet pointer used outside its scope (or null / empty)")
        n = context.end - context.offset
        if context.op.device:
            out = bytearray(n)
            if n:
                eng.d2h(out, context.pointer + context.offset, n)
            return bytes(out)
        return C.string_at(context.pointer + context.offset, n)
    return bytes(memoryview(context.octet_vector)[context.offset:context.end])


class DeflateState:
    """deflate-state (deflate.lisp:4-62): only the slots a caller can observe"""
    format = FORMATS["deflate"]
    format_name = "deflate"

    def __init__(self, output_buffer=None):
        self.output_buffer = output_buffer if output_buffer is not None else bytearray(0)
        self.output_offset = 0
        self.finished = False
        self.output_overflow = False
        self.input_underrun = False
        self._calls = 0
        self.result = None
        self._seen = bytearray()   # every input octet given to this state (resume replays it)
        self._full = None          # scratch vector of the last replay: output octets from _full_out on
        self._full_out = 0
        self._delivered = 0        # output octets handed out so far
        # resume base: a proven block boundary (tbz_result.in_consumed) from which replays start
        self._base_in = 0          # input octets before it
        self._base_out = 0         # output octets before it
        self._base_ck = None       # checksum of those output octets (None: the format's initial value)
        self._cand = None          # (in, out, ck) of the latest boundary reported, adopted once its output is delivered
        self._no_base = False      # the stream needs history across boundaries (or failed): replay from 0


class ZlibState(DeflateState):
    """zlib-state (zlib.lisp:3-12)"""
    format = FORMATS["zlib"]
    format_name = "zlib"


class GzipState(DeflateState):
    """gzip-state (gzip.lisp:3-28)"""
    format = FORMATS["gzip"]
    format_name = "gzip"


def make_deflate_state(output_buffer=None):
    return DeflateState(output_buffer)


def make_zlib_state(output_buffer=None):
    return ZlibState(output_buffer)


def make_gzip_state(output_buffer=None):
    return GzipState(output_buffer)


def finished(state):
    """api.lisp:67-68"""
    return state.finished


def input_underrun(state):
    """api.lisp:69-70"""
    return state.input_underrun


def output_overflow(state):
    """api.lisp:71-72"""
    return state.output_overflow


def replace_output_buffer(state, buffer):
    """api.lisp:12-21"""
    if not (state.output_offset == 0 or state.output_overflow):
        raise ThreeBzError(-19, "can't switch buffers without filling old one yet.")
    state.output_buffer = buffer
    state.output_offset = 0
    state.output_overflow = False


def _chain_checksum(eng, fmt, d_buf, n, ck):
    """checksum of n more output octets at d_buf, continuing from ck (None = initial value)"""
    if fmt == FORMATS["zlib"]:
        a = 1 if ck is None else ck
        s1, s2 = eng.adler32_device(d_buf, n, a & 0xFFFF, a >> 16)
        return s1 | (s2 << 16)
    return eng.crc32_device(d_buf, n, 0 if ck is None else ck)


def _replay(eng, state):
    """decode everything the state has seen from its resume base on (one engine call into a device scratch
    buffer).  Leaves the octets in state._full (they continue the output at state._full_out), the status in
    state._full_status / _full_flags and the latest proven boundary in state._cand."""
    deflate = FORMATS["deflate"]
    while True:
        base_in, base_out = state._base_in, state._base_out
        tail = bytes(state._seen[base_in:])
        fmt = state.format if base_in == 0 else deflate   # past the container header: blocks only
        status, flags, full, cand = 0, 0, bytearray(0), None
        size = eng.inflate_size(tail, fmt)
        res = size
        status = size.status
        if status >= 0:
            n = int(size.out_total)
            d_in, d_out = eng.malloc(len(tail) + 64), eng.malloc(n + 64)
            try:
                eng.h2d(d_in, tail)
                # (one octet of slack: a stored block cut off exactly at the end of a FULL buffer is output-overflow in
                # the reference, deflate.lisp:538-573 — this scratch buffer must never be the reason for a status)
                res = eng.inflate_device(d_in, len(tail), d_out, n + 1, fmt)
                status, flags = res.status, res.flags
                got = int(res.out_len) if status >= 0 else 0
                if status == _lib.FINISHED and base_in and state.format != deflate:
                    # the engine decoded raw blocks: the container's trailer is checked here as zlib.lisp:80-95 /
                    # gzip.lisp:78-106 do (checksum of ALL output = the base's, continued over these octets)
                    ck = _chain_checksum(eng, state.format, d_out, got, state._base_ck)
                    end = int(res.in_consumed)
                    have = len(tail) - end
                    flags |= 2
                    if state.format == FORMATS["zlib"]:
                        if have < 4:
                            status = _lib.INPUT_UNDERRUN
                        elif int.from_bytes(tail[end:end + 4], "big") != ck:
                            status = -11
                    else:
                        if have < 4:
                            status = _lib.INPUT_UNDERRUN
                        elif int.from_bytes(tail[end:end + 4], "little") != ck:
                            status = -16
                        elif have < 8:
                            status = _lib.INPUT_UNDERRUN
                elif status == _lib.INPUT_UNDERRUN and not (flags & 2) and res.in_consumed > 0 and not state._no_base:
                    b_out = int(res.boundary_out)
                    ck = None if state.format == deflate else _chain_checksum(eng, state.format, d_out, b_out,
                                                                              state._base_ck)
                    cand = (base_in + int(res.in_consumed), base_out + b_out, ck)
                if status >= 0 and got:
                    full = bytearray(got)
                    eng.d2h(full, d_out, got)
            finally:
                eng.free(d_in)
                eng.free(d_out)
        if status < 0 and base_in:
            # blocks that reach back across the boundary (sync-flush history) fail as "distance before start";
            # whatever the reason, the whole stream decides: replay from the first octet, for good
            state._base_in, state._base_out, state._base_ck, state._cand, state._no_base = 0, 0, None, None, True
            continue
        state.result = res
        state._full, state._full_out = full, base_out
        state._full_status, state._full_flags = status, flags
        if cand is not None:
            state._cand = cand
        return


def decompress(context, state, engine=None):
    """api.lisp:3-10.  One call over everything the context holds, on the device.

    All three outcomes are reported as the reference does — finished / input-underrun / output-overflow flags,
    octet count, the correct prefix in the buffer.

    RESUMING (the chunked protocol of deflate.lisp:114-137: more input after input-underrun, a new buffer after
    output-overflow) is done by REPLAY on the device from the last proven block boundary: the state keeps the
    input octets it has been given; a call that brings new input decodes again from the resume base (one engine
    call into a device scratch buffer) and hands out the octets beyond those already delivered; a call that only
    brings a new output buffer hands out the next slice.  The resume base is the boundary the engine reports for
    an unfinished stream (tbz_result.in_consumed / boundary_out: the chain of blocks landed there), adopted once
    all output before it has been delivered; the tail is then decoded as raw deflate, the checksum continues from
    the base's (tbz_adler32_device / tbz_crc32_device chain) and the container trailer is compared here.  A tail
    that fails for any reason — blocks that copy from before the boundary, as after Z_SYNC_FLUSH, fail as
    "distance before start" — sends the state back to replaying from the first octet, so the answer is always
    the whole stream's.  Cost per call that brings input: O(octets since the last flush boundary) for
    flush-delimited streams, O(prefix) otherwise.  Same flags, counts and octets as the reference call by call on
    valid streams (tests: case_chunked_resume).  Deviation: a stream that turns out to be INVALID is reported
    when the replay first meets the error, which can be a call earlier than the reference (which first hands out
    the output before the error, and a checksum mismatch only after the last octet)."""
    eng = engine or default_engine()
    first = state._calls == 0
    state._calls += 1
    state.input_underrun = False
    state.output_overflow = False
    if first and isinstance(context, OctetPointerContext) and valid_octet_pointer(context.op):
        # foreign memory goes to the engine as it is: a host pointer through tbz_inflate, a device pointer through
        # tbz_inflate_device (no staging copy of the input); the octets are only copied if the state must resume
        out = state.output_buffer
        n_in = context.end - context.offset
        if context.op.device:
            d_out = eng.malloc(len(out) + 64)
            try:
                res = eng.inflate_device(context.pointer + context.offset, n_in, d_out, len(out), state.format)
                if res.status >= 0 and res.out_len:
                    eng.d2h(out, d_out, int(res.out_len))
            finally:
                eng.free(d_out)
        else:
            res = _lib.Result()
            eng._check(eng.lib.tbz_inflate(eng._ctx, state.format, context.pointer + context.offset, n_in,
                                           _addr(out), len(out), C.byref(res)))
        new = b"" if res.status in (_lib.FINISHED,) or res.status < 0 else _context_octets(eng, context)
    else:
        new = _context_octets(eng, context)
        res = None
    if first:
        out = state.output_buffer
        if res is None:
            res = eng.inflate(new, state.format, out)
        state.result = res
        state._seen = bytearray(new)
        state._full = None
        if res.status < 0:
            raise ThreeBzError(res.status, eng.strerror(res.status))
        state.finished = res.status == _lib.FINISHED
        state.input_underrun = res.status == _lib.INPUT_UNDERRUN
        state.output_overflow = res.status == _lib.OUTPUT_OVERFLOW
        state.output_offset = res.out_len
        state._delivered = res.out_len
        context.offset = context.end if not state.finished else context.offset + res.in_consumed
        # the reference's early returns: zlib header underrun and every gzip header/trailer underrun
        # `(return-from … 0)` (zlib.lisp:113-114, gzip.lisp:86,:99,:116…); otherwise output-offset
        if state.input_underrun and state.format == FORMATS["gzip"] and (res.flags & 2):
            return 0  # final block decoded but crc32 / ISIZE cut off: (return-from decompress-gzip 0)
        return res.out_len
    if state.finished:
        return state.output_offset
    # ---- resume by replay from the base
    if new or state._full is None:
        state._seen += new
        context.offset = context.end
        c = state._cand
        if c is not None and c[1] <= state._delivered and c[0] > state._base_in:
            state._base_in, state._base_out, state._base_ck = c   # everything before it has been handed out
        state._cand = None
        _replay(eng, state)
    full, status = state._full, state._full_status
    if status < 0:
        raise ThreeBzError(status, eng.strerror(status))
    avail = state._full_out + len(full)
    off = state.output_offset
    give = max(0, min(len(state.output_buffer) - off, avail - state._delivered))
    src = state._delivered - state._full_out
    state.output_buffer[off:off + give] = full[src:src + give]
    state.output_offset = off + give
    state._delivered += give
    pending = avail - state._delivered
    stored_cut = status == _lib.INPUT_UNDERRUN and (state._full_flags & 4) and \
        state.output_offset == len(state.output_buffer)
    if pending > 0 or stored_cut:
        # (stored_cut: the input ran out inside a stored block just where this buffer is full — the reference
        # asks for output space first there, deflate.lisp:538-573)
        state.output_overflow = True
        return state.output_offset
    state.finished = status == _lib.FINISHED
    state.input_underrun = status == _lib.INPUT_UNDERRUN
    if state.input_underrun and state.format == FORMATS["gzip"] and (state._full_flags & 2):
        return 0
    return state.output_offset


def decompress_vector(compressed, format="zlib", start=0, end=None, output=None, engine=None):
    """api.lisp:23-65 — returns (buffer, count).

    With `output`: a single call; not finished => error "incomplete ~a stream" / "not enough space
    to decompress ~a stream" (api.lisp:41-47).  Without: the reference grows 32 KiB buffers by
    doubling and gathers (api.lisp:48-65); here the size comes from the engine's count pass, so the
    result buffer is allocated exactly once."""
    fmt = FORMATS[format] if isinstance(format, str) else format
    name = format if isinstance(format, str) else {0: "deflate", 1: "zlib", 2: "gzip"}[fmt]
    end = len(compressed) if end is None else end
    eng = engine or default_engine()
    if output is not None:
        res = eng.inflate(compressed, fmt, output, start=start, end=end)
        if res.status < 0:
            raise ThreeBzError(res.status, eng.strerror(res.status))
        if res.status != _lib.FINISHED:
            if res.status == _lib.INPUT_UNDERRUN:
                raise ThreeBzError(-20, "incomplete %s stream" % name)
            raise ThreeBzError(-21, "not enough space to decompress %s stream" % name)
        return output, res.out_len
    q = eng.inflate_size(compressed, fmt, start=start, end=end)
    if q.status < 0:
        raise ThreeBzError(q.status, eng.strerror(q.status))
    if q.status == _lib.INPUT_UNDERRUN:  # (assert (not (ds-input-underrun state))) api.lisp:55
        raise ThreeBzError(-20, "incomplete %s stream" % name)
    buf = bytearray(q.out_total)
    res = eng.inflate(compressed, fmt, buf, start=start, end=end)
    if res.status < 0:
        raise ThreeBzError(res.status, eng.strerror(res.status))
    if res.status != _lib.FINISHED:
        raise ThreeBzError(-20, "incomplete %s stream" % name)
    return buf, res.out_len


# ------------------------------------------------------------------------------------------------
# Multi-member gzip.  3bz decodes the first member and stops (gzip.lisp:277-286): the
# caller is expected to call again with :start at the next member.  A file of many members is a batch
# of independent streams once their starts are known — and they are not (a member's length is only
# known after decoding it).  Same move as for flush markers: every `1f 8b 08` is a CANDIDATE start;
# all candidate ranges are decoded in one batch call, and a range is a member iff it FINISHED having
# consumed exactly its octets (header, blocks, CRC32, ISIZE — all verified by the engine).  The walk
# from offset 0 accepts proven members; a candidate inside a member's data (false magic) shows as a
# range that did not finish, and the member is decoded again over the merged range.
# ------------------------------------------------------------------------------------------------
GZIP_MAGIC = b"\x1f\x8b\x08"


def decompress_gzip_members(compressed, start=0, end=None, engine=None):
    """every member of a gzip file, in order: a list of bytearrays.  Each member is exactly what
    `(decompress-vector v :format :gzip :start member-offset)` returns; a damaged or incomplete member
    raises what that call would.  Octets after the last member that do not start a member are ignored
    (as gzip(1) does)."""
    eng = engine or default_engine()
    end = len(compressed) if end is None else end
    data = bytes(compressed[start:end])
    cands, p = [0], data.find(GZIP_MAGIC, 1)   # offset 0 is taken as given: the engine reports a bad magic there
    while p >= 0:
        cands.append(p)
        p = data.find(GZIP_MAGIC, p + 1)
    bounds = cands + [len(data)]

    def isize_hint(lo, hi):  # ISIZE (gzip.lisp:96-101, unchecked there) sizes the buffer; a wrong one just fails the range
        n = int.from_bytes(data[hi - 4:hi], "little") if hi - lo >= 18 else 0
        return n if n <= 1032 * (hi - lo) + 64 else 0

    ins = [data[bounds[i]:bounds[i + 1]] for i in range(len(cands))]
    outs = [bytearray(isize_hint(bounds[i], bounds[i + 1])) for i in range(len(cands))]
    res = eng.inflate_batch(ins, FORMATS["gzip"], outs)
    members, i = [], 0
    while i < len(cands):
        lo = bounds[i]
        if res[i].status == _lib.FINISHED and res[i].in_consumed == bounds[i + 1] - lo:
            del outs[i][res[i].out_len:]   # (ISIZE is not checked by 3bz: it may overstate)
            members.append(outs[i])
            i += 1
            continue
        # not a whole member: a later candidate lies inside this member's data, or the member is damaged,
        # or garbage follows it.  Decode from `lo` to the end (the ordinary one-stream call decides).
        q = eng.inflate_size(data, FORMATS["gzip"], start=lo)
        if q.status < 0:
            raise ThreeBzError(q.status, eng.strerror(q.status))
        if q.status == _lib.INPUT_UNDERRUN:
            raise ThreeBzError(-20, "incomplete gzip stream")
        buf = bytearray(q.out_total)
        r = eng.inflate(data, FORMATS["gzip"], buf, start=lo)
        if r.status < 0:
            raise ThreeBzError(r.status, eng.strerror(r.status))
        if r.status != _lib.FINISHED:
            raise ThreeBzError(-20, "incomplete gzip stream")
        members.append(buf)
        nxt = lo + r.in_consumed
        while i < len(cands) and bounds[i] < nxt:
            i += 1
        if i < len(cands) and bounds[i] != nxt:
            break  # what follows the member is not a member
    return members
