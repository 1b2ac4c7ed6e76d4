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
        self._full = None          # scratch vector of the last replay
        self._delivered = 0        # output octets handed out so far


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


def decompress(context, state, engine=None):
    """api.lisp:3-10.  One call over everything the context holds, on the device.

    All three outcomes are reported as the reference does — finished / input-underrun / output-overflow flags,
    octet count, the correct prefix in the buffer.

    RESUMING (the chunked protocol of deflate.lisp:114-137: more input after input-underrun, a new buffer after
    output-overflow) is done by REPLAY on the device: the state keeps every input octet it has been given; a call
    that brings new input decodes the whole prefix again (one engine call into a scratch vector) and hands out
    the octets beyond those already delivered; a call that only brings a new output buffer hands out the next
    slice of the scratch vector.  Same flags, counts and octets as the reference call by call on valid streams
    (tests: case_chunked_resume); cost O(prefix) per call that brings input, so it suits a few large chunks —
    a device-resident session that restarts at the last block boundary is SURVEY §8f-2, next.  Deviation: a
    stream that turns out to be INVALID is reported when the replay first meets the error, which can be a call
    earlier than the reference (which first hands out the output before the error, and a checksum
    mismatch only after the last octet)."""
    eng = engine or default_engine()
    first = state._calls == 0
    state._calls += 1
    state.input_underrun = False
    state.output_overflow = False
    new = bytes(memoryview(context.octet_vector)[context.offset:context.end])
    if first:
        out = state.output_buffer
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
    # ---- resume by replay
    if new or state._full is None:
        state._seen += new
        context.offset = context.end
        size = eng.inflate_size(bytes(state._seen), state.format)
        if size.status < 0:
            raise ThreeBzError(size.status, eng.strerror(size.status))
        full = bytearray(size.out_total)
        res = eng.inflate(bytes(state._seen), state.format, full)
        state.result = res
        state._full = full
        state._full_status = res.status
        state._full_flags = res.flags
    full, status = state._full, state._full_status
    if status < 0:
        raise ThreeBzError(status, eng.strerror(status))
    avail = len(full)
    off = state.output_offset
    give = max(0, min(len(state.output_buffer) - off, avail - state._delivered))
    state.output_buffer[off:off + give] = full[state._delivered:state._delivered + give]
    state.output_offset = off + give
    state._delivered += give
    pending = avail - state._delivered
    if pending > 0:
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
