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

    def inflate_alloc(self, data, fmt, start=0, end=None):
        """one decode, the result buffer allocated once the size is known: returns (result, bytearray or None)"""
        end = len(data) if end is None else end
        res = _lib.Result()
        box = {}

        def alloc(_user, n):
            box["buf"] = bytearray(n)
            return C.addressof((C.c_char * n).from_buffer(box["buf"])) if n else 0

        cb = _lib.ALLOC_FN(alloc)
        base = _addr(data)
        self._check(self.lib.tbz_inflate_alloc(self._ctx, fmt, (base or 0) + start if base else None, end - start, cb,
                                               None, C.byref(res)))
        return res, box.get("buf")

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

    def inflate_to_device(self, data, fmt, start=0, end=None):
        """one decode of host octets into device memory the caller owns (self.free): (result, device pointer or None)"""
        end = len(data) if end is None else end
        res, p = _lib.Result(), C.c_void_p()
        base = _addr(data)
        self._check(self.lib.tbz_inflate_to_device(self._ctx, fmt, (base or 0) + start if base else None, end - start,
                                                   C.byref(p), C.byref(res)))
        return res, p.value

    @staticmethod
    def inflate_batch_multi(engines, datas, fmt, outs):
        """n streams over several contexts (one per device): tbz_inflate_batch_multi — LPT assignment in C, one host thread
        per context, results in stream order"""
        n, k = len(datas), len(engines)
        lib = engines[0].lib
        ctxs = (C.c_void_p * k)(*[e._ctx for e in engines])
        ins = (C.c_void_p * n)(*[_addr(d) for d in datas])
        il = (C.c_size_t * n)(*[len(d) for d in datas])
        os_ = (C.c_void_p * n)(*[_addr(o) for o in outs])
        ol = (C.c_size_t * n)(*[len(o) for o in outs])
        res = (_lib.Result * n)()
        engines[0]._check(lib.tbz_inflate_batch_multi(ctxs, k, fmt, n, ins, il, os_, ol, res))
        return list(res)

    @staticmethod
    def inflate_batch_multi_device(engines, parts, fmt):
        """streams already resident on the devices: parts[k] = (d_in, in_offs, in_lens, d_out, out_offs, out_caps) for
        engines[k] — per context the arguments of inflate_batch_device — all contexts at once (tbz_inflate_batch_multi_device).
        Returns one list of results per context."""
        k = len(engines)
        lib = engines[0].lib
        u64p, vp = C.POINTER(C.c_uint64), C.c_void_p
        ctxs = (vp * k)(*[e._ctx for e in engines])
        ns = (C.c_size_t * k)(*[len(q[1]) for q in parts])
        arrs = [[(C.c_uint64 * max(1, len(q[j])))(*q[j]) for j in (1, 2, 4, 5)] for q in parts]
        col = lambda j: (u64p * k)(*[C.cast(a[j], u64p) for a in arrs])
        res = [(_lib.Result * max(1, len(q[1])))() for q in parts]
        rp = (C.POINTER(_lib.Result) * k)(*[C.cast(r, C.POINTER(_lib.Result)) for r in res])
        engines[0]._check(lib.tbz_inflate_batch_multi_device(ctxs, k, fmt, ns, (vp * k)(*[q[0] for q in parts]), col(0), col(1),
                                                             (vp * k)(*[q[3] for q in parts]), col(2), col(3), rp))
        return [list(r)[:len(q[1])] for r, q in zip(res, parts)]

    @staticmethod
    def inflate_sharded_multi(engines, data, fmt, out):
        """ONE flush-delimited stream over several contexts in one call (tbz_inflate_sharded_multi): (result, sharded?)"""
        k = len(engines)
        ctxs = (C.c_void_p * k)(*[e._ctx for e in engines])
        res, sh = _lib.Result(), C.c_int(0)
        engines[0]._check(engines[0].lib.tbz_inflate_sharded_multi(ctxs, k, fmt, _addr(data), len(data), _addr(out),
                                                                   len(out) if out is not None else 0, C.byref(res), C.byref(sh)))
        return res, bool(sh.value)

    def assign_streams(self, sizes, parts):
        """tbz_assign_streams: owner[i] of stream i among `parts` contexts / ranks"""
        n = len(sizes)
        owner = (C.c_uint32 * n)()
        self._check(self.lib.tbz_assign_streams((C.c_size_t * n)(*sizes), n, parts, owner))
        return list(owner)

    # ---- device buffers (raw device pointers as ints, e.g. torch.Tensor.data_ptr())
    def inflate_device(self, d_in, in_len, d_out, out_cap, fmt):
        res = _lib.Result()
        self._check(self.lib.tbz_inflate_device(self._ctx, fmt, d_in, in_len, d_out, out_cap, C.byref(res)))
        return res

    @staticmethod
    def u64_array(v):
        """the C ABI's offset / length arrays; a caller that decodes the same batch shape again keeps them"""
        return v if isinstance(v, C.Array) else (C.c_uint64 * len(v))(*v)

    def inflate_batch_device(self, d_in, in_offs, in_lens, d_out, out_offs, out_caps, fmt, raw=False):
        """raw=True: the tbz_result records as the ctypes array the library filled (indexable, not copied)"""
        n = len(in_offs)
        a = self.u64_array
        res = (_lib.Result * n)()
        self._check(self.lib.tbz_inflate_batch_device(self._ctx, fmt, n, d_in, a(in_offs), a(in_lens), d_out,
                                                      a(out_offs), a(out_caps), res))
        return res if raw else list(res)

    def inflate_gzip_members_device(self, d_in, in_len, d_out, out_cap, max_members):
        """every member of a concatenated gzip file, input and output in HBM: (results, in offsets, out offsets)"""
        res = (_lib.Result * max_members)()
        io, oo = (C.c_uint64 * max_members)(), (C.c_uint64 * max_members)()
        n = C.c_size_t()
        self._check(self.lib.tbz_inflate_gzip_members_device(self._ctx, d_in, in_len, d_out, out_cap, max_members, res, io, oo,
                                                             C.byref(n)))
        return res[:n.value], list(io[:n.value]), list(oo[:n.value])

    def inflate_gzip_members(self, data, start=0, end=None, max_members=1 << 20):
        """the same over host octets: (results, in offsets, list of bytearrays — None for a member that failed)"""
        end = len(data) if end is None else end
        res = (_lib.Result * max_members)()
        io = (C.c_uint64 * max_members)()
        n = C.c_size_t()
        bufs = []

        def alloc(_user, k):
            bufs.append(bytearray(k))
            return C.addressof((C.c_char * k).from_buffer(bufs[-1])) if k else 0

        cb = _lib.ALLOC_FN(alloc)
        base = _addr(data)
        self._check(self.lib.tbz_inflate_gzip_members(self._ctx, (base or 0) + start if base else None, end - start, cb, None,
                                                      max_members, res, io, C.byref(n)))
        out, it = [], iter(bufs)
        for r in res[:n.value]:
            out.append(next(it) if r.status >= 0 else None)
        return res[:n.value], list(io[:n.value]), out

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

    # ---- sessions (the chunked protocol with the state on the device)
    def session_create(self, fmt):
        p = C.c_void_p()
        self._check(self.lib.tbz_session_create(self._ctx, fmt, C.byref(p)))
        return p

    def session_destroy(self, sess):
        if sess and getattr(self, "_ctx", None):
            self.lib.tbz_session_destroy(sess)

    def session_feed(self, sess, addr, n, on_device=False):
        self._check(self.lib.tbz_session_feed(sess, addr, n, 1 if on_device else 0))

    def session_decompress(self, sess, out_addr, cap):
        res = _lib.Result()
        self._check(self.lib.tbz_session_decompress(sess, out_addr, cap, C.byref(res)))
        return res

    def session_stats(self, sess):
        """(engine calls made by the session, input octets handed to them — re-decoded ones counted again)"""
        a, b = C.c_uint64(), C.c_uint64()
        self._check(self.lib.tbz_session_stats(sess, C.byref(a), C.byref(b)))
        return a.value, b.value

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


class OctetStreamContext:
    """make-octet-stream-context (io-common.lisp:47-63): a binary input file stream + context-boxes.  The reference
    reads such a context octet by octet ("very slow", README.md); here what the boxes span, from offset to end, is read
    in one go and handed to the state's session (the Lisp shim does the same with one READ-SEQUENCE)."""

    def __init__(self, octet_stream, start=0, offset=0, end=None):
        if not valid_octet_stream(octet_stream):   # (assert (valid-octet-stream file-stream)) io-common.lisp:55
            raise ThreeBzError(-22, "not an open binary input stream")
        self.octet_stream = octet_stream
        self.start = start
        self.offset = offset
        if end is None:   # (file-length file-stream)
            here = octet_stream.tell()
            end = octet_stream.seek(0, 2)
            octet_stream.seek(here)
        self.end = end


def valid_octet_stream(os_):
    """io-common.lisp:65-69: an open input stream of octets"""
    try:
        return (not os_.closed) and os_.readable() and os_.seekable() and isinstance(os_.read(0), (bytes, bytearray))
    except Exception:
        return False


def make_octet_stream_context(file_stream, start=0, offset=0, end=None):
    return OctetStreamContext(file_stream, start, offset, end)


def resync_file_stream(context):
    """%resync-file-stream (io-common.lisp:57-63): put the stream's file position where the context stands — after a
    finished stream that is just behind its trailer, so that the caller can go on reading what follows.  A no-op for
    contexts that are not streams, as the reference's default method is."""
    if isinstance(context, OctetStreamContext):
        context.octet_stream.seek(context.offset)


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
        self._sess = None          # tbz_session: the resumable part of the state lives on the device
        self._eng = None
        self._fed = 0              # input octets given to the session so far

    def __del__(self):
        try:
            if self._sess is not None and self._eng is not None:
                self._eng.session_destroy(self._sess)
                self._sess = None
        except Exception:
            pass


class ZlibState(DeflateState):
    """zlib-state (zlib.lisp:3-12)"""
    format = FORMATS["zlib"]
    format_name = "zlib"


class GzipState(DeflateState):
    """gzip-state (gzip.lisp:3-28).  The metadata slots are filled as the header arrives, with the reference's values
    (gzip.lisp:123-241; keywords as strings): compression_method "deflate", flags a list of "text" / "header-crc" /
    "extra" / "name" / "comment" in the reference's (pushed) order, extra octets, name / comment strings (utf-8 with
    substitution, as the reference's :errorp nil decodes them), operating_system, mtime_unix / mtime_universal (None when the header's MTIME is 0),
    compression_level "maximum" / "fastest" / the XFL octet."""
    format = FORMATS["gzip"]
    format_name = "gzip"

    def __init__(self, output_buffer=None):
        super().__init__(output_buffer)
        self.compression_method = None
        self.flags = None
        self.extra = None
        self.name = None
        self.comment = None
        self.operating_system = None
        self.mtime_unix = None
        self.mtime_universal = None
        self.compression_level = "default"
        self._hdr = bytearray()    # the header's octets as far as they have arrived
        self._hdr_done = False

    def gzip_meta(self):
        return {k: getattr(self, k) for k in ("compression_method", "flags", "extra", "name", "comment",
                                              "operating_system", "mtime_unix", "mtime_universal", "compression_level")}


_GZ_OS = ("fat", "amiga", "vms", "unix", "vm/cms", "atari-tos", "hpfs", "macintosh", "z-system", "cp/m", "tops-20",
          "ntfs", "qdos", "acorn-riscos")


def _note_gzip_header(eng, state, octets):
    """gzip.lisp:123-241: the metadata slots, from the header octets the state has been given so far (host side:
    tbz_gzip_header_parse; the device only skips the header)"""
    if state._hdr_done:
        return
    state._hdr += octets[:max(0, 70000 - len(state._hdr))]
    h = _lib.GzipHeader()
    hb = bytes(state._hdr)
    eng.lib.tbz_gzip_header_parse(_addr(hb), len(hb), C.byref(h))
    stage = h.stage    # how far the parse got: 2 cm+flg, 3 mtime, 4 xfl+os, 5 extra, 6 name, 7 comment

    def text(off, n):
        # gzip.lisp:214-217, :236-239: (babel:octets-to-string ... :encoding :utf-8 :errorp nil) — with :errorp nil babel
        # substitutes what is not utf-8 and returns, so the reference's iso-8859-1 branch is never reached.  (How many
        # replacement characters an invalid run yields is babel's business and not pinned by any reference test.)
        return bytes(hb[off:off + n]).decode("utf-8", errors="replace")
    if stage >= 2:
        state.compression_method = "deflate"
        state.flags = [k for bit, k in ((4, "comment"), (3, "name"), (2, "extra"), (1, "header-crc"), (0, "text"))
                       if h.flg >> bit & 1]
    if stage >= 3 and h.mtime:
        state.mtime_unix = h.mtime
        state.mtime_universal = h.mtime + 2208988800
    if stage >= 4:
        state.compression_level = {2: "maximum", 4: "fastest"}.get(h.xfl, h.xfl)
        state.operating_system = _GZ_OS[h.os] if h.os <= 13 else ("unknown", h.os)
    if stage >= 5 and h.flg & 4:
        state.extra = hb[h.extra_off:h.extra_off + h.extra_len]
    if stage >= 6 and h.flg & 8:
        state.name = text(h.name_off, h.name_len)
    if stage >= 7 and h.flg & 16:
        state.comment = text(h.comment_off, h.comment_len)
    if h.status != _lib.INPUT_UNDERRUN:
        state._hdr_done = True   # complete, or an error the decode will raise


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


def _out_addr(buf, offset):
    if len(buf) - offset <= 0:
        return None
    if isinstance(buf, bytearray) or isinstance(buf, memoryview):
        return C.addressof((C.c_char * len(buf)).from_buffer(buf)) + offset
    return buf.ctypes.data + offset  # numpy uint8 array


def decompress(context, state, engine=None):
    """api.lisp:3-10: decode what the context holds into the state's output buffer, from output-offset on.

    All three outcomes are reported as the reference does — finished / input-underrun / output-overflow flags, the
    octet count, the octets in the buffer — and the call can be repeated as in the reference's chunked protocol
    (deflate.lisp:114-137): with a context that brings more input after input-underrun, with a new buffer
    (replace-output-buffer) after output-overflow.  The resumable part of the state is a tbz_session on the device
    (include/tbz_amd.h): unconsumed input, the 32 KiB window, octets decoded beyond the buffer's end; a call costs the
    new input plus the one block it continues.  A stream that turns out to be invalid raises in the call in which the
    reference would have met the error, after the output before it has been handed out.

    A pointer context's memory goes to the session as it is: a host pointer is copied once to the device, a device
    pointer (octet-pointer over HBM) device to device."""
    eng = engine or default_engine()
    if state._sess is None:
        state._sess = eng.session_create(state.format)
        state._eng = eng
    state._calls += 1
    state.input_underrun = False
    state.output_overflow = False
    if state.finished:
        return state.output_offset
    fed_before = state._fed
    start_offset = context.offset
    n_in = context.end - context.offset
    if isinstance(context, OctetPointerContext):
        if not valid_octet_pointer(context.op):   # (assert (valid-octet-pointer (op context))) io-mmap.lisp:66
            raise ThreeBzError(-22, "octet pointer used outside its scope (or null / empty)")
        if n_in > 0:
            eng.session_feed(state._sess, context.pointer + context.offset, n_in, on_device=context.op.device)
            if isinstance(state, GzipState) and not state._hdr_done:
                head = min(n_in, 70000)
                if context.op.device:
                    hb = bytearray(head)
                    eng.d2h(hb, context.pointer + context.offset, head)
                else:
                    hb = C.string_at(context.pointer + context.offset, head)
                _note_gzip_header(eng, state, bytes(hb))
    elif isinstance(context, OctetStreamContext):
        if not valid_octet_stream(context.octet_stream):   # (assert (valid-octet-stream …)) io.lisp:71
            raise ThreeBzError(-22, "not an open binary input stream")
        data = b""
        if n_in > 0:
            context.octet_stream.seek(context.offset)
            data = context.octet_stream.read(n_in)
            n_in = len(data)   # (a file shorter than the boxes say: what is there)
        if n_in > 0:
            eng.session_feed(state._sess, _addr(data), n_in)
            if isinstance(state, GzipState):
                _note_gzip_header(eng, state, data)
    elif n_in > 0:
        mv = memoryview(context.octet_vector)[context.offset:context.end]
        data = bytes(mv)   # (pinned for the duration of the call, as cffi:with-pointer-to-vector-data does)
        eng.session_feed(state._sess, _addr(data), n_in)
        if isinstance(state, GzipState):
            _note_gzip_header(eng, state, data)
    if n_in > 0:
        state._fed += n_in
        context.offset = context.offset + n_in if isinstance(context, OctetStreamContext) else context.end
    out = state.output_buffer
    off = state.output_offset
    res = eng.session_decompress(state._sess, _out_addr(out, off), max(0, len(out) - off))
    state.result = res
    state.output_offset = off + int(res.out_len)
    if res.status < 0:
        raise ThreeBzError(res.status, eng.strerror(res.status))
    state.finished = res.status == _lib.FINISHED
    state.input_underrun = res.status == _lib.INPUT_UNDERRUN
    state.output_overflow = res.status == _lib.OUTPUT_OVERFLOW
    if state.finished:
        # the context stands just behind the stream (its trailer included), as the reference leaves it
        context.offset = start_offset + max(0, int(res.in_consumed) - fed_before)
        resync_file_stream(context)   # (io.lisp:102-104: the stream's position follows the boxes)
    # the reference's early return: gzip's final block decoded but crc32 / ISIZE cut off -> (return-from decompress-gzip 0)
    if state.input_underrun and state.format == FORMATS["gzip"] and (res.flags & 2):
        return 0
    return state.output_offset


def decompress_vector(compressed, format="zlib", start=0, end=None, output=None, engine=None):
    """api.lisp:23-65 — returns (buffer, count).

    With `output`: a single call; not finished => error "incomplete ~a stream" / "not enough space
    to decompress ~a stream" (api.lisp:41-47).  Without: the reference grows 32 KiB buffers by
    doubling and gathers (api.lisp:48-65); here the engine decodes once and the result buffer is
    allocated when the size is known (tbz_inflate_alloc: one input copy, one Huffman pass)."""
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
    res, buf = eng.inflate_alloc(compressed, fmt, start=start, end=end)
    if res.status < 0:
        raise ThreeBzError(res.status, eng.strerror(res.status))
    if res.status != _lib.FINISHED:   # (assert (not (ds-input-underrun state))) api.lisp:55
        raise ThreeBzError(-20, "incomplete %s stream" % name)
    return (buf if buf is not None else bytearray(0)), res.out_len


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
    (as gzip(1) does).  One call of tbz_inflate_gzip_members: candidates, batch and walk run inside the library."""
    eng = engine or default_engine()
    res, _offs, bufs = eng.inflate_gzip_members(compressed, start=start, end=end)
    members = []
    for r, b in zip(res, bufs):
        if r.status < 0:
            raise ThreeBzError(r.status, eng.strerror(r.status))
        if r.status != _lib.FINISHED:
            raise ThreeBzError(-20, "incomplete gzip stream")
        members.append(b)
    return members
