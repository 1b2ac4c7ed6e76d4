"""ctypes binding of the CPU oracle (oracle/tbz_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package (3bz_amd/) never imports it.

The Python surface mirrors 3bz's exported API (package.lisp:13-27) so parity
tests read like the reference's own REPL tests.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

DEFLATE, ZLIB, GZIP = 0, 1, 2
FORMATS = {"deflate": DEFLATE, "zlib": ZLIB, "gzip": GZIP}

ERR_NAMES = {
    -1: "btype", -2: "stored-len", -3: "oversubscribed", -4: "incomplete", -5: "repeat-no-prev",
    -6: "repeat-overrun", -7: "invalid-node", -8: "no-window", -9: "zlib-header", -10: "zlib-dict",
    -11: "adler", -12: "gzip-magic", -13: "gzip-method", -14: "gzip-flags", -15: "gzip-hcrc",
    -16: "crc", -17: "state", -18: "tree-overflow", -19: "replace-buffer", -20: "incomplete-stream",
    -21: "no-space", -22: "end-node",
}


class OracleError(Exception):
    def __init__(self, code, msg=""):
        super().__init__("oracle error %d (%s) %s" % (code, ERR_NAMES.get(code, "?"), msg))
        self.code = code


def build(force=False):
    so = os.path.join(_HERE, "libtbz_oracle.so")
    src = os.path.join(_HERE, "tbz_oracle.c")
    if force or not os.path.exists(so) or (
            os.path.exists(src) and os.path.getmtime(so) < os.path.getmtime(src)):
        subprocess.check_call(["make", "-C", _HERE, "libtbz_oracle.so"], stdout=subprocess.DEVNULL)
    return so


class _Ctx(C.Structure):
    _fields_ = [("vec", C.c_void_p), ("start", C.c_size_t), ("end", C.c_size_t), ("offset", C.c_size_t)]


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.tbzo_make_state.restype = C.c_void_p
        L.tbzo_make_state.argtypes = [C.c_int, C.c_void_p, C.c_size_t]
        L.tbzo_free_state.argtypes = [C.c_void_p]
        L.tbzo_decompress.restype = C.c_int64
        L.tbzo_decompress.argtypes = [C.POINTER(_Ctx), C.c_void_p]
        L.tbzo_replace_output_buffer.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        for f in ("tbzo_finished", "tbzo_input_underrun", "tbzo_output_overflow"):
            getattr(L, f).argtypes = [C.c_void_p]
        L.tbzo_output_offset.restype = C.c_int64
        L.tbzo_output_offset.argtypes = [C.c_void_p]
        L.tbzo_errmsg.restype = C.c_char_p
        L.tbzo_errmsg.argtypes = [C.c_void_p]
        L.tbzo_get_gzip_meta.argtypes = [C.c_void_p, C.c_void_p]
        L.tbzo_checksum.restype = C.c_uint32
        L.tbzo_checksum.argtypes = [C.c_void_p]
        L.tbzo_decompress_vector_into.restype = C.c_int64
        L.tbzo_decompress_vector_into.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p,
                                                  C.c_size_t]
        L.tbzo_decompress_vector.restype = C.c_int64
        L.tbzo_decompress_vector.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_int,
                                             C.POINTER(C.c_void_p)]
        L.tbzo_free.argtypes = [C.c_void_p]
        L.tbzo_adler32.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32),
                                   C.POINTER(C.c_uint32)]
        L.tbzo_crc32.restype = C.c_uint32
        L.tbzo_crc32.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32]
        L.tbzo_debug_build_trees.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_int),
                                             C.c_void_p, C.POINTER(C.c_int)]
        L.tbzo_set_fresh_tables.argtypes = [C.c_int]
        _LIB = L
    return _LIB


def set_fresh_tables(on):
    """test switch (not reference behaviour): an all-zero alphabet invalidates its table instead of leaving the
    previous block's entries in it — what the device path documents it does"""
    lib().tbzo_set_fresh_tables(1 if on else 0)


def _addr(buf):
    """address of a bytes / bytearray / numpy uint8 array without copying"""
    if buf is None:
        return None
    if isinstance(buf, bytes):
        return C.cast(C.c_char_p(buf), C.c_void_p).value
    if isinstance(buf, bytearray):
        return C.addressof((C.c_char * len(buf)).from_buffer(buf)) if len(buf) else None
    return buf.ctypes.data  # numpy


class OctetVectorContext:
    """make-octet-vector-context (io-common.lisp:40-45)"""

    def __init__(self, vector, start=0, offset=None, end=None):
        self.vector = vector
        self._c = _Ctx(_addr(vector), start, len(vector) if end is None else end,
                       start if offset is None else offset)

    @property
    def offset(self):
        return self._c.offset


def make_octet_vector_context(vector, start=0, offset=None, end=None):
    return OctetVectorContext(vector, start, offset, end)


class _GzipMeta(C.Structure):
    _fields_ = [("have_cm", C.c_int), ("have_mtime", C.c_int), ("have_os", C.c_int), ("have_extra", C.c_int),
                ("have_name", C.c_int), ("have_comment", C.c_int), ("flg", C.c_uint32), ("mtime", C.c_uint32),
                ("xfl", C.c_uint32), ("os", C.c_uint32), ("extra", C.c_void_p), ("name", C.c_void_p),
                ("comment", C.c_void_p), ("extra_len", C.c_size_t), ("name_len", C.c_size_t),
                ("comment_len", C.c_size_t)]


class State:
    """deflate-state / zlib-state / gzip-state (deflate.lisp:4-62, zlib.lisp:3-12, gzip.lisp:3-28)"""

    def __init__(self, fmt, output_buffer=None):
        self.fmt = fmt
        self.output_buffer = output_buffer
        n = len(output_buffer) if output_buffer is not None else 0
        self._p = lib().tbzo_make_state(fmt, _addr(output_buffer), n)

    def __del__(self):
        if getattr(self, "_p", None):
            lib().tbzo_free_state(self._p)
            self._p = None

    @property
    def output_offset(self):
        return lib().tbzo_output_offset(self._p)

    @property
    def checksum(self):
        return lib().tbzo_checksum(self._p)

    def gzip_meta(self):
        """the gzip-state's metadata slots as decompress-gzip has filled them so far, with the reference's values
        (gzip.lisp:123-241): keywords as strings, name / comment decoded as utf-8 with substitution (gzip.lisp:209-217: :errorp nil)"""
        m = _GzipMeta()
        lib().tbzo_get_gzip_meta(self._p, C.byref(m))

        def text(p, n):
            b = C.string_at(p, n) if n else b""
            # (babel:octets-to-string ... :encoding :utf-8 :errorp nil) substitutes and returns: gzip.lisp:214-217, :236-239;
            # the iso-8859-1 branch behind it is dead code.  The number of substitutes per invalid run is unpinned.
            return b.decode("utf-8", errors="replace")
        out = {"compression_method": None, "flags": None, "extra": None, "name": None, "comment": None,
               "operating_system": None, "mtime_unix": None, "mtime_universal": None, "compression_level": "default"}
        if m.have_cm:
            out["compression_method"] = "deflate"
            out["flags"] = [k for bit, k in ((4, "comment"), (3, "name"), (2, "extra"), (1, "header-crc"), (0, "text"))
                            if m.flg >> bit & 1]    # (push ...) in bit order 0..4: the list reads the other way round
        if m.have_mtime and m.mtime:
            out["mtime_unix"] = m.mtime
            out["mtime_universal"] = m.mtime + 2208988800   # (encode-universal-time 0 0 0 1 1 1970 0)
        if m.have_os:
            out["compression_level"] = {2: "maximum", 4: "fastest"}.get(m.xfl, m.xfl)
            names = ("fat", "amiga", "vms", "unix", "vm/cms", "atari-tos", "hpfs", "macintosh", "z-system", "cp/m",
                     "tops-20", "ntfs", "qdos", "acorn-riscos")
            out["operating_system"] = names[m.os] if m.os <= 13 else ("unknown", m.os)
        if m.have_extra:
            out["extra"] = C.string_at(m.extra, m.extra_len) if m.extra_len else b""
        if m.have_name:
            out["name"] = text(m.name, m.name_len)
        if m.have_comment:
            out["comment"] = text(m.comment, m.comment_len)
        return out


def make_deflate_state(output_buffer=None):
    return State(DEFLATE, output_buffer)


def make_zlib_state(output_buffer=None):
    return State(ZLIB, output_buffer)


def make_gzip_state(output_buffer=None):
    return State(GZIP, output_buffer)


def decompress(context, state):
    """api.lisp:3-10"""
    r = lib().tbzo_decompress(C.byref(context._c), state._p)
    if r < 0:
        raise OracleError(r, lib().tbzo_errmsg(state._p).decode())
    return r


def replace_output_buffer(state, buffer):
    """api.lisp:12-21"""
    r = lib().tbzo_replace_output_buffer(state._p, _addr(buffer), len(buffer))
    if r < 0:
        raise OracleError(r, lib().tbzo_errmsg(state._p).decode())
    state.output_buffer = buffer


def finished(state):
    return bool(lib().tbzo_finished(state._p))


def input_underrun(state):
    return bool(lib().tbzo_input_underrun(state._p))


def output_overflow(state):
    return bool(lib().tbzo_output_overflow(state._p))


def decompress_vector(compressed, format="zlib", start=0, end=None, output=None):
    """api.lisp:23-65 — returns (buffer, count)"""
    fmt = FORMATS[format] if isinstance(format, str) else format
    end = len(compressed) if end is None else end
    if output is not None:
        r = lib().tbzo_decompress_vector_into(_addr(compressed), start, end, fmt, _addr(output), len(output))
        if r < 0:
            raise OracleError(r)
        return output, r
    p = C.c_void_p()
    r = lib().tbzo_decompress_vector(_addr(compressed), start, end, fmt, C.byref(p))
    if r < 0:
        raise OracleError(r)
    out = C.string_at(p.value, r) if r else b""
    lib().tbzo_free(p)
    return out, r


def adler32(buf, s1=1, s2=0, end=None):
    """checksums.lisp:167-174 — returns (s1, s2)"""
    o1, o2 = C.c_uint32(), C.c_uint32()
    lib().tbzo_adler32(_addr(buf), len(buf) if end is None else end, s1, s2, C.byref(o1), C.byref(o2))
    return o1.value, o2.value


def crc32(buf, crc=0, end=None):
    """checksums.lisp:196-210"""
    return lib().tbzo_crc32(_addr(buf), len(buf) if end is None else end, crc)
