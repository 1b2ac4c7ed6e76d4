"""ctypes binding of lib3bz_amd.so (the C ABI of include/tbz_amd.h).

The product path is the HIP library and nothing else: if it is missing or cannot be loaded this
module raises — there is no CPU fallback of any kind.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_NAME = "lib3bz_amd.so"

FORMATS = {"deflate": 0, "zlib": 1, "gzip": 2}
FINISHED, INPUT_UNDERRUN, OUTPUT_OVERFLOW = 0, 1, 2


class Result(C.Structure):
    """struct tbz_result"""
    _fields_ = [("status", C.c_int32), ("segments", C.c_uint32), ("out_len", C.c_uint64),
                ("out_total", C.c_uint64), ("in_consumed", C.c_uint64), ("adler32", C.c_uint32),
                ("crc32", C.c_uint32), ("trailer_check", C.c_uint32), ("trailer_isize", C.c_uint32),
                ("flags", C.c_uint32), ("reserved", C.c_uint32), ("boundary_out", C.c_uint64)]


class GzipHeader(C.Structure):
    """struct tbz_gzip_header"""
    _fields_ = [("status", C.c_int32), ("header_len", C.c_uint32), ("cm", C.c_uint32), ("flg", C.c_uint32),
                ("mtime", C.c_uint32), ("xfl", C.c_uint32), ("os", C.c_uint32), ("extra_off", C.c_uint32),
                ("extra_len", C.c_uint32), ("name_off", C.c_uint32), ("name_len", C.c_uint32),
                ("comment_off", C.c_uint32), ("comment_len", C.c_uint32), ("hcrc_present", C.c_uint32),
                ("hcrc", C.c_uint32), ("stage", C.c_uint32)]


class Timings(C.Structure):
    """struct tbz_timings"""
    _fields_ = [("scan_ms", C.c_float), ("huff_ms", C.c_float), ("lz_ms", C.c_float), ("cksum_ms", C.c_float),
                ("total_ms", C.c_float), ("huff_launches", C.c_uint32), ("fixup_rounds", C.c_uint32),
                ("token_words", C.c_uint64), ("n_segments", C.c_uint64), ("n_groups", C.c_uint64),
                ("find_ms", C.c_float), ("resolve_ms", C.c_float), ("k1_gang", C.c_uint32), ("k2_kinds", C.c_uint32),
                ("n_candidates", C.c_uint64), ("n_hgroups", C.c_uint64), ("scratch_bytes", C.c_uint64),
                ("h2d_copies", C.c_uint32), ("passes", C.c_uint32),
                ("h2d_ms", C.c_float), ("host_decode_ms", C.c_float), ("d2h_ms", C.c_float), ("reserved4", C.c_uint32)]


assert C.sizeof(Result) == 64
ALLOC_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_size_t)   # tbz_alloc_fn

# every symbol include/tbz_amd.h declares
SYMBOLS = [
    "tbz_ctx_create", "tbz_ctx_destroy", "tbz_ctx_trim", "tbz_abi_version", "tbz_strerror", "tbz_last_error",
    "tbz_device_count", "tbz_inflate", "tbz_inflate_size", "tbz_inflate_alloc", "tbz_inflate_batch", "tbz_inflate_device",
    "tbz_inflate_batch_device", "tbz_adler32_device", "tbz_crc32_device", "tbz_device_malloc",
    "tbz_device_free", "tbz_memcpy_h2d", "tbz_memcpy_d2h", "tbz_last_timings",
    "tbz_session_create", "tbz_session_destroy", "tbz_session_feed", "tbz_session_decompress", "tbz_session_stats",
    "tbz_gzip_header_parse", "tbz_inflate_gzip_members", "tbz_inflate_gzip_members_device",
    "tbz_inflate_to_device", "tbz_assign_streams", "tbz_inflate_batch_multi", "tbz_inflate_batch_multi_device",
    "tbz_inflate_sharded_plan", "tbz_inflate_sharded_verdict", "tbz_inflate_sharded_multi",
]


class LibraryMissing(RuntimeError):
    pass


def default_path():
    return os.path.join(_HERE, LIB_NAME)


def load(path=None):
    """Load the library and declare prototypes.  `path` is only ever passed by tests (the CPU
    lane-emulation build of the same sources); the product always loads lib3bz_amd.so."""
    path = path or default_path()
    if not os.path.exists(path):
        raise LibraryMissing(
            "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % path)
    L = C.CDLL(path)
    vp, sz, u64p = C.c_void_p, C.c_size_t, C.POINTER(C.c_uint64)
    L.tbz_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.tbz_ctx_destroy.argtypes = [vp]
    L.tbz_ctx_destroy.restype = None
    L.tbz_ctx_trim.argtypes = [vp]
    L.tbz_strerror.restype = C.c_char_p
    L.tbz_strerror.argtypes = [C.c_int]
    L.tbz_last_error.restype = C.c_char_p
    L.tbz_last_error.argtypes = [vp]
    L.tbz_inflate.argtypes = [vp, C.c_int, vp, sz, vp, sz, C.POINTER(Result)]
    L.tbz_inflate_size.argtypes = [vp, C.c_int, vp, sz, C.POINTER(Result)]
    L.tbz_inflate_alloc.argtypes = [vp, C.c_int, vp, sz, ALLOC_FN, vp, C.POINTER(Result)]
    L.tbz_inflate_batch.argtypes = [vp, C.c_int, sz, C.POINTER(vp), C.POINTER(sz), C.POINTER(vp), C.POINTER(sz),
                                    C.POINTER(Result)]
    L.tbz_inflate_device.argtypes = [vp, C.c_int, vp, sz, vp, sz, C.POINTER(Result)]
    L.tbz_inflate_batch_device.argtypes = [vp, C.c_int, sz, vp, u64p, u64p, vp, u64p, u64p, C.POINTER(Result)]
    L.tbz_adler32_device.argtypes = [vp, vp, sz, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32),
                                     C.POINTER(C.c_uint32)]
    L.tbz_crc32_device.argtypes = [vp, vp, sz, C.c_uint32, C.POINTER(C.c_uint32)]
    L.tbz_device_malloc.argtypes = [vp, sz, C.POINTER(vp)]
    L.tbz_device_free.argtypes = [vp, vp]
    L.tbz_memcpy_h2d.argtypes = [vp, vp, vp, sz]
    L.tbz_memcpy_d2h.argtypes = [vp, vp, vp, sz]
    L.tbz_last_timings.argtypes = [vp, C.POINTER(Timings)]
    L.tbz_gzip_header_parse.argtypes = [vp, sz, C.POINTER(GzipHeader)]
    L.tbz_session_create.argtypes = [vp, C.c_int, C.POINTER(vp)]
    L.tbz_session_destroy.argtypes = [vp]
    L.tbz_session_destroy.restype = None
    L.tbz_session_feed.argtypes = [vp, vp, sz, C.c_int]
    L.tbz_session_decompress.argtypes = [vp, vp, sz, C.POINTER(Result)]
    L.tbz_session_stats.argtypes = [vp, u64p, u64p]
    L.tbz_inflate_gzip_members_device.argtypes = [vp, vp, sz, vp, sz, sz, C.POINTER(Result), u64p, u64p, C.POINTER(sz)]
    L.tbz_inflate_gzip_members.argtypes = [vp, vp, sz, ALLOC_FN, vp, sz, C.POINTER(Result), u64p, C.POINTER(sz)]
    L.tbz_inflate_to_device.argtypes = [vp, C.c_int, vp, sz, C.POINTER(vp), C.POINTER(Result)]
    L.tbz_assign_streams.argtypes = [C.POINTER(sz), sz, sz, C.POINTER(C.c_uint32)]
    L.tbz_inflate_batch_multi.argtypes = [C.POINTER(vp), sz, C.c_int, sz, C.POINTER(vp), C.POINTER(sz), C.POINTER(vp), C.POINTER(sz),
                                          C.POINTER(Result)]
    u64pp, vpp = C.POINTER(u64p), C.POINTER(vp)
    L.tbz_inflate_batch_multi_device.argtypes = [C.POINTER(vp), sz, C.c_int, C.POINTER(sz), vpp, u64pp, u64pp, vpp, u64pp, u64pp,
                                                 C.POINTER(C.POINTER(Result))]
    L.tbz_inflate_sharded_plan.argtypes = [vp, sz, sz, u64p]
    L.tbz_inflate_sharded_verdict.argtypes = [C.c_int, vp, sz, sz, u64p, C.POINTER(Result), C.POINTER(C.c_uint32), u64p, u64p,
                                              C.POINTER(C.c_uint32), u64p, C.POINTER(C.c_int)]
    L.tbz_inflate_sharded_multi.argtypes = [C.POINTER(vp), sz, C.c_int, vp, sz, vp, sz, C.POINTER(Result), C.POINTER(C.c_int)]
    for s in SYMBOLS:
        getattr(L, s)  # AttributeError if the ABI is incomplete
    return L
