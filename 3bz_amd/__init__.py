"""3bz_amd — MI355X-native inflate engine behind 3bz's octet-vector API.

`import 3bz_amd` is not valid Python syntax (the name starts with a digit); load it with
    importlib.import_module("3bz_amd")
The package holds only what the hot path needs: csrc/ (HIP kernels + host engine + C ABI),
the ctypes binding and the host-side mirror of the reference's API.
"""
from .api import (Engine, EngineError, ThreeBzError, decompress, decompress_gzip_members, decompress_vector,  # noqa: F401
                  default_engine,
                  finished, input_underrun, make_deflate_state, make_gzip_state, make_octet_pointer_context,
                  make_octet_stream_context, make_octet_vector_context, resync_file_stream, valid_octet_pointer,
                  valid_octet_stream, with_octet_pointer,
                  make_zlib_state, output_overflow, replace_output_buffer, set_default_engine)
from ._lib import FORMATS, Result, Timings  # noqa: F401
