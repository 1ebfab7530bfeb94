"""ctypes binding of the C ABI in include/ipx.h (libipx.so, built by __graft_entry__.build()).

The shared library is the product; this module only declares its entry points. There is
no CPU fallback: if the library is missing, or no HIP device is visible, the calls raise.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# IPX_LIB names another build of the same sources next to this file (scripts/: IPX_LIB=libipx_diag.so, the -DIPX_DIAG
# build with the kernel variants the shipped library does not carry)
LIB_PATH = os.path.join(_HERE, os.environ.get("IPX_LIB") or "libipx.so")

#: every symbol include/ipx.h declares (tests check the library exports all of them)
SYMBOLS = [
    "ipx_last_error", "ipx_version", "ipx_device_count", "ipx_create", "ipx_destroy", "ipx_dim",
    "ipx_ntotal", "ipx_reserve", "ipx_set_id_base", "ipx_add", "ipx_search", "ipx_search_noalloc", "ipx_search_aspect", "ipx_search_pooled", "ipx_pool", "ipx_add_pooled",
    "ipx_merge_topk", "ipx_merge_topk_packed", "ipx_packed_block_bytes", "ipx_set_param", "ipx_read_trace", "ipx_check", "ipx_set_timing", "ipx_last_timing", "ipx_last_geometry",
    "ipx_last_bf16_stats", "ipx_outlier_rows", "ipx_last_exact_queries",
    "ipx_write_ranking", "ipx_write_ranking_lookup", "ipx_format_score",
]

IPX_F32, IPX_F16, IPX_BF16 = 0, 1, 2
IPX_MAX_K = 2048            # depth of one pass (include/ipx.h)
IPX_MAX_DEPTH = 1 << 20     # depth a call accepts

_lib = None


class IpxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"ipx error {code}: {msg}")
        self.code = code


def lib():
    """Load libipx.so once and declare argument types."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback for the search path.")
    # torch ships its own HIP runtime; whichever runtime is loaded first serves the whole process.
    # Load torch's first, so tensors handed to the library and the library's own calls share it
    # (loaded the other way round, torch finds no GPU afterwards).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, f32p, i64p = C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p
    L.ipx_last_error.restype = C.c_char_p
    L.ipx_version.restype = C.c_char_p
    L.ipx_device_count.restype = i32
    L.ipx_create.argtypes = [i32, i32, C.POINTER(vp)]
    L.ipx_destroy.argtypes = [vp]
    L.ipx_dim.argtypes = [vp]
    L.ipx_ntotal.argtypes = [vp]
    L.ipx_ntotal.restype = i64
    L.ipx_reserve.argtypes = [vp, i64]
    L.ipx_set_id_base.argtypes = [vp, i64]
    L.ipx_add.argtypes = [vp, vp, i64, i32, i32, vp]
    L.ipx_search.argtypes = [vp, f32p, i64, i32, f32p, i64p, i32, vp]
    L.ipx_search_noalloc.argtypes = [vp, f32p, i64, i32, f32p, i64p, vp]
    L.ipx_search_aspect.argtypes = [vp, f32p, f32p, i32, i32, i64, i32, f32p, i64p, i32, vp]
    L.ipx_search_pooled.argtypes = [vp, f32p, i64, f32p, f32p, i32, i32, i64, i32, f32p, i64p, f32p, i32, vp]
    L.ipx_pool.argtypes = [vp, f32p, i64, f32p, f32p, i32, i32, i64, f32p, f32p, i32, vp]
    L.ipx_add_pooled.argtypes = [vp, f32p, i64, f32p, f32p, i32, i32, i64, i32, vp]
    L.ipx_merge_topk.argtypes = [f32p, i64p, i32, i64, i32, i32, f32p, i64p, i32, i32, vp]
    L.ipx_merge_topk_packed.argtypes = [vp, i32, i64, i32, i32, f32p, i64p, i32, vp]
    L.ipx_packed_block_bytes.argtypes = [i64, i32]
    L.ipx_packed_block_bytes.restype = C.c_size_t
    L.ipx_set_param.argtypes = [vp, C.c_char_p, i64]
    L.ipx_read_trace.argtypes = [vp, vp, i32]
    L.ipx_check.argtypes = [vp, vp]
    L.ipx_set_timing.argtypes = [vp, i32]
    L.ipx_last_timing.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.ipx_last_geometry.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
    L.ipx_last_bf16_stats.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(i32), C.POINTER(i64)]
    L.ipx_outlier_rows.argtypes = [vp]
    L.ipx_last_exact_queries.argtypes = [vp]
    L.ipx_last_exact_queries.restype = i64
    L.ipx_write_ranking.argtypes = [C.c_char_p, vp, i32, vp, i32, vp, vp, i64, i32]
    L.ipx_write_ranking_lookup.argtypes = [C.c_char_p, vp, i32, vp, vp, i32, i64, vp, vp, i64, i32]
    L.ipx_format_score.argtypes = [C.c_float, C.c_char_p, C.c_size_t]
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise IpxError(rc, lib().ipx_last_error().decode("utf-8", "replace"))
