"""ctypes binding of libpt_hip.so (C ABI: include/pt_api.h).

This is the only way the Python host side reaches the GPU path; there is no fallback.  If the
library is missing the import of this module's `lib()` raises -- loudly, by design.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PT_HIP_LIB") or os.path.join(_HERE, "libpt_hip.so")   # PT_HIP_LIB: profiling/ablation builds only

NOIDX = 0xFFFFFFFF
MAX_K = 32
F32, F16, F64 = 0, 1, 2
DIST_UNIFORM, DIST_CLUSTERED = 0, 1
BLEND_MEAN, BLEND_INV_D2 = 0, 1
OK, ERR_ARG, ERR_HIP, ERR_STATE, ERR_NOMEM, ERR_UNSUPPORTED = 0, -1, -2, -3, -4, -5

# every symbol include/pt_api.h declares (tests check the .so exports all of them)
SYMBOLS = [
    "pt_ctx_create", "pt_ctx_destroy", "pt_set_stream", "pt_set_param", "pt_last_error", "pt_stats", "pt_synchronize",
    "pt_build_aos", "pt_build_soa", "pt_build_soa_indexed", "pt_set_attributes", "pt_set_attributes_range", "pt_set_attributes_local", "pt_build_synth", "pt_rebuild",
    "pt_num_source", "pt_query_aos", "pt_query_soa", "pt_targets_synth", "pt_targets_soa", "pt_targets_aos", "pt_num_targets", "pt_query_resident", "pt_query_blend_resident", "pt_query_resident_host",
    "pt_resident_target_ids", "pt_resident_target_xyz", "pt_resident_source_xyz", "pt_blend", "pt_blend_dev", "pt_blend_weighted", "pt_blend_weighted_dev", "pt_pca_normals",
    "pt_pca_normals_dev", "pt_merge_candidates_dev", "pt_slab_need_dev", "pt_pack_requests_dev", "pt_query_bounded_dev",
    "pt_bake_texture", "pt_texture_pad", "pt_host_alloc", "pt_host_free", "pt_upload_begin", "pt_upload_range", "pt_upload_end", "pt_stream_query",
    "pt_comm_unique_id", "pt_comm_init", "pt_comm_destroy", "pt_comm_abort", "pt_exchange_merge_dev", "pt_exchange_merge_local", "pt_query_exchange_blend",
]


class Stats(C.Structure):
    _fields_ = [
        ("n_source", C.c_uint64), ("n_target", C.c_uint64), ("k", C.c_int32), ("_pad", C.c_int32),
        ("ms_build", C.c_double), ("ms_sort_targets", C.c_double), ("ms_query", C.c_double),
        ("ms_blend", C.c_double), ("ms_pca", C.c_double),
        ("bytes_alg_build", C.c_uint64), ("bytes_alg_query", C.c_uint64),
        ("grid_dim", C.c_int32 * 3), ("n_levels", C.c_int32),
        ("cell_size", C.c_double), ("n_cells", C.c_uint64), ("device_bytes", C.c_uint64),
        ("ms_kernel", C.c_double * 8), ("n_leftover", C.c_uint64), ("rho_occupied", C.c_double),
        ("n_refine", C.c_int32), ("bbox_guess", C.c_int32), ("ms_bake", C.c_double),
        ("n_nodes", C.c_uint32), ("refine_levels", C.c_int32), ("max_cell_points", C.c_uint32), ("n_wave", C.c_uint32),
        ("pass1_pooled", C.c_int32), ("stream_skipped", C.c_int32), ("stream_revisited", C.c_int32), ("pass2_pooled", C.c_int32),
        ("uniform_probe", C.c_int32), ("dup_leaves", C.c_int32), ("presort_refine", C.c_int32), ("n_sorts", C.c_int32), ("ordered_input", C.c_int32),
    ]


class ExchangeStats(C.Structure):
    _fields_ = [("crossing", C.c_uint64), ("answered", C.c_uint64), ("bytes_sent", C.c_uint64), ("bytes_received", C.c_uint64), ("ms", C.c_double)]


class PtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libpt_hip error %d: %s" % (code, msg))
        self.code = code


_lib = None


def lib():
    """Load libpt_hip.so.  `import torch` first if torch is used in the same process, so that both
    share one HIP runtime (both link libamdhip64.so.7)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libpt_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C 3d-reconstruction-from-point-cloud_amd/csrc`. There is no CPU fallback." % LIB_PATH)
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    p, u64, i32, dbl = C.c_void_p, C.c_uint64, C.c_int, C.c_double
    sig = {
        "pt_ctx_create": (i32, [C.POINTER(p), C.POINTER(C.c_int), i32]),
        "pt_ctx_destroy": (None, [p]),
        "pt_set_stream": (i32, [p, p]),
        "pt_set_param": (i32, [p, C.c_char_p, dbl]),
        "pt_last_error": (C.c_char_p, [p]),
        "pt_stats": (i32, [p, C.POINTER(Stats)]),
        "pt_synchronize": (i32, [p]),
        "pt_build_aos": (i32, [p, p, u64]),
        "pt_build_soa": (i32, [p, p, i32, p, p, u64, i32]),
        "pt_build_soa_indexed": (i32, [p, p, i32, p, u64, i32]),
        "pt_set_attributes": (i32, [p, p, p, u64, i32]),
        "pt_set_attributes_range": (i32, [p, u64, u64, p, p, u64]),
        "pt_set_attributes_local": (i32, [p, p, p, i32]),
        "pt_build_synth": (i32, [p, u64, u64, i32, i32, i32, dbl, dbl]),
        "pt_rebuild": (i32, [p]),
        "pt_num_source": (u64, [p]),
        "pt_query_aos": (i32, [p, p, u64, i32, p, p]),
        "pt_query_soa": (i32, [p, p, i32, u64, i32, i32, p, p]),
        "pt_targets_synth": (i32, [p, u64, u64, i32, i32, i32, dbl, dbl]),
        "pt_num_targets": (u64, [p]),
        "pt_targets_soa": (i32, [p, p, i32, u64, i32]),
        "pt_targets_aos": (i32, [p, p, u64]),
        "pt_query_resident": (i32, [p, i32, p, p]),
        "pt_query_blend_resident": (i32, [p, i32, i32, p, p, p, p]),
        "pt_query_resident_host": (i32, [p, i32, i32, p, p, p, p]),
        "pt_resident_target_ids": (i32, [p, p]),
        "pt_resident_target_xyz": (i32, [p, p]),
        "pt_resident_source_xyz": (i32, [p, p, C.POINTER(C.c_int)]),
        "pt_blend": (i32, [p, p, p, u64, i32, i32, p, p]),
        "pt_blend_dev": (i32, [p, p, p, u64, i32, i32, p, p]),
        "pt_blend_weighted": (i32, [p, p, p, u64, i32, p, p]),
        "pt_blend_weighted_dev": (i32, [p, p, p, u64, i32, p, p]),
        "pt_pca_normals": (i32, [p, p, u64, i32, p]),
        "pt_pca_normals_dev": (i32, [p, p, u64, i32, p]),
        "pt_merge_candidates_dev": (i32, [p, p, p, i32, u64, i32, p, p]),
        "pt_slab_need_dev": (i32, [p, p, i32, p, u64, i32, i32, p, i32, i32, p]),
        "pt_pack_requests_dev": (i32, [p, p, i32, p, u64, i32, i32, p, i32, i32, p, p, p]),
        "pt_query_bounded_dev": (i32, [p, p, i32, p, u64, i32, p, p]),
        "pt_bake_texture": (i32, [p, p, u64, p, u64, p, i32, i32, i32, p]),
        "pt_texture_pad": (i32, [p, p, i32, i32, p]),
        "pt_host_alloc": (p, [u64]),
        "pt_host_free": (None, [p]),
        "pt_upload_begin": (i32, [p, u64, i32, i32]),
        "pt_upload_range": (i32, [p, u64, u64, p, p, p, p, p]),
        "pt_upload_end": (i32, [p]),
        "pt_stream_query": (i32, [p, p, i32, u64, u64, u64, i32, p, p]),
        "pt_comm_unique_id": (i32, [p]),
        "pt_comm_init": (i32, [p, i32, i32, p]),
        "pt_comm_destroy": (i32, [p]),
        "pt_comm_abort": (i32, [p]),
        "pt_exchange_merge_dev": (i32, [p, p, i32, u64, i32, i32, p, p, p, i32, p, p, p]),
        "pt_exchange_merge_local": (i32, [p, i32, p, i32, p, i32, i32, p, p, p, i32, p, p]),
        "pt_query_exchange_blend": (i32, [p, p, i32, u64, i32, i32, p, i32, p, p, p, p, p]),
    }
    assert sorted(sig) == sorted(SYMBOLS)
    for name, (res, args) in sig.items():
        f = getattr(L, name)      # AttributeError if the .so lacks a declared symbol
        f.restype = res
        f.argtypes = args
    _lib = L
    return L
