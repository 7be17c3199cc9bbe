"""ctypes binding of libpgv_hip.so (include/pgv_hip.h).

The shared library is the product; this module only declares its prototypes so
the Python harness (tests/, bench.py, __graft_entry__.py) can call through the
same C ABI a Postgres backend would.  There is no fallback: if the library is
missing, import fails loudly; if there is no GPU, every call returns
PGV_ERR_DEVICE and is raised as PgvError.
"""
import ctypes as C
import os

# The Python harness keeps data in HBM through PyTorch, whose wheel bundles its
# own HIP runtime (same soname, libamdhip64.so.7).  Two HIP runtimes cannot
# share one process, so when torch is installed it is imported FIRST and the
# dynamic loader then binds libpgv_hip.so to that already-loaded runtime.  A
# Postgres backend has no torch: there the library binds to /opt/rocm's
# runtime through its RUNPATH.  Nothing else from torch is used here.
try:
    import torch  # noqa: F401
except ImportError:  # pragma: no cover
    torch = None

_HERE = os.path.dirname(os.path.abspath(__file__))
# PGV_HIP_LIB points at another build of the same library (kernel experiments: tools/ablate_tile.sh)
LIB_PATH = os.environ.get("PGV_HIP_LIB") or os.path.join(_HERE, "lib", "libpgv_hip.so")

PGV_OK, PGV_ERR_ARG, PGV_ERR_DIMS, PGV_ERR_DEVICE, PGV_ERR_NOMEM, PGV_ERR_STATE, PGV_ERR_DATA = range(7)
PGV_F32, PGV_F16 = 0, 1
PGV_L2SQ, PGV_NEG_IP, PGV_L1 = 0, 1, 2
PGV_OPS_L2, PGV_OPS_IP, PGV_OPS_COSINE = 0, 1, 2
PGV_BOUND_STATISTICAL, PGV_BOUND_WORST_CASE = 0, 1

# every symbol include/pgv_hip.h declares (tests check the library exports each)
SYMBOLS = [
    "pgv_last_error", "pgv_abi_version", "pgv_device_count", "pgv_ctx_create", "pgv_ctx_destroy",
    "pgv_ctx_sync", "pgv_ctx_stream", "pgv_timer_start", "pgv_timer_stop", "pgv_ctx_set_profiling", "pgv_ctx_set_exact_scan", "pgv_index_share", "pgv_pinned_alloc", "pgv_pinned_free",
    "pgv_ctx_reset_stats", "pgv_ctx_get_stats", "pgv_index_upload", "pgv_index_free",
    "pgv_index_rows", "pgv_index_lists", "pgv_rank_lists", "pgv_scan_lists", "pgv_search_batch", "pgv_scan_batch",
    "pgv_assign", "pgv_kmeans", "pgv_lloyd_partial", "pgv_lloyd_finish", "pgv_kmeanspp_init",
    "pgv_distance_batch", "pgv_cosine_distance_batch", "pgv_bit_distance_batch", "pgv_hnsw_upload", "pgv_hnsw_free", "pgv_hnsw_score", "pgv_hnsw_set_graph", "pgv_hnsw_search",
    "pgv_hnsw_build_search", "pgv_hnsw_build_neighbors", "pgv_hnsw_link_begin", "pgv_hnsw_link_prepare",
    "pgv_hnsw_link_apply", "pgv_hnsw_link_end", "pgv_hnsw_build_search_keep", "pgv_hnsw_build_select_kept", "pgv_hnsw_score_pairs", "pgv_hnsw_score_groups", "pgv_hnsw_update_graph",
    "pgv_query_begin", "pgv_query_end", "pgv_query_rank", "pgv_query_scan", "pgv_query_more", "pgv_query_lists",
    "pgv_comm_unique_id", "pgv_comm_create", "pgv_comm_create_custom", "pgv_comm_destroy", "pgv_comm_size",
    "pgv_comm_rank", "pgv_kmeans_sharded", "pgv_search_batch_sharded",
    "pgv_device_memory", "pgv_pinned_register", "pgv_pinned_unregister", "pgv_index_export", "pgv_index_import",
    "pgv_index_tids", "pgv_hnsw_export", "pgv_hnsw_import", "pgv_hnsw_share", "pgv_hnsw_device", "pgv_exact_topk", "pgv_ctx_set_bound",
    "pgv_hnsw_upload_payload", "pgv_hnsw_get_payload", "pgv_builder_begin", "pgv_builder_add", "pgv_builder_set_centers", "pgv_builder_rows", "pgv_builder_finish", "pgv_builder_free", "pgv_index_drain",
    "pgv_index_set_overlap",
]


class PgvError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("libpgv_hip error %d: %s" % (code, message))
        self.code = code
        self.message = message


class PgvStats(C.Structure):
    _fields_ = [("scan_ms", C.c_double), ("scan_launches", C.c_int64),
                ("scan_pairs", C.c_double), ("scan_rows", C.c_double),
                ("aux_ms", C.c_double), ("aux_launches", C.c_int64), ("aux_pairs", C.c_double),
                ("assign_redo_rows", C.c_double), ("assign_rows", C.c_double), ("assign_recheck_rows", C.c_double), ("scan_unique_rows", C.c_double), ("scan_redo_queries", C.c_double), ("scan_widened_queries", C.c_double)]


ALL_REDUCE_F32 = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
ALL_GATHER = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)


class PgvCollectives(C.Structure):
    _fields_ = [("all_reduce_sum_f32", ALL_REDUCE_F32), ("all_gather", ALL_GATHER), ("state", C.c_void_p)]


NEXT_DOUBLE = C.CFUNCTYPE(C.c_double, C.c_void_p)
NEXT_U32 = C.CFUNCTYPE(C.c_uint32, C.c_void_p)


class PgvRng(C.Structure):
    _fields_ = [("next_double", C.c_void_p), ("next_u32", C.c_void_p),
                ("state", C.c_void_p), ("seed", C.c_uint64)]


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libpgv_hip.so not built (%s): run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C pgvector_amd/csrc` -- there is no CPU fallback" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    P, I, I64 = C.c_void_p, C.c_int, C.c_int64
    lib.pgv_last_error.restype = C.c_char_p
    lib.pgv_last_error.argtypes = []
    lib.pgv_abi_version.restype = I
    lib.pgv_device_count.restype = I
    lib.pgv_ctx_create.argtypes = [I, P, C.POINTER(P)]
    lib.pgv_ctx_destroy.argtypes = [P]
    lib.pgv_ctx_destroy.restype = None
    lib.pgv_ctx_sync.argtypes = [P]
    lib.pgv_ctx_stream.argtypes = [P]
    lib.pgv_ctx_stream.restype = P
    lib.pgv_timer_start.argtypes = [P]
    lib.pgv_timer_stop.argtypes = [P, C.POINTER(C.c_float)]
    lib.pgv_ctx_set_profiling.argtypes = [P, I]
    lib.pgv_ctx_set_exact_scan.argtypes = [P, I]
    lib.pgv_ctx_reset_stats.argtypes = [P]
    lib.pgv_ctx_get_stats.argtypes = [P, C.POINTER(PgvStats)]
    lib.pgv_index_upload.argtypes = [P, I, I, I, I, P, P, P, P, C.POINTER(P)]
    lib.pgv_index_free.argtypes = [P]
    lib.pgv_index_share.argtypes = [P, P, C.POINTER(P)]
    lib.pgv_index_set_overlap.argtypes = [P, I]
    lib.pgv_index_free.restype = None
    lib.pgv_index_export.argtypes = [P, P]
    lib.pgv_index_import.argtypes = [P, P, C.POINTER(P)]
    lib.pgv_index_tids.argtypes = [P, P, I64, P]
    lib.pgv_hnsw_export.argtypes = [P, P]
    lib.pgv_hnsw_share.argtypes = [P, P, C.POINTER(P)]
    lib.pgv_hnsw_device.argtypes = [P]
    lib.pgv_hnsw_upload_payload.argtypes = [P, I, I, I, P, I64, P, I, C.POINTER(P)]
    lib.pgv_hnsw_get_payload.argtypes = [P, P, I, P]
    lib.pgv_exact_topk.argtypes = [P, I, I, I, P, I, P, I64, I, P, P]
    lib.pgv_ctx_set_bound.argtypes = [P, I]
    lib.pgv_builder_begin.argtypes = [P, I, I, I, I, P, I64, C.POINTER(P)]
    lib.pgv_builder_add.argtypes = [P, P, P, I64]
    lib.pgv_builder_set_centers.argtypes = [P, P]
    lib.pgv_builder_rows.argtypes = [P]
    lib.pgv_builder_rows.restype = I64
    lib.pgv_builder_finish.argtypes = [P, C.POINTER(P), P, P]
    lib.pgv_builder_free.argtypes = [P]
    lib.pgv_builder_free.restype = None
    lib.pgv_index_drain.argtypes = [P, I64, P, P]
    lib.pgv_hnsw_import.argtypes = [P, P, C.POINTER(P)]
    lib.pgv_device_memory.argtypes = [I, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.pgv_pinned_register.argtypes = [P, C.c_size_t]
    lib.pgv_pinned_unregister.argtypes = [P]
    lib.pgv_pinned_unregister.restype = None
    lib.pgv_index_rows.argtypes = [P]
    lib.pgv_index_rows.restype = I64
    lib.pgv_index_lists.argtypes = [P]
    lib.pgv_rank_lists.argtypes = [P, P, I, I, P, P]
    lib.pgv_scan_lists.argtypes = [P, P, P, I, P, P, I64, C.POINTER(I64)]
    lib.pgv_search_batch.argtypes = [P, P, I, I, I, P, P, P]
    lib.pgv_scan_batch.argtypes = [P, P, I, P, I, I, P, P, P]
    lib.pgv_query_begin.argtypes = [P, C.POINTER(P)]
    lib.pgv_query_end.argtypes = [P]
    lib.pgv_query_end.restype = None
    lib.pgv_query_rank.argtypes = [P, P, I]
    lib.pgv_query_scan.argtypes = [P, I, I, I, P, P, P, C.POINTER(I), C.POINTER(I64)]
    lib.pgv_query_more.argtypes = [P, I, I, P, P, P, C.POINTER(I)]
    lib.pgv_query_lists.argtypes = [P, P, I]
    lib.pgv_comm_unique_id.argtypes = [P]
    lib.pgv_comm_create.argtypes = [P, I, I, P, C.POINTER(P)]
    lib.pgv_comm_create_custom.argtypes = [P, I, I, C.POINTER(PgvCollectives), C.POINTER(P)]
    lib.pgv_comm_destroy.argtypes = [P]
    lib.pgv_comm_destroy.restype = None
    lib.pgv_comm_size.argtypes = [P]
    lib.pgv_comm_rank.argtypes = [P]
    lib.pgv_kmeans_sharded.argtypes = [P, I, I, I, P, I, I, I, C.POINTER(PgvRng), P, P, C.POINTER(I)]
    lib.pgv_search_batch_sharded.argtypes = [P, P, P, I, I, I, P, P]
    lib.pgv_assign.argtypes = [P, I, I, I, P, I, P, I64, P, P]
    lib.pgv_kmeans.argtypes = [P, I, I, I, P, I, I, I, C.POINTER(PgvRng), P, P, C.POINTER(I)]
    lib.pgv_lloyd_partial.argtypes = [P, I, I, I, P, I, P, I, P, P, P, P]
    lib.pgv_lloyd_finish.argtypes = [P, I, I, I, I, P, P, C.POINTER(PgvRng), P]
    lib.pgv_kmeanspp_init.argtypes = [P, I, I, I, P, I, I, C.POINTER(PgvRng), P]
    lib.pgv_distance_batch.argtypes = [P, I, I, I, P, P, I64, P]
    lib.pgv_hnsw_upload.argtypes = [P, I, I, I, P, I64, C.POINTER(P)]
    lib.pgv_hnsw_free.argtypes = [P]
    lib.pgv_hnsw_free.restype = None
    lib.pgv_hnsw_score.argtypes = [P, P, I, P, P, I64, P]
    lib.pgv_cosine_distance_batch.argtypes = [P, I, I, P, P, I64, P]
    lib.pgv_bit_distance_batch.argtypes = [P, I, I, P, P, I64, P]
    lib.pgv_hnsw_set_graph.argtypes = [P, I, C.c_int32, P, P, P]
    lib.pgv_hnsw_search.argtypes = [P, P, I, I, I, P, P, P]
    lib.pgv_hnsw_build_search.argtypes = [P, P, P, I, I, I, P, P, P]
    lib.pgv_hnsw_build_neighbors.argtypes = [P, P, P, I, I, I, P, P, P, P, P]
    lib.pgv_hnsw_score_pairs.argtypes = [P, P, P, I64, P]
    lib.pgv_hnsw_update_graph.argtypes = [P, C.c_int32, P, I, P, P]
    return lib


lib = _load()


def check(rc):
    if rc != PGV_OK:
        raise PgvError(rc, lib.pgv_last_error().decode("utf-8", "replace"))
