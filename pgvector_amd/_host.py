"""ctypes binding of libpgv_host.so (pgvector_amd/host/pgv_host.h): the C host
glue above the libpgv_hip ABI.  Harness only, like _lib.py."""
import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import check  # noqa: F401

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libpgv_host.so")


class HnswGraphStruct(C.Structure):
    _fields_ = [("nelements", C.c_int64), ("m", C.c_int), ("entry", C.c_int32),
                ("levels", C.c_void_p), ("nbr_start", C.c_void_p), ("nbr", C.c_void_p)]


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError("libpgv_host.so not built: make -C pgvector_amd/host")
    lib = C.CDLL(LIB_PATH)
    lib.pgv_host_last_error.restype = C.c_char_p
    lib.pgv_host_hnsw_search.argtypes = [C.c_void_p, C.POINTER(HnswGraphStruct), C.c_int, C.c_int, C.c_void_p,
                                         C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    return lib


lib = _load()


def host_check(rc):
    if rc != _lib.PGV_OK:
        raise _lib.PgvError(rc, lib.pgv_host_last_error().decode("utf-8", "replace"))


def hnsw_graph(levels, nbr_start, nbr, m, entry):
    """levels [n] int32; nbr laid out like HnswNeighborTupleData per element (src/hnsw.h:384-392)"""
    levels = np.ascontiguousarray(levels, dtype=np.int32)
    nbr_start = np.ascontiguousarray(nbr_start, dtype=np.int64)
    nbr = np.ascontiguousarray(nbr, dtype=np.int32)
    g = HnswGraphStruct()
    g.nelements, g.m, g.entry = len(levels), m, entry
    g.levels, g.nbr_start, g.nbr = levels.ctypes.data, nbr_start.ctypes.data, nbr.ctypes.data
    g._keep = (levels, nbr_start, nbr)
    return g


def hnsw_search(mirror, graph, queries, ef_search, k):
    """mirror: api.Hnsw; queries: host array already normalised for cosine"""
    queries = np.ascontiguousarray(queries)
    nq = queries.shape[0]
    elem = np.empty((nq, k), dtype=np.int64)
    dist = np.empty((nq, k), dtype=np.float32)
    scored = np.empty(nq, dtype=np.int64)
    host_check(lib.pgv_host_hnsw_search(mirror.h, C.byref(graph), mirror.dtype, mirror.dim,
                                        C.c_void_p(queries.ctypes.data), nq, ef_search, k,
                                        C.c_void_p(elem.ctypes.data), C.c_void_p(dist.ctypes.data),
                                        C.c_void_p(scored.ctypes.data)))
    return elem, dist, scored
