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


class HnswBuilt(C.Structure):
    _fields_ = [("n", C.c_int64), ("m", C.c_int), ("entry", C.c_int32), ("levels", C.c_void_p),
                ("nbr_start", C.c_void_p), ("nbr", C.c_void_p), ("dup_of", C.c_void_p),
                ("nelements", C.c_int64), ("device_pairs", C.c_int64), ("batches", C.c_int64),
                ("deferred_updates", C.c_int64), ("phase_secs", C.c_double * 8)]


class HnswImage(C.Structure):
    _fields_ = [("dtype", C.c_int), ("dim", C.c_int), ("m", C.c_int), ("ef_construction", C.c_int), ("n", C.c_int64),
                ("entry", C.c_int32), ("vectors", C.c_void_p), ("levels", C.c_void_p), ("nbr_start", C.c_void_p),
                ("nbr", C.c_void_p), ("heaptids", C.c_void_p), ("element_tids", C.c_void_p)]


class Rel(C.Structure):
    _fields_ = [("pages", C.c_void_p), ("nblocks", C.c_uint32), ("cap", C.c_uint32), ("generation", C.c_uint64)]


class IvfImage(C.Structure):
    _fields_ = [("dtype", C.c_int), ("dim", C.c_int), ("lists", C.c_int), ("nrows", C.c_int64),
                ("centers", C.c_void_p), ("list_offsets", C.c_void_p), ("vectors", C.c_void_p),
                ("tids", C.c_void_p), ("start_pages", C.c_void_p)]


DEAD_FN = C.CFUNCTYPE(C.c_int, C.c_uint64, C.c_void_p)  # IndexBulkDeleteCallback


def _load():
    # the HNSW build replays independent selections on OpenMP threads; between parallel regions the main
    # thread talks to the GPU, so idle workers should sleep instead of spinning (measured: 4.6 s -> 3.3 s
    # for a 100 k x 1536 build).  libgomp reads this when it is loaded.
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")
    if not os.path.exists(LIB_PATH):
        raise ImportError("libpgv_host.so not built: make -C pgvector_amd/host")
    lib = C.CDLL(LIB_PATH)
    lib.pgv_host_last_error.restype = C.c_char_p
    lib.pgv_host_hnsw_search.argtypes = [C.c_void_p, C.POINTER(HnswGraphStruct), C.c_int, C.c_int, C.c_void_p,
                                         C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    P, I, I64 = C.c_void_p, C.c_int, C.c_int64
    lib.pgv_host_hnsw_build.argtypes = [P, I, I, P, I64, I, I, P, I, C.POINTER(HnswBuilt)]
    lib.pgv_host_hnsw_built_free.argtypes = [C.POINTER(HnswBuilt)]
    lib.pgv_host_hnsw_built_free.restype = None
    lib.pgv_host_hnsw_write_index.argtypes = [C.POINTER(Rel), I, I, I, I, I64, P, P, P, P, P, P, C.c_int32]
    lib.pgv_host_hnsw_stage.argtypes = [C.POINTER(Rel), I, C.POINTER(HnswImage)]
    lib.pgv_host_hnsw_image_free.argtypes = [C.POINTER(HnswImage)]
    lib.pgv_host_hnsw_image_free.restype = None
    lib.pgv_rel_init.argtypes = [C.POINTER(Rel)]
    lib.pgv_rel_init.restype = None
    lib.pgv_rel_free.argtypes = [C.POINTER(Rel)]
    lib.pgv_rel_free.restype = None
    lib.pgv_host_ivf_write_index.argtypes = [C.POINTER(Rel), I, I, I, P, P, P, P]
    lib.pgv_host_ivf_insert.argtypes = [C.POINTER(Rel), I, I, P, C.c_uint64]
    lib.pgv_host_ivf_bulkdelete.argtypes = [C.POINTER(Rel), DEAD_FN, P, C.POINTER(I64), C.POINTER(I64)]
    lib.pgv_host_ivf_mirror_open.argtypes = [P, I, I, C.POINTER(P)]
    lib.pgv_host_ivf_mirror_get.argtypes = [P, C.POINTER(Rel), C.POINTER(P), C.POINTER(C.POINTER(IvfImage))]
    lib.pgv_host_ivf_mirror_restages.argtypes = [P]
    lib.pgv_host_ivf_mirror_restages.restype = I64
    lib.pgv_host_ivf_mirror_close.argtypes = [P]
    lib.pgv_host_ivf_mirror_close.restype = None
    lib.pgv_host_ivf_stage.argtypes = [C.POINTER(Rel), I, C.POINTER(IvfImage)]
    lib.pgv_host_ivf_image_free.argtypes = [C.POINTER(IvfImage)]
    lib.pgv_host_ivf_image_free.restype = None
    lib.pgv_host_ivf_beginscan.argtypes = [P, C.POINTER(IvfImage), I, I, I, I, C.POINTER(P)]
    lib.pgv_host_ivf_rescan.argtypes = [P, P]
    lib.pgv_host_ivf_gettuple.argtypes = [P, C.POINTER(C.c_uint64), C.POINTER(C.c_double)]
    lib.pgv_host_pool_create.argtypes = [P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(P)]
    lib.pgv_host_pool_search.argtypes = [P, P, P, P]
    lib.pgv_host_pool_stats.argtypes = [P, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    lib.pgv_host_pool_stats.restype = None
    lib.pgv_host_pool_destroy.argtypes = [P]
    lib.pgv_host_pool_destroy.restype = None
    lib.pgv_host_ivf_endscan.argtypes = [P]
    lib.pgv_host_ivf_endscan.restype = None
    lib.pgv_host_ivf_build.argtypes = [P, I, I, I, I, P, P, I64, P, I, P, C.POINTER(Rel)]
    lib.pgv_host_ivf_build_mirror.argtypes = [P, I, I, I, I, P, P, I64, P, I, P, C.POINTER(Rel), C.POINTER(P)]
    lib.pgv_host_ivf_build_phases.argtypes = [C.POINTER(C.c_double)]
    lib.pgv_host_ivf_build_phases.restype = None
    lib.pgv_host_float_to_half.argtypes = [C.c_float]
    lib.pgv_host_float_to_half.restype = C.c_uint16
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


_NP = {0: np.float32, 1: np.float16}


def hnsw_build(mirror, rows, m, ef_construction, rng=None, max_batch=1024):
    """pgv_host_hnsw_build: mirror = api.Hnsw holding exactly `rows`; returns a dict with levels,
    nbr_start, nbr, entry, dup_of (numpy copies) and the counters"""
    rows = np.ascontiguousarray(rows, dtype=_NP[mirror.dtype])
    b = HnswBuilt()
    host_check(lib.pgv_host_hnsw_build(mirror.h, mirror.dtype, rows.shape[1], C.c_void_p(rows.ctypes.data),
                                       rows.shape[0], m, ef_construction,
                                       C.byref(rng) if rng is not None else None, max_batch, C.byref(b)))
    n = b.n

    def copy(ptr, ctype, count):
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=(max(count, 1),))[:count].copy()
    nbr_start = copy(b.nbr_start, C.c_int64, n + 1)
    out = {"levels": copy(b.levels, C.c_int32, n), "nbr_start": nbr_start,
           "nbr": copy(b.nbr, C.c_int32, int(nbr_start[-1]) if n else 0), "dup_of": copy(b.dup_of, C.c_int32, n),
           "entry": b.entry, "m": b.m, "nelements": b.nelements, "device_pairs": b.device_pairs, "batches": b.batches,
           "deferred_updates": b.deferred_updates,
           "phase_secs": dict(zip(("search", "pairs", "select", "records", "update", "patch", "pairlist", "free"), list(b.phase_secs)))}
    lib.pgv_host_hnsw_built_free(C.byref(b))
    return out


class Relation:
    """an IVFFlat index as an array of 8 KB pages (pgv_rel)"""

    def __init__(self):
        self.rel = Rel()
        lib.pgv_rel_init(C.byref(self.rel))

    def __del__(self):
        try:
            lib.pgv_rel_free(C.byref(self.rel))
        except Exception:
            pass

    @property
    def nblocks(self):
        return self.rel.nblocks

    def page(self, blk):
        return np.ctypeslib.as_array(C.cast(self.rel.pages, C.POINTER(C.c_uint8)),
                                     shape=(self.rel.nblocks * 8192,))[blk * 8192:(blk + 1) * 8192]

    def write_index(self, dtype, centers, list_offsets, vectors, tids):
        centers = np.ascontiguousarray(centers, dtype=_NP[dtype])
        vectors = np.ascontiguousarray(vectors, dtype=_NP[dtype])
        list_offsets = np.ascontiguousarray(list_offsets, dtype=np.int64)
        tids = np.ascontiguousarray(tids, dtype=np.uint64)
        host_check(lib.pgv_host_ivf_write_index(C.byref(self.rel), dtype, centers.shape[1], centers.shape[0],
                                                C.c_void_p(centers.ctypes.data), C.c_void_p(list_offsets.ctypes.data),
                                                C.c_void_p(vectors.ctypes.data), C.c_void_p(tids.ctypes.data)))

    def insert(self, dtype, list_id, vector, tid):
        vector = np.ascontiguousarray(vector, dtype=_NP[dtype])
        host_check(lib.pgv_host_ivf_insert(C.byref(self.rel), dtype, list_id, C.c_void_p(vector.ctypes.data), tid))

    @property
    def generation(self):
        return self.rel.generation

    def bulkdelete(self, dead_tids):
        """ambulkdelete with a callback that reports the given heap TIDs dead -> (removed, remaining)"""
        dead = set(int(t) for t in dead_tids)
        cb = DEAD_FN(lambda tid, _state: 1 if tid in dead else 0)
        removed, remaining = C.c_int64(), C.c_int64()
        host_check(lib.pgv_host_ivf_bulkdelete(C.byref(self.rel), cb, None, C.byref(removed), C.byref(remaining)))
        return removed.value, remaining.value

    def stage(self, dtype):
        return StagedImage(self, dtype)

    def write_hnsw(self, dtype, m, ef_construction, vectors, tids, levels, nbr_start, nbr, entry, dup_of=None):
        """FlushPages of an HNSW build (pgv_host_hnsw_write_index)"""
        vectors = np.ascontiguousarray(vectors, dtype=_NP[dtype])
        tids = np.ascontiguousarray(tids, dtype=np.uint64)
        levels = np.ascontiguousarray(levels, dtype=np.int32)
        nbr_start = np.ascontiguousarray(nbr_start, dtype=np.int64)
        nbr = np.ascontiguousarray(nbr, dtype=np.int32)
        dup = None if dup_of is None else np.ascontiguousarray(dup_of, dtype=np.int32)
        host_check(lib.pgv_host_hnsw_write_index(
            C.byref(self.rel), dtype, vectors.shape[1], m, ef_construction, vectors.shape[0],
            C.c_void_p(vectors.ctypes.data), C.c_void_p(tids.ctypes.data), C.c_void_p(levels.ctypes.data),
            C.c_void_p(nbr_start.ctypes.data), C.c_void_p(nbr.ctypes.data),
            None if dup is None else C.c_void_p(dup.ctypes.data), int(entry)))

    def stage_hnsw(self, dtype):
        """pgv_host_hnsw_stage -> dict of numpy copies"""
        img = HnswImage()
        host_check(lib.pgv_host_hnsw_stage(C.byref(self.rel), dtype, C.byref(img)))
        n = img.n

        def copy(ptr, ctype, count, shape=None):
            a = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=(max(count, 1),))[:count].copy()
            return a.reshape(shape) if shape else a
        et = C.c_float if dtype == 0 else C.c_uint16
        nbr_start = copy(img.nbr_start, C.c_int64, n + 1)
        out = {"n": n, "dim": img.dim, "m": img.m, "ef_construction": img.ef_construction, "entry": img.entry,
               "vectors": copy(img.vectors, et, n * img.dim, (n, img.dim)).view(_NP[dtype]),
               "levels": copy(img.levels, C.c_int32, n), "nbr_start": nbr_start,
               "nbr": copy(img.nbr, C.c_int32, int(nbr_start[-1]) if n else 0),
               "heaptids": copy(img.heaptids, C.c_uint64, n * 10, (n, 10)),
               "element_tids": copy(img.element_tids, C.c_uint64, n)}
        lib.pgv_host_hnsw_image_free(C.byref(img))
        return out

    def build(self, ctx, ops, dtype, lists, rows, tids, samples, rng=None):
        rows = np.ascontiguousarray(rows, dtype=_NP[dtype])
        samples = np.ascontiguousarray(samples, dtype=_NP[dtype])
        tids = np.ascontiguousarray(tids, dtype=np.uint64)
        host_check(lib.pgv_host_ivf_build(ctx.h, ops, dtype, rows.shape[1], lists, C.c_void_p(rows.ctypes.data),
                                          C.c_void_p(tids.ctypes.data), rows.shape[0],
                                          C.c_void_p(samples.ctypes.data) if len(samples) else None, len(samples),
                                          C.byref(rng) if rng is not None else None, C.byref(self.rel)))


    def build_mirror(self, ctx, ops, dtype, lists, rows, tids, samples, rng=None):
        """pgv_host_ivf_build_mirror: the pages AND the device mirror of the new index in one go -> api.IvfIndex"""
        from . import api
        rows = np.ascontiguousarray(rows, dtype=_NP[dtype])
        samples = np.ascontiguousarray(samples, dtype=_NP[dtype])
        tids = np.ascontiguousarray(tids, dtype=np.uint64)
        h = C.c_void_p()
        host_check(lib.pgv_host_ivf_build_mirror(ctx.h, ops, dtype, rows.shape[1], lists, C.c_void_p(rows.ctypes.data),
                                                 C.c_void_p(tids.ctypes.data), rows.shape[0],
                                                 C.c_void_p(samples.ctypes.data) if len(samples) else None, len(samples),
                                                 C.byref(rng) if rng is not None else None, C.byref(self.rel), C.byref(h)))
        ix = api.IvfIndex.__new__(api.IvfIndex)
        ix.ctx, ix.dtype, ix.dim, ix.h = ctx, dtype, rows.shape[1], h
        ix.metric = api.PGV_L2SQ if ops == api.PGV_OPS_L2 else api.PGV_NEG_IP
        ix.nlists = lists
        ctx._adopt(ix)
        return ix


class StagedImage:
    """contiguous list-major image staged out of the pages (pgv_host_ivf_stage)"""

    def __init__(self, relation, dtype):
        self.img = IvfImage()
        host_check(lib.pgv_host_ivf_stage(C.byref(relation.rel), dtype, C.byref(self.img)))
        i = self.img
        self.dtype, self.dim, self.lists, self.nrows = dtype, i.dim, i.lists, i.nrows

        def view(ptr, ctype, shape):
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=shape)
        et = C.c_float if dtype == 0 else C.c_uint16
        self.centers = view(i.centers, et, (i.lists, i.dim)).view(_NP[dtype])
        self.list_offsets = view(i.list_offsets, C.c_int64, (i.lists + 1,))
        self.vectors = view(i.vectors, et, (max(i.nrows, 0), i.dim)).view(_NP[dtype]) if i.nrows else \
            np.zeros((0, i.dim), _NP[dtype])
        self.tids = view(i.tids, C.c_uint64, (max(i.nrows, 0),)) if i.nrows else np.zeros(0, np.uint64)
        self.start_pages = view(i.start_pages, C.c_uint32, (i.lists,))

    def __del__(self):
        try:
            lib.pgv_host_ivf_image_free(C.byref(self.img))
        except Exception:
            pass


class _BorrowedIndex:
    """a pgv_index owned by a Mirror"""

    def __init__(self, h):
        self.h = h


class _BorrowedImage:
    def __init__(self, img_ptr, dtype):
        self.img = img_ptr.contents
        self.dtype = dtype
        self.nrows, self.lists, self.dim = self.img.nrows, self.img.lists, self.img.dim


class Mirror:
    """device mirror of a Relation that restages itself when the pages changed (pgv_host_ivf_mirror_*)"""

    def __init__(self, ctx, metric, dtype):
        self.dtype = dtype
        h = C.c_void_p()
        host_check(lib.pgv_host_ivf_mirror_open(ctx.h, metric, dtype, C.byref(h)))
        self.h = h
        ctx._adopt(self)

    def get(self, relation):
        """-> (index handle wrapper, staged image wrapper), valid until the next get()/close()"""
        ih = C.c_void_p()
        img = C.POINTER(IvfImage)()
        host_check(lib.pgv_host_ivf_mirror_get(self.h, C.byref(relation.rel), C.byref(ih), C.byref(img)))
        return _BorrowedIndex(ih), _BorrowedImage(img, self.dtype)

    @property
    def restages(self):
        return lib.pgv_host_ivf_mirror_restages(self.h)

    def close(self):
        if self.h:
            lib.pgv_host_ivf_mirror_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class IvfScan:
    """ivfflatbeginscan ... endscan over a staged image and its device mirror"""

    def __init__(self, mirror, staged, probes, max_probes=0, iterative=False, normalize_query=False):
        self.staged, self.mirror = staged, mirror
        h = C.c_void_p()
        host_check(lib.pgv_host_ivf_beginscan(mirror.h, C.byref(staged.img), probes, max_probes, int(iterative),
                                              int(normalize_query), C.byref(h)))
        self.h = h

    def rescan(self, query):
        q = None if query is None else np.ascontiguousarray(query, dtype=_NP[self.staged.dtype])
        self._q = q
        host_check(lib.pgv_host_ivf_rescan(self.h, None if q is None else C.c_void_p(q.ctypes.data)))

    def fetch(self, limit=None):
        """pull heap TIDs like the executor does (amgettuple until exhausted or LIMIT)"""
        tids, dists = [], []
        tid, dist = C.c_uint64(), C.c_double()
        while limit is None or len(tids) < limit:
            rc = lib.pgv_host_ivf_gettuple(self.h, C.byref(tid), C.byref(dist))
            if rc < 0:
                host_check(_lib.PGV_ERR_DEVICE)
            if rc == 0:
                break
            tids.append(tid.value)
            dists.append(dist.value)
        return np.array(tids, dtype=np.uint64), np.array(dists)

    def close(self):
        if self.h:
            lib.pgv_host_ivf_endscan(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Pool:
    """pgv_host_pool_*: backends hand in one query each; arrivals within max_wait_us share one pgv_search_batch"""

    def __init__(self, index, probes, k, max_batch=1024, max_wait_us=50, lanes=2, device=0):
        self.k, self.dtype, self.dim = k, index.dtype, index.dim
        h = C.c_void_p()
        host_check(lib.pgv_host_pool_create(index.h, device, index.dtype, index.dim, probes, k, max_batch, max_wait_us,
                                            lanes, C.byref(h)))
        self.h = h

    def search(self, query):
        """blocking, thread-safe: (tids [k] uint64, distances [k] float32)"""
        q = np.ascontiguousarray(query, dtype=_NP[self.dtype])
        tid = np.empty(self.k, dtype=np.uint64)
        dist = np.empty(self.k, dtype=np.float32)
        host_check(lib.pgv_host_pool_search(self.h, q.ctypes.data, tid.ctypes.data, dist.ctypes.data))
        return tid, dist

    def stats(self):
        b, q = C.c_int64(), C.c_int64()
        lib.pgv_host_pool_stats(self.h, C.byref(b), C.byref(q))
        return {"batches": b.value, "queries": q.value}

    def close(self):
        if self.h:
            lib.pgv_host_pool_destroy(self.h)
            self.h = None


# ---------------------------------------------------------------- backends that are processes (tools/)
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BACKEND_EXE = os.path.join(_ROOT, "build", "tools", "pgv_backend")
BACKENDS_SO = os.path.join(_ROOT, "build", "tools", "libbackends.so")
_drv = None


def backends_driver():
    """tools/backends_driver.c as a shared object (built by __graft_entry__.build())"""
    global _drv
    if _drv is None:
        if not (os.path.exists(BACKENDS_SO) and os.path.exists(BACKEND_EXE)):
            raise ImportError("build/tools/libbackends.so / pgv_backend not built: run __graft_entry__.build()")
        d = C.CDLL(BACKENDS_SO)
        P, I = C.c_void_p, C.c_int
        D = C.c_double
        d.backends_run.argtypes = [P, I, I, I, P, I, C.c_size_t, I, I, D, C.POINTER(D), C.c_char_p, C.c_size_t]
        d.pool_run.argtypes = [P, I, I, I, I, I, P, I, C.c_size_t, I, I, I, I, I, D, C.POINTER(D), C.c_char_p, C.c_size_t]
        d.backends_run_processes.argtypes = [P, C.c_char_p, I, I, I, I, I, P, I, C.c_size_t, I, I, I, I, I, I, I, I,
                                             C.c_char_p, I, D, I, P, P, C.POINTER(D), C.c_char_p, C.c_size_t]
        _drv = d
    return _drv


def write_index_image(name, metric, dtype, dim, centers, list_offsets, vectors, tids):
    """a pgvb_image segment (tools/pgv_backend_shm.h) under /dev/shm for `pgv_backend owner`; returns its shm name"""
    centers = np.ascontiguousarray(centers)
    vectors = np.ascontiguousarray(vectors)
    offs = np.ascontiguousarray(list_offsets, dtype=np.int64)
    tids = None if tids is None else np.ascontiguousarray(tids, dtype=np.uint64)
    head = 4096
    parts, at = [], head
    for a in (centers, offs, vectors, tids):
        if a is None:
            parts.append(0)
            continue
        parts.append(at)
        at = (at + a.nbytes + 4095) & ~4095
    path = "/dev/shm/" + name.lstrip("/")
    with open(path, "wb") as f:
        f.truncate(at)
    mm = np.memmap(path, dtype=np.uint8, mode="r+")
    hdr = np.zeros(8, dtype=np.uint64)
    hdr[0] = 0x7067765f696d6167
    hdr[1] = (int(metric) & 0xffffffff) | (int(dtype) << 32)
    hdr[2] = (int(dim) & 0xffffffff) | (int(centers.shape[0]) << 32)
    hdr[3] = int(vectors.shape[0])
    hdr[4:8] = parts
    mm[:64] = hdr.view(np.uint8)
    for off, a in zip(parts, (centers, offs, vectors, tids)):
        if a is not None:
            mm[off:off + a.nbytes] = a.view(np.uint8).reshape(-1)
    mm.flush()
    del mm
    return "/" + name.lstrip("/")


def run_backend_threads(index, queries, probes, k, nbackends, per_thread, device=0, deadline_s=60.0):
    """N backends as THREADS of this process (one pgv_ctx + pgv_index_share view + pgv_query each), one query at a time
    each.  A backend not back after deadline_s raises with the call it sits in."""
    d = backends_driver()
    q = np.ascontiguousarray(queries)
    out = (C.c_double * 4)()
    err = C.create_string_buffer(1024)
    rc = d.backends_run(index.h, device, nbackends, per_thread, q.ctypes.data, q.shape[0], q.strides[0], probes, min(k, 64),
                        float(deadline_s), out, err, len(err))
    if rc != 0:
        raise _lib.PgvError(rc, "backends_run: " + err.value.decode("utf-8", "replace"))
    return {"qps": out[0], "latency_us_p50": out[1], "latency_us_p90": out[2]}


def run_pooled_threads(index, queries, probes, k, nclients, per_thread, max_batch=1024, max_wait_us=50, lanes=2, device=0,
                       deadline_s=60.0):
    """N client THREADS behind pgv_host_pool_* (lane threads in this process), one query each at a time"""
    d = backends_driver()
    q = np.ascontiguousarray(queries)
    out = (C.c_double * 4)()
    err = C.create_string_buffer(1024)
    rc = d.pool_run(index.h, device, index.dtype, index.dim, nclients, per_thread, q.ctypes.data, q.shape[0], q.strides[0],
                    probes, min(k, 64), max_batch, max_wait_us, lanes, float(deadline_s), out, err, len(err))
    if rc != 0:
        raise _lib.PgvError(rc, "pool_run: " + err.value.decode("utf-8", "replace"))
    return {"qps": out[0], "latency_us_p50": out[1], "latency_us_p90": out[2], "mean_batch": out[3]}


def run_backend_processes(index, queries, probes, k, mode, nclients, per_client, warmup=5, max_batch=1024,
                          max_wait_us=50, lanes=2, server_processes=True, verify=False, image_shm=None, device=0,
                          deadline_s=60.0, chaos=0):
    """N backend PROCESSES against one device mirror.  mode 0: every process imports the mirror and runs
    pgv_query_*; mode 1: GPU-less clients behind the shared-memory pooler.  Returns a dict (qps, latencies, mean
    batch, HBM that went to the children) and, with verify, the answers [nclients, per_client, k]."""
    d = backends_driver()
    q = np.ascontiguousarray(queries)
    out = (C.c_double * 8)()
    err = C.create_string_buffer(512)
    k = min(k, 64)
    ans_t = np.zeros((nclients, per_client, k), dtype=np.uint64) if verify else None
    ans_d = np.zeros((nclients, per_client, k), dtype=np.float32) if verify else None
    rc = d.backends_run_processes(index.h if index is not None else None, image_shm.encode() if image_shm else None,
                                  device, mode, nclients, per_client, warmup, q.ctypes.data, q.shape[0], q.strides[0],
                                  index.dtype if index is not None else (0 if q.dtype == np.float32 else 1), q.shape[1],
                                  probes, k, max_batch, max_wait_us, lanes, 1 if server_processes else 0,
                                  BACKEND_EXE.encode(), 1 if verify else 0, float(deadline_s), int(chaos),
                                  ans_t.ctypes.data if verify else None, ans_d.ctypes.data if verify else None,
                                  out, err, len(err))
    if rc != 0:
        raise _lib.PgvError(rc, "backends_run_processes: " + err.value.decode("utf-8", "replace"))
    res = {"qps": out[0], "latency_us_p50": out[1], "latency_us_p90": out[2], "mean_batch": out[3],
           "hbm_bytes_taken_by_children": out[4], "processes": int(out[5]), "clients_completed": int(out[6]),
           "clients_failed": int(out[7])}
    return (res, ans_t, ans_d) if verify else res
