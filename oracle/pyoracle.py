"""ctypes face of oracle/liboracle.so (and oracle/_ref/libpgvref.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never by pgvector_amd.  See pgv_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORA_F32, ORA_F16 = 0, 1
OPS_L2, OPS_IP, OPS_COSINE, OPS_L1 = 0, 1, 2, 3
NP_OF = {ORA_F32: np.float32, ORA_F16: np.float16}


def build(native=False):
    """(re)build the restatement from its own sources; returns the .so path.  The -march=native build is
    ALWAYS redone on the host that uses it (a copy made elsewhere may use instructions this CPU lacks, or
    predate the sources)."""
    target = "liboracle_native.so" if native else "liboracle.so"
    subprocess.run(["make", "-s"] + (["-B"] if native else []) + ["-C", HERE, target], check=True)
    return os.path.join(HERE, target)


class Prng(C.Structure):
    _fields_ = [("s0", C.c_uint64), ("s1", C.c_uint64)]


class IvfIndexStruct(C.Structure):
    _fields_ = [("ops", C.c_int), ("dtype", C.c_int), ("dim", C.c_int), ("nlists", C.c_int),
                ("centers", C.c_void_p), ("list_offsets", C.c_void_p), ("vectors", C.c_void_p),
                ("tids", C.c_void_p)]


def _p(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


class Oracle:
    def __init__(self, native=False, path=None):
        if path is None:
            path = os.path.join(HERE, "liboracle_native.so" if native else "liboracle.so")
            if native or not os.path.exists(path):
                path = build(native)
        self.path = path
        L = self.lib = C.CDLL(path)
        F, D, I, P, I64 = C.c_float, C.c_double, C.c_int, C.c_void_p, C.c_int64
        for name, res in [("ora_vector_l2_squared", F), ("ora_vector_inner_product", F),
                          ("ora_vector_cosine_similarity", D), ("ora_vector_l1", F),
                          ("ora_halfvec_l2_squared", F), ("ora_halfvec_inner_product", F),
                          ("ora_halfvec_cosine_similarity", D), ("ora_halfvec_l1", F),
                          ("ora_halfvec_l2_squared_default", F), ("ora_halfvec_inner_product_default", F)]:
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = [I, P, P]
        for name in ["ora_l2_distance", "ora_l2_squared_distance", "ora_inner_product",
                     "ora_negative_inner_product", "ora_cosine_distance", "ora_spherical_distance",
                     "ora_l1_distance", "ora_halfvec_l2_distance", "ora_halfvec_l2_squared_distance",
                     "ora_halfvec_inner_product_f8", "ora_halfvec_negative_inner_product",
                     "ora_halfvec_cosine_distance", "ora_halfvec_spherical_distance",
                     "ora_halfvec_l1_distance"]:
            fn = getattr(L, name)
            fn.restype = I
            fn.argtypes = [I, P, I, P, C.POINTER(D)]
        L.ora_last_error.restype = C.c_char_p
        L.ora_vector_norm.restype = D
        L.ora_vector_norm.argtypes = [I, P]
        L.ora_halfvec_l2_norm.restype = D
        L.ora_halfvec_l2_norm.argtypes = [I, P]
        L.ora_l2_normalize.argtypes = [I, P, P]
        L.ora_halfvec_l2_normalize.argtypes = [I, P, P]
        L.ora_half_to_float.restype = F
        L.ora_half_to_float.argtypes = [C.c_uint16]
        L.ora_float_to_half.restype = C.c_uint16
        L.ora_float_to_half.argtypes = [F]
        L.ora_halfvec_uses_f16c.restype = I
        L.ora_index_distance.restype = D
        L.ora_index_distance.argtypes = [I, I, I, P, P]
        L.ora_kmeans_distance.restype = D
        L.ora_kmeans_distance.argtypes = [I, I, I, P, P]
        L.ora_prng_seed.argtypes = [C.POINTER(Prng), C.c_uint64]
        L.ora_prng_u32.restype = C.c_uint32
        L.ora_prng_u32.argtypes = [C.POINTER(Prng)]
        L.ora_prng_double.restype = D
        L.ora_prng_double.argtypes = [C.POINTER(Prng)]
        L.ora_ivf_get_scan_lists.restype = I
        L.ora_ivf_get_scan_lists.argtypes = [C.POINTER(IvfIndexStruct), P, I, P, P]
        L.ora_ivf_get_scan_items.restype = I64
        L.ora_ivf_get_scan_items.argtypes = [C.POINTER(IvfIndexStruct), P, P, I, P, P]
        L.ora_ivf_search.restype = I
        L.ora_ivf_search.argtypes = [C.POINTER(IvfIndexStruct), P, I, I, P, P]
        L.ora_pages_meta.argtypes = [P, C.c_uint32, C.POINTER(I), C.POINTER(I)]
        L.ora_pages_search.argtypes = [P, C.c_uint32, I, I, P, I, I, P, P, C.POINTER(C.c_int64)]
        L.ora_ivf_assign.restype = None
        L.ora_ivf_assign.argtypes = [I, I, I, P, I, P, I64, P, P]
        L.ora_ivf_num_samples.restype = I
        L.ora_ivf_num_samples.argtypes = [I, I64]
        L.ora_kmeans_init_centers.restype = None
        L.ora_kmeans_init_centers.argtypes = [I, I, I, P, I, P, I, P, C.POINTER(Prng)]
        L.ora_kmeans_compute_new_centers.restype = None
        L.ora_kmeans_compute_new_centers.argtypes = [I, I, I, P, I, P, P, I, P, C.POINTER(Prng)]
        L.ora_kmeans.restype = I
        L.ora_kmeans.argtypes = [I, I, I, P, I, P, I, C.POINTER(Prng), P]
        L.ora_kmeans_lloyd_assign.restype = None
        L.ora_kmeans_lloyd_assign.argtypes = [I, I, I, P, I, P, I, P, P]
        if hasattr(L, "ora_bench_search"):
            L.ora_bench_cpus.restype = I
            L.ora_bench_alloc.restype = P
            L.ora_bench_alloc.argtypes = [C.c_size_t]
            L.ora_bench_free.restype = None
            L.ora_bench_free.argtypes = [P, C.c_size_t]
            L.ora_bench_spread_copy.argtypes = [P, P, C.c_size_t, I]
            L.ora_bench_search.argtypes = [C.POINTER(IvfIndexStruct), P, C.c_uint32, I, I, P, C.c_size_t, I, I, I, I, D,
                                           P, P, P, P]
            L.ora_bench_assign.argtypes = [I, I, I, P, I, P, I64, I, P, C.POINTER(D)]
        self.has_hnsw = hasattr(L, "ora_hnsw_build")
        if self.has_hnsw:
            L.ora_hnsw_build.restype = P
            L.ora_hnsw_build.argtypes = [I, I, I, P, I64, I, I, C.c_uint64]
            L.ora_hnsw_build_parallel.restype = P
            L.ora_hnsw_build_parallel.argtypes = [I, I, I, P, I64, I, I, C.c_uint64, I]
            L.ora_hnsw_import.restype = P
            L.ora_hnsw_import.argtypes = [I, I, I, P, I64, I, P, P, P, C.c_int32]
            L.ora_hnsw_free.argtypes = [P]
            L.ora_hnsw_free.restype = None
            L.ora_hnsw_num_elements.restype = I64
            L.ora_hnsw_num_elements.argtypes = [P]
            L.ora_hnsw_entry_point.restype = I
            L.ora_hnsw_entry_point.argtypes = [P, C.POINTER(I)]
            L.ora_hnsw_m.restype = I
            L.ora_hnsw_m.argtypes = [P]
            L.ora_hnsw_level.restype = I
            L.ora_hnsw_level.argtypes = [P, I64]
            L.ora_hnsw_neighbors.restype = I
            L.ora_hnsw_neighbors.argtypes = [P, I64, I, P]
            L.ora_hnsw_element_row.restype = I64
            L.ora_hnsw_element_row.argtypes = [P, I64]
            L.ora_hnsw_search.restype = I
            L.ora_hnsw_search.argtypes = [P, P, I, I, P, P, C.POINTER(I64)]

    # ---- helpers -------------------------------------------------------
    @staticmethod
    def arr(x, dtype):
        return np.ascontiguousarray(x, dtype=NP_OF[dtype])

    def last_error(self):
        return self.lib.ora_last_error().decode()

    def sql(self, name, a, b, half=False):
        """an fmgr-level wrapper; returns (rc, value)"""
        dt = ORA_F16 if half else ORA_F32
        a, b = self.arr(a, dt), self.arr(b, dt)
        out = C.c_double()
        rc = getattr(self.lib, name)(len(a), _p(a), len(b), _p(b), C.byref(out))
        return rc, out.value

    @staticmethod
    def pack_bits(bits):
        """a PostgreSQL bit string ('0101...' or a 0/1 array) -> (nbits, packed bytes, first bit = MSB of byte 0)"""
        if isinstance(bits, str):
            bits = np.array([c == "1" for c in bits], dtype=np.uint8)
        bits = np.asarray(bits, dtype=np.uint8)
        return len(bits), np.ascontiguousarray(np.packbits(bits)) if len(bits) else np.zeros(0, np.uint8)

    def bit_sql(self, name, a, b):
        """ora_hamming_distance / ora_jaccard_distance on bit strings; returns (rc, value)"""
        na, pa = self.pack_bits(a)
        nb, pb = self.pack_bits(b)
        fn = getattr(self.lib, name)
        fn.restype = C.c_int
        fn.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_double)]
        out = C.c_double()
        pa = np.concatenate([pa, np.zeros(8, np.uint8)])  # keep the pointers valid for empty strings
        pb = np.concatenate([pb, np.zeros(8, np.uint8)])
        rc = fn(na, _p(pa), nb, _p(pb), C.byref(out))
        return rc, out.value

    def bit_rows(self, name, query, rows):
        """packed rows [n x bytes] against one packed query -> float64 [n] (ora_bit_hamming / ora_bit_jaccard)"""
        fn = getattr(self.lib, name)
        fn.restype = C.c_uint64 if name == "ora_bit_hamming" else C.c_double
        fn.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p]
        rows = np.ascontiguousarray(rows, dtype=np.uint8)
        query = np.ascontiguousarray(query, dtype=np.uint8)
        return np.array([float(fn(rows.shape[1], _p(r), _p(query))) for r in rows], dtype=np.float64)

    def kernel(self, name, a, b, half=False):
        dt = ORA_F16 if half else ORA_F32
        a, b = self.arr(a, dt), self.arr(b, dt)
        assert len(a) == len(b)
        return getattr(self.lib, name)(len(a), _p(a), _p(b))

    def prng(self, seed):
        st = Prng()
        self.lib.ora_prng_seed(C.byref(st), seed)
        return st

    def index_struct(self, ops, dtype, centers, list_offsets, vectors, tids=None):
        """keeps the arrays alive on the returned object"""
        centers = self.arr(centers, dtype)
        vectors = self.arr(vectors, dtype)
        list_offsets = np.ascontiguousarray(list_offsets, dtype=np.int64)
        s = IvfIndexStruct()
        s.ops, s.dtype, s.dim, s.nlists = ops, dtype, centers.shape[1], centers.shape[0]
        s.centers, s.list_offsets, s.vectors = centers.ctypes.data, list_offsets.ctypes.data, vectors.ctypes.data
        if tids is not None:
            tids = np.ascontiguousarray(tids, dtype=np.uint64)
            s.tids = tids.ctypes.data
        s._keep = (centers, vectors, list_offsets, tids)
        return s

    def get_scan_lists(self, ix, query, maxprobes):
        q = None if query is None else self.arr(query, ix.dtype)
        m = min(maxprobes, ix.nlists)
        lists = np.empty(m, dtype=np.int32)
        dist = np.empty(m, dtype=np.float64)
        n = self.lib.ora_ivf_get_scan_lists(C.byref(ix), _p(q), maxprobes, _p(lists), _p(dist))
        return lists[:n], dist[:n]

    def get_scan_items(self, ix, query, lists):
        q = None if query is None else self.arr(query, ix.dtype)
        lists = np.ascontiguousarray(lists, dtype=np.int32)
        off = ix._keep[2]
        total = int(sum(off[l + 1] - off[l] for l in lists))
        dist = np.empty(max(total, 1), dtype=np.float64)
        slot = np.empty(max(total, 1), dtype=np.int64)
        n = self.lib.ora_ivf_get_scan_items(C.byref(ix), _p(q), _p(lists), len(lists), _p(dist), _p(slot))
        return dist[:n], slot[:n]

    def search(self, ix, query, probes, k):
        q = None if query is None else self.arr(query, ix.dtype)
        tids = np.empty(k, dtype=np.uint64)
        dist = np.empty(k, dtype=np.float64)
        n = self.lib.ora_ivf_search(C.byref(ix), _p(q), probes, k, _p(tids), _p(dist))
        return tids[:n], dist[:n]

    def pages_search(self, pages_ptr, nblocks, ops, dtype, query, probes, k):
        """ivfflatgettuple's first batch over an array of 8 KB pages (oracle_pages.c); pages_ptr = address of
        block 0 -> (tids, distances, tuples scanned)"""
        q = None if query is None else self.arr(query, dtype)
        tids = np.empty(k, dtype=np.uint64)
        dist = np.empty(k, dtype=np.float64)
        scanned = C.c_int64()
        n = self.lib.ora_pages_search(C.c_void_p(pages_ptr), nblocks, ops, dtype, _p(q), probes, k, _p(tids), _p(dist),
                                      C.byref(scanned))
        if n < 0:
            raise ValueError("not an IVFFlat page image")
        return tids[:n], dist[:n], scanned.value

    # ---- bench.py's thread runners (oracle_bench.c) ---------------------
    def spread(self, a, nthreads):
        """a copy of the array in pages spread round the machine's memory nodes (first touch by pinned threads);
        returns (ndarray view, release())"""
        a = np.ascontiguousarray(a)
        nbytes = max(a.nbytes, 1)
        p = self.lib.ora_bench_alloc(nbytes)
        if not p:
            raise MemoryError("ora_bench_alloc(%d)" % nbytes)
        self.lib.ora_bench_spread_copy(p, a.ctypes.data, a.nbytes, nthreads)
        buf = (C.c_char * nbytes).from_address(p)
        view = np.frombuffer(buf, dtype=a.dtype, count=a.size).reshape(a.shape)
        return view, (lambda: self.lib.ora_bench_free(p, nbytes))

    def bench_search(self, ix, queries, probes, k, nthreads, seconds, pages=None, ops=0, dtype=0):
        """nthreads pinned threads answering the queries round-robin for `seconds` (each at least once) ->
        (answers [(tids, dist)] per query, queries answered, wall seconds)"""
        q = np.ascontiguousarray(queries)
        nq = q.shape[0]
        tids = np.zeros((nq, k), dtype=np.uint64)
        dist = np.zeros((nq, k), dtype=np.float64)
        cnt = np.zeros(nq, dtype=np.int32)
        stats = np.zeros(2, dtype=np.float64)
        if pages is None:
            rc = self.lib.ora_bench_search(C.byref(ix), None, 0, ix.ops, ix.dtype, _p(q), q.strides[0], nq, probes, k,
                                           nthreads, seconds, _p(tids), _p(dist), _p(cnt), _p(stats))
        else:
            ptr, nblocks = pages
            rc = self.lib.ora_bench_search(None, ptr, nblocks, ops, dtype, _p(q), q.strides[0], nq, probes, k, nthreads,
                                           seconds, _p(tids), _p(dist), _p(cnt), _p(stats))
        if rc != 0:
            raise RuntimeError("ora_bench_search rc %d" % rc)
        answers = [(tids[i, :cnt[i]].copy(), dist[i, :cnt[i]].copy()) for i in range(nq)]
        return answers, int(stats[0]), float(stats[1])

    def bench_assign(self, ops, dtype, centers, rows, nthreads):
        """rows split between nthreads pinned workers, each running the argmin loop -> (lists, wall seconds)"""
        centers, rows = self.arr(centers, dtype), self.arr(rows, dtype)
        out = np.empty(rows.shape[0], dtype=np.int32)
        secs = C.c_double()
        rc = self.lib.ora_bench_assign(ops, dtype, rows.shape[1], _p(centers), centers.shape[0], _p(rows), rows.shape[0],
                                       nthreads, _p(out), C.byref(secs))
        if rc != 0:
            raise RuntimeError("ora_bench_assign rc %d" % rc)
        return out, secs.value

    def assign(self, ops, dtype, centers, rows):
        centers, rows = self.arr(centers, dtype), self.arr(rows, dtype)
        n = rows.shape[0]
        out = np.empty(n, dtype=np.int32)
        dist = np.empty(n, dtype=np.float64)
        self.lib.ora_ivf_assign(ops, dtype, centers.shape[1], _p(centers), centers.shape[0], _p(rows), n,
                                _p(out), _p(dist))
        return out, dist

    def kmeans_init_centers(self, ops, dtype, samples, k, rng):
        samples = self.arr(samples, dtype)
        centers = np.zeros((k, samples.shape[1]), dtype=NP_OF[dtype])
        self.lib.ora_kmeans_init_centers(ops, dtype, samples.shape[1], _p(samples), samples.shape[0],
                                         _p(centers), k, None, C.byref(rng))
        return centers

    def kmeans_compute_new_centers(self, ops, dtype, samples, closest, k, rng):
        samples = self.arr(samples, dtype)
        closest = np.ascontiguousarray(closest, dtype=np.int32)
        centers = np.zeros((k, samples.shape[1]), dtype=NP_OF[dtype])
        counts = np.zeros(k, dtype=np.int32)
        self.lib.ora_kmeans_compute_new_centers(ops, dtype, samples.shape[1], _p(samples), samples.shape[0],
                                                _p(closest), _p(centers), k, _p(counts), C.byref(rng))
        return centers, counts

    def kmeans(self, ops, dtype, samples, k, rng):
        samples = self.arr(samples, dtype)
        n, dim = samples.shape if samples.ndim == 2 else (0, 0)
        centers = np.zeros((k, dim), dtype=NP_OF[dtype])
        closest = np.zeros(max(n, 1), dtype=np.int32)
        it = self.lib.ora_kmeans(ops, dtype, dim, _p(samples) if n else None, n, _p(centers), k,
                                 C.byref(rng), _p(closest))
        return centers, closest[:n], it

    def lloyd_assign(self, ops, dtype, samples, centers):
        samples, centers = self.arr(samples, dtype), self.arr(centers, dtype)
        n = samples.shape[0]
        closest = np.empty(n, dtype=np.int32)
        dist = np.empty(n, dtype=np.float32)
        self.lib.ora_kmeans_lloyd_assign(ops, dtype, samples.shape[1], _p(samples), n, _p(centers),
                                         centers.shape[0], _p(closest), _p(dist))
        return closest, dist


class HnswGraph:
    """in-memory HNSW graph built by the oracle's restatement of the reference build"""

    def __init__(self, ora, ops, dtype, rows, m=16, ef_construction=64, seed=0, threads=0):
        """threads = 0: the serial build (src/hnswbuild.c without parallel workers); threads >= 1: the reference's
        parallel build restated (per-element locks, one shared graph) -- one thread gives the serial graph"""
        self.ora, self.ops, self.dtype = ora, ops, dtype
        self.rows = ora.arr(rows, dtype)
        if threads > 0:
            self.h = ora.lib.ora_hnsw_build_parallel(ops, dtype, self.rows.shape[1], _p(self.rows), self.rows.shape[0],
                                                     m, ef_construction, seed, int(threads))
        else:
            self.h = ora.lib.ora_hnsw_build(ops, dtype, self.rows.shape[1], _p(self.rows), self.rows.shape[0],
                                            m, ef_construction, seed)
        self.m = m

    @classmethod
    def from_tuples(cls, ora, ops, dtype, values, m, levels, nbr_start, nbr, entry):
        """a graph built elsewhere (the GPU build, a staged index) in the index's neighbor-tuple layout; `values` are
        the index values (normalised for cosine), element e answers with row e.  Only search() applies."""
        self = cls.__new__(cls)
        self.ora, self.ops, self.dtype, self.m = ora, ops, dtype, m
        self.rows = ora.arr(values, dtype)
        levels = np.ascontiguousarray(levels, dtype=np.int32)
        nbr_start = np.ascontiguousarray(nbr_start, dtype=np.int64)
        nbr = np.ascontiguousarray(nbr, dtype=np.int32)
        self.h = ora.lib.ora_hnsw_import(ops, dtype, self.rows.shape[1], _p(self.rows), self.rows.shape[0], m,
                                         _p(levels), _p(nbr_start), _p(nbr), int(entry))
        return self

    def close(self):
        if self.h:
            self.ora.lib.ora_hnsw_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def nelements(self):
        return self.ora.lib.ora_hnsw_num_elements(self.h)

    def search(self, query, ef_search, k):
        q = self.ora.arr(query, self.dtype)
        rows = np.empty(k, dtype=np.int64)
        dist = np.empty(k, dtype=np.float64)
        scored = C.c_int64()
        n = self.ora.lib.ora_hnsw_search(self.h, _p(q), ef_search, k, _p(rows), _p(dist), C.byref(scored))
        return rows[:n], dist[:n], scored.value

    def export(self):
        """flat arrays for a device/host mirror: per element row, level, heap-tid-free neighbor table.
        neighbors[e, lc, :] padded with -1; layer 0 has 2m slots, upper layers m (src/hnsw.h:127)"""
        L = self.ora.lib
        n = self.nelements
        lvl = C.c_int()
        entry = L.ora_hnsw_entry_point(self.h, C.byref(lvl))
        levels = np.array([L.ora_hnsw_level(self.h, e) for e in range(n)], dtype=np.int32)
        rows = np.array([L.ora_hnsw_element_row(self.h, e) for e in range(n)], dtype=np.int64)
        maxl = int(levels.max()) if n else 0
        nbr = np.full((n, maxl + 1, 2 * self.m), -1, dtype=np.int32)
        buf = np.empty(2 * self.m, dtype=np.int32)
        for e in range(n):
            for lc in range(levels[e] + 1):
                c = L.ora_hnsw_neighbors(self.h, e, lc, _p(buf))
                nbr[e, lc, :c] = buf[:c]
        return {"entry": entry, "entry_level": lvl.value, "levels": levels, "rows": rows, "neighbors": nbr}

    def export_tuples(self):
        """neighbor tuples as the index stores them (src/hnsw.h:384-392, src/hnswutils.c:786):
        (level + 2) * m slots per element, layer lc at (level - lc) * m, -1 = invalid TID"""
        ex = self.export()
        n, m = len(ex["levels"]), self.m
        sizes = (ex["levels"].astype(np.int64) + 2) * m
        start = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        nbr = np.full(int(start[-1]), -1, dtype=np.int32)
        for e in range(n):
            lvl = int(ex["levels"][e])
            for lc in range(lvl + 1):
                lm = 2 * m if lc == 0 else m
                o = int(start[e]) + (lvl - lc) * m
                nbr[o:o + lm] = ex["neighbors"][e, lc, :lm]
        ex.update({"nbr_start": start, "nbr": nbr})
        return ex


class Ref:
    """oracle/_ref/libpgvref.so: the reference's src/halfutils.c compiled unmodified"""

    def __init__(self):
        path = os.path.join(HERE, "_ref", "libpgvref.so")
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        L = self.lib = C.CDLL(path)
        L.pgvref_init()
        for name, res in [("pgvref_halfvec_l2_squared", C.c_float), ("pgvref_halfvec_inner_product", C.c_float),
                          ("pgvref_halfvec_cosine_similarity", C.c_double), ("pgvref_halfvec_l1", C.c_float)]:
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
        L.pgvref_half_to_float.restype = C.c_float
        L.pgvref_half_to_float.argtypes = [C.c_uint16]
        L.pgvref_float_to_half.restype = C.c_uint16
        L.pgvref_float_to_half.argtypes = [C.c_float]
        if hasattr(L, "pgvref_bit_init"):  # src/bitutils.c compiled unmodified as well
            L.pgvref_bit_init()
            L.pgvref_bit_hamming.restype = C.c_uint64
            L.pgvref_bit_hamming.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p]
            L.pgvref_bit_jaccard.restype = C.c_double
            L.pgvref_bit_jaccard.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p]

    def kernel(self, name, a, b):
        a = np.ascontiguousarray(a, dtype=np.float16)
        b = np.ascontiguousarray(b, dtype=np.float16)
        return getattr(self.lib, name)(len(a), _p(a), _p(b))


class Ref32:
    """oracle/_ref/libpgvref32.so: the reference's src/vector.c compiled unmodified (oracle/ref_glue32.c);
    its SQL-callable fp32 distance functions by name"""

    FUNCTIONS = ("l2_distance", "vector_l2_squared_distance", "inner_product", "vector_negative_inner_product",
                 "cosine_distance", "vector_spherical_distance", "l1_distance", "vector_norm")

    def __init__(self):
        path = os.path.join(HERE, "_ref", "libpgvref32.so")
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        L = self.lib = C.CDLL(path)
        L.pgvref32_call.restype = C.c_int
        L.pgvref32_rows.restype = C.c_int
        L.pgvref32_l2_normalize.restype = C.c_int
        L.pgvref32_last_error.restype = C.c_char_p

    def call(self, name, a, b=None):
        """-> (rc, float8 value); rc 1 = the reference raised ERROR, text in last_error()"""
        a = np.ascontiguousarray(a, dtype=np.float32)
        if b is not None:
            b = np.ascontiguousarray(b, dtype=np.float32)
        out = C.c_double()
        rc = self.lib.pgvref32_call(name.encode(), len(a), _p(a), 0 if b is None else len(b), _p(b), C.byref(out))
        return rc, out.value

    def rows(self, name, query, rows):
        query = np.ascontiguousarray(query, dtype=np.float32)
        rows = np.ascontiguousarray(rows, dtype=np.float32)
        out = np.empty(rows.shape[0], dtype=np.float64)
        rc = self.lib.pgvref32_rows(name.encode(), rows.shape[1], _p(query), _p(rows), C.c_long(rows.shape[0]), _p(out))
        if rc != 0:
            raise RuntimeError(self.last_error())
        return out

    def l2_normalize(self, a):
        a = np.ascontiguousarray(a, dtype=np.float32)
        out = np.empty_like(a)
        rc = self.lib.pgvref32_l2_normalize(len(a), _p(a), _p(out))
        return rc, out

    def last_error(self):
        return self.lib.pgvref32_last_error().decode()
