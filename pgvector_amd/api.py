"""Thin Python face of the C ABI, for the test/bench harness.

Buffers may be numpy arrays (host memory) or torch tensors (host or HBM); the
library itself decides per pointer whether to stage it.  Nothing here computes
a distance: every function forwards to libpgv_hip.so.
"""
import ctypes as C
import weakref

import numpy as np

from . import _lib
from ._lib import (PGV_ERR_ARG, PGV_ERR_STATE, PGV_F16, PGV_F32, PGV_L1, PGV_L2SQ, PGV_NEG_IP, PGV_OPS_COSINE,  # noqa: F401
                   PGV_OPS_IP, PGV_OPS_L2, PgvError, PgvRng, PgvStats, check, lib)

_NP_OF = {PGV_F32: np.float32, PGV_F16: np.float16}


def _is_torch(x):
    return type(x).__module__.startswith("torch")


def ptr(x):
    """raw address of a numpy array / torch tensor (None -> NULL)"""
    if x is None:
        return None
    if _is_torch(x):
        assert x.is_contiguous(), "tensor must be contiguous"
        return C.c_void_p(x.data_ptr())
    assert isinstance(x, np.ndarray) and x.flags["C_CONTIGUOUS"], "need a C-contiguous array"
    return C.c_void_p(x.ctypes.data)


def _on_device(x):
    return _is_torch(x) and x.is_cuda


def _empty_like_kind(ref, shape, np_dtype):
    """output buffer living where `ref` lives"""
    if _on_device(ref):
        import torch
        return torch.empty(shape, dtype=getattr(torch, np.dtype(np_dtype).name), device=ref.device)
    return np.empty(shape, dtype=np_dtype)


def as_dtype(x, dtype):
    """host arrays: make sure element type and layout match what the ABI reads"""
    if _is_torch(x):
        return x.contiguous()
    return np.ascontiguousarray(x, dtype=_NP_OF[dtype])


def make_rng(seed=0, next_double=None, next_u32=None, state=None):
    """pgv_rng: the library's own generator (seed) or caller callbacks (C function pointers)"""
    r = PgvRng()
    r.next_double = C.cast(next_double, C.c_void_p) if next_double is not None else None
    r.next_u32 = C.cast(next_u32, C.c_void_p) if next_u32 is not None else None
    r.state = state
    r.seed = seed
    return r


class Context:
    def __init__(self, device=0, stream=None):
        """stream: None -> a private stream; an int hipStream_t handle -> enqueue on that stream
        (0, the handle of the default stream, is passed as PGV_DEFAULT_STREAM)"""
        h = C.c_void_p()
        if stream is None:
            arg = None
        elif stream == 0:
            arg = C.c_void_p(-1)  # PGV_DEFAULT_STREAM
        else:
            arg = C.c_void_p(stream)
        check(lib.pgv_ctx_create(device, arg, C.byref(h)))
        self.h = h
        self.device = device
        self._children = []  # weakrefs: index/hnsw handles must be freed before their context

    def _adopt(self, child):
        self._children.append(weakref.ref(child))

    def close(self):
        if self.h:
            for ref in self._children:
                child = ref()
                if child is not None:
                    child.close()
            self._children = []
            lib.pgv_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        check(lib.pgv_ctx_sync(self.h))

    def stream(self):
        return lib.pgv_ctx_stream(self.h)

    def timer_start(self):
        check(lib.pgv_timer_start(self.h))

    def timer_stop(self):
        ms = C.c_float()
        check(lib.pgv_timer_stop(self.h, C.byref(ms)))
        return ms.value

    def set_profiling(self, on):
        check(lib.pgv_ctx_set_profiling(self.h, 1 if on else 0))

    def set_exact_scan(self, on):
        """Keep batched L2 scans on the exact vector-ALU kernels (no matrix-core pre-filter)."""
        check(lib.pgv_ctx_set_exact_scan(self.h, 1 if on else 0))

    def set_bound(self, worst_case):
        """pgv_ctx_set_bound: the rounding bound the MFMA L2 paths prove completeness with (statistical / worst case)"""
        check(lib.pgv_ctx_set_bound(self.h, _lib.PGV_BOUND_WORST_CASE if worst_case else _lib.PGV_BOUND_STATISTICAL))

    def reset_stats(self):
        check(lib.pgv_ctx_reset_stats(self.h))

    def stats(self):
        s = PgvStats()
        check(lib.pgv_ctx_get_stats(self.h, C.byref(s)))
        return {"scan_ms": s.scan_ms, "scan_launches": s.scan_launches,
                "scan_pairs": s.scan_pairs, "scan_rows": s.scan_rows,
                "aux_ms": s.aux_ms, "aux_launches": s.aux_launches, "aux_pairs": s.aux_pairs,
                "assign_redo_rows": s.assign_redo_rows, "assign_rows": s.assign_rows,
                "assign_recheck_rows": s.assign_recheck_rows, "scan_unique_rows": s.scan_unique_rows, "scan_redo_queries": s.scan_redo_queries,
                "scan_widened_queries": s.scan_widened_queries}


class IvfIndex:
    """device mirror of one IVFFlat index (pgv_index_upload)"""

    def __init__(self, ctx, metric, dtype, dim, centers, list_offsets, vectors, tids=None):
        self.ctx, self.metric, self.dtype, self.dim = ctx, metric, dtype, dim
        centers = as_dtype(centers, dtype)
        vectors = as_dtype(vectors, dtype)
        if not _is_torch(list_offsets):
            list_offsets = np.ascontiguousarray(list_offsets, dtype=np.int64)
        if tids is not None and not _is_torch(tids):
            tids = np.ascontiguousarray(tids, dtype=np.uint64)
        self.nlists = int(list_offsets.shape[0]) - 1
        h = C.c_void_p()
        check(lib.pgv_index_upload(ctx.h, metric, dtype, dim, self.nlists, ptr(centers),
                                   ptr(list_offsets), ptr(vectors), ptr(tids), C.byref(h)))
        self.h = h
        ctx._adopt(self)

    def set_overlap(self, lanes):
        """pgv_index_set_overlap: consecutive search_batch calls run on `lanes` internal streams in turn (one batch's
        ranking / planning / top-k under the other's scan); device outputs are complete after ctx.sync().  1 = off."""
        check(lib.pgv_index_set_overlap(self.h, int(lanes)))

    def share(self, ctx):
        """a second handle for another context (its own stream and scratch) on the same device: pgv_index_share.
        The device arrays go with the last handle that is closed."""
        v = IvfIndex.__new__(IvfIndex)
        v.ctx, v.metric, v.dtype, v.dim, v.nlists = ctx, self.metric, self.dtype, self.dim, self.nlists
        h = C.c_void_p()
        check(lib.pgv_index_share(self.h, ctx.h, C.byref(h)))
        v.h = h
        ctx._adopt(v)
        return v

    def export(self):
        """pgv_index_export: the 256-byte handle another PROCESS imports (pgv_index_import) to scan this mirror"""
        buf = C.create_string_buffer(256)
        check(lib.pgv_index_export(self.h, buf))
        return bytes(buf.raw)

    @classmethod
    def from_handle(cls, ctx, handle, metric=None, dtype=None, dim=None):
        """pgv_index_import: map a mirror another process exported (no copy)"""
        v = cls.__new__(cls)
        h = C.c_void_p()
        buf = C.create_string_buffer(bytes(handle), 256)
        check(lib.pgv_index_import(ctx.h, buf, C.byref(h)))
        v.ctx, v.metric, v.dtype, v.dim, v.h = ctx, metric, dtype, dim, h
        v.nlists = lib.pgv_index_lists(h)
        ctx._adopt(v)
        return v

    def close(self):
        if self.h:
            lib.pgv_index_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def rows(self):
        return lib.pgv_index_rows(self.h)

    def tids(self, slots):
        """pgv_index_tids: heap TIDs of row slots, for a caller without a TID table (a backend that imported)"""
        slots = np.ascontiguousarray(slots, dtype=np.int64)
        out = np.empty(slots.shape, dtype=np.uint64)
        check(lib.pgv_index_tids(self.h, ptr(slots), int(slots.size), ptr(out)))
        return out

    def rank_lists(self, queries, maxprobes, want_dist=True):
        queries = as_dtype(queries, self.dtype)
        nq = int(queries.shape[0])
        lists = _empty_like_kind(queries, (nq, maxprobes), np.int32)
        dist = _empty_like_kind(queries, (nq, maxprobes), np.float32) if want_dist else None
        check(lib.pgv_rank_lists(self.h, ptr(queries), nq, maxprobes, ptr(lists), ptr(dist)))
        return lists, dist

    def scan_lists(self, query, lists):
        if query is not None:
            query = as_dtype(query, self.dtype)
        lists = np.ascontiguousarray(lists, dtype=np.int32)
        count = C.c_int64()
        # first call sizes the output (capacity 0 only reports the count)
        rc = lib.pgv_scan_lists(self.h, ptr(query), ptr(lists), len(lists), None, None, 0, C.byref(count))
        if rc != _lib.PGV_OK and count.value == 0:
            check(rc)
        m = count.value
        dist = np.empty(m, dtype=np.float32)
        slot = np.empty(m, dtype=np.int64)
        if m:
            check(lib.pgv_scan_lists(self.h, ptr(query), ptr(lists), len(lists), ptr(dist), ptr(slot),
                                     m, C.byref(count)))
        return dist, slot

    def search_batch(self, queries, probes, k, want_tid=False, out=None):
        queries = as_dtype(queries, self.dtype)
        nq = int(queries.shape[0])
        if out is None:
            dist = _empty_like_kind(queries, (nq, k), np.float32)
            slot = _empty_like_kind(queries, (nq, k), np.int64)
            tid = _empty_like_kind(queries, (nq, k), np.uint64 if not _on_device(queries) else np.int64) \
                if want_tid else None
        else:
            dist, slot, tid = out
        check(lib.pgv_search_batch(self.h, ptr(queries), nq, probes, k, ptr(dist), ptr(slot), ptr(tid)))
        return dist, slot, tid


def _scan_batch(self, queries, probe_lists, k, want_tid=False, out=None):
    """GetScanItems + sorted head for probe lists chosen elsewhere (pgv_scan_batch)"""
    queries = as_dtype(queries, self.dtype)
    nq = int(queries.shape[0])
    if not _is_torch(probe_lists):
        probe_lists = np.ascontiguousarray(probe_lists, dtype=np.int32)
    probes = int(probe_lists.shape[1])
    if out is None:
        dist = _empty_like_kind(queries, (nq, k), np.float32)
        slot = _empty_like_kind(queries, (nq, k), np.int64)
        tid = _empty_like_kind(queries, (nq, k), np.uint64 if not _on_device(queries) else np.int64) \
            if want_tid else None
    else:
        dist, slot, tid = out
    check(lib.pgv_scan_batch(self.h, ptr(queries), nq, ptr(probe_lists), probes, k, ptr(dist), ptr(slot),
                             ptr(tid)))
    return dist, slot, tid


IvfIndex.scan_batch = _scan_batch


class Query:
    """one backend's index scan, device-resident between calls (pgv_query_*): rank() = GetScanLists,
    scan() = GetScanItems + the sorted head, more() = deeper into the same batch's sorted stream"""

    def __init__(self, index):
        self.index = index
        h = C.c_void_p()
        check(lib.pgv_query_begin(index.h, C.byref(h)))
        self.h = h
        index.ctx._adopt(self)

    def close(self):
        if self.h:
            lib.pgv_query_end(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def rank(self, query, max_probes):
        if query is not None:
            query = as_dtype(query, self.index.dtype)
        self._q = query
        check(lib.pgv_query_rank(self.h, ptr(query), int(max_probes)))

    def lists(self, n):
        out = np.empty(n, dtype=np.int32)
        check(lib.pgv_query_lists(self.h, ptr(out), n))
        return out

    def scan(self, first, nprobes, head, want_tid=True):
        dist = np.empty(head, dtype=np.float32)
        slot = np.empty(head, dtype=np.int64)
        tid = np.empty(head, dtype=np.uint64) if want_tid else None
        count, total = C.c_int(), C.c_int64()
        check(lib.pgv_query_scan(self.h, int(first), int(nprobes), int(head), ptr(dist), ptr(slot), ptr(tid),
                                 C.byref(count), C.byref(total)))
        n = count.value
        return dist[:n], slot[:n], (tid[:n] if want_tid else None), total.value

    def more(self, skip, count, want_tid=True):
        dist = np.empty(count, dtype=np.float32)
        slot = np.empty(count, dtype=np.int64)
        tid = np.empty(count, dtype=np.uint64) if want_tid else None
        got = C.c_int()
        check(lib.pgv_query_more(self.h, int(skip), int(count), ptr(dist), ptr(slot), ptr(tid), C.byref(got)))
        n = got.value
        return dist[:n], slot[:n], (tid[:n] if want_tid else None)


class Comm:
    """one process per GPU: the library's communicator (pgv_comm_*).  backend "rccl": RCCL over xGMI, the
    group id travels through torch.distributed once; backend "host": the two collectives are callbacks into
    torch.distributed -- through host memory under gloo (functional runs only), through device staging tensors under
    torch's nccl backend (RCCL driven by torch: what bench.py falls back to when pgv_comm_create fails)."""

    def __init__(self, ctx, backend="rccl"):
        import torch
        import torch.distributed as dist
        self.ctx = ctx
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        h = C.c_void_p()
        if backend == "rccl":
            ident = torch.zeros(128, dtype=torch.uint8)
            if self.world > 1:
                if self.rank == 0:
                    buf = (C.c_uint8 * 128)()
                    check(lib.pgv_comm_unique_id(buf))
                    ident = torch.tensor(list(buf), dtype=torch.uint8)
                on_dev = dist.get_backend() == "nccl"
                t = ident.cuda() if on_dev else ident
                dist.broadcast(t, 0)
                ident = t.cpu()
            if self.world == 1:  # a group of one: still through RCCL, which exercises the plumbing on one GPU
                buf = (C.c_uint8 * 128)()
                check(lib.pgv_comm_unique_id(buf))
                ident = torch.tensor(list(buf), dtype=torch.uint8)
            raw = (C.c_uint8 * 128)(*ident.tolist())
            check(lib.pgv_comm_create(ctx.h, self.world, self.rank, raw, C.byref(h)))
        else:
            hip = C.CDLL("libamdhip64.so")
            hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
            hip.hipStreamSynchronize.argtypes = [C.c_void_p]
            world = self.world
            # the staging tensors live where torch.distributed's backend wants them: host memory for gloo, this
            # device for nccl (= RCCL driven by torch: the fallback when pgv_comm_create itself cannot be had)
            on_dev = dist.is_initialized() and dist.get_backend() == "nccl"
            where = torch.device("cuda", ctx.device) if on_dev else torch.device("cpu")
            to_stage, from_stage = (3, 3) if on_dev else (2, 1)   # hipMemcpyKind: D2D / D2H, H2D

            def all_reduce(_state, buf, count, stream):
                try:
                    hip.hipStreamSynchronize(stream)
                    stage = torch.empty(count, dtype=torch.float32, device=where)
                    hip.hipMemcpy(stage.data_ptr(), buf, count * 4, to_stage)
                    dist.all_reduce(stage)
                    if on_dev:
                        torch.cuda.synchronize(where)
                    hip.hipMemcpy(buf, stage.data_ptr(), count * 4, from_stage)
                    return 0
                except Exception:
                    return 1

            def all_gather(_state, send, recv, nbytes, stream):
                try:
                    hip.hipStreamSynchronize(stream)
                    mine = torch.empty(nbytes, dtype=torch.uint8, device=where)
                    hip.hipMemcpy(mine.data_ptr(), send, nbytes, to_stage)
                    full = torch.empty(nbytes * world, dtype=torch.uint8, device=where)
                    dist.all_gather_into_tensor(full, mine) if on_dev else dist.all_gather(list(full.split(nbytes)), mine)
                    if on_dev:
                        torch.cuda.synchronize(where)
                    hip.hipMemcpy(recv, full.data_ptr(), nbytes * world, from_stage)
                    return 0
                except Exception:
                    return 1
            self._cbs = (_lib.ALL_REDUCE_F32(all_reduce), _lib.ALL_GATHER(all_gather))
            coll = _lib.PgvCollectives(self._cbs[0], self._cbs[1], None)
            self._coll = coll
            check(lib.pgv_comm_create_custom(ctx.h, self.world, self.rank, C.byref(coll), C.byref(h)))
        self.h = h
        ctx._adopt(self)

    def close(self):
        if self.h:
            lib.pgv_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def kmeans(self, ops, dtype, dim, samples_local, k, rng=None, max_iterations=500, want_closest=True):
        samples_local = as_dtype(samples_local, dtype)
        n = int(samples_local.shape[0])
        centers = _empty_like_kind(samples_local, (k, dim), _NP_OF[dtype])
        closest = _empty_like_kind(samples_local, (n,), np.int32) if want_closest and n else None
        iters = C.c_int()
        check(lib.pgv_kmeans_sharded(self.h, ops, dtype, dim, ptr(samples_local) if n else None, n, k, max_iterations,
                                     C.byref(rng) if rng is not None else None, ptr(centers), ptr(closest),
                                     C.byref(iters)))
        return centers, closest, iters.value

    def search_batch(self, index, queries, probes, k, out=None):
        queries = as_dtype(queries, index.dtype)
        nq = int(queries.shape[0])
        if out is None:
            dist_ = _empty_like_kind(queries, (nq, k), np.float32)
            tid = _empty_like_kind(queries, (nq, k), np.uint64 if not _on_device(queries) else np.int64)
        else:
            dist_, tid = out
        check(lib.pgv_search_batch_sharded(self.h, index.h, ptr(queries), nq, probes, k, ptr(dist_), ptr(tid)))
        return dist_, tid


def assign(ctx, metric, dtype, dim, centers, rows, want_dist=True):
    centers = as_dtype(centers, dtype)
    rows = as_dtype(rows, dtype)
    n = int(rows.shape[0])
    out = _empty_like_kind(rows, (n,), np.int32)
    dist = _empty_like_kind(rows, (n,), np.float32) if want_dist else None
    check(lib.pgv_assign(ctx.h, metric, dtype, dim, ptr(centers), int(centers.shape[0]), ptr(rows), n,
                         ptr(out), ptr(dist)))
    return out, dist


def distance_batch(ctx, metric, dtype, dim, query, rows):
    query = as_dtype(query, dtype)
    rows = as_dtype(rows, dtype)
    n = int(rows.shape[0])
    out = _empty_like_kind(rows, (n,), np.float32)
    check(lib.pgv_distance_batch(ctx.h, metric, dtype, dim, ptr(query), ptr(rows), n, ptr(out)))
    return out


def exact_topk(ctx, metric, dtype, dim, queries, rows, k, out=None):
    """pgv_exact_topk: the index-less `ORDER BY <op> LIMIT k` for a batch of queries -> (dist [nq x k], idx [nq x k])"""
    queries = as_dtype(queries, dtype)
    rows = as_dtype(rows, dtype)
    nq, n = int(queries.shape[0]), int(rows.shape[0])
    dist, idx = out if out is not None else (_empty_like_kind(queries, (nq, k), np.float32),
                                            _empty_like_kind(queries, (nq, k), np.int64))
    check(lib.pgv_exact_topk(ctx.h, metric, dtype, dim, ptr(queries), nq, ptr(rows) if n else None, n, int(k),
                             ptr(dist), ptr(idx)))
    return dist, idx


def kmeans(ctx, ops, dtype, dim, samples, k, rng=None, max_iterations=500, want_closest=True):
    samples = as_dtype(samples, dtype)
    n = int(samples.shape[0])
    centers = _empty_like_kind(samples, (k, dim), _NP_OF[dtype])
    closest = _empty_like_kind(samples, (n,), np.int32) if want_closest and n else None
    iters = C.c_int()
    check(lib.pgv_kmeans(ctx.h, ops, dtype, dim, ptr(samples) if n else None, n, k, max_iterations,
                         C.byref(rng) if rng is not None else None, ptr(centers), ptr(closest),
                         C.byref(iters)))
    return centers, closest, iters.value


def kmeanspp_init(ctx, ops, dtype, dim, samples, k, rng=None):
    samples = as_dtype(samples, dtype)
    centers = _empty_like_kind(samples, (k, dim), _NP_OF[dtype])
    check(lib.pgv_kmeanspp_init(ctx.h, ops, dtype, dim, ptr(samples), int(samples.shape[0]), k,
                                C.byref(rng) if rng is not None else None, ptr(centers)))
    return centers


def lloyd_partial(ctx, ops, dtype, dim, samples, centers, closest):
    """closest is updated in place; returns (sums [k x dim] fp32, counts [k], changes)"""
    samples = as_dtype(samples, dtype)
    centers = as_dtype(centers, dtype)
    k = int(centers.shape[0])
    n = int(samples.shape[0])
    sums = _empty_like_kind(samples, (k, dim), np.float32)
    counts = _empty_like_kind(samples, (k,), np.int32)
    changes = _empty_like_kind(samples, (1,), np.int64)
    check(lib.pgv_lloyd_partial(ctx.h, ops, dtype, dim, ptr(samples), n, ptr(centers), k, ptr(closest),
                                ptr(sums), ptr(counts), ptr(changes)))
    return sums, counts, changes


def lloyd_finish(ctx, ops, dtype, dim, sums, counts, rng=None, like=None):
    k = int(counts.shape[0])
    centers = _empty_like_kind(like if like is not None else sums, (k, dim), _NP_OF[dtype])
    check(lib.pgv_lloyd_finish(ctx.h, ops, dtype, dim, k, ptr(sums), ptr(counts),
                               C.byref(rng) if rng is not None else None, ptr(centers)))
    return centers


def cosine_distance_batch(ctx, dtype, dim, query, rows):
    """cosine_distance of one query against n rows, float8 like the operator (pgv_cosine_distance_batch)"""
    query, rows = as_dtype(query, dtype), as_dtype(rows, dtype)
    n = int(rows.shape[0])
    out = _empty_like_kind(rows, (n,), np.float64)
    check(lib.pgv_cosine_distance_batch(ctx.h, dtype, dim, ptr(query), ptr(rows), n, ptr(out)))
    return out


PGV_BIT_HAMMING, PGV_BIT_JACCARD = 0, 1


def bit_distance_batch(ctx, metric, nbits, query, rows):
    """hamming / jaccard distance of one packed bit string against n packed rows [n x (nbits + 7) // 8] uint8"""
    if not _is_torch(rows):
        rows = np.ascontiguousarray(rows, dtype=np.uint8)
        query = np.ascontiguousarray(query, dtype=np.uint8)
    n = int(rows.shape[0])
    out = _empty_like_kind(rows, (n,), np.float64)
    check(lib.pgv_bit_distance_batch(ctx.h, metric, nbits, ptr(query) if nbits else None, ptr(rows), n, ptr(out)))
    return out


class IvfBuilder:
    """pgv_builder_*: heap rows assigned and kept on the device, finish() = the tuplesort by list as a device gather
    whose result is the mirror itself"""

    def __init__(self, ctx, metric, dtype, dim, centers, expected_rows=0, nlists=None):
        """centers None (and nlists given): they come with set_centers(); add() only uploads until then"""
        self.ctx, self.metric, self.dtype, self.dim = ctx, metric, dtype, dim
        if centers is not None:
            centers = as_dtype(centers, dtype)
            nlists = int(centers.shape[0])
        self.nlists = int(nlists)
        h = C.c_void_p()
        check(lib.pgv_builder_begin(ctx.h, metric, dtype, dim, self.nlists, ptr(centers), int(expected_rows), C.byref(h)))
        self.h = h

    def set_centers(self, centers):
        centers = as_dtype(centers, self.dtype)
        if int(centers.shape[0]) != self.nlists:
            raise ValueError("set_centers: %d centers for %d lists" % (centers.shape[0], self.nlists))
        check(lib.pgv_builder_set_centers(self.h, ptr(centers)))

    def add(self, rows, tids=None):
        rows = as_dtype(rows, self.dtype)
        if tids is not None and not _is_torch(tids):
            tids = np.ascontiguousarray(tids, dtype=np.uint64)
        check(lib.pgv_builder_add(self.h, ptr(rows), ptr(tids), int(rows.shape[0])))

    @property
    def rows(self):
        return lib.pgv_builder_rows(self.h)

    def finish(self, want_lists=False):
        """-> (IvfIndex, list_offsets [lists + 1], lists [rows, heap order] or None)"""
        n = self.rows
        off = np.empty(self.nlists + 1, dtype=np.int64)
        lists = np.empty(n, dtype=np.int32) if want_lists else None
        h = C.c_void_p()
        check(lib.pgv_builder_finish(self.h, C.byref(h), ptr(off), ptr(lists)))
        ix = IvfIndex.__new__(IvfIndex)
        ix.ctx, ix.metric, ix.dtype, ix.dim, ix.h = self.ctx, self.metric, self.dtype, self.dim, h
        ix.nlists = self.nlists
        self.ctx._adopt(ix)
        return ix, off, lists

    def close(self):
        if self.h:
            lib.pgv_builder_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_SINK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p)


def drain_index(index, chunk_rows=0):
    """pgv_index_drain -> (vectors [n x dim], tids [n]) in list-major order (host copies; a test helper)"""
    n = index.rows
    npdt = _NP_OF[index.dtype]
    vec = np.empty((n, index.dim), dtype=npdt)
    tids = np.zeros(n, dtype=np.uint64)
    row_bytes = index.dim * vec.itemsize
    seen = []

    def sink(_arg, first, count, vptr, tptr):
        C.memmove(vec.ctypes.data + first * row_bytes, vptr, count * row_bytes)
        if tptr:
            C.memmove(tids.ctypes.data + first * 8, tptr, count * 8)
        seen.append((first, count))
        return 0
    cb = _SINK(sink)
    check(lib.pgv_index_drain(index.h, int(chunk_rows), cb, None))
    return vec, tids, seen


class Hnsw:
    """device mirror of an HNSW index's element vectors (pgv_hnsw_upload)"""

    def __init__(self, ctx, metric, dtype, dim, elements, payload=None):
        """payload: [n x w] uint32 per-element words kept next to the vectors (pgv_hnsw_upload_payload)"""
        self.ctx, self.metric, self.dtype, self.dim = ctx, metric, dtype, dim
        elements = as_dtype(elements, dtype)
        h = C.c_void_p()
        if payload is None:
            check(lib.pgv_hnsw_upload(ctx.h, metric, dtype, dim, ptr(elements), int(elements.shape[0]),
                                      C.byref(h)))
            self.payload_words = 0
        else:
            payload = np.ascontiguousarray(payload, dtype=np.uint32)
            self.payload_words = int(payload.shape[1])
            check(lib.pgv_hnsw_upload_payload(ctx.h, metric, dtype, dim, ptr(elements), int(elements.shape[0]),
                                              ptr(payload), 4 * self.payload_words, C.byref(h)))
        self.h = h
        ctx._adopt(self)

    def get_payload(self, elements, words=None):
        """pgv_hnsw_get_payload: the payload rows of result elements -> [n x w] uint32 (zeros for slots < 0)"""
        elements = np.ascontiguousarray(elements, dtype=np.int64).ravel()
        w = words or self.payload_words
        out = np.empty((elements.size, w), dtype=np.uint32)
        check(lib.pgv_hnsw_get_payload(self.h, ptr(elements), int(elements.size), ptr(out)))
        return out

    def close(self):
        if self.h:
            lib.pgv_hnsw_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def export(self):
        """pgv_hnsw_export: the handle another PROCESS imports (pgv_hnsw_import) to search this mirror"""
        buf = C.create_string_buffer(256)
        check(lib.pgv_hnsw_export(self.h, buf))
        return bytes(buf.raw)

    @classmethod
    def from_handle(cls, ctx, handle, dtype):
        """pgv_hnsw_import: map the elements and the graph another process exported (no copy, read-only)"""
        v = cls.__new__(cls)
        h = C.c_void_p()
        buf = C.create_string_buffer(bytes(handle), 256)
        check(lib.pgv_hnsw_import(ctx.h, buf, C.byref(h)))
        v.ctx, v.metric, v.dtype, v.dim, v.h = ctx, None, dtype, None, h
        v.payload_words = 0
        ctx._adopt(v)
        return v

    def share(self, ctx):
        """pgv_hnsw_share: a view of this mirror for another context of the same process (own stream and scratch; the
        vectors and the graph stay this mirror's).  Close the view before the mirror."""
        v = Hnsw.__new__(Hnsw)
        h = C.c_void_p()
        check(lib.pgv_hnsw_share(self.h, ctx.h, C.byref(h)))
        v.ctx, v.metric, v.dtype, v.dim, v.h = ctx, self.metric, self.dtype, self.dim, h
        v.payload_words = self.payload_words
        v._owner = self   # keeps the owner alive
        ctx._adopt(v)
        return v

    def update_graph(self, entry, elements, tuple_offsets, tuples):
        """pgv_hnsw_update_graph: new entry point and the rewritten neighbor tuples of `elements` (tuple i =
        tuples[tuple_offsets[i] : tuple_offsets[i + 1]]); through a view the patch lands in the owner's graph"""
        elements = np.ascontiguousarray(elements, dtype=np.int32)
        tuple_offsets = np.ascontiguousarray(tuple_offsets, dtype=np.int64)
        tuples = np.ascontiguousarray(tuples, dtype=np.int32)
        check(lib.pgv_hnsw_update_graph(self.h, int(entry), ptr(elements), int(elements.size), ptr(tuple_offsets),
                                        ptr(tuples)))

    def set_graph(self, m, entry, levels, nbr_start, nbr):
        """the graph a scan walks (pgv_hnsw_set_graph): per element slot its level and neighbor tuple"""
        levels = np.ascontiguousarray(levels, dtype=np.int32) if not _is_torch(levels) else levels
        nbr_start = np.ascontiguousarray(nbr_start, dtype=np.int64) if not _is_torch(nbr_start) else nbr_start
        nbr = np.ascontiguousarray(nbr, dtype=np.int32) if not _is_torch(nbr) else nbr
        check(lib.pgv_hnsw_set_graph(self.h, int(m), int(entry), ptr(levels), ptr(nbr_start), ptr(nbr)))

    def search(self, queries, ef_search, k, want_scored=True):
        """hnswgettuple's first batch on the device (pgv_hnsw_search) ->
        (element slots [nq x k] nearest first / -1, distances [nq x k] / +inf, scored [nq] or None)"""
        queries = as_dtype(queries, self.dtype)
        nq = int(queries.shape[0])
        elem = _empty_like_kind(queries, (nq, k), np.int64)
        dist = _empty_like_kind(queries, (nq, k), np.float32)
        scored = _empty_like_kind(queries, (nq,), np.int64) if want_scored else None
        check(lib.pgv_hnsw_search(self.h, ptr(queries), nq, int(ef_search), int(k), ptr(elem), ptr(dist),
                                  ptr(scored)))
        return elem, dist, scored

    def score(self, queries, slot, query_of=None):
        queries = as_dtype(queries, self.dtype)
        slot = np.ascontiguousarray(slot, dtype=np.int32) if not _is_torch(slot) else slot
        if query_of is not None and not _is_torch(query_of):
            query_of = np.ascontiguousarray(query_of, dtype=np.int32)
        n = int(slot.shape[0])
        out = _empty_like_kind(slot, (n,), np.float32)
        check(lib.pgv_hnsw_score(self.h, ptr(queries), int(queries.shape[0]), ptr(slot), ptr(query_of),
                                 n, ptr(out)))
        return out
