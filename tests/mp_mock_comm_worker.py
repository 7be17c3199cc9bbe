"""Two (or more) gloo ranks on the CPU through the C multi-GPU entry points -- pgv_comm_create_custom,
pgv_kmeans_sharded, pgv_search_batch_sharded -- of the stand-in device (tests/c/mock_hip.c, which issues the product's
collectives in the product's order: see its "multi-GPU" section), with the collectives as gloo callbacks, exactly the
way api.Comm(backend="host") hands them to libpgv_hip.  Launched by tests/test_sharded_cpath_gloo.py through
torch.distributed.run; rank 0 prints 'MOCK-COMM-OK'."""
import ctypes as C
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402  (the checker)

ALL_REDUCE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
ALL_GATHER = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)


class Collectives(C.Structure):
    _fields_ = [("all_reduce_sum_f32", ALL_REDUCE), ("all_gather", ALL_GATHER), ("state", C.c_void_p)]


class Rng(C.Structure):
    _fields_ = [("next_double", C.c_void_p), ("next_u32", C.c_void_p), ("state", C.c_void_p), ("seed", C.c_uint64)]


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    lib = C.CDLL(os.environ["PGV_MOCK_LIB"])
    P, I = C.c_void_p, C.c_int
    lib.pgv_last_error.restype = C.c_char_p
    lib.pgv_ctx_create.argtypes = [I, P, C.POINTER(P)]
    lib.pgv_comm_create_custom.argtypes = [P, I, I, C.POINTER(Collectives), C.POINTER(P)]
    lib.pgv_comm_size.argtypes = [P]
    lib.pgv_comm_rank.argtypes = [P]
    lib.pgv_kmeans_sharded.argtypes = [P, I, I, I, P, I, I, I, C.POINTER(Rng), P, P, C.POINTER(I)]
    lib.pgv_kmeans.argtypes = [P, I, I, I, P, I, I, I, C.POINTER(Rng), P, P, C.POINTER(I)]
    lib.pgv_assign.argtypes = [P, I, I, I, P, I, P, C.c_int64, P, P]
    lib.pgv_index_upload.argtypes = [P, I, I, I, I, P, P, P, P, C.POINTER(P)]
    lib.pgv_search_batch.argtypes = [P, P, I, I, I, P, P, P]
    lib.pgv_search_batch_sharded.argtypes = [P, P, P, I, I, I, P, P]
    lib.pgv_index_free.argtypes = [P]
    lib.pgv_comm_destroy.argtypes = [P]
    calls = {"all_reduce": 0, "all_reduce_floats": 0, "all_gather": 0, "all_gather_bytes": 0}

    def all_reduce(_state, buf, count, _stream):
        try:
            t = torch.frombuffer((C.c_float * count).from_address(buf), dtype=torch.float32)
            dist.all_reduce(t)
            calls["all_reduce"] += 1
            calls["all_reduce_floats"] += count
            return 0
        except Exception:  # noqa: BLE001
            return 1

    def all_gather(_state, send, recv, nbytes, _stream):
        try:
            mine = torch.frombuffer((C.c_uint8 * nbytes).from_address(send), dtype=torch.uint8).clone()
            parts = [torch.empty(nbytes, dtype=torch.uint8) for _ in range(world)]
            dist.all_gather(parts, mine)
            full = torch.cat(parts)   # (kept in a name: the buffer must outlive the copy)
            C.memmove(recv, full.data_ptr(), nbytes * world)
            calls["all_gather"] += 1
            calls["all_gather_bytes"] += nbytes
            return 0
        except Exception:  # noqa: BLE001
            return 1

    cbs = (ALL_REDUCE(all_reduce), ALL_GATHER(all_gather))
    coll = Collectives(cbs[0], cbs[1], None)
    ctx, cm = P(), P()
    assert lib.pgv_ctx_create(0, None, C.byref(ctx)) == 0
    assert lib.pgv_comm_create_custom(ctx, world, rank, C.byref(coll), C.byref(cm)) == 0, lib.pgv_last_error()
    assert lib.pgv_comm_size(cm) == world and lib.pgv_comm_rank(cm) == rank

    rng = np.random.default_rng(5)
    n, dim, k = 6000, 24, 30
    means = rng.random((k, dim), dtype=np.float32)
    data = (means[rng.integers(0, k, n)] + 0.05 * rng.standard_normal((n, dim))).astype(np.float32)
    per = (n + world - 1) // world
    mine = np.ascontiguousarray(data[rank * per:(rank + 1) * per])

    def p(a):
        return a.ctypes.data_as(P)

    # ---- k-means: sharded samples, the same centers on every rank, as good as the oracle's ElkanKmeans
    centers = np.zeros((k, dim), np.float32)
    closest = np.zeros(mine.shape[0], np.int32)
    iters = I()
    seed = Rng(None, None, None, 9)
    assert lib.pgv_kmeans_sharded(cm, 0, 0, dim, p(mine), mine.shape[0], k, 500, C.byref(seed), p(centers), p(closest),
                                  C.byref(iters)) == 0, lib.pgv_last_error()
    allc = [torch.empty(k, dim) for _ in range(world)]
    dist.all_gather(allc, torch.from_numpy(centers))
    assert all(torch.equal(allc[0], c) for c in allc), "centers differ between the ranks"
    # per Lloyd iteration ONE all-reduce of sums | counts | changes: k * dim + k + 1 floats
    assert calls["all_reduce"] == iters.value and calls["all_reduce_floats"] == iters.value * (k * dim + k + 1), calls
    # sample counts once, then per k-means++ round the weight totals and the candidate row
    assert calls["all_gather"] == 1 + 1 + 2 * (k - 1), calls
    ora = po.Oracle()
    want_c, _, _ = ora.kmeans(po.OPS_L2, po.ORA_F32, data, k, ora.prng(9))

    def inertia(c):
        d = ((data[:, None, :].astype(np.float64) - np.asarray(c)[None, :, :].astype(np.float64)) ** 2).sum(-1)
        return float(d.min(axis=1).sum())
    # the same call on ONE rank holding every sample, same seed: the same draws pick the same k-means++ centers, the
    # sums differ in their last bits only (partial sums per rank, then the all-reduce)
    solo, one, its = P(), np.zeros((k, dim), np.float32), I()
    assert lib.pgv_comm_create_custom(ctx, 1, 0, None, C.byref(solo)) == 0
    seed1 = Rng(None, None, None, 9)
    assert lib.pgv_kmeans_sharded(solo, 0, 0, dim, p(data), n, k, 500, C.byref(seed1), p(one), None, C.byref(its)) == 0
    lib.pgv_comm_destroy(solo)
    assert inertia(centers) <= 1.01 * inertia(one), (inertia(centers), inertia(one))
    assert inertia(centers) <= 1.6 * inertia(want_c), (inertia(centers), inertia(want_c))   # (another local optimum than Elkan's)
    assert 1 <= iters.value <= 500
    # the local assignment is the oracle's argmin under the final centers
    want_l, _ = ora.assign(po.OPS_L2, po.ORA_F32, centers, mine)
    assert (np.asarray(want_l) != closest).mean() < 0.01

    # ---- a rank without samples
    part = np.ascontiguousarray(data) if rank == 0 else np.zeros((0, dim), np.float32)
    c2 = np.zeros((k, dim), np.float32)
    it2 = I()
    seed2 = Rng(None, None, None, 3)
    assert lib.pgv_kmeans_sharded(cm, 0, 0, dim, p(part), part.shape[0], k, 500, C.byref(seed2), p(c2), None, C.byref(it2)) == 0
    assert inertia(c2) <= 1.6 * inertia(want_c)

    # ---- list scan: lists sharded l % world, the same answers as the unsharded index and as the oracle
    lists, _ = ora.assign(po.OPS_L2, po.ORA_F32, centers, data)
    lists = np.asarray(lists)
    order = np.argsort(lists, kind="stable")
    off = np.concatenate([[0], np.cumsum(np.bincount(lists, minlength=k))]).astype(np.int64)
    tids = order.astype(np.uint64)
    rows = np.ascontiguousarray(data[order])
    whole = P()
    assert lib.pgv_index_upload(ctx, 0, 0, dim, k, p(centers), p(off), p(rows), p(tids), C.byref(whole)) == 0
    own = (lists[order] % world) == rank
    lens = np.where(np.arange(k) % world == rank, np.diff(off), 0)
    loff = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    lrows, ltids = np.ascontiguousarray(rows[own]), np.ascontiguousarray(tids[own])
    local = P()
    assert lib.pgv_index_upload(ctx, 0, 0, dim, k, p(centers), p(loff), p(lrows), p(ltids), C.byref(local)) == 0
    nq, kk = 37, 10
    queries = (means[rng.integers(0, k, nq)] + 0.05 * rng.standard_normal((nq, dim))).astype(np.float32)
    ix = ora.index_struct(po.OPS_L2, po.ORA_F32, centers, off, rows, tids)
    for probes in (1, 4, k):
        gd, gt = np.zeros((nq, kk), np.float32), np.zeros((nq, kk), np.uint64)
        before = calls["all_gather"]
        assert lib.pgv_search_batch_sharded(cm, local, p(queries), nq, probes, kk, p(gd), p(gt)) == 0, lib.pgv_last_error()
        assert calls["all_gather"] - before == 3          # probe-list slices, head distances, head tids
        wd, wt = np.zeros((nq, kk), np.float32), np.zeros((nq, kk), np.uint64)
        assert lib.pgv_search_batch(whole, p(queries), nq, probes, kk, p(wd), None, p(wt)) == 0
        np.testing.assert_allclose(gd, wd, rtol=1e-6)
        for i in range(nq):
            assert sorted(gt[i].tolist()) == sorted(wt[i].tolist()), (probes, i)
            ot, od = ora.search(ix, queries[i], probes, kk)
            np.testing.assert_allclose(gd[i][:len(od)], od, rtol=1e-4, atol=1e-6)
    lib.pgv_index_free(whole)
    lib.pgv_index_free(local)
    lib.pgv_comm_destroy(cm)
    dist.barrier()
    if rank == 0:
        print("MOCK-COMM-OK")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
