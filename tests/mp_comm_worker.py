"""Two ranks on ONE GPU (functional run of the library's multi-GPU path): the collectives of pgv_comm
are host callbacks over gloo, everything else is the real C path.  Launched by test_gpu_round2.py
through torch.distributed.run; rank 0 prints 'COMM-OK' when every check passed."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from pgvector_amd import api  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    ctx = api.Context(0, stream=0)
    comm = api.Comm(ctx, backend="host")
    rng = np.random.default_rng(5)
    n, dim, k = 6000, 48, 40
    means = rng.random((k, dim), dtype=np.float32)
    data = (means[rng.integers(0, k, n)] + 0.05 * rng.standard_normal((n, dim))).astype(np.float32)
    per = (n + world - 1) // world
    mine = np.ascontiguousarray(data[rank * per:(rank + 1) * per])

    # k-means: sharded samples, same centers on every rank, as good as the single-GPU run
    centers, closest, iters = comm.kmeans(api.PGV_OPS_L2, api.PGV_F32, dim, mine, k, api.make_rng(seed=9))
    allc = [torch.empty(k, dim) for _ in range(world)]
    dist.all_gather(allc, torch.from_numpy(centers))
    assert all(torch.equal(allc[0], c) for c in allc), "centers differ between the ranks"
    single, sc, siters = api.kmeans(ctx, api.PGV_OPS_L2, api.PGV_F32, dim, data, k, api.make_rng(seed=9))

    def inertia(c):
        d = ((data[:, None, :].astype(np.float64) - c[None, :, :].astype(np.float64)) ** 2).sum(-1)
        return float(d.min(axis=1).sum())
    assert inertia(centers) <= 1.03 * inertia(single), (inertia(centers), inertia(single))
    assert 1 <= iters <= 500
    # the local assignment is the argmin under the final centers
    want, _ = api.assign(ctx, api.PGV_L2SQ, api.PGV_F32, dim, centers, mine)
    assert (want != closest).mean() < 0.01

    # spherical opclass, one rank without samples
    unit = data / np.linalg.norm(data, axis=1, keepdims=True)
    part = np.ascontiguousarray(unit) if rank == 0 else np.zeros((0, dim), np.float32)
    c2, _, it2 = comm.kmeans(api.PGV_OPS_IP, api.PGV_F32, dim, part, k, api.make_rng(seed=3))
    np.testing.assert_allclose(np.linalg.norm(c2.astype(np.float64), axis=1), 1.0, rtol=1e-5)

    # list scan: lists sharded l % world, same answers as the unsharded index
    lists, _ = api.assign(ctx, api.PGV_L2SQ, api.PGV_F32, dim, single, data)
    order = np.argsort(lists, kind="stable")
    off = np.concatenate([[0], np.cumsum(np.bincount(lists, minlength=k))]).astype(np.int64)
    tids = order.astype(np.uint64)
    whole = api.IvfIndex(ctx, api.PGV_L2SQ, api.PGV_F32, dim, single, off, data[order], tids)
    own = (lists[order] % world) == rank
    lens = np.where(np.arange(k) % world == rank, np.diff(off), 0)
    loff = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    local = api.IvfIndex(ctx, api.PGV_L2SQ, api.PGV_F32, dim, single, loff, np.ascontiguousarray(data[order][own]),
                         np.ascontiguousarray(tids[own]))
    queries = (means[rng.integers(0, k, 37)] + 0.05 * rng.standard_normal((37, dim))).astype(np.float32)
    for probes in (1, 4, k):
        gd, gt = comm.search_batch(local, queries, probes, 10)
        wd, _, wt = whole.search_batch(queries, probes, 10, want_tid=True)
        np.testing.assert_allclose(gd, wd, rtol=1e-6)
        for i in range(len(queries)):
            assert sorted(gt[i].tolist()) == sorted(wt[i].tolist()), (probes, i)
    whole.close()
    local.close()
    comm.close()
    ctx.close()
    dist.barrier()
    if rank == 0:
        print("COMM-OK")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
