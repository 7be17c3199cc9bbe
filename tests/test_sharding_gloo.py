"""The N > 1 path on CPU: world_size-2 gloo processes run the same sharding /
merge / all-reduce code bench.py runs on GPUs, with the oracle standing in for
libpgv_hip as the per-rank compute.  Results must equal the single-process run."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_problem():
    from helpers import CpuIvf, gen
    from oracle import pyoracle as po
    ora = po.Oracle()
    data = gen(3000, 16, seed=5, dist="clustered", clusters=12)
    ivf = CpuIvf(ora, po.OPS_L2, po.ORA_F32, data, 12)
    queries = gen(9, 16, seed=6, dist="clustered", clusters=12)
    return ora, po, data, ivf, queries


def _local_search(ora, po, centers, off, vecs, tids, queries, probes, k):
    ix = ora.index_struct(po.OPS_L2, po.ORA_F32, centers, off, vecs, tids)
    d = np.full((len(queries), k), np.inf, dtype=np.float32)
    t = np.full((len(queries), k), -1, dtype=np.int64)
    for i, q in enumerate(queries):
        tt, dd = ora.search(ix, q, probes, k)
        d[i, :len(dd)] = dd
        t[i, :len(tt)] = tt.astype(np.int64)
    return torch.from_numpy(d), torch.from_numpy(t)


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pgvector_amd import sharding
        ora, po, data, ivf, queries = _make_problem()
        # --- scan: lists sharded l % world, local top-k, all-gather merge
        v, t, off = sharding.local_index_arrays(torch.from_numpy(ivf.vectors), torch.from_numpy(ivf.tids.astype(np.int64)),
                                                torch.from_numpy(ivf.list_offsets), rank, world)
        assert int(off[-1]) == v.shape[0]
        ld, lt = _local_search(ora, po, ivf.centers, off.numpy(), v.numpy(), t.numpy().astype(np.uint64), queries, 4, 7)
        md, mt = sharding.merge_topk(ld, lt, 7)
        # --- k-means: samples sharded by row, all-reduce of sums/counts/changes
        samples = torch.from_numpy(data[:1200])
        lo, hi = sharding.row_shard(1200, rank, world)
        init = torch.from_numpy(np.ascontiguousarray(data[:12]))

        def partial(s, c, closest):
            new, _ = ora.lloyd_assign(po.OPS_L2, po.ORA_F32, s.numpy(), c.numpy())
            changes = int((new != closest.numpy()).sum())
            closest.copy_(torch.from_numpy(new))
            sums = np.zeros((12, 16), dtype=np.float32)
            np.add.at(sums, new, s.numpy())
            return torch.from_numpy(sums), torch.from_numpy(np.bincount(new, minlength=12).astype(np.int32)), \
                torch.tensor([changes], dtype=torch.int64)

        def finish(sums, counts, it):
            c = sums.numpy() / np.maximum(counts.numpy(), 1)[:, None].astype(np.float32)
            return torch.from_numpy(c.astype(np.float32))
        centers, closest, iters = sharding.sharded_kmeans(samples[lo:hi].contiguous(), init, partial, finish, 50)
        # --- assignment gather
        mine, _ = ora.assign(po.OPS_L2, po.ORA_F32, centers.numpy(), data[slice(*sharding.row_shard(3000, rank, world))])
        every = sharding.gather_assignments(torch.from_numpy(mine), 3000, world)
        if rank == 0:
            torch.save({"md": md, "mt": mt, "centers": centers, "iters": iters, "lists": every}, out)
    finally:
        dist.destroy_process_group()


def test_world2_matches_single_process(tmp_path):
    out = str(tmp_path / "w2.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    from pgvector_amd import sharding
    ora, po, data, ivf, queries = _make_problem()
    sd, st = _local_search(ora, po, ivf.centers, ivf.list_offsets, ivf.vectors, ivf.tids, queries, 4, 7)
    np.testing.assert_array_equal(got["md"].numpy(), sd.numpy())
    np.testing.assert_array_equal(got["mt"].numpy(), st.numpy())
    # single-process Lloyd from the same init
    centers = np.ascontiguousarray(data[:12])
    prev = None
    for it in range(50):
        new, _ = ora.lloyd_assign(po.OPS_L2, po.ORA_F32, data[:1200], centers)
        sums = np.zeros((12, 16), dtype=np.float32)
        np.add.at(sums, new, data[:1200])
        centers = (sums / np.maximum(np.bincount(new, minlength=12), 1)[:, None].astype(np.float32)).astype(np.float32)
        changed = prev is None or (new != prev).any()
        prev = new
        if not changed and it != 0:
            break
    assert got["iters"] == it + 1
    np.testing.assert_allclose(got["centers"].numpy(), centers, rtol=1e-5, atol=1e-6)
    want, _ = ora.assign(po.OPS_L2, po.ORA_F32, got["centers"].numpy(), data)
    np.testing.assert_array_equal(got["lists"].numpy(), want)


def test_local_index_arrays_partition():
    from pgvector_amd import sharding
    off = torch.tensor([0, 3, 3, 7, 8, 12])
    vec = torch.arange(12, dtype=torch.float32).view(12, 1)
    tid = torch.arange(12)
    seen = []
    for r in range(3):
        v, t, o = sharding.local_index_arrays(vec, tid, off, r, 3)
        assert o.numel() == 6 and int(o[-1]) == v.shape[0]
        for l in range(5):
            n = int(o[l + 1] - o[l])
            assert n == (int(off[l + 1] - off[l]) if l % 3 == r else 0)
        seen += t.tolist()
    assert sorted(seen) == list(range(12))
