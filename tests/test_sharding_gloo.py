"""The harness side of the N > 1 path on CPU: world_size-2 gloo processes run the bookkeeping bench.py runs on GPUs
(row shards -> assignment -> rows to the owners of their lists -> local images; exact ground truth merged across
ranks), with the oracle standing in for libpgv_hip as the per-rank compute.  Results must equal the single-process
run.  The path's own collectives (pgv_kmeans_sharded, pgv_search_batch_sharded) are HIP code inside libpgv_hip:
their 2-rank test needs a GPU (tests/test_gpu_round2.py::test_comm_two_ranks_on_one_gpu, two processes, gloo
callbacks through pgv_comm_create_custom)."""
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
    d = np.full((len(queries), k), np.inf, dtype=np.float64)
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
        # --- build: this rank's heap rows, assigned locally, every row sent to the owner of its list
        lo, hi = sharding.row_shard(3000, rank, world)
        mine, _ = ora.assign(po.OPS_L2, po.ORA_F32, ivf.centers, data[lo:hi])
        every = sharding.gather_assignments(torch.from_numpy(mine), 3000, world)
        # lists to ranks by rows (bench.py's default): the map comes from the all-reduced list sizes, the same on every rank
        owners = sharding.plan_owners(sharding.global_list_sizes(torch.from_numpy(mine), 12), world, "balanced")
        v, t, off = sharding.exchange_rows(torch.from_numpy(data[lo:hi]), torch.arange(lo, hi, dtype=torch.int64),
                                           torch.from_numpy(mine), 12, owners)
        assert int(off[-1]) == v.shape[0]
        # --- scan of the local image + the exact ground truth merged across the ranks
        ld, lt = _local_search(ora, po, ivf.centers, off.numpy(), v.numpy(), t.numpy().astype(np.uint64), queries, 12, 7)
        md = sharding.merge_exact_topk(ld, 7)
        torch.save({"v": v, "t": t, "off": off, "md": md, "lists": every, "owners": owners}, out + ".%d" % rank)
    finally:
        dist.destroy_process_group()


def test_world2_matches_single_process(tmp_path):
    out = str(tmp_path / "w2.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    from pgvector_amd import sharding
    ora, po, data, ivf, queries = _make_problem()
    want_lists, _ = ora.assign(po.OPS_L2, po.ORA_F32, ivf.centers, data)
    # the serial image: list-major, heap order inside a list
    order = np.argsort(want_lists, kind="stable")
    gvec, gtid = torch.from_numpy(data[order]), torch.from_numpy(order.astype(np.int64))
    goff = torch.zeros(13, dtype=torch.int64)
    goff[1:] = torch.cumsum(torch.from_numpy(np.bincount(want_lists, minlength=12)), 0)
    sd, _ = _local_search(ora, po, ivf.centers, goff.numpy(), gvec.numpy(), gtid.numpy().astype(np.uint64), queries, 12, 7)
    owners = sharding.plan_owners(torch.from_numpy(np.bincount(want_lists, minlength=12)), 2, "balanced")
    rows = []
    for r in range(2):
        got = torch.load(out + ".%d" % r)
        np.testing.assert_array_equal(got["lists"].numpy(), want_lists)
        np.testing.assert_array_equal(got["owners"].numpy(), owners.numpy())     # every rank derived the same map
        v, t, off = sharding.local_index_arrays(gvec, gtid, goff, r, 2, owners)
        rows.append(int(off[-1]))
        np.testing.assert_array_equal(got["off"].numpy(), off.numpy())
        np.testing.assert_array_equal(got["t"].numpy(), t.numpy())
        np.testing.assert_array_equal(got["v"].numpy(), v.numpy())
        np.testing.assert_array_equal(got["md"].numpy(), sd.numpy())  # every list probed: the exact top-k
    assert sum(rows) == 3000 and max(rows) <= 1.05 * 1500


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


def test_plan_owners_balances_rows_and_keeps_modulo_selectable():
    """lists longest first onto the lightest rank: k-means list sizes are skewed (round 4's 8-rank record: l % 8 left one
    rank with 61 % of the mean); the balanced map keeps the heaviest rank within 5 % of the mean, is a pure function of
    the sizes, and `modulo` is still l % world"""
    from pgvector_amd import sharding
    rng = np.random.default_rng(8)
    for nlists, world in ((256, 8), (4096, 8), (1000, 4), (512, 2), (13, 3)):
        sizes = torch.from_numpy((rng.lognormal(0.0, 1.0, nlists) * 1000).astype(np.int64) + 1)
        own = sharding.plan_owners(sizes, world, "balanced")
        assert own.shape == (nlists,) and int(own.min()) >= 0 and int(own.max()) < world
        load = torch.bincount(own, weights=sizes.double(), minlength=world)
        mod = torch.bincount(torch.arange(nlists) % world, weights=sizes.double(), minlength=world)
        if nlists >= 256:
            assert float(load.max() / load.mean()) <= 1.05, (nlists, world, load)
        assert float(load.max()) <= float(mod.max()) + 1e-9
        assert torch.equal(own, sharding.plan_owners(sizes.clone(), world, "balanced"))
        assert torch.equal(sharding.plan_owners(sizes, world, "modulo"), torch.arange(nlists) % world)
    assert torch.equal(sharding.plan_owners(torch.tensor([5, 1, 1]), 1, "balanced"), torch.zeros(3, dtype=torch.int64))
    with pytest.raises(ValueError):
        sharding.plan_owners(torch.tensor([1, 2]), 2, "random")
