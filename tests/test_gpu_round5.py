"""Round 5 on the GPU: overlapping batches (pgv_index_set_overlap) answer exactly what stream-ordered batches answer."""
import os

import numpy as np
import pytest

from helpers import gen
from pgvector_amd import api

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = api.Context(0)
    yield c
    c.close()


def _index(ctx, oracle, n=60000, dim=128, lists=64, seed=11):
    from oracle import pyoracle as po
    from helpers import CpuIvf, gen
    data = gen(n, dim, seed=seed, dist="clustered", clusters=lists)
    ivf = CpuIvf(oracle, po.OPS_L2, po.ORA_F32, data, lists)
    ix = api.IvfIndex(ctx, ivf.metric, api.PGV_F32, dim, ivf.centers, ivf.list_offsets, ivf.vectors, ivf.tids)
    return ix, ivf


def test_overlapping_batches_answer_what_stream_ordered_batches_answer(ctx, oracle):
    """six different 512-query batches (the matrix-core scan: 512 x 8 probes over 64 lists), device buffers in and out:
    stream-ordered first, then on 2 and on 3 lanes with nothing but ctx.sync() at the end -- identical distances, slots
    and TIDs; host buffers through a lane too; the statistics of the lanes count as the context's; back to one lane"""
    import torch
    from helpers import gen
    ix, ivf = _index(ctx, oracle)
    dev = torch.device("cuda", 0)
    nb, nq, probes, k = 6, 512, 8, 10
    qs = [torch.from_numpy(gen(nq, 128, seed=100 + b, dist="clustered", clusters=64)).to(dev) for b in range(nb)]
    try:
        want = []
        for b in range(nb):
            d, s, t = ix.search_batch(qs[b], probes, k, want_tid=True)
            ctx.sync()
            want.append((d.cpu().numpy(), s.cpu().numpy(), t.cpu().numpy()))
        for lanes in (2, 3):
            ix.set_overlap(lanes)
            ctx.set_profiling(True)
            ctx.reset_stats()
            outs = [tuple(torch.empty((nq, k), device=dev, dtype=dt) for dt in (torch.float32, torch.int64, torch.int64))
                    for _ in range(nb)]
            for rep in range(3):                      # the lanes are reused: a lane's next batch waits for its previous one
                for b in range(nb):
                    ix.search_batch(qs[b], probes, k, want_tid=True, out=outs[b])
            ctx.sync()
            st = ctx.stats()
            ctx.set_profiling(False)
            assert st["scan_launches"] == 3 * nb and st["scan_pairs"] > 0      # the lanes' work is the context's
            for b in range(nb):
                np.testing.assert_array_equal(outs[b][0].cpu().numpy(), want[b][0])
                np.testing.assert_array_equal(outs[b][1].cpu().numpy(), want[b][1])
                np.testing.assert_array_equal(outs[b][2].cpu().numpy(), want[b][2])
            # host buffers: synchronous, as without lanes
            d, s, t = ix.search_batch(qs[0].cpu().numpy(), probes, k, want_tid=True)
            np.testing.assert_array_equal(np.asarray(d), want[0][0])
            np.testing.assert_array_equal(np.asarray(t).astype(np.int64), want[0][2])
        ix.set_overlap(1)
        d, s, t = ix.search_batch(qs[1], probes, k, want_tid=True)
        ctx.sync()
        np.testing.assert_array_equal(d.cpu().numpy(), want[1][0])
        with pytest.raises(Exception):
            ix.set_overlap(9)
    finally:
        ix.close()


def test_device_queries_written_just_before_the_call_are_seen_by_the_lane(ctx, oracle):
    """the lane starts behind what the context's stream holds at the call: a query buffer filled by a kernel on that
    stream right before search_batch is read after the fill, not before"""
    import torch
    from helpers import gen
    tctx = api.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    ix, ivf = _index(tctx, oracle, seed=12)
    dev = torch.device("cuda", 0)
    try:
        q = torch.from_numpy(gen(512, 128, seed=300, dist="clustered", clusters=64)).to(dev)
        d0, _, t0 = ix.search_batch(q, 8, 10, want_tid=True)
        tctx.sync()
        ix.set_overlap(2)
        for _ in range(4):
            buf = torch.zeros_like(q)
            big = torch.randn((4096, 4096), device=dev)
            for _ in range(3):
                big = big @ big * 1e-3                # keeps torch's stream busy: the copy below is queued behind it
            buf.copy_(q)
            d1, _, t1 = ix.search_batch(buf, 8, 10, want_tid=True)
            tctx.sync()
            torch.cuda.synchronize()
            np.testing.assert_array_equal(d1.cpu().numpy(), d0.cpu().numpy())
            np.testing.assert_array_equal(t1.cpu().numpy(), t0.cpu().numpy())
    finally:
        ix.close()
        tctx.close()


# ------------------------------------------------------------------ pgv_exact_topk on 128 x 128 tiles (kernels_dense.hip)
@pytest.mark.parametrize("ops,dt,dim,n,nq,k,dist", [
    ("l2", "f32", 1536, 3000, 300, 10, "clustered"),   # ragged query tile (300 = 2 x 128 + 44), ragged row tile
    ("l2", "f32", 100, 5000, 128, 10, "normal"),       # a row of 25 vectors: the last slice is partly zeros
    ("l2", "f32", 7, 1000, 129, 40, "normal"),         # one slice, k' = 160
    ("l2", "f32", 2000, 129, 256, 10, "uniform"),      # the largest indexed dimension, rows barely past one tile
    ("ip", "f32", 768, 4000, 200, 10, "clustered"),
    ("l2", "f16", 3072, 2000, 130, 10, "clustered"),
    ("ip", "f16", 1024, 2400, 128, 25, "clustered"),
    ("l2", "f32", 8, 4000, 160, 40, "int"),            # exact ties: lower row index first, as the sequential scan
])
def test_exact_topk_on_128_tiles_is_the_sequential_scan(ctx, oracle, ops, dt, dim, n, nq, k, dist):
    """from 128 queries on pgv_exact_topk runs mfma_dense_kernel (128 queries x 128 rows per workgroup, four accumulator
    chains by quarters of the row) + the exact tail: ids are the oracle's per-row l2 / inner-product calls + ascending
    sort (src/vector.c:579-620) up to float ties, distances to 1e-5; PGV_NO_DENSE128 cannot be flipped inside a process,
    so the old kernel is the comparison of tests/test_gpu_round3.py (< 128 queries)"""
    from oracle import pyoracle as po
    from helpers import assert_topk_equiv, gen
    odt = po.ORA_F32 if dt == "f32" else po.ORA_F16
    oops = po.OPS_L2 if ops == "l2" else po.OPS_IP
    rows = gen(n, dim, seed=951, dist=dist, dtype=odt)
    queries = gen(nq, dim, seed=952, dist=dist, dtype=odt)
    if dist == "int":
        rows[500:530] = rows[17]
        queries[3] = rows[17]
    metric = api.PGV_L2SQ if ops == "l2" else api.PGV_NEG_IP
    d, idx = api.exact_topk(ctx, metric, api.PGV_F32 if dt == "f32" else api.PGV_F16, dim, queries, rows, k)
    s = oracle.index_struct(oops, odt, rows[:1], np.array([0, n], dtype=np.int64), rows, np.arange(n, dtype=np.uint64))
    for i in range(0, nq, 3):
        wt, wd = oracle.search(s, queries[i], 1, k)
        scale = 0.0
        if metric == api.PGV_NEG_IP:
            scale = 1e-5 * float(np.max(np.abs(rows.astype(np.float64)) @ np.abs(queries[i].astype(np.float64))))
        assert_topk_equiv(idx[i].tolist(), d[i], wt.tolist(), wd, atol=max(scale, 1e-30),
                          what="dense128 %s %s dim %d q %d" % (ops, dt, dim, i))
        if dist == "int":
            assert idx[i].tolist() == wt.astype(np.int64).tolist()      # ties resolved towards the lower row index


# ------------------------------------------------------------------ the 64-query form of the batched list scan
@pytest.mark.parametrize("ops,dt,dim,lists,probes,nq", [
    ("l2", "f32", 128, 16, 8, 700),      # 700 x 8 / 16 = 350 queries per list: every task group is full (64)
    ("ip", "f32", 96, 32, 6, 300),       # 56 per list: groups of 64 with ragged tails, some lists under 33 (lower tile only)
    ("l2", "f16", 264, 24, 8, 260),      # halfvec, a row that is not whole slices
    ("l2", "f32", 1536, 40, 10, 200),    # the headline's row shape, 50 per list
    ("l2", "f32", 8, 12, 6, 150),        # exact ties (integer data): the recheck decides, lower position first
])
def test_the_64_query_scan_form_answers_like_the_oracle(ctx, oracle, ops, dt, dim, lists, probes, nq):
    """mfma_scan_kernel<.., 64> (two tiles per wavefront, chains by quarters of the row; chosen for halfvec batches with more
    than 12 queries per list on average, forced here for fp32 too by PGV_SCAN_WIDE=1 in the fixture's environment when set):
    heads equal the oracle's GetScanLists + GetScanItems + sort for every query (src/ivfscan.c:47-187)"""
    import torch
    from oracle import pyoracle as po
    from helpers import CpuIvf, assert_topk_equiv, gen
    odt = po.ORA_F32 if dt == "f32" else po.ORA_F16
    oops = po.OPS_L2 if ops == "l2" else po.OPS_IP
    n = 6000
    dist = "int" if dim == 8 else "clustered"
    data = gen(n, dim, seed=971, dist=dist, dtype=odt, clusters=lists)
    ivf = CpuIvf(oracle, oops, odt, data, lists)
    ix = api.IvfIndex(ctx, ivf.metric, api.PGV_F32 if dt == "f32" else api.PGV_F16, dim, ivf.centers, ivf.list_offsets,
                      ivf.vectors, ivf.tids)
    queries = gen(nq, dim, seed=972, dist=dist, dtype=odt, clusters=lists)
    try:
        assert nq * probes / lists > 12
        ctx.set_profiling(True)
        ctx.reset_stats()
        d, s, t = ix.search_batch(queries, probes, 10, want_tid=True)
        st = ctx.stats()
        ctx.set_profiling(False)
        assert st["scan_launches"] >= 1
        for i in range(nq):
            wt, wd = oracle.search(ivf.struct, queries[i], probes, 10)
            scale = 0.0
            if ops == "ip":
                scale = 1e-5 * float(np.max(np.abs(data.astype(np.float64)) @ np.abs(queries[i].astype(np.float64))))
            assert_topk_equiv(np.asarray(t[i])[:len(wt)].astype(np.uint64).tolist(), np.asarray(d[i])[:len(wt)], wt.tolist(), wd,
                              atol=max(scale, 1e-30), what="wide scan %s %s dim %d q %d" % (ops, dt, dim, i))
    finally:
        ix.close()


def test_the_64_query_scan_form_for_fp32_too_in_a_process_that_forces_it():
    """the default picks the 64-query form from 12 queries per list on (fp32 too since round 6: a few per cent); PGV_SCAN_WIDE=1 -- read
    once per process -- forces it everywhere: five shapes incl. fp32 L2 / IP and exact ties, every query against the oracle"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "mp_wide_scan_worker.py")], capture_output=True, text=True,
                       timeout=600, env=dict(os.environ, PGV_SCAN_WIDE="1"))
    assert r.returncode == 0 and "WIDE-OK 5" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])


# ---------------------------------------------------------------------------------------------------------------------
# SelectNeighbors of the elements being inserted, on the device (pgv_hnsw_build_neighbors)
def _host_select(ids, dist, tri, lm):
    """src/hnswutils.c:1064-1165 for a list without cached flags, candidates nearest first, pair distances in the
    (u, v < u) triangle: (neighbors, distances, closer flags) in r's order"""
    nw = len(ids)
    if nw <= lm:
        return ids[::-1].tolist(), dist[::-1].tolist(), [0] * nw
    chosen, looked = [], 0
    for j in range(nw):
        if len(chosen) >= lm:
            break
        looked = j + 1
        if all(tri[j * (j - 1) // 2 + r] > dist[j] for r in chosen):
            chosen.append(j)
    order = list(chosen)
    for x in range(looked):
        if len(order) >= lm:
            break
        if x not in chosen:
            order.append(x)
    return [int(ids[x]) for x in order], [float(dist[x]) for x in order], [1 if i < len(chosen) else 0 for i in range(len(order))]


@pytest.mark.parametrize("metric,dist_kind,dim", [(api.PGV_L2SQ, "clustered", 48), (api.PGV_L2SQ, "int10", 8), (api.PGV_NEG_IP, "normal", 96)])
def test_select_neighbors_on_the_device_is_the_sweep_over_the_same_lists(ctx, metric, dist_kind, dim):
    """pgv_hnsw_build_neighbors = pgv_hnsw_build_search + CheckElementCloser's distances + the sweep of SelectNeighbors
    (src/hnswutils.c:1040-1165): against the same sweep done here over the lists and pair distances the two older calls
    return -- ids, distances (bitwise: the same kernels), closer flags, counts, every layer."""
    import ctypes as C
    from pgvector_amd import _host
    n, m, efc = 5000, 6, 40
    data = gen(n, dim, seed=801, dist=dist_kind) if dist_kind != "clustered" else gen(n, dim, seed=801, dist="clustered", clusters=25)
    mirror = api.Hnsw(ctx, metric, api.PGV_F32, dim, data)
    # a graph over the first 4000 rows; the last 1000 are "being inserted"
    head = api.Hnsw(ctx, metric, api.PGV_F32, dim, data[:4000])
    built = _host.hnsw_build(head, data[:4000], m, efc, api.make_rng(seed=3), max_batch=64)
    head.close()
    levels = np.concatenate([built["levels"], np.zeros(1000, np.int32)])
    rngl = np.random.default_rng(5)
    new_levels = np.minimum((-np.log(rngl.random(1000)) / np.log(m)).astype(np.int32), int(built["levels"].max()))
    levels[4000:] = new_levels
    nbr_start = np.zeros(n + 1, np.int64)
    nbr_start[1:] = np.cumsum((levels.astype(np.int64) + 2) * m)
    nbr = np.full(int(nbr_start[-1]), -1, np.int32)
    nbr[:len(built["nbr"])] = built["nbr"]
    mirror.set_graph(m, built["entry"], levels, nbr_start, nbr)
    elems = np.arange(4000, 5000, dtype=np.int32)
    lcap = int(min(new_levels.max(), levels[built["entry"]])) + 1
    per = len(elems) * lcap
    ids = np.empty((per, efc), np.int32)
    dist = np.empty((per, efc), np.float32)
    cnt = np.empty(per, np.int32)
    api.check(api.lib.pgv_hnsw_build_search(mirror.h, api.ptr(elems), api.ptr(new_levels), len(elems), efc, lcap, api.ptr(ids),
                                            api.ptr(dist), api.ptr(cnt)))
    oi = np.empty((per, 2 * m), np.int32)
    od = np.empty((per, 2 * m), np.float32)
    oc = np.empty((per, 2 * m), np.uint8)
    on = np.empty(per, np.int32)
    pairs = C.c_int64()
    api.check(api.lib.pgv_hnsw_build_neighbors(mirror.h, api.ptr(elems), api.ptr(new_levels), len(elems), efc, lcap, api.ptr(oi),
                                               api.ptr(od), api.ptr(oc), api.ptr(on), C.byref(pairs)))
    thinned = 0
    want_pairs = 0
    for g in range(per):
        q, lc = divmod(g, lcap)
        lm = 2 * m if lc == 0 else m
        nw = 0 if lc > new_levels[q] else int(cnt[g])
        tri = None
        if nw > lm:
            thinned += 1
            want_pairs += nw * (nw - 1) // 2
            gi = ids[g, :nw]
            a = np.concatenate([np.full(u, gi[u], np.int32) for u in range(1, nw)])
            b = np.concatenate([gi[:u] for u in range(1, nw)])
            tri = np.empty(len(a), np.float32)
            api.check(api.lib.pgv_hnsw_score_pairs(mirror.h, api.ptr(a), api.ptr(b), len(a), api.ptr(tri)))
        wi, wd, wc = _host_select(ids[g, :nw], dist[g, :nw], tri, lm)
        assert on[g] == len(wi), (g, on[g], len(wi))
        assert oi[g, :on[g]].tolist() == wi, (g, lc, nw)
        assert od[g, :on[g]].tolist() == wd
        assert oc[g, :on[g]].tolist() == wc
    assert thinned > 300 and pairs.value == want_pairs
    mirror.close()


@pytest.mark.parametrize("max_batch,m", [(1, 6), (64, 8), (512, 8), (256, 24), (128, 36)])
def test_hnsw_build_on_the_device_builds_the_graph_of_the_host_replay(ctx, max_batch, m):
    """the whole build three ways: the graph updates on the device (default: pgv_hnsw_link_*, HnswUpdateConnection for
    every list a batch links into replayed by GPU lanes), the host-side replay on OpenMP threads (PGV_HNSW_HOST_LINK=1),
    and that with SelectNeighbors of the new elements on the host as well (PGV_HNSW_HOST_SELECT=1): the same graph, tuple
    for tuple, the same duplicates, the same entry point.  m = 36 (lists of 72, ef_construction 72) takes the forms for
    large m: one lane per list for the replay (hnsw_link_kernel<200>, hnsw_link_core.h as compiled for the device) and for the
    new elements' selection (hnsw_select_kernel)."""
    from pgvector_amd import _host
    n, dim, efc = (1500, 16, 24) if max_batch == 1 else (20000 if m < 30 else 8000, 64, max(48, 2 * m))
    data = gen(n, dim, seed=811, dist="clustered", clusters=40)
    data[n // 2:n // 2 + 40] = data[7]        # duplicates: eleven heap TIDs fit an element (HNSW_HEAPTIDS 10), the rest link
    graphs = []
    for env in ({}, {"PGV_HNSW_HOST_LINK": "1"}, {"PGV_HNSW_HOST_SELECT": "1"}):
        os.environ.update(env)
        try:
            mirror = api.Hnsw(ctx, api.PGV_L2SQ, api.PGV_F32, dim, data)
            built = _host.hnsw_build(mirror, data, m, efc, api.make_rng(seed=9), max_batch=max_batch)
            # the mirror holds the graph it was built with: a search walks it
            elem, _, _ = mirror.search(data[:16], 40, 5)
            assert (elem[:, 0] >= 0).all()
            mirror.close()
        finally:
            for k in env:
                os.environ.pop(k, None)
        graphs.append(built)
    a = graphs[0]
    if m <= 8:      # (a list that is not thinned keeps the search's order: the identical rows are not met first)
        assert (a["dup_of"] >= 0).sum() >= 9
    for b in graphs[1:]:
        assert a["entry"] == b["entry"] and a["batches"] == b["batches"] and a["nelements"] == b["nelements"]
        np.testing.assert_array_equal(a["levels"], b["levels"])
        np.testing.assert_array_equal(a["dup_of"], b["dup_of"])
        np.testing.assert_array_equal(a["nbr"], b["nbr"])
