"""GPU parity: libpgv_hip (through its C ABI) against the CPU oracle on the same
seeded inputs.  Integer/index results exact, distances within 1e-5 relative
(north_star), row ids identical wherever the reference's order is determined."""
import ctypes as C
import math
import os

import numpy as np
import pytest

import pgvector_amd
from oracle import pyoracle as po
from pgvector_amd import api

from helpers import RTOL, CpuIvf, assert_close, assert_topk_equiv, gen, golden, normalize_rows

pytestmark = pytest.mark.gpu

DT = {po.ORA_F32: api.PGV_F32, po.ORA_F16: api.PGV_F16}


def ora_dist(oracle, ops, dtype, q, rows):
    return np.array([oracle.lib.ora_index_distance(ops, dtype, rows.shape[1], po._p(r), po._p(q)) for r in rows])


# ------------------------------------------------------------ raw distances
@pytest.mark.parametrize("dtype", [po.ORA_F32, po.ORA_F16])
@pytest.mark.parametrize("dim", [1, 2, 3, 5, 9, 16, 31, 100, 128, 384, 768, 1000, 1536, 2000, 3072, 4000])
def test_distance_batch_all_metrics(ctx, oracle, dtype, dim):
    n = 257 if dim > 1000 else 1031
    rows = gen(n, dim, seed=dim, dist="normal", dtype=dtype)
    q = gen(1, dim, seed=dim + 1, dist="normal", dtype=dtype)[0]
    for metric, ops in [(api.PGV_L2SQ, po.OPS_L2), (api.PGV_NEG_IP, po.OPS_IP), (api.PGV_L1, po.OPS_L1)]:
        got = api.distance_batch(ctx, metric, DT[dtype], dim, q, rows)
        want = ora_dist(oracle, ops, dtype, q, rows)
        # inner products of normal data cancel: tolerance relative to the magnitude of the terms
        scale = np.abs(rows.astype(np.float64)) @ np.abs(q.astype(np.float64)) if metric == api.PGV_NEG_IP else 0.0
        assert_close(got, want, rtol=RTOL, atol=RTOL * np.max(scale), what="metric %d dim %d" % (metric, dim))


def test_known_answers_through_the_gpu(ctx, oracle):
    """the reference's SQL known answers (test/expected/vector_type.out, halfvec.out):
    kernel value from the GPU, float8 post-processing as the fmgr wrapper does it"""
    n_checked = 0
    for case in golden("distance_known_answers.json")["cases"]:
        f = case["func"]
        if "error" in case or f not in ("l2_distance", "inner_product", "negative_inner_product", "l1_distance"):
            continue
        dtype = po.ORA_F16 if case["type"] == "halfvec" else po.ORA_F32
        a = np.asarray(case["args"][0], dtype=po.NP_OF[dtype])
        b = np.asarray(case["args"][1], dtype=po.NP_OF[dtype])
        metric = {"l2_distance": api.PGV_L2SQ, "l1_distance": api.PGV_L1}.get(f, api.PGV_NEG_IP)
        raw = float(api.distance_batch(ctx, metric, DT[dtype], len(a), b, a[None, :])[0])
        got = {"l2_distance": lambda v: math.sqrt(v), "inner_product": lambda v: -v}.get(f, lambda v: v)(raw)
        want = {"inf": math.inf, "-inf": -math.inf, "nan": math.nan}.get(case["expect"], case["expect"])
        assert got == want or (math.isnan(got) and math.isnan(want)), (case, got)
        n_checked += 1
    assert n_checked >= 20


def test_overflow_is_not_trapped(ctx):
    """l2_distance([3e38],[-3e38]) = Infinity (vector_type.out:387-391)"""
    rows = np.array([[-3e38], [3e38]], dtype=np.float32)
    q = np.array([3e38], dtype=np.float32)
    got = api.distance_batch(ctx, api.PGV_L2SQ, api.PGV_F32, 1, q, rows)
    assert np.isinf(got[0]) and got[1] == 0.0


def test_device_and_host_buffers_agree(ctx):
    import torch
    rows = gen(3000, 768, seed=5)
    q = gen(1, 768, seed=6)[0]
    host = api.distance_batch(ctx, api.PGV_L2SQ, api.PGV_F32, 768, q, rows)
    dev = api.distance_batch(ctx, api.PGV_L2SQ, api.PGV_F32, 768, torch.from_numpy(q).cuda(),
                             torch.from_numpy(rows).cuda())
    ctx.sync()
    np.testing.assert_array_equal(host, dev.cpu().numpy())


# ------------------------------------------------------------------ IVF scan
def _upload(ctx, ivf):
    return api.IvfIndex(ctx, ivf.metric, DT[ivf.dtype], ivf.vectors.shape[1], ivf.centers, ivf.list_offsets,
                        ivf.vectors, ivf.tids)


@pytest.fixture(scope="module")
def small_ivf(oracle):
    data = gen(20000, 64, seed=11, dist="clustered", clusters=40)
    return CpuIvf(oracle, po.OPS_L2, po.ORA_F32, data, 50)


def test_rank_lists_matches_get_scan_lists(ctx, oracle, small_ivf):
    ix = _upload(ctx, small_ivf)
    queries = gen(33, 64, seed=12, dist="clustered", clusters=40)
    for probes in (1, 7, 50):
        lists, dist = ix.rank_lists(queries, probes)
        for i, q in enumerate(queries):
            wl, wd = oracle.get_scan_lists(small_ivf.struct, q, probes)
            assert_topk_equiv(lists[i], dist[i], wl, wd, what="rank_lists probes=%d q=%d" % (probes, i))


def test_rank_lists_tie_rule(ctx):
    """equal distances on the boundary keep the lower list id (src/ivfscan.c:92)"""
    centers = np.array([[0, 0], [1, 0], [1, 0], [5, 5], [0.5, 0], [1, 0]], dtype=np.float32)
    off = np.arange(7, dtype=np.int64)
    ix = api.IvfIndex(ctx, api.PGV_L2SQ, api.PGV_F32, 2, centers, off, centers)
    lists, dist = ix.rank_lists(np.array([[0, 0]], dtype=np.float32), 3)
    assert lists[0].tolist() == [0, 4, 1] and dist[0].tolist() == [0.0, 0.25, 1.0]
    lists, _ = ix.rank_lists(np.array([[1, 0]], dtype=np.float32), 2)
    assert lists[0].tolist() == [1, 2]


def test_scan_lists_order_and_values(ctx, oracle, small_ivf):
    ix = _upload(ctx, small_ivf)
    q = gen(1, 64, seed=13, dist="clustered", clusters=40)[0]
    lists, _ = oracle.get_scan_lists(small_ivf.struct, q, 5)
    dist, slot = ix.scan_lists(q, lists)
    # unsorted: exactly the order the reference feeds its tuplesort
    want_slots = np.concatenate([np.arange(small_ivf.list_offsets[l], small_ivf.list_offsets[l + 1]) for l in lists])
    np.testing.assert_array_equal(slot, want_slots)
    assert_close(dist, ora_dist(oracle, po.OPS_L2, po.ORA_F32, q, small_ivf.vectors[want_slots]), what="scan_lists")
    # after the host-side sort it is GetScanItems
    wd, ws = oracle.get_scan_items(small_ivf.struct, q, lists)
    order = np.argsort(dist.astype(np.float64), kind="stable")
    assert_topk_equiv(slot[order], dist[order], ws, wd, what="scan items sorted")


def test_scan_lists_null_query_and_empty_lists(ctx, oracle):
    data = gen(50, 8, seed=1)
    centers = gen(6, 8, seed=2)
    ivf = CpuIvf(oracle, po.OPS_L2, po.ORA_F32, data, 6, centers=centers)
    ix = _upload(ctx, ivf)
    dist, slot = ix.scan_lists(None, np.arange(6))  # NULL query: ZeroDistance, all rows
    assert len(dist) == 50 and (dist == 0).all() and sorted(slot.tolist()) == list(range(50))
    empty = [l for l in range(6) if ivf.list_offsets[l + 1] == ivf.list_offsets[l]]
    dist, slot = ix.scan_lists(data[0], np.array(empty, dtype=np.int32))
    assert len(dist) == 0


@pytest.mark.parametrize("ops,dtype,dim", [(po.OPS_L2, po.ORA_F32, 64), (po.OPS_IP, po.ORA_F32, 96),
                                           (po.OPS_COSINE, po.ORA_F32, 100), (po.OPS_L2, po.ORA_F16, 128),
                                           (po.OPS_COSINE, po.ORA_F16, 72)])
def test_search_batch_matches_reference_scan(ctx, oracle, ops, dtype, dim):
    data = gen(12000, dim, seed=21, dist="clustered", clusters=30, dtype=dtype)
    ivf = CpuIvf(oracle, ops, dtype, data, 40)
    ix = _upload(ctx, ivf)
    queries = gen(64, dim, seed=22, dist="clustered", clusters=30, dtype=dtype)
    gq = normalize_rows(oracle, queries, dtype) if ops == po.OPS_COSINE else queries  # GetScanValue :222-229
    for probes, k in [(1, 10), (4, 10), (40, 25)]:
        dist, slot, tid = ix.search_batch(gq, probes, k, want_tid=True)
        for i in range(len(queries)):
            wt, wd = oracle.search(ivf.struct, queries[i], probes, k)
            got_t = tid[i][slot[i] >= 0]
            assert_topk_equiv(got_t.tolist(), dist[i][:len(got_t)], wt.tolist(), wd,
                              what="ops %d probes %d q %d" % (ops, probes, i))


@pytest.mark.parametrize("ops,dtype,dim,nq,probes", [(po.OPS_L2, po.ORA_F32, 256, 200, 4), (po.OPS_IP, po.ORA_F32, 768, 90, 3),
                                                     (po.OPS_L2, po.ORA_F16, 512, 150, 5), (po.OPS_L2, po.ORA_F32, 1536, 64, 6)])
def test_search_batch_many_queries_per_list(ctx, oracle, ops, dtype, dim, nq, probes):
    """lists probed by more than 8 queries of the batch take the tile kernel (row tiles in LDS
    by async DMA, 16 queries per pass); ragged list lengths, more than 16 queries per list"""
    n, lists = 3000, 12
    data = gen(n, dim, seed=141, dist="clustered", clusters=lists, dtype=dtype)
    ivf = CpuIvf(oracle, ops, dtype, data, lists)
    ix = _upload(ctx, ivf)
    queries = gen(nq, dim, seed=142, dist="clustered", clusters=lists, dtype=dtype)
    dist, slot, tid = ix.search_batch(queries, probes, 10, want_tid=True)
    for i in range(nq):
        wt, wd = oracle.search(ivf.struct, queries[i], probes, 10)
        assert_topk_equiv(tid[i][slot[i] >= 0].tolist(), dist[i][:len(wt)], wt.tolist(), wd,
                          what="tile ops %d dim %d q %d" % (ops, dim, i))


def test_scan_batch_with_given_probe_lists(ctx, oracle, small_ivf):
    """pgv_scan_batch = the GetScanItems half of pgv_search_batch (multi-GPU splits the two)"""
    ix = _upload(ctx, small_ivf)
    queries = gen(40, 64, seed=151, dist="clustered", clusters=40)
    lists, _ = ix.rank_lists(queries, 6)
    d1, s1, t1 = ix.search_batch(queries, 6, 12, want_tid=True)
    d2, s2, t2 = ix.scan_batch(queries, lists, 12, want_tid=True)
    np.testing.assert_array_equal(s1, s2)
    np.testing.assert_array_equal(d1, d2)
    np.testing.assert_array_equal(t1, t2)


def test_search_batch_exact_when_probing_every_list(ctx, oracle):
    """probes = lists is an exact scan: returned row ids must equal brute force"""
    data = gen(5000, 32, seed=31)
    ivf = CpuIvf(oracle, po.OPS_L2, po.ORA_F32, data, 16)
    ix = _upload(ctx, ivf)
    queries = gen(20, 32, seed=32)
    dist, slot, _ = ix.search_batch(queries, 16, 10)
    for i, q in enumerate(queries):
        d = ((ivf.vectors.astype(np.float64) - q) ** 2).sum(axis=1)
        want = np.argsort(d, kind="stable")[:10]
        assert_topk_equiv(slot[i].tolist(), dist[i], want.tolist(), d[want], what="exact q %d" % i)


def test_search_batch_short_lists_pad(ctx, oracle):
    data = gen(30, 4, seed=41)
    ivf = CpuIvf(oracle, po.OPS_L2, po.ORA_F32, data, 10, centers=gen(10, 4, seed=42))
    ix = _upload(ctx, ivf)
    dist, slot, _ = ix.search_batch(data[:3], 1, 20)
    for i in range(3):
        lists, _ = oracle.get_scan_lists(ivf.struct, data[i], 1)
        m = int(ivf.list_offsets[lists[0] + 1] - ivf.list_offsets[lists[0]])
        assert (slot[i][m:] == -1).all() and np.isinf(dist[i][m:]).all() and (slot[i][:m] >= 0).all()


def test_full_size_properties_on_device(ctx):
    """BASELINE-scale shapes through size-independent properties (no CPU oracle at this size):
    (1) probing every list is an exact scan: the batched path (tile / group kernels, top-k select)
        must return exactly what a brute-force pgv_distance_batch + sort returns;
    (2) the unsorted single-query scan holds the same multiset of distances;
    (3) results do not depend on the batch a query travels in (idempotence)."""
    import torch
    n, dim, lists, k = 300_000, 768, 300, 10
    g = torch.Generator(device="cuda")
    g.manual_seed(7)
    data = torch.rand((n, dim), generator=g, device="cuda")
    centers = data[torch.randperm(n, generator=g, device="cuda")[:lists]].contiguous()
    assign, _ = api.assign(ctx, api.PGV_L2SQ, api.PGV_F32, dim, centers, data, want_dist=False)
    order = torch.argsort(assign.long(), stable=True)
    off = torch.zeros(lists + 1, dtype=torch.int64, device="cuda")
    off[1:] = torch.cumsum(torch.bincount(assign.long(), minlength=lists), 0)
    vectors = data[order].contiguous()
    ix = api.IvfIndex(ctx, api.PGV_L2SQ, api.PGV_F32, dim, centers, off, vectors, order)
    queries = torch.rand((48, dim), generator=g, device="cuda")
    dist, slot, tid = ix.search_batch(queries, lists, k, want_tid=True)
    ctx.sync()
    for i in (0, 17, 47):
        brute = api.distance_batch(ctx, api.PGV_L2SQ, api.PGV_F32, dim, queries[i].contiguous(), vectors)
        ctx.sync()
        bd, bi = torch.sort(brute, stable=True)
        assert torch.equal(slot[i], bi[:k]), (slot[i], bi[:k])
        torch.testing.assert_close(dist[i], bd[:k], rtol=RTOL, atol=0)
        assert torch.equal(tid[i], order[bi[:k]])
    # idempotence: the same query alone, and inside another batch
    d1, s1, _ = ix.search_batch(queries[17:18].contiguous(), 12, k)
    d2, s2, _ = ix.search_batch(queries[10:30].contiguous(), 12, k)
    ctx.sync()
    # (a query alone runs the per-query kernels, inside a batch of 20 the list-major ones: the same rows in the same
    # order; the two kernel families sum a row's terms in different lane orders, so distances agree to the last bits)
    assert torch.equal(s1[0], s2[7])
    torch.testing.assert_close(d1[0], d2[7], rtol=1e-6, atol=0)
    # unsorted scan of a few lists == the same rows scored by distance_batch
    probe, _ = ix.rank_lists(queries[3:4].contiguous(), 5)
    ctx.sync()
    sd, ss = ix.scan_lists(queries[3].cpu().numpy(), probe[0].cpu().numpy())
    brute = api.distance_batch(ctx, api.PGV_L2SQ, api.PGV_F32, dim, queries[3].contiguous(), vectors)
    ctx.sync()
    np.testing.assert_allclose(sd, brute.cpu().numpy()[ss], rtol=RTOL)
    ix.close()


# ---------------------------------------------------------------------- build
@pytest.mark.parametrize("ops,dtype,dim,k", [(po.OPS_L2, po.ORA_F32, 48, 37), (po.OPS_IP, po.ORA_F32, 130, 200),
                                             (po.OPS_L2, po.ORA_F16, 64, 129), (po.OPS_L2, po.ORA_F32, 3, 5)])
def test_assign_matches_add_tuple_to_sort(ctx, oracle, ops, dtype, dim, k):
    rows = gen(3001, dim, seed=51, dist="clustered", dtype=dtype)
    centers = gen(k, dim, seed=52, dist="clustered", dtype=dtype)
    metric = api.PGV_L2SQ if ops == po.OPS_L2 else api.PGV_NEG_IP
    got, gd = api.assign(ctx, metric, DT[dtype], dim, centers, rows)
    want, wd = oracle.assign(ops, dtype, centers, rows)
    assert_close(gd, wd, atol=RTOL * dim, what="assign distance")
    diff = np.nonzero(got != want)[0]
    for r in diff:  # a different list is only acceptable on a float-level tie
        d = np.array([oracle.lib.ora_index_distance(ops, dtype, dim, po._p(rows[r]), po._p(centers[c]))
                      for c in (got[r], want[r])])
        assert abs(d[0] - d[1]) <= 4 * RTOL * max(abs(d).max(), 1e-30), (r, got[r], want[r], d)
    assert len(diff) <= max(1, len(rows) // 500)


def test_assign_exact_on_integer_data(ctx, oracle):
    """integer-valued data makes fp32 arithmetic exact: list ids must be identical,
    including the first-minimum-wins rule on real ties (src/ivfbuild.c:187-191)"""
    rows = gen(2000, 16, seed=61, dist="int")
    centers = gen(300, 16, seed=62, dist="int")
    centers[7] = centers[3]  # duplicate center: the lower id must win
    got, gd = api.assign(ctx, api.PGV_L2SQ, api.PGV_F32, 16, centers, rows)
    want, wd = oracle.assign(po.OPS_L2, po.ORA_F32, centers, rows)
    np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(gd.astype(np.float64), wd)


def test_kmeanspp_picks_match_on_exact_data(ctx, oracle):
    """integer data -> identical distances -> the D^2 sampling walk (src/ivfkmeans.c:77-84)
    must pick the same samples given the same pg_prng stream"""
    samples = gen(1500, 8, seed=71, dist="int")
    st = oracle.prng(99)
    import ctypes as C
    rng = api.make_rng(next_double=oracle.lib.ora_prng_double_cb, next_u32=oracle.lib.ora_prng_u32_cb,
                       state=C.cast(C.pointer(st), C.c_void_p))
    got = api.kmeanspp_init(ctx, api.PGV_OPS_L2, api.PGV_F32, 8, samples, 40, rng)
    want = oracle.kmeans_init_centers(po.OPS_L2, po.ORA_F32, samples, 40, oracle.prng(99))
    np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("ops,dtype", [(po.OPS_L2, po.ORA_F32), (po.OPS_IP, po.ORA_F32), (po.OPS_L2, po.ORA_F16),
                                       (po.OPS_COSINE, po.ORA_F16)])
def test_one_lloyd_step_from_given_centers(ctx, oracle, ops, dtype):
    """assignment identical (up to float ties), new centers within 1e-5 (SURVEY hard part 3)"""
    dim, k = 24, 30
    samples = gen(4000, dim, seed=81, dist="clustered", clusters=k, dtype=dtype)
    if ops != po.OPS_L2:
        samples = normalize_rows(oracle, samples, dtype)
    centers = np.ascontiguousarray(samples[np.random.default_rng(82).choice(len(samples), k, replace=False)])
    closest = np.full(len(samples), -1, dtype=np.int32)
    pops = {po.OPS_L2: api.PGV_OPS_L2, po.OPS_IP: api.PGV_OPS_IP, po.OPS_COSINE: api.PGV_OPS_COSINE}[ops]
    sums, counts, changes = api.lloyd_partial(ctx, pops, DT[dtype], dim, samples, centers, closest)
    want_closest, _ = oracle.lloyd_assign(ops, dtype, samples, centers)
    assert (closest != want_closest).sum() <= 2
    assert int(changes[0]) == len(samples)
    np.testing.assert_array_equal(counts, np.bincount(closest, minlength=k))
    # sums in sample order are bit-exact against the reference's SumCenters given the same assignment
    want_centers, want_counts = oracle.kmeans_compute_new_centers(ops, dtype, samples, closest, k, oracle.prng(1))
    got_centers = api.lloyd_finish(ctx, pops, DT[dtype], dim, sums, counts, api.make_rng(seed=1))
    np.testing.assert_array_equal(counts, want_counts)
    assert_close(got_centers.astype(np.float64), want_centers.astype(np.float64), rtol=RTOL,
                 atol=1e-3 if dtype == po.ORA_F16 else 1e-7, what="new centers")


def test_kmeans_end_to_end_quality_and_rules(ctx, oracle):
    """statistical parity (SURVEY hard part 3): same stopping rule, inertia no worse than
    the reference's Elkan k-means from the same seed stream by more than a few %"""
    import ctypes as C
    dim, k = 16, 25
    samples = gen(5000, dim, seed=91, dist="clustered", clusters=k)
    st = oracle.prng(7)
    rng = api.make_rng(next_double=oracle.lib.ora_prng_double_cb, next_u32=oracle.lib.ora_prng_u32_cb,
                       state=C.cast(C.pointer(st), C.c_void_p))
    centers, closest, iters = api.kmeans(ctx, api.PGV_OPS_L2, api.PGV_F32, dim, samples, k, rng)
    wc, wcl, wit = oracle.kmeans(po.OPS_L2, po.ORA_F32, samples, k, oracle.prng(7))
    assert 1 <= iters <= 500 and np.isfinite(centers).all()

    def inertia(c, a):
        return float(((samples.astype(np.float64) - c[a].astype(np.float64)) ** 2).sum())
    assert inertia(centers, closest) <= 1.05 * inertia(wc, wcl)
    # converged: one more assignment changes nothing
    again, _ = oracle.lloyd_assign(po.OPS_L2, po.ORA_F32, samples, centers)
    assert (again != closest).mean() < 0.002


def test_kmeans_degenerate_inputs(ctx):
    """test/t/008_ivfflat_centers.pl: duplicates / more lists than points; no samples -> RandomCenters"""
    data = np.tile(np.array([[1, 2, 3]], dtype=np.float32), (30, 1))
    centers, closest, iters = api.kmeans(ctx, api.PGV_OPS_L2, api.PGV_F32, 3, data, 5, api.make_rng(seed=1))
    assert np.isfinite(centers).all() and iters >= 1 and (closest == closest[0]).all()
    centers, _, iters = api.kmeans(ctx, api.PGV_OPS_L2, api.PGV_F32, 3, np.zeros((0, 3), np.float32), 4,
                                   api.make_rng(seed=1))
    assert iters == 0 and (centers >= 0).all() and (centers < 1).all()
    centers, _, _ = api.kmeans(ctx, api.PGV_OPS_COSINE, api.PGV_F32, 3, np.zeros((0, 3), np.float32), 4,
                               api.make_rng(seed=1))
    np.testing.assert_allclose(np.linalg.norm(centers, axis=1), 1.0, rtol=1e-6)


# ----------------------------------------------------------------------- HNSW
def test_hnsw_score_gathers(ctx, oracle):
    dim = 200
    elems = gen(5000, dim, seed=101, dist="normal")
    queries = gen(7, dim, seed=102, dist="normal")
    rng = np.random.default_rng(103)
    slot = rng.integers(0, 5000, 999).astype(np.int32)
    qof = rng.integers(0, 7, 999).astype(np.int32)
    for metric, ops in [(api.PGV_L2SQ, po.OPS_L2), (api.PGV_NEG_IP, po.OPS_IP), (api.PGV_L1, po.OPS_L1)]:
        h = api.Hnsw(ctx, metric, api.PGV_F32, dim, elems)
        got = h.score(queries, slot, qof)
        want = np.array([oracle.lib.ora_index_distance(ops, po.ORA_F32, dim, po._p(queries[q]), po._p(elems[s]))
                         for s, q in zip(slot, qof)])
        assert_close(got, want, rtol=RTOL, atol=RTOL * 50, what="hnsw score metric %d" % metric)
        h.close()


@pytest.mark.parametrize("ops,metric,dist,dtype,dim,m,ef", [
    (po.OPS_L2, api.PGV_L2SQ, "int10", "f32", 8, 8, 40),
    (po.OPS_L2, api.PGV_L2SQ, "int", "f32", 24, 8, 40),
    (po.OPS_L2, api.PGV_L2SQ, "normal", "f32", 100, 16, 100),
    (po.OPS_COSINE, api.PGV_NEG_IP, "normal", "f32", 256, 8, 40),
    (po.OPS_L1, api.PGV_L1, "int10", "f32", 8, 5, 17),
    (po.OPS_L2, api.PGV_L2SQ, "normal", "f16", 136, 8, 64),
    # the other two fp16 instantiations of the search kernel, and fp16 rows of whole 1 KiB slices (four rows per trip)
    (po.OPS_IP, api.PGV_NEG_IP, "normal", "f16", 256, 8, 40),
    (po.OPS_L1, api.PGV_L1, "int10", "f16", 8, 5, 17),
    (po.OPS_L2, api.PGV_L2SQ, "normal", "f16", 1536, 16, 64),
])
def test_hnsw_search_on_device(ctx, oracle, ops, metric, dist, dtype, dim, m, ef):
    """pgv_hnsw_search: the whole first batch of an HNSW scan in one kernel launch (greedy descent
    + HnswSearchLayer, src/hnswscan.c:25-56, src/hnswutils.c:824-987) against the oracle's search on
    the same graph; on integer-valued data the arithmetic is exact and the traversal must be
    identical wherever no two candidates are equally far ("int10": exact arithmetic, few ties)"""
    n = 3000
    data = gen(n, dim, seed=211, dist=dist)
    odt, gdt = (po.ORA_F32, api.PGV_F32) if dtype == "f32" else (po.ORA_F16, api.PGV_F16)
    if dtype == "f16":
        data = data.astype(np.float16).astype(np.float32)
    g = po.HnswGraph(oracle, ops, po.ORA_F32, data, m=m, ef_construction=32, seed=7)
    ex = g.export_tuples()
    stored = data[ex["rows"]]
    if ops == po.OPS_COSINE:
        stored = normalize_rows(oracle, np.ascontiguousarray(stored), po.ORA_F32)
    stored_dev = stored.astype(np.float16) if dtype == "f16" else stored
    mirror = api.Hnsw(ctx, metric, gdt, dim, stored_dev)
    mirror.set_graph(m, ex["entry"], ex["levels"], ex["nbr_start"], ex["nbr"])
    queries = gen(48, dim, seed=212, dist=dist)
    if dtype == "f16":
        queries = queries.astype(np.float16).astype(np.float32)
    gq = normalize_rows(oracle, queries, po.ORA_F32) if ops == po.OPS_COSINE else queries
    gq_dev = gq.astype(np.float16) if dtype == "f16" else gq
    k = min(10, ef)
    elem, gd, scored = mirror.search(gq_dev, ef, k)
    same_scored = 0
    for i, q in enumerate(queries):
        rows, wd, wscored = g.search(q, ef, k)
        got_rows = ex["rows"][elem[i][elem[i] >= 0]]
        # integer-valued data is full of equal distances; the order in which tied candidates are
        # expanded is unspecified (pairing-heap internals in the reference, a binary heap in the
        # oracle, array order here), so the walks -- and the scored counts -- legitimately differ
        # there and only the result is compared
        same_scored += int(scored[i] == wscored)
        if dtype == "f32":
            assert_topk_equiv(got_rows.tolist(), gd[i][:len(got_rows)], rows.tolist(), wd,
                              what="hnsw device search ops %d q %d" % (ops, i))
        else:  # the oracle graph search runs in fp32 on the fp16-rounded values: same values, same order
            assert len(got_rows) == len(rows)
            assert_close(gd[i][:len(rows)], wd, rtol=RTOL, atol=1e-6, what="hnsw f16 device search q %d" % i)
    assert dist == "int" or same_scored >= 40, same_scored
    if dist == "int10":
        assert same_scored >= 44, same_scored
    # the host-driven search (pgv_host_hnsw_search) walks the same graph with the same distances
    if dtype == "f32" and dist == "normal":
        from pgvector_amd import _host
        graph = _host.hnsw_graph(ex["levels"], ex["nbr_start"], ex["nbr"], m, ex["entry"])
        helem, hd, hscored = _host.hnsw_search(mirror, graph, gq, ef, k)
        assert np.array_equal(hscored, scored)
        assert np.array_equal(hd, gd)
    mirror.close()


def test_hnsw_search_on_device_edge_cases(ctx, oracle):
    """empty index, single element, k == ef_search == 1, argument checks"""
    dim = 16
    one = gen(1, dim, seed=3)
    mirror = api.Hnsw(ctx, api.PGV_L2SQ, api.PGV_F32, dim, one)
    with pytest.raises(api.PgvError):
        mirror.search(one, 10, 5)  # no graph yet
    mirror.set_graph(4, 0, np.zeros(1, np.int32), np.array([0, 8], np.int64), np.full(8, -1, np.int32))
    elem, d, scored = mirror.search(gen(3, dim, seed=4), 1, 1)
    assert elem.ravel().tolist() == [0, 0, 0] and scored.tolist() == [1, 1, 1]
    elem, d, scored = mirror.search(gen(3, dim, seed=4), 10, 5)
    assert (elem[:, 0] == 0).all() and (elem[:, 1:] == -1).all() and np.isinf(d[:, 1:]).all()
    with pytest.raises(api.PgvError):
        mirror.search(one, 5, 6)  # k > ef_search
    with pytest.raises(api.PgvError):
        mirror.search(one, 1001, 5)
    mirror.set_graph(4, -1, np.zeros(1, np.int32), np.array([0, 8], np.int64), np.full(8, -1, np.int32))
    elem, d, scored = mirror.search(gen(2, dim, seed=4), 10, 5)
    assert (elem == -1).all() and scored.tolist() == [0, 0]
    mirror.close()


@pytest.mark.parametrize("ops,metric,dist", [(po.OPS_L2, api.PGV_L2SQ, "int"), (po.OPS_L2, api.PGV_L2SQ, "normal"),
                                              (po.OPS_COSINE, api.PGV_NEG_IP, "normal"), (po.OPS_L1, api.PGV_L1, "int")])
def test_hnsw_search_with_gpu_candidate_scoring(ctx, oracle, ops, metric, dist):
    """HnswSearchLayer replayed by the C host glue with every candidate batch scored on the GPU
    (src/hnswutils.c:888-976) against the oracle's search on the same graph"""
    from pgvector_amd import _host
    dim, n = 24, 3000
    data = gen(n, dim, seed=111, dist=dist)
    g = po.HnswGraph(oracle, ops, po.ORA_F32, data, m=8, ef_construction=32, seed=5)
    ex = g.export_tuples()
    stored = data[ex["rows"]]
    if ops == po.OPS_COSINE:
        stored = normalize_rows(oracle, np.ascontiguousarray(stored), po.ORA_F32)
    mirror = api.Hnsw(ctx, metric, api.PGV_F32, dim, stored)
    graph = _host.hnsw_graph(ex["levels"], ex["nbr_start"], ex["nbr"], 8, ex["entry"])
    queries = gen(40, dim, seed=112, dist=dist)
    gq = normalize_rows(oracle, queries, po.ORA_F32) if ops == po.OPS_COSINE else queries
    elem, gd, scored = _host.hnsw_search(mirror, graph, gq, 40, 10)
    same_scored = 0
    for i, q in enumerate(queries):
        rows, wd, wscored = g.search(q, 40, 10)
        got_rows = ex["rows"][elem[i][elem[i] >= 0]]
        if dist == "int":  # exact arithmetic: identical traversal
            assert scored[i] == wscored
        same_scored += int(scored[i] == wscored)
        assert_topk_equiv(got_rows.tolist(), gd[i][:len(got_rows)], rows.tolist(), wd,
                          what="hnsw ops %d q %d" % (ops, i))
    assert same_scored >= 36
    mirror.close()


# ------------------------------------------------- pages -> mirror -> amgettuple
@pytest.mark.parametrize("ops,dtype,dim", [(po.OPS_L2, po.ORA_F32, 40), (po.OPS_COSINE, po.ORA_F32, 20),
                                           (po.OPS_IP, po.ORA_F16, 64)])
def test_index_built_staged_and_scanned_through_the_host_glue(ctx, oracle, ops, dtype, dim):
    """BuildIndex on the GPU -> 8 KB pages -> staging -> device mirror -> ivfflatgettuple,
    against the oracle scanning the same staged image"""
    from pgvector_amd import _host
    n, lists = 6000, 24
    heap = gen(n, dim, seed=121, dist="clustered", clusters=lists, dtype=dtype)
    tids = ((np.arange(n, dtype=np.uint64) // 50) << np.uint64(16)) | (np.arange(n, dtype=np.uint64) % 50 + 1)
    pops = {po.OPS_L2: api.PGV_OPS_L2, po.OPS_IP: api.PGV_OPS_IP, po.OPS_COSINE: api.PGV_OPS_COSINE}[ops]
    # rows and samples go in as the heap holds them: pgv_host_ivf_build normalises what the opclass's
    # NORM procs normalise (BuildCallback / SampleCallback, src/ivfbuild.c:148-156, :174-180)
    rows = heap
    samples = rows[np.random.default_rng(1).choice(n, 1200, replace=False)]
    rel = _host.Relation()
    rel.build(ctx, pops, DT[dtype], lists, rows, tids, samples, api.make_rng(seed=3))
    img = rel.stage(DT[dtype])
    assert img.nrows == n and sorted(img.tids.tolist()) == sorted(tids.tolist())
    metric = api.PGV_L2SQ if ops == po.OPS_L2 else api.PGV_NEG_IP
    mirror = api.IvfIndex(ctx, metric, DT[dtype], dim, img.centers, img.list_offsets, img.vectors, img.tids)
    ix = oracle.index_struct(ops, dtype, img.centers, img.list_offsets, img.vectors, img.tids)
    # every row sits in the list of its nearest center
    want_lists, _ = oracle.assign(ops, dtype, img.centers, img.vectors)
    have_lists = np.repeat(np.arange(lists), np.diff(img.list_offsets))
    assert (want_lists != have_lists).mean() < 0.002
    scan = _host.IvfScan(mirror, img, probes=3, normalize_query=(ops == po.OPS_COSINE))
    for q in gen(12, dim, seed=122, dist="clustered", clusters=lists, dtype=dtype):
        scan.rescan(q)
        got_t, got_d = scan.fetch(limit=15)
        wt, wd = oracle.search(ix, q, 3, 15)
        assert_topk_equiv(got_t.tolist(), got_d, wt.tolist(), wd, what="amgettuple")
    # exhausting the scan returns every tuple of the probed lists, ascending
    scan.rescan(heap[0])
    all_t, all_d = scan.fetch()
    lists_probed, _ = oracle.get_scan_lists(ix, normalize_rows(oracle, heap[:1], dtype)[0] if ops == po.OPS_COSINE else heap[0], 3)
    assert len(all_t) == sum(int(img.list_offsets[l + 1] - img.list_offsets[l]) for l in lists_probed)
    assert (np.diff(all_d) >= 0).all()
    # NULL query: all tuples of the first `probes` lists at distance 0 (ZeroDistance)
    scan.rescan(None)
    nt, nd = scan.fetch()
    assert len(nt) == int(img.list_offsets[3]) and (nd == 0).all()
    scan.close()
    # iterative scan (relaxed_order): batches of `probes` lists up to max_probes
    it = _host.IvfScan(mirror, img, probes=2, max_probes=6, iterative=True, normalize_query=(ops == po.OPS_COSINE))
    q = gen(1, dim, seed=123, dist="clustered", clusters=lists, dtype=dtype)[0]
    it.rescan(q)
    got_t, got_d = it.fetch()
    gq = normalize_rows(oracle, q[None, :], dtype)[0] if ops == po.OPS_COSINE else q
    pl, _ = oracle.get_scan_lists(ix, gq, 6)
    want = []
    for b in range(0, 6, 2):
        d, s_ = oracle.get_scan_items(ix, gq, pl[b:b + 2])
        want.append((img.tids[s_], d))
    assert len(got_t) == sum(len(w[0]) for w in want)
    at = 0
    for wt, wd in want:  # each batch sorted on its own
        assert_topk_equiv(got_t[at:at + len(wt)].tolist(), got_d[at:at + len(wt)], wt.tolist(), wd, what="iterative batch")
        at += len(wt)
    it.close()
    mirror.close()


# -------------------------------------------------------------- API contracts
def test_argument_errors(ctx):
    with pytest.raises(pgvector_amd.PgvError) as e:
        api.distance_batch(ctx, api.PGV_L2SQ, api.PGV_F32, 16001, np.zeros(16001, np.float32),
                           np.zeros((1, 16001), np.float32))
    assert e.value.code == pgvector_amd._lib.PGV_ERR_DIMS
    centers = np.zeros((3, 4), np.float32)
    ix = api.IvfIndex(ctx, api.PGV_L2SQ, api.PGV_F32, 4, centers, np.array([0, 1, 2, 3]), centers)
    with pytest.raises(pgvector_amd.PgvError):
        ix.rank_lists(np.zeros((1, 4), np.float32), 4)  # maxprobes > lists must be clamped by the caller
    with pytest.raises(pgvector_amd.PgvError):
        api.IvfIndex(ctx, api.PGV_L2SQ, api.PGV_F32, 4, centers, np.array([0, 2, 1, 3]), centers)


def test_c_driver(tmp_path):
    """the boundary from C, without Python or torch in the process: tests/c/abi_driver.c is compiled
    with gcc against include/pgv_hip.h + the host glue and run as its own program (build, stage, scan
    through amgettuple, batched search, exact scan, on-device HNSW search, all checked against
    brute force inside the driver)"""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "pgvector_amd", "lib")
    exe = str(tmp_path / "abi_driver")
    subprocess.run(["gcc", "-O1", "-Wall", "-I", os.path.join(root, "include"), "-I",
                    os.path.join(root, "pgvector_amd", "host"), os.path.join(root, "tests", "c", "abi_driver.c"),
                    "-o", exe, "-L", libdir, "-lpgv_host", "-lpgv_hip", "-lm", "-Wl,-rpath," + libdir], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "C-DRIVER OK" in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-2000:])


def test_mirror_follows_inserts_and_vacuum(ctx, oracle):
    """the device mirror is restaged when (and only when) the index pages changed: scans see an
    inserted row at once and stop returning vacuumed rows (src/ivfinsert.c:72-181,
    src/ivfvacuum.c:18-143)"""
    from pgvector_amd import _host
    dim, n, lists = 64, 2000, 8
    data = gen(n, dim, seed=301, dist="clustered", clusters=8)
    tids = ((np.arange(n, dtype=np.uint64) + 1) << np.uint64(16)) | np.uint64(1)
    rel = _host.Relation()
    rel.build(ctx, api.PGV_OPS_L2, api.PGV_F32, lists, data, tids, data)
    mirror = _host.Mirror(ctx, api.PGV_L2SQ, api.PGV_F32)

    def top(query, k=5):
        ix, img = mirror.get(rel)
        scan = _host.IvfScan(ix, img, lists)
        scan.rescan(query)
        t, d = scan.fetch(k)
        scan.close()
        return t, d

    q = data[17] + 0.001
    t0, d0 = top(q)
    assert int(t0[0]) == int(tids[17]) and mirror.restages == 1
    top(q)
    assert mirror.restages == 1  # unchanged pages: the mirror is reused
    # INSERT: a row right at the query goes into the nearest list and must come back first
    centers = mirror.get(rel)[1].img
    cen = np.ctypeslib.as_array(C.cast(centers.centers, C.POINTER(C.c_float)), shape=(lists, dim))
    lst = int(np.argmin(((cen - q) ** 2).sum(1)))
    new_tid = (np.uint64(999999) << np.uint64(16)) | np.uint64(7)
    rel.insert(api.PGV_F32, lst, q, int(new_tid))
    t1, d1 = top(q)
    assert int(t1[0]) == int(new_tid) and d1[0] == 0.0 and mirror.restages == 2
    assert int(t1[1]) == int(tids[17])
    # VACUUM: both go away
    removed, remaining = rel.bulkdelete([int(new_tid), int(tids[17])])
    assert (removed, remaining) == (2, n - 1)
    t2, d2 = top(q)
    assert mirror.restages == 3
    assert int(new_tid) not in set(map(int, t2)) and int(tids[17]) not in set(map(int, t2))
    np.testing.assert_array_equal(t2[:3], t0[1:4])
    mirror.close()


def _recall_of_graph(ctx, metric, data, levels, nbr_start, nbr, m, entry, queries, ef, k):
    mirror = api.Hnsw(ctx, metric, api.PGV_F32, data.shape[1], data)
    mirror.set_graph(m, entry, levels, nbr_start, nbr)
    elem, _, _ = mirror.search(queries, ef, k)
    mirror.close()
    d = ((queries[:, None, :].astype(np.float64) - data[None, :, :].astype(np.float64)) ** 2).sum(-1)
    kth = np.sort(d, axis=1)[:, k - 1]
    hits = sum(int((d[i, e[e >= 0]] <= kth[i] + 1e-9).sum()) for i, e in enumerate(elem))
    return hits / (len(queries) * k)


def test_hnsw_build_serial_is_the_reference_loop(ctx, oracle):
    """pgv_host_hnsw_build with max_batch = 1 (InsertTupleInMemory one tuple at a time,
    src/hnswbuild.c:436-476) on data with exact fp32 arithmetic and the oracle's pg_prng stream:
    same levels, same entry point, same neighbor tuples as the oracle's build"""
    from pgvector_amd import _host
    n, dim, m, efc = 1200, 8, 6, 24
    data = gen(n, dim, seed=401, dist="int10")
    g = po.HnswGraph(oracle, po.OPS_L2, po.ORA_F32, data, m=m, ef_construction=efc, seed=11)
    ex = g.export_tuples()
    assert np.array_equal(ex["rows"], np.arange(n))  # no duplicate vectors in this data
    st = oracle.prng(11)
    rng = api.make_rng(next_double=oracle.lib.ora_prng_double_cb, next_u32=oracle.lib.ora_prng_u32_cb,
                       state=C.cast(C.pointer(st), C.c_void_p))
    mirror = api.Hnsw(ctx, api.PGV_L2SQ, api.PGV_F32, dim, data)
    built = _host.hnsw_build(mirror, data, m, efc, rng, max_batch=1)
    assert built["nelements"] == n and built["batches"] == n
    np.testing.assert_array_equal(built["levels"], ex["levels"])
    np.testing.assert_array_equal(built["nbr_start"], ex["nbr_start"])
    assert built["entry"] == ex["entry"]
    # a tie between two candidates' distances may be walked in a different order (unspecified in the
    # reference); everything else must be identical
    same = np.array([np.array_equal(built["nbr"][built["nbr_start"][e]:built["nbr_start"][e + 1]],
                                    ex["nbr"][ex["nbr_start"][e]:ex["nbr_start"][e + 1]]) for e in range(n)])
    assert same.mean() >= 0.99, same.mean()
    # the mirror now holds the built graph: search it
    q = gen(32, dim, seed=402, dist="int10")
    elem, d, _ = mirror.search(q, 40, 10)
    for i in range(len(q)):
        rows, wd, _ = g.search(q[i], 40, 10)
        assert_topk_equiv(elem[i][elem[i] >= 0].tolist(), d[i][:len(rows)], rows.tolist(), wd, what="built graph q %d" % i)
    mirror.close()


@pytest.mark.parametrize("max_batch", [16, 256])
def test_hnsw_build_batched_quality(ctx, oracle, max_batch):
    """batched concurrent inserts (the reference's parallel build interleaves its workers the same
    way): the graph differs from the serial one but must search as well"""
    from pgvector_amd import _host
    n, dim, m, efc = 6000, 32, 8, 40
    data = gen(n, dim, seed=411, dist="clustered", clusters=30)
    queries = gen(64, dim, seed=412, dist="clustered", clusters=30)
    mirror = api.Hnsw(ctx, api.PGV_L2SQ, api.PGV_F32, dim, data)
    built = _host.hnsw_build(mirror, data, m, efc, api.make_rng(seed=5), max_batch=max_batch)
    assert built["nelements"] + int((built["dup_of"] >= 0).sum()) == n
    assert built["batches"] < n / 4
    # structural invariants: neighbors are linked elements at or above the layer, no self loops, lists within lm
    lv, st, nb = built["levels"], built["nbr_start"], built["nbr"]
    for e in range(0, n, 37):
        for lc in range(lv[e] + 1):
            lm = 2 * m if lc == 0 else m
            o = st[e] + (lv[e] - lc) * m
            ids = nb[o:o + lm]
            ids = ids[ids >= 0]
            assert len(set(ids.tolist())) == len(ids) and e not in ids
            assert (lv[ids] >= lc).all() and (built["dup_of"][ids] < 0).all()
    mirror.close()
    got = _recall_of_graph(ctx, api.PGV_L2SQ, data, lv, st, nb, m, built["entry"], queries, 40, 10)
    g = po.HnswGraph(oracle, po.OPS_L2, po.ORA_F32, data, m=m, ef_construction=efc, seed=5)
    ex = g.export_tuples()
    want = _recall_of_graph(ctx, api.PGV_L2SQ, data[ex["rows"]], ex["levels"], ex["nbr_start"], ex["nbr"], m,
                            ex["entry"], queries, 40, 10)
    assert got >= want - 0.03 and got >= 0.5, (got, want)


def test_hnsw_build_attaches_duplicates(ctx, oracle):
    """FindDuplicateInMemory (src/hnswbuild.c:313-364): an identical vector takes a heap-TID slot of
    the element already in the graph instead of becoming an element (up to 10 TIDs per element)"""
    from pgvector_amd import _host
    base = gen(150, 12, seed=421, dist="int10")
    data = np.ascontiguousarray(np.repeat(base, 13, axis=0))  # 13 copies: 10 fit one element, 3 the next
    rng_perm = np.random.default_rng(3).permutation(len(data))
    data = np.ascontiguousarray(data[rng_perm])
    g = po.HnswGraph(oracle, po.OPS_L2, po.ORA_F32, data, m=6, ef_construction=24, seed=2)
    st = oracle.prng(2)
    rng = api.make_rng(next_double=oracle.lib.ora_prng_double_cb, next_u32=oracle.lib.ora_prng_u32_cb,
                       state=C.cast(C.pointer(st), C.c_void_p))
    mirror = api.Hnsw(ctx, api.PGV_L2SQ, api.PGV_F32, 12, data)
    built = _host.hnsw_build(mirror, data, 6, 24, rng, max_batch=1)
    mirror.close()
    assert built["nelements"] == g.nelements
    dup = built["dup_of"]
    assert (dup >= 0).sum() == len(data) - g.nelements
    for r in np.nonzero(dup >= 0)[0][:200]:
        assert np.array_equal(data[r], data[dup[r]]) and dup[dup[r]] < 0


# ---------------------------------------------- operator-path cosine, bit vectors
@pytest.mark.parametrize("dtype", [po.ORA_F32, po.ORA_F16])
@pytest.mark.parametrize("dim", [1, 3, 16, 129, 768, 1536, 2000])
def test_cosine_distance_batch(ctx, oracle, dtype, dim):
    """cosine_distance without an index (src/vector.c:649-697, src/halfvec.c:652-700): float8 results"""
    n = 300
    rows = gen(n, dim, seed=501, dist="normal", dtype=dtype)
    q = gen(1, dim, seed=502, dist="normal", dtype=dtype)[0]
    rows[7] = 0  # a zero vector: NaN (0 / 0), test/expected/vector_type.out:437-441
    rows[8] = q  # identical: exactly 0 after the clamp unless rounding leaves it just below 1
    rows[9] = -q
    got = api.cosine_distance_batch(ctx, DT[dtype], dim, q, rows)
    name = "ora_cosine_distance" if dtype == po.ORA_F32 else "ora_halfvec_cosine_distance"
    want = np.array([oracle.sql(name, r, q, half=dtype == po.ORA_F16)[1] for r in rows])
    assert np.isnan(got[7]) and np.isnan(want[7])
    ok = ~np.isnan(want)
    # 1 - similarity: compare the similarities to 1e-5 relative
    assert_close(1.0 - got[ok], 1.0 - want[ok], rtol=RTOL, atol=1e-6, what="cosine dim %d" % dim)
    assert abs(got[8]) <= 2e-6 and abs(got[9] - 2.0) <= 2e-6


def test_cosine_distance_known_answers(ctx):
    """the reference's own cosine answers (test/expected/vector_type.out / halfvec.out)"""
    cases = [c for c in golden("distance_known_answers.json")["cases"] if c["func"] == "cosine_distance"]
    assert len(cases) >= 6
    for c in cases:
        dt = api.PGV_F32 if c["type"] == "vector" else api.PGV_F16
        if c.get("error") or len(c["args"][0]) != len(c["args"][1]):
            continue
        npt = np.float32 if dt == api.PGV_F32 else np.float16
        a = np.array(c["args"][0], dtype=npt)
        b = np.array(c["args"][1], dtype=npt)
        got = api.cosine_distance_batch(ctx, dt, len(a), b, a[None, :])[0]
        want = float(c["expect"])
        if math.isnan(want):
            assert math.isnan(got), c
        else:
            assert abs(got - want) <= 1e-6, (c, got)


@pytest.mark.parametrize("nbits", [1, 3, 8, 13, 64, 127, 128, 129, 513, 1536, 4000, 64000])
def test_bit_distance_batch(ctx, oracle, nbits):
    """hamming_distance / jaccard_distance (src/bitvec.c:45-70): exact float8 results"""
    rng = np.random.default_rng(nbits)
    n = 257
    bits = rng.integers(0, 2, (n + 1, nbits), dtype=np.uint8)
    bits[3] = 0          # jaccard with an all-zero row: 1
    bits[4] = bits[n]    # identical to the query
    packed = np.packbits(bits, axis=1)
    rows, q = np.ascontiguousarray(packed[:n]), np.ascontiguousarray(packed[n])
    got_h = api.bit_distance_batch(ctx, api.PGV_BIT_HAMMING, nbits, q, rows)
    got_j = api.bit_distance_batch(ctx, api.PGV_BIT_JACCARD, nbits, q, rows)
    np.testing.assert_array_equal(got_h, oracle.bit_rows("ora_bit_hamming", q, rows))
    np.testing.assert_array_equal(got_j, oracle.bit_rows("ora_bit_jaccard", q, rows))
    assert got_h[4] == 0 and got_j[3] == 1.0


def test_bit_distance_known_answers_and_edges(ctx, oracle):
    for c in golden("bit_known_answers.json")["cases"]:
        if "error" in c:
            continue
        na, pa = oracle.pack_bits(c["a"])
        nb, pb = oracle.pack_bits(c["b"])
        metric = api.PGV_BIT_HAMMING if c["func"] == "hamming_distance" else api.PGV_BIT_JACCARD
        got = api.bit_distance_batch(ctx, metric, na, pb, pa.reshape(1, -1))
        assert got[0] == c["value"], (c, got)
    with pytest.raises(api.PgvError):
        api.bit_distance_batch(ctx, 7, 8, np.zeros(1, np.uint8), np.zeros((1, 1), np.uint8))


def test_hnsw_build_pages_stage_search_end_to_end(ctx, oracle):
    """CREATE INDEX ... USING hnsw through the whole chain: GPU build -> FlushPages layout (hnsw_pages.c) ->
    staged back from the pages like a scan reads them -> device mirror -> pgv_hnsw_search; the results must be
    those of searching the built arrays directly (only the slot numbering differs)"""
    from pgvector_amd import _host
    n, dim, m, efc = 2500, 48, 8, 32
    data = gen(n, dim, seed=431, dist="clustered", clusters=12)
    data[100] = data[7]  # one duplicate vector: shares element 7's tuple
    tids = ((np.arange(n, dtype=np.uint64) + 1) << np.uint64(16)) | np.uint64(1)
    mirror = api.Hnsw(ctx, api.PGV_L2SQ, api.PGV_F32, dim, data)
    built = _host.hnsw_build(mirror, data, m, efc, api.make_rng(seed=9), max_batch=32)
    assert built["dup_of"][100] == 7
    queries = gen(24, dim, seed=432, dist="clustered", clusters=12)
    e1, d1, s1 = mirror.search(queries, 40, 10)
    mirror.close()
    rel = _host.Relation()
    rel.write_hnsw(api.PGV_F32, m, efc, data, tids, built["levels"], built["nbr_start"], built["nbr"], built["entry"],
                   built["dup_of"])
    img = rel.stage_hnsw(api.PGV_F32)
    assert img["n"] == n - 1
    m2 = api.Hnsw(ctx, api.PGV_L2SQ, api.PGV_F32, dim, img["vectors"])
    m2.set_graph(img["m"], img["entry"], img["levels"], img["nbr_start"], img["nbr"])
    e2, d2, s2 = m2.search(queries, 40, 10)
    m2.close()
    np.testing.assert_array_equal(d1, d2)
    np.testing.assert_array_equal(s1, s2)
    # same heap rows: slot -> first heap TID
    np.testing.assert_array_equal(tids[e1], img["heaptids"][e2, 0])
    # the duplicate's heap TID rides on element 7's tuple
    s7 = int(np.nonzero(img["heaptids"][:, 0] == tids[7])[0][0])
    assert img["heaptids"][s7, 1] == tids[100]
