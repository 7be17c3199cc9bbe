"""GPU parity, round 2: the fused single-query path (pgv_query_*), the MFMA assignment, the tile
kernel instantiations BASELINE's shapes select, spherical k-means, the reference's own index-order
transcripts through the GPU, HNSW at configs[3]'s shape.  Same contract as test_gpu_parity.py:
integers/indexes exact, distances within 1e-5 relative, ties compared as sets."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import pyoracle as po
from pgvector_amd import api

from helpers import RTOL, CpuIvf, assert_close, assert_topk_equiv, gen, golden, normalize_rows

pytestmark = pytest.mark.gpu

DT = {po.ORA_F32: api.PGV_F32, po.ORA_F16: api.PGV_F16}
POPS = {po.OPS_L2: api.PGV_OPS_L2, po.OPS_IP: api.PGV_OPS_IP, po.OPS_COSINE: api.PGV_OPS_COSINE}


def _upload(ctx, ivf):
    return api.IvfIndex(ctx, ivf.metric, DT[ivf.dtype], ivf.vectors.shape[1], ivf.centers, ivf.list_offsets,
                        ivf.vectors, ivf.tids)


def ora_dist(oracle, ops, dtype, q, rows):
    return np.array([oracle.lib.ora_index_distance(ops, dtype, rows.shape[1], po._p(r), po._p(q)) for r in rows])


# ------------------------------------------------ one query at a time (amgettuple)
@pytest.mark.parametrize("ops,dtype,dim", [
    (po.OPS_L2, po.ORA_F32, 64),     # 16 lanes per row
    (po.OPS_L2, po.ORA_F32, 100),    # 25 vectors: partial last slice
    (po.OPS_IP, po.ORA_F32, 256),    # whole 1 KiB rows, query in registers
    (po.OPS_L2, po.ORA_F32, 1536),   # BASELINE headline row shape (6 slices)
    (po.OPS_L2, po.ORA_F16, 512),
    (po.OPS_L2, po.ORA_F16, 3072),   # configs[4] row shape
    (po.OPS_IP, po.ORA_F16, 72),
    (po.OPS_L2, po.ORA_F32, 3),
])
def test_query_path_is_gettuple(ctx, oracle, ops, dtype, dim):
    """pgv_query_rank / scan / more against GetScanLists + GetScanItems + the ascending tuplesort
    stream of the oracle (src/ivfscan.c:47-187), iterative batches included"""
    n, lists = (3000, 24) if dim > 1000 else (9000, 40)
    data = gen(n, dim, seed=301, dist="clustered", clusters=lists, dtype=dtype)
    ivf = CpuIvf(oracle, ops, dtype, data, lists)
    ix = _upload(ctx, ivf)
    qh = api.Query(ix)
    queries = gen(9, dim, seed=302, dist="clustered", clusters=lists, dtype=dtype)
    for qi, q in enumerate(queries):
        probes, maxp = (1, 5) if qi % 3 == 0 else ((3, 9) if qi % 3 == 1 else (lists, lists))
        qh.rank(q, maxp)
        wl, wd = oracle.get_scan_lists(ivf.struct, q, maxp)
        gl = qh.lists(maxp)
        assert_topk_equiv(gl.tolist(), ora_dist(oracle, ops, dtype, q, ivf.centers[gl]), wl.tolist(), wd,
                          what="query rank dim %d q %d" % (dim, qi))
        # first batch: head of the sorted stream, then deeper
        wd_all, ws_all = oracle.get_scan_items(ivf.struct, q, gl[:probes])
        d, s, t, total = qh.scan(0, probes, 32)
        assert total == len(ws_all)
        m = min(32, total)
        assert_topk_equiv(s.tolist(), d, ws_all[:m].tolist(), wd_all[:m], what="query head dim %d q %d" % (dim, qi))
        np.testing.assert_array_equal(t, ivf.tids[s])
        if total > 32:
            d2, s2, t2 = qh.more(32, 200)
            m2 = min(232, total)
            assert_topk_equiv(np.concatenate([s, s2]).tolist(), np.concatenate([d, d2]), ws_all[:m2].tolist(),
                              wd_all[:m2], what="query more dim %d q %d" % (dim, qi))
        # iterative scan: the next batch of `probes` lists, sorted on its own (src/ivfscan.c:400-406)
        if maxp >= 2 * probes:
            wd_b, ws_b = oracle.get_scan_items(ivf.struct, q, gl[probes:2 * probes])
            d, s, _, total = qh.scan(probes, probes, 64)
            assert total == len(ws_b)
            m = min(64, total)
            assert_topk_equiv(s.tolist(), d, ws_b[:m].tolist(), wd_b[:m], what="query batch 2 dim %d q %d" % (dim, qi))
    # NULL query: ZeroDistance (src/ivfscan.c:192-196) -- the first lists, every tuple at distance 0, page order
    qh.rank(None, 3)
    assert qh.lists(3).tolist() == [0, 1, 2]
    d, s, _, total = qh.scan(0, 3, 50)
    assert total == int(ivf.list_offsets[3]) and (d == 0).all() and s.tolist() == list(range(min(50, total)))
    qh.close()
    ix.close()


def test_query_path_ties_and_padding(ctx, oracle):
    """equal distances keep insertion order (page-chain order), short batches report their true count,
    NaN distances sort last like float8"""
    vec = np.array([[0, 0], [1, 0], [1, 0], [0, 1], [np.nan, 0], [2, 2], [1, 0]], dtype=np.float32)
    centers = np.array([[0, 0], [5, 5]], dtype=np.float32)
    off = np.array([0, 7, 7], dtype=np.int64)
    tids = np.arange(100, 107, dtype=np.uint64)
    ix = api.IvfIndex(ctx, api.PGV_L2SQ, api.PGV_F32, 2, centers, off, vec, tids)
    qh = api.Query(ix)
    qh.rank(np.array([0, 0], dtype=np.float32), 2)
    assert qh.lists(2).tolist() == [0, 1]
    d, s, t, total = qh.scan(0, 2, 64)
    assert total == 7 and s.tolist() == [0, 1, 2, 3, 6, 5, 4] and np.isnan(d[-1]) and t.tolist() == (s + 100).tolist()
    d, s, _, total = qh.scan(1, 1, 8)  # an empty list
    assert total == 0 and len(s) == 0
    with pytest.raises(api.PgvError):
        qh.scan(0, 3, 8)
    qh.close()
    ix.close()


def test_gettuple_pulls_deep_into_a_batch(ctx, oracle):
    """the executor may pull every tuple: head (64) -> device refills (to 1024) -> the whole batch on the host;
    the stream must be the oracle's sorted order throughout"""
    from pgvector_amd import _host
    n, dim, lists = 5000, 32, 4
    data = gen(n, dim, seed=311)
    ivf = CpuIvf(oracle, po.OPS_L2, po.ORA_F32, data, lists)
    rel = _host.Relation()
    rel.write_index(api.PGV_F32, ivf.centers, ivf.list_offsets, ivf.vectors, ivf.tids)
    img = rel.stage(api.PGV_F32)
    mirror = api.IvfIndex(ctx, api.PGV_L2SQ, api.PGV_F32, dim, img.centers, img.list_offsets, img.vectors, img.tids)
    scan = _host.IvfScan(mirror, img, probes=2)
    q = gen(1, dim, seed=312)[0]
    scan.rescan(q)
    got_t, got_d = scan.fetch()
    wt, wd = oracle.search(ivf.struct, q, 2, n)
    assert len(got_t) == len(wt) > 1100
    assert_topk_equiv(got_t.tolist(), got_d, wt.tolist(), wd, what="deep pull")
    scan.close()
    mirror.close()


# --------------------------------------------------------- tile kernel shapes of BASELINE
@pytest.mark.parametrize("ops,dtype,dim,nq,probes", [(po.OPS_L2, po.ORA_F16, 3072, 70, 5),    # tile<half, L2, 6>: configs[4]
                                                     (po.OPS_IP, po.ORA_F32, 1536, 70, 5),    # tile<float, IP, 6>: configs[2]
                                                     (po.OPS_IP, po.ORA_F16, 1024, 100, 4)])  # tile<half, IP, 2>
def test_tile_kernel_baseline_shapes(ctx, oracle, ops, dtype, dim, nq, probes):
    n, lists = 2400, 10
    data = gen(n, dim, seed=321, dist="clustered", clusters=lists, dtype=dtype)
    ivf = CpuIvf(oracle, ops, dtype, data, lists)
    ix = _upload(ctx, ivf)
    queries = gen(nq, dim, seed=322, dist="clustered", clusters=lists, dtype=dtype)
    dist, slot, tid = ix.search_batch(queries, probes, 10, want_tid=True)
    for i in range(nq):
        wt, wd = oracle.search(ivf.struct, queries[i], probes, 10)
        assert_topk_equiv(tid[i][slot[i] >= 0].tolist(), dist[i][:len(wt)], wt.tolist(), wd,
                          what="tile ops %d dim %d q %d" % (ops, dim, i))
    ix.close()


# ----------------------------------------------------------------- MFMA assignment
def _check_assign(oracle, ops, dtype, dim, rows, centers, got, gd, max_ties):
    want, wd = oracle.assign(ops, dtype, centers, rows)
    scale = 1.0
    if ops != po.OPS_L2:
        scale = float(np.max(np.abs(rows.astype(np.float64)) @ np.abs(centers.astype(np.float64)).T))
    assert_close(gd, wd, rtol=RTOL, atol=RTOL * scale if ops != po.OPS_L2 else 1e-30, what="assign distance")
    diff = np.nonzero(got != want)[0]
    for r in diff:  # a different list only on a float-level tie
        d = np.array([oracle.lib.ora_index_distance(ops, dtype, dim, po._p(rows[r]), po._p(centers[c]))
                      for c in (got[r], want[r])])
        assert abs(d[0] - d[1]) <= 4 * RTOL * max(abs(d).max(), scale if ops != po.OPS_L2 else 1e-30), (r, got[r], want[r], d)
    assert len(diff) <= max_ties, (len(diff), max_ties)


@pytest.mark.parametrize("ops,dtype,dim,k,n,dist", [
    (po.OPS_L2, po.ORA_F32, 96, 300, 2500, "clustered"),   # pre-filter + exact recheck
    (po.OPS_L2, po.ORA_F32, 130, 257, 1111, "uniform"),    # near ties: the exact fallback list is exercised; dim tail
    (po.OPS_IP, po.ORA_F32, 100, 200, 3001, "normal"),     # fp32 MFMA is the reference's own arithmetic
    (po.OPS_L2, po.ORA_F16, 200, 513, 1500, "clustered"),  # 32x32x16 f16 MFMA, 256-wide tiles, ragged edges
    (po.OPS_IP, po.ORA_F16, 64, 64, 700, "normal"),
    (po.OPS_L2, po.ORA_F32, 1536, 128, 600, "clustered"),  # BASELINE row shape
])
def test_assign_on_the_matrix_cores(ctx, oracle, ops, dtype, dim, k, n, dist):
    rows = gen(n, dim, seed=331, dist=dist, dtype=dtype, clusters=40)
    centers = gen(k, dim, seed=332, dist=dist, dtype=dtype, clusters=40)
    metric = api.PGV_L2SQ if ops == po.OPS_L2 else api.PGV_NEG_IP
    got, gd = api.assign(ctx, metric, DT[dtype], dim, centers, rows)
    _check_assign(oracle, ops, dtype, dim, rows, centers, got, gd, max(2, n // 400))


def test_assign_mfma_exact_ties_and_specials(ctx, oracle):
    """integer data: exact arithmetic in the pre-filter too, so every tie reaches the exact recheck and the
    lowest id wins; a NaN / inf center is never selected; a row of NaN stays in list 0"""
    rows = gen(1500, 32, seed=341, dist="int")
    centers = gen(200, 32, seed=342, dist="int")
    for a, b in [(9, 3), (150, 3), (77, 20), (199, 198)]:
        centers[a] = centers[b]  # duplicates inside and across the 128-wide center tiles
    got, gd = api.assign(ctx, api.PGV_L2SQ, api.PGV_F32, 32, centers, rows)
    want, wd = oracle.assign(po.OPS_L2, po.ORA_F32, centers, rows)
    np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(gd.astype(np.float64), wd)
    got, gd = api.assign(ctx, api.PGV_NEG_IP, api.PGV_F32, 32, centers, rows)
    want, wd = oracle.assign(po.OPS_IP, po.ORA_F32, centers, rows)
    np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(gd.astype(np.float64), wd)
    c2 = centers.copy()
    c2[5, 3] = np.nan
    c2[60, 0] = np.inf
    r2 = rows.copy()
    r2[10, :] = np.nan
    for metric, ops in [(api.PGV_L2SQ, po.OPS_L2), (api.PGV_NEG_IP, po.OPS_IP)]:
        got, _ = api.assign(ctx, metric, api.PGV_F32, 32, c2, r2)
        want, _ = oracle.assign(ops, po.ORA_F32, c2, r2)
        np.testing.assert_array_equal(got, want)
        assert got[10] == 0 and 5 not in got.tolist()


# ----------------------------------------------------------------- spherical k-means
def _unit_lattice(n, dim, seed):
    """unit vectors with four entries of +-0.5: every inner product is a multiple of 0.25, exact in fp32"""
    rng = np.random.default_rng(seed)
    x = np.zeros((n, dim), dtype=np.float32)
    for i in range(n):
        x[i, rng.choice(dim, 4, replace=False)] = rng.choice([-0.5, 0.5], 4)
    return x


@pytest.mark.parametrize("ops", [po.OPS_IP, po.OPS_COSINE])
def test_kmeanspp_spherical_picks_match_on_exact_data(ctx, oracle, ops):
    """InitCenters under vector_spherical_distance (src/ivfkmeans.c:24-91, src/vector.c:705-722): exact inner
    products -> the same acos(ip)/pi -> the same D^2 walk given the same pg_prng stream"""
    samples = _unit_lattice(1200, 16, seed=351)
    st = oracle.prng(77)
    rng = api.make_rng(next_double=oracle.lib.ora_prng_double_cb, next_u32=oracle.lib.ora_prng_u32_cb,
                       state=C.cast(C.pointer(st), C.c_void_p))
    got = api.kmeanspp_init(ctx, POPS[ops], api.PGV_F32, 16, samples, 30, rng)
    want = oracle.kmeans_init_centers(ops, po.ORA_F32, samples, 30, oracle.prng(77))
    np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("ops,dtype,k", [(po.OPS_IP, po.ORA_F32, 25), (po.OPS_COSINE, po.ORA_F32, 80),
                                         (po.OPS_L2, po.ORA_F16, 25), (po.OPS_IP, po.ORA_F16, 70)])
def test_kmeans_end_to_end_other_opclasses(ctx, oracle, ops, dtype, k):
    """statistical parity for the spherical and halfvec builds (k >= 64 takes the MFMA assignment): same
    stopping rule, objective within a few % of the reference's Elkan run from the same seed stream"""
    dim = 24
    samples = gen(6000, dim, seed=361, dist="clustered", clusters=k, dtype=dtype)
    if ops != po.OPS_L2:
        samples = normalize_rows(oracle, samples, dtype)
    st = oracle.prng(11)
    rng = api.make_rng(next_double=oracle.lib.ora_prng_double_cb, next_u32=oracle.lib.ora_prng_u32_cb,
                       state=C.cast(C.pointer(st), C.c_void_p))
    centers, closest, iters = api.kmeans(ctx, POPS[ops], DT[dtype], dim, samples, k, rng)
    wc, wcl, wit = oracle.kmeans(ops, dtype, samples, k, oracle.prng(11))
    assert 1 <= iters <= 500 and np.isfinite(centers.astype(np.float32)).all()
    s64 = samples.astype(np.float64)

    def objective(c, a):
        c = c.astype(np.float64)
        if ops == po.OPS_L2:
            return float(((s64 - c[a]) ** 2).sum())
        return float((1.0 - (s64 * c[a]).sum(axis=1)).sum())  # spherical: 1 - cos
    assert objective(centers, closest) <= 1.05 * objective(wc, wcl) + 1e-6
    if ops != po.OPS_L2:
        np.testing.assert_allclose(np.linalg.norm(centers.astype(np.float64), axis=1), 1.0,
                                   rtol=2e-3 if dtype == po.ORA_F16 else 1e-6)
    again, _ = oracle.lloyd_assign(ops, dtype, samples, centers)
    assert (again != closest).mean() < 0.004


# -------------------------------------------- the reference's transcripts through the GPU
IVF_CASES = [c for c in golden("index_order.json")["cases"] if c["am"] == "ivfflat"]


@pytest.mark.parametrize("case", IVF_CASES, ids=[c["source"].split("/")[-1] for c in IVF_CASES])
def test_ivfflat_transcripts_on_the_gpu(ctx, oracle, case):
    """test/expected/ivfflat_vector.out / ivfflat_halfvec.out: CREATE INDEX over the first three rows
    (pgv_host_ivf_build: k-means + assignment on the GPU, page writer), INSERT of the fourth, then the
    ORDER BY query through stage -> mirror -> ivfflatgettuple; the row order must be the transcript's"""
    from pgvector_amd import _host
    dtype = po.ORA_F16 if case["type"] == "halfvec" else po.ORA_F32
    ops = {"l2": po.OPS_L2, "ip": po.OPS_IP, "cosine": po.OPS_COSINE}[case["ops"]]
    rows = np.asarray(case["rows"], dtype=po.NP_OF[dtype])
    dim = rows.shape[1]
    tids = (np.arange(len(rows), dtype=np.uint64) << np.uint64(16)) | np.uint64(1)
    built = rows[:3]
    samples = built
    if ops == po.OPS_COSINE:
        samples = samples[np.abs(samples.astype(np.float32)).sum(axis=1) > 0]
    if ops != po.OPS_L2:
        samples = normalize_rows(oracle, np.ascontiguousarray(samples), dtype)
    rel = _host.Relation()
    rel.build(ctx, POPS[ops], DT[dtype], case["lists"], built, tids[:3], samples, api.make_rng(seed=42))
    # aminsert of the later rows: nearest list under FUNCTION 1, appended (src/ivfinsert.c:72-181)
    img0 = rel.stage(DT[dtype])
    for r in range(3, len(rows)):
        v = rows[r]
        if ops == po.OPS_COSINE:
            if not np.abs(v.astype(np.float32)).sum() > 0:
                continue
            v = normalize_rows(oracle, v[None, :], dtype)[0]
        lst, _ = api.assign(ctx, api.PGV_L2SQ if ops == po.OPS_L2 else api.PGV_NEG_IP, DT[dtype], dim,
                            np.ascontiguousarray(img0.centers), v[None, :])
        rel.insert(DT[dtype], int(lst[0]), v, int(tids[r]))
    img = rel.stage(DT[dtype])
    metric = api.PGV_L2SQ if ops == po.OPS_L2 else api.PGV_NEG_IP
    mirror = api.IvfIndex(ctx, metric, DT[dtype], dim, img.centers, img.list_offsets, img.vectors, img.tids)

    def run(query, probes, limit=None, **kw):
        scan = _host.IvfScan(mirror, img, probes=probes, normalize_query=(ops == po.OPS_COSINE), **kw)
        scan.rescan(None if query is None else np.asarray(query, dtype=po.NP_OF[dtype]))
        t, _ = scan.fetch(limit)
        scan.close()
        return [rows[int(x) >> 16].astype(np.float32).tolist() for x in t]

    if "self_nearest" in case:
        for v in case["self_nearest"]:
            assert run(v, case["probes"], limit=1) == [[float(x) for x in v]]
    elif case.get("iterative"):
        got = run(case["query"], case["probes"], max_probes=case["max_probes"], iterative=True)
        assert got == [[float(x) for x in v] for v in case["expect"]]
    else:
        got = run(case["query"], case["probes"])
        if "expect_count" in case:
            assert len(got) == case["expect_count"]
        else:
            assert got == [[float(x) for x in v] for v in case["expect"]]
    mirror.close()


HNSW_CASES = [c for c in golden("index_order.json")["cases"] if c["am"] == "hnsw"]


@pytest.mark.parametrize("case", HNSW_CASES, ids=[c["source"].split("/")[-1] for c in HNSW_CASES])
def test_hnsw_transcripts_on_the_gpu(ctx, oracle, case):
    """test/expected/hnsw_vector.out: the graph built by the GPU build loop, searched on the device"""
    from pgvector_amd import _host
    ops = {"l2": po.OPS_L2, "ip": po.OPS_IP, "cosine": po.OPS_COSINE, "l1": po.OPS_L1}[case["ops"]]
    metric = {po.OPS_L2: api.PGV_L2SQ, po.OPS_L1: api.PGV_L1}.get(ops, api.PGV_NEG_IP)
    rows = np.asarray(case["rows"], dtype=np.float32)
    stored = rows
    keep = np.arange(len(rows))
    if ops == po.OPS_COSINE:
        keep = np.nonzero(np.abs(rows).sum(axis=1) > 0)[0]
        stored = normalize_rows(oracle, np.ascontiguousarray(rows[keep]), po.ORA_F32)
    mirror = api.Hnsw(ctx, metric, api.PGV_F32, rows.shape[1], stored)
    g = _host.hnsw_build(mirror, stored, 16, 64, api.make_rng(seed=1), max_batch=1)
    mirror.set_graph(16, g["entry"], g["levels"], g["nbr_start"], g["nbr"])
    q = np.asarray(case["query"], dtype=np.float32)[None, :]
    if ops == po.OPS_COSINE:
        q = normalize_rows(oracle, q, po.ORA_F32)
    elem, dist, _ = mirror.search(q, 40, len(stored))
    got = [rows[keep[e]].tolist() for e in elem[0] if e >= 0]
    assert got == [[float(x) for x in v] for v in case["expect"]]
    mirror.close()


# ----------------------------------------------------------- HNSW at configs[3]'s shape
def test_hnsw_cosine_1536_m16_ef100(ctx, oracle):
    """BASELINE configs[3] scaled in rows only: vector_cosine_ops, 1536-d, m 16, ef_search 100, the oracle's
    graph walked on the device"""
    n, dim, m, ef = 2500, 1536, 16, 100
    data = gen(n, dim, seed=371, dist="clustered", clusters=50)
    g = po.HnswGraph(oracle, po.OPS_COSINE, po.ORA_F32, data, m=m, ef_construction=64, seed=9)
    ex = g.export_tuples()
    stored = normalize_rows(oracle, np.ascontiguousarray(data[ex["rows"]]), po.ORA_F32)
    mirror = api.Hnsw(ctx, api.PGV_NEG_IP, api.PGV_F32, dim, stored)
    mirror.set_graph(m, ex["entry"], ex["levels"], ex["nbr_start"], ex["nbr"])
    queries = gen(24, dim, seed=372, dist="clustered", clusters=50)
    gq = normalize_rows(oracle, queries, po.ORA_F32)
    elem, gd, scored = mirror.search(gq, ef, 10)
    same = 0
    for i, q in enumerate(queries):
        rows, wd, wscored = g.search(q, ef, 10)
        got_rows = ex["rows"][elem[i][elem[i] >= 0]]
        assert_topk_equiv(got_rows.tolist(), gd[i][:len(got_rows)], rows.tolist(), wd, rtol=RTOL,
                          what="hnsw cosine 1536 q %d" % i)
        same += int(scored[i] == wscored)
    assert same >= 20, same
    mirror.close()


# ------------------------------------------------ full-size assignment through a size-independent property
@pytest.mark.parametrize("tname,metric,n,k,dim", [("f16", api.PGV_L2SQ, 300_000, 1024, 3072),
                                                  ("f16", api.PGV_NEG_IP, 300_000, 1024, 1024),
                                                  ("f32", api.PGV_L2SQ, 400_000, 1000, 768)])
def test_assign_at_scale_is_the_fp64_argmin(ctx, tname, metric, n, k, dim):
    """BASELINE-scale assignment (no CPU oracle at this size): the chosen center of every row of two slabs must be
    the float64 argmin up to the float tolerance of the distances.  More workgroups than CUs: a workgroup follows
    another on the same CU, which is what exposed the missing LDS-DMA wait of the fp16 MFMA pipeline in round 2
    (short stages: ds_read overtook the fill, ~2 % of the rows went to a neighbouring center)."""
    import torch
    g = torch.Generator(device="cuda")
    g.manual_seed(5)
    tdt = torch.float16 if tname == "f16" else torch.float32
    dt = api.PGV_F16 if tname == "f16" else api.PGV_F32
    means = torch.rand((k // 4, dim), generator=g, device="cuda")
    rows = torch.empty((n, dim), device="cuda", dtype=tdt)
    for lo in range(0, n, 1 << 16):
        hi = min(n, lo + (1 << 16))
        comp = torch.randint(0, k // 4, (hi - lo,), generator=g, device="cuda")
        rows[lo:hi] = (means[comp] + 0.1 * torch.randn((hi - lo, dim), generator=g, device="cuda")).to(tdt)
    centers = rows[torch.randperm(n, generator=g, device="cuda")[:k]].contiguous()
    got, gd = api.assign(ctx, metric, dt, dim, centers, rows, want_dist=True)
    ctx.sync()
    c64 = centers.double()
    for lo in (0, n - 5000):
        r64 = rows[lo:lo + 5000].double()
        ref = torch.cdist(r64, c64).pow(2) if metric == api.PGV_L2SQ else -(r64 @ c64.T)
        best = ref.min(dim=1).values
        mine = ref.gather(1, got[lo:lo + 5000].long()[:, None])[:, 0]
        scale = best.abs() + (1.0 if metric == api.PGV_L2SQ else float(dim))
        assert int(((mine - best) > 1e-5 * scale).sum()) == 0
        torch.testing.assert_close(gd[lo:lo + 5000].double(), mine, rtol=1e-5, atol=1e-5 * float(scale.max()))


# ------------------------------------------------------------------ multi-GPU path in the library
def test_comm_group_of_one_goes_through_rccl(ctx, oracle):
    """pgv_comm_create with a unique id and one rank: librccl is resolved, a communicator initialised and every
    collective of pgv_kmeans_sharded / pgv_search_batch_sharded issued through it -- results identical to the
    single-GPU entry points"""
    comm = api.Comm(ctx, backend="rccl")
    data = gen(5000, 32, seed=381, dist="clustered", clusters=25)
    c1, cl1, it1 = api.kmeans(ctx, api.PGV_OPS_L2, api.PGV_F32, 32, data, 25, api.make_rng(seed=4))
    c2, cl2, it2 = comm.kmeans(api.PGV_OPS_L2, api.PGV_F32, 32, data, 25, api.make_rng(seed=4))
    np.testing.assert_array_equal(c1, c2)
    np.testing.assert_array_equal(cl1, cl2)
    assert it1 == it2
    ivf = CpuIvf(oracle, po.OPS_L2, po.ORA_F32, data, 25, centers=c1)
    ix = _upload(ctx, ivf)
    q = gen(20, 32, seed=382, dist="clustered", clusters=25)
    d1, _, t1 = ix.search_batch(q, 3, 10, want_tid=True)
    d2, t2 = comm.search_batch(ix, q, 3, 10)
    np.testing.assert_array_equal(d1, d2)
    np.testing.assert_array_equal(t1, t2)
    ix.close()
    comm.close()


def test_comm_two_ranks_on_one_gpu():
    """the whole multi-GPU path of the library with two processes sharing this GPU (collectives: host callbacks
    over gloo; RCCL itself needs two GPUs and is the driver's to run)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29611",
                        os.path.join(root, "tests", "mp_comm_worker.py")],
                       cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0 and "COMM-OK" in r.stdout, (r.returncode, r.stdout[-1000:], r.stderr[-3000:])


# ------------------------------------------------- the batched list scan on the matrix cores
def _batch_vs_oracle(ctx, oracle, ivf, queries, probes, k, what, expect_redo=None, gq=None):
    ix = _upload(ctx, ivf)
    gq = queries if gq is None else gq  # what GetScanValue hands on (normalised for cosine); the oracle takes the raw datum
    ctx.set_profiling(True)
    ctx.reset_stats()
    dist, slot, tid = ix.search_batch(gq, probes, k, want_tid=True)
    redo = ctx.stats()["scan_redo_queries"]
    ctx.set_profiling(False)
    ctx.set_exact_scan(True)
    try:
        edist, eslot, _ = ix.search_batch(gq, probes, k, want_tid=True)
    finally:
        ctx.set_exact_scan(False)
    for i in range(len(queries)):
        wt, wd = oracle.search(ivf.struct, queries[i], probes, k)
        have = slot[i] >= 0
        assert have.sum() == len(wt), (what, i, have.sum(), len(wt))
        assert_topk_equiv(tid[i][have].tolist(), dist[i][:len(wt)], wt.tolist(), wd, what="%s q %d" % (what, i))
        # and the same head as the exact kernels give, tie for tie apart from last-bit differences
        assert_topk_equiv(slot[i][have].tolist(), dist[i][:len(wt)], eslot[i][eslot[i] >= 0].tolist(),
                          edist[i][:len(wt)].astype(np.float64), what="%s vs exact q %d" % (what, i))
    if expect_redo == "none":
        assert redo == 0, redo
    elif expect_redo == "all":
        assert redo == len(queries), redo
    ix.close()
    return redo


@pytest.mark.parametrize("ops,dtype,dim,n,lists,nq,probes,k,dist", [
    (po.OPS_L2, po.ORA_F32, 1536, 3000, 12, 150, 4, 10, "clustered"),   # headline row shape; ragged query groups (150 = 4 x 32 + 22)
    (po.OPS_L2, po.ORA_F32, 100, 6000, 8, 90, 3, 10, "uniform"),        # partial last slice, near ties
    (po.OPS_L2, po.ORA_F32, 3, 4000, 5, 80, 2, 64, "normal"),           # one vector per row; k = 64 -> k' = 256
    (po.OPS_L2, po.ORA_F16, 3072, 2400, 10, 70, 5, 10, "clustered"),    # configs[4] row shape
    (po.OPS_L2, po.ORA_F16, 72, 5000, 6, 64, 6, 40, "normal"),
    (po.OPS_L2, po.ORA_F32, 40, 6000, 6, 70, 3, 100, "normal"),        # k = 100 -> k' = 164
    (po.OPS_L2, po.ORA_F32, 24, 6000, 5, 60, 2, 192, "uniform"),       # the largest head of this path: k' = 256
    (po.OPS_IP, po.ORA_F32, 1536, 2400, 10, 70, 5, 10, "clustered"),    # configs[2]: the MFMA value is the result
    (po.OPS_IP, po.ORA_F16, 1024, 2400, 10, 100, 4, 10, "clustered"),
    (po.OPS_COSINE, po.ORA_F32, 200, 4000, 10, 90, 3, 10, "clustered"),
])
def test_batched_scan_on_the_matrix_cores(ctx, oracle, ops, dtype, dim, n, lists, nq, probes, k, dist):
    """search_batch with >8 queries per list goes through mfma_scan_kernel (+ the exact tail for L2): the head
    must be GetScanItems + tuplesort's (src/ivfscan.c:124-187) and the exact kernels' head"""
    data = gen(n, dim, seed=401, dist=dist, clusters=lists, dtype=dtype)
    ivf = CpuIvf(oracle, ops, dtype, data, lists)
    queries = gen(nq, dim, seed=402, dist=dist, clusters=lists, dtype=dtype)
    gq = normalize_rows(oracle, queries, dtype) if ops == po.OPS_COSINE else None  # GetScanValue (src/ivfscan.c:222-229)
    _batch_vs_oracle(ctx, oracle, ivf, queries, probes, k, "mfma scan ops %d dim %d" % (ops, dim), gq=gq)


def test_batched_scan_mfma_ties_short_lists_and_specials(ctx, oracle):
    """exact ties keep stream order (integer data: the expansion is exact, nothing is redone); lists shorter than
    k'; an empty list; a NaN row and an inf row (flagged -> exact pass); a huge-norm row (the bound grows until
    every query is redone) -- results never change"""
    dim, lists = 8, 6
    data = gen(900, dim, seed=411, dist="int")
    data[100:140] = data[60]            # 40 equal rows
    centers = gen(lists, dim, seed=412, dist="int")
    centers[5] = 1000.0                 # nothing lands here: an empty list
    ivf = CpuIvf(oracle, po.OPS_L2, po.ORA_F32, data, lists, centers=centers)
    queries = gen(60, dim, seed=413, dist="int")
    queries[7] = data[60]
    # k = 10 -> k' = 40 candidates, inside the run of 41 equal rows for the queries near data[60]: the boundary
    # is a tie, those queries (and only such) take the exact pass
    redo = _batch_vs_oracle(ctx, oracle, ivf, queries, lists, 10, "int ties")
    assert redo <= 6, redo
    _batch_vs_oracle(ctx, oracle, ivf, queries, lists, 64, "int ties k=64", expect_redo="none")
    # fewer tuples than k (and than k'): a 30-row index
    small = CpuIvf(oracle, po.OPS_L2, po.ORA_F32, data[:30], 2, centers=centers[:2])
    _batch_vs_oracle(ctx, oracle, small, queries[:40], 2, 50, "short")
    # specials
    bad = gen(900, dim, seed=414, dist="normal")
    bad[5, 0] = np.nan
    bad[6, 1] = np.inf
    ivf2 = CpuIvf(oracle, po.OPS_L2, po.ORA_F32, bad, 3, centers=gen(3, dim, seed=415, dist="normal"))
    redo = _batch_vs_oracle(ctx, oracle, ivf2, gen(40, dim, seed=416, dist="normal"), 3, 10, "nan/inf rows")
    assert redo == 40   # max |x|^2 is inf: no bound, every query takes the exact pass
    big = gen(900, dim, seed=417, dist="normal")
    big[11] *= 3e4
    ivf3 = CpuIvf(oracle, po.OPS_L2, po.ORA_F32, big, 3, centers=gen(3, dim, seed=415, dist="normal"))
    _batch_vs_oracle(ctx, oracle, ivf3, gen(40, dim, seed=418, dist="normal"), 3, 10, "huge norm", expect_redo="all")


def test_batched_scan_mfma_at_scale_is_the_exact_scan(ctx):
    """workgroups that follow one another on a CU (LDS reuse, many tasks per workgroup): 200k x 768 rows,
    2000 queries -- slot for slot the exact kernels' answer, and (almost) nothing redone"""
    import torch
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    n, dim, lists, nq, probes = 200_000, 768, 100, 2000, 5
    means = torch.rand((50, dim), generator=g, device=dev)
    data = means[torch.randint(0, 50, (n,), generator=g, device=dev)] + 0.1 * torch.randn((n, dim), generator=g, device=dev)
    centers, _, _ = api.kmeans(ctx, api.PGV_OPS_L2, api.PGV_F32, dim, data[:20000].contiguous(), lists,
                               api.make_rng(seed=3), want_closest=False)
    assign, _ = api.assign(ctx, api.PGV_L2SQ, api.PGV_F32, dim, centers, data, want_dist=False)
    order = torch.argsort(assign.long(), stable=True)
    off = torch.zeros(lists + 1, dtype=torch.int64, device=dev)
    off[1:] = torch.cumsum(torch.bincount(assign.long(), minlength=lists), 0)
    ix = api.IvfIndex(ctx, api.PGV_L2SQ, api.PGV_F32, dim, centers, off, data[order].contiguous(), order)
    queries = (means[torch.randint(0, 50, (nq,), generator=g, device=dev)] + 0.1 * torch.randn((nq, dim), generator=g, device=dev))
    ctx.set_profiling(True)
    ctx.reset_stats()
    d, s, _ = ix.search_batch(queries, probes, 10, want_tid=True)
    redo = ctx.stats()["scan_redo_queries"]
    ctx.set_profiling(False)
    ctx.set_exact_scan(True)
    try:
        ed, es, _ = ix.search_batch(queries, probes, 10, want_tid=True)
    finally:
        ctx.set_exact_scan(False)
    d, s, ed, es = (np.asarray(x.cpu() if hasattr(x, "cpu") else x) for x in (d, s, ed, es))
    diff = np.nonzero((s != es).any(axis=1))[0]
    for i in diff:  # only a last-bit difference between two summation orders may swap neighbours
        assert sorted(s[i].tolist()) == sorted(es[i].tolist()) or np.allclose(d[i], ed[i], rtol=1e-6), i
    assert len(diff) <= nq // 100, len(diff)
    np.testing.assert_allclose(d, ed, rtol=1e-5)
    assert redo <= nq // 100, redo
    ix.close()


# ------------------------------------------------- center ranking of a batch on the matrix cores
@pytest.mark.parametrize("ops,dtype,dim,lists,dist", [
    (po.OPS_L2, po.ORA_F32, 1536, 300, "clustered"),   # headline row shape
    (po.OPS_L2, po.ORA_F32, 100, 1000, "uniform"),     # near ties between centers
    (po.OPS_L2, po.ORA_F16, 72, 70, "normal"),
    (po.OPS_IP, po.ORA_F32, 256, 200, "normal"),
    (po.OPS_IP, po.ORA_F16, 1024, 129, "clustered"),
])
def test_rank_lists_on_the_matrix_cores(ctx, oracle, ops, dtype, dim, lists, dist):
    """GetScanLists (src/ivfscan.c:47-118) for a batch: mfma_scan_kernel over the centers (+ exact recheck of
    maxprobes + 16 candidates for L2); maxprobes + 16 > 256 and pgv_ctx_set_exact_scan stay on the exact kernels"""
    centers = gen(lists, dim, seed=421, dist=dist, dtype=dtype, clusters=30)
    centers[7] = centers[3]                     # equal centers: the lower id first
    data = gen(lists * 3, dim, seed=422, dist=dist, dtype=dtype, clusters=30)
    ivf = CpuIvf(oracle, ops, dtype, data, lists, centers=centers)
    ix = _upload(ctx, ivf)
    queries = gen(45, dim, seed=423, dist=dist, dtype=dtype, clusters=30)
    queries[5] = centers[3]
    for probes in (1, 10, min(lists, 64), min(lists, 250)):
        lists_got, dist_got = ix.rank_lists(queries, probes)
        ctx.set_exact_scan(True)
        try:
            lists_exact, _ = ix.rank_lists(queries, probes)
        finally:
            ctx.set_exact_scan(False)
        for i, q in enumerate(queries):
            wl, wd = oracle.get_scan_lists(ivf.struct, q, probes)
            atol = 1e-30 if ops == po.OPS_L2 else RTOL * float(np.abs(q.astype(np.float64)) @ np.abs(centers.astype(np.float64)).max(axis=0))
            assert_topk_equiv(lists_got[i], dist_got[i], wl, wd, what="mfma rank ops %d probes %d q %d" % (ops, probes, i),
                              atol=atol)
        same = (lists_got == lists_exact).all(axis=1).mean()
        assert same >= 0.9, same   # only last-bit near-ties may order differently
    q5, _ = ix.rank_lists(queries, 4)
    assert q5[5].tolist()[:2] == [3, 7] or ops != po.OPS_L2
    ix.close()


def test_rank_lists_mfma_specials(ctx, oracle):
    """a center with a huge norm widens the bound until every query is redone exactly; a NaN center sorts last"""
    dim, lists = 16, 80
    centers = gen(lists, dim, seed=431, dist="normal")
    centers[9] *= 1e5
    centers[11, 0] = np.nan
    data = gen(400, dim, seed=432, dist="normal")
    ivf = CpuIvf(oracle, po.OPS_L2, po.ORA_F32, data, lists, centers=centers)
    ix = _upload(ctx, ivf)
    queries = gen(40, dim, seed=433, dist="normal")
    for probes in (5, lists):
        got, gd = ix.rank_lists(queries, probes)
        for i, q in enumerate(queries):
            wl, wd = oracle.get_scan_lists(ivf.struct, q, probes)
            assert_topk_equiv(got[i], gd[i], wl, wd, what="rank specials probes %d q %d" % (probes, i))
    ix.close()


# ------------------------------------------------- one device mirror, several backends
def test_index_shared_by_contexts_on_their_own_streams(ctx, oracle):
    """pgv_index_share: a second context (own stream, own scratch) scans the same device arrays; batches and
    single-query scans from two threads at once give the answers of the owner's serial scan"""
    import threading
    n, dim, lists, probes = 6000, 96, 30, 4
    data = gen(n, dim, seed=441, dist="clustered", clusters=lists)
    ivf = CpuIvf(oracle, po.OPS_L2, po.ORA_F32, data, lists)
    ix = _upload(ctx, ivf)
    queries = gen(400, dim, seed=442, dist="clustered", clusters=lists)
    want_d, want_s, _ = ix.search_batch(queries, probes, 10, want_tid=True)
    others = [api.Context(0) for _ in range(3)]   # streams of their own
    views = [ix.share(c) for c in others]
    got = [None] * len(views)
    single = [None] * len(views)

    def backend(i):
        v = views[i]
        for _ in range(5):
            got[i] = v.search_batch(queries, probes, 10, want_tid=True)
        qh = api.Query(v)
        out = []
        for q in queries[:60]:
            qh.rank(q, probes)
            d, s, t, total = qh.scan(0, probes, 10)
            out.append((d.copy(), s.copy()))
        qh.close()
        single[i] = out
    threads = [threading.Thread(target=backend, args=(i,)) for i in range(len(views))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for i in range(len(views)):
        np.testing.assert_array_equal(got[i][1], want_s)
        np.testing.assert_array_equal(got[i][0], want_d)
        for j, (d, s) in enumerate(single[i]):
            m = len(s)
            assert_topk_equiv(s.tolist(), d, want_s[j][:m].tolist(), want_d[j][:m].astype(np.float64),
                              what="shared index, backend %d query %d" % (i, j))
    # the device arrays go with the LAST handle: the uploaded index may be dropped first
    ix.close()
    d, s, _ = views[0].search_batch(queries, probes, 10, want_tid=True)
    np.testing.assert_array_equal(s, want_s)
    for v in views:
        v.close()
    for c in others:
        c.close()


def test_pooler_batches_backends_and_answers_like_search_batch(ctx, oracle):
    """pgv_host_pool_*: 24 threads hand in one query each, again and again; every answer is the row
    pgv_search_batch gives for that query (= GetScanItems + tuplesort, checked against the oracle above), and the
    queries did travel in batches"""
    import threading
    from pgvector_amd import _host
    n, dim, lists, probes, k = 6000, 96, 30, 4, 10
    data = gen(n, dim, seed=451, dist="clustered", clusters=lists)
    ivf = CpuIvf(oracle, po.OPS_L2, po.ORA_F32, data, lists)
    ix = _upload(ctx, ivf)
    queries = gen(240, dim, seed=452, dist="clustered", clusters=lists)
    want_d, _, want_t = ix.search_batch(queries, probes, k, want_tid=True)
    for i in (0, 17, 101):   # the batched rows are the reference's
        wt, wd = oracle.search(ivf.struct, queries[i], probes, k)
        assert_topk_equiv(want_t[i].tolist(), want_d[i], wt.tolist(), wd, what="pool baseline q %d" % i)
    pool = _host.Pool(ix, probes, k, max_batch=16, max_wait_us=200, lanes=2)
    got = {}
    errors = []

    def backend(t):
        try:
            for j in range(t, len(queries), 24):
                got[j] = pool.search(queries[j])
        except Exception as e:  # noqa: BLE001
            errors.append(e)
    threads = [threading.Thread(target=backend, args=(t,)) for t in range(24)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    st = pool.stats()
    assert st["queries"] == len(queries) and st["batches"] < len(queries), st
    for j in range(len(queries)):
        tid, dist = got[j]
        # a query may run in a batch small enough for the exact kernels or in one the MFMA path takes: same head
        assert_topk_equiv(tid.tolist(), dist, want_t[j].tolist(), want_d[j].astype(np.float64), what="pool q %d" % j)
    # one query alone in its batch
    tid, dist = pool.search(queries[3])
    assert_topk_equiv(tid.tolist(), dist, want_t[3].tolist(), want_d[3].astype(np.float64), what="pool single")
    pool.close()
    ix.close()
