"""GPU parity, round 3: the device mirror and the pooler across PROCESSES (a Postgres backend is a process,
src/ivfscan.c:252-296), the deterministic completeness bound of the MFMA L2 scan, the exact-scan top-k.
Same contract as test_gpu_parity.py: integers/indexes exact, distances within 1e-5 relative, ties as sets."""
import os

import numpy as np
import pytest

from oracle import pyoracle as po
from pgvector_amd import _host, api

from helpers import CpuIvf, assert_topk_equiv, gen, normalize_rows

pytestmark = pytest.mark.gpu


def _upload(ctx, ivf):
    dt = {po.ORA_F32: api.PGV_F32, po.ORA_F16: api.PGV_F16}[ivf.dtype]
    return api.IvfIndex(ctx, ivf.metric, dt, ivf.vectors.shape[1], ivf.centers, ivf.list_offsets, ivf.vectors, ivf.tids)


def _check_answers(oracle, ivf, queries, probes, k, ans_t, ans_d, what):
    nq = queries.shape[0]
    want = [oracle.search(ivf.struct, q, probes, k) for q in queries]
    for c in range(ans_t.shape[0]):
        for j in range(ans_t.shape[1]):
            qi = (c * 31 + j) % nq  # the query client c asks in iteration j (tools/pgv_backend.c)
            wt, wd = want[qi]
            got_t = ans_t[c, j][:len(wt)]
            assert (ans_t[c, j][len(wt):] == np.uint64(0xFFFFFFFFFFFFFFFF)).all(), what
            assert_topk_equiv(got_t.tolist(), ans_d[c, j][:len(wt)], wt.tolist(), wd, what="%s client %d query %d" % (what, c, j))


@pytest.mark.parametrize("owner", ["this_process", "another_process"])
def test_mirror_and_pooler_across_processes(ctx, oracle, owner):
    """ONE device mirror, many PROCESSES: the owner exports the index (pgv_index_export), lane servers and
    independent backends import it (pgv_index_import: hipIpc, no copy); GPU-less client processes hand single
    queries to the pooler through a shared segment (pgv_host_pool_*).  Every answer is the oracle's
    GetScanLists + GetScanItems + sorted head (src/ivfscan.c:47-187)."""
    n, dim, lists, probes, k = 20000, 96, 40, 4, 10
    data = gen(n, dim, seed=801, dist="clustered", clusters=lists)
    ivf = CpuIvf(oracle, po.OPS_L2, po.ORA_F32, data, lists)
    queries = gen(48, dim, seed=802, dist="clustered", clusters=lists)
    ix = _upload(ctx, ivf)
    img = None
    try:
        if owner == "another_process":
            img = _host.write_index_image("pgv_test_img_%d" % os.getpid(), ivf.metric, api.PGV_F32, dim, ivf.centers,
                                          ivf.list_offsets, ivf.vectors, ivf.tids)
        src = None if img else ix
        # (a) 6 GPU-less clients behind the pooler, its two lanes led by two more processes
        res, at, ad = _host.run_backend_processes(src, queries, probes, k, 1, 6, 30, max_batch=4, max_wait_us=100,
                                                  lanes=2, server_processes=True, verify=True, image_shm=img)
        assert res["processes"] == 6 + 2 + (1 if img else 0)
        assert 1.0 <= res["mean_batch"] <= 4.0
        _check_answers(oracle, ivf, queries, probes, k, at, ad, "pooled/" + owner)
        # (b) 4 independent backends: each imports the mirror and runs pgv_query_rank + pgv_query_scan on its own stream
        res, at, ad = _host.run_backend_processes(src, queries, probes, k, 0, 4, 30, verify=True, image_shm=img)
        _check_answers(oracle, ivf, queries, probes, k, at, ad, "independent/" + owner)
        if not img:
            # (c) lanes led by threads of the owning process (pgv_index_share views), clients still processes
            res, at, ad = _host.run_backend_processes(ix, queries, probes, k, 1, 5, 20, max_batch=8, max_wait_us=100,
                                                      lanes=2, server_processes=False, verify=True)
            _check_answers(oracle, ivf, queries, probes, k, at, ad, "lane threads")
    finally:
        if img:
            os.unlink("/dev/shm/" + img.lstrip("/"))
        ix.close()


def test_imported_mirror_occupies_hbm_once(ctx):
    """four processes scanning an 0.8 GB mirror take the HBM of their contexts and scratch, not of four copies"""
    n, dim, lists, probes, k = 200000, 1024, 64, 4, 10
    rng = np.random.default_rng(5)
    centers = rng.standard_normal((lists, dim), dtype=np.float32)
    vectors = rng.standard_normal((n, dim), dtype=np.float32)
    offs = (np.arange(lists + 1, dtype=np.int64) * (n // lists))
    offs[-1] = n
    tids = np.arange(n, dtype=np.uint64)
    ix = api.IvfIndex(ctx, api.PGV_L2SQ, api.PGV_F32, dim, centers, offs, vectors, tids)
    index_bytes = n * dim * 4
    queries = vectors[rng.choice(n, 16, replace=False)]
    want_d, _, want_t = ix.search_batch(queries, lists, k, want_tid=True)  # every list probed: the query finds itself
    res, at, ad = _host.run_backend_processes(ix, queries, lists, k, 0, 4, 8, verify=True)
    assert res["hbm_bytes_taken_by_children"] < 2 * index_bytes, res  # four copies would be >= 4 x
    for c in range(4):
        for j in range(8):
            assert at[c, j].tolist() == np.asarray(want_t)[(c * 31 + j) % 16].tolist()
    ix.close()


def test_export_import_argument_errors(ctx):
    data = gen(500, 16, seed=1, dist="normal")
    centers = data[:4].copy()
    offs = np.array([0, 100, 200, 300, 500], dtype=np.int64)
    ix = api.IvfIndex(ctx, api.PGV_L2SQ, api.PGV_F32, 16, centers, offs, data, None)
    h = ix.export()
    assert len(h) == 256
    with pytest.raises(api.PgvError) as e:  # the exporting process shares, it does not import
        api.IvfIndex.from_handle(ctx, h)
    assert e.value.code == api.PGV_ERR_STATE
    with pytest.raises(api.PgvError) as e:
        api.IvfIndex.from_handle(ctx, bytes(256))
    assert e.value.code == api.PGV_ERR_ARG
    ix.close()


def test_index_tids_of_scanned_slots(ctx, oracle):
    """pgv_index_tids: a backend that imported the mirror keeps no TID table; the slots of a whole-batch scan
    (pgv_scan_lists: one run per list) come back as the heap TIDs the index was staged with"""
    n, dim, lists = 6000, 32, 12
    data = gen(n, dim, seed=811, dist="clustered", clusters=lists)
    ivf = CpuIvf(oracle, po.OPS_L2, po.ORA_F32, data, lists)
    ix = _upload(ctx, ivf)
    rng = np.random.default_rng(1)
    slots = np.concatenate([np.arange(ivf.list_offsets[l], ivf.list_offsets[l + 1]) for l in (7, 2, 9)] +
                           [rng.integers(0, n, 300)]).astype(np.int64)
    got = ix.tids(slots)
    assert got.tolist() == np.asarray(ivf.tids, dtype=np.uint64)[slots].tolist()
    assert ix.tids(np.zeros(0, np.int64)).size == 0
    with pytest.raises(api.PgvError) as e:
        ix.tids(np.array([0, n], dtype=np.int64))
    assert e.value.code == api.PGV_ERR_ARG
    ix.close()


@pytest.mark.parametrize("dtype", ["f32", "f16"])
def test_hnsw_mirror_across_processes(ctx, oracle, tmp_path, dtype):
    """pgv_hnsw_export / pgv_hnsw_import: another PROCESS maps the elements and the graph (hipIpc, no copy) and its
    hnswgettuple batch (src/hnswscan.c:16-66, HnswSearchLayer src/hnswutils.c:838-951) equals the oracle's walk"""
    import subprocess
    import sys
    odt, gdt = (po.ORA_F32, api.PGV_F32) if dtype == "f32" else (po.ORA_F16, api.PGV_F16)
    n, dim, m = 3000, 48, 8
    data = gen(n, dim, seed=821, dist="normal")
    if dtype == "f16":
        data = data.astype(np.float16)
    g = po.HnswGraph(oracle, po.OPS_L2, odt, data, m=m, ef_construction=40, seed=5)
    ex = g.export_tuples()
    # per-element payload: what hnswgettuple needs of an element (src/hnswscan.c:293-311) -- here 10 heap TIDs' worth
    # of words and a count, derived from the row so that the importing process can be checked
    rows_of = np.asarray(ex["rows"], dtype=np.int64)
    payload = np.zeros((len(rows_of), 21), dtype=np.uint32)
    payload[:, 0] = 1 + (rows_of % 10)
    payload[:, 1:] = (rows_of[:, None] * 31 + np.arange(20)[None, :]).astype(np.uint32)
    mirror = api.Hnsw(ctx, api.PGV_L2SQ, gdt, dim, data[ex["rows"]], payload=payload)
    with pytest.raises(api.PgvError):  # nothing to search before the graph is set
        mirror.export()
    mirror.set_graph(m, ex["entry"], ex["levels"], ex["nbr_start"], ex["nbr"])
    queries = gen(12, dim, seed=822, dist="normal")
    if dtype == "f16":
        queries = queries.astype(np.float16)
    handle = mirror.export()
    with pytest.raises(api.PgvError) as e:  # the exporting process searches its own mirror
        api.Hnsw.from_handle(ctx, handle, gdt)
    assert e.value.code == api.PGV_ERR_STATE
    job, res = str(tmp_path / "job.npz"), str(tmp_path / "res.npz")
    np.savez(job, handle=np.frombuffer(handle, dtype=np.uint8), queries=queries, dtype=gdt, ef=40, k=10, words=21)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "mp_hnsw_import_worker.py"), job, res], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = np.load(res)
    assert int(out["readonly"]) == api.PGV_ERR_STATE  # an imported mirror is read-only
    own_elem, own_dist, _ = mirror.search(queries, 40, 10)
    assert out["elem"].tolist() == np.asarray(own_elem).tolist()
    # the importer read the payload of its results out of the exporter's allocation: no table of its own
    flat = out["elem"].ravel()
    want = np.where(flat[:, None] >= 0, payload[np.clip(flat, 0, None)], 0)
    assert out["payload"].tolist() == want.tolist()
    assert mirror.get_payload(np.array([0, -1, 5])).tolist() == [payload[0].tolist(), [0] * 21, payload[5].tolist()]
    for i, q in enumerate(queries):
        rows, wd, _ = g.search(q, 40, 10)
        el = out["elem"][i]
        assert_topk_equiv(ex["rows"][el[el >= 0]].tolist(), out["dist"][i][:len(rows)], rows.tolist(), wd,
                          what="imported hnsw q%d" % i)
    mirror.close()


# ------------------------------------------------- the exact scan for a batch of queries
DT = {po.ORA_F32: api.PGV_F32, po.ORA_F16: api.PGV_F16}


def _ora_exact_topk(oracle, ops, dtype, rows, q, k):
    """the sequential scan as the oracle has it: GetScanItems + the sorted head over ONE list holding every row
    (the same per-row FUNCTION 1 calls and ascending sort an index-less ORDER BY ... LIMIT k makes)"""
    n = rows.shape[0]
    s = oracle.index_struct(ops, dtype, rows[:1], np.array([0, n], dtype=np.int64), rows, np.arange(n, dtype=np.uint64))
    return oracle.search(s, q, 1, k)


@pytest.mark.parametrize("ops,dtype,dim,n,nq,k,dist", [
    (po.OPS_L2, po.ORA_F32, 128, 10000, 100, 10, "uniform"),     # BASELINE configs[0]: 10k x 128, 100 queries
    (po.OPS_L2, po.ORA_F32, 1536, 3000, 70, 10, "clustered"),    # matrix cores + exact tail, ragged last query group
    (po.OPS_L2, po.ORA_F32, 96, 5000, 20, 10, "normal"),         # < 64 queries: the vector-ALU kernels
    (po.OPS_L2, po.ORA_F32, 3, 4000, 64, 64, "normal"),          # k = 64 -> k' = 256
    (po.OPS_L2, po.ORA_F32, 40, 3000, 64, 300, "normal"),        # k' would exceed 256: exact kernels
    (po.OPS_L2, po.ORA_F16, 3072, 2000, 66, 10, "clustered"),
    (po.OPS_L2, po.ORA_F16, 72, 5000, 9, 40, "normal"),
    (po.OPS_IP, po.ORA_F32, 768, 4000, 100, 10, "clustered"),    # the MFMA value is the result
    (po.OPS_IP, po.ORA_F16, 1024, 2400, 64, 25, "clustered"),
    (po.OPS_IP, po.ORA_F32, 50, 2000, 5, 10, "normal"),
])
def test_exact_topk_is_the_sequential_scan(ctx, oracle, ops, dtype, dim, n, nq, k, dist):
    """pgv_exact_topk against the oracle's per-row distance calls + ascending sort (src/vector.c:579-620, the
    executor's top-N): ids exact up to float ties, distances to 1e-5"""
    rows = gen(n, dim, seed=901, dist=dist, dtype=dtype)
    queries = gen(nq, dim, seed=902, dist=dist, dtype=dtype)
    metric = api.PGV_L2SQ if ops == po.OPS_L2 else api.PGV_NEG_IP
    dist_, idx = api.exact_topk(ctx, metric, DT[dtype], dim, queries, rows, k)
    scale = 0.0
    for i in range(nq):
        wt, wd = _ora_exact_topk(oracle, ops, dtype, rows, queries[i], k)
        if metric == api.PGV_NEG_IP:  # inner products cancel: tolerance relative to the size of the terms
            scale = 1e-5 * float(np.max(np.abs(rows.astype(np.float64)) @ np.abs(queries[i].astype(np.float64))))
        assert_topk_equiv(idx[i].tolist(), dist_[i], wt.tolist(), wd, atol=max(scale, 1e-30),
                          what="exact topk ops %d dim %d q %d" % (ops, dim, i))


def test_exact_topk_l1_ties_and_edges(ctx, oracle):
    """L1 (l1_distance, src/vector.c:715-725) goes through the exact kernels; exact ties come back by row index;
    fewer rows than k pads with +inf / -1; n = 0; device buffers give what host buffers give"""
    import torch
    dim = 24
    rows = gen(3000, dim, seed=911, dist="normal")
    queries = gen(70, dim, seed=912, dist="normal")
    d, ix = api.exact_topk(ctx, api.PGV_L1, api.PGV_F32, dim, queries, rows, 10)
    for i in range(0, 70, 7):
        want = np.array([oracle.lib.ora_index_distance(po.OPS_L1, po.ORA_F32, dim, po._p(r), po._p(queries[i])) for r in rows])
        order = np.argsort(want, kind="stable")[:10]
        assert_topk_equiv(ix[i].tolist(), d[i], order.tolist(), want[order], what="l1 q %d" % i)
    # integer data: many exact ties, every one resolved towards the lower row index, on both kernel families
    irows = gen(4000, 8, seed=913, dist="int")
    irows[500:560] = irows[17]
    for nq in (8, 96):
        iq = gen(nq, 8, seed=914, dist="int")
        iq[3] = irows[17]
        d, ix = api.exact_topk(ctx, api.PGV_L2SQ, api.PGV_F32, 8, iq, irows, 40)
        for i in range(nq):
            want = ((irows.astype(np.float64) - iq[i].astype(np.float64)) ** 2).sum(1)  # exact in fp32 too
            order = np.lexsort((np.arange(4000), want))[:40]
            assert ix[i].tolist() == order.tolist(), (nq, i)
            assert d[i].tolist() == want[order].tolist()
    # fewer rows than k, and none
    d, ix = api.exact_topk(ctx, api.PGV_L2SQ, api.PGV_F32, dim, queries, rows[:6], 10)
    assert (ix[:, 6:] == -1).all() and np.isinf(d[:, 6:]).all() and (ix[:, :6] >= 0).all()
    assert sorted(ix[0, :6].tolist()) == list(range(6))
    d, ix = api.exact_topk(ctx, api.PGV_L2SQ, api.PGV_F32, dim, queries[:5], rows[:0], 3)
    assert (ix == -1).all() and np.isinf(d).all()
    # device buffers
    hd, hi = api.exact_topk(ctx, api.PGV_L2SQ, api.PGV_F32, dim, queries, rows, 10)
    dd, di = api.exact_topk(ctx, api.PGV_L2SQ, api.PGV_F32, dim, torch.from_numpy(queries).cuda(),
                            torch.from_numpy(rows).cuda(), 10)
    assert di.cpu().numpy().tolist() == hi.tolist() and dd.cpu().numpy().tolist() == hd.tolist()
    with pytest.raises(api.PgvError) as e:
        api.exact_topk(ctx, api.PGV_L2SQ, api.PGV_F32, dim, queries, rows, 5000)
    assert e.value.code == api.PGV_ERR_ARG


# ------------------------------------------------- the completeness bound of the MFMA L2 paths
def _search_with_bound(ctx, ix, queries, probes, k, worst_case):
    ctx.set_bound(worst_case)
    ctx.set_profiling(True)
    ctx.reset_stats()
    try:
        dist, slot, tid = ix.search_batch(queries, probes, k, want_tid=True)
        st = ctx.stats()
        redo = st["scan_redo_queries"]
        _search_with_bound.widened = st["scan_widened_queries"]
    finally:
        ctx.set_profiling(False)
        ctx.set_bound(True)      # the default
    return dist, slot, tid, redo


def test_the_two_bounds_where_they_differ(ctx, oracle):
    """rows at squared distances spread over [0, 0.12 R^2] from queries of norm R = 30: the gap between the k-th and the
    k'-th candidate (30 rows of ~5000 per list: ~0.65) clears twice the statistical bound (1.9e-5 (|q| + |x|)^2 = 0.07
    at 1536-d) AND twice the deterministic one of the four-chain scan kernel (gamma_(d/4+4) 2 |q||x| + ... = 0.09), so
    both settle every query -- where round 3's deterministic form, (gamma_(d+1) + gamma_(d+2)) (|q| + |x|)^2 = 0.66 on
    one accumulator chain, could settle none (checked numerically below) -- and both return the oracle's rows.  With the
    rows ten times denser (~90 rows inside the deterministic band: more than k' = 40 holds, fewer than 256) the queries
    are flagged and SETTLED BY THE WIDER CANDIDATE SET of batch_fix_kernel: nobody takes the exact pass."""
    dim, n, nq, lists, k = 1536, 10000, 128, 2, 10
    rng = np.random.default_rng(77)
    base = rng.standard_normal(dim).astype(np.float32)
    base *= 30.0 / np.linalg.norm(base)                                     # R = 30
    dirs = rng.standard_normal((n, dim)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    queries = np.ascontiguousarray(base[None, :] + 0.01 * rng.standard_normal((nq, dim)).astype(np.float32))
    u = 2.0 ** -24
    old_worst = ((dim + 1) * u + (dim + 2) * u) * (30.0 + 31.8) ** 2       # round 3's eps
    redo, widened = {}, {}
    for spread in (0.12, 0.012):
        radius = np.sqrt(rng.random(n) * spread * 900.0).astype(np.float32)[:, None]
        data = np.ascontiguousarray(base[None, :] + radius * dirs)
        ivf = CpuIvf(oracle, po.OPS_L2, po.ORA_F32, data, lists)
        ix = _upload(ctx, ivf)
        gaps = []
        for wc in (False, True):
            dist, slot, tid, redo[(spread, wc)] = _search_with_bound(ctx, ix, queries, lists, k, wc)
            widened[(spread, wc)] = _search_with_bound.widened
            for i in range(nq):
                wt, wd = oracle.search(ivf.struct, queries[i], lists, k)
                assert_topk_equiv(tid[i].tolist(), dist[i], wt.tolist(), wd, what="spread %s bound %s q %d" % (spread, wc, i))
                if spread == 0.12 and not wc:
                    d40 = np.sort(((data.astype(np.float64) - queries[i].astype(np.float64)) ** 2).sum(1))[:40]
                    gaps.append(d40[39] - d40[9])
        if spread == 0.12:
            assert redo[(spread, False)] == 0 and redo[(spread, True)] == 0, redo      # both settle every query
            assert 2.0 * old_worst > np.median(gaps), (old_worst, np.median(gaps))      # round 3's form could not
            assert widened[(spread, False)] == 0 and widened[(spread, True)] == 0, widened
        else:
            assert redo[(spread, False)] == 0 and redo[(spread, True)] == 0, redo       # no exact pass ...
            assert widened[(spread, True)] >= nq // 2, widened                          # ... the wider set settled them
            assert widened[(spread, False)] <= widened[(spread, True)], widened
        ix.close()


def test_mfma_l2_under_catastrophic_cancellation(ctx, oracle):
    """rows = b + tiny perturbations with |b| = 1000: |q|^2 + |x|^2 - 2 q.x loses every digit of distances ~ 1e-2
    (the expansion's rounding is ~ 1), so no bound can settle anything: every query must be flagged and redone on
    the exact kernels, whose results are the reference's.  Same through the assignment pre-filter and pgv_exact_topk."""
    dim, n, nq, lists, k = 256, 6000, 96, 3, 10
    rng = np.random.default_rng(78)
    base = rng.standard_normal(dim).astype(np.float32)
    base *= 1000.0 / np.linalg.norm(base)
    data = np.ascontiguousarray(base[None, :] + 0.01 * rng.standard_normal((n, dim)).astype(np.float32))
    queries = np.ascontiguousarray(base[None, :] + 0.01 * rng.standard_normal((nq, dim)).astype(np.float32))
    centers = np.ascontiguousarray(base[None, :] + 0.01 * rng.standard_normal((lists, dim)).astype(np.float32))
    ivf = CpuIvf(oracle, po.OPS_L2, po.ORA_F32, data, lists, centers=centers)
    ix = _upload(ctx, ivf)
    for wc in (False, True):
        dist, slot, tid, redo = _search_with_bound(ctx, ix, queries, lists, k, wc)
        assert redo == nq, (wc, redo)
        for i in range(nq):
            wt, wd = oracle.search(ivf.struct, queries[i], lists, k)
            assert_topk_equiv(tid[i].tolist(), dist[i], wt.tolist(), wd, what="cancel bound %s q %d" % (wc, i))
        ctx.set_bound(wc)
        try:
            d, idx = api.exact_topk(ctx, api.PGV_L2SQ, api.PGV_F32, dim, queries, data, k)
            # 600 centers so that the assignment goes through the MFMA pre-filter
            many = np.ascontiguousarray(base[None, :] + 0.01 * rng.standard_normal((600, dim)).astype(np.float32))
            got, _ = api.assign(ctx, api.PGV_L2SQ, api.PGV_F32, dim, many, data[:2048])
        finally:
            ctx.set_bound(True)
        for i in range(0, nq, 5):
            wt, wd = _ora_exact_topk(oracle, po.OPS_L2, po.ORA_F32, data, queries[i], k)
            assert_topk_equiv(idx[i].tolist(), d[i], wt.tolist(), wd, what="cancel exact_topk %s q %d" % (wc, i))
        want, _ = oracle.assign(po.OPS_L2, po.ORA_F32, many, data[:2048])
        for r in np.nonzero(np.asarray(got) != np.asarray(want))[0]:  # only float-level ties may differ
            dg = oracle.lib.ora_index_distance(po.OPS_L2, po.ORA_F32, dim, po._p(data[r]), po._p(many[got[r]]))
            dw = oracle.lib.ora_index_distance(po.OPS_L2, po.ORA_F32, dim, po._p(data[r]), po._p(many[want[r]]))
            assert abs(dg - dw) <= 1e-5 * dw, (wc, r, dg, dw)
    ix.close()


# ------------------------------------------------- a few queries at a time (mq_*_kernel)
@pytest.mark.parametrize("ops,dtype,dim,n,lists,nq,probes,k,dist", [
    (po.OPS_L2, po.ORA_F32, 1536, 6000, 60, 2, 10, 10, "clustered"),    # nq <= 4: rank and scan both per query
    (po.OPS_L2, po.ORA_F32, 96, 20000, 200, 7, 10, 10, "clustered"),    # share 0.35
    (po.OPS_L2, po.ORA_F32, 100, 20000, 400, 16, 10, 64, "uniform"),    # share 0.4; near ties
    (po.OPS_L2, po.ORA_F16, 3072, 3000, 100, 3, 8, 10, "clustered"),
    (po.OPS_IP, po.ORA_F32, 768, 8000, 160, 6, 8, 25, "clustered"),
    (po.OPS_COSINE, po.ORA_F16, 200, 8000, 160, 5, 10, 10, "clustered"),
    (po.OPS_L2, po.ORA_F32, 5, 9000, 300, 12, 10, 1000, "normal"),      # deep head, more than some queries have tuples
    (po.OPS_L2, po.ORA_F32, 64, 24000, 600, 22, 10, 10, "clustered"),   # per-query center ranking up to 24 queries
    (po.OPS_IP, po.ORA_F16, 128, 24000, 800, 30, 10, 10, "clustered"),  # past it: dense ranking, per-query scan
])
def test_small_batches_take_the_per_query_kernels(ctx, oracle, ops, dtype, dim, n, lists, nq, probes, k, dist):
    """nq * probes <= 0.4 lists (or nq <= 4): pgv_search_batch runs mq_rank / mq_lists / mq_scan / mq_head -- the
    head must be GetScanLists + GetScanItems + tuplesort's (src/ivfscan.c:47-187), like every other path's"""
    data = gen(n, dim, seed=931, dist=dist, clusters=lists, dtype=dtype)
    ivf = CpuIvf(oracle, ops, dtype, data, lists)
    queries = gen(nq, dim, seed=932, dist=dist, clusters=lists, dtype=dtype)
    gq = normalize_rows(oracle, queries, dtype) if ops == po.OPS_COSINE else queries
    ix = _upload(ctx, ivf)
    dist_, slot, tid = ix.search_batch(gq, probes, k, want_tid=True)
    ranked, rdist = ix.rank_lists(gq, probes)
    for i in range(nq):
        wt, wd = oracle.search(ivf.struct, queries[i], probes, k)
        have = slot[i] >= 0
        assert have.sum() == len(wt) and (tid[i][~have] == np.uint64(0xFFFFFFFFFFFFFFFF)).all()
        assert np.isinf(dist_[i][~have]).all()
        assert_topk_equiv(tid[i][have].tolist(), dist_[i][:len(wt)], wt.tolist(), wd, what="mq q %d" % i)
        wl, wld = oracle.get_scan_lists(ivf.struct, gq[i], probes)  # GetScanLists sees what GetScanValue normalised
        assert_topk_equiv(ranked[i].tolist(), rdist[i], wl.tolist(), wld, what="mq lists q %d" % i)
    ix.close()


def test_small_batch_ties_empty_lists_and_device_buffers(ctx, oracle):
    """integer data (exact ties keep stream order), an empty list among the probed, torch tensors in and out"""
    import torch
    dim, lists = 8, 40
    data = gen(3000, dim, seed=941, dist="int")
    data[100:140] = data[60]
    centers = gen(lists, dim, seed=942, dist="int")
    centers[5] = 1000.0                  # nothing lands here: an empty list
    ivf = CpuIvf(oracle, po.OPS_L2, po.ORA_F32, data, lists, centers=centers)
    queries = gen(3, dim, seed=943, dist="int")
    queries[1] = data[60]
    ix = _upload(ctx, ivf)
    for probes in (5, lists):            # lists = 40 probes x 3 queries: share 3 but nq <= 4 -> still per query
        d, s, t = ix.search_batch(queries, probes, 50, want_tid=True)
        for i in range(3):
            wt, wd = oracle.search(ivf.struct, queries[i], probes, 50)
            have = s[i] >= 0
            assert_topk_equiv(t[i][have].tolist(), d[i][:len(wt)], wt.tolist(), wd, what="mq ties q %d" % i)
            # equal distances come back in stream order (probe order, then position in the list)
            dd, ss = d[i][have], s[i][have]
            for a in range(len(dd) - 1):
                if dd[a] == dd[a + 1]:
                    la = np.searchsorted(ivf.list_offsets, ss[a], side="right") - 1
                    lb = np.searchsorted(ivf.list_offsets, ss[a + 1], side="right") - 1
                    assert la != lb or ss[a] < ss[a + 1]
    qd = torch.from_numpy(queries).cuda()
    dd, sd, td = ix.search_batch(qd, 5, 50, want_tid=True)
    hd, hs, ht = ix.search_batch(queries, 5, 50, want_tid=True)
    assert sd.cpu().numpy().tolist() == hs.tolist() and dd.cpu().numpy().tolist() == hd.tolist()
    ix.close()


# ------------------------------------------------- the build's tuplesort on the device
@pytest.mark.parametrize("ops,dtype,dim,n,lists", [
    (po.OPS_L2, po.ORA_F32, 1536, 5000, 40),
    (po.OPS_L2, po.ORA_F32, 30, 20000, 300),     # padded rows (30 floats -> 32), short-varlena territory on pages
    (po.OPS_IP, po.ORA_F16, 257, 9000, 64),      # padded fp16 rows
    (po.OPS_L2, po.ORA_F32, 8, 4000, 2000),      # more lists than rows per list: many empty lists
])
def test_builder_is_assign_plus_the_tuplesort_by_list(ctx, oracle, ops, dtype, dim, n, lists):
    """pgv_builder_add / finish / pgv_index_drain against AddTupleToSort + the sort on the list column
    (src/ivfbuild.c:161-219, :606-615): the oracle's assignment, a stable sort by list (heap order inside a list)"""
    data = gen(n, dim, seed=951, dist="clustered", clusters=min(lists, 64), dtype=dtype)
    centers = np.ascontiguousarray(data[np.random.default_rng(3).choice(n, lists, replace=False)])
    metric = api.PGV_L2SQ if ops == po.OPS_L2 else api.PGV_NEG_IP
    tids = (np.arange(n, dtype=np.uint64) * np.uint64(7)) + np.uint64(11)
    b = api.IvfBuilder(ctx, metric, DT[dtype], dim, centers, expected_rows=n // 3)   # grows twice
    for lo in range(0, n, 3000):
        b.add(data[lo:lo + 3000], tids[lo:lo + 3000])
    assert b.rows == n
    ix, off, got_lists = b.finish(want_lists=True)
    b.close()
    want_lists, _ = oracle.assign(ops, dtype, centers, data)
    want_lists = np.asarray(want_lists)
    diff = np.nonzero(got_lists != want_lists)[0]
    for r in diff:  # only float-level ties may differ
        dg = oracle.lib.ora_index_distance(ops, dtype, dim, po._p(data[r]), po._p(centers[got_lists[r]]))
        dw = oracle.lib.ora_index_distance(ops, dtype, dim, po._p(data[r]), po._p(centers[want_lists[r]]))
        assert abs(dg - dw) <= 1e-5 * max(abs(dw), 1e-30), (r, dg, dw)
    order = np.argsort(got_lists, kind="stable")
    assert off.tolist() == np.concatenate([[0], np.cumsum(np.bincount(got_lists, minlength=lists))]).tolist()
    vec, dt, seen = api.drain_index(ix, chunk_rows=1777)
    assert [s[0] for s in seen] == list(range(0, n, 1777)) and sum(s[1] for s in seen) == n
    assert dt.tolist() == tids[order].tolist()
    assert np.array_equal(vec.view(np.uint8), np.ascontiguousarray(data[order]).view(np.uint8))
    # and the mirror answers like one uploaded from the same arrays
    ref = api.IvfIndex(ctx, metric, DT[dtype], dim, centers, off, data[order], tids[order])
    q = gen(20, dim, seed=952, dist="clustered", clusters=min(lists, 64), dtype=dtype)
    p = min(lists, 6)
    d1, s1, t1 = ix.search_batch(q, p, 10, want_tid=True)
    d2, s2, t2 = ref.search_batch(q, p, 10, want_tid=True)
    assert s1.tolist() == s2.tolist() and t1.tolist() == t2.tolist() and d1.tolist() == d2.tolist()
    ref.close()
    ix.close()


def test_builder_without_tids_and_empty(ctx):
    centers = gen(5, 16, seed=1, dist="normal")
    b = api.IvfBuilder(ctx, api.PGV_L2SQ, api.PGV_F32, 16, centers)
    ix, off, _ = b.finish()
    assert ix.rows == 0 and off.tolist() == [0] * 6
    ix.close()
    b.close()
    data = gen(700, 16, seed=2, dist="normal")
    b = api.IvfBuilder(ctx, api.PGV_L2SQ, api.PGV_F32, 16, centers)
    b.add(data)                      # no TIDs: heap positions
    ix, off, lists = b.finish(want_lists=True)
    vec, tids, _ = api.drain_index(ix)
    assert tids.tolist() == np.argsort(lists, kind="stable").tolist()
    assert np.array_equal(vec, data[tids.astype(np.int64)])
    ix.close()
    b.close()


@pytest.mark.parametrize("dtype,dim", [(po.ORA_F32, 96), (po.ORA_F16, 250)])
def test_builder_without_centers_uploads_beside_kmeans(ctx, dtype, dim):
    """pgv_builder_begin(centers = NULL): rows are only copied (on the builder's own stream) while pgv_kmeans computes
    the centers on the context from another host thread; set_centers + finish then give exactly what a builder that
    knew the centers from the start gives"""
    import threading
    n, lists = 40000, 64
    data = gen(n, dim, seed=971, dist="clustered", clusters=32, dtype=dtype)
    tids = np.arange(n, dtype=np.uint64) + np.uint64(5)
    samples = np.ascontiguousarray(data[::8])
    box = {}

    def run_kmeans():
        try:
            box["centers"] = api.kmeans(ctx, api.PGV_OPS_L2, DT[dtype], dim, samples, lists, api.make_rng(seed=9))[0]
        except Exception as e:  # noqa: BLE001
            box["error"] = e

    b = api.IvfBuilder(ctx, api.PGV_L2SQ, DT[dtype], dim, None, expected_rows=n // 2, nlists=lists)
    th = threading.Thread(target=run_kmeans)
    th.start()
    for lo in range(0, n, 2500):
        b.add(data[lo:lo + 2500], tids[lo:lo + 2500])
    with pytest.raises(api.PgvError):
        b.finish()                     # no centers yet
    th.join()
    assert "error" not in box, box.get("error")
    centers = box["centers"]
    b.set_centers(centers)
    with pytest.raises(api.PgvError):
        b.set_centers(centers)         # once
    b.add(data[:100], tids[:100] + np.uint64(1000000))   # and rows that come after the centers
    ix, off, lists_got = b.finish(want_lists=True)
    b.close()
    ref_b = api.IvfBuilder(ctx, api.PGV_L2SQ, DT[dtype], dim, centers, expected_rows=n + 100)
    ref_b.add(data, tids)
    ref_b.add(data[:100], tids[:100] + np.uint64(1000000))
    ref, ref_off, ref_lists = ref_b.finish(want_lists=True)
    ref_b.close()
    assert off.tolist() == ref_off.tolist() and lists_got.tolist() == ref_lists.tolist()
    v1, t1, _ = api.drain_index(ix)
    v2, t2, _ = api.drain_index(ref)
    assert t1.tolist() == t2.tolist() and np.array_equal(v1.view(np.uint8), v2.view(np.uint8))
    ix.close()
    ref.close()


@pytest.mark.parametrize("ops,dtype,dim,n,lists", [
    (po.OPS_L2, po.ORA_F32, 1536, 4000, 16),      # one tuple per page
    (po.OPS_L2, po.ORA_F32, 20, 30000, 50),       # short varlena headers, ~80 tuples per page
    (po.OPS_COSINE, po.ORA_F16, 384, 12000, 30),  # normalised rows, zero-norm rows dropped
    (po.OPS_IP, po.ORA_F32, 100, 9000, 200),      # more than one list page; short lists
])
def test_build_mirror_pages_and_mirror_agree(ctx, oracle, ops, dtype, dim, n, lists):
    """pgv_host_ivf_build_mirror: the pages (InsertTuples, src/ivfbuild.c:271-331) are written from the device
    mirror's rows as they drain; staging those pages gives the mirror's arrays back, and the oracle walking the pages
    answers like the mirror does"""
    from pgvector_amd import _host
    heap = gen(n, dim, seed=961, dist="clustered", clusters=min(lists, 32), dtype=dtype)
    if ops == po.OPS_COSINE:
        heap[17] = 0
        heap[4000] = 0
    tids = ((np.arange(n, dtype=np.uint64) // 50) << np.uint64(16)) | (np.arange(n, dtype=np.uint64) % 50 + 1)
    pops = {po.OPS_L2: api.PGV_OPS_L2, po.OPS_IP: api.PGV_OPS_IP, po.OPS_COSINE: api.PGV_OPS_COSINE}[ops]
    samples = heap[np.random.default_rng(1).choice(n, min(n, 50 * lists), replace=False)]
    rel = _host.Relation()
    mirror = rel.build_mirror(ctx, pops, DT[dtype], lists, heap, tids, samples, api.make_rng(seed=3))
    img = rel.stage(DT[dtype])
    kept = n - (2 if ops == po.OPS_COSINE else 0)
    assert img.nrows == kept == mirror.rows
    vec, mt, _ = api.drain_index(mirror, chunk_rows=2048)
    assert mt.tolist() == img.tids.tolist()
    assert np.array_equal(np.ascontiguousarray(vec).view(np.uint8), np.ascontiguousarray(img.vectors).view(np.uint8))
    # heap order inside every list (the tuplesort's input order, which InsertTuples keeps)
    pos = {int(t): i for i, t in enumerate(tids.tolist())}
    for l in range(lists):
        run = [pos[int(t)] for t in img.tids[img.list_offsets[l]:img.list_offsets[l + 1]].tolist()]
        assert run == sorted(run)
    queries = gen(10, dim, seed=962, dist="clustered", clusters=min(lists, 32), dtype=dtype)
    gq = normalize_rows(oracle, queries, dtype) if ops == po.OPS_COSINE else queries
    d, s, t = mirror.search_batch(gq, 4, 10, want_tid=True)
    for i in range(len(queries)):
        wt, wd, _ = oracle.pages_search(rel.rel.pages, int(rel.nblocks), ops, dtype, gq[i], 4, 10)  # what GetScanValue hands on
        assert_topk_equiv(t[i][:len(wt)].tolist(), d[i][:len(wt)], wt.tolist(), wd, what="pages vs mirror q %d" % i)
    mirror.close()


def test_hnsw_device_search_is_the_oracles_walk_at_100k(ctx, oracle):
    """configs[3]'s row shape at 100 000 rows: the graph is built on the GPU (pgv_host_hnsw_build), the oracle's
    HnswSearchLayer / GetScanItems restatement (src/hnswutils.c:838-951, src/hnswscan.c:25-56) walks the SAME graph
    (ora_hnsw_import) -- hnswgettuple's first batch must come out identical up to float ties"""
    import torch
    from pgvector_amd import _host
    n, dim, m, efc = 100_000, 1536, 16, 64
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    comps = torch.rand((64, dim), generator=g, device=dev)
    data = comps[torch.randint(0, 64, (n,), generator=g, device=dev)] + 0.1 * torch.randn((n, dim), generator=g, device=dev)
    data = (data / data.norm(dim=1, keepdim=True)).contiguous()
    q = comps[torch.randint(0, 64, (24,), generator=g, device=dev)] + 0.1 * torch.randn((24, dim), generator=g, device=dev)
    q = (q / q.norm(dim=1, keepdim=True)).contiguous()
    host_rows = data.cpu().numpy()
    mirror = api.Hnsw(ctx, api.PGV_NEG_IP, api.PGV_F32, dim, data)
    built = _host.hnsw_build(mirror, host_rows, m, efc, api.make_rng(seed=1), max_batch=1024)
    assert built["nelements"] == n
    walk = po.HnswGraph.from_tuples(oracle, po.OPS_COSINE, po.ORA_F32, host_rows, m, built["levels"], built["nbr_start"],
                                    built["nbr"], built["entry"])
    qh = q.cpu().numpy()
    for ef in (40, 100):
        elem, dist, _ = mirror.search(q, ef, 10)
        eh, dh = elem.cpu().numpy(), dist.cpu().numpy()
        for i in range(len(qh)):
            wr, wd, _ = walk.search(qh[i], ef, 10)
            assert_topk_equiv(eh[i][eh[i] >= 0].tolist(), dh[i][:len(wr)], wr.tolist(), wd, what="hnsw 100k ef %d q %d" % (ef, i))
    walk.close()
    mirror.close()
