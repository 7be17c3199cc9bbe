"""GPU parity, round 3: the device mirror and the pooler across PROCESSES (a Postgres backend is a process,
src/ivfscan.c:252-296), the deterministic completeness bound of the MFMA L2 scan, the exact-scan top-k.
Same contract as test_gpu_parity.py: integers/indexes exact, distances within 1e-5 relative, ties as sets."""
import os

import numpy as np
import pytest

from oracle import pyoracle as po
from pgvector_amd import _host, api

from helpers import CpuIvf, assert_topk_equiv, gen

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
    mirror = api.Hnsw(ctx, api.PGV_L2SQ, gdt, dim, data[ex["rows"]])
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
    np.savez(job, handle=np.frombuffer(handle, dtype=np.uint8), queries=queries, dtype=gdt, ef=40, k=10)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "mp_hnsw_import_worker.py"), job, res], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = np.load(res)
    assert int(out["readonly"]) == api.PGV_ERR_STATE  # an imported mirror is read-only
    own_elem, own_dist, _ = mirror.search(queries, 40, 10)
    assert out["elem"].tolist() == np.asarray(own_elem).tolist()
    for i, q in enumerate(queries):
        rows, wd, _ = g.search(q, 40, 10)
        el = out["elem"][i]
        assert_topk_equiv(ex["rows"][el[el >= 0]].tolist(), out["dist"][i][:len(rows)], rows.tolist(), wd,
                          what="imported hnsw q%d" % i)
    mirror.close()
