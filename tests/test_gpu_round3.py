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
