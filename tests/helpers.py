"""Shared test helpers: synthetic data, a CPU-built IVFFlat image, comparisons."""
import json
import os

import numpy as np

from oracle import pyoracle as po

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# tolerance north_star states for floating point: 1e-5 relative on distances
RTOL = 1e-5

NP_OF = po.NP_OF


def golden(name):
    return json.load(open(os.path.join(GOLDEN, name)))


def gen(n, dim, seed, dist="uniform", dtype=po.ORA_F32, clusters=16):
    rng = np.random.default_rng(seed)
    if dist == "uniform":
        x = rng.random((n, dim), dtype=np.float32)
    elif dist == "normal":
        x = rng.standard_normal((n, dim)).astype(np.float32)
    elif dist == "clustered":
        c = rng.random((clusters, dim), dtype=np.float32)
        x = c[rng.integers(0, clusters, n)] + 0.1 * rng.standard_normal((n, dim)).astype(np.float32)
    elif dist == "int":
        x = rng.integers(-8, 9, (n, dim)).astype(np.float32)
    elif dist == "int10":
        # exact fp32 arithmetic (dim <= 8: every partial sum of squares < 2^24) with few equal distances
        x = rng.integers(0, 1024, (n, dim)).astype(np.float32)
    else:
        raise ValueError(dist)
    return np.ascontiguousarray(x.astype(NP_OF[dtype]))


def normalize_rows(ora, x, dtype):
    out = np.zeros_like(x)
    for i in range(x.shape[0]):
        if dtype == po.ORA_F32:
            ora.lib.ora_l2_normalize(x.shape[1], po._p(x[i]), po._p(out[i]))
        else:
            ora.lib.ora_halfvec_l2_normalize(x.shape[1], po._p(x[i]), po._p(out[i]))
    return out


class CpuIvf:
    """IVFFlat image built the way the reference builds it, on the CPU oracle:
    centers (given or k-means), every row assigned by AddTupleToSort's argmin,
    rows laid out list-major in heap order inside a list (src/ivfbuild.c:271-331)."""

    def __init__(self, ora, ops, dtype, data, nlists, centers=None, seed=3):
        self.ops, self.dtype = ops, dtype
        data = np.ascontiguousarray(data)
        heap_ids = np.arange(data.shape[0], dtype=np.int64)
        if ops == po.OPS_COSINE:
            # zero-norm rows are not indexed, the rest are stored normalised (src/ivfbuild.c:174-180)
            norms = np.array([ora.lib.ora_vector_norm(data.shape[1], po._p(r)) if dtype == po.ORA_F32
                              else ora.lib.ora_halfvec_l2_norm(data.shape[1], po._p(r)) for r in data])
            keep = norms > 0
            data, heap_ids = normalize_rows(ora, np.ascontiguousarray(data[keep]), dtype), heap_ids[keep]
        if centers is None:
            ns = ora.lib.ora_ivf_num_samples(nlists, data.shape[0])
            rng = np.random.default_rng(seed)
            samples = data[rng.choice(data.shape[0], min(ns, data.shape[0]), replace=False)]
            if ops in (po.OPS_IP, po.OPS_COSINE):
                samples = normalize_rows(ora, np.ascontiguousarray(samples), dtype)
                samples = samples[np.abs(samples.astype(np.float32)).sum(axis=1) > 0]
            centers, _, it = ora.kmeans(ops, dtype, samples, nlists, ora.prng(seed))
            assert it >= 0
        self.centers = np.ascontiguousarray(centers)
        lists, _ = ora.assign(ops, dtype, self.centers, data)
        order = np.argsort(lists, kind="stable")
        self.vectors = np.ascontiguousarray(data[order])
        self.tids = (heap_ids[order].astype(np.uint64) << np.uint64(16)) | np.uint64(1)
        self.heap_ids = heap_ids[order]
        counts = np.bincount(lists, minlength=nlists)
        self.list_offsets = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        self.nlists = nlists
        self.struct = ora.index_struct(ops, dtype, self.centers, self.list_offsets, self.vectors, self.tids)

    @property
    def metric(self):
        return 0 if self.ops == po.OPS_L2 else 1  # PGV_L2SQ / PGV_NEG_IP


def assert_close(got, want, rtol=RTOL, atol=0.0, what=""):
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    both_nan = np.isnan(got) & np.isnan(want)
    both_inf = np.isinf(got) & np.isinf(want) & (np.sign(got) == np.sign(want))
    ok = both_nan | both_inf | (np.abs(got - want) <= atol + rtol * np.abs(want))
    if not ok.all():
        bad = np.argwhere(~ok)[:5]
        raise AssertionError("%s: %d/%d mismatches, first %s got %s want %s" % (
            what, (~ok).sum(), ok.size, bad.tolist(), got[~ok][:5], want[~ok][:5]))


def assert_topk_equiv(got_ids, got_dist, want_ids, want_dist, rtol=RTOL, what="", atol=1e-30):
    """Row ids must match wherever the reference's own order is determined beyond
    the float tolerance; inside a run of (near-)equal distances any order is
    accepted (the reference's tuplesort leaves tie order unspecified,
    test/t/003_ivfflat_vector_build_recall.pl:85-90 is tie-tolerant too)."""
    got_ids, want_ids = list(got_ids), list(want_ids)
    assert len(got_ids) == len(want_ids), (what, len(got_ids), len(want_ids))
    # atol: inner products that cancel to almost nothing carry the rounding of their terms (callers pass
    # rtol x the size of the terms), not of the result
    assert_close(got_dist, want_dist, rtol=rtol, atol=atol, what=what + " distances")
    want_dist = np.asarray(want_dist, dtype=np.float64)
    i, n = 0, len(want_ids)
    while i < n:
        j = i + 1
        while j < n and abs(want_dist[j] - want_dist[j - 1]) <= 4 * rtol * max(abs(want_dist[j]), 1e-30) + 4 * atol:
            j += 1
        if j == n:
            # the last run may be cut by k: its members need only come from the tie class
            assert set(got_ids[:i]) == set(want_ids[:i]), what
            break
        assert sorted(got_ids[i:j]) == sorted(want_ids[i:j]), (what, i, j, got_ids[i:j], want_ids[i:j])
        i = j
