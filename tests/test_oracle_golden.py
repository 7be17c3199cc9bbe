"""CPU tests pinning the oracle: reference known-answer vectors, the reference's
own fp16 kernels (oracle/_ref), and the index-order transcripts."""
import math
import os

import ctypes as C

import numpy as np
import pytest

from oracle import pyoracle as po

from helpers import CpuIvf, gen, golden, normalize_rows

SQL_VECTOR = {"l2_distance": "ora_l2_distance", "inner_product": "ora_inner_product",
              "negative_inner_product": "ora_negative_inner_product",
              "cosine_distance": "ora_cosine_distance", "l1_distance": "ora_l1_distance"}
SQL_HALF = {"l2_distance": "ora_halfvec_l2_distance", "inner_product": "ora_halfvec_inner_product_f8",
            "negative_inner_product": "ora_halfvec_negative_inner_product",
            "cosine_distance": "ora_halfvec_cosine_distance", "l1_distance": "ora_halfvec_l1_distance"}


def _expect(v):
    return {"inf": math.inf, "-inf": -math.inf, "nan": math.nan}.get(v, v) if isinstance(v, str) else v


def _apply_wrap(value, wrap):
    if not wrap:
        return value
    if wrap[0] == "round":
        return round(value, wrap[1])
    if wrap[0] == "real":
        return float(np.float32(value))
    raise AssertionError(wrap)


CASES = golden("distance_known_answers.json")["cases"]


@pytest.mark.parametrize("case", CASES, ids=[c["source"].split("/")[-1] for c in CASES])
def test_known_answers(oracle, case):
    half = case["type"] == "halfvec"
    dt = po.ORA_F16 if half else po.ORA_F32
    func, args = case["func"], case["args"]
    if func in ("vector_norm", "l2_norm"):
        a = oracle.arr(args[0], dt)
        got = (oracle.lib.ora_halfvec_l2_norm if half else oracle.lib.ora_vector_norm)(len(a), po._p(a))
        assert _apply_wrap(got, case.get("wrap")) == pytest.approx(_expect(case["expect"]), rel=1e-6)
        return
    if func == "l2_normalize":
        a = oracle.arr(args[0], dt)
        out = np.zeros_like(a)
        rc = (oracle.lib.ora_halfvec_l2_normalize if half else oracle.lib.ora_l2_normalize)(len(a), po._p(a), po._p(out))
        if "error" in case:
            assert rc != 0
        else:
            assert rc == 0
            # psql prints the shortest decimal that round-trips in the element type
            want = np.asarray(case["expect"], dtype=po.NP_OF[dt])
            np.testing.assert_array_equal(out, want)
        return
    rc, got = oracle.sql((SQL_HALF if half else SQL_VECTOR)[func], args[0], args[1], half=half)
    if "error" in case:
        assert rc == 1 and oracle.last_error() == case["error"]
        return
    assert rc == 0
    want = _expect(case["expect"])
    if isinstance(want, float) and math.isnan(want):
        assert math.isnan(got)
    else:
        assert got == want, (case, got)


def test_cross_type_equality(oracle):
    """test/t/034_distance_functions.pl:36-52: on integer-valued 5-d vectors halfvec
    results equal vector results exactly"""
    rng = np.random.default_rng(34)
    data = rng.integers(0, 10, (500, 5)).astype(np.float32)
    queries = rng.integers(0, 10, (20, 5)).astype(np.float32)
    pairs = [("ora_l2_distance", "ora_halfvec_l2_distance"), ("ora_inner_product", "ora_halfvec_inner_product_f8"),
             ("ora_cosine_distance", "ora_halfvec_cosine_distance"), ("ora_l1_distance", "ora_halfvec_l1_distance")]
    for q in queries:
        for r in data[:100]:
            for fv, fh in pairs:
                _, a = oracle.sql(fv, r, q)
                _, b = oracle.sql(fh, r, q, half=True)
                assert a == b or (math.isnan(a) and math.isnan(b)), (fv, r, q, a, b)


@pytest.mark.skipif(not os.path.exists(os.path.join(po.HERE, "_ref", "libpgvref.so")),
                    reason="oracle/_ref not built (reference tree absent)")
def test_fp16_kernels_match_reference_object(oracle):
    """the restated fp16 kernels against the reference's own halfutils.c, bit for bit"""
    ref = po.Ref()
    rng = np.random.default_rng(16)
    for dim in (1, 2, 3, 7, 8, 9, 15, 16, 17, 64, 100, 127, 128, 1000, 3072):
        for scale in (1.0, 100.0, 1e-3):
            a = (rng.standard_normal(dim) * scale).astype(np.float16)
            b = (rng.standard_normal(dim) * scale).astype(np.float16)
            for mine, theirs in [("ora_halfvec_l2_squared", "pgvref_halfvec_l2_squared"),
                                 ("ora_halfvec_inner_product", "pgvref_halfvec_inner_product"),
                                 ("ora_halfvec_cosine_similarity", "pgvref_halfvec_cosine_similarity"),
                                 ("ora_halfvec_l1", "pgvref_halfvec_l1")]:
                x, y = oracle.kernel(mine, a, b, half=True), ref.kernel(theirs, a, b)
                assert x == y or (math.isnan(x) and math.isnan(y)), (mine, dim, scale, x, y)
    # every binary16 value converts like the reference's HalfToFloat4, and back
    for h in range(0, 65536, 7):
        x, y = oracle.lib.ora_half_to_float(h), ref.lib.pgvref_half_to_float(h)
        assert x == y or (math.isnan(x) and math.isnan(y))
    for f in np.concatenate([rng.standard_normal(2000) * 10.0 ** rng.integers(-9, 6, 2000),
                             [0.0, -0.0, 65504.0, 65519.9, 65520.0, 1e-8, 5.96e-8, 2.98e-8, np.inf, -np.inf]]):
        f = float(np.float32(f))
        assert oracle.lib.ora_float_to_half(f) == ref.lib.pgvref_float_to_half(f), f


def test_fp32_kernels_against_float64(oracle):
    """the fp32 kernels stay within fp32 round-off of an exact float64 evaluation"""
    rng = np.random.default_rng(32)
    for dim in (1, 3, 9, 128, 768, 1536, 2000):
        a, b = rng.random(dim, dtype=np.float32), rng.random(dim, dtype=np.float32)
        a64, b64 = a.astype(np.float64), b.astype(np.float64)
        assert oracle.kernel("ora_vector_l2_squared", a, b) == pytest.approx(((a64 - b64) ** 2).sum(), rel=1e-5)
        assert oracle.kernel("ora_vector_inner_product", a, b) == pytest.approx((a64 * b64).sum(), rel=1e-5)
        assert oracle.kernel("ora_vector_l1", a, b) == pytest.approx(np.abs(a64 - b64).sum(), rel=1e-5)


def _rows_as(dtype, rows):
    return np.asarray(rows, dtype=po.NP_OF[dtype])


INDEX_CASES = [c for c in golden("index_order.json")["cases"] if c["am"] == "ivfflat"]


@pytest.mark.parametrize("case", INDEX_CASES, ids=[c["source"].split("/")[-1] for c in INDEX_CASES])
def test_ivfflat_transcripts(oracle, case):
    """test/expected/ivfflat_vector.out / ivfflat_halfvec.out replayed on the oracle's
    restatement of the build (k-means + assignment) and scan loops"""
    dtype = po.ORA_F16 if case["type"] == "halfvec" else po.ORA_F32
    ops = {"l2": po.OPS_L2, "ip": po.OPS_IP, "cosine": po.OPS_COSINE}[case["ops"]]
    rows = _rows_as(dtype, case["rows"])
    # the index is created over the first three rows + NULL, the 4th is inserted afterwards
    # (src/ivfinsert.c: nearest list, appended) -- with lists=1 the layout is the same
    built_on = rows[:3]
    if ops == po.OPS_COSINE:
        built_on = built_on[np.abs(built_on.astype(np.float32)).sum(axis=1) > 0]
    samples = built_on
    if ops in (po.OPS_IP, po.OPS_COSINE):
        from helpers import normalize_rows
        samples = normalize_rows(oracle, np.ascontiguousarray(samples), dtype)
        samples = samples[np.abs(samples.astype(np.float32)).sum(axis=1) > 0]
    centers, _, it = oracle.kmeans(ops, dtype, samples, case["lists"], oracle.prng(42))
    assert it >= 0
    ivf = CpuIvf(oracle, ops, dtype, rows, case["lists"], centers=centers)
    stored = ivf.vectors
    orig = rows[ivf.heap_ids]

    def run(query, probes, k=100):
        tids, dist = oracle.search(ivf.struct, None if query is None else _rows_as(dtype, query), probes, k)
        slots = [int(np.where(ivf.tids == t)[0][0]) for t in tids]
        return [orig[s].astype(np.float32).tolist() for s in slots]

    if "self_nearest" in case:
        for v in case["self_nearest"]:
            assert run(v, case["probes"], k=1) == [[float(x) for x in v]]
        return
    if case.get("iterative"):
        # relaxed_order: batches of `probes` lists until max_probes lists were scanned
        # (src/ivfscan.c:400-406); every batch is sorted on its own
        lists, _ = oracle.get_scan_lists(ivf.struct, _rows_as(dtype, case["query"]), case["max_probes"])
        got = []
        for i in range(0, len(lists), case["probes"]):
            d, s = oracle.get_scan_items(ivf.struct, _rows_as(dtype, case["query"]), lists[i:i + case["probes"]])
            got += [orig[x].astype(np.float32).tolist() for x in s]
        assert got == [[float(x) for x in v] for v in case["expect"]]
        return
    got = run(case["query"], case["probes"])
    if "expect_count" in case:
        assert len(got) == case["expect_count"]
    else:
        assert got == [[float(x) for x in v] for v in case["expect"]]
    assert stored.shape[0] == len(ivf.heap_ids)


def test_scan_lists_rules(oracle):
    """GetScanLists (src/ivfscan.c:47-118): ascending, strict-< replacement, clamp to lists"""
    centers = np.array([[0, 0], [1, 0], [1, 0], [5, 5], [0.5, 0]], dtype=np.float32)
    off = np.arange(6, dtype=np.int64)
    ix = oracle.index_struct(po.OPS_L2, po.ORA_F32, centers, off, centers)
    lists, dist = oracle.get_scan_lists(ix, np.array([1, 0], dtype=np.float32), 2)
    assert lists.tolist() == [1, 2] and dist.tolist() == [0.0, 0.0]
    lists, dist = oracle.get_scan_lists(ix, np.array([0, 0], dtype=np.float32), 3)
    assert lists.tolist() == [0, 4, 1]  # the tie between lists 1 and 2 keeps the earlier one
    lists, _ = oracle.get_scan_lists(ix, np.array([0, 0], dtype=np.float32), 99)
    assert len(lists) == 5
    lists, dist = oracle.get_scan_lists(ix, None, 3)  # NULL query: ZeroDistance
    assert dist.tolist() == [0.0, 0.0, 0.0]


def test_assign_first_minimum_wins(oracle):
    centers = np.array([[1, 1], [1, 1], [0, 0]], dtype=np.float32)
    rows = np.array([[1, 1], [0, 0], [0.5, 0.5]], dtype=np.float32)
    lists, dist = oracle.assign(po.OPS_L2, po.ORA_F32, centers, rows)
    assert lists.tolist() == [0, 2, 0] and dist.tolist() == [0.0, 0.0, 0.5]


def test_kmeans_recall_floor(oracle):
    """test/t/003_ivfflat_vector_build_recall.pl, scaled down: 3-d uniform data, k=20,
    probes = lists must give recall 1.0 (exhaustive) and probes=10% a decent floor"""
    data = gen(4000, 3, seed=1)
    ivf = CpuIvf(oracle, po.OPS_L2, po.ORA_F32, data, 20)
    queries = gen(10, 3, seed=2)
    for q in queries:
        exact = np.argsort(((data.astype(np.float64) - q) ** 2).sum(axis=1), kind="stable")[:20]
        tids, _ = oracle.search(ivf.struct, q, 20, 20)
        got = set(int(t >> np.uint64(16)) for t in tids)
        assert got == set(exact.tolist())


def test_kmeans_handles_duplicates_and_little_data(oracle):
    """test/t/008_ivfflat_centers.pl: more lists than distinct points must not fail"""
    data = np.tile(np.array([[1, 2, 3]], dtype=np.float32), (30, 1))
    centers, closest, it = oracle.kmeans(po.OPS_L2, po.ORA_F32, data, 5, oracle.prng(1))
    assert it >= 1 and np.isfinite(centers).all()
    centers, closest, it = oracle.kmeans(po.OPS_L2, po.ORA_F32, np.zeros((0, 3), np.float32), 4, oracle.prng(1))
    assert it == 0 and centers.shape == (4, 3) and (centers >= 0).all() and (centers < 1).all()


def test_prng_is_deterministic(oracle):
    a, b = oracle.prng(42), oracle.prng(42)
    xs = [oracle.lib.ora_prng_double(a) for _ in range(5)]
    assert xs == [oracle.lib.ora_prng_double(b) for _ in range(5)]
    assert all(0.0 <= x < 1.0 for x in xs) and len(set(xs)) == 5


HNSW_CASES = [c for c in golden("index_order.json")["cases"] if c["am"] == "hnsw"]


@pytest.mark.parametrize("case", HNSW_CASES, ids=[c["source"].split("/")[-1] for c in HNSW_CASES])
def test_hnsw_transcripts(oracle, case):
    """test/expected/hnsw_vector.out replayed on the restated in-memory build + search"""
    ops = {"l2": po.OPS_L2, "ip": po.OPS_IP, "cosine": po.OPS_COSINE, "l1": po.OPS_L1}[case["ops"]]
    rows = np.asarray(case["rows"], dtype=np.float32)
    g = po.HnswGraph(oracle, ops, po.ORA_F32, rows, m=16, ef_construction=64, seed=1)
    got, _, _ = g.search(np.asarray(case["query"], dtype=np.float32), 40, 100)
    assert [rows[r].tolist() for r in got] == [[float(x) for x in v] for v in case["expect"]]


def test_hnsw_recall_floor(oracle):
    """test/t/012_hnsw_vector_build_recall.pl:94-95, scaled: 3-d uniform data, k = 20,
    defaults m=16 ef_construction=64 ef_search=40 -> recall >= 0.99 (L2)"""
    data = gen(3000, 3, seed=12)
    g = po.HnswGraph(oracle, po.OPS_L2, po.ORA_F32, data, seed=12)
    hits = total = 0
    for q in gen(20, 3, seed=13):
        d = ((data.astype(np.float64) - q) ** 2).sum(axis=1)
        kth = np.sort(d)[19]
        rows, dist, scored = g.search(q, 40, 20)
        hits += int((d[rows] <= kth).sum())
        total += 20
        assert scored > 20 and (np.diff(dist) >= 0).all()
    assert hits / total >= 0.99


def test_hnsw_duplicates_fold_into_one_element(oracle):
    """src/hnswbuild.c:339-364: bytewise-equal vectors share an element (up to 10 heap tids)"""
    data = np.tile(np.array([[1, 2, 3]], dtype=np.float32), (25, 1))
    data = np.vstack([data, gen(50, 3, seed=3)])
    g = po.HnswGraph(oracle, po.OPS_L2, po.ORA_F32, data, seed=2)
    assert g.nelements == 50 + 3  # 25 duplicates -> ceil(25 / 10) elements
    rows, dist, _ = g.search(np.array([1, 2, 3], dtype=np.float32), 40, 25)
    assert sorted(rows.tolist()) == list(range(25)) and (dist == 0).all()


def test_hnsw_import_walks_like_the_graph_it_was_exported_from(oracle):
    """ora_hnsw_import (a graph given as the index's neighbor tuples: what the GPU build and the page stager
    produce) must answer like the in-memory graph: same rows, same distances, same number of scored elements"""
    for ops in (po.OPS_L2, po.OPS_COSINE):
        data = gen(1500, 12, seed=31, dist="normal")
        g = po.HnswGraph(oracle, ops, po.ORA_F32, data, m=8, ef_construction=40, seed=4)
        ex = g.export_tuples()
        values = data[ex["rows"]]
        if ops == po.OPS_COSINE:
            values = normalize_rows(oracle, np.ascontiguousarray(values), po.ORA_F32)
        h = po.HnswGraph.from_tuples(oracle, ops, po.ORA_F32, values, 8, ex["levels"], ex["nbr_start"], ex["nbr"], ex["entry"])
        for q in gen(25, 12, seed=32, dist="normal"):
            rows, dist, scored = g.search(q, 40, 10)
            erows, edist, escored = h.search(q, 40, 10)
            assert ex["rows"][erows].tolist() == rows.tolist() and escored == scored
            np.testing.assert_array_equal(edist, dist)


def test_hnsw_parallel_build_with_one_thread_is_the_serial_graph(oracle):
    """ora_hnsw_build_parallel restates the reference's parallel in-memory build (src/hnswbuild.c:366-480: element
    locks, entry lock + gate); with a single inserter nothing is concurrent and the graph must be the serial
    build's, element for element -- duplicates and zero-norm rows included"""
    for ops, dim in ((po.OPS_L2, 12), (po.OPS_COSINE, 7)):
        data = gen(1200, dim, seed=41, dist="normal")
        data[100:112] = data[100]          # 12 bytewise duplicates: two elements (10 heap tids + 2)
        if ops == po.OPS_COSINE:
            data[5] = 0                    # not indexed
        a = po.HnswGraph(oracle, ops, po.ORA_F32, data, m=8, ef_construction=40, seed=6)
        b = po.HnswGraph(oracle, ops, po.ORA_F32, data, m=8, ef_construction=40, seed=6, threads=1)
        ea, eb = a.export(), b.export()
        assert a.nelements == b.nelements and ea["entry"] == eb["entry"]
        np.testing.assert_array_equal(ea["levels"], eb["levels"])
        np.testing.assert_array_equal(ea["rows"], eb["rows"])
        np.testing.assert_array_equal(ea["neighbors"], eb["neighbors"])
        q = gen(1, dim, seed=42, dist="normal")[0]
        assert a.search(q, 40, 10)[0].tolist() == b.search(q, 40, 10)[0].tolist()


def test_hnsw_parallel_build_with_threads_keeps_the_invariants_and_the_recall(oracle):
    """several inserters: which neighbors an element finds depends on the interleaving (as in the reference), so the
    graph is checked by what must hold for every interleaving -- levels from the one seeded stream, list lengths
    within m / 2m, valid ids, no self links, every neighbor on a layer it has, an entry point of the top level,
    each element reachable by search -- and by its recall against the serial graph's"""
    data = gen(4000, 16, seed=51, dist="normal")
    serial = po.HnswGraph(oracle, po.OPS_L2, po.ORA_F32, data, m=8, ef_construction=40, seed=8)
    par = po.HnswGraph(oracle, po.OPS_L2, po.ORA_F32, data, m=8, ef_construction=40, seed=8, threads=6)
    es, ep = serial.export(), par.export()
    assert par.nelements == serial.nelements == 4000
    np.testing.assert_array_equal(ep["levels"], es["levels"])
    assert ep["entry_level"] == int(ep["levels"].max())
    nb = ep["neighbors"]
    for e in range(0, 4000, 7):
        for lc in range(int(ep["levels"][e]) + 1):
            ids = nb[e, lc][nb[e, lc] >= 0]
            assert len(ids) <= (16 if lc == 0 else 8) and e not in ids and len(set(ids.tolist())) == len(ids)
            assert (ids < 4000).all() and (ep["levels"][ids] >= lc).all()
        assert (nb[e, int(ep["levels"][e]) + 1:] < 0).all()
    hits = {"serial": 0, "parallel": 0}
    queries = gen(40, 16, seed=52, dist="normal")
    for q in queries:
        d = ((data.astype(np.float64) - q) ** 2).sum(axis=1)
        kth = np.sort(d)[9]
        hits["serial"] += int((d[serial.search(q, 40, 10)[0]] <= kth).sum())
        hits["parallel"] += int((d[par.search(q, 40, 10)[0]] <= kth).sum())
    assert hits["parallel"] >= hits["serial"] - 12      # 3 points of 400
    # every element can be found again (it was linked into the graph)
    for e in range(0, 4000, 97):
        rows, dist, _ = par.search(data[e], 40, 1)
        assert dist[0] == 0.0


# ------------------------------------------------------------------ bit vectors
def test_bit_distances_match_the_reference_known_answers(oracle):
    """hamming_distance / jaccard_distance (src/bitvec.c:45-70, src/bitutils.c:49-131) against every
    known answer of test/expected/bit.out, error texts included"""
    cases = golden("bit_known_answers.json")["cases"]
    assert len(cases) >= 26
    for c in cases:
        rc, value = oracle.bit_sql("ora_" + c["func"], c["a"], c["b"])
        if "error" in c:
            assert rc != 0 and oracle.last_error() == c["error"], (c, oracle.last_error())
        else:
            assert rc == 0 and value == c["value"], (c, value)


def test_bit_oracle_matches_the_compiled_reference(oracle):
    """the restatement against oracle/_ref (the reference's src/bitutils.c compiled unmodified, whichever of
    its Default / AVX-512 variants this CPU dispatches to) on random bit vectors of every tail length"""
    try:
        ref = po.Ref()
    except FileNotFoundError:
        pytest.skip("oracle/_ref not built")
    if not hasattr(ref.lib, "pgvref_bit_hamming"):
        pytest.skip("oracle/_ref predates the bit kernels")
    rng = np.random.default_rng(5)
    for nbytes in list(range(0, 20)) + [63, 64, 65, 127, 128, 192, 500, 2000]:
        a = rng.integers(0, 256, nbytes + 8, dtype=np.uint8)
        b = rng.integers(0, 256, nbytes + 8, dtype=np.uint8)
        assert oracle.lib.ora_bit_hamming(nbytes, po._p(a), po._p(b)) == ref.lib.pgvref_bit_hamming(nbytes, po._p(a), po._p(b))
        oracle.lib.ora_bit_jaccard.restype = C.c_double
        assert oracle.lib.ora_bit_jaccard(nbytes, po._p(a), po._p(b)) == ref.lib.pgvref_bit_jaccard(nbytes, po._p(a), po._p(b))


def test_bench_runners_give_the_single_thread_answers(oracle):
    """oracle_bench.c (bench.py's cpu_baseline): pinned threads round ora_ivf_search / ora_ivf_assign and the
    spread placement copy change where and how often things run, never what they return"""
    import numpy as np
    from helpers import CpuIvf, gen
    from oracle import pyoracle as po
    data = gen(3000, 24, seed=61, dist="clustered", clusters=8)
    ivf = CpuIvf(oracle, po.OPS_L2, po.ORA_F32, data, 8)
    queries = gen(10, 24, seed=62, dist="clustered", clusters=8)
    vec, release = oracle.spread(ivf.vectors, 3)
    assert np.array_equal(vec, ivf.vectors)
    s = oracle.index_struct(po.OPS_L2, po.ORA_F32, ivf.centers, ivf.list_offsets, vec, ivf.tids)
    answers, done, secs = oracle.bench_search(s, queries, 3, 5, 4, 0.05)
    assert done >= 10 and secs > 0
    for i, q in enumerate(queries):
        wt, wd = oracle.search(ivf.struct, q, 3, 5)
        assert answers[i][0].tolist() == wt.tolist() and answers[i][1].tolist() == wd.tolist()
    release()
    got, secs = oracle.bench_assign(po.OPS_L2, po.ORA_F32, ivf.centers, data, 3)
    want, _ = oracle.assign(po.OPS_L2, po.ORA_F32, ivf.centers, data)
    assert got.tolist() == np.asarray(want).tolist() and secs > 0
    assert oracle.lib.ora_bench_cpus() >= 1


# ---------------------------------------------------------------------------------------------------------------------
# fp32: the reference's src/vector.c ITSELF, compiled unmodified into oracle/_ref/libpgvref32.so (oracle/Makefile,
# oracle/ref_glue32.c over the declaration-only stand-ins of ext/shim).  Its kernels are static; they are reached the way
# SQL reaches them, through l2_distance / vector_l2_squared_distance / inner_product / ... (src/vector.c:579-780).
REF32 = os.path.exists(os.path.join(os.path.dirname(po.__file__), "_ref", "libpgvref32.so"))
REF_NAME = {"l2_distance": "l2_distance", "inner_product": "inner_product",
            "negative_inner_product": "vector_negative_inner_product", "cosine_distance": "cosine_distance",
            "l1_distance": "l1_distance"}


@pytest.mark.skipif(not REF32, reason="oracle/_ref/libpgvref32.so not built (reference tree absent)")
def test_fp32_kernels_match_the_references_own_vector_c_bit_for_bit(oracle):
    """the restated fp32 kernels and wrappers against the reference's compiled src/vector.c: every dimension 1..70
    (all vector-loop tails), the BASELINE dimensions, the maximum 16 000, three magnitudes"""
    ref = po.Ref32()
    rng = np.random.default_rng(32)
    pairs = [("ora_l2_squared_distance", "vector_l2_squared_distance"), ("ora_l2_distance", "l2_distance"),
             ("ora_inner_product", "inner_product"), ("ora_negative_inner_product", "vector_negative_inner_product"),
             ("ora_cosine_distance", "cosine_distance"), ("ora_spherical_distance", "vector_spherical_distance"),
             ("ora_l1_distance", "l1_distance")]
    checked = 0
    for dim in list(range(1, 71)) + [100, 127, 128, 129, 255, 256, 500, 768, 1000, 1536, 2000, 3072, 4096, 16000]:
        for scale in (1.0, 1e3, 1e-3):
            a = (rng.standard_normal(dim) * scale).astype(np.float32)
            b = (rng.standard_normal(dim) * scale).astype(np.float32)
            for mine, theirs in pairs:
                rc, x = oracle.sql(mine, a, b)
                rc2, y = ref.call(theirs, a, b)
                assert rc == 0 and rc2 == 0
                assert x == y or (math.isnan(x) and math.isnan(y)), (mine, dim, scale, x, y)
                checked += 1
            n1 = oracle.lib.ora_vector_norm(dim, po._p(a))
            rc2, n2 = ref.call("vector_norm", a)
            assert rc2 == 0 and n1 == n2, (dim, scale, n1, n2)
            out = np.zeros_like(a)
            assert oracle.lib.ora_l2_normalize(dim, po._p(a), po._p(out)) == 0
            rc2, out2 = ref.l2_normalize(a)
            assert rc2 == 0
            np.testing.assert_array_equal(out, out2)
    assert checked > 1500


@pytest.mark.skipif(not REF32, reason="oracle/_ref/libpgvref32.so not built (reference tree absent)")
def test_the_references_own_vector_c_gives_the_known_answers_of_its_tests():
    """the fixture the oracle is pinned with (test/expected/vector_type.out) holds for the compiled reference too --
    including overflow to Infinity, NaN, the clamp of cosine_distance, and the dimension-mismatch ERROR text"""
    ref = po.Ref32()
    seen = 0
    for case in CASES:
        if case["type"] != "vector":
            continue
        func, args = case["func"], case["args"]
        if func == "vector_norm":
            rc, got = ref.call("vector_norm", args[0])
            assert rc == 0 and _apply_wrap(got, case.get("wrap")) == pytest.approx(_expect(case["expect"]), rel=1e-6)
        elif func == "l2_normalize":
            rc, out = ref.l2_normalize(args[0])
            if "error" in case:
                assert rc == 1
            else:
                assert rc == 0
                np.testing.assert_array_equal(out, np.asarray(case["expect"], dtype=np.float32))
        elif func in REF_NAME:
            rc, got = ref.call(REF_NAME[func], args[0], args[1])
            if "error" in case:
                assert rc == 1 and ref.last_error() == case["error"]
            else:
                want = _expect(case["expect"])
                assert rc == 0
                assert (math.isnan(got) and math.isnan(want)) or got == want, (case, got)
        else:
            continue
        seen += 1
    assert seen >= 30


@pytest.mark.skipif(not REF32, reason="oracle/_ref/libpgvref32.so not built (reference tree absent)")
def test_oracle_scan_distances_are_the_references_on_real_valued_rows(oracle):
    """GetScanItems' per-tuple call (FUNCTION 1 of the opclass) over 2000 rows x 1536: what the oracle's list scan
    computes equals what the reference's vector_l2_squared_distance / vector_negative_inner_product return, bitwise"""
    ref = po.Ref32()
    rng = np.random.default_rng(7)
    rows = rng.standard_normal((2000, 1536)).astype(np.float32)
    q = rng.standard_normal(1536).astype(np.float32)
    for ops, name in ((po.OPS_L2, "vector_l2_squared_distance"), (po.OPS_IP, "vector_negative_inner_product")):
        want = ref.rows(name, q, rows)
        oracle.lib.ora_index_distance.restype = C.c_double
        got = np.array([oracle.lib.ora_index_distance(ops, po.ORA_F32, 1536, po._p(rows[i]), po._p(q)) for i in range(len(rows))])
        np.testing.assert_array_equal(got, want)
