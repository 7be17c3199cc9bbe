"""CPU tests of the C host glue that needs no GPU: the IVFFlat on-disk format
(writer + stager round trip, page capacities, short varlena headers, insert)."""
import numpy as np
import pytest

from pgvector_amd import _host

from helpers import gen


def _layout(n, lists, seed):
    rng = np.random.default_rng(seed)
    sizes = rng.multinomial(n, np.ones(lists) / lists)
    return np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)


@pytest.mark.parametrize("dtype,dim", [(0, 3), (0, 29), (0, 30), (0, 128), (0, 768), (0, 1536), (0, 2000),
                                       (1, 3), (1, 61), (1, 62), (1, 3072), (1, 4000)])
def test_write_then_stage_round_trip(dtype, dim):
    n, lists = 57, 5
    np_t = np.float32 if dtype == 0 else np.float16
    centers = gen(lists, dim, seed=1).astype(np_t)
    vectors = gen(n, dim, seed=2).astype(np_t)
    off = _layout(n, lists, 3)
    off[2] = off[1]  # an empty list
    tids = ((np.arange(n, dtype=np.uint64) + 7) << np.uint64(16)) | np.uint64(3)
    rel = _host.Relation()
    rel.write_index(dtype, centers, off, vectors, tids)
    img = rel.stage(dtype)
    assert (img.dim, img.lists, img.nrows) == (dim, lists, n)
    np.testing.assert_array_equal(img.centers.view(np.uint8), centers.view(np.uint8))
    np.testing.assert_array_equal(img.list_offsets, off)
    np.testing.assert_array_equal(img.vectors.view(np.uint8), vectors.view(np.uint8))
    np.testing.assert_array_equal(img.tids, tids)


def test_page_capacities_match_the_reference_format():
    """8192-byte pages, 24-byte header, 8-byte special, 4-byte line pointers:
    15 entry tuples per page at 128-d fp32, 2 at 768-d, 1 at 1536-d fp32 and 3072-d fp16"""
    for dtype, dim, per_page in [(0, 128, 15), (0, 768, 2), (0, 1536, 1), (1, 3072, 1)]:
        n = per_page * 4
        np_t = np.float32 if dtype == 0 else np.float16
        rel = _host.Relation()
        rel.write_index(dtype, np.zeros((1, dim), np_t), np.array([0, n]), np.zeros((n, dim), np_t),
                        np.arange(n, dtype=np.uint64))
        # block 0 meta, block 1 list page, then the entry chain
        assert rel.nblocks == 2 + 4, (dim, rel.nblocks)
    # meta page fields (src/ivfflat.h:46-52,251-257)
    page0 = rel.page(0)
    magic, version = np.frombuffer(page0[24:32].tobytes(), dtype=np.uint32)
    dims, lists = np.frombuffer(page0[32:36].tobytes(), dtype=np.uint16)
    assert (magic, version, dims, lists) == (0x14FF1A7, 1, 3072, 1)
    # special space: nextblkno / page id 0xFF84
    page2 = rel.page(2)
    assert np.frombuffer(page2[8190:8192].tobytes(), dtype=np.uint16)[0] == 0xFF84
    assert np.frombuffer(page2[8184:8188].tobytes(), dtype=np.uint32)[0] == 3


def test_too_many_dimensions_for_a_page():
    rel = _host.Relation()
    with pytest.raises(Exception):
        rel.write_index(0, np.zeros((1, 2100), np.float32), np.array([0, 0]), np.zeros((0, 2100), np.float32),
                        np.zeros(0, np.uint64))


def test_insert_appends_and_extends_the_chain():
    dim = 768
    rel = _host.Relation()
    base = gen(3, dim, seed=5)
    rel.write_index(0, gen(2, dim, seed=4), np.array([0, 2, 3]), base, np.arange(3, dtype=np.uint64))
    before = rel.nblocks
    extra = gen(5, dim, seed=6)
    for i, v in enumerate(extra):
        rel.insert(0, 1, v, 100 + i)
    img = rel.stage(0)
    np.testing.assert_array_equal(img.list_offsets, [0, 2, 8])
    np.testing.assert_array_equal(img.vectors[3:], extra)
    np.testing.assert_array_equal(img.tids[3:], np.arange(100, 105))
    assert rel.nblocks == before + 2  # 1 + 5 tuples at 2 per page = 3 pages, one existed


def test_host_float_to_half_matches_the_oracle(oracle):
    rng = np.random.default_rng(9)
    vals = np.concatenate([rng.standard_normal(5000) * 10.0 ** rng.integers(-10, 6, 5000),
                           [0.0, -0.0, 65504.0, 65519.9, 65520.0, 65536.0, 1e-8, 5.96e-8, 2.98e-8, 2.9802322e-8,
                            6.1e-5, 6.097e-5, np.inf, -np.inf]])
    for f in vals:
        f = float(np.float32(f))
        assert _host.lib.pgv_host_float_to_half(f) == oracle.lib.ora_float_to_half(f), f


@pytest.mark.parametrize("dtype,dim", [(0, 3), (0, 128), (0, 768), (1, 1024)])
def test_bulkdelete_removes_dead_tuples_and_reuses_the_page(dtype, dim):
    """ivfflatbulkdelete (src/ivfvacuum.c:18-143): dead heap TIDs vanish, the survivors keep their
    order, the counters are the reference's (tuples_removed / num_index_tuples), the list's insert
    page moves to the first page that lost tuples and the next insert lands there"""
    n, lists = 90, 4
    np_t = np.float32 if dtype == 0 else np.float16
    centers = gen(lists, dim, seed=1).astype(np_t)
    vectors = gen(n, dim, seed=2).astype(np_t)
    off = _layout(n, lists, 5)
    tids = ((np.arange(n, dtype=np.uint64) + 1) << np.uint64(16)) | np.uint64(1)
    rel = _host.Relation()
    rel.write_index(dtype, centers, off, vectors, tids)
    gen0 = rel.generation
    blocks0 = rel.nblocks
    dead = tids[::3]
    removed, remaining = rel.bulkdelete(dead)
    assert (removed, remaining) == (len(dead), n - len(dead))
    assert rel.generation > gen0 and rel.nblocks == blocks0  # pages are emptied, not unlinked
    img = rel.stage(dtype)
    keep = np.ones(n, bool)
    keep[::3] = False
    np.testing.assert_array_equal(img.tids, tids[keep])
    np.testing.assert_array_equal(img.vectors.view(np.uint8), vectors[keep].view(np.uint8))
    want_off = np.concatenate([[0], np.cumsum([keep[off[i]:off[i + 1]].sum() for i in range(lists)])])
    np.testing.assert_array_equal(img.list_offsets, want_off)
    # nothing dead: no change, no invalidation
    g1 = rel.generation
    assert rel.bulkdelete([]) == (0, n - len(dead)) and rel.generation == g1
    # the freed space is reused: inserting into a list that lost tuples does not extend the relation
    lst = int(np.argmax(np.diff(off)))
    new_tid = np.uint64(0xABCDEF0001)
    rel.insert(dtype, lst, vectors[0], int(new_tid))
    assert rel.nblocks == blocks0
    img2 = rel.stage(dtype)
    assert img2.nrows == n - len(dead) + 1 and int(new_tid) in set(int(t) for t in img2.tids[img2.list_offsets[lst]:img2.list_offsets[lst + 1]])


# ------------------------------------------------------------------- HNSW pages
def _random_graph(n, m, seed):
    """a random but well-formed neighbor table in the HnswNeighborTupleData layout"""
    rng = np.random.default_rng(seed)
    levels = np.minimum((-np.log(rng.random(n)) / np.log(m)).astype(np.int32), 5)
    nbr_start = np.concatenate([[0], np.cumsum((levels.astype(np.int64) + 2) * m)])
    nbr = np.full(int(nbr_start[-1]), -1, np.int32)
    for e in range(n):
        for lc in range(levels[e] + 1):
            lm = 2 * m if lc == 0 else m
            pool = np.nonzero(levels >= lc)[0]
            pool = pool[pool != e]
            k = int(rng.integers(0, min(lm, len(pool)) + 1))
            o = int(nbr_start[e]) + (int(levels[e]) - lc) * m
            nbr[o:o + k] = rng.choice(pool, k, replace=False)
    entry = int(np.argmax(levels))
    return levels, nbr_start, nbr, entry


@pytest.mark.parametrize("dtype,dim,m", [(0, 3, 4), (0, 128, 16), (0, 1536, 16), (1, 3072, 8), (0, 2000, 5)])
def test_hnsw_write_then_stage_round_trip(dtype, dim, m):
    """FlushPages (src/hnswbuild.c:300-312) then the scan-side walk: every element comes back with its
    level, vector, heap TIDs and neighbors; slots follow page order (newest element first)"""
    n = 83
    np_t = np.float32 if dtype == 0 else np.float16
    vectors = gen(n, dim, seed=7).astype(np_t)
    tids = ((np.arange(n, dtype=np.uint64) + 5) << np.uint64(16)) | np.uint64(2)
    levels, nbr_start, nbr, entry = _random_graph(n, m, 11)
    # rows 10 and 20 are duplicates of element 4: only their heap TIDs are kept
    dup = np.full(n, -1, np.int32)
    dup[[10, 20]] = 4
    for e in range(n):  # nobody may point at a duplicate
        seg = nbr[nbr_start[e]:nbr_start[e + 1]]
        seg[np.isin(seg, [10, 20])] = -1
    rel = _host.Relation()
    rel.write_hnsw(dtype, m, 64, vectors, tids, levels, nbr_start, nbr, entry, dup)
    img = rel.stage_hnsw(dtype)
    kept = [e for e in range(n - 1, -1, -1) if dup[e] < 0]  # page order
    assert img["n"] == len(kept) and (img["dim"], img["m"], img["ef_construction"]) == (dim, m, 64)
    slot_of = {e: s for s, e in enumerate(kept)}
    assert img["entry"] == slot_of[entry]
    np.testing.assert_array_equal(img["levels"], levels[kept])
    np.testing.assert_array_equal(img["vectors"].view(np.uint8), vectors[kept].view(np.uint8))
    np.testing.assert_array_equal(img["heaptids"][:, 0], tids[kept])
    s4 = slot_of[4]
    assert img["heaptids"][s4, 1] == tids[10] and img["heaptids"][s4, 2] == tids[20]
    assert (img["heaptids"][s4, 3:] == np.uint64(0xFFFFFFFFFFFFFFFF)).all()
    for s, e in enumerate(kept):
        raw = nbr[nbr_start[e]:nbr_start[e + 1]]
        # an invalid TID ends a layer's list (HnswLoadNeighborTids breaks, src/hnswutils.c:785-786): per layer slice
        # (level .. 1: m entries, layer 0: 2 m) the entries before the first -1 survive, the rest reads as -1
        want, at = [], 0
        for lc in range(int(levels[e]), -1, -1):
            ln = 2 * m if lc == 0 else m
            sl = list(raw[at:at + ln])
            cut = sl.index(-1) if -1 in sl else ln
            want += [slot_of[x] for x in sl[:cut]] + [-1] * (ln - cut)
            at += ln
        assert at == len(raw)
        np.testing.assert_array_equal(img["nbr"][img["nbr_start"][s]:img["nbr_start"][s + 1]], np.array(want, np.int32))


def test_hnsw_page_layout_matches_the_reference_format():
    """src/hnsw.h:40-47, 334-392: meta page fields, page id 0xFF90, element tuple = 72-byte header + the
    vector varlena (6224 bytes at 1536-d fp32), neighbor tuple = 4 + 6 * (level + 2) * m,
    element and neighbor tuple on the same page"""
    m, n = 16, 12
    for dim, per_page in [(1536, 1), (128, 10)]:
        vectors = gen(n, dim, seed=1).astype(np.float32)
        levels = np.zeros(n, np.int32)
        nbr_start = np.arange(n + 1, dtype=np.int64) * 2 * m
        nbr = np.full(n * 2 * m, -1, np.int32)
        nbr[0] = 1
        rel = _host.Relation()
        rel.write_hnsw(0, m, 64, vectors, np.arange(1, n + 1, dtype=np.uint64) << np.uint64(16) | np.uint64(1), levels,
                       nbr_start, nbr, 3)
        assert rel.nblocks == 1 + -(-n // per_page), (dim, rel.nblocks)
        page0 = rel.page(0)
        magic, version, dims = np.frombuffer(page0[24:36].tobytes(), dtype=np.uint32)
        mm, efc = np.frombuffer(page0[36:40].tobytes(), dtype=np.uint16)
        assert (magic, version, dims, mm, efc) == (0xA953A953, 1, dim, m, 64)
        entry_blk = np.frombuffer(page0[40:44].tobytes(), dtype=np.uint32)[0]
        entry_off, = np.frombuffer(page0[44:46].tobytes(), dtype=np.uint16)
        entry_level, = np.frombuffer(page0[46:48].tobytes(), dtype=np.int16)
        insert_page, = np.frombuffer(page0[48:52].tobytes(), dtype=np.uint32)
        assert entry_level == 0 and insert_page == rel.nblocks - 1 and entry_blk >= 1 and entry_off >= 1
        page1 = rel.page(1)
        assert np.frombuffer(page1[8190:8192].tobytes(), dtype=np.uint16)[0] == 0xFF90
        lp1, lp2 = np.frombuffer(page1[24:32].tobytes(), dtype=np.uint32)
        assert lp1 >> 17 == ((72 + 8 + 4 * dim + 7) // 8) * 8 and lp2 >> 17 == 200  # 6224 at 1536-d
        off1 = int(lp1 & 0x7FFF)
        etup = page1[off1:off1 + 80]
        assert etup[0] == 1 and etup[1] == 0 and etup[2] == 0 and etup[3] == 1  # type, level, deleted, version
        nb_hi, nb_lo, nb_off = np.frombuffer(etup[64:70].tobytes(), dtype=np.uint16)
        assert (int(nb_hi) << 16 | int(nb_lo), nb_off) == (1, 2)  # its neighbor tuple: same page, next offset
        vl_len, = np.frombuffer(etup[72:76].tobytes(), dtype=np.uint32)
        assert vl_len >> 2 == 8 + 4 * dim and np.frombuffer(etup[76:78].tobytes(), dtype=np.int16)[0] == dim
        ntup = page1[int(lp2 & 0x7FFF):int(lp2 & 0x7FFF) + 8]
        assert ntup[0] == 2 and np.frombuffer(ntup[2:4].tobytes(), dtype=np.uint16)[0] == 2 * m


def test_hnsw_stage_rejects_other_relations():
    rel = _host.Relation()
    rel.write_index(0, np.zeros((1, 4), np.float32), np.array([0, 1]), np.zeros((1, 4), np.float32),
                    np.array([1], np.uint64))
    with pytest.raises(Exception):
        rel.stage_hnsw(0)


def test_oracle_walks_the_pages_like_it_walks_the_arrays(oracle):
    """oracle_pages.c (GetScanLists / GetScanItems over the 8 KB page image, src/ivfscan.c:47-187) against
    oracle_ivf.c over the contiguous arrays of the same index: identical tids and distances -- for rows with a
    4-byte varlena header (dim 40), with the 1-byte short header (dim 3) and halfvec"""
    from oracle import pyoracle as po
    from helpers import CpuIvf
    for ops, dtype, dim in [(po.OPS_L2, po.ORA_F32, 40), (po.OPS_L2, po.ORA_F32, 3), (po.OPS_IP, po.ORA_F16, 24)]:
        data = gen(3000, dim, seed=61, dist="clustered", clusters=10, dtype=dtype)
        ivf = CpuIvf(oracle, ops, dtype, data, 12)
        rel = _host.Relation()
        rel.write_index(0 if dtype == po.ORA_F32 else 1, ivf.centers, ivf.list_offsets, ivf.vectors, ivf.tids)
        for q in gen(8, dim, seed=62, dist="clustered", clusters=10, dtype=dtype):
            for probes in (1, 3, 12):
                wt, wd = oracle.search(ivf.struct, q, probes, 25)
                gt, gd, scanned = oracle.pages_search(rel.rel.pages, rel.nblocks, ops, dtype, q, probes, 25)
                np.testing.assert_array_equal(gt, wt)
                np.testing.assert_array_equal(gd, wd)
        # NULL query: the first lists, every tuple at distance 0
        gt, gd, scanned = oracle.pages_search(rel.rel.pages, rel.nblocks, ops, dtype, None, 2, 10 ** 6)
        assert scanned == int(ivf.list_offsets[2]) and (gd == 0).all()


def test_host_normalize_matches_the_oracle_bit_for_bit(oracle):
    """pgv_host_normalize_value (what the host scan does to a cosine query and the host build to cosine rows:
    l2_normalize / halfvec_l2_normalize, src/vector.c:785-819, src/halfvec.c:724-759) against the oracle's restatement --
    which tests/test_oracle_golden.py holds to the reference's compiled src/vector.c bit for bit."""
    import ctypes as C
    lib = _host.lib
    lib.pgv_host_normalize_value.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lib.pgv_host_normalize_value.restype = C.c_int
    rng = np.random.default_rng(5)
    for dtype, np_t, fn, dims in ((0, np.float32, oracle.lib.ora_l2_normalize, [1, 2, 3, 7, 8, 16, 31, 32, 100, 768, 1536, 2000]),
                                  (1, np.float16, oracle.lib.ora_halfvec_l2_normalize, [1, 2, 3, 7, 8, 16, 31, 32, 100, 768, 3072, 4000])):
        for dim in dims:
            for _ in range(20):
                a = (rng.standard_normal(dim) * 10.0 ** rng.integers(-2, 2)).astype(np_t)
                got, want = np.empty(dim, np_t), np.empty(dim, np_t)
                assert lib.pgv_host_normalize_value(dtype, dim, a.ctypes.data, got.ctypes.data) == 1
                fn(dim, a.ctypes.data, want.ctypes.data)
                np.testing.assert_array_equal(got.view(np.uint8), want.view(np.uint8))
        zero = np.zeros(8, np_t)
        out = np.ones(8, np_t)
        assert lib.pgv_host_normalize_value(dtype, 8, zero.ctypes.data, out.ctypes.data) == 0 and not out.any()
