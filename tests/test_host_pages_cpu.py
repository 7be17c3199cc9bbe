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
