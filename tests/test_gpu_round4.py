"""GPU tests of round 4: what the deterministic completeness bound of the MFMA L2 scan assumes about the matrix
pipeline's arithmetic, pinned on the hardware; the admission gate under load; pooled clients that die."""
import os

import numpy as np
import pytest

from pgvector_amd import api

pytestmark = pytest.mark.gpu


def _ip_values(ctx, dtype, rows, queries):
    """-q.x of every (query, row) pair as mfma_scan_kernel computes it: an inner-product index of ONE list, the whole
    batch probing it (the matrix-core scan's values are the result for inner product: no recheck in between)"""
    n, dim = rows.shape
    ix = api.IvfIndex(ctx, api.PGV_NEG_IP, dtype, dim, rows[:1].copy(), np.array([0, n], dtype=np.int64), rows,
                      np.arange(n, dtype=np.uint64))
    dist, slot, _ = ix.search_batch(queries, 1, n)
    ix.close()
    out = np.zeros((queries.shape[0], n), dtype=np.float32)
    for i in range(queries.shape[0]):
        out[i, slot[i]] = dist[i]
    return -out


@pytest.mark.parametrize("nq", [8, 64])          # the 16-wide and the 32-wide MFMA shapes of the kernel
def test_mfma_fp32_chain_rounds_to_nearest(ctx, nq):
    """scan_bound (pgv_internal.h) charges one unit roundoff per operation of an accumulator chain.  Chain 0 of the
    four-chain kernel adds the products of elements 0, 4, 8, 12, ... in that order:
      * c + p with c = 1 and p = 1.5 x 2^-24 (a quarter ulp above the half-way point): round-to-nearest gives 1 + 2^-23,
        a truncating adder gives 1;
      * a product (1 + 2^-12)(1 + 3 x 2^-12) = 1 + 2^-10 + 1.5 x 2^-23: nearest is 1 + 2^-10 + 2^-22, truncation
        1 + 2^-10 + 2^-23.
    Either failing means the pipeline rounds worse than the bound assumes."""
    dim, n = 64, 32
    rows = np.zeros((n, dim), dtype=np.float32)
    queries = np.zeros((nq, dim), dtype=np.float32)
    rows[0, 0], rows[0, 8] = 1.0, 1.0
    queries[:, 0], queries[:, 8] = 1.0, np.float32(1.5 * 2.0 ** -24)
    rows[1, 4] = np.float32(1.0 + 3 * 2.0 ** -12)
    queries[:, 4] = np.float32(1.0 + 2.0 ** -12)
    v = _ip_values(ctx, api.PGV_F32, rows, queries)
    assert (v[:, 0] == np.float32(1.0 + 2.0 ** -23)).all(), v[:, 0].astype(np.float64) - 1.0
    want = np.float32(1.0 + 2.0 ** -10 + 2.0 ** -22)
    assert (v[:, 1] == want).all(), (v[:, 1].astype(np.float64) - 1.0 - 2.0 ** -10) * 2.0 ** 23


@pytest.mark.parametrize("nq", [8, 64])
def test_mfma_fp16_chain_rounds_to_nearest(ctx, nq):
    """fp16 products are exact in fp32; what can round is the accumulation.  Elements 0 and 128 belong to the same
    chain in both kernel shapes (the chains repeat every 64 / 128 halves): 1 + 1.5 x 2^-24 again."""
    dim, n = 256, 32
    rows = np.zeros((n, dim), dtype=np.float16)
    queries = np.zeros((nq, dim), dtype=np.float16)
    rows[0, 0], rows[0, 128] = 1.0, np.float16(2.0 ** -12)
    queries[:, 0], queries[:, 128] = 1.0, np.float16(1.5 * 2.0 ** -12)
    v = _ip_values(ctx, api.PGV_F16, rows, queries)
    assert (v[:, 0] == np.float32(1.0 + 2.0 ** -23)).all(), v[:, 0].astype(np.float64) - 1.0
