"""Run by tests/test_gpu_round5.py in a process of its own with PGV_SCAN_WIDE=1 (the library reads it once): the 64-query
form of the batched list scan for fp32 AND fp16 shapes against the oracle.  Prints 'WIDE-OK <cases>' on success."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from helpers import CpuIvf, assert_topk_equiv, gen  # noqa: E402
from oracle import pyoracle as po  # noqa: E402
from pgvector_amd import api  # noqa: E402

CASES = [("l2", "f32", 128, 16, 8, 700), ("ip", "f32", 96, 32, 6, 300), ("l2", "f16", 264, 24, 8, 260),
         ("l2", "f32", 1536, 40, 10, 200), ("l2", "f32", 8, 12, 6, 150)]


def main():
    assert os.environ.get("PGV_SCAN_WIDE") == "1"
    oracle = po.Oracle()
    ctx = api.Context(0)
    for ops, dt, dim, lists, probes, nq in CASES:
        odt = po.ORA_F32 if dt == "f32" else po.ORA_F16
        oops = po.OPS_L2 if ops == "l2" else po.OPS_IP
        dist = "int" if dim == 8 else "clustered"
        data = gen(6000, dim, seed=971, dist=dist, dtype=odt, clusters=lists)
        ivf = CpuIvf(oracle, oops, odt, data, lists)
        ix = api.IvfIndex(ctx, ivf.metric, api.PGV_F32 if dt == "f32" else api.PGV_F16, dim, ivf.centers, ivf.list_offsets,
                          ivf.vectors, ivf.tids)
        queries = gen(nq, dim, seed=972, dist=dist, dtype=odt, clusters=lists)
        d, s, t = ix.search_batch(queries, probes, 10, want_tid=True)
        for i in range(nq):
            wt, wd = oracle.search(ivf.struct, queries[i], probes, 10)
            scale = 0.0
            if ops == "ip":
                scale = 1e-5 * float(np.max(np.abs(data.astype(np.float64)) @ np.abs(queries[i].astype(np.float64))))
            assert_topk_equiv(np.asarray(t[i])[:len(wt)].astype(np.uint64).tolist(), np.asarray(d[i])[:len(wt)], wt.tolist(), wd,
                              atol=max(scale, 1e-30), what="wide scan %s %s dim %d q %d" % (ops, dt, dim, i))
        ix.close()
    ctx.close()
    print("WIDE-OK %d" % len(CASES))


if __name__ == "__main__":
    main()
