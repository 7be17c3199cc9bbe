"""A/B timing of pgv_hnsw_search on fp16 elements (the halfvec HNSW opclasses): a random 32-regular layer-0 graph over
clustered vectors is enough to time the search kernel (run once per library: PGV_HIP_LIB=... python tools/hnsw_f16_ab.py)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pgvector_amd import api  # noqa: E402

n, dim, m, nq, ef, k = 200_000, int(os.environ.get("DIM", 1536)), 16, 4096, 100, 10
rng = np.random.default_rng(5)
centers = rng.standard_normal((64, dim)).astype(np.float32)
rows = (centers[rng.integers(0, 64, n)] + 0.35 * rng.standard_normal((n, dim)).astype(np.float32)).astype(np.float16)
queries = (centers[rng.integers(0, 64, nq)] + 0.35 * rng.standard_normal((nq, dim)).astype(np.float32)).astype(np.float16)
ctx = api.Context(0)
out = {"lib": os.environ.get("PGV_HIP_LIB", "tree"), "n": n, "dim": dim, "nq": nq, "ef": ef}
for name, dtype, data, q in (("f16", api.PGV_F16, rows, queries), ("f32", api.PGV_F32, rows.astype(np.float32), queries.astype(np.float32))):
    h = api.Hnsw(ctx, api.PGV_L2SQ, dtype, dim, data)
    nbr = rng.integers(0, n, (n, 2 * m), dtype=np.int32)
    h.set_graph(m, 0, np.zeros(n, np.int32), np.arange(n + 1, dtype=np.int64) * 2 * m, nbr.ravel())
    e0, d0, s0 = h.search(q, ef, k)
    ctx.sync()
    best = 1e9
    for _ in range(5):
        t = time.perf_counter()
        e, d, s = h.search(q, ef, k)
        ctx.sync()
        best = min(best, time.perf_counter() - t)
    out[name] = {"ms": round(best * 1e3, 3), "qps": round(nq / best), "scored_mean": float(np.mean(s)),
                 "checksum": int(np.asarray(e).astype(np.int64).sum()), "same_as_first": bool((np.asarray(e) == np.asarray(e0)).all())}
    h.close()
print(json.dumps(out))
