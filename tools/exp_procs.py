#!/usr/bin/env python3
"""Round-3 experiment: N backend PROCESSES, one query at a time each, on one device mirror (a 200 k x 256 index:
small enough that the device work is short and the scheduling of the processes' queues shows).  Run under different
GPU_MAX_HW_QUEUES; prints one JSON object."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pgvector_amd import _host, api  # noqa: E402

rng = np.random.default_rng(3)
n, dim, lists, nq, probes, k = 200000, 256, 100, 64, 5, 10
data = rng.standard_normal((n, dim), dtype=np.float32)
centers = data[rng.choice(n, lists, replace=False)].copy()
ctx = api.Context(0)
lst, _ = api.assign(ctx, api.PGV_L2SQ, api.PGV_F32, dim, centers, data)
order = np.argsort(lst, kind="stable")
off = np.zeros(lists + 1, dtype=np.int64)
np.cumsum(np.bincount(lst, minlength=lists), out=off[1:])
ix = api.IvfIndex(ctx, api.PGV_L2SQ, api.PGV_F32, dim, centers, off, data[order], order.astype(np.uint64))
queries = data[rng.choice(n, nq, replace=False)] + 0.01
out = {"GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES")}
for nb in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "4,16,32").split(",")]:
    try:
        out[str(nb)] = _host.run_backend_processes(ix, queries, probes, k, 0, nb, 200)
    except Exception as e:  # noqa: BLE001
        out[str(nb)] = {"error": repr(e)}
print(json.dumps(out))
