#!/usr/bin/env python3
"""pgv_exact_topk at 1 M x 1536 fp32 x 1024 queries: ms per batch; PGV_DENSE_KEEP=0/1 A/B."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pgvector_amd import api  # noqa: E402

dev = torch.device("cuda", 0)
ctx = api.Context(0, stream=torch.cuda.current_stream().cuda_stream)
g = torch.Generator(device=dev)
g.manual_seed(61)
n, dim, nq, k = 1_000_000, 1536, 1024, 10
rows = torch.rand((n, dim), generator=g, device=dev)
queries = torch.rand((nq, dim), generator=g, device=dev)
od = torch.empty((nq, k), device=dev, dtype=torch.float32)
oi = torch.empty((nq, k), device=dev, dtype=torch.int64)
out = {"PGV_DENSE_KEEP": os.environ.get("PGV_DENSE_KEEP")}
for metric, name in ((api.PGV_L2SQ, "l2"), (api.PGV_NEG_IP, "ip")):
    for _ in range(2):
        api.exact_topk(ctx, metric, api.PGV_F32, dim, queries, rows, k, out=(od, oi))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(6):
        api.exact_topk(ctx, metric, api.PGV_F32, dim, queries, rows, k, out=(od, oi))
    torch.cuda.synchronize()
    s = (time.perf_counter() - t0) / 6
    out[name] = {"ms_per_batch": s * 1e3, "tflops": 2.0 * n * nq * dim / s / 1e12}
    # exactness against float64 on a few queries
    q64 = queries[:8].double()
    best = None
    for lo in range(0, n, 100_000):
        v = rows[lo:lo + 100_000].double()
        d = ((q64 * q64).sum(1)[:, None] + (v * v).sum(1)[None, :] - 2 * q64 @ v.T) if name == "l2" else -(q64 @ v.T)
        cand = d if best is None else torch.cat([best, d], 1)
        idx = torch.arange(lo, lo + v.shape[0], device=dev)[None, :].expand(8, -1)
        ci = idx if best is None else torch.cat([bi, idx], 1)
        top = torch.topk(cand, k, dim=1, largest=False)
        best, bi = top.values, torch.gather(ci, 1, top.indices)
    out[name]["ids_equal_float64_top10_of_8_queries"] = bool((torch.sort(bi, 1).values == torch.sort(oi[:8], 1).values).all().item())
print(json.dumps(out))
