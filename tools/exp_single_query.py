#!/usr/bin/env python3
"""Round-3 experiment: N independent backends, one query at a time each (pgv_query_rank + pgv_query_scan), against
one device mirror -- as threads of one process (tools/backends_driver.c) and as processes (tools/pgv_backend.c).
Run under different GPU_MAX_HW_QUEUES to see whether the 16 -> 32 backend regression is the runtime's stream ->
hardware-queue mapping.  Prints one JSON object."""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pgvector_amd import _host, api  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=500_000)
    ap.add_argument("--lists", type=int, default=500)
    ap.add_argument("--dim", type=int, default=1536)
    ap.add_argument("--threads", default="1,8,16,32")
    ap.add_argument("--procs", default="")
    ap.add_argument("--per", type=int, default=400)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(2)
    n, dim, lists, probes, k = args.rows, args.dim, args.lists, 10, 10
    ncomp = max(lists // 4, 1)
    means = torch.rand((ncomp, dim), generator=g, device=dev)
    data = torch.empty((n, dim), device=dev)
    for lo in range(0, n, 1 << 17):
        hi = min(n, lo + (1 << 17))
        comp = torch.randint(0, ncomp, (hi - lo,), generator=g, device=dev)
        data[lo:hi] = means[comp] + 0.1 * torch.randn((hi - lo, dim), generator=g, device=dev)
    ctx = api.Context(0, stream=0)
    centers, _, _ = api.kmeans(ctx, api.PGV_OPS_L2, api.PGV_F32, dim,
                               data[torch.randperm(n, generator=g, device=dev)[:50 * lists]].contiguous(), lists,
                               api.make_rng(seed=3), want_closest=False)
    assign, _ = api.assign(ctx, api.PGV_L2SQ, api.PGV_F32, dim, centers, data, want_dist=False)
    order = torch.argsort(assign.long(), stable=True)
    off = torch.zeros(lists + 1, dtype=torch.int64, device=dev)
    off[1:] = torch.cumsum(torch.bincount(assign.long(), minlength=lists), 0)
    vectors = data[order].contiguous()
    del data
    ix = api.IvfIndex(ctx, api.PGV_L2SQ, api.PGV_F32, dim, centers, off, vectors, order)
    comp = torch.randint(0, ncomp, (256,), generator=g, device=dev)
    qh = np.ascontiguousarray((means[comp] + 0.1 * torch.randn((256, dim), generator=g, device=dev)).cpu().numpy())
    out = {"GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"), "rows": n, "lists": lists, "threads": {}, "processes": {}}
    for nb in [int(x) for x in args.threads.split(",") if x]:
        try:
            r = _host.run_backend_threads(ix, qh, probes, k, nb, args.per)
            out["threads"][str(nb)] = {"rc": 0, "qps": r["qps"], "p50_us": r["latency_us_p50"], "p90_us": r["latency_us_p90"]}
        except Exception as e:  # noqa: BLE001
            out["threads"][str(nb)] = {"error": repr(e)}
        print("threads", nb, out["threads"][str(nb)], file=sys.stderr, flush=True)
    for nb in [int(x) for x in args.procs.split(",") if x]:
        try:
            r = _host.run_backend_processes(ix, qh, probes, k, 0, nb, args.per)
            out["processes"][str(nb)] = r
        except Exception as e:  # noqa: BLE001
            out["processes"][str(nb)] = {"error": repr(e)}
        print("processes", nb, out["processes"][str(nb)], file=sys.stderr, flush=True)
    print(json.dumps(out))
    ix.close()
    ctx.close()


if __name__ == "__main__":
    main()
