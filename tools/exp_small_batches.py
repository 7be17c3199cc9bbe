#!/usr/bin/env python3
"""pgv_search_batch on batches of a few queries (4 / 16 / 64) against the headline index shape (rows x dim fp32,
lists, probes): ms per batch, and -- under `rocprofv3 --kernel-trace --stats` -- which kernels the time goes to.
`python tools/exp_small_batches.py [--batches 16 --reps 200]` (on the GPU)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1000000)
    ap.add_argument("--dim", type=int, default=1536)
    ap.add_argument("--lists", type=int, default=1000)
    ap.add_argument("--probes", type=int, default=10)
    ap.add_argument("--batches", default="4,16,64")
    ap.add_argument("--reps", type=int, default=200)
    a = ap.parse_args()
    import torch

    from pgvector_amd import api
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    comps = torch.rand((a.lists, a.dim), device=dev, generator=g)
    ctx = api.Context(0, stream=0)
    b = api.IvfBuilder(ctx, api.PGV_L2SQ, api.PGV_F32, a.dim, comps, expected_rows=a.rows)
    for lo in range(0, a.rows, 100000):
        hi = min(a.rows, lo + 100000)
        pick = torch.randint(0, a.lists, (hi - lo,), device=dev, generator=g)
        b.add((comps[pick] + 0.1 * torch.randn((hi - lo, a.dim), device=dev, generator=g)).contiguous())
    ix, off, _ = b.finish()
    b.close()
    pool = 16
    for nb in [int(x) for x in a.batches.split(",")]:
        qs = []
        for _ in range(pool):
            pick = torch.randint(0, a.lists, (nb,), device=dev, generator=g)
            qs.append((comps[pick] + 0.1 * torch.randn((nb, a.dim), device=dev, generator=g)).contiguous())
        od = torch.empty((nb, 10), device=dev, dtype=torch.float32)
        os_ = torch.empty((nb, 10), device=dev, dtype=torch.int64)
        ot = torch.empty((nb, 10), device=dev, dtype=torch.int64)
        for j in range(10):
            ix.search_batch(qs[j % pool], a.probes, 10, want_tid=True, out=(od, os_, ot))
        ctx.sync()
        t0 = time.perf_counter()
        for j in range(a.reps):
            ix.search_batch(qs[j % pool], a.probes, 10, want_tid=True, out=(od, os_, ot))
        ctx.sync()
        s = (time.perf_counter() - t0) / a.reps
        rows = float(np.mean(np.diff(off))) * a.probes * nb
        print(json.dumps({"batch": nb, "ms_per_batch": round(s * 1e3, 4), "qps": round(nb / s),
                          "rows_of_probed_lists_GB": round(rows * a.dim * 4 / 1e9, 3),
                          "TBps_if_every_list_is_read_once_per_query": round(rows * a.dim * 4 / s / 1e12, 2)}), flush=True)
    ix.close()
    ctx.close()


if __name__ == "__main__":
    main()
