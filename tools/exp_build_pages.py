#!/usr/bin/env python3
"""The product IVFFlat build through 8 KB pages (pgv_host_ivf_build_mirror), timed by phase: rows x dim fp32 rows of a
Gaussian mixture, vector_l2_ops.  `python tools/exp_build_pages.py [--rows 1000000 --dim 1536 --lists 1000 --reps 3]`
(on the GPU); prints one JSON object per repetition."""
import argparse
import ctypes
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
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    import torch

    from pgvector_amd import _host, api
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    comps = torch.rand((a.lists, a.dim), device=dev, generator=g)
    rows = np.empty((a.rows, a.dim), dtype=np.float32)
    for lo in range(0, a.rows, 100000):
        hi = min(a.rows, lo + 100000)
        pick = torch.randint(0, a.lists, (hi - lo,), device=dev, generator=g)
        rows[lo:hi] = (comps[pick] + 0.1 * torch.randn((hi - lo, a.dim), device=dev, generator=g)).cpu().numpy()
    tids = (np.arange(a.rows, dtype=np.uint64) << np.uint64(16)) | np.uint64(1)
    rng = np.random.default_rng(1)
    samples = rows[np.sort(rng.choice(a.rows, min(max(50 * a.lists, 10000), a.rows), replace=False))]
    ctx = api.Context(0, stream=0)
    for rep in range(a.reps):
        rel = _host.Relation()
        t0 = time.perf_counter()
        pix = rel.build_mirror(ctx, api.PGV_OPS_L2, api.PGV_F32, a.lists, rows, tids, samples, api.make_rng(seed=2))
        ctx.sync()
        secs = time.perf_counter() - t0
        ph = (ctypes.c_double * 5)()
        _host.lib.pgv_host_ivf_build_phases(ph)
        _host.lib.pgv_host_ivf_writer_wait_secs.restype = ctypes.c_double
        print(json.dumps({"rep": rep, "build_secs_pages": secs, "pages": int(rel.nblocks),
                          "page_writer_waited_for_zeroed_pages": round(_host.lib.pgv_host_ivf_writer_wait_secs(), 4),
                          "phases": dict(zip(("normalise", "kmeans_left_after_upload", "upload_beside_kmeans", "assign_and_order_by_list_on_device",
                                              "page_writer"), [round(float(x), 4) for x in ph]))}), flush=True)
        if rep == a.reps - 1:
            # the drain alone (pinned bounce buffers + D2H, a sink that does nothing), by piece size
            from pgvector_amd.api import _SINK
            from pgvector_amd._lib import check
            cb = _SINK(lambda _a, first, count, v, t: 0)
            for chunk in (0, 1 << 14, 1 << 12, 1 << 10):
                for _ in range(2):
                    t0 = time.perf_counter()
                    check(api.lib.pgv_index_drain(pix.h, chunk, cb, None))
                    secs = time.perf_counter() - t0
                print(json.dumps({"drain_only_chunk_rows": chunk or "default (64 MB)", "secs": round(secs, 4),
                                  "GBps": round(a.rows * a.dim * 4 / secs / 1e9, 1)}), flush=True)
        pix.close()
        del rel
    ctx.close()


if __name__ == "__main__":
    main()
