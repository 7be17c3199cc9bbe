#!/usr/bin/env python3
"""tools/tune_hard.py -- where does bench.py's mid-difficulty data set (gen_hard) put recall@10 at probes 10?
One row per (sigma, alpha, zipf): k-means iterations, list imbalance, recall at probes 1 / 10 / 32 / 100."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0]] + sys.argv[1:]
import bench  # noqa: E402
from pgvector_amd import api  # noqa: E402


def main():
    grid = [tuple(float(x) for x in a.split(",")) for a in sys.argv[1:]] or [(0.4, 1.0, 0.8), (0.55, 1.0, 0.8), (0.7, 1.0, 0.8), (1.0, 1.0, 0.8)]
    n, dim, lists, k = 1_000_000, 1536, 1000, 10
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    ctx = api.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    for sigma, alpha, zipf in grid:
        bench.HARD.update(sigma=sigma, alpha=alpha, zipf=zipf)
        data, means = bench.gen_hard(n, dim, lists // 4, 50, dev)
        ctx.set_profiling(True)
        ctx.reset_stats()
        t0 = time.perf_counter()
        centers, offsets, vectors, tids, iters, bt, index = bench.build_index(ctx, data, lists, 0, 1, 0, api.PGV_F32, api.PGV_OPS_L2, api.PGV_L2SQ)
        st = ctx.stats()
        ctx.set_profiling(False)
        del data
        q, _ = bench.gen_hard(256, dim, lists // 4, 150, dev, means=means)
        ed, _ = bench.exact_topk_fp64(vectors, q, k, api.PGV_L2SQ)
        sizes = (offsets[1:] - offsets[:-1]).double()
        row = {"sigma": sigma, "alpha": alpha, "zipf": zipf, "kmeans_iterations": int(iters), "build_secs": round(bt["total"], 3),
               "kmeans_secs": round(bt["kmeans"], 3), "lists_max_over_mean": round(float(sizes.max() / sizes.mean()), 2), "recall": {},
               "phases": {a: round(b, 3) for a, b in bt.items()},
               "assign_rechecked": round(st["assign_recheck_rows"] / max(st["assign_rows"], 1), 4),
               "assign_redone": round(st["assign_redo_rows"] / max(st["assign_rows"], 1), 5)}
        for p in (1, 10, 32, 100):
            gd, _, _ = index.search_batch(q, p, k)
            row["recall"][p] = round(bench.recall_at_k(gd, ed, k), 4)
        print(json.dumps(row), flush=True)
        index.close()
        del vectors, centers
        torch.cuda.empty_cache()
    ctx.close()


if __name__ == "__main__":
    main()
