#!/usr/bin/env python3
"""HNSW graph quality at scale (round 3): is the GPU build's lock-step batching free, and how does a GPU-built graph
compare with the graph the reference's serial algorithm builds on the same rows?

Two steps, because the serial CPU build of 500 k x 1536 takes a quarter of an hour on one core:

  python tools/hnsw_quality.py --oracle-build 500000 [--threads 8] [--oracle-recall]       (CPU only, anywhere)
      builds the graph with the oracle's restatement of the reference's in-memory build
      (src/hnswbuild.c:376-476; serial, or with --threads the parallel build with per-element locks) and saves it to build/hnsw_oracle_<rows>.npz (git-ignored; travels with gpurun)
  python tools/hnsw_quality.py --rows 500000 --batches 16,256,1024 --small-rows 50000   (on the GPU)
      same rows (numpy generator, same seed): (a) pgv_host_hnsw_build with max_batch 16 / 256 / 1024 and, on the first
      small-rows rows, also 1 (one insert at a time: the serial algorithm on the device); (b) every graph -- the
      oracle's included, uploaded with pgv_hnsw_set_graph -- searched by pgv_hnsw_search at ef_search 40 / 100 / 200,
      recall@10 against exact float64 inner products.  Prints one JSON object (kept as profiles/r03_hnsw_quality.json).

vector_cosine_ops, m 16, ef_construction 64: BASELINE configs[3]'s shape."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def make_rows(rows, dim, nq, seed=0):
    """unit vectors of a 64-component mixture (sigma 0.1) -- generated in slabs so that a prefix of the rows is the
    same whatever the total"""
    rng = np.random.default_rng(seed)
    comps = rng.random((64, dim), dtype=np.float32)
    data = np.empty((rows, dim), dtype=np.float32)
    for lo in range(0, rows, 50000):
        hi = min(rows, lo + 50000)
        g = np.random.default_rng(seed * 7919 + 1 + lo // 50000)
        data[lo:hi] = comps[g.integers(0, 64, hi - lo)] + 0.1 * g.standard_normal((hi - lo, dim)).astype(np.float32)
    data /= np.linalg.norm(data, axis=1, keepdims=True)
    g = np.random.default_rng(seed + 12345)
    q = comps[g.integers(0, 64, nq)] + 0.1 * g.standard_normal((nq, dim)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    return np.ascontiguousarray(data), np.ascontiguousarray(q.astype(np.float32))


def oracle_path(rows):
    return os.path.join(ROOT, "build", "hnsw_oracle_%d.npz" % rows)


def oracle_build(a):
    from oracle import pyoracle as po
    ora = po.Oracle(native=True)
    data, _ = make_rows(a.oracle_build, a.dim, 8)
    t0 = time.perf_counter()
    g = po.HnswGraph(ora, po.OPS_COSINE, po.ORA_F32, data, m=a.m, ef_construction=a.ef_construction, seed=1,
                     threads=a.threads)
    secs = time.perf_counter() - t0
    ex = g.export_tuples()
    os.makedirs(os.path.dirname(oracle_path(a.oracle_build)), exist_ok=True)
    np.savez_compressed(oracle_path(a.oracle_build), rows=ex["rows"].astype(np.int32), levels=ex["levels"].astype(np.int8),
                        nbr_start=ex["nbr_start"], nbr=ex["nbr"], entry=ex["entry"], build_secs=secs, n=a.oracle_build,
                        dim=a.dim, m=a.m, ef_construction=a.ef_construction, threads=a.threads)
    print("oracle graph of %d rows built in %.1f s (%s) -> %s" % (
        a.oracle_build, secs, "serial" if a.threads == 0 else "%d inserter threads, per-element locks" % a.threads,
        oracle_path(a.oracle_build)))
    if a.oracle_recall:
        # the oracle's own search over its graph against exact float64 inner products (CPU only): a record of the
        # graph's quality that needs no GPU
        _, q = make_rows(a.oracle_build, a.dim, a.queries)
        q64 = q.astype(np.float64)
        best = np.full((a.queries, a.k), -2.0)
        for lo in range(0, a.oracle_build, 50000):
            ip = q64 @ data[lo:lo + 50000].astype(np.float64).T
            best = -np.sort(-np.concatenate([best, ip], axis=1), axis=1)[:, :a.k]
        kth = best[:, -1]
        rec = {}
        for ef in (40, 100, 200):
            hits = 0
            for i in range(a.queries):
                rows, _, _ = g.search(q[i], ef, a.k)
                hits += int(((data[rows].astype(np.float64) @ q64[i]) >= kth[i] - 1e-9).sum())
            rec[str(ef)] = hits / (a.queries * a.k)
        print(json.dumps({"oracle_rows": a.oracle_build, "threads": a.threads, "build_secs": secs, "recall_at_10": rec}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--oracle-build", type=int, default=0)
    ap.add_argument("--threads", type=int, default=0, help="oracle build: 0 = serial, N = the reference's parallel "
                    "in-memory build restated with N inserter threads (ora_hnsw_build_parallel)")
    ap.add_argument("--oracle-recall", action="store_true", help="oracle build: also search the graph with the oracle "
                    "and print recall@k at ef_search 40 / 100 / 200 (CPU only)")
    ap.add_argument("--rows", type=int, default=500000)
    ap.add_argument("--small-rows", type=int, default=50000)
    ap.add_argument("--dim", type=int, default=1536)
    ap.add_argument("--m", type=int, default=16)
    ap.add_argument("--ef-construction", type=int, default=64)
    ap.add_argument("--queries", type=int, default=1000)
    ap.add_argument("--batches", default="16,256,1024")
    ap.add_argument("--k", type=int, default=10)
    a = ap.parse_args()
    if a.oracle_build:
        return oracle_build(a)
    import torch

    from pgvector_amd import _host, api
    dev = torch.device("cuda:0")
    data, q = make_rows(a.rows, a.dim, a.queries)
    ctx = api.Context(0, stream=0)
    dd = torch.from_numpy(data).to(dev)
    qd = torch.from_numpy(q).to(dev)
    out = {"rows": a.rows, "dim": a.dim, "m": a.m, "ef_construction": a.ef_construction, "k": a.k, "queries": a.queries,
           "data": "unit vectors of a 64-component Gaussian mixture, sigma 0.1", "graphs": {}}

    def exact_kth(n):
        q64 = qd.double()
        best = torch.full((a.queries, a.k), -2.0, dtype=torch.float64, device=dev)
        for lo in range(0, n, 100000):
            best = torch.topk(torch.cat([best, q64 @ dd[lo:min(n, lo + 100000)].double().T], dim=1), a.k, dim=1).values
        return best[:, -1]

    def evaluate(mirror, rows_of_elem, kth, n):
        """recall@k and throughput at three ef_search values; rows_of_elem maps element slots to data rows"""
        res = {}
        q64 = qd.double()
        reps = qd.repeat(max(1, 10000 // a.queries), 1).contiguous()
        for ef in (40, 100, 200):
            mirror.search(reps[:64].contiguous(), ef, a.k)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            elem, _, scored = mirror.search(reps, ef, a.k)
            torch.cuda.synchronize()
            s = time.perf_counter() - t0
            e = elem[:a.queries]
            rows = rows_of_elem[e.clamp(min=0)] if rows_of_elem is not None else e.clamp(min=0)
            ip = (q64[:, None, :] * dd[rows].double()).sum(-1)
            hits = ((ip >= kth[:, None] - 1e-9) & (e >= 0)).sum().item()
            res[str(ef)] = {"recall_at_10": hits / (a.queries * a.k), "qps": reps.shape[0] / s,
                            "scored_elements_per_query": float(scored.float().mean().item())}
        return res

    for n, batches in ((a.small_rows, [1] + [int(b) for b in a.batches.split(",")]), (a.rows, [int(b) for b in a.batches.split(",")])):
        if n <= 0 or n > a.rows:
            continue
        kth = exact_kth(n)
        for mb in batches:
            mirror = api.Hnsw(ctx, api.PGV_NEG_IP, api.PGV_F32, a.dim, dd[:n].contiguous())
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            built = _host.hnsw_build(mirror, data[:n], a.m, a.ef_construction, api.make_rng(seed=1), max_batch=mb)
            torch.cuda.synchronize()
            secs = time.perf_counter() - t0
            keep = torch.from_numpy(np.nonzero(built["dup_of"] < 0)[0]).to(dev)  # element slot -> data row
            rec = {"built_by": "pgv_host_hnsw_build, max_batch %d" % mb, "rows": n, "build_secs": secs,
                   "elements": int(built["nelements"]), "ef_search": evaluate(mirror, keep, kth, n)}
            out["graphs"]["gpu_rows%d_batch%d" % (n, mb)] = rec
            print(json.dumps({"gpu_rows%d_batch%d" % (n, mb): rec}), file=sys.stderr, flush=True)
            mirror.close()
        path = oracle_path(n)
        if os.path.exists(path):
            z = np.load(path)
            assert int(z["n"]) == n and int(z["dim"]) == a.dim and int(z["m"]) == a.m
            rows_of = z["rows"].astype(np.int64)
            mirror = api.Hnsw(ctx, api.PGV_NEG_IP, api.PGV_F32, a.dim, dd[:n][torch.from_numpy(rows_of).to(dev)].contiguous())
            mirror.set_graph(a.m, int(z["entry"]), z["levels"].astype(np.int32), z["nbr_start"], z["nbr"])
            rec = {"built_by": "the oracle's serial restatement of the reference build (CPU, %.0f s on one core)" % float(z["build_secs"]),
                   "rows": n, "elements": int(len(rows_of)),
                   "ef_search": evaluate(mirror, torch.from_numpy(rows_of).to(dev), kth, n)}
            out["graphs"]["oracle_rows%d" % n] = rec
            print(json.dumps({"oracle_rows%d" % n: rec}), file=sys.stderr, flush=True)
            mirror.close()
    print(json.dumps(out))
    ctx.close()


if __name__ == "__main__":
    main()
