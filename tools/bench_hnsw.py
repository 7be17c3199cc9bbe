#!/usr/bin/env python3
"""HNSW candidate-batch scoring (BASELINE configs[3], scaled): vector_cosine_ops, m = 16,
ef_search = 100, 1536-d unit vectors.  The graph is built by the CPU oracle's restatement of
the reference build (it is test infrastructure: the product path only SEARCHES).  Two searches
are timed: pgv_hnsw_search, the whole first batch of a scan in one kernel launch (one workgroup
per query), and the C host glue's lock-step HnswSearchLayer with every expansion step's
candidates scored by one pgv_hnsw_score launch.  Prints one JSON line.

usage: python tools/bench_hnsw.py [--rows 50000] [--dim 1536] [--queries 1000]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from pgvector_amd import _host, api  # noqa: E402
from oracle import pyoracle as po  # noqa: E402


def big(a):
    """BASELINE configs[3] at full size without the CPU oracle: data generated on the device (harness),
    graph built by pgv_host_hnsw_build, searched by pgv_hnsw_search, recall against an exact fp64
    scan done with torch on the device (harness)."""
    import torch
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    comps = torch.rand((64, a.dim), generator=g, device=dev)
    pick = torch.randint(0, 64, (a.rows,), generator=g, device=dev)
    data = comps[pick] + 0.1 * torch.randn((a.rows, a.dim), generator=g, device=dev)
    data = (data / data.norm(dim=1, keepdim=True)).contiguous()
    qpick = torch.randint(0, 64, (a.queries,), generator=g, device=dev)
    q = comps[qpick] + 0.1 * torch.randn((a.queries, a.dim), generator=g, device=dev)
    q = (q / q.norm(dim=1, keepdim=True)).contiguous()
    host_rows = data.cpu().numpy()
    ctx = api.Context(0, stream=0)
    mirror = api.Hnsw(ctx, api.PGV_NEG_IP, api.PGV_F32, a.dim, data)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    built = _host.hnsw_build(mirror, host_rows, a.m, a.ef_construction, api.make_rng(seed=1), max_batch=a.gpu_build or 1024)
    torch.cuda.synchronize()
    build_s = time.perf_counter() - t0
    reps = max(1, 20000 // a.queries)
    qd = q.repeat(reps, 1).contiguous()
    mirror.search(qd[:64].contiguous(), a.ef_search, a.k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    elem, dist, scored = mirror.search(qd, a.ef_search, a.k)
    torch.cuda.synchronize()
    dev_s = time.perf_counter() - t0
    # exact top-k: fp64 inner products in row chunks
    best = torch.full((a.queries, a.k), -2.0, dtype=torch.float64, device=dev)
    q64 = q.double()
    for lo in range(0, a.rows, 100000):
        ip = q64 @ data[lo:lo + 100000].double().T
        best = torch.topk(torch.cat([best, ip], dim=1), a.k, dim=1).values
    kth = best[:, -1]
    e = elem[:a.queries]
    got_ip = (q64[:, None, :] * data[e.clamp(min=0)].double()).sum(-1)
    hits = ((got_ip >= kth[:, None] - 1e-9) & (e >= 0)).sum().item()
    # ef_search sweep on the same graph (hnsw.ef_search is the user's recall knob, src/hnsw.c:74-77)
    sweep = {}
    for ef in (40, 100, 200, 400, 800):
        if ef < a.k:
            continue
        mirror.search(qd[:64].contiguous(), ef, a.k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e2, _, sc2 = mirror.search(qd, ef, a.k)
        torch.cuda.synchronize()
        s2 = time.perf_counter() - t0
        e2 = e2[:a.queries]
        gip = (q64[:, None, :] * data[e2.clamp(min=0)].double()).sum(-1)
        h2 = ((gip >= kth[:, None] - 1e-9) & (e2 >= 0)).sum().item()
        sweep[str(ef)] = {"qps": qd.shape[0] / s2, "recall_at_k": h2 / (a.queries * a.k),
                          "scored_elements_per_query": float(sc2.float().mean().item())}
    print(json.dumps({
        "ef_search_sweep": sweep,
        "metric": "HNSW QPS (pgv_hnsw_search, graph built by pgv_host_hnsw_build)", "value": qd.shape[0] / dev_s,
        "unit": "queries/s", "config": {"rows": a.rows, "dim": a.dim, "m": a.m, "ef_construction": a.ef_construction,
                                        "ef_search": a.ef_search, "k": a.k, "queries_in_flight": int(qd.shape[0]),
                                        "ops": "vector_cosine_ops"},
        "recall_at_k": hits / (a.queries * a.k), "scored_elements_per_query": float(scored.float().mean().item()),
        "algorithmic_GBps": float(scored.sum().item()) * a.dim * 4 / dev_s / 1e9,
        "gpu_build": {"secs": build_s, "max_batch": a.gpu_build or 1024, "batches": built["batches"],
                      "elements": built["nelements"], "pairs_scored": built["device_pairs"],
                      "deferred_updates": built["deferred_updates"], "phase_secs": built["phase_secs"]}}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=50000)
    ap.add_argument("--dim", type=int, default=1536)
    ap.add_argument("--queries", type=int, default=1000)
    ap.add_argument("--m", type=int, default=16)
    ap.add_argument("--ef-construction", type=int, default=64)
    ap.add_argument("--ef-search", type=int, default=100)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--graph", default="", help="npz file to load the graph from / save it to")
    ap.add_argument("--build-only", action="store_true")
    ap.add_argument("--skip-host-search", action="store_true")
    ap.add_argument("--gpu-build", type=int, default=0, help="also build the graph on the GPU with this max_batch")
    ap.add_argument("--big", action="store_true", help="full-size run without the CPU oracle (see big())")
    a = ap.parse_args()
    if a.big:
        return big(a)
    rng = np.random.default_rng(0)
    comps = rng.random((64, a.dim), dtype=np.float32)
    data = comps[rng.integers(0, 64, a.rows)] + 0.1 * rng.standard_normal((a.rows, a.dim)).astype(np.float32)
    queries = comps[rng.integers(0, 64, a.queries)] + 0.1 * rng.standard_normal((a.queries, a.dim)).astype(np.float32)
    ora = po.Oracle(native=True)
    g = None
    if a.graph and os.path.exists(a.graph):
        # a graph built earlier by this same script (--graph FILE --build-only) from the same seed
        z = np.load(a.graph)
        ex = {key: z[key] for key in ("rows", "levels", "nbr_start", "nbr")}
        ex["entry"] = int(z["entry"])
        build_s = float(z["build_s"])
        assert int(z["rows_n"]) == a.rows and int(z["dim"]) == a.dim and int(z["m"]) == a.m
    else:
        t0 = time.perf_counter()
        g = po.HnswGraph(ora, po.OPS_COSINE, po.ORA_F32, data, m=a.m, ef_construction=a.ef_construction, seed=1)
        build_s = time.perf_counter() - t0
        ex = g.export_tuples()
        if a.graph:
            np.savez(a.graph, rows=ex["rows"], levels=ex["levels"], nbr_start=ex["nbr_start"], nbr=ex["nbr"],
                     entry=ex["entry"], build_s=build_s, rows_n=a.rows, dim=a.dim, m=a.m)
    if a.build_only:
        print("graph built in %.1f s" % build_s)
        return
    unit = data / np.linalg.norm(data.astype(np.float64), axis=1, keepdims=True)
    stored = np.ascontiguousarray(unit[ex["rows"]].astype(np.float32))
    qn = np.ascontiguousarray((queries / np.linalg.norm(queries.astype(np.float64), axis=1, keepdims=True)).astype(np.float32))

    ctx = api.Context(0, stream=0)  # the library runs on torch's default stream: its outputs are ordered with torch
    mirror = api.Hnsw(ctx, api.PGV_NEG_IP, api.PGV_F32, a.dim, stored)
    graph = _host.hnsw_graph(ex["levels"], ex["nbr_start"], ex["nbr"], a.m, ex["entry"])
    _host.hnsw_search(mirror, graph, qn[:8], a.ef_search, a.k)  # warm up
    t0 = time.perf_counter()
    elem, dist, scored = _host.hnsw_search(mirror, graph, qn, a.ef_search, a.k)
    gpu_s = time.perf_counter() - t0

    # the on-device search: queries resident in HBM, timed with a device sync on both sides
    import torch
    mirror.set_graph(a.m, ex["entry"], ex["levels"], ex["nbr_start"], ex["nbr"])
    qd = torch.from_numpy(qn).cuda()
    reps = max(1, 20000 // a.queries)
    qd = qd.repeat(reps, 1).contiguous()
    mirror.search(qd[:64], a.ef_search, a.k)
    ctx.sync()
    t0 = time.perf_counter()
    delem, ddist, dscored = mirror.search(qd, a.ef_search, a.k)
    ctx.sync()
    dev_s = time.perf_counter() - t0
    delem_h = delem[:a.queries].cpu().numpy()
    same = float((np.sort(delem_h, axis=1) == np.sort(elem, axis=1)).all(axis=1).mean())

    # CREATE INDEX on the GPU: the same rows through pgv_host_hnsw_build, searched the same way
    gpu_build = None
    if a.gpu_build:
        m2 = api.Hnsw(ctx, api.PGV_NEG_IP, api.PGV_F32, a.dim, stored)
        t0 = time.perf_counter()
        built = _host.hnsw_build(m2, stored, a.m, a.ef_construction, api.make_rng(seed=1), max_batch=a.gpu_build)
        gb_s = time.perf_counter() - t0
        belem, _, bscored = m2.search(qd[:a.queries].contiguous(), a.ef_search, a.k)
        gpu_build = {"secs": gb_s, "max_batch": a.gpu_build, "batches": built["batches"],
                     "elements": built["nelements"], "pairs_scored": built["device_pairs"],
                     "elem": belem.cpu().numpy(), "scored": float(bscored.float().mean().item())}
        m2.close()

    # exact ground truth on the same unit vectors (cosine distance order = -ip order)
    ip = qn.astype(np.float64) @ stored.astype(np.float64).T
    kth = -np.sort(-ip, axis=1)[:, a.k - 1]
    hits = 0
    for i in range(a.queries):
        e = elem[i][elem[i] >= 0]
        hits += int((ip[i, e] >= kth[i] - 1e-9).sum())
    recall = hits / (a.queries * a.k)
    dhits = 0
    for i in range(a.queries):
        e = delem_h[i][delem_h[i] >= 0]
        dhits += int((ip[i, e] >= kth[i] - 1e-9).sum())
    if gpu_build:
        bh = 0
        be = gpu_build.pop("elem")
        for i in range(a.queries):
            e = be[i][be[i] >= 0]
            bh += int((ip[i, e] >= kth[i] - 1e-9).sum())
        gpu_build["recall_at_k_of_its_graph"] = bh / (a.queries * a.k)
        gpu_build["scored_elements_per_query"] = gpu_build.pop("scored")

    # the oracle's search, one thread (one backend)
    t0 = time.perf_counter()
    n_cpu = min(a.queries, 200) if g is not None else 0
    cpu_scored = 0
    for i in range(n_cpu):
        _, _, sc = g.search(queries[i], a.ef_search, a.k)
        cpu_scored += sc
    cpu_s = time.perf_counter() - t0
    print(json.dumps({
        "metric": "HNSW QPS (candidate-batch scoring on GPU, graph walk on host)", "value": a.queries / gpu_s,
        "unit": "queries/s", "config": {"rows": a.rows, "dim": a.dim, "m": a.m, "ef_construction": a.ef_construction,
                                        "ef_search": a.ef_search, "k": a.k, "queries_in_lock_step": a.queries,
                                        "ops": "vector_cosine_ops"},
        "device_search": {"value": qd.shape[0] / dev_s, "unit": "queries/s", "queries_in_flight": int(qd.shape[0]),
                          "recall_at_k": dhits / (a.queries * a.k),
                          "scored_elements_per_query": float(dscored.float().mean().item()),
                          "algorithmic_GBps": float(dscored.sum().item()) * a.dim * 4 / dev_s / 1e9,
                          "same_result_set_as_host_search": same},
        "recall_at_k": recall, "scored_elements_per_query": float(scored.mean()),
        "algorithmic_GBps": float(scored.sum()) * a.dim * 4 / gpu_s / 1e9,
        "cpu_baseline": ({"value": n_cpu / cpu_s, "unit": "queries/s", "cores": 1, "kind": "port",
                          "sample": "%d queries, oracle HnswSearchLayer on one thread" % n_cpu,
                          "scored_elements_per_query": cpu_scored / n_cpu} if n_cpu else None),
        "gpu_build": gpu_build, "graph_build_secs_cpu_oracle": build_s, "elements": int(len(ex["levels"]))}))


if __name__ == "__main__":
    main()
