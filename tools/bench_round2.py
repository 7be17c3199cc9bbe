#!/usr/bin/env python3
"""Round-2 micro-benchmarks on one MI355X: the MFMA assignment (AddTupleToSort's argmin,
src/ivfbuild.c:183-192) and the fused single-query path (ivfflatgettuple, src/ivfscan.c:361-414).
Prints one JSON object; `--what assign,query` selects."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pgvector_amd import api  # noqa: E402


def timed(ctx, fn, reps=3):
    fn()
    ctx.sync()
    best = 1e30
    for _ in range(reps):
        ctx.timer_start()
        fn()
        best = min(best, ctx.timer_stop())
    return best


CASES = []


def bench_assign(ctx, out):
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    res = []
    for name, n, k, dim, tdt, dt, metric in [
        ("headline L2 fp32", 1_000_000, 1000, 1536, torch.float32, api.PGV_F32, api.PGV_L2SQ),
        ("headline IP fp32", 1_000_000, 1000, 1536, torch.float32, api.PGV_F32, api.PGV_NEG_IP),
        ("c2 L2 fp32", 1_000_000, 1000, 768, torch.float32, api.PGV_F32, api.PGV_L2SQ),
        ("c5 share L2 fp16", 1_250_000, 4096, 3072, torch.float16, api.PGV_F16, api.PGV_L2SQ),
        ("c5 share IP fp16", 1_250_000, 4096, 3072, torch.float16, api.PGV_F16, api.PGV_NEG_IP),
        ("c3 share IP fp32", 1_250_000, 4096, 1536, torch.float32, api.PGV_F32, api.PGV_NEG_IP),
        ("lloyd 50k L2 fp32", 50_000, 1000, 1536, torch.float32, api.PGV_F32, api.PGV_L2SQ),
        ("uniform L2 fp32 (near ties)", 200_000, 1000, 768, torch.float32, api.PGV_F32, api.PGV_L2SQ),
    ]:
        if CASES and not any(c in name for c in CASES):
            continue
        means = torch.rand((max(k // 4, 1), dim), generator=g, device=dev)
        if "uniform" in name:
            rows = torch.rand((n, dim), generator=g, device=dev).to(tdt)
            centers = torch.rand((k, dim), generator=g, device=dev).to(tdt)
        else:
            rows = torch.empty((n, dim), device=dev, dtype=tdt)
            for lo in range(0, n, 1 << 17):
                hi = min(n, lo + (1 << 17))
                comp = torch.randint(0, means.shape[0], (hi - lo,), generator=g, device=dev)
                rows[lo:hi] = (means[comp] + 0.1 * torch.randn((hi - lo, dim), generator=g, device=dev)).to(tdt)
            centers = rows[torch.randperm(n, generator=g, device=dev)[:k]].contiguous()
        lists = torch.empty(n, dtype=torch.int32, device=dev)
        dist = torch.empty(n, dtype=torch.float32, device=dev)

        def run():
            from pgvector_amd._lib import check, lib
            check(lib.pgv_assign(ctx.h, metric, dt, dim, api.ptr(centers), k, api.ptr(rows), n, api.ptr(lists),
                                 api.ptr(dist)))
        ctx.set_profiling(True)
        ctx.reset_stats()
        ms = timed(ctx, run)
        st = ctx.stats()
        ctx.set_profiling(False)
        macs = float(n) * k * dim
        # spot check against torch on a slab
        sl = slice(0, 4096)
        r32, c32 = rows[sl].float(), centers.float()
        if metric == api.PGV_L2SQ:
            ref = torch.cdist(r32.double(), c32.double()).pow(2)
        else:
            ref = -(r32.double() @ c32.double().T)
        rv, ri = ref.min(dim=1)
        got = lists[sl].long()
        gv = ref.gather(1, got[:, None])[:, 0]
        bad = int(((gv - rv).abs() > 1e-5 * (rv.abs() + (1.0 if metric == api.PGV_L2SQ else float(dim)))).sum().item())
        res.append({"case": name, "n": n, "k": k, "dim": dim, "ms": ms, "tmac_per_s": macs / ms / 1e9,
                    "tflops_2flop": 2 * macs / ms / 1e9, "mismatch_vs_fp64_4096": bad,
                    "redo_fraction": st["assign_redo_rows"] / st["assign_rows"] if st["assign_rows"] else None,
                    "recheck_fraction": st["assign_recheck_rows"] / st["assign_rows"] if st["assign_rows"] else None})
        print(res[-1], file=sys.stderr, flush=True)
        del rows, centers, lists, dist
    out["assign"] = res


def bench_query(ctx, out):
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(2)
    n, dim, lists, probes = 1_000_000, 1536, 1000, 10
    means = torch.rand((250, dim), generator=g, device=dev)
    data = torch.empty((n, dim), device=dev)
    for lo in range(0, n, 1 << 17):
        hi = min(n, lo + (1 << 17))
        comp = torch.randint(0, 250, (hi - lo,), generator=g, device=dev)
        data[lo:hi] = means[comp] + 0.1 * torch.randn((hi - lo, dim), generator=g, device=dev)
    centers, _, _ = api.kmeans(ctx, api.PGV_OPS_L2, api.PGV_F32, dim,
                               data[torch.randperm(n, generator=g, device=dev)[:50000]].contiguous(), lists,
                               api.make_rng(seed=3), want_closest=False)
    assign, _ = api.assign(ctx, api.PGV_L2SQ, api.PGV_F32, dim, centers, data, want_dist=False)
    order = torch.argsort(assign.long(), stable=True)
    off = torch.zeros(lists + 1, dtype=torch.int64, device=dev)
    off[1:] = torch.cumsum(torch.bincount(assign.long(), minlength=lists), 0)
    vectors = data[order].contiguous()
    del data
    ix = api.IvfIndex(ctx, api.PGV_L2SQ, api.PGV_F32, dim, centers, off, vectors, order)
    comp = torch.randint(0, 250, (512,), generator=g, device=dev)
    queries = (means[comp] + 0.1 * torch.randn((512, dim), generator=g, device=dev)).cpu().numpy()
    qh = api.Query(ix)
    res = {}
    for head in (10, 64):
        for i in range(20):
            qh.rank(queries[i], probes)
            qh.scan(0, probes, head)
        lat = []
        for i in range(400):
            q = queries[i % 512]
            t0 = time.perf_counter()
            qh.rank(q, probes)
            d, s, t, total = qh.scan(0, probes, head)
            lat.append(time.perf_counter() - t0)
        lat = np.array(lat) * 1e6
        res["rank+scan head=%d" % head] = {"p50_us": float(np.percentile(lat, 50)), "p90_us": float(np.percentile(lat, 90)),
                                            "mean_us": float(lat.mean()), "rows_per_query": int(total)}
    # parity of the fused path with the batched path on the same queries
    bd, bs, bt = ix.search_batch(queries[:64], probes, 10, want_tid=True)
    same = 0
    for i in range(64):
        qh.rank(queries[i], probes)
        d, s, t, _ = qh.scan(0, probes, 10)
        same += int(np.array_equal(s, bs[i]))
    res["fused_equals_batched_of_64"] = same
    # the old two-call path for comparison
    lat = []
    for i in range(100):
        q = queries[i % 512]
        t0 = time.perf_counter()
        l, _ = ix.rank_lists(q[None, :], probes, want_dist=False)
        ix.scan_lists(q, l[0])
        lat.append(time.perf_counter() - t0)
    res["legacy rank_lists+scan_lists (+host sort not included)"] = {"p50_us": float(np.percentile(np.array(lat) * 1e6, 50))}
    # GPU time of the two kernels alone
    def both():
        qh.rank(queries[0], probes)
        qh.scan(0, probes, 10)
    res["gpu_ms_rank_plus_scan"] = timed(ctx, both, reps=5)
    # through the C host glue: ivfflatgettuple for LIMIT 10
    from pgvector_amd import _host
    img = _host.IvfImage()
    cen = centers.cpu().numpy()
    offh = off.cpu().numpy()
    tids = order.cpu().numpy().astype(np.uint64)
    img.dtype, img.dim, img.lists, img.nrows = api.PGV_F32, dim, lists, n
    img.centers, img.list_offsets, img.tids = cen.ctypes.data, offh.ctypes.data, tids.ctypes.data
    img.vectors = None

    class Staged:
        pass
    st = Staged()
    st.img, st.dtype = img, api.PGV_F32
    scan = _host.IvfScan(ix, st, probes=probes)
    for i in range(20):
        scan.rescan(queries[i])
        scan.fetch(limit=10)
    lat = []
    for i in range(300):
        t0 = time.perf_counter()
        scan.rescan(queries[i % 512])
        scan.fetch(limit=10)
        lat.append(time.perf_counter() - t0)
    lat = np.array(lat) * 1e6
    res["pgv_host_ivf_gettuple x10 (python ctypes loop)"] = {"p50_us": float(np.percentile(lat, 50)),
                                                               "p90_us": float(np.percentile(lat, 90))}
    scan.close()
    qh.close()
    ix.close()
    out["query"] = res
    print(res, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--what", default="assign,query")
    ap.add_argument("--cases", default="", help="comma-separated substrings of the assign cases to run (default: all)")
    ap.add_argument("--bound", default="default", choices=("default", "statistical", "worst"),
                    help="pgv_ctx_set_bound before the runs (default: whatever the library starts with)")
    args = ap.parse_args()
    CASES.extend(c for c in args.cases.split(",") if c)
    ctx = api.Context(0, stream=0)
    if args.bound != "default":
        ctx.set_bound(args.bound == "worst")
    out = {"bound": args.bound}
    for w in args.what.split(","):
        {"assign": bench_assign, "query": bench_query}[w](ctx, out)
    print(json.dumps(out))
    ctx.close()


if __name__ == "__main__":
    main()
