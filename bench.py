#!/usr/bin/env python3
"""bench.py -- IVFFlat QPS @ recall@10 (+ index build seconds) on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it
is launched through torch.distributed.run with one rank per GPU.  A step is one
batch of queries through the whole hot path (GetScanLists + GetScanItems + the
head of the sorted stream) with everything already resident in HBM.  Rank 0
prints ONE JSON line.

Workload (config.workload): BASELINE.json's headline -- 1,000,000 x 1536-d
fp32, vector_l2_ops, lists = 1000, probes = 10, k = 10, synthetic Gaussian
mixture (250 components, sigma 0.1) so that recall is meaningful.  With N GPUs
the lists are sharded l % N, the query batch grows to N x batch ("weak": the
per-GPU scan work per step is fixed), per-rank top-k are merged with one
all-gather.

Extra objects on the line: `roofline` (the list-scan kernel, timed with HIP
events on its own stream inside the timed region) and `cpu_baseline` (the
oracle's restatement of ivfflatgettuple compiled with the reference's flags and
-march=native, timed on the host cores of this box on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import pgvector_amd  # noqa: E402
from pgvector_amd import api, sharding  # noqa: E402

WORKLOADS = {
    # name: rows, dim, lists, probes, element type, opclass
    "headline": (1_000_000, 1536, 1000, 10, "f32", "l2"),   # BASELINE.json metric: IVFFlat 1M x 1536d
    "c2": (1_000_000, 768, 1000, 10, "f32", "l2"),          # configs[1]
    # one GPU's share of configs[2] (10M x 1536 fp32 vector_ip_ops, lists 4096, 8 GPUs, probes 64)
    "c3shard": (1_250_000, 1536, 512, 8, "f32", "ip"),
    # one GPU's share of configs[4] (10M x 3072 fp16 halfvec_l2_ops, lists 4096, 8 GPUs, probes 64)
    "c5shard": (1_250_000, 3072, 512, 8, "f16", "l2"),
    "small": (100_000, 256, 100, 10, "f32", "l2"),          # quick functional run
    "smallh": (100_000, 512, 100, 10, "f16", "ip"),
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
SCAN_KERNEL_TAG = "tile_v3"  # which kernel build the PMC traffic entries in profiles/traffic.json belong to


def pmc_traffic(workload, batch):
    """HBM bytes per list-scan launch from the committed rocprofv3 PMC passes (separate runs,
    MI355X_MICROARCH.md HBM section); None when no pass matches this kernel/workload."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        for e in t["entries"]:
            if e["kernel"] == SCAN_KERNEL_TAG and e["workload"] == workload and e["batch"] == batch:
                return e["traffic_bytes_per_launch"]
    except Exception:
        pass
    return None


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print(*a, file=sys.stderr, flush=True)


def dbg(*a):
    if os.environ.get("PGV_BENCH_DEBUG"):
        print("[rank %s]" % os.environ.get("RANK", "0"), *a, file=sys.stderr, flush=True)


def gen_mixture(n, dim, components, sigma, seed, device, means=None):
    """seeded Gaussian mixture, generated on the device in slabs"""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    if means is None:
        means = torch.rand((components, dim), generator=g, device=device, dtype=torch.float32)
    out = torch.empty((n, dim), device=device, dtype=torch.float32)
    slab = 1 << 17
    for lo in range(0, n, slab):
        hi = min(n, lo + slab)
        comp = torch.randint(0, means.shape[0], (hi - lo,), generator=g, device=device)
        out[lo:hi] = means[comp]
        out[lo:hi].add_(torch.randn((hi - lo, dim), generator=g, device=device, dtype=torch.float32), alpha=sigma)
    return out, means


def build_index(ctx, data, lists, seed, world, rank, dtype, ops, metric):
    """IVFFlat build on the GPU(s): sample, k-means, assign every row, lay out list-major.
    Returns (index handle pieces, build seconds split by phase)."""
    n, dim = data.shape
    dev = data.device
    t = {}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    # numSamples = max(50 * lists, 10000) capped by the rows (src/ivfbuild.c:446-455)
    ns = min(max(50 * lists, 10000), n)
    g = torch.Generator(device=dev)
    g.manual_seed(seed + 1)
    samples = data[torch.randperm(n, generator=g, device=dev)[:ns]].contiguous()
    if ops != api.PGV_OPS_L2:
        # spherical opclasses: SampleRows normalises the samples (src/ivfbuild.c:154-155)
        s32 = samples.float()
        samples = (s32 / s32.norm(dim=1, keepdim=True).clamp_min(1e-30)).to(samples.dtype).contiguous()
    if world == 1:
        centers, _, iters = api.kmeans(ctx, ops, dtype, dim, samples, lists,
                                       api.make_rng(seed=seed + 2), want_closest=False)
    else:
        dbg("kmeans++ init")
        init = api.kmeanspp_init(ctx, ops, dtype, dim, samples, lists, api.make_rng(seed=seed + 2))
        dbg("kmeans++ done")
        lo, hi = sharding.row_shard(ns, rank, world)
        local = samples[lo:hi].contiguous()

        def partial(s, c, closest):
            return api.lloyd_partial(ctx, ops, dtype, dim, s, c, closest)

        def finish(sums, counts, it):
            return api.lloyd_finish(ctx, ops, dtype, dim, sums, counts,
                                    api.make_rng(seed=seed + 3 + it), like=local)
        centers, _, iters = sharding.sharded_kmeans(local, init, partial, finish,
                                                    on_iter=(lambda *a: dbg("iter", *a)) if os.environ.get("PGV_BENCH_DEBUG") else None)
    ctx.sync()
    torch.cuda.synchronize()
    t["kmeans"] = time.perf_counter() - t0
    dbg("kmeans done", iters)
    t1 = time.perf_counter()
    lo, hi = sharding.row_shard(n, rank, world)
    local_lists, _ = api.assign(ctx, metric, dtype, dim, centers, data[lo:hi], want_dist=False)
    ctx.sync()
    all_lists = sharding.gather_assignments(local_lists, n, world)
    torch.cuda.synchronize()
    t["assign"] = time.perf_counter() - t1
    t2 = time.perf_counter()
    order = torch.argsort(all_lists.to(torch.int64), stable=True)
    counts = torch.bincount(all_lists.to(torch.int64), minlength=lists)
    offsets = torch.zeros(lists + 1, dtype=torch.int64, device=dev)
    offsets[1:] = torch.cumsum(counts, 0)
    vectors = data[order]
    tids = order.to(torch.int64)
    if world > 1:
        vectors, tids, offsets = sharding.local_index_arrays(vectors, tids, offsets, rank, world)
    torch.cuda.synchronize()
    t["layout"] = time.perf_counter() - t2
    t["total"] = time.perf_counter() - t0
    return centers, offsets, vectors, tids, iters, t


def recall_at_k(ivf_dist, exact_dist, k):
    """tie-tolerant recall (test/t/003_ivfflat_vector_build_recall.pl:85-90): a returned row
    counts when its distance is within the exact k-th distance"""
    kth = exact_dist[:, k - 1:k]
    hit = (ivf_dist <= kth * (1 + 1e-6) + 1e-12).sum(dim=1).clamp(max=k)
    return float(hit.float().mean().item() / k)


def cpu_baseline(centers, offsets, vectors, tids, queries, probes, k, dtype, ops, budget_s=12.0):
    """the oracle (= the reference's loops and kernels restated, built with the reference's
    flags + -march=native) answering the same queries on the host cores of this box"""
    from concurrent.futures import ThreadPoolExecutor

    from oracle import pyoracle as po
    ora = po.Oracle(native=True)
    ix = ora.index_struct(po.OPS_L2 if ops == api.PGV_OPS_L2 else po.OPS_IP,
                          po.ORA_F32 if dtype == api.PGV_F32 else po.ORA_F16, centers, offsets, vectors, tids)
    cores = min(os.cpu_count() or 1, 64)
    nq = queries.shape[0]

    def worker(w):
        done = 0
        t_end = time.perf_counter() + budget_s
        i = w
        while time.perf_counter() < t_end:
            ora.search(ix, queries[i % nq], probes, k)
            done += 1
            i += cores
        return done
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        total = sum(ex.map(worker, range(cores)))
    el = time.perf_counter() - t0
    # one thread alone = one Postgres backend
    t0 = time.perf_counter()
    single = 0
    while time.perf_counter() - t0 < 3.0:
        ora.search(ix, queries[single % nq], probes, k)
        single += 1
    single_qps = single / (time.perf_counter() - t0)
    return {"value": total / el, "unit": "queries/s", "cores": cores, "kind": "port",
            "single_thread_qps": single_qps,
            "sample": "%d queries in %.1f s on %d threads (%d more on 1 thread), same index and query "
                      "distribution; fmgr/bufmgr/tuplesort overheads of a real server not included"
                      % (total, el, cores, single)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="headline", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=1024, help="queries per step per GPU")
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--probes", type=int, default=0)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--host-io", action="store_true", help="also time the batch with host-memory queries/results")
    ap.add_argument("--recall-queries", type=int, default=256)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for a functional "
                                                     "multi-rank run on a single GPU)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
    assert world == args.gpus or world == 1, "launch N > 1 through torch.distributed.run"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    n, dim, lists, probes, tname, oname = WORKLOADS[args.workload]
    dtype = api.PGV_F32 if tname == "f32" else api.PGV_F16
    tdtype = torch.float32 if tname == "f32" else torch.float16
    ops = api.PGV_OPS_L2 if oname == "l2" else api.PGV_OPS_IP
    metric = api.PGV_L2SQ if oname == "l2" else api.PGV_NEG_IP
    esize = 4 if tname == "f32" else 2
    if args.probes:
        probes = args.probes
    k = args.k
    ctx = api.Context(local_rank, stream=torch.cuda.current_stream().cuda_stream)

    # ---------------------------------------------------------------- setup
    components = max(lists // 4, 1)
    data, means = gen_mixture(n, dim, components, 0.1, args.seed, dev)
    data = data.to(tdtype)
    log("data: %d x %d %s generated" % (n, dim, tname))
    centers, offsets, vectors, tids, iters, build_t = build_index(ctx, data, lists, args.seed, world, rank,
                                                                  dtype, ops, metric)
    log("build: %s (k-means iterations %d)" % ({a: round(b, 3) for a, b in build_t.items()}, iters))
    index = api.IvfIndex(ctx, metric, dtype, dim, centers, offsets, vectors, tids.view(torch.int64))
    local_rows = int(vectors.shape[0])
    del data

    total_batch = args.batch * world
    pool = 8
    queries, _ = gen_mixture(total_batch * pool, dim, components, 0.1, args.seed + 100, dev, means=means)
    queries = queries.to(tdtype).view(pool, total_batch, dim)

    out_d = torch.empty((total_batch, k), device=dev, dtype=torch.float32)
    out_s = torch.empty((total_batch, k), device=dev, dtype=torch.int64)
    out_t = torch.empty((total_batch, k), device=dev, dtype=torch.int64)

    def step(i):
        q = queries[i % pool]
        if world == 1:
            index.search_batch(q, probes, k, want_tid=True, out=(out_d, out_s, out_t))
            return out_d, out_t
        # N GPUs: each rank ranks its own slice of the batch against the replicated centers,
        # the probe lists are all-gathered, each rank scans the lists it owns, and the
        # per-rank top-k are merged -- per-GPU work per step does not grow with N
        mine = q[rank * args.batch:(rank + 1) * args.batch]
        lists_mine, _ = index.rank_lists(mine, probes, want_dist=False)
        lists_all = sharding.gather_probe_lists(lists_mine)
        index.scan_batch(q, lists_all, k, want_tid=True, out=(out_d, out_s, out_t))
        return sharding.merge_topk(out_d, out_t, k)

    # ---------------------------------------------------------------- recall
    rq = min(args.recall_queries, total_batch)
    if world == 1:
        exact_d, _, _ = index.search_batch(queries[0][:rq].contiguous(), lists, k, want_tid=False)
        got_d, _, _ = index.search_batch(queries[0][:rq].contiguous(), probes, k, want_tid=False)
    else:
        ed, _, et = index.search_batch(queries[0][:rq].contiguous(), lists, k, want_tid=True)
        exact_d, _ = sharding.merge_topk(ed, et, k)
        gd, _, gt = index.search_batch(queries[0][:rq].contiguous(), probes, k, want_tid=True)
        got_d, _ = sharding.merge_topk(gd, gt, k)
    ctx.sync()
    recall = recall_at_k(got_d, exact_d, k)
    log("recall@%d = %.4f at probes=%d" % (k, recall, probes))

    # ----------------------------------------------------------------- timed
    for i in range(args.warmup):
        step(i)
    ctx.set_profiling(True)
    ctx.reset_stats()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    stats = ctx.stats()
    ctx.set_profiling(False)
    if world > 1:
        el = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        elapsed = float(el.item())

    qps = total_batch * args.steps / elapsed
    launches = max(stats["scan_launches"], 1)
    algo_bytes = stats["scan_pairs"] * esize * dim     # 4*d (fp32) / 2*d (fp16) bytes per scored vector
    stream_bytes = stats["scan_rows"] * esize * dim    # rows actually streamed (shared by a query group)
    scan_s = stats["scan_ms"] / 1e3
    achieved = algo_bytes / scan_s / 1e9 if scan_s > 0 else 0.0
    roofline = {
        "kernel": "tile_scan_kernel / scan_kernel (IVFFlat list scan, src/ivfscan.c:123-187)",
        "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": achieved / HBM_PEAK_GBS,
        "traffic": pmc_traffic(args.workload, args.batch) if world == 1 else None,
        "algorithmic_bytes_per_launch": algo_bytes / launches,
        "streamed_bytes_per_launch": stream_bytes / launches,
        "streamed_GBps": stream_bytes / scan_s / 1e9 if scan_s > 0 else 0.0,
        "frac_streamed": (stream_bytes / scan_s / 1e9 if scan_s > 0 else 0.0) / HBM_PEAK_GBS,  # physical bytes vs the HBM peak
        "avg_launch_ms": stats["scan_ms"] / launches, "launches": launches,
        # the batched kernel is bound by fp32 vector-ALU issue once rows are shared by many queries:
        # 3 flop per element (subtract, multiply, add) for L2, 2 for inner product
        "valu_tflops": stats["scan_pairs"] * dim * (3.0 if oname == "l2" else 2.0) / scan_s / 1e12 if scan_s > 0 else 0.0,
        "valu_peak_tflops": 157.3,
        "note": "achieved = elem_size*dim bytes per (query,row) pair / kernel time (HIP events on the launch "
                "stream); rows probed by several queries of a batch are read from HBM once per query "
                "group, so achieved may exceed the physical rate -- streamed_GBps is the physical one",
    }
    line = {
        "metric": "QPS @ recall@10 (IVFFlat, 1M x 1536d)" if args.workload == "headline"
                  else "QPS @ recall@10 (IVFFlat)",
        "value": qps, "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": tname, "data": "synthetic",
        "config": {"workload": "%s: IVFFlat %s_%s_ops %d x %d %s, lists=%d, probes=%d, k=%d, "
                               "batch=%d queries/step/GPU, Gaussian mixture (%d components, sigma 0.1)"
                               % (args.workload, "vector" if tname == "f32" else "halfvec", oname, n, dim, tname,
                                  lists, probes, k, args.batch, components),
                   "rows": n, "dim": dim, "lists": lists, "probes": probes, "k": k,
                   "batch_per_gpu": args.batch, "parallelism": "lists sharded l %% %d, top-k all-gather" % world,
                   "local_rows": local_rows},
        "recall_at_10": recall, "build_secs": build_t["total"],
        "build_phases_secs": build_t, "kmeans_iterations": iters,
        "roofline": roofline,
        "center_rank_ms_per_step": stats["aux_ms"] / args.steps,
        "scan_ms_per_step": stats["scan_ms"] / args.steps,
    }
    if rank == 0 and world == 1 and args.host_io:
        # what a Postgres backend sees: queries and results in HOST memory, i.e. H2D of the batch, the same
        # kernels, D2H of k x (distance, slot, tid) per query and a stream sync inside every call.
        # Reported next to `value`, never as `value`.
        qh = [queries[j].cpu().numpy() for j in range(min(pool, 4))]
        for j in range(2):
            index.search_batch(qh[j % len(qh)], probes, k, want_tid=True)
        t0 = time.perf_counter()
        for j in range(args.steps):
            index.search_batch(qh[j % len(qh)], probes, k, want_tid=True)
        host_s = time.perf_counter() - t0
        line["host_buffers"] = {"value": args.batch * args.steps / host_s, "unit": "queries/s",
                                "ms_per_step": host_s / args.steps * 1e3,
                                "h2d_bytes_per_step": int(args.batch * dim * esize),
                                "d2h_bytes_per_step": int(args.batch * k * 20)}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            line["cpu_baseline"] = cpu_baseline(centers.cpu().numpy(), offsets.cpu().numpy(), vectors.cpu().numpy(),
                                                tids.cpu().numpy().astype(np.uint64),
                                                queries[1][:256].cpu().numpy(), probes, k, dtype, ops)
        except Exception as e:  # the baseline must never sink the GPU number
            line["cpu_baseline"] = {"value": None, "unit": "queries/s", "cores": 0, "kind": "port",
                                    "sample": "failed: %r" % (e,)}
    if rank == 0:
        print(json.dumps(line), flush=True)
    index.close()
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
