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
every rank generates only ITS rows of the same global data set (slab-keyed
generators), the build shards samples and heap rows by row and moves each row
to the owner of its list (l % N), the query batch grows to N x batch ("weak":
the per-GPU scan work per step is fixed), per-rank top-k are merged with one
all-gather inside the library.  `--workload c3 / c5 --gpus 8` are BASELINE
configs[2] / configs[4] at full size (10 M rows).

How the line is produced (round 3's driver run lost most of it behind one hung section): the process the driver
starts does the MANDATORY part itself, in this order -- build, recall against float64, the timed steps, oracle parity +
cpu_baseline -- and from there on the line exists.  Every other section runs in a CHILD of this script
(`--section NAME`), one at a time, each in its own session with its own budget (killed as a process group when it is
not back), writing what it has measured to a file after every step: configs (other_configs + exact_scan), hnsw,
build (pages + CPU build baseline), the PMC traffic passes, sweeps, and the sections that start backends LAST.  A
failed section is a `failures` entry and exit code 2 -- with the whole line printed; one that only ran out of its budget (a slow
box) is a `failures` entry with the prefix "budget:" and leaves the exit code alone.

What the line carries besides the contract's fields (rank 0, N = 1):
  parity       the GPU's answers for `parity_checked_queries` queries compared with the CPU oracle's
               (tid, distance) for the SAME index and queries, tie-tolerant; a mismatch exits non-zero
  recall_at_10 against an exact float64 brute force over all rows (N > 1: every rank over the rows
               it holds, merged), never against the GPU itself
  roofline     the list-scan kernel, timed with HIP events on its own stream inside the timed region:
               frac = bytes actually streamed from HBM / kernel time / 8 TB/s (physical, <= 1);
               the per-(query,row)-pair figure of SURVEY 8d is kept as algorithmic_GBps; passes =
               streamed / unique rows; traffic = HBM bytes per launch from a live rocprofv3 PMC pass
               of this same script (FETCH_SIZE x 2 per the gfx950 note + WRITE_SIZE), or null
  cpu_baseline kind "reference": the reference's OWN compiled scan (src/ivfscan.c + ivfutils.c + vector.c, unpatched, its
               own flags; oracle/_ref/ref_scan_bench_v4|v3) as backend processes over the 8 KB page image of this index on
               this box's host cores, its answers checked against the oracle's; `port_value` beside it: the oracle's
               restatement (bare loops over contiguous arrays, pinned threads); bounded samples
               (.page_image: the port over the 8 KB pages the product build wrote)
  other_configs  c2, c3shard, c5shard (BASELINE configs[1], one GPU's share of [2] and [4]): QPS, recall, the scan
               kernel's roofline, oracle parity each
  hnsw         BASELINE configs[3] at full size (1 M x 1536, GPU-built graph): ef_search 40 / 100 / 200
  exact_scan   BASELINE configs[0] (10 k x 128, 100 queries) through pgv_exact_topk beside the oracle's loop
  build        kernel-only build (k-means + assignment, data resident in HBM) and build_secs_pages:
               the product path pgv_host_ivf_build -> 8 KB pages -> stage -> upload from host memory
  cpu_build_baseline  the oracle's IvfflatKmeans (one thread, like the reference) on the build's own sample and
               its assignment loop at 1 / all threads (extrapolated), beside build_secs
  batch_sweep  batch 1 (the amgettuple path: pgv_query_* and the C host glue), 4, 16, 64, 256
  probes_sweep probes 1 / 10 / 32 / 100 with recall each; `uniform`: the same on U[0,1)^d data
  bound_modes  the statistical and the worst-case completeness bound of the MFMA L2 paths side by side
  concurrent_backends  N independent backends / N pooled clients against ONE device mirror, as threads of one
               process and as PROCESSES (pgv_index_export / import, pool state in a shared segment)
  sections     seconds, exit code and budget of every child
"""
import argparse
import ctypes
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import threading  # noqa: E402

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
    # BASELINE configs[2] / configs[4] at full size: for --gpus 8 (1.25 M rows per GPU; one GPU holds them too, 61 GB)
    "c3": (10_000_000, 1536, 4096, 64, "f32", "ip"),
    "c5": (10_000_000, 3072, 4096, 64, "f16", "l2"),
    # configs[2]'s shape at reduced rows, for a functional N-rank run on ONE GPU (--gpus 8 --backend gloo)
    "c3small": (320_000, 1536, 256, 16, "f32", "ip"),
    # the headline's shape on the mid-difficulty data set (gen_hard): recall@10 well below 1 at probes 10
    "hard": (1_000_000, 1536, 1000, 10, "f32", "l2"),
    "hardsmall": (100_000, 256, 100, 10, "f32", "l2"),
    "small": (100_000, 256, 100, 10, "f32", "l2"),          # quick functional run
    "smallh": (100_000, 512, 100, 10, "f16", "ip"),
}
PLACEMENT = {"policy": "balanced"}   # --placement: how lists go to ranks (pgvector_amd/sharding.py plan_owners)
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
RTOL = 1e-5            # north_star: float tolerance of the distances


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print(*a, file=sys.stderr, flush=True)


def dbg(*a):
    if os.environ.get("PGV_BENCH_DEBUG"):
        print("[rank %s]" % os.environ.get("RANK", "0"), *a, file=sys.stderr, flush=True)


SLAB = 1 << 17


def gen_mixture(n, dim, components, sigma, seed, device, means=None, lo=0, hi=None):
    """seeded Gaussian mixture, rows [lo, hi) of n, generated on the device in slabs of 2^17 rows.  Every slab has
    a generator of its own keyed by (seed, slab index): any rank produces exactly its own rows of the same global
    data set without generating anybody else's (N GPUs never hold the whole set on one rank)."""
    hi = n if hi is None else hi
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    if means is None:
        means = torch.rand((components, dim), generator=g, device=device, dtype=torch.float32)
    out = torch.empty((hi - lo, dim), device=device, dtype=torch.float32)
    for s0 in range(lo // SLAB * SLAB, hi, SLAB):
        s1 = min(n, s0 + SLAB)
        g.manual_seed(seed * 1000003 + 7919 * (s0 // SLAB) + 1)
        comp = torch.randint(0, means.shape[0], (s1 - s0,), generator=g, device=device)
        slab = means[comp]
        slab.add_(torch.randn((s1 - s0, dim), generator=g, device=device, dtype=torch.float32), alpha=sigma)
        a, b = max(lo, s0), min(hi, s1)
        out[a - lo:b - lo] = slab[a - s0:b - s0]
    return out, means


HARD = {"sigma": 3.0, "alpha": 0.6, "zipf": 0.8}   # --hard-sigma / --hard-alpha / --hard-zipf


def hard_scales(dim, alpha, device):
    """per-dimension standard deviations ~ (j + 1)^(-alpha / 2), mean variance 1: a power-law spectrum (what PCA of real
    embeddings looks like: a few dozen directions carry most of the variance -- low intrinsic dimension)"""
    sc = torch.arange(1, dim + 1, device=device, dtype=torch.float32).pow(-alpha / 2.0)
    return sc * (dim / (sc * sc).sum()).sqrt()


def gen_hard(n, dim, components, seed, device, means=None, lo=0, hi=None, sigma=None, alpha=None, zipf=None):
    """the mid-difficulty data set (SURVEY 8d "Synthetic inputs"): a Gaussian mixture whose components
    OVERLAP (component spread `sigma` x the spread of the means, against 0.1 x a unit cube for the easy mixture), with a
    power-law variance per dimension and Zipf-weighted components (unbalanced lists).  Slab-keyed like gen_mixture.
    Returns (rows [lo, hi), means)."""
    sigma = HARD["sigma"] if sigma is None else sigma
    alpha = HARD["alpha"] if alpha is None else alpha
    zipf = HARD["zipf"] if zipf is None else zipf
    hi = n if hi is None else hi
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    sc = hard_scales(dim, alpha, device)
    if means is None:
        means = torch.randn((components, dim), generator=g, device=device, dtype=torch.float32) * sc
    w = torch.arange(1, means.shape[0] + 1, device=device, dtype=torch.float32).pow(-zipf)
    out = torch.empty((hi - lo, dim), device=device, dtype=torch.float32)
    for s0 in range(lo // SLAB * SLAB, hi, SLAB):
        s1 = min(n, s0 + SLAB)
        g.manual_seed(seed * 1000003 + 7919 * (s0 // SLAB) + 1)
        comp = torch.multinomial(w, s1 - s0, replacement=True, generator=g)
        slab = means[comp]
        slab.add_(torch.randn((s1 - s0, dim), generator=g, device=device, dtype=torch.float32) * sc, alpha=sigma)
        a, b = max(lo, s0), min(hi, s1)
        out[a - lo:b - lo] = slab[a - s0:b - s0]
    return out, means


def gen_uniform(n, dim, seed, device):
    """U[0,1)^d: what every reference test uses (test/t/003_ivfflat_vector_build_recall.pl:60)"""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    return torch.rand((n, dim), generator=g, device=device, dtype=torch.float32)


def sample_rows(data, lists, seed, ops, n_global=None, world=1):
    """numSamples = max(50 * lists, 10000) capped by the rows (src/ivfbuild.c:446-455), this rank's share of them
    drawn from its own rows; spherical opclasses normalise the samples (SampleCallback, :148-156)"""
    n = data.shape[0]
    ns = min(max(50 * lists, 10000), n_global or n)
    ns = min((ns + world - 1) // world, n)
    g = torch.Generator(device=data.device)
    g.manual_seed(seed + 1)
    samples = data[torch.randperm(n, generator=g, device=data.device)[:ns]].contiguous()
    if ops != api.PGV_OPS_L2:
        s32 = samples.float()
        samples = (s32 / s32.norm(dim=1, keepdim=True).clamp_min(1e-30)).to(samples.dtype).contiguous()
    return samples


def build_index(ctx, data, lists, seed, world, rank, dtype, ops, metric, comm=None, row_lo=0, n_global=None, sharded=None):
    """IVFFlat build on the GPU(s), data resident in HBM: sample, k-means, assign every row, lay out list-major.
    `data` holds this rank's heap rows [row_lo, row_lo + len) of n_global.  With N ranks the k-means samples and
    the heap rows are sharded by row (pgv_kmeans_sharded does the Lloyd exchanges inside the library), every rank
    assigns its own rows, and one all-to-all brings each row to the rank that owns its list (l % N).
    Returns the local image pieces and the seconds per phase."""
    n, dim = data.shape
    t = {}
    sharded = world > 1 if sharded is None else sharded
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    samples = sample_rows(data, lists, seed + 31 * rank, ops, n_global, world)
    if not sharded:
        centers, _, iters = api.kmeans(ctx, ops, dtype, dim, samples, lists,
                                       api.make_rng(seed=seed + 2), want_closest=False)
    else:
        centers, _, iters = comm.kmeans(ops, dtype, dim, samples, lists,
                                        api.make_rng(seed=seed + 2), want_closest=False)
    ctx.sync()
    torch.cuda.synchronize()
    t["kmeans"] = time.perf_counter() - t0
    t1 = time.perf_counter()
    if not sharded:
        # the product's own build: rows assigned where they are kept, the sort by list a device gather whose result
        # IS the mirror (pgv_builder_*); TIDs = heap positions
        b = api.IvfBuilder(ctx, metric, dtype, dim, centers, expected_rows=n)
        b.add(data)
        ctx.sync()
        torch.cuda.synchronize()
        t["assign"] = time.perf_counter() - t1
        t2 = time.perf_counter()
        index, offsets_h, lists_h = b.finish(want_lists=True)
        ctx.sync()
        torch.cuda.synchronize()
        t["layout"] = time.perf_counter() - t2
        t["total"] = time.perf_counter() - t0
        b.close()
        # the harness's own copy of the same layout (float64 ground truth, the oracle's arrays): not part of the build
        lists64 = torch.from_numpy(lists_h.astype(np.int64)).to(data.device)
        order = torch.argsort(lists64, stable=True)
        offsets = torch.from_numpy(offsets_h).to(data.device)
        return centers, offsets, data[order], order, iters, t, index
    local_lists, _ = api.assign(ctx, metric, dtype, dim, centers, data, want_dist=False)
    ctx.sync()
    torch.cuda.synchronize()
    t["assign"] = time.perf_counter() - t1
    t2 = time.perf_counter()
    row_ids = torch.arange(row_lo, row_lo + n, dtype=torch.int64, device=data.device)
    # which rank gets which list: by rows (longest list first onto the lightest rank) unless --placement modulo
    owners = sharding.plan_owners(sharding.global_list_sizes(local_lists, lists), world, PLACEMENT["policy"])
    vectors, tids, offsets = sharding.exchange_rows(data, row_ids, local_lists, lists, owners)
    torch.cuda.synchronize()
    t["layout"] = time.perf_counter() - t2   # includes the all-to-all of the rows
    t["total"] = time.perf_counter() - t0
    index = api.IvfIndex(ctx, metric, dtype, dim, centers, offsets, vectors, tids.view(torch.int64))
    return centers, offsets, vectors, tids, iters, t, index


def exact_topk_fp64(vectors, queries, k, metric):
    """ground truth: exact float64 brute force over every row, on the device, in slabs
    -> (distances [nq x k] float64, row slots [nq x k])"""
    q = queries.double()
    nq = q.shape[0]
    best_d = torch.full((nq, k), float("inf"), dtype=torch.float64, device=q.device)
    best_i = torch.full((nq, k), -1, dtype=torch.int64, device=q.device)
    qq = (q * q).sum(1)[:, None]
    slab = 1 << 16
    for lo in range(0, vectors.shape[0], slab):
        v = vectors[lo:lo + slab].double()
        if metric == api.PGV_L2SQ:
            d = qq + (v * v).sum(1)[None, :] - 2.0 * (q @ v.T)
        else:
            d = -(q @ v.T)
        cand_d = torch.cat([best_d, d], dim=1)
        cand_i = torch.cat([best_i, torch.arange(lo, lo + v.shape[0], device=q.device)[None, :].expand(nq, -1)], dim=1)
        top = torch.topk(cand_d, k, dim=1, largest=False)
        best_d, best_i = top.values, torch.gather(cand_i, 1, top.indices)
    return best_d, best_i


def recall_at_k(got_dist, exact_dist, k):
    """tie-tolerant recall (test/t/003_ivfflat_vector_build_recall.pl:85-90): a returned row counts
    when its distance is within the exact k-th distance (float tolerance of the kernel)"""
    kth = exact_dist[:, k - 1:k]
    tol = RTOL * kth.abs() + 1e-12
    hit = (got_dist.double() <= kth + tol).sum(dim=1).clamp(max=k)
    return float(hit.double().mean().item() / k)


def topk_equiv(got_ids, got_d, want_ids, want_d, rtol=RTOL):
    """the reference's own tie-tolerant rule: ids must match wherever the order is determined beyond
    the float tolerance; inside a run of (near-)equal distances any order is accepted.
    Returns None or a description of the mismatch."""
    got_ids, want_ids = list(got_ids), list(want_ids)
    if len(got_ids) != len(want_ids):
        return "lengths %d vs %d" % (len(got_ids), len(want_ids))
    gd, wd = np.asarray(got_d, np.float64), np.asarray(want_d, np.float64)
    bad = np.abs(gd - wd) > rtol * np.abs(wd) + 1e-30
    if bad.any():
        return "distance %r vs %r" % (gd[bad][:3].tolist(), wd[bad][:3].tolist())
    i, n = 0, len(want_ids)
    while i < n:
        j = i + 1
        while j < n and abs(wd[j] - wd[j - 1]) <= 4 * rtol * max(abs(wd[j]), 1e-30):
            j += 1
        if j == n:
            if set(got_ids[:i]) != set(want_ids[:i]):
                return "ids before the last tie run differ"
            break
        if sorted(got_ids[i:j]) != sorted(want_ids[i:j]):
            return "ids %r vs %r at %d..%d" % (got_ids[i:j], want_ids[i:j], i, j)
        i = j
    return None


def cpu_quota():
    """CPUs this container may actually burn: the cgroup's CFS quota (v2 cpu.max, v1 cpu.cfs_quota_us), or None"""
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if quota == "max" else float(quota) / float(period)
    except Exception:  # noqa: BLE001
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / per
    except Exception:  # noqa: BLE001
        return None


def cpu_threads(ora):
    """threads for the CPU baselines: the CPUs allowed by affinity, capped by the cgroup quota (pinning 256 threads
    under a 16-CPU quota only buys throttling) and by 256"""
    cpus = ora.lib.ora_bench_cpus()
    quota = cpu_quota()
    cores = cpus if quota is None else min(cpus, max(1, int(quota)))
    return max(1, min(cores, 256)), cpus, quota


def cpu_baseline(centers, offsets, vectors, tids, queries, probes, k, dtype, ops, budget_s=12.0, pages=None):
    """the oracle (= the reference's loops and kernels restated, built with the reference's flags +
    -march=native) answering the same queries on the host cores of this box: one pinned thread per "backend"
    (oracle/oracle_bench.c), the index re-homed in pieces spread round the memory nodes (what numactl --interleave
    gives shared_buffers) -- an array filled by one thread sits on one node and caps the scaling at ~5 x.
    Returns the baseline record and the oracle's (tids, distances) per query for the parity check."""
    from oracle import pyoracle as po
    ora = po.Oracle(native=True)
    oops = po.OPS_L2 if ops == api.PGV_OPS_L2 else po.OPS_IP
    odt = po.ORA_F32 if dtype == api.PGV_F32 else po.ORA_F16
    cores, cpus, quota = cpu_threads(ora)
    svec, release = ora.spread(vectors, cores)
    ix = ora.index_struct(oops, odt, centers, offsets, svec, tids)
    esz = 4 if dtype == api.PGV_F32 else 2
    dim = vectors.shape[1]
    off = np.asarray(offsets)
    # bytes the scan of one query reads: its probed lists' rows + every center (GetScanLists)
    rows_per_query = float(np.mean([sum(off[l + 1] - off[l] for l in ora.get_scan_lists(ix, q, probes)[0])
                                    for q in queries[:32]]))
    bytes_per_query = (rows_per_query + centers.shape[0]) * dim * esz
    answers, total, el = ora.bench_search(ix, queries, probes, k, cores, budget_s)
    _, single, single_el = ora.bench_search(ix, queries[:64], probes, k, 1, 3.0)
    rec = {"value": total / el, "unit": "queries/s", "cores": cores, "kind": "port",
           "single_thread_qps": single / single_el, "single_thread_ms_per_query": single_el / single * 1e3,
           "aggregate_GBps": total / el * bytes_per_query / 1e9, "single_thread_GBps": single / single_el * bytes_per_query / 1e9,
           "scaling_over_one_thread": (total / el) / (single / single_el),
           "placement": "one pthread per backend pinned to allowed CPU t * %d / %d (sched_getaffinity: %d CPUs allowed, "
                        "os.cpu_count %s, cgroup CPU quota %s); index rows copied into 2 MB pieces first-touched round-robin by those "
                        "threads (interleaved over the memory nodes)" % (cpus, cores, cpus, os.cpu_count(), quota),
           "layout": "contiguous list-major arrays (an upper bound of the reference: no 8 KB page walk, "
                     "no fmgr/bufmgr/tuplesort overheads)",
           "sample": "%d queries in %.1f s on %d threads (%d more on 1 thread), same index and queries as the "
                     "parity check" % (total, el, cores, single)}
    release()
    page_answers = None
    if pages is not None:
        # the same loops over the emulated 8 KB page image (oracle_pages.c): page headers, line
        # pointers, IndexTuple headers, nextblkno chains, one 1536-d tuple per page
        ptr, nblocks = pages
        pa, ptotal, pel = ora.bench_search(None, queries, probes, k, cores, max(budget_s / 2.0, 3.0),
                                           pages=(ptr, nblocks), ops=oops, dtype=odt)
        _, psingle, psingle_el = ora.bench_search(None, queries[:64], probes, k, 1, 3.0, pages=(ptr, nblocks), ops=oops,
                                                  dtype=odt)
        page_answers = [(t, d, None) for t, d in pa]
        rec["contiguous"] = {key: rec[key] for key in ("value", "single_thread_qps", "single_thread_ms_per_query",
                                                       "aggregate_GBps", "scaling_over_one_thread", "layout", "sample")}
        rec.update({"value": ptotal / pel, "single_thread_qps": psingle / psingle_el,
                    "single_thread_ms_per_query": psingle_el / psingle * 1e3,
                    "aggregate_GBps": ptotal / pel * bytes_per_query / 1e9,
                    "single_thread_GBps": psingle / psingle_el * bytes_per_query / 1e9,
                    "scaling_over_one_thread": (ptotal / pel) / (psingle / psingle_el),
                    "layout": "emulated 8 KB page image (oracle_pages.c walks meta / list / entry pages like "
                              "src/ivfscan.c:47-187; no buffer pins, fmgr or tuplesort copies: still an upper bound); "
                              "the image is placed by the list-parallel page writer that built it",
                    "sample": "%d queries in %.1f s on %d threads (%d more on 1 thread) over the %d pages the product "
                              "build wrote" % (ptotal, pel, cores, psingle, nblocks)})
    return rec, answers, page_answers


def reference_scan_program():
    """oracle/_ref/ref_scan_bench_v4 where the host has AVX-512 (what the reference's -march=native enables on it), else
    _v3 (AVX2); None when neither was built (no reference tree where build() ran)"""
    flags = ""
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("flags"):
                flags = ln
                break
    except OSError:
        pass
    have = set(flags.split())
    order = ["v4", "v3"] if {"avx512f", "avx512bw", "avx512cd", "avx512dq", "avx512vl"} <= have else ["v3"]
    for isa in order:
        exe = os.path.join(ROOT, "oracle", "_ref", "ref_scan_bench_" + isa)
        if os.path.exists(exe) and os.access(exe, os.X_OK):
            return exe, "x86-64-" + isa
    return None, None


def reference_baseline(centers, offsets, vectors, tids, queries, probes, k, dtype, ops, oracle_answers, cores,
                       secs=10.0, secs_single=5.0):
    """cpu_baseline of kind "reference": the reference's OWN compiled scan (src/ivfscan.c:47-187,361-414 with its
    ivfutils.c / vector.c / halfvec.c / halfutils.c, unpatched, its own flags; oracle/ref_scan_bench.c) answering the
    parity queries over the 8 KB page image of THIS index, one single-threaded backend process per connection: `cores`
    processes for `secs`, then one for `secs_single`.  The pages are written by the product's page writer
    (pgv_host_ivf_write_index, byte-identical to the reference's own build: tests/c/ext_driver.c) from the same arrays the
    oracle reads.  Returns the record (with the reference's answers checked against the oracle's) or raises."""
    from pgvector_amd import _host
    exe, isa = reference_scan_program()
    if exe is None:
        raise RuntimeError("oracle/_ref/ref_scan_bench_v* not built (the reference tree was absent where build() ran)")
    nq, dim = queries.shape
    procs = max(1, min(int(cores), 40))
    t0 = time.perf_counter()
    rel = _host.Relation()
    page_tids = (np.asarray(tids).astype(np.uint64) << np.uint64(16)) | np.uint64(1)
    rel.write_index(dtype, centers, offsets, vectors, page_tids)
    nblocks = int(rel.nblocks)
    t_write = time.perf_counter() - t0
    # the page image goes where there is room for it (8 GB at the headline: a container's /dev/shm may be 64 MB)
    need = nblocks * 8192 + (64 << 20)
    shm = next((c for c in ("/dev/shm", "/tmp", ROOT) if os.path.isdir(c) and shutil.disk_usage(c).free > need), None)
    if shm is None:
        raise RuntimeError("no directory with %d MB free for the page image" % (need >> 20))
    d = tempfile.mkdtemp(prefix="pgv_refscan_", dir=shm)
    try:
        t0 = time.perf_counter()
        arr = np.ctypeslib.as_array((ctypes.c_uint8 * (nblocks * 8192)).from_address(rel.rel.pages))
        arr.tofile(os.path.join(d, "pages.bin"))
        del arr, rel
        np.ascontiguousarray(queries).tofile(os.path.join(d, "queries.bin"))
        t_dump = time.perf_counter() - t0
        cmd = [exe, os.path.join(d, "pages.bin"), os.path.join(d, "queries.bin"), str(dim), str(nq), str(probes), str(k),
               str(procs), str(secs), str(secs_single), os.path.join(d, "answers.bin"),
               "f32" if dtype == api.PGV_F32 else "f16", "l2" if ops == api.PGV_OPS_L2 else "ip"]
        t0 = time.perf_counter()
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=secs + secs_single + 240)
        t_run = time.perf_counter() - t0
        if r.returncode != 0:
            raise RuntimeError("ref_scan_bench exit code %d: %s" % (r.returncode, r.stderr[-300:]))
        rec = json.loads(r.stdout.strip().splitlines()[-1])
        got = np.fromfile(os.path.join(d, "answers.bin"), dtype=np.uint64).reshape(nq, k)
    finally:
        shutil.rmtree(d, ignore_errors=True)
    bad = []
    for i in range(nq):
        wt, wd = oracle_answers[i]
        g = (got[i][:len(wt)] >> np.uint64(16)).tolist()
        why = topk_equiv(g, wd, wt.tolist(), wd)
        if why:
            bad.append((i, why))
    return {"value": rec["qps"], "unit": "queries/s", "cores": procs, "kind": "reference",
            "single_thread_qps": rec["single_qps"], "isa": isa,
            "scaling_over_one_process": rec["qps"] / rec["single_qps"] if rec["single_qps"] else None,
            "program": "oracle/_ref/%s: the reference's src/ivfscan.c + ivfutils.c + vector.c + halfvec.c + halfutils.c, unpatched, "
                       "compiled with its own flags (Makefile:30 + -O2) at %s (its -march=native cannot travel to this host), as %d "
                       "single-threaded backend processes over the stand-in server runtime of tests/c (buffer manager, fmgr, slots, "
                       "tuplesort -- all lighter than PostgreSQL's: an upper bound on the reference inside a real server)"
                       % (os.path.basename(exe), isa, procs),
            "sample": "%d queries in %.1f s by %d processes (%d more in %.1f s by 1), %d pages, same index and queries as the parity check"
                      % (rec["queries"], rec["secs"], procs, rec["single_queries"], rec["single_secs"], nblocks),
            "pages": nblocks, "page_write_secs": t_write, "page_dump_secs": t_dump, "program_secs": t_run,
            "program_load_secs": rec["load_secs"], "answer_pass_secs": rec["answer_pass_secs"],
            "answers_against_oracle": {"checked_queries": nq, "mismatches": len(bad), "first": repr(bad[0]) if bad else None,
                                       "rule": "the reference returns heap TIDs in order; they must be the oracle's wherever "
                                               "its distances differ beyond 1e-5 relative"}}


def cpu_build_kmeans(host_samples, lists, dtype, ops, seed, out):
    """SURVEY 8d, first half (a background thread while the GPU sections go on; the oracle's C releases the GIL and
    uses ONE core, like the reference: IvfflatKmeans is single-threaded whatever max_parallel_maintenance_workers
    is): InitCenters + ElkanKmeans restated (src/ivfkmeans.c:23-91, 246-485) on the very sample the GPU build used."""
    try:
        from oracle import pyoracle as po
        ora = po.Oracle(native=True)
        oops = po.OPS_L2 if ops == api.PGV_OPS_L2 else po.OPS_IP
        odt = po.ORA_F32 if dtype == api.PGV_F32 else po.ORA_F16
        t0 = time.perf_counter()
        centers, _, iters = ora.kmeans(oops, odt, host_samples, lists, ora.prng(seed))
        out.update({"kind": "port", "kmeans_secs_one_thread": time.perf_counter() - t0, "kmeans_iterations": int(iters),
                    "kmeans_sample_rows": int(host_samples.shape[0]), "_centers": centers})
    except Exception as e:  # noqa: BLE001
        out["error"] = repr(e)


def cpu_build_assign(host_rows, dtype, ops, out):
    """second half, in the foreground (it takes every core for a second): the argmin loop of src/ivfbuild.c:183-192 on
    a 100k-row subsample at 1 thread and at every core (the reference's parallel build splits the heap scan between
    workers, src/ivfbuild.c:830-966), extrapolated to all rows"""
    centers = out.pop("_centers", None)
    if centers is None:
        return
    try:
        from oracle import pyoracle as po
        ora = po.Oracle(native=True)
        oops = po.OPS_L2 if ops == api.PGV_OPS_L2 else po.OPS_IP
        odt = po.ORA_F32 if dtype == api.PGV_F32 else po.ORA_F16
        n = host_rows.shape[0]
        sub = np.ascontiguousarray(host_rows[:: max(n // 100_000, 1)][:100_000])
        cores = cpu_threads(ora)[0]
        one_rows = sub[:4000]
        _, one_s = ora.bench_assign(oops, odt, centers, one_rows, 1)
        _, all_s = ora.bench_assign(oops, odt, centers, sub, cores)
        km_s = out["kmeans_secs_one_thread"]
        out.update({
            "assign_rows_per_s_one_thread": one_rows.shape[0] / one_s,
            "assign_rows_per_s_all_threads": sub.shape[0] / all_s, "assign_threads": cores,
            "assign_sample_rows": int(sub.shape[0]),
            "assign_secs_extrapolated_one_thread": n / (one_rows.shape[0] / one_s),
            "assign_secs_extrapolated_all_threads": n / (sub.shape[0] / all_s),
            "build_secs_extrapolated_one_thread": km_s + n / (one_rows.shape[0] / one_s),
            "build_secs_extrapolated_all_threads": km_s + n / (sub.shape[0] / all_s),
            "note": "k-means measured in full on one thread (as the reference runs it); assignment measured on a "
                    "subsample and scaled to %d rows (labelled extrapolated); no heap scan, tuplesort or page writes "
                    "on the CPU side" % n})
    except Exception as e:  # noqa: BLE001
        out["error"] = repr(e)


def concurrent_backends(index, device, qhost, queries, probes, k, args, dev, out, flush):
    """what a server looks like to the device: N backends against ONE uploaded index.  (a) backends as threads of a
    plain C driver, one query at a time each (pgv_query_rank + pgv_query_scan, the amgettuple path); the same as
    PROCESSES that import the mirror; clients behind the pooler as threads and as processes; (b) two submitters of
    1024-query batches on two streams.  Every row has a deadline of its own and says where it was stuck; `out` is
    flushed after every row, so a run that is cut still carries the rows it finished."""
    from pgvector_amd import _host
    qh = np.ascontiguousarray(qhost)
    row_deadline = 25.0
    out["single_query"] = {}
    for nb in (1, 2, 4, 8, 16, 32):
        log("  backends: %d threads" % nb)
        try:
            out["single_query"][str(nb)] = _host.run_backend_threads(index, qh, probes, k, nb, 400, device, row_deadline)
        except Exception as e:  # noqa: BLE001
            out["single_query"][str(nb)] = {"error": repr(e)}
            flush()
            raise   # a stuck thread still sits in the library: nothing after it in this process can be trusted
        flush()
    out["single_query"]["driver"] = ("tools/backends_driver.c: THREADS of one process over the C ABI, one pgv_ctx + "
                                     "pgv_index_share view + pgv_query per backend, pgv_query_rank + pgv_query_scan per query; "
                                     "at most PGV_MAX_INFLIGHT_SCANS (16) scans in flight per process")
    # the same with PROCESSES, which is what a Postgres backend is: every process imports the ONE device mirror
    # (pgv_index_export / pgv_index_import: hipIpc, no copy) and scans it on a context and stream of its own
    out["single_query_processes"] = {}
    # (32 own-context processes -- 7-10 k QPS on these boxes, profiles/r03/processes_*.json, DESIGN 4.8b -- only with
    # --all-process-rows: that row costs the most and teaches the least in a driver run)
    proc_rows = (1, 4, 8, 16, 32) if args.all_process_rows else (1, 4, 8, 16)
    if os.environ.get("PGV_BENCH_PROC_ROWS"):   # e.g. "5,6,7,8": where the own-context cliff sits (profiles/r06/own_context_cliff.md)
        proc_rows = tuple(int(x) for x in os.environ["PGV_BENCH_PROC_ROWS"].split(","))
    for nb in proc_rows:
        log("  backends: %d processes" % nb)
        try:
            out["single_query_processes"][str(nb)] = _host.run_backend_processes(index, qh, probes, k, 0, nb, 300,
                                                                                 deadline_s=row_deadline)
        except Exception as e:  # noqa: BLE001
            out["single_query_processes"][str(nb)] = {"error": repr(e)}
            flush()
            break
        flush()
    out["single_query_processes"]["driver"] = ("tools/pgv_backend.c `query`: one PROCESS per backend (fork + exec), the "
                                               "mirror imported from the owner's export handle, pgv_query_rank + "
                                               "pgv_query_scan per query; HBM holds the index once")
    # (a') the same kind of clients behind the host glue's pooler (ivf_pool.c): one query each, batched on arrival
    out["pooled_single_query"] = {}
    for nc in (16, 64, 256):
        log("  backends: %d pooled client threads" % nc)
        try:
            out["pooled_single_query"][str(nc)] = _host.run_pooled_threads(index, qh, probes, k, nc, max(40, 6000 // nc),
                                                                           1024, 50, 3, device, row_deadline)
        except Exception as e:  # noqa: BLE001
            out["pooled_single_query"][str(nc)] = {"error": repr(e)}
            flush()
            raise
        flush()
    out["pooled_single_query"]["pool"] = ("pgv_host_pool_*: client THREADS block in pgv_host_pool_search with one query each; "
                                          "max_batch 1024, max_wait 50 us, 3 lanes (contexts), one scan at a time + linger 300 us; host buffers in and out")
    # ... and with GPU-less client PROCESSES in front of two lane-server processes: the pool's slots, lane words and
    # payload ring live in a shared segment (non-private futexes, a robust process-shared mutex)
    out["pooled_single_query_processes"] = {}
    for nc in (16, 64, 256, 1024):
        log("  backends: %d pooled client processes" % nc)
        try:
            out["pooled_single_query_processes"][str(nc)] = _host.run_backend_processes(
                index, qh, probes, k, 1, nc, max(200, 48000 // nc), max_batch=1024, max_wait_us=50, lanes=3,
                server_processes=True, deadline_s=row_deadline)
        except Exception as e:  # noqa: BLE001
            out["pooled_single_query_processes"][str(nc)] = {"error": repr(e)}
            flush()
            break
        flush()
    out["pooled_single_query_processes"]["pool"] = ("tools/pgv_backend.c `client` x N + `serve` x 3 lanes: every client and "
                                                    "every lane leader is a process; the leaders import the mirror; "
                                                    "one scan at a time + linger 300 us, baton wake (ivf_pool.c); the "
                                                    "processes are confined to as many CPUs as the cgroup's quota pays "
                                                    "for (tools/backends_driver.c pinned_cpus: a quota without a cpuset "
                                                    "makes CFS bandwidth accounting the bottleneck, 200 us of system "
                                                    "time per query)")
    # (b) batches from two submitters
    log("  backends: two batch submitters")
    ctx2 = api.Context(device)
    v2 = index.share(ctx2)
    bufs = [tuple(torch.empty((args.batch, k), device=dev, dtype=dt) for dt in (torch.float32, torch.int64, torch.int64))
            for _ in range(2)]
    handles = [index, v2]
    pool = queries.shape[0]

    def step(j):
        handles[j % 2].search_batch(queries[j % pool][:args.batch], probes, k, want_tid=True, out=bufs[j % 2])
    s = timed_steps(step, 40, warmup=4)
    out["batches_from_two_submitters"] = {"qps": args.batch / s, "ms_per_batch": s * 1e3,
                                          "note": "alternate %d-query batches on two contexts (two streams): the second "
                                                  "one's ranking / planning / top-k run under the first one's scan" % args.batch}
    flush()
    v2.close()
    ctx2.close()
    return out


def hnsw_section(ctx, dev, args, failures, rows=1_000_000, dim=1536, m=16, efc=64, k=10, nq=1000, efs=(40, 100, 200)):
    """BASELINE configs[3]: vector_cosine_ops HNSW, 1 M x 1536, m 16 (src/hnswbuild.c:376-476 build loop,
    src/hnswscan.c:25-56 + src/hnswutils.c:824-987 scan).  The graph is built on the GPU by pgv_host_hnsw_build,
    every scan's first batch walked on the device by pgv_hnsw_search, ef_search 40 (the reference default) / 100 / 200.
    Parity: the oracle's HnswSearchLayer restatement walks the SAME graph (ora_hnsw_import) for 32 queries at ef 100;
    recall against an exact float64 scan."""
    from pgvector_amd import _host
    g = torch.Generator(device=dev)
    g.manual_seed(args.seed + 21)
    comps = torch.rand((64, dim), generator=g, device=dev)
    data = torch.empty((rows, dim), device=dev)
    for lo in range(0, rows, SLAB):
        hi = min(rows, lo + SLAB)
        data[lo:hi] = comps[torch.randint(0, 64, (hi - lo,), generator=g, device=dev)] + \
            0.1 * torch.randn((hi - lo, dim), generator=g, device=dev)
        data[lo:hi] /= data[lo:hi].norm(dim=1, keepdim=True)  # HnswFormIndexValue normalises (src/hnswutils.c:406-428)
    q = comps[torch.randint(0, 64, (nq,), generator=g, device=dev)] + 0.1 * torch.randn((nq, dim), generator=g, device=dev)
    q = (q / q.norm(dim=1, keepdim=True)).contiguous()
    host_rows = data.cpu().numpy()
    mirror = api.Hnsw(ctx, api.PGV_NEG_IP, api.PGV_F32, dim, data)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    built = _host.hnsw_build(mirror, host_rows, m, efc, api.make_rng(seed=1), max_batch=1024)
    torch.cuda.synchronize()
    build_s = time.perf_counter() - t0
    # exact top-k in float64, in row slabs
    q64 = q.double()
    best = torch.full((nq, k), -2.0, dtype=torch.float64, device=dev)
    for lo in range(0, rows, 100_000):
        best = torch.topk(torch.cat([best, q64 @ data[lo:lo + 100_000].double().T], dim=1), k, dim=1).values
    kth = best[:, -1]
    qd = q.repeat(max(1, 20000 // nq), 1).contiguous()
    sweep, keep = {}, None
    def run_ef(ef):
        mirror.search(qd[:64].contiguous(), ef, k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        elem, gd, scored = mirror.search(qd, ef, k)
        torch.cuda.synchronize()
        dev_s = time.perf_counter() - t0
        e = elem[:nq]
        got_ip = (q64[:, None, :] * data[e.clamp(min=0)].double()).sum(-1)
        recall = float(((got_ip >= kth[:, None] - 1e-9) & (e >= 0)).sum().item()) / (nq * k)
        gbps = float(scored.sum().item()) * dim * 4 / dev_s / 1e9
        sweep[str(ef)] = {"qps": qd.shape[0] / dev_s, "recall_at_10": recall,
                          "scored_elements_per_query": float(scored.float().mean().item()),
                          "scored_rows_GBps": gbps, "frac_of_hbm_peak": gbps / HBM_PEAK_GBS}
        return e, gd, recall
    for ef in efs:
        e, gd, _ = run_ef(ef)
        if ef == 100:
            keep = (e.cpu().numpy(), gd[:nq].cpu().numpy())
    # a usable operating point: the smallest ef_search of the ladder (up to the reference's maximum of 1000,
    # src/hnsw.h HNSW_MAX_EF_SEARCH) whose recall@10 reaches 0.95, and the QPS there
    at95 = None
    try:
        for ef in sorted(set(efs) | {400, 600, 800, 1000}):
            rec = sweep[str(ef)]["recall_at_10"] if str(ef) in sweep else run_ef(ef)[2]
            if rec >= 0.95:
                at95 = {"ef_search": ef, "qps": sweep[str(ef)]["qps"], "recall_at_10": rec,
                        "frac_of_hbm_peak": sweep[str(ef)]["frac_of_hbm_peak"]}
                break
        if at95 is None:
            top = max(int(e_) for e_ in sweep)
            at95 = {"ef_search": None, "highest": {"ef_search": top, "recall_at_10": sweep[str(top)]["recall_at_10"],
                                                    "qps": sweep[str(top)]["qps"]},
                    "note": "recall@10 %.4f at ef_search %d (the reference's maximum is 1000): 0.95 is not reached on this data "
                            "with m %d / ef_construction %d" % (sweep[str(top)]["recall_at_10"], top, m, efc)}
    except Exception as e_:  # noqa: BLE001
        at95 = {"error": repr(e_)}
    out = {"workload": "HNSW vector_cosine_ops %d x %d f32, m %d, ef_construction %d, k %d (BASELINE configs[3])"
                       % (rows, dim, m, efc, k),
           "queries_in_flight": int(qd.shape[0]), "ef_search": sweep, "recall_0.95_at": at95,
           "recall_ground_truth": "exact float64 inner products over all rows, %d queries" % nq,
           "roofline_note": "scored_rows_GBps = element rows gathered x 6 KB / search seconds: a 6 GB mirror does not "
                            "fit the 256 MB MALL, so this is HBM traffic (gathers of whole rows)",
           "build_secs": build_s, "build": {"batches": built["batches"], "elements": built["nelements"],
                                            "pairs_scored": built["device_pairs"], "phase_secs": built["phase_secs"]}}
    if not args.no_cpu_baseline and keep is not None:
        from oracle import pyoracle as po
        ora = po.Oracle(native=True)
        if int((built["dup_of"] < 0).sum()) != built["nelements"]:
            raise RuntimeError("element bookkeeping: %d kept rows, %d elements" % (int((built["dup_of"] < 0).sum()), built["nelements"]))
        walk = po.HnswGraph.from_tuples(ora, po.OPS_COSINE, po.ORA_F32, host_rows, m, built["levels"], built["nbr_start"],
                                        built["nbr"], built["entry"])
        eh, dh = keep
        qh = q.cpu().numpy()
        bad, checked = [], 64
        t0 = time.perf_counter()
        for i in range(checked):
            wr, wd, _ = walk.search(qh[i], 100, k)
            # both sides report the index's FUNCTION 1 value (vector_negative_inner_product on normalised rows)
            why = topk_equiv(eh[i][eh[i] >= 0].tolist(), dh[i][:len(wr)], wr.tolist(), wd)
            if why:
                bad.append((i, why))
        cpu_s = (time.perf_counter() - t0) / checked
        out["parity"] = {"against": "oracle HnswSearchLayer / GetScanItems restatement walking the same GPU-built graph "
                                    "(ora_hnsw_import), ef_search 100", "checked_queries": checked, "mismatches": len(bad)}
        out["cpu_search_single_thread_qps"] = 1.0 / cpu_s
        if bad:
            failures.append("hnsw: %d of %d queries differ from the oracle's walk, first: %r" % (len(bad), checked, bad[0]))
        walk.close()
        # the CPU side of the build: the reference's PARALLEL in-memory build restated (ora_hnsw_build_parallel:
        # per-element locks, src/hnswbuild.c:366-480) on a bounded prefix of the same rows, every core the container
        # grants.  An insert's cost grows with the graph, so the rate of a 20 k-row graph FLATTERS the CPU at 1 M.
        try:
            threads, _, _ = cpu_threads(ora)
            sub = min(20000, int(host_rows.shape[0]))
            t0 = time.perf_counter()
            cg = po.HnswGraph(ora, po.OPS_COSINE, po.ORA_F32, np.ascontiguousarray(host_rows[:sub]), m=m,
                              ef_construction=efc, seed=1, threads=threads)
            cs = time.perf_counter() - t0
            cg.close()
            out["cpu_build_baseline"] = {
                "kind": "port", "rows": sub, "threads": threads, "secs": cs, "rows_per_sec": sub / cs,
                "gpu_rows_per_sec_at_full_size": rows / build_s,
                "note": "oracle parallel build of the first %d rows; per-insert cost grows with the graph (1 M rows "
                        "on 8 cores: 486 s, profiles/r03_hnsw_quality.md), so this rate is an upper bound for the "
                        "CPU at %d rows" % (sub, rows)}
        except Exception as e:  # noqa: BLE001
            out["cpu_build_baseline"] = {"error": repr(e)}
    mirror.close()
    return out


def roofline_record(stats, esize, dim, tname, kernel):
    """the list-scan kernel's physical roofline from the library's device-side accounting (HIP events on the launch
    stream): frac = row bytes actually streamed / kernel seconds / 8 TB/s"""
    launches = max(stats["scan_launches"], 1)
    algo_bytes = stats["scan_pairs"] * esize * dim
    stream_bytes = stats["scan_rows"] * esize * dim
    unique_bytes = stats["scan_unique_rows"] * esize * dim
    scan_s = stats["scan_ms"] / 1e3
    sgbps = stream_bytes / scan_s / 1e9 if scan_s > 0 else 0.0
    gbps = unique_bytes / scan_s / 1e9 if scan_s > 0 else 0.0   # every probed row once: the floor of any batched scan
    return {"kernel": kernel, "bound": "hbm", "achieved": gbps, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": gbps / HBM_PEAK_GBS, "traffic": None,
            "achieved_streamed": sgbps, "frac_streamed": sgbps / HBM_PEAK_GBS,
            "streamed_bytes_per_launch": stream_bytes / launches, "unique_bytes_per_launch": unique_bytes / launches,
            "passes": stream_bytes / unique_bytes if unique_bytes > 0 else None,
            "algorithmic_bytes_per_launch": algo_bytes / launches,
            "algorithmic_GBps": algo_bytes / scan_s / 1e9 if scan_s > 0 else 0.0,
            "avg_launch_ms": stats["scan_ms"] / launches, "launches": launches,
            "useful_tflops": stats["scan_pairs"] * dim * 2.0 / scan_s / 1e12 if scan_s > 0 else 0.0,
            "mfma_peak_tflops": 157.3 if tname == "f32" else 2500.0}


def run_workload(ctx, dev, name, args, failures, steps=10, parity_queries=64):
    """one of BASELINE's other IVFFlat configs end to end on this GPU: build, recall against float64, the timed
    batch loop with the scan kernel's roofline, parity of `parity_queries` queries with the CPU oracle"""
    n, dim, lists, probes, tname, oname = WORKLOADS[name]
    dtype = api.PGV_F32 if tname == "f32" else api.PGV_F16
    tdtype = torch.float32 if tname == "f32" else torch.float16
    ops = api.PGV_OPS_L2 if oname == "l2" else api.PGV_OPS_IP
    metric = api.PGV_L2SQ if oname == "l2" else api.PGV_NEG_IP
    esize = 4 if tname == "f32" else 2
    k, batch, pool = args.k, args.batch, 4
    components = max(lists // 4, 1)
    hard = name.startswith("hard")
    if hard:
        data, means = gen_hard(n, dim, components, args.seed + 50, dev)
    else:
        data, means = gen_mixture(n, dim, components, 0.1, args.seed + 50, dev)
    data = data.to(tdtype)
    ctx.set_profiling(True)
    ctx.reset_stats()
    centers, offsets, vectors, tids, iters, build_t, index = build_index(ctx, data, lists, args.seed, 1, 0, dtype, ops, metric)
    build_stats = ctx.stats()
    ctx.set_profiling(False)
    del data
    if hard:
        queries, _ = gen_hard(batch * pool, dim, components, args.seed + 150, dev, means=means)
    else:
        queries, _ = gen_mixture(batch * pool, dim, components, 0.1, args.seed + 150, dev, means=means)
    queries = queries.to(tdtype).view(pool, batch, dim)
    od = torch.empty((batch, k), device=dev, dtype=torch.float32)
    os_ = torch.empty((batch, k), device=dev, dtype=torch.int64)
    ot = torch.empty((batch, k), device=dev, dtype=torch.int64)
    rq = min(args.recall_queries, batch)
    rqueries = queries[1][:rq].contiguous()
    exact_d, _ = exact_topk_fp64(vectors, rqueries, k, metric)
    gd, _, _ = index.search_batch(rqueries, probes, k, want_tid=True)
    recall = recall_at_k(gd, exact_d, k)

    def step(j):
        index.search_batch(queries[j % pool], probes, k, want_tid=True, out=(od, os_, ot))
    for j in range(3):
        step(j)
    # the same steps once before any settling (what a cold measurement would print) ...
    s_unsettled = timed_steps(step, steps, warmup=0)
    # ... then untimed steps for --settle-ms like the headline's timed region (the float64 ground truth above leaves the
    # chip at whatever clocks a dense fp64 pass ends with; round 5's c2 line was measured without this)
    t_settle, settle_steps = time.perf_counter(), 0
    while args.settle_ms > 0 and (time.perf_counter() - t_settle) * 1e3 < args.settle_ms and settle_steps < 4096:
        for j in range(8):
            step(settle_steps + j)
        settle_steps += 8
        torch.cuda.synchronize()
    ctx.set_profiling(True)
    ctx.reset_stats()
    s = timed_steps(step, steps, warmup=0)
    stats = ctx.stats()
    ctx.set_profiling(False)
    what = ("overlapping Gaussian mixture (%d Zipf(%.1f)-weighted components, component sigma %.2f x the means' spread, "
            "per-dimension variance ~ j^-%.1f)" % (components, HARD["zipf"], HARD["sigma"], HARD["alpha"])) if hard else \
        "Gaussian mixture (%d components, sigma 0.1)" % components
    out = {"workload": "%s: IVFFlat %s_%s_ops %d x %d %s, lists=%d, probes=%d, k=%d, batch=%d, %s"
                       % (name, "vector" if tname == "f32" else "halfvec", oname, n, dim, tname, lists, probes, k, batch, what),
           "qps": batch / s, "ms_per_step": s * 1e3, "steps": steps, "settle_steps": settle_steps,
           "qps_unsettled": batch / s_unsettled, "recall_at_10": recall,
           "recall_ground_truth": "exact float64 brute force over all %d rows, %d queries" % (n, rq),
           "build_secs": build_t["total"], "build_phases_secs": build_t, "kmeans_iterations": iters,
           "scan_ms_per_step": stats["scan_ms"] / steps, "scan_redo_queries_per_step": stats["scan_redo_queries"] / steps,
           "scan_widened_queries_per_step": stats["scan_widened_queries"] / steps,
           "roofline": roofline_record(stats, esize, dim, tname, "mfma_scan_kernel (IVFFlat list scan)")}
    if oname == "l2":
        out["bound_statistical"] = bound_mode_run(ctx, step, steps, batch)
    if hard:
        # what the easy mixture cannot show: recall below 1, lists of unequal length, Lloyd running for dozens of
        # iterations, the completeness bounds with something to do
        sizes = (offsets[1:] - offsets[:-1]).double()
        out["list_rows_max_over_mean"] = float(sizes.max().item() / sizes.mean().item())
        out["list_rows_min"], out["list_rows_max"] = int(sizes.min().item()), int(sizes.max().item())
        out["build_assign"] = {"rows": build_stats["assign_rows"],
                               "rechecked_fraction": build_stats["assign_recheck_rows"] / max(build_stats["assign_rows"], 1),
                               "redone_fraction": build_stats["assign_redo_rows"] / max(build_stats["assign_rows"], 1)}
        sweep = {}
        for p in (1, 10, 32, 100):
            if p > lists:
                continue
            ctx.set_profiling(True)
            ctx.reset_stats()
            sp = timed_steps(lambda j: index.search_batch(queries[j % pool], p, k, want_tid=True, out=(od, os_, ot)), 5)
            st = ctx.stats()
            ctx.set_profiling(False)
            gdp, _, _ = index.search_batch(rqueries, p, k)
            sweep[str(p)] = {"qps": batch / sp, "ms_per_step": sp * 1e3, "recall_at_10": recall_at_k(gdp, exact_d, k),
                             "frac": roofline_record(st, esize, dim, tname, "")["frac"],
                             "passes": roofline_record(st, esize, dim, tname, "")["passes"],
                             "useful_tflops": roofline_record(st, esize, dim, tname, "")["useful_tflops"],
                             "redo_queries_per_step": st["scan_redo_queries"] / 7, "widened_queries_per_step": st["scan_widened_queries"] / 7}
        out["probes_sweep"] = sweep
        # the assignment of every row under both bounds, same centers (the build above ran under the default)
        ab = {}
        for mode in ("worst_case", "statistical"):
            ctx.set_bound(mode == "worst_case")
            ctx.set_profiling(True)
            ctx.reset_stats()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            api.assign(ctx, metric, dtype, dim, centers, vectors, want_dist=False)
            ctx.sync()
            torch.cuda.synchronize()
            st = ctx.stats()
            ab[mode] = {"secs": time.perf_counter() - t0, "rechecked_fraction": st["assign_recheck_rows"] / max(st["assign_rows"], 1),
                        "redone_fraction": st["assign_redo_rows"] / max(st["assign_rows"], 1)}
            ctx.set_profiling(False)
        ctx.set_bound(True)
        out["assign_bounds"] = ab
    if not args.no_cpu_baseline:
        from oracle import pyoracle as po
        ora = po.Oracle(native=True)
        ix = ora.index_struct(po.OPS_L2 if oname == "l2" else po.OPS_IP, po.ORA_F32 if tname == "f32" else po.ORA_F16,
                              centers.cpu().numpy(), offsets.cpu().numpy(), vectors.cpu().numpy(),
                              tids.cpu().numpy().astype(np.uint64))
        pq = queries[2][:parity_queries].contiguous()
        pd, _, pt = index.search_batch(pq, probes, k, want_tid=True)
        pd, pt, pqh = pd.cpu().numpy(), pt.cpu().numpy(), pq.cpu().numpy()
        bad = []
        t0 = time.perf_counter()
        for i in range(parity_queries):
            wt, wd = ora.search(ix, pqh[i], probes, k)
            why = topk_equiv(pt[i][:len(wt)].astype(np.uint64).tolist(), pd[i][:len(wt)], wt.tolist(), wd)
            if why:
                bad.append((i, why))
        out["parity"] = {"against": "CPU oracle, same index, same queries", "checked_queries": parity_queries,
                         "mismatches": len(bad)}
        out["cpu_single_thread_qps"] = parity_queries / (time.perf_counter() - t0)
        if bad:
            failures.append("%s: %d of %d queries differ from the oracle, first: %r" % (name, len(bad), parity_queries, bad[0]))
    index.close()
    return out


def run_full_config(ctx, dev, name, args, failures, steps=5, parity_queries=64, host_cap_bytes=24 << 30):
    """BASELINE configs[2] / configs[4] at their REAL shape on one GPU (10 M rows, lists 4096, probes 64; 61 GB of rows
    + the 61 GB heap-order copy of the build fit one 288 GB MI355X).  The rows exist only as seeded slabs of 2^17 rows
    that are regenerated whenever a pass needs them (sampling, the build, float64 ground truth, the oracle's rows):
      k-means  numSamples = 50 * lists = 204 800 (src/ivfbuild.c:446-455), k = 4096
      assign   every row, as the builder takes it (AddTupleToSort, src/ivfbuild.c:161-219)
      scan     1024-query batches, 64 probes (src/ivfscan.c:47-187)
      recall   against exact float64 brute force over all 10 M rows
      parity   >= 64 queries against the CPU oracle over the SAME index: the oracle ranks all 4096 centers itself and is
               handed the rows of every list it (or the GPU) probes -- lists nobody probes are never read by either side,
               so they stay on the device (61 GB of host copies would buy nothing); the parity queries are drawn from
               16 mixture components to keep that union under `host_cap_bytes`."""
    n, dim, lists, probes, tname, oname = WORKLOADS[name]
    dtype = api.PGV_F32 if tname == "f32" else api.PGV_F16
    tdtype = torch.float32 if tname == "f32" else torch.float16
    ops = api.PGV_OPS_L2 if oname == "l2" else api.PGV_OPS_IP
    metric = api.PGV_L2SQ if oname == "l2" else api.PGV_NEG_IP
    esize = 4 if tname == "f32" else 2
    k, batch, pool = args.k, args.batch, 3
    components = max(lists // 4, 1)
    seed = args.seed + 70
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    means = torch.rand((components, dim), generator=g, device=dev, dtype=torch.float32)
    nslabs = (n + SLAB - 1) // SLAB

    def slab(i):
        lo, hi = i * SLAB, min(n, (i + 1) * SLAB)
        rows, _ = gen_mixture(n, dim, components, 0.1, seed, dev, means=means, lo=lo, hi=hi)
        return lo, rows.to(tdtype)
    free0, total0 = torch.cuda.mem_get_info(dev)
    out = {"workload": "%s: IVFFlat %s_%s_ops %d x %d %s, lists=%d, probes=%d, k=%d, batch=%d, Gaussian mixture (%d "
                       "components, sigma 0.1), ONE GPU" % (name, "vector" if tname == "f32" else "halfvec", oname, n, dim,
                                                            tname, lists, probes, k, batch, components),
           "rows": n, "dim": dim, "lists": lists, "probes": probes}
    t_all = time.perf_counter()
    # ---- sampling pass
    ns = min(max(50 * lists, 10000), n)
    g.manual_seed(seed + 1)
    pick = torch.sort(torch.randperm(n, generator=g, device=dev)[:ns]).values
    samples = torch.empty((ns, dim), device=dev, dtype=tdtype)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    bounds = torch.searchsorted(pick, torch.arange(0, nslabs + 1, device=dev) * SLAB).tolist()
    for i in range(nslabs):
        a, b = bounds[i], bounds[i + 1]
        if b > a:
            lo, rows = slab(i)
            samples[a:b] = rows[pick[a:b] - lo]
    if ops != api.PGV_OPS_L2:
        s32 = samples.float()
        samples = (s32 / s32.norm(dim=1, keepdim=True).clamp_min(1e-30)).to(tdtype).contiguous()
    torch.cuda.synchronize()
    out["generate_pass_secs"] = time.perf_counter() - t0
    # ---- k-means
    t0 = time.perf_counter()
    centers, _, iters = api.kmeans(ctx, ops, dtype, dim, samples, lists, api.make_rng(seed=seed + 2), want_closest=False)
    ctx.sync()
    torch.cuda.synchronize()
    out["kmeans_secs"], out["kmeans_iterations"], out["kmeans_samples"] = time.perf_counter() - t0, int(iters), ns
    log("%s: k-means k=%d on %d samples: %.2f s, %d iterations" % (name, lists, ns, out["kmeans_secs"], iters))
    del samples
    # ---- the heap rows into the builder, assigned as they arrive
    ctx.set_profiling(True)
    ctx.reset_stats()
    b = api.IvfBuilder(ctx, metric, dtype, dim, centers, expected_rows=n)
    assign_s = 0.0
    for i in range(nslabs):
        _, rows = slab(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        b.add(rows)
        ctx.sync()
        assign_s += time.perf_counter() - t0
        del rows
    bst = ctx.stats()
    out["assign_secs"] = assign_s
    out["assign_tflops"] = 2.0 * n * lists * dim / assign_s / 1e12
    out["assign_rechecked_fraction"] = bst["assign_recheck_rows"] / bst["assign_rows"] if bst["assign_rows"] else None
    t0 = time.perf_counter()
    index, offsets_h, lists_h = b.finish(want_lists=True)
    ctx.sync()
    out["layout_secs"] = time.perf_counter() - t0
    b.close()
    torch.cuda.empty_cache()
    out["build_secs"] = out["kmeans_secs"] + out["assign_secs"] + out["layout_secs"]
    sizes = np.diff(offsets_h)
    out["list_rows"] = {"min": int(sizes.min()), "mean": float(sizes.mean()), "max": int(sizes.max())}
    free1, _ = torch.cuda.mem_get_info(dev)
    out["hbm_bytes_in_use_after_build"] = int(total0 - free1)
    log("%s: assign %.2f s (%.0f TFLOP/s), layout %.2f s" % (name, assign_s, out["assign_tflops"], out["layout_secs"]))
    # ---- queries
    queries, _ = gen_mixture(batch * pool, dim, components, 0.1, seed + 150, dev, means=means)
    queries = queries.to(tdtype).view(pool, batch, dim)
    od = torch.empty((batch, k), device=dev, dtype=torch.float32)
    os_ = torch.empty((batch, k), device=dev, dtype=torch.int64)
    ot = torch.empty((batch, k), device=dev, dtype=torch.int64)

    def step(j):
        index.search_batch(queries[j % pool], probes, k, want_tid=True, out=(od, os_, ot))
    for j in range(2):
        step(j)
    ctx.reset_stats()
    s = timed_steps(step, steps, warmup=0)
    stats = ctx.stats()
    ctx.set_profiling(False)
    out.update({"qps": batch / s, "ms_per_step": s * 1e3, "steps": steps,
                "scan_ms_per_step": stats["scan_ms"] / steps, "center_rank_ms": stats["aux_ms"] / steps,
                "scan_redo_queries_per_step": stats["scan_redo_queries"] / steps,
                "roofline": roofline_record(stats, esize, dim, tname, "mfma_scan_kernel (IVFFlat list scan)")})
    log("%s: %.0f QPS, %.2f ms/step (scan %.2f, center rank %.3f), frac %.2f" % (
        name, out["qps"], out["ms_per_step"], out["scan_ms_per_step"], out["center_rank_ms"], out["roofline"]["frac"]))
    # ---- parity queries: from 16 components, so that the lists they probe fit the host cap
    rq = min(args.recall_queries, batch)
    rqueries = queries[1][:rq].contiguous()
    g.manual_seed(seed + 5)
    few = means[torch.randperm(components, generator=g, device=dev)[:16]]
    pq, _ = gen_mixture(parity_queries, dim, 16, 0.1, seed + 170, dev, means=few)
    pq = pq.to(tdtype).contiguous()
    gd, _, _ = index.search_batch(rqueries, probes, k, want_tid=True)
    pd, _, pt = index.search_batch(pq, probes, k, want_tid=True)
    ctx.sync()
    want_lists = None
    ora = ix_rank = None
    if not args.no_cpu_baseline:
        from oracle import pyoracle as po
        ora = po.Oracle(native=True)
        o_ops, o_dt = (po.OPS_L2 if oname == "l2" else po.OPS_IP), (po.ORA_F32 if tname == "f32" else po.ORA_F16)
        centers_h = centers.cpu().numpy()
        pqh = pq.cpu().numpy()
        ix_rank = ora.index_struct(o_ops, o_dt, centers_h, np.zeros(lists + 1, np.int64), np.zeros((1, dim), centers_h.dtype))
        probed = set()
        for i in range(parity_queries):
            probed.update(int(x) for x in ora.get_scan_lists(ix_rank, pqh[i], probes)[0])
        gl = index.rank_lists(pq, probes, want_dist=False)
        gl = gl[0] if isinstance(gl, tuple) else gl
        probed.update(int(x) for x in np.asarray(gl.cpu() if api._is_torch(gl) else gl).ravel())
        want_lists = np.array(sorted(probed), dtype=np.int64)
        need = int(sizes[want_lists].sum()) * dim * esize
        out["parity_lists_on_host"] = {"lists": len(want_lists), "bytes": need}
        if need > host_cap_bytes:
            failures.append("%s: the parity queries probe %d lists = %.1f GB, over the host cap" % (name, len(want_lists), need / 2**30))
            want_lists = None
    # ---- one more pass over the slabs: float64 ground truth for recall, and the oracle's rows
    is_wanted = None
    if want_lists is not None:
        is_wanted = torch.zeros(lists, dtype=torch.bool, device=dev)
        is_wanted[torch.from_numpy(want_lists).to(dev)] = True
    lists_dev = torch.from_numpy(lists_h.astype(np.int64)).to(dev)
    q64 = rqueries.double()
    qq = (q64 * q64).sum(1)[:, None]
    best = torch.full((rq, k), float("inf"), dtype=torch.float64, device=dev)
    host_rows, host_ids, host_list = [], [], []
    t0 = time.perf_counter()
    for i in range(nslabs):
        lo, rows = slab(i)
        for c0 in range(0, rows.shape[0], 1 << 15):
            v = rows[c0:c0 + (1 << 15)].double()
            d = (qq + (v * v).sum(1)[None, :] - 2.0 * (q64 @ v.T)) if metric == api.PGV_L2SQ else -(q64 @ v.T)
            best = torch.topk(torch.cat([best, d], dim=1), k, dim=1, largest=False).values
        if is_wanted is not None:
            sl = lists_dev[lo:lo + rows.shape[0]]
            m = torch.nonzero(is_wanted[sl]).squeeze(1)
            if m.numel():
                host_rows.append(rows[m].cpu().numpy())
                host_ids.append((m + lo).cpu().numpy())
                host_list.append(sl[m].cpu().numpy())
        del rows
    out["ground_truth_pass_secs"] = time.perf_counter() - t0
    out["recall_at_10"] = recall_at_k(gd, best, k)
    out["recall_ground_truth"] = "exact float64 brute force over all %d rows, %d queries" % (n, rq)
    log("%s: recall@10 %.4f" % (name, out["recall_at_10"]))
    if want_lists is not None:
        hl = np.concatenate(host_list)
        order = np.argsort(hl, kind="stable")           # list-major, heap order inside a list: the mirror's own order
        vec = np.concatenate(host_rows)[order]
        ids = np.concatenate(host_ids)[order].astype(np.uint64)
        del host_rows
        red_off = np.zeros(lists + 1, np.int64)
        np.cumsum(np.bincount(hl, minlength=lists), out=red_off[1:])
        ix = ora.index_struct(o_ops, o_dt, centers_h, red_off, vec, ids)
        threads = cpu_threads(ora)[0]
        t0 = time.perf_counter()
        answers, _, _ = ora.bench_search(ix, pqh, probes, k, threads, 0.0)
        cpu_s = time.perf_counter() - t0
        pdh, pth = pd.cpu().numpy(), pt.cpu().numpy()
        bad = []
        for i in range(parity_queries):
            wt, wd = answers[i]
            why = topk_equiv(pth[i][:len(wt)].astype(np.uint64).tolist(), pdh[i][:len(wt)], wt.tolist(), wd)
            if why:
                bad.append((i, why))
        out["parity"] = {"against": "CPU oracle over the same centers and the rows of every list either side probes "
                                    "(%d lists, %.1f GB on the host)" % (len(want_lists), vec.nbytes / 2**30),
                         "checked_queries": parity_queries, "mismatches": len(bad)}
        out["cpu_qps_all_threads"] = parity_queries / cpu_s
        out["cpu_threads"] = threads
        if bad:
            failures.append("%s: %d of %d queries differ from the oracle, first: %r" % (name, len(bad), parity_queries, bad[0]))
        log("%s: parity %d mismatches of %d (oracle %.1f QPS on %d threads)" % (name, len(bad), parity_queries,
                                                                               out["cpu_qps_all_threads"], threads))
    out["total_secs"] = time.perf_counter() - t_all
    index.close()
    return out


def bound_mode_run(ctx, step, steps, batch):
    """the same timed loop with PGV_BOUND_STATISTICAL (include/pgv_hip.h; the default is the deterministic bound): QPS and
    queries redone exactly per step"""
    ctx.set_bound(False)
    try:
        for j in range(2):
            step(j)
        ctx.set_profiling(True)
        ctx.reset_stats()
        s = timed_steps(step, steps, warmup=0)
        st = ctx.stats()
    finally:
        ctx.set_profiling(False)
        ctx.set_bound(True)
    return {"qps": batch / s, "ms_per_step": s * 1e3, "scan_redo_queries_per_step": st["scan_redo_queries"] / steps,
            "scan_widened_queries_per_step": st["scan_widened_queries"] / steps,
            "bound": "8 sqrt(d + 4) 2^-24 (|q| + |x|max)^2, probabilistic (pgv_ctx_set_bound(PGV_BOUND_STATISTICAL))"}


def exact_scan_section(ctx, dev, args, failures):
    """BASELINE configs[0]: the index-less `ORDER BY embedding <-> q LIMIT 10` over 10 k x 128 for 100 queries --
    pgv_exact_topk against the oracle's per-row l2 calls + sort (src/vector.c:579-589), and a 1 M x 1536 case for
    the kernel's rate"""
    out = {}
    g = torch.Generator(device=dev)
    g.manual_seed(args.seed + 61)
    for label, n, dim, nq in (("config1_10k_x_128", 10_000, 128, 100), ("1M_x_1536", 1_000_000, 1536, 1024)):
        rows = torch.rand((n, dim), generator=g, device=dev)
        queries = torch.rand((nq, dim), generator=g, device=dev)
        od = torch.empty((nq, args.k), device=dev, dtype=torch.float32)
        oi = torch.empty((nq, args.k), device=dev, dtype=torch.int64)
        s = timed_steps(lambda j: api.exact_topk(ctx, api.PGV_L2SQ, api.PGV_F32, dim, queries, rows, args.k, out=(od, oi)), 10)
        rec = {"rows": n, "dim": dim, "queries": nq, "k": args.k, "ms_per_batch": s * 1e3, "qps": nq / s,
               "rows_GBps": n * dim * 4 * ((nq + 31) // 32) / s / 1e9,
               "path": "pgv_exact_topk: MFMA expansion + exact tail, data resident in HBM"}
        if n <= 100_000:
            qh, rh = queries.cpu().numpy(), rows.cpu().numpy()
            sh = timed_steps(lambda j: api.exact_topk(ctx, api.PGV_L2SQ, api.PGV_F32, dim, qh, rh, args.k), 5)
            rec["host_buffers_ms_per_batch"] = sh * 1e3
            if not args.no_cpu_baseline:
                from oracle import pyoracle as po
                ora = po.Oracle(native=True)
                ix = ora.index_struct(po.OPS_L2, po.ORA_F32, rh[:1], np.array([0, n], dtype=np.int64), rh,
                                      np.arange(n, dtype=np.uint64))
                gd, gi = od.cpu().numpy(), oi.cpu().numpy()
                bad = []
                t0 = time.perf_counter()
                for i in range(nq):
                    wt, wd = ora.search(ix, qh[i], 1, args.k)
                    why = topk_equiv(gi[i].astype(np.uint64).tolist(), gd[i], wt.tolist(), wd)
                    if why:
                        bad.append((i, why))
                cpu_s = time.perf_counter() - t0
                rec["parity"] = {"against": "oracle: one l2 call per row + ascending sort", "checked_queries": nq,
                                 "mismatches": len(bad)}
                rec["cpu_baseline"] = {"value": nq / cpu_s, "unit": "queries/s", "cores": 1, "kind": "port",
                                       "sample": "the same %d queries, one thread" % nq}
                if bad:
                    failures.append("exact scan: %d of %d queries differ from the oracle, first: %r" % (len(bad), nq, bad[0]))
        out[label] = rec
        del rows, queries
    return out


def timed_steps(fn, steps, warmup=2):
    for i in range(warmup):
        fn(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        fn(warmup + i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


# ---------------------------------------------------------------------------------------------------------------------
# The line the driver parses.  Round 4's line had grown to ~21 KB (sweeps, backends, notes) and the driver's record of it
# came back `parsed: null`.  What goes to fd 1 now is ONE compact object, capped at LINE_CAP bytes: the contract fields,
# roofline, cpu_baseline, parity, one summary per other config.  Everything measured (the former line, unabridged) goes
# to bench_detail.json beside this script (and to gpurun_out/ when that exists) -- never to stdout.
LINE_CAP = 4096
DETAIL_NAME = "bench_detail.json"


def _r(x, sig=5):
    """floats to `sig` significant digits (bytes on the line are the budget); everything else untouched"""
    if isinstance(x, bool) or x is None:
        return x
    if isinstance(x, float):
        if x != x or x in (float("inf"), float("-inf")):
            return None
        return float("%.*g" % (sig, x))
    return x


def _clean(x):
    if isinstance(x, dict):
        return {str(k): _clean(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_clean(v) for v in x]
    return _r(x)


def _pick(d, keys):
    return {k: _r(d[k]) for k in keys if isinstance(d, dict) and k in d}


def compact_line(full):
    """the <= LINE_CAP-byte object of the contract, cut from the full record `full` (which is left untouched)"""
    line = _pick(full, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                        "scaling", "vs_baseline", "dtype", "data"))
    cfg = full.get("config", {})
    line["config"] = _pick(cfg, ("workload", "rows", "dim", "lists", "probes", "k", "batch_per_gpu", "settle_steps"))
    if isinstance(line["config"].get("workload"), str):
        line["config"]["workload"] = line["config"]["workload"][:120]
    if full.get("n_gpus", 1) > 1:
        line["config"]["parallelism"] = str(cfg.get("parallelism", ""))[:80]
        mg = full.get("multi_gpu", {})
        line["multi_gpu"] = _pick(mg, ("comm_size", "pgv_comm_size", "backend", "communicator", "launcher",
                                       "kmeans_allreduce_bytes_per_iteration", "search_allgather_bytes_per_step", "placement",
                                       "rows_per_rank_min", "rows_per_rank_mean", "rows_per_rank_max",
                                       "slowest_rank_scan_ms", "measured_on"))
    ro = full.get("roofline", {})
    line["roofline"] = _pick(ro, ("bound", "achieved", "peak", "unit", "frac", "frac_streamed", "traffic", "avg_launch_ms",
                                  "launches", "passes", "unique_bytes_per_launch", "streamed_bytes_per_launch",
                                  "algorithmic_bytes_per_launch"))
    line["roofline"]["kernel"] = str(ro.get("kernel", "")).split(" ")[0]
    cb = full.get("cpu_baseline")
    if isinstance(cb, dict):
        line["cpu_baseline"] = _pick(cb, ("value", "unit", "cores", "kind", "single_thread_qps", "isa", "port_value",
                                          "port_single_thread_qps"))
        if isinstance(cb.get("answers_against_oracle"), dict):
            line["cpu_baseline"]["answers_vs_oracle_mismatches"] = cb["answers_against_oracle"].get("mismatches")
        line["cpu_baseline"]["sample"] = str(cb.get("sample", ""))[:100]
        if isinstance(cb.get("page_image"), dict):
            line["cpu_baseline"]["page_image_qps"] = _r(cb["page_image"].get("value"))
    pa = full.get("parity")
    if isinstance(pa, dict):
        line["parity"] = {"mismatches": pa.get("mismatches"), "checked": full.get("parity_checked_queries")}
        if "page_built_index_mismatches" in pa:
            line["parity"]["page_built_index_mismatches"] = pa["page_built_index_mismatches"]
    for key in ("value_unsettled", "recall_at_10", "build_secs", "kmeans_iterations", "center_rank_ms_per_step", "scan_ms_per_step", "overlap_lanes"):
        if key in full:
            line[key] = _r(full[key])
    oc = {}
    for name, c in (full.get("other_configs") or {}).items():
        if not isinstance(c, dict):
            continue
        o = {"qps": _r(c.get("qps")), "recall": _r(c.get("recall_at_10")),
             "frac": _r((c.get("roofline") or {}).get("frac")),
             "parity_mismatches": (c.get("parity") or {}).get("mismatches"),
             "parity_checked": (c.get("parity") or {}).get("checked_queries")}
        for extra in ("build_secs", "rows", "kmeans_secs", "kmeans_iterations", "assign_secs", "center_rank_ms"):
            if extra in c:
                o[extra] = _r(c[extra])
        oc[name] = o
    for name in ("c3", "c5"):     # configs[2] / configs[4] at their real 10 M-row shape, one GPU
        c = full.get("full_" + name)
        if isinstance(c, dict):
            oc[name] = {"qps": _r(c.get("qps")), "recall": _r(c.get("recall_at_10")),
                        "frac": _r((c.get("roofline") or {}).get("frac")),
                        "parity_mismatches": (c.get("parity") or {}).get("mismatches"),
                        "parity_checked": (c.get("parity") or {}).get("checked_queries"),
                        "rows": c.get("rows"), "build_secs": _r(c.get("build_secs")), "kmeans_secs": _r(c.get("kmeans_secs")),
                        "kmeans_iterations": c.get("kmeans_iterations"), "assign_secs": _r(c.get("assign_secs")),
                        "center_rank_ms": _r(c.get("center_rank_ms"))}
    hd = full.get("hard")
    if isinstance(hd, dict) and "qps" in hd:
        oc["hard"] = {"qps": _r(hd.get("qps")), "recall": _r(hd.get("recall_at_10"), 4), "frac": _r((hd.get("roofline") or {}).get("frac"), 3),
                      "parity_mismatches": (hd.get("parity") or {}).get("mismatches"),
                      "parity_checked": (hd.get("parity") or {}).get("checked_queries"),
                      "build_secs": _r(hd.get("build_secs"), 3), "kmeans_iterations": hd.get("kmeans_iterations"),
                      "lists_max_over_mean": _r(hd.get("list_rows_max_over_mean"), 3),
                      "recall_by_probes": {p: _r(v.get("recall_at_10"), 3) for p, v in (hd.get("probes_sweep") or {}).items()},
                      "qps_by_probes": {p: _r(v.get("qps"), 3) for p, v in (hd.get("probes_sweep") or {}).items()}}
    hn = full.get("hnsw")
    if isinstance(hn, dict):
        ef = (hn.get("ef_search") or {}).get("100") or {}
        a95 = hn.get("recall_0.95_at") or {}
        oc["c4_hnsw"] = {"qps": _r(ef.get("qps")), "recall": _r(ef.get("recall_at_10")), "frac": _r(ef.get("frac_of_hbm_peak")),
                         "ef_for_recall_0.95": a95.get("ef_search"), "qps_at_recall_0.95": _r(a95.get("qps")),
                         "highest_ef": _pick(a95.get("highest") or {}, ("ef_search", "recall_at_10", "qps")),
                         "parity_mismatches": (hn.get("parity") or {}).get("mismatches"),
                         "parity_checked": (hn.get("parity") or {}).get("checked_queries"),
                         "build_secs": _r(hn.get("build_secs"))}
    ex = (full.get("exact_scan") or {}).get("config1_10k_x_128")
    if isinstance(ex, dict):
        oc["c1_exact"] = {"qps": _r(ex.get("qps")), "parity_mismatches": (ex.get("parity") or {}).get("mismatches"),
                          "parity_checked": (ex.get("parity") or {}).get("checked_queries")}
    if oc:
        line["other_configs"] = oc
    bs = full.get("batch_sweep")
    if isinstance(bs, dict):
        line["batch_ms"] = {b: _r(v.get("ms_per_step", (v.get("latency_us_p50") or 0) / 1e3), 4)
                            for b, v in bs.items() if isinstance(v, dict)}
    be = full.get("concurrent_backends")
    if isinstance(be, dict):
        s = {}
        for key, short in (("single_query_processes", "own_ctx_procs_qps"), ("pooled_single_query_processes", "pooled_procs_qps")):
            rows = be.get(key)
            if isinstance(rows, dict):
                s[short] = {n: _r(v.get("qps"), 4) for n, v in rows.items() if isinstance(v, dict) and "qps" in v}
        top = (be.get("pooled_single_query_processes") or {}).get("256")
        if isinstance(top, dict):
            s["pooled_256_p50_us"] = _r(top.get("latency_us_p50"), 4)
        top = (be.get("pooled_single_query_processes") or {}).get("1024")
        if isinstance(top, dict):
            s["pooled_1024_p50_us"] = _r(top.get("latency_us_p50"), 4)
        if s:
            line["backends"] = s
    ob = full.get("overlapped_batches")
    if isinstance(ob, dict):
        line["overlapped_batches"] = _pick(ob, ("lanes", "qps", "ms_per_step", "scan_ms_per_launch"))
    if "build_secs_pages" in full:
        line["build_secs_pages"] = _r(full["build_secs_pages"])
    cbb = full.get("cpu_build_baseline")
    if isinstance(cbb, dict):
        line["cpu_build_secs_all_threads_extrapolated"] = _r(cbb.get("build_secs_extrapolated_all_threads"))
    line["bench_wall_secs"] = _r(full.get("bench_wall_secs"))
    line["detail"] = DETAIL_NAME
    fails = full.get("failures")
    if fails:
        line["failures"] = [str(f)[:160] for f in fails[:6]]
        if len(fails) > 6:
            line["failures"].append("... %d more in %s" % (len(fails) - 6, DETAIL_NAME))
    # the cap is a contract: shed the optional summaries, least important first, until the line fits
    for victim in ("backends", "batch_ms", "cpu_build_secs_all_threads_extrapolated", "build_secs_pages", "overlapped_batches", "other_configs",
                   "failures"):
        if len(json.dumps(line, default=str)) < LINE_CAP:
            break
        if victim == "failures" and "failures" in line:
            line["failures"] = [str(f)[:60] for f in line["failures"][:3]]
        else:
            line.pop(victim, None)
    return line


def write_detail(full):
    """the unabridged record, beside bench.py and (when the GPU box's scratch dir exists) under gpurun_out/"""
    here = os.path.dirname(os.path.abspath(__file__))
    wrote = []
    for d in (here, os.path.join(here, "gpurun_out")):
        if not os.path.isdir(d):
            continue
        try:
            p = os.path.join(d, DETAIL_NAME)
            with open(p + ".tmp", "w") as f:
                json.dump(full, f, default=str, indent=1)
            os.replace(p + ".tmp", p)
            wrote.append(p)
        except OSError:
            pass
    return wrote


def emit_line(fd, full):
    """detail to its file(s), the compact line to stderr for the log, and LAST (and alone) on fd 1"""
    wrote = write_detail(full)
    text = json.dumps(_clean(compact_line(full)), default=str, allow_nan=False)
    assert "\n" not in text
    log("detail record: %s" % (", ".join(wrote) or "could not be written"))
    log("line (%d bytes): %s" % (len(text), text))
    os.write(fd, (text + "\n").encode())


# ---------------------------------------------------------------------------------------------------------------------
# How the ONE JSON line is kept safe (round 3's driver run lost everything behind a section that hung):
#   1. the parent process does only what the line cannot do without, in this order: build, recall, the timed steps,
#      oracle parity + cpu_baseline.  From there on the line exists and is complete as far as the contract goes.
#   2. everything else runs in CHILD processes of this same script (`--section NAME`), one after the other, each in a
#      session of its own with a budget of its own; a child that is not back in time is killed with its whole process
#      group and the next one starts.  A child writes what it has measured to a file after every step, so a cut
#      section still contributes the rows it finished.  Order: other_configs + exact_scan, hnsw, build (pages, CPU
#      build baseline), PMC traffic, sweeps, and the sections that start backends LAST.
#   3. a failed section is a `failures` entry and the exit code is 2 -- with the full line printed; a section that was cut
#      or not started for lack of time is a `failures` entry ("budget: ...") too, but the exit code stays 0: incomplete,
#      not wrong.
#   4. the watchdog is the last resort for the parent itself: it prints the line as far as it has got and exits 2 (3
#      without a line).
WATCH = {"line": None, "fd": None, "rank": 0, "section": "setup", "done": False}

SECTION_BUDGET_S = {"configs": 110, "hard": 80, "hnsw": 100, "build": 110, "sweeps": 100, "backends": 100, "c3full": 90, "c5full": 90}
SECTION_ORDER = ("configs", "hard", "hnsw", "build", "traffic", "c3full", "c5full", "sweeps", "backends")


def watchdog(deadline_s):
    t_end = time.perf_counter() + deadline_s
    while time.perf_counter() < t_end:
        if WATCH["done"]:
            return
        time.sleep(0.5)
    if WATCH["done"]:
        return
    line = WATCH["line"]
    if line is not None and WATCH["rank"] == 0:
        for _ in range(5):
            try:
                snap = dict(line)
                snap.setdefault("failures", [])
                snap["failures"] = list(snap["failures"]) + [
                    "watchdog: '%s' had not returned after %d s; the line ends here" % (WATCH["section"], deadline_s)]
                snap["bench_wall_secs"] = deadline_s
                emit_line(WATCH["fd"], snap)
                break
            except Exception:  # noqa: BLE001  (the dict was being written to: try again)
                time.sleep(0.05)
    # a cut line is a failed run, like any other `failures` entry (other ranks just leave)
    os._exit((2 if WATCH["rank"] == 0 else 0) if line is not None else 3)


class SectionOut:
    """what a section child has measured so far, rewritten (atomically) after every step"""

    def __init__(self, path):
        self.path = path
        self.data = {"failures": [], "_at": "start"}

    def flush(self):
        if not self.path:
            return
        tmp = self.path + ".tmp"
        with open(tmp, "w") as f:
            json.dump(self.data, f, default=str)
        os.replace(tmp, self.path)

    def at(self, what):
        self.data["_at"] = what
        log("[%s] %s" % (time.strftime("%H:%M:%S"), what))
        self.flush()

    def put(self, key, value):
        self.data[key] = value
        self.flush()


def run_group(cmd, budget_s, env=None, cwd=None, stdout=None):
    """a child in a session of its own; not back after budget_s: SIGKILL to its whole process group, at most 5 s of
    waiting for the corpse (a process stuck in the driver is left to init).  Returns (rc or None, timed_out, secs)."""
    import signal
    t0 = time.perf_counter()
    p = subprocess.Popen(cmd, stdin=subprocess.DEVNULL, stdout=stdout if stdout is not None else 2, stderr=2,
                         start_new_session=True, env=env, cwd=cwd)
    timed_out = False
    try:
        rc = p.wait(timeout=budget_s)
    except subprocess.TimeoutExpired:
        timed_out = True
        try:
            os.killpg(p.pid, signal.SIGKILL)
        except OSError:
            pass
        try:
            rc = p.wait(timeout=5)
        except subprocess.TimeoutExpired:
            rc = None
    else:
        try:
            os.killpg(p.pid, signal.SIGKILL)   # stragglers of a child that is itself gone (backend processes)
        except OSError:
            pass
    return rc, timed_out, time.perf_counter() - t0


def section_cmd(name, args, path):
    cmd = [sys.executable, os.path.abspath(__file__), "--section", name, "--section-out", path, "--workload", args.workload,
           "--batch", str(args.batch), "--k", str(args.k), "--seed", str(args.seed), "--recall-queries", str(args.recall_queries),
           "--watchdog-secs", "0"]
    if args.probes:
        cmd += ["--probes", str(args.probes)]
    if args.no_cpu_baseline:
        cmd.append("--no-cpu-baseline")
    if args.all_process_rows:
        cmd.append("--all-process-rows")
    return cmd


def run_section(name, args, budget_s, line, failures):
    """one optional section in a child of this script; merges what it wrote into the line"""
    fd, path = tempfile.mkstemp(prefix="pgv_section_%s_" % name, suffix=".json", dir="/tmp")
    os.close(fd)
    os.unlink(path)
    cmd = section_cmd(name, args, path)
    log("[%s] section %s: start (budget %d s)" % (time.strftime("%H:%M:%S"), name, budget_s))
    rc, timed_out, secs = run_group(cmd, budget_s)
    data = {}
    try:
        with open(path) as f:
            data = json.load(f)
    except Exception:  # noqa: BLE001
        pass
    for junk in (path, path + ".tmp"):
        try:
            os.unlink(junk)
        except OSError:
            pass
    for key, val in data.items():
        if key not in ("failures", "_at"):
            line[key] = val
    failures.extend("%s: %s" % (name, f) for f in data.get("failures", []))
    line.setdefault("sections", {})[name] = {"secs": secs, "rc": rc, "timed_out": timed_out, "budget_secs": budget_s}
    if timed_out:
        failures.append("budget: section %s: not back after %d s, killed (it had reached: %s)" % (name, budget_s, data.get("_at", "nothing")))
    elif rc != 0:
        failures.append("section %s: exit code %r (it had reached: %s)" % (name, rc, data.get("_at", "nothing")))
    log("[%s] section %s: %.1f s, rc %r%s" % (time.strftime("%H:%M:%S"), name, secs, rc, ", CUT" if timed_out else ""))


def live_traffic(args, scan_ms, budget_s=110):
    """HBM bytes per list-scan launch, measured NOW: two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE;
    never combined, MI355X_MICROARCH.md) of this same script on a short run.  Launches of the scan
    kernel whose duration is within 35 % of this run's average are the timed list scans."""
    exe = shutil.which("rocprofv3")
    if not exe:
        return None, "rocprofv3 not on PATH"
    out = {}
    base = [sys.executable, os.path.abspath(__file__), "--workload", args.workload, "--batch", str(args.batch),
            "--steps", "4", "--warmup", "1", "--seed", str(args.seed), "--child", "--watchdog-secs", "0",
            "--overlap", str(args.overlap)]
    if args.probes:
        base += ["--probes", str(args.probes)]
    env = dict(os.environ, TMPDIR="/tmp")
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="pgv_pmc_", dir="/tmp")
        try:
            log("[%s] traffic: rocprofv3 --pmc %s" % (time.strftime("%H:%M:%S"), counter))
            with open(os.path.join(d, "stdout.txt"), "w") as so:
                rc, timed_out, _ = run_group([exe, "--pmc", counter, "--kernel-trace", "-d", d, "-o", "pmc", "--output-format",
                                              "csv", "--"] + base, budget_s, env=env, cwd="/tmp", stdout=so)
            if timed_out:
                return None, "%s pass not back after %d s (killed)" % (counter, budget_s)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if rc != 0 or not files:
                return None, "%s pass failed (rc %r)" % (counter, rc)
            import csv
            vals = []
            for row in csv.DictReader(open(files[0])):
                if row.get("Counter_Name") != counter:
                    continue
                name = row["Kernel_Name"]
                if "tile_scan_kernel" not in name and "scan_kernel" not in name:
                    continue
                dur = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6
                if abs(dur - scan_ms) <= 0.35 * scan_ms:
                    vals.append(float(row["Counter_Value"]))
            if not vals:
                return None, "no %s rows matched the scan kernel" % counter
            out[counter] = (sum(vals) / len(vals), len(vals))
        except Exception as e:  # profiling must never sink the number
            return None, "%s pass: %r" % (counter, e)
        finally:
            shutil.rmtree(d, ignore_errors=True)
    # rocprofv3 reports both in KB; wide coalesced reads are tallied at half their size on gfx950
    traffic = 2.0 * out["FETCH_SIZE"][0] * 1024.0 + out["WRITE_SIZE"][0] * 1024.0
    return traffic, "live rocprofv3 --pmc passes of this script: 2 x FETCH_SIZE (%d launches) + WRITE_SIZE (%d)" % (
        out["FETCH_SIZE"][1], out["WRITE_SIZE"][1])


class Headline:
    """the headline workload resident on one GPU: data, the built index, the query pool"""


def headline_setup(args, dev, ctx, world=1, rank=0, comm=None, keep_host_rows=False, sharded=False):
    H = Headline()
    H.n, H.dim, H.lists, H.probes, H.tname, H.oname = WORKLOADS[args.workload]
    H.dtype = api.PGV_F32 if H.tname == "f32" else api.PGV_F16
    H.tdtype = torch.float32 if H.tname == "f32" else torch.float16
    H.ops = api.PGV_OPS_L2 if H.oname == "l2" else api.PGV_OPS_IP
    H.metric = api.PGV_L2SQ if H.oname == "l2" else api.PGV_NEG_IP
    H.esize = 4 if H.tname == "f32" else 2
    if args.probes:
        H.probes = args.probes
    H.k = args.k
    H.components = max(H.lists // 4, 1)
    row_lo, row_hi = sharding.row_shard(H.n, rank, world)
    H.hard = args.workload.startswith("hard")
    if H.hard:
        data, H.means = gen_hard(H.n, H.dim, H.components, args.seed, dev, lo=row_lo, hi=row_hi)
    else:
        data, H.means = gen_mixture(H.n, H.dim, H.components, 0.1, args.seed, dev, lo=row_lo, hi=row_hi)
    data = data.to(H.tdtype)
    log("data: rows [%d, %d) of %d x %d %s generated" % (row_lo, row_hi, H.n, H.dim, H.tname))
    ctx.set_profiling(True)
    ctx.reset_stats()
    H.centers, H.offsets, H.vectors, H.tids, H.iters, H.build_t, H.index = build_index(
        ctx, data, H.lists, args.seed, world, rank, H.dtype, H.ops, H.metric, comm, row_lo=row_lo, n_global=H.n, sharded=sharded)
    H.build_stats = ctx.stats()
    ctx.set_profiling(False)
    log("build: %s (k-means iterations %d)" % ({a: round(b, 3) for a, b in H.build_t.items()}, H.iters))
    H.local_rows = int(H.vectors.shape[0])
    H.host_rows = data.cpu().numpy() if keep_host_rows else None
    del data
    H.total_batch = args.batch * world
    H.pool = 8
    if H.hard:
        queries, _ = gen_hard(H.total_batch * H.pool, H.dim, H.components, args.seed + 100, dev, means=H.means)
    else:
        queries, _ = gen_mixture(H.total_batch * H.pool, H.dim, H.components, 0.1, args.seed + 100, dev, means=H.means)
    H.queries = queries.to(H.tdtype).view(H.pool, H.total_batch, H.dim)
    return H


# ------------------------------------------------------------------------------------------------ the section children
def section_configs(args, dev, ctx, out):
    """BASELINE's other IVFFlat configs on this GPU (c2 = configs[1], one GPU's share of configs[2] and configs[4]),
    then configs[0]: the exact scan"""
    fails = out.data["failures"]
    oc = {}
    for wname in ("c2", "c3shard", "c5shard"):
        out.at("other_configs." + wname)
        try:
            oc[wname] = run_workload(ctx, dev, wname, args, fails)
            log("%s: %.0f QPS, recall %.4f, roofline frac %.2f" % (
                wname, oc[wname]["qps"], oc[wname]["recall_at_10"], oc[wname]["roofline"]["frac"]))
        except Exception as e:  # noqa: BLE001
            oc[wname] = {"error": repr(e)}
            fails.append("other_configs.%s: %r" % (wname, e))
        out.put("other_configs", oc)
        torch.cuda.empty_cache()
    out.at("exact_scan")
    try:
        out.put("exact_scan", exact_scan_section(ctx, dev, args, fails))
    except Exception as e:  # noqa: BLE001
        out.put("exact_scan", {"error": repr(e)})
        fails.append("exact_scan: %r" % (e,))


def section_hard(args, dev, ctx, out):
    """the headline's shape on the mid-difficulty data set: recall below 1, unbalanced lists, dozens of Lloyd iterations"""
    fails = out.data["failures"]
    out.at("other_configs.hard")
    try:
        r = run_workload(ctx, dev, "hard", args, fails)
        log("hard: %.0f QPS, recall %.4f at probes 10, k-means %d iterations, lists max / mean %.2f" % (
            r["qps"], r["recall_at_10"], r["kmeans_iterations"], r["list_rows_max_over_mean"]))
        out.put("hard", r)
    except Exception as e:  # noqa: BLE001
        out.put("hard", {"error": repr(e)})
        fails.append("hard: %r" % (e,))


def section_hnsw(args, dev, ctx, out):
    """BASELINE configs[3] at full size"""
    out.at("hnsw: 1 M x 1536 build + ef_search 40 / 100 / 200")
    out.put("hnsw", hnsw_section(ctx, dev, args, out.data["failures"]))


def section_build(args, dev, ctx, out):
    """the product build path through 8 KB pages (+ the oracle walking those very pages: parity and the page-image CPU
    baseline), and the CPU build baseline beside build_secs"""
    from pgvector_amd import _host
    fails = out.data["failures"]
    out.at("build: headline data + kernel-only build")
    H = headline_setup(args, dev, ctx, keep_host_rows=True)
    n, dim, lists, probes, k = H.n, H.dim, H.lists, H.probes, H.k
    rq = min(args.recall_queries, H.total_batch)
    rqueries = H.queries[1][:rq].contiguous()
    exact_d, _ = exact_topk_fp64(H.vectors, rqueries, k, H.metric)
    gd0, _, _ = H.index.search_batch(rqueries, probes, k, want_tid=True)
    recall = recall_at_k(gd0, exact_d, k)
    out.at("build: pgv_host_ivf_build_mirror through pages")
    t0 = time.perf_counter()
    host_tids = (np.arange(n, dtype=np.uint64) << np.uint64(16)) | np.uint64(1)
    g = np.random.default_rng(args.seed + 1)
    ns = min(max(50 * lists, 10000), n)
    host_samples = H.host_rows[np.sort(g.choice(n, ns, replace=False))]
    t_sample = time.perf_counter() - t0
    # the CPU k-means (one core, like the reference) starts now and runs beside the GPU work below
    cpu_build, cpu_build_thread = {}, None
    if not args.no_cpu_baseline:
        cpu_build_thread = threading.Thread(target=cpu_build_kmeans,
                                            args=(host_samples, lists, H.dtype, H.ops, args.seed + 2, cpu_build))
        cpu_build_thread.start()
    rel = _host.Relation()
    t0 = time.perf_counter()
    pix = rel.build_mirror(ctx, H.ops, H.dtype, lists, H.host_rows, host_tids, host_samples, api.make_rng(seed=args.seed + 2))
    ctx.sync()
    t_build = time.perf_counter() - t0
    ph = (ctypes.c_double * 5)()
    _host.lib.pgv_host_ivf_build_phases(ph)
    # the old route to a mirror of these pages (a backend that finds the index on disk): stage + upload
    t0 = time.perf_counter()
    img = rel.stage(H.dtype)
    t_stage = time.perf_counter() - t0
    t0 = time.perf_counter()
    pix2 = api.IvfIndex(ctx, H.metric, H.dtype, dim, img.centers, img.list_offsets, img.vectors, img.tids)
    ctx.sync()
    t_upload = time.perf_counter() - t0
    # the page-built index answers like an index should: recall against the same exact ground truth, and the
    # mirror that came with the build equals the one staged out of the pages
    gd, gs, gt = pix.search_batch(rqueries, probes, k, want_tid=True)
    gd2, gs2, gt2 = pix2.search_batch(rqueries, probes, k, want_tid=True)
    same_mirror = bool(torch.equal(gs, gs2) and torch.equal(gt, gt2) and torch.equal(gd, gd2))
    if not same_mirror:
        fails.append("the mirror pgv_host_ivf_build_mirror returns differs from the one staged out of its pages")
    pix2.close()
    prec = recall_at_k(gd, exact_d, k)
    out.data["build_secs_pages"] = t_build
    out.put("build_pages", {
        "path": "pgv_host_ivf_build_mirror: host rows -> k-means on the GPU (helper thread) WHILE the rows go to the "
                "device (pgv_builder_add without centers: copies on the builder's stream) -> assignment of all rows + "
                "order by list on the device = the mirror (pgv_builder_set_centers / _finish) -> 8 KB pages "
                "written from the mirror's rows as they come back (pgv_index_drain), page array zeroed in the "
                "background meanwhile; no host sort, no staging pass, no second upload",
        "build_secs": t_build, "sample_secs": t_sample,
        "build_phases_secs": dict(zip(("normalise", "kmeans_left_after_upload", "upload_beside_kmeans",
                                       "assign_and_order_by_list_on_device", "page_writer"), [float(x) for x in ph])),
        "mirror_from_pages_secs": {"stage": t_stage, "upload": t_upload,
                                   "note": "what a backend pays that finds the index on disk (not part of the build)"},
        "mirror_equals_staged_pages": same_mirror,
        "pages": int(rel.nblocks), "page_bytes": int(rel.nblocks) * 8192,
        "recall_at_10": prec, "kernel_only_build_secs_this_process": H.build_t["total"]})
    if prec < recall - 0.02:
        fails.append("page-built index recall %.4f below the torch-laid-out index's %.4f" % (prec, recall))
    if not args.no_cpu_baseline:
        out.at("build: the oracle walking the pages the product build wrote")
        from oracle import pyoracle as po
        ora = po.Oracle(native=True)
        oops = po.OPS_L2 if H.ops == api.PGV_OPS_L2 else po.OPS_IP
        odt = po.ORA_F32 if H.dtype == api.PGV_F32 else po.ORA_F16
        cores, cpus, quota = cpu_threads(ora)
        pq = min(256, H.total_batch)
        pqueries = H.queries[1][:pq].contiguous()
        pqh = pqueries.cpu().numpy()
        pages = (rel.rel.pages, int(rel.nblocks))
        pa, ptotal, pel = ora.bench_search(None, pqh, probes, k, cores, 6.0, pages=pages, ops=oops, dtype=odt)
        _, psingle, psingle_el = ora.bench_search(None, pqh[:64], probes, k, 1, 3.0, pages=pages, ops=oops, dtype=odt)
        gd2, _, gt2 = pix.search_batch(pqueries, probes, k, want_tid=True)
        ctx.sync()
        gd2, gt2 = gd2.cpu().numpy(), gt2.cpu().numpy()
        bad2 = []
        for i in range(pq):
            wt, wd = pa[i]
            why = topk_equiv(gt2[i][:len(wt)].astype(np.uint64).tolist(), gd2[i][:len(wt)], wt.tolist(), wd)
            if why:
                bad2.append((i, why))
        if bad2:
            fails.append("page-built index: %d of %d queries differ from the oracle walking the same pages, first: %r"
                         % (len(bad2), pq, bad2[0]))
        out.put("cpu_baseline_page_image", {
            "value": ptotal / pel, "unit": "queries/s", "cores": cores, "kind": "port",
            "single_thread_qps": psingle / psingle_el,
            "layout": "emulated 8 KB page image (oracle_pages.c walks meta / list / entry pages like "
                      "src/ivfscan.c:47-187; no buffer pins, fmgr or tuplesort copies: still an upper bound)",
            "sample": "%d queries in %.1f s on %d threads (%d more on 1 thread) over the %d pages the product build wrote"
                      % (ptotal, pel, cores, psingle, int(rel.nblocks)),
            "parity_of_the_page_built_index": {"checked_queries": pq, "mismatches": len(bad2)}})
    pix.close()
    if cpu_build_thread is not None:
        out.at("build: CPU build baseline (k-means on one core, assignment subsample)")
        cpu_build_thread.join()
        cpu_build_assign(H.host_rows, H.dtype, H.ops, cpu_build)
        cpu_build["gpu_build_secs"] = H.build_t["total"]
        cpu_build["gpu_build_secs_pages"] = t_build
        if "error" in cpu_build:
            fails.append("cpu_build_baseline: %s" % cpu_build["error"])
        out.put("cpu_build_baseline", cpu_build)
    H.index.close()


def section_sweeps(args, dev, ctx, out):
    """batch 1 (the amgettuple path) / 4 / 16 / 64 / 256 / the headline batch, probes 1 / 10 / 32 / 100, the two
    completeness bounds side by side, uniform data"""
    fails = out.data["failures"]
    out.at("sweeps: headline data + build")
    H = headline_setup(args, dev, ctx)
    index, queries, probes, k, dim, n, lists = H.index, H.queries, H.probes, H.k, H.dim, H.n, H.lists
    total_batch, pool = H.total_batch, H.pool
    out_d = torch.empty((total_batch, k), device=dev, dtype=torch.float32)
    out_s = torch.empty((total_batch, k), device=dev, dtype=torch.int64)
    out_t = torch.empty((total_batch, k), device=dev, dtype=torch.int64)
    rq = min(args.recall_queries, total_batch)
    rqueries = queries[1][:rq].contiguous()
    exact_d, _ = exact_topk_fp64(H.vectors, rqueries, k, H.metric)

    def step(i):
        index.search_batch(queries[i % pool], probes, k, want_tid=True, out=(out_d, out_s, out_t))

    out.at("sweeps: batch 1 (pgv_query_*)")
    sweep = {}
    # batch 1 is what amgettuple issues: the device-resident single-query path, host-memory query in,
    # k (distance, tid) out
    qhost = queries[2][:256].cpu().numpy()
    qh = api.Query(index)
    for j in range(20):
        qh.rank(qhost[j], probes)
        qh.scan(0, probes, k)
    lat = []
    for j in range(300):
        t0 = time.perf_counter()
        qh.rank(qhost[j % 256], probes)
        qh.scan(0, probes, k)
        lat.append(time.perf_counter() - t0)
    lat = np.array(lat)
    # parity of that path with the batched one
    bd, bs, _ = index.search_batch(qhost[:32], probes, k)
    same = 0
    for j in range(32):
        qh.rank(qhost[j], probes)
        d1, s1, _, _ = qh.scan(0, probes, k)
        same += int(np.array_equal(s1, bs[j]))
    if same != 32:
        fails.append("single-query path differs from the batched path on %d of 32 queries" % (32 - same))
    qh.close()
    sweep["1"] = {"qps": 1.0 / lat.mean(), "latency_us_p50": float(np.percentile(lat, 50) * 1e6),
                  "latency_us_p90": float(np.percentile(lat, 90) * 1e6),
                  "path": "pgv_query_rank + pgv_query_scan (head of k read back through pinned memory)",
                  "equals_batched_path": "%d/32" % same}
    # the same through the C host glue: ivfflatrescan + k x ivfflatgettuple
    try:
        from pgvector_amd import _host
        img = _host.IvfImage()
        cen, offh = H.centers.cpu().numpy(), H.offsets.cpu().numpy()
        tidh = H.tids.cpu().numpy().astype(np.uint64)
        img.dtype, img.dim, img.lists, img.nrows = H.dtype, dim, lists, n
        img.centers, img.list_offsets, img.tids, img.vectors = cen.ctypes.data, offh.ctypes.data, tidh.ctypes.data, None

        class _Staged:
            pass
        st = _Staged()
        st.img, st.dtype = img, H.dtype
        scan = _host.IvfScan(index, st, probes=probes)
        for j in range(20):
            scan.rescan(qhost[j])
            scan.fetch(limit=k)
        lat2 = []
        for j in range(300):
            t0 = time.perf_counter()
            scan.rescan(qhost[j % 256])
            scan.fetch(limit=k)
            lat2.append(time.perf_counter() - t0)
        scan.close()
        lat2 = np.array(lat2)
        sweep["1"]["gettuple_latency_us_p50"] = float(np.percentile(lat2, 50) * 1e6)
        sweep["1"]["gettuple_path"] = ("pgv_host_ivf_rescan + %d x pgv_host_ivf_gettuple (C host glue; the loop "
                                       "itself is Python/ctypes)" % k)
    except Exception as e:  # noqa: BLE001
        sweep["1"]["gettuple_error"] = repr(e)
    out.put("batch_sweep", sweep)
    out.at("sweeps: batches 4 .. %d" % args.batch)
    for b in (4, 16, 64, 256):
        qb = queries[3][:b].contiguous()
        od = torch.empty((b, k), device=dev, dtype=torch.float32)
        os_ = torch.empty((b, k), device=dev, dtype=torch.int64)
        ot = torch.empty((b, k), device=dev, dtype=torch.int64)
        s = timed_steps(lambda j: index.search_batch(qb, probes, k, want_tid=True, out=(od, os_, ot)), 20)
        sweep[str(b)] = {"qps": b / s, "ms_per_step": s * 1e3}
    ctx.set_profiling(True)
    ctx.reset_stats()
    s_head = timed_steps(step, 20, warmup=3)
    head_stats = ctx.stats()
    ctx.set_profiling(False)
    sweep[str(args.batch)] = {"qps": total_batch / s_head, "ms_per_step": s_head * 1e3,
                              "note": "this child's own run of the headline step"}
    out.put("batch_sweep", sweep)

    out.at("sweeps: probes 1 / 10 / 32 / 100")
    psweep = {}
    for p in (1, 10, 32, 100):
        if p > lists:
            continue
        s = timed_steps(lambda j: index.search_batch(queries[j % pool], p, k, want_tid=True, out=(out_d, out_s, out_t)), 6)
        gd, _, _ = index.search_batch(rqueries, p, k)
        psweep[str(p)] = {"qps": total_batch / s, "ms_per_step": s * 1e3, "recall_at_10": recall_at_k(gd, exact_d, k)}
    out.put("probes_sweep", psweep)

    if H.metric == api.PGV_L2SQ:
        out.at("sweeps: completeness bounds side by side")
        try:
            st = bound_mode_run(ctx, step, 10, total_batch)
            out.put("bound_modes", {
                "worst_case": {"qps": total_batch / s_head, "ms_per_step": s_head * 1e3,
                               "scan_redo_queries_per_step": head_stats["scan_redo_queries"] / 20,
                               "scan_widened_queries_per_step": head_stats["scan_widened_queries"] / 20,
                               "bound": "deterministic (default): gamma_(d/4+4) 2 |q||x|max + gamma_(d/64+10) |x|max^2 + "
                                        "2 gamma_(d+2) distance, u = 2^-24, four accumulator chains per output"},
                "statistical": st,
                "cost_of_worst_case": 1.0 - (total_batch / s_head) / st["qps"],
                "note": "same index, same queries, same results; uniform data and the other configs carry the same pair "
                        "(bound_statistical beside the default)"})
        except Exception as e:  # noqa: BLE001
            out.put("bound_modes", {"error": repr(e)})
            fails.append("bound_modes: %r" % (e,))

    out.at("sweeps: uniform data")
    try:
        index.close()
        del H.vectors
        udata = gen_uniform(n, dim, args.seed + 7, dev).to(H.tdtype)
        uc, uo, uv, ut, uit, ubt, uix = build_index(ctx, udata, lists, args.seed, 1, 0, H.dtype, H.ops, H.metric)
        del udata
        uq = gen_uniform(total_batch, dim, args.seed + 8, dev).to(H.tdtype)
        ued, _ = exact_topk_fp64(uv, uq[:rq], k, H.metric)
        ures = {}
        for p in (10, 100):
            s = timed_steps(lambda j: uix.search_batch(uq, p, k, want_tid=True, out=(out_d, out_s, out_t)), 5)
            gd, _, _ = uix.search_batch(uq[:rq].contiguous(), p, k)
            ures[str(p)] = {"qps": total_batch / s, "recall_at_10": recall_at_k(gd, ued, k)}
        ubound = bound_mode_run(ctx, lambda j: uix.search_batch(uq, 10, k, want_tid=True, out=(out_d, out_s, out_t)),
                                5, total_batch) if H.metric == api.PGV_L2SQ else None
        # the default (deterministic) bound on the same index and batch, with its counters (which of
        # the two -- band recheck or k' widening -- a cost on concentrated distances comes from)
        ctx.set_profiling(True)
        ctx.reset_stats()
        s_def = timed_steps(lambda j: uix.search_batch(uq, 10, k, want_tid=True, out=(out_d, out_s, out_t)), 5, warmup=2)
        st_def = ctx.stats()
        ctx.set_profiling(False)
        udef = {"qps": total_batch / s_def, "ms_per_step": s_def * 1e3,
                "scan_redo_queries_per_step": st_def["scan_redo_queries"] / 7, "scan_widened_queries_per_step": st_def["scan_widened_queries"] / 7,
                "scan_ms_per_step": st_def["scan_ms"] / 7}
        out.put("uniform", {"data": "U[0,1)^%d (test/t/003_ivfflat_vector_build_recall.pl:60)" % dim,
                            "bound_default_probes_10": udef,
                            "bound_statistical_probes_10": ubound,
                            "build_secs": ubt["total"], "kmeans_iterations": uit, "probes": ures,
                            "note": "uniform high-d data has no cluster structure: IVF recall at 1 % of the lists "
                                    "is low by construction (the reference skips such cases, t/003:101-104)"})
        uix.close()
    except Exception as e:  # noqa: BLE001
        out.put("uniform", {"error": repr(e)})
        fails.append("uniform: %r" % (e,))


def section_backends(args, dev, ctx, out):
    """several backends on ONE device mirror: threads, processes, pooled clients.  LAST in the run, and a child: a
    backend that wedges costs this section its budget and nothing else."""
    out.at("backends: headline data + build")
    H = headline_setup(args, dev, ctx)
    qhost = H.queries[2][:256].cpu().numpy()
    cb = {}
    out.data["concurrent_backends"] = cb
    out.at("backends: rows")
    try:
        concurrent_backends(H.index, 0, qhost, H.queries, H.probes, H.k, args, dev, cb, out.flush)
    except Exception as e:  # noqa: BLE001
        cb["error"] = repr(e)
        out.data["failures"].append("concurrent_backends: %r" % (e,))
    out.flush()


def section_full(which):
    def run(args, dev, ctx, out):
        """BASELINE configs[2] (c3) / configs[4] (c5) at their real 10 M-row shape on this one GPU"""
        out.at("full_configs." + which)
        out.put("full_" + which, run_full_config(ctx, dev, which, args, out.data["failures"]))
    return run


SECTIONS = {"configs": section_configs, "hard": section_hard, "hnsw": section_hnsw, "build": section_build, "sweeps": section_sweeps,
            "backends": section_backends, "c3full": section_full("c3"), "c5full": section_full("c5")}


def section_main(args):
    import traceback
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    ctx = api.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    out = SectionOut(args.section_out)
    rc = 0
    try:
        SECTIONS[args.section](args, dev, ctx, out)
        out.data["_at"] = "done"
    except Exception as e:  # noqa: BLE001
        traceback.print_exc()
        out.data["failures"].append("%r (at: %s)" % (e, out.data["_at"]))
        rc = 1
    out.flush()
    if not args.section_out:
        print(json.dumps(out.data, default=str), file=sys.stderr)
    sys.stderr.flush()
    if args.soft_exit:
        sys.exit(rc)   # (under rocprofv3: its tool writes the traces from an exit handler)
    os._exit(rc)   # no interpreter teardown: a thread that is still inside the library cannot hold the exit up


# ---------------------------------------------------------------------------------------------------------------------
# `python bench.py --gpus N` starts its own N ranks (one process per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in
# the environment exactly as torch.distributed.run would set them; under torch.distributed.run WORLD_SIZE is already
# there and nothing is started from here).  The launcher (this process) never touches a device: it finds a free port,
# starts the ranks in sessions of their own, waits for every rank to report that it JOINED the group
# (--startup-timeout), relays rank 0's ONE line after checking that the line really describes N ranks, and takes every
# rank down with its process group when one of them fails or the watchdog's time is up.  Fewer devices than ranks over
# RCCL, a group that formed with another size, a rank that died: a `failures` entry and exit code 2 -- never a 1-GPU
# number under "n_gpus": N.
LAUNCH_ENV = "PGV_BENCH_LAUNCH_DIR"


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_mark_ready(rank):
    """a rank tells the launcher that the group formed with it in it (a file: the launcher holds no group membership)"""
    d = os.environ.get(LAUNCH_ENV)
    if d:
        try:
            with open(os.path.join(d, "ready.%d" % rank), "w") as f:
                f.write(str(os.getpid()))
        except OSError:
            pass


def launch_failure(fd, args, failures, extra=None):
    """the contract's fields with no value, what went wrong, exit code 2"""
    n, dim, lists, probes, tname, oname = WORKLOADS[args.workload]
    line = {"metric": "QPS @ recall@10 (IVFFlat, 1M x 1536d)" if args.workload == "headline" else "QPS @ recall@10 (IVFFlat)",
            "value": None, "unit": "queries/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": None, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": tname,
            "data": "synthetic", "config": {"workload": args.workload, "rows": n, "dim": dim, "lists": lists,
                                            "probes": args.probes or probes, "k": args.k, "batch_per_gpu": args.batch},
            "failures": list(failures)}
    if extra:
        line.update(extra)
    for f in failures:
        log("FAILED: " + f)
    os.write(fd, (json.dumps(_clean(line), default=str) + "\n").encode())
    sys.exit(2)


def dry_launch_rank(args, json_fd, world, rank):
    """--dry-launch: form the group, agree on its size, say so.  With --backend gloo no device is touched (the
    launcher's CPU test); with nccl every rank takes its device and the all-reduce runs over RCCL."""
    use_dev = args.backend == "nccl"
    fault = os.environ.get("PGV_BENCH_TEST_FAULT", "")   # the launcher's own tests: "die:R" / "hang:R" before joining
    if fault == "die:%d" % rank:
        os._exit(7)
    if fault == "hang:%d" % rank:
        time.sleep(3600)
    if use_dev:
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    else:
        dist.init_process_group(args.backend, rank=rank, world_size=world)
    launch_mark_ready(rank)
    one = torch.ones(1, dtype=torch.int64, device="cuda" if use_dev else "cpu")
    dist.all_reduce(one)
    ranks = [torch.zeros(1, dtype=torch.int64, device=one.device) for _ in range(world)]
    dist.all_gather(ranks, torch.tensor([rank], dtype=torch.int64, device=one.device))
    agreed = int(one.item())
    ok = agreed == world == args.gpus and [int(r.item()) for r in ranks] == list(range(world))
    if rank == 0:
        line = {"dry_launch": True, "n_gpus": world, "world_agreed": agreed, "ranks_seen": [int(r.item()) for r in ranks],
                "backend": args.backend, "launcher": os.environ.get("PGV_BENCH_LAUNCHER", "external (WORLD_SIZE was set)"),
                "master": "%s:%s" % (os.environ.get("MASTER_ADDR"), os.environ.get("MASTER_PORT"))}
        if not ok:
            line["failures"] = ["the group agreed on %d ranks, --gpus is %d" % (agreed, args.gpus)]
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 2)


def self_launch(args, json_fd):
    import signal
    n = args.gpus
    if not args.dry_launch or args.backend == "nccl":
        # counted without creating a context in the launcher (the library's own count; torch would initialise HIP here)
        ndev = int(pgvector_amd.lib.pgv_device_count())
        if ndev < 1:
            return launch_failure(json_fd, args, ["--gpus %d: no gfx950 device on this node" % n])
        if args.backend == "nccl" and ndev < n:
            return launch_failure(json_fd, args, [
                "--gpus %d over RCCL needs %d devices, this node has %d (a functional N-rank run on fewer devices: --backend gloo)"
                % (n, n, ndev)], {"devices": ndev})
    else:
        ndev = 0
    port = free_port()
    ldir = tempfile.mkdtemp(prefix="pgv_launch_", dir="/tmp")
    argv = [a for a in sys.argv[1:]]
    procs, t0 = [], time.perf_counter()
    log("[launcher] starting %d ranks of `%s` (backend %s, 127.0.0.1:%d, %d devices)" % (n, " ".join(argv), args.backend, port, ndev))
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PGV_BENCH_LAUNCHER="bench.py self-launch")
        env[LAUNCH_ENV] = ldir
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", "1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, stdin=subprocess.DEVNULL,
                                      stdout=subprocess.PIPE if r == 0 else 2, stderr=2, start_new_session=True, env=env))
    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    reader.start()

    def kill_all():
        for p in procs:
            if p.poll() is None:
                try:
                    os.killpg(p.pid, signal.SIGKILL)
                except OSError:
                    pass
        for p in procs:
            try:
                p.wait(timeout=5)
            except subprocess.TimeoutExpired:
                pass

    failures, joined = [], False
    overall = (args.watchdog_secs if args.watchdog_secs > 0 else 3600) + 60
    try:
        while True:
            rcs = [p.poll() for p in procs]
            if all(rc is not None for rc in rcs):
                break
            el = time.perf_counter() - t0
            if not joined:
                ready = sum(os.path.exists(os.path.join(ldir, "ready.%d" % r)) for r in range(n))
                if ready == n:
                    joined = True
                    log("[launcher] all %d ranks joined the group after %.1f s" % (n, el))
                elif el > args.startup_timeout:
                    failures.append("only %d of %d ranks joined the group within %.0f s (--startup-timeout)" % (ready, n, args.startup_timeout))
                    break
            bad = [(r, rc) for r, rc in enumerate(rcs) if rc not in (None, 0)]
            if bad:
                # a rank is gone: the others would wait in a collective for ever; rank 0 gets a moment to print its line
                time.sleep(3.0 if rcs[0] is None else 0.0)
                failures.append("rank %d exited with code %r; the other ranks were stopped" % bad[0])
                break
            if el > overall:
                failures.append("the ranks were not back after %d s; stopped" % overall)
                break
            time.sleep(0.1)
    finally:
        kill_all()
        reader.join(timeout=5)
        shutil.rmtree(ldir, ignore_errors=True)
    rcs = [p.returncode for p in procs]
    text = (out0[0] if out0 else b"").decode(errors="replace").strip().splitlines()
    line = None
    for cand in reversed(text):
        try:
            line = json.loads(cand)
            break
        except ValueError:
            continue
    if line is None:
        return launch_failure(json_fd, args, failures + ["rank 0 printed no line (exit codes by rank: %r)" % (rcs,)])
    # the line must describe N ranks: what it says, and what the library's communicator saw
    if line.get("n_gpus") != n:
        failures.append("rank 0's line says n_gpus %r, --gpus is %d" % (line.get("n_gpus"), n))
    if not args.dry_launch and n > 1 and (line.get("multi_gpu") or {}).get("pgv_comm_size") != n:
        failures.append("pgv_comm size %r, --gpus is %d" % ((line.get("multi_gpu") or {}).get("pgv_comm_size"), n))
    if any(rc != 0 for rc in rcs) and not failures and not line.get("failures"):
        failures.append("exit codes by rank: %r" % (rcs,))
    line["launcher"] = {"kind": "bench.py self-launch (one process per rank, started by the process the driver ran)",
                        "ranks": n, "devices": ndev, "joined": joined, "exit_codes": rcs,
                        "secs": round(time.perf_counter() - t0, 1)}
    if failures:
        line["failures"] = list(line.get("failures", [])) + failures
        for f in failures:
            log("FAILED: " + f)
    os.write(json_fd, (json.dumps(line, default=str) + "\n").encode())
    sys.exit(2 if line.get("failures") else 0)


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
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skips the oracle: no cpu_baseline, no TID parity")
    ap.add_argument("--no-reference-baseline", action="store_true", help="cpu_baseline stays the oracle's port (no "
                    "oracle/_ref/ref_scan_bench run)")
    ap.add_argument("--cpu-port-secs", type=float, default=6.0, help="seconds of the oracle's threads (cpu_baseline.port)")
    ap.add_argument("--cpu-reference-secs", type=float, default=10.0, help="seconds of the reference's backend processes")
    ap.add_argument("--hard-sigma", type=float, default=HARD["sigma"], help="gen_hard: component spread / spread of the means")
    ap.add_argument("--hard-alpha", type=float, default=HARD["alpha"], help="gen_hard: per-dimension variance ~ j^-alpha")
    ap.add_argument("--hard-zipf", type=float, default=HARD["zipf"], help="gen_hard: component weights ~ c^-zipf")
    ap.add_argument("--no-sweeps", action="store_true", help="skip every optional section (the child processes)")
    ap.add_argument("--no-traffic", action="store_true", help="skip the live rocprofv3 PMC passes")
    ap.add_argument("--sections", default=",".join(SECTION_ORDER),
                    help="optional sections to run, in this order (default: all of %s)" % ",".join(SECTION_ORDER))
    ap.add_argument("--all-process-rows", action="store_true", help="also 32 own-context backend processes")
    ap.add_argument("--budget-secs", type=int, default=420, help="no optional section STARTS once the run is this old")
    ap.add_argument("--watchdog-secs", type=int, default=720, help="last resort for the parent process: print the line as "
                    "far as it has got and exit 2 (0: no watchdog)")
    ap.add_argument("--child", action="store_true", help="(internal) the short run the PMC passes profile")
    ap.add_argument("--section", default=None, choices=sorted(SECTIONS), help="(internal) run ONE optional section and "
                    "write its JSON to --section-out")
    ap.add_argument("--section-out", default=None)
    ap.add_argument("--soft-exit", action="store_true", help="(internal) a section child ends through the interpreter's "
                    "normal exit (profilers write their traces from exit handlers)")
    ap.add_argument("--host-io", action="store_true", help="also time the batch with host-memory queries/results")
    ap.add_argument("--recall-queries", type=int, default=256)
    ap.add_argument("--exact-scan", action="store_true", help="A/B: keep the batched L2 scan on the vector-ALU kernels "
                    "(pgv_ctx_set_exact_scan)")
    ap.add_argument("--overlap", type=int, default=1, help="streams consecutive batches of the TIMED loop alternate on "
                    "(pgv_index_set_overlap; 1, the default: one stream, stream-ordered -- the scan kernel's launches are then "
                    "timed undisturbed; the overlapped form is measured right after as `overlapped_batches`)")
    ap.add_argument("--overlap-lanes", type=int, default=3, help="lanes of the `overlapped_batches` measurement (0: skip it)")
    ap.add_argument("--settle-ms", type=float, default=300.0,
                    help="untimed steps for this long BEFORE the W warmup steps: the float64 ground truth just above leaves the "
                         "chip at whatever clocks a dense fp64 pass ends with, and a 27 ms timed region would measure that "
                         "transient instead of the scan's steady state (0: none; the count is reported as config.settle_steps)")
    ap.add_argument("--placement", default="balanced", choices=("balanced", "modulo"),
                    help="N GPUs: lists to ranks by rows (LPT) or l %% N")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for a functional "
                                                     "multi-rank run on a single GPU)")
    ap.add_argument("--sharded-path", action="store_true", help="--gpus 1 through the N-GPU code path: torch.distributed over "
                    "RCCL with ONE rank, the library's communicator, pgv_kmeans_sharded, the row exchange, "
                    "pgv_search_batch_sharded (the closest a one-GPU box gets to the multi-GPU run; no optional sections)")
    ap.add_argument("--dry-launch", action="store_true", help="only start the --gpus N ranks, form the group, agree on its "
                    "size and print a short line (no device needed with --backend gloo: the launcher's own test)")
    ap.add_argument("--startup-timeout", type=float, default=300.0, help="self-launch: seconds every rank has to join the "
                    "group (the first `import torch` on a fresh box takes 1-2 minutes)")
    ap.add_argument("--no-self-launch", action="store_true", help="--gpus N > 1 without WORLD_SIZE in the environment is an "
                    "error instead of starting the N ranks from here")
    args = ap.parse_args()
    HARD.update(sigma=args.hard_sigma, alpha=args.hard_alpha, zipf=args.hard_zipf)
    PLACEMENT["policy"] = args.placement
    t_program = time.perf_counter()
    # fd 1 carries the ONE JSON line and nothing else: libraries that greet on stdout (RCCL's version banner)
    # are pointed at stderr for the rest of the run
    json_fd = os.dup(1)
    os.dup2(2, 1)
    if args.section:
        os.close(json_fd)
        return section_main(args)
    if args.child:
        args.no_cpu_baseline = args.no_sweeps = args.no_traffic = True
    WATCH["fd"] = json_fd
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or args.dry_launch):
        # `python bench.py --gpus N` as the driver types it: the N ranks are started from HERE (the reference's analogue:
        # a parallel build launches its own workers, src/ivfbuild.c:830-966) -- never a silent 1-GPU run that says N
        if args.no_self_launch:
            return launch_failure(json_fd, args, ["--gpus %d without WORLD_SIZE and --no-self-launch: nothing started the ranks"
                                                  % args.gpus])
        return self_launch(args, json_fd)
    if args.watchdog_secs > 0:
        threading.Thread(target=watchdog, args=(args.watchdog_secs,), daemon=True).start()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    WATCH["rank"] = rank
    if world != args.gpus:
        # whoever launched this (torch.distributed.run, the self-launcher) formed another group than --gpus says
        if rank == 0:
            launch_failure(json_fd, args, ["WORLD_SIZE is %d but --gpus is %d: refusing to report one as the other"
                                           % (world, args.gpus)])
        sys.exit(2)
    sharded = world > 1 or args.sharded_path
    if sharded or args.dry_launch:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(free_port()) if world == 1 else "29500")
        if args.dry_launch:
            return dry_launch_rank(args, json_fd, world, rank)
        ndev = torch.cuda.device_count()
        if args.backend == "nccl" and ndev < world:
            # RCCL wants one device per rank; `--backend gloo` is the functional N-ranks-on-fewer-devices run
            if rank == 0:
                launch_failure(json_fd, args, ["--gpus %d over RCCL needs %d devices, this node has %d" % (world, world, ndev)])
            sys.exit(2)
        local_rank = local_rank % max(ndev, 1)
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
        if dist.get_world_size() != args.gpus:
            if rank == 0:
                launch_failure(json_fd, args, ["the process group has %d ranks, --gpus is %d" % (dist.get_world_size(), args.gpus)])
            sys.exit(2)
        launch_mark_ready(rank)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    ctx = api.Context(local_rank, stream=torch.cuda.current_stream().cuda_stream)
    comm, comm_kind = None, None
    if sharded:
        comm_kind = "rccl" if args.backend == "nccl" else "host"
        if comm_kind == "rccl":
            # the library's own RCCL communicator; if ANY rank cannot have it, every rank takes the callbacks into
            # torch.distributed instead (still RCCL, driven by torch) -- agreed on, so that no rank waits in a collective
            # the others never enter
            try:
                comm = api.Comm(ctx, backend="rccl")
                ok = 1
            except Exception as e:   # noqa: BLE001
                log("rank %d: pgv_comm_create failed (%r)" % (rank, e))
                comm, ok = None, 0
            flag = torch.tensor([ok], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0:
                if comm is not None:
                    comm.close()
                comm = api.Comm(ctx, backend="host")
                comm_kind = "torch.distributed nccl callbacks (pgv_comm_create failed on some rank)"
        else:
            comm = api.Comm(ctx, backend="host")
    if args.exact_scan:
        ctx.set_exact_scan(True)
    failures = []
    if comm_kind is not None and comm_kind.startswith("torch.distributed"):
        failures.append("comm: " + comm_kind)
    if comm is not None and comm.world != args.gpus:
        failures.append("the library's communicator has %d ranks, --gpus is %d" % (comm.world, args.gpus))

    # ---------------------------------------------------------------- setup
    WATCH["section"] = "data + build"
    H = headline_setup(args, dev, ctx, world, rank, comm, sharded=sharded)
    n, dim, lists, probes, tname, oname = H.n, H.dim, H.lists, H.probes, H.tname, H.oname
    dtype, ops, metric, esize, k = H.dtype, H.ops, H.metric, H.esize, H.k
    centers, offsets, vectors, tids, index = H.centers, H.offsets, H.vectors, H.tids, H.index
    iters, build_t, build_stats, components = H.iters, H.build_t, H.build_stats, H.components
    total_batch, pool, queries = H.total_batch, H.pool, H.queries

    out_d = torch.empty((total_batch, k), device=dev, dtype=torch.float32)
    out_s = torch.empty((total_batch, k), device=dev, dtype=torch.int64)
    out_t = torch.empty((total_batch, k), device=dev, dtype=torch.int64)

    def step(i):
        q = queries[i % pool]
        if not sharded:
            index.search_batch(q, probes, k, want_tid=True, out=(out_d, out_s, out_t))
            return out_d, out_t
        # N GPUs (pgv_search_batch_sharded): each rank ranks its own slice of the batch against the
        # replicated centers, the probe lists are all-gathered, each rank scans the lists it owns, the
        # per-rank top-k are all-gathered and merged on the device -- per-GPU work per step does not grow with N
        comm.search_batch(index, q, probes, k, out=(out_d, out_t))
        return out_d, out_t

    # ---------------------------------------------------------------- recall vs exact fp64
    WATCH["section"] = "recall against float64"
    rq = min(args.recall_queries, total_batch)
    rqueries = queries[1][:rq].contiguous()
    # exact float64 brute force over every row; with N ranks each one scans the rows it holds and the per-rank
    # exact top-k are merged (never the GPUs' own fp32 scan as its own ground truth)
    exact_d, _ = exact_topk_fp64(vectors, rqueries, k, metric)
    if not sharded:
        got_d, got_s, got_t = index.search_batch(rqueries, probes, k, want_tid=True)
        recall_truth = "exact float64 brute force over all %d rows, %d queries" % (n, rq)
    else:
        exact_d = sharding.merge_exact_topk(exact_d, k)
        got_d, got_t = comm.search_batch(index, rqueries, probes, k)
        recall_truth = "exact float64 brute force, every rank over the rows it holds, merged over %d ranks, %d queries" % (world, rq)
    ctx.sync()
    recall = recall_at_k(got_d, exact_d, k)
    log("recall@%d = %.4f at probes=%d (%s)" % (k, recall, probes, recall_truth))

    # ----------------------------------------------------------------- timed
    WATCH["section"] = "timed steps"
    if not sharded and args.overlap > 1:
        # consecutive batches on `overlap` internal streams of the library (pgv_index_set_overlap): one batch's center
        # ranking / planning / top-k / recheck run under the other's list scan; every batch is complete inside the timed
        # region (the synchronize below waits for all streams)
        index.set_overlap(args.overlap)
    # the K steps once BEFORE any settling (the un-settled number beside `value`)
    for i in range(args.warmup):
        step(i)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t_u = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    torch.cuda.synchronize()
    unsettled_qps = total_batch * args.steps / (time.perf_counter() - t_u)   # (this rank's clock: detail only)
    settle_steps = 0
    if args.settle_ms > 0:
        # (every rank runs the same number: the sharded step has collectives in it)
        t_settle = time.perf_counter()
        while True:
            for i in range(8):
                step(settle_steps + i)
            settle_steps += 8
            torch.cuda.synchronize()
            done = torch.tensor([1.0 if (time.perf_counter() - t_settle) * 1e3 >= args.settle_ms else 0.0], device=dev)
            if world > 1:
                dist.all_reduce(done, op=dist.ReduceOp.MIN)
            if done.item() > 0 or settle_steps >= 4096:
                break
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
    if not sharded and args.overlap > 1:
        index.set_overlap(1)     # everything below reads its answers in stream order again
    overlapped = None
    if not sharded and args.overlap == 1 and args.overlap_lanes > 1 and not args.child:
        # the same K steps once more with one caller's consecutive batches on `overlap_lanes` internal streams
        # (pgv_index_set_overlap: list scans take turns, everything else of a batch runs under another batch's scan).
        # Not `value`: the scan launches of the timed region above are timed without company, these are not.
        WATCH["section"] = "overlapped batches"
        index.set_overlap(args.overlap_lanes)
        for i in range(max(args.warmup, args.overlap_lanes)):
            step(i)
        ctx.set_profiling(True)
        ctx.reset_stats()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(args.steps):
            step(args.warmup + i)
        torch.cuda.synchronize()
        el2 = time.perf_counter() - t1
        st2 = ctx.stats()
        ctx.set_profiling(False)
        index.set_overlap(1)
        overlapped = {"lanes": args.overlap_lanes, "qps": total_batch * args.steps / el2, "ms_per_step": el2 / args.steps * 1e3,
                      "scan_ms_per_launch": st2["scan_ms"] / max(st2["scan_launches"], 1), "steps": args.steps}
        log("overlapped batches (%d lanes): %.0f QPS, %.3f ms/step, scan %.3f ms/launch" % (
            args.overlap_lanes, overlapped["qps"], overlapped["ms_per_step"], overlapped["scan_ms_per_launch"]))
    if world > 1:
        el = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        elapsed = float(el.item())

    qps = total_batch * args.steps / elapsed
    launches = max(stats["scan_launches"], 1)
    algo_bytes = stats["scan_pairs"] * esize * dim     # 4*d (fp32) / 2*d (fp16) bytes per scored vector
    stream_bytes = stats["scan_rows"] * esize * dim    # rows actually streamed (a row shared by a query group counts once)
    unique_bytes = stats["scan_unique_rows"] * esize * dim  # rows of the lists somebody probes: one ideal pass
    scan_s = stats["scan_ms"] / 1e3
    streamed_gbps = stream_bytes / scan_s / 1e9 if scan_s > 0 else 0.0
    unique_gbps = unique_bytes / scan_s / 1e9 if scan_s > 0 else 0.0
    avg_launch_ms = stats["scan_ms"] / launches
    roofline = {
        "kernel": ("tile_scan_kernel / scan_kernel" if args.exact_scan else "mfma_scan_kernel")
                  + " (IVFFlat list scan, GetScanItems, src/ivfscan.c:123-187)",
        "bound": "hbm", "achieved": unique_gbps, "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": unique_gbps / HBM_PEAK_GBS,
        "achieved_streamed": streamed_gbps, "frac_streamed": streamed_gbps / HBM_PEAK_GBS,
        "traffic": None, "traffic_source": None,
        "streamed_bytes_per_launch": stream_bytes / launches,
        "unique_bytes_per_launch": unique_bytes / launches,
        "passes": stream_bytes / unique_bytes if unique_bytes > 0 else None,
        "algorithmic_bytes_per_launch": algo_bytes / launches,
        "algorithmic_GBps": algo_bytes / scan_s / 1e9 if scan_s > 0 else 0.0,
        "avg_launch_ms": avg_launch_ms, "launches": launches,
        # useful multiply-adds (one per element of every scored (query, row) pair) against the dense
        # MFMA peak of the dtype; the kernel pads a task to 16 or 32 queries, so the matrix cores do more
        "useful_tflops": stats["scan_pairs"] * dim * 2.0 / scan_s / 1e12 if scan_s > 0 else 0.0,
        "mfma_peak_tflops": 157.3 if tname == "f32" else 2500.0,
        "measured_stream_ceiling_GBps": 6200.0,
        "note": "achieved/frac = bytes of the rows some query of the batch probes, each counted ONCE (the floor of any "
                "batched scan), per kernel second (HIP events on the launch stream) against the 8 TB/s peak; "
                "achieved_streamed counts a row once per group of <= 32 queries that shares it (what the kernel "
                "really reads; passes = streamed / unique); the per-(query,row)-pair figure of SURVEY 8d "
                "(algorithmic_GBps) exceeds the physical rate because a row is scored for many queries on the matrix cores.  "
                "measured_stream_ceiling_GBps: what a kernel that only stages the same 128-row tasks into LDS reaches on "
                "this part, any access pattern (tools/stream_patterns.hip, profiles/r02b_stream_patterns.txt, DESIGN.md 4.1c)",
    }
    line = WATCH["line"] = {
        "metric": "QPS @ recall@10 (IVFFlat, 1M x 1536d)" if args.workload == "headline"
                  else "QPS @ recall@10 (IVFFlat)",
        "value": qps, "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": tname, "data": "synthetic",
        "config": {"workload": "%s: IVFFlat %s_%s_ops %d x %d %s, lists=%d, probes=%d, k=%d, "
                               "batch=%d queries/step/GPU, %s"
                               % (args.workload, "vector" if tname == "f32" else "halfvec", oname, n, dim, tname,
                                  lists, probes, k, args.batch,
                                  "overlapping Gaussian mixture (gen_hard: sigma %.2f, variance ~ j^-%.1f, Zipf %.1f)"
                                  % (HARD["sigma"], HARD["alpha"], HARD["zipf"]) if H.hard
                                  else "Gaussian mixture (%d components, sigma 0.1)" % components),
                   "rows": n, "dim": dim, "lists": lists, "probes": probes, "k": k,
                   "batch_per_gpu": args.batch, "settle_steps": settle_steps, "parallelism": "lists sharded over %d ranks (%s by rows); k-means all-reduce, probe-list and "
                                  "top-k all-gathers inside libpgv_hip (RCCL on the library's stream)" % (world, args.placement),
                   "local_rows": H.local_rows},
        "value_unsettled": unsettled_qps,
        "recall_at_10": recall, "recall_ground_truth": recall_truth,
        "build_secs": build_t["total"], "build_phases_secs": build_t, "kmeans_iterations": iters,
        "build_assign": {"rows": build_stats["assign_rows"],
                         "rechecked_fraction": build_stats["assign_recheck_rows"] / build_stats["assign_rows"]
                         if build_stats["assign_rows"] else None,
                         "redone_fraction": build_stats["assign_redo_rows"] / build_stats["assign_rows"]
                         if build_stats["assign_rows"] else None,
                         "note": "L2 assignment = MFMA pre-filter + exact recheck (k-means iterations and the heap rows)"},
        "roofline": roofline,
        "center_rank_ms_per_step": stats["aux_ms"] / args.steps,
        "scan_ms_per_step": stats["scan_ms"] / args.steps,
        "scan_redo_queries_per_step": stats["scan_redo_queries"] / args.steps,
        "scan_widened_queries_per_step": stats["scan_widened_queries"] / args.steps,
        "scan_path": "exact vector-ALU kernels (--exact-scan)" if args.exact_scan else "auto",
        "overlap_lanes": args.overlap if not sharded else 1,
    }
    if overlapped:
        line["overlapped_batches"] = overlapped
    if sharded:
        # what the first real N-GPU run needs to be read: the communicator's size, the exchanges per step and per
        # Lloyd iteration in bytes, the build's phases (build_phases_secs: kmeans = k-means++ + Lloyd with
        # one fused all-reduce per iteration; assign; layout = the all-to-all of the rows to their lists' owners)
        # every rank's share and its own scan time: a step ends with the slowest rank
        mine = torch.tensor([float(H.local_rows), stats["scan_ms"] / max(args.steps, 1)], dtype=torch.float64,
                            device=dev if args.backend == "nccl" else torch.device("cpu"))
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        rows_all = [float(t[0].item()) for t in every]
        scan_all = [float(t[1].item()) for t in every]
        line["multi_gpu"] = {
            "placement": "%s (pgvector_amd/sharding.py plan_owners)" % args.placement,
            "rows_per_rank_min": min(rows_all), "rows_per_rank_mean": sum(rows_all) / world, "rows_per_rank_max": max(rows_all),
            "rows_per_rank_max_over_mean": max(rows_all) / (sum(rows_all) / world),
            "slowest_rank_scan_ms": max(scan_all), "scan_ms_by_rank": scan_all,
            "measured_on": ("ONE rank (--sharded-path): a rehearsal of the N-GPU code path on one device, not a scaling number"
                            if world == 1 else "RCCL over xGMI") if args.backend == "nccl" else
                           "UNMEASURED ON HARDWARE: %d ranks on one GPU over %s (functional run)" % (world, args.backend),
            "comm_size": comm.world, "pgv_comm_size": comm.world, "torch_world_size": dist.get_world_size(),
            "backend": args.backend, "communicator": comm_kind,
            "launcher": os.environ.get("PGV_BENCH_LAUNCHER", "external (WORLD_SIZE was set: torch.distributed.run)"),
            "kmeans_allreduce_bytes_per_iteration": int(lists * dim * 4 + lists * 4 + 8),
            "kmeans_iterations": iters,
            "search_allgather_bytes_per_step": {"probe_lists": int(total_batch * probes * 4),
                                                "topk": int(total_batch * k * 12 * world)},
            "rows_per_rank": H.local_rows}
    log("timed: %.0f QPS, %.3f ms/step, scan %.3f ms/launch (frac %.2f)" % (qps, elapsed / args.steps * 1e3, avg_launch_ms,
                                                                            roofline["frac"]))

    single = rank == 0 and not sharded
    if single and args.host_io:
        # what a Postgres backend sees for a batch: queries and results in HOST memory
        qh = [queries[j].cpu().numpy() for j in range(min(pool, 4))]
        s = timed_steps(lambda j: index.search_batch(qh[j % len(qh)], probes, k, want_tid=True), args.steps)
        line["host_buffers"] = {"value": args.batch / s, "unit": "queries/s", "ms_per_step": s * 1e3,
                                "h2d_bytes_per_step": int(args.batch * dim * esize),
                                "d2h_bytes_per_step": int(args.batch * k * 20)}

    # --------------------------------------------------- parity with the CPU oracle + its speed (mandatory, in-process)
    if single and not args.no_cpu_baseline:
        WATCH["section"] = "oracle parity + cpu_baseline"
        try:
            pq = min(256, total_batch)
            pqueries = queries[1][:pq].contiguous()
            pd, ps, pt = index.search_batch(pqueries, probes, k, want_tid=True)
            ctx.sync()
            hc, ho, hv = centers.cpu().numpy(), offsets.cpu().numpy(), vectors.cpu().numpy()
            ht, hq = tids.cpu().numpy().astype(np.uint64), pqueries.cpu().numpy()
            base, answers, _ = cpu_baseline(hc, ho, hv, ht, hq, probes, k, dtype, ops, budget_s=args.cpu_port_secs)
            line["cpu_baseline"] = base
            if not args.no_reference_baseline:
                # the reference's own compiled scan over the page image of this index: `value` becomes
                # ITS rate ("kind": "reference"), the port's bare loops stay beside it
                WATCH["section"] = "cpu_baseline: the reference's own scan"
                try:
                    ref = reference_baseline(hc, ho, hv, ht, hq, probes, k, dtype, ops, answers, base["cores"],
                                             secs=args.cpu_reference_secs, secs_single=args.cpu_reference_secs / 2.0)
                    ref["port_value"] = base["value"]
                    ref["port_single_thread_qps"] = base["single_thread_qps"]
                    ref["port"] = base
                    ref["stand_in_overheads"] = {
                        "reference_over_port_all_cores": ref["value"] / base["value"] if base["value"] else None,
                        "reference_over_port_one_core": ref["single_thread_qps"] / base["single_thread_qps"]
                        if base["single_thread_qps"] else None,
                        "note": "port = the oracle's bare loops over contiguous arrays (no pages, no fmgr, no tuplesort), "
                                "threads of one process; reference = the compiled reference over 8 KB pages through the "
                                "stand-in's buffer manager, index_getattr, fmgr, slot and tuplesort, one process per backend"}
                    line["cpu_baseline"] = ref
                    if ref["answers_against_oracle"]["mismatches"]:
                        failures.append("the reference's own scan differs from the oracle on %d of %d queries, first: %s" % (
                            ref["answers_against_oracle"]["mismatches"], pq, ref["answers_against_oracle"]["first"]))
                    log("reference scan: %.0f QPS on %d processes, %.1f on one (%s); port %.0f / %.1f" % (
                        ref["value"], ref["cores"], ref["single_thread_qps"], ref["isa"], base["value"], base["single_thread_qps"]))
                except Exception as e:   # noqa: BLE001  (the port's number stands, the line says why it is not the reference's)
                    base["reference_error"] = repr(e)
                    failures.append("incomplete: cpu_baseline kind 'reference' did not run (the port's number stands): %r" % (e,))
            del hv
            pd, pt = pd.cpu().numpy(), pt.cpu().numpy()
            bad = []
            for i in range(pq):
                wt, wd = answers[i]
                why = topk_equiv(pt[i][:len(wt)].astype(np.uint64).tolist(), pd[i][:len(wt)], wt.tolist(), wd)
                if why:
                    bad.append((i, why))
            line["parity_checked_queries"] = pq
            line["parity"] = {"against": "CPU oracle (oracle/, restated src/ivfscan.c:47-187), same index, same queries",
                              "rule": "row ids identical where the order is determined beyond 1e-5 relative, "
                                      "distances within 1e-5 relative",
                              "mismatches": len(bad)}
            if bad:
                failures.append("parity: %d of %d queries differ from the oracle, first: %r" % (len(bad), pq, bad[0]))
            log("parity: %d mismatches of %d; cpu_baseline %.0f QPS on %d threads" % (len(bad), pq, base["value"], base["cores"]))
        except Exception as e:  # an oracle that cannot run must not pass for parity
            line["cpu_baseline"] = {"value": None, "unit": "queries/s", "cores": 0, "kind": "port",
                                    "sample": "failed: %r" % (e,)}
            line["parity_checked_queries"] = 0
            failures.append("oracle parity / cpu_baseline did not run: %r" % (e,))
    line["mandatory_secs"] = time.perf_counter() - t_program

    # ------------------------------------------------------------- optional sections: children, one at a time
    if single and not (args.no_sweeps and args.no_traffic):
        wanted = [s for s in args.sections.split(",") if s]
        for name in SECTION_ORDER:
            if name not in wanted:
                continue
            if name == "traffic":
                if args.no_traffic:
                    continue
            elif args.no_sweeps or (name in ("configs", "hard", "hnsw", "c3full", "c5full") and args.workload != "headline"):
                continue
            WATCH["section"] = "section " + name
            if time.perf_counter() - t_program > args.budget_secs:
                failures.append("budget: section %s: not started, the run was %d s old (--budget-secs %d)"
                                % (name, time.perf_counter() - t_program, args.budget_secs))
                continue
            if name == "traffic":
                t0 = time.perf_counter()
                traffic, src = live_traffic(args, avg_launch_ms)
                roofline["traffic"] = traffic
                roofline["traffic_source"] = src
                if traffic and stream_bytes > 0:
                    roofline["traffic_over_streamed"] = traffic / (stream_bytes / launches)
                if traffic is None:
                    failures.append("incomplete: traffic: " + src)
                line.setdefault("sections", {})["traffic"] = {"secs": time.perf_counter() - t0}
                continue
            run_section(name, args, SECTION_BUDGET_S[name], line, failures)
        if "cpu_baseline" in line and "cpu_baseline_page_image" in line:
            line["cpu_baseline"]["page_image"] = line.pop("cpu_baseline_page_image")
            line.setdefault("parity", {})["page_built_index_mismatches"] = \
                line["cpu_baseline"]["page_image"]["parity_of_the_page_built_index"]["mismatches"]
        if "batch_sweep" in line:
            line["batch_sweep"][str(args.batch)]["timed_region_qps"] = qps

    if failures:
        line["failures"] = failures
    line["bench_wall_secs"] = time.perf_counter() - t_program   # everything: data, builds, sections, CPU baselines, PMC passes
    WATCH["done"] = True
    if rank == 0:
        emit_line(json_fd, line)
    index.close()
    if comm is not None:
        comm.close()
    ctx.close()
    if sharded:
        dist.destroy_process_group()
    if failures:
        # an optional section that ran out of its budget on a slow box leaves the measurement INCOMPLETE (it is listed in
        # `failures`, prefix "budget:"), not wrong: only everything else -- a parity mismatch, a crashed section, a baseline
        # that did not run -- is a failed run
        hard = [f for f in failures if not str(f).startswith(("budget:", "incomplete:"))]
        log(("FAILED: " if hard else "INCOMPLETE: ") + "; ".join(failures))
        if hard:
            sys.exit(2)


if __name__ == "__main__":
    main()
