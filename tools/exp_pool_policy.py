#!/usr/bin/env python3
"""Round-5 experiment on one MI355X: (a) the pooler's batch policy -- one scan at a time + linger (default) against
lanes scanning side by side (PGV_POOL_OVERLAP=1), 2 / 3 lanes, linger 60 / 120 / 250 us -- with 16 / 64 / 256 GPU-less
client PROCESSES; (b) own-context backend processes 1 / 2 / 4 / 8 / 16 with the container's cpu.stat (throttled periods,
user / system time) and the GPU's clocks read around every row.  Headline index (1 M x 1536 fp32, lists 1000, probes 10).
Prints one JSON object."""
import json
import os
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from pgvector_amd import _host, api  # noqa: E402


def cpu_stat():
    try:
        return {a: int(b) for a, b in (ln.split() for ln in open("/sys/fs/cgroup/cpu.stat"))}
    except Exception:  # noqa: BLE001
        return {}


def delta(a, b):
    return {k: b[k] - a[k] for k in b if k in a}


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "pool,procs"
    args = type("A", (), {})()
    args.workload, args.probes, args.k, args.seed, args.batch = "headline", 0, 10, 0, 1024
    dev = torch.device("cuda", 0)
    ctx = api.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    H = bench.headline_setup(args, dev, ctx)
    qh = np.ascontiguousarray(H.queries[0][:256].cpu().numpy())
    out = {"cpu_max": open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else None}
    if "pool" in what:
        out["pool"] = {}
        variants = (("exclusive_2lanes_linger120", {}, 2), ("exclusive_3lanes_linger120", {}, 3),
                    ("exclusive_3lanes_linger250", {"PGV_POOL_LINGER_US": "250"}, 3),
                    ("exclusive_3lanes_linger60", {"PGV_POOL_LINGER_US": "60"}, 3),
                    ("overlap_2lanes (round 4)", {"PGV_POOL_OVERLAP": "1"}, 2))
        if "poolquick" in what:
            variants = (("exclusive_3lanes_linger120", {}, 3), ("exclusive_3lanes_no_linger", {"PGV_POOL_LINGER_US": "0"}, 3))
        for label, env, lanes in variants:
            for kk in ("PGV_POOL_OVERLAP", "PGV_POOL_LINGER_US"):
                os.environ.pop(kk, None)
            os.environ.update(env)
            row = {}
            for nc in ((8, 16, 32, 64, 256) if "poolquick" in what else (16, 64, 256)):
                try:
                    row[str(nc)] = _host.run_backend_processes(H.index, qh, H.probes, 10, 1, nc, max(60, 12000 // nc),
                                                               max_batch=1024, max_wait_us=50, lanes=lanes,
                                                               server_processes=True, deadline_s=40.0)
                except Exception as e:  # noqa: BLE001
                    row[str(nc)] = {"error": repr(e)}
                print(label, nc, row[str(nc)], file=sys.stderr, flush=True)
            out["pool"][label] = row
    if "procs" in what:
        out["procs"] = {}
        counts = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 2, 4, 8, 16]
        if len(sys.argv) > 3:      # KEY=VALUE,... for the CHILDREN only (this process has its HIP runtime up already)
            for kv in sys.argv[3].split(","):
                os.environ[kv.split("=")[0]] = kv.split("=")[1]
            out["children_env"] = sys.argv[3]
        for nb in counts:
            c0, t0 = cpu_stat(), time.time()
            try:
                r = _host.run_backend_processes(H.index, qh, H.probes, 10, 0, nb, 600, deadline_s=40.0)
            except Exception as e:  # noqa: BLE001
                r = {"error": repr(e)}
            r["wall_s"] = time.time() - t0
            r["cpu_stat_delta"] = delta(c0, cpu_stat())
            out["procs"][str(nb)] = r
            print("procs", nb, r, file=sys.stderr, flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
