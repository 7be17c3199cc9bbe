#!/usr/bin/env python3
"""tools/exp_pool.py -- the pooler's policy space on the headline index: pooled client PROCESSES x
lanes x linger.  usage: exp_pool.py [nclients,lanes,linger_us,overlap,max_wait_us ...]"""
import json
import os
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from pgvector_amd import _host, api  # noqa: E402


def main():
    rows = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]] or [(256, 3, 120, 0, 50, -1), (256, 1, 120, 0, 50, -1)]
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    ctx = api.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    args = types.SimpleNamespace(workload="headline", probes=0, k=10, seed=0, batch=1024)
    H = bench.headline_setup(args, dev, ctx)
    qh = H.queries[2][:256].cpu().numpy()
    def throttled():
        try:
            return {k: int(v) for k, v in (ln.split() for ln in open("/sys/fs/cgroup/cpu.stat")) if k in ("nr_throttled", "throttled_usec", "usage_usec")}
        except Exception:  # noqa: BLE001
            return {}
    for row in rows:
        nc, lanes, linger, overlap, max_wait = row[:5]
        fanout = row[5] if len(row) > 5 else -1
        if fanout >= 0:
            os.environ["PGV_POOL_WAKE_FANOUT"] = str(fanout)
        else:
            os.environ.pop("PGV_POOL_WAKE_FANOUT", None)
        t0 = throttled()
        os.environ["PGV_POOL_LINGER_US"] = str(linger)
        if overlap:
            os.environ["PGV_POOL_OVERLAP"] = "1"
        else:
            os.environ.pop("PGV_POOL_OVERLAP", None)
        try:
            r = _host.run_backend_processes(H.index, qh, H.probes, H.k, 1, nc, (row[6] if len(row) > 6 else max(60, 24000 // nc)), max_batch=1024,
                                            max_wait_us=max_wait, lanes=lanes, server_processes=True, deadline_s=60.0)
        except Exception as e:  # noqa: BLE001
            r = {"error": repr(e)}
        t1 = throttled()
        r.update(nclients=nc, lanes=lanes, linger_us=linger, overlap=overlap, max_wait_us=max_wait, wake_fanout=fanout,
                 cgroup={k: t1.get(k, 0) - t0.get(k, 0) for k in t1})
        print(json.dumps(r), flush=True)
    H.index.close()
    ctx.close()


if __name__ == "__main__":
    main()
