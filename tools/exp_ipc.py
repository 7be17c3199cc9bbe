"""round-3 experiment: does the mirror cross processes (hipIpc), from torch's HIP runtime and from /opt/rocm's?"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from pgvector_amd import api, _host, _lib  # noqa: E402

out = {}
rng = np.random.default_rng(3)
n, dim, lists, nq, probes, k = 200000, 256, 100, 64, 5, 10
data = rng.standard_normal((n, dim), dtype=np.float32)
centers = data[rng.choice(n, lists, replace=False)].copy()
ctx = api.Context(0)
lst, _ = api.assign(ctx, api.PGV_L2SQ, api.PGV_F32, dim, centers, data)
order = np.argsort(lst, kind="stable")
vectors = data[order]
tids = (order.astype(np.uint64) + 1000)
offs = np.zeros(lists + 1, dtype=np.int64)
np.cumsum(np.bincount(lst, minlength=lists), out=offs[1:])
ix = api.IvfIndex(ctx, api.PGV_L2SQ, api.PGV_F32, dim, centers, offs, vectors, tids)
queries = data[rng.choice(n, nq, replace=False)] + 0.01
want_d, _, want_t = ix.search_batch(queries, probes, k, want_tid=True)
want_t = np.asarray(want_t)


def check(ans_t, nclients, per):
    bad = 0
    for c in range(nclients):
        for j in range(per):
            qi = (c * 31 + j) % nq
            if not np.array_equal(ans_t[c, j], want_t[qi]):
                bad += 1
    return bad


for label, image in (("owner_is_python_torch_runtime", None), ("owner_is_a_process", "img")):
    for mode, nclients in ((1, 6), (0, 4)):
        key = "%s/mode%d" % (label, mode)
        try:
            img = None
            if image:
                img = _host.write_index_image("pgv_img_%d" % os.getpid(), api.PGV_L2SQ, api.PGV_F32, dim, centers, offs,
                                              vectors, tids)
            t = time.time()
            res, at, ad = _host.run_backend_processes(None if image else ix, queries, probes, k, mode, nclients, 40,
                                                      max_batch=16, max_wait_us=200, lanes=2, server_processes=True,
                                                      verify=True, image_shm=img)
            res["wall_s"] = time.time() - t
            res["wrong_answers"] = check(at, nclients, 40)
            out[key] = res
        except Exception as e:  # noqa: BLE001
            out[key] = {"error": repr(e)}
        finally:
            if image:
                try:
                    os.unlink("/dev/shm/pgv_img_%d" % os.getpid())
                except OSError:
                    pass
# lanes as threads of this process, clients as processes
try:
    res, at, ad = _host.run_backend_processes(ix, queries, probes, k, 1, 6, 40, max_batch=16, max_wait_us=200, lanes=2,
                                              server_processes=False, verify=True)
    res["wrong_answers"] = check(at, 6, 40)
    out["lane_threads_in_owner/mode1"] = res
except Exception as e:  # noqa: BLE001
    out["lane_threads_in_owner/mode1"] = {"error": repr(e)}
print(json.dumps(out, indent=1))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "exp_ipc.json"), "w"), indent=1)
