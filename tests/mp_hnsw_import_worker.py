"""child process of test_hnsw_mirror_across_processes: imports an HNSW mirror by handle and searches it.
argv: <npz with handle, queries, dtype, ef, k> <npz to write>"""
import sys

import numpy as np

from pgvector_amd import api


def main():
    job = np.load(sys.argv[1])
    ctx = api.Context(0)
    h = api.Hnsw.from_handle(ctx, job["handle"].tobytes(), int(job["dtype"]))
    elem, dist, scored = h.search(job["queries"], int(job["ef"]), int(job["k"]))
    readonly = 0
    try:
        h.set_graph(8, 0, np.zeros(1, np.int32), np.zeros(2, np.int64), np.zeros(1, np.int32))
    except api.PgvError as e:
        readonly = e.code
    pay = h.get_payload(np.asarray(elem), words=int(job["words"])) if int(job["words"]) else np.zeros((0, 0), np.uint32)
    np.savez(sys.argv[2], elem=np.asarray(elem), dist=np.asarray(dist), readonly=readonly, payload=pay)
    h.close()
    ctx.close()


if __name__ == "__main__":
    main()
