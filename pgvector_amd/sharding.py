"""Multi-GPU partitioning of the IVFFlat path as the HARNESS needs it (bench.py, tests): which rank owns which
list, how a rank-local index image is cut out of a global one or assembled from row shards, how per-rank exact
answers merge into a ground truth.

The path's own exchanges live in the C library: pgv_kmeans_sharded (one fused all-reduce per Lloyd iteration) and
pgv_search_batch_sharded (probe-list and top-k all-gathers, device merge) over pgv_comm (RCCL, or the caller's
collectives) -- include/pgv_hip.h.  What is left here is index/tensor bookkeeping on torch tensors plus the one
exchange the BUILD needs that the library does not own: rows move to the rank that owns their list
(src/ivfbuild.c:830-966: parallel workers feed one shared tuplesort; here every rank sorts its own lists).

Scheme
  scan   lists are disjoint, so every list lives on ONE rank.  Which one: by ROWS (default, `plan_owners`): lists
         longest first, each onto the rank that holds the fewest rows so far -- a step ends with the slowest rank, and
         k-means lists are far from equal (c3small, 8 ranks: l % 8 left rank 0 with 24 443 rows against a mean of
         40 000; `balanced` keeps max / mean within a per cent).  `modulo` (l % world) stays selectable.  The map is a
         pure function of the global list sizes, so every rank computes the same one.  Centers are replicated (<= 50 MB).
  build  heap rows and k-means samples are sharded by row; after assignment one all-to-all brings every row to the
         owner of its list, which lays its lists out list-major in heap (global row) order.
"""
import heapq

import torch
import torch.distributed as dist


def world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def grouped():
    """a process group exists: the collectives below run even with ONE rank (bench.py --sharded-path: the N-GPU code
    path rehearsed on one device must issue the same torch.distributed calls)"""
    return dist.is_available() and dist.is_initialized()


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def _stage(t):
    """gloo moves host memory: device tensors take a detour through the CPU (functional
    multi-rank runs on one GPU and the CPU tests); nccl/RCCL works on HBM directly"""
    if t.is_cuda and dist.get_backend() == "gloo":
        return t.cpu()
    return t


def _all_gather(t):
    st = _stage(t.contiguous())
    out = [torch.empty_like(st) for _ in range(world())]
    dist.all_gather(out, st)
    return [o.to(t.device) for o in out]


def _all_reduce_sum(t):
    st = _stage(t)
    dist.all_reduce(st, op=dist.ReduceOp.SUM)
    if st is not t:
        t.copy_(st)
    return t


def owner_of_list(list_ids, world_size, owners=None):
    """rank that stores each list: `owners` (plan_owners' map) when given, else round-robin"""
    if owners is not None:
        return owners.to(list_ids.device)[list_ids]
    return list_ids % world_size


def plan_owners(list_sizes, world_size, policy="balanced"):
    """owner[nlists] from the GLOBAL rows per list.  balanced: longest list first onto the rank with the fewest rows so
    far (ties: lower rank) -- LPT, max load <= 4/3 of the optimum and in practice within a per cent of the mean for
    hundreds of lists per rank; modulo: l % world.  Deterministic: every rank derives the same map."""
    sizes = [int(x) for x in list_sizes.tolist()]
    n = len(sizes)
    if policy == "modulo" or world_size == 1:
        return torch.arange(n, dtype=torch.int64) % world_size
    if policy != "balanced":
        raise ValueError("placement policy %r" % (policy,))
    heap = [(0, r) for r in range(world_size)]
    owners = [0] * n
    for l in sorted(range(n), key=lambda i: (-sizes[i], i)):
        load, r = heapq.heappop(heap)
        owners[l] = r
        heapq.heappush(heap, (load + sizes[l], r))
    return torch.tensor(owners, dtype=torch.int64)


def global_list_sizes(lists_local, nlists):
    """rows per list over all ranks (one all-reduce of nlists counts)"""
    counts = torch.bincount(lists_local.to(torch.int64), minlength=nlists)
    return _all_reduce_sum(counts) if grouped() else counts


def row_shard(n, r, world_size):
    """contiguous row range [lo, hi) of rank r"""
    per = (n + world_size - 1) // world_size
    lo = min(n, r * per)
    return lo, min(n, lo + per)


def local_index_arrays(vectors_sorted, tids_sorted, list_offsets, r, world_size, owners=None):
    """Cut rank r's image out of a global list-major image.

    The local image keeps ALL `nlists` entries in list_offsets (lists owned by
    other ranks become empty), so center ranking -- done against the replicated
    centers -- yields the global probe set and the scan simply finds nothing in
    foreign lists.  Returns (vectors, tids, list_offsets) of the local image.
    """
    nlists = list_offsets.numel() - 1
    lens = list_offsets[1:] - list_offsets[:-1]
    ids = torch.arange(nlists, device=list_offsets.device)
    mine = owner_of_list(ids, world_size, owners) == r
    local_lens = torch.where(mine, lens, torch.zeros_like(lens))
    local_off = torch.zeros(nlists + 1, dtype=torch.int64, device=list_offsets.device)
    local_off[1:] = torch.cumsum(local_lens, 0)
    # rows of owned lists, in list order (they are already contiguous per list)
    row_list = torch.repeat_interleave(ids, lens)
    keep = mine[row_list]
    return vectors_sorted[keep].contiguous(), tids_sorted[keep].contiguous(), local_off


def gather_assignments(local_lists, n, world_size=None):
    """concatenate every rank's list ids (row shards of equal size except the last)"""
    w = world_size or world()
    if w == 1:
        return local_lists
    per = (n + w - 1) // w
    padded = torch.full((per,), -1, dtype=local_lists.dtype, device=local_lists.device)
    padded[: local_lists.numel()] = local_lists
    return torch.cat(_all_gather(padded))[:n]


def exchange_rows(vectors_local, tids_local, lists_local, nlists, owners=None):
    """Row-sharded build -> list-sharded image.  Every rank holds some heap rows (vectors [m x d], global row ids
    tids [m], assigned lists [m]); row i goes to the owner of its list (`owners`: plan_owners' map; None: l % world).  Returns the local image
    (vectors, tids, list_offsets[nlists + 1]): owned lists in list order, rows of a list in global row order
    (the heap order a serial build feeds its tuplesort, src/ivfbuild.c:271-331); foreign lists empty.
    One variable-size all-to-all for the rows and two small ones for tids / list ids."""
    w, dev = world(), vectors_local.device
    lists64 = lists_local.to(torch.int64)
    if grouped():
        dest = owner_of_list(lists64, w, owners)
        order = torch.argsort(dest, stable=True)
        send_counts = torch.bincount(dest, minlength=w)
        recv_counts = torch.empty_like(send_counts)
        sc = _stage(send_counts)
        rc = torch.empty_like(sc)
        dist.all_to_all_single(rc, sc)
        recv_counts = rc.to(dev)
        ss, rs = send_counts.tolist(), recv_counts.tolist()

        def a2a(t):
            st = _stage(t[order].contiguous())
            out = torch.empty((sum(rs),) + tuple(st.shape[1:]), dtype=st.dtype, device=st.device)
            dist.all_to_all_single(out, st, output_split_sizes=rs, input_split_sizes=ss)
            return out.to(dev)
        vectors_local, tids_local, lists64 = a2a(vectors_local), a2a(tids_local), a2a(lists64)
    # list-major, heap order inside a list: sort by (list, global row id)
    order = torch.argsort(tids_local, stable=True)
    order = order[torch.argsort(lists64[order], stable=True)]
    counts = torch.bincount(lists64, minlength=nlists)
    offsets = torch.zeros(nlists + 1, dtype=torch.int64, device=dev)
    offsets[1:] = torch.cumsum(counts, 0)
    return vectors_local[order].contiguous(), tids_local[order].contiguous(), offsets


def merge_exact_topk(local_dist, k):
    """ground truth over row-sharded data: every rank's exact [nq x k] distances (ascending, +inf padded) ->
    the k smallest per query over all ranks, on every rank"""
    if not grouped():
        return local_dist
    d = torch.cat(_all_gather(local_dist), dim=1)
    return torch.sort(d, dim=1).values[:, :k].contiguous()
