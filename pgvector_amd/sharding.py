"""Multi-GPU partitioning of the IVFFlat path: which rank owns which list, how
a rank-local index image is cut out of the global one, how per-rank top-k
answers are merged, and the k-means iteration with its all-reduce.

Pure index/tensor bookkeeping on torch tensors (CPU or HBM) plus
torch.distributed collectives (backend "nccl" = RCCL over xGMI on the GPU box,
"gloo" in the CPU tests).  Distances are never computed here: the callers pass
in the functions that do (libpgv_hip on a GPU; the tests plug in the oracle).

Scheme
  scan   lists are disjoint, so list l lives on rank l % world.  Centers are
         replicated (<= 50 MB); every rank ranks all centers, scans the probed
         lists it owns and contributes a local top-k; one all-gather of
         k x (distance, tid) per query and a k-way merge finish the query.
  build  k-means samples and heap rows are sharded by row, centers replicated:
         per Lloyd iteration one all-reduce of the per-center fp32 sums [k x d],
         counts [k] and the change counter; assignment of heap rows needs no
         collective beyond gathering the list ids.
"""
import torch
import torch.distributed as dist


def world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


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


def owner_of_list(list_ids, world_size):
    """rank that stores each list (round-robin keeps sizes balanced for k-means lists)"""
    return list_ids % world_size


def row_shard(n, r, world_size):
    """contiguous row range [lo, hi) of rank r"""
    per = (n + world_size - 1) // world_size
    lo = min(n, r * per)
    return lo, min(n, lo + per)


def local_index_arrays(vectors_sorted, tids_sorted, list_offsets, r, world_size):
    """Cut rank r's image out of a global list-major image.

    The local image keeps ALL `nlists` entries in list_offsets (lists owned by
    other ranks become empty), so center ranking -- done against the replicated
    centers -- yields the global probe set and the scan simply finds nothing in
    foreign lists.  Returns (vectors, tids, list_offsets) of the local image.
    """
    nlists = list_offsets.numel() - 1
    lens = list_offsets[1:] - list_offsets[:-1]
    ids = torch.arange(nlists, device=list_offsets.device)
    mine = owner_of_list(ids, world_size) == r
    local_lens = torch.where(mine, lens, torch.zeros_like(lens))
    local_off = torch.zeros(nlists + 1, dtype=torch.int64, device=list_offsets.device)
    local_off[1:] = torch.cumsum(local_lens, 0)
    # rows of owned lists, in list order (they are already contiguous per list)
    row_list = torch.repeat_interleave(ids, lens)
    keep = mine[row_list]
    return vectors_sorted[keep].contiguous(), tids_sorted[keep].contiguous(), local_off


def gather_probe_lists(local_lists):
    """every rank ranked its own slice of the query batch against the replicated centers;
    concatenate the slices' probe lists in rank order -> [nq_total x probes] on every rank"""
    if world() == 1:
        return local_lists
    return torch.cat(_all_gather(local_lists), dim=0)


def merge_topk(local_dist, local_tid, k):
    """All-gather every rank's [nq x k] (distance, tid) and keep the k nearest per
    query: the final top-k merge (ascending; ties: lower rank first, then the
    rank's own order, which is deterministic).  +inf / -1 padding sorts last."""
    w = world()
    if w == 1:
        return local_dist, local_tid
    d = torch.cat(_all_gather(local_dist), dim=1)
    t = torch.cat(_all_gather(local_tid), dim=1)
    order = torch.sort(d, dim=1, stable=True).indices[:, :k]
    return torch.gather(d, 1, order), torch.gather(t, 1, order)


def allreduce_lloyd(sums, counts, changes):
    """the one exchange of a Lloyd iteration: k*d*4 + k*4 + 8 bytes per rank"""
    if world() > 1:
        _all_reduce_sum(sums)
        _all_reduce_sum(counts)
        _all_reduce_sum(changes)
    return sums, counts, changes


def sharded_kmeans(samples_local, init_centers, partial_fn, finish_fn, max_iterations=500, on_iter=None):
    """Lloyd iterations with sharded samples and replicated centers.

    partial_fn(samples_local, centers, closest) -> (sums [k x d] fp32, counts [k] i32, changes [1] i64)
        (closest updated in place; libpgv_hip's pgv_lloyd_partial)
    finish_fn(sums, counts, iteration) -> centers   (pgv_lloyd_finish; same on every rank)
    Stops like src/ivfkmeans.c:482-483: an iteration other than the first with no
    reassignment anywhere.  Returns (centers, closest_local, iterations).
    """
    centers = init_centers
    closest = torch.full((samples_local.shape[0],), -1, dtype=torch.int32, device=samples_local.device)
    it = 0
    for it in range(max_iterations):
        sums, counts, changes = partial_fn(samples_local, centers, closest)
        local_changes = int(changes.item()) if on_iter else 0
        sums, counts, changes = allreduce_lloyd(sums, counts, changes)
        if on_iter:
            on_iter(it, local_changes, int(changes.item()), int(counts.sum().item()))
        centers = finish_fn(sums, counts, it)
        if int(changes.item()) == 0 and it != 0:
            break
    return centers, closest, it + 1


def gather_assignments(local_lists, n, world_size=None):
    """concatenate every rank's list ids (row shards of equal size except the last)"""
    w = world_size or world()
    if w == 1:
        return local_lists
    per = (n + w - 1) // w
    padded = torch.full((per,), -1, dtype=local_lists.dtype, device=local_lists.device)
    padded[: local_lists.numel()] = local_lists
    return torch.cat(_all_gather(padded))[:n]
