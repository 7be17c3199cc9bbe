// kernels_select.hip -- the bookkeeping either side of the streaming kernel:
//   * plan_*: turn "query q probes lists L[q][0..P)" (the output of GetScanLists,
//     src/ivfscan.c:114-115) into list-major work: for every list the queries
//     that probe it, cut into ScanTasks, plus where each (query, list) run of
//     distances lands in the query's output segment -- the order the reference
//     feeds its tuplesort (src/ivfscan.c:134-179).
//   * topk_kernel: the head of the ascending tuplesort stream
//     (src/ivfscan.c:182, :238-247) / the bounded heap of GetScanLists
//     (:76-106): k smallest of a segment by (value, position), exact and
//     deterministic (radix select on order-preserving keys, then a bitonic sort
//     of the k survivors).
#include "pgv_device.h"
#include "pgv_select.h"

namespace pgv {

namespace {

// ---------------------------------------------------------------- planning

__global__ void plan_count_kernel(const int32_t *__restrict__ probe_lists,
                                  const int64_t *__restrict__ list_off, int nq, int probes,
                                  int *__restrict__ cnt, int64_t *__restrict__ probe_off,
                                  int64_t *__restrict__ seg_len) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    int64_t run = 0;
    for (int p = 0; p < probes; p++) {
        const int l = probe_lists[(size_t)q * probes + p];
        probe_off[(size_t)q * probes + p] = run;
        run += list_off[l + 1] - list_off[l];
        atomicAdd(&cnt[l], 1);
    }
    seg_len[q] = run;
}

// single-block exclusive scan of an int64 sequence produced on the fly: every wavefront scans its 64 values with
// shuffles, the 16 wavefront totals are scanned by the first lanes -- two barriers per 1024 values (the
// Hillis-Steele form over LDS took twenty, and a lone workgroup pays for each)
template <typename F>
__device__ void block_exclusive_scan(int n, F value, int64_t *out, int64_t *scratch /*[blockDim / 64 + 1]*/) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    int64_t carry = 0;  // kept identically by every thread
    for (int base = 0; base < n; base += blockDim.x) {
        const int i = base + threadIdx.x;
        const int64_t v = i < n ? value(i) : 0;
        int64_t incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int64_t t = __shfl_up(incl, d);
            if (lane >= d) incl += t;
        }
        if (lane == 63) scratch[wave] = incl;
        __syncthreads();
        int64_t before = 0, total = 0;
        for (int w = 0; w < nwaves; w++) {
            const int64_t t = scratch[w];
            if (w < wave) before += t;
            total += t;
        }
        if (i < n) out[i] = carry + before + incl - v;
        carry += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) out[n] = carry;
    __syncthreads();
}

__global__ __launch_bounds__(1024) void plan_scan_kernel(
    const int *__restrict__ cnt, const int64_t *__restrict__ list_off,
    const int64_t *__restrict__ seg_len, int nq, int nlists, int qt, int rows_per_task,
    int64_t *__restrict__ seg_start, int64_t *__restrict__ pair_start,
    int64_t *__restrict__ task_start, int64_t *__restrict__ totals /*[2]: out elems, tasks*/) {
    __shared__ int64_t scratch[1024 / 64 + 1];
    block_exclusive_scan(nq, [&](int i) { return seg_len[i]; }, seg_start, scratch);
    block_exclusive_scan(nlists, [&](int i) { return (int64_t)cnt[i]; }, pair_start, scratch);
    block_exclusive_scan(
        nlists,
        [&](int i) {
            const int64_t len = list_off[i + 1] - list_off[i];
            const int64_t ng = (cnt[i] + qt - 1) / qt;
            const int64_t nc = (len + rows_per_task - 1) / rows_per_task;
            return ng * nc;
        },
        task_start, scratch);
    if (threadIdx.x == 0) {
        totals[0] = seg_start[nq];
        totals[1] = task_start[nlists];
        *reinterpret_cast<int *>(totals + 2) = (int)task_start[nlists];  // what the scan kernels read
    }
}

__global__ void plan_pairs_kernel(const int32_t *__restrict__ probe_lists,
                                  const int64_t *__restrict__ list_off,
                                  const int64_t *__restrict__ probe_off,
                                  const int64_t *__restrict__ seg_start,
                                  const int64_t *__restrict__ pair_start, int *__restrict__ fill,
                                  int nq, int probes, ScanPair *__restrict__ pairs) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)nq * probes) return;
    const int q = (int)(i / probes);
    const int l = probe_lists[i];
    const int64_t pos = pair_start[l] + atomicAdd(&fill[l], 1);
    ScanPair pr;
    pr.out_rel = seg_start[q] + probe_off[i] - list_off[l];
    pr.query = q;
    pr.pad = 0;
    pairs[pos] = pr;
}

// one thread per task: the list by bisection of task_start (the lists' tasks are consecutive), then chunk-major
// inside the list -- tasks that stream the same rows sit next to each other in the queue
__global__ void plan_tasks_kernel(const int *__restrict__ cnt, const int64_t *__restrict__ list_off,
                                  const int64_t *__restrict__ pair_start,
                                  const int64_t *__restrict__ task_start, int nlists, int qt,
                                  int rows_per_task, ScanTask *__restrict__ tasks) {
    const int64_t ntasks = task_start[nlists];
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < ntasks; t += (int64_t)gridDim.x * blockDim.x) {
        int lo = 0, hi = nlists - 1;  // last list whose first task is <= t (lists without tasks share a start)
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (task_start[mid] <= t)
                lo = mid;
            else
                hi = mid - 1;
        }
        const int l = lo;
        const int64_t len = list_off[l + 1] - list_off[l];
        const int ng = (cnt[l] + qt - 1) / qt;
        const int64_t rel = t - task_start[l];
        const int c = (int)(rel / ng), g = (int)(rel % ng);
        ScanTask task;
        task.row0 = list_off[l] + (int64_t)c * rows_per_task;
        const int64_t left = len - (int64_t)c * rows_per_task;
        task.nrows = (int)(left < rows_per_task ? left : rows_per_task);
        task.pair0 = (int)(pair_start[l] + (int64_t)g * qt);
        const int pl = cnt[l] - g * qt;
        task.npairs = pl < qt ? pl : qt;
        task.pad = ng > 1 ? 1 : 0;  // the chunk is streamed once per query group: worth keeping in the caches
        tasks[t] = task;
    }
}

// float8 ordering: NaN after everything, strict
__device__ __forceinline__ bool dist_before(float a, float b) { return float_to_key(a) < float_to_key(b); }

// ------------------------------------------------------------------- top-k
// block_topk (pgv_select.h) does the work; this kernel runs it for one segment per workgroup

__global__ __launch_bounds__(kSelThreads) void topk_kernel(
    const float *__restrict__ vals, const int64_t *__restrict__ seg_start, int64_t fixed_len,
    int k, int kp, int cap, float *__restrict__ out_val, int64_t *__restrict__ out_pos,
    int32_t *__restrict__ zero_word) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned long long *ent = reinterpret_cast<unsigned long long *>(smem);  // [cap >= kp]
    SelShared *s = reinterpret_cast<SelShared *>(smem + (size_t)cap * 8);

    const int seg = blockIdx.x;
    // a counter the NEXT kernel in the stream starts from zero (the flagged-query count of the exact tail):
    // cleared here instead of by a memset launch of its own
    if (zero_word && seg == 0 && threadIdx.x == 0) *zero_word = 0;
    const int64_t base = seg_start ? seg_start[seg] : (int64_t)seg * fixed_len;
    const int64_t m = seg_start ? seg_start[seg + 1] - base : fixed_len;
    const float *v = vals + base;
    block_topk_auto(v, m, k, kp, cap, ent, s);  // segments up to 20 480 values: one trip to memory, selection from registers

    for (int i = threadIdx.x; i < k; i += kSelThreads) {
        const unsigned long long e = ent[i];
        const bool have = e != ~0ull && (int64_t)i < (m < k ? m : (int64_t)k);
        out_val[(size_t)seg * k + i] = have ? key_to_float((unsigned)(e >> 32)) : INFINITY;
        out_pos[(size_t)seg * k + i] = have ? (int64_t)(unsigned)(e & 0xffffffffu) : -1;
    }
}

// position inside a query's segment -> row slot (and heap TID)
__global__ void positions_to_slots_kernel(const int32_t *__restrict__ probe_lists,
                                          const int64_t *__restrict__ probe_off,
                                          const int64_t *__restrict__ list_off,
                                          const uint64_t *__restrict__ tids, int nq, int probes,
                                          int k, const int64_t *__restrict__ pos,
                                          int64_t *__restrict__ out_slot,
                                          uint64_t *__restrict__ out_tid) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)nq * k) return;
    const int q = (int)(i / k);
    const int64_t p = pos[i];
    int64_t slot = -1;
    if (p >= 0) {
        const int64_t *off = probe_off + (size_t)q * probes;
        int lo = 0, hi = probes - 1;  // last probe whose offset <= p
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (off[mid] <= p)
                lo = mid;
            else
                hi = mid - 1;
        }
        const int l = probe_lists[(size_t)q * probes + lo];
        slot = list_off[l] + (p - off[lo]);
    }
    if (out_slot) out_slot[i] = slot;
    if (out_tid) out_tid[i] = (slot >= 0 && tids) ? tids[slot] : ~0ull;
}

__global__ void iota_slots_kernel(const int32_t *__restrict__ lists,
                                  const int64_t *__restrict__ probe_off,
                                  const int64_t *__restrict__ list_off,
                                  int64_t *__restrict__ out_slot) {
    const int p = blockIdx.x;
    const int l = lists[p];
    const int64_t beg = list_off[l], len = list_off[l + 1] - beg;
    for (int64_t i = threadIdx.x; i < len; i += blockDim.x) out_slot[probe_off[p] + i] = beg + i;
}

// profiling only: add this batch's pair and streamed-row counts to device accumulators
// (read once by pgv_ctx_get_stats; no host round trip inside the timed region)
__global__ __launch_bounds__(256) void plan_stats_kernel(const int *__restrict__ cnt,
                                                         const int64_t *__restrict__ list_off,
                                                         const int64_t *__restrict__ totals, int nlists,
                                                         int qt, double *__restrict__ acc) {
    __shared__ double red[256], red_u[256];
    double rows = 0.0, uniq = 0.0;
    for (int l = threadIdx.x; l < nlists; l += 256) {
        const double len = (double)(list_off[l + 1] - list_off[l]);
        rows += (double)((cnt[l] + qt - 1) / qt) * len;
        uniq += cnt[l] > 0 ? len : 0.0;
    }
    red[threadIdx.x] = rows;
    red_u[threadIdx.x] = uniq;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            red[threadIdx.x] += red[threadIdx.x + s];
            red_u[threadIdx.x] += red_u[threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        acc[0] += (double)totals[0];  // (row, query) pairs
        acc[1] += red[0];             // rows streamed
        acc[5] += red_u[0];           // rows of the lists at least one query probes (what one pass would stream)
    }
}

// the final top-k merge of a sharded scan: per query the k nearest of the R ranks' sorted (distance, tid)
// heads, ascending; ties: lower rank first, then the rank's own order (+inf / ~0 padding sorts last)
__global__ void merge_heads_kernel(const float *__restrict__ dist_all, const uint64_t *__restrict__ tid_all, int nranks,
                                   int nq, int k, float *__restrict__ out_dist, uint64_t *__restrict__ out_tid) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    int at[16];
    for (int r = 0; r < nranks; r++) at[r] = 0;
    for (int i = 0; i < k; i++) {
        int best = -1;
        float bd = INFINITY;
        for (int r = 0; r < nranks; r++) {
            if (at[r] >= k) continue;
            const float d = dist_all[((size_t)r * nq + q) * k + at[r]];
            if (best < 0 || dist_before(d, bd)) {
                best = r;
                bd = d;
            }
        }
        out_dist[(size_t)q * k + i] = bd;
        out_tid[(size_t)q * k + i] = tid_all[((size_t)best * nq + q) * k + at[best]];
        at[best]++;
    }
}

__global__ void cast_pos_kernel(const int64_t *__restrict__ pos, int64_t n,
                                int32_t *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (int32_t)pos[i];
}

}  // namespace

int launch_merge_heads(pgv_ctx *ctx, const float *dist_all, const uint64_t *tid_all, int nranks, int nq, int k,
                       float *out_dist, uint64_t *out_tid) {
    if (nq <= 0) return PGV_OK;
    if (nranks > 16) PGV_FAIL(PGV_ERR_ARG, "merge: more than 16 ranks");
    hipLaunchKernelGGL(merge_heads_kernel, dim3((nq + 127) / 128), dim3(128), 0, ctx->stream, dist_all, tid_all, nranks,
                       nq, k, out_dist, out_tid);
    PGV_HIP(hipGetLastError());
    return PGV_OK;
}

int launch_cast_pos_to_i32(pgv_ctx *ctx, const int64_t *pos, int64_t n, int32_t *out) {
    if (n <= 0) return PGV_OK;
    hipLaunchKernelGGL(cast_pos_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       ctx->stream, pos, n, out);
    PGV_HIP(hipGetLastError());
    return PGV_OK;
}

int launch_plan_batch(pgv_ctx *ctx, const pgv_index *ix, const int32_t *probe_lists, int nq,
                      int probes, int qt, int rows_per_task, bool read_totals, PlanResult *res) {
    const int nlists = ix->nlists;
    const size_t npairs = (size_t)nq * probes;

    // Host-side bounds size the task queue and the output without a readback:
    //   outputs  <= nq * (rows of the `probes` longest lists)
    //   tasks     = sum_l chunks(l) * groups(l),  groups(l) <= cnt_l / qt + 1,  sum_l cnt_l = nq * probes
    //            <= max_chunks * (nq * probes / qt) + sum_l chunks(l)
    int64_t max_chunks = 0, sum_chunks = 0;
    for (int l = 0; l < nlists; l++) {
        const int64_t nc = (ix->h_offsets[l + 1] - ix->h_offsets[l] + rows_per_task - 1) / rows_per_task;
        sum_chunks += nc;
        if (nc > max_chunks) max_chunks = nc;
    }
    res->ntasks_bound = max_chunks * (int64_t)(npairs / qt) + sum_chunks;
    res->out_bound = (int64_t)nq * ix->len_prefix[probes];
    if (res->ntasks_bound > 0x7fffffff) PGV_FAIL(PGV_ERR_ARG, "plan: too many tasks");

    // plan_a: cnt[nlists] | fill[nlists]   (ints, zeroed every call)
    PGV_TRY(ctx->plan_a.ensure(sizeof(int) * 2 * (size_t)nlists));
    // plan_b: probe_off[nq*probes] | seg_len[nq] | seg_start[nq+1] | pair_start[nlists+1]
    //         | task_start[nlists+1] | totals[2] | ntasks (int)
    const size_t nb = npairs + (size_t)nq + (size_t)nq + 1 + 2 * ((size_t)nlists + 1) + 3;
    PGV_TRY(ctx->plan_b.ensure(sizeof(int64_t) * nb));
    PGV_TRY(ctx->pairs.ensure(sizeof(ScanPair) * npairs));
    PGV_TRY(ctx->tasks.ensure(sizeof(ScanTask) * (size_t)(res->ntasks_bound > 0 ? res->ntasks_bound : 1) + 16));
    int *cnt = ctx->plan_a.as<int>();
    int *fill = cnt + nlists;
    int64_t *probe_off = ctx->plan_b.as<int64_t>();
    int64_t *seg_len = probe_off + npairs;
    int64_t *seg_start = seg_len + nq;
    int64_t *pair_start = seg_start + nq + 1;
    int64_t *task_start = pair_start + nlists + 1;
    int64_t *totals = task_start + nlists + 1;
    ScanTask *tasks = ctx->tasks.as<ScanTask>();

    PGV_HIP(hipMemsetAsync(cnt, 0, sizeof(int) * 2 * (size_t)nlists, ctx->stream));
    hipLaunchKernelGGL(plan_count_kernel, dim3((nq + 255) / 256), dim3(256), 0, ctx->stream,
                       probe_lists, ix->list_offsets, nq, probes, cnt, probe_off, seg_len);
    hipLaunchKernelGGL(plan_scan_kernel, dim3(1), dim3(1024), 0, ctx->stream, cnt,
                       ix->list_offsets, seg_len, nq, nlists, qt, rows_per_task, seg_start,
                       pair_start, task_start, totals);
    hipLaunchKernelGGL(plan_pairs_kernel, dim3((unsigned)((npairs + 255) / 256)), dim3(256), 0,
                       ctx->stream, probe_lists, ix->list_offsets, probe_off, seg_start,
                       pair_start, fill, nq, probes, ctx->pairs.as<ScanPair>());
    {
        const int64_t want = (res->ntasks_bound + 255) / 256;
        const int64_t cap = (int64_t)ctx->num_cus * 8;
        hipLaunchKernelGGL(plan_tasks_kernel, dim3((unsigned)(want < 1 ? 1 : (want > cap ? cap : want))), dim3(256), 0,
                           ctx->stream, cnt, ix->list_offsets, pair_start, task_start, nlists, qt, rows_per_task, tasks);
    }
    PGV_HIP(hipGetLastError());

    res->ntasks = res->ntasks_bound;
    res->total_out = res->out_bound;
    if (read_totals) {  // profiling: exact pair / streamed-row counts, accumulated on the device
        PGV_TRY(ctx->stats_dev.ensure(8 * sizeof(double)));
        hipLaunchKernelGGL(plan_stats_kernel, dim3(1), dim3(256), 0, ctx->stream, cnt,
                           ix->list_offsets, totals, nlists, qt, ctx->stats_dev.as<double>());
        PGV_HIP(hipGetLastError());
    }
    res->tasks = tasks;
    res->pairs = ctx->pairs.as<ScanPair>();
    res->ntasks_dev = reinterpret_cast<int *>(totals + 2);
    res->seg_start = seg_start;
    res->probe_off = probe_off;
    return PGV_OK;
}

int launch_topk_segments(pgv_ctx *ctx, const float *vals, const int64_t *seg_start, int nseg,
                         int64_t fixed_len, int k, float *out_val, int64_t *out_pos, int32_t *zero_word) {
    if (nseg <= 0 || k <= 0) return PGV_OK;
    if (k > 4096) PGV_FAIL(PGV_ERR_ARG, "top-k: k = %d exceeds the supported 4096", k);
    int kp = 1;
    while (kp < k) kp <<= 1;
    if (kp < 2) kp = 2;
    const int cap = kp > kFastCap ? kp : kFastCap;
    const size_t lds = (size_t)cap * 8 + sizeof(SelShared);
    hipLaunchKernelGGL(topk_kernel, dim3(nseg), dim3(kSelThreads), lds, ctx->stream, vals,
                       seg_start, fixed_len, k, kp, cap, out_val, out_pos, zero_word);
    PGV_HIP(hipGetLastError());
    return PGV_OK;
}

int launch_positions_to_slots(pgv_ctx *ctx, const pgv_index *ix, const int32_t *probe_lists,
                              const int64_t *probe_off, int nq, int probes, int k,
                              const int64_t *pos, int64_t *out_slot, uint64_t *out_tid) {
    const int64_t total = (int64_t)nq * k;
    if (total <= 0) return PGV_OK;
    hipLaunchKernelGGL(positions_to_slots_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256),
                       0, ctx->stream, probe_lists, probe_off, ix->list_offsets, ix->tids, nq,
                       probes, k, pos, out_slot, out_tid);
    PGV_HIP(hipGetLastError());
    return PGV_OK;
}

int launch_iota_slots(pgv_ctx *ctx, const pgv_index *ix, const int32_t *lists_dev, int nlists,
                      const int64_t *probe_off, int64_t *out_slot) {
    if (nlists <= 0) return PGV_OK;
    hipLaunchKernelGGL(iota_slots_kernel, dim3(nlists), dim3(256), 0, ctx->stream, lists_dev,
                       probe_off, ix->list_offsets, out_slot);
    PGV_HIP(hipGetLastError());
    return PGV_OK;
}

}  // namespace pgv
