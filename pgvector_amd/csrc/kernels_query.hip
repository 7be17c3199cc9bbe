// kernels_query.hip -- ivfflatgettuple's data path for ONE query, nothing but the head read back.
//
// A Postgres backend scans an index for one query at a time (src/ivfscan.c:361-414): the first
// amgettuple call runs GetScanLists (:47-118), then GetScanItems (:123-187) scores every tuple of
// the next `probes` lists and sorts them, and the executor usually pulls only LIMIT k of them.
// The batched path (plan kernels + tile scan + segmented top-k) needs ~10 launches and two host
// round trips for that; here it is a chain of four small launches on one queue with no host
// involvement in between (five where the device is not behind a large PCIe BAR):
//
//   (the host stores the padded query straight into a fine-grained device row: hipDeviceAttributeIsLargeBar;
//    otherwise query_stage_kernel copies it from pinned host memory -- one more launch, ~2 us)
//   query_rank_kernel    distance to every center, spread over all CUs
//   query_lists_kernel   the max_probes nearest (ascending, lower id on ties) -> device memory
//   query_scan_kernel    the ~N*probes/lists rows of the next `probes` lists split evenly over all
//                        workgroups (rows are ~6 KB, a wavefront keeps two in flight): pure
//                        streaming, distances kept in HBM in tuplesort input order
//   query_head_kernel    selects and sorts the head of the stream and writes (distance, slot, tid)
//                        into pinned host memory, the sequence word last -- no list of 10 000
//                        distances crosses PCIe, no host sort, no stream synchronise
//
// A dependent kernel boundary costs ~1.5 us on this chip (MI355X_MICROARCH.md price list), less
// than an in-kernel "last workgroup selects" hand-off (write-through stores, drain, ~1500 tickets,
// acquire: ~5 us measured), and it keeps the scan kernel a clean streaming kernel.
#include "pgv_device.h"
#include "pgv_select.h"

#include <type_traits>

namespace pgv {

namespace {

constexpr int kQThreads = kSelThreads;  // 256: block_topk's geometry
constexpr int kQWaves = kQThreads / kWave;
constexpr int kMaxBatchLists = 256;     // probes per GetScanItems batch the fused path handles
constexpr int kHeadCap = 1024;          // entries block_topk may keep

// ---- scoring: rows of whole 1 KiB slices keep the query in registers and two rows in flight
template <typename T, int METRIC, int NCH> struct WholeRows {
    Raw16 q[NCH];
    __device__ __forceinline__ void load_query(const char *qp, int lane) {
#pragma unroll
        for (int c = 0; c < NCH; c++) q[c] = load16(qp + ((size_t)c * kWave + lane) * sizeof(Raw16));
    }
    // distances of rows a and b: even lanes return a's, odd lanes b's
    __device__ __forceinline__ float pair(const char *a, const char *b, int lane) const {
        Raw16 ra[NCH], rb[NCH];
#pragma unroll
        for (int c = 0; c < NCH; c++) ra[c] = load16(a + ((size_t)c * kWave + lane) * sizeof(Raw16));
#pragma unroll
        for (int c = 0; c < NCH; c++) rb[c] = load16(b + ((size_t)c * kWave + lane) * sizeof(Raw16));
        float sa = 0.f, sb = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; c++) sa = accum_slice<T, METRIC>(sa, ra[c], q[c]);
#pragma unroll
        for (int c = 0; c < NCH; c++) sb = accum_slice<T, METRIC>(sb, rb[c], q[c]);
        return finish<METRIC>(wave_sum2(sa, sb));
    }
};

// any row length: 1 << lg lanes per row, 64 >> lg rows per wavefront step; the total is valid
// in the last lane of each group
template <typename T, int METRIC>
__device__ __forceinline__ float group_distance(const char *row, const char *qp, int nvec, int lane_in, int lg) {
    const int lpr = 1 << lg;
    float acc0 = 0.f, acc1 = 0.f;
    int v = lane_in;
    for (; v + lpr < nvec; v += 2 * lpr) {
        const Raw16 a0 = load16(row + (size_t)v * sizeof(Raw16));
        const Raw16 a1 = load16(row + (size_t)(v + lpr) * sizeof(Raw16));
        acc0 = accum_slice<T, METRIC>(acc0, a0, load16(qp + (size_t)v * sizeof(Raw16)));
        acc1 = accum_slice<T, METRIC>(acc1, a1, load16(qp + (size_t)(v + lpr) * sizeof(Raw16)));
    }
    if (v < nvec) acc0 = accum_slice<T, METRIC>(acc0, load16(row + (size_t)v * sizeof(Raw16)), load16(qp + (size_t)v * sizeof(Raw16)));
    return finish<METRIC>(group_sum_to_last(acc0 + acc1, lg));
}

// rows [first, end) given by a functor row_ptr(j) -> const char*, distances to out[j]
template <typename T, int METRIC, int NCH, typename RowPtr>
__device__ __forceinline__ void score_rows(RowPtr row_ptr, int64_t first, int64_t end, const char *qp, int nvec, int lg,
                                           float *out) {
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    if constexpr (NCH > 0) {
        WholeRows<T, METRIC, NCH> w;
        w.load_query(qp, lane);
        for (int64_t j = first + 2 * wave; j < end; j += 2 * kQWaves) {
            const bool two = j + 1 < end;
            const float d = w.pair(row_ptr(j), row_ptr(two ? j + 1 : j), lane);
            if (lane == 0 || (lane == 1 && two)) out[j + lane] = d;
        }
    } else {
        const int rpw = kWave >> lg;  // rows per wavefront step
        const int g = lane >> lg, lane_in = lane & ((1 << lg) - 1);
        for (int64_t j0 = first + (int64_t)wave * rpw; j0 < end; j0 += (int64_t)kQWaves * rpw) {
            const int64_t j = j0 + g;
            const bool valid = j < end;
            const float d = group_distance<T, METRIC>(row_ptr(valid ? j : end - 1), qp, nvec, lane_in, lg);
            if (valid && lane_in == (1 << lg) - 1) out[j] = d;
        }
    }
}

// ------------------------------------------------------------------- GetScanLists
template <typename T, int METRIC, int NCH>
__global__ __launch_bounds__(kQThreads) void query_rank_kernel(const char *__restrict__ centers, int nlists, int nvec,
                                                               int lg, const char *__restrict__ qp,
                                                               float *__restrict__ cdist, int per) {
    const size_t row_bytes = (size_t)nvec * sizeof(Raw16);
    for (int64_t first = (int64_t)blockIdx.x * per; first < nlists; first += (int64_t)gridDim.x * per) {
        const int64_t end = first + per < nlists ? first + per : nlists;
        score_rows<T, METRIC, NCH>([&](int64_t j) { return centers + (size_t)j * row_bytes; }, first, end, qp, nvec, lg,
                                   cdist);
    }
}

// the bounded heap of GetScanLists (src/ivfscan.c:76-106): max_probes nearest centers, ascending
__global__ __launch_bounds__(kQThreads) void query_lists_kernel(const float *__restrict__ cdist, int nlists,
                                                                int max_probes, int kp, int32_t *__restrict__ out_lists) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned long long *ent = reinterpret_cast<unsigned long long *>(smem);
    SelShared *s = reinterpret_cast<SelShared *>(smem + (size_t)kHeadCap * 8);
    block_topk_auto(cdist, nlists, max_probes, kp, kHeadCap, ent, s);
    for (int i = threadIdx.x; i < max_probes; i += kQThreads) out_lists[i] = (int32_t)(unsigned)(ent[i] & 0xffffffffu);
}

// ------------------------------------------------------------------- GetScanItems
struct QueryHead {        // what the host reads back (pinned host memory)
    long long total;      // tuples in this batch (= tuplesort's input count)
    int count;            // entries below: min(head, total)
    unsigned seq;         // written last: the call this record answers
};

// where each probed list starts in the batch (offs) and its first row slot (lbeg): lists in probe
// order, page-chain order inside -- the order the reference feeds its tuplesort
struct BatchMap {
    int64_t offs[kMaxBatchLists + 1];
    int64_t lbeg[kMaxBatchLists];
    __device__ void build(const int32_t *probe_lists, int nprobes, const int64_t *list_off) {
        if ((int)threadIdx.x < nprobes) {
            const int l = probe_lists[threadIdx.x];
            lbeg[threadIdx.x] = list_off[l];
            offs[threadIdx.x + 1] = list_off[l + 1] - list_off[l];
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            offs[0] = 0;
            for (int p = 0; p < nprobes; p++) offs[p + 1] += offs[p];
        }
        __syncthreads();
    }
    __device__ __forceinline__ int64_t slot_of(int64_t j, int nprobes) const {  // batch position -> row slot
        int lo = 0, hi = nprobes - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (offs[mid] <= j)
                lo = mid;
            else
                hi = mid - 1;
        }
        return lbeg[lo] + (j - offs[lo]);
    }
};

template <typename T, int METRIC, int NCH>
__global__ __launch_bounds__(kQThreads) void query_scan_kernel(
    const char *__restrict__ vectors, const int64_t *__restrict__ list_off, const int32_t *__restrict__ probe_lists,
    int nprobes, int nvec, int lg, const char *__restrict__ qp, float *__restrict__ seg, int per) {
    __shared__ BatchMap map;
    const size_t row_bytes = (size_t)nvec * sizeof(Raw16);
    map.build(probe_lists, nprobes, list_off);
    const int64_t m = map.offs[nprobes];
    // runs of `per` rows per workgroup, round-robin until the batch is covered (the grid was sized
    // from the host's upper bound of m)
    for (int64_t first = (int64_t)blockIdx.x * per; first < m; first += (int64_t)gridDim.x * per) {
        const int64_t end = first + per < m ? first + per : m;
        if (qp) {
            score_rows<T, METRIC, NCH>([&](int64_t j) { return vectors + (size_t)map.slot_of(j, nprobes) * row_bytes; },
                                       first, end, qp, nvec, lg, seg);
        } else {
            // NULL query: ZeroDistance (src/ivfscan.c:192-196), every tuple at distance 0
            for (int64_t j = first + threadIdx.x; j < end; j += kQThreads) seg[j] = 0.f;
        }
    }
}

// entries [skip, skip + count) of the ascending tuplesort stream of the batch whose distances are
// in seg[0, m) (tuplesort_performsort + the executor's pulls, src/ivfscan.c:182, :408-413): the
// head right after the scan (skip = 0), deeper windows when the executor pulls on (one workgroup;
// k = skip + count <= kHeadCap).  Results go to pinned host memory, the sequence word last.
__global__ __launch_bounds__(kQThreads) void query_head_kernel(
    const float *__restrict__ seg, const int64_t *__restrict__ list_off, const uint64_t *__restrict__ tids,
    const int32_t *__restrict__ probe_lists, int nprobes, int skip, int count, int kp, QueryHead *__restrict__ hdr,
    float *__restrict__ head_dist, int64_t *__restrict__ head_slot, uint64_t *__restrict__ head_tid, unsigned seq) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ BatchMap map;
    map.build(probe_lists, nprobes, list_off);
    const int64_t m = map.offs[nprobes];
    unsigned long long *ent = reinterpret_cast<unsigned long long *>(smem);
    SelShared *s = reinterpret_cast<SelShared *>(smem + (size_t)kHeadCap * 8);
    const int k = skip + count;
    block_topk_auto(seg, m, k, kp, kHeadCap, ent, s);
    const int have = (int)(m < k ? m : k) - skip;
    for (int i = threadIdx.x; i < have; i += kQThreads) {
        const unsigned long long e = ent[skip + i];
        const int64_t slot = map.slot_of((int64_t)(unsigned)(e & 0xffffffffu), nprobes);
        head_dist[i] = key_to_float((unsigned)(e >> 32));
        head_slot[i] = slot;
        head_tid[i] = tids ? tids[slot] : ~0ull;
    }
    // every wavefront's stores have left before the record is declared complete; the host polls seq
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        hdr->total = m;
        hdr->count = have > 0 ? have : 0;
        __hip_atomic_store(&hdr->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// the query payload from pinned host memory into its padded device row (one launch on the same
// queue instead of a copy-engine transfer with its cross-queue hand-off)
// (the host side has already padded it: whole 16-byte vectors, one PCIe round trip per thread)
__global__ __launch_bounds__(1024) void query_stage_kernel(const Raw16 *__restrict__ src, Raw16 *__restrict__ dst, int nvec) {
    for (int i = threadIdx.x; i < nvec; i += 1024) dst[i] = src[i];
}

// NULL query: ZeroDistance ranks the first max_probes lists (strict `<` never replaces, src/ivfscan.c:92)
__global__ void query_iota_kernel(int32_t *__restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = i;
}

// ------------------------------------------------------------------- a few queries at a time
// A batch too small to share rows between its queries (nq * probes well below the list count: every probed list
// belongs to one query) gains nothing from the list-major plan of the batched path and pays its dozen launches.
// The same four kernels as above, one grid row (blockIdx.y) per query, results into the caller's device arrays:
// rank, lists, scan, head -- pgv_search_batch for 16 queries in four launches.
template <typename T, int METRIC, int NCH>
__global__ __launch_bounds__(kQThreads) void mq_rank_kernel(const char *__restrict__ centers, int nlists, int nvec, int lg,
                                                            const char *__restrict__ queries, float *__restrict__ cdist,
                                                            int64_t cd_stride, int per) {
    const size_t row_bytes = (size_t)nvec * sizeof(Raw16);
    const char *qp = queries + (size_t)blockIdx.y * row_bytes;
    float *cd = cdist + (size_t)blockIdx.y * cd_stride;
    for (int64_t first = (int64_t)blockIdx.x * per; first < nlists; first += (int64_t)gridDim.x * per) {
        const int64_t end = first + per < nlists ? first + per : nlists;
        score_rows<T, METRIC, NCH>([&](int64_t j) { return centers + (size_t)j * row_bytes; }, first, end, qp, nvec, lg, cd);
    }
}

__global__ __launch_bounds__(kQThreads) void mq_lists_kernel(const float *__restrict__ cdist, int64_t cd_stride, int nlists,
                                                             int max_probes, int kp, int32_t *__restrict__ out_lists,
                                                             float *__restrict__ out_dist) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned long long *ent = reinterpret_cast<unsigned long long *>(smem);
    SelShared *s = reinterpret_cast<SelShared *>(smem + (size_t)kHeadCap * 8);
    const int q = blockIdx.x;
    block_topk_auto(cdist + (size_t)q * cd_stride, nlists, max_probes, kp, kHeadCap, ent, s);
    for (int i = threadIdx.x; i < max_probes; i += kQThreads) {
        out_lists[(size_t)q * max_probes + i] = (int32_t)(unsigned)(ent[i] & 0xffffffffu);
        if (out_dist) out_dist[(size_t)q * max_probes + i] = key_to_float((unsigned)(ent[i] >> 32));
    }
}

template <typename T, int METRIC, int NCH>
__global__ __launch_bounds__(kQThreads) void mq_scan_kernel(
    const char *__restrict__ vectors, const int64_t *__restrict__ list_off, const int32_t *__restrict__ probe_lists,
    int nprobes, int nvec, int lg, const char *__restrict__ queries, float *__restrict__ seg, int64_t seg_stride, int per) {
    __shared__ BatchMap map;
    const size_t row_bytes = (size_t)nvec * sizeof(Raw16);
    const int q = blockIdx.y;
    map.build(probe_lists + (size_t)q * nprobes, nprobes, list_off);
    const int64_t m = map.offs[nprobes];
    const char *qp = queries + (size_t)q * row_bytes;
    float *sg = seg + (size_t)q * seg_stride;
    for (int64_t first = (int64_t)blockIdx.x * per; first < m; first += (int64_t)gridDim.x * per) {
        const int64_t end = first + per < m ? first + per : m;
        score_rows<T, METRIC, NCH>([&](int64_t j) { return vectors + (size_t)map.slot_of(j, nprobes) * row_bytes; },
                                   first, end, qp, nvec, lg, sg);
    }
}

// the head of each query's stream into the caller's [nq x k] arrays (+inf / -1 / ~0 past the tuples there are)
__global__ __launch_bounds__(kQThreads) void mq_head_kernel(
    const float *__restrict__ seg, int64_t seg_stride, const int64_t *__restrict__ list_off,
    const uint64_t *__restrict__ tids, const int32_t *__restrict__ probe_lists, int nprobes, int k, int kp,
    float *__restrict__ out_dist, int64_t *__restrict__ out_slot, uint64_t *__restrict__ out_tid,
    double *__restrict__ stats) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ BatchMap map;
    const int q = blockIdx.x;
    map.build(probe_lists + (size_t)q * nprobes, nprobes, list_off);
    const int64_t m = map.offs[nprobes];
    unsigned long long *ent = reinterpret_cast<unsigned long long *>(smem);
    SelShared *s = reinterpret_cast<SelShared *>(smem + (size_t)kHeadCap * 8);
    block_topk_auto(seg + (size_t)q * seg_stride, m, k, kp, kHeadCap, ent, s);
    const int have = (int)(m < k ? m : k);
    for (int i = threadIdx.x; i < k; i += kQThreads) {
        const size_t o = (size_t)q * k + i;
        if (i < have) {
            const unsigned long long e = ent[i];
            const int64_t slot = map.slot_of((int64_t)(unsigned)(e & 0xffffffffu), nprobes);
            out_dist[o] = key_to_float((unsigned)(e >> 32));
            if (out_slot) out_slot[o] = slot;
            if (out_tid) out_tid[o] = tids ? tids[slot] : ~0ull;
        } else {
            out_dist[o] = INFINITY;
            if (out_slot) out_slot[o] = -1;
            if (out_tid) out_tid[o] = ~0ull;
        }
    }
    if (stats && threadIdx.x == 0) {  // profiling: pairs scored = rows streamed = rows of probed lists (nothing is shared)
        atomicAdd(&stats[0], (double)m);
        atomicAdd(&stats[1], (double)m);
        atomicAdd(&stats[5], (double)m);
    }
}

// ------------------------------------------------------------------- exact tail of the MFMA L2 scan
// mfma_scan_kernel's L2 values are |x|^2 - 2 q.x (the expansion of the distance less its per-query constant |q|^2):
// they pick k' candidates per query, and this
// kernel (one workgroup per query) re-evaluates those with the reference's arithmetic
// (sum of (q - x)^2, src/vector.c:172-185), sorts them the way the tuplesort would (distance, then
// position in the stream) and emits the first k.  A query is flagged for the full exact pass unless
// the k'-th approximate value clears the k-th one by twice the rounding bound of the expansion, i.e.
// unless no row outside the candidates can belong to the head.
constexpr int kRecheckCap = 256;  // k' <= 256: rank_sort_entries' reach

template <typename T>
__global__ __launch_bounds__(kQThreads) void batch_recheck_kernel(
    const char *__restrict__ vectors, const uint64_t *__restrict__ tids, int nvec, int lg,
    const char *__restrict__ queries, int kprime, int k, const float *__restrict__ approx_val,
    const int64_t *__restrict__ cand_pos, const int64_t *__restrict__ cand_slot,
    const int64_t *__restrict__ list_off, const int32_t *__restrict__ probe_lists,
    const int64_t *__restrict__ probe_off, int probes,
    const int64_t *__restrict__ seg_start, int64_t fixed_len,
    const unsigned *__restrict__ row_norm_max, ScanBound bound, int nq, float *__restrict__ out_dist,
    int64_t *__restrict__ out_slot, uint64_t *__restrict__ out_tid, int32_t *__restrict__ out_i32,
    int32_t *__restrict__ flags) {
    __shared__ float exact[kRecheckCap];
    __shared__ __attribute__((aligned(16))) unsigned long long ent[kRecheckCap];
    const int q = blockIdx.x;
    const size_t row_bytes = (size_t)nvec * sizeof(Raw16);
    const int64_t m = seg_start ? seg_start[q + 1] - seg_start[q] : fixed_len;  // rows the candidates were picked from
    const int ncand = (int)(m < kprime ? m : (int64_t)kprime);
    const int kk = ncand < k ? ncand : k;
    // Only candidates within 2 eps of the k-th approximate value can be in the head: the k smallest-approx rows have
    // exact <= approx[k] + eps, every row with approx > approx[k] + 2 eps has exact > approx[k] + eps.  The candidates
    // come sorted (float8 order, NaN last), so the band is a prefix; it is usually k + a few of the k' rows.
    const float *av = approx_val + (size_t)q * kprime;
    // |q|^2 for the bound (the approximate values carry none: it would shift a query's values alike)
    __shared__ float qn_part[kQWaves];
    {
        const char *qrow = queries + (size_t)q * row_bytes;
        float a = 0.f;
        for (int v = threadIdx.x; v < nvec; v += kQThreads) {
            const Raw16 x = load16(qrow + (size_t)v * sizeof(Raw16));
            a = accum_slice<T, 1>(a, x, x);
        }
        for (int m2 = 32; m2 > 0; m2 >>= 1) a += __shfl_xor(a, m2);
        if ((threadIdx.x & (kWave - 1)) == 0) qn_part[threadIdx.x >> 6] = a;
    }
    __syncthreads();
    float qn = 0.f;
    for (int w = 0; w < kQWaves; w++) qn += qn_part[w];
    const float rn = __uint_as_float(*row_norm_max);
    // (pgv_internal.h, ScanBound; the norms computed here and there carry a relative error of a few u themselves:
    // one part in a thousand on top covers it)
    const float cross = 2.f * sqrtf(qn * rn);
    const float eps = 1.001f * (bound.g_sq * (qn + rn + cross) + bound.g_dot * cross + bound.g_norm * rn);
    // NaN / inf anywhere: everything is in the band
    unsigned band = 0u;
    if (kk > 0) {
        const float edge = av[kk - 1] + 2.f * eps;
        band = float_to_key(edge + bound.g_ref * fabsf(edge + qn));
    }
    const int cnt = __syncthreads_count((int)threadIdx.x < ncand && float_to_key(av[threadIdx.x]) <= band);
    // the candidates' row slots: given (the center ranking: a center's position is its id), or worked out here from
    // the positions in the query's segment (the list scan; what positions_to_slots_kernel does for the exact paths)
    __shared__ int64_t slots[kRecheckCap];
    if ((int)threadIdx.x < cnt) {
        const int64_t p = cand_pos[(size_t)q * kprime + threadIdx.x];
        int64_t slot = p;
        if (cand_slot) {
            slot = cand_slot[(size_t)q * kprime + threadIdx.x];
        } else {
            const int64_t *off = probe_off + (size_t)q * probes;
            int lo = 0, hi = probes - 1;  // last probe whose offset <= p
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (off[mid] <= p)
                    lo = mid;
                else
                    hi = mid - 1;
            }
            slot = list_off[probe_lists[(size_t)q * probes + lo]] + (p - off[lo]);
        }
        slots[threadIdx.x] = slot;
    }
    __syncthreads();
    score_rows<T, 0, 0>([&](int64_t j) { return vectors + (size_t)slots[j] * row_bytes; }, 0, cnt,
                        queries + (size_t)q * row_bytes, nvec, lg, exact);
    __syncthreads();
    if ((int)threadIdx.x < kRecheckCap)
        ent[threadIdx.x] = (int)threadIdx.x < cnt
                               ? ((unsigned long long)float_to_key(exact[threadIdx.x]) << 32) |
                                     (unsigned)cand_pos[(size_t)q * kprime + threadIdx.x]
                               : ~0ull;
    __syncthreads();
    // sort by (exact distance, position in the stream): every thread ranks its own entry
    const unsigned long long mine = ent[threadIdx.x];
    const int64_t my_slot = (int)threadIdx.x < cnt ? slots[threadIdx.x] : -1;
    int rank = 0;
    {
        const ulonglong2 *e2 = reinterpret_cast<const ulonglong2 *>(ent);
        const int n2 = (cnt + 1) / 2;
#pragma unroll 8
        for (int j = 0; j < n2; j++) {
            const ulonglong2 o = e2[j];
            rank += o.x < mine ? 1 : 0;
            rank += o.y < mine ? 1 : 0;
        }
    }
    if ((int)threadIdx.x < cnt && rank < k) {
        const size_t o = (size_t)q * k + rank;
        out_dist[o] = key_to_float((unsigned)(mine >> 32));
        if (out_slot) out_slot[o] = my_slot;
        if (out_i32) out_i32[o] = (int32_t)my_slot;
        if (out_tid) out_tid[o] = tids ? tids[my_slot] : ~0ull;
    }
    if ((int)threadIdx.x >= kk && (int)threadIdx.x < k) {  // fewer tuples than the head asked for
        const size_t o = (size_t)q * k + threadIdx.x;
        out_dist[o] = INFINITY;
        if (out_slot) out_slot[o] = -1;
        if (out_i32) out_i32[o] = -1;
        if (out_tid) out_tid[o] = ~0ull;
    }
    if (threadIdx.x == 0) {
        // the band reaches the end of the candidates and there are rows beyond them: one of those may be in it too
        const int32_t flag = (m > ncand && cnt == ncand) ? 1 : 0;
        flags[q] = flag;
        if (flag) flags[nq + 1 + atomicAdd(&flags[nq], 1)] = q;  // [nq] flags | count | list of flagged queries
    }
}

// The flagged queries, start to end in one launch (they are rare -- none on ordinary data -- so the three launches
// this used to be were three dependent kernel boundaries for nothing).  A workgroup takes a flagged query and
//   (1) WIDENS its candidate set: the kWide = 256 smallest approximate values of the segment (still in seg_vals),
//       the rounding band around the k-th of them, the exact distances of the rows in the band.  When the band ends
//       inside the 256 (or the segment has no more rows), the head is complete: sort by (exact distance, position),
//       write the query's output row.  This settles a band that was a few dozen candidates too wide for k' at ~1 %
//       of the cost of (2) (fp16 at 3072 dimensions under the deterministic bound: one query in nine).
//   (2) otherwise scores every row of its segment with the exact form (the whole segment, what the exact kernels would
//       have done), selects the head like topk_kernel and writes the query's output row.
// Workgroups without a flagged query leave at once.
constexpr int kWide = kRecheckCap;

template <typename T>
__global__ __launch_bounds__(kQThreads) void batch_fix_kernel(
    const char *__restrict__ vectors, const int64_t *__restrict__ list_off, const uint64_t *__restrict__ tids, int nvec,
    int lg, const char *__restrict__ queries, const int32_t *__restrict__ probe_lists,
    const int64_t *__restrict__ probe_off, int probes, const int64_t *__restrict__ seg_start, int64_t fixed_len,
    const int32_t *__restrict__ flags, int nq, float *__restrict__ seg_vals, int k, int kp, int cap,
    const unsigned *__restrict__ row_norm_max, ScanBound bound, int widen,
    float *__restrict__ out_dist, int64_t *__restrict__ out_slot, uint64_t *__restrict__ out_tid,
    int32_t *__restrict__ out_i32, double *__restrict__ stats) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned long long *ent = reinterpret_cast<unsigned long long *>(smem);  // [cap >= max(kp, kWide)]
    SelShared *sel = reinterpret_cast<SelShared *>(smem + (size_t)cap * 8);
    __shared__ float exact[kWide];
    __shared__ int64_t slots[kWide];
    __shared__ float qn_part[kQWaves];
    const int nflag = flags[nq];
    const size_t row_bytes = (size_t)nvec * sizeof(Raw16);
    for (int f = blockIdx.x; f < nflag; f += gridDim.x) {
        const int q = flags[nq + 1 + f];
        // probe_lists == null: the rows are one dense run (the centers), every query's segment is all of them
        const int64_t base = probe_lists ? seg_start[q] : (int64_t)q * fixed_len;
        const int64_t m = probe_lists ? seg_start[q + 1] - base : fixed_len;
        const int64_t *off = probe_off + (size_t)q * probes;
        const int32_t *pl = probe_lists + (size_t)q * probes;
        auto slot_of = [&](int64_t j) -> int64_t {  // position in the segment -> row slot
            if (!probe_lists) return j;
            int lo = 0, hi = probes - 1;
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (off[mid] <= j)
                    lo = mid;
                else
                    hi = mid - 1;
            }
            return list_off[pl[lo]] + (j - off[lo]);
        };
        float *v = seg_vals + base;
        const char *qrow = queries + (size_t)q * row_bytes;
        bool settled = false;
        if (widen && k <= kWide / 2) {
            // (1) the kWide smallest approximate values, ascending by (value, position)
            const int nw = (int)(m < kWide ? m : (int64_t)kWide);
            block_topk([v](int64_t i) { return v[i]; }, m, nw, kWide, cap, ent, sel);
            {
                float a = 0.f;
                for (int x = threadIdx.x; x < nvec; x += kQThreads) {
                    const Raw16 e = load16(qrow + (size_t)x * sizeof(Raw16));
                    a = accum_slice<T, 1>(a, e, e);
                }
                for (int m2 = 32; m2 > 0; m2 >>= 1) a += __shfl_xor(a, m2);
                if ((threadIdx.x & (kWave - 1)) == 0) qn_part[threadIdx.x >> 6] = a;
            }
            __syncthreads();
            float qn = 0.f;
            for (int w = 0; w < kQWaves; w++) qn += qn_part[w];
            const float rn = __uint_as_float(*row_norm_max);
            const float cross = 2.f * sqrtf(qn * rn);
            const float eps = 1.001f * (bound.g_sq * (qn + rn + cross) + bound.g_dot * cross + bound.g_norm * rn);
            const int kk = nw < k ? nw : k;
            unsigned band = 0u;
            if (kk > 0) {
                const float edge = key_to_float((unsigned)(ent[kk - 1] >> 32)) + 2.f * eps;
                band = float_to_key(edge + bound.g_ref * fabsf(edge + qn));
            }
            const int cnt = __syncthreads_count((int)threadIdx.x < nw && (unsigned)(ent[threadIdx.x] >> 32) <= band);
            settled = !(m > nw && cnt == nw);  // (block-uniform)
            if (settled) {
                const unsigned my_pos = (int)threadIdx.x < cnt ? (unsigned)(ent[threadIdx.x] & 0xffffffffu) : 0u;
                if ((int)threadIdx.x < cnt) slots[threadIdx.x] = slot_of((int64_t)my_pos);
                __syncthreads();
                score_rows<T, 0, 0>([&](int64_t j) { return vectors + (size_t)slots[j] * row_bytes; }, 0, cnt, qrow, nvec, lg,
                                    exact);
                __syncthreads();
                if ((int)threadIdx.x < kWide)
                    ent[threadIdx.x] = (int)threadIdx.x < cnt
                                           ? ((unsigned long long)float_to_key(exact[threadIdx.x]) << 32) | my_pos
                                           : ~0ull;
                __syncthreads();
                const unsigned long long mine = ent[threadIdx.x];
                int rank = 0;
                for (int j = 0; j < cnt; j++) rank += ent[j] < mine ? 1 : 0;
                const int have = (int)(m < k ? m : (int64_t)k);
                if ((int)threadIdx.x < cnt && rank < k) {
                    const size_t o = (size_t)q * k + rank;
                    const int64_t slot = slots[threadIdx.x];
                    out_dist[o] = key_to_float((unsigned)(mine >> 32));
                    if (out_slot) out_slot[o] = slot;
                    if (out_i32) out_i32[o] = (int32_t)slot;
                    if (out_tid) out_tid[o] = tids ? tids[slot] : ~0ull;
                }
                if ((int)threadIdx.x >= have && (int)threadIdx.x < k) {
                    const size_t o = (size_t)q * k + threadIdx.x;
                    out_dist[o] = INFINITY;
                    if (out_slot) out_slot[o] = -1;
                    if (out_i32) out_i32[o] = -1;
                    if (out_tid) out_tid[o] = ~0ull;
                }
                if (stats && threadIdx.x == 0) atomicAdd(&stats[7], 1.0);  // profiling: settled by the wider candidate set
            }
            __syncthreads();  // ent / exact / slots are reused
        }
        if (settled) continue;
        if (stats && threadIdx.x == 0) atomicAdd(&stats[6], 1.0);  // profiling: queries that took the exact pass
        score_rows<T, 0, 0>([&](int64_t j) { return vectors + (size_t)slot_of(j) * row_bytes; }, 0, m, qrow, nvec, lg, v);
        __threadfence();
        __syncthreads();
        // the values were written by this workgroup's other wavefronts a moment ago: agent-scope loads, so that a
        // line of the vector L1 that was cached before (a neighbouring segment's selection) cannot serve them stale
        block_topk([v](int64_t i) { return __hip_atomic_load(v + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }, m, k, kp,
                   cap, ent, sel);
        for (int i = threadIdx.x; i < k; i += kQThreads) {
            const unsigned long long e = ent[i];
            const bool have = e != ~0ull && (int64_t)i < (m < k ? m : (int64_t)k);
            const int64_t slot = have ? slot_of((int64_t)(unsigned)(e & 0xffffffffu)) : -1;
            out_dist[(size_t)q * k + i] = have ? key_to_float((unsigned)(e >> 32)) : INFINITY;
            if (out_slot) out_slot[(size_t)q * k + i] = slot;
            if (out_i32) out_i32[(size_t)q * k + i] = (int32_t)slot;
            if (out_tid) out_tid[(size_t)q * k + i] = (have && tids) ? tids[slot] : ~0ull;
        }
        __syncthreads();  // ent / sel are reused by the next flagged query
    }
}

constexpr size_t kQueryLds = (size_t)kHeadCap * 8 + sizeof(SelShared);

// rows per workgroup run and workgroups: one wavefront step per run (two rows per wavefront for
// whole-slice rows) so that every row is in flight at once; at most 8 workgroups per CU
void run_geometry(pgv_ctx *ctx, const RowGeom &g, int64_t nrows, int *per, int *grid) {
    const bool whole = g.lpr_log2 == 6 && g.nvec % kWave == 0;
    const int step = kQWaves * (whole ? 2 : (kWave >> g.lpr_log2));
    int64_t runs = (nrows + step - 1) / step;
    const int64_t cap = (int64_t)ctx->num_cus * 8;
    *per = step;
    *grid = (int)(runs < 1 ? 1 : (runs > cap ? cap : runs));
}

int pow2_at_least(int k) {
    int kp = 2;
    while (kp < k) kp <<= 1;
    return kp;
}

// whole-slice row shapes get the register-resident query; everything else the generic loop
template <typename T, int METRIC, typename F> int dispatch_nch(const RowGeom &g, F f) {
    if (g.lpr_log2 == 6 && g.nvec % kWave == 0) {
        switch (g.nvec / kWave) {
            case 1: return f(std::integral_constant<int, 1>());
            case 2: return f(std::integral_constant<int, 2>());
            case 3: return f(std::integral_constant<int, 3>());
            case 4: return f(std::integral_constant<int, 4>());
            case 6: return f(std::integral_constant<int, 6>());
        }
    }
    return f(std::integral_constant<int, 0>());
}

template <typename F> int dispatch_metric(pgv_metric metric, pgv_dtype dtype, F f) {
#define PGV_QM(T)                                                                     \
    switch (metric) {                                                                 \
        case PGV_L2SQ: return f((T *)nullptr, std::integral_constant<int, 0>());      \
        case PGV_NEG_IP: return f((T *)nullptr, std::integral_constant<int, 1>());    \
        case PGV_L1: return f((T *)nullptr, std::integral_constant<int, 2>());        \
    }
    if (dtype == PGV_F32) {
        PGV_QM(float)
    } else {
        PGV_QM(__half)
    }
#undef PGV_QM
    PGV_FAIL(PGV_ERR_ARG, "query: unknown metric %d", (int)metric);
}

}  // namespace

int query_max_batch_lists() { return kMaxBatchLists; }
int query_head_cap() { return kHeadCap; }
size_t query_head_bytes(int head) { return sizeof(QueryHead) + (size_t)head * (4 + 8 + 8) + 64; }

int launch_query_stage(pgv_ctx *ctx, const void *src_pinned, void *dst_dev, int nvec) {
    hipLaunchKernelGGL(query_stage_kernel, dim3(1), dim3(1024), 0, ctx->stream, static_cast<const Raw16 *>(src_pinned),
                       static_cast<Raw16 *>(dst_dev), nvec);
    PGV_HIP(hipGetLastError());
    return PGV_OK;
}

int launch_query_iota(pgv_ctx *ctx, int32_t *out, int n) {
    hipLaunchKernelGGL(query_iota_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, out, n);
    PGV_HIP(hipGetLastError());
    return PGV_OK;
}

// one query against n contiguous rows, out[j] = distance to row j: the streaming form of query_rank_kernel (whole
// rows in flight per wavefront, the query in registers) -- a k-means++ round over the samples, pgv_distance_batch
int launch_one_query_rows(pgv_ctx *ctx, pgv_metric metric, pgv_dtype dtype, const RowGeom &g, const void *rows, int n,
                          const void *q_dev, float *out) {
    if (n <= 0) return PGV_OK;
    int per, grid;
    run_geometry(ctx, g, n, &per, &grid);
    return dispatch_metric(metric, dtype, [&](auto *tp, auto mc) {
        using T = std::remove_pointer_t<decltype(tp)>;
        constexpr int M = decltype(mc)::value;
        return dispatch_nch<T, M>(g, [&](auto nc) {
            constexpr int NCH = decltype(nc)::value;
            hipLaunchKernelGGL((query_rank_kernel<T, M, NCH>), dim3(grid), dim3(kQThreads), 0, ctx->stream,
                               static_cast<const char *>(rows), n, g.nvec, g.lpr_log2, static_cast<const char *>(q_dev),
                               out, per);
            PGV_HIP(hipGetLastError());
            return PGV_OK;
        });
    });
}

int launch_query_rank(pgv_ctx *ctx, const pgv_index *ix, const void *q_dev, float *cdist, int max_probes,
                      int32_t *out_lists) {
    const RowGeom &g = ix->geom;
    int per, grid;
    run_geometry(ctx, g, ix->nlists, &per, &grid);
    const int kp = pow2_at_least(max_probes);
    PGV_TRY(dispatch_metric(ix->metric, ix->dtype, [&](auto *tp, auto mc) {
        using T = std::remove_pointer_t<decltype(tp)>;
        constexpr int M = decltype(mc)::value;
        return dispatch_nch<T, M>(g, [&](auto nc) {
            constexpr int NCH = decltype(nc)::value;
            hipLaunchKernelGGL((query_rank_kernel<T, M, NCH>), dim3(grid), dim3(kQThreads), 0, ctx->stream,
                               static_cast<const char *>(ix->centers), ix->nlists, g.nvec, g.lpr_log2,
                               static_cast<const char *>(q_dev), cdist, per);
            PGV_HIP(hipGetLastError());
            return PGV_OK;
        });
    }));
    hipLaunchKernelGGL(query_lists_kernel, dim3(1), dim3(kQThreads), kQueryLds, ctx->stream, cdist, ix->nlists,
                       max_probes, kp, out_lists);
    PGV_HIP(hipGetLastError());
    return PGV_OK;
}

int launch_query_scan(pgv_ctx *ctx, const pgv_index *ix, const void *q_dev, const int32_t *probe_lists, int nprobes,
                      int64_t rows_bound, float *seg) {
    const RowGeom &g = ix->geom;
    int per, grid;
    run_geometry(ctx, g, rows_bound, &per, &grid);
    return dispatch_metric(ix->metric, ix->dtype, [&](auto *tp, auto mc) {
        using T = std::remove_pointer_t<decltype(tp)>;
        constexpr int M = decltype(mc)::value;
        return dispatch_nch<T, M>(g, [&](auto nc) {
            constexpr int NCH = decltype(nc)::value;
            hipLaunchKernelGGL((query_scan_kernel<T, M, NCH>), dim3(grid), dim3(kQThreads), 0, ctx->stream,
                               static_cast<const char *>(ix->vectors), ix->list_offsets, probe_lists, nprobes, g.nvec,
                               g.lpr_log2, static_cast<const char *>(q_dev), seg, per);
            PGV_HIP(hipGetLastError());
            return PGV_OK;
        });
    });
}

int launch_query_head(pgv_ctx *ctx, const pgv_index *ix, const float *seg, const int32_t *probe_lists, int nprobes,
                      int skip, int count, void *head_rec, unsigned seq) {
    const int kp = pow2_at_least(skip + count);
    QueryHead *hdr = static_cast<QueryHead *>(head_rec);
    char *base = static_cast<char *>(head_rec) + 64;
    int64_t *h_slot = reinterpret_cast<int64_t *>(base);
    uint64_t *h_tid = reinterpret_cast<uint64_t *>(base + (size_t)count * 8);
    float *h_dist = reinterpret_cast<float *>(base + (size_t)count * 16);
    hipLaunchKernelGGL(query_head_kernel, dim3(1), dim3(kQThreads), kQueryLds, ctx->stream, seg, ix->list_offsets,
                       ix->tids, probe_lists, nprobes, skip, count, kp, hdr, h_dist, h_slot, h_tid, seq);
    PGV_HIP(hipGetLastError());
    return PGV_OK;
}

// a few queries at a time (see mq_*_kernel): GetScanLists for nq staged queries
int launch_multi_rank(pgv_ctx *ctx, const pgv_index *ix, const void *q_dev, int nq, float *cdist, int64_t cd_stride,
                      int max_probes, int32_t *out_lists, float *out_dist) {
    const RowGeom &g = ix->geom;
    int per, grid;
    run_geometry(ctx, g, ix->nlists, &per, &grid);
    const int kp = pow2_at_least(max_probes);
    PGV_TRY(dispatch_metric(ix->metric, ix->dtype, [&](auto *tp, auto mc) {
        using T = std::remove_pointer_t<decltype(tp)>;
        constexpr int M = decltype(mc)::value;
        return dispatch_nch<T, M>(g, [&](auto nc) {
            constexpr int NCH = decltype(nc)::value;
            hipLaunchKernelGGL((mq_rank_kernel<T, M, NCH>), dim3(grid, nq), dim3(kQThreads), 0, ctx->stream,
                               static_cast<const char *>(ix->centers), ix->nlists, g.nvec, g.lpr_log2,
                               static_cast<const char *>(q_dev), cdist, cd_stride, per);
            PGV_HIP(hipGetLastError());
            return PGV_OK;
        });
    }));
    hipLaunchKernelGGL(mq_lists_kernel, dim3(nq), dim3(kQThreads), kQueryLds, ctx->stream, cdist, cd_stride, ix->nlists,
                       max_probes, kp, out_lists, out_dist);
    PGV_HIP(hipGetLastError());
    return PGV_OK;
}

// ... and GetScanItems + the head of the sorted stream, every query over its own lists
int launch_multi_scan(pgv_ctx *ctx, const pgv_index *ix, const void *q_dev, int nq, const int32_t *probe_lists, int nprobes,
                      int64_t rows_bound, float *seg, int64_t seg_stride, int k, float *out_dist, int64_t *out_slot,
                      uint64_t *out_tid) {
    const RowGeom &g = ix->geom;
    int per, grid;
    run_geometry(ctx, g, rows_bound, &per, &grid);
    // nq grid rows share the device: a row needs no more workgroups than keep every CU busy
    const int cap = (ctx->num_cus * 8 + nq - 1) / nq;
    if (grid > cap) grid = cap < 1 ? 1 : cap;
    PGV_TRY(dispatch_metric(ix->metric, ix->dtype, [&](auto *tp, auto mc) {
        using T = std::remove_pointer_t<decltype(tp)>;
        constexpr int M = decltype(mc)::value;
        return dispatch_nch<T, M>(g, [&](auto nc) {
            constexpr int NCH = decltype(nc)::value;
            hipLaunchKernelGGL((mq_scan_kernel<T, M, NCH>), dim3(grid, nq), dim3(kQThreads), 0, ctx->stream,
                               static_cast<const char *>(ix->vectors), ix->list_offsets, probe_lists, nprobes, g.nvec,
                               g.lpr_log2, static_cast<const char *>(q_dev), seg, seg_stride, per);
            PGV_HIP(hipGetLastError());
            return PGV_OK;
        });
    }));
    double *stats = (ctx->profiling && ctx->stats_dev.p) ? ctx->stats_dev.as<double>() : nullptr;
    hipLaunchKernelGGL(mq_head_kernel, dim3(nq), dim3(kQThreads), kQueryLds, ctx->stream, seg, seg_stride, ix->list_offsets,
                       ix->tids, probe_lists, nprobes, k, pow2_at_least(k), out_dist, out_slot, out_tid, stats);
    PGV_HIP(hipGetLastError());
    return PGV_OK;
}

int launch_batch_recheck(pgv_ctx *ctx, const ExactRows &xr, const void *q_dev, int nq, int kprime, int k,
                         const float *approx_val, const int64_t *cand_pos, const int64_t *cand_slot,
                         const int64_t *seg_start, int64_t fixed_len, const ScanBound &bound,
                         float *out_dist, int64_t *out_slot, uint64_t *out_tid, int32_t *flags, int32_t *out_i32,
                         const int32_t *probe_lists, const int64_t *probe_off, int probes) {
    if (nq <= 0) return PGV_OK;
    if (kprime > kRecheckCap || k > kprime) PGV_FAIL(PGV_ERR_ARG, "recheck: k' = %d outside k..%d", kprime, kRecheckCap);
#define PGV_RECHECK(T)                                                                                              \
    hipLaunchKernelGGL(batch_recheck_kernel<T>, dim3(nq), dim3(kQThreads), 0, ctx->stream,                           \
                       static_cast<const char *>(xr.vectors), xr.tids, xr.geom.nvec, xr.geom.lpr_log2,               \
                       static_cast<const char *>(q_dev), kprime, k, approx_val, cand_pos, cand_slot, xr.list_offsets, \
                       probe_lists, probe_off, probes, seg_start, fixed_len, xr.norm_max, bound, nq, out_dist, out_slot, out_tid, out_i32, flags)
    if (xr.dtype == PGV_F32)
        PGV_RECHECK(float);
    else
        PGV_RECHECK(__half);
#undef PGV_RECHECK
    PGV_HIP(hipGetLastError());
    return PGV_OK;
}

int launch_batch_fix(pgv_ctx *ctx, const ExactRows &xr, const void *q_dev, int nq, const int32_t *probe_lists,
                     const int64_t *probe_off, int probes, const int64_t *seg_start, int64_t fixed_len,
                     const int32_t *flags, float *seg_vals, int k, const ScanBound &bound, float *out_dist, int64_t *out_slot,
                     uint64_t *out_tid, int32_t *out_i32) {
    if (nq <= 0) return PGV_OK;
    double *stats = (ctx->profiling && ctx->stats_dev.p && probe_lists) ? ctx->stats_dev.as<double>() : nullptr;
    if (k > 4096) PGV_FAIL(PGV_ERR_ARG, "top-k: k = %d exceeds the supported 4096", k);
    int kp = 2;
    while (kp < k) kp <<= 1;
    const int cap = kp > kFastCap ? kp : kFastCap;
    const int widen = ctx->no_widen ? 0 : 1;
    const size_t lds = (size_t)cap * 8 + sizeof(SelShared);
    const int grid = nq < ctx->num_cus ? nq : ctx->num_cus;
#define PGV_FIX(T)                                                                                                   \
    hipLaunchKernelGGL(batch_fix_kernel<T>, dim3(grid), dim3(kQThreads), lds, ctx->stream,                           \
                       static_cast<const char *>(xr.vectors), xr.list_offsets, xr.tids, xr.geom.nvec,                \
                       xr.geom.lpr_log2, static_cast<const char *>(q_dev), probe_lists, probe_off, probes, seg_start, \
                       fixed_len, flags, nq, seg_vals, k, kp, cap, xr.norm_max, bound, widen, out_dist, out_slot, out_tid,    \
                       out_i32, stats)
    if (xr.dtype == PGV_F32)
        PGV_FIX(float);
    else
        PGV_FIX(__half);
#undef PGV_FIX
    PGV_HIP(hipGetLastError());
    return PGV_OK;
}

}  // namespace pgv
