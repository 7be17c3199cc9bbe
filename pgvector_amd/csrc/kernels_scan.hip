// kernels_scan.hip -- HBM-streaming distance kernels for gfx950.
//
// scan_kernel: the body of GetScanItems (src/ivfscan.c:157-173), GetScanLists
// (:69-74) and k-means++'s distance-to-the-newest-center loop
// (src/ivfkmeans.c:52-71): rows are streamed once from HBM with coalesced
// 16-byte loads and scored against a small group of queries held in LDS.
//
//   * one wavefront (64 lanes) cooperates on rows: `1 << lpr_log2` adjacent
//     lanes own one row, each lane a 16-byte slice per trip, R independent
//     rows per lane in flight; partial sums are folded with DPP at the end.
//   * a workgroup (4 waves) owns a ScanTask = a run of rows of one list x up
//     to QT queries that probe that list; the rows are read from HBM once
//     for all QT queries (query batching is where QPS beyond the single-query
//     roofline comes from).
//   * workgroups are persistent and pull tasks from a device counter, so
//     ragged list lengths balance themselves.
//   * fp32 accumulate in every case, fp16 rows converted exactly
//     (src/halfutils.c:46-78 does the same with cvtph_ps).
#include "pgv_device.h"

#include <type_traits>

namespace pgv {

namespace {

constexpr int kScanThreads = 256;
constexpr int kScanWaves = kScanThreads / kWave;

// waves per SIMD the register allocator must leave room for (kernel variant table below)
constexpr int scan_min_waves(int qt, int r) {
    return qt * r >= 64 ? 2 : (qt * r >= 32 ? 3 : 4);
}

template <typename T, int METRIC, int QT, int R, int THREADS, int PF>
__global__ __launch_bounds__(THREADS, scan_min_waves(QT, R)) void scan_kernel(
    const char *__restrict__ rows, const char *__restrict__ queries,
    const ScanTask *__restrict__ tasks, const int *__restrict__ ntasks_ptr,
    int *__restrict__ task_counter, const ScanPair *__restrict__ pairs, float *__restrict__ out,
    int nvec, int lpr_log2, int nchunks) {
    constexpr int N = VecTraits<T>::N;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // layout: [QT * nvec] Raw16 query slices | [QT] int64 out_rel | int task id
    Raw16 *lds_q = reinterpret_cast<Raw16 *>(smem);
    int64_t *lds_rel = reinterpret_cast<int64_t *>(smem + (size_t)QT * nvec * sizeof(Raw16));
    int *lds_task = reinterpret_cast<int *>(lds_rel + QT);

    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    const int lpr = 1 << lpr_log2;
    const int sub = lane & (lpr - 1);  // which slice of the row
    const int rsub = lane >> lpr_log2; // which row of the wave-load
    const int rpw = kWave >> lpr_log2; // rows per wave-load
    const size_t row_bytes = (size_t)nvec * sizeof(Raw16);
    const int ntasks = *ntasks_ptr;

    for (;;) {
        if (threadIdx.x == 0) *lds_task = atomicAdd(task_counter, 1);
        __syncthreads();
        const int t = *lds_task;
        if (t >= ntasks) {
            // every workgroup ends here exactly once; the last one leaves both words zero for the next launch
            // (no memset launch per scan: k-means++ issues one scan per center)
            if (threadIdx.x == 0 && atomicAdd(task_counter + 1, 1) == (int)gridDim.x - 1) {
                atomicExch(task_counter + 1, 0);
                atomicExch(task_counter, 0);
            }
            return;
        }
        const ScanTask task = tasks[t];

        // stage this task's queries (L2-resident) into LDS
        for (int i = threadIdx.x; i < QT * nvec; i += THREADS) {
            int q = i / nvec, v = i - q * nvec;
            int qq = q < task.npairs ? q : task.npairs - 1;
            int qid = pairs[task.pair0 + qq].query;
            lds_q[i] = load16(queries + (size_t)qid * row_bytes + (size_t)v * sizeof(Raw16));
        }
        if (threadIdx.x < QT) {
            int qq = threadIdx.x < task.npairs ? threadIdx.x : task.npairs - 1;
            lds_rel[threadIdx.x] = pairs[task.pair0 + qq].out_rel;
        }
        __syncthreads();

        const char *task_rows = rows + (size_t)task.row0 * row_bytes;
        for (int b = wave; b * R * rpw < task.nrows; b += THREADS / kWave) {
            float acc[R][QT];
#pragma unroll
            for (int r = 0; r < R; r++)
#pragma unroll
                for (int q = 0; q < QT; q++) acc[r][q] = 0.f;

            const char *rp[R];
#pragma unroll
            for (int r = 0; r < R; r++) {
                int row = b * R * rpw + r * rpw + rsub;
                row = row < task.nrows ? row : task.nrows - 1;  // tail rows are computed, not stored
                rp[r] = task_rows + (size_t)row * row_bytes;
            }

            // Software-pipelined trip over the row: the loads of slice c+1 are in flight
            // while slice c is scored.  Loads are never predicated (a branch around a load
            // makes hipcc drain vmcnt per load).  Whole slices run without any masking; only
            // a ragged last slice (dim not a multiple of the lane group) clamps its address
            // and zeroes the operands of the lanes past the end.
            const int nfull = nvec >> lpr_log2;
            auto score = [&](const Raw16(&rowv)[R], int vq, bool ok, auto masked) {
                float rf[R][N];
#pragma unroll
                for (int r = 0; r < R; r++) {
                    Unpacked<T> u(rowv[r]);
#pragma unroll
                    for (int e = 0; e < N; e++) rf[r][e] = (decltype(masked)::value && !ok) ? 0.f : u.v[e];
                }
#pragma unroll
                for (int q = 0; q < QT; q++) {
                    Unpacked<T> uq(lds_q[q * nvec + vq]);
                    if constexpr (decltype(masked)::value) {
#pragma unroll
                        for (int e = 0; e < N; e++) uq.v[e] = ok ? uq.v[e] : 0.f;
                    }
#pragma unroll
                    for (int r = 0; r < R; r++)
#pragma unroll
                        for (int e = 0; e < N; e++)
                            acc[r][q] = accum<METRIC>(acc[r][q], rf[r][e], uq.v[e]);
                }
            };
            Raw16 cur[R], nxt[R];
            {
                const int vi = sub < nvec ? sub : nvec - 1;
#pragma unroll
                for (int r = 0; r < R; r++) cur[r] = load16(rp[r] + (size_t)vi * sizeof(Raw16));
            }
            for (int c = 0; c < nfull; c++) {
                const int vi = c * lpr + sub;
                int vn = vi + lpr;
                vn = vn < nvec ? vn : nvec - 1;
#pragma unroll
                for (int r = 0; r < R; r++) nxt[r] = load16(rp[r] + (size_t)vn * sizeof(Raw16));
                score(cur, vi, true, std::false_type{});
#pragma unroll
                for (int r = 0; r < R; r++) cur[r] = nxt[r];
            }
            if (nfull < nchunks) {
                const int vi = nfull * lpr + sub;
                const bool ok = vi < nvec;
                score(cur, ok ? vi : nvec - 1, ok, std::true_type{});
            }

#pragma unroll
            for (int r = 0; r < R; r++) {
                const int row = b * R * rpw + r * rpw + rsub;
#pragma unroll
                for (int q = 0; q < QT; q++) {
                    float s = group_sum_to_last(acc[r][q], lpr_log2);
                    if (sub == lpr - 1 && row < task.nrows && q < task.npairs)
                        out[lds_rel[q] + task.row0 + row] = finish<METRIC>(s);
                }
            }
        }
        __syncthreads();  // LDS is rewritten by the next task
    }
}

// Gathered scoring (HNSW candidate batches): pair i = (slot[i], query_of[i]).
// Both operands come from global memory; queries stay L2-resident.
template <typename T, int METRIC, int R>
__global__ __launch_bounds__(kScanThreads) void score_gather_kernel(
    const char *__restrict__ rows, const char *__restrict__ queries,
    const int32_t *__restrict__ slot, const int32_t *__restrict__ query_of, int64_t npairs,
    float *__restrict__ out, int nvec, int lpr_log2, int nchunks) {
    constexpr int N = VecTraits<T>::N;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    const int lpr = 1 << lpr_log2;
    const int sub = lane & (lpr - 1);
    const int rsub = lane >> lpr_log2;
    const int rpw = kWave >> lpr_log2;
    const size_t row_bytes = (size_t)nvec * sizeof(Raw16);

    const int64_t base = ((int64_t)blockIdx.x * kScanWaves + wave) * R * rpw;
    if (base >= npairs) return;

    float acc[R];
    const char *rp[R];
    const char *qp[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        int64_t p = base + r * rpw + rsub;
        p = p < npairs ? p : npairs - 1;
        acc[r] = 0.f;
        rp[r] = rows + (size_t)slot[p] * row_bytes;
        qp[r] = queries + (size_t)(query_of ? query_of[p] : 0) * row_bytes;
    }
    for (int c = 0; c < nchunks; c++) {
        const int vi = c * lpr + sub;
        const bool ok = vi < nvec;
        const int vc = ok ? vi : nvec - 1;  // never predicate a load (see scan_kernel)
        Raw16 rv[R], qv[R];
#pragma unroll
        for (int r = 0; r < R; r++) {
            rv[r] = load16(rp[r] + (size_t)vc * sizeof(Raw16));
            qv[r] = load16(qp[r] + (size_t)vc * sizeof(Raw16));
        }
#pragma unroll
        for (int r = 0; r < R; r++) {
            Unpacked<T> ur(rv[r]);
            Unpacked<T> uq(qv[r]);
#pragma unroll
            for (int e = 0; e < N; e++)
                acc[r] = accum<METRIC>(acc[r], ok ? ur.v[e] : 0.f, ok ? uq.v[e] : 0.f);
        }
    }
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int64_t p = base + r * rpw + rsub;
        float s = group_sum_to_last(acc[r], lpr_log2);
        if (sub == lpr - 1 && p < npairs) out[p] = finish<METRIC>(s);
    }
}

// Pair distances inside groups of rows (the HNSW build: CheckElementCloser's distances between the candidates of one
// neighbor list, src/hnswutils.c:1040-1059): group g is the rows ids[at .. at + n), wanted are the pairs (u, v < u) for
// u >= from, written u ascending then v from pair_at[g] on.  score_gather_kernel reads both rows of every pair from L2
// -- a 64-candidate list's 2016 pairs are 4032 row reads of 64 rows, and that traffic is what bounds it; here a worker (the
// lanes that share a row) takes a TILE of 4 x 4 pairs: 8 row reads for 16 pairs.  Each pair keeps its own accumulator
// and the element order, the lane split and the cross-lane sum of score_gather_kernel: the values are bit for bit the
// same.  One workgroup per group, its workers take the group's tiles round-robin.
template <typename T, int METRIC>
__global__ __launch_bounds__(kScanThreads) void score_groups_kernel(
    const char *__restrict__ rows, const int32_t *__restrict__ ids, const int64_t *__restrict__ ids_at, int64_t ids_stride,
    const int32_t *__restrict__ n_arr, const int32_t *__restrict__ from_arr, const int64_t *__restrict__ pair_at, int ngroups,
    float *__restrict__ out, int nvec, int lpr_log2, int nchunks) {
    constexpr int N = VecTraits<T>::N;
    constexpr int TU = 4, TV = 4;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    const int lpr = 1 << lpr_log2;
    const int sub = lane & (lpr - 1);
    const int rsub = lane >> lpr_log2;
    const int rpw = kWave >> lpr_log2;
    const int worker = wave * rpw + rsub, nworkers = kScanWaves * rpw;
    const size_t row_bytes = (size_t)nvec * sizeof(Raw16);
    for (int g = blockIdx.x; g < ngroups; g += gridDim.x) {
        const int64_t p0 = pair_at[g];
        if (pair_at[g + 1] == p0) continue;
        const int64_t at = ids_at ? ids_at[g] : (int64_t)g * ids_stride;
        const int n = n_arr ? n_arr[g] : (int)(ids_at[g + 1] - at);
        int from = from_arr ? from_arr[g] : 1;
        if (from < 1) from = 1;
        if (n <= from) continue;
        const int32_t *gi = ids + at;
        const int64_t base = (int64_t)from * (from - 1) / 2;
        int t = 0;  // tiles of this group, in order: u-blocks from `from`, each against the v-blocks below its last row
        for (int u0 = from; u0 < n; u0 += TU) {
            const int ulast = (u0 + TU - 1 < n - 1) ? u0 + TU - 1 : n - 1;
            for (int v0 = 0; v0 < ulast; v0 += TV, t++) {
                if (t % nworkers != worker) continue;
                const char *up[TU], *vp[TV];
#pragma unroll
                for (int i = 0; i < TU; i++) up[i] = rows + (size_t)gi[u0 + i < n ? u0 + i : n - 1] * row_bytes;
#pragma unroll
                for (int j = 0; j < TV; j++) vp[j] = rows + (size_t)gi[v0 + j < n ? v0 + j : n - 1] * row_bytes;
                float acc[TU][TV];
#pragma unroll
                for (int i = 0; i < TU; i++)
#pragma unroll
                    for (int j = 0; j < TV; j++) acc[i][j] = 0.f;
                for (int c = 0; c < nchunks; c++) {
                    const int vi = c * lpr + sub;
                    const bool ok = vi < nvec;
                    const int vc = ok ? vi : nvec - 1;  // never predicate a load (see scan_kernel)
                    Raw16 ur[TU], vr[TV];
#pragma unroll
                    for (int i = 0; i < TU; i++) ur[i] = load16(up[i] + (size_t)vc * sizeof(Raw16));
#pragma unroll
                    for (int j = 0; j < TV; j++) vr[j] = load16(vp[j] + (size_t)vc * sizeof(Raw16));
#pragma unroll
                    for (int i = 0; i < TU; i++) {
                        Unpacked<T> a(ur[i]);
#pragma unroll
                        for (int j = 0; j < TV; j++) {
                            Unpacked<T> b(vr[j]);
#pragma unroll
                            for (int e = 0; e < N; e++) acc[i][j] = accum<METRIC>(acc[i][j], ok ? a.v[e] : 0.f, ok ? b.v[e] : 0.f);
                        }
                    }
                }
#pragma unroll
                for (int i = 0; i < TU; i++)
#pragma unroll
                    for (int j = 0; j < TV; j++) {
                        const float sum = group_sum_to_last(acc[i][j], lpr_log2);
                        const int u = u0 + i, v = v0 + j;
                        if (sub == lpr - 1 && u < n && v < u) out[p0 + (int64_t)u * (u - 1) / 2 - base + v] = finish<METRIC>(sum);
                    }
            }
        }
    }
}

template <typename T, int METRIC, int QT, int R, int THREADS, int PF>
int launch_scan_t(pgv_ctx *ctx, const RowGeom &g, const void *rows, const void *queries,
                  const ScanTask *tasks, const int *ntasks_dev, int ntasks_bound,
                  const ScanPair *pairs, float *out) {
    if (ntasks_bound <= 0) return PGV_OK;
    PGV_TRY(ctx->counters.ensure(256));
    if (!ctx->counters_clean) {
        PGV_HIP(hipMemsetAsync(ctx->counters.p, 0, 256, ctx->stream));
        ctx->counters_clean = true;
    }
    int *counter = ctx->counters.as<int>() + 10;  // words 10, 11: claimed tasks, workgroups done (the kernel re-zeroes them)
    size_t lds = (size_t)QT * g.nvec * sizeof(Raw16) + QT * sizeof(int64_t) + 16;
    // enough resident workgroups to cover HBM latency, never more than there is work
    int per_cu = (int)(160 * 1024 / (lds + 256));
    const int by_threads = 2048 / THREADS;
    if (per_cu > by_threads) per_cu = by_threads;
    if (per_cu < 1) per_cu = 1;
    int grid = ctx->num_cus * per_cu;
    if (grid > ntasks_bound) grid = ntasks_bound;
    auto kern = scan_kernel<T, METRIC, QT, R, THREADS, PF>;
    if (lds > 64 * 1024)
        PGV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(THREADS), lds, ctx->stream,
                       static_cast<const char *>(rows), static_cast<const char *>(queries), tasks,
                       ntasks_dev, counter, pairs, out, g.nvec, g.lpr_log2, g.nchunks);
    PGV_HIP(hipGetLastError());
    return PGV_OK;
}

// Variant table: (queries per task, rows in flight per lane, workgroup size).
// Wide groups need a big LDS image of the queries (QT x row bytes), which limits a
// CU to one or two workgroups -- so those workgroups are made of more waves.
template <typename T, int METRIC>
int launch_scan_m(pgv_ctx *ctx, const RowGeom &g, const void *rows, const void *queries,
                  const ScanTask *tasks, const int *ntasks_dev, int ntasks_bound,
                  const ScanPair *pairs, int qt, float *out) {
#define PGV_SCAN(QT, R, TH, PF) \
    return launch_scan_t<T, METRIC, QT, R, TH, PF>(ctx, g, rows, queries, tasks, ntasks_dev, ntasks_bound, pairs, out)
    switch (qt) {
        case 1:
            PGV_SCAN(1, 8, 256, 1);
        case 2:
            PGV_SCAN(2, 4, 256, 1);
        case 4:
            PGV_SCAN(4, 4, 256, 1);
        case 8:
            PGV_SCAN(8, 4, 256, 1);
        case 16:
            // (16, 8, 512) and (16, 4, 1024) spill
            PGV_SCAN(16, 4, 512, 1);
        default:
            PGV_FAIL(PGV_ERR_ARG, "scan: unsupported query group size %d", qt);
    }
#undef PGV_SCAN
}

template <typename T>
int launch_scan_d(pgv_ctx *ctx, pgv_metric metric, const RowGeom &g, const void *rows,
                  const void *queries, const ScanTask *tasks, const int *ntasks_dev,
                  int ntasks_bound, const ScanPair *pairs, int qt, float *out) {
    switch (metric) {
        case PGV_L2SQ:
            return launch_scan_m<T, 0>(ctx, g, rows, queries, tasks, ntasks_dev, ntasks_bound,
                                       pairs, qt, out);
        case PGV_NEG_IP:
            return launch_scan_m<T, 1>(ctx, g, rows, queries, tasks, ntasks_dev, ntasks_bound,
                                       pairs, qt, out);
        case PGV_L1:
            return launch_scan_m<T, 2>(ctx, g, rows, queries, tasks, ntasks_dev, ntasks_bound,
                                       pairs, qt, out);
    }
    PGV_FAIL(PGV_ERR_ARG, "scan: unknown metric %d", (int)metric);
}

template <typename T, int METRIC>
int launch_gather_t(pgv_ctx *ctx, const RowGeom &g, const void *rows, const void *queries,
                    const int32_t *slot, const int32_t *query_of, int64_t npairs, float *out) {
    constexpr int R = 4;
    if (npairs <= 0) return PGV_OK;
    const int rpw = kWave >> g.lpr_log2;
    const int64_t per_block = (int64_t)kScanWaves * R * rpw;
    int64_t grid = (npairs + per_block - 1) / per_block;
    hipLaunchKernelGGL((score_gather_kernel<T, METRIC, R>), dim3((unsigned)grid),
                       dim3(kScanThreads), 0, ctx->stream, static_cast<const char *>(rows),
                       static_cast<const char *>(queries), slot, query_of, npairs, out, g.nvec,
                       g.lpr_log2, g.nchunks);
    PGV_HIP(hipGetLastError());
    return PGV_OK;
}

template <typename T, int METRIC>
int launch_groups_t(pgv_ctx *ctx, const RowGeom &g, const void *rows, const int32_t *ids, const int64_t *ids_at,
                    int64_t ids_stride, const int32_t *n_arr, const int32_t *from_arr, const int64_t *pair_at, int ngroups,
                    float *out) {
    if (ngroups <= 0) return PGV_OK;
    const int cap = ctx->num_cus * 16;
    hipLaunchKernelGGL((score_groups_kernel<T, METRIC>), dim3(ngroups < cap ? ngroups : cap), dim3(kScanThreads), 0, ctx->stream,
                       static_cast<const char *>(rows), ids, ids_at, ids_stride, n_arr, from_arr, pair_at, ngroups, out, g.nvec,
                       g.lpr_log2, g.nchunks);
    PGV_HIP(hipGetLastError());
    return PGV_OK;
}

}  // namespace

// Lanes per row: the power of two that wastes the fewest lane-trips.
RowGeom row_geom(int dim, pgv_dtype t) {
    RowGeom g;
    g.ld = padded_dim(dim, t);
    g.nvec = g.ld * elem_size(t) / kVecBytes;
    int best_lg = 6;
    long best_cost = -1;
    for (int lg = 6; lg >= 0; lg--) {
        int lpr = 1 << lg;
        int trips = (g.nvec + lpr - 1) / lpr;
        long waste = (long)trips * lpr - g.nvec;
        // wasted lane-trips per row, then prefer wider groups (more bytes per instruction)
        long cost = waste * 64 / lpr;
        if (trips > 64) continue;  // keep the trip count bounded for huge rows
        if (best_cost < 0 || cost < best_cost) {
            best_cost = cost;
            best_lg = lg;
        }
    }
    g.lpr_log2 = best_lg;
    int lpr = 1 << best_lg;
    g.nchunks = (g.nvec + lpr - 1) / lpr;
    return g;
}

// Queries per task: the smallest power of two >= wanted, capped by the LDS image of
// the group (<= 64 KB up to 8 queries, <= 128 KB for 16).  Groups of 16 leave room
// for one workgroup per CU only and measured slower than two passes of 8 (DESIGN.md
// section 6), so 8 is the cap.
int scan_group_size(const RowGeom &g, pgv_dtype, int wanted) {
    constexpr int cap = 8;
    int qt = 16;
    while (qt > 1 && (qt > cap || (size_t)qt * g.nvec * sizeof(Raw16) > (qt > 8 ? 128u : 64u) * 1024 - 256)) qt >>= 1;
    while (qt > 1 && qt / 2 >= wanted) qt >>= 1;
    return qt;
}

int launch_scan(pgv_ctx *ctx, pgv_metric metric, pgv_dtype dtype, const RowGeom &g,
                const void *rows, const void *queries, const ScanTask *tasks,
                const int *ntasks_dev, int ntasks_bound, const ScanPair *pairs, int qt,
                float *out) {
    if (dtype == PGV_F32)
        return launch_scan_d<float>(ctx, metric, g, rows, queries, tasks, ntasks_dev,
                                    ntasks_bound, pairs, qt, out);
    return launch_scan_d<__half>(ctx, metric, g, rows, queries, tasks, ntasks_dev, ntasks_bound,
                                 pairs, qt, out);
}

int launch_score_gather(pgv_ctx *ctx, pgv_metric metric, pgv_dtype dtype, const RowGeom &g,
                        const void *rows, const void *queries, const int32_t *slot,
                        const int32_t *query_of, int64_t npairs, float *out) {
#define PGV_GATHER(T)                                                                         \
    switch (metric) {                                                                         \
        case PGV_L2SQ:                                                                        \
            return launch_gather_t<T, 0>(ctx, g, rows, queries, slot, query_of, npairs, out); \
        case PGV_NEG_IP:                                                                      \
            return launch_gather_t<T, 1>(ctx, g, rows, queries, slot, query_of, npairs, out); \
        case PGV_L1:                                                                          \
            return launch_gather_t<T, 2>(ctx, g, rows, queries, slot, query_of, npairs, out); \
    }
    if (dtype == PGV_F32) {
        PGV_GATHER(float)
    } else {
        PGV_GATHER(__half)
    }
#undef PGV_GATHER
    PGV_FAIL(PGV_ERR_ARG, "score: unknown metric %d", (int)metric);
}

int launch_score_groups(pgv_ctx *ctx, pgv_metric metric, pgv_dtype dtype, const RowGeom &g, const void *rows,
                        const int32_t *ids, const int64_t *ids_at, int64_t ids_stride, const int32_t *n_arr,
                        const int32_t *from_arr, const int64_t *pair_at, int ngroups, float *out) {
#define PGV_GROUPS(T)                                                                                                   \
    switch (metric) {                                                                                                   \
        case PGV_L2SQ:                                                                                                  \
            return launch_groups_t<T, 0>(ctx, g, rows, ids, ids_at, ids_stride, n_arr, from_arr, pair_at, ngroups, out); \
        case PGV_NEG_IP:                                                                                                \
            return launch_groups_t<T, 1>(ctx, g, rows, ids, ids_at, ids_stride, n_arr, from_arr, pair_at, ngroups, out); \
        case PGV_L1:                                                                                                    \
            return launch_groups_t<T, 2>(ctx, g, rows, ids, ids_at, ids_stride, n_arr, from_arr, pair_at, ngroups, out); \
    }
    if (dtype == PGV_F32) {
        PGV_GROUPS(float)
    } else {
        PGV_GROUPS(__half)
    }
#undef PGV_GROUPS
    PGV_FAIL(PGV_ERR_ARG, "score: unknown metric %d", (int)metric);
}

}  // namespace pgv
