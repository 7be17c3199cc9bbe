// kernels_tile.hip -- the list scan for rows probed by MANY queries of a batch.
//
// scan_kernel (kernels_scan.hip) keeps a group of <= 8 queries in LDS and streams
// the rows through registers; a list probed by 9..16 queries is streamed twice.
// Here the roles are swapped: a tile of rows is brought into LDS once by the
// asynchronous global->LDS DMA (global_load_lds, no registers involved, the next
// tile in flight while the current one is scored), and every wavefront scores the
// whole tile against ITS OWN two queries, which it holds in registers.  A
// 512-thread workgroup therefore serves up to 16 queries per pass over the rows:
// the body of GetScanItems (src/ivfscan.c:157-173) for 16 backends' queries with
// each index tuple read from HBM once.
//
// Requirements: rows are a whole number of 1 KiB slices (64 lanes x 16 B), i.e.
// dim a multiple of 256 (fp32) / 512 (fp16) -- every BASELINE config.  Other
// shapes use scan_kernel.
#include "pgv_device.h"

namespace pgv {

namespace {

constexpr int kTileThreads = 1024;
constexpr int kTileWaves = kTileThreads / kWave;  // 16: four per SIMD, so a SIMD always has a busy wave
constexpr int kQW = 1;                            // queries per wavefront -> 16 per workgroup
#ifndef PGV_TILE_ABLATE
#define PGV_TILE_ABLATE 0
#endif
#ifndef PGV_TILE_CAP
#define PGV_TILE_CAP 3
#endif
constexpr int kRB = 3;                            // rows scored together (ILP for the LDS reads)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// LDS reads in inline asm.  They are hidden from hipcc on purpose: with a global->LDS DMA
// in flight it cannot prove that a ds_read of the tile being scored does not alias the
// tile being filled, and drains vmcnt(0) before every read -- which would serialise the
// DMA of the next tile behind the scoring of this one.  The buffers are disjoint by
// construction (double buffering, barrier between fill and use), so the only wait these
// reads need is their own lgkmcnt: lds_issue4 starts four reads, lds_wait_* is the single
// wait before the first consumer and names every destination register so that nothing
// that uses them can be scheduled above it (cdna_hip_programming.md 5.7, form ii).
__device__ __forceinline__ void lds_issue3(unsigned a0, unsigned a1, unsigned a2, u32x4 &v0, u32x4 &v1,
                                           u32x4 &v2) {
    asm volatile(
        "ds_read_b128 %0, %3\n\t"
        "ds_read_b128 %1, %4\n\t"
        "ds_read_b128 %2, %5"
        : "=&v"(v0), "=&v"(v1), "=&v"(v2)
        : "v"(a0), "v"(a1), "v"(a2)
        : "memory");
}

template <int NS> __device__ __forceinline__ void lds_wait(u32x4 (&v)[NS][3]);
#define PGV_V3(c) "+v"(v[c][0]), "+v"(v[c][1]), "+v"(v[c][2])
template <> __device__ __forceinline__ void lds_wait<1>(u32x4 (&v)[1][3]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : PGV_V3(0)::"memory");
}
template <> __device__ __forceinline__ void lds_wait<2>(u32x4 (&v)[2][3]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : PGV_V3(0), PGV_V3(1)::"memory");
}
template <> __device__ __forceinline__ void lds_wait<3>(u32x4 (&v)[3][3]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : PGV_V3(0), PGV_V3(1), PGV_V3(2)::"memory");
}
template <> __device__ __forceinline__ void lds_wait<4>(u32x4 (&v)[4][3]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : PGV_V3(0), PGV_V3(1), PGV_V3(2), PGV_V3(3)::"memory");
}
template <> __device__ __forceinline__ void lds_wait<6>(u32x4 (&v)[6][3]) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : PGV_V3(0), PGV_V3(1), PGV_V3(2), PGV_V3(3), PGV_V3(4), PGV_V3(5)::"memory");
}
#undef PGV_V3

// what a wavefront does in one pass over a tile: which query of the task it serves, which of
// that query's servers it is, and how many servers the query has
struct Slot {
    int q, rank, servers;
    int64_t rel;  // out index of (this query, row 0 of the task)
    __device__ __forceinline__ void init(int wave, int n, int first) {
        const int j = wave % n;
        q = first + j;
        rank = wave / n;
        servers = (kTileWaves - 1 - j) / n + 1;
    }
};

template <typename T, int METRIC, int NCH>
__global__ __launch_bounds__(kTileThreads) void tile_scan_kernel(
    const char *__restrict__ rows, const char *__restrict__ queries,
    const ScanTask *__restrict__ tasks, const int *__restrict__ ntasks_ptr,
    int *__restrict__ task_counter, const ScanPair *__restrict__ pairs, float *__restrict__ out,
    int tile_rows) {
    constexpr size_t ROWB = (size_t)NCH * 1024;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // layout: [2][tile_rows * ROWB] row tiles | int task id
    const size_t tile_bytes = (size_t)tile_rows * ROWB;
    int *lds_task = reinterpret_cast<int *>(smem + 2 * tile_bytes);

    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // provably wave-uniform
    const int ntasks = *ntasks_ptr;
#ifdef PGV_TILE_TIMING  // experiment build: where a workgroup's wall time goes (printed for a few of them)
    unsigned long long tm_start = wall_clock64(), tm_setup = 0, tm_score = 0, tm_wait = 0, tm_mark;
    int tm_tasks = 0, tm_tiles = 0;
#define PGV_TM(x) x
#else
#define PGV_TM(x)
#endif

    for (;;) {
        PGV_TM(tm_mark = wall_clock64();)
        if (threadIdx.x == 0) *lds_task = atomicAdd(task_counter, 1);
        __syncthreads();
        const int t = *lds_task;
#ifdef PGV_TILE_TIMING
        if (t >= ntasks) {
            if (threadIdx.x == 0 && (blockIdx.x < 3 || blockIdx.x == 200))
                printf("wg %d: total %llu setup %llu score %llu wait %llu (x10ns) tasks %d tiles %d\n", blockIdx.x,
                       wall_clock64() - tm_start, tm_setup, tm_score, tm_wait, tm_tasks, tm_tiles);
            return;
        }
        tm_tasks++;
#else
        if (t >= ntasks) return;
#endif
        const ScanTask task = tasks[t];

        // Query j of the task is served by wavefront j and, when the task has fewer than 16
        // queries, also by the otherwise idle wavefronts j + n, j + 2n, ...: the servers of a
        // query take the row batches of a tile round-robin.  With 9..12 queries that would leave
        // the single-server queries four batches per tile next to two for the others, so those
        // tasks run two passes per tile instead: queries 0..7 with two servers each, then the
        // remaining 1..4 with 4..16 servers each (three batch times instead of four).  A
        // wavefront keeps the query of each pass in registers for the whole task.
        const int np = task.npairs;
        const bool two_pass = np > 8 && np <= 12;
        Slot slot[2];
        Raw16 qreg[2][NCH];
        slot[0].init(wave, two_pass ? 8 : np, 0);
        slot[1].init(wave, two_pass ? np - 8 : 1, 8);
#pragma unroll
        for (int s = 0; s < 2; s++) {
            if (s == 1 && !two_pass) break;
            const ScanPair pr = pairs[task.pair0 + slot[s].q];
            slot[s].rel = pr.out_rel + task.row0;
            const char *qp = queries + (size_t)pr.query * ROWB + (size_t)lane * sizeof(Raw16);
#pragma unroll
            for (int c = 0; c < NCH; c++) qreg[s][c] = load16(qp + (size_t)c * 1024);
        }

        // DMA one tile: 1 KiB slice i of the tile goes to LDS offset i * 1024 (+ lane * 16,
        // added by the hardware); slices are dealt round-robin to the wavefronts
        auto issue_tile = [&](int ti, char *dst) {
            const int r0 = ti * tile_rows;
            const int nr = task.nrows - r0 < tile_rows ? task.nrows - r0 : tile_rows;
            const char *src = rows + ((size_t)task.row0 + r0) * ROWB + (size_t)lane * sizeof(Raw16);
            for (int i = wave; i < nr * NCH; i += kTileWaves)
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void *)(src + (size_t)i * 1024),
                    (__attribute__((address_space(3))) void *)(dst + (size_t)i * 1024), 16, 0, 0);
        };

        const int ntiles = (task.nrows + tile_rows - 1) / tile_rows;
        issue_tile(0, smem);
        __syncthreads();  // vmcnt(0) + barrier: tile 0 (and the query registers) have landed
        PGV_TM(tm_setup += wall_clock64() - tm_mark;)

        for (int ti = 0; ti < ntiles; ti++) {
            char *cur = smem + (size_t)(ti & 1) * tile_bytes;
            char *nxt = smem + (size_t)((ti + 1) & 1) * tile_bytes;
#if PGV_TILE_ABLATE != 1 && PGV_TILE_ABLATE != 3  // ablation build 1: score without streaming (tools/ablate_tile.sh)
            if (ti + 1 < ntiles) issue_tile(ti + 1, nxt);  // in flight while `cur` is scored
#endif

            const int r_base = ti * tile_rows;
            const int rows_here = task.nrows - r_base < tile_rows ? task.nrows - r_base : tile_rows;
            PGV_TM(tm_mark = wall_clock64(); tm_tiles++;)
            // LDS byte address of this lane's slice of row 0 of the tile being scored
            const unsigned lrow = (unsigned)(size_t)(__attribute__((address_space(3))) char *)cur +
                                  (unsigned)lane * (unsigned)sizeof(Raw16);
            auto score_pass = [&](const Slot &sl, const Raw16 (&q)[NCH]) {
                for (int r0 = sl.rank * kRB; r0 < rows_here; r0 += sl.servers * kRB) {
                    f32x2 acc[kRB];
#pragma unroll
                    for (int i = 0; i < kRB; i++) acc[i] = f32x2{0.f, 0.f};
                    // several slices of the three rows are requested from LDS at once (one wait
                    // per round); rows past the end of a ragged tile read stale LDS and are
                    // never stored.  128 VGPRs per lane bound the slices per round.
                    constexpr int CAP = PGV_TILE_CAP;
                    constexpr int NS = NCH <= CAP ? NCH : (NCH % 4 == 0 && CAP >= 4 ? 4 : (NCH % 3 == 0 && CAP >= 3 ? 3 : (NCH % 2 == 0 ? 2 : 1)));
#pragma unroll
                    for (int h = 0; h < NCH / NS; h++) {
                        u32x4 rv[NS][3];
#pragma unroll
                        for (int c = 0; c < NS; c++) {
                            const unsigned a = lrow + (unsigned)r0 * (unsigned)ROWB + (unsigned)(h * NS + c) * 1024u;
                            lds_issue3(a, a + (unsigned)ROWB, a + 2u * (unsigned)ROWB, rv[c][0], rv[c][1], rv[c][2]);
                        }
                        lds_wait<NS>(rv);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int c = 0; c < NS; c++) {
#pragma unroll
                            for (int i = 0; i < kRB; i++) {
                                Raw16 raw;
#pragma unroll
                                for (int w = 0; w < 4; w++) raw.w[w] = rv[c][i][w];
                                accum_slice2<T, METRIC>(acc[i], raw, q[h * NS + c]);
                            }
                        }
                    }
                    // lanes 0, 1, 2 end up with the totals of rows r0, r0 + 1, r0 + 2
                    const float sum = wave_sum3(acc[0].x + acc[0].y, acc[1].x + acc[1].y, acc[2].x + acc[2].y);
#if PGV_TILE_ABLATE == 3 || PGV_TILE_ABLATE == 4  // no stores (the condition is never true)
                    if (sum == 12345.678f) out[0] = sum;
#else
                    if (lane < kRB && r0 + lane < rows_here) out[sl.rel + r_base + r0 + lane] = finish<METRIC>(sum);
#endif
                }
            };
#if PGV_TILE_ABLATE != 2  // ablation build 2: stream without scoring
            score_pass(slot[0], qreg[0]);
            if (two_pass) score_pass(slot[1], qreg[1]);
#endif
            PGV_TM(tm_score += wall_clock64() - tm_mark; tm_mark = wall_clock64();)
            __syncthreads();  // next tile landed (vmcnt(0)); everyone is done reading `cur`
            PGV_TM(tm_wait += wall_clock64() - tm_mark;)
        }
    }
}

template <typename T, int METRIC, int NCH>
int launch_tile_t(pgv_ctx *ctx, const void *rows, const void *queries, const ScanTask *tasks,
                  const int *ntasks_dev, int ntasks_bound, const ScanPair *pairs, int tile_rows,
                  float *out) {
    if (ntasks_bound <= 0) return PGV_OK;
    PGV_TRY(ctx->counters.ensure(256));
    int *counter = ctx->counters.as<int>();
    PGV_HIP(hipMemsetAsync(counter, 0, sizeof(int), ctx->stream));
    const size_t lds = 2 * (size_t)tile_rows * NCH * 1024 + 16;
    auto kern = tile_scan_kernel<T, METRIC, NCH>;
    PGV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int grid = ctx->num_cus * (int)(160 * 1024 / (lds + 256) > 0 ? 160 * 1024 / (lds + 256) : 1);
    if (grid > ntasks_bound) grid = ntasks_bound;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kTileThreads), lds, ctx->stream,
                       static_cast<const char *>(rows), static_cast<const char *>(queries), tasks,
                       ntasks_dev, counter, pairs, out, tile_rows);
    PGV_HIP(hipGetLastError());
    return PGV_OK;
}

template <typename T, int METRIC>
int launch_tile_n(pgv_ctx *ctx, int nch, const void *rows, const void *queries,
                  const ScanTask *tasks, const int *ntasks_dev, int ntasks_bound,
                  const ScanPair *pairs, int tile_rows, float *out) {
#define PGV_TILE(NCH)                                                                          \
    case NCH:                                                                                  \
        return launch_tile_t<T, METRIC, NCH>(ctx, rows, queries, tasks, ntasks_dev,            \
                                             ntasks_bound, pairs, tile_rows, out)
    switch (nch) {
        PGV_TILE(1);
        PGV_TILE(2);
        PGV_TILE(3);
        PGV_TILE(4);
        PGV_TILE(6);
        PGV_TILE(8);
    }
#undef PGV_TILE
    PGV_FAIL(PGV_ERR_ARG, "tile scan: unsupported row size (%d KiB)", nch);
}

}  // namespace

// rows must be whole 1 KiB slices and one of the instantiated sizes
bool tile_scan_supported(const RowGeom &g) {
    if (g.lpr_log2 != 6 || g.nvec % kWave != 0) return false;
    const int nch = g.nvec / kWave;
    return nch == 1 || nch == 2 || nch == 3 || nch == 4 || nch == 6 || nch == 8;
}

int tile_scan_queries_per_task() { return kTileWaves * kQW; }

// rows per LDS tile: two tiles fill ~150 KiB of the CU's 160 KiB; a multiple of 2 * kRB so
// that two servers of a query split a tile evenly
int tile_scan_tile_rows(const RowGeom &g) {
    const size_t row_bytes = (size_t)g.nvec * sizeof(Raw16);
    int tr = (int)((150 * 1024 / 2) / row_bytes);
    tr = tr / (2 * kRB) * (2 * kRB);
    if (tr > 60) tr = 60;
    if (tr < kRB) tr = kRB;
    return tr;
}

int launch_tile_scan(pgv_ctx *ctx, pgv_metric metric, pgv_dtype dtype, const RowGeom &g,
                     const void *rows, const void *queries, const ScanTask *tasks,
                     const int *ntasks_dev, int ntasks_bound, const ScanPair *pairs, float *out) {
    if (!tile_scan_supported(g)) PGV_FAIL(PGV_ERR_ARG, "tile scan: row shape not supported");
    const int nch = g.nvec / kWave;
    const int tr = tile_scan_tile_rows(g);
#define PGV_TILE_M(T)                                                                           \
    switch (metric) {                                                                           \
        case PGV_L2SQ:                                                                          \
            return launch_tile_n<T, 0>(ctx, nch, rows, queries, tasks, ntasks_dev, ntasks_bound, \
                                       pairs, tr, out);                                         \
        case PGV_NEG_IP:                                                                        \
            return launch_tile_n<T, 1>(ctx, nch, rows, queries, tasks, ntasks_dev, ntasks_bound, \
                                       pairs, tr, out);                                         \
        case PGV_L1:                                                                            \
            return launch_tile_n<T, 2>(ctx, nch, rows, queries, tasks, ntasks_dev, ntasks_bound, \
                                       pairs, tr, out);                                         \
    }
    if (dtype == PGV_F32) {
        PGV_TILE_M(float)
    } else {
        PGV_TILE_M(__half)
    }
#undef PGV_TILE_M
    PGV_FAIL(PGV_ERR_ARG, "tile scan: unknown metric %d", (int)metric);
}

}  // namespace pgv
