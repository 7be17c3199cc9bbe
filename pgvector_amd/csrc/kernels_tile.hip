// kernels_tile.hip -- the list scan for rows probed by MANY queries of a batch.
//
// scan_kernel (kernels_scan.hip) keeps a group of <= 8 queries in LDS and streams
// the rows through registers; a list probed by 9..16 queries is streamed twice.
// Here the roles are swapped: a tile of rows is brought into LDS once by the
// asynchronous global->LDS DMA (global_load_lds, no registers involved, the next
// tile in flight while the current one is scored), and every wavefront scores the
// tile against ITS OWN two queries, which it holds in registers -- each row slice
// read from LDS is used twice, without which the LDS is as busy as the vector ALU.
// A 512-thread workgroup therefore serves up to 16 queries per pass over the rows:
// the body of GetScanItems (src/ivfscan.c:157-173) for 16 backends' queries with
// each index tuple read from HBM once.  Two such workgroups share a CU (half of
// its LDS each): while one sits in its tile barrier or waits for its DMA, the other
// has the ALUs.
//
// Requirements: rows are a whole number of 1 KiB slices (64 lanes x 16 B), i.e.
// dim a multiple of 256 (fp32) / 512 (fp16) -- every BASELINE config.  Other
// shapes use scan_kernel.
//
// (The ablation / per-wavefront timing builds that produced DESIGN.md's "DMA only 1.06 ms,
// scoring only 1.00 ms" figures lived in this file through round 1 -- git history, commit
// 088d32a -- and are gone from the product source.)
#include "pgv_device.h"

#include <cstdlib>

namespace pgv {

namespace {

constexpr int kTileThreads = 512;
constexpr int kTileGroupsPerCu = 1024 / kTileThreads;  // workgroups sharing a CU (and its LDS)
constexpr int kTileWaves = kTileThreads / kWave;       // 8: with two workgroups, four per SIMD
constexpr int kQW = 2;                                  // queries per wavefront -> 16 per workgroup
constexpr int kDmaFront = 1;  // DMA instructions of the next tile issued before the first row is scored

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// LDS reads in inline asm.  They are hidden from hipcc on purpose: with a global->LDS DMA
// in flight it cannot prove that a ds_read of the tile being scored does not alias the
// tile being filled, and drains vmcnt(0) before every read -- which would serialise the
// DMA of the next tile behind the scoring of this one.  The buffers are disjoint by
// construction (double buffering, barrier between fill and use), so the only wait these
// reads need is their own lgkmcnt.  One asm statement starts the reads of all NS slices of a
// row and waits for them, so nothing that uses the registers can be scheduled above the wait
// (cdna_hip_programming.md 5.7).
template <int NS> __device__ __forceinline__ void lds_read_row(unsigned a, u32x4 (&v)[NS]);
template <> __device__ __forceinline__ void lds_read_row<1>(unsigned a, u32x4 (&v)[1]) {
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v[0]) : "v"(a) : "memory");
}
template <> __device__ __forceinline__ void lds_read_row<2>(unsigned a, u32x4 (&v)[2]) {
    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]) : "v"(a) : "memory");
}
template <> __device__ __forceinline__ void lds_read_row<3>(unsigned a, u32x4 (&v)[3]) {
    asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %3 offset:1024\n\tds_read_b128 %2, %3 offset:2048\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]) : "v"(a) : "memory");
}
template <> __device__ __forceinline__ void lds_read_row<4>(unsigned a, u32x4 (&v)[4]) {
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\t"
                 "ds_read_b128 %3, %4 offset:3072\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]) : "v"(a) : "memory");
}
template <> __device__ __forceinline__ void lds_read_row<6>(unsigned a, u32x4 (&v)[6]) {
    asm volatile("ds_read_b128 %0, %6\n\tds_read_b128 %1, %6 offset:1024\n\tds_read_b128 %2, %6 offset:2048\n\t"
                 "ds_read_b128 %3, %6 offset:3072\n\tds_read_b128 %4, %6 offset:4096\n\tds_read_b128 %5, %6 offset:5120\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]) : "v"(a) : "memory");
}

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

    for (;;) {
        if (threadIdx.x == 0) *lds_task = atomicAdd(task_counter, 1);
        __syncthreads();
        const int t = *lds_task;
        if (t >= ntasks) return;
        const ScanTask task = tasks[t];

        // Queries 2j and 2j + 1 of the task form pair j, served by wavefront j and, when the task
        // has fewer than 8 pairs, also by the otherwise idle wavefronts j + n, j + 2n, ...: the
        // servers of a pair take the rows of a tile round-robin.  An odd task's last pair
        // repeats its query and drops the copy.  The two queries live in this wavefront's
        // registers for the whole task.
        const int np = task.npairs;
        const int npr = (np + 1) >> 1;
        const int my_pair = wave % npr;
        const int my_rank = wave / npr;                              // which of its servers this wave is
        const int servers = (kTileWaves - 1 - my_pair) / npr + 1;
        const bool q1_valid = 2 * my_pair + 1 < np;
        Raw16 qreg[kQW][NCH];
        int64_t my_rel = 0;  // out index of (the query this LANE reports: even lanes the first, odd the second; row 0 of the task)
#pragma unroll
        for (int s = 0; s < kQW; s++) {
            const ScanPair pr = pairs[task.pair0 + 2 * my_pair + (s == 1 && q1_valid ? 1 : 0)];
            if ((lane & 1) == s) my_rel = pr.out_rel + task.row0;
            const char *qp = queries + (size_t)pr.query * ROWB + (size_t)lane * sizeof(Raw16);
#pragma unroll
            for (int c = 0; c < NCH; c++) qreg[s][c] = load16(qp + (size_t)c * 1024);
        }

        // DMA: 1 KiB slice i of a tile goes to LDS offset i * 1024 (+ lane * 16, added by the
        // hardware); slices are dealt round-robin to the wavefronts
        auto rows_in = [&](int ti) {
            const int left = task.nrows - ti * tile_rows;
            return left < tile_rows ? left : tile_rows;
        };
        auto tile_src = [&](int ti) {
            return rows + ((size_t)task.row0 + (size_t)ti * tile_rows) * ROWB + (size_t)lane * sizeof(Raw16);
        };
        auto dma_slice = [&](const char *src, char *dst, int i) {
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void *)(src + (size_t)i * 1024),
                (__attribute__((address_space(3))) void *)(dst + (size_t)i * 1024), 16, 0, 0);
        };

        // The distances of a tile leave in ONE store per wavefront, one tile late: lane 2i
        // (2i + 1) collects the first (second) query's distance to the i-th row this wavefront
        // scored, and the store is issued after the next tile's barrier, so that its write
        // acknowledgement is long in when the wavefront next drains vmcnt.
        float res = 0.f;
        const bool lane_reports = (lane & 1) == 0 || q1_valid;
        auto store_tile = [&](int tp) {
            const int row = my_rank + (lane >> 1) * servers;
            if (lane_reports && row < rows_in(tp)) out[my_rel + (int64_t)tp * tile_rows + row] = res;
        };

        const int ntiles = (task.nrows + tile_rows - 1) / tile_rows;
        {
            const char *src = tile_src(0);
            for (int i = wave; i < rows_in(0) * NCH; i += kTileWaves) dma_slice(src, smem, i);
        }
        // tile 0 (and the query registers) have landed: the DMA wait is spelled out, __syncthreads()
        // by itself does not drain an LDS-DMA
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();

        for (int ti = 0; ti < ntiles; ti++) {
            char *cur = smem + (size_t)(ti & 1) * tile_bytes;
            char *nxt = smem + (size_t)((ti + 1) & 1) * tile_bytes;
            if (ti > 0) store_tile(ti - 1);
            // The next tile is streamed while `cur` is scored.  Its DMA instructions are not
            // issued in one burst after the barrier -- with every wavefront doing that at once
            // the vector-memory queue backs up and all of them stall on the issue -- but dealt
            // out between the rows: kDmaFront of them up front, one more per row
            // scored, the rest after the last row.
            int dma_i = wave, dma_n = 0;
            const char *dma_src = nullptr;
            if (ti + 1 < ntiles) {
                dma_n = rows_in(ti + 1) * NCH;
                dma_src = tile_src(ti + 1);
            }
            auto issue_next = [&](int count) {
                for (int c = 0; c < count && dma_i < dma_n; c++, dma_i += kTileWaves) dma_slice(dma_src, nxt, dma_i);
            };
            issue_next(kDmaFront);

            const int rows_here = rows_in(ti);
            // LDS byte address of this lane's slice of row 0 of the tile being scored
            const unsigned lrow = (unsigned)(size_t)(__attribute__((address_space(3))) char *)cur +
                                  (unsigned)lane * (unsigned)sizeof(Raw16);
            int nth = 0;  // rows this wavefront has scored in this tile
            for (int r0 = my_rank; r0 < rows_here; r0 += servers, nth++) {
                issue_next(1);
                u32x4 rv[NCH];  // a whole row per round
                lds_read_row<NCH>(lrow + (unsigned)r0 * (unsigned)ROWB, rv);
                __builtin_amdgcn_sched_barrier(0);
                f32x2 acc0 = {0.f, 0.f}, acc1 = {0.f, 0.f};
#pragma unroll
                for (int c = 0; c < NCH; c++) {
                    Raw16 raw;
#pragma unroll
                    for (int w = 0; w < 4; w++) raw.w[w] = rv[c][w];
                    accum_slice2<T, METRIC>(acc0, raw, qreg[0][c]);
                    accum_slice2<T, METRIC>(acc1, raw, qreg[1][c]);
                }
                // even lanes end up with the first query's total, odd lanes with the second's
                const float sum = wave_sum2(acc0.x + acc0.y, acc1.x + acc1.y);
                if ((lane >> 1) == nth) res = finish<METRIC>(sum);
            }
            issue_next(1 << 20);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // next tile landed (this wavefront's share)
            __syncthreads();                                   // ... everyone's; and everyone is done reading `cur`
        }
        store_tile(ntiles - 1);
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
    }
#undef PGV_TILE
    PGV_FAIL(PGV_ERR_ARG, "tile scan: unsupported row size (%d KiB)", nch);
}

}  // namespace

// rows must be whole 1 KiB slices and one of the instantiated sizes
bool tile_scan_supported(const RowGeom &g) {
    if (g.lpr_log2 != 6 || g.nvec % kWave != 0) return false;
    const int nch = g.nvec / kWave;
    // 8 KiB rows would be 2048 fp32 / 4096 fp16 dimensions, beyond the index limits
    // (src/ivfflat.h:37, src/ivfutils.c:401)
    return nch == 1 || nch == 2 || nch == 3 || nch == 4 || nch == 6;
}

int tile_scan_queries_per_task() { return kTileWaves * kQW; }

// rows per LDS tile: the two tiles of each of a CU's workgroups fill ~150 KiB of its 160 KiB
int tile_scan_tile_rows(const RowGeom &g) {
    const size_t row_bytes = (size_t)g.nvec * sizeof(Raw16);
    int tr = (int)((150 * 1024 / 2 / kTileGroupsPerCu) / row_bytes);
    tr = tr / 2 * 2;
    if (tr > 32) tr = 32;  // a wavefront reports up to tile_rows rows x 2 queries from its 64 lanes
    if (tr < 2) tr = 2;
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
