// pgv_abi_common.h -- what the translation units of the C ABI (pgv_abi*.hip) share: argument checks, staging of
// host-or-device arrays, profiling hooks, the host-planned dense scan, the library's own random source, and the
// prototypes of the few area functions another area calls.  The helpers sit in an anonymous namespace ON PURPOSE: every
// unit gets its own copy (they are small, and dense_scan's cached plan lives on the context, not in the unit).
#pragma once

#include "pgv_internal.h"
#include "pgv_gate.h"
#include <dlfcn.h>
#include <linux/futex.h>
#include <sys/syscall.h>
#include <unistd.h>
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <new>

using namespace pgv;

namespace {

int check_common(pgv_dtype dtype, int dim) {
    if (dtype != PGV_F32 && dtype != PGV_F16) PGV_FAIL(PGV_ERR_ARG, "unknown dtype %d", (int)dtype);
    // VECTOR_MAX_DIM / HALFVEC_MAX_DIM (src/vector.h:10, src/halfvec.h:61)
    if (dim < 1 || dim > 16000) PGV_FAIL(PGV_ERR_DIMS, "dimensions %d outside 1..16000", dim);
    return PGV_OK;
}

int check_metric(pgv_metric m) {
    if (m != PGV_L2SQ && m != PGV_NEG_IP && m != PGV_L1) PGV_FAIL(PGV_ERR_ARG, "unknown metric %d", (int)m);
    return PGV_OK;
}

// rows that cannot stay in the 256 MB last-level cache between two batches anyway (four times its size and up) are
// fetched non-temporally by the MFMA scan; smaller sets keep the default policy and the cache residency it gives them
bool rows_stream_past_caches(const RowGeom &g, pgv_dtype dtype, int64_t nrows) {
    return (size_t)nrows * (size_t)g.ld * elem_size(dtype) >= ((size_t)1 << 30);
}

// rows [n x dim] tightly packed (host or device) -> device rows [n x ld], zero padded.
// When the source already lives on the device with ld == dim it is used in place.
int stage_rows(pgv_ctx *ctx, const void *src, int64_t n, int dim, pgv_dtype dtype,
               const RowGeom &g, DBuf &scratch, const void **out) {
    const size_t es = elem_size(dtype);
    const bool dev = is_device_ptr(src);
    if (dev && g.ld == dim) {
        *out = src;
        return PGV_OK;
    }
    const size_t bytes = (size_t)n * g.ld * es;
    PGV_TRY(scratch.ensure(bytes ? bytes : 16));
    if (n == 0) {
        *out = scratch.p;
        return PGV_OK;
    }
    if (g.ld == dim) {
        PGV_HIP(hipMemcpyAsync(scratch.p, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    } else {
        PGV_HIP(hipMemsetAsync(scratch.p, 0, bytes, ctx->stream));
        PGV_HIP(hipMemcpy2DAsync(scratch.p, (size_t)g.ld * es, src, (size_t)dim * es,
                                 (size_t)dim * es, (size_t)n,
                                 dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice,
                                 ctx->stream));
    }
    if (!dev) PGV_HIP(hipStreamSynchronize(ctx->stream));  // the caller may reuse src right away
    *out = scratch.p;
    return PGV_OK;
}

// device rows [n x ld] -> caller rows [n x dim] (host or device)
int unstage_rows(pgv_ctx *ctx, const void *src_dev, int64_t n, int dim, pgv_dtype dtype,
                 const RowGeom &g, void *dst) {
    const size_t es = elem_size(dtype);
    if (n == 0) return PGV_OK;
    const bool dev = is_device_ptr(dst);
    PGV_HIP(hipMemcpy2DAsync(dst, (size_t)dim * es, src_dev, (size_t)g.ld * es, (size_t)dim * es,
                             (size_t)n, dev ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost,
                             ctx->stream));
    if (!dev) PGV_HIP(hipStreamSynchronize(ctx->stream));
    return PGV_OK;
}

// flat array host-or-device -> device
int stage_flat(pgv_ctx *ctx, const void *src, size_t bytes, DBuf &scratch, const void **out) {
    if (is_device_ptr(src)) {
        *out = src;
        return PGV_OK;
    }
    PGV_TRY(scratch.ensure(bytes ? bytes : 16));
    if (bytes) {
        PGV_HIP(hipMemcpyAsync(scratch.p, src, bytes, hipMemcpyHostToDevice, ctx->stream));
        PGV_HIP(hipStreamSynchronize(ctx->stream));
    }
    *out = scratch.p;
    return PGV_OK;
}

// An output the caller gave us: computed straight into it when it is device
// memory, otherwise into scratch and copied back by finish().
struct OutArg {
    void *user = nullptr;
    void *dev = nullptr;
    size_t bytes = 0;
    bool direct = false;
    int init(void *user_ptr, size_t nbytes, DBuf &scratch) {
        user = user_ptr;
        bytes = nbytes;
        if (!user_ptr) {
            dev = nullptr;
            return PGV_OK;
        }
        if (is_device_ptr(user_ptr)) {
            direct = true;
            dev = user_ptr;
            return PGV_OK;
        }
        PGV_TRY(scratch.ensure(nbytes ? nbytes : 16));
        dev = scratch.p;
        return PGV_OK;
    }
    template <typename T> T *as() const { return static_cast<T *>(dev); }
    // returns true via *need_sync when a device->host copy was enqueued
    int finish(pgv_ctx *ctx, bool *need_sync) const {
        if (user && !direct && bytes) {
            PGV_HIP(hipMemcpyAsync(user, dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
            *need_sync = true;
        }
        return PGV_OK;
    }
};

// h_a is pinned staging for small host-planned tables; the copy out of it is asynchronous, so
// it is only rewritten once that copy has been consumed
int staging_acquire(pgv_ctx *ctx) {
    if (ctx->h_a_pending) {
        PGV_HIP(hipEventSynchronize(ctx->h_a_busy));
        ctx->h_a_pending = false;
    }
    return PGV_OK;
}
int staging_release(pgv_ctx *ctx) {
    if (!ctx->h_a_busy) PGV_HIP(hipEventCreateWithFlags(&ctx->h_a_busy, hipEventDisableTiming));
    PGV_HIP(hipEventRecord(ctx->h_a_busy, ctx->stream));
    ctx->h_a_pending = true;
    return PGV_OK;
}

int sync_if(pgv_ctx *ctx, bool need) {
    if (need) PGV_HIP(hipStreamSynchronize(ctx->stream));
    return PGV_OK;
}

// ------------------------------------------------------------ profiling hooks
struct ScanTimer {
    pgv_ctx *ctx;
    size_t slot = (size_t)-1;
    int begin(double pairs, double rows, bool aux = false) {
        if (!ctx->profiling) return PGV_OK;
        if (ctx->ev_used + 2 > ctx->ev_pool.size()) {
            for (int i = 0; i < 2; i++) {
                hipEvent_t e;
                PGV_HIP(hipEventCreate(&e));
                ctx->ev_pool.push_back(e);
            }
        }
        slot = ctx->ev_used;
        ctx->ev_used += 2;
        if (ctx->ev_is_aux.size() < ctx->ev_used / 2) ctx->ev_is_aux.resize(ctx->ev_used / 2);
        ctx->ev_is_aux[slot / 2] = aux ? 1 : 0;
        if (aux) {
            ctx->aux_launches += 1;
            ctx->aux_pairs += pairs;
        } else {
            ctx->scan_launches += 1;
            ctx->scan_pairs += pairs;
            ctx->scan_rows += rows;
        }
        PGV_HIP(hipEventRecord(ctx->ev_pool[slot], ctx->stream));
        return PGV_OK;
    }
    int end() {
        if (slot == (size_t)-1) return PGV_OK;
        PGV_HIP(hipEventRecord(ctx->ev_pool[slot + 1], ctx->stream));
        return PGV_OK;
    }
};

int resolve_events(pgv_ctx *ctx) {
    if (ctx->ev_used == 0) return PGV_OK;
    PGV_HIP(hipStreamSynchronize(ctx->stream));
    for (size_t i = 0; i + 1 < ctx->ev_used; i += 2) {
        float ms = 0.f;
        PGV_HIP(hipEventElapsedTime(&ms, ctx->ev_pool[i], ctx->ev_pool[i + 1]));
        if (ctx->ev_is_aux[i / 2])
            ctx->aux_ms += ms;
        else
            ctx->scan_ms += ms;
    }
    ctx->ev_used = 0;
    return PGV_OK;
}

// --------------------------------------------------- dense scan (host-planned)
// rows [0, nrows) x queries [0, nq): out[q * out_stride + r].  Used for center
// ranking, exact scans and k-means++ rounds; tasks are planned on the host since
// their shape depends only on sizes.
int rows_per_task_for(pgv_ctx *ctx, int64_t total_rows, int64_t groups) {
    // aim at >= 8 tasks per CU, 32..256 rows each
    int64_t want_tasks = (int64_t)ctx->num_cus * 8;
    int64_t ch = (total_rows * groups + want_tasks - 1) / want_tasks;
    ch = (ch + 31) / 32 * 32;
    if (ch < 32) ch = 32;
    if (ch > 256) ch = 256;
    return (int)ch;
}

static bool dense_keep() {
    static int v = -1;
    if (v < 0) {
        const char *e = getenv("PGV_DENSE_KEEP");
        v = e ? atoi(e) : 1;
    }
    return v != 0;
}

int dense_scan(pgv_ctx *ctx, pgv_metric metric, pgv_dtype dtype, const RowGeom &g,
               const void *rows_dev, int64_t nrows, const void *queries_dev, int nq,
               int64_t out_stride, float *out_dev, bool mfma = false, const float *row_norms = nullptr,
               const float *query_norms = nullptr) {
    if (nrows <= 0 || nq <= 0) return PGV_OK;
    // one query against contiguous rows (a k-means++ round, pgv_distance_batch): no plan, no task counter -- the
    // single-query path's streaming kernel, whole rows in flight (k-means of the headline build: 0.137 -> 0.104 s)
    if (nq == 1 && !mfma && nrows <= 0x7fffffff) {
        ScanTimer timer{ctx};
        PGV_TRY(timer.begin((double)nrows, (double)nrows, true));
        PGV_TRY(launch_one_query_rows(ctx, metric, dtype, g, rows_dev, (int)nrows, queries_dev, out_dev));
        PGV_TRY(timer.end());
        return PGV_OK;
    }
    // many queries against the same rows (center ranking of a batch): the tile kernel serves
    // 16 queries per pass over the rows, the MFMA kernel 32 (L2: the expansion with the norms given,
    // an approximation the caller rechecks)
    const bool use_tile = !mfma && nq > 8 && tile_scan_supported(g);
    const int qt = mfma ? mfma_scan_queries_per_task()
                        : (use_tile ? tile_scan_queries_per_task() : scan_group_size(g, dtype, nq));
    const int ngroups = (nq + qt - 1) / qt;
    int ch = mfma ? mfma_scan_rows_per_task() : rows_per_task_for(ctx, nrows, ngroups);
    if (use_tile) {
        // whole tiles, and long enough runs to amortise a task's prologue (query registers,
        // first tile) when the rows are few but the query groups many
        const int tr = tile_scan_tile_rows(g);
        ch = (ch + tr - 1) / tr * tr;
        if (ch < 10 * tr && (int64_t)ngroups * ((nrows + 10 * tr - 1) / (10 * tr)) >= ctx->num_cus) ch = 10 * tr;
    }
    const int64_t nchunks = (nrows + ch - 1) / ch;
    const int64_t ntasks = nchunks * ngroups;
    if (ntasks > 0x7fffffff) PGV_FAIL(PGV_ERR_ARG, "scan: too many tasks");

    const size_t tb = sizeof(ScanTask) * (size_t)ntasks, pb = sizeof(ScanPair) * (size_t)nq;
    // a dense plan depends on (rows, queries, stride, rows per task, queries per task) only: a batch loop repeats it
    // (the center ranking of every batch; the 999 rounds of a k-means++ seeding, where re-planning cost a host-built
    // table, a copy-engine transfer and an event wait per round), so the last one stays on the device
    DBuf &plan_buf = ctx->dense_plan;
    const int plan_kind = (mfma ? 1 << 30 : 0) | (use_tile ? 1 << 29 : 0) | (qt << 12) | ch;
    const bool cached = ctx->dense_plan.p && ctx->dense_plan_rows == nrows && ctx->dense_plan_nq == nq &&
                        ctx->dense_plan_stride == out_stride && ctx->dense_plan_kind == plan_kind;
    if (!cached) {
        PGV_TRY(staging_acquire(ctx));
        PGV_TRY(ctx->h_a.ensure(tb + pb + 16));
        ScanTask *ht = ctx->h_a.as<ScanTask>();
        ScanPair *hp = reinterpret_cast<ScanPair *>(reinterpret_cast<char *>(ht) + tb);
        int *hn = reinterpret_cast<int *>(reinterpret_cast<char *>(hp) + pb);
        for (int q = 0; q < nq; q++) {
            hp[q].out_rel = (int64_t)q * out_stride;
            hp[q].query = q;
            hp[q].pad = 0;
        }
        int64_t t = 0;
        for (int64_t c = 0; c < nchunks; c++)
            for (int gidx = 0; gidx < ngroups; gidx++) {
                ht[t].row0 = c * ch;
                int64_t left = nrows - c * ch;
                ht[t].nrows = (int)(left < ch ? left : ch);
                ht[t].pair0 = gidx * qt;
                int pl = nq - gidx * qt;
                ht[t].npairs = pl < qt ? pl : qt;
                // a chunk that several query groups stream (consecutive tasks) is worth keeping in the caches
                ht[t].pad = (ngroups > 1 && dense_keep()) ? 1 : 0;
                t++;
            }
        *hn = (int)ntasks;
        ctx->dense_plan_rows = -1;  // (not valid while it is being replaced)
        PGV_TRY(plan_buf.ensure(tb + pb + 16));
        PGV_HIP(hipMemcpyAsync(plan_buf.p, ht, tb + pb + 16, hipMemcpyHostToDevice, ctx->stream));
        PGV_TRY(staging_release(ctx));
        ctx->dense_plan_rows = nrows;
        ctx->dense_plan_nq = nq;
        ctx->dense_plan_stride = out_stride;
        ctx->dense_plan_kind = plan_kind;
    }
    const ScanTask *dt = plan_buf.as<ScanTask>();
    const ScanPair *dp = reinterpret_cast<const ScanPair *>(plan_buf.as<char>() + tb);
    const int *dn = reinterpret_cast<const int *>(plan_buf.as<char>() + tb + pb);

    ScanTimer timer{ctx};
    PGV_TRY(timer.begin((double)nrows * nq, (double)nrows * ngroups, true));
    if (mfma)
        PGV_TRY(launch_mfma_scan(ctx, metric, dtype, g, rows_dev, queries_dev, dt, dn, (int)ntasks, dp, row_norms,
                                 query_norms, out_dev, rows_stream_past_caches(g, dtype, nrows)));
    else if (use_tile)
        PGV_TRY(launch_tile_scan(ctx, metric, dtype, g, rows_dev, queries_dev, dt, dn, (int)ntasks, dp, out_dev));
    else
        PGV_TRY(launch_scan(ctx, metric, dtype, g, rows_dev, queries_dev, dt, dn, (int)ntasks, dp, qt,
                            out_dev));
    PGV_TRY(timer.end());
    return PGV_OK;
}

// -------------------------------------------------- library-owned random source
struct Xoro {
    uint64_t s0, s1;
    static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    static uint64_t splitmix(uint64_t &st) {
        uint64_t v = (st += 0x9E3779B97f4A7C15ull);
        v = (v ^ (v >> 30)) * 0xBF58476D1CE4E5B9ull;
        v = (v ^ (v >> 27)) * 0x94D049BB133111EBull;
        return v ^ (v >> 31);
    }
    explicit Xoro(uint64_t seed) {
        s0 = splitmix(seed);
        s1 = splitmix(seed);
        if (!s0 && !s1) s0 = 1;
    }
    uint64_t next() {
        uint64_t a = s0, x = s1 ^ a, out = rotl(a * 5, 7) * 9;
        s0 = rotl(a, 24) ^ x ^ (x << 16);
        s1 = rotl(x, 37);
        return out;
    }
};

struct Rng {
    const pgv_rng *user;
    Xoro own;
    explicit Rng(const pgv_rng *r) : user(r), own(r ? r->seed : 0) {}
    double next_double() {
        if (user && user->next_double) return user->next_double(user->state);
        return std::ldexp((double)(own.next() >> 12), -52);
    }
    uint32_t next_u32() {
        if (user && user->next_u32) return user->next_u32(user->state);
        return (uint32_t)(own.next() >> 32);
    }
};

}  // namespace

// ---- shared by the IVFFlat scans and the exact scan: the candidates of the MFMA L2 paths
namespace {

struct ApproxScratch {
    float *cand_val = nullptr;   // [nq x kprime] approximate values, ascending
    int64_t *cand_pos = nullptr; // [nq x kprime] positions in the query's segment (center ids for the ranking)
    int32_t *flags = nullptr;    // [nq] flags | count | list of flagged queries
    int carve(pgv_ctx *ctx, DBuf &buf, int nq, int kprime) {
        (void)ctx;
        const size_t nk = (size_t)nq * kprime;
        const size_t a1 = (sizeof(float) * nk + 15) & ~(size_t)15, a2 = a1 + sizeof(int64_t) * nk,
                     a3 = a2 + sizeof(int32_t) * (2 * (size_t)nq + 1);
        PGV_TRY(buf.ensure(a3));
        char *b = buf.as<char>();
        cand_val = reinterpret_cast<float *>(b);
        cand_pos = reinterpret_cast<int64_t *>(b + a1);
        flags = reinterpret_cast<int32_t *>(b + a2);  // flags[nq], the count, is cleared by the candidates' top-k launch
        return PGV_OK;
    }
};

// k' of the MFMA L2 selections: the head asked for and a margin the rounding bound clears easily
// Device memory that another process may map (pgv_index_export / pgv_hnsw_export: hipIpcGetMemHandle, dmabuf mode).
// The runtime carves allocations below 2 MiB out of shared 2 MiB blocks; the handle of such a fragment is the BLOCK's,
// and on this pool's driver (round 6, two boxes, every run) hipIpcGetMemHandle of a small mirror returned "invalid
// argument" for good once other fragments of the worker had been exported and freed (profiles/r06/ipc_export_small.md;
// round 5 had seen it once in ~350 exports).  A mirror that can be exported therefore owns whole blocks: the size is
// rounded up to a multiple of 2 MiB (a 20 KB test index costs 2 MiB of a 288 GB part).
static const size_t kExportGranule = (size_t)2 << 20;
static inline hipError_t malloc_exportable(void **p, size_t bytes) {
    const size_t rounded = (bytes + kExportGranule - 1) / kExportGranule * kExportGranule;
    return hipMalloc(p, rounded ? rounded : kExportGranule);
}

static int approx_candidates(int k) {
    if (k <= 8) return 32;
    if (4 * k > 256) return k + 64;
    int kp = 64;
    while (kp < 4 * k) kp <<= 1;
    return kp;
}

static bool spherical(pgv_ops ops) { return ops == PGV_OPS_IP || ops == PGV_OPS_COSINE; }

static int check_ops(pgv_ops ops) {
    if (ops != PGV_OPS_L2 && ops != PGV_OPS_IP && ops != PGV_OPS_COSINE)
        PGV_FAIL(PGV_ERR_ARG, "unknown opclass family %d", (int)ops);
    return PGV_OK;
}

}  // namespace

// ---- area functions another area calls (defined without `static` in the unit named)
extern "C" {
// pgv_abi_ivf.hip
int rank_lists_dev(pgv_index *ix, const void *q_dev, int nq, int maxprobes, int32_t *out_lists_dev, float *out_dist_dev);
int scan_batch_dev(pgv_index *ix, const void *q_dev, int nq, const int32_t *probe_lists, int probes, int k, float *out_dist,
                   int64_t *out_slot, uint64_t *out_tid);
int check_batch_args(pgv_index *ix, const void *queries, int nq, int probes, int k, float *out_dist, uint64_t *out_tid,
                     const char *who);
// pgv_abi_build.hip
int lloyd_partial_dev(pgv_ctx *ctx, pgv_ops ops, pgv_dtype dtype, const RowGeom &g, const void *samples_dev, int n,
                      const void *centers_dev, int k, int32_t *closest_io, float *sums, int32_t *counts,
                      unsigned long long *changes);
int lloyd_finish_dev(pgv_ctx *ctx, pgv_ops ops, pgv_dtype dtype, const RowGeom &g, int dim, int k, const float *sums_dev,
                     const int32_t *counts_dev, const int32_t *counts_host, Rng &rng, void *centers_dev);
int check_centers_dev(pgv_ctx *ctx, pgv_ops ops, pgv_dtype dtype, const RowGeom &g, int dim, int k, const void *centers_dev);
}
