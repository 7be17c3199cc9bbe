// pgv_abi.hip -- extern "C" entry points of libpgv_hip (include/pgv_hip.h):
// argument checking, host<->device staging, work planning and kernel launches.
// There is no CPU fallback anywhere in this file: without a GPU every entry
// point reports PGV_ERR_DEVICE.
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

// ===================================================================== utils
namespace pgv {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int DBuf::ensure(size_t bytes) {
    if (bytes <= cap && p) return PGV_OK;
    if (p) {
        (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    size_t want = bytes < 256 ? 256 : bytes;
    want += want / 4;  // headroom so slowly growing requests do not reallocate every call
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) {
        p = nullptr;
        (void)hipGetLastError();
        set_error("hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
        return PGV_ERR_NOMEM;
    }
    cap = want;
    return PGV_OK;
}
void DBuf::release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
}
int HBuf::ensure(size_t bytes) {
    if (bytes <= cap && p) return PGV_OK;
    if (p) {
        (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
    }
    size_t want = bytes < 256 ? 256 : bytes;
    want += want / 4;
    hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
    if (e != hipSuccess) {
        p = nullptr;
        (void)hipGetLastError();
        set_error("hipHostMalloc(%zu) failed: %s", want, hipGetErrorString(e));
        return PGV_ERR_NOMEM;
    }
    cap = want;
    return PGV_OK;
}
void HBuf::release() {
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
}

bool is_device_ptr(const void *p) {
    if (!p) return false;
    hipPointerAttribute_t a;
    hipError_t e = hipPointerGetAttributes(&a, p);
    if (e != hipSuccess) {
        (void)hipGetLastError();  // plain malloc'ed memory: not known to the runtime
        return false;
    }
    return a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged;
}

}  // namespace pgv

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

// =================================================================== context
extern "C" {

const char *pgv_last_error(void) { return pgv::g_err; }
int pgv_abi_version(void) { return PGV_ABI_VERSION; }

int pgv_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

int pgv_device_memory(int device, uint64_t *free_bytes, uint64_t *total_bytes) {
    int n = pgv_device_count();
    if (n <= 0) PGV_FAIL(PGV_ERR_DEVICE, "no HIP device available (libpgv_hip has no CPU path)");
    if (device < 0 || device >= n) PGV_FAIL(PGV_ERR_ARG, "device %d out of range 0..%d", device, n - 1);
    PGV_HIP(hipSetDevice(device));
    size_t f = 0, t = 0;
    PGV_HIP(hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = t;
    return PGV_OK;
}

int pgv_pinned_alloc(size_t bytes, void **out) {
    if (!out) PGV_FAIL(PGV_ERR_ARG, "pgv_pinned_alloc: out is NULL");
    *out = nullptr;
    if (pgv_device_count() <= 0) PGV_FAIL(PGV_ERR_DEVICE, "no HIP device available (libpgv_hip has no CPU path)");
    if (hipHostMalloc(out, bytes ? bytes : 16, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        PGV_FAIL(PGV_ERR_NOMEM, "hipHostMalloc(%zu) failed", bytes);
    }
    return PGV_OK;
}

void pgv_pinned_free(void *p) {
    if (p) (void)hipHostFree(p);
}

int pgv_pinned_register(void *p, size_t bytes) {
    if (!p || bytes == 0) PGV_FAIL(PGV_ERR_ARG, "pgv_pinned_register: empty range");
    if (pgv_device_count() <= 0) PGV_FAIL(PGV_ERR_DEVICE, "no HIP device available (libpgv_hip has no CPU path)");
    hipError_t e = hipHostRegister(p, bytes, hipHostRegisterPortable);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        PGV_FAIL(PGV_ERR_NOMEM, "hipHostRegister(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
    }
    return PGV_OK;
}

void pgv_pinned_unregister(void *p) {
    if (p) (void)hipHostUnregister(p);
}

int pgv_ctx_create(int device, void *stream, pgv_ctx **out) {
    if (!out) PGV_FAIL(PGV_ERR_ARG, "pgv_ctx_create: out is NULL");
    *out = nullptr;
    int n = pgv_device_count();
    if (n <= 0) PGV_FAIL(PGV_ERR_DEVICE, "no HIP device available (libpgv_hip has no CPU path)");
    if (device < 0 || device >= n) PGV_FAIL(PGV_ERR_ARG, "device %d out of range 0..%d", device, n - 1);
    PGV_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    PGV_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        PGV_FAIL(PGV_ERR_DEVICE, "device %d is %s; libpgv_hip is built for gfx950 only", device,
                 prop.gcnArchName);
    pgv_ctx *ctx = new (std::nothrow) pgv_ctx();
    if (!ctx) PGV_FAIL(PGV_ERR_NOMEM, "out of host memory");
    ctx->device = device;
    {
        const char *e = getenv("PGV_NO_WIDEN");
        ctx->no_widen = e && atoi(e) != 0;
    }
    ctx->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (stream == PGV_DEFAULT_STREAM) {
        ctx->stream = nullptr;  // the legacy default stream
    } else if (stream) {
        ctx->stream = static_cast<hipStream_t>(stream);
    } else {
        hipError_t e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
        if (e != hipSuccess) {
            delete ctx;
            PGV_FAIL(PGV_ERR_DEVICE, "hipStreamCreate failed: %s", hipGetErrorString(e));
        }
        ctx->own_stream = true;
    }
    if (hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess) {
        pgv_ctx_destroy(ctx);
        PGV_FAIL(PGV_ERR_DEVICE, "hipEventCreate failed");
    }
    *out = ctx;
    return PGV_OK;
}

void pgv_ctx_destroy(pgv_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    DBuf *d[] = {&ctx->q_stage, &ctx->rows_stage, &ctx->centers_stage, &ctx->out_stage,
                 &ctx->out_stage2, &ctx->idx_stage, &ctx->tasks, &ctx->pairs, &ctx->counters,
                 &ctx->plan_a, &ctx->plan_b, &ctx->plan_c, &ctx->plan_d, &ctx->dist_mat,
                 &ctx->sel_a, &ctx->sel_b, &ctx->km_a, &ctx->km_b, &ctx->km_c, &ctx->km_d,
                 &ctx->km_e, &ctx->km_f, &ctx->km_g, &ctx->stats_dev, &ctx->mf_a, &ctx->mf_b, &ctx->mf_c,
                 &ctx->zeros, &ctx->ms_a, &ctx->ms_b, &ctx->dense_plan, &ctx->xt_norms, &ctx->mf_d};
    for (DBuf *b : d) b->release();
    ctx->h_a.release();
    ctx->h_b.release();
    ctx->h_c.release();
    for (hipEvent_t e : ctx->ev_pool) (void)hipEventDestroy(e);
    if (ctx->h_a_busy) (void)hipEventDestroy(ctx->h_a_busy);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int pgv_ctx_sync(pgv_ctx *ctx) {
    if (!ctx) PGV_FAIL(PGV_ERR_ARG, "ctx is NULL");
    PGV_HIP(hipStreamSynchronize(ctx->stream));
    for (pgv_ctx *c : ctx->children) PGV_HIP(hipStreamSynchronize(c->stream));  // lanes of overlapping batches
    return PGV_OK;
}

void *pgv_ctx_stream(pgv_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

int pgv_timer_start(pgv_ctx *ctx) {
    if (!ctx) PGV_FAIL(PGV_ERR_ARG, "ctx is NULL");
    PGV_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    return PGV_OK;
}

int pgv_timer_stop(pgv_ctx *ctx, float *out_ms) {
    if (!ctx || !out_ms) PGV_FAIL(PGV_ERR_ARG, "ctx/out_ms is NULL");
    PGV_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    PGV_HIP(hipEventSynchronize(ctx->ev1));
    PGV_HIP(hipEventElapsedTime(out_ms, ctx->ev0, ctx->ev1));
    return PGV_OK;
}

int pgv_ctx_set_profiling(pgv_ctx *ctx, int on) {
    if (!ctx) PGV_FAIL(PGV_ERR_ARG, "ctx is NULL");
    PGV_TRY(resolve_events(ctx));
    ctx->profiling = on != 0;
    if (ctx->profiling && !ctx->stats_dev.p) {
        PGV_TRY(ctx->stats_dev.ensure(8 * sizeof(double)));
        PGV_HIP(hipMemsetAsync(ctx->stats_dev.p, 0, 8 * sizeof(double), ctx->stream));
    }
    for (pgv_ctx *c : ctx->children) PGV_TRY(pgv_ctx_set_profiling(c, on));
    return PGV_OK;
}

int pgv_ctx_set_exact_scan(pgv_ctx *ctx, int on) {
    if (!ctx) PGV_FAIL(PGV_ERR_ARG, "pgv_ctx_set_exact_scan: ctx is NULL");
    ctx->no_mfma_scan = on != 0;
    for (pgv_ctx *c : ctx->children) c->no_mfma_scan = ctx->no_mfma_scan;
    return PGV_OK;
}

int pgv_ctx_set_bound(pgv_ctx *ctx, int mode) {
    if (!ctx) PGV_FAIL(PGV_ERR_ARG, "pgv_ctx_set_bound: ctx is NULL");
    if (mode != PGV_BOUND_STATISTICAL && mode != PGV_BOUND_WORST_CASE)
        PGV_FAIL(PGV_ERR_ARG, "pgv_ctx_set_bound: unknown mode %d", mode);
    ctx->bound_mode = mode;
    ctx->assign_bound_mode = mode;
    for (pgv_ctx *c : ctx->children) c->bound_mode = c->assign_bound_mode = mode;
    return PGV_OK;
}

int pgv_ctx_reset_stats(pgv_ctx *ctx) {
    if (!ctx) PGV_FAIL(PGV_ERR_ARG, "ctx is NULL");
    PGV_TRY(resolve_events(ctx));
    ctx->scan_ms = 0;
    ctx->scan_launches = 0;
    ctx->scan_pairs = 0;
    ctx->scan_rows = 0;
    if (ctx->stats_dev.p) PGV_HIP(hipMemsetAsync(ctx->stats_dev.p, 0, 8 * sizeof(double), ctx->stream));
    ctx->aux_ms = 0;
    ctx->aux_launches = 0;
    ctx->aux_pairs = 0;
    for (pgv_ctx *c : ctx->children) PGV_TRY(pgv_ctx_reset_stats(c));
    return PGV_OK;
}

int pgv_ctx_get_stats(pgv_ctx *ctx, pgv_stats *out) {
    if (!ctx || !out) PGV_FAIL(PGV_ERR_ARG, "ctx/out is NULL");
    PGV_TRY(resolve_events(ctx));
    double dev_acc[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    if (ctx->stats_dev.p) {
        PGV_HIP(hipMemcpyAsync(dev_acc, ctx->stats_dev.p, sizeof(dev_acc), hipMemcpyDeviceToHost, ctx->stream));
        PGV_HIP(hipStreamSynchronize(ctx->stream));
    }
    out->scan_ms = ctx->scan_ms;
    out->scan_launches = ctx->scan_launches;
    out->scan_pairs = ctx->scan_pairs + dev_acc[0];
    out->scan_rows = ctx->scan_rows + dev_acc[1];
    out->aux_ms = ctx->aux_ms;
    out->aux_launches = ctx->aux_launches;
    out->aux_pairs = ctx->aux_pairs;
    out->assign_redo_rows = dev_acc[2];
    out->assign_rows = dev_acc[3];
    out->assign_recheck_rows = dev_acc[4];
    out->scan_unique_rows = dev_acc[5];
    out->scan_redo_queries = dev_acc[6];
    out->scan_widened_queries = dev_acc[7];
    for (pgv_ctx *c : ctx->children) {  // what the lanes of overlapping batches did counts as this context's
        pgv_stats cs;
        PGV_TRY(pgv_ctx_get_stats(c, &cs));
        out->scan_ms += cs.scan_ms;
        out->scan_launches += cs.scan_launches;
        out->scan_pairs += cs.scan_pairs;
        out->scan_rows += cs.scan_rows;
        out->aux_ms += cs.aux_ms;
        out->aux_launches += cs.aux_launches;
        out->aux_pairs += cs.aux_pairs;
        out->assign_redo_rows += cs.assign_redo_rows;
        out->assign_rows += cs.assign_rows;
        out->assign_recheck_rows += cs.assign_recheck_rows;
        out->scan_unique_rows += cs.scan_unique_rows;
        out->scan_redo_queries += cs.scan_redo_queries;
        out->scan_widened_queries += cs.scan_widened_queries;
    }
    return PGV_OK;
}

// ============================================================== IVFFlat index

namespace {
// where each device array of an IVFFlat mirror sits inside its one allocation
struct IndexLayout {
    size_t centers, vectors, offsets, tids, row_norms, center_norms, bytes;
    bool has_tids, has_norms;
};
IndexLayout index_layout(int nlists, int64_t n, size_t row_bytes, bool has_tids, bool l2) {
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    IndexLayout L{};
    size_t at = 0;
    L.centers = at; at = up(at + (size_t)nlists * row_bytes);
    L.vectors = at; at = up(at + (size_t)(n > 0 ? n : 1) * row_bytes);
    L.offsets = at; at = up(at + sizeof(int64_t) * ((size_t)nlists + 1));
    L.has_tids = has_tids;
    L.tids = at; if (has_tids) at = up(at + sizeof(uint64_t) * (size_t)n);
    L.has_norms = l2;
    L.row_norms = at; if (l2 && n > 0) at = up(at + sizeof(float) * ((size_t)n + 1));
    L.center_norms = at; if (l2) at = up(at + sizeof(float) * ((size_t)nlists + 1));
    L.bytes = at;
    return L;
}
void index_carve(pgv_index *ix, const IndexLayout &L) {
    char *b = static_cast<char *>(ix->arena);
    ix->centers = b + L.centers;
    ix->vectors = b + L.vectors;
    ix->list_offsets = reinterpret_cast<int64_t *>(b + L.offsets);
    ix->tids = L.has_tids ? reinterpret_cast<uint64_t *>(b + L.tids) : nullptr;
    ix->row_norms = L.has_norms && ix->nrows > 0 ? reinterpret_cast<float *>(b + L.row_norms) : nullptr;
    ix->center_norms = L.has_norms ? reinterpret_cast<float *>(b + L.center_norms) : nullptr;
}
// len_prefix / max_list_len from h_offsets
void index_host_tables(pgv_index *ix) {
    const int nlists = ix->nlists;
    std::vector<int64_t> lens((size_t)nlists);
    int64_t maxlen = 0;
    for (int l = 0; l < nlists; l++) {
        lens[l] = ix->h_offsets[l + 1] - ix->h_offsets[l];
        if (lens[l] > maxlen) maxlen = lens[l];
    }
    ix->max_list_len = maxlen;
    std::sort(lens.begin(), lens.end(), [](int64_t a, int64_t b) { return a > b; });
    ix->len_prefix.assign((size_t)nlists + 1, 0);
    for (int l = 0; l < nlists; l++) ix->len_prefix[l + 1] = ix->len_prefix[l] + lens[l];
}
}  // namespace

extern "C++" {
namespace {

// The mirror of an index whose list offsets are known: one allocation, host tables, the norms the MFMA paths want.
// `fill` enqueues (on ctx->stream) whatever brings centers / vectors / tids into the carved arrays.
template <typename Fill>
int index_create(pgv_ctx *ctx, pgv_metric metric, pgv_dtype dtype, int dim, int nlists, const std::vector<int64_t> &off,
                 bool has_tids, Fill fill, pgv_index **out) {
    const int64_t n = off[nlists];
    pgv_index *ix = new (std::nothrow) pgv_index();
    if (!ix) PGV_FAIL(PGV_ERR_NOMEM, "out of host memory");
    ix->ctx = ctx;
    ix->refs = new (std::nothrow) int(1);
    if (!ix->refs) {
        delete ix;
        PGV_FAIL(PGV_ERR_NOMEM, "out of host memory");
    }
    ix->metric = metric;
    ix->dtype = dtype;
    ix->dim = dim;
    ix->nlists = nlists;
    ix->nrows = n;
    ix->geom = row_geom(dim, dtype);
    ix->h_offsets = off;
    index_host_tables(ix);
    const size_t row_bytes = (size_t)ix->geom.ld * elem_size(dtype);

    auto fail = [&](int rc) {
        pgv_index_free(ix);
        return rc;
    };
    // one allocation for the whole mirror (a single IPC handle exports it): centers | vectors | list_offsets |
    // tids | row_norms | center_norms, each part 256-byte aligned
    IndexLayout lay = index_layout(nlists, n, row_bytes, has_tids && n > 0, metric == PGV_L2SQ);
    if (hipMalloc(&ix->arena, lay.bytes) != hipSuccess) {
        (void)hipGetLastError();
        set_error("hipMalloc(%zu) for the index mirror failed", lay.bytes);
        return fail(PGV_ERR_NOMEM);
    }
    ix->arena_bytes = lay.bytes;
    index_carve(ix, lay);
    int rc;
    if ((rc = fill(ix)) != PGV_OK) return fail(rc);
    if (hipMemcpyAsync(ix->list_offsets, ix->h_offsets.data(), sizeof(int64_t) * off.size(),
                       hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
        return fail((set_error("copy of list_offsets failed"), PGV_ERR_DEVICE));
    if (ix->row_norms) {
        // |x|^2 per row and the largest of them: the MFMA scan's expansion of the L2 distance
        if (hipMemsetAsync(ix->row_norms + n, 0, sizeof(float), ctx->stream) != hipSuccess)
            return fail((set_error("memset of row_norms failed"), PGV_ERR_DEVICE));
        if ((rc = launch_row_norms(ctx, dtype, ix->geom, ix->vectors, n, ix->row_norms,
                                   reinterpret_cast<unsigned *>(ix->row_norms + n))) != PGV_OK)
            return fail(rc);
    }
    if (ix->center_norms) {
        if (hipMemsetAsync(ix->center_norms + nlists, 0, sizeof(float), ctx->stream) != hipSuccess)
            return fail((set_error("memset of center_norms failed"), PGV_ERR_DEVICE));
        if ((rc = launch_row_norms(ctx, dtype, ix->geom, ix->centers, nlists, ix->center_norms,
                                   reinterpret_cast<unsigned *>(ix->center_norms + nlists))) != PGV_OK)
            return fail(rc);
    }
    if (hipStreamSynchronize(ctx->stream) != hipSuccess)
        return fail((set_error("index upload failed: %s", hipGetErrorString(hipGetLastError())), PGV_ERR_DEVICE));
    *out = ix;
    return PGV_OK;
}

// tightly packed rows (host or device) into padded device rows
int put_rows_on(hipStream_t stream, const RowGeom &g, pgv_dtype dtype, int dim, void *dst, const void *src, int64_t rows) {
    if (rows == 0) return PGV_OK;
    const size_t es = elem_size(dtype), row_bytes = (size_t)g.ld * es;
    const bool dev = is_device_ptr(src);
    if (g.ld == dim) {
        PGV_HIP(hipMemcpyAsync(dst, src, (size_t)rows * row_bytes, dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice,
                               stream));
    } else {
        PGV_HIP(hipMemsetAsync(dst, 0, (size_t)rows * row_bytes, stream));
        PGV_HIP(hipMemcpy2DAsync(dst, row_bytes, src, (size_t)dim * es, (size_t)dim * es, (size_t)rows,
                                 dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, stream));
    }
    return PGV_OK;
}

int put_rows(pgv_ctx *ctx, const RowGeom &g, pgv_dtype dtype, int dim, void *dst, const void *src, int64_t rows) {
    return put_rows_on(ctx->stream, g, dtype, dim, dst, src, rows);
}

}  // namespace
}  // extern "C++"

int pgv_index_upload(pgv_ctx *ctx, pgv_metric metric, pgv_dtype dtype, int dim, int nlists,
                     const void *centers, const int64_t *list_offsets, const void *vectors,
                     const uint64_t *tids, pgv_index **out) {
    if (!ctx || !out) PGV_FAIL(PGV_ERR_ARG, "pgv_index_upload: ctx/out is NULL");
    *out = nullptr;
    PGV_TRY(check_common(dtype, dim));
    PGV_TRY(check_metric(metric));
    // IVFFLAT_MAX_LISTS (src/ivfflat.h:56)
    if (nlists < 1 || nlists > 32768) PGV_FAIL(PGV_ERR_ARG, "lists %d outside 1..32768", nlists);
    if (!centers || !list_offsets) PGV_FAIL(PGV_ERR_ARG, "centers/list_offsets is NULL");
    PGV_HIP(hipSetDevice(ctx->device));

    std::vector<int64_t> off((size_t)nlists + 1);
    if (is_device_ptr(list_offsets)) {
        PGV_HIP(hipMemcpy(off.data(), list_offsets, sizeof(int64_t) * off.size(), hipMemcpyDeviceToHost));
    } else {
        memcpy(off.data(), list_offsets, sizeof(int64_t) * off.size());
    }
    if (off[0] != 0) PGV_FAIL(PGV_ERR_ARG, "list_offsets[0] must be 0");
    for (int l = 0; l < nlists; l++)
        if (off[l + 1] < off[l]) PGV_FAIL(PGV_ERR_ARG, "list_offsets not ascending at list %d", l);
    const int64_t n = off[nlists];
    if (n > 0 && !vectors) PGV_FAIL(PGV_ERR_ARG, "vectors is NULL");
    return index_create(ctx, metric, dtype, dim, nlists, off, tids != nullptr, [&](pgv_index *ix) -> int {
        PGV_TRY(put_rows(ctx, ix->geom, dtype, dim, ix->centers, centers, nlists));
        PGV_TRY(put_rows(ctx, ix->geom, dtype, dim, ix->vectors, vectors, n));
        if (ix->tids)
            PGV_HIP(hipMemcpyAsync(ix->tids, tids, sizeof(uint64_t) * (size_t)n,
                                   is_device_ptr(tids) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ctx->stream));
        return PGV_OK;
    }, out);
}

// ------------------------------------------------------------ the build's tuplesort on the device
struct pgv_builder {
    pgv_ctx *ctx = nullptr;
    pgv_metric metric = PGV_L2SQ;
    pgv_dtype dtype = PGV_F32;
    int dim = 0, nlists = 0;
    RowGeom geom{};
    DBuf centers;  // [nlists x ld]
    DBuf rows;     // [cap x ld] heap order
    DBuf tids;     // [cap]
    DBuf lists;    // [cap] int32
    int64_t n = 0, cap = 0;
    int64_t assigned = 0;  // rows [0, assigned) have their list id
    bool has_tids = true;
    // centers not known yet (pgv_builder_begin with centers == NULL): rows are only copied, on a stream of the
    // builder's own, so that the k-means which is still computing the centers on the context's stream (from another
    // host thread) and the upload of the heap overlap; pgv_builder_set_centers ends this state
    bool deferred = false;
    hipStream_t copy_stream = nullptr;
    hipStream_t stream() const { return deferred ? copy_stream : ctx->stream; }
};

static int builder_reserve(pgv_builder *b, int64_t want) {
    if (want <= b->cap) return PGV_OK;
    int64_t cap = b->cap ? b->cap + b->cap / 2 : want;
    if (cap < want) cap = want;
    const size_t row_bytes = (size_t)b->geom.ld * elem_size(b->dtype);
    DBuf rows, tids, lists;
    PGV_TRY(rows.ensure(row_bytes * (size_t)cap));
    int rc = tids.ensure(sizeof(uint64_t) * (size_t)cap);
    if (rc == PGV_OK) rc = lists.ensure(sizeof(int32_t) * (size_t)cap);
    if (rc == PGV_OK && b->n > 0) {
        hipError_t e = hipMemcpyAsync(rows.p, b->rows.p, row_bytes * (size_t)b->n, hipMemcpyDeviceToDevice, b->stream());
        if (e == hipSuccess) e = hipMemcpyAsync(tids.p, b->tids.p, sizeof(uint64_t) * (size_t)b->n, hipMemcpyDeviceToDevice, b->stream());
        if (e == hipSuccess) e = hipMemcpyAsync(lists.p, b->lists.p, sizeof(int32_t) * (size_t)b->n, hipMemcpyDeviceToDevice, b->stream());
        if (e == hipSuccess) e = hipStreamSynchronize(b->stream());
        if (e != hipSuccess) {
            set_error("growing the builder failed: %s", hipGetErrorString(e));
            rc = PGV_ERR_DEVICE;
        }
    }
    if (rc != PGV_OK) {
        rows.release();
        tids.release();
        lists.release();
        return rc;
    }
    b->rows.release();
    b->tids.release();
    b->lists.release();
    b->rows = rows;
    b->tids = tids;
    b->lists = lists;
    b->cap = cap;
    return PGV_OK;
}

// rows [assigned, n) to their nearest center
static int builder_assign_pending(pgv_builder *b) {
    if (b->assigned >= b->n) return PGV_OK;
    const size_t row_bytes = (size_t)b->geom.ld * elem_size(b->dtype);
    PGV_TRY(launch_argmin(b->ctx, b->metric, b->dtype, b->geom, b->rows.as<char>() + (size_t)b->assigned * row_bytes,
                          b->n - b->assigned, b->centers.p, b->nlists, b->lists.as<int32_t>() + b->assigned, nullptr));
    b->assigned = b->n;
    return PGV_OK;
}

int pgv_builder_begin(pgv_ctx *ctx, pgv_metric metric, pgv_dtype dtype, int dim, int nlists, const void *centers,
                      int64_t expected_rows, pgv_builder **out) {
    if (!ctx || !out) PGV_FAIL(PGV_ERR_ARG, "pgv_builder_begin: ctx/out is NULL");
    *out = nullptr;
    PGV_TRY(check_common(dtype, dim));
    PGV_TRY(check_metric(metric));
    if (nlists < 1 || nlists > 32768) PGV_FAIL(PGV_ERR_ARG, "lists %d outside 1..32768", nlists);
    if (expected_rows < 0) PGV_FAIL(PGV_ERR_ARG, "expected_rows < 0");
    PGV_HIP(hipSetDevice(ctx->device));
    pgv_builder *b = new (std::nothrow) pgv_builder();
    if (!b) PGV_FAIL(PGV_ERR_NOMEM, "out of host memory");
    b->ctx = ctx;
    b->metric = metric;
    b->dtype = dtype;
    b->dim = dim;
    b->nlists = nlists;
    b->geom = row_geom(dim, dtype);
    int rc = b->centers.ensure((size_t)b->geom.ld * elem_size(dtype) * (size_t)nlists);
    if (rc == PGV_OK && centers) {
        rc = put_rows(ctx, b->geom, dtype, dim, b->centers.p, centers, nlists);
        if (rc == PGV_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = PGV_ERR_DEVICE;  // the caller may reuse centers
    } else if (rc == PGV_OK) {
        b->deferred = true;
        if (hipStreamCreateWithFlags(&b->copy_stream, hipStreamNonBlocking) != hipSuccess) {
            set_error("pgv_builder_begin: no stream for the upload");
            rc = PGV_ERR_DEVICE;
        }
    }
    if (rc == PGV_OK && expected_rows > 0) rc = builder_reserve(b, expected_rows);
    if (rc != PGV_OK) {
        pgv_builder_free(b);
        return rc;
    }
    *out = b;
    return PGV_OK;
}

void pgv_builder_free(pgv_builder *b) {
    if (!b) return;
    if (b->copy_stream) {
        (void)hipStreamSynchronize(b->copy_stream);
        (void)hipStreamDestroy(b->copy_stream);
    }
    if (b->ctx) (void)hipStreamSynchronize(b->ctx->stream);
    b->centers.release();
    b->rows.release();
    b->tids.release();
    b->lists.release();
    delete b;
}

int64_t pgv_builder_rows(const pgv_builder *b) { return b ? b->n : -1; }

int pgv_builder_add(pgv_builder *b, const void *rows, const uint64_t *tids, int64_t n) {
    if (!b) PGV_FAIL(PGV_ERR_ARG, "pgv_builder_add: builder is NULL");
    if (n < 0) PGV_FAIL(PGV_ERR_ARG, "n < 0");
    if (n == 0) return PGV_OK;
    if (!rows) PGV_FAIL(PGV_ERR_ARG, "rows is NULL");
    if (b->n + n > 0xffffffffll) PGV_FAIL(PGV_ERR_ARG, "pgv_builder_add: more than 2^32 rows");
    if (b->n > 0 && (tids != nullptr) != b->has_tids) PGV_FAIL(PGV_ERR_ARG, "pgv_builder_add: tids given for some batches only");
    pgv_ctx *ctx = b->ctx;
    PGV_HIP(hipSetDevice(ctx->device));
    PGV_TRY(builder_reserve(b, b->n + n));
    b->has_tids = tids != nullptr;
    const size_t row_bytes = (size_t)b->geom.ld * elem_size(b->dtype);
    char *dst = b->rows.as<char>() + (size_t)b->n * row_bytes;
    hipStream_t stream = b->stream();
    PGV_TRY(put_rows_on(stream, b->geom, b->dtype, b->dim, dst, rows, n));
    if (tids)
        PGV_HIP(hipMemcpyAsync(b->tids.as<uint64_t>() + b->n, tids, sizeof(uint64_t) * (size_t)n,
                               is_device_ptr(tids) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, stream));
    b->n += n;
    // AddTupleToSort's argmin (src/ivfbuild.c:183-192) for this batch (and what an earlier centerless phase left),
    // where the rows now are
    if (!b->deferred) PGV_TRY(builder_assign_pending(b));
    // host buffers may be reused by the caller right away; device rows must have arrived before the caller's stream
    // moves on
    if (b->deferred || !is_device_ptr(rows) || (tids && !is_device_ptr(tids))) PGV_HIP(hipStreamSynchronize(stream));
    return PGV_OK;
}

int pgv_builder_set_centers(pgv_builder *b, const void *centers) {
    if (!b || !centers) PGV_FAIL(PGV_ERR_ARG, "pgv_builder_set_centers: builder/centers is NULL");
    if (!b->deferred) PGV_FAIL(PGV_ERR_STATE, "pgv_builder_set_centers: the builder has its centers");
    pgv_ctx *ctx = b->ctx;
    PGV_HIP(hipSetDevice(ctx->device));
    PGV_HIP(hipStreamSynchronize(b->copy_stream));  // every row has arrived; from here on the context's stream is used
    b->deferred = false;
    PGV_TRY(put_rows(ctx, b->geom, b->dtype, b->dim, b->centers.p, centers, b->nlists));
    PGV_HIP(hipStreamSynchronize(ctx->stream));
    return PGV_OK;
}

int pgv_builder_finish(pgv_builder *b, pgv_index **out_index, int64_t *out_offsets, int32_t *out_lists) {
    if (!b || !out_index) PGV_FAIL(PGV_ERR_ARG, "pgv_builder_finish: builder/out_index is NULL");
    *out_index = nullptr;
    pgv_ctx *ctx = b->ctx;
    PGV_HIP(hipSetDevice(ctx->device));
    if (b->deferred) PGV_FAIL(PGV_ERR_STATE, "pgv_builder_finish: no centers (pgv_builder_set_centers)");
    PGV_TRY(builder_assign_pending(b));
    const int64_t n = b->n;
    const int nlists = b->nlists;
    int list_bits = 1;
    while ((1 << list_bits) < nlists) list_bits++;
    const size_t sort_bytes = n > 0 ? build_sort_scratch_bytes(n, 32 + list_bits) : 0;
    // scratch: keys_tmp | keys_sorted | counts | offsets | bad | sort scratch
    const size_t kb = sizeof(unsigned long long) * (size_t)(n > 0 ? n : 1), cb = sizeof(unsigned long long) * (size_t)nlists,
                 ob = sizeof(int64_t) * ((size_t)nlists + 1);
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t o_sorted = up(kb), o_counts = o_sorted + up(kb), o_off = o_counts + up(cb), o_bad = o_off + up(ob),
                 o_sort = o_bad + 256;
    DBuf scratch;
    PGV_TRY(scratch.ensure(o_sort + sort_bytes + 256));
    char *sp = scratch.as<char>();
    auto *keys_tmp = reinterpret_cast<unsigned long long *>(sp);
    auto *keys_sorted = reinterpret_cast<unsigned long long *>(sp + o_sorted);
    auto *counts = reinterpret_cast<unsigned long long *>(sp + o_counts);
    auto *offsets_dev = reinterpret_cast<int64_t *>(sp + o_off);
    int *bad = reinterpret_cast<int *>(sp + o_bad);
    int rc = launch_build_order(ctx, b->lists.as<int32_t>(), n, nlists, keys_tmp, keys_sorted, counts, offsets_dev, bad,
                                sp + o_sort, sort_bytes);
    std::vector<int64_t> off((size_t)nlists + 1);
    int bad_h = 0;
    if (rc == PGV_OK) {
        hipError_t e = hipMemcpyAsync(off.data(), offsets_dev, ob, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(&bad_h, bad, sizeof(int), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess && out_lists && n > 0)
            e = hipMemcpyAsync(out_lists, b->lists.p, sizeof(int32_t) * (size_t)n,
                               is_device_ptr(out_lists) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) {
            set_error("pgv_builder_finish: %s", hipGetErrorString(e));
            rc = PGV_ERR_DEVICE;
        }
    }
    if (rc == PGV_OK && (bad_h || off[nlists] != n)) {
        set_error("pgv_builder_finish: assignment produced a list id outside 0..%d", nlists - 1);
        rc = PGV_ERR_STATE;
    }
    if (rc == PGV_OK)
        rc = index_create(ctx, b->metric, b->dtype, b->dim, nlists, off, true, [&](pgv_index *ix) -> int {
            const size_t row_bytes = (size_t)b->geom.ld * elem_size(b->dtype);
            PGV_HIP(hipMemcpyAsync(ix->centers, b->centers.p, row_bytes * (size_t)nlists, hipMemcpyDeviceToDevice, ctx->stream));
            // rows and heap TIDs (heap positions when none were given) into list-major order, heap order inside a list
            return launch_build_gather(ctx, b->rows.p, keys_sorted, n, b->geom.nvec, ix->vectors,
                                       b->has_tids ? b->tids.as<uint64_t>() : nullptr, ix->tids);
        }, out_index);
    scratch.release();
    if (rc != PGV_OK) return rc;
    if (out_offsets) memcpy(out_offsets, off.data(), ob);
    // the heap-order copy has served
    b->rows.release();
    b->tids.release();
    b->lists.release();
    b->n = b->cap = b->assigned = 0;
    return PGV_OK;
}

// the mirror's rows, list-major, back to the host in pieces: double-buffered D2H into pinned memory, the sink called
// for piece i while piece i + 1 is on its way
int pgv_index_drain(pgv_index *ix, int64_t chunk_rows, pgv_rows_sink sink, void *arg) {
    if (!ix || !sink) PGV_FAIL(PGV_ERR_ARG, "pgv_index_drain: index/sink is NULL");
    pgv_ctx *ctx = ix->ctx;
    PGV_HIP(hipSetDevice(ctx->device));
    const int64_t n = ix->nrows;
    if (n == 0) return PGV_OK;
    const size_t es = elem_size(ix->dtype), tight = (size_t)ix->dim * es, padded = (size_t)ix->geom.ld * es;
    // 64 MB pieces: long enough for the link's full rate, short enough that pinning the two bounce buffers (which
    // costs ~30 ms at 2 x 256 MB) does not show
    if (chunk_rows <= 0) chunk_rows = (int64_t)std::max<size_t>(1, ((size_t)64 << 20) / tight);
    if (chunk_rows > n) chunk_rows = n;
    void *buf[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
    const size_t piece = tight * (size_t)chunk_rows + sizeof(uint64_t) * (size_t)chunk_rows;
    int rc = PGV_OK;
    for (int i = 0; i < 2 && rc == PGV_OK; i++) {
        if (hipHostMalloc(&buf[i], piece, hipHostMallocDefault) != hipSuccess) rc = PGV_ERR_NOMEM;
        if (rc == PGV_OK && hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) != hipSuccess) rc = PGV_ERR_DEVICE;
    }
    auto issue = [&](int64_t c, int slot) -> int {
        const int64_t r0 = c * chunk_rows, cnt = std::min(chunk_rows, n - r0);
        char *dst = static_cast<char *>(buf[slot]);
        const char *src = static_cast<const char *>(ix->vectors) + (size_t)r0 * padded;
        if (padded == tight)
            PGV_HIP(hipMemcpyAsync(dst, src, tight * (size_t)cnt, hipMemcpyDeviceToHost, ctx->stream));
        else
            PGV_HIP(hipMemcpy2DAsync(dst, tight, src, padded, tight, (size_t)cnt, hipMemcpyDeviceToHost, ctx->stream));
        if (ix->tids)
            PGV_HIP(hipMemcpyAsync(dst + tight * (size_t)chunk_rows, ix->tids + r0, sizeof(uint64_t) * (size_t)cnt,
                                   hipMemcpyDeviceToHost, ctx->stream));
        PGV_HIP(hipEventRecord(ev[slot], ctx->stream));
        return PGV_OK;
    };
    const int64_t nchunks = (n + chunk_rows - 1) / chunk_rows;
    if (rc == PGV_OK) rc = issue(0, 0);
    for (int64_t c = 0; c < nchunks && rc == PGV_OK; c++) {
        const int slot = (int)(c & 1);
        if (c + 1 < nchunks) rc = issue(c + 1, slot ^ 1);
        if (rc != PGV_OK) break;
        if (hipEventSynchronize(ev[slot]) != hipSuccess) {
            set_error("pgv_index_drain: copy failed");
            rc = PGV_ERR_DEVICE;
            break;
        }
        const int64_t r0 = c * chunk_rows, cnt = std::min(chunk_rows, n - r0);
        const char *p = static_cast<const char *>(buf[slot]);
        const int src = sink(arg, r0, cnt, p, ix->tids ? reinterpret_cast<const uint64_t *>(p + tight * (size_t)chunk_rows) : nullptr);
        if (src != 0) {
            set_error("pgv_index_drain: the sink returned %d", src);
            rc = PGV_ERR_STATE;
        }
    }
    (void)hipStreamSynchronize(ctx->stream);
    for (int i = 0; i < 2; i++) {
        if (ev[i]) (void)hipEventDestroy(ev[i]);
        if (buf[i]) (void)hipHostFree(buf[i]);
    }
    if (rc == PGV_ERR_NOMEM) set_error("pgv_index_drain: pinned buffers (2 x %zu bytes) could not be allocated", piece);
    return rc;
}

static void index_drop_lanes(pgv_index *ix);

int pgv_index_set_overlap(pgv_index *ix, int lanes) {
    if (!ix) PGV_FAIL(PGV_ERR_ARG, "pgv_index_set_overlap: index is NULL");
    if (lanes < 1 || lanes > 4) PGV_FAIL(PGV_ERR_ARG, "pgv_index_set_overlap: lanes %d outside 1..4", lanes);
    PGV_HIP(hipSetDevice(ix->ctx->device));
    PGV_TRY(pgv_ctx_sync(ix->ctx));
    index_drop_lanes(ix);
    if (lanes == 1) return PGV_OK;
    PGV_HIP(hipEventCreateWithFlags(&ix->lane_event, hipEventDisableTiming));
    PGV_HIP(hipEventCreateWithFlags(&ix->lane_scan_done, hipEventDisableTiming));
    for (int i = 0; i < lanes; i++) {
        pgv_ctx *lc = nullptr;
        pgv_index *v = nullptr;
        int rc = pgv_ctx_create(ix->ctx->device, nullptr, &lc);
        if (rc == PGV_OK) rc = pgv_index_share(ix, lc, &v);
        if (rc != PGV_OK) {
            if (lc) pgv_ctx_destroy(lc);
            index_drop_lanes(ix);
            return rc;
        }
        lc->no_mfma_scan = ix->ctx->no_mfma_scan;
        lc->bound_mode = ix->ctx->bound_mode;
        lc->assign_bound_mode = ix->ctx->assign_bound_mode;
        if (ix->ctx->profiling) (void)pgv_ctx_set_profiling(lc, 1);
        lc->scan_gate = ix->lane_scan_done;
        ix->lanes.push_back(v);
        ix->ctx->children.push_back(lc);
    }
    return PGV_OK;
}

int pgv_index_share(pgv_index *ix, pgv_ctx *ctx, pgv_index **out) {
    if (!ix || !ctx || !out) PGV_FAIL(PGV_ERR_ARG, "pgv_index_share: index/ctx/out is NULL");
    *out = nullptr;
    if (ctx->device != ix->ctx->device)
        PGV_FAIL(PGV_ERR_ARG, "pgv_index_share: the index lives on device %d, the context on %d", ix->ctx->device, ctx->device);
    pgv_index *v = new (std::nothrow) pgv_index(*ix);  // same device arrays, host tables copied
    if (!v) PGV_FAIL(PGV_ERR_NOMEM, "out of host memory");
    v->ctx = ctx;
    v->lanes.clear();  // (the lanes of overlapping batches belong to the handle they were set on)
    v->lane_next = 0;
    v->lane_event = nullptr;
    v->lane_scan_done = nullptr;
    __atomic_add_fetch(ix->refs, 1, __ATOMIC_RELAXED);
    *out = v;
    return PGV_OK;
}

// What crosses the process boundary: the shape of the mirror and the IPC handle of its one allocation.
struct IndexHandleWire {
    uint64_t magic;
    uint32_t abi, pid;
    int32_t device, metric, dtype, dim, nlists, has_tids;
    int64_t nrows;
    uint64_t arena_bytes;
    hipIpcMemHandle_t mem;
};
static_assert(sizeof(IndexHandleWire) <= PGV_INDEX_HANDLE_BYTES, "pgv_index_handle too small");
static constexpr uint64_t kIndexHandleMagic = 0x7067765f69786831ull;  // "pgv_ixh1"

int pgv_index_export(pgv_index *ix, pgv_index_handle *out) {
    if (!ix || !out) PGV_FAIL(PGV_ERR_ARG, "pgv_index_export: index/out is NULL");
    if (!ix->arena) PGV_FAIL(PGV_ERR_STATE, "pgv_index_export: the index has no device arrays");
    if (ix->imported) PGV_FAIL(PGV_ERR_STATE, "pgv_index_export: export from the process that uploaded the index");
    PGV_HIP(hipSetDevice(ix->ctx->device));
    IndexHandleWire w;
    memset(&w, 0, sizeof(w));
    w.magic = kIndexHandleMagic;
    w.abi = PGV_ABI_VERSION;
    w.pid = (uint32_t)getpid();
    w.device = ix->ctx->device;
    w.metric = ix->metric;
    w.dtype = ix->dtype;
    w.dim = ix->dim;
    w.nlists = ix->nlists;
    w.has_tids = ix->tids != nullptr;
    w.nrows = ix->nrows;
    w.arena_bytes = ix->arena_bytes;
    hipError_t e = hipIpcGetMemHandle(&w.mem, ix->arena);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        PGV_FAIL(PGV_ERR_DEVICE,
                 "hipIpcGetMemHandle failed: %s (the driver here shares memory by dmabuf: HSA_ENABLE_IPC_MODE_LEGACY=0 "
                 "must be in the environment of every process)", hipGetErrorString(e));
    }
    memset(out, 0, sizeof(*out));
    memcpy(out->bytes, &w, sizeof(w));
    return PGV_OK;
}

int pgv_index_import(pgv_ctx *ctx, const pgv_index_handle *handle, pgv_index **out) {
    if (!ctx || !handle || !out) PGV_FAIL(PGV_ERR_ARG, "pgv_index_import: ctx/handle/out is NULL");
    *out = nullptr;
    IndexHandleWire w;
    memcpy(&w, handle->bytes, sizeof(w));
    if (w.magic != kIndexHandleMagic || w.abi != PGV_ABI_VERSION)
        PGV_FAIL(PGV_ERR_ARG, "pgv_index_import: not a handle of this library version");
    if (w.pid == (uint32_t)getpid())
        PGV_FAIL(PGV_ERR_STATE, "pgv_index_import: the handle was exported by this process (use pgv_index_share)");
    if (w.device != ctx->device)
        PGV_FAIL(PGV_ERR_ARG, "pgv_index_import: the index lives on device %d, the context on %d", w.device, ctx->device);
    PGV_TRY(check_common((pgv_dtype)w.dtype, w.dim));
    PGV_TRY(check_metric((pgv_metric)w.metric));
    if (w.nlists < 1 || w.nlists > 32768 || w.nrows < 0) PGV_FAIL(PGV_ERR_ARG, "pgv_index_import: corrupt handle");
    PGV_HIP(hipSetDevice(ctx->device));
    pgv_index *ix = new (std::nothrow) pgv_index();
    if (!ix) PGV_FAIL(PGV_ERR_NOMEM, "out of host memory");
    ix->refs = new (std::nothrow) int(1);
    if (!ix->refs) {
        delete ix;
        PGV_FAIL(PGV_ERR_NOMEM, "out of host memory");
    }
    ix->ctx = ctx;
    ix->metric = (pgv_metric)w.metric;
    ix->dtype = (pgv_dtype)w.dtype;
    ix->dim = w.dim;
    ix->nlists = w.nlists;
    ix->nrows = w.nrows;
    ix->geom = row_geom(w.dim, ix->dtype);
    ix->imported = true;
    const size_t row_bytes = (size_t)ix->geom.ld * elem_size(ix->dtype);
    IndexLayout lay = index_layout(w.nlists, w.nrows, row_bytes, w.has_tids != 0, ix->metric == PGV_L2SQ);
    if (lay.bytes != w.arena_bytes) {
        pgv_index_free(ix);
        PGV_FAIL(PGV_ERR_ARG, "pgv_index_import: handle describes %llu bytes, this library lays the mirror out in %zu",
                 (unsigned long long)w.arena_bytes, lay.bytes);
    }
    hipError_t e = hipIpcOpenMemHandle(&ix->arena, w.mem, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        ix->arena = nullptr;
        pgv_index_free(ix);
        PGV_FAIL(PGV_ERR_DEVICE, "hipIpcOpenMemHandle failed: %s (is the exporting process alive, and "
                 "HSA_ENABLE_IPC_MODE_LEGACY=0 set in both?)", hipGetErrorString(e));
    }
    ix->arena_bytes = lay.bytes;
    index_carve(ix, lay);
    // the host-side tables come from the mirror itself
    ix->h_offsets.assign((size_t)w.nlists + 1, 0);
    if (hipMemcpy(ix->h_offsets.data(), ix->list_offsets, sizeof(int64_t) * ix->h_offsets.size(),
                  hipMemcpyDeviceToHost) != hipSuccess) {
        set_error("pgv_index_import: reading list_offsets failed: %s", hipGetErrorString(hipGetLastError()));
        pgv_index_free(ix);
        return PGV_ERR_DEVICE;
    }
    if (ix->h_offsets[0] != 0 || ix->h_offsets[w.nlists] != w.nrows) {
        pgv_index_free(ix);
        PGV_FAIL(PGV_ERR_DATA, "pgv_index_import: the shared mirror does not match its handle");
    }
    index_host_tables(ix);
    *out = ix;
    return PGV_OK;
}

// the device arrays go with the last handle on them (the uploaded index or a pgv_index_share view); an imported
// mirror is unmapped from this process, the exporter's allocation stays
static void index_drop_lanes(pgv_index *ix) {
    for (pgv_index *v : ix->lanes) {
        pgv_ctx *lc = v->ctx;
        if (ix->ctx) {
            auto &ch = ix->ctx->children;
            ch.erase(std::remove(ch.begin(), ch.end(), lc), ch.end());
        }
        pgv_index_free(v);  // (a view: gives its reference back)
        pgv_ctx_destroy(lc);
    }
    ix->lanes.clear();
    if (ix->lane_event) (void)hipEventDestroy(ix->lane_event);
    ix->lane_event = nullptr;
    if (ix->lane_scan_done) (void)hipEventDestroy(ix->lane_scan_done);
    ix->lane_scan_done = nullptr;
}

void pgv_index_free(pgv_index *ix) {
    if (!ix) return;
    if (!ix->lanes.empty()) index_drop_lanes(ix);
    if (ix->ctx) (void)hipStreamSynchronize(ix->ctx->stream);
    if (ix->refs && __atomic_sub_fetch(ix->refs, 1, __ATOMIC_ACQ_REL) > 0) {
        delete ix;
        return;
    }
    if (ix->arena) {
        if (ix->imported)
            (void)hipIpcCloseMemHandle(ix->arena);
        else
            (void)hipFree(ix->arena);
    }
    delete ix->refs;
    delete ix;
}

int pgv_index_tids(pgv_index *ix, const int64_t *slots, int64_t n, uint64_t *out) {
    if (!ix || !out || (n > 0 && !slots)) PGV_FAIL(PGV_ERR_ARG, "pgv_index_tids: index/slots/out is NULL");
    if (!ix->tids) PGV_FAIL(PGV_ERR_STATE, "pgv_index_tids: the index was uploaded without tids");
    if (is_device_ptr(slots) || is_device_ptr(out)) PGV_FAIL(PGV_ERR_ARG, "pgv_index_tids: host arrays only");
    PGV_HIP(hipSetDevice(ix->ctx->device));
    // a scan's slots come in runs (one per list): one copy per run of consecutive slots
    int64_t i = 0;
    while (i < n) {
        if (slots[i] < 0 || slots[i] >= ix->nrows) PGV_FAIL(PGV_ERR_ARG, "pgv_index_tids: slot %lld out of range", (long long)slots[i]);
        int64_t j = i + 1;
        while (j < n && slots[j] == slots[j - 1] + 1) j++;
        if (slots[j - 1] >= ix->nrows) PGV_FAIL(PGV_ERR_ARG, "pgv_index_tids: slot %lld out of range", (long long)slots[j - 1]);
        PGV_HIP(hipMemcpyAsync(out + i, ix->tids + slots[i], sizeof(uint64_t) * (size_t)(j - i), hipMemcpyDeviceToHost,
                               ix->ctx->stream));
        i = j;
    }
    PGV_HIP(hipStreamSynchronize(ix->ctx->stream));
    return PGV_OK;
}

int64_t pgv_index_rows(const pgv_index *ix) { return ix ? ix->nrows : -1; }
int pgv_index_lists(const pgv_index *ix) { return ix ? ix->nlists : -1; }

// scratch of an approximate (MFMA) L2 pass over nq queries keeping kprime candidates each
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


// device-side core of GetScanLists for nq staged queries
static int rank_lists_dev(pgv_index *ix, const void *q_dev, int nq, int maxprobes,
                          int32_t *out_lists_dev, float *out_dist_dev) {
    pgv_ctx *ctx = ix->ctx;
    // distance matrix [nq x nlists], then the maxprobes smallest per row
    PGV_TRY(ctx->dist_mat.ensure(sizeof(float) * (size_t)nq * ix->nlists));
    float *mat = ctx->dist_mat.as<float>();
    PGV_TRY(ctx->sel_a.ensure(sizeof(int64_t) * (size_t)nq * maxprobes));
    float *dist = out_dist_dev;
    if (!dist) {
        PGV_TRY(ctx->sel_b.ensure(sizeof(float) * (size_t)nq * maxprobes));
        dist = ctx->sel_b.as<float>();
    }
    int64_t *pos = ctx->sel_a.as<int64_t>();
    // a batch against a few hundred centers or more: the matrix cores.  Inner product: the values are
    // the result.  L2: the expansion picks maxprobes + 16 candidates, their exact distances decide, and a
    // query whose candidates cannot be proven complete is redone exactly (same scheme as the list scan)
    // a handful of queries: one grid row per query over the centers, selection per query (two launches)
    // (measured on 1000 centers x 1536: ahead of the dense plan + top-k + position cast up to ~24 queries, level at 32)
    if (nq <= 24 && maxprobes <= query_head_cap()) {
        const int64_t cd_stride = ((int64_t)ix->nlists + 7) / 4 * 4;  // 16-byte aligned rows + a float4 of slack
        PGV_TRY(ctx->dist_mat.ensure(sizeof(float) * (size_t)nq * cd_stride));
        return launch_multi_rank(ctx, ix, q_dev, nq, ctx->dist_mat.as<float>(), cd_stride, maxprobes, out_lists_dev,
                                 out_dist_dev);
    }
    const int cand = maxprobes + 16 < ix->nlists ? maxprobes + 16 : ix->nlists;
    const bool mfma = nq >= 128 && ix->nlists >= 64 && !ctx->no_mfma_scan &&
                      (ix->metric == PGV_NEG_IP || (ix->metric == PGV_L2SQ && ix->center_norms && cand <= 256));
    if (mfma && ix->metric == PGV_L2SQ) {
        ApproxScratch sc;
        PGV_TRY(sc.carve(ctx, ctx->ms_b, nq, cand));
        PGV_TRY(dense_scan(ctx, ix->metric, ix->dtype, ix->geom, ix->centers, ix->nlists, q_dev, nq, ix->nlists, mat,
                           true, ix->center_norms, nullptr));
        PGV_TRY(launch_topk_segments(ctx, mat, nullptr, nq, ix->nlists, cand, sc.cand_val, sc.cand_pos, sc.flags + nq));
        const ExactRows xr{ix->centers, nullptr, nullptr, ix->geom, ix->dtype,
                           reinterpret_cast<const unsigned *>(ix->center_norms + ix->nlists)};
        // a center's position in the matrix row is its id: cand_pos serves as the slots
        // the center ids leave as the int32 list ids the callers want (no conversion pass)
        PGV_TRY(launch_batch_recheck(ctx, xr, q_dev, nq, cand, maxprobes, sc.cand_val, sc.cand_pos, sc.cand_pos, nullptr,
                                     ix->nlists, scan_bound(ctx, ix->geom.ld), dist, nullptr, nullptr, sc.flags,
                                     out_lists_dev));
        PGV_TRY(launch_batch_fix(ctx, xr, q_dev, nq, nullptr, nullptr, 0, nullptr, ix->nlists, sc.flags, mat, maxprobes,
                                 scan_bound(ctx, ix->geom.ld), dist, nullptr, nullptr, out_lists_dev));
        return PGV_OK;
    } else {
        PGV_TRY(dense_scan(ctx, ix->metric, ix->dtype, ix->geom, ix->centers, ix->nlists, q_dev, nq, ix->nlists, mat,
                           mfma, nullptr, nullptr));
        PGV_TRY(launch_topk_segments(ctx, mat, nullptr, nq, ix->nlists, maxprobes, dist, pos));
    }
    PGV_TRY(launch_cast_pos_to_i32(ctx, pos, (int64_t)nq * maxprobes, out_lists_dev));
    return PGV_OK;
}

int pgv_rank_lists(pgv_index *ix, const void *queries, int nq, int maxprobes, int32_t *out_lists,
                   float *out_dist) {
    if (!ix || !out_lists) PGV_FAIL(PGV_ERR_ARG, "pgv_rank_lists: index/out_lists is NULL");
    if (nq < 0) PGV_FAIL(PGV_ERR_ARG, "nq < 0");
    if (maxprobes < 1 || maxprobes > ix->nlists)
        PGV_FAIL(PGV_ERR_ARG, "maxprobes %d outside 1..lists (%d)", maxprobes, ix->nlists);
    if (nq == 0) return PGV_OK;
    if (!queries) PGV_FAIL(PGV_ERR_ARG, "queries is NULL");
    pgv_ctx *ctx = ix->ctx;
    PGV_HIP(hipSetDevice(ctx->device));
    const void *q_dev;
    PGV_TRY(stage_rows(ctx, queries, nq, ix->dim, ix->dtype, ix->geom, ctx->q_stage, &q_dev));
    OutArg ol, od;
    PGV_TRY(ol.init(out_lists, sizeof(int32_t) * (size_t)nq * maxprobes, ctx->out_stage));
    PGV_TRY(od.init(out_dist, sizeof(float) * (size_t)nq * maxprobes, ctx->out_stage2));
    PGV_TRY(rank_lists_dev(ix, q_dev, nq, maxprobes, ol.as<int32_t>(), od.as<float>()));
    bool need = false;
    PGV_TRY(ol.finish(ctx, &need));
    PGV_TRY(od.finish(ctx, &need));
    return sync_if(ctx, need);
}

int pgv_scan_lists(pgv_index *ix, const void *query, const int32_t *lists, int nlists,
                   float *out_dist, int64_t *out_slot, int64_t capacity, int64_t *out_count) {
    if (!ix || !out_count) PGV_FAIL(PGV_ERR_ARG, "pgv_scan_lists: index/out_count is NULL");
    if (nlists < 0 || (nlists > 0 && !lists)) PGV_FAIL(PGV_ERR_ARG, "bad list array");
    pgv_ctx *ctx = ix->ctx;
    PGV_HIP(hipSetDevice(ctx->device));

    std::vector<int32_t> hl((size_t)nlists);
    if (nlists) {
        if (is_device_ptr(lists))
            PGV_HIP(hipMemcpy(hl.data(), lists, sizeof(int32_t) * (size_t)nlists, hipMemcpyDeviceToHost));
        else
            memcpy(hl.data(), lists, sizeof(int32_t) * (size_t)nlists);
    }
    int64_t m = 0;
    for (int p = 0; p < nlists; p++) {
        if (hl[p] < 0 || hl[p] >= ix->nlists) PGV_FAIL(PGV_ERR_ARG, "list id %d out of range", hl[p]);
        m += ix->h_offsets[hl[p] + 1] - ix->h_offsets[hl[p]];
    }
    *out_count = m;
    if (m > capacity) PGV_FAIL(PGV_ERR_ARG, "output capacity %lld < %lld tuples", (long long)capacity, (long long)m);
    if (m == 0) return PGV_OK;
    if (!out_dist || !out_slot) PGV_FAIL(PGV_ERR_ARG, "out_dist/out_slot is NULL");

    OutArg od, os;
    PGV_TRY(od.init(out_dist, sizeof(float) * (size_t)m, ctx->out_stage));
    PGV_TRY(os.init(out_slot, sizeof(int64_t) * (size_t)m, ctx->out_stage2));

    // plan on the host: per probed list a run of chunks, all for the one query
    const int ch = rows_per_task_for(ctx, m, 1);
    int64_t ntasks = 0;
    for (int p = 0; p < nlists; p++) {
        int64_t len = ix->h_offsets[hl[p] + 1] - ix->h_offsets[hl[p]];
        ntasks += (len + ch - 1) / ch;
    }
    const size_t tb = sizeof(ScanTask) * (size_t)ntasks, pb = sizeof(ScanPair) * (size_t)nlists,
                 ob = sizeof(int64_t) * (size_t)nlists, lb = sizeof(int32_t) * (size_t)nlists;
    PGV_TRY(staging_acquire(ctx));
    PGV_TRY(ctx->h_a.ensure(tb + pb + ob + lb + 16));
    char *hb = ctx->h_a.as<char>();
    ScanTask *ht = reinterpret_cast<ScanTask *>(hb);
    ScanPair *hp = reinterpret_cast<ScanPair *>(hb + tb);
    int64_t *hoff = reinterpret_cast<int64_t *>(hb + tb + pb);
    int32_t *hlist = reinterpret_cast<int32_t *>(hb + tb + pb + ob);
    int *hn = reinterpret_cast<int *>(hb + tb + pb + ob + lb);
    int64_t t = 0, run = 0;
    for (int p = 0; p < nlists; p++) {
        const int64_t beg = ix->h_offsets[hl[p]], len = ix->h_offsets[hl[p] + 1] - beg;
        hp[p].out_rel = run - beg;
        hp[p].query = 0;
        hp[p].pad = 0;
        hoff[p] = run;
        hlist[p] = hl[p];
        for (int64_t c = 0; c * ch < len; c++) {
            ht[t].row0 = beg + c * ch;
            int64_t left = len - c * ch;
            ht[t].nrows = (int)(left < ch ? left : ch);
            ht[t].pair0 = p;
            ht[t].npairs = 1;
            ht[t].pad = 0;
            t++;
        }
        run += len;
    }
    *hn = (int)ntasks;
    const size_t total = tb + pb + ob + lb + 16;
    PGV_TRY(ctx->tasks.ensure(total));
    PGV_HIP(hipMemcpyAsync(ctx->tasks.p, hb, total, hipMemcpyHostToDevice, ctx->stream));
    char *db = ctx->tasks.as<char>();

    PGV_TRY(launch_iota_slots(ctx, ix, reinterpret_cast<int32_t *>(db + tb + pb + ob), nlists,
                              reinterpret_cast<int64_t *>(db + tb + pb), os.as<int64_t>()));
    if (query == nullptr) {
        // ZeroDistance (src/ivfscan.c:192-196): every tuple at distance 0
        PGV_HIP(hipMemsetAsync(od.dev, 0, sizeof(float) * (size_t)m, ctx->stream));
    } else {
        const void *q_dev;
        PGV_TRY(stage_rows(ctx, query, 1, ix->dim, ix->dtype, ix->geom, ctx->q_stage, &q_dev));
        ScanTimer timer{ctx};
        PGV_TRY(timer.begin((double)m, (double)m));
        PGV_TRY(launch_scan(ctx, ix->metric, ix->dtype, ix->geom, ix->vectors, q_dev,
                            reinterpret_cast<ScanTask *>(db), reinterpret_cast<int *>(db + tb + pb + ob + lb),
                            (int)ntasks, reinterpret_cast<ScanPair *>(db + tb), 1, od.as<float>()));
        PGV_TRY(timer.end());
    }
    bool need = true;  // h_a must be consumed before the next call rewrites it
    PGV_TRY(od.finish(ctx, &need));
    PGV_TRY(os.finish(ctx, &need));
    return sync_if(ctx, need);
}

// k' of the MFMA L2 paths: the candidates kept per query by the expansion's values.  4 k rounded UP to the power of two
// the selection pads to anyway (k = 10: 64 instead of 40 at no cost in topk_kernel, and the recheck reads only the
// rounding band's prefix) -- which is what lets the deterministic band of a 3072-d halfvec scan (~50 candidates wide)
// fit without the widening pass; k + 64 past 64
static int approx_candidates(int k) {
    if (k <= 8) return 32;
    if (4 * k > 256) return k + 64;
    int kp = 64;
    while (kp < 4 * k) kp <<= 1;
    return kp;
}

// GetScanItems + head of the sorted stream for staged queries and device probe lists
// lanes of overlapping batches: this stream's list scan starts when the previous lane's has ended
static int scan_turn_begin(pgv_ctx *ctx) {
    if (ctx->scan_gate) PGV_HIP(hipStreamWaitEvent(ctx->stream, ctx->scan_gate, 0));
    return PGV_OK;
}
static int scan_turn_end(pgv_ctx *ctx) {
    if (ctx->scan_gate) PGV_HIP(hipEventRecord(ctx->scan_gate, ctx->stream));
    return PGV_OK;
}

static int scan_batch_dev(pgv_index *ix, const void *q_dev, int nq, const int32_t *probe_lists, int probes,
                          int k, float *out_dist, int64_t *out_slot, uint64_t *out_tid) {
    pgv_ctx *ctx = ix->ctx;
    // invert to list-major work.  Queries per list on average decides how wide a group is
    // worth.  Lists probed by more than 8 queries go to the tile kernel (16 queries per pass
    // over the rows) when the row shape allows it.
    const double share = (double)nq * probes / (double)ix->nlists;
    // Too few queries to share rows between them (every probed list belongs to one query): the list-major plan
    // gains nothing and costs a dozen launches.  Each query scans its own lists (mq_scan_kernel) and selects
    // its own head (mq_head_kernel): two launches, the single-query kernels with one grid row per query.
    if ((share <= 0.4 || nq <= 4) && nq <= 1024 && probes <= query_max_batch_lists() && k <= query_head_cap()) {
        const int64_t bound = ix->len_prefix[probes];  // rows of the `probes` longest lists
        const int64_t seg_stride = (bound + 7) / 4 * 4;
        PGV_TRY(ctx->plan_d.ensure(sizeof(float) * (size_t)nq * seg_stride));
        OutArg od, os, ot;
        PGV_TRY(od.init(out_dist, sizeof(float) * (size_t)nq * k, ctx->out_stage));
        PGV_TRY(os.init(out_slot, sizeof(int64_t) * (size_t)nq * k, ctx->out_stage2));
        PGV_TRY(ot.init(out_tid, sizeof(uint64_t) * (size_t)nq * k, ctx->sel_b));
        PGV_TRY(scan_turn_begin(ctx));
        ScanTimer timer{ctx};
        PGV_TRY(timer.begin(0.0, 0.0));  // pairs / rows are added up on the device by mq_head_kernel
        PGV_TRY(launch_multi_scan(ctx, ix, q_dev, nq, probe_lists, probes, bound, ctx->plan_d.as<float>(), seg_stride, k,
                                  od.as<float>(), os.as<int64_t>(), ot.as<uint64_t>()));
        PGV_TRY(timer.end());
        PGV_TRY(scan_turn_end(ctx));
        bool need = false;
        PGV_TRY(od.finish(ctx, &need));
        PGV_TRY(os.finish(ctx, &need));
        PGV_TRY(ot.finish(ctx, &need));
        return sync_if(ctx, need);
    }
    // ... and to the matrix cores (32 queries per pass) for L2 / inner product heads of up to 192
    const bool use_mfma = share > 3.0 && k <= 192 && !ctx->no_mfma_scan &&
                          (ix->metric == PGV_NEG_IP || (ix->metric == PGV_L2SQ && ix->row_norms));
    const bool use_tile = !use_mfma && tile_scan_supported(ix->geom) && share > 8.0;
    const int qt = use_mfma ? mfma_scan_queries_per_task()
                            : (use_tile ? tile_scan_queries_per_task()
                                        : scan_group_size(ix->geom, ix->dtype, (int)std::ceil(share)));
    constexpr int rpt_tiles = 20;  // tiles per task (measured best of 10 / 20 / 40 / 80 on the headline batch)
    const int rows_per_task = use_mfma ? mfma_scan_rows_per_task()
                                       : (use_tile ? rpt_tiles * tile_scan_tile_rows(ix->geom)
                                                   : (qt >= 16 ? 256 : (qt >= 4 ? 128 : 64)));
    PlanResult plan;
    PGV_TRY(launch_plan_batch(ctx, ix, probe_lists, nq, probes, qt, rows_per_task, ctx->profiling, &plan));

    // MFMA L2: scratch for the candidates' exact tail
    const bool approx = use_mfma && ix->metric == PGV_L2SQ;
    int kprime = k;
    ApproxScratch sc;
    if (approx) {
        // 32 .. 256 candidates: the head asked for and a margin the rounding bound clears easily (4 k while that
        // fits batch_recheck_kernel's 256, k + 64 beyond)
        kprime = approx_candidates(k);
        PGV_TRY(sc.carve(ctx, ctx->ms_a, nq, kprime));
    }
    float *cand_val = sc.cand_val;
    int64_t *cand_pos = sc.cand_pos;
    int32_t *flags = sc.flags;

    // GetScanItems: one streaming pass
    PGV_TRY(ctx->plan_d.ensure(sizeof(float) * (size_t)(plan.out_bound > 0 ? plan.out_bound : 1)));
    float *seg_vals = ctx->plan_d.as<float>();
    if (plan.ntasks_bound > 0) {
        PGV_TRY(scan_turn_begin(ctx));
        ScanTimer timer{ctx};
        PGV_TRY(timer.begin(0.0, 0.0));  // pairs / rows of this launch are accumulated on the device
        if (use_mfma)
            PGV_TRY(launch_mfma_scan(ctx, ix->metric, ix->dtype, ix->geom, ix->vectors, q_dev, plan.tasks,
                                     plan.ntasks_dev, (int)plan.ntasks_bound, plan.pairs, ix->row_norms, nullptr,
                                     seg_vals, rows_stream_past_caches(ix->geom, ix->dtype, ix->nrows)));
        else if (use_tile)
            PGV_TRY(launch_tile_scan(ctx, ix->metric, ix->dtype, ix->geom, ix->vectors, q_dev, plan.tasks,
                                     plan.ntasks_dev, (int)plan.ntasks_bound, plan.pairs, seg_vals));
        else
            PGV_TRY(launch_scan(ctx, ix->metric, ix->dtype, ix->geom, ix->vectors, q_dev, plan.tasks,
                                plan.ntasks_dev, (int)plan.ntasks_bound, plan.pairs, qt, seg_vals));
        PGV_TRY(timer.end());
        PGV_TRY(scan_turn_end(ctx));
    }

    // head of the sorted stream
    OutArg od, os, ot;
    PGV_TRY(od.init(out_dist, sizeof(float) * (size_t)nq * k, ctx->out_stage));
    PGV_TRY(os.init(out_slot, sizeof(int64_t) * (size_t)nq * k, ctx->out_stage2));
    PGV_TRY(ot.init(out_tid, sizeof(uint64_t) * (size_t)nq * k, ctx->sel_b));
    PGV_TRY(ctx->sel_a.ensure(sizeof(int64_t) * (size_t)nq * k));
    int64_t *pos = ctx->sel_a.as<int64_t>();
    if (approx) {
        // k' candidates by the expansion, their exact distances, the head; queries whose candidate
        // set cannot be proven complete (flags) take the exact pass over their whole segment
        const ScanBound gamma = scan_bound(ctx, ix->geom.ld);
        PGV_TRY(launch_topk_segments(ctx, seg_vals, plan.seg_start, nq, 0, kprime, cand_val, cand_pos, flags + nq));
        const ExactRows xr{ix->vectors, ix->tids, ix->list_offsets, ix->geom, ix->dtype,
                           reinterpret_cast<const unsigned *>(ix->row_norms + ix->nrows)};
        // (the candidates' positions become row slots inside the recheck)
        PGV_TRY(launch_batch_recheck(ctx, xr, q_dev, nq, kprime, k, cand_val, cand_pos, nullptr, plan.seg_start, 0,
                                     gamma, od.as<float>(), os.as<int64_t>(), ot.as<uint64_t>(), flags, nullptr,
                                     probe_lists, plan.probe_off, probes));
        PGV_TRY(launch_batch_fix(ctx, xr, q_dev, nq, probe_lists, plan.probe_off, probes, plan.seg_start, 0, flags,
                                 seg_vals, k, gamma, od.as<float>(), os.as<int64_t>(), ot.as<uint64_t>()));
    } else {
        PGV_TRY(launch_topk_segments(ctx, seg_vals, plan.seg_start, nq, 0, k, od.as<float>(), pos));
        PGV_TRY(launch_positions_to_slots(ctx, ix, probe_lists, plan.probe_off, nq, probes, k, pos,
                                          os.as<int64_t>(), ot.as<uint64_t>()));
    }
    bool need = false;
    PGV_TRY(od.finish(ctx, &need));
    PGV_TRY(os.finish(ctx, &need));
    PGV_TRY(ot.finish(ctx, &need));
    return sync_if(ctx, need);
}

static int check_batch_args(pgv_index *ix, const void *queries, int nq, int probes, int k, float *out_dist,
                            uint64_t *out_tid, const char *who) {
    if (!ix) PGV_FAIL(PGV_ERR_ARG, "%s: index is NULL", who);
    if (nq < 0 || k < 1) PGV_FAIL(PGV_ERR_ARG, "bad nq/k");
    if (probes < 1 || probes > ix->nlists)
        PGV_FAIL(PGV_ERR_ARG, "probes %d outside 1..lists (%d)", probes, ix->nlists);
    if (out_tid && !ix->tids) PGV_FAIL(PGV_ERR_STATE, "index was uploaded without tids");
    if (nq > 0 && (!queries || !out_dist)) PGV_FAIL(PGV_ERR_ARG, "queries/out_dist is NULL");
    return PGV_OK;
}

int pgv_search_batch(pgv_index *ix, const void *queries, int nq, int probes, int k, float *out_dist,
                     int64_t *out_slot, uint64_t *out_tid) {
    PGV_TRY(check_batch_args(ix, queries, nq, probes, k, out_dist, out_tid, "pgv_search_batch"));
    if (nq == 0) return PGV_OK;
    if (!ix->lanes.empty()) {
        // overlapping batches: this one runs on the next lane's stream, behind whatever the caller's stream holds now
        // (device-side queries may still be on their way) and beside the batch the previous call put on another lane
        pgv_index *lane = ix->lanes[ix->lane_next++ % ix->lanes.size()];
        PGV_HIP(hipSetDevice(ix->ctx->device));
        PGV_HIP(hipEventRecord(ix->lane_event, ix->ctx->stream));
        PGV_HIP(hipStreamWaitEvent(lane->ctx->stream, ix->lane_event, 0));
        return pgv_search_batch(lane, queries, nq, probes, k, out_dist, out_slot, out_tid);
    }
    pgv_ctx *ctx = ix->ctx;
    PGV_HIP(hipSetDevice(ctx->device));
    const void *q_dev;
    PGV_TRY(stage_rows(ctx, queries, nq, ix->dim, ix->dtype, ix->geom, ctx->q_stage, &q_dev));
    // GetScanLists for the whole batch
    PGV_TRY(ctx->idx_stage.ensure(sizeof(int32_t) * (size_t)nq * probes));
    int32_t *probe_lists = ctx->idx_stage.as<int32_t>();
    PGV_TRY(rank_lists_dev(ix, q_dev, nq, probes, probe_lists, nullptr));
    return scan_batch_dev(ix, q_dev, nq, probe_lists, probes, k, out_dist, out_slot, out_tid);
}

int pgv_scan_batch(pgv_index *ix, const void *queries, int nq, const int32_t *probe_lists, int probes, int k,
                   float *out_dist, int64_t *out_slot, uint64_t *out_tid) {
    PGV_TRY(check_batch_args(ix, queries, nq, probes, k, out_dist, out_tid, "pgv_scan_batch"));
    if (nq == 0) return PGV_OK;
    if (!probe_lists) PGV_FAIL(PGV_ERR_ARG, "probe_lists is NULL");
    pgv_ctx *ctx = ix->ctx;
    PGV_HIP(hipSetDevice(ctx->device));
    // the planner indexes list_offsets with these ids: host-side lists are checked here; lists that are
    // already on the device must come from pgv_rank_lists (ids in range, distinct per query)
    if (!is_device_ptr(probe_lists)) {
        for (size_t i = 0; i < (size_t)nq * probes; i++)
            if (probe_lists[i] < 0 || probe_lists[i] >= ix->nlists)
                PGV_FAIL(PGV_ERR_ARG, "probe list id %d out of range 0..%d", probe_lists[i], ix->nlists - 1);
    }
    const void *q_dev, *pl_dev;
    PGV_TRY(stage_rows(ctx, queries, nq, ix->dim, ix->dtype, ix->geom, ctx->q_stage, &q_dev));
    PGV_TRY(stage_flat(ctx, probe_lists, sizeof(int32_t) * (size_t)nq * probes, ctx->idx_stage, &pl_dev));
    return scan_batch_dev(ix, q_dev, nq, static_cast<const int32_t *>(pl_dev), probes, k, out_dist, out_slot,
                          out_tid);
}

// ------------------------------------------------------- one query at a time
namespace {

struct QueryHeadHost {  // mirrors QueryHead of kernels_query.hip
    long long total;
    int count;
    unsigned seq;
};

// Admission of single-query scans (threads of ONE process; a Postgres backend is a process of its own and has one
// scan in flight at most).  Measured on MI355X (profiles/r03_single_query_concurrency.md): the device runs ~2.5
// kernels of different streams at a time (4 hardware queues), 16 backends reach 48 k QPS and every backend beyond
// that LOWERS the total (32: 32 k, 64: 16 k -- co-running kernels slow each other down and the runtime interleaves
// barrier packets for every stream switch on a queue).  So at most g_scan_gate_width scan+head pairs are in flight
// per process; the others sleep on a futex.  PGV_MAX_INFLIGHT_SCANS overrides the width (0 = no gate).
static int g_scan_gate_width = -1;
static PgvGate g_scan_gate;  // pgv_gate.h (round 3's version lost wake-ups: the hang of BENCH_r03)
static void scan_gate_enter() {
    if (g_scan_gate_width < 0) {
        const char *e = getenv("PGV_MAX_INFLIGHT_SCANS");
        __atomic_store_n(&g_scan_gate_width, e ? atoi(e) : 16, __ATOMIC_RELAXED);
    }
    g_scan_gate.enter(g_scan_gate_width);
}
static void scan_gate_leave() { g_scan_gate.leave(g_scan_gate_width); }
struct ScanGate {
    ScanGate() { scan_gate_enter(); }
    ~ScanGate() { scan_gate_leave(); }
};

// the head record lands in pinned host memory; its seq word is written last.  Spin on it for a
// while (the kernel's own stores are the fastest completion signal there is), then fall back to
// a stream synchronise.
int wait_head(pgv_ctx *ctx, pgv_query *q, unsigned seq) {
    volatile QueryHeadHost *h = static_cast<volatile QueryHeadHost *>(q->head_pinned);
    for (int spin = 0; spin < 200000; spin++) {
        if (__atomic_load_n(&h->seq, __ATOMIC_ACQUIRE) == seq) return PGV_OK;
        __builtin_ia32_pause();
    }
    PGV_HIP(hipStreamSynchronize(ctx->stream));
    if (__atomic_load_n(&h->seq, __ATOMIC_ACQUIRE) != seq) PGV_FAIL(PGV_ERR_DEVICE, "query kernel did not report");
    return PGV_OK;
}

void copy_head(pgv_query *q, int stride, int n, float *out_dist, int64_t *out_slot, uint64_t *out_tid) {
    const char *base = static_cast<const char *>(q->head_pinned) + 64;
    if (out_slot) memcpy(out_slot, base, sizeof(int64_t) * (size_t)n);
    if (out_tid) memcpy(out_tid, base + (size_t)stride * 8, sizeof(uint64_t) * (size_t)n);
    if (out_dist) memcpy(out_dist, base + (size_t)stride * 16, sizeof(float) * (size_t)n);
}

}  // namespace

int pgv_query_begin(pgv_index *ix, pgv_query **out) {
    if (!ix || !out) PGV_FAIL(PGV_ERR_ARG, "pgv_query_begin: index/out is NULL");
    *out = nullptr;
    pgv_ctx *ctx = ix->ctx;
    PGV_HIP(hipSetDevice(ctx->device));
    pgv_query *q = new (std::nothrow) pgv_query();
    if (!q) PGV_FAIL(PGV_ERR_NOMEM, "out of host memory");
    q->ix = ix;
    const int cap = query_head_cap();
    const size_t state_bytes = sizeof(int32_t) * (size_t)cap + sizeof(float) * ((size_t)ix->nlists + 4);  // + one float4 of slack
    const size_t row_bytes = (size_t)ix->geom.ld * elem_size(ix->dtype);
    q->head_bytes = query_head_bytes(cap);
    int rc = q->state.ensure(state_bytes);
    if (rc == PGV_OK) rc = q->q_dev.ensure(row_bytes);
    if (rc == PGV_OK && hipHostMalloc(&q->q_pinned, row_bytes, hipHostMallocDefault) != hipSuccess) rc = PGV_ERR_NOMEM;
    if (rc == PGV_OK && hipHostMalloc(&q->head_pinned, q->head_bytes, hipHostMallocDefault) != hipSuccess)
        rc = PGV_ERR_NOMEM;
    if (rc == PGV_OK && hipMemsetAsync(q->state.p, 0, state_bytes, ctx->stream) != hipSuccess) rc = PGV_ERR_DEVICE;
    if (rc != PGV_OK) {
        set_error("pgv_query_begin: allocation failed");
        pgv_query_end(q);
        return rc;
    }
    memset(q->head_pinned, 0, q->head_bytes);
    q->lists = q->state.as<int32_t>();
    q->cdist = reinterpret_cast<float *>(q->lists + cap);
    *out = q;
    return PGV_OK;
}

void pgv_query_end(pgv_query *q) {
    if (!q) return;
    if (q->ix && q->ix->ctx) (void)hipStreamSynchronize(q->ix->ctx->stream);
    q->state.release();
    q->seg.release();
    q->q_dev.release();
    if (q->q_pinned) (void)hipHostFree(q->q_pinned);
    if (q->head_pinned) (void)hipHostFree(q->head_pinned);
    delete q;
}

int pgv_query_rank(pgv_query *q, const void *query, int max_probes) {
    if (!q) PGV_FAIL(PGV_ERR_ARG, "pgv_query_rank: q is NULL");
    pgv_index *ix = q->ix;
    pgv_ctx *ctx = ix->ctx;
    if (max_probes < 1 || max_probes > ix->nlists)
        PGV_FAIL(PGV_ERR_ARG, "maxprobes %d outside 1..lists (%d)", max_probes, ix->nlists);
    if (max_probes > query_head_cap())
        PGV_FAIL(PGV_ERR_ARG, "pgv_query_rank handles up to %d probes; use pgv_rank_lists", query_head_cap());
    PGV_HIP(hipSetDevice(ctx->device));
    q->max_probes = max_probes;
    q->is_null = query == nullptr;
    q->cur_n = 0;
    if (q->is_null) return launch_query_iota(ctx, q->lists, max_probes);
    const size_t es = elem_size(ix->dtype);
    const size_t row_bytes = (size_t)ix->geom.ld * es;
    if (is_device_ptr(query)) {
        PGV_HIP(hipMemsetAsync(q->q_dev.p, 0, row_bytes, ctx->stream));
        PGV_HIP(hipMemcpyAsync(q->q_dev.p, query, (size_t)ix->dim * es, hipMemcpyDeviceToDevice, ctx->stream));
    } else {
        // the previous query's kernels have finished reading q_pinned: every pgv_query_scan waits for its head
        // (a rank that no scan followed is waited for here)
        if (q->rank_pending) PGV_HIP(hipStreamSynchronize(ctx->stream));
        memcpy(q->q_pinned, query, (size_t)ix->dim * es);
        if (row_bytes > (size_t)ix->dim * es) memset(static_cast<char *>(q->q_pinned) + (size_t)ix->dim * es, 0, row_bytes - (size_t)ix->dim * es);
        PGV_TRY(launch_query_stage(ctx, q->q_pinned, q->q_dev.p, ix->geom.nvec));
    }
    q->rank_pending = true;
    return launch_query_rank(ctx, ix, q->q_dev.p, q->cdist, max_probes, q->lists);
}

int pgv_query_scan(pgv_query *q, int first, int nprobes, int head, float *out_dist, int64_t *out_slot,
                   uint64_t *out_tid, int *out_count, int64_t *out_total) {
    if (!q || !out_count) PGV_FAIL(PGV_ERR_ARG, "pgv_query_scan: q/out_count is NULL");
    pgv_index *ix = q->ix;
    pgv_ctx *ctx = ix->ctx;
    if (q->max_probes <= 0) PGV_FAIL(PGV_ERR_STATE, "pgv_query_scan before pgv_query_rank");
    if (first < 0 || nprobes < 1 || first + nprobes > q->max_probes)
        PGV_FAIL(PGV_ERR_ARG, "lists [%d, %d) outside the %d ranked", first, first + nprobes, q->max_probes);
    if (nprobes > query_max_batch_lists())
        PGV_FAIL(PGV_ERR_ARG, "pgv_query_scan handles up to %d lists per batch; use pgv_scan_lists", query_max_batch_lists());
    if (head < 1 || head > query_head_cap()) PGV_FAIL(PGV_ERR_ARG, "head %d outside 1..%d", head, query_head_cap());
    if (out_tid && !ix->tids) PGV_FAIL(PGV_ERR_STATE, "index was uploaded without tids");
    PGV_HIP(hipSetDevice(ctx->device));
    const int64_t bound = ix->len_prefix[nprobes];  // rows of the nprobes longest lists
    PGV_TRY(q->seg.ensure(sizeof(float) * (size_t)(bound + 4)));  // + one float4 of slack for the vector loads of the selection
    const unsigned seq = ++q->seq ? q->seq : ++q->seq;  // never 0: the cleared record's value
    ScanGate gate;  // held until the head is back (every return below)
    PGV_TRY(launch_query_scan(ctx, ix, q->is_null ? nullptr : q->q_dev.p, q->lists + first, nprobes, bound,
                              q->seg.as<float>()));
    PGV_TRY(launch_query_head(ctx, ix, q->seg.as<float>(), q->lists + first, nprobes, 0, head, q->head_pinned, seq));
    q->cur_first = first;
    q->cur_n = nprobes;
    PGV_TRY(wait_head(ctx, q, seq));
    q->rank_pending = false;
    const QueryHeadHost *h = static_cast<const QueryHeadHost *>(q->head_pinned);
    *out_count = h->count;
    if (out_total) *out_total = h->total;
    copy_head(q, head, h->count, out_dist, out_slot, out_tid);
    return PGV_OK;
}

int pgv_query_more(pgv_query *q, int skip, int count, float *out_dist, int64_t *out_slot, uint64_t *out_tid,
                   int *out_count) {
    if (!q || !out_count) PGV_FAIL(PGV_ERR_ARG, "pgv_query_more: q/out_count is NULL");
    pgv_index *ix = q->ix;
    pgv_ctx *ctx = ix->ctx;
    if (q->cur_n <= 0) PGV_FAIL(PGV_ERR_STATE, "pgv_query_more before pgv_query_scan");
    if (skip < 0 || count < 1 || skip + count > query_head_cap())
        PGV_FAIL(PGV_ERR_ARG, "skip + count = %d exceeds %d; fetch the batch with pgv_scan_lists", skip + count,
                 query_head_cap());
    if (out_tid && !ix->tids) PGV_FAIL(PGV_ERR_STATE, "index was uploaded without tids");
    PGV_HIP(hipSetDevice(ctx->device));
    const unsigned seq = ++q->seq ? q->seq : ++q->seq;
    PGV_TRY(launch_query_head(ctx, ix, q->seg.as<float>(), q->lists + q->cur_first, q->cur_n, skip, count,
                              q->head_pinned, seq));
    PGV_TRY(wait_head(ctx, q, seq));
    const QueryHeadHost *h = static_cast<const QueryHeadHost *>(q->head_pinned);
    *out_count = h->count;
    copy_head(q, count, h->count, out_dist, out_slot, out_tid);
    return PGV_OK;
}

int pgv_query_lists(pgv_query *q, int32_t *out_lists, int n) {
    if (!q || !out_lists) PGV_FAIL(PGV_ERR_ARG, "pgv_query_lists: NULL argument");
    if (n < 0 || n > q->max_probes) PGV_FAIL(PGV_ERR_ARG, "%d lists asked, %d ranked", n, q->max_probes);
    pgv_ctx *ctx = q->ix->ctx;
    PGV_HIP(hipSetDevice(ctx->device));
    PGV_HIP(hipMemcpyAsync(out_lists, q->lists, sizeof(int32_t) * (size_t)n,
                           is_device_ptr(out_lists) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, ctx->stream));
    PGV_HIP(hipStreamSynchronize(ctx->stream));
    q->rank_pending = false;
    return PGV_OK;
}

// ================================================================= build side

int pgv_assign(pgv_ctx *ctx, pgv_metric metric, pgv_dtype dtype, int dim, const void *centers, int k,
               const void *rows, int64_t n, int32_t *out_list, float *out_dist) {
    if (!ctx || !out_list) PGV_FAIL(PGV_ERR_ARG, "pgv_assign: ctx/out_list is NULL");
    PGV_TRY(check_common(dtype, dim));
    PGV_TRY(check_metric(metric));
    if (k < 1 || !centers) PGV_FAIL(PGV_ERR_ARG, "need at least one center");
    if (n < 0) PGV_FAIL(PGV_ERR_ARG, "n < 0");
    if (n == 0) return PGV_OK;
    if (!rows) PGV_FAIL(PGV_ERR_ARG, "rows is NULL");
    PGV_HIP(hipSetDevice(ctx->device));
    const RowGeom g = row_geom(dim, dtype);
    const void *c_dev;
    PGV_TRY(stage_rows(ctx, centers, k, dim, dtype, g, ctx->centers_stage, &c_dev));

    const bool rows_dev = is_device_ptr(rows);
    const bool out_dev = is_device_ptr(out_list);
    const bool dist_dev = out_dist && is_device_ptr(out_dist);
    const size_t es = elem_size(dtype);
    // host rows are staged in slabs (BuildCallback batches); device rows go in one piece
    const int64_t slab = (rows_dev && g.ld == dim) ? n : (int64_t)1 << 18;
    bool need = false;
    for (int64_t r0 = 0; r0 < n; r0 += slab) {
        const int64_t cnt = n - r0 < slab ? n - r0 : slab;
        const void *r_dev;
        PGV_TRY(stage_rows(ctx, static_cast<const char *>(rows) + (size_t)r0 * dim * es, cnt, dim,
                           dtype, g, ctx->rows_stage, &r_dev));
        int32_t *idx = out_list + r0;
        float *val = out_dist ? out_dist + r0 : nullptr;
        if (!out_dev) {
            PGV_TRY(ctx->out_stage.ensure(sizeof(int32_t) * (size_t)cnt));
            idx = ctx->out_stage.as<int32_t>();
        }
        if (out_dist && !dist_dev) {
            PGV_TRY(ctx->out_stage2.ensure(sizeof(float) * (size_t)cnt));
            val = ctx->out_stage2.as<float>();
        }
        PGV_TRY(launch_argmin(ctx, metric, dtype, g, r_dev, cnt, c_dev, k, idx, val));
        if (!out_dev) {
            PGV_HIP(hipMemcpyAsync(out_list + r0, idx, sizeof(int32_t) * (size_t)cnt, hipMemcpyDeviceToHost, ctx->stream));
            need = true;
        }
        if (out_dist && !dist_dev) {
            PGV_HIP(hipMemcpyAsync(out_dist + r0, val, sizeof(float) * (size_t)cnt, hipMemcpyDeviceToHost, ctx->stream));
            need = true;
        }
        if (need && r0 + slab < n) PGV_HIP(hipStreamSynchronize(ctx->stream));  // scratch is reused
    }
    return sync_if(ctx, need);
}

int pgv_distance_batch(pgv_ctx *ctx, pgv_metric metric, pgv_dtype dtype, int dim, const void *query,
                       const void *rows, int64_t n, float *out) {
    if (!ctx || !out) PGV_FAIL(PGV_ERR_ARG, "pgv_distance_batch: ctx/out is NULL");
    PGV_TRY(check_common(dtype, dim));
    PGV_TRY(check_metric(metric));
    if (n < 0) PGV_FAIL(PGV_ERR_ARG, "n < 0");
    if (n == 0) return PGV_OK;
    if (!query || !rows) PGV_FAIL(PGV_ERR_ARG, "query/rows is NULL");
    PGV_HIP(hipSetDevice(ctx->device));
    const RowGeom g = row_geom(dim, dtype);
    const void *q_dev, *r_dev;
    PGV_TRY(stage_rows(ctx, query, 1, dim, dtype, g, ctx->q_stage, &q_dev));
    PGV_TRY(stage_rows(ctx, rows, n, dim, dtype, g, ctx->rows_stage, &r_dev));
    OutArg od;
    PGV_TRY(od.init(out, sizeof(float) * (size_t)n, ctx->out_stage));
    PGV_TRY(dense_scan(ctx, metric, dtype, g, r_dev, n, q_dev, 1, 0, od.as<float>()));
    bool need = false;
    PGV_TRY(od.finish(ctx, &need));
    return sync_if(ctx, need);
}

// The sequential scan + top-N heapsort of `ORDER BY embedding <op> $1 LIMIT k` without an index (the per-row
// l2_distance / vector_negative_inner_product / l1_distance calls of src/vector.c:579-697 and their halfvec twins),
// for a batch of queries against the same rows: one dense "list".  L2 and inner product run on the matrix cores
// (L2: candidates by the expansion, the reference's sum((q - x)^2) for those, queries that cannot be proven
// complete redone exactly -- the scheme of the list scan), L1 and small batches on the vector-ALU kernels.
int pgv_exact_topk(pgv_ctx *ctx, pgv_metric metric, pgv_dtype dtype, int dim, const void *queries, int nq,
                   const void *rows, int64_t n, int k, float *out_dist, int64_t *out_idx) {
    if (!ctx || !out_dist || !out_idx) PGV_FAIL(PGV_ERR_ARG, "pgv_exact_topk: ctx/out_dist/out_idx is NULL");
    PGV_TRY(check_common(dtype, dim));
    PGV_TRY(check_metric(metric));
    if (nq < 0 || n < 0) PGV_FAIL(PGV_ERR_ARG, "pgv_exact_topk: nq/n < 0");
    if (k < 1 || k > 4096) PGV_FAIL(PGV_ERR_ARG, "pgv_exact_topk: k %d outside 1..4096", k);
    if (n > 0xffffffffll) PGV_FAIL(PGV_ERR_ARG, "pgv_exact_topk: more than 2^32 rows");
    if (nq == 0) return PGV_OK;
    if (!queries || (n > 0 && !rows)) PGV_FAIL(PGV_ERR_ARG, "pgv_exact_topk: queries/rows is NULL");
    PGV_HIP(hipSetDevice(ctx->device));
    const RowGeom g = row_geom(dim, dtype);
    const size_t row_bytes = (size_t)g.ld * elem_size(dtype);
    const void *q_dev, *r_dev = nullptr;
    PGV_TRY(stage_rows(ctx, queries, nq, dim, dtype, g, ctx->q_stage, &q_dev));
    if (n > 0) PGV_TRY(stage_rows(ctx, rows, n, dim, dtype, g, ctx->rows_stage, &r_dev));
    OutArg od, oi;
    PGV_TRY(od.init(out_dist, sizeof(float) * (size_t)nq * k, ctx->out_stage));
    PGV_TRY(oi.init(out_idx, sizeof(int64_t) * (size_t)nq * k, ctx->out_stage2));

    // the distance matrix of a query chunk stays under 4 GiB (whole 128-query tiles of the dense kernel when it is that
    // large: 1024 queries x 1 M rows are ONE pass, every row tile read from HBM once)
    int chunk = n > 0 ? (int)std::min<int64_t>(nq, std::max<int64_t>(1, ((int64_t)1 << 30) / n)) : nq;
    if (chunk >= 128)
        chunk = chunk / 128 * 128;
    else if (chunk >= 64)
        chunk = chunk / 32 * 32;
    static const bool no_dense = [] {
        const char *e = getenv("PGV_NO_DENSE128");
        return e && atoi(e) != 0;
    }();
    const int kprime = approx_candidates(k);
    const bool l2_mfma = metric == PGV_L2SQ && kprime <= 256 && n > kprime;
    const bool mfma_ok = !ctx->no_mfma_scan && n >= 64 && (metric == PGV_NEG_IP || l2_mfma);
    float *norms = nullptr;
    if (mfma_ok && l2_mfma && nq >= 64) {
        PGV_TRY(ctx->xt_norms.ensure(sizeof(float) * ((size_t)n + 1)));
        norms = ctx->xt_norms.as<float>();
        PGV_HIP(hipMemsetAsync(norms + n, 0, sizeof(float), ctx->stream));
        PGV_TRY(launch_row_norms(ctx, dtype, g, r_dev, n, norms, reinterpret_cast<unsigned *>(norms + n)));
    }
    for (int q0 = 0; q0 < nq; q0 += chunk) {
        const int cn = std::min(chunk, nq - q0);
        const char *qp = static_cast<const char *>(q_dev) + (size_t)q0 * row_bytes;
        float *cd = od.as<float>() + (size_t)q0 * k;
        int64_t *ci = oi.as<int64_t>() + (size_t)q0 * k;
        PGV_TRY(ctx->dist_mat.ensure(sizeof(float) * std::max<size_t>((size_t)cn * (size_t)n, 4)));
        float *mat = ctx->dist_mat.as<float>();
        const bool mfma = mfma_ok && cn >= 64;
        // 128 queries x 128 rows per workgroup (kernels_dense.hip) from 128 queries on: the rows are streamed once per
        // 128 queries instead of once per 32
        const bool dense128 = mfma && cn >= 128 && n >= 128 && !no_dense;
        if (mfma && metric == PGV_L2SQ) {
            ApproxScratch sc;
            PGV_TRY(sc.carve(ctx, ctx->ms_b, cn, kprime));
            // the candidates are proven complete with the rounding bound of the kernel that produced the values
            const ScanBound bound = dense128 ? scan_bound_chain(ctx, g.ld, dense_chain_length(g, dtype)) : scan_bound(ctx, g.ld);
            if (dense128)
                PGV_TRY(launch_mfma_dense(ctx, metric, dtype, g, r_dev, n, qp, cn, norms, mat, n));
            else
                PGV_TRY(dense_scan(ctx, metric, dtype, g, r_dev, n, qp, cn, n, mat, true, norms, nullptr));
            PGV_TRY(launch_topk_segments(ctx, mat, nullptr, cn, n, kprime, sc.cand_val, sc.cand_pos, sc.flags + cn));
            const ExactRows xr{r_dev, nullptr, nullptr, g, dtype, reinterpret_cast<const unsigned *>(norms + n)};
            // a row's position in the matrix row is its index: cand_pos serves as the slots
            PGV_TRY(launch_batch_recheck(ctx, xr, qp, cn, kprime, k, sc.cand_val, sc.cand_pos, sc.cand_pos, nullptr, n,
                                         bound, cd, ci, nullptr, sc.flags));
            PGV_TRY(launch_batch_fix(ctx, xr, qp, cn, nullptr, nullptr, 0, nullptr, n, sc.flags, mat, k, bound,
                                     cd, ci, nullptr));
        } else {
            if (dense128)
                PGV_TRY(launch_mfma_dense(ctx, metric, dtype, g, r_dev, n, qp, cn, nullptr, mat, n));
            else
                PGV_TRY(dense_scan(ctx, metric, dtype, g, r_dev, n, qp, cn, n, mat, mfma, nullptr, nullptr));
            PGV_TRY(launch_topk_segments(ctx, mat, nullptr, cn, n, k, cd, ci));
        }
    }
    bool need = false;
    PGV_TRY(od.finish(ctx, &need));
    PGV_TRY(oi.finish(ctx, &need));
    return sync_if(ctx, need);
}

int pgv_cosine_distance_batch(pgv_ctx *ctx, pgv_dtype dtype, int dim, const void *query, const void *rows,
                              int64_t n, double *out) {
    if (!ctx || !out) PGV_FAIL(PGV_ERR_ARG, "pgv_cosine_distance_batch: ctx/out is NULL");
    PGV_TRY(check_common(dtype, dim));
    if (n < 0) PGV_FAIL(PGV_ERR_ARG, "n < 0");
    if (n == 0) return PGV_OK;
    if (!query || !rows) PGV_FAIL(PGV_ERR_ARG, "query/rows is NULL");
    PGV_HIP(hipSetDevice(ctx->device));
    const RowGeom g = row_geom(dim, dtype);
    const void *q_dev, *r_dev;
    PGV_TRY(stage_rows(ctx, query, 1, dim, dtype, g, ctx->q_stage, &q_dev));
    PGV_TRY(stage_rows(ctx, rows, n, dim, dtype, g, ctx->rows_stage, &r_dev));
    OutArg od;
    PGV_TRY(od.init(out, sizeof(double) * (size_t)n, ctx->out_stage));
    PGV_TRY(launch_cosine(ctx, dtype, g, r_dev, q_dev, n, od.as<double>()));
    bool need = false;
    PGV_TRY(od.finish(ctx, &need));
    return sync_if(ctx, need);
}

int pgv_bit_distance_batch(pgv_ctx *ctx, pgv_bit_metric metric, int nbits, const void *query, const void *rows,
                           int64_t n, double *out) {
    if (!ctx || !out) PGV_FAIL(PGV_ERR_ARG, "pgv_bit_distance_batch: ctx/out is NULL");
    if (metric != PGV_BIT_HAMMING && metric != PGV_BIT_JACCARD) PGV_FAIL(PGV_ERR_ARG, "unknown bit metric %d", (int)metric);
    if (nbits < 0 || nbits > 64000 * 8) PGV_FAIL(PGV_ERR_DIMS, "bit length %d out of range", nbits);
    if (n < 0) PGV_FAIL(PGV_ERR_ARG, "n < 0");
    if (n == 0) return PGV_OK;
    if (!rows || (nbits > 0 && !query)) PGV_FAIL(PGV_ERR_ARG, "query/rows is NULL");
    PGV_HIP(hipSetDevice(ctx->device));
    OutArg od;
    PGV_TRY(od.init(out, sizeof(double) * (size_t)n, ctx->out_stage));
    const int bytes = (nbits + 7) / 8;  // VARBITBYTES
    if (bytes == 0) {
        // empty bit strings: hamming 0, jaccard 1 (no common bit), src/bitutils.c:71, :127-128
        std::vector<double> v((size_t)n, metric == PGV_BIT_HAMMING ? 0.0 : 1.0);
        PGV_HIP(hipMemcpyAsync(od.as<double>(), v.data(), sizeof(double) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
        PGV_HIP(hipStreamSynchronize(ctx->stream));
    } else {
        // bytes are staged like fp16 elements of a (bytes / 2)-dimensional row would be: zero-padded to whole
        // 16-byte vectors (an odd byte count is padded by the 2-D copy as well)
        RowGeom g;
        g.ld = (bytes + 15) / 16 * 16;  // padded row length in BYTES
        g.nvec = g.ld / 16;
        g.lpr_log2 = 6;
        while (g.lpr_log2 > 0 && (1 << (g.lpr_log2 - 1)) >= g.nvec) g.lpr_log2--;
        g.nchunks = (g.nvec + (1 << g.lpr_log2) - 1) >> g.lpr_log2;
        auto stage = [&](const void *src, int64_t cnt, DBuf &scratch, const void **outp) -> int {
            const bool dev = is_device_ptr(src);
            if (dev && g.ld == bytes) {
                *outp = src;
                return PGV_OK;
            }
            PGV_TRY(scratch.ensure((size_t)cnt * g.ld));
            PGV_HIP(hipMemsetAsync(scratch.p, 0, (size_t)cnt * g.ld, ctx->stream));
            PGV_HIP(hipMemcpy2DAsync(scratch.p, (size_t)g.ld, src, (size_t)bytes, (size_t)bytes, (size_t)cnt,
                                     dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ctx->stream));
            if (!dev) PGV_HIP(hipStreamSynchronize(ctx->stream));
            *outp = scratch.p;
            return PGV_OK;
        };
        const void *q_dev, *r_dev;
        PGV_TRY(stage(query, 1, ctx->q_stage, &q_dev));
        PGV_TRY(stage(rows, n, ctx->rows_stage, &r_dev));
        PGV_TRY(launch_bit_distance(ctx, metric == PGV_BIT_HAMMING ? 0 : 1, g, r_dev, q_dev, n, od.as<double>()));
    }
    bool need = false;
    PGV_TRY(od.finish(ctx, &need));
    return sync_if(ctx, need);
}

// --------------------------------------------------------------------- k-means

static bool spherical(pgv_ops ops) { return ops == PGV_OPS_IP || ops == PGV_OPS_COSINE; }

static int check_ops(pgv_ops ops) {
    if (ops != PGV_OPS_L2 && ops != PGV_OPS_IP && ops != PGV_OPS_COSINE)
        PGV_FAIL(PGV_ERR_ARG, "unknown opclass family %d", (int)ops);
    return PGV_OK;
}

// k-means++ on staged (padded, device) samples into padded device centers
static int kmeanspp_dev(pgv_ctx *ctx, pgv_ops ops, pgv_dtype dtype, const RowGeom &g,
                        const void *samples_dev, int n, int k, Rng &rng, void *centers_dev) {
    const size_t row_bytes = (size_t)g.ld * elem_size(dtype);
    const int nblocks = kmpp_block_count(n);
    // km_a: weight[n] | raw[n]   km_b: block_sums[nblocks] | draws[k]   km_c: picked[k]
    PGV_TRY(ctx->km_a.ensure(sizeof(float) * 2 * (size_t)n));
    PGV_TRY(ctx->km_b.ensure(sizeof(double) * ((size_t)nblocks + (size_t)k)));
    PGV_TRY(ctx->km_c.ensure(sizeof(int32_t) * (size_t)k));
    float *weight = ctx->km_a.as<float>();
    float *raw = weight + n;
    double *block_sums = ctx->km_b.as<double>();
    double *draws_dev = block_sums + nblocks;
    int32_t *picked = ctx->km_c.as<int32_t>();

    // the reference draws RandomInt() once, then one RandomDouble() per further center
    // (src/ivfkmeans.c:36, :77): pre-draw them in that order
    const uint32_t first = rng.next_u32() % (uint32_t)n;
    PGV_TRY(ctx->h_b.ensure(sizeof(double) * (size_t)k + sizeof(float) * (size_t)n));
    double *h_draws = ctx->h_b.as<double>();
    for (int i = 0; i + 1 < k; i++) h_draws[i] = rng.next_double();
    float *h_w = reinterpret_cast<float *>(h_draws + k);
    for (int j = 0; j < n; j++) h_w[j] = 3.402823466e+38f;  // FLT_MAX (:39-40)
    PGV_HIP(hipMemcpyAsync(draws_dev, h_draws, sizeof(double) * (size_t)k, hipMemcpyHostToDevice, ctx->stream));
    PGV_HIP(hipMemcpyAsync(weight, h_w, sizeof(float) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    PGV_HIP(hipMemcpyAsync(centers_dev, static_cast<const char *>(samples_dev) + (size_t)first * row_bytes,
                           row_bytes, hipMemcpyDeviceToDevice, ctx->stream));
    const int32_t first_i = (int32_t)first;
    PGV_HIP(hipMemcpyAsync(picked, &first_i, sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    PGV_HIP(hipStreamSynchronize(ctx->stream));

    const pgv_metric km = spherical(ops) ? PGV_NEG_IP : PGV_L2SQ;
    for (int i = 0; i + 1 < k; i++) {
        // distance of every sample to the newest center only (:52-60)
        const void *center_i = static_cast<const char *>(centers_dev) + (size_t)i * row_bytes;
        PGV_TRY(dense_scan(ctx, km, dtype, g, samples_dev, n, center_i, 1, 0, raw));
        PGV_TRY(launch_kmpp_update(ctx, raw, weight, n, spherical(ops) ? 1 : 0, block_sums));
        PGV_TRY(launch_kmpp_pick(ctx, g, samples_dev, n, weight, block_sums, draws_dev, i, centers_dev, picked));
    }
    return PGV_OK;
}

// assignment + per-center sums/counts for staged samples; all outputs device
static int lloyd_partial_dev(pgv_ctx *ctx, pgv_ops ops, pgv_dtype dtype, const RowGeom &g,
                             const void *samples_dev, int n, const void *centers_dev, int k,
                             int32_t *closest_io, float *sums /*[k x ld]*/, int32_t *counts,
                             unsigned long long *changes) {
    // km_d: closest_new[n] | offsets[k+1] | members[n]
    PGV_TRY(ctx->km_d.ensure(sizeof(int32_t) * (2 * (size_t)n + (size_t)k + 1)));
    int32_t *closest_new = ctx->km_d.as<int32_t>();
    int32_t *offsets = closest_new + n;
    int32_t *members = offsets + k + 1;
    PGV_HIP(hipMemsetAsync(counts, 0, sizeof(int32_t) * (size_t)k, ctx->stream));
    PGV_HIP(hipMemsetAsync(changes, 0, sizeof(unsigned long long), ctx->stream));
    if (n > 0) {
        PGV_TRY(launch_argmin_mode(ctx, spherical(ops) ? 3 : 0, dtype, g, samples_dev, n, centers_dev, k,
                                   closest_new, nullptr));
        PGV_TRY(launch_changes_hist(ctx, closest_new, closest_io, n, counts, changes));
    }
    PGV_TRY(launch_members(ctx, closest_io, n, k, counts, offsets, members));
    PGV_TRY(launch_center_sums(ctx, dtype, g, samples_dev, offsets, members, k, sums));
    return PGV_OK;
}

// centers from (all-reduced) sums/counts; counts_host tells which clusters are empty
static int lloyd_finish_dev(pgv_ctx *ctx, pgv_ops ops, pgv_dtype dtype, const RowGeom &g, int dim, int k,
                            const float *sums_dev, const int32_t *counts_dev, const int32_t *counts_host,
                            Rng &rng, void *centers_dev) {
    // empty clusters take dim RandomDouble() draws each, in center order (src/ivfkmeans.c:222-227)
    int nempty = 0;
    for (int c = 0; c < k; c++)
        if (counts_host[c] <= 0) nempty++;
    PGV_TRY(ctx->km_e.ensure(sizeof(int32_t) * (size_t)k + sizeof(float) * ((size_t)nempty * dim + 1)));
    int32_t *refill_row = ctx->km_e.as<int32_t>();
    float *refill = reinterpret_cast<float *>(refill_row + k);
    if (nempty > 0) {
        PGV_TRY(ctx->h_b.ensure(sizeof(int32_t) * (size_t)k + sizeof(float) * (size_t)nempty * dim));
        int32_t *h_row = ctx->h_b.as<int32_t>();
        float *h_fill = reinterpret_cast<float *>(h_row + k);
        int e = 0;
        for (int c = 0; c < k; c++) {
            h_row[c] = -1;
            if (counts_host[c] <= 0) {
                for (int d = 0; d < dim; d++) h_fill[(size_t)e * dim + d] = (float)rng.next_double();
                h_row[c] = e++;
            }
        }
        PGV_HIP(hipMemcpyAsync(refill_row, h_row, sizeof(int32_t) * (size_t)k + sizeof(float) * (size_t)nempty * dim,
                               hipMemcpyHostToDevice, ctx->stream));
        PGV_HIP(hipStreamSynchronize(ctx->stream));
    }
    PGV_TRY(launch_finish_centers(ctx, dtype, g, k, dim, sums_dev, counts_dev, refill, refill_row, centers_dev));
    if (spherical(ops)) {
        PGV_TRY(ctx->km_f.ensure(64));
        int32_t *flag = ctx->km_f.as<int32_t>();
        PGV_HIP(hipMemsetAsync(flag, 0, sizeof(int32_t), ctx->stream));
        PGV_TRY(launch_normalize_rows(ctx, dtype, g, centers_dev, k, dim, flag));
    }
    return PGV_OK;
}

static int check_centers_dev(pgv_ctx *ctx, pgv_ops ops, pgv_dtype dtype, const RowGeom &g, int dim, int k,
                             const void *centers_dev) {
    PGV_TRY(ctx->km_f.ensure(64));
    int32_t *flag = ctx->km_f.as<int32_t>();
    PGV_HIP(hipMemsetAsync(flag, 0, sizeof(int32_t), ctx->stream));
    PGV_TRY(launch_check_centers(ctx, dtype, g, centers_dev, k, dim, ops == PGV_OPS_COSINE ? 1 : 0, flag));
    int32_t h = 0;
    PGV_HIP(hipMemcpyAsync(&h, flag, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    PGV_HIP(hipStreamSynchronize(ctx->stream));
    // messages of src/ivfkmeans.c:507-510, :533
    if (h & 2) PGV_FAIL(PGV_ERR_DATA, "NaN detected. Please report a bug.");
    if (h & 4) PGV_FAIL(PGV_ERR_DATA, "Infinite value detected. Please report a bug.");
    if (h & 8) PGV_FAIL(PGV_ERR_DATA, "Zero norm detected. Please report a bug.");
    return PGV_OK;
}

int pgv_kmeanspp_init(pgv_ctx *ctx, pgv_ops ops, pgv_dtype dtype, int dim, const void *samples, int n,
                      int k, const pgv_rng *rng, void *out_centers) {
    if (!ctx || !out_centers) PGV_FAIL(PGV_ERR_ARG, "pgv_kmeanspp_init: ctx/out_centers is NULL");
    PGV_TRY(check_common(dtype, dim));
    PGV_TRY(check_ops(ops));
    if (k < 1 || n < 1 || !samples) PGV_FAIL(PGV_ERR_ARG, "need samples and k >= 1");
    PGV_HIP(hipSetDevice(ctx->device));
    const RowGeom g = row_geom(dim, dtype);
    const void *s_dev;
    PGV_TRY(stage_rows(ctx, samples, n, dim, dtype, g, ctx->rows_stage, &s_dev));
    PGV_TRY(ctx->centers_stage.ensure((size_t)k * g.ld * elem_size(dtype)));
    PGV_HIP(hipMemsetAsync(ctx->centers_stage.p, 0, (size_t)k * g.ld * elem_size(dtype), ctx->stream));
    Rng r(rng);
    PGV_TRY(kmeanspp_dev(ctx, ops, dtype, g, s_dev, n, k, r, ctx->centers_stage.p));
    PGV_TRY(unstage_rows(ctx, ctx->centers_stage.p, k, dim, dtype, g, out_centers));
    return pgv_ctx_sync(ctx);
}

int pgv_lloyd_partial(pgv_ctx *ctx, pgv_ops ops, pgv_dtype dtype, int dim, const void *samples, int n,
                      const void *centers, int k, int32_t *io_closest, float *out_sums,
                      int32_t *out_counts, int64_t *out_changes) {
    if (!ctx || !io_closest || !out_sums || !out_counts || !out_changes)
        PGV_FAIL(PGV_ERR_ARG, "pgv_lloyd_partial: NULL argument");
    PGV_TRY(check_common(dtype, dim));
    PGV_TRY(check_ops(ops));
    if (k < 1 || n < 0 || !centers || (n > 0 && !samples)) PGV_FAIL(PGV_ERR_ARG, "bad sizes");
    PGV_HIP(hipSetDevice(ctx->device));
    const RowGeom g = row_geom(dim, dtype);
    const void *s_dev, *c_dev;
    PGV_TRY(stage_rows(ctx, samples, n, dim, dtype, g, ctx->rows_stage, &s_dev));
    PGV_TRY(stage_rows(ctx, centers, k, dim, dtype, g, ctx->centers_stage, &c_dev));
    // closest is in/out
    int32_t *closest_dev = io_closest;
    const bool closest_is_dev = is_device_ptr(io_closest);
    if (!closest_is_dev) {
        PGV_TRY(ctx->idx_stage.ensure(sizeof(int32_t) * (size_t)(n > 0 ? n : 1)));
        closest_dev = ctx->idx_stage.as<int32_t>();
        if (n) PGV_HIP(hipMemcpyAsync(closest_dev, io_closest, sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    }
    // sums are produced padded [k x ld]; hand back [k x dim]
    PGV_TRY(ctx->km_g.ensure(sizeof(float) * (size_t)k * g.ld + 64));
    float *sums_pad = ctx->km_g.as<float>();
    unsigned long long *changes_dev = reinterpret_cast<unsigned long long *>(sums_pad + (size_t)k * g.ld);
    OutArg oc;
    PGV_TRY(oc.init(out_counts, sizeof(int32_t) * (size_t)k, ctx->out_stage));
    PGV_TRY(lloyd_partial_dev(ctx, ops, dtype, g, s_dev, n, c_dev, k, closest_dev, sums_pad, oc.as<int32_t>(), changes_dev));
    const bool sums_dev = is_device_ptr(out_sums);
    PGV_HIP(hipMemcpy2DAsync(out_sums, sizeof(float) * (size_t)dim, sums_pad, sizeof(float) * (size_t)g.ld,
                             sizeof(float) * (size_t)dim, (size_t)k,
                             sums_dev ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, ctx->stream));
    const bool ch_dev = is_device_ptr(out_changes);
    PGV_HIP(hipMemcpyAsync(out_changes, changes_dev, sizeof(int64_t), ch_dev ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, ctx->stream));
    bool need = !sums_dev || !ch_dev;
    if (!closest_is_dev && n) {
        PGV_HIP(hipMemcpyAsync(io_closest, closest_dev, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
        need = true;
    }
    PGV_TRY(oc.finish(ctx, &need));
    return sync_if(ctx, need);
}

int pgv_lloyd_finish(pgv_ctx *ctx, pgv_ops ops, pgv_dtype dtype, int dim, int k, const float *sums,
                     const int32_t *counts, const pgv_rng *rng, void *out_centers) {
    if (!ctx || !sums || !counts || !out_centers) PGV_FAIL(PGV_ERR_ARG, "pgv_lloyd_finish: NULL argument");
    PGV_TRY(check_common(dtype, dim));
    PGV_TRY(check_ops(ops));
    if (k < 1) PGV_FAIL(PGV_ERR_ARG, "k < 1");
    PGV_HIP(hipSetDevice(ctx->device));
    const RowGeom g = row_geom(dim, dtype);
    // sums arrive [k x dim] fp32; the kernel wants [k x ld]
    PGV_TRY(ctx->km_g.ensure(sizeof(float) * (size_t)k * g.ld + 64));
    float *sums_pad = ctx->km_g.as<float>();
    const bool sums_dev = is_device_ptr(sums);
    PGV_HIP(hipMemsetAsync(sums_pad, 0, sizeof(float) * (size_t)k * g.ld, ctx->stream));
    PGV_HIP(hipMemcpy2DAsync(sums_pad, sizeof(float) * (size_t)g.ld, sums, sizeof(float) * (size_t)dim,
                             sizeof(float) * (size_t)dim, (size_t)k,
                             sums_dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ctx->stream));
    const void *counts_dev;
    std::vector<int32_t> h_counts((size_t)k);
    if (is_device_ptr(counts)) {
        counts_dev = counts;
        PGV_HIP(hipMemcpyAsync(h_counts.data(), counts, sizeof(int32_t) * (size_t)k, hipMemcpyDeviceToHost, ctx->stream));
        PGV_HIP(hipStreamSynchronize(ctx->stream));
    } else {
        memcpy(h_counts.data(), counts, sizeof(int32_t) * (size_t)k);
        PGV_TRY(stage_flat(ctx, counts, sizeof(int32_t) * (size_t)k, ctx->out_stage, &counts_dev));
    }
    PGV_TRY(ctx->centers_stage.ensure((size_t)k * g.ld * elem_size(dtype)));
    Rng r(rng);
    PGV_TRY(lloyd_finish_dev(ctx, ops, dtype, g, dim, k, sums_pad, static_cast<const int32_t *>(counts_dev),
                             h_counts.data(), r, ctx->centers_stage.p));
    PGV_TRY(unstage_rows(ctx, ctx->centers_stage.p, k, dim, dtype, g, out_centers));
    return pgv_ctx_sync(ctx);
}

int pgv_kmeans(pgv_ctx *ctx, pgv_ops ops, pgv_dtype dtype, int dim, const void *samples, int n, int k,
               int max_iterations, const pgv_rng *rng, void *out_centers, int32_t *out_closest,
               int *out_iters) {
    if (!ctx || !out_centers) PGV_FAIL(PGV_ERR_ARG, "pgv_kmeans: ctx/out_centers is NULL");
    PGV_TRY(check_common(dtype, dim));
    PGV_TRY(check_ops(ops));
    if (k < 1 || k > 32768) PGV_FAIL(PGV_ERR_ARG, "lists %d outside 1..32768", k);
    if (n < 0 || (n > 0 && !samples)) PGV_FAIL(PGV_ERR_ARG, "bad samples");
    // spherical opclasses need dim > 1 (src/ivfbuild.c:375-378)
    if (spherical(ops) && dim < 2) PGV_FAIL(PGV_ERR_DIMS, "dimensions must be greater than one for this opclass");
    if (max_iterations <= 0) max_iterations = 500;  // src/ivfkmeans.c:347
    PGV_HIP(hipSetDevice(ctx->device));
    const RowGeom g = row_geom(dim, dtype);
    const size_t row_bytes = (size_t)g.ld * elem_size(dtype);
    Rng r(rng);
    PGV_TRY(ctx->centers_stage.ensure((size_t)k * row_bytes));
    void *centers_dev = ctx->centers_stage.p;
    PGV_HIP(hipMemsetAsync(centers_dev, 0, (size_t)k * row_bytes, ctx->stream));
    int iters = 0;

    if (n == 0) {
        // RandomCenters (src/ivfkmeans.c:110-133): as if every cluster were empty
        std::vector<int32_t> zero((size_t)k, 0);
        PGV_TRY(ctx->km_g.ensure(sizeof(float) * (size_t)k * g.ld + sizeof(int32_t) * (size_t)k));
        float *sums = ctx->km_g.as<float>();
        int32_t *counts = reinterpret_cast<int32_t *>(sums + (size_t)k * g.ld);
        PGV_HIP(hipMemsetAsync(sums, 0, sizeof(float) * (size_t)k * g.ld + sizeof(int32_t) * (size_t)k, ctx->stream));
        PGV_TRY(lloyd_finish_dev(ctx, ops, dtype, g, dim, k, sums, counts, zero.data(), r, centers_dev));
    } else {
        const void *s_dev;
        PGV_TRY(stage_rows(ctx, samples, n, dim, dtype, g, ctx->rows_stage, &s_dev));
        PGV_TRY(kmeanspp_dev(ctx, ops, dtype, g, s_dev, n, k, r, centers_dev));

        // km_g: sums[k x ld] | counts[k] | changes | closest[n]
        PGV_TRY(ctx->km_g.ensure(sizeof(float) * (size_t)k * g.ld + sizeof(int32_t) * ((size_t)k + (size_t)n) + 64));
        float *sums = ctx->km_g.as<float>();
        int32_t *counts = reinterpret_cast<int32_t *>(sums + (size_t)k * g.ld);
        unsigned long long *changes = reinterpret_cast<unsigned long long *>(counts + k + (k & 1));
        int32_t *closest = reinterpret_cast<int32_t *>(changes + 1);
        PGV_HIP(hipMemsetAsync(closest, 0xff, sizeof(int32_t) * (size_t)n, ctx->stream));  // -1: everything "changes" first
        PGV_TRY(staging_acquire(ctx));
        PGV_TRY(ctx->h_a.ensure(sizeof(int32_t) * (size_t)k + 16));
        for (int it = 0; it < max_iterations; it++) {
            iters = it + 1;
            PGV_TRY(lloyd_partial_dev(ctx, ops, dtype, g, s_dev, n, centers_dev, k, closest, sums, counts, changes));
            // counts (which clusters are empty) and the change count steer the host
            int32_t *h_counts = ctx->h_a.as<int32_t>();
            unsigned long long *h_changes = reinterpret_cast<unsigned long long *>(h_counts + k + (k & 1));
            PGV_HIP(hipMemcpyAsync(h_counts, counts, sizeof(int32_t) * (size_t)k, hipMemcpyDeviceToHost, ctx->stream));
            PGV_HIP(hipMemcpyAsync(h_changes, changes, sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
            PGV_HIP(hipStreamSynchronize(ctx->stream));
            const unsigned long long nchanges = *h_changes;
            PGV_TRY(lloyd_finish_dev(ctx, ops, dtype, g, dim, k, sums, counts, h_counts, r, centers_dev));
            // stop when an iteration other than the first reassigns nothing (src/ivfkmeans.c:482-483)
            if (nchanges == 0 && it != 0) break;
        }
        if (out_closest) {
            const bool dev = is_device_ptr(out_closest);
            PGV_HIP(hipMemcpyAsync(out_closest, closest, sizeof(int32_t) * (size_t)n,
                                   dev ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, ctx->stream));
        }
    }
    PGV_TRY(check_centers_dev(ctx, ops, dtype, g, dim, k, centers_dev));
    PGV_TRY(unstage_rows(ctx, centers_dev, k, dim, dtype, g, out_centers));
    if (out_iters) *out_iters = iters;
    return pgv_ctx_sync(ctx);
}

// ================================================================ multi-GPU
// One process per GPU; the collectives are RCCL calls (resolved with dlsym: the library carries no
// link-time dependency on librccl) or the caller's callbacks, always on the context's stream.

namespace {

struct PgvNcclId {  // ncclUniqueId: passed by value
    char internal[PGV_COMM_ID_BYTES];
};

struct RcclApi {
    void *lib = nullptr;
    int (*GetUniqueId)(void *) = nullptr;
    int (*CommInitRank)(void **, int, PgvNcclId, int) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};

int load_rccl(RcclApi **out) {
    static RcclApi api;
    if (!api.lib) {
        void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) PGV_FAIL(PGV_ERR_DEVICE, "librccl not found: %s", dlerror());
        api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
        api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
        api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
        api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(dlsym(h, "ncclAllReduce"));
        api.AllGather = reinterpret_cast<decltype(api.AllGather)>(dlsym(h, "ncclAllGather"));
        api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
        if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllReduce || !api.AllGather)
            PGV_FAIL(PGV_ERR_DEVICE, "librccl lacks an expected entry point");
        api.lib = h;
    }
    *out = &api;
    return PGV_OK;
}

constexpr int kNcclUint8 = 1, kNcclFloat32 = 7, kNcclSum = 0;  // rccl.h: ncclDataType_t / ncclRedOp_t

}  // namespace

struct pgv_comm {
    pgv_ctx *ctx = nullptr;
    int nranks = 1, rank = 0;
    RcclApi *rccl = nullptr;
    void *nccl = nullptr;  // ncclComm_t
    pgv_collectives custom{};
    bool has_custom = false;
    pgv::DBuf a, b, c, d, e, f;  // exchange buffers
    long long *host_rec = nullptr;  // pinned: {changes, empty clusters, sequence} of the Lloyd iteration in flight
    long long seq = 0;
};

namespace {

int comm_all_gather(pgv_comm *cm, const void *send, void *recv, size_t bytes) {
    if (cm->nranks == 1 && !cm->nccl) {
        if (send != recv) PGV_HIP(hipMemcpyAsync(recv, send, bytes, hipMemcpyDeviceToDevice, cm->ctx->stream));
        return PGV_OK;
    }
    if (cm->has_custom && cm->nranks > 1) {
        if (cm->custom.all_gather(cm->custom.state, send, recv, bytes, (void *)cm->ctx->stream) != 0)
            PGV_FAIL(PGV_ERR_DEVICE, "all-gather callback failed");
        return PGV_OK;
    }
    const int rc = cm->rccl->AllGather(send, recv, bytes, kNcclUint8, cm->nccl, cm->ctx->stream);
    if (rc != 0) PGV_FAIL(PGV_ERR_DEVICE, "ncclAllGather: %s", cm->rccl->GetErrorString ? cm->rccl->GetErrorString(rc) : "error");
    return PGV_OK;
}

int comm_all_reduce_f32(pgv_comm *cm, float *buf, size_t count) {
    if (cm->nranks == 1 && !cm->nccl) return PGV_OK;
    if (cm->has_custom && cm->nranks > 1) {
        if (cm->custom.all_reduce_sum_f32(cm->custom.state, buf, count, (void *)cm->ctx->stream) != 0)
            PGV_FAIL(PGV_ERR_DEVICE, "all-reduce callback failed");
        return PGV_OK;
    }
    const int rc = cm->rccl->AllReduce(buf, buf, count, kNcclFloat32, kNcclSum, cm->nccl, cm->ctx->stream);
    if (rc != 0) PGV_FAIL(PGV_ERR_DEVICE, "ncclAllReduce: %s", cm->rccl->GetErrorString ? cm->rccl->GetErrorString(rc) : "error");
    return PGV_OK;
}

int comm_new(pgv_ctx *ctx, int nranks, int rank, pgv_comm **out) {
    if (!ctx || !out) PGV_FAIL(PGV_ERR_ARG, "pgv_comm_create: ctx/out is NULL");
    *out = nullptr;
    if (nranks < 1 || nranks > 16 || rank < 0 || rank >= nranks) PGV_FAIL(PGV_ERR_ARG, "rank %d of %d", rank, nranks);
    pgv_comm *cm = new (std::nothrow) pgv_comm();
    if (!cm) PGV_FAIL(PGV_ERR_NOMEM, "out of host memory");
    cm->ctx = ctx;
    cm->nranks = nranks;
    cm->rank = rank;
    if (hipHostMalloc((void **)&cm->host_rec, 64, hipHostMallocDefault) != hipSuccess) {
        delete cm;
        PGV_FAIL(PGV_ERR_NOMEM, "pinned allocation failed");
    }
    memset(cm->host_rec, 0, 64);
    *out = cm;
    return PGV_OK;
}

}  // namespace

int pgv_comm_unique_id(void *out_id) {
    if (!out_id) PGV_FAIL(PGV_ERR_ARG, "out_id is NULL");
    RcclApi *api;
    PGV_TRY(load_rccl(&api));
    const int rc = api->GetUniqueId(out_id);
    if (rc != 0) PGV_FAIL(PGV_ERR_DEVICE, "ncclGetUniqueId failed (%d)", rc);
    return PGV_OK;
}

int pgv_comm_create(pgv_ctx *ctx, int nranks, int rank, const void *unique_id, pgv_comm **out) {
    pgv_comm *cm;
    PGV_TRY(comm_new(ctx, nranks, rank, &cm));
    if (nranks > 1 && !unique_id) {
        pgv_comm_destroy(cm);
        PGV_FAIL(PGV_ERR_ARG, "unique_id is NULL");
    }
    if (unique_id) {  // a group of one given an id still goes through RCCL (exercises the plumbing)
        int rc = load_rccl(&cm->rccl);
        if (rc != PGV_OK) {
            pgv_comm_destroy(cm);
            return rc;
        }
        const hipError_t he = hipSetDevice(ctx->device);
        if (he != hipSuccess) {  // (PGV_HIP would return past the destroy: the communicator's buffers would leak)
            (void)hipGetLastError();
            pgv_comm_destroy(cm);
            PGV_FAIL(PGV_ERR_DEVICE, "pgv_comm_create: hipSetDevice(%d): %s", ctx->device, hipGetErrorString(he));
        }
        PgvNcclId id;
        memcpy(&id, unique_id, sizeof(id));
        const int nrc = cm->rccl->CommInitRank(&cm->nccl, nranks, id, rank);
        if (nrc != 0) {
            pgv_comm_destroy(cm);
            PGV_FAIL(PGV_ERR_DEVICE, "ncclCommInitRank failed (%d)", nrc);
        }
    }
    *out = cm;
    return PGV_OK;
}

int pgv_comm_create_custom(pgv_ctx *ctx, int nranks, int rank, const pgv_collectives *coll, pgv_comm **out) {
    if (nranks > 1 && (!coll || !coll->all_reduce_sum_f32 || !coll->all_gather))
        PGV_FAIL(PGV_ERR_ARG, "pgv_comm_create_custom: both collectives are needed");
    pgv_comm *cm;
    PGV_TRY(comm_new(ctx, nranks, rank, &cm));
    if (coll) {
        cm->custom = *coll;
        cm->has_custom = true;
    }
    *out = cm;
    return PGV_OK;
}

void pgv_comm_destroy(pgv_comm *cm) {
    if (!cm) return;
    if (cm->ctx) (void)hipStreamSynchronize(cm->ctx->stream);
    if (cm->nccl && cm->rccl) (void)cm->rccl->CommDestroy(cm->nccl);
    pgv::DBuf *bufs[] = {&cm->a, &cm->b, &cm->c, &cm->d, &cm->e, &cm->f};
    for (pgv::DBuf *b : bufs) b->release();
    if (cm->host_rec) (void)hipHostFree(cm->host_rec);
    delete cm;
}

int pgv_comm_size(const pgv_comm *cm) { return cm ? cm->nranks : 0; }
int pgv_comm_rank(const pgv_comm *cm) { return cm ? cm->rank : -1; }

int pgv_kmeans_sharded(pgv_comm *cm, pgv_ops ops, pgv_dtype dtype, int dim, const void *samples, int n, int k,
                       int max_iterations, const pgv_rng *rng, void *out_centers, int32_t *out_closest, int *out_iters) {
    if (!cm || !out_centers) PGV_FAIL(PGV_ERR_ARG, "pgv_kmeans_sharded: comm/out_centers is NULL");
    PGV_TRY(check_common(dtype, dim));
    PGV_TRY(check_ops(ops));
    if (k < 1 || k > 32768) PGV_FAIL(PGV_ERR_ARG, "lists %d outside 1..32768", k);
    if (n < 0 || (n > 0 && !samples)) PGV_FAIL(PGV_ERR_ARG, "bad samples");
    if (spherical(ops) && dim < 2) PGV_FAIL(PGV_ERR_DIMS, "dimensions must be greater than one for this opclass");
    if (max_iterations <= 0) max_iterations = 500;
    pgv_ctx *ctx = cm->ctx;
    const int R = cm->nranks;
    PGV_HIP(hipSetDevice(ctx->device));
    const RowGeom g = row_geom(dim, dtype);
    const size_t row_bytes = (size_t)g.ld * elem_size(dtype);
    Rng r(rng);

    // every rank's sample count (once; the only host round trip before the iterations)
    PGV_TRY(cm->a.ensure(sizeof(int64_t) * (size_t)(R + 1)));
    int64_t *cnt_dev = cm->a.as<int64_t>();
    const int64_t mine = n;
    PGV_HIP(hipMemcpyAsync(cnt_dev + R, &mine, sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
    PGV_TRY(comm_all_gather(cm, cnt_dev + R, cnt_dev, sizeof(int64_t)));
    std::vector<int64_t> cnt((size_t)R);
    PGV_HIP(hipMemcpyAsync(cnt.data(), cnt_dev, sizeof(int64_t) * (size_t)R, hipMemcpyDeviceToHost, ctx->stream));
    PGV_HIP(hipStreamSynchronize(ctx->stream));
    int64_t n_total = 0;
    for (int q = 0; q < R; q++) n_total += cnt[q];
    if (n_total > 0x7fffffff) PGV_FAIL(PGV_ERR_ARG, "too many samples");

    PGV_TRY(ctx->centers_stage.ensure((size_t)k * row_bytes));
    void *centers_dev = ctx->centers_stage.p;
    PGV_HIP(hipMemsetAsync(centers_dev, 0, (size_t)k * row_bytes, ctx->stream));
    int iters = 0;
    std::vector<int32_t> ones((size_t)k, 1);

    // km_g: rec = sums[k x ld] | tail[k + 1] (fp32)   then counts[k] | changes | closest[n]
    const size_t rec_floats = (size_t)k * g.ld + (size_t)k + 1;
    PGV_TRY(ctx->km_g.ensure(sizeof(float) * (rec_floats + 1) + sizeof(int32_t) * ((size_t)k + (size_t)n + 4) + 64));
    float *sums = ctx->km_g.as<float>();
    float *tail = sums + (size_t)k * g.ld;
    int32_t *counts = reinterpret_cast<int32_t *>(sums + ((rec_floats + 1) & ~(size_t)1));
    unsigned long long *changes = reinterpret_cast<unsigned long long *>(counts + k + (k & 1));
    int32_t *closest = reinterpret_cast<int32_t *>(changes + 1);

    if (n_total == 0) {
        // RandomCenters (src/ivfkmeans.c:110-133): as if every cluster were empty; same draws on every rank
        std::vector<int32_t> zero((size_t)k, 0);
        PGV_HIP(hipMemsetAsync(sums, 0, sizeof(float) * rec_floats, ctx->stream));
        PGV_HIP(hipMemsetAsync(counts, 0, sizeof(int32_t) * (size_t)k, ctx->stream));
        PGV_TRY(lloyd_finish_dev(ctx, ops, dtype, g, dim, k, sums, counts, zero.data(), r, centers_dev));
    } else {
        const void *s_dev = nullptr;
        if (n > 0) PGV_TRY(stage_rows(ctx, samples, n, dim, dtype, g, ctx->rows_stage, &s_dev));

        // ---- k-means++ (src/ivfkmeans.c:23-91) over the sharded sample
        const int nblocks = n > 0 ? kmpp_block_count(n) : 0;
        PGV_TRY(ctx->km_a.ensure(sizeof(float) * 2 * (size_t)(n > 0 ? n : 1)));
        PGV_TRY(ctx->km_b.ensure(sizeof(double) * ((size_t)nblocks + (size_t)k + 1)));
        float *weight = ctx->km_a.as<float>();
        float *raw = weight + (n > 0 ? n : 1);
        double *block_sums = ctx->km_b.as<double>();
        double *draws_dev = block_sums + nblocks;
        // cm->b: my total | totals[R]      cm->c: my candidate row | gathered rows [R]      cm->d: owner
        PGV_TRY(cm->b.ensure(sizeof(double) * (size_t)(R + 1)));
        PGV_TRY(cm->c.ensure(row_bytes * (size_t)(R + 1)));
        PGV_TRY(cm->d.ensure(64));
        double *my_total = cm->b.as<double>();
        double *totals = my_total + 1;
        char *send_row = cm->c.as<char>();
        char *rows_all = send_row + row_bytes;
        int32_t *owner = cm->d.as<int32_t>();

        const int64_t first = (int64_t)(r.next_u32() % (uint32_t)n_total);
        PGV_TRY(ctx->h_b.ensure(sizeof(double) * (size_t)k + sizeof(float) * (size_t)(n > 0 ? n : 1)));
        double *h_draws = ctx->h_b.as<double>();
        for (int i = 0; i + 1 < k; i++) h_draws[i] = r.next_double();
        float *h_w = reinterpret_cast<float *>(h_draws + k);
        for (int j = 0; j < n; j++) h_w[j] = 3.402823466e+38f;  // FLT_MAX (:39-40)
        PGV_HIP(hipMemcpyAsync(draws_dev, h_draws, sizeof(double) * (size_t)k, hipMemcpyHostToDevice, ctx->stream));
        if (n > 0) PGV_HIP(hipMemcpyAsync(weight, h_w, sizeof(float) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
        // the first center: the sample RandomInt() % numSamples names, wherever it lives
        int first_owner = 0;
        int64_t at = first;
        while (first_owner < R - 1 && at >= cnt[first_owner]) at -= cnt[first_owner++];
        if (first_owner == cm->rank)
            PGV_HIP(hipMemcpyAsync(send_row, static_cast<const char *>(s_dev) + (size_t)at * row_bytes, row_bytes,
                                   hipMemcpyDeviceToDevice, ctx->stream));
        else
            PGV_HIP(hipMemsetAsync(send_row, 0, row_bytes, ctx->stream));
        PGV_HIP(hipMemcpyAsync(owner, &first_owner, sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
        PGV_HIP(hipStreamSynchronize(ctx->stream));  // h_b and first_owner may go out of scope / be rewritten
        PGV_TRY(comm_all_gather(cm, send_row, rows_all, row_bytes));
        PGV_TRY(launch_kmpp_take_row(ctx, g, rows_all, owner, centers_dev, -1));

        const pgv_metric km = spherical(ops) ? PGV_NEG_IP : PGV_L2SQ;
        for (int i = 0; i + 1 < k; i++) {
            const void *center_i = static_cast<const char *>(centers_dev) + (size_t)i * row_bytes;
            if (n > 0) {
                PGV_TRY(dense_scan(ctx, km, dtype, g, s_dev, n, center_i, 1, 0, raw));
                PGV_TRY(launch_kmpp_update(ctx, raw, weight, n, spherical(ops) ? 1 : 0, block_sums));
                PGV_TRY(launch_kmpp_total(ctx, block_sums, nblocks, my_total));
            } else {
                PGV_HIP(hipMemsetAsync(my_total, 0, sizeof(double), ctx->stream));
            }
            PGV_TRY(comm_all_gather(cm, my_total, totals, sizeof(double)));
            PGV_TRY(launch_kmpp_pick_sharded(ctx, g, s_dev, n, weight, block_sums, totals, R, cm->rank, draws_dev, i,
                                             send_row, owner));
            PGV_TRY(comm_all_gather(cm, send_row, rows_all, row_bytes));
            PGV_TRY(launch_kmpp_take_row(ctx, g, rows_all, owner, centers_dev, i));
        }

        // ---- Lloyd iterations: one fused all-reduce each, the host follows through pinned memory
        if (n > 0) PGV_HIP(hipMemsetAsync(closest, 0xff, sizeof(int32_t) * (size_t)n, ctx->stream));
        for (int it = 0; it < max_iterations; it++) {
            iters = it + 1;
            PGV_TRY(lloyd_partial_dev(ctx, ops, dtype, g, s_dev, n, centers_dev, k, closest, sums, counts, changes));
            PGV_TRY(launch_lloyd_pack(ctx, counts, changes, k, tail));
            PGV_TRY(comm_all_reduce_f32(cm, sums, rec_floats));
            const long long seq = ++cm->seq;
            PGV_TRY(launch_lloyd_unpack(ctx, tail, k, counts, changes, cm->host_rec, seq));
            // enqueue nothing that depends on the host's decision before the record is in
            volatile long long *rec = cm->host_rec;
            bool seen = false;
            for (long spin = 0; spin < 50000000L; spin++) {
                if (__atomic_load_n(&rec[2], __ATOMIC_ACQUIRE) == seq) {
                    seen = true;
                    break;
                }
                __builtin_ia32_pause();
            }
            if (!seen) {
                PGV_HIP(hipStreamSynchronize(ctx->stream));
                if (__atomic_load_n(&rec[2], __ATOMIC_ACQUIRE) != seq)
                    PGV_FAIL(PGV_ERR_DEVICE, "Lloyd iteration did not report");
            }
            const unsigned long long nchanges = (unsigned long long)rec[0];
            const int32_t *h_counts = ones.data();
            std::vector<int32_t> real_counts;
            if (rec[1] > 0) {
                // empty clusters take draws from the rng in center order (src/ivfkmeans.c:222-227): the rare slow path
                real_counts.resize((size_t)k);
                PGV_HIP(hipMemcpyAsync(real_counts.data(), counts, sizeof(int32_t) * (size_t)k, hipMemcpyDeviceToHost,
                                       ctx->stream));
                PGV_HIP(hipStreamSynchronize(ctx->stream));
                h_counts = real_counts.data();
            }
            PGV_TRY(lloyd_finish_dev(ctx, ops, dtype, g, dim, k, sums, counts, h_counts, r, centers_dev));
            if (nchanges == 0 && it != 0) break;
        }
        if (out_closest && n > 0) {
            const bool dev = is_device_ptr(out_closest);
            PGV_HIP(hipMemcpyAsync(out_closest, closest, sizeof(int32_t) * (size_t)n,
                                   dev ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, ctx->stream));
        }
    }
    PGV_TRY(check_centers_dev(ctx, ops, dtype, g, dim, k, centers_dev));
    PGV_TRY(unstage_rows(ctx, centers_dev, k, dim, dtype, g, out_centers));
    if (out_iters) *out_iters = iters;
    return pgv_ctx_sync(ctx);
}

int pgv_search_batch_sharded(pgv_comm *cm, pgv_index *ix, const void *queries, int nq, int probes, int k,
                             float *out_dist, uint64_t *out_tid) {
    if (!cm) PGV_FAIL(PGV_ERR_ARG, "pgv_search_batch_sharded: comm is NULL");
    PGV_TRY(check_batch_args(ix, queries, nq, probes, k, out_dist, out_tid, "pgv_search_batch_sharded"));
    if (!ix->tids) PGV_FAIL(PGV_ERR_STATE, "a sharded index needs heap tids (row slots are rank-local)");
    if (!out_tid) PGV_FAIL(PGV_ERR_ARG, "out_tid is NULL");
    if (nq == 0) return PGV_OK;
    pgv_ctx *ctx = ix->ctx;
    if (ctx != cm->ctx) PGV_FAIL(PGV_ERR_ARG, "index and communicator belong to different contexts");
    const int R = cm->nranks;
    PGV_HIP(hipSetDevice(ctx->device));
    const void *q_dev;
    PGV_TRY(stage_rows(ctx, queries, nq, ix->dim, ix->dtype, ix->geom, ctx->q_stage, &q_dev));
    const size_t row_bytes = (size_t)ix->geom.ld * elem_size(ix->dtype);

    // GetScanLists: every rank ranks its slice of the batch against the replicated centers
    const int per = (nq + R - 1) / R;
    const int lo = cm->rank * per < nq ? cm->rank * per : nq;
    const int hi = lo + per < nq ? lo + per : nq;
    const size_t slice_bytes = sizeof(int32_t) * (size_t)per * probes;
    PGV_TRY(cm->a.ensure(slice_bytes * (size_t)(R + 1)));
    int32_t *lists_mine = cm->a.as<int32_t>();
    int32_t *lists_all = lists_mine + (size_t)per * probes;
    PGV_HIP(hipMemsetAsync(lists_mine, 0, slice_bytes, ctx->stream));
    if (hi > lo)
        PGV_TRY(rank_lists_dev(ix, static_cast<const char *>(q_dev) + (size_t)lo * row_bytes, hi - lo, probes, lists_mine,
                               nullptr));
    PGV_TRY(comm_all_gather(cm, lists_mine, lists_all, slice_bytes));

    // GetScanItems: the probed lists this rank owns, for the whole batch
    const size_t head = (size_t)nq * k;
    PGV_TRY(cm->e.ensure((sizeof(float) + sizeof(uint64_t)) * head * (size_t)(R + 1)));
    float *dist_mine = cm->e.as<float>();
    float *dist_all = dist_mine + head;
    uint64_t *tid_mine = reinterpret_cast<uint64_t *>(dist_all + head * R);
    uint64_t *tid_all = tid_mine + head;
    PGV_TRY(scan_batch_dev(ix, q_dev, nq, lists_all, probes, k, dist_mine, nullptr, tid_mine));
    PGV_TRY(comm_all_gather(cm, dist_mine, dist_all, sizeof(float) * head));
    PGV_TRY(comm_all_gather(cm, tid_mine, tid_all, sizeof(uint64_t) * head));

    // the final top-k merge
    OutArg od, ot;
    PGV_TRY(od.init(out_dist, sizeof(float) * head, ctx->out_stage));
    PGV_TRY(ot.init(out_tid, sizeof(uint64_t) * head, ctx->out_stage2));
    PGV_TRY(launch_merge_heads(ctx, dist_all, tid_all, R, nq, k, od.as<float>(), ot.as<uint64_t>()));
    bool need = false;
    PGV_TRY(od.finish(ctx, &need));
    PGV_TRY(ot.finish(ctx, &need));
    return sync_if(ctx, need);
}

// ======================================================================= HNSW

int pgv_hnsw_upload(pgv_ctx *ctx, pgv_metric metric, pgv_dtype dtype, int dim, const void *elements,
                    int64_t n, pgv_hnsw **out) {
    return pgv_hnsw_upload_payload(ctx, metric, dtype, dim, elements, n, nullptr, 0, out);
}

// where the per-element payload starts inside the elements' allocation
static size_t hnsw_payload_offset(int64_t n, size_t row_bytes) {
    return ((size_t)(n > 0 ? n : 1) * row_bytes + 255) & ~(size_t)255;
}

int pgv_hnsw_upload_payload(pgv_ctx *ctx, pgv_metric metric, pgv_dtype dtype, int dim, const void *elements,
                            int64_t n, const void *payload, int payload_bytes, pgv_hnsw **out) {
    if (!ctx || !out) PGV_FAIL(PGV_ERR_ARG, "pgv_hnsw_upload: ctx/out is NULL");
    *out = nullptr;
    PGV_TRY(check_common(dtype, dim));
    PGV_TRY(check_metric(metric));
    if (n < 0 || (n > 0 && !elements)) PGV_FAIL(PGV_ERR_ARG, "bad elements");
    if (payload_bytes < 0 || payload_bytes > 4096 || (payload_bytes & 3) || (payload_bytes > 0 && n > 0 && !payload))
        PGV_FAIL(PGV_ERR_ARG, "payload: 0..4096 bytes per element in whole words, got %d", payload_bytes);
    PGV_HIP(hipSetDevice(ctx->device));
    pgv_hnsw *h = new (std::nothrow) pgv_hnsw();
    if (!h) PGV_FAIL(PGV_ERR_NOMEM, "out of host memory");
    h->ctx = ctx;
    h->metric = metric;
    h->dtype = dtype;
    h->dim = dim;
    h->n = n;
    h->geom = row_geom(dim, dtype);
    const size_t es = elem_size(dtype), row_bytes = (size_t)h->geom.ld * es;
    const size_t bytes = (size_t)(n > 0 ? n : 1) * row_bytes;
    // the payload (what a scan needs to turn an element into heap TIDs) rides in the same allocation, so that the one
    // IPC handle of the elements carries it to importing processes
    const size_t pay_off = hnsw_payload_offset(n, row_bytes), pay_total = (size_t)payload_bytes * (size_t)(n > 0 ? n : 0);
    if (hipMalloc(&h->elements, payload_bytes > 0 ? pay_off + (pay_total ? pay_total : 4) : bytes) != hipSuccess) {
        delete h;
        PGV_FAIL(PGV_ERR_NOMEM, "hipMalloc(%zu) for hnsw elements failed", bytes);
    }
    h->payload_bytes = payload_bytes;
    h->payload = payload_bytes > 0 ? static_cast<char *>(h->elements) + pay_off : nullptr;
    if (pay_total) {
        hipError_t e = hipMemcpyAsync(h->payload, payload, pay_total,
                                      is_device_ptr(payload) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ctx->stream);
        if (e != hipSuccess) {
            pgv_hnsw_free(h);
            PGV_FAIL(PGV_ERR_DEVICE, "hnsw payload upload failed: %s", hipGetErrorString(e));
        }
    }
    if (n > 0) {
        const bool dev = is_device_ptr(elements);
        hipError_t e;
        if (h->geom.ld == dim) {
            e = hipMemcpyAsync(h->elements, elements, (size_t)n * row_bytes,
                               dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ctx->stream);
        } else {
            e = hipMemsetAsync(h->elements, 0, bytes, ctx->stream);
            if (e == hipSuccess)
                e = hipMemcpy2DAsync(h->elements, row_bytes, elements, (size_t)dim * es, (size_t)dim * es,
                                     (size_t)n, dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ctx->stream);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) {
            pgv_hnsw_free(h);
            PGV_FAIL(PGV_ERR_DEVICE, "hnsw upload failed: %s", hipGetErrorString(e));
        }
    }
    *out = h;
    return PGV_OK;
}

// Searches on this handle's stream see the mirror's last patch, whichever stream ran it (a device-side wait).
static int hnsw_graph_acquire(pgv_hnsw *h) {
    pgv_hnsw *o = h->view_of ? h->view_of : h;
    if (o->graph_ev_set) PGV_HIP(hipStreamWaitEvent(h->ctx->stream, o->graph_ev, 0));
    return PGV_OK;
}

// a view follows its owner: the graph may have been (re)set and the entry point moved since the view was made
static void hnsw_view_refresh(pgv_hnsw *h) {
    const pgv_hnsw *o = h->view_of;
    if (!o) return;
    h->graph = o->graph;
    h->levels = o->levels;
    h->nbr_start = o->nbr_start;
    h->nbr = o->nbr;
    h->m = o->m;
    h->entry = o->entry;
    h->graph_bytes = o->graph_bytes;
    h->nbr_total = o->nbr_total;
}

int pgv_hnsw_device(const pgv_hnsw *h) { return h && h->ctx ? h->ctx->device : -1; }

int pgv_hnsw_share(pgv_hnsw *h, pgv_ctx *ctx, pgv_hnsw **out) {
    if (!h || !ctx || !out) PGV_FAIL(PGV_ERR_ARG, "pgv_hnsw_share: mirror/ctx/out is NULL");
    *out = nullptr;
    if (ctx->device != h->ctx->device) PGV_FAIL(PGV_ERR_ARG, "pgv_hnsw_share: the mirror lives on another device");
    pgv_hnsw *v = new (std::nothrow) pgv_hnsw();
    if (!v) PGV_FAIL(PGV_ERR_NOMEM, "out of host memory");
    v->ctx = ctx;
    v->metric = h->metric;
    v->dtype = h->dtype;
    v->dim = h->dim;
    v->n = h->n;
    v->geom = h->geom;
    v->elements = h->elements;
    v->payload = h->payload;
    v->payload_bytes = h->payload_bytes;
    v->view_of = h->view_of ? h->view_of : h;
    hnsw_view_refresh(v);
    *out = v;
    return PGV_OK;
}

void pgv_hnsw_free(pgv_hnsw *h) {
    if (!h) return;
    if (h->ctx) (void)hipStreamSynchronize(h->ctx->stream);
    if (h->view_of) {  // the owner's allocations stay
        h->bitmaps.release();
        delete h;
        return;
    }
    if (h->graph_ev) (void)hipEventDestroy(h->graph_ev);
    if (h->imported) {
        if (h->elements) (void)hipIpcCloseMemHandle(h->elements);
        if (h->graph) (void)hipIpcCloseMemHandle(h->graph);
    } else {
        if (h->elements) (void)hipFree(h->elements);
        if (h->graph) (void)hipFree(h->graph);
    }
    h->bitmaps.release();
    delete h;
}

struct HnswHandleWire {
    uint64_t magic;
    uint32_t abi, pid;
    int32_t device, metric, dtype, dim, m, entry;
    int64_t n, nbr_total;
    uint64_t graph_bytes;
    int32_t payload_bytes, pad;
    hipIpcMemHandle_t elements, graph;
};
static_assert(sizeof(HnswHandleWire) <= PGV_INDEX_HANDLE_BYTES, "pgv_index_handle too small for an HNSW mirror");
static constexpr uint64_t kHnswHandleMagic = 0x7067765f686e7731ull;  // "pgv_hnw1"

int pgv_hnsw_export(pgv_hnsw *h, pgv_index_handle *out) {
    if (!h || !out) PGV_FAIL(PGV_ERR_ARG, "pgv_hnsw_export: handle/out is NULL");
    if (h->imported || h->view_of) PGV_FAIL(PGV_ERR_STATE, "pgv_hnsw_export: export from the process that uploaded the mirror");
    if (!h->elements || !h->graph || h->m == 0)
        PGV_FAIL(PGV_ERR_STATE, "pgv_hnsw_export: needs a non-empty mirror with its graph set (pgv_hnsw_set_graph)");
    PGV_HIP(hipSetDevice(h->ctx->device));
    PGV_HIP(hipStreamSynchronize(h->ctx->stream));
    HnswHandleWire w;
    memset(&w, 0, sizeof(w));
    w.magic = kHnswHandleMagic;
    w.abi = PGV_ABI_VERSION;
    w.pid = (uint32_t)getpid();
    w.device = h->ctx->device;
    w.metric = h->metric;
    w.dtype = h->dtype;
    w.dim = h->dim;
    w.m = h->m;
    w.entry = h->entry;
    w.n = h->n;
    w.nbr_total = h->nbr_total;
    w.graph_bytes = h->graph_bytes;
    w.payload_bytes = h->payload_bytes;
    hipError_t e = hipIpcGetMemHandle(&w.elements, h->elements);
    if (e == hipSuccess) e = hipIpcGetMemHandle(&w.graph, h->graph);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        PGV_FAIL(PGV_ERR_DEVICE, "hipIpcGetMemHandle failed: %s (HSA_ENABLE_IPC_MODE_LEGACY=0 must be set)", hipGetErrorString(e));
    }
    memset(out, 0, sizeof(*out));
    memcpy(out->bytes, &w, sizeof(w));
    return PGV_OK;
}

int pgv_hnsw_import(pgv_ctx *ctx, const pgv_index_handle *handle, pgv_hnsw **out) {
    if (!ctx || !handle || !out) PGV_FAIL(PGV_ERR_ARG, "pgv_hnsw_import: ctx/handle/out is NULL");
    *out = nullptr;
    HnswHandleWire w;
    memcpy(&w, handle->bytes, sizeof(w));
    if (w.magic != kHnswHandleMagic || w.abi != PGV_ABI_VERSION)
        PGV_FAIL(PGV_ERR_ARG, "pgv_hnsw_import: not an HNSW handle of this library version");
    if (w.pid == (uint32_t)getpid())
        PGV_FAIL(PGV_ERR_STATE, "pgv_hnsw_import: the handle was exported by this process");
    if (w.device != ctx->device)
        PGV_FAIL(PGV_ERR_ARG, "pgv_hnsw_import: the mirror lives on device %d, the context on %d", w.device, ctx->device);
    PGV_TRY(check_common((pgv_dtype)w.dtype, w.dim));
    PGV_TRY(check_metric((pgv_metric)w.metric));
    if (w.n < 1 || w.m < 2 || w.m > 100 || w.entry < -1 || w.entry >= w.n || w.nbr_total < 0 || w.payload_bytes < 0 ||
        w.payload_bytes > 4096)
        PGV_FAIL(PGV_ERR_ARG, "pgv_hnsw_import: corrupt handle");
    PGV_HIP(hipSetDevice(ctx->device));
    pgv_hnsw *h = new (std::nothrow) pgv_hnsw();
    if (!h) PGV_FAIL(PGV_ERR_NOMEM, "out of host memory");
    h->ctx = ctx;
    h->metric = (pgv_metric)w.metric;
    h->dtype = (pgv_dtype)w.dtype;
    h->dim = w.dim;
    h->n = w.n;
    h->geom = row_geom(w.dim, h->dtype);
    h->m = w.m;
    h->entry = w.entry;
    h->imported = true;
    h->nbr_total = w.nbr_total;
    h->graph_bytes = w.graph_bytes;
    hipError_t e = hipIpcOpenMemHandle(&h->elements, w.elements, hipIpcMemLazyEnablePeerAccess);
    if (e == hipSuccess) e = hipIpcOpenMemHandle(&h->graph, w.graph, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        if (h->elements && !h->graph) { (void)hipIpcCloseMemHandle(h->elements); }
        h->elements = h->graph = nullptr;
        delete h;
        PGV_FAIL(PGV_ERR_DEVICE, "hipIpcOpenMemHandle failed: %s", hipGetErrorString(e));
    }
    const size_t lb = ((size_t)h->n * sizeof(int32_t) + 15) / 16 * 16;
    const size_t sb = ((size_t)(h->n + 1) * sizeof(int64_t) + 15) / 16 * 16;
    char *base = static_cast<char *>(h->graph);
    h->levels = reinterpret_cast<const int32_t *>(base);
    h->nbr_start = reinterpret_cast<const int64_t *>(base + lb);
    h->nbr = reinterpret_cast<int32_t *>(base + lb + sb);
    h->payload_bytes = w.payload_bytes;
    h->payload = w.payload_bytes > 0
                     ? static_cast<char *>(h->elements) + hnsw_payload_offset(h->n, (size_t)h->geom.ld * elem_size(h->dtype))
                     : nullptr;
    *out = h;
    return PGV_OK;
}

// the payload rows of the given element slots (a scan's results) -> host memory; slots < 0 give zero bytes
int pgv_hnsw_get_payload(pgv_hnsw *h, const int64_t *elements, int n, void *out) {
    if (!h || !out || (n > 0 && !elements)) PGV_FAIL(PGV_ERR_ARG, "pgv_hnsw_get_payload: handle/elements/out is NULL");
    if (n < 0) PGV_FAIL(PGV_ERR_ARG, "n < 0");
    if (h->payload_bytes <= 0) PGV_FAIL(PGV_ERR_STATE, "pgv_hnsw_get_payload: the mirror was uploaded without a payload");
    if (n == 0) return PGV_OK;
    pgv_ctx *ctx = h->ctx;
    PGV_HIP(hipSetDevice(ctx->device));
    const void *e_dev;
    PGV_TRY(stage_flat(ctx, elements, sizeof(int64_t) * (size_t)n, ctx->idx_stage, &e_dev));
    OutArg oo;
    PGV_TRY(oo.init(out, (size_t)h->payload_bytes * (size_t)n, ctx->out_stage));
    PGV_TRY(launch_gather_words(ctx, h->payload, h->payload_bytes / 4, h->n, static_cast<const int64_t *>(e_dev), n,
                                oo.as<uint32_t>()));
    bool need = false;
    PGV_TRY(oo.finish(ctx, &need));
    return sync_if(ctx, need);
}

int pgv_hnsw_score(pgv_hnsw *h, const void *queries, int nq, const int32_t *slot, const int32_t *query_of,
                   int64_t npairs, float *out) {
    if (!h || !out) PGV_FAIL(PGV_ERR_ARG, "pgv_hnsw_score: handle/out is NULL");
    if (npairs < 0 || nq < 1) PGV_FAIL(PGV_ERR_ARG, "bad sizes");
    if (npairs == 0) return PGV_OK;
    if (!queries || !slot) PGV_FAIL(PGV_ERR_ARG, "queries/slot is NULL");
    pgv_ctx *ctx = h->ctx;
    PGV_HIP(hipSetDevice(ctx->device));
    const void *q_dev, *s_dev, *qo_dev = nullptr;
    PGV_TRY(stage_rows(ctx, queries, nq, h->dim, h->dtype, h->geom, ctx->q_stage, &q_dev));
    PGV_TRY(stage_flat(ctx, slot, sizeof(int32_t) * (size_t)npairs, ctx->idx_stage, &s_dev));
    if (query_of) PGV_TRY(stage_flat(ctx, query_of, sizeof(int32_t) * (size_t)npairs, ctx->plan_d, &qo_dev));
    OutArg od;
    PGV_TRY(od.init(out, sizeof(float) * (size_t)npairs, ctx->out_stage));
    PGV_TRY(launch_score_gather(ctx, h->metric, h->dtype, h->geom, h->elements, q_dev,
                                static_cast<const int32_t *>(s_dev), static_cast<const int32_t *>(qo_dev),
                                npairs, od.as<float>()));
    bool need = false;
    PGV_TRY(od.finish(ctx, &need));
    return sync_if(ctx, need);
}

int pgv_hnsw_set_graph(pgv_hnsw *h, int m, int32_t entry, const int32_t *levels, const int64_t *nbr_start,
                       const int32_t *nbr) {
    if (!h) PGV_FAIL(PGV_ERR_ARG, "pgv_hnsw_set_graph: handle is NULL");
    if (h->imported || h->view_of) PGV_FAIL(PGV_ERR_STATE, "pgv_hnsw_set_graph: an imported mirror / a view is read-only");
    if (m < 2 || m > 100) PGV_FAIL(PGV_ERR_ARG, "m must be 2..100 (src/hnsw.h:55-56), got %d", m);
    if (entry < -1 || entry >= h->n) PGV_FAIL(PGV_ERR_ARG, "entry point %d out of range", (int)entry);
    if (h->n > 0 && (!levels || !nbr_start || !nbr)) PGV_FAIL(PGV_ERR_ARG, "levels/nbr_start/nbr is NULL");
    pgv_ctx *ctx = h->ctx;
    PGV_HIP(hipSetDevice(ctx->device));
    PGV_HIP(hipStreamSynchronize(ctx->stream));  // no search may still be reading the old graph
    if (h->graph) {
        PGV_HIP(hipFree(h->graph));
        h->graph = nullptr;
    }
    h->m = m;
    h->entry = entry;
    if (h->n == 0) return PGV_OK;
    // total neighbor slots: the last offset (it may live on either side)
    int64_t total = 0;
    PGV_HIP(hipMemcpy(&total, nbr_start + h->n, sizeof(int64_t), hipMemcpyDefault));
    if (total < 0) PGV_FAIL(PGV_ERR_ARG, "nbr_start is not an offset array");
    const size_t lb = ((size_t)h->n * sizeof(int32_t) + 15) / 16 * 16;
    const size_t sb = ((size_t)(h->n + 1) * sizeof(int64_t) + 15) / 16 * 16;
    const size_t nb = (size_t)(total > 0 ? total : 1) * sizeof(int32_t);
    if (hipMalloc(&h->graph, lb + sb + nb) != hipSuccess)
        PGV_FAIL(PGV_ERR_NOMEM, "hipMalloc(%zu) for the hnsw graph failed", lb + sb + nb);
    char *base = static_cast<char *>(h->graph);
    PGV_HIP(hipMemcpyAsync(base, levels, (size_t)h->n * sizeof(int32_t), hipMemcpyDefault, ctx->stream));
    PGV_HIP(hipMemcpyAsync(base + lb, nbr_start, (size_t)(h->n + 1) * sizeof(int64_t), hipMemcpyDefault, ctx->stream));
    if (total > 0)
        PGV_HIP(hipMemcpyAsync(base + lb + sb, nbr, (size_t)total * sizeof(int32_t), hipMemcpyDefault, ctx->stream));
    PGV_HIP(hipStreamSynchronize(ctx->stream));
    h->levels = reinterpret_cast<const int32_t *>(base);
    h->nbr_start = reinterpret_cast<const int64_t *>(base + lb);
    h->nbr = reinterpret_cast<int32_t *>(base + lb + sb);
    h->graph_bytes = lb + sb + nb;
    h->nbr_total = total;
    return PGV_OK;
}

int pgv_hnsw_search(pgv_hnsw *h, const void *queries, int nq, int ef_search, int k, int64_t *out_elem,
                    float *out_dist, int64_t *out_scored) {
    if (!h || !out_elem || !out_dist) PGV_FAIL(PGV_ERR_ARG, "pgv_hnsw_search: handle/out is NULL");
    hnsw_view_refresh(h);
    if (nq < 0) PGV_FAIL(PGV_ERR_ARG, "bad query count");
    if (ef_search < 1 || ef_search > 1000)
        PGV_FAIL(PGV_ERR_ARG, "hnsw.ef_search must be 1..1000 (src/hnsw.c:93-94), got %d", ef_search);
    if (k < 1 || k > ef_search) PGV_FAIL(PGV_ERR_ARG, "k must be 1..ef_search, got %d", k);
    if (h->m == 0) PGV_FAIL(PGV_ERR_ARG, "pgv_hnsw_search needs pgv_hnsw_set_graph first");
    if (nq == 0) return PGV_OK;
    if (!queries) PGV_FAIL(PGV_ERR_ARG, "queries is NULL");
    pgv_ctx *ctx = h->ctx;
    PGV_HIP(hipSetDevice(ctx->device));
    PGV_TRY(hnsw_graph_acquire(h));
    const void *q_dev;
    PGV_TRY(stage_rows(ctx, queries, nq, h->dim, h->dtype, h->geom, ctx->q_stage, &q_dev));
    int words = 0;
    const int grid = hnsw_search_grid(ctx, nq, h->n, &words);
    PGV_TRY(h->bitmaps.ensure((size_t)grid * words * sizeof(uint32_t)));
    PGV_TRY(ctx->counters.ensure(256));
    OutArg oe, od, os;
    PGV_TRY(oe.init(out_elem, sizeof(int64_t) * (size_t)nq * k, ctx->out_stage2));
    PGV_TRY(od.init(out_dist, sizeof(float) * (size_t)nq * k, ctx->out_stage));
    PGV_TRY(os.init(out_scored, sizeof(int64_t) * (size_t)nq, ctx->sel_b));
    HnswSearchArgs a;
    a.queries = q_dev;
    a.nq = nq;
    a.ef = ef_search;
    a.k = k;
    a.out_elem = oe.as<int64_t>();
    a.out_dist = od.as<float>();
    a.out_scored = out_scored ? os.as<int64_t>() : nullptr;
    PGV_TRY(launch_hnsw_search(ctx, h->metric, h->dtype, h->geom, h->elements, h->n, h->levels, h->nbr_start,
                               h->nbr, h->m, h->entry, a, h->bitmaps.as<uint32_t>(), words, grid,
                               ctx->counters.as<int>()));
    bool need = false;
    PGV_TRY(oe.finish(ctx, &need));
    PGV_TRY(od.finish(ctx, &need));
    PGV_TRY(os.finish(ctx, &need));
    return sync_if(ctx, need);
}

int pgv_hnsw_build_search(pgv_hnsw *h, const int32_t *elements, const int32_t *insert_levels, int nq,
                          int ef_construction, int layer_cap, int32_t *out_ids, float *out_dist, int32_t *out_count) {
    if (!h || !out_ids || !out_dist || !out_count) PGV_FAIL(PGV_ERR_ARG, "pgv_hnsw_build_search: handle/out is NULL");
    hnsw_view_refresh(h);
    if (nq < 0 || layer_cap < 1) PGV_FAIL(PGV_ERR_ARG, "bad sizes");
    if (ef_construction < 4 || ef_construction > 1000)
        PGV_FAIL(PGV_ERR_ARG, "ef_construction must be 4..1000 (src/hnsw.h:58-59), got %d", ef_construction);
    if (h->m == 0) PGV_FAIL(PGV_ERR_ARG, "pgv_hnsw_build_search needs pgv_hnsw_set_graph first");
    if (nq == 0) return PGV_OK;
    if (!elements || !insert_levels) PGV_FAIL(PGV_ERR_ARG, "elements/insert_levels is NULL");
    pgv_ctx *ctx = h->ctx;
    PGV_HIP(hipSetDevice(ctx->device));
    PGV_TRY(hnsw_graph_acquire(h));
    const void *e_dev, *l_dev;
    PGV_TRY(stage_flat(ctx, elements, sizeof(int32_t) * (size_t)nq, ctx->idx_stage, &e_dev));
    PGV_TRY(stage_flat(ctx, insert_levels, sizeof(int32_t) * (size_t)nq, ctx->plan_d, &l_dev));
    int words = 0;
    const int grid = hnsw_search_grid(ctx, nq, h->n, &words);
    PGV_TRY(h->bitmaps.ensure((size_t)grid * words * sizeof(uint32_t)));
    PGV_TRY(ctx->counters.ensure(256));
    const size_t per = (size_t)nq * layer_cap;
    OutArg oi, od, oc;
    PGV_TRY(oi.init(out_ids, sizeof(int32_t) * per * ef_construction, ctx->out_stage2));
    PGV_TRY(od.init(out_dist, sizeof(float) * per * ef_construction, ctx->out_stage));
    PGV_TRY(oc.init(out_count, sizeof(int32_t) * per, ctx->sel_b));
    HnswSearchArgs a;
    a.qids = static_cast<const int32_t *>(e_dev);
    a.qlevels = static_cast<const int32_t *>(l_dev);
    a.nq = nq;
    a.ef = ef_construction;
    a.k = 0;
    a.lw_ids = oi.as<int32_t>();
    a.lw_dist = od.as<float>();
    a.lw_cnt = oc.as<int32_t>();
    a.lcap = layer_cap;
    PGV_TRY(launch_hnsw_search(ctx, h->metric, h->dtype, h->geom, h->elements, h->n, h->levels, h->nbr_start,
                               h->nbr, h->m, h->entry, a, h->bitmaps.as<uint32_t>(), words, grid,
                               ctx->counters.as<int>()));
    bool need = false;
    PGV_TRY(oi.finish(ctx, &need));
    PGV_TRY(od.finish(ctx, &need));
    PGV_TRY(oc.finish(ctx, &need));
    return sync_if(ctx, need);
}

int pgv_hnsw_score_pairs(pgv_hnsw *h, const int32_t *a, const int32_t *b, int64_t npairs, float *out) {
    if (!h || !out) PGV_FAIL(PGV_ERR_ARG, "pgv_hnsw_score_pairs: handle/out is NULL");
    if (npairs < 0) PGV_FAIL(PGV_ERR_ARG, "bad sizes");
    if (npairs == 0) return PGV_OK;
    if (!a || !b) PGV_FAIL(PGV_ERR_ARG, "a/b is NULL");
    pgv_ctx *ctx = h->ctx;
    PGV_HIP(hipSetDevice(ctx->device));
    const void *a_dev, *b_dev;
    PGV_TRY(stage_flat(ctx, a, sizeof(int32_t) * (size_t)npairs, ctx->idx_stage, &a_dev));
    PGV_TRY(stage_flat(ctx, b, sizeof(int32_t) * (size_t)npairs, ctx->plan_d, &b_dev));
    OutArg od;
    PGV_TRY(od.init(out, sizeof(float) * (size_t)npairs, ctx->out_stage));
    // the element mirror is its own query array: pair i = (row a[i], "query" b[i])
    PGV_TRY(launch_score_gather(ctx, h->metric, h->dtype, h->geom, h->elements, h->elements,
                                static_cast<const int32_t *>(a_dev), static_cast<const int32_t *>(b_dev), npairs,
                                od.as<float>()));
    bool need = false;
    PGV_TRY(od.finish(ctx, &need));
    return sync_if(ctx, need);
}

int pgv_hnsw_score_groups(pgv_hnsw *h, const int32_t *ids, const int64_t *ids_start, const int32_t *from,
                          const int64_t *pair_start, int ngroups, int64_t nids, int64_t npairs, float *out) {
    if (!h || !out) PGV_FAIL(PGV_ERR_ARG, "pgv_hnsw_score_groups: handle/out is NULL");
    if (ngroups < 0 || nids < 0 || npairs < 0) PGV_FAIL(PGV_ERR_ARG, "bad sizes");
    if (ngroups == 0 || npairs == 0) return PGV_OK;
    if (!ids || !ids_start || !from || !pair_start) PGV_FAIL(PGV_ERR_ARG, "ids/ids_start/from/pair_start is NULL");
    pgv_ctx *ctx = h->ctx;
    PGV_HIP(hipSetDevice(ctx->device));
    // the four small tables in one staging buffer, the expanded slot arrays in two scratch buffers
    const size_t b_ids = (sizeof(int32_t) * (size_t)nids + 15) & ~(size_t)15,
                 b_start = sizeof(int64_t) * ((size_t)ngroups + 1),
                 b_from = (sizeof(int32_t) * (size_t)ngroups + 15) & ~(size_t)15;
    PGV_TRY(ctx->km_a.ensure(b_ids + 2 * b_start + b_from));
    char *tab = ctx->km_a.as<char>();
    auto put = [&](void *dst, const void *src, size_t bytes) -> int {
        PGV_HIP(hipMemcpyAsync(dst, src, bytes, is_device_ptr(src) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice,
                               ctx->stream));
        return PGV_OK;
    };
    PGV_TRY(put(tab, ids, sizeof(int32_t) * (size_t)nids));
    PGV_TRY(put(tab + b_ids, ids_start, b_start));
    PGV_TRY(put(tab + b_ids + b_start, pair_start, b_start));
    PGV_TRY(put(tab + b_ids + 2 * b_start, from, sizeof(int32_t) * (size_t)ngroups));
    PGV_TRY(ctx->idx_stage.ensure(sizeof(int32_t) * (size_t)npairs));
    PGV_TRY(ctx->plan_d.ensure(sizeof(int32_t) * (size_t)npairs));
    int32_t *a_dev = ctx->idx_stage.as<int32_t>(), *b_dev = ctx->plan_d.as<int32_t>();
    PGV_TRY(launch_expand_groups(ctx, reinterpret_cast<const int32_t *>(tab),
                                 reinterpret_cast<const int64_t *>(tab + b_ids),
                                 reinterpret_cast<const int32_t *>(tab + b_ids + 2 * b_start),
                                 reinterpret_cast<const int64_t *>(tab + b_ids + b_start), ngroups, a_dev, b_dev));
    OutArg od;
    PGV_TRY(od.init(out, sizeof(float) * (size_t)npairs, ctx->out_stage));
    PGV_TRY(launch_score_gather(ctx, h->metric, h->dtype, h->geom, h->elements, h->elements, a_dev, b_dev, npairs,
                                od.as<float>()));
    bool need = true;  // the host tables above must have been read before the caller reuses them
    PGV_TRY(od.finish(ctx, &need));
    return sync_if(ctx, need);
}

int pgv_hnsw_update_graph(pgv_hnsw *h, int32_t entry, const int32_t *elements, int nupd,
                          const int64_t *tuple_offsets, const int32_t *tuples) {
    if (!h) PGV_FAIL(PGV_ERR_ARG, "pgv_hnsw_update_graph: handle is NULL");
    if (h->imported) PGV_FAIL(PGV_ERR_STATE, "pgv_hnsw_update_graph: an imported mirror is read-only");
    hnsw_view_refresh(h);
    // through a view (pgv_hnsw_share) the patch lands in the owner's arrays, on the view's stream
    pgv_hnsw *o = h->view_of ? h->view_of : h;
    if (o->imported) PGV_FAIL(PGV_ERR_STATE, "pgv_hnsw_update_graph: an imported mirror is read-only");
    if (h->m == 0) PGV_FAIL(PGV_ERR_ARG, "pgv_hnsw_update_graph needs pgv_hnsw_set_graph first");
    if (entry < -1 || entry >= h->n) PGV_FAIL(PGV_ERR_ARG, "entry point %d out of range", (int)entry);
    if (nupd < 0 || (nupd > 0 && (!elements || !tuple_offsets || !tuples))) PGV_FAIL(PGV_ERR_ARG, "bad update");
    pgv_ctx *ctx = h->ctx;
    PGV_HIP(hipSetDevice(ctx->device));
    h->entry = entry;
    o->entry = entry;
    if (nupd == 0) return PGV_OK;
    if (is_device_ptr(tuple_offsets)) PGV_FAIL(PGV_ERR_ARG, "tuple_offsets must be host memory");
    const int64_t total = tuple_offsets[nupd];
    for (int i = 0; i < nupd; i++)
        if (tuple_offsets[i] < 0 || tuple_offsets[i + 1] < tuple_offsets[i])
            PGV_FAIL(PGV_ERR_ARG, "tuple_offsets is not an offset array");
    // an earlier patch that ran on another stream comes first
    PGV_TRY(hnsw_graph_acquire(h));
    const void *id_dev, *tp_dev, *of_dev;
    PGV_TRY(stage_flat(ctx, elements, sizeof(int32_t) * (size_t)nupd, ctx->idx_stage, &id_dev));
    PGV_TRY(stage_flat(ctx, tuples, sizeof(int32_t) * (size_t)(total > 0 ? total : 1), ctx->plan_d, &tp_dev));
    PGV_TRY(stage_flat(ctx, tuple_offsets, sizeof(int64_t) * (size_t)(nupd + 1), ctx->plan_c, &of_dev));
    PGV_TRY(launch_hnsw_patch(ctx, h->nbr, h->nbr_start, h->n, static_cast<const int32_t *>(id_dev),
                              static_cast<const int64_t *>(of_dev), static_cast<const int32_t *>(tp_dev), nupd));
    // later launches on this stream see the patched graph; searches on other streams (the owner's, other views') wait
    // for this event on the device.  The caller keeps searches that READ the old tuples away from the patch: they have
    // returned (every search ends with a stream synchronize) before it calls this.
    if (!o->graph_ev) PGV_HIP(hipEventCreateWithFlags(&o->graph_ev, hipEventDisableTiming));
    PGV_HIP(hipEventRecord(o->graph_ev, ctx->stream));
    o->graph_ev_set = true;
    return PGV_OK;
}

}  // extern "C"
