// pgv_abi.hip -- libpgv_hip's shared utilities (error text, device / pinned buffers) and the context entry points of
// include/pgv_hip.h.  The other areas: pgv_abi_ivf.hip (IVFFlat mirror, builder, scans, one query at a time),
// pgv_abi_build.hip (assignment, exact scans, k-means), pgv_abi_comm.hip (multi-GPU), pgv_abi_hnsw.hip.
#include "pgv_abi_common.h"




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
    bytes += 64;  // slack every buffer has: the selections read whole 16-byte vectors around a ragged segment
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

}  // extern "C"
