// pgv_abi_hnsw.hip -- extern "C" entry points of libpgv_hip (include/pgv_hip.h): the HNSW mirror, its graph, searches and the build's scoring.
// Split out of pgv_abi.hip in round 5 (one unit per area, so that an edit recompiles one of them).
#include "pgv_abi_common.h"

extern "C" {

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
    if (malloc_exportable(&h->elements, payload_bytes > 0 ? pay_off + (pay_total ? pay_total : 4) : bytes) != hipSuccess) {
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

// PGV_HNSW_PAIRS_GATHER=1: the build's pair distances by the older gathered kernel (one pair a lane group, both rows from
// L2) instead of score_groups_kernel's 4 x 4 tiles -- the same values bit for bit, for A/B runs
static bool hnsw_pairs_by_gather() {
    static int v = -1;
    if (v < 0) {
        const char *e = getenv("PGV_HNSW_PAIRS_GATHER");
        v = e && atoi(e) != 0 ? 1 : 0;
    }
    return v != 0;
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

static void hnsw_link_free(pgv_hnsw *o);

void pgv_hnsw_free(pgv_hnsw *h) {
    if (!h) return;
    if (h->ctx) (void)hipStreamSynchronize(h->ctx->stream);
    if (h->view_of) {  // the owner's allocations stay
        h->bitmaps.release();
        delete h;
        return;
    }
    hnsw_link_free(h);
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
    if (malloc_exportable(&h->graph, lb + sb + nb) != hipSuccess)
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

// the searches of a batch into lw_* (device arrays [nq x layer_cap x ef] / [nq x layer_cap]), on ctx's stream
static int hnsw_search_into(pgv_hnsw *h, const int32_t *e_dev, const int32_t *l_dev, int nq, int ef_construction, int layer_cap,
                            int32_t *lw_ids, float *lw_dist, int32_t *lw_cnt) {
    pgv_ctx *ctx = h->ctx;
    int words = 0;
    const int grid = hnsw_search_grid(ctx, nq, h->n, &words);
    PGV_TRY(h->bitmaps.ensure((size_t)grid * words * sizeof(uint32_t)));
    PGV_TRY(ctx->counters.ensure(256));
    HnswSearchArgs a;
    a.qids = e_dev;
    a.qlevels = l_dev;
    a.nq = nq;
    a.ef = ef_construction;
    a.k = 0;
    a.lw_ids = lw_ids;
    a.lw_dist = lw_dist;
    a.lw_cnt = lw_cnt;
    a.lcap = layer_cap;
    return launch_hnsw_search(ctx, h->metric, h->dtype, h->geom, h->elements, h->n, h->levels, h->nbr_start, h->nbr, h->m,
                              h->entry, a, h->bitmaps.as<uint32_t>(), words, grid, ctx->counters.as<int>());
}

// SelectNeighbors over the candidate lists lw_* of a batch (device arrays), on ctx's stream; outputs as pgv_hnsw_build_neighbors
static int hnsw_select_from(pgv_hnsw *h, const int32_t *lw_ids, const float *lw_dist, const int32_t *lw_cnt, const int32_t *l_dev,
                            int nq, int ef_construction, int layer_cap, int32_t *out_ids, float *out_dist, uint8_t *out_closer,
                            int32_t *out_count, int64_t *out_pairs) {
    pgv_ctx *ctx = h->ctx;
    const int m = h->m, stride = 2 * m;
    const size_t per = (size_t)nq * layer_cap;
    const int ngroups = (int)per;
    PGV_TRY(ctx->km_e.ensure(sizeof(int64_t) * (per + 1)));
    int64_t *pair_start = ctx->km_e.as<int64_t>();
    // which lists SelectNeighbors has to thin, and where each one's pair triangle goes; the total comes back (8 bytes:
    // it sizes the next launches)
    PGV_TRY(launch_hnsw_select_plan(ctx, lw_cnt, ngroups, layer_cap, m, pair_start));
    int64_t npairs = 0;
    PGV_HIP(hipMemcpyAsync(&npairs, pair_start + ngroups, sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
    PGV_HIP(hipStreamSynchronize(ctx->stream));
    if (npairs < 0 || npairs > (int64_t)per * ef_construction * (ef_construction - 1) / 2)
        PGV_FAIL(PGV_ERR_STATE, "hnsw select: %lld pairs planned", (long long)npairs);
    if (npairs > 0) {
        PGV_TRY(ctx->dist_mat.ensure(sizeof(float) * (size_t)npairs));
        // CheckElementCloser's HnswGetDistance (src/hnswutils.c:1040-1059), a list's triangle in 4 x 4 tiles
        if (hnsw_pairs_by_gather()) {
            PGV_TRY(ctx->km_f.ensure(sizeof(int32_t) * (size_t)npairs));
            PGV_TRY(ctx->km_g.ensure(sizeof(int32_t) * (size_t)npairs));
            PGV_TRY(launch_hnsw_select_pairs(ctx, lw_ids, lw_cnt, pair_start, ngroups, ef_construction, ctx->km_f.as<int32_t>(),
                                             ctx->km_g.as<int32_t>()));
            PGV_TRY(launch_score_gather(ctx, h->metric, h->dtype, h->geom, h->elements, h->elements, ctx->km_f.as<int32_t>(),
                                        ctx->km_g.as<int32_t>(), npairs, ctx->dist_mat.as<float>()));
        } else
            PGV_TRY(launch_score_groups(ctx, h->metric, h->dtype, h->geom, h->elements, lw_ids, nullptr, ef_construction, lw_cnt,
                                        nullptr, pair_start, ngroups, ctx->dist_mat.as<float>()));
    } else
        PGV_TRY(ctx->dist_mat.ensure(16));
    OutArg oi, od, oc, on;
    PGV_TRY(oi.init(out_ids, sizeof(int32_t) * per * stride, ctx->out_stage));
    PGV_TRY(od.init(out_dist, sizeof(float) * per * stride, ctx->out_stage2));
    PGV_TRY(oc.init(out_closer, per * stride, ctx->sel_a));
    PGV_TRY(on.init(out_count, sizeof(int32_t) * per, ctx->sel_b));
    PGV_TRY(launch_hnsw_select(ctx, lw_ids, lw_dist, lw_cnt, l_dev, pair_start, ctx->dist_mat.as<float>(), ngroups, layer_cap,
                               ef_construction, m, stride, oi.as<int32_t>(), od.as<float>(), oc.as<uint8_t>(), on.as<int32_t>()));
    if (out_pairs) *out_pairs = npairs;
    bool need = false;
    PGV_TRY(oi.finish(ctx, &need));
    PGV_TRY(od.finish(ctx, &need));
    PGV_TRY(oc.finish(ctx, &need));
    PGV_TRY(on.finish(ctx, &need));
    return sync_if(ctx, need);
}

static int hnsw_build_args(pgv_hnsw *h, const char *who, int nq, int ef_construction, int layer_cap) {
    if (nq < 0 || layer_cap < 1) PGV_FAIL(PGV_ERR_ARG, "%s: bad sizes", who);
    if (ef_construction < 4 || ef_construction > 1000)
        PGV_FAIL(PGV_ERR_ARG, "ef_construction must be 4..1000 (src/hnsw.h:58-59), got %d", ef_construction);
    if (h->m == 0) PGV_FAIL(PGV_ERR_ARG, "%s needs pgv_hnsw_set_graph first", who);
    const int stride = 2 * h->m;
    if ((size_t)nq * layer_cap > 0x7fffffff / (size_t)(ef_construction > stride ? ef_construction : stride))
        PGV_FAIL(PGV_ERR_ARG, "%s: batch too large", who);
    return PGV_OK;
}

int pgv_hnsw_build_neighbors(pgv_hnsw *h, const int32_t *elements, const int32_t *insert_levels, int nq,
                             int ef_construction, int layer_cap, int32_t *out_ids, float *out_dist, uint8_t *out_closer,
                             int32_t *out_count, int64_t *out_pairs) {
    if (!h || !out_ids || !out_dist || !out_closer || !out_count)
        PGV_FAIL(PGV_ERR_ARG, "pgv_hnsw_build_neighbors: handle/out is NULL");
    hnsw_view_refresh(h);
    PGV_TRY(hnsw_build_args(h, "pgv_hnsw_build_neighbors", nq, ef_construction, layer_cap));
    if (out_pairs) *out_pairs = 0;
    if (nq == 0) return PGV_OK;
    if (!elements || !insert_levels) PGV_FAIL(PGV_ERR_ARG, "elements/insert_levels is NULL");
    pgv_ctx *ctx = h->ctx;
    PGV_HIP(hipSetDevice(ctx->device));
    PGV_TRY(hnsw_graph_acquire(h));
    const size_t per = (size_t)nq * layer_cap;
    const void *e_dev, *l_dev;
    PGV_TRY(stage_flat(ctx, elements, sizeof(int32_t) * (size_t)nq, ctx->idx_stage, &e_dev));
    PGV_TRY(stage_flat(ctx, insert_levels, sizeof(int32_t) * (size_t)nq, ctx->plan_c, &l_dev));
    // the candidate lists stay on the device: km_b ids | km_c distances | km_d counts
    PGV_TRY(ctx->km_b.ensure(sizeof(int32_t) * per * ef_construction));
    PGV_TRY(ctx->km_c.ensure(sizeof(float) * per * ef_construction));
    PGV_TRY(ctx->km_d.ensure(sizeof(int32_t) * per));
    PGV_TRY(hnsw_search_into(h, static_cast<const int32_t *>(e_dev), static_cast<const int32_t *>(l_dev), nq, ef_construction,
                             layer_cap, ctx->km_b.as<int32_t>(), ctx->km_c.as<float>(), ctx->km_d.as<int32_t>()));
    return hnsw_select_from(h, ctx->km_b.as<int32_t>(), ctx->km_c.as<float>(), ctx->km_d.as<int32_t>(),
                            static_cast<const int32_t *>(l_dev), nq, ef_construction, layer_cap, out_ids, out_dist, out_closer,
                            out_count, out_pairs);
}

// pgv_hnsw_build_neighbors in two halves, for a caller that overlaps them: the searches of a batch, whose candidate
// lists are KEPT on the device in one of two slots of the mirror's build state (pgv_hnsw_link_begin), and the selection
// over a kept slot -- from any view of the mirror, on that view's stream.  The searches of batch n + 2 (other slot, one
// view) then run beside the selection of batch n + 1 (another view).
int pgv_hnsw_build_search_keep(pgv_hnsw *h, const int32_t *elements, const int32_t *insert_levels, int nq, int ef_construction,
                               int layer_cap, int slot) {
    if (!h) PGV_FAIL(PGV_ERR_ARG, "pgv_hnsw_build_search_keep: handle is NULL");
    hnsw_view_refresh(h);
    pgv_hnsw *o = h->view_of ? h->view_of : h;
    if (!o->link) PGV_FAIL(PGV_ERR_STATE, "pgv_hnsw_build_search_keep needs pgv_hnsw_link_begin");
    if (slot < 0 || slot > 1) PGV_FAIL(PGV_ERR_ARG, "slot must be 0 or 1");
    PGV_TRY(hnsw_build_args(h, "pgv_hnsw_build_search_keep", nq, ef_construction, layer_cap));
    HnswLinkState::Kept &K = o->link->kept[slot];
    K.nq = 0;
    if (nq == 0) return PGV_OK;
    if (!elements || !insert_levels) PGV_FAIL(PGV_ERR_ARG, "elements/insert_levels is NULL");
    pgv_ctx *ctx = h->ctx;
    PGV_HIP(hipSetDevice(ctx->device));
    PGV_TRY(hnsw_graph_acquire(h));
    const size_t per = (size_t)nq * layer_cap;
    PGV_TRY(K.ids.ensure(sizeof(int32_t) * per * ef_construction));
    PGV_TRY(K.dist.ensure(sizeof(float) * per * ef_construction));
    PGV_TRY(K.cnt.ensure(sizeof(int32_t) * per));
    PGV_TRY(K.elems.ensure(sizeof(int32_t) * 2 * (size_t)nq));
    int32_t *e_dev = K.elems.as<int32_t>(), *l_dev = e_dev + nq;
    PGV_HIP(hipMemcpyAsync(e_dev, elements, sizeof(int32_t) * (size_t)nq, hipMemcpyDefault, ctx->stream));
    PGV_HIP(hipMemcpyAsync(l_dev, insert_levels, sizeof(int32_t) * (size_t)nq, hipMemcpyDefault, ctx->stream));
    PGV_TRY(hnsw_search_into(h, e_dev, l_dev, nq, ef_construction, layer_cap, K.ids.as<int32_t>(), K.dist.as<float>(),
                             K.cnt.as<int32_t>()));
    PGV_HIP(hipStreamSynchronize(ctx->stream));  // the searches are over when this returns: the tuples may be rewritten
    K.nq = nq;
    K.ef = ef_construction;
    K.lcap = layer_cap;
    return PGV_OK;
}

int pgv_hnsw_build_select_kept(pgv_hnsw *h, int slot, int32_t *out_ids, float *out_dist, uint8_t *out_closer, int32_t *out_count,
                               int64_t *out_pairs) {
    if (!h || !out_ids || !out_dist || !out_closer || !out_count)
        PGV_FAIL(PGV_ERR_ARG, "pgv_hnsw_build_select_kept: handle/out is NULL");
    // (no view refresh: the entry point may be moving under pgv_hnsw_link_apply on another thread, and nothing here looks
    // at the graph -- the kept lists, the element rows and m, which is fixed for the build)
    pgv_hnsw *o = h->view_of ? h->view_of : h;
    if (!o->link) PGV_FAIL(PGV_ERR_STATE, "pgv_hnsw_build_select_kept needs pgv_hnsw_link_begin");
    if (slot < 0 || slot > 1) PGV_FAIL(PGV_ERR_ARG, "slot must be 0 or 1");
    if (out_pairs) *out_pairs = 0;
    const HnswLinkState::Kept &K = o->link->kept[slot];
    if (K.nq == 0) return PGV_OK;
    PGV_HIP(hipSetDevice(h->ctx->device));
    h->m = o->m;
    const int32_t *l_dev = K.elems.as<int32_t>() + K.nq;
    return hnsw_select_from(h, K.ids.as<int32_t>(), K.dist.as<float>(), K.cnt.as<int32_t>(), l_dev, K.nq, K.ef, K.lcap, out_ids,
                            out_dist, out_closer, out_count, out_pairs);
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
    OutArg od;
    PGV_TRY(od.init(out, sizeof(float) * (size_t)npairs, ctx->out_stage));
    if (hnsw_pairs_by_gather()) {
        PGV_TRY(ctx->idx_stage.ensure(sizeof(int32_t) * (size_t)npairs));
        PGV_TRY(ctx->plan_d.ensure(sizeof(int32_t) * (size_t)npairs));
        int32_t *a_dev = ctx->idx_stage.as<int32_t>(), *b_dev = ctx->plan_d.as<int32_t>();
        PGV_TRY(launch_expand_groups(ctx, reinterpret_cast<const int32_t *>(tab),
                                     reinterpret_cast<const int64_t *>(tab + b_ids),
                                     reinterpret_cast<const int32_t *>(tab + b_ids + 2 * b_start),
                                     reinterpret_cast<const int64_t *>(tab + b_ids + b_start), ngroups, a_dev, b_dev));
        PGV_TRY(launch_score_gather(ctx, h->metric, h->dtype, h->geom, h->elements, h->elements, a_dev, b_dev, npairs,
                                    od.as<float>()));
    } else
        PGV_TRY(launch_score_groups(ctx, h->metric, h->dtype, h->geom, h->elements, reinterpret_cast<const int32_t *>(tab),
                                    reinterpret_cast<const int64_t *>(tab + b_ids), 0, nullptr,
                                    reinterpret_cast<const int32_t *>(tab + b_ids + 2 * b_start),
                                    reinterpret_cast<const int64_t *>(tab + b_ids + b_start), ngroups, od.as<float>()));
    bool need = true;  // the host tables above must have been read before the caller reuses them
    PGV_TRY(od.finish(ctx, &need));
    return sync_if(ctx, need);
}

// ------------------------------------------------------------------ the build's graph updates on the device
static void hnsw_link_free(pgv_hnsw *o) {
    HnswLinkState *L = o->link;
    if (!L) return;
    if (L->nb_dist) (void)hipFree(L->nb_dist);
    if (L->nb_flag) (void)hipFree(L->nb_flag);
    if (L->list_count) (void)hipFree(L->list_count);
    if (L->stats_dev) (void)hipFree(L->stats_dev);
    if (L->stats_host) (void)hipHostFree(L->stats_host);
    L->rec.release();
    L->links.release();
    L->ids.release();
    L->pa.release();
    L->pb.release();
    L->tri.release();
    L->mm.release();
    L->loc.release();
    L->sel.release();
    for (int i = 0; i < 2; i++) {
        L->kept[i].ids.release();
        L->kept[i].dist.release();
        L->kept[i].cnt.release();
        L->kept[i].elems.release();
    }
    delete L;
    o->link = nullptr;
}

int pgv_hnsw_link_begin(pgv_hnsw *h) {
    if (!h) PGV_FAIL(PGV_ERR_ARG, "pgv_hnsw_link_begin: handle is NULL");
    if (h->imported || h->view_of) PGV_FAIL(PGV_ERR_STATE, "pgv_hnsw_link_begin: an imported mirror / a view is read-only");
    if (h->m == 0) PGV_FAIL(PGV_ERR_ARG, "pgv_hnsw_link_begin needs pgv_hnsw_set_graph first");
    pgv_ctx *ctx = h->ctx;
    PGV_HIP(hipSetDevice(ctx->device));
    hnsw_link_free(h);
    HnswLinkState *L = new (std::nothrow) HnswLinkState();
    if (!L) PGV_FAIL(PGV_ERR_NOMEM, "out of host memory");
    const size_t total = (size_t)(h->nbr_total > 0 ? h->nbr_total : 1);
    L->nlists = total / (size_t)h->m + 1;  // every list starts at a multiple of m
    if (hipMalloc(&L->nb_dist, total * sizeof(float)) != hipSuccess || hipMalloc(&L->nb_flag, total) != hipSuccess ||
        hipMalloc(&L->list_count, 2 * L->nlists * sizeof(int)) != hipSuccess) {
        if (L->nb_dist) (void)hipFree(L->nb_dist);
        if (L->nb_flag) (void)hipFree(L->nb_flag);
        delete L;
        PGV_FAIL(PGV_ERR_NOMEM, "hipMalloc(%zu) for the hnsw build state failed", total * 5 + 2 * L->nlists * sizeof(int));
    }
    L->list_rec = L->list_count + L->nlists;
    if (hipMalloc(&L->stats_dev, 4 * sizeof(int64_t)) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void **>(&L->stats_host), 4 * sizeof(int64_t), hipHostMallocDefault) != hipSuccess) {
        h->link = L;
        hnsw_link_free(h);
        PGV_FAIL(PGV_ERR_NOMEM, "no room for the hnsw build's counters");
    }
    memset(L->stats_host, 0, 4 * sizeof(int64_t));
    PGV_HIP(hipMemsetAsync(L->stats_dev, 0, 4 * sizeof(int64_t), ctx->stream));
    h->link = L;
    PGV_HIP(hipMemsetAsync(L->nb_dist, 0, total * sizeof(float), ctx->stream));
    PGV_HIP(hipMemsetAsync(L->nb_flag, 0, total, ctx->stream));
    PGV_HIP(hipMemsetAsync(L->list_count, 0, 2 * L->nlists * sizeof(int), ctx->stream));
    PGV_HIP(hipStreamSynchronize(ctx->stream));
    return PGV_OK;
}

int pgv_hnsw_link_prepare(pgv_hnsw *h, const int32_t *elements, const uint8_t *linked, int nq, int layer_cap,
                          const int32_t *sel_ids, const float *sel_dist, const uint8_t *sel_closer, const int32_t *sel_count,
                          int64_t *out_pairs) {
    if (!h || !h->link) PGV_FAIL(PGV_ERR_STATE, "pgv_hnsw_link_prepare needs pgv_hnsw_link_begin");
    if (nq < 0 || layer_cap < 1) PGV_FAIL(PGV_ERR_ARG, "bad sizes");
    HnswLinkState *L = h->link;
    pgv_ctx *ctx = h->ctx;
    PGV_HIP(hipSetDevice(ctx->device));
    if (out_pairs) *out_pairs = 0;
    L->nrec = 0;
    L->npairs = 0;
    L->nq = nq;
    L->lcap = layer_cap;
    L->prepared = true;
    if (nq == 0) return PGV_OK;
    if (!elements || !linked || !sel_ids || !sel_dist || !sel_closer || !sel_count)
        PGV_FAIL(PGV_ERR_ARG, "the new elements' lists are NULL");
    const int m = h->m;
    const size_t per = (size_t)nq * layer_cap, stride = 2 * (size_t)m;
    if (per * stride > 0x7fffffff) PGV_FAIL(PGV_ERR_ARG, "pgv_hnsw_link_prepare: batch too large");
    // a searcher on another stream has left the tuples as the last patch / link made them
    PGV_TRY(hnsw_graph_acquire(h));
    // the batch's own lists (pgv_hnsw_build_neighbors' output), kept until _apply puts them in place
    {
        const size_t b_ids = sizeof(int32_t) * per * stride, b_dist = sizeof(float) * per * stride, b_cnt = sizeof(int32_t) * per,
                     b_el = sizeof(int32_t) * (size_t)nq, b_cl = (per * stride + 15) & ~(size_t)15, b_ln = ((size_t)nq + 15) & ~(size_t)15;
        PGV_TRY(L->sel.ensure(b_ids + b_dist + b_cnt + b_el + b_cl + b_ln));
        char *base = L->sel.as<char>();
        L->d_sel_ids = reinterpret_cast<int32_t *>(base);
        L->d_sel_dist = reinterpret_cast<float *>(base + b_ids);
        L->d_sel_cnt = reinterpret_cast<int32_t *>(base + b_ids + b_dist);
        L->d_elems = reinterpret_cast<int32_t *>(base + b_ids + b_dist + b_cnt);
        L->d_sel_closer = reinterpret_cast<uint8_t *>(base + b_ids + b_dist + b_cnt + b_el);
        L->d_linked = L->d_sel_closer + b_cl;
        PGV_HIP(hipMemcpyAsync(L->d_sel_ids, sel_ids, b_ids, hipMemcpyDefault, ctx->stream));
        PGV_HIP(hipMemcpyAsync(L->d_sel_dist, sel_dist, b_dist, hipMemcpyDefault, ctx->stream));
        PGV_HIP(hipMemcpyAsync(L->d_sel_cnt, sel_count, b_cnt, hipMemcpyDefault, ctx->stream));
        PGV_HIP(hipMemcpyAsync(L->d_elems, elements, b_el, hipMemcpyDefault, ctx->stream));
        PGV_HIP(hipMemcpyAsync(L->d_sel_closer, sel_closer, per * stride, hipMemcpyDefault, ctx->stream));
        PGV_HIP(hipMemcpyAsync(L->d_linked, linked, (size_t)nq, hipMemcpyDefault, ctx->stream));
    }
    // records: at most one per request
    const size_t cap = per * stride, n1 = cap + 1;
    const size_t b64 = sizeof(int64_t) * (5 * n1 + 4), b32 = sizeof(int32_t) * (7 * n1 + 8);
    PGV_TRY(L->rec.ensure(b64 + b32));
    char *base = L->rec.as<char>();
    L->rec_off = reinterpret_cast<int64_t *>(base);
    L->rec_pos = L->rec_off + n1;
    L->ids_start = L->rec_pos + n1;
    L->pair_start = L->ids_start + n1;
    L->mm_start = L->pair_start + n1;
    L->totals = L->mm_start + n1;
    L->rec_owner = reinterpret_cast<int32_t *>(base + b64);
    L->rec_lc = L->rec_owner + n1;
    L->rec_nstart = L->rec_lc + n1;
    L->rec_from = L->rec_nstart + n1;
    L->rec_wait = L->rec_from + n1;
    L->rec_list = L->rec_wait + n1;
    L->rec_fill = reinterpret_cast<int *>(L->rec_list + n1);
    L->blocked = L->rec_fill + n1;
    L->nrec_dev = L->blocked + 2;
    PGV_TRY(L->links.ensure((sizeof(int32_t) + sizeof(float)) * cap));
    L->d_link_elem = L->links.as<int32_t>();
    L->d_link_dist = reinterpret_cast<float *>(L->d_link_elem + cap);
    PGV_HIP(hipMemsetAsync(L->nrec_dev, 0, sizeof(int), ctx->stream));
    PGV_TRY(launch_hnsw_link_group(ctx, 0, L->d_elems, L->d_linked, nq, layer_cap, m, L->d_sel_ids, L->d_sel_dist, L->d_sel_cnt,
                                   h->levels, h->nbr_start, L->list_count, L->list_rec, L->nrec_dev, L->rec_owner, L->rec_lc,
                                   L->rec_list, nullptr, nullptr, nullptr, nullptr));
    int nrec = 0;
    PGV_HIP(hipMemcpyAsync(&nrec, L->nrec_dev, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    PGV_HIP(hipStreamSynchronize(ctx->stream));  // (also: the caller's arrays have been read)
    if (nrec < 0 || (size_t)nrec > cap) PGV_FAIL(PGV_ERR_STATE, "pgv_hnsw_link_prepare: %d records", nrec);
    L->nrec = nrec;
    if (nrec == 0) return PGV_OK;
    PGV_TRY(launch_hnsw_link_size(ctx, h->nbr, L->nb_flag, h->levels, h->nbr_start, m, L->rec_owner, L->rec_lc, L->rec_list,
                                  L->list_count, L->rec_off, nrec, 0, L->rec_pos, L->rec_nstart, L->rec_from, nullptr,
                                  L->ids_start, L->pair_start));
    PGV_TRY(launch_hnsw_link_scan(ctx, L->rec_off, L->ids_start, L->pair_start, nrec, L->totals));
    PGV_HIP(hipMemsetAsync(L->rec_fill, 0, sizeof(int) * (size_t)nrec, ctx->stream));
    PGV_TRY(launch_hnsw_link_group(ctx, 1, L->d_elems, L->d_linked, nq, layer_cap, m, L->d_sel_ids, L->d_sel_dist, L->d_sel_cnt,
                                   h->levels, h->nbr_start, L->list_count, L->list_rec, L->nrec_dev, L->rec_owner, L->rec_lc,
                                   L->rec_list, L->rec_off, L->rec_fill, L->d_link_elem, L->d_link_dist));
    int64_t totals[3] = {0, 0, 0};
    PGV_HIP(hipMemcpyAsync(totals, L->totals, sizeof(totals), hipMemcpyDeviceToHost, ctx->stream));
    PGV_HIP(hipStreamSynchronize(ctx->stream));
    const int64_t nlinks = totals[0], nids = totals[1], npairs = totals[2];
    const int64_t lm0 = 2 * (int64_t)m;
    if (nlinks < nrec || (size_t)nlinks > cap || nids < nlinks || nids > nlinks + (int64_t)nrec * lm0 || npairs < 0)
        PGV_FAIL(PGV_ERR_STATE, "pgv_hnsw_link_prepare: %lld links / %lld ids / %lld pairs planned for %d records",
                 (long long)nlinks, (long long)nids, (long long)npairs, nrec);
    PGV_TRY(L->ids.ensure(sizeof(int32_t) * (size_t)(nids > 0 ? nids : 1)));
    PGV_TRY(L->tri.ensure(sizeof(float) * (size_t)(npairs > 0 ? npairs : 1)));
    const bool gather = hnsw_pairs_by_gather();
    if (gather) {
        PGV_TRY(L->pa.ensure(sizeof(int32_t) * (size_t)(npairs > 0 ? npairs : 1)));
        PGV_TRY(L->pb.ensure(sizeof(int32_t) * (size_t)(npairs > 0 ? npairs : 1)));
    }
    // the id lists (and, for the gathered form, the slot pairs)
    PGV_TRY(launch_hnsw_link_pairs(ctx, h->nbr, L->rec_pos, L->rec_nstart, L->rec_from, L->rec_off, L->d_link_elem, L->d_link_dist,
                                   L->rec_list, L->list_count, nrec, 0, L->ids_start, L->ids.as<int32_t>(), L->pair_start,
                                   gather ? L->pa.as<int32_t>() : nullptr, gather ? L->pb.as<int32_t>() : nullptr));
    if (npairs > 0) {
        if (gather)
            PGV_TRY(launch_score_gather(ctx, h->metric, h->dtype, h->geom, h->elements, h->elements, L->pa.as<int32_t>(),
                                        L->pb.as<int32_t>(), npairs, L->tri.as<float>()));
        else
            PGV_TRY(launch_score_groups(ctx, h->metric, h->dtype, h->geom, h->elements, L->ids.as<int32_t>(), L->ids_start, 0,
                                        nullptr, L->rec_from, L->pair_start, nrec, L->tri.as<float>()));
    }
    L->npairs = npairs;
    if (out_pairs) *out_pairs = npairs;
    return PGV_OK;  // the scoring runs on; pgv_hnsw_link_apply is ordered behind it on the same stream
}

int pgv_hnsw_link_apply(pgv_hnsw *h, int32_t entry) {
    if (!h || !h->link || !h->link->prepared) PGV_FAIL(PGV_ERR_STATE, "pgv_hnsw_link_apply needs pgv_hnsw_link_prepare");
    if (entry < -1 || entry >= h->n) PGV_FAIL(PGV_ERR_ARG, "entry point %d out of range", (int)entry);
    HnswLinkState *L = h->link;
    pgv_ctx *ctx = h->ctx;
    PGV_HIP(hipSetDevice(ctx->device));
    L->prepared = false;
    const int nrec = L->nrec, m = h->m;
    if (nrec > 0) {
        PGV_TRY(L->loc.ensure(sizeof(int16_t) * (size_t)nrec * (2 * (size_t)m + 1)));
        PGV_HIP(hipMemsetAsync(L->blocked, 0, 2 * sizeof(int), ctx->stream));  // [0] stopped in the first round, [1] in the second
        PGV_TRY(launch_hnsw_link_replay(ctx, h->nbr, L->nb_dist, L->nb_flag, m, nrec, 0, L->rec_lc, L->rec_off, L->d_link_dist,
                                        L->rec_pos, L->rec_nstart, L->rec_from, L->ids_start, L->ids.as<int32_t>(), L->pair_start,
                                        L->tri.as<float>(), nullptr, nullptr, L->rec_wait, L->loc.as<int16_t>(), L->blocked));
        // The replays that stepped outside the pairs fetched for them: their lists' member-member triangles, then the rest
        // of their newcomers.  Everything is enqueued without waiting for a count: the triangles' room is the bound (every
        // record, a full list), launches cover every record and the ones that did not stop leave at once -- the host is
        // back before the first kernel has run, and the searches of the next batch but one start on the device the
        // moment the last one is through (hnsw_graph_acquire on their stream).
        const size_t lm0 = 2 * (size_t)m, mm_bound = (size_t)nrec * (lm0 * (lm0 - 1) / 2);
        const bool async = !hnsw_pairs_by_gather() && mm_bound * sizeof(float) <= ((size_t)512 << 20);
        PGV_TRY(launch_hnsw_link_size(ctx, h->nbr, L->nb_flag, h->levels, h->nbr_start, m, L->rec_owner, L->rec_lc, L->rec_list,
                                      L->list_count, L->rec_off, nrec, 1, L->rec_pos, L->rec_nstart, L->rec_from, L->rec_wait,
                                      nullptr, L->mm_start));
        PGV_TRY(launch_hnsw_link_scan(ctx, L->mm_start, nullptr, nullptr, nrec, L->totals));
        if (async) {
            PGV_TRY(L->mm.ensure(sizeof(float) * (mm_bound > 0 ? mm_bound : 1)));
            PGV_TRY(launch_score_groups(ctx, h->metric, h->dtype, h->geom, h->elements, L->ids.as<int32_t>(), L->ids_start, 0,
                                        L->rec_nstart, nullptr, L->mm_start, nrec, L->mm.as<float>()));
        } else {
            int64_t npairs2 = 0;
            PGV_HIP(hipMemcpyAsync(&npairs2, L->totals, sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
            PGV_HIP(hipStreamSynchronize(ctx->stream));
            if (npairs2 < 0 || (size_t)npairs2 > mm_bound)
                PGV_FAIL(PGV_ERR_STATE, "pgv_hnsw_link_apply: %lld member pairs planned for %d records", (long long)npairs2, nrec);
            PGV_TRY(L->mm.ensure(sizeof(float) * (size_t)(npairs2 > 0 ? npairs2 : 1)));
            if (npairs2 > 0 && hnsw_pairs_by_gather()) {
                PGV_TRY(L->pa.ensure(sizeof(int32_t) * (size_t)npairs2));
                PGV_TRY(L->pb.ensure(sizeof(int32_t) * (size_t)npairs2));
                PGV_TRY(launch_hnsw_link_pairs(ctx, h->nbr, L->rec_pos, L->rec_nstart, L->rec_from, L->rec_off, L->d_link_elem,
                                               L->d_link_dist, L->rec_list, L->list_count, nrec, 1, L->ids_start,
                                               L->ids.as<int32_t>(), L->mm_start, L->pa.as<int32_t>(), L->pb.as<int32_t>()));
                PGV_TRY(launch_score_gather(ctx, h->metric, h->dtype, h->geom, h->elements, h->elements, L->pa.as<int32_t>(),
                                            L->pb.as<int32_t>(), npairs2, L->mm.as<float>()));
            } else if (npairs2 > 0)
                PGV_TRY(launch_score_groups(ctx, h->metric, h->dtype, h->geom, h->elements, L->ids.as<int32_t>(), L->ids_start, 0,
                                            L->rec_nstart, nullptr, L->mm_start, nrec, L->mm.as<float>()));
        }
        PGV_TRY(launch_hnsw_link_replay(ctx, h->nbr, L->nb_dist, L->nb_flag, m, nrec, 1, L->rec_lc, L->rec_off, L->d_link_dist,
                                        L->rec_pos, L->rec_nstart, L->rec_from, L->ids_start, L->ids.as<int32_t>(), L->pair_start,
                                        L->tri.as<float>(), L->mm_start, L->mm.as<float>(), L->rec_wait, L->loc.as<int16_t>(),
                                        L->blocked + 1));
        // the counts: added up on the device, copied to pinned memory behind the launches (read at the end)
        PGV_TRY(launch_hnsw_link_stats(ctx, L->blocked, L->totals, L->stats_dev));
        PGV_HIP(hipMemcpyAsync(L->stats_host, L->stats_dev, 3 * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
    }
    // the batch's own elements (their lists were selected with their searches)
    if (L->nq > 0)
        PGV_TRY(launch_hnsw_link_new(ctx, h->nbr, L->nb_dist, L->nb_flag, h->levels, h->nbr_start, m, L->d_elems, L->d_linked,
                                     L->nq, L->lcap, L->d_sel_ids, L->d_sel_dist, L->d_sel_closer, L->d_sel_cnt));
    h->entry = entry;
    // searches on other streams (the helper's view) wait for this on the device
    if (!h->graph_ev) PGV_HIP(hipEventCreateWithFlags(&h->graph_ev, hipEventDisableTiming));
    PGV_HIP(hipEventRecord(h->graph_ev, ctx->stream));
    h->graph_ev_set = true;
    return PGV_OK;
}

int pgv_hnsw_link_end(pgv_hnsw *h, int32_t *out_nbr, int64_t *out_pairs, int64_t *out_deferred) {
    if (!h) PGV_FAIL(PGV_ERR_ARG, "pgv_hnsw_link_end: handle is NULL");
    if (out_pairs) *out_pairs = 0;
    if (out_deferred) *out_deferred = 0;
    if (!h->link) return PGV_OK;
    pgv_ctx *ctx = h->ctx;
    PGV_HIP(hipSetDevice(ctx->device));
    PGV_TRY(hnsw_graph_acquire(h));
    if (out_nbr && h->nbr_total > 0)
        PGV_HIP(hipMemcpyAsync(out_nbr, h->nbr, sizeof(int32_t) * (size_t)h->nbr_total, hipMemcpyDefault, ctx->stream));
    PGV_HIP(hipStreamSynchronize(ctx->stream));
    const int64_t deferred = h->link->stats_host[0], pairs2 = h->link->stats_host[1], still = h->link->stats_host[2];
    hnsw_link_free(h);
    if (still != 0) PGV_FAIL(PGV_ERR_STATE, "pgv_hnsw_link_end: %lld list updates were left waiting for distances", (long long)still);
    if (out_pairs) *out_pairs = pairs2;
    if (out_deferred) *out_deferred = deferred;
    return PGV_OK;
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
