// pgv_abi_ivf.hip -- extern "C" entry points of libpgv_hip (include/pgv_hip.h): the IVFFlat mirror, the build's tuplesort on the device, list scans, one query at a time.
// Split out of pgv_abi.hip in round 5 (one unit per area, so that an edit recompiles one of them).
#include "pgv_abi_common.h"

extern "C" {

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
    if (malloc_exportable(&ix->arena, lay.bytes) != hipSuccess) {
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


// device-side core of GetScanLists for nq staged queries
int rank_lists_dev(pgv_index *ix, const void *q_dev, int nq, int maxprobes,
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

int scan_batch_dev(pgv_index *ix, const void *q_dev, int nq, const int32_t *probe_lists, int probes,
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
    // lists probed by more than 32 queries of the batch are streamed once per group of `qt` queries: from ~12 queries per
    // list on average (configs[2] / [4]: 1024 x 64 probes over 4096 lists = 16) enough lists pass 32 for the 64-query
    // form of the kernel to pay (PGV_SCAN_WIDE = 0 / 1 forces it off / on: A/B)
    static const int wide_env = [] {
        const char *e = getenv("PGV_SCAN_WIDE");
        return e ? atoi(e) : -1;
    }();
    // measured (10 M rows, lists 4096, probes 64, 1024 queries): 3072-d fp16 88.5 k -> 93.3 k QPS (scan 10.81 -> 10.19 ms); 1536-d
    // fp32: HBM traffic -16 % (passes 1.26 -> 1.06) but the scan only 11.83 -> 11.43 ms, 81.7 k -> 84.4 k QPS (round 6; one GPU's
    // share 612 k -> 640 k) -- a task of two fp32 tiles per stage fill runs the matrix pipes at ~90 %, so the 64-query form
    // buys a few per cent there, not the traffic's 16; below ~12 queries per list it costs (headline, share 10: -3 %)
    const bool wide = use_mfma && (wide_env < 0 ? share > 12.0 : wide_env != 0);
    const int qt = use_mfma ? (wide ? mfma_scan_queries_per_task_wide() : mfma_scan_queries_per_task())
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
                                     seg_vals, rows_stream_past_caches(ix->geom, ix->dtype, ix->nrows), qt));
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
        // (the 64-query form of the scan keeps its four chains as consecutive quarters of the row: whole 128-byte slices)
        const ScanBound gamma = wide ? scan_bound_chain(ctx, ix->geom.ld, dense_chain_length(ix->geom, ix->dtype))
                                     : scan_bound(ctx, ix->geom.ld);
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

int check_batch_args(pgv_index *ix, const void *queries, int nq, int probes, int k, float *out_dist,
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

// PGV_QUERY_DIRECT=0 keeps the staging kernel (A/B switch)
static bool query_direct_enabled() {
    static const bool on = [] {
        const char *e = getenv("PGV_QUERY_DIRECT");
        return !(e && e[0] == '0');
    }();
    return on;
}

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
    int large_bar = 0;
    if (rc == PGV_OK && query_direct_enabled() &&
        hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, ctx->device) == hipSuccess && large_bar) {
        // a device row the host can store into (the whole of HBM is behind the PCIe BAR): the query then needs no
        // staging kernel.  Optional: without it the pinned row + query_stage_kernel carry the query
        if (hipExtMallocWithFlags(&q->q_direct, row_bytes, hipDeviceMallocFinegrained) != hipSuccess) {
            (void)hipGetLastError();
            q->q_direct = nullptr;
        }
    }
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
    if (q->q_direct) (void)hipFree(q->q_direct);
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
        q->q_row = q->q_dev.p;
    } else {
        // the previous query's kernels have finished reading q_pinned: every pgv_query_scan waits for its head
        // (a rank that no scan followed is waited for here)
        if (q->rank_pending) PGV_HIP(hipStreamSynchronize(ctx->stream));
        void *dst = q->q_direct ? q->q_direct : q->q_pinned;
        memcpy(dst, query, (size_t)ix->dim * es);
        if (row_bytes > (size_t)ix->dim * es) memset(static_cast<char *>(dst) + (size_t)ix->dim * es, 0, row_bytes - (size_t)ix->dim * es);
        if (q->q_direct) {
            // the stores drain to the device ahead of the doorbell write of the launch below (posted writes, in order)
            __builtin_ia32_sfence();
            q->q_row = q->q_direct;
        } else {
            PGV_TRY(launch_query_stage(ctx, q->q_pinned, q->q_dev.p, ix->geom.nvec));
            q->q_row = q->q_dev.p;
        }
    }
    q->rank_pending = true;
    return launch_query_rank(ctx, ix, q->q_row, q->cdist, max_probes, q->lists);
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
    PGV_TRY(launch_query_scan(ctx, ix, q->is_null ? nullptr : q->q_row, q->lists + first, nprobes, bound,
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

}  // extern "C"
