// pgv_abi_comm.hip -- extern "C" entry points of libpgv_hip (include/pgv_hip.h): multi-GPU (pgv_comm_*, sharded k-means and search).
// Split out of pgv_abi.hip in round 5 (one unit per area, so that an edit recompiles one of them).
#include "pgv_abi_common.h"

extern "C" {

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

}  // extern "C"
