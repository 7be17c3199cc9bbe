// pgv_abi_build.hip -- extern "C" entry points of libpgv_hip (include/pgv_hip.h): assignment, distance batches, the exact scan, k-means.
// Split out of pgv_abi.hip in round 5 (one unit per area, so that an edit recompiles one of them).
#include "pgv_abi_common.h"

extern "C" {

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
int lloyd_partial_dev(pgv_ctx *ctx, pgv_ops ops, pgv_dtype dtype, const RowGeom &g,
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
int lloyd_finish_dev(pgv_ctx *ctx, pgv_ops ops, pgv_dtype dtype, const RowGeom &g, int dim, int k,
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

int check_centers_dev(pgv_ctx *ctx, pgv_ops ops, pgv_dtype dtype, const RowGeom &g, int dim, int k,
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

}  // extern "C"
