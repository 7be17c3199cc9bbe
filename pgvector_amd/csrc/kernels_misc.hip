// kernels_misc.hip -- the distance operators that no index scan needs but the executor
// does, one query against n contiguous rows (the exact-scan shape, one fmgr call per heap row
// in the reference):
//   cosine_distance   src/vector.c:649-697 / src/halfvec.c:652-700 (VectorCosineSimilarity's three
//                     fp32 accumulators, the division and the clamp in double)
//   hamming_distance, jaccard_distance   src/bitvec.c:45-70 over src/bitutils.c:49-131
// Both are plain HBM streaming: rows are read once with 16-byte loads, the query stays in L2.
#include "pgv_device.h"

namespace pgv {

namespace {

constexpr int kMiscThreads = 256;
constexpr int kMiscWaves = kMiscThreads / kWave;

// lanes cooperating on a row like scan_kernel: (1 << lpr_log2) adjacent lanes, nchunks trips
template <typename T>
__global__ __launch_bounds__(kMiscThreads) void cosine_kernel(const char *__restrict__ rows, const char *__restrict__ query,
                                                              int64_t n, double *__restrict__ out, int nvec,
                                                              int lpr_log2, int nchunks) {
    constexpr int N = VecTraits<T>::N;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    const int lpr = 1 << lpr_log2;
    const int sub = lane & (lpr - 1);
    const int rsub = lane >> lpr_log2;
    const int rpw = kWave >> lpr_log2;
    const size_t row_bytes = (size_t)nvec * sizeof(Raw16);
    const int64_t i = ((int64_t)blockIdx.x * kMiscWaves + wave) * rpw + rsub;
    const bool valid = i < n;
    const char *rp = rows + (size_t)(valid ? i : n - 1) * row_bytes;
    // similarity += ax[i] * bx[i]; norma += ax[i] * ax[i]; normb += bx[i] * bx[i]  (vector.c:656-662)
    float sim = 0.f, na = 0.f, nb = 0.f;
    for (int c = 0; c < nchunks; c++) {
        const int vi = c * lpr + sub;
        const bool ok = vi < nvec;
        const int vc = ok ? vi : nvec - 1;  // never predicate a load (see scan_kernel)
        const Raw16 rv = load16(rp + (size_t)vc * sizeof(Raw16));
        const Raw16 qv = load16(query + (size_t)vc * sizeof(Raw16));
        Unpacked<T> ur(rv);
        Unpacked<T> uq(qv);
#pragma unroll
        for (int e = 0; e < N; e++) {
            const float a = ok ? ur.v[e] : 0.f, b = ok ? uq.v[e] : 0.f;
            sim = __builtin_fmaf(a, b, sim);
            na = __builtin_fmaf(a, a, na);
            nb = __builtin_fmaf(b, b, nb);
        }
    }
    sim = group_sum_to_last(sim, lpr_log2);
    na = group_sum_to_last(na, lpr_log2);
    nb = group_sum_to_last(nb, lpr_log2);
    if (sub == lpr - 1 && valid) {
        // Use sqrt(a * b) over sqrt(a) * sqrt(b) (vector.c:664-665), then cosine_distance's clamp (:688-694)
        double similarity = (double)sim / sqrt((double)na * (double)nb);
        if (similarity > 1.0)
            similarity = 1.0;
        else if (similarity < -1.0)
            similarity = -1.0;
        out[i] = 1.0 - similarity;  // NaN (a zero vector: 0 / 0) passes through both comparisons
    }
}

// bit rows padded with zero bytes to whole 16-byte vectors: padding adds nothing to any count.
// MODE 0: popcount(a ^ b); MODE 1: 1 - |a & b| / (|a| + |b| - |a & b|), 1 when the intersection is empty.
template <int MODE>
__global__ __launch_bounds__(kMiscThreads) void bit_kernel(const char *__restrict__ rows, const char *__restrict__ query,
                                                           int64_t n, double *__restrict__ out, int nvec, int lpr_log2,
                                                           int nchunks) {
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    const int lpr = 1 << lpr_log2;
    const int sub = lane & (lpr - 1);
    const int rsub = lane >> lpr_log2;
    const int rpw = kWave >> lpr_log2;
    const size_t row_bytes = (size_t)nvec * sizeof(Raw16);
    const int64_t i = ((int64_t)blockIdx.x * kMiscWaves + wave) * rpw + rsub;
    const bool valid = i < n;
    const char *rp = rows + (size_t)(valid ? i : n - 1) * row_bytes;
    // counts stay far below 2^24 (at most 64000 bits, src/ivfutils.c:416), so fp32 sums of them are exact
    float c0 = 0.f, c1 = 0.f, c2 = 0.f;
    for (int c = 0; c < nchunks; c++) {
        const int vi = c * lpr + sub;
        const bool ok = vi < nvec;
        const int vc = ok ? vi : nvec - 1;
        const Raw16 rv = load16(rp + (size_t)vc * sizeof(Raw16));
        const Raw16 qv = load16(query + (size_t)vc * sizeof(Raw16));
        int x = 0, y = 0, z = 0;
#pragma unroll
        for (int w = 0; w < 4; w++) {
            if (MODE == 0) {
                x += __popc(rv.w[w] ^ qv.w[w]);
            } else {
                x += __popc(rv.w[w] & qv.w[w]);
                y += __popc(rv.w[w]);
                z += __popc(qv.w[w]);
            }
        }
        if (ok) {
            c0 += (float)x;
            c1 += (float)y;
            c2 += (float)z;
        }
    }
    c0 = group_sum_to_last(c0, lpr_log2);
    if (MODE == 1) {
        c1 = group_sum_to_last(c1, lpr_log2);
        c2 = group_sum_to_last(c2, lpr_log2);
    }
    if (sub == lpr - 1 && valid) {
        if (MODE == 0)
            out[i] = (double)c0;
        else
            out[i] = c0 == 0.f ? 1.0 : 1.0 - (double)c0 / ((double)c1 + (double)c2 - (double)c0);
    }
}

}  // namespace

int launch_cosine(pgv_ctx *ctx, pgv_dtype dtype, const RowGeom &g, const void *rows, const void *query, int64_t n,
                  double *out) {
    if (n <= 0) return PGV_OK;
    const int rpw = kWave >> g.lpr_log2;
    const int64_t per_block = (int64_t)kMiscWaves * rpw;
    const unsigned grid = (unsigned)((n + per_block - 1) / per_block);
    if (dtype == PGV_F32)
        hipLaunchKernelGGL(cosine_kernel<float>, dim3(grid), dim3(kMiscThreads), 0, ctx->stream,
                           static_cast<const char *>(rows), static_cast<const char *>(query), n, out, g.nvec,
                           g.lpr_log2, g.nchunks);
    else
        hipLaunchKernelGGL(cosine_kernel<__half>, dim3(grid), dim3(kMiscThreads), 0, ctx->stream,
                           static_cast<const char *>(rows), static_cast<const char *>(query), n, out, g.nvec,
                           g.lpr_log2, g.nchunks);
    PGV_HIP(hipGetLastError());
    return PGV_OK;
}

int launch_bit_distance(pgv_ctx *ctx, int mode, const RowGeom &g, const void *rows, const void *query, int64_t n,
                        double *out) {
    if (n <= 0) return PGV_OK;
    const int rpw = kWave >> g.lpr_log2;
    const int64_t per_block = (int64_t)kMiscWaves * rpw;
    const unsigned grid = (unsigned)((n + per_block - 1) / per_block);
    if (mode == 0)
        hipLaunchKernelGGL(bit_kernel<0>, dim3(grid), dim3(kMiscThreads), 0, ctx->stream, static_cast<const char *>(rows),
                           static_cast<const char *>(query), n, out, g.nvec, g.lpr_log2, g.nchunks);
    else
        hipLaunchKernelGGL(bit_kernel<1>, dim3(grid), dim3(kMiscThreads), 0, ctx->stream, static_cast<const char *>(rows),
                           static_cast<const char *>(query), n, out, g.nvec, g.lpr_log2, g.nchunks);
    PGV_HIP(hipGetLastError());
    return PGV_OK;
}

}  // namespace pgv
