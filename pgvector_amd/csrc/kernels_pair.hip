// kernels_pair.hip -- rows x centers, both large: nearest center per row.
//
// The argmin loop of AddTupleToSort (src/ivfbuild.c:183-192) for a batch of
// heap rows and the assignment half of a k-means iteration
// (src/ivfkmeans.c:391-451 without Elkan's pruning).  n*k*dim work: the one
// compute-bound piece of the path.  This version keeps the reference's exact
// per-pair form sum((a-b)^2) / sum(a*b) on the fp32 vector ALUs with a classic
// LDS-tiled register blocking (128 rows x 128 centers per workgroup, 8x8 per
// lane) -- an MFMA formulation (|a|^2 - 2ab + |b|^2) changes rounding
// (SURVEY hard part 4) and is left for a later, separately verified variant.
//
// Selection rule: first strictly-smaller distance wins scanning centers in
// ascending id (src/ivfbuild.c:187-191); a distance that is NaN or +inf is
// never selected (it is not < DBL_MAX), leaving list 0 like the reference.
#include "pgv_device.h"

#include <cfloat>

namespace pgv {

namespace {

constexpr int BM = 128;  // rows per workgroup
constexpr int BN = 128;  // centers per tile
constexpr int BK = 16;   // dimension slice staged in LDS
constexpr int PAD = 4;
constexpr int kPairThreads = 256;
constexpr int kListedChunk = 256;  // centers per workgroup when only listed rows are redone

// MODE 0: L2 squared   1: negative inner product   2: L1
// MODE 3: spherical k-means = -clamp(ip, -1, 1): same argmin as acos(ip)/pi
//         (src/vector.c:703-722) including the ties its clamp creates
template <int MODE> __device__ __forceinline__ float pair_accum(float acc, float a, float b) {
    if constexpr (MODE == 0) {
        float d = a - b;
        return fmaf(d, d, acc);
    } else if constexpr (MODE == 2) {
        return acc + fabsf(a - b);
    } else {
        return fmaf(a, b, acc);
    }
}
template <int MODE> __device__ __forceinline__ float pair_finish(float acc) {
    if constexpr (MODE == 1) return -acc;
    if constexpr (MODE == 3) {
        float v = acc;
        if (v > 1.f)
            v = 1.f;
        else if (v < -1.f)
            v = -1.f;
        return -v;
    }
    return acc;
}

template <typename T, int MODE>
__global__ __launch_bounds__(kPairThreads) void argmin_kernel(
    const char *__restrict__ rows, int64_t n, const char *__restrict__ centers, int k, int nvec,
    int32_t *__restrict__ out_idx, float *__restrict__ out_val, const int32_t *__restrict__ row_list,
    const int *__restrict__ row_count, unsigned long long *__restrict__ packed) {
    // row_list != nullptr: only the *row_count rows it names (the rows the MFMA pre-filter of
    // kernels_mfma.hip could not decide).  The grid is sized for the worst case, and because the
    // list is usually short the centers are split over blockIdx.y (kListedChunk each): every
    // workgroup folds its (distance key, center id) minimum into packed[row] with a 64-bit
    // atomicMin -- the same lexicographic "first strictly smaller wins".
    if (row_list) n = *row_count;
    if ((int64_t)blockIdx.x * BM >= n) return;
    const int c_begin = row_list ? (int)blockIdx.y * kListedChunk : 0;
    const int c_end = row_list ? (c_begin + kListedChunk < k ? c_begin + kListedChunk : k) : k;
    constexpr int N = VecTraits<T>::N;            // elements per 16-byte vector
    constexpr int VPT = BK / N;                   // vectors per row per k-slice (4 or 2)
    constexpr int LOADS = BM * VPT / kPairThreads; // 16-byte loads per thread per operand (2 or 1)
    __shared__ float As[BK][BM + PAD];
    __shared__ float Bs[BK][BN + PAD];

    const int tid = threadIdx.x;
    const int tx = tid & 15;  // center sub-tile
    const int ty = tid >> 4;  // row sub-tile
    const int64_t row_base = (int64_t)blockIdx.x * BM;
    const size_t row_bytes = (size_t)nvec * sizeof(Raw16);
    const int ld = nvec * N;

    float best_val[8];
    int best_idx[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        best_val[i] = INFINITY;
        best_idx[i] = 0;
    }

    for (int cb = c_begin; cb < c_end; cb += BN) {
        float acc[8][8];
#pragma unroll
        for (int i = 0; i < 8; i++)
#pragma unroll
            for (int j = 0; j < 8; j++) acc[i][j] = 0.f;

        for (int kb = 0; kb < ld; kb += BK) {
            // global -> LDS, transposed to [k][row]; rows/centers past the end are clamped
#pragma unroll
            for (int l = 0; l < LOADS; l++) {
                const int idx = tid + l * kPairThreads;
                const int r = idx / VPT;
                const int v = idx % VPT;
                const int vi = kb / N + v;
                Raw16 ra = raw16_zero(), rb = raw16_zero();
                if (vi < nvec) {
                    int64_t ar = row_base + r;
                    ar = ar < n ? ar : n - 1;
                    if (row_list) ar = row_list[ar];
                    int br = cb + r;
                    br = br < k ? br : k - 1;
                    ra = load16(rows + (size_t)ar * row_bytes + (size_t)vi * sizeof(Raw16));
                    rb = load16(centers + (size_t)br * row_bytes + (size_t)vi * sizeof(Raw16));
                }
                Unpacked<T> ua(ra), ub(rb);
#pragma unroll
                for (int e = 0; e < N; e++) {
                    As[v * N + e][r] = ua.v[e];
                    Bs[v * N + e][r] = ub.v[e];
                }
            }
            __syncthreads();
#pragma unroll
            for (int kk = 0; kk < BK; kk++) {
                float a[8], b[8];
                const float4 a0 = *reinterpret_cast<const float4 *>(&As[kk][ty * 8]);
                const float4 a1 = *reinterpret_cast<const float4 *>(&As[kk][ty * 8 + 4]);
                const float4 b0 = *reinterpret_cast<const float4 *>(&Bs[kk][tx * 8]);
                const float4 b1 = *reinterpret_cast<const float4 *>(&Bs[kk][tx * 8 + 4]);
                a[0] = a0.x; a[1] = a0.y; a[2] = a0.z; a[3] = a0.w;
                a[4] = a1.x; a[5] = a1.y; a[6] = a1.z; a[7] = a1.w;
                b[0] = b0.x; b[1] = b0.y; b[2] = b0.z; b[3] = b0.w;
                b[4] = b1.x; b[5] = b1.y; b[6] = b1.z; b[7] = b1.w;
#pragma unroll
                for (int i = 0; i < 8; i++)
#pragma unroll
                    for (int j = 0; j < 8; j++) acc[i][j] = pair_accum<MODE>(acc[i][j], a[i], b[j]);
            }
            __syncthreads();
        }

        // fold this tile of centers into the running (value, id) minimum
#pragma unroll
        for (int i = 0; i < 8; i++) {
            float v = INFINITY;
            int id = 0x7fffffff;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int c = cb + tx * 8 + j;
                const float d = pair_finish<MODE>(acc[i][j]);
                if (c < k && d < v) {
                    v = d;
                    id = c;
                }
            }
            // lexicographic (value, id) min over the 16 lanes that share these rows
#pragma unroll
            for (int m = 1; m < 16; m <<= 1) {
                const float ov = __shfl_xor(v, m, 16);
                const int oid = __shfl_xor(id, m, 16);
                if (ov < v || (ov == v && oid < id)) {
                    v = ov;
                    id = oid;
                }
            }
            if (v < best_val[i]) {  // strict: an equal distance in a later tile loses
                best_val[i] = v;
                best_idx[i] = id;
            }
        }
    }

    if (tx == 0) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int64_t r = row_base + ty * 8 + i;
            if (r < n) {
                if (row_list) {
                    if (best_val[i] < INFINITY)
                        atomicMin(&packed[row_list[r]],
                                  ((unsigned long long)float_to_key(best_val[i]) << 32) | (unsigned)best_idx[i]);
                } else {
                    out_idx[r] = best_idx[i];
                    if (out_val) out_val[r] = best_val[i] == INFINITY ? FLT_MAX : best_val[i];
                }
            }
        }
    }
}

template <typename T, int MODE>
int launch_argmin_t(pgv_ctx *ctx, const RowGeom &g, const void *rows, int64_t n,
                    const void *centers, int k, int32_t *out_idx, float *out_val,
                    const int32_t *row_list = nullptr, const int *row_count = nullptr,
                    unsigned long long *packed = nullptr) {
    if (n <= 0) return PGV_OK;
    const int64_t grid = (n + BM - 1) / BM;
    const int chunks = row_list ? (k + kListedChunk - 1) / kListedChunk : 1;
    hipLaunchKernelGGL((argmin_kernel<T, MODE>), dim3((unsigned)grid, (unsigned)chunks), dim3(kPairThreads), 0,
                       ctx->stream, static_cast<const char *>(rows), n,
                       static_cast<const char *>(centers), k, g.nvec, out_idx, out_val, row_list, row_count, packed);
    PGV_HIP(hipGetLastError());
    return PGV_OK;
}

}  // namespace

// mode: 0..2 = pgv_metric, 3 = spherical k-means
int launch_argmin_mode(pgv_ctx *ctx, int mode, pgv_dtype dtype, const RowGeom &g,
                       const void *rows, int64_t n, const void *centers, int k,
                       int32_t *out_idx, float *out_val) {
    if (k <= 0) PGV_FAIL(PGV_ERR_ARG, "argmin: need at least one center");
    // the dense contractions go to the matrix cores (kernels_mfma.hip); L1 and small problems stay here
    if (mfma_argmin_supported(mode, n, k)) return launch_argmin_mfma(ctx, mode, dtype, g, rows, n, centers, k, out_idx, out_val);
#define PGV_ARGMIN(T)                                                                     \
    switch (mode) {                                                                       \
        case 0:                                                                           \
            return launch_argmin_t<T, 0>(ctx, g, rows, n, centers, k, out_idx, out_val);  \
        case 1:                                                                           \
            return launch_argmin_t<T, 1>(ctx, g, rows, n, centers, k, out_idx, out_val);  \
        case 2:                                                                           \
            return launch_argmin_t<T, 2>(ctx, g, rows, n, centers, k, out_idx, out_val);  \
        case 3:                                                                           \
            return launch_argmin_t<T, 3>(ctx, g, rows, n, centers, k, out_idx, out_val);  \
    }
    if (dtype == PGV_F32) {
        PGV_ARGMIN(float)
    } else {
        PGV_ARGMIN(__half)
    }
#undef PGV_ARGMIN
    PGV_FAIL(PGV_ERR_ARG, "argmin: unknown mode %d", mode);
}

// the rows named by row_list[0 .. *row_count) only (device memory; n bounds *row_count): the
// lexicographic (distance key, center id) minimum of each is folded into packed[row], which the
// caller preset to ~0
int launch_argmin_listed(pgv_ctx *ctx, int mode, pgv_dtype dtype, const RowGeom &g, const void *rows, int64_t n,
                         const void *centers, int k, const int32_t *row_list, const int *row_count,
                         unsigned long long *packed) {
    if (mode != 0) PGV_FAIL(PGV_ERR_ARG, "listed argmin: L2 only");
    if (dtype == PGV_F32)
        return launch_argmin_t<float, 0>(ctx, g, rows, n, centers, k, nullptr, nullptr, row_list, row_count, packed);
    return launch_argmin_t<__half, 0>(ctx, g, rows, n, centers, k, nullptr, nullptr, row_list, row_count, packed);
}

int launch_argmin(pgv_ctx *ctx, pgv_metric metric, pgv_dtype dtype, const RowGeom &g,
                  const void *rows, int64_t n, const void *centers, int k, int32_t *out_idx,
                  float *out_val) {
    return launch_argmin_mode(ctx, (int)metric, dtype, g, rows, n, centers, k, out_idx, out_val);
}

}  // namespace pgv
