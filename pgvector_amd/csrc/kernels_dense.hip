// kernels_dense.hip -- every query x every row on the matrix cores, 128 x 128 per workgroup: pgv_exact_topk's scan.
//
// The index-less `ORDER BY v <-> q LIMIT k` for a batch of queries (the reference calls l2_distance once per
// row and query, src/vector.c:579-589) is a dense contraction: nq x n x dim multiply-adds.  Until round 5 it ran on the
// list scan's kernel (mfma_scan_kernel: 32 queries x 128 rows per workgroup, one 32 x 32 tile per wavefront), which
// streams the rows once per 32 queries and multiplies for ~0.5 us per 20 KB stage fill: 1 M x 1536 x 1024 queries took
// 42 ms (75 TFLOP/s, 48 % of the fp32 MFMA peak), bound by the fill's round trip.  Here a workgroup owns 128 queries x
// 128 rows (the assignment kernel's tile, kernels_mfma.hip): four wavefronts of 2 x 2 tiles of 32 x 32, 4 x the
// multiplies per stage, rows streamed once per 128 queries -- and the workgroups that share a row tile (one per query
// tile) sit next to each other on ONE XCD, so the tile comes from HBM once and from that XCD's L2 after.
//
//   METRIC 1 (negative inner product): the value IS the reference's arithmetic in a different summation order.
//   METRIC 0 (L2): |x|^2 - 2 q.x with the rows' norms given -- the expansion less its per-query constant, an
//   approximation that only picks candidates; pgv_exact_topk re-evaluates sum((q - x)^2) for the k' best and proves that
//   nothing outside them can matter (batch_recheck_kernel / batch_fix_kernel), with the rounding bound of THIS kernel:
//   a dot product is accumulated in FOUR chains (the slices of a row in four consecutive quarters, each summed by the
//   matrix pipeline on its own, the quarters added at the end), so each chain holds at most ceil(slices / 4) x 32 (fp32)
//   products -- dense_chain_length() -- and the deterministic bound stays as narrow as the list scan's (pgv_internal.h).
//
// Staging, swizzle and operand reads are the assignment kernel's: 128-byte slices of 128 + 128 rows brought into LDS by
// the global->LDS DMA, slot p of row i holding 16-byte vector p ^ ((i >> 1) & 7), next slice in flight under the
// current one's MFMAs.
#include "pgv_device.h"
#include "pgv_internal.h"

namespace pgv {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int kDenseSlice = 128;            // bytes of a row per stage
constexpr int kDenseTile = 128;             // queries and rows per workgroup
constexpr int kDenseStage = 2 * kDenseTile * kDenseSlice;  // 32 KB: [128 query rows | 128 data rows] x 128 B

template <typename T> struct DenseMma;
template <> struct DenseMma<float> {
    static __device__ __forceinline__ void run(f32x16 &acc, const u32x4 &a, const u32x4 &b) {
#pragma unroll
        for (int e = 0; e < 4; e++)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a[e]), __uint_as_float(b[e]), acc, 0, 0, 0);
    }
};
template <> struct DenseMma<__half> {
    static __device__ __forceinline__ void run(f32x16 &acc, const u32x4 &a, const u32x4 &b) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0,
                                                     0, 0);
    }
};

// out[q * out_stride + r] for q in [0, nq), r in [0, n).  Grid: 8 * ceil(row_tiles / 8) * query_tiles workgroups; ids go
// round the 8 XCDs, so XCD x runs ids x, x + 8, ...: its s-th workgroup takes query tile s % qtiles of row tile
// x + 8 * (s / qtiles) -- the qtiles workgroups of one row tile are consecutive on one XCD.
template <typename T, int METRIC>
__global__ __launch_bounds__(256, 2) void mfma_dense_kernel(const char *__restrict__ rows, int64_t n,
                                                             const char *__restrict__ queries, int nq, int nvec,
                                                             const float *__restrict__ row_norms,
                                                             const char *__restrict__ zeros16, float *__restrict__ out,
                                                             int64_t out_stride, int qtiles, int quarter) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;  // 2 x 2 wavefronts, 64 queries x 64 rows each
    const int l31 = lane & 31, half = lane >> 5;
    const unsigned sw = (unsigned)(l31 >> 1) & 7u;

    const int64_t w = blockIdx.x, x = w & 7, sq = w >> 3;
    const int qt = (int)(sq % qtiles);
    const int64_t row_tile = x + 8 * (sq / qtiles);
    const int64_t row_base = row_tile * kDenseTile;
    if (row_base >= n) return;  // the padded tail of the XCD round-robin
    const int q_base = qt * kDenseTile;
    const size_t row_bytes = (size_t)nvec * sizeof(Raw16);
    const int nslices = (nvec + 7) / 8;

    // DMA role: this wavefront fills 64 consecutive rows of the concatenation [128 query rows | 128 data rows], 8 per
    // instruction; a lane brings slot (lane & 7) of row (lane >> 3) of its group.  Rows past the end repeat the last one.
    const int crow0 = wave * 64;
    const bool fills_q = crow0 < kDenseTile;
    const int drow = lane >> 3, dpos = lane & 7;
    const char *src_base = fills_q ? queries + (size_t)q_base * row_bytes : rows + (size_t)row_base * row_bytes;
    const int64_t src_limit = fills_q ? (int64_t)(nq - q_base < kDenseTile ? nq - q_base : kDenseTile)
                                      : (n - row_base < kDenseTile ? n - row_base : kDenseTile);
    const char *src[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int64_t i = (fills_q ? crow0 : crow0 - kDenseTile) + 8 * j + drow;
        src[j] = src_base + (size_t)(i < src_limit ? i : src_limit - 1) * row_bytes;
    }
    auto issue_stage = [&](int sl, int buf) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int v = dpos ^ ((4 * j + (drow >> 1)) & 7);  // slot p of row i holds vector p ^ ((i >> 1) & 7)
            const int vi = sl * 8 + v;
            const char *p = vi < nvec ? src[j] + (size_t)vi * sizeof(Raw16) : zeros16;
            char *dst = smem + (size_t)buf * kDenseStage + (size_t)(crow0 + 8 * j) * kDenseSlice;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)p,
                                             (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
        }
    };

    f32x16 acc[2][2], sum[2][2];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                acc[a][b][r] = 0.f;
                sum[a][b][r] = 0.f;
            }

    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;
    const unsigned a_lane = (unsigned)(wm * 64 + l31) * kDenseSlice;
    const unsigned b_lane = (unsigned)(kDenseTile + wn * 64 + l31) * kDenseSlice;

    issue_stage(0, 0);
    int in_quarter = 0;
    for (int sl = 0; sl < nslices; sl++) {
        // the slice has landed for every wavefront and nobody still reads the other buffer (the wait is spelled out:
        // __syncthreads() alone does not drain an LDS-DMA)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (sl + 1 < nslices) issue_stage(sl + 1, (sl + 1) & 1);
        const unsigned sbase = lds0 + (unsigned)(sl & 1) * kDenseStage;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const unsigned xo = (((unsigned)(2 * c + half)) ^ sw) << 4;
            u32x4 a[2], b[2];
            asm volatile(
                "ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:4096\n\t"
                "ds_read_b128 %2, %5\n\tds_read_b128 %3, %5 offset:4096\n\t"
                "s_waitcnt lgkmcnt(0)"
                : "=&v"(a[0]), "=&v"(a[1]), "=&v"(b[0]), "=&v"(b[1])
                : "v"(sbase + a_lane + xo), "v"(sbase + b_lane + xo)
                : "memory");
#pragma unroll
            for (int tm = 0; tm < 2; tm++)
#pragma unroll
                for (int tn = 0; tn < 2; tn++) DenseMma<T>::run(acc[tm][tn], a[tm], b[tn]);
        }
        // a quarter of the row is one accumulator chain: closed here, the next one starts from zero
        if (++in_quarter == quarter) {
            in_quarter = 0;
#pragma unroll
            for (int tm = 0; tm < 2; tm++)
#pragma unroll
                for (int tn = 0; tn < 2; tn++) {
                    sum[tm][tn] += acc[tm][tn];
#pragma unroll
                    for (int r = 0; r < 16; r++) acc[tm][tn][r] = 0.f;
                }
        }
    }
#pragma unroll
    for (int tm = 0; tm < 2; tm++)
#pragma unroll
        for (int tn = 0; tn < 2; tn++) sum[tm][tn] += acc[tm][tn];  // (the open chain; zeros when the last one closed)

    // a lane holds, per tile, 16 queries' values for ONE row: register r is query (r & 3) + 8 (r >> 2) + 4 half; the 32
    // lanes of a half-wave write 32 consecutive rows of one query: 128-byte stores
#pragma unroll
    for (int tn = 0; tn < 2; tn++) {
        const int64_t r = row_base + wn * 64 + tn * 32 + l31;
        if (r >= n) continue;
        const float rn = METRIC == 0 ? row_norms[r] : 0.f;
#pragma unroll
        for (int tm = 0; tm < 2; tm++)
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int q = q_base + wm * 64 + tm * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
                if (q < nq) out[(size_t)q * out_stride + r] = METRIC == 0 ? fmaf(-2.f, sum[tm][tn][e], rn) : -sum[tm][tn][e];
            }
    }
}

}  // namespace

// slices per accumulator chain and the products one chain holds at most (what the rounding bound is computed from)
static int dense_quarter(const RowGeom &g) { return ((g.nvec + 7) / 8 + 3) / 4; }
int dense_chain_length(const RowGeom &g, pgv_dtype dtype) {
    return dense_quarter(g) * (kDenseSlice / elem_size(dtype));
}

int launch_mfma_dense(pgv_ctx *ctx, pgv_metric metric, pgv_dtype dtype, const RowGeom &g, const void *rows, int64_t n,
                      const void *queries, int nq, const float *row_norms, float *out, int64_t out_stride) {
    if (n <= 0 || nq <= 0) return PGV_OK;
    if (metric == PGV_L2SQ && !row_norms) PGV_FAIL(PGV_ERR_ARG, "dense scan: L2 needs the rows' norms");
    if (metric != PGV_L2SQ && metric != PGV_NEG_IP) PGV_FAIL(PGV_ERR_ARG, "dense scan: metric %d", (int)metric);
    if (!ctx->zeros.p) {
        PGV_TRY(ctx->zeros.ensure(256));
        PGV_HIP(hipMemsetAsync(ctx->zeros.p, 0, 256, ctx->stream));
    }
    const int qtiles = (nq + kDenseTile - 1) / kDenseTile;
    const int64_t row_tiles = (n + kDenseTile - 1) / kDenseTile;
    const int64_t grid = 8 * ((row_tiles + 7) / 8) * qtiles;
    if (grid > 0x7fffffff) PGV_FAIL(PGV_ERR_ARG, "dense scan: too many workgroups");
    const size_t lds = 2 * (size_t)kDenseStage;
    const int quarter = dense_quarter(g);
#define PGV_DENSE(T, M)                                                                                                   \
    {                                                                                                                     \
        auto kern = mfma_dense_kernel<T, M>;                                                                              \
        PGV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,     \
                                    (int)lds));                                                                           \
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, ctx->stream, static_cast<const char *>(rows), n,   \
                           static_cast<const char *>(queries), nq, g.nvec, row_norms,                                     \
                           static_cast<const char *>(ctx->zeros.p), out, out_stride, qtiles, quarter);                    \
    }
    if (dtype == PGV_F32) {
        if (metric == PGV_L2SQ) PGV_DENSE(float, 0) else PGV_DENSE(float, 1)
    } else {
        if (metric == PGV_L2SQ) PGV_DENSE(__half, 0) else PGV_DENSE(__half, 1)
    }
#undef PGV_DENSE
    PGV_HIP(hipGetLastError());
    return PGV_OK;
}

}  // namespace pgv
