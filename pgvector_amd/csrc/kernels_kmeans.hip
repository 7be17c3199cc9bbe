// kernels_kmeans.hip -- the k-means pieces of IVFFlat build that are not a
// distance kernel: k-means++ weight bookkeeping and the D^2 pick
// (src/ivfkmeans.c:64-84), per-center fp32 sums in sample order and the
// center update (src/ivfkmeans.c:151-236, src/ivfutils.c:301-361), the
// spherical renormalisation (src/vector.c:785-819, src/halfvec.c:724-759) and
// CheckCenters (src/ivfkmeans.c:490-547).  All HBM-bound byte shuffling; the
// distances themselves come from kernels_scan.hip / kernels_pair.hip.
#include "pgv_device.h"

#include <cfloat>
#include <climits>

namespace pgv {

namespace {

constexpr int kKmThreads = 256;

// ---------------------------------------------------------------- k-means++

// raw[j] = FUNCTION-1-style kernel value of (sample j, newest center):
//   spherical == 0: L2 squared          -> distance = sqrt((double) raw)      (vector.c:588)
//   spherical == 1: negative inner prod -> distance = acos(clamp(ip)) / pi    (vector.c:713-721)
// weight[j] = min(weight[j], (float) distance^2) (ivfkmeans.c:64-68); per-block sums in double.
__global__ __launch_bounds__(kKmThreads) void kmpp_update_kernel(
    const float *__restrict__ raw, float *__restrict__ weight, int n, int spherical,
    double *__restrict__ block_sums) {
    __shared__ double red[kKmThreads];
    const int j = blockIdx.x * kKmThreads + threadIdx.x;
    double w = 0.0;
    if (j < n) {
        double distance;
        if (spherical) {
            double ip = -(double)raw[j];
            if (ip > 1)
                ip = 1;
            else if (ip < -1)
                ip = -1;
            distance = acos(ip) / 3.14159265358979323846;
        } else {
            distance = sqrt((double)raw[j]);
        }
        distance *= distance;
        float cur = weight[j];
        if (distance < (double)cur) {
            cur = (float)distance;
            weight[j] = cur;
        }
        w = (double)cur;
    }
    // in-order tree: the same association every run
    red[threadIdx.x] = w;
    __syncthreads();
    for (int s = kKmThreads / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) block_sums[blockIdx.x] = red[0];
}

// The reference's walk `choice -= v[j]; if (choice <= 0) break;` over [begin, end), kept in its
// sequential association but without a data-dependent exit in the serial part: lane 0 writes the
// running value after every step of a 256-entry batch (independent LDS loads, one dependent
// subtraction each), then the block looks for the first non-positive one in parallel.
// Returns that index (or `end`); *carry is the running value on entry to it.
struct WalkLds {
    double v[kKmThreads];
    double r[kKmThreads];
    double carry;
    int first;
};

template <typename V>
__device__ int walk_first_nonpositive(const V *__restrict__ vals, int begin, int end, WalkLds &w) {
    const int tid = threadIdx.x;
    for (int base = begin; base < end; base += kKmThreads) {
        const int i = base + tid;
        w.v[tid] = i < end ? (double)vals[i] : 0.0;
        if (tid == 0) w.first = INT_MAX;
        __syncthreads();
        if (tid == 0) {
            double c = w.carry;
#pragma unroll 16
            for (int j = 0; j < kKmThreads; j++) {
                c -= w.v[j];
                w.r[j] = c;
            }
        }
        __syncthreads();
        if (i < end && w.r[tid] <= 0) atomicMin(&w.first, tid);
        __syncthreads();
        const int f = w.first;
        if (tid == 0) w.carry = f == INT_MAX ? w.r[kKmThreads - 1] : (f == 0 ? w.carry : w.r[f - 1]);
        __syncthreads();
        if (f != INT_MAX) return base + f;
    }
    return end;
}

// choice = sum * RandomDouble(); walk the weights until it is used up
// (ivfkmeans.c:77-84); the chosen sample becomes center `next`.
__global__ __launch_bounds__(kKmThreads) void kmpp_pick_kernel(
    const char *__restrict__ samples, int n, const float *__restrict__ weight,
    const double *__restrict__ block_sums, int nblocks, const double *__restrict__ draws,
    int round, char *__restrict__ centers, int nvec, int32_t *__restrict__ picked) {
    __shared__ WalkLds w;
    // the total: block sums added in block order
    double sum = 0.0;
    for (int base = 0; base < nblocks; base += kKmThreads) {
        const int i = base + threadIdx.x;
        __syncthreads();
        w.v[threadIdx.x] = i < nblocks ? block_sums[i] : 0.0;
        __syncthreads();
        if (threadIdx.x == 0) {
#pragma unroll 16
            for (int j = 0; j < kKmThreads; j++) sum += w.v[j];
        }
    }
    if (threadIdx.x == 0) w.carry = sum * draws[round];
    __syncthreads();
    // skip whole blocks while the walk cannot end inside them, then walk that block's weights
    // (and on, should rounding carry the walk past it); the last sample ends it regardless
    const int b = walk_first_nonpositive(block_sums, 0, nblocks - 1, w);
    const int chosen = walk_first_nonpositive(weight, b * kKmThreads, n - 1, w);
    if (threadIdx.x == 0) picked[round + 1] = chosen;
    const size_t row_bytes = (size_t)nvec * sizeof(Raw16);
    const Raw16 *src = reinterpret_cast<const Raw16 *>(samples + (size_t)chosen * row_bytes);
    Raw16 *dst = reinterpret_cast<Raw16 *>(centers + (size_t)(round + 1) * row_bytes);
    for (int v = threadIdx.x; v < nvec; v += kKmThreads) dst[v] = src[v];
}

// The same pick with the samples sharded by row over `nranks` processes (global sample order =
// rank order): every rank knows every rank's weight total (all-gathered); the walk first steps
// over whole ranks, then -- on the rank it ends in -- over that rank's blocks and weights.  The
// owner writes the chosen row into its send slot, everybody else zeros; after the all-gather of
// the slots every rank copies slot *owner_out into centers[round + 1].
__global__ __launch_bounds__(kKmThreads) void kmpp_pick_sharded_kernel(
    const char *__restrict__ samples, int n, const float *__restrict__ weight,
    const double *__restrict__ block_sums, int nblocks, const double *__restrict__ totals, int nranks, int rank,
    const double *__restrict__ draws, int round, char *__restrict__ send_row, int nvec, int32_t *__restrict__ owner_out) {
    __shared__ WalkLds w;
    __shared__ int owner;
    if (threadIdx.x == 0) {
        double grand = 0.0;
        for (int r = 0; r < nranks; r++) grand += totals[r];
        double choice = grand * draws[round];
        int o = nranks - 1;
        for (int r = 0; r < nranks - 1; r++) {
            // a rank without weight cannot hold the choice; the last rank with samples ends the walk regardless
            if (choice - totals[r] <= 0 && totals[r] > 0) {
                o = r;
                break;
            }
            choice -= totals[r];
        }
        owner = o;
        w.carry = choice;
        *owner_out = o;
    }
    __syncthreads();
    const size_t row_bytes = (size_t)nvec * sizeof(Raw16);
    Raw16 *dst = reinterpret_cast<Raw16 *>(send_row);
    if (owner != rank || n <= 0) {
        for (int v = threadIdx.x; v < nvec; v += kKmThreads) dst[v] = raw16_zero();
        return;
    }
    const int b = walk_first_nonpositive(block_sums, 0, nblocks - 1, w);
    const int chosen = walk_first_nonpositive(weight, b * kKmThreads, n - 1, w);
    const Raw16 *src = reinterpret_cast<const Raw16 *>(samples + (size_t)chosen * row_bytes);
    for (int v = threadIdx.x; v < nvec; v += kKmThreads) dst[v] = src[v];
}

// this rank's weight total: block sums added in block order (the association kmpp_pick_kernel uses)
__global__ void kmpp_total_kernel(const double *__restrict__ block_sums, int nblocks, double *__restrict__ out) {
    double sum = 0.0;
    for (int i = 0; i < nblocks; i++) sum += block_sums[i];
    *out = sum;
}

// centers[round + 1] = the owner's slot of the gathered rows
__global__ void kmpp_take_row_kernel(const char *__restrict__ gathered, const int32_t *__restrict__ owner, int nvec,
                                     char *__restrict__ centers, int round) {
    const size_t row_bytes = (size_t)nvec * sizeof(Raw16);
    const Raw16 *src = reinterpret_cast<const Raw16 *>(gathered + (size_t)*owner * row_bytes);
    Raw16 *dst = reinterpret_cast<Raw16 *>(centers + (size_t)(round + 1) * row_bytes);
    for (int v = blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += gridDim.x * blockDim.x) dst[v] = src[v];
}

// the all-reduced Lloyd record [sums k x ld | counts k | changes 1] (all fp32: counts and the change
// count are exact below 2^24) back into the integer counters the finish step reads; the pinned
// host record gets (changes, empty clusters), its sequence word last -- what steers the host loop
__global__ __launch_bounds__(kKmThreads) void lloyd_unpack_kernel(const float *__restrict__ tail, int k,
                                                                  int32_t *__restrict__ counts,
                                                                  unsigned long long *__restrict__ changes,
                                                                  long long *__restrict__ host_rec, long long seq) {
    __shared__ int empty;
    if (threadIdx.x == 0) empty = 0;
    __syncthreads();
    int mine = 0;
    for (int c = threadIdx.x; c < k; c += kKmThreads) {
        const int v = (int)tail[c];
        counts[c] = v;
        mine += v <= 0 ? 1 : 0;
    }
    if (mine) atomicAdd(&empty, mine);
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long ch = (unsigned long long)tail[k];
        *changes = ch;
        if (host_rec) {
            host_rec[0] = (long long)ch;
            host_rec[1] = empty;
            __hip_atomic_store(&host_rec[2], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// counts / changes of the local partial step appended to the sums as fp32, ready for one all-reduce
__global__ __launch_bounds__(kKmThreads) void lloyd_pack_kernel(const int32_t *__restrict__ counts,
                                                                const unsigned long long *__restrict__ changes, int k,
                                                                float *__restrict__ tail) {
    for (int c = blockIdx.x * kKmThreads + threadIdx.x; c <= k; c += gridDim.x * kKmThreads)
        tail[c] = c < k ? (float)counts[c] : (float)*changes;
}

// ------------------------------------------------------------- Lloyd pieces

__global__ __launch_bounds__(kKmThreads) void changes_hist_kernel(
    const int32_t *__restrict__ closest_new, int32_t *__restrict__ closest_io, int n,
    int32_t *__restrict__ counts, unsigned long long *__restrict__ changes) {
    const int j = blockIdx.x * kKmThreads + threadIdx.x;
    bool changed = false;
    if (j < n) {
        const int c = closest_new[j];
        changed = closest_io[j] != c;
        closest_io[j] = c;
        atomicAdd(&counts[c], 1);
    }
    const unsigned long long bal = __ballot(changed);
    if ((threadIdx.x & (kWave - 1)) == 0 && bal) atomicAdd(changes, (unsigned long long)__popcll(bal));
}

__global__ __launch_bounds__(1024) void offsets_kernel(const int32_t *__restrict__ counts, int k,
                                                       int32_t *__restrict__ offsets) {
    __shared__ int scratch[1024];
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < k; base += 1024) {
        const int i = base + threadIdx.x;
        const int v = i < k ? counts[i] : 0;
        scratch[threadIdx.x] = v;
        __syncthreads();
        for (int s = 1; s < 1024; s <<= 1) {
            int t = threadIdx.x >= (unsigned)s ? scratch[threadIdx.x - s] : 0;
            __syncthreads();
            scratch[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < k) offsets[i] = carry + scratch[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry += scratch[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) offsets[k] = carry;
}

// one wavefront per center: its members in ascending sample order (a stable
// compaction), so the sums below add in exactly the order SumCenters does
__global__ __launch_bounds__(kKmThreads) void members_kernel(
    const int32_t *__restrict__ closest, int n, int k, const int32_t *__restrict__ offsets,
    int32_t *__restrict__ members) {
    const int c = blockIdx.x * (kKmThreads / kWave) + (threadIdx.x >> 6);
    if (c >= k) return;
    const int lane = threadIdx.x & (kWave - 1);
    int at = offsets[c];
    for (int base = 0; base < n; base += kWave) {
        const int j = base + lane;
        const bool mine = j < n && closest[j] == c;
        const unsigned long long bal = __ballot(mine);
        if (mine) members[at + __popcll(bal & ((1ull << lane) - 1ull))] = j;
        at += __popcll(bal);
    }
}

// sums[c][d] = fp32 sum over members of c in sample order (ivfutils.c:340-361);
// lanes map to dimensions, so every lane runs the reference's scalar chain.
template <typename T>
__global__ __launch_bounds__(kKmThreads) void center_sums_kernel(
    const char *__restrict__ samples, const int32_t *__restrict__ offsets,
    const int32_t *__restrict__ members, int nvec, float *__restrict__ sums) {
    constexpr int N = VecTraits<T>::N;
    const int c = blockIdx.y;
    const int v = blockIdx.x * kKmThreads + threadIdx.x;
    if (v >= nvec) return;
    const size_t row_bytes = (size_t)nvec * sizeof(Raw16);
    float acc[N];
#pragma unroll
    for (int e = 0; e < N; e++) acc[e] = 0.f;
    const int beg = offsets[c], end = offsets[c + 1];
    for (int m = beg; m < end; m++) {
        const Raw16 r = load16(samples + (size_t)members[m] * row_bytes + (size_t)v * sizeof(Raw16));
        Unpacked<T> u(r);
#pragma unroll
        for (int e = 0; e < N; e++) acc[e] += u.v[e];
    }
    float *dst = sums + ((size_t)c * nvec + v) * N;
#pragma unroll
    for (int e = 0; e < N; e++) dst[e] = acc[e];
}

template <typename T> __device__ __forceinline__ void store_elem(char *row, int d, float x);
template <> __device__ __forceinline__ void store_elem<float>(char *row, int d, float x) {
    reinterpret_cast<float *>(row)[d] = x;  // VectorUpdateCenter, ivfutils.c:301-311
}
template <> __device__ __forceinline__ void store_elem<__half>(char *row, int d, float x) {
    // HalfvecUpdateCenter: Float4ToHalfUnchecked = round-to-nearest-even (halfutils.h:146-152)
    reinterpret_cast<__half *>(row)[d] = __float2half_rn(x);
}

// centers[c] = sums[c] / count (inf clamped to +-FLT_MAX), or the refill row for
// an empty cluster (ivfkmeans.c:205-231)
template <typename T>
__global__ __launch_bounds__(kKmThreads) void finish_centers_kernel(
    const float *__restrict__ sums, const int32_t *__restrict__ counts,
    const float *__restrict__ refill, const int32_t *__restrict__ refill_row, int dim, int ld,
    char *__restrict__ centers) {
    const int c = blockIdx.y;
    const int d = blockIdx.x * kKmThreads + threadIdx.x;
    if (d >= ld) return;
    const size_t row_bytes = (size_t)ld * sizeof(T);
    float x = 0.f;
    if (d < dim) {
        const int cnt = counts[c];
        if (cnt > 0) {
            x = sums[(size_t)c * ld + d];
            if (isinf(x)) x = x > 0 ? FLT_MAX : -FLT_MAX;
            x /= (float)cnt;
        } else {
            x = refill[(size_t)refill_row[c] * dim + d];
        }
    }
    store_elem<T>(centers + (size_t)c * row_bytes, d, x);
}

template <typename T> __device__ __forceinline__ float load_elem(const char *row, int d);
template <> __device__ __forceinline__ float load_elem<float>(const char *row, int d) {
    return reinterpret_cast<const float *>(row)[d];
}
template <> __device__ __forceinline__ float load_elem<__half>(const char *row, int d) {
    return __half2float(reinterpret_cast<const __half *>(row)[d]);
}

// l2_normalize / halfvec_l2_normalize in place, one wavefront per row: norm in
// double, element = (float)(x / norm), zero rows stay zero.  flag gets bit 0 on
// an overflowing element (float_overflow_error in the reference).
template <typename T>
__global__ __launch_bounds__(kKmThreads) void normalize_rows_kernel(char *__restrict__ rows,
                                                                    int64_t n, int dim, int ld,
                                                                    int32_t *__restrict__ flag) {
    const int64_t r = (int64_t)blockIdx.x * (kKmThreads / kWave) + (threadIdx.x >> 6);
    if (r >= n) return;
    const int lane = threadIdx.x & (kWave - 1);
    char *row = rows + (size_t)r * ld * sizeof(T);
    double acc = 0.0;
    for (int d = lane; d < dim; d += kWave) {
        const double x = (double)load_elem<T>(row, d);
        acc += x * x;
    }
    for (int m = 32; m > 0; m >>= 1) acc += __shfl_xor(acc, m);
    const double norm = sqrt(acc);
    if (!(norm > 0)) return;
    for (int d = lane; d < dim; d += kWave) {
        const float y = (float)((double)load_elem<T>(row, d) / norm);
        store_elem<T>(row, d, y);
        if (isinf(load_elem<T>(row, d))) atomicOr(flag, 1);
    }
}

// CheckElements + CheckNorms (ivfkmeans.c:490-536): bit 1 = NaN, bit 2 = inf, bit 3 = zero norm
template <typename T>
__global__ __launch_bounds__(kKmThreads) void check_centers_kernel(const char *__restrict__ rows,
                                                                   int k, int dim, int ld,
                                                                   int check_zero_norm,
                                                                   int32_t *__restrict__ flag) {
    const int c = blockIdx.x * (kKmThreads / kWave) + (threadIdx.x >> 6);
    if (c >= k) return;
    const int lane = threadIdx.x & (kWave - 1);
    const char *row = rows + (size_t)c * ld * sizeof(T);
    double acc = 0.0;
    int bad = 0;
    for (int d = lane; d < dim; d += kWave) {
        const float x = load_elem<T>(row, d);
        if (isnan(x)) bad |= 2;
        if (isinf(x)) bad |= 4;
        acc += (double)x * (double)x;
    }
    for (int m = 32; m > 0; m >>= 1) acc += __shfl_xor(acc, m);
    if (check_zero_norm && lane == 0 && sqrt(acc) == 0) bad |= 8;
    if (bad) atomicOr(flag, bad);
}

}  // namespace

int kmpp_block_count(int n) { return (n + kKmThreads - 1) / kKmThreads; }

int launch_kmpp_update(pgv_ctx *ctx, const float *raw, float *weight, int n, int spherical,
                       double *block_sums) {
    hipLaunchKernelGGL(kmpp_update_kernel, dim3(kmpp_block_count(n)), dim3(kKmThreads), 0,
                       ctx->stream, raw, weight, n, spherical, block_sums);
    PGV_HIP(hipGetLastError());
    return PGV_OK;
}

int launch_kmpp_pick(pgv_ctx *ctx, const RowGeom &g, const void *samples, int n,
                     const float *weight, const double *block_sums, const double *draws,
                     int round, void *centers, int32_t *picked) {
    hipLaunchKernelGGL(kmpp_pick_kernel, dim3(1), dim3(kKmThreads), 0, ctx->stream,
                       static_cast<const char *>(samples), n, weight, block_sums,
                       kmpp_block_count(n), draws, round, static_cast<char *>(centers), g.nvec,
                       picked);
    PGV_HIP(hipGetLastError());
    return PGV_OK;
}

int launch_kmpp_total(pgv_ctx *ctx, const double *block_sums, int nblocks, double *out) {
    hipLaunchKernelGGL(kmpp_total_kernel, dim3(1), dim3(1), 0, ctx->stream, block_sums, nblocks, out);
    PGV_HIP(hipGetLastError());
    return PGV_OK;
}

int launch_kmpp_pick_sharded(pgv_ctx *ctx, const RowGeom &g, const void *samples, int n, const float *weight,
                             const double *block_sums, const double *totals, int nranks, int rank, const double *draws,
                             int round, void *send_row, int32_t *owner_out) {
    hipLaunchKernelGGL(kmpp_pick_sharded_kernel, dim3(1), dim3(kKmThreads), 0, ctx->stream,
                       static_cast<const char *>(samples), n, weight, block_sums, kmpp_block_count(n), totals, nranks,
                       rank, draws, round, static_cast<char *>(send_row), g.nvec, owner_out);
    PGV_HIP(hipGetLastError());
    return PGV_OK;
}

int launch_kmpp_take_row(pgv_ctx *ctx, const RowGeom &g, const void *gathered, const int32_t *owner, void *centers,
                         int round) {
    hipLaunchKernelGGL(kmpp_take_row_kernel, dim3(1), dim3(kKmThreads), 0, ctx->stream,
                       static_cast<const char *>(gathered), owner, g.nvec, static_cast<char *>(centers), round);
    PGV_HIP(hipGetLastError());
    return PGV_OK;
}

int launch_lloyd_pack(pgv_ctx *ctx, const int32_t *counts, const unsigned long long *changes, int k, float *tail) {
    hipLaunchKernelGGL(lloyd_pack_kernel, dim3((k + kKmThreads) / kKmThreads), dim3(kKmThreads), 0, ctx->stream, counts,
                       changes, k, tail);
    PGV_HIP(hipGetLastError());
    return PGV_OK;
}

int launch_lloyd_unpack(pgv_ctx *ctx, const float *tail, int k, int32_t *counts, unsigned long long *changes,
                        long long *host_rec, long long seq) {
    hipLaunchKernelGGL(lloyd_unpack_kernel, dim3(1), dim3(kKmThreads), 0, ctx->stream, tail, k, counts, changes,
                       host_rec, seq);
    PGV_HIP(hipGetLastError());
    return PGV_OK;
}

int launch_changes_hist(pgv_ctx *ctx, const int32_t *closest_new, int32_t *closest_io, int n,
                        int32_t *counts, unsigned long long *changes) {
    if (n <= 0) return PGV_OK;
    hipLaunchKernelGGL(changes_hist_kernel, dim3((n + kKmThreads - 1) / kKmThreads),
                       dim3(kKmThreads), 0, ctx->stream, closest_new, closest_io, n, counts,
                       changes);
    PGV_HIP(hipGetLastError());
    return PGV_OK;
}

int launch_members(pgv_ctx *ctx, const int32_t *closest, int n, int k, const int32_t *counts,
                   int32_t *offsets, int32_t *members) {
    hipLaunchKernelGGL(offsets_kernel, dim3(1), dim3(1024), 0, ctx->stream, counts, k, offsets);
    const int per = kKmThreads / kWave;
    hipLaunchKernelGGL(members_kernel, dim3((k + per - 1) / per), dim3(kKmThreads), 0, ctx->stream,
                       closest, n, k, offsets, members);
    PGV_HIP(hipGetLastError());
    return PGV_OK;
}

int launch_center_sums(pgv_ctx *ctx, pgv_dtype dtype, const RowGeom &g, const void *samples,
                       const int32_t *offsets, const int32_t *members, int k, float *sums) {
    dim3 grid((g.nvec + kKmThreads - 1) / kKmThreads, k);
    if (dtype == PGV_F32)
        hipLaunchKernelGGL(center_sums_kernel<float>, grid, dim3(kKmThreads), 0, ctx->stream,
                           static_cast<const char *>(samples), offsets, members, g.nvec, sums);
    else
        hipLaunchKernelGGL(center_sums_kernel<__half>, grid, dim3(kKmThreads), 0, ctx->stream,
                           static_cast<const char *>(samples), offsets, members, g.nvec, sums);
    PGV_HIP(hipGetLastError());
    return PGV_OK;
}

int launch_finish_centers(pgv_ctx *ctx, pgv_dtype dtype, const RowGeom &g, int k, int dim,
                          const float *sums, const int32_t *counts, const float *refill,
                          const int32_t *refill_row, void *centers) {
    dim3 grid((g.ld + kKmThreads - 1) / kKmThreads, k);
    if (dtype == PGV_F32)
        hipLaunchKernelGGL(finish_centers_kernel<float>, grid, dim3(kKmThreads), 0, ctx->stream,
                           sums, counts, refill, refill_row, dim, g.ld,
                           static_cast<char *>(centers));
    else
        hipLaunchKernelGGL(finish_centers_kernel<__half>, grid, dim3(kKmThreads), 0, ctx->stream,
                           sums, counts, refill, refill_row, dim, g.ld,
                           static_cast<char *>(centers));
    PGV_HIP(hipGetLastError());
    return PGV_OK;
}

int launch_normalize_rows(pgv_ctx *ctx, pgv_dtype dtype, const RowGeom &g, void *rows, int64_t n,
                          int dim, int32_t *flag) {
    if (n <= 0) return PGV_OK;
    const int per = kKmThreads / kWave;
    dim3 grid((unsigned)((n + per - 1) / per));
    if (dtype == PGV_F32)
        hipLaunchKernelGGL(normalize_rows_kernel<float>, grid, dim3(kKmThreads), 0, ctx->stream,
                           static_cast<char *>(rows), n, dim, g.ld, flag);
    else
        hipLaunchKernelGGL(normalize_rows_kernel<__half>, grid, dim3(kKmThreads), 0, ctx->stream,
                           static_cast<char *>(rows), n, dim, g.ld, flag);
    PGV_HIP(hipGetLastError());
    return PGV_OK;
}

int launch_check_centers(pgv_ctx *ctx, pgv_dtype dtype, const RowGeom &g, const void *centers,
                         int k, int dim, int check_zero_norm, int32_t *flag) {
    const int per = kKmThreads / kWave;
    dim3 grid((k + per - 1) / per);
    if (dtype == PGV_F32)
        hipLaunchKernelGGL(check_centers_kernel<float>, grid, dim3(kKmThreads), 0, ctx->stream,
                           static_cast<const char *>(centers), k, dim, g.ld, check_zero_norm,
                           flag);
    else
        hipLaunchKernelGGL(check_centers_kernel<__half>, grid, dim3(kKmThreads), 0, ctx->stream,
                           static_cast<const char *>(centers), k, dim, g.ld, check_zero_norm,
                           flag);
    PGV_HIP(hipGetLastError());
    return PGV_OK;
}

}  // namespace pgv
