// kernels_build.hip -- the tuplesort of the IVFFlat build on the device.
//
// The reference's build feeds (list id, heap TID, vector) tuples to a tuplesort ordered by list id
// (src/ivfbuild.c:161-219 AddTupleToSort, :606-615 tuplesort_begin_heap on the list column) and InsertTuples
// (:271-331) walks the sorted stream list by list.  Rows that pgv_builder_add has assigned are still in HBM in heap
// order; this file brings them into list-major order there -- the order of the device mirror AND of the index's
// pages -- so that no host sort and no second upload exist:
//
//   build_keys_kernel     key = (list id << 32) | heap position, and the list's length counted
//   (rocPRIM radix sort of the 64-bit keys: ascending by list, heap order inside a list -- stable by construction)
//   build_offsets_kernel  exclusive scan of the lengths -> list_offsets
//   build_gather_kernel   row i of the mirror = staged row (key_i & 0xffffffff), 16 bytes per lane, plus its TID
//
// HBM-bound byte moving (one read + one write of every row); no distance is computed here.
#include <cstring>

#include "pgv_device.h"

#include <rocprim/rocprim.hpp>

namespace pgv {

namespace {

__global__ __launch_bounds__(256) void build_keys_kernel(const int32_t *__restrict__ lists, int64_t n, int nlists,
                                                         unsigned long long *__restrict__ keys,
                                                         unsigned long long *__restrict__ counts, int *__restrict__ bad) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int l = lists[i];
    if (l < 0 || l >= nlists) {  // cannot happen with pgv_builder_add's own assignment; a guard for the scatter
        *bad = 1;
        keys[i] = ~0ull;
        return;
    }
    keys[i] = ((unsigned long long)(unsigned)l << 32) | (unsigned long long)(unsigned)i;
    atomicAdd(&counts[l], 1ull);
}

__global__ __launch_bounds__(1024) void build_offsets_kernel(const unsigned long long *__restrict__ counts, int nlists,
                                                             int64_t *__restrict__ offsets) {
    __shared__ int64_t part[1024];
    // thread t owns lists [t * per, (t + 1) * per)
    const int per = (nlists + 1023) / 1024;
    int64_t local = 0;
    for (int j = 0; j < per; j++) {
        const int l = threadIdx.x * per + j;
        if (l < nlists) local += (int64_t)counts[l];
    }
    part[threadIdx.x] = local;
    __syncthreads();
    for (int st = 1; st < 1024; st <<= 1) {
        const int64_t t = threadIdx.x >= (unsigned)st ? part[threadIdx.x - st] : 0;
        __syncthreads();
        part[threadIdx.x] += t;
        __syncthreads();
    }
    int64_t run = part[threadIdx.x] - local;
    for (int j = 0; j < per; j++) {
        const int l = threadIdx.x * per + j;
        if (l < nlists) {
            offsets[l] = run;
            run += (int64_t)counts[l];
        }
    }
    if (threadIdx.x == 1023) offsets[nlists] = part[1023];
}

// one wavefront per row: 64 lanes x 16 bytes per trip
__global__ __launch_bounds__(256) void build_gather_kernel(const Raw16 *__restrict__ src, const unsigned long long *__restrict__ keys,
                                                           int64_t n, int nvec, Raw16 *__restrict__ dst,
                                                           const uint64_t *__restrict__ src_tids, uint64_t *__restrict__ dst_tids) {
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t i = (int64_t)blockIdx.x * (blockDim.x / kWave) + (threadIdx.x >> 6);
    if (i >= n) return;
    const int64_t from = (int64_t)(keys[i] & 0xffffffffull);
    const Raw16 *s = src + (size_t)from * nvec;
    Raw16 *d = dst + (size_t)i * nvec;
    for (int v = lane; v < nvec; v += kWave) d[v] = s[v];
    if (lane == 0 && dst_tids) dst_tids[i] = src_tids ? src_tids[from] : (uint64_t)from;
}

__global__ __launch_bounds__(256) void gather_words_kernel(const uint32_t *__restrict__ src, int words, int64_t nrows,
                                                           const int64_t *__restrict__ idx, int n, uint32_t *__restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)n * words) return;
    const int i = (int)(t / words), w = (int)(t % words);
    const int64_t r = idx[i];
    out[t] = (r >= 0 && r < nrows) ? src[(size_t)r * words + w] : 0u;
}

}  // namespace

int launch_gather_words(pgv_ctx *ctx, const void *src, int words_per_row, int64_t nrows, const int64_t *idx, int n,
                        uint32_t *out) {
    const int64_t total = (int64_t)n * words_per_row;
    if (total <= 0) return PGV_OK;
    hipLaunchKernelGGL(gather_words_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream,
                       static_cast<const uint32_t *>(src), words_per_row, nrows, idx, n, out);
    PGV_HIP(hipGetLastError());
    return PGV_OK;
}

size_t build_sort_scratch_bytes(int64_t n, int key_bits) {
    size_t bytes = 0;
    (void)rocprim::radix_sort_keys(nullptr, bytes, static_cast<unsigned long long *>(nullptr),
                                   static_cast<unsigned long long *>(nullptr), (size_t)n, 0, (unsigned)key_bits);
    return bytes;
}

// lists [n] (device) -> keys_sorted [n], offsets [nlists + 1] (device); counts [nlists] and keys_tmp [n] are scratch
int launch_build_order(pgv_ctx *ctx, const int32_t *lists, int64_t n, int nlists, unsigned long long *keys_tmp,
                       unsigned long long *keys_sorted, unsigned long long *counts, int64_t *offsets, int *bad,
                       void *sort_scratch, size_t sort_scratch_bytes) {
    PGV_HIP(hipMemsetAsync(counts, 0, sizeof(unsigned long long) * (size_t)nlists, ctx->stream));
    PGV_HIP(hipMemsetAsync(bad, 0, sizeof(int), ctx->stream));
    if (n > 0) {
        hipLaunchKernelGGL(build_keys_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, lists, n, nlists,
                           keys_tmp, counts, bad);
        PGV_HIP(hipGetLastError());
        int list_bits = 1;
        while ((1 << list_bits) < nlists) list_bits++;
        PGV_HIP(rocprim::radix_sort_keys(sort_scratch, sort_scratch_bytes, keys_tmp, keys_sorted, (size_t)n, 0,
                                         (unsigned)(32 + list_bits), ctx->stream));
    }
    hipLaunchKernelGGL(build_offsets_kernel, dim3(1), dim3(1024), 0, ctx->stream, counts, nlists, offsets);
    PGV_HIP(hipGetLastError());
    return PGV_OK;
}

int launch_build_gather(pgv_ctx *ctx, const void *src_rows, const unsigned long long *keys_sorted, int64_t n, int nvec,
                        void *dst_rows, const uint64_t *src_tids, uint64_t *dst_tids) {
    if (n <= 0) return PGV_OK;
    hipLaunchKernelGGL(build_gather_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, ctx->stream,
                       static_cast<const Raw16 *>(src_rows), keys_sorted, n, nvec, static_cast<Raw16 *>(dst_rows), src_tids,
                       dst_tids);
    PGV_HIP(hipGetLastError());
    return PGV_OK;
}

}  // namespace pgv
