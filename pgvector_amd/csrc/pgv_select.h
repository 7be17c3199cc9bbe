// pgv_select.h -- k smallest of a sequence by (value, position), one 256-thread workgroup.
//
// The head of the reference's ascending tuplesort stream (src/ivfscan.c:182, :238-247) and the
// bounded heap of GetScanLists (:76-106): exact and deterministic.  Keys are order-preserving
// uint images of the floats (NaN last, like float8 ordering), ties go to the lower position (a
// deterministic refinement of tuplesort's unspecified tie order; for GetScanLists it IS the
// reference's rule: the strict `<` at :92 keeps the earlier list).
//
// Shared by topk_kernel (kernels_select.hip: one segment per workgroup) and the single-query
// kernels (kernels_query.hip: the last workgroup to finish selects).
#pragma once

#include "pgv_device.h"


namespace pgv {

namespace {

constexpr int kSelThreads = 256;
constexpr int kBins = 2048;
constexpr int kFastCap = 1024;  // candidates the threshold pre-filter may keep

struct SelShared {
    unsigned hist[kBins];
    unsigned scan[kSelThreads];
    unsigned wave_tot[kSelThreads / kWave];
    unsigned count;      // entries collected so far
    unsigned bin;        // selected bin of the current pass
    unsigned remaining;  // how many of the selected bin are still needed
    unsigned running;    // ordered pass: equal keys seen so far
};

// find the bin holding the `want`-th (0-based) smallest element; leaves the bin in s->bin and
// the rank inside it in s->remaining
__device__ inline void pick_bin(SelShared *s, int nbins, unsigned want) {
    // each thread owns nbins/kSelThreads consecutive bins
    const int per = nbins / kSelThreads;
    unsigned local = 0;
    for (int j = 0; j < per; j++) local += s->hist[threadIdx.x * per + j];
    unsigned *scan = s->scan;
    scan[threadIdx.x] = local;
    __syncthreads();
    for (int st = 1; st < kSelThreads; st <<= 1) {
        unsigned t = threadIdx.x >= (unsigned)st ? scan[threadIdx.x - st] : 0;
        __syncthreads();
        scan[threadIdx.x] += t;
        __syncthreads();
    }
    const unsigned before = scan[threadIdx.x] - local;
    if (want >= before && want < before + local) {
        unsigned acc = before;
        for (int j = 0; j < per; j++) {
            const unsigned h = s->hist[threadIdx.x * per + j];
            if (want < acc + h) {
                s->bin = threadIdx.x * per + j;
                s->remaining = want - acc;  // rank inside the bin
                break;
            }
            acc += h;
        }
    }
    __syncthreads();
}

// bitonic sort of ent[0, kp), kp a power of two
__device__ inline void sort_entries(unsigned long long *ent, int kp) {
    for (int size = 2; size <= kp; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = threadIdx.x; i < kp / 2; i += kSelThreads) {
                const int lo = 2 * i - (i & (stride - 1));
                const int hi = lo + stride;
                const bool up = (lo & size) == 0;
                const unsigned long long a = ent[lo], b = ent[hi];
                if ((a > b) == up) {
                    ent[lo] = b;
                    ent[hi] = a;
                }
            }
            __syncthreads();
        }
}

// <= 256 distinct entries: every thread counts how many are smaller than its own and moves it
// there -- one pass over LDS broadcasts instead of the bitonic network's dozens of barriers
__device__ inline void rank_sort_entries(unsigned long long *ent, int n) {
    const unsigned long long mine = (int)threadIdx.x < n ? ent[threadIdx.x] : ~0ull;
    int rank = 0;
    // entries past n are ~0 (never smaller than a real entry): whole 16-byte reads, several in flight
    const ulonglong2 *e2 = reinterpret_cast<const ulonglong2 *>(ent);
    const int n2 = (n + 1) / 2;
#pragma unroll 8
    for (int j = 0; j < n2; j++) {
        const ulonglong2 o = e2[j];
        rank += o.x < mine ? 1 : 0;
        rank += o.y < mine ? 1 : 0;
    }
    __syncthreads();
    if ((int)threadIdx.x < n) ent[rank] = mine;
    __syncthreads();
}

// rank of `mine` among the 256 per-thread minima (ties: lower thread first): LDS broadcasts, 16 bytes a
// read, a few reads in flight (a scalar loop of dependent ds_reads costs ~60 cycles an element)
__device__ inline unsigned rank_among_minima(const unsigned *mins, unsigned mine) {
    const uint4 *m4 = reinterpret_cast<const uint4 *>(mins);
    unsigned rank = 0;
#pragma unroll 4
    for (int j4 = 0; j4 < kSelThreads / 4; j4++) {
        const uint4 o = m4[j4];
        const int j = 4 * j4, t = (int)threadIdx.x;
        rank += (o.x < mine || (o.x == mine && j < t)) ? 1u : 0u;
        rank += (o.y < mine || (o.y == mine && j + 1 < t)) ? 1u : 0u;
        rank += (o.z < mine || (o.z == mine && j + 2 < t)) ? 1u : 0u;
        rank += (o.w < mine || (o.w == mine && j + 3 < t)) ? 1u : 0u;
    }
    return rank;
}

// The k smallest of load(0 .. m) by (key, position), ascending, left in ent[0, min(k, m)); the
// rest of ent[0, kp) is ~0.  kp = k rounded up to a power of two (>= 2), cap >= max(kp, kFastCap)
// entries of LDS behind `ent`.  An entry is (key << 32) | position.
// (everything here is inlined into its kernel: as a called function the selection saved and restored registers
// through scratch memory on the one workgroup's critical path, and its kernels needed a private segment)
template <typename Load>
__device__ __forceinline__ void block_topk(Load load, int64_t m, int k, int kp, int cap, unsigned long long *ent,
                                           SelShared *s) {
    for (int i = threadIdx.x; i < cap; i += kSelThreads) ent[i] = ~0ull;
    if (threadIdx.x == 0) s->count = 0;
    __syncthreads();

    int sort_n = kp;     // how many entries the final sort covers
    bool done = false;   // block-uniform: the fast path produced the candidates
    if (m > k && k <= kSelThreads / 2) {
        // Fast path for the usual "small k of a long sequence": every thread takes the minimum
        // of its strided share; the k-th smallest of those 256 minima is a real element, hence
        // an upper bound T0 of the k-th smallest overall.  One more sweep (L2-resident by now)
        // keeps everything <= T0 -- a few dozen candidates -- which are then sorted by
        // (key, position) exactly like the general path.  Falls through to the radix select
        // if the candidates do not fit (long runs of equal keys).
        unsigned *mins = s->hist;  // scratch
        unsigned mine = 0xffffffffu;
        // sweeps: eight independent loads in flight per thread and iteration, the ragged end as one more
        // full batch with clamped addresses; the loop stays rolled -- a single workgroup running straight-line
        // code once is bound by instruction fetch, so compact beats unrolled here (measured)
        constexpr int U = 8;
#pragma unroll 1
        for (int64_t i = threadIdx.x; i < m; i += (int64_t)U * kSelThreads) {
            float v[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int64_t at = i + (int64_t)u * kSelThreads;
                v[u] = load(at < m ? at : m - 1);
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const unsigned key = i + (int64_t)u * kSelThreads < m ? float_to_key(v[u]) : 0xffffffffu;
                mine = key < mine ? key : mine;
            }
        }
        mins[threadIdx.x] = mine;
        __syncthreads();
        const unsigned rank = rank_among_minima(mins, mine);
        if (rank == (unsigned)(k - 1)) s->bin = mine;  // exactly one thread has this rank
        __syncthreads();
        const unsigned t0 = s->bin;
#pragma unroll 1
        for (int64_t i = threadIdx.x; i < m; i += (int64_t)U * kSelThreads) {
            float v[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int64_t at = i + (int64_t)u * kSelThreads;
                v[u] = load(at < m ? at : m - 1);
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int64_t at = i + (int64_t)u * kSelThreads;
                const unsigned key = float_to_key(v[u]);
                if (at < m && key <= t0) {
                    const unsigned slot = atomicAdd(&s->count, 1u);
                    if (slot < (unsigned)cap) ent[slot] = ((unsigned long long)key << 32) | (unsigned)at;
                }
            }
        }
        __syncthreads();
        const unsigned got = s->count;
        if (got <= (unsigned)cap) {
            done = true;
            sort_n = kp;
            while (sort_n < (int)got) sort_n <<= 1;
            if (got <= (unsigned)kSelThreads) {
                rank_sort_entries(ent, (int)got);
                return;
            }
        } else {
            __syncthreads();
            for (int i = threadIdx.x; i < cap; i += kSelThreads) ent[i] = ~0ull;
            if (threadIdx.x == 0) s->count = 0;
            __syncthreads();
        }
    }

    if (done) {
        // candidates are in ent[0, got)
    } else if (m <= k) {
        for (int64_t i = threadIdx.x; i < m; i += kSelThreads)
            ent[i] = ((unsigned long long)float_to_key(load(i)) << 32) | (unsigned)i;
        __syncthreads();
    } else {
        // three radix passes (11 + 11 + 10 bits) pin down the k-th smallest key exactly
        unsigned prefix = 0;
        unsigned want = (unsigned)(k - 1);
        unsigned count_eq = 0;
        for (int pass = 0; pass < 3; pass++) {
            const int shift = pass == 0 ? 21 : (pass == 1 ? 10 : 0);
            const int nbins = pass == 2 ? 1024 : 2048;
            for (int i = threadIdx.x; i < kBins; i += kSelThreads) s->hist[i] = 0;
            __syncthreads();
            for (int64_t i = threadIdx.x; i < m; i += kSelThreads) {
                const unsigned key = float_to_key(load(i));
                const bool in = pass == 0 || (pass == 1 ? (key >> 21) == (prefix >> 21)
                                                        : (key >> 10) == (prefix >> 10));
                if (in) atomicAdd(&s->hist[(key >> shift) & (nbins - 1)], 1u);
            }
            __syncthreads();
            pick_bin(s, nbins, want);
            prefix |= s->bin << shift;
            want = s->remaining;
            count_eq = s->hist[s->bin];
            __syncthreads();
        }
        const unsigned thr = prefix;      // the k-th smallest key
        const unsigned need_eq = want + 1; // how many keys == thr belong to the top k
        if (threadIdx.x == 0) s->running = 0;
        __syncthreads();
        const bool ordered = count_eq > need_eq;  // ties on the boundary: lowest positions win
        const int64_t padded = (m + kSelThreads - 1) / kSelThreads * kSelThreads;
        for (int64_t i = threadIdx.x; i < padded; i += kSelThreads) {
            const unsigned key = i < m ? float_to_key(load(i)) : 0xffffffffu;
            const bool valid = i < m;
            if (valid && key < thr) {
                const unsigned at = atomicAdd(&s->count, 1u);
                ent[at] = ((unsigned long long)key << 32) | (unsigned)i;
            }
            if (!ordered) {
                if (valid && key == thr) {
                    const unsigned at = atomicAdd(&s->count, 1u);
                    ent[at] = ((unsigned long long)key << 32) | (unsigned)i;
                }
            } else {
                // position-ordered rank among equal keys (block-wide, tile by tile)
                const bool eq = valid && key == thr;
                const unsigned long long bal = __ballot(eq);
                const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
                const unsigned before_lane = __popcll(bal & ((1ull << lane) - 1ull));
                if (lane == 0) s->wave_tot[wave] = (unsigned)__popcll(bal);
                __syncthreads();
                unsigned before_wave = 0, tile_tot = 0;
                for (int w = 0; w < kSelThreads / kWave; w++) {
                    if (w < wave) before_wave += s->wave_tot[w];
                    tile_tot += s->wave_tot[w];
                }
                const unsigned rank = s->running + before_wave + before_lane;
                if (eq && rank < need_eq) {
                    const unsigned at = atomicAdd(&s->count, 1u);
                    ent[at] = ((unsigned long long)key << 32) | (unsigned)i;
                }
                __syncthreads();
                if (threadIdx.x == 0) s->running += tile_tot;
                __syncthreads();
            }
        }
        __syncthreads();
    }

    sort_entries(ent, sort_n);
}

// The single-query path's selections (1000 center distances; ~12 000 tuple distances of a batch): the whole
// sequence is fetched ONCE as twelve 16-byte loads per thread, all in flight together -- one memory round
// trip instead of one per sweep iteration -- and both sweeps run from registers.  v must be 16-byte
// aligned with at least ((m + 3) & ~3) floats readable.  Same result as block_topk.
constexpr int kSmallVec = 12;
// ... and twenty for the segments past 12 288 (round 6: the headline's segments average 11.8 k rows, so two queries in
// five were over the old limit and took the general path's two sweeps from memory)
constexpr int kLargeVec = 20;
constexpr int kSmallMax = kLargeVec * 4 * kSelThreads;  // 20480
constexpr int kExtractMax = 16;  // k up to which the threshold comes from per-wavefront extraction

// NVEC float4 per thread: 1 (m <= 1024: the centers of a typical index), 4 (m <= 4096) or 12.  Returns false when the
// general path has to run instead (long runs of equal keys, or too few finite values).
// `skip` (0..3): v is the segment's start rounded DOWN to 16 bytes, its first `skip` floats belong to whoever is in front
// (they read as +inf and positions are counted from the segment's own first element); m counts them in.
template <int NVEC>
__device__ __forceinline__ bool block_topk_small(const float *v, int m, int skip, int k, int kp, int cap,
                                                 unsigned long long *ent, SelShared *s) {
    // A lone workgroup runs at whatever clock an otherwise idle chip grants: what counts here is the
    // number of instructions on the critical path (measured: ~300 instructions per microsecond), so the
    // sweeps compare floats (one v_min / v_cmp per value, keys only where a value is kept) and the rank
    // step compares unique 32-bit composites.
    for (int i = threadIdx.x; i < cap; i += kSelThreads) ent[i] = ~0ull;
    if (threadIdx.x == 0) s->count = 0;
    const int last4 = ((m + 3) & ~3) - 4;  // the last whole float4
    float4 r[NVEC];
#pragma unroll
    for (int b = 0; b < NVEC; b++) {
        const int at = b * 4 * kSelThreads + 4 * (int)threadIdx.x;
        r[b] = *reinterpret_cast<const float4 *>(v + (at < last4 ? at : last4));
    }
    // values past m read as +inf from here on; NaN never wins a float minimum (it sorts last anyway)
    float fmine = INFINITY;
#pragma unroll
    for (int b = 0; b < NVEC; b++) {
        const int at = b * 4 * kSelThreads + 4 * (int)threadIdx.x;
        if (at + 3 >= m) {  // the ragged end (and everything clamped past it)
            if (at + 0 >= m) r[b].x = INFINITY;
            if (at + 1 >= m) r[b].y = INFINITY;
            if (at + 2 >= m) r[b].z = INFINITY;
            if (at + 3 >= m) r[b].w = INFINITY;
        }
        if (b == 0 && at == 0) {  // the ragged front
            if (skip > 0) r[b].x = INFINITY;
            if (skip > 1) r[b].y = INFINITY;
            if (skip > 2) r[b].z = INFINITY;
        }
        fmine = fminf(fminf(fmine, fminf(r[b].x, r[b].y)), fminf(r[b].z, r[b].w));
    }
    // unique composite: the key with its low byte replaced by the thread id.  The thread of composite
    // rank k - 1 bounds k minima by (its key | 0xff): a valid (marginally looser) threshold.
    const unsigned key_mine = float_to_key(fmine);
    const unsigned comp = (key_mine & ~0xffu) | threadIdx.x;
    unsigned *mins = s->hist;
    if (k <= kExtractMax) {
        // small k: every wavefront pulls its k smallest composites out one at a time (a DPP minimum per round, ~15
        // instructions), then the 4k survivors are ranked -- the same composite of rank k - 1 as below for a
        // third of the instructions
        const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
        unsigned v = comp, keep = 0xffffffffu;
#pragma unroll 1
        for (int j = 0; j < k; j++) {
            const unsigned mn = wave_min_u32(v);
            keep = lane == j ? mn : keep;
            v = v == mn ? 0xffffffffu : v;
        }
        const int n4 = (kSelThreads / kWave) * k;
        __syncthreads();
        if (lane < k) mins[wave * k + lane] = keep;
        if ((int)threadIdx.x >= n4 && (int)threadIdx.x < ((n4 + 3) & ~3)) mins[threadIdx.x] = 0xffffffffu;
        __syncthreads();
        if ((int)threadIdx.x < n4) {
            const unsigned own = mins[threadIdx.x];
            const uint4 *m4 = reinterpret_cast<const uint4 *>(mins);
            unsigned rank = 0;
            for (int j4 = 0; j4 < (n4 + 3) / 4; j4++) {
                const uint4 o = m4[j4];
                rank += (o.x < own) + (o.y < own) + (o.z < own) + (o.w < own);
            }
            if (rank == (unsigned)(k - 1)) s->bin = own | 0xffu;
        }
    } else {
        __syncthreads();
        mins[threadIdx.x] = comp;
        __syncthreads();
        unsigned rank = 0;
        {
            const uint4 *m4 = reinterpret_cast<const uint4 *>(mins);
#pragma unroll 4
            for (int j4 = 0; j4 < kSelThreads / 4; j4++) {
                const uint4 o = m4[j4];
                rank += (o.x < comp) + (o.y < comp) + (o.z < comp) + (o.w < comp);
            }
        }
        if (rank == (unsigned)(k - 1)) s->bin = key_mine | 0xffu;
    }
    __syncthreads();
    const unsigned t0 = s->bin;
    // a finite threshold means k finite values exist at or below it: NaN and +inf entries (which the float
    // comparisons below never admit) cannot belong to the answer.  Otherwise: the general path.
    const bool general = t0 >= float_to_key(INFINITY);
    if (!general) {
        // count first (branch-free), reserve the slots with ONE atomic per thread that has any, then write:
        // 64 returning atomics on one LDS word, each inside a divergent branch, cost more than the sweep
        const float t0f = key_to_float(t0);
        unsigned cnt = 0;
#pragma unroll
        for (int b = 0; b < NVEC; b++)
            cnt += (r[b].x <= t0f) + (r[b].y <= t0f) + (r[b].z <= t0f) + (r[b].w <= t0f);
        if (cnt) {
            unsigned slot = atomicAdd(&s->count, cnt);
#pragma unroll
            for (int b = 0; b < NVEC; b++) {
                const int at = b * 4 * kSelThreads + 4 * (int)threadIdx.x;
                const float e[4] = {r[b].x, r[b].y, r[b].z, r[b].w};
                if (fminf(fminf(e[0], e[1]), fminf(e[2], e[3])) <= t0f) {
#pragma unroll
                    for (int c = 0; c < 4; c++)
                        if (e[c] <= t0f) {
                            if (slot < (unsigned)cap)
                                ent[slot] = ((unsigned long long)float_to_key(e[c]) << 32) | (unsigned)(at + c - skip);
                            slot++;
                        }
                }
            }
        }
    }
    __syncthreads();
    const unsigned got = s->count;
    if (!general && got <= (unsigned)kSelThreads) {
        rank_sort_entries(ent, (int)got);
    } else if (!general && got <= (unsigned)cap) {
        int sort_n = kp;
        while (sort_n < (int)got) sort_n <<= 1;
        sort_entries(ent, sort_n);
    } else {
        __syncthreads();
        return false;
    }
    return true;
}

// picks the register-resident form when it applies.  v may start anywhere (a segment of a packed stream): the vector
// loads start at the 16-byte boundary below it; at least 3 floats in front of a ragged v and (m + 3) & ~3 from the
// boundary on must be readable (every DBuf has the slack; a segment that starts a buffer is aligned).
__device__ __forceinline__ void block_topk_auto(const float *v, int64_t m, int k, int kp, int cap, unsigned long long *ent,
                                                SelShared *s) {
    bool done = false;  // block-uniform
    const int skip = (int)((reinterpret_cast<uintptr_t>(v) >> 2) & 3);
    const int64_t ma = m + skip;
    if (m > k && ma <= kSmallMax && k <= kSelThreads / 2) {
        const float *va = v - skip;
        if (ma <= 4 * kSelThreads)
            done = block_topk_small<1>(va, (int)ma, skip, k, kp, cap, ent, s);
        else if (ma <= 16 * kSelThreads)
            done = block_topk_small<4>(va, (int)ma, skip, k, kp, cap, ent, s);
        else if (ma <= kSmallVec * 4 * kSelThreads)
            done = block_topk_small<kSmallVec>(va, (int)ma, skip, k, kp, cap, ent, s);
        else
            done = block_topk_small<kLargeVec>(va, (int)ma, skip, k, kp, cap, ent, s);
    }
    if (!done) block_topk([v](int64_t i) { return v[i]; }, m, k, kp, cap, ent, s);
}

}  // namespace

}  // namespace pgv
