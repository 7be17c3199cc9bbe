// kernels_hnsw.hip -- hnswgettuple's first batch entirely on the device: the greedy
// descent through the upper layers and HnswSearchLayer on layer 0
// (src/hnswscan.c:25-56, src/hnswutils.c:824-987), one workgroup per query.
//
// The reference keeps two pairing heaps per layer search: C, the candidates still to
// expand (nearest first), and W, the ef nearest elements found (furthest first).
// Every element enters both at once (:936-960); W evicts its furthest when it
// overflows (:967-973); the search stops when the nearest unexpanded candidate is
// strictly farther than W's furthest (:894).  A candidate that has been evicted from
// W is at least as far as W's furthest from then on, so it can never be expanded
// before the search stops: C is, for the purposes of the result, "the entries of W
// not expanded yet".  Here W is therefore ONE ascending array in LDS whose entries
// carry an `expanded` flag:
//   next candidate  = first unexpanded entry (none left <=> the reference's loop ends)
//   neighbours      = the candidate's neighbour tuple at this layer, in tuple order,
//                     minus the visited ones (visited set: a bitmap in HBM)
//   scoring         = all unvisited neighbours at once, every wavefront of the
//                     workgroup taking rows (the distances do not depend on heap state)
//   admission       = a stable merge of the scored batch into W truncated to ef, which
//                     is exactly what admitting them one by one with the reference's
//                     `eDistance < f->distance || wlen < ef` test produces: an element
//                     rejected or evicted on the way would sit past position ef in
//                     the merged order as well.
// One refinement keeps this exact for distance ties: an element that was admitted and
// then evicted from W at a distance EQUAL to W's new furthest is still a live
// candidate in the reference (its stop test is strict).  Such elements go to a small
// tie list T (they all share one distance); T is dropped as soon as W's furthest
// gets nearer, and its members are expanded after W's own unexpanded entries.  The
// ORDER among equal distances is the one thing left open -- it is unspecified in the
// reference as well (pairing-heap internals).
#include "pgv_device.h"

#include <type_traits>

#include <cstdlib>

namespace pgv {

namespace {

constexpr int kHnswThreads = 256;
constexpr int kHnswWaves = kHnswThreads / kWave;
constexpr uint32_t kExpanded = 0x80000000u;
constexpr int kTieCap = 64;  // evicted candidates tied with W's furthest that are remembered

struct HnswDev {
    const char *rows;  // [n x nvec] 16-byte vectors
    int nvec, lpr_log2, nchunks;
    const int32_t *levels;     // [n]
    const int64_t *nbr_start;  // [n + 1]
    const int32_t *nbr;        // neighbour tuples, HnswNeighborTupleData layout
    int m;
    int32_t entry;
    int64_t n;
};

// One launch's queries and outputs.  A scan passes query vectors and wants the k nearest of
// layer 0.  The build (HnswFindElementNeighbors, src/hnswutils.c:1280-1357) passes element slots
// as queries plus each element's insert level: layers above it are descended greedily (ef = 1),
// layers at or below it are searched with ef_construction and their whole W is returned.
struct HnswRun {
    const char *queries;      // [nq x nvec] 16-byte vectors, or NULL with qids
    const int32_t *qids;      // query i = element qids[i] of the mirror (build), or NULL
    const int32_t *qlevels;   // insert level per query (build), or NULL = 0
    int nq, ef, k;
    int64_t *out_elem;        // [nq x k] or NULL
    float *out_dist;          // [nq x k] or NULL
    int64_t *out_scored;      // [nq] or NULL
    int32_t *lw_ids;          // [nq x lcap x ef] W of every searched layer <= the insert level, nearest first; or NULL
    float *lw_dist;           // [nq x lcap x ef]
    int32_t *lw_cnt;          // [nq x lcap] |W| (0 for layers not searched)
    int lcap;
    int rows_per_trip;        // rows a lane group scores at once: 1, 2 or 4 (PGV_HNSW_ROWS_PER_TRIP, for A/B runs)
};

template <typename T, int METRIC>
__global__ __launch_bounds__(kHnswThreads) void hnsw_search_kernel(
    HnswDev g, HnswRun run, uint32_t *__restrict__ bitmaps, int words, int *__restrict__ qcounter) {
    const int nq = run.nq, ef = run.ef, k = run.k;
    int64_t *__restrict__ out_elem = run.out_elem;
    float *__restrict__ out_dist = run.out_dist;
    int64_t *__restrict__ out_scored = run.out_scored;
    constexpr int N = VecTraits<T>::N;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lm0 = 2 * g.m;
    // layout: query row | W keys [2][ef] | W ids [2][ef] | batch keys [lm0] | batch ids [lm0] | scalars
    Raw16 *lq = reinterpret_cast<Raw16 *>(smem);
    uint32_t *wk = reinterpret_cast<uint32_t *>(smem + (size_t)g.nvec * sizeof(Raw16));
    uint32_t *wi = wk + 2 * ef;
    uint32_t *bk = wi + 2 * ef;
    int32_t *bi = reinterpret_cast<int32_t *>(bk + lm0);
    uint32_t *ti = reinterpret_cast<uint32_t *>(bi + lm0);  // tie list ids [kTieCap]
    int *sc = reinterpret_cast<int *>(ti + kTieCap);
    // sc: [0] query, [1] |W|, [2] batch size, [3] candidate element (-1: none), [4] W buffer,
    //     [5] |T|, [6] next unread entry of T, [7] key shared by T's entries

    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lpr = 1 << g.lpr_log2;
    const int sub = lane & (lpr - 1);
    const int rsub = lane >> g.lpr_log2;
    const int rpw = kWave >> g.lpr_log2;
    const size_t row_bytes = (size_t)g.nvec * sizeof(Raw16);
    uint32_t *bitmap = bitmaps + (size_t)blockIdx.x * words;

    // distance of the query (in LDS) to the rows bi[0 .. nb): every wavefront takes R x rpw rows per trip -- R rows' loads
    // in flight per lane group.  A walk is a chain of ~100 such batches, each waiting for its rows: with one row per trip
    // a batch of 32 neighbors was eight dependent round trips to HBM per wavefront (same box, 400 k x 1536, ef 100:
    // 352 k QPS with R = 1, 456 k with R = 2; 1 M x 1536: 366-403 k with R = 2, 385-413 k with R = 4, the default;
    // profiles/r05/hnsw_build_device_link.md).  Each row keeps its own accumulator and element order: the distances do not
    // depend on R.
    auto score_rows = [&](auto rc, int nb) __attribute__((always_inline)) {
        constexpr int R = decltype(rc)::value;
        const int step = kHnswWaves * rpw;
        for (int base = wave * rpw; base < nb; base += R * step) {
            int idx[R];
            bool valid[R];
            const char *rp[R];
            float acc[R];
#pragma unroll
            for (int r = 0; r < R; r++) {
                idx[r] = base + rsub + r * step;
                valid[r] = idx[r] < nb;
                rp[r] = g.rows + (size_t)bi[valid[r] ? idx[r] : nb - 1] * row_bytes;
                acc[r] = 0.f;
            }
#pragma unroll 2
            for (int c = 0; c < g.nchunks; c++) {
                const int vi = c * lpr + sub;
                const bool ok = vi < g.nvec;
                const int vc = ok ? vi : g.nvec - 1;  // never predicate a load (see scan_kernel)
                Raw16 rv[R];
#pragma unroll
                for (int r = 0; r < R; r++) rv[r] = load16(rp[r] + (size_t)vc * sizeof(Raw16));
                const Raw16 qv = lq[vc];
                Unpacked<T> uq(qv);
#pragma unroll
                for (int r = 0; r < R; r++) {
                    Unpacked<T> ur(rv[r]);
#pragma unroll
                    for (int e = 0; e < N; e++) acc[r] = accum<METRIC>(acc[r], ok ? ur.v[e] : 0.f, ok ? uq.v[e] : 0.f);
                }
            }
#pragma unroll
            for (int r = 0; r < R; r++) {
                const float sum = group_sum_to_last(acc[r], g.lpr_log2);
                if (sub == lpr - 1 && valid[r]) bk[idx[r]] = float_to_key(finish<METRIC>(sum));
            }
        }
    };
    // (always inlined: as a called function -- what the fp16 instantiations got -- every batch of a walk saved and
    // restored registers through 208 bytes of scratch)
    auto score_batch = [&](int nb) __attribute__((always_inline)) {
        if (run.rows_per_trip >= 4)
            score_rows(std::integral_constant<int, 4>{}, nb);
        else if (run.rows_per_trip == 2)
            score_rows(std::integral_constant<int, 2>{}, nb);
        else
            score_rows(std::integral_constant<int, 1>{}, nb);
    };

    for (;;) {
        if (tid == 0) sc[0] = atomicAdd(qcounter, 1);
        __syncthreads();
        const int qi = sc[0];
        if (qi >= nq) return;

        int64_t scored = 0;
        const int qlevel = run.qlevels ? run.qlevels[qi] : 0;
        if (run.lw_cnt)
            for (int l = tid; l < run.lcap; l += kHnswThreads) run.lw_cnt[(size_t)qi * run.lcap + l] = 0;
        if (g.entry < 0) {  // empty index
            if (out_elem)
                for (int j = tid; j < k; j += kHnswThreads) {
                    out_elem[(size_t)qi * k + j] = -1;
                    out_dist[(size_t)qi * k + j] = __uint_as_float(0x7f800000u);
                }
            if (tid == 0 && out_scored) out_scored[qi] = 0;
            __syncthreads();
            continue;
        }

        // query into LDS; the entry point is the first batch
        {
            const char *qrow = run.qids ? g.rows + (size_t)run.qids[qi] * row_bytes : run.queries + (size_t)qi * row_bytes;
            for (int v = tid; v < g.nvec; v += kHnswThreads) lq[v] = load16(qrow + (size_t)v * sizeof(Raw16));
        }
        if (tid == 0) bi[0] = g.entry;
        __syncthreads();
        score_batch(1);
        __syncthreads();
        if (tid == 0) {
            wk[0] = bk[0];
            wi[0] = (uint32_t)g.entry;
            sc[1] = 1;
            sc[4] = 0;
        }
        const int top = g.levels[g.entry];

        for (int lc = top; lc >= 0; lc--) {
            const int ef_l = lc <= qlevel ? ef : 1;
            const int lm = lc == 0 ? lm0 : g.m;
            // a fresh visited set holding the entry points (src/hnswutils.c:866-885); the entry
            // points are the previous layer's W, all unexpanded again
            for (int w = tid; w < words; w += kHnswThreads) bitmap[w] = 0u;
            __syncthreads();
            {
                const int cur = sc[4], wn = sc[1];
                for (int j = tid; j < wn; j += kHnswThreads) {
                    const uint32_t id = wi[cur * ef + j] & ~kExpanded;
                    wi[cur * ef + j] = id;
                    atomicOr(&bitmap[id >> 5], 1u << (id & 31));
                }
            }
            if (tid == 0) sc[5] = sc[6] = 0;
            __syncthreads();
            if (lc == 0) scored += sc[1];  // its entry points count too (src/hnswutils.c:872-873)

            for (;;) {
                // ---- wavefront 0: next candidate and its unvisited neighbours, in tuple order
                if (wave == 0) {
                    const int cur = sc[4], wn = sc[1];
                    int found = -1;
                    for (int base = 0; base < wn && found < 0; base += kWave) {
                        const int j = base + lane;
                        const bool open = j < wn && !(wi[cur * ef + j] & kExpanded);
                        const unsigned long long bal = __ballot(open);
                        if (bal) found = base + __ffsll((long long)bal) - 1;
                    }
                    int nb = 0;
                    int cand = -1;
                    if (found >= 0) {
                        cand = (int)wi[cur * ef + found];
                        if (lane == 0) wi[cur * ef + found] = (uint32_t)cand | kExpanded;
                    } else if (sc[6] < sc[5]) {  // W exhausted: an evicted candidate tied with its furthest
                        cand = (int)ti[sc[6]];
                        if (lane == 0) sc[6] = sc[6] + 1;
                    }
                    if (cand >= 0) {
                        const uint32_t c = (uint32_t)cand;
                        // neighbour tuple slice of layer lc (src/hnswutils.c:786)
                        const int64_t start = g.nbr_start[c] + (int64_t)(g.levels[c] - lc) * g.m;
                        for (int base = 0; base < lm; base += kWave) {
                            const int j = base + lane;
                            const int32_t e = j < lm ? g.nbr[start + j] : -1;
                            bool fresh = false;
                            if (e >= 0) {
                                const uint32_t bit = 1u << (e & 31);
                                fresh = !(atomicOr(&bitmap[e >> 5], bit) & bit);
                            }
                            const unsigned long long bal = __ballot(fresh);
                            if (fresh) bi[nb + __popcll(bal & ((1ull << lane) - 1ull))] = e;
                            nb += __popcll(bal);
                        }
                    }
                    // "make robust to issues" (src/hnswutils.c:947-949): an element below this layer
                    // is scored and stays visited but is never admitted; only upper layers can see one
                    if (lane == 0) {
                        sc[2] = nb;
                        sc[3] = cand;
                    }
                }
                __syncthreads();
                const int nb = sc[2];
                if (sc[3] < 0) break;  // every entry of W expanded: the reference's C is exhausted or too far
                if (nb == 0) {
                    __syncthreads();
                    continue;
                }
                if (lc == 0) scored += nb;  // so->tuples: only the layer-0 search counts (src/hnswscan.c:52-55)

                // ---- everyone: score the batch
                score_batch(nb);
                __syncthreads();

                // ---- wavefront 0: stable merge of the batch into W, truncated to ef_l
                if (wave == 0) {
                    const int cur = sc[4], nxt = cur ^ 1, wn = sc[1];
                    const uint32_t *ck = wk + cur * ef, *ci = wi + cur * ef;
                    uint32_t *nk = wk + nxt * ef, *ni = wi + nxt * ef;
                    int nbv = nb;
                    if (lc > 0) {  // drop batch entries that do not reach this layer, keeping the order
                        int kept = 0;
                        for (int base = 0; base < nb; base += kWave) {
                            const int i = base + lane;
                            const bool keep = i < nb && g.levels[bi[i < nb ? i : 0]] >= lc;
                            const uint32_t key = i < nb ? bk[i] : 0u;
                            const int32_t id = i < nb ? bi[i] : 0;
                            const unsigned long long bal = __ballot(keep);
                            const int at = kept + __popcll(bal & ((1ull << lane) - 1ull));
                            // in-place compaction is safe: a chunk only writes at or below its own reads
                            if (keep) {
                                bk[at] = key;
                                bi[at] = id;
                            }
                            kept += __popcll(bal);
                        }
                        nbv = kept;
                    }
                    const int nb = nbv;
                    const int wn_new = wn + nb < ef_l ? wn + nb : ef_l;
                    // old entries move up by the number of batch entries strictly nearer
                    for (int j = lane; j < wn; j += kWave) {
                        const uint32_t key = ck[j];
                        int s = 0;
                        for (int b = 0; b < nb; b++) s += bk[b] < key;
                        if (j + s < ef_l) {
                            nk[j + s] = key;
                            ni[j + s] = ci[j];
                        }
                    }
                    // batch entries go after every old entry that is not farther and after the
                    // batch entries that are nearer or equal-and-earlier
                    for (int i = lane; i < nb; i += kWave) {
                        const uint32_t key = bk[i];
                        int lo = 0, hi = wn;  // first old entry with key > this one
                        while (lo < hi) {
                            const int mid = (lo + hi) >> 1;
                            if (ck[mid] <= key)
                                lo = mid + 1;
                            else
                                hi = mid;
                        }
                        int s = lo;
                        for (int b = 0; b < nb; b++) s += (bk[b] < key) || (bk[b] == key && b < i);
                        if (s < ef_l) {
                            nk[s] = key;
                            ni[s] = (uint32_t)bi[i];
                        }
                    }
                    // The tie list.  With W full, its furthest key K is nk[ef_l - 1].  T keeps the
                    // unexpanded candidates that were pushed out of W at exactly K: old entries, and
                    // batch entries that had been admitted when their turn came (fewer than ef_l
                    // entries not farther among W and the batch entries before them).  Entries of an
                    // older T stay only if K has not moved.
                    if (wn_new == ef_l && wn + nb > ef_l) {
                        const uint32_t K = nk[ef_l - 1];
                        int tn = sc[5];
                        if (tn > 0 && (uint32_t)sc[7] != K) tn = 0;
                        const int t0 = tn > 0 ? sc[6] : 0;
                        for (int base = 0; base < wn; base += kWave) {
                            const int j = base + lane;
                            bool tie = false;
                            uint32_t id = 0;
                            if (j < wn && ck[j] == K && !(ci[j] & kExpanded)) {
                                int sft = 0;
                                for (int b = 0; b < nb; b++) sft += bk[b] < K;
                                tie = j + sft >= ef_l;
                                id = ci[j];
                            }
                            const unsigned long long bal = __ballot(tie);
                            const int at = tn + __popcll(bal & ((1ull << lane) - 1ull));
                            if (tie && at < kTieCap) ti[at] = id;
                            tn += __popcll(bal);
                        }
                        for (int base = 0; base < nb; base += kWave) {
                            const int i = base + lane;
                            bool tie = false;
                            if (i < nb && bk[i] == K) {
                                int lo = 0, hi = wn;
                                while (lo < hi) {
                                    const int mid = (lo + hi) >> 1;
                                    if (ck[mid] <= K)
                                        lo = mid + 1;
                                    else
                                        hi = mid;
                                }
                                int before = lo, pos = lo;
                                for (int b = 0; b < nb; b++) {
                                    before += b < i && bk[b] <= K;
                                    pos += (bk[b] < K) || (bk[b] == K && b < i);
                                }
                                tie = before < ef_l && pos >= ef_l;  // admitted in its turn, pushed out later
                            }
                            const unsigned long long bal = __ballot(tie);
                            const int at = tn + __popcll(bal & ((1ull << lane) - 1ull));
                            if (tie && at < kTieCap) ti[at] = (uint32_t)bi[i];
                            tn += __popcll(bal);
                        }
                        if (lane == 0) {
                            sc[5] = tn < kTieCap ? tn : kTieCap;
                            sc[6] = t0;
                            sc[7] = (int)K;
                        }
                    }
                    if (lane == 0) {
                        sc[1] = wn_new;
                        sc[4] = nxt;
                    }
                }
                __syncthreads();
            }
            __syncthreads();
            if (run.lw_ids && lc <= qlevel && lc < run.lcap) {  // this layer's W: the candidates SelectNeighbors sees
                const int cur = sc[4], wn = sc[1];
                const size_t o = ((size_t)qi * run.lcap + lc) * ef;
                for (int j = tid; j < wn; j += kHnswThreads) {
                    run.lw_ids[o + j] = (int32_t)(wi[cur * ef + j] & ~kExpanded);
                    run.lw_dist[o + j] = key_to_float(wk[cur * ef + j]);
                }
                if (tid == 0) run.lw_cnt[(size_t)qi * run.lcap + lc] = wn;
            }
        }

        // nearest first (src/hnswscan.c:293-311)
        if (out_elem) {
            const int cur = sc[4], wn = sc[1];
            for (int j = tid; j < k; j += kHnswThreads) {
                const bool have = j < wn;
                out_elem[(size_t)qi * k + j] = have ? (int64_t)(wi[cur * ef + j] & ~kExpanded) : -1;
                out_dist[(size_t)qi * k + j] = have ? key_to_float(wk[cur * ef + j]) : __uint_as_float(0x7f800000u);
            }
        }
        if (tid == 0 && out_scored) out_scored[qi] = scored;
        __syncthreads();
    }
}

// neighbour tuples of a few elements rewritten in place (the build's graph grows batch by batch);
// a tuple the caller sized wrongly or an element out of range is skipped
__global__ __launch_bounds__(256) void hnsw_patch_kernel(int32_t *__restrict__ nbr, const int64_t *__restrict__ nbr_start,
                                                          int64_t n, const int32_t *__restrict__ ids,
                                                          const int64_t *__restrict__ packed_off,
                                                          const int32_t *__restrict__ packed, int nupd) {
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= nupd) return;
    const int32_t e = ids[wave];
    if (e < 0 || e >= n) return;
    const int64_t dst = nbr_start[e], len = nbr_start[e + 1] - dst, src = packed_off[wave];
    if (packed_off[wave + 1] - src != len) return;
    for (int64_t j = lane; j < len; j += 64) nbr[dst + j] = packed[src + j];
}

template <typename T, int METRIC>
int launch_hnsw_t(pgv_ctx *ctx, const HnswDev &g, const HnswRun &run, uint32_t *bitmaps, int words, int grid,
                  int *counter) {
    const size_t lds = (size_t)g.nvec * sizeof(Raw16) + (size_t)run.ef * 16 + (size_t)g.m * 16 + kTieCap * 4 + 64;
    if (lds > 150 * 1024)
        PGV_FAIL(PGV_ERR_ARG, "hnsw search: ef %d / m %d need %zu bytes of LDS", run.ef, g.m, lds);
    auto kern = hnsw_search_kernel<T, METRIC>;
    PGV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kHnswThreads), lds, ctx->stream, g, run, bitmaps, words, counter);
    PGV_HIP(hipGetLastError());
    return PGV_OK;
}


// ------------------------------------------------------------------ SelectNeighbors for the elements being inserted
// HnswFindElementNeighbors ends each layer with SelectNeighbors(w, lm, ...) (src/hnswutils.c:1064-1165, Algorithm 4 of
// the HNSW paper with the reference's extras) over the layer's candidate list W, and CheckElementCloser (:1040-1059)
// inside it compares the candidate's distance to the element with its distances to the neighbors chosen so far.  For a
// NEW element the list has no cached `closer` flags (closerSet is false: every check is computed) and W arrives ordered
// by the search, so the selection is the plain greedy sweep, nearest candidate first:
//     closer(e) = no neighbor r chosen so far has d(e, r) <= d(e, q)          (ties are NOT closer: `<=` at :1053)
//     chosen while |r| < lm; the rejected ones ("pruned connections", :1146-1148) fill r up to lm, nearest first
// and a list of at most lm candidates is taken whole, in W's own (furthest first) order (:1072-1073).
//
// Three launches on the searches' stream: which lists need thinning and where their pair triangles go (one workgroup:
// an exclusive scan), the (u, v < u) slot pairs of those lists for score_gather_kernel, and -- after the scoring -- the
// sweep itself, one lane per list: its inner loop is a dependent chain of comparisons, the lists of a batch are
// thousands, and a wavefront's 64 lists share nothing, so lanes are the parallelism.

// pairs of list g = (query, layer): cnt * (cnt - 1) / 2 when it has to be thinned, else none; start[] by exclusive scan
// (256 threads: a larger workgroup waits for a whole CU's worth of room while another stream's searches fill the chip)
__global__ __launch_bounds__(256) void hnsw_select_plan_kernel(const int32_t *__restrict__ cnt, int ngroups, int lcap, int m,
                                                                int64_t *__restrict__ pair_start) {
    constexpr int T = 256, PER = 8;
    __shared__ int64_t wave_tot[T / 64];
    __shared__ int64_t carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int base = 0; base < ngroups; base += T * PER) {
        const int g0 = base + (int)threadIdx.x * PER;
        int64_t v[PER], mine = 0;
#pragma unroll
        for (int t = 0; t < PER; t++) {
            const int g = g0 + t;
            v[t] = 0;
            if (g < ngroups) {
                const int lm = (g % lcap) == 0 ? 2 * m : m;
                const int64_t c = cnt[g];
                v[t] = c > lm ? c * (c - 1) / 2 : 0;
            }
            mine += v[t];
        }
        int64_t incl = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int64_t t = __shfl_up(incl, d);
            if (lane >= d) incl += t;
        }
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        int64_t before = 0, total = 0;
        for (int w = 0; w < T / 64; w++) {
            const int64_t t = wave_tot[w];
            if (w < wave) before += t;
            total += t;
        }
        const int64_t carry = carry_s;
        int64_t run = carry + before + incl - mine;
#pragma unroll
        for (int t = 0; t < PER; t++) {
            if (g0 + t < ngroups) pair_start[g0 + t] = run;
            run += v[t];
        }
        __syncthreads();
        if (threadIdx.x == 0) carry_s = carry + total;
        __syncthreads();
    }
    if (threadIdx.x == 0) pair_start[ngroups] = carry_s;
}

// the pairs (u, v < u) of every list that is thinned, u ascending then v: a[] = slot of u, b[] = slot of v
__global__ __launch_bounds__(256) void hnsw_select_pairs_kernel(const int32_t *__restrict__ lw_ids, const int32_t *__restrict__ cnt,
                                                                 const int64_t *__restrict__ pair_start, int ngroups, int ef,
                                                                 int32_t *__restrict__ a, int32_t *__restrict__ b) {
    for (int g = blockIdx.x; g < ngroups; g += gridDim.x) {
        int64_t at = pair_start[g];
        if (pair_start[g + 1] == at) continue;
        const int32_t *gi = lw_ids + (size_t)g * ef;
        const int n = cnt[g];
        for (int u = 1; u < n; u++) {
            const int32_t iu = gi[u];
            for (int v = threadIdx.x; v < u; v += blockDim.x) {
                a[at + v] = iu;
                b[at + v] = gi[v];
            }
            at += u;
        }
    }
}

// the sweep: list g's candidates are lw_ids / lw_dist[g * ef ..), nearest first; tri = its pair distances.
// out_* [g * stride ..): the neighbors in the order the reference's r holds them, their distances, their closer flags
__global__ __launch_bounds__(64) void hnsw_select_kernel(const int32_t *__restrict__ lw_ids, const float *__restrict__ lw_dist,
                                                         const int32_t *__restrict__ cnt, const int32_t *__restrict__ qlevels,
                                                         const int64_t *__restrict__ pair_start, const float *__restrict__ tri_all,
                                                         int ngroups, int lcap, int ef, int m, int stride,
                                                         int32_t *__restrict__ out_ids, float *__restrict__ out_dist,
                                                         uint8_t *__restrict__ out_closer, int32_t *__restrict__ out_cnt) {
    const int g = blockIdx.x * 64 + threadIdx.x;
    if (g >= ngroups) return;
    const int lc = g % lcap, q = g / lcap;
    const int lm = lc == 0 ? 2 * m : m;
    const int nw = lc > qlevels[q] ? 0 : cnt[g];  // (layers above the insert level are descended, not linked)
    const int32_t *ids = lw_ids + (size_t)g * ef;
    const float *dist = lw_dist + (size_t)g * ef;
    int32_t *oi = out_ids + (size_t)g * stride;
    float *od = out_dist + (size_t)g * stride;
    uint8_t *oc = out_closer + (size_t)g * stride;
    if (nw <= lm) {
        // taken whole, in W's order: furthest first
        for (int i = 0; i < nw; i++) {
            oi[i] = ids[nw - 1 - i];
            od[i] = dist[nw - 1 - i];
            oc[i] = 0;
        }
        out_cnt[g] = nw;
        return;
    }
    const float *tri = tri_all + pair_start[g];
    int rn = 0;
    // the chosen ones' candidate indexes live in the output (their ids are written there anyway): oi[i] holds the
    // INDEX until the end, then the element
    int j = 0;
    for (; j < nw && rn < lm; j++) {
        const float de = dist[j];
        bool closer = true;
        for (int i = 0; i < rn; i++) {
            const int r = oi[i];  // r < j: chosen earlier, nearer or equal
            if (tri[(int64_t)j * (j - 1) / 2 + r] <= de) {
                closer = false;
                break;
            }
        }
        if (closer) oi[rn++] = j;
    }
    // the candidates looked at are 0 .. j - 1; the rejected among them, nearest first, fill r up to lm.  Walk them by
    // merging "all of 0 .. j - 1" against the chosen (ascending) indexes.
    const int chosen = rn;
    {
        int ci = 0;
        for (int x = 0; x < j && rn < lm; x++) {
            if (ci < chosen && oi[ci] == x) {
                ci++;
                continue;
            }
            oi[rn++] = x;
        }
    }
    for (int i = 0; i < rn; i++) {
        const int x = oi[i];
        od[i] = dist[x];
        oc[i] = i < chosen ? 1 : 0;
        oi[i] = ids[x];
    }
    out_cnt[g] = rn;
}


// the same sweep, one WAVEFRONT per list, for ef_construction <= 64: lane j holds candidate j, the list's pair triangle sits
// in LDS, and "no neighbor chosen so far is at distance <= mine" is one comparison per chosen lane and a ballot.  What
// stays sequential is the walk over the candidates, nearest first.
__global__ __launch_bounds__(256) void hnsw_select_wave_kernel(const int32_t *__restrict__ lw_ids, const float *__restrict__ lw_dist,
                                                               const int32_t *__restrict__ cnt, const int32_t *__restrict__ qlevels,
                                                               const int64_t *__restrict__ pair_start,
                                                               const float *__restrict__ tri_all, int ngroups, int lcap, int ef,
                                                               int m, int stride, int32_t *__restrict__ out_ids,
                                                               float *__restrict__ out_dist, uint8_t *__restrict__ out_closer,
                                                               int32_t *__restrict__ out_cnt) {
    __shared__ float tri_lds[4][64 * 63 / 2 + 64];  // (+ 64: the lanes that are not chosen index past their row harmlessly)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int g = blockIdx.x * 4 + wv;
    if (g >= ngroups) return;
    const int lc = g % lcap, q = g / lcap;
    const int lm = lc == 0 ? 2 * m : m;
    const int nw = lc > qlevels[q] ? 0 : cnt[g];
    int32_t *oi = out_ids + (size_t)g * stride;
    float *od = out_dist + (size_t)g * stride;
    uint8_t *oc = out_closer + (size_t)g * stride;
    const int32_t id = lane < nw ? lw_ids[(size_t)g * ef + lane] : -1;
    const float dist = lane < nw ? lw_dist[(size_t)g * ef + lane] : 0.f;
    if (nw <= lm) {
        // taken whole, in W's order: furthest first
        if (lane < nw) {
            oi[nw - 1 - lane] = id;
            od[nw - 1 - lane] = dist;
            oc[nw - 1 - lane] = 0;
        }
        if (lane == 0) out_cnt[g] = nw;
        return;
    }
    const float *tri_g = tri_all + pair_start[g];
    const int np = nw * (nw - 1) / 2;
    for (int i = lane; i < np; i += 64) tri_lds[wv][i] = tri_g[i];
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    bool chosen = false;
    int pos = -1, rn = 0, nrej = 0, rpos = -1;
    for (int j = 0; j < nw && rn < lm; j++) {
        const float dj = __shfl(dist, j);
        const bool decides = chosen && tri_lds[wv][j * (j - 1) / 2 + lane] <= dj;  // (chosen lanes are < j)
        if (__ballot(decides)) {
            if (lane == j) rpos = nrej;
            nrej++;
        } else {
            if (lane == j) {
                chosen = true;
                pos = rn;
            }
            rn++;
        }
    }
    // the rejected among the candidates looked at fill r up to lm, nearest first (:1146-1148)
    const int nchosen = rn;
    if (rpos >= 0 && nchosen + rpos < lm) pos = nchosen + rpos;
    const int total = nchosen + nrej < lm ? nchosen + nrej : lm;
    if (pos >= 0) {
        oi[pos] = id;
        od[pos] = dist;
        oc[pos] = chosen ? 1 : 0;
    }
    if (lane == 0) out_cnt[g] = total;
}

}  // namespace

int hnsw_search_grid(pgv_ctx *ctx, int nq, int64_t n, int *words_out) {
    const int words = (int)((n + 31) / 32) + 1;
    int grid = ctx->num_cus * 4;
    const int64_t cap = (int64_t)(1ll << 30) / ((int64_t)words * 4);  // visited bitmaps: <= 1 GiB
    if (grid > cap) grid = cap > 0 ? (int)cap : 1;
    if (grid > nq) grid = nq;
    *words_out = words;
    return grid;
}

int launch_hnsw_search(pgv_ctx *ctx, pgv_metric metric, pgv_dtype dtype, const RowGeom &geom, const void *rows,
                       int64_t n, const int32_t *levels, const int64_t *nbr_start, const int32_t *nbr, int m,
                       int32_t entry, const HnswSearchArgs &a, uint32_t *bitmaps, int words, int grid, int *counter) {
    HnswDev g;
    g.rows = static_cast<const char *>(rows);
    g.nvec = geom.nvec;
    g.lpr_log2 = geom.lpr_log2;
    g.nchunks = geom.nchunks;
    g.levels = levels;
    g.nbr_start = nbr_start;
    g.nbr = nbr;
    g.m = m;
    g.entry = entry;
    g.n = n;
    HnswRun run;
    run.queries = static_cast<const char *>(a.queries);
    run.qids = a.qids;
    run.qlevels = a.qlevels;
    run.nq = a.nq;
    run.ef = a.ef;
    run.k = a.k;
    run.out_elem = a.out_elem;
    run.out_dist = a.out_dist;
    run.out_scored = a.out_scored;
    run.lw_ids = a.lw_ids;
    run.lw_dist = a.lw_dist;
    run.lw_cnt = a.lw_cnt;
    run.lcap = a.lcap;
    {
        static const int per_trip = getenv("PGV_HNSW_ROWS_PER_TRIP") ? atoi(getenv("PGV_HNSW_ROWS_PER_TRIP")) : 4;
        run.rows_per_trip = per_trip >= 4 ? 4 : (per_trip == 1 ? 1 : 2);
    }
    PGV_HIP(hipMemsetAsync(counter, 0, sizeof(int), ctx->stream));
#define PGV_HNSW_M(T)                                                                      \
    switch (metric) {                                                                      \
        case PGV_L2SQ:                                                                     \
            return launch_hnsw_t<T, 0>(ctx, g, run, bitmaps, words, grid, counter);        \
        case PGV_NEG_IP:                                                                   \
            return launch_hnsw_t<T, 1>(ctx, g, run, bitmaps, words, grid, counter);        \
        case PGV_L1:                                                                       \
            return launch_hnsw_t<T, 2>(ctx, g, run, bitmaps, words, grid, counter);        \
    }
    if (dtype == PGV_F32) {
        PGV_HNSW_M(float)
    } else {
        PGV_HNSW_M(__half)
    }
#undef PGV_HNSW_M
    PGV_FAIL(PGV_ERR_ARG, "hnsw search: unknown metric %d", (int)metric);
}

int launch_hnsw_patch(pgv_ctx *ctx, int32_t *nbr, const int64_t *nbr_start, int64_t n, const int32_t *ids,
                      const int64_t *packed_off, const int32_t *packed, int nupd) {
    if (nupd <= 0) return PGV_OK;
    hipLaunchKernelGGL(hnsw_patch_kernel, dim3((nupd + 3) / 4), dim3(256), 0, ctx->stream, nbr, nbr_start, n, ids,
                       packed_off, packed, nupd);
    PGV_HIP(hipGetLastError());
    return PGV_OK;
}

// pgv_hnsw_score_groups: the (u, v) pairs of every group written out as the slot arrays score_gather_kernel reads.
// One workgroup per group (grid-stride), a row u per step, the row's v spread over the threads (coalesced).
__global__ __launch_bounds__(256) void expand_groups_kernel(const int32_t *__restrict__ ids,
                                                            const int64_t *__restrict__ ids_start,
                                                            const int32_t *__restrict__ from,
                                                            const int64_t *__restrict__ pair_start, int ngroups,
                                                            int32_t *__restrict__ a, int32_t *__restrict__ b) {
    for (int g = blockIdx.x; g < ngroups; g += gridDim.x) {
        const int32_t *gi = ids + ids_start[g];
        const int n = (int)(ids_start[g + 1] - ids_start[g]);
        const int f = from[g] < 1 ? 1 : from[g];
        int64_t at = pair_start[g];
        for (int u = f; u < n; u++) {
            const int32_t iu = gi[u];
            for (int v = threadIdx.x; v < u; v += blockDim.x) {
                a[at + v] = iu;
                b[at + v] = gi[v];
            }
            at += u;
        }
    }
}

int launch_expand_groups(pgv_ctx *ctx, const int32_t *ids, const int64_t *ids_start, const int32_t *from,
                         const int64_t *pair_start, int ngroups, int32_t *a, int32_t *b) {
    if (ngroups <= 0) return PGV_OK;
    const int cap = ctx->num_cus * 16;
    hipLaunchKernelGGL(expand_groups_kernel, dim3(ngroups < cap ? ngroups : cap), dim3(256), 0, ctx->stream, ids,
                       ids_start, from, pair_start, ngroups, a, b);
    PGV_HIP(hipGetLastError());
    return PGV_OK;
}

// SelectNeighbors of a batch's new elements (see hnsw_select_kernel): plan -> *out_total_dev pairs at pair_start[ngroups]
int launch_hnsw_select_plan(pgv_ctx *ctx, const int32_t *cnt, int ngroups, int lcap, int m, int64_t *pair_start) {
    if (ngroups <= 0) return PGV_OK;
    hipLaunchKernelGGL(hnsw_select_plan_kernel, dim3(1), dim3(256), 0, ctx->stream, cnt, ngroups, lcap, m, pair_start);
    PGV_HIP(hipGetLastError());
    return PGV_OK;
}

int launch_hnsw_select_pairs(pgv_ctx *ctx, const int32_t *lw_ids, const int32_t *cnt, const int64_t *pair_start, int ngroups,
                             int ef, int32_t *a, int32_t *b) {
    if (ngroups <= 0) return PGV_OK;
    const int cap = ctx->num_cus * 16;
    hipLaunchKernelGGL(hnsw_select_pairs_kernel, dim3(ngroups < cap ? ngroups : cap), dim3(256), 0, ctx->stream, lw_ids, cnt,
                       pair_start, ngroups, ef, a, b);
    PGV_HIP(hipGetLastError());
    return PGV_OK;
}

int launch_hnsw_select(pgv_ctx *ctx, const int32_t *lw_ids, const float *lw_dist, const int32_t *cnt, const int32_t *qlevels,
                       const int64_t *pair_start, const float *tri, int ngroups, int lcap, int ef, int m, int stride,
                       int32_t *out_ids, float *out_dist, uint8_t *out_closer, int32_t *out_cnt) {
    if (ngroups <= 0) return PGV_OK;
    static const bool serial = getenv("PGV_HNSW_SELECT_SERIAL") && atoi(getenv("PGV_HNSW_SELECT_SERIAL")) != 0;
    if (ef <= 64 && !serial) {
        hipLaunchKernelGGL(hnsw_select_wave_kernel, dim3((ngroups + 3) / 4), dim3(256), 0, ctx->stream, lw_ids, lw_dist, cnt,
                           qlevels, pair_start, tri, ngroups, lcap, ef, m, stride, out_ids, out_dist, out_closer, out_cnt);
        PGV_HIP(hipGetLastError());
        return PGV_OK;
    }
    hipLaunchKernelGGL(hnsw_select_kernel, dim3((ngroups + 63) / 64), dim3(64), 0, ctx->stream, lw_ids, lw_dist, cnt, qlevels,
                       pair_start, tri, ngroups, lcap, ef, m, stride, out_ids, out_dist, out_closer, out_cnt);
    PGV_HIP(hipGetLastError());
    return PGV_OK;
}

}  // namespace pgv
