// kernels_hnsw_link.hip -- the in-memory HNSW build's graph updates on the device: a batch of new elements linked into
// the neighbor lists they chose (HnswUpdateNeighborsInMemory -> HnswUpdateConnection, src/hnswbuild.c:376-405,
// src/hnswutils.c:1183-1231), every list replayed by one lane with the reference's SelectNeighbors (hnsw_link_core.h).
//
// The graph state lives next to the mirror's neighbor tuples (pgv_hnsw::nbr, the array the searches read): for every
// tuple slot the neighbor's distance to the owner (nb_dist) and its cached `closer` flag (nb_flag bit 0; bit 1 of a
// list's first slot = the list's closerSet).  A list's length is its run of non-negative slots.
//
//   group     the batch's link requests (new element q chose neighbor `owner` on layer lc) grouped by list: a counter per
//             list (slot position / m: every list starts at a multiple of m), the first request of a list makes its
//             record; a second sweep files each request under its record; the record's newcomers are then put in heap
//             order -- the order the reference's loop links them in -- whatever order the atomics ran in
//   prepare   per record (owner, layer, newcomers): where the list is, its members, which pair distances its
//             selections can look up -- the whole triangle when the list has no cached flags, otherwise the pairs that
//             involve a newcomer (the rule of host/hnsw_build.c step 4) -- sizes, two scans, then the id lists and the
//             (u, v) slot pairs for score_gather_kernel
//   replay    one lane per record: appends while the list has room, SelectNeighbors + replace after; a replay that
//             needs a member-member pair that was not fetched stops there (rec_wait) and is continued by a second
//             launch once those triangles are scored
//   new       the batch's own elements' lists (SelectNeighbors ran with their searches: hnsw_select_kernel) into place
#include "pgv_device.h"

#include <cstdlib>

#define PGV_LINK_FN __device__ __forceinline__
#include "hnsw_link_core.h"

namespace pgv {

namespace {

__device__ __forceinline__ int64_t group_pairs_dev(int n, int from) {
    if (from < 1) from = 1;
    return n > from ? ((int64_t)n * (n - 1) - (int64_t)from * (from - 1)) / 2 : 0;
}

// request (q, lc, i): element elems[q] chose sel_ids[(q * lcap + lc) * 2m + i] as a neighbor on layer lc
__device__ __forceinline__ bool link_request(int t, int nq, int lcap, int stride, const int32_t *elems, const uint8_t *linked,
                                             const int32_t *levels, const int32_t *sel_cnt, int *q, int *lc, int *i) {
    const int g = t / stride;
    *i = t - g * stride;
    *q = g / lcap;
    *lc = g - *q * lcap;
    if (*q >= nq || !linked[*q]) return false;
    return *lc <= levels[elems[*q]] && *i < sel_cnt[g];
}

__global__ __launch_bounds__(256) void hnsw_link_count_kernel(const int32_t *__restrict__ elems, const uint8_t *__restrict__ linked,
                                                               int nq, int lcap, int m, const int32_t *__restrict__ sel_ids,
                                                               const int32_t *__restrict__ sel_cnt,
                                                               const int32_t *__restrict__ levels,
                                                               const int64_t *__restrict__ nbr_start, int *__restrict__ list_count,
                                                               int *__restrict__ list_rec, int *__restrict__ nrec,
                                                               int32_t *__restrict__ rec_owner, int32_t *__restrict__ rec_lc,
                                                               int32_t *__restrict__ rec_list) {
    const int stride = 2 * m;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    int q, lc, i;
    if (t >= nq * lcap * stride || !link_request(t, nq, lcap, stride, elems, linked, levels, sel_cnt, &q, &lc, &i)) return;
    const int owner = sel_ids[t];
    const int lidx = (int)((nbr_start[owner] + (int64_t)(levels[owner] - lc) * m) / m);
    if (atomicAdd(&list_count[lidx], 1) == 0) {
        const int rec = atomicAdd(nrec, 1);
        list_rec[lidx] = rec;
        rec_owner[rec] = owner;
        rec_lc[rec] = lc;
        rec_list[rec] = lidx;
    }
}

__global__ __launch_bounds__(256) void hnsw_link_fill_kernel(const int32_t *__restrict__ elems, const uint8_t *__restrict__ linked,
                                                              int nq, int lcap, int m, const int32_t *__restrict__ sel_ids,
                                                              const float *__restrict__ sel_dist,
                                                              const int32_t *__restrict__ sel_cnt,
                                                              const int32_t *__restrict__ levels,
                                                              const int64_t *__restrict__ nbr_start,
                                                              const int *__restrict__ list_rec, const int64_t *__restrict__ rec_off,
                                                              int *__restrict__ rec_fill, int32_t *__restrict__ link_elem,
                                                              float *__restrict__ link_dist) {
    const int stride = 2 * m;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    int q, lc, i;
    if (t >= nq * lcap * stride || !link_request(t, nq, lcap, stride, elems, linked, levels, sel_cnt, &q, &lc, &i)) return;
    const int owner = sel_ids[t];
    const int lidx = (int)((nbr_start[owner] + (int64_t)(levels[owner] - lc) * m) / m);
    const int rec = list_rec[lidx];
    const int64_t at = rec_off[rec] + atomicAdd(&rec_fill[rec], 1);
    link_elem[at] = elems[q];
    link_dist[at] = sel_dist[t];
}

// pass 0: every record; pass 1: the records whose replay stopped (rec_wait < nlocal) ask for their member triangles
__global__ __launch_bounds__(256) void hnsw_link_size_kernel(const int32_t *__restrict__ nbr, const uint8_t *__restrict__ nb_flag,
                                                              const int32_t *__restrict__ levels,
                                                              const int64_t *__restrict__ nbr_start, int m,
                                                              const int32_t *__restrict__ rec_owner,
                                                              const int32_t *__restrict__ rec_lc,
                                                              const int32_t *__restrict__ rec_list,
                                                              const int *__restrict__ list_count,
                                                              int64_t *__restrict__ rec_off, int nrec, int pass,
                                                              int64_t *__restrict__ rec_pos, int32_t *__restrict__ rec_nstart,
                                                              int32_t *__restrict__ rec_from, const int32_t *__restrict__ rec_wait,
                                                              int64_t *__restrict__ size_ids, int64_t *__restrict__ size_pairs) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nrec) return;
    if (pass == 1) {
        const int nstart = rec_nstart[k];
        const int nlocal = nstart + (int)(rec_off[k + 1] - rec_off[k]);
        size_pairs[k] = rec_wait[k] < nlocal ? group_pairs_dev(nstart, 1) : 0;
        return;
    }
    const int owner = rec_owner[k], lc = rec_lc[k];
    const int lm = lc == 0 ? 2 * m : m;
    const int64_t pos = nbr_start[owner] + (int64_t)(levels[owner] - lc) * m;
    int nstart = 0;
    while (nstart < lm && nbr[pos + nstart] >= 0) nstart++;
    const bool closer_set = (nb_flag[pos] & 2) != 0;
    const int nnew = list_count[rec_list[k]];  // (pass 0: rec_off is this launch's to size -- the scan makes it offsets)
    const int nlocal = nstart + nnew;
    rec_off[k] = nnew;
    // cached flags: only the pairs that involve a newcomer; none: the list's next selection computes everything
    int from = closer_set ? nstart : 1;
    if (from < 1) from = 1;
    rec_pos[k] = pos;
    rec_nstart[k] = nstart;
    rec_from[k] = from;
    size_ids[k] = nlocal;
    size_pairs[k] = nlocal > lm ? group_pairs_dev(nlocal, from) : 0;  // a list that cannot overflow runs no selection
}

// exclusive scans of up to three int64 sequences in place (one workgroup); totals[0 .. 2] = their sums.  A thread takes
// 16 consecutive values per trip (4 K values a trip: a batch's records are a few tens of thousands).  256 threads, not
// 1024: the searches of the next batch fill the chip while this runs, and a workgroup of sixteen wavefronts waits until
// one CU has room for all of them -- measured 435 us a launch, most of it waiting to start
constexpr int kLinkScanThreads = 256;
__global__ __launch_bounds__(kLinkScanThreads) void hnsw_link_scan_kernel(int64_t *__restrict__ a, int64_t *__restrict__ b,
                                                               int64_t *__restrict__ c, int n, int64_t *__restrict__ totals) {
    constexpr int PER = 16;
    __shared__ int64_t wave_tot[kLinkScanThreads / 64];
    __shared__ int64_t carry_s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int which = 0; which < 3; which++) {
        int64_t *x = which == 0 ? a : (which == 1 ? b : c);
        if (!x) continue;
        if (threadIdx.x == 0) carry_s = 0;
        __syncthreads();
        for (int base = 0; base < n; base += kLinkScanThreads * PER) {
            const int i0 = base + (int)threadIdx.x * PER;
            int64_t v[PER];
            int64_t mine = 0;
#pragma unroll
            for (int t = 0; t < PER; t++) {
                v[t] = i0 + t < n ? x[i0 + t] : 0;
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
            for (int w = 0; w < kLinkScanThreads / 64; w++) {
                const int64_t t = wave_tot[w];
                if (w < wave) before += t;
                total += t;
            }
            const int64_t carry = carry_s;
            int64_t run = carry + before + incl - mine;
#pragma unroll
            for (int t = 0; t < PER; t++) {
                if (i0 + t < n) x[i0 + t] = run;
                run += v[t];
            }
            __syncthreads();
            if (threadIdx.x == 0) carry_s = carry + total;
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            x[n] = carry_s;
            totals[which] = carry_s;
        }
        __syncthreads();
    }
}

// pass 0: the record's id list (members in slot order, then the newcomers in link order) and its pairs (u >= from, v < u);
// pass 1: the member triangle (u < nstart) of the records that wait
__global__ __launch_bounds__(256) void hnsw_link_pairs_kernel(const int32_t *__restrict__ nbr, const int64_t *__restrict__ rec_pos,
                                                               const int32_t *__restrict__ rec_nstart,
                                                               const int32_t *__restrict__ rec_from,
                                                               const int64_t *__restrict__ rec_off,
                                                               int32_t *__restrict__ link_elem, float *__restrict__ link_dist,
                                                               const int32_t *__restrict__ rec_list, int *__restrict__ list_count,
                                                               int nrec, int pass,
                                                               const int64_t *__restrict__ ids_start, int32_t *__restrict__ ids,
                                                               const int64_t *__restrict__ pair_start, int32_t *__restrict__ a,
                                                               int32_t *__restrict__ b) {
    for (int k = blockIdx.x; k < nrec; k += gridDim.x) {
        const int nstart = rec_nstart[k];
        int32_t *gi = ids + ids_start[k];
        int64_t at = pair_start[k];
        const int64_t np = pair_start[k + 1] - at;
        if (pass == 0) {
            const int nnew = (int)(rec_off[k + 1] - rec_off[k]);
            const int64_t pos = rec_pos[k];
            if (threadIdx.x == 0) {
                // the newcomers in heap order (the reference links them in that order; the atomics filed them in any):
                // ids of one batch, a handful per list
                int32_t *ne = link_elem + rec_off[k];
                float *nd = link_dist + rec_off[k];
                for (int x = 1; x < nnew; x++) {
                    const int32_t e = ne[x];
                    const float d = nd[x];
                    int y = x;
                    while (y > 0 && ne[y - 1] > e) {
                        ne[y] = ne[y - 1];
                        nd[y] = nd[y - 1];
                        y--;
                    }
                    ne[y] = e;
                    nd[y] = d;
                }
                list_count[rec_list[k]] = 0;  // (the table is all zeros again when the batch is through)
            }
            __syncthreads();
            for (int j = threadIdx.x; j < nstart; j += blockDim.x) gi[j] = nbr[pos + j];
            for (int j = threadIdx.x; j < nnew; j += blockDim.x) gi[nstart + j] = link_elem[rec_off[k] + j];
            if (np == 0 || !a) {  // (no pairs wanted, or the tiled scoring reads the id lists itself)
                __syncthreads();
                continue;
            }
            __syncthreads();
            const int n = nstart + nnew;
            for (int u = rec_from[k]; u < n; u++) {
                const int32_t iu = gi[u];
                for (int v = threadIdx.x; v < u; v += blockDim.x) {
                    a[at + v] = iu;
                    b[at + v] = gi[v];
                }
                at += u;
            }
            __syncthreads();
        } else {
            if (np == 0) continue;
            for (int u = 1; u < nstart; u++) {
                const int32_t iu = gi[u];
                for (int v = threadIdx.x; v < u; v += blockDim.x) {
                    a[at + v] = iu;
                    b[at + v] = gi[v];
                }
                at += u;
            }
        }
    }
}

struct LinkArgs {
    int32_t *nbr;
    float *nb_dist;
    uint8_t *nb_flag;
    int m, nrec, pass;
    const int32_t *rec_lc;
    const int64_t *rec_off;
    const float *link_dist;
    const int64_t *rec_pos;
    const int32_t *rec_nstart, *rec_from;
    const int64_t *ids_start;
    const int32_t *ids;
    const int64_t *pair_start;
    const float *tri;
    const int64_t *mm_start;  // pass 1
    const float *mm;
    int32_t *rec_wait;   // [nrec] the first local not linked yet (nlocal: done)
    int16_t *loc_save;   // [nrec x (2m + 1)]
    int *blocked;        // records that stopped in this launch
};

template <int CAP>
__global__ __launch_bounds__(64) void hnsw_link_kernel(LinkArgs g) {
    const int k = blockIdx.x * 64 + threadIdx.x;
    if (k >= g.nrec) return;
    const int nstart = g.rec_nstart[k];
    const int nlocal = (int)(g.ids_start[k + 1] - g.ids_start[k]);
    int first = nstart;
    if (g.pass == 1) {
        first = g.rec_wait[k];
        if (first >= nlocal) return;
    }
    const int lm = g.rec_lc[k] == 0 ? 2 * g.m : g.m;
    const int64_t pos = g.rec_pos[k];
    int32_t le[CAP + 1];
    float ld[CAP + 1];
    uint8_t lf[CAP + 1];
    int16_t loc[CAP + 1];
    uint64_t key[CAP + 1];
    uint8_t scratch[5 * (CAP + 1)];
    int len = 0;
    while (len < lm && g.nbr[pos + len] >= 0) {
        le[len] = g.nbr[pos + len];
        ld[len] = g.nb_dist[pos + len];
        lf[len] = g.nb_flag[pos + len] & 1;
        len++;
    }
    uint8_t closer_set = (g.nb_flag[pos] >> 1) & 1;
    int16_t *save = g.loc_save + (size_t)k * (2 * g.m + 1);
    if (g.pass == 0)
        for (int j = 0; j < len; j++) loc[j] = (int16_t)j;
    else
        for (int j = 0; j < len; j++) loc[j] = save[j];
    pgv_link_pairs ps;
    ps.tri = g.tri + g.pair_start[k];
    ps.from = g.rec_from[k];
    ps.base = ps.from * (ps.from - 1) / 2;
    ps.mm = g.pass == 1 ? g.mm + g.mm_start[k] : nullptr;
    const int stop = pgv_link_replay(le, ld, lf, loc, &len, &closer_set, lm, g.ids + g.ids_start[k],
                                     g.link_dist + g.rec_off[k], nstart, nlocal, first, &ps, key, scratch);
    for (int j = 0; j < len; j++) {
        g.nbr[pos + j] = le[j];
        g.nb_dist[pos + j] = ld[j];
        g.nb_flag[pos + j] = (uint8_t)(lf[j] | (j == 0 ? (closer_set << 1) : 0));
    }
    g.rec_wait[k] = stop;
    if (stop < nlocal) {
        atomicAdd(g.blocked, 1);
        for (int j = 0; j < len; j++) save[j] = loc[j];
    }
}

// The same replay, one WAVEFRONT per record, for lists of up to 63 entries (m <= 31): lane i holds candidate i of the
// list (element, distance, closer flag, local), the newcomer sits in lane lm.  What is sequential in SelectNeighbors --
// candidates are looked at nearest first, each decision depends on the ones before -- stays a loop; what is not runs
// across the lanes: the sort is a rank (every lane counts the keys above its own), CheckElementCloser against the
// neighbors chosen so far is one comparison per lane and a ballot, the record's pair distances are read from LDS, where
// the wavefront put them in one coalesced sweep.  Decisions, flags, the dropped candidate and the slot the newcomer takes
// are hnsw_link_core.h's, step for step (tests: the graphs of the two kernels and of the host replay are the same).  One
// difference that changes nothing: a check that meets BOTH a deciding pair and a pair that was not fetched stops for
// the missing one here, whatever their order in the reference's loop -- the list then waits for its member triangle and
// comes to the same decision.
constexpr int kLinkTriCap = 1024;  // pair distances of a record kept in LDS (a full 33-candidate triangle is 528)

__global__ __launch_bounds__(256) void hnsw_link_wave_kernel(LinkArgs g) {
    __shared__ float tri_lds[4][kLinkTriCap];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int k = blockIdx.x * 4 + wv;
    if (k >= g.nrec) return;
    const int nstart = g.rec_nstart[k];
    const int nlocal = (int)(g.ids_start[k + 1] - g.ids_start[k]);
    int first = nstart;
    if (g.pass == 1) {
        first = g.rec_wait[k];
        if (first >= nlocal) return;
    }
    const int lm = g.rec_lc[k] == 0 ? 2 * g.m : g.m;
    const int64_t pos = g.rec_pos[k];
    // the list: lane j holds item j
    int32_t ce = lane < lm ? g.nbr[pos + lane] : -1;
    int len = __popcll(__ballot(ce >= 0));  // (a list is a run of non-negative slots)
    float cd = lane < len ? g.nb_dist[pos + lane] : 0.f;
    const uint8_t raw = lane < len ? g.nb_flag[pos + lane] : 0;
    int cf = raw & 1;
    int closer_set = (__shfl((int)raw, 0) >> 1) & 1;
    int16_t *save = g.loc_save + (size_t)k * (2 * g.m + 1);
    int loc = g.pass == 0 ? lane : (lane < len ? (int)save[lane] : 0);
    // the record's pair distances: LDS when they fit
    const int64_t np = g.pair_start[k + 1] - g.pair_start[k];
    const float *tri_g = g.tri + g.pair_start[k];
    const bool tri_in_lds = np <= kLinkTriCap;
    if (tri_in_lds)
        for (int i = lane; i < (int)np; i += 64) tri_lds[wv][i] = tri_g[i];
    const float *mm = g.pass == 1 ? g.mm + g.mm_start[k] : nullptr;
    const int from = g.rec_from[k], base = from * (from - 1) / 2;
    const int32_t *ids = g.ids + g.ids_start[k];
    const float *newdist = g.link_dist + g.rec_off[k];
    __builtin_amdgcn_s_waitcnt(0xc07f);  // the LDS copy has landed before the first lookup (one wavefront: no barrier)
    __builtin_amdgcn_wave_barrier();

    int stop = nlocal;
    for (int u = first; u < nlocal; u++) {
        const int32_t ne = ids[u];
        const float nd = newdist[u - nstart];
        if (len < lm) {
            if (lane == len) {
                ce = ne;
                cd = nd;
                cf = 0;
                loc = u;
            }
            len++;
            continue;
        }
        // the newcomer as candidate lm
        const int nc = lm + 1;
        if (lane == lm) {
            ce = ne;
            cd = nd;
            cf = 0;
            loc = u;
        }
        const bool active = lane < nc;
        // CompareCandidateDistances as one key; rank = candidates before this one in the (descending) order
        uint64_t key = 0;
        {
            float d0 = cd + 0.0f;
            uint32_t ub = __float_as_uint(d0);
            ub ^= (ub >> 31) ? 0xFFFFFFFFu : 0x80000000u;
            key = ((uint64_t)ub << 32) | (uint32_t)ce;
        }
        int rank = 0;
        for (int j = 0; j < nc; j++) {
            const uint64_t kj = ((uint64_t)(uint32_t)__shfl((int)(key >> 32), j) << 32) | (uint32_t)__shfl((int)(key & 0xffffffffu), j);
            rank += kj > key ? 1 : 0;
        }
        const bool must_calculate = !closer_set;
        bool in_r = false, in_added = false, processed = false;
        int rn = 0, nadded = 0, nrej = 0, rej_idx = -1, newflag = cf;
        bool removed_any = false, missing = false;
        for (int p = nc - 1; p >= 0 && rn < lm; p--) {
            const int e = __ffsll((unsigned long long)__ballot(active && rank == p)) - 1;  // the closest remaining
            const float de = __shfl(cd, e);
            const int le = __shfl(loc, e);
            int closer = __shfl(cf, e);
            int which = 0;  // 0: the cached flag stands; 1: check against r; 2: check against added
            if (must_calculate)
                which = 1;
            else if (nadded > 0) {
                if (closer)
                    which = 2;
                else if (removed_any)
                    which = 1;
            } else if (e == lm)
                which = 1;
            if (which) {
                const bool member = which == 1 ? in_r : in_added;
                bool miss = false, decides = false;
                if (member) {
                    const int hi = le > loc ? le : loc, lo = le > loc ? loc : le;
                    float d = 0.f;
                    if (hi >= from) {
                        const int at = hi * (hi - 1) / 2 - base + lo;
                        d = tri_in_lds ? tri_lds[wv][at] : tri_g[at];
                    } else if (mm)
                        d = mm[hi * (hi - 1) / 2 + lo];
                    else
                        miss = true;
                    decides = !miss && d <= de;
                }
                if (__ballot(miss)) {
                    missing = true;
                    break;
                }
                const int was = closer;
                closer = __ballot(decides) ? 0 : 1;
                if (which == 2) {
                    if (!closer) removed_any = true;
                } else if (!must_calculate) {
                    // (checked against r because it was a reject after a removal, or the newcomer)
                    if (closer) {
                        if (lane == e) in_added = true;
                        nadded++;
                    }
                }
                (void)was;
            }
            if (lane == e) {
                processed = true;
                newflag = closer;
                if (closer)
                    in_r = true;
                else
                    rej_idx = nrej;
            }
            if (closer)
                rn++;
            else
                nrej++;
        }
        if (missing) {
            stop = u;  // nothing has been changed: this newcomer and the later ones wait for the member triangle
            break;
        }
        if (processed) cf = newflag;
        closer_set = 1;
        // r is filled up with the rejected in the order they were rejected; the first one left over is dropped, or --
        // none left -- the furthest candidate (:1146-1157)
        const int want = lm - rn;
        unsigned long long dm = __ballot(active && rej_idx == want);
        if (!dm) dm = __ballot(active && rank == 0);
        const int dropped = __ffsll(dm) - 1;
        if (dropped != lm) {
            // the list keeps its members' places, the newcomer takes the dropped one's (:1211-1227)
            const int32_t e2 = __shfl(ce, lm);
            const float d2 = __shfl(cd, lm);
            const int f2 = __shfl(cf, lm);
            const int l2 = __shfl(loc, lm);
            if (lane == dropped) {
                ce = e2;
                cd = d2;
                cf = f2;
                loc = l2;
            }
        }
    }
    if (lane < len) {
        g.nbr[pos + lane] = ce;
        g.nb_dist[pos + lane] = cd;
        g.nb_flag[pos + lane] = (uint8_t)(cf | (lane == 0 ? (closer_set << 1) : 0));
    }
    if (lane == 0) g.rec_wait[k] = stop;
    if (stop < nlocal) {
        if (lane == 0) atomicAdd(g.blocked, 1);
        if (lane < len) save[lane] = (int16_t)loc;
    }
}

// a batch's counts added to the build's: lists that needed the second round, member pairs scored for them, updates that
// were STILL waiting after it (must stay 0)
__global__ void hnsw_link_stats_kernel(const int *__restrict__ blocked, const int64_t *__restrict__ totals,
                                       int64_t *__restrict__ stats) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        stats[0] += blocked[0];
        stats[1] += blocked[0] > 0 ? totals[0] : 0;
        stats[2] += blocked[1];
    }
}

// the batch's own elements: list (q, lc) of the lists hnsw_select_kernel made, into its place in the tuples
__global__ __launch_bounds__(256) void hnsw_link_new_kernel(int32_t *__restrict__ nbr, float *__restrict__ nb_dist,
                                                            uint8_t *__restrict__ nb_flag, const int32_t *__restrict__ levels,
                                                            const int64_t *__restrict__ nbr_start, int m,
                                                            const int32_t *__restrict__ elems, const uint8_t *__restrict__ linked,
                                                            int nq, int lcap, const int32_t *__restrict__ sel_ids,
                                                            const float *__restrict__ sel_dist,
                                                            const uint8_t *__restrict__ sel_closer,
                                                            const int32_t *__restrict__ sel_cnt) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= nq * lcap) return;
    const int q = g / lcap, lc = g % lcap;
    if (!linked[q]) return;
    const int e = elems[q], lv = levels[e];
    if (lc > lv) return;
    const int64_t pos = nbr_start[e] + (int64_t)(lv - lc) * m;
    const int rn = sel_cnt[g], stride = 2 * m;
    for (int i = 0; i < rn; i++) {
        nbr[pos + i] = sel_ids[(size_t)g * stride + i];
        nb_dist[pos + i] = sel_dist[(size_t)g * stride + i];
        nb_flag[pos + i] = sel_closer[(size_t)g * stride + i] & 1;  // closerSet stays 0: not sorted deterministically (:1143-1144)
    }
}

}  // namespace

int launch_hnsw_link_group(pgv_ctx *ctx, int step, const int32_t *elems, const uint8_t *linked, int nq, int lcap, int m,
                           const int32_t *sel_ids, const float *sel_dist, const int32_t *sel_cnt, const int32_t *levels,
                           const int64_t *nbr_start, int *list_count, int *list_rec, int *nrec, int32_t *rec_owner,
                           int32_t *rec_lc, int32_t *rec_list, const int64_t *rec_off, int *rec_fill, int32_t *link_elem,
                           float *link_dist) {
    const int64_t n = (int64_t)nq * lcap * 2 * m;
    if (n <= 0) return PGV_OK;
    const dim3 grid((unsigned)((n + 255) / 256));
    if (step == 0)
        hipLaunchKernelGGL(hnsw_link_count_kernel, grid, dim3(256), 0, ctx->stream, elems, linked, nq, lcap, m, sel_ids, sel_cnt,
                           levels, nbr_start, list_count, list_rec, nrec, rec_owner, rec_lc, rec_list);
    else
        hipLaunchKernelGGL(hnsw_link_fill_kernel, grid, dim3(256), 0, ctx->stream, elems, linked, nq, lcap, m, sel_ids, sel_dist,
                           sel_cnt, levels, nbr_start, list_rec, rec_off, rec_fill, link_elem, link_dist);
    PGV_HIP(hipGetLastError());
    return PGV_OK;
}

int launch_hnsw_link_size(pgv_ctx *ctx, const int32_t *nbr, const uint8_t *nb_flag, const int32_t *levels,
                          const int64_t *nbr_start, int m, const int32_t *rec_owner, const int32_t *rec_lc,
                          const int32_t *rec_list, const int *list_count, int64_t *rec_off, int nrec, int pass, int64_t *rec_pos,
                          int32_t *rec_nstart, int32_t *rec_from, const int32_t *rec_wait, int64_t *size_ids,
                          int64_t *size_pairs) {
    if (nrec <= 0) return PGV_OK;
    hipLaunchKernelGGL(hnsw_link_size_kernel, dim3((nrec + 255) / 256), dim3(256), 0, ctx->stream, nbr, nb_flag, levels,
                       nbr_start, m, rec_owner, rec_lc, rec_list, list_count, rec_off, nrec, pass, rec_pos, rec_nstart, rec_from,
                       rec_wait, size_ids, size_pairs);
    PGV_HIP(hipGetLastError());
    return PGV_OK;
}

int launch_hnsw_link_scan(pgv_ctx *ctx, int64_t *a, int64_t *b, int64_t *c, int n, int64_t *totals) {
    hipLaunchKernelGGL(hnsw_link_scan_kernel, dim3(1), dim3(kLinkScanThreads), 0, ctx->stream, a, b, c, n, totals);
    PGV_HIP(hipGetLastError());
    return PGV_OK;
}

int launch_hnsw_link_stats(pgv_ctx *ctx, const int *blocked, const int64_t *totals, int64_t *stats) {
    hipLaunchKernelGGL(hnsw_link_stats_kernel, dim3(1), dim3(64), 0, ctx->stream, blocked, totals, stats);
    PGV_HIP(hipGetLastError());
    return PGV_OK;
}

int launch_hnsw_link_pairs(pgv_ctx *ctx, const int32_t *nbr, const int64_t *rec_pos, const int32_t *rec_nstart,
                           const int32_t *rec_from, const int64_t *rec_off, int32_t *link_elem, float *link_dist,
                           const int32_t *rec_list, int *list_count, int nrec, int pass,
                           const int64_t *ids_start, int32_t *ids, const int64_t *pair_start, int32_t *a, int32_t *b) {
    if (nrec <= 0) return PGV_OK;
    const int cap = ctx->num_cus * 16;
    hipLaunchKernelGGL(hnsw_link_pairs_kernel, dim3(nrec < cap ? nrec : cap), dim3(256), 0, ctx->stream, nbr, rec_pos, rec_nstart,
                       rec_from, rec_off, link_elem, link_dist, rec_list, list_count, nrec, pass, ids_start, ids, pair_start, a, b);
    PGV_HIP(hipGetLastError());
    return PGV_OK;
}

int launch_hnsw_link_replay(pgv_ctx *ctx, int32_t *nbr, float *nb_dist, uint8_t *nb_flag, int m, int nrec, int pass,
                            const int32_t *rec_lc, const int64_t *rec_off, const float *link_dist, const int64_t *rec_pos,
                            const int32_t *rec_nstart, const int32_t *rec_from, const int64_t *ids_start, const int32_t *ids,
                            const int64_t *pair_start, const float *tri, const int64_t *mm_start, const float *mm,
                            int32_t *rec_wait, int16_t *loc_save, int *blocked) {
    if (nrec <= 0) return PGV_OK;
    LinkArgs g;
    g.nbr = nbr;
    g.nb_dist = nb_dist;
    g.nb_flag = nb_flag;
    g.m = m;
    g.nrec = nrec;
    g.pass = pass;
    g.rec_lc = rec_lc;
    g.rec_off = rec_off;
    g.link_dist = link_dist;
    g.rec_pos = rec_pos;
    g.rec_nstart = rec_nstart;
    g.rec_from = rec_from;
    g.ids_start = ids_start;
    g.ids = ids;
    g.pair_start = pair_start;
    g.tri = tri;
    g.mm_start = mm_start;
    g.mm = mm;
    g.rec_wait = rec_wait;
    g.loc_save = loc_save;
    g.blocked = blocked;
    // lists of up to 63 entries: a wavefront per record (PGV_HNSW_LINK_SERIAL=1: the one-lane-per-record form below,
    // which also serves the larger m)
    static const bool serial = getenv("PGV_HNSW_LINK_SERIAL") && atoi(getenv("PGV_HNSW_LINK_SERIAL")) != 0;
    if (2 * m + 1 <= 64 && !serial) {
        hipLaunchKernelGGL(hnsw_link_wave_kernel, dim3((nrec + 3) / 4), dim3(256), 0, ctx->stream, g);
        PGV_HIP(hipGetLastError());
        return PGV_OK;
    }
    const dim3 grid((nrec + 63) / 64);
    // the lists' working copies are per-lane arrays: sized for the m in use (2m + 1 entries)
    if (2 * m <= 32)
        hipLaunchKernelGGL(hnsw_link_kernel<32>, grid, dim3(64), 0, ctx->stream, g);
    else if (2 * m <= 64)
        hipLaunchKernelGGL(hnsw_link_kernel<64>, grid, dim3(64), 0, ctx->stream, g);
    else
        hipLaunchKernelGGL(hnsw_link_kernel<PGV_LINK_LMAX>, grid, dim3(64), 0, ctx->stream, g);
    PGV_HIP(hipGetLastError());
    return PGV_OK;
}

int launch_hnsw_link_new(pgv_ctx *ctx, int32_t *nbr, float *nb_dist, uint8_t *nb_flag, const int32_t *levels,
                         const int64_t *nbr_start, int m, const int32_t *elems, const uint8_t *linked, int nq, int lcap,
                         const int32_t *sel_ids, const float *sel_dist, const uint8_t *sel_closer, const int32_t *sel_cnt) {
    const int n = nq * lcap;
    if (n <= 0) return PGV_OK;
    hipLaunchKernelGGL(hnsw_link_new_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, nbr, nb_dist, nb_flag, levels,
                       nbr_start, m, elems, linked, nq, lcap, sel_ids, sel_dist, sel_closer, sel_cnt);
    PGV_HIP(hipGetLastError());
    return PGV_OK;
}

}  // namespace pgv
