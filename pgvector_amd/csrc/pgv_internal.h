// pgv_internal.h -- shared declarations of libpgv_hip (host side + launchers).
// gfx950 only; no other backend is supported or compiled.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>

#include "../../include/pgv_hip.h"

namespace pgv {

// ------------------------------------------------------------------ errors
void set_error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));

#define PGV_FAIL(code, ...)            \
    do {                               \
        ::pgv::set_error(__VA_ARGS__); \
        return (code);                 \
    } while (0)

#define PGV_HIP(call)                                                                  \
    do {                                                                               \
        hipError_t e__ = (call);                                                       \
        if (e__ != hipSuccess) {                                                       \
            ::pgv::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__),   \
                             __FILE__, __LINE__);                                      \
            return e__ == hipErrorOutOfMemory ? PGV_ERR_NOMEM : PGV_ERR_DEVICE;        \
        }                                                                              \
    } while (0)

#define PGV_TRY(call)          \
    do {                       \
        int rc__ = (call);     \
        if (rc__ != PGV_OK)    \
            return rc__;       \
    } while (0)

// ------------------------------------------------------------- device memory
struct DBuf {  // growable device allocation, reused across calls
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes);
    void release();
    template <typename T> T *as() const { return static_cast<T *>(p); }
};

struct HBuf {  // growable pinned host allocation
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes);
    void release();
    template <typename T> T *as() const { return static_cast<T *>(p); }
};

bool is_device_ptr(const void *p);

constexpr int kVecBytes = 16;  // every row is padded to whole 16-byte vectors in HBM

inline int elem_size(pgv_dtype t) { return t == PGV_F32 ? 4 : 2; }
// padded row length in elements
inline int padded_dim(int dim, pgv_dtype t) {
    int per = kVecBytes / elem_size(t);
    return (dim + per - 1) / per * per;
}

// geometry of the streaming kernels for one row length (see kernels_scan.hip)
struct RowGeom {
    int ld;         // padded elements per row
    int nvec;       // 16-byte vectors per row
    int lpr_log2;   // lanes cooperating on one row = 1 << lpr_log2
    int nchunks;    // loop trips over the row
};
RowGeom row_geom(int dim, pgv_dtype t);

// one unit of streaming work: rows [row0, row0 + nrows) scored against
// pairs [pair0, pair0 + npairs)
struct ScanTask {
    int64_t row0;
    int32_t nrows;
    int32_t pair0;
    int32_t npairs;
    int32_t pad;
};
// one (query, destination) of a task: distance of row r goes to out[out_rel + r]
struct ScanPair {
    int64_t out_rel;
    int32_t query;
    int32_t pad;
};

}  // namespace pgv

// --------------------------------------------------------------- the context
struct pgv_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int num_cus = 256;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // scratch (device)
    pgv::DBuf q_stage, rows_stage, centers_stage, out_stage, out_stage2, idx_stage;
    pgv::DBuf tasks, pairs, counters, plan_a, plan_b, plan_c, plan_d, dist_mat, sel_a, sel_b;
    pgv::DBuf km_a, km_b, km_c, km_d, km_e, km_f, km_g;
    pgv::DBuf ms_b;  // MFMA center ranking: the same scratch as ms_a
    pgv::DBuf dense_plan;  // the last dense (every row x every query) MFMA task list, reused while its shape repeats
    int64_t dense_plan_rows = -1, dense_plan_stride = -1;
    int dense_plan_nq = -1, dense_plan_kind = -1;
    bool counters_clean = false;  // ctx->counters starts zeroed; mfma_scan_kernel leaves its words zero again
    pgv::DBuf ms_a;  // MFMA list scan: query norms | candidate values, positions, slots | flags
    pgv::DBuf mf_d;  // MFMA assignment split over center parts: the parts' candidates per row
    pgv::DBuf mf_a, mf_b, mf_c, zeros;  // MFMA assignment: norms, pre-filter candidates, redo list; 16 zero bytes
    pgv::DBuf stats_dev;  // profiling: {pairs, rows streamed} of the batched list scans, as doubles
    // scratch (pinned host); h_a_busy marks the last async copy out of h_a (waited before reuse)
    pgv::HBuf h_a, h_b, h_c;
    hipEvent_t h_a_busy = nullptr;
    bool h_a_pending = false;
    // per-kernel profiling (pgv_ctx_set_profiling): HIP event pairs around the
    // streaming kernel, resolved lazily in pgv_ctx_get_stats
    bool profiling = false;
    bool no_mfma_scan = false;  // pgv_ctx_set_exact_scan
    bool no_widen = false;      // PGV_NO_WIDEN=1: flagged queries go straight to the exact pass (experiments)
    int bound_mode = 1;         // the scans' bound: PGV_BOUND_WORST_CASE unless pgv_ctx_set_bound says otherwise
    int assign_bound_mode = 1;  // the assignment pre-filter's: PGV_BOUND_WORST_CASE too since round 6 (pgv_ctx_set_bound)
    pgv::DBuf xt_norms;         // pgv_exact_topk: |row|^2 of the caller's rows
    std::vector<hipEvent_t> ev_pool;  // start/stop pairs
    size_t ev_used = 0;
    double scan_ms = 0.0;
    int64_t scan_launches = 0;
    double scan_pairs = 0.0;  // (row, query) pairs scored by timed launches
    double scan_rows = 0.0;   // rows streamed by timed launches
    std::vector<char> ev_is_aux;  // per event pair: counts towards aux_* instead of scan_*
    double aux_ms = 0.0;
    int64_t aux_launches = 0;
    double aux_pairs = 0.0;
    // lanes of indexes whose batches overlap (pgv_index_set_overlap): contexts with streams of their own that belong to
    // an index of this context; pgv_ctx_sync waits for them, the settings and the statistics include them
    std::vector<pgv_ctx *> children;
    // a lane's list scans take turns with the other lanes': two HBM-bound scans side by side only share the bandwidth,
    // while everything else of a batch (ranking, planning, top-k, recheck) fits under another batch's scan.  The event
    // belongs to the index (pgv_index::lane_scan_done); null outside lanes.
    hipEvent_t scan_gate = nullptr;
};

struct pgv_index {
    pgv_ctx *ctx = nullptr;
    pgv_metric metric = PGV_L2SQ;
    pgv_dtype dtype = PGV_F32;
    int dim = 0;
    int nlists = 0;
    int64_t nrows = 0;
    pgv::RowGeom geom{};
    void *centers = nullptr;          // [nlists x ld]
    void *vectors = nullptr;          // [nrows x ld]
    int64_t *list_offsets = nullptr;  // device [nlists + 1]
    uint64_t *tids = nullptr;         // device [nrows] or null
    float *row_norms = nullptr;       // device [nrows] |x|^2 then one word: bits of the largest (L2 indexes; the MFMA scan)
    float *center_norms = nullptr;    // the same for the centers [nlists + 1]
    int *refs = nullptr;              // handles (the uploaded index + its pgv_index_share views) on the device arrays
    // every device array above lives in ONE allocation, so that one hipIpcMemHandle carries the whole mirror to
    // another process (pgv_index_export / pgv_index_import)
    void *arena = nullptr;
    size_t arena_bytes = 0;
    bool imported = false;            // arena was opened with hipIpcOpenMemHandle: closed, not freed
    std::vector<int64_t> h_offsets;   // host copy
    std::vector<int64_t> len_prefix;  // len_prefix[p] = rows in the p longest lists (output size bound)
    int64_t max_list_len = 0;
    // pgv_index_set_overlap: views of this index on contexts (streams, scratch) of their own; consecutive
    // pgv_search_batch calls take them in turn
    std::vector<pgv_index *> lanes;
    unsigned lane_next = 0;
    hipEvent_t lane_event = nullptr;  // orders a lane's start after what the caller's stream holds at the call
    hipEvent_t lane_scan_done = nullptr;  // the lanes' list scans run one after the other (pgv_ctx::scan_gate)
};

// one backend's index scan (pgv_query_*): everything between amrescan and amendscan that lives on the device
struct pgv_query {
    pgv_index *ix = nullptr;
    pgv::DBuf state;      // lists[head cap] | cdist[nlists]
    pgv::DBuf seg;        // distances of the current GetScanItems batch, tuplesort input order
    pgv::DBuf q_dev;      // the query row, padded
    void *q_pinned = nullptr;     // pinned host staging of the query payload
    const void *q_row = nullptr;  // where the kernels read the current query: q_direct or q_dev
    void *q_direct = nullptr;     // fine-grained device row the host writes through the BAR (no staging kernel), or null
    void *head_pinned = nullptr;  // pinned, device-written: QueryHead + head arrays
    size_t head_bytes = 0;
    int32_t *lists = nullptr;
    float *cdist = nullptr;
    int max_probes = 0;   // lists ranked by the last pgv_query_rank
    bool is_null = false;
    bool rank_pending = false;     // kernels that read q_pinned may still be in flight
    int cur_first = 0, cur_n = 0;  // the batch whose distances are in seg
    unsigned seq = 0;
};

struct pgv_hnsw {
    pgv_ctx *ctx = nullptr;
    pgv_metric metric = PGV_L2SQ;
    pgv_dtype dtype = PGV_F32;
    int dim = 0;
    int64_t n = 0;
    pgv::RowGeom geom{};
    void *elements = nullptr;  // [n x ld]
    // the graph (pgv_hnsw_set_graph): one allocation holding levels | nbr_start | nbr
    void *graph = nullptr;
    const int32_t *levels = nullptr;
    const int64_t *nbr_start = nullptr;
    int32_t *nbr = nullptr;
    int m = 0;
    int32_t entry = -1;
    pgv::DBuf bitmaps;  // visited sets of the search workgroups
    size_t graph_bytes = 0;
    int64_t nbr_total = 0;
    bool imported = false;  // elements / graph were opened with hipIpcOpenMemHandle (a read-only view)
    pgv_hnsw *view_of = nullptr;  // pgv_hnsw_share: a view of that mirror in the same process (own context / stream)
    // the last pgv_hnsw_update_graph of the mirror (through it or through a view), recorded on the stream that ran it:
    // searches on any other stream wait for it on the device (owner's field; views look at view_of's)
    hipEvent_t graph_ev = nullptr;
    bool graph_ev_set = false;
    char *payload = nullptr;  // [n x payload_bytes] behind the elements, same allocation (pgv_hnsw_upload_payload)
    int payload_bytes = 0;
    struct HnswLinkState *link = nullptr;  // pgv_hnsw_link_begin .. _end: the in-memory build's graph state (owner only)
};

// the state of a build whose graph updates run on the device (kernels_hnsw_link.hip): per tuple slot the neighbor's
// distance and closer flag, and the scratch of the batch in flight
struct HnswLinkState {
    float *nb_dist = nullptr;     // [nbr_total]
    uint8_t *nb_flag = nullptr;   // [nbr_total] bit 0 closer; bit 1 of a list's first slot: closerSet
    int *list_count = nullptr;    // [nlists] link requests of the batch in flight per list (slot position / m); zero between batches
    int *list_rec = nullptr;      // [nlists] the list's record in that batch
    size_t nlists = 0;
    int64_t *stats_dev = nullptr, *stats_host = nullptr;  // [3] lists deferred | member pairs scored | updates left waiting (sums)
    pgv::DBuf rec, links, ids, pa, pb, tri, mm, loc, sel;
    int nrec = 0, nq = 0, lcap = 0;
    int64_t npairs = 0;   // of the batch prepared last
    bool prepared = false;
    // where the pieces of `rec` / `links` / `sel` are (set by prepare)
    int32_t *rec_owner = nullptr, *rec_lc = nullptr, *rec_nstart = nullptr, *rec_from = nullptr, *rec_wait = nullptr,
            *rec_list = nullptr;
    int64_t *rec_off = nullptr, *rec_pos = nullptr, *ids_start = nullptr, *pair_start = nullptr, *mm_start = nullptr,
            *totals = nullptr;
    int *rec_fill = nullptr, *blocked = nullptr, *nrec_dev = nullptr;
    int32_t *d_link_elem = nullptr;
    float *d_link_dist = nullptr;
    int32_t *d_sel_ids = nullptr, *d_sel_cnt = nullptr, *d_elems = nullptr;
    float *d_sel_dist = nullptr;
    uint8_t *d_sel_closer = nullptr, *d_linked = nullptr;
    // pgv_hnsw_build_search_keep: the candidate lists of two batches (ids | distances | counts; elements | insert levels)
    struct Kept {
        pgv::DBuf ids, dist, cnt, elems;
        int nq = 0, ef = 0, lcap = 0;
    } kept[2];
};

namespace pgv {

// How far an MFMA L2 value a = |x|^2 - 2 q.x can be from the true s = |x|^2 - 2 q.x of the same pair (the distance
// less |q|^2, which shifts all values of a query alike), and what the candidate selection must allow for.
//
// ScanBound -- the list scan / center ranking / exact top-k (mfma_scan_kernel + batch_recheck_kernel):
//     eps(q, x) = g_sq (|q| + |x|)^2  +  g_dot 2 |q||x|  +  g_norm |x|^2 ,      band = a_k + 2 eps + g_ref |a_k + 2 eps + |q|^2|
//   statistical    g_sq = 8 sqrt(dim + 4) u, the rest 0: the probabilistic bound of a length-dim fp32 summation (Higham &
//                  Mary 2019: lambda sqrt(n) u fails with probability ~ exp(-lambda^2 / 2) per sum, lambda = 8), u = 2^-24
//   worst case     deterministic (Higham, Accuracy and Stability of Numerical Algorithms, Lemma 3.1 / eq. 3.5: a sum of n
//                  rounded products in ANY order errs by at most gamma_n sum |a_i b_i|, gamma_n = n u / (1 - n u)), with
//                  u = 2^-24 per operation: the matrix pipeline rounds products and accumulator additions to nearest,
//                  which tests/test_gpu_round4.py pins on the hardware (half-way cases through both MFMA shapes):
//                    g_dot   gamma_(dim/4 + 4): the kernel keeps FOUR independent accumulators per output, each adds
//                            dim / 4 products (Cauchy-Schwarz: sum |q_i x_i| <= |q||x|), two more additions join them
//                            and the final fma(-2, dot, |x|^2) rounds once
//                    g_norm  gamma_(dim/64 + 10): row_norms_kernel's per-lane chains of dim / 64 fmas + 6 shuffle
//                            additions, + the final fma
//                    g_ref   2 gamma_(dim + 2): the exact form sum((q - x)^2) that decides among the candidates (here
//                            and in the reference) is itself rounded -- all terms positive, so RELATIVE to the distance;
//                            a row outside the band must stay outside when both its and the k-th row's exact values move
//                  |x| is the largest row norm of the index (one word, kept with the norms).
// ArgminBound -- the build's L2 pre-filter (mfma_argmin_kernel, ONE accumulator chain per output: 128 accumulators a
//   lane leave no room for four): gamma (|c|^2 + 2 |a||c|) + gamma_x d, DETERMINISTIC by default since round 6:
//   gamma = gamma_(dim+1), 5-10 x wider than the statistical 8 sqrt(dim + 4) u.  What made that affordable is not a
//   tighter band but cheaper ambiguity: a row's merged list carries kWide = 8 candidates and `dropped`, the smallest value
//   no list kept (keep_best), so the exact recheck evaluates whatever lies inside the band and only a row whose band holds
//   MORE than the lists do (its 8th value or `dropped` inside it) goes to the all-centers exact kernel -- round 5 sent
//   every row whose 4th value was inside (10 % of 3072-d rows: 32 -> 123 ms).  Measured (profiles/r06/assign_bounds.md):
//   worst case over statistical +4 % (1 M x 1000 x 1536 fp32), +10 % (1.25 M x 4096 x 3072 fp16), redo 0 - 0.06 %.
//   One unit roundoff per product of the chain is what the bound charges; tools/mfma_numerics.py shows the fp32 matrix
//   instructions to BE an fmaf chain bit for bit and the fp16 ones to lose at most 3.2 u per 16 products (charged: 16 u).
struct ScanBound {
    float g_sq, g_dot, g_norm, g_ref;
};
inline float gamma_n(double n, double v) { return (float)(n * v / (1.0 - n * v)); }
inline ScanBound scan_bound(const pgv_ctx *ctx, int dim) {
    if (ctx->bound_mode == 0) return {8.f * std::sqrt((float)dim + 4.f) * 5.9604645e-8f, 0.f, 0.f, 0.f};
    constexpr double v = 5.9604645e-8;  // 2^-24: round to nearest (tests/test_gpu_round4.py)
    return {0.f, gamma_n(dim / 4.0 + 4.0, v), gamma_n(dim / 64.0 + 10.0, v), 2.f * gamma_n(dim + 2.0, v)};
}
// the same with the products per accumulator chain given (mfma_dense_kernel: quarters of whole 128-byte slices)
inline ScanBound scan_bound_chain(const pgv_ctx *ctx, int dim, int chain) {
    ScanBound b = scan_bound(ctx, dim);
    if (ctx->bound_mode != 0) b.g_dot = gamma_n(chain + 4.0, 5.9604645e-8);
    return b;
}
struct ExpansionBound {
    float gamma;        // of |x|^2 + 2 |q||x| (expansion terms)
    float gamma_exact;  // of (|q| + |x|)^2 (the exact value's own rounding; 0 in the statistical model)
};
inline ExpansionBound argmin_bound(const pgv_ctx *ctx, int dim) {
    constexpr float u = 5.9604645e-8f;
    if (ctx->assign_bound_mode == 0) return {8.f * std::sqrt((float)dim + 4.f) * u, 0.f};
    const float n1 = (float)(dim + 1) * u, n2 = (float)(dim + 2) * u;
    return {n1 / (1.f - n1), n2 / (1.f - n2)};
}

// ---------------------------------------------------------------- launchers
// kernels_scan.hip: stream rows, score them against small groups of queries
int launch_scan(pgv_ctx *ctx, pgv_metric metric, pgv_dtype dtype, const RowGeom &g,
                const void *rows, const void *queries, const ScanTask *tasks,
                const int *ntasks_dev, int ntasks_bound, const ScanPair *pairs, int qt,
                float *out);
int scan_group_size(const RowGeom &g, pgv_dtype dtype, int wanted);
// gathered variant (HNSW candidate scoring): pair i = (slot[i], query_of[i])
int launch_expand_groups(pgv_ctx *ctx, const int32_t *ids, const int64_t *ids_start, const int32_t *from,
                         const int64_t *pair_start, int ngroups, int32_t *a, int32_t *b);
int launch_score_gather(pgv_ctx *ctx, pgv_metric metric, pgv_dtype dtype, const RowGeom &g,
                        const void *rows, const void *queries, const int32_t *slot,
                        const int32_t *query_of, int64_t npairs, float *out);
// the pairs (u, v < u), u >= from, inside groups of rows, in 4 x 4 tiles (bit for bit score_gather's values).  Group g is
// ids[ids_at[g] ..) (ids_at NULL: g * ids_stride), n_arr[g] rows (NULL: ids_at[g + 1] - ids_at[g]), from_arr[g] (NULL: 1);
// its pairs go to out[pair_at[g] ..); pair_at[g + 1] == pair_at[g]: nothing wanted
int launch_score_groups(pgv_ctx *ctx, pgv_metric metric, pgv_dtype dtype, const RowGeom &g, const void *rows,
                        const int32_t *ids, const int64_t *ids_at, int64_t ids_stride, const int32_t *n_arr,
                        const int32_t *from_arr, const int64_t *pair_at, int ngroups, float *out);

// kernels_misc.hip: operator-path cosine distance and bit-vector distances, one query x n rows
int launch_cosine(pgv_ctx *ctx, pgv_dtype dtype, const RowGeom &g, const void *rows, const void *query, int64_t n,
                  double *out);
int launch_bit_distance(pgv_ctx *ctx, int mode, const RowGeom &g, const void *rows, const void *query, int64_t n,
                        double *out);

// kernels_hnsw.hip: the whole first batch of an HNSW scan, one workgroup per query
int hnsw_search_grid(pgv_ctx *ctx, int nq, int64_t n, int *words_out);
struct HnswSearchArgs {
    const void *queries = nullptr;     // staged query rows, or
    const int32_t *qids = nullptr;     // element slots used as queries (build)
    const int32_t *qlevels = nullptr;  // insert level per query (build)
    int nq = 0, ef = 0, k = 0;
    int64_t *out_elem = nullptr;
    float *out_dist = nullptr;
    int64_t *out_scored = nullptr;
    int32_t *lw_ids = nullptr;         // per-layer W (build)
    float *lw_dist = nullptr;
    int32_t *lw_cnt = nullptr;
    int lcap = 0;
};
int launch_hnsw_search(pgv_ctx *ctx, pgv_metric metric, pgv_dtype dtype, const RowGeom &geom, const void *rows,
                       int64_t n, const int32_t *levels, const int64_t *nbr_start, const int32_t *nbr, int m,
                       int32_t entry, const HnswSearchArgs &a, uint32_t *bitmaps, int words, int grid, int *counter);
int launch_hnsw_patch(pgv_ctx *ctx, int32_t *nbr, const int64_t *nbr_start, int64_t n, const int32_t *ids,
                      const int64_t *packed_off, const int32_t *packed, int nupd);

// kernels_tile.hip: row tiles in LDS (async DMA), queries in registers: 16 queries per pass
bool tile_scan_supported(const RowGeom &g);
int tile_scan_queries_per_task();
int tile_scan_tile_rows(const RowGeom &g);
int launch_tile_scan(pgv_ctx *ctx, pgv_metric metric, pgv_dtype dtype, const RowGeom &g,
                     const void *rows, const void *queries, const ScanTask *tasks,
                     const int *ntasks_dev, int ntasks_bound, const ScanPair *pairs, float *out);

// kernels_pair.hip: n rows x k centers, both large: argmin per row
int launch_argmin(pgv_ctx *ctx, pgv_metric metric, pgv_dtype dtype, const RowGeom &g,
                  const void *rows, int64_t n, const void *centers, int k, int32_t *out_idx,
                  float *out_val);
// mode 0..2 = pgv_metric, 3 = spherical k-means (-clamp(ip))
int launch_argmin_mode(pgv_ctx *ctx, int mode, pgv_dtype dtype, const RowGeom &g,
                       const void *rows, int64_t n, const void *centers, int k,
                       int32_t *out_idx, float *out_val);

int launch_argmin_listed(pgv_ctx *ctx, int mode, pgv_dtype dtype, const RowGeom &g, const void *rows, int64_t n,
                         const void *centers, int k, const int32_t *row_list, const int *row_count,
                         unsigned long long *packed);
// kernels_hnsw.hip: SelectNeighbors of a batch's new elements on the device (plan -> pairs -> [score_gather] -> sweep)
int launch_hnsw_select_plan(pgv_ctx *ctx, const int32_t *cnt, int ngroups, int lcap, int m, int64_t *pair_start);
int launch_hnsw_select_pairs(pgv_ctx *ctx, const int32_t *lw_ids, const int32_t *cnt, const int64_t *pair_start, int ngroups,
                             int ef, int32_t *a, int32_t *b);
int launch_hnsw_select(pgv_ctx *ctx, const int32_t *lw_ids, const float *lw_dist, const int32_t *cnt, const int32_t *qlevels,
                       const int64_t *pair_start, const float *tri, int ngroups, int lcap, int ef, int m, int stride,
                       int32_t *out_ids, float *out_dist, uint8_t *out_closer, int32_t *out_cnt);
// kernels_hnsw_link.hip: a batch linked into the neighbor lists it chose, on the device
int launch_hnsw_link_group(pgv_ctx *ctx, int step, const int32_t *elems, const uint8_t *linked, int nq, int lcap, int m,
                           const int32_t *sel_ids, const float *sel_dist, const int32_t *sel_cnt, const int32_t *levels,
                           const int64_t *nbr_start, int *list_count, int *list_rec, int *nrec, int32_t *rec_owner,
                           int32_t *rec_lc, int32_t *rec_list, const int64_t *rec_off, int *rec_fill, int32_t *link_elem,
                           float *link_dist);
int launch_hnsw_link_size(pgv_ctx *ctx, const int32_t *nbr, const uint8_t *nb_flag, const int32_t *levels,
                          const int64_t *nbr_start, int m, const int32_t *rec_owner, const int32_t *rec_lc,
                          const int32_t *rec_list, const int *list_count, int64_t *rec_off, int nrec, int pass, int64_t *rec_pos,
                          int32_t *rec_nstart, int32_t *rec_from, const int32_t *rec_wait, int64_t *size_ids,
                          int64_t *size_pairs);
int launch_hnsw_link_scan(pgv_ctx *ctx, int64_t *a, int64_t *b, int64_t *c, int n, int64_t *totals);
int launch_hnsw_link_stats(pgv_ctx *ctx, const int *blocked, const int64_t *totals, int64_t *stats);
int launch_hnsw_link_pairs(pgv_ctx *ctx, const int32_t *nbr, const int64_t *rec_pos, const int32_t *rec_nstart,
                           const int32_t *rec_from, const int64_t *rec_off, int32_t *link_elem, float *link_dist,
                           const int32_t *rec_list, int *list_count, int nrec, int pass,
                           const int64_t *ids_start, int32_t *ids, const int64_t *pair_start, int32_t *a, int32_t *b);
int launch_hnsw_link_replay(pgv_ctx *ctx, int32_t *nbr, float *nb_dist, uint8_t *nb_flag, int m, int nrec, int pass,
                            const int32_t *rec_lc, const int64_t *rec_off, const float *link_dist, const int64_t *rec_pos,
                            const int32_t *rec_nstart, const int32_t *rec_from, const int64_t *ids_start, const int32_t *ids,
                            const int64_t *pair_start, const float *tri, const int64_t *mm_start, const float *mm,
                            int32_t *rec_wait, int16_t *loc_save, int *blocked);
int launch_hnsw_link_new(pgv_ctx *ctx, int32_t *nbr, float *nb_dist, uint8_t *nb_flag, const int32_t *levels,
                         const int64_t *nbr_start, int m, const int32_t *elems, const uint8_t *linked, int nq, int lcap,
                         const int32_t *sel_ids, const float *sel_dist, const uint8_t *sel_closer, const int32_t *sel_cnt);
// kernels_mfma.hip: the same on the matrix cores (ip / spherical directly, L2 as pre-filter + exact recheck)
bool mfma_argmin_supported(int mode, int64_t n, int k);
int launch_argmin_mfma(pgv_ctx *ctx, int mode, pgv_dtype dtype, const RowGeom &g, const void *rows, int64_t n,
                       const void *centers, int k, int32_t *out_idx, float *out_val);

// kernels_mfma.hip: the batched list scan on the matrix cores (<= 128 rows x <= 32 queries per task)
int mfma_scan_rows_per_task();
int mfma_scan_queries_per_task();
// kernels_dense.hip: every query x every row, 128 x 128 tiles (pgv_exact_topk); out[q * out_stride + row]
int launch_mfma_dense(pgv_ctx *ctx, pgv_metric metric, pgv_dtype dtype, const RowGeom &g, const void *rows, int64_t n,
                      const void *queries, int nq, const float *row_norms, float *out, int64_t out_stride);
int dense_chain_length(const RowGeom &g, pgv_dtype dtype);  // products per accumulator chain, at most
int launch_row_norms(pgv_ctx *ctx, pgv_dtype dtype, const RowGeom &g, const void *rows, int64_t n, float *out,
                     unsigned *max_bits);
int launch_mfma_scan(pgv_ctx *ctx, pgv_metric metric, pgv_dtype dtype, const RowGeom &g, const void *rows,
                     const void *queries, const ScanTask *tasks, const int *ntasks_dev, int ntasks_bound,
                     const ScanPair *pairs, const float *row_norms, const float *query_norms, float *out, bool stream_rows,
                     int queries_per_task = 32);
int mfma_scan_queries_per_task_wide();  // 64: tasks of the batches whose lists are probed by many queries each

// kernels_build.hip: rows of 32-bit words gathered by index (the HNSW mirror's per-element payload)
int launch_gather_words(pgv_ctx *ctx, const void *src, int words_per_row, int64_t nrows, const int64_t *idx, int n,
                        uint32_t *out);
// kernels_build.hip: the build's tuplesort on the device (order by list, heap order inside; gather)
size_t build_sort_scratch_bytes(int64_t n, int key_bits);
int launch_build_order(pgv_ctx *ctx, const int32_t *lists, int64_t n, int nlists, unsigned long long *keys_tmp,
                       unsigned long long *keys_sorted, unsigned long long *counts, int64_t *offsets, int *bad,
                       void *sort_scratch, size_t sort_scratch_bytes);
int launch_build_gather(pgv_ctx *ctx, const void *src_rows, const unsigned long long *keys_sorted, int64_t n, int nvec,
                        void *dst_rows, const uint64_t *src_tids, uint64_t *dst_tids);

// kernels_select.hip: planning + top-k selection
struct PlanResult {
    ScanTask *tasks = nullptr;
    ScanPair *pairs = nullptr;
    int *ntasks_dev = nullptr;
    int64_t ntasks = 0;           // exact when the totals were read back, else the bound
    int64_t total_out = 0;        // sum of the queries' segment lengths (exact or bound, likewise)
    int64_t ntasks_bound = 0;
    int64_t out_bound = 0;
    int64_t *seg_start = nullptr; // device [nq + 1]
    int64_t *probe_off = nullptr; // device [nq x probes]
};
int launch_plan_batch(pgv_ctx *ctx, const pgv_index *ix, const int32_t *probe_lists, int nq,
                      int probes, int qt, int rows_per_task, bool read_totals, PlanResult *res);
int launch_topk_segments(pgv_ctx *ctx, const float *vals, const int64_t *seg_start, int nseg,
                         int64_t fixed_len, int k, float *out_val, int64_t *out_pos, int32_t *zero_word = nullptr);
int launch_positions_to_slots(pgv_ctx *ctx, const pgv_index *ix, const int32_t *probe_lists,
                              const int64_t *probe_off, int nq, int probes, int k,
                              const int64_t *pos, int64_t *out_slot, uint64_t *out_tid);
// kernels_query.hip: the exact tail of the MFMA L2 scans.  The rows the candidates come
// from: an index's tuples (list_offsets set) or its centers (one dense run)
struct ExactRows {
    const void *vectors;
    const uint64_t *tids;          // or null
    const int64_t *list_offsets;   // or null
    pgv::RowGeom geom;
    pgv_dtype dtype;
    const unsigned *norm_max;      // bits of the largest |row|^2
};
int launch_batch_recheck(pgv_ctx *ctx, const ExactRows &xr, const void *q_dev, int nq, int kprime, int k,
                         const float *approx_val, const int64_t *cand_pos, const int64_t *cand_slot,
                         const int64_t *seg_start, int64_t fixed_len, const ScanBound &bound,
                         float *out_dist, int64_t *out_slot, uint64_t *out_tid, int32_t *flags,
                         int32_t *out_i32 = nullptr, const int32_t *probe_lists = nullptr,
                         const int64_t *probe_off = nullptr, int probes = 0);  // cand_slot null: slots from the positions
// the flagged queries start to end: exact scores of the whole segment, head, output row (out_slot: row slots, or
// center ids for the dense form)
int launch_batch_fix(pgv_ctx *ctx, const ExactRows &xr, const void *q_dev, int nq, const int32_t *probe_lists,
                     const int64_t *probe_off, int probes, const int64_t *seg_start, int64_t fixed_len,
                     const int32_t *flags, float *seg_vals, int k, const ScanBound &bound, float *out_dist, int64_t *out_slot,
                     uint64_t *out_tid, int32_t *out_i32 = nullptr);
int launch_iota_slots(pgv_ctx *ctx, const pgv_index *ix, const int32_t *lists_dev, int nlists,
                      const int64_t *probe_off, int64_t *out_slot);

// kernels_query.hip: one query at a time ([stage,] rank, lists, scan, head: four or five launches, no host round trip)
int query_max_batch_lists();
int query_head_cap();
size_t query_head_bytes(int head);
int launch_query_stage(pgv_ctx *ctx, const void *src_pinned, void *dst_dev, int nvec);
// the same kernels for a few queries at a time (one grid row per query, results in device arrays)
int launch_multi_rank(pgv_ctx *ctx, const pgv_index *ix, const void *q_dev, int nq, float *cdist, int64_t cd_stride,
                      int max_probes, int32_t *out_lists, float *out_dist);
int launch_multi_scan(pgv_ctx *ctx, const pgv_index *ix, const void *q_dev, int nq, const int32_t *probe_lists, int nprobes,
                      int64_t rows_bound, float *seg, int64_t seg_stride, int k, float *out_dist, int64_t *out_slot,
                      uint64_t *out_tid);
int launch_query_iota(pgv_ctx *ctx, int32_t *out, int n);
int launch_one_query_rows(pgv_ctx *ctx, pgv_metric metric, pgv_dtype dtype, const RowGeom &g, const void *rows, int n,
                          const void *q_dev, float *out);
int launch_query_rank(pgv_ctx *ctx, const pgv_index *ix, const void *q_dev, float *cdist, int max_probes,
                      int32_t *out_lists);
int launch_query_scan(pgv_ctx *ctx, const pgv_index *ix, const void *q_dev, const int32_t *probe_lists, int nprobes,
                      int64_t rows_bound, float *seg);
int launch_query_head(pgv_ctx *ctx, const pgv_index *ix, const float *seg, const int32_t *probe_lists, int nprobes,
                      int skip, int count, void *head_rec, unsigned seq);

int launch_merge_heads(pgv_ctx *ctx, const float *dist_all, const uint64_t *tid_all, int nranks, int nq, int k,
                       float *out_dist, uint64_t *out_tid);

// kernels_select.hip (small helpers)
int launch_cast_pos_to_i32(pgv_ctx *ctx, const int64_t *pos, int64_t n, int32_t *out);

// kernels_kmeans.hip
int kmpp_block_count(int n);
int launch_kmpp_update(pgv_ctx *ctx, const float *raw, float *weight, int n, int spherical,
                       double *block_sums);
int launch_kmpp_pick(pgv_ctx *ctx, const RowGeom &g, const void *samples, int n,
                     const float *weight, const double *block_sums, const double *draws,
                     int round, void *centers, int32_t *picked);
int launch_kmpp_total(pgv_ctx *ctx, const double *block_sums, int nblocks, double *out);
int launch_kmpp_pick_sharded(pgv_ctx *ctx, const RowGeom &g, const void *samples, int n, const float *weight,
                             const double *block_sums, const double *totals, int nranks, int rank, const double *draws,
                             int round, void *send_row, int32_t *owner_out);
int launch_kmpp_take_row(pgv_ctx *ctx, const RowGeom &g, const void *gathered, const int32_t *owner, void *centers,
                         int round);
int launch_lloyd_pack(pgv_ctx *ctx, const int32_t *counts, const unsigned long long *changes, int k, float *tail);
int launch_lloyd_unpack(pgv_ctx *ctx, const float *tail, int k, int32_t *counts, unsigned long long *changes,
                        long long *host_rec, long long seq);
int launch_changes_hist(pgv_ctx *ctx, const int32_t *closest_new, int32_t *closest_io, int n,
                        int32_t *counts, unsigned long long *changes);
int launch_members(pgv_ctx *ctx, const int32_t *closest, int n, int k, const int32_t *counts,
                   int32_t *offsets, int32_t *members);
int launch_center_sums(pgv_ctx *ctx, pgv_dtype dtype, const RowGeom &g, const void *samples,
                       const int32_t *offsets, const int32_t *members, int k, float *sums);
int launch_finish_centers(pgv_ctx *ctx, pgv_dtype dtype, const RowGeom &g, int k, int dim,
                          const float *sums, const int32_t *counts, const float *refill,
                          const int32_t *refill_row, void *centers);
int launch_normalize_rows(pgv_ctx *ctx, pgv_dtype dtype, const RowGeom &g, void *rows, int64_t n,
                          int dim, int32_t *flag);
int launch_check_centers(pgv_ctx *ctx, pgv_dtype dtype, const RowGeom &g, const void *centers,
                         int k, int dim, int check_zero_norm, int32_t *flag);

}  // namespace pgv
