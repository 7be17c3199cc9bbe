/*
 * pgv_hip.h -- C ABI of libpgv_hip.so: the MI355X (gfx950) implementation of
 * pgvector's distance hot path.
 *
 * This is the drop-in boundary.  pgvector has no batch seam of its own: every
 * distance goes through a per-pair fmgr call (src/ivfflat.h:307 `distfunc`,
 * src/hnswutils.c:524-528 `HnswGetDistance`).  Each entry point below replaces
 * one of the LOOPS around that call; the reference loop it replaces is cited
 * on the declaration.  Host code inside the extension (C)
 * stages vectors out of 8 KB pages into contiguous arrays, calls these
 * functions, and feeds the results back into the same Postgres structures the
 * reference uses (tuplesort, pairing heaps, page writers).
 *
 * Conventions
 *   - plain C: opaque handles, plain pointers and sizes; no C++/torch types.
 *   - every function returns PGV_OK or a PGV_ERR_* code; the message of the
 *     last failure on the calling thread is pgv_last_error().
 *   - buffer arguments may be HOST or DEVICE pointers; the library asks the HIP
 *     runtime which (hipPointerGetAttributes) and stages host buffers through
 *     its own pinned/device scratch.  A Postgres backend passes host memory; a
 *     harness that already keeps data in HBM passes device memory and pays no
 *     PCIe traffic.
 *   - a handle is used by one thread at a time (a Postgres backend is single
 *     threaded); different handles may be used concurrently.
 *   - all work is enqueued on the context's HIP stream; functions that return
 *     results into host memory synchronise that stream before returning,
 *     functions whose outputs are device pointers do not (pgv_ctx_sync does).
 *   - vectors are `dim` consecutive elements of `dtype` (float or IEEE
 *     binary16), row-major, densely packed: the payload of the reference's
 *     Vector / HalfVector varlenas (src/vector.h:18-24, src/halfvec.h:68-74)
 *     without the 8-byte header.
 *   - distances come back as the fp32 value of the reference's kernel
 *     (src/vector.c:560-617, src/halfutils.c:29-121); the float8 widening,
 *     sqrt, negation-for-display etc. of the fmgr wrappers is exact host work.
 *   - there is NO CPU fallback: without a usable gfx950 device every call
 *     fails with PGV_ERR_DEVICE.
 */
#ifndef PGV_HIP_H
#define PGV_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PGV_ABI_VERSION 1

/* status codes */
#define PGV_OK 0
#define PGV_ERR_ARG 1			/* bad argument (NULL, negative size, ...) */
#define PGV_ERR_DIMS 2			/* dimension limits: 1..16000 (src/vector.h:10) */
#define PGV_ERR_DEVICE 3		/* no usable GPU / HIP runtime error */
#define PGV_ERR_NOMEM 4			/* device or host allocation failed */
#define PGV_ERR_STATE 5			/* handle in the wrong state */
#define PGV_ERR_DATA 6			/* data error the reference would elog (NaN center, ...) */

/* element type of an indexed column: vector (float4) or halfvec (binary16) */
typedef enum pgv_dtype
{
	PGV_F32 = 0,
	PGV_F16 = 1
}			pgv_dtype;

/*
 * Distance computed by a kernel = FUNCTION 1 of the operator class
 * (src/ivfflat.h:40 IVFFLAT_DISTANCE_PROC, src/hnsw.h:37 HNSW_DISTANCE_PROC):
 *   PGV_L2SQ    vector_l2_squared_distance / halfvec_l2_squared_distance   (sql/vector.sql:409,822)
 *   PGV_NEG_IP  vector_negative_inner_product / halfvec_...                (sql/vector.sql:415,422,828,835)
 *   PGV_L1      l1_distance (hnsw vector_l1_ops, sql/vector.sql:445)
 */
typedef enum pgv_metric
{
	PGV_L2SQ = 0,
	PGV_NEG_IP = 1,
	PGV_L1 = 2
}			pgv_metric;

/*
 * Operator-class family, for the build path where FUNCTION 3/4 matter too:
 *   PGV_OPS_L2      k-means on l2_distance, nothing normalised
 *   PGV_OPS_IP      spherical k-means (samples and centers normalised), rows stored as-is
 *   PGV_OPS_COSINE  spherical k-means, rows stored normalised, zero rows skipped
 * (sql/vector.sql:406-425; src/ivfbuild.c:69-73,154-155,174-180)
 */
typedef enum pgv_ops
{
	PGV_OPS_L2 = 0,
	PGV_OPS_IP = 1,
	PGV_OPS_COSINE = 2
}			pgv_ops;

typedef struct pgv_ctx pgv_ctx;		/* one GPU + stream + scratch */
typedef struct pgv_index pgv_index;	/* device mirror of one IVFFlat index */
typedef struct pgv_hnsw pgv_hnsw;	/* device mirror of one HNSW index's element vectors */

/*
 * Source of RandomDouble()/RandomInt() (src/ivfflat.h:86-94).  Inside Postgres
 * the callbacks wrap pg_prng_double/pg_prng_uint32 on pg_global_prng_state so
 * the stream of draws is the reference's.  NULL callbacks select the library's
 * own xoroshiro128** seeded with `seed`.
 */
typedef struct pgv_rng
{
	double		(*next_double) (void *state);	/* uniform [0,1) */
	uint32_t	(*next_u32) (void *state);
	void	   *state;
	uint64_t	seed;
}			pgv_rng;

/* ------------------------------------------------------------------ context */

const char *pgv_last_error(void);
int			pgv_abi_version(void);

/* number of HIP devices visible (0 when there is no GPU / no driver) */
int			pgv_device_count(void);
/* free / total HBM of a device in bytes as the driver reports them now (all processes' allocations counted) */
int			pgv_device_memory(int device, uint64_t *free_bytes, uint64_t *total_bytes);
/*
 * Page-locked host memory (hipHostMalloc) for buffers handed to the library again and again -- the lanes of
 * the host glue's pooler (pgv_host_pool_*), a backend's query staging.  Plain host memory works everywhere
 * too; pinned memory makes the copies in and out asynchronous and about twice as fast.
 */
int			pgv_pinned_alloc(size_t bytes, void **out);
void		pgv_pinned_free(void *p);
/*
 * Page-lock memory the caller already owns -- a range of a shared-memory segment (Postgres' DSM / the main shared
 * memory block) that several processes fill and the process that owns the GPU context hands to the library: the
 * pooler's query and result lanes (pgv_host_pool_*).  Registration is per process and optional: an unregistered
 * range is staged through the library's own pinned scratch.
 */
int			pgv_pinned_register(void *p, size_t bytes);
void		pgv_pinned_unregister(void *p);

/*
 * Create the per-backend GPU context: lazily from _PG_init (src/vector.c:57-65)
 * or on first use.  `stream` is an existing hipStream_t to enqueue on (as
 * void*), PGV_DEFAULT_STREAM for the device's default stream (whose handle is
 * itself NULL), or NULL to create a private non-blocking stream.  A caller that
 * produces inputs / consumes outputs in HBM on its own stream must pass that
 * stream, or bracket calls with its own synchronisation.
 */
#define PGV_DEFAULT_STREAM ((void *) (intptr_t) -1)	/* the device's default (NULL) hipStream_t */
int			pgv_ctx_create(int device, void *stream, pgv_ctx * *out);
void		pgv_ctx_destroy(pgv_ctx * ctx);
int			pgv_ctx_sync(pgv_ctx * ctx);
/* the hipStream_t all work of this context is enqueued on */
void	   *pgv_ctx_stream(pgv_ctx * ctx);
/* time the enclosed GPU work with HIP events on the context's stream (ms) */
int			pgv_timer_start(pgv_ctx * ctx);
int			pgv_timer_stop(pgv_ctx * ctx, float *out_ms);

/*
 * Optional per-kernel accounting for the roofline report: when on, every
 * launch of the row-streaming distance kernel is bracketed by HIP events on the
 * context's stream; pgv_ctx_get_stats synchronises, resolves them and returns
 * the totals since the last pgv_ctx_reset_stats.
 */
typedef struct pgv_stats
{
	/* the IVFFlat list scan (GetScanItems): pgv_scan_lists / pgv_search_batch */
	double		scan_ms;		/* GPU time inside the row-streaming kernel */
	int64_t		scan_launches;
	double		scan_pairs;		/* (row, query) distances it produced */
	double		scan_rows;		/* rows it streamed (a row shared by a query group counts once) */
	/* the same kernel on other inputs: center ranking, exact scans, k-means++ rounds */
	double		aux_ms;
	int64_t		aux_launches;
	double		aux_pairs;
	/* L2 assignment on the matrix cores (pgv_assign / k-means): rows it handled, how many of them the
	 * pre-filter left to the exact recheck of its 4 candidates, and how many the exact kernel redid in full */
	double		assign_redo_rows;
	double		assign_rows;
	double		assign_recheck_rows;
	/* batched list scans: rows of the lists at least one query of the batch probes -- what a single pass
	 * over the probed part of the index would stream (scan_rows / scan_unique_rows = passes) */
	double		scan_unique_rows;
	/* L2 batches scanned on the matrix cores: queries whose k' candidates could not be proven to hold
	 * the whole head and took the exact pass over their segment as well */
	double		scan_redo_queries;
	/* ... and those that a wider candidate set (256 instead of k') did settle, without the exact pass */
	double		scan_widened_queries;
}			pgv_stats;
int			pgv_ctx_set_profiling(pgv_ctx * ctx, int on);
/*
 * The batched list scan (pgv_search_batch / pgv_scan_batch) picks L2 candidates with the matrix cores'
 * |q|^2 + |x|^2 - 2 q.x and then evaluates the reference's sum((q - x)^2) for those only; results are the
 * same as evaluating every row exactly.  on != 0 keeps every row on the exact vector-ALU kernels (A/B
 * measurements, and indexes whose rows mix magnitudes so widely that every query would be redone).
 */
int			pgv_ctx_set_exact_scan(pgv_ctx * ctx, int on);
/*
 * How the MFMA L2 paths (batched list scan, center ranking, assignment pre-filter, pgv_exact_topk) bound the
 * distance between the expansion's value |x|^2 - 2 q.x and the true one when they decide that a candidate set is
 * complete.  Row ids "must match the reference CPU path": with either bound every decision the bound cannot settle
 * goes to the exact kernels; what differs is whether the bound itself can be exceeded.
 *   PGV_BOUND_WORST_CASE (default)  deterministic: the list scan keeps FOUR independent accumulator chains per output,
 *       so its dot product errs by at most gamma_(dim/4 + 4) |q||x| (Higham, Lemma 3.1: any order of n rounded
 *       products), gamma_n = n u / (1 - n u) with u = 2^-24 (the matrix pipeline rounds to nearest: pinned on the
 *       hardware by tests/test_gpu_round4.py); + the row norm's own chain; + the rounding of the exact form
 *       sum((q - x)^2) -- the reference's and ours -- which is relative to the DISTANCE, not to (|q| + |x|)^2.  At 1536
 *       dimensions: 2.3e-5 x 2 |q||x|max, narrower than the statistical band, which is why it can be the default (bench.py `bound_modes`: both side by side; round 3's form of it,
 *       (gamma_(dim+1) + gamma_(dim+2)) (|q| + |x|max)^2 on one chain, flagged every query and cost 98 %).
 *   PGV_BOUND_STATISTICAL  8 sqrt(dim + 4) 2^-24 (|q| + |x|max)^2 -- the probabilistic model of a length-dim fp32
 *       summation (fails with probability ~ e^-32 per sum).
 * A context starts with the deterministic bound EVERYWHERE (round 6), the build's assignment pre-filter included.  That
 * kernel has one accumulator chain per output (128 accumulators a lane leave no room for four), so its band is
 * gamma_(dim+1) (|c|^2 + 2 |a||c|) + gamma_(dim+2) d, 5-10 x wider than the statistical one; what makes it affordable is
 * that an ambiguous row costs an exact recheck of the candidates inside the band (a row's merged list carries 8 of them
 * and the smallest value no list kept) instead of a pass over all centers: +4 % at 1 M x 1000 x 1536 fp32, +10 % at
 * 1.25 M x 4096 x 3072 fp16 over the statistical bound (round 5's form of it: 32 -> 123 ms).  pgv_ctx_set_bound(mode)
 * sets scans and assignment alike.
 * pgv_ctx_set_exact_scan(ctx, 1) remains the mode that uses no expansion at all.
 */
#define PGV_BOUND_STATISTICAL 0
#define PGV_BOUND_WORST_CASE 1
int			pgv_ctx_set_bound(pgv_ctx * ctx, int mode);
int			pgv_ctx_reset_stats(pgv_ctx * ctx);
int			pgv_ctx_get_stats(pgv_ctx * ctx, pgv_stats * out);

/* ------------------------------------------------------- IVFFlat scan side */

/*
 * Upload (or adopt, when the pointers are device memory the index is copied
 * device-to-device) a list-major image of an IVFFlat index:
 *   centers       [nlists x dim]   the Vector inside each IvfflatListData (src/ivfflat.h:270-275)
 *   list_offsets  [nlists + 1]     row range of list l = [off[l], off[l+1]) -- its page chain, flattened
 *   vectors       [n x dim]        index tuple payloads in page-chain order (src/ivfscan.c:139-179)
 *   tids          [n] or NULL      heap TIDs (ItemPointerData widened to 64 bit); NULL -> results carry row slots
 * Replaces the ReadBuffer/LockBuffer page walks of GetScanLists/GetScanItems.
 */
int			pgv_index_upload(pgv_ctx * ctx, pgv_metric metric, pgv_dtype dtype, int dim, int nlists,
							 const void *centers, const int64_t *list_offsets,
							 const void *vectors, const uint64_t *tids, pgv_index * *out);
void		pgv_index_free(pgv_index * index);
/*
 * A second handle on an uploaded index for another context (= another backend's stream and scratch) on the same
 * device: many backends scan ONE device mirror concurrently, each on its own stream -- what shared_buffers is to
 * the reference's scans.  The index is read-only after upload, so no locking is involved.  The handles share the
 * device arrays, which are released by the last pgv_index_free among them (in any order).
 */
int			pgv_index_share(pgv_index * index, pgv_ctx * ctx, pgv_index * *out);
/*
 * Overlapping batches.  A batch of pgv_search_batch is ~25 dependent launches on one stream: center ranking, planning,
 * the list scan (which streams the probed lists once and is HBM-bound), top-k, exact recheck.  Two submitters on two
 * contexts already overlap one batch's ranking / planning / top-k with the other's scan (the scans themselves cannot
 * overlap: both want the whole HBM); lanes > 1 gives ONE caller the same: consecutive pgv_search_batch calls on this
 * handle run on `lanes` internal streams in turn (each with scratch of its own; the index stays one).  With lanes > 1
 *   - a call returns when its work is enqueued; device-side outputs are complete after pgv_ctx_sync(the index's
 *     context) (which waits for the lanes too), NOT merely in the order of that context's stream;
 *   - device-side queries are read no earlier than what the context's stream held at the call (an event orders it);
 *   - host-side outputs make the call synchronous, as always (nothing overlaps then);
 *   - statistics, profiling and the bound / exact-scan settings of the context include the lanes.
 * lanes = 1 (the default) restores the stream-ordered behaviour.  Views (pgv_index_share / _import) have lanes of their
 * own or none.  1 <= lanes <= 4.
 */
int			pgv_index_set_overlap(pgv_index * index, int lanes);
/*
 * The same across PROCESSES -- a Postgres backend is a process (src/ivfscan.c:252-296 runs in each), and its parallel
 * build shares state through a DSM segment (src/ivfbuild.c:830-966).  The whole mirror (centers, vectors, offsets,
 * TIDs, norms) is one device allocation; pgv_index_export wraps its hipIpcMemHandle and the mirror's shape into a
 * fixed-size, position-independent handle that the owner publishes in shared memory; every other process maps the
 * SAME HBM with pgv_index_import and scans it on its own context (stream + scratch).  The index occupies HBM once,
 * whatever the number of backends.
 *   - the exporting process must keep its index (and itself) alive while others have it mapped: inside the server
 *     that is the background worker that staged the index, not a backend that may exit;
 *   - an importer releases its mapping with pgv_index_free; a handle cannot be imported by the process that
 *     exported it (PGV_ERR_STATE: that process uses pgv_index_share);
 *   - the ROCm driver shares memory through dmabuf here: HSA_ENABLE_IPC_MODE_LEGACY=0 in every process.
 */
#define PGV_INDEX_HANDLE_BYTES 256
typedef struct pgv_index_handle
{
	unsigned char bytes[PGV_INDEX_HANDLE_BYTES];
}			pgv_index_handle;
int			pgv_index_export(pgv_index * index, pgv_index_handle * out);
int			pgv_index_import(pgv_ctx * ctx, const pgv_index_handle * handle, pgv_index * *out);
/*
 * Heap TIDs of row slots (what pgv_scan_lists hands back), for a caller that keeps no TID table of its own -- a
 * backend that imported the mirror.  slots / out are host arrays; runs of consecutive slots (a list) are one copy.
 */
int			pgv_index_tids(pgv_index * index, const int64_t *slots, int64_t n, uint64_t *out);
int64_t		pgv_index_rows(const pgv_index * index);
int			pgv_index_lists(const pgv_index * index);

/*
 * GetScanLists (src/ivfscan.c:47-118) for a batch of queries: distance from each
 * query to every center, keep the `maxprobes` nearest, ascending.  Ties on the
 * boundary keep the lower list id (the reference's strict `<` at :92).
 *   queries    [nq x dim]
 *   out_lists  [nq x maxprobes] list ids (maxprobes is clamped to nlists by the caller, :271-278)
 *   out_dist   [nq x maxprobes] or NULL
 */
int			pgv_rank_lists(pgv_index * index, const void *queries, int nq, int maxprobes,
						   int32_t *out_lists, float *out_dist);

/*
 * GetScanItems (src/ivfscan.c:123-187) without the final sort: the distance of
 * every tuple of the given lists to one query, in the order the reference
 * feeds its tuplesort (lists in the given order, page-chain order inside).
 *   out_dist  [m]  m = sum of the list lengths (query NULL -> all 0: ZeroDistance :192-196)
 *   out_slot  [m]  row slot of each tuple (index into vectors/tids)
 *   capacity  size of the out arrays; *out_count = m (PGV_ERR_ARG if m > capacity)
 * The caller widens to float8 and runs the same tuplesort as the reference.
 */
int			pgv_scan_lists(pgv_index * index, const void *query, const int32_t *lists, int nlists,
						   float *out_dist, int64_t *out_slot, int64_t capacity, int64_t *out_count);

/*
 * Fused query path for many backends' queries at once: GetScanLists +
 * GetScanItems + the head of the sorted tuplesort stream (src/ivfscan.c:360-414),
 * i.e. for each query the k nearest tuples of its `probes` nearest lists,
 * ascending (ties: lower insertion position first).
 *   out_dist  [nq x k]   +inf padded when fewer than k tuples were scanned
 *   out_slot  [nq x k]   row slots, -1 padded
 *   out_tid   [nq x k]   or NULL; heap TIDs when the index was uploaded with tids
 */
int			pgv_search_batch(pgv_index * index, const void *queries, int nq, int probes, int k,
							 float *out_dist, int64_t *out_slot, uint64_t *out_tid);

/*
 * The GetScanItems half of pgv_search_batch on its own: the probe lists were chosen
 * elsewhere (pgv_rank_lists, possibly on another GPU: with the index sharded by list,
 * every rank ranks a slice of the queries against the replicated centers, the probe lists
 * are all-gathered, and each rank scans the lists it owns).  Lists that are empty in this
 * image contribute nothing.
 *   probe_lists [nq x probes] list ids, ascending by distance per query
 */
int			pgv_scan_batch(pgv_index * index, const void *queries, int nq, const int32_t *probe_lists,
						   int probes, int k, float *out_dist, int64_t *out_slot, uint64_t *out_tid);

/*
 * One backend's index scan, device-resident between the amgettuple calls (src/ivfscan.c:361-414).
 * This is the path a Postgres backend binds: one query at a time, LIMIT k tuples pulled.
 *   pgv_query_rank   GetScanLists (src/ivfscan.c:47-118): ranks every center against the query
 *                    and keeps the max_probes nearest ON THE DEVICE (one launch, nothing read back).
 *                    query NULL = ZeroDistance (:192-196): lists 0 .. max_probes-1.
 *   pgv_query_scan   GetScanItems + tuplesort_performsort (:123-187) for ranked lists
 *                    [first, first + nprobes): every tuple scored (one launch, the rows spread over
 *                    all CUs), the distances stay in HBM in tuplesort input order, and only the
 *                    `head` nearest come back, ascending (ties: lower insertion position first):
 *                    out_dist / out_slot / out_tid [head], *out_count = min(head, tuples),
 *                    *out_total = tuples in the batch (what the reference's tuplesort holds).
 *   pgv_query_more   entries [skip, skip + count) of the same batch's sorted stream, for an
 *                    executor that pulls past the head; skip + count <= 1024.  Beyond that the
 *                    caller fetches the whole batch with pgv_scan_lists (pgv_query_lists gives the
 *                    ranked list ids) and sorts it like the reference does.
 * Limits of the fused path: max_probes <= 1024, nprobes <= 256, head <= 1024 (PGV_ERR_ARG
 * otherwise: use pgv_rank_lists / pgv_scan_lists).  Iterative scans (:400-406) call
 * pgv_query_scan again with first += nprobes.  out_tid needs an index uploaded with tids.
 */
typedef struct pgv_query pgv_query;
int			pgv_query_begin(pgv_index * index, pgv_query * *out);
void		pgv_query_end(pgv_query * q);
int			pgv_query_rank(pgv_query * q, const void *query, int max_probes);
int			pgv_query_scan(pgv_query * q, int first, int nprobes, int head, float *out_dist, int64_t *out_slot,
						   uint64_t *out_tid, int *out_count, int64_t *out_total);
int			pgv_query_more(pgv_query * q, int skip, int count, float *out_dist, int64_t *out_slot,
						   uint64_t *out_tid, int *out_count);
int			pgv_query_lists(pgv_query * q, int32_t *out_lists, int n);

/* ------------------------------------------------------ IVFFlat build side */

/*
 * The argmin loop of AddTupleToSort (src/ivfbuild.c:183-192) for a batch of
 * heap rows: list id of the nearest center under FUNCTION 1, first
 * strictly-smallest wins.
 *   centers [k x dim], rows [n x dim], out_list [n], out_dist [n] or NULL
 */
int			pgv_assign(pgv_ctx * ctx, pgv_metric metric, pgv_dtype dtype, int dim,
					   const void *centers, int k, const void *rows, int64_t n,
					   int32_t *out_list, float *out_dist);

/*
 * IvfflatKmeans (src/ivfkmeans.c:553-570): k-means++ seeding (:23-91) followed
 * by Lloyd iterations to the reference's stopping rule (:347, :482-483), new
 * centers as fp32 sums in sample order / count (:179-236), empty clusters
 * refilled from the rng (:222-227), spherical variant for ip/cosine (:234-235),
 * CheckCenters (:490-547).  Elkan's bounds (:391-476) are a CPU-side pruning of
 * exactly this computation and are not reproduced: every sample is scored
 * against every center each iteration.
 *   samples      [n x dim] as left by SampleRows (already normalised for ip/cosine, src/ivfbuild.c:154-155)
 *   out_centers  [k x dim] in `dtype`
 *   out_closest  [n] final assignment, or NULL
 *   out_iters    iterations run, or NULL
 * n == 0 -> RandomCenters (:110-133).
 */
int			pgv_kmeans(pgv_ctx * ctx, pgv_ops ops, pgv_dtype dtype, int dim,
					   const void *samples, int n, int k, int max_iterations, const pgv_rng * rng,
					   void *out_centers, int32_t *out_closest, int *out_iters);

/*
 * The two halves of one Lloyd iteration, exposed so that a multi-GPU build can
 * put its all-reduce between them (samples sharded by row, centers replicated):
 *   pgv_lloyd_partial: assignment of this rank's samples to `centers` + this
 *       rank's per-center fp32 sums [k x dim], counts [k] and the number of
 *       samples whose center changed versus io_closest (updated in place).
 *   pgv_lloyd_finish:  centers = sums / counts (inf clamp, empty -> rng, dtype
 *       rounding, renormalise) from the all-reduced sums/counts.
 * All pointers follow the host-or-device rule; with device pointers nothing
 * synchronises, so RCCL can be enqueued on the same stream in between.
 */
int			pgv_lloyd_partial(pgv_ctx * ctx, pgv_ops ops, pgv_dtype dtype, int dim,
							  const void *samples, int n, const void *centers, int k,
							  int32_t *io_closest, float *out_sums, int32_t *out_counts,
							  int64_t *out_changes);
int			pgv_lloyd_finish(pgv_ctx * ctx, pgv_ops ops, pgv_dtype dtype, int dim, int k,
							 const float *sums, const int32_t *counts, const pgv_rng * rng,
							 void *out_centers);

/*
 * InitCenters (src/ivfkmeans.c:23-91) alone: k-means++ seeding under the
 * FUNCTION 3 distance.  out_centers [k x dim].
 */
int			pgv_kmeanspp_init(pgv_ctx * ctx, pgv_ops ops, pgv_dtype dtype, int dim,
							  const void *samples, int n, int k, const pgv_rng * rng,
							  void *out_centers);

/* ------------------------------------------- multi-GPU: one process per GPU */

/*
 * IVFFlat shards naturally: k-means samples and heap rows by row, lists by l % nranks,
 * centers replicated.  The exchange steps of the path live HERE, in the library, enqueued on the
 * context's stream between the kernels that produce and consume their buffers (the reference runs
 * its parallel build inside the extension too, src/ivfbuild.c:830-966):
 *   k-means     per k-means++ round one all-gather of the ranks' weight totals and one of the candidate
 *               rows; per Lloyd iteration ONE all-reduce of sums [k x dim] | counts [k] | changes as a
 *               single fp32 buffer (k*d*4 + k*4 + 4 bytes), no host synchronisation: the host follows
 *               the iteration through a 24-byte record in pinned memory
 *   list scan   all-gather of the probe lists every rank chose for its slice of the query batch, and
 *               of every rank's k x (distance, tid) per query, merged on the device
 * pgv_comm_create uses RCCL over xGMI (librccl is resolved at run time; nothing is linked).
 * pgv_comm_create_custom takes the two collectives from the caller instead -- a Postgres parallel
 * build can back them with its DSM segment, the CPU-side tests with gloo.
 * HNSW search does not shard: replicas only.
 */
typedef struct pgv_comm pgv_comm;

#define PGV_COMM_ID_BYTES 128
/* rank 0 makes the id of a new group (ncclGetUniqueId); the host hands it to every rank by whatever channel it has */
int			pgv_comm_unique_id(void *out_id);
int			pgv_comm_create(pgv_ctx * ctx, int nranks, int rank, const void *unique_id, pgv_comm * *out);

/* both operate on DEVICE buffers and must be ordered after the work already enqueued on `stream` and
 * before whatever is enqueued next (a host implementation synchronises the stream on entry) */
typedef struct pgv_collectives
{
	int			(*all_reduce_sum_f32) (void *state, float *buf, size_t count, void *stream);	/* in place */
	int			(*all_gather) (void *state, const void *send, void *recv, size_t bytes_per_rank, void *stream);
	void	   *state;
}			pgv_collectives;
int			pgv_comm_create_custom(pgv_ctx * ctx, int nranks, int rank, const pgv_collectives * coll, pgv_comm * *out);
void		pgv_comm_destroy(pgv_comm * comm);
int			pgv_comm_size(const pgv_comm * comm);
int			pgv_comm_rank(const pgv_comm * comm);

/*
 * pgv_kmeans with the samples sharded by row: samples_local [n_local x dim] are this rank's rows of the
 * sample (global order = rank order), every rank gets the same centers.  Every rank must pass the same
 * rng stream (seed or callbacks in the same state): the draws steer replicated decisions.
 *   out_centers [k x dim], out_closest_local [n_local] or NULL
 */
int			pgv_kmeans_sharded(pgv_comm * comm, pgv_ops ops, pgv_dtype dtype, int dim,
							   const void *samples_local, int n_local, int k, int max_iterations,
							   const pgv_rng * rng, void *out_centers, int32_t *out_closest_local, int *out_iters);

/*
 * pgv_search_batch over an index whose lists are sharded l % nranks (`local_index` holds all list
 * offsets, foreign lists empty, replicated centers, and was uploaded with tids): every rank passes the
 * SAME batch of nq queries, ranks its slice of it against the centers, scans the probed lists it owns
 * for all of them, and ends up with the merged k nearest (distance, tid) of every query.
 *   out_dist [nq x k], out_tid [nq x k]  (ties: lower rank first)
 */
int			pgv_search_batch_sharded(pgv_comm * comm, pgv_index * local_index, const void *queries, int nq,
									 int probes, int k, float *out_dist, uint64_t *out_tid);

/*
 * The build's tuplesort on the device.  The reference's BuildCallback assigns every heap row to its nearest center
 * and feeds (list, tid, vector) to a tuplesort ordered by list (AddTupleToSort src/ivfbuild.c:161-219, the sort
 * :606-615); InsertTuples (:271-331) then walks the sorted stream list by list into pages.  With the rows in HBM
 * anyway (pgv_assign staged them), the sort is a device gather and its result IS the device mirror:
 *
 *   pgv_builder_begin   centers of the index (after IvfflatKmeans); expected_rows sizes the first allocation.
 *                       centers == NULL: they come later -- until pgv_builder_set_centers, pgv_builder_add only copies
 *                       its rows to the device, on a stream of the builder's own, so that the upload of the heap and a
 *                       pgv_kmeans still running on the context (called from another host thread) overlap; the rows
 *                       are assigned in one piece once the centers are there.  (That stream is not ordered after the
 *                       context's: device-resident rows handed to such a builder must be complete before the call.)
 *   pgv_builder_add     a batch of heap rows (host or device, tightly packed) with their TIDs (or NULL: heap
 *                       positions): copied to the device, assigned there (same kernel as pgv_assign), kept in heap order
 *   pgv_builder_finish  list-major order (ascending list, heap order inside a list -- what the tuplesort delivers
 *                       to InsertTuples), the mirror made from it without a second upload; out_offsets [lists + 1]
 *                       and out_lists [rows, heap order] (either may be NULL) go back to the host
 *   pgv_index_drain     the mirror's rows in list-major order, piece by piece through pinned memory (the next
 *                       piece's copy in flight while the sink runs): what the page writer consumes
 *
 * The builder holds the heap-order copy until finish (2 x the rows in HBM for a moment: 12 GB at 1 M x 1536).
 */
typedef struct pgv_builder pgv_builder;
int			pgv_builder_begin(pgv_ctx * ctx, pgv_metric metric, pgv_dtype dtype, int dim, int nlists, const void *centers,
							  int64_t expected_rows, pgv_builder * *out);
int			pgv_builder_add(pgv_builder * b, const void *rows, const uint64_t *tids, int64_t n);
/* the centers of a builder begun without them (PGV_ERR_STATE otherwise); called once, before pgv_builder_finish */
int			pgv_builder_set_centers(pgv_builder * b, const void *centers);
int64_t		pgv_builder_rows(const pgv_builder * b);
int			pgv_builder_finish(pgv_builder * b, pgv_index * *out_index, int64_t *out_offsets, int32_t *out_lists);
void		pgv_builder_free(pgv_builder * b);
/* sink(arg, first_slot, count, vectors [count x dim] tightly packed, tids [count] or NULL) -> 0 to go on */
typedef int (*pgv_rows_sink) (void *arg, int64_t first_slot, int64_t count, const void *vectors, const uint64_t *tids);
int			pgv_index_drain(pgv_index * index, int64_t chunk_rows, pgv_rows_sink sink, void *arg);

/* ------------------------------------------------- generic candidate batch */

/*
 * One query against n contiguous rows: the exact-scan shape (one fmgr call
 * per heap row, src/vector.c:579-589) and the building block of HNSW candidate
 * scoring.  out[n] = kernel value (L2 squared / -ip / L1).
 */
int			pgv_distance_batch(pgv_ctx * ctx, pgv_metric metric, pgv_dtype dtype, int dim,
							   const void *query, const void *rows, int64_t n, float *out);

/*
 * The exact scan for a BATCH of queries: `ORDER BY embedding <op> $1 LIMIT k` without an index is one fmgr call per
 * heap row (l2_distance src/vector.c:579-589, vector_negative_inner_product :610-620, l1_distance :715-725; the
 * halfvec twins src/halfvec.c:605-748) feeding the executor's top-N heapsort.  nq queries against the same n rows:
 * out_dist [nq x k] ascending kernel values (L2 squared / -ip / L1: the sqrt and the sign stay float8 post-ops of
 * the caller, both monotonic), out_idx [nq x k] row indexes, ties by lower index; entries beyond n are +inf / -1.
 * queries / rows / outputs may be host or device memory.  L2 and inner product batches of 64 queries or more run on
 * the matrix cores (L2 with the exact tail of the list scan, see pgv_ctx_set_bound), everything else on the exact
 * vector-ALU kernels.  k <= 4096.
 */
int			pgv_exact_topk(pgv_ctx * ctx, pgv_metric metric, pgv_dtype dtype, int dim, const void *queries, int nq,
						   const void *rows, int64_t n, int k, float *out_dist, int64_t *out_idx);

/*
 * cosine_distance as the operator computes it without an index (`<=>` in a sequential scan or an
 * executor recheck; src/vector.c:649-697, src/halfvec.c:652-700): three fp32 accumulators in one
 * pass, `(double) sim / sqrt((double) na * (double) nb)`, clamped to [-1, 1], 1 - similarity.  A
 * zero vector gives NaN like the reference (test/expected/vector_type.out:437-441).  out [n] float8.
 */
int			pgv_cosine_distance_batch(pgv_ctx * ctx, pgv_dtype dtype, int dim,
									  const void *query, const void *rows, int64_t n, double *out);

/*
 * hamming_distance / jaccard_distance (`<~>`, `<%>`; src/bitvec.c:45-70 over src/bitutils.c:49-131)
 * of one bit string against n bit strings of the same length: VARBITS payloads, (nbits + 7) / 8
 * bytes each, first bit = most significant bit of byte 0, rows contiguous.  out [n] float8 (exact).
 */
typedef enum pgv_bit_metric
{
	PGV_BIT_HAMMING = 0,
	PGV_BIT_JACCARD = 1
}			pgv_bit_metric;

int			pgv_bit_distance_batch(pgv_ctx * ctx, pgv_bit_metric metric, int nbits,
								   const void *query, const void *rows, int64_t n, double *out);

/* --------------------------------------------------------------- HNSW side */

/*
 * Device mirror of an HNSW index's element vectors (the `Vector data` at byte
 * 72 of every HnswElementTupleData, src/hnsw.h:372-382), addressed by a dense
 * element slot the host keeps per (blkno, offno).
 */
int			pgv_hnsw_upload(pgv_ctx * ctx, pgv_metric metric, pgv_dtype dtype, int dim,
							const void *elements, int64_t n, pgv_hnsw * *out);
void		pgv_hnsw_free(pgv_hnsw * h);
/*
 * The same with a PAYLOAD per element (0..4096 bytes in whole words) kept next to the vectors: what a scan needs to
 * turn a result element into heap TIDs (HnswElementTuple's heaptids[HNSW_HEAPTIDS] and their count,
 * src/hnsw.h:142-153, hnswgettuple src/hnswscan.c:293-311).  It rides in the elements' allocation, so an importing
 * process (pgv_hnsw_import) sees it without a table of its own; pgv_hnsw_get_payload brings the payload rows of a
 * scan's result elements to the host (element slots < 0 give zero bytes).
 */
int			pgv_hnsw_upload_payload(pgv_ctx * ctx, pgv_metric metric, pgv_dtype dtype, int dim, const void *elements,
									int64_t n, const void *payload, int payload_bytes, pgv_hnsw * *out);
int			pgv_hnsw_get_payload(pgv_hnsw * h, const int64_t *elements, int n, void *out);

/*
 * The HNSW mirror across processes, like pgv_index_export / pgv_index_import: the element vectors and the graph
 * (set with pgv_hnsw_set_graph) stay in the exporting process's HBM, importers map them read-only and search on
 * their own context (stream, visited bitmaps).  Graph patches by the owner (pgv_hnsw_update_graph) are seen by the
 * importers; the entry point travels inside the handle, so a new entry point -- like a pgv_hnsw_set_graph by the owner,
 * which reallocates the graph -- needs a new export.
 */
int			pgv_hnsw_export(pgv_hnsw * h, pgv_index_handle * out);
/*
 * A view of a mirror for ANOTHER context of the same process (its own stream, scratch and visited bitmaps; the
 * element rows, the payload and the graph stay the owner's and follow the owner's pgv_hnsw_set_graph /
 * pgv_hnsw_update_graph): pgv_index_share's twin.  Searches, scoring and pgv_hnsw_update_graph work through a view
 * (the patch lands in the owner's graph); pgv_hnsw_set_graph and pgv_hnsw_export do not.  The HNSW build runs the next
 * batch's searches and the graph patches on views while the current batch's lists are replayed
 * (pgvector_amd/host/hnsw_build.c).  One thread per handle at a time; free the views before the owner.
 */
int			pgv_hnsw_share(pgv_hnsw * h, pgv_ctx * ctx, pgv_hnsw * *out);
int			pgv_hnsw_device(const pgv_hnsw * h);	/* the device the mirror lives on (-1: NULL) */
int			pgv_hnsw_import(pgv_ctx * ctx, const pgv_index_handle * handle, pgv_hnsw * *out);

/*
 * The candidate-scoring loop of HnswSearchLayer (src/hnswutils.c:908-934,
 * HnswLoadElementImpl :533-571): distances from queries to gathered elements.
 * Pair i scores element slot[i] against query query_of[i] (query_of == NULL ->
 * all pairs use query 0).  One call serves one expansion step of many
 * concurrent searches.
 *   queries [nq x dim], slot [npairs], query_of [npairs] or NULL, out [npairs]
 */
int			pgv_hnsw_score(pgv_hnsw * h, const void *queries, int nq,
						   const int32_t *slot, const int32_t *query_of, int64_t npairs, float *out);

/*
 * The graph the scan walks, flattened: the meta page's m and entry point
 * (src/hnswutils.c:298-328) and, per element slot, its level and neighbor tuple
 * (src/hnsw.h:384-392: (level + 2) * m slots, layer lc starting at
 * (level - lc) * m, src/hnswutils.c:786; -1 = invalid TID).  Copied to the device.
 *   levels [n], nbr_start [n + 1] (offsets into nbr), nbr [nbr_start[n]]
 */
int			pgv_hnsw_set_graph(pgv_hnsw * h, int m, int32_t entry, const int32_t *levels,
							   const int64_t *nbr_start, const int32_t *nbr);

/*
 * hnswgettuple's first batch on the device (GetScanItems, src/hnswscan.c:25-56):
 * greedy descent with ef = 1 through the upper layers, then HnswSearchLayer with
 * ef_search on layer 0 (src/hnswutils.c:824-987), one workgroup per query, no host
 * round trips.  Needs pgv_hnsw_set_graph.  Queries already normalised for cosine
 * (src/hnswscan.c:92-114).
 *   queries    [nq x dim]
 *   out_elem   [nq x k] element slots nearest first, -1 padded (k <= ef_search)
 *   out_dist   [nq x k] FUNCTION 1 distances, +inf padded
 *   out_scored [nq] or NULL: so->tuples, the number of elements scored on layer 0
 */
int			pgv_hnsw_search(pgv_hnsw * h, const void *queries, int nq, int ef_search, int k,
							int64_t *out_elem, float *out_dist, int64_t *out_scored);

/* ------------------------------------------------------ HNSW build side */

/*
 * HnswFindElementNeighbors' searches (src/hnswutils.c:1280-1357) for a batch of elements being
 * inserted: from the current entry point, greedy descent (ef = 1) through the layers above each
 * element's insert level, then HnswSearchLayer with ef_construction on every layer from
 * min(insert level, entry level) down to 0.  The elements are slots of the mirror (their vectors
 * are the queries); none of them is linked into the graph yet, so none can be found.  Returns each
 * searched layer's W -- the candidate list SelectNeighbors works on -- nearest first.
 *   elements [nq], insert_levels [nq]
 *   out_ids / out_dist [nq x layer_cap x ef_construction], out_count [nq x layer_cap] (0 = layer not searched)
 */
int			pgv_hnsw_build_search(pgv_hnsw * h, const int32_t *elements, const int32_t *insert_levels, int nq,
								  int ef_construction, int layer_cap,
								  int32_t *out_ids, float *out_dist, int32_t *out_count);

/*
 * HnswFindElementNeighbors whole (src/hnswutils.c:1280-1357) for a batch of elements being inserted: the searches of
 * pgv_hnsw_build_search, then SelectNeighbors (:1064-1165; CheckElementCloser :1040-1059) over every searched layer's
 * candidate list ON THE DEVICE -- the candidate lists and their pair distances never leave it.  A new element's list has
 * no cached closer flags, so the selection is the plain sweep: nearest candidate first, closer(e) = no neighbor chosen so
 * far is at distance <= d(e, element) from e; chosen while fewer than lm (2m on layer 0, m above), then the rejected ones
 * fill up to lm, nearest first (:1146-1148); a list of at most lm candidates is taken whole in W's furthest-first order.
 *   elements [nq], insert_levels [nq]
 *   out_ids / out_dist / out_closer [nq x layer_cap x 2m]: the neighbors in the order the reference's r holds them (what
 *       AddConnections stores, :1243-1251), their distances, their `closer` flags; out_count [nq x layer_cap]
 *   out_pairs (or NULL): pair distances computed (profiling)
 */
int			pgv_hnsw_build_neighbors(pgv_hnsw * h, const int32_t *elements, const int32_t *insert_levels, int nq,
									 int ef_construction, int layer_cap,
									 int32_t *out_ids, float *out_dist, uint8_t *out_closer, int32_t *out_count,
									 int64_t *out_pairs);

/*
 * pgv_hnsw_build_neighbors in two halves, for a build that overlaps them (between pgv_hnsw_link_begin and _end): the
 * searches of a batch, whose candidate lists are kept on the device in slot 0 or 1 of the mirror's build state -- when it
 * returns the searches are over and the tuples may be rewritten --, and SelectNeighbors over a kept slot, from any view
 * of the mirror on that view's stream (outputs as pgv_hnsw_build_neighbors).  The searches of batch n + 2 (one view, the
 * other slot) then run beside the selection of batch n + 1 (another view).
 */
int			pgv_hnsw_build_search_keep(pgv_hnsw * h, const int32_t *elements, const int32_t *insert_levels, int nq,
									   int ef_construction, int layer_cap, int slot);
int			pgv_hnsw_build_select_kept(pgv_hnsw * h, int slot, int32_t *out_ids, float *out_dist, uint8_t *out_closer,
									   int32_t *out_count, int64_t *out_pairs);

/*
 * The graph updates of the in-memory build on the device: HnswUpdateNeighborsInMemory -> HnswUpdateConnection
 * (src/hnswbuild.c:376-405, src/hnswutils.c:1183-1231) for a batch of new elements at once -- every neighbor list they
 * chose replayed by one GPU lane with the reference's SelectNeighbors (sorted candidates, cached closer flags, pruned
 * connections kept, the dropped neighbor's place handed to the newcomer), the pair distances CheckElementCloser looks up
 * scored on the device next to it.  Between _begin and _end the mirror carries, per neighbor-tuple slot, the neighbor's
 * distance and closer flag; the tuples themselves (what the searches read) are rewritten in place.
 *
 *   pgv_hnsw_link_begin    after pgv_hnsw_set_graph with every tuple empty (-1): allocates the state
 *   pgv_hnsw_link_prepare  the batch: elements [nq], linked [nq] (0: a duplicate, left out) and their neighbor lists as
 *                          pgv_hnsw_build_neighbors returned them ([nq x layer_cap x 2m] + counts).  Every (new element,
 *                          chosen neighbor, layer) is a link request; they are grouped by neighbor list on the device,
 *                          the newcomers of a list in heap order (as the reference's loop would meet them).  Scores
 *                          every pair the lists' selections can look up -- a list without cached flags: all of them;
 *                          otherwise those that involve a newcomer -- and returns while that runs
 *   pgv_hnsw_link_apply    replays the lists, then puts the batch's own elements in place; sets the entry point.  The
 *                          caller keeps searches that read the old tuples away (they have returned); later searches on
 *                          any stream see the new ones
 *                          Nothing in it waits for the device: it returns with everything enqueued, and a search posted
 *                          right behind it (another view) starts on the device when the last kernel is through
 *   pgv_hnsw_link_end      out_nbr [total slots] (or NULL): the finished tuples; out_pairs / out_deferred (or NULL): the
 *                          member-member pair distances scored for, and the number of, lists whose replay needed the
 *                          second round, over the whole build; frees the state
 * pgv_hnsw_link_prepare's out_pairs (or NULL): pair distances scored for the batch's first round.
 */
int			pgv_hnsw_link_begin(pgv_hnsw * h);
int			pgv_hnsw_link_prepare(pgv_hnsw * h, const int32_t *elements, const uint8_t *linked, int nq, int layer_cap,
								  const int32_t *sel_ids, const float *sel_dist, const uint8_t *sel_closer,
								  const int32_t *sel_count, int64_t *out_pairs);
int			pgv_hnsw_link_apply(pgv_hnsw * h, int32_t entry);
int			pgv_hnsw_link_end(pgv_hnsw * h, int32_t *out_nbr, int64_t *out_pairs, int64_t *out_deferred);

/*
 * Distances between pairs of elements of the mirror, out[i] = d(a[i], b[i]): CheckElementCloser's
 * HnswGetDistance between a candidate and an already selected neighbor (src/hnswutils.c:1040-1059),
 * batched over every pair a batch of inserts can need.
 */
int			pgv_hnsw_score_pairs(pgv_hnsw * h, const int32_t *a, const int32_t *b, int64_t npairs, float *out);
/*
 * The same distances for pairs given as GROUPS (the lists a batch of inserts touches): group g holds the
 * element slots ids[ids_start[g] .. ids_start[g + 1]) = its locals 0 .. n-1 and asks for the pairs (u, v), v < u,
 * of the locals u >= from[g], u ascending then v ascending (from = 1: the whole triangle; from = the number of old
 * members: only the pairs that involve a newcomer).  The pairs are expanded on the device, so a list of n slots
 * crosses the bus instead of up to n (n - 1) / 2 pairs of them.
 *   ids [ids_start[ngroups]], ids_start [ngroups + 1], from [ngroups], pair_start [ngroups + 1] (all host or device)
 *   out [pair_start[ngroups]]: group g's distances start at pair_start[g]
 */
int			pgv_hnsw_score_groups(pgv_hnsw * h, const int32_t *ids, const int64_t *ids_start, const int32_t *from,
								  const int64_t *pair_start, int ngroups, int64_t nids, int64_t npairs, float *out);


/*
 * The graph after a batch of inserts: new entry point and the rewritten neighbor tuples
 * (HnswUpdateNeighborsInMemory / AddConnections, src/hnswbuild.c:376-431).  Element elements[i]'s
 * whole tuple is tuples[tuple_offsets[i] .. tuple_offsets[i + 1]) in the layout of pgv_hnsw_set_graph;
 * tuple_offsets is host memory.  Ordered on the device with later searches through the mirror AND through its views
 * (pgv_hnsw_share), whichever of them ran the patch; searches that still read the old tuples must have returned before
 * the call (every search call ends synchronized), and patches of one mirror are issued one at a time (the caller's
 * business when several threads hold views: pgv_host_hnsw_build gives them all to one helper thread).
 */
int			pgv_hnsw_update_graph(pgv_hnsw * h, int32_t entry, const int32_t *elements, int nupd,
								  const int64_t *tuple_offsets, const int32_t *tuples);

#ifdef __cplusplus
}
#endif
#endif							/* PGV_HIP_H */
