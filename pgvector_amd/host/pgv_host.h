/*
 * pgv_host.h -- C host glue above the libpgv_hip ABI: the parts of pgvector's
 * index access methods that stay on the CPU when the distance loops move to the
 * GPU.  Each function mirrors a reference callback or helper (cited) but works
 * on plain arrays / an emulated 8 KB page image instead of a Relation, because
 * no PostgreSQL headers exist in this build environment (SURVEY hard part 1).
 * The logic is the logic a maintainer pastes into src/ivfscan.c, src/ivfbuild.c
 * and src/hnswutils.c; it is compiled and tested here.
 *
 * Depends only on include/pgv_hip.h.  No CPU distance code: every distance comes
 * from libpgv_hip.
 */
#ifndef PGV_HOST_H
#define PGV_HOST_H

#include <stddef.h>
#include <stdint.h>

#include "../../include/pgv_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

const char *pgv_host_last_error(void);
/* threads the host glue's copy loops use: OpenMP's, capped by the container's CPU quota and by 64 */
int			pgv_host_threads(void);

/* ------------------------------------------------------------------- HNSW */

/*
 * What a scan learns from the index about the graph, flattened: the meta page
 * (src/hnswutils.c:298-328: m, entry point) and, per element, its level and
 * neighbor tuple (src/hnsw.h:384-392: (level + 2) * m index TIDs, layer lc
 * starting at (level - lc) * m, src/hnswutils.c:786).  Elements are addressed by
 * the dense slot the device mirror uses.
 */
typedef struct pgv_hnsw_graph
{
	int64_t		nelements;
	int			m;
	int32_t		entry;			/* entry point slot, -1 for an empty index */
	const int32_t *levels;		/* [nelements] */
	const int64_t *nbr_start;	/* [nelements + 1]: offset of an element's neighbor tuple in nbr */
	const int32_t *nbr;			/* neighbor slots, -1 = invalid TID; layout of HnswNeighborTupleData */
}			pgv_hnsw_graph;

/*
 * hnswgettuple's first batch (src/hnswscan.c:25-56, :189-331) for nq queries at
 * once: greedy descent with ef = 1 on the upper layers, HnswSearchLayer with
 * ef_search on layer 0 (src/hnswutils.c:824-987).  All searches advance in lock
 * step; the unvisited neighbors of every search's current candidate are scored
 * by ONE pgv_hnsw_score call per step, then each search replays the reference's
 * heap logic (:908-976) on its distances -- the distances do not depend on heap
 * state, so the outcome is the reference's.
 *
 *   queries    [nq x dim] host memory, already normalised for cosine (src/hnswscan.c:92-114)
 *   out_elem   [nq x k] element slots nearest first, -1 padded
 *   out_dist   [nq x k] FUNCTION 1 distances, +inf padded
 *   out_scored [nq] or NULL: so->tuples, the number of scored elements
 */
int			pgv_host_hnsw_search(pgv_hnsw * mirror, const pgv_hnsw_graph * graph, pgv_dtype dtype, int dim,
								 const void *queries, int nq, int ef_search, int k,
								 int64_t *out_elem, float *out_dist, int64_t *out_scored);

/*
 * The in-memory phase of CREATE INDEX ... USING hnsw (src/hnswbuild.c:436-476, :376-431) with
 * every distance on the GPU: elements are inserted in batches of up to max_batch (all of a batch
 * search the graph as it stood when the batch began, like the reference's parallel workers racing
 * each other; max_batch = 1 is the reference's serial loop, the cap is 2048), see hnsw_build.c.  A batch never
 * exceeds 1/16 of the elements already linked and ends at the first element taller than the entry point.
 * With max_batch >= 64 the loop is a pipeline once batches are full: two helper threads (each with its own pgv_ctx
 * and a pgv_hnsw_share view of the mirror, made and freed by this call) run the NEXT batch's searches and candidate
 * pair distances, the graph patches and the list records' pair scoring while the calling thread (and its OpenMP team,
 * <= 16 threads) replays the current batch; a batch that runs ahead does not see the batch before it.  The mirror
 * must not be used by other threads during the call.
 *
 *   mirror  pgv_hnsw_upload of ALL n element vectors (what HnswFormIndexValue produced: normalised
 *           for cosine, zero-norm rows left out by the caller); the graph is (re)set by this call
 *   rows    the same vectors in host memory (FindDuplicateInMemory compares bytes, :313-364)
 *   rng     HnswInitElement's level draws (src/hnswutils.c:243-270), one per row in row order
 * Result (malloc'ed; free with pgv_host_hnsw_built_free): levels, nbr_start, nbr in the layout of
 * pgv_hnsw_set_graph, the entry point, dup_of[row] = element that took the row's heap TID or -1.
 */
typedef struct pgv_hnsw_built
{
	int64_t		n;
	int			m;
	int32_t		entry;
	int32_t    *levels;			/* [n] */
	int64_t    *nbr_start;		/* [n + 1] */
	int32_t    *nbr;
	int32_t    *dup_of;			/* [n] */
	int64_t		nelements;		/* elements linked into the graph (n minus duplicates) */
	int64_t		device_pairs;	/* element pairs scored for SelectNeighbors */
	int64_t		batches;
	int64_t		deferred_updates;	/* lists whose re-selection needed a second distance launch */
	double		phase_secs[8];	/* wall time: search, pair scoring, SelectNeighbors, list records, list updates, graph patch,
								 * pair lists, record release */
}			pgv_hnsw_built;

int			pgv_host_hnsw_build(pgv_hnsw * mirror, pgv_dtype dtype, int dim, const void *rows, int64_t n, int m,
								int ef_construction, const pgv_rng * rng, int max_batch, pgv_hnsw_built * out);
void		pgv_host_hnsw_built_free(pgv_hnsw_built * built);
/*
 * Inside a server: a callback pgv_host_hnsw_build polls between batches, on the calling thread (per thread; NULL
 * removes it).  Nonzero ends the build with PGV_ERR_STATE ("cancelled ..."), everything joined and freed; the callback
 * must not longjmp.  The build's helper threads and OpenMP team run with all signals blocked.
 */
typedef int (*pgv_host_cancel_check) (void *arg);
void		pgv_host_hnsw_set_cancel_check(pgv_host_cancel_check check, void *arg);

/*
 * The HNSW index in its on-disk form (src/hnsw.h:40-47, 334-392), either side of the device path.
 *
 * pgv_host_hnsw_write_index: FlushPages (src/hnswbuild.c:300-312) from a built graph -- meta page,
 *   element pages (newest element first, element and neighbor tuple on one page when they fit),
 *   neighbor tuples.  Rows with dup_of[row] >= 0 contribute only their heap TID to that element.
 *   rel is declared below (an array of 8 KB pages).
 * pgv_host_hnsw_stage: one walk over the element pages instead of a tuple at a time
 *   (HnswLoadElement / HnswLoadNeighborTids, src/hnswutils.c:533-571, :761-794): dense slots in page
 *   order, vectors, levels, heap TIDs (UINT64_MAX = invalid) and the neighbor table that
 *   pgv_hnsw_upload + pgv_hnsw_set_graph take.  Deleted elements are left out.
 */
struct pgv_rel;

typedef struct pgv_hnsw_image
{
	pgv_dtype	dtype;
	int			dim;
	int			m;
	int			ef_construction;
	int64_t		n;
	int32_t		entry;			/* slot of the entry point, -1 for an empty index */
	void	   *vectors;		/* [n x dim] */
	int32_t    *levels;			/* [n] */
	int64_t    *nbr_start;		/* [n + 1] */
	int32_t    *nbr;
	uint64_t   *heaptids;		/* [n x 10] (block << 16) | offset */
	uint64_t   *element_tids;	/* [n] index TID of each slot */
}			pgv_hnsw_image;

int			pgv_host_hnsw_write_index(struct pgv_rel * rel, pgv_dtype dtype, int dim, int m, int ef_construction,
									  int64_t n, const void *vectors, const uint64_t *tids, const int32_t *levels,
									  const int64_t *nbr_start, const int32_t *nbr, const int32_t *dup_of,
									  int32_t entry);
int			pgv_host_hnsw_stage(const struct pgv_rel * rel, pgv_dtype dtype, pgv_hnsw_image * out);
void		pgv_host_hnsw_image_free(pgv_hnsw_image * img);

/* ---------------------------------------------------------------- IVFFlat */

/*
 * A relation as an array of 8 KB pages in the exact on-disk format of an IVFFlat
 * index (src/ivfflat.h:46-52, 251-275): block 0 the meta page, blocks >= 1 list
 * pages chained by nextblkno, then per list a chain of entry pages holding
 * standard IndexTuples.  Stands in for Relation + buffer manager.
 */
typedef struct pgv_rel
{
	uint8_t    *pages;
	uint32_t	nblocks;
	uint32_t	cap;
	uint64_t	generation;		/* bumped by every page change: stands in for relcache invalidation */
}			pgv_rel;

#define PGV_BLCKSZ 8192
#define PGV_INVALID_BLOCK 0xFFFFFFFFu

double		pgv_host_ivf_writer_wait_secs(void);	/* diagnostics: the last page writer's wait for its zeroed page array */
void		pgv_rel_init(pgv_rel * rel);
void		pgv_rel_free(pgv_rel * rel);

/*
 * The page-writing tail of CREATE INDEX (src/ivfbuild.c:485-556 CreateMetaPage /
 * CreateListPages, :271-331 InsertTuples, src/ivfutils.c:229-262 IvfflatUpdateList)
 * from data already sorted by list:
 *   centers      [lists x dim] payloads (become the Vector/HalfVector in IvfflatListData)
 *   list_offsets [lists + 1]   rows of list l = [off[l], off[l+1])
 *   vectors      [n x dim]     payloads in sort order;  tids [n] heap TIDs ((block << 16) | offset)
 */
int			pgv_host_ivf_write_index(pgv_rel * rel, pgv_dtype dtype, int dim, int lists,
									 const void *centers, const int64_t *list_offsets,
									 const void *vectors, const uint64_t *tids);

/*
 * The same in steps a build overlaps with the device's work (ivf_pages.c): begin -- the page array for at most
 * max_rows tuples, zeroed by background threads; layout -- meta page, list pages and every entry page's final
 * header / line pointers / chain link from the list lengths alone; fill -- the tuples of any run of slots of the
 * sorted stream, rows in parallel; end.  pgv_host_ivf_write_index is begin + layout + one fill + end.
 */
typedef struct pgv_ivf_writer pgv_ivf_writer;
int			pgv_host_ivf_writer_begin(pgv_rel * rel, pgv_dtype dtype, int dim, int lists, int64_t max_rows,
									  pgv_ivf_writer * *out);
int			pgv_host_ivf_writer_layout(pgv_ivf_writer * w, const void *centers, const int64_t *list_offsets);
int			pgv_host_ivf_writer_fill(pgv_ivf_writer * w, int64_t first_slot, int64_t count, const void *vectors,
									 const uint64_t *tids);
int			pgv_host_ivf_writer_end(pgv_ivf_writer * w);

/* single-row insert (src/ivfinsert.c:72-181): append to the list's insert page, extending the chain */
int			pgv_host_ivf_insert(pgv_rel * rel, pgv_dtype dtype, int list, const void *vector, uint64_t tid);

/*
 * ivfflatbulkdelete (src/ivfvacuum.c:18-143): walk every list's entry pages, drop the index
 * tuples whose heap TID the callback reports dead (PageIndexMultiDelete: the survivors keep
 * their order and are renumbered), and point the list's insertPage at the first page that lost
 * tuples so that later inserts refill it (IvfflatUpdateList with no original insert page,
 * src/ivfutils.c:252-261).
 */
typedef int (*pgv_host_dead_fn) (uint64_t tid, void *state);
int			pgv_host_ivf_bulkdelete(pgv_rel * rel, pgv_host_dead_fn dead, void *state,
									int64_t *tuples_removed, int64_t *num_index_tuples);

/*
 * Staging: walk the page chains exactly as GetScanLists / GetScanItems do
 * (src/ivfscan.c:58-111, :139-179), strip IndexTuple and varlena headers (both the
 * 4-byte and the 1-byte short form index_form_tuple produces for dim <= 29) and
 * emit the contiguous list-major image pgv_index_upload takes.  Output arrays are
 * malloc'ed; free with pgv_host_ivf_image_free.
 */
typedef struct pgv_ivf_image
{
	pgv_dtype	dtype;
	int			dim;
	int			lists;
	int64_t		nrows;
	void	   *centers;		/* [lists x dim] */
	int64_t    *list_offsets;	/* [lists + 1] */
	void	   *vectors;		/* [nrows x dim] */
	uint64_t   *tids;			/* [nrows] */
	uint32_t   *start_pages;	/* [lists] list->startPage */
}			pgv_ivf_image;

int			pgv_host_ivf_stage(const pgv_rel * rel, pgv_dtype dtype, pgv_ivf_image * out);
void		pgv_host_ivf_image_free(pgv_ivf_image * img);

/*
 * Device mirror lifecycle: the staged image and its pgv_index, rebuilt
 * whenever the relation's pages changed since the last staging (insert, vacuum, rebuild) --
 * the role a relcache invalidation callback plays inside a server.  pgv_host_ivf_mirror_get
 * returns the current index and image; both stay valid until the next _get or _close.
 */
typedef struct pgv_ivf_mirror pgv_ivf_mirror;

int			pgv_host_ivf_mirror_open(pgv_ctx * ctx, pgv_metric metric, pgv_dtype dtype, pgv_ivf_mirror * *out);
int			pgv_host_ivf_mirror_get(pgv_ivf_mirror * mirror, const pgv_rel * rel,
									pgv_index * *out_index, const pgv_ivf_image * *out_image);
int64_t		pgv_host_ivf_mirror_restages(const pgv_ivf_mirror * mirror);	/* how many times it (re)staged */
void		pgv_host_ivf_mirror_close(pgv_ivf_mirror * mirror);

/*
 * ivfflatbeginscan / ivfflatgettuple / ivfflatendscan (src/ivfscan.c:252-431) on a
 * staged mirror: probes / max_probes / iterative_scan are the GUCs of
 * src/ivfflat.c:38-59, normalize_query is set for opclasses with a NORM_PROC
 * (cosine).  gettuple returns 1 and a heap TID, or 0 when the scan is exhausted.
 */
typedef struct pgv_ivf_scan pgv_ivf_scan;

int			pgv_host_ivf_beginscan(pgv_index * mirror, const pgv_ivf_image * img, int probes, int max_probes,
								   int iterative, int normalize_query, pgv_ivf_scan * *out);
/* amrescan: a new ORDER BY value (NULL = SQL NULL: every tuple at distance 0, :192-196) */
int			pgv_host_ivf_rescan(pgv_ivf_scan * scan, const void *query);
int			pgv_host_ivf_gettuple(pgv_ivf_scan * scan, uint64_t *out_tid, double *out_distance);

/*
 * A pooler in front of the batched scan (ivf_pool.c): backends hand in one query each (what ivfflatgettuple's
 * caller has, src/ivfscan.c:361-414) and block; queries that arrive within max_wait_us of a batch's first one, or
 * until max_batch are waiting, share one pgv_search_batch -- every pass over a probed list serves all of them.
 * `lanes` batches are in flight at once.
 *
 * Backends are PROCESSES (src/ivfscan.c:252-296 runs in each), so all state the two sides exchange lives in one
 * position-independent shared segment the caller provides -- inside the server a DSM / ShmemInitStruct block, the
 * pattern of the reference's parallel build (src/ivfbuild.c:830-966):
 *   pgv_host_pool_shm_bytes / _shm_init   size and format a segment (no GPU involved)
 *   pgv_host_pool_attach / _detach        a process's handle on a formatted segment mapped at any address
 *   pgv_host_pool_search                  a backend's query: touches the segment only, needs NO GPU context
 *   pgv_host_pool_serve                   the leader loop of one lane, run by whoever owns a GPU context and a handle
 *                                         on the device mirror (a background worker: the owner's pgv_index_share
 *                                         view, or a pgv_index_import view in another process); returns at shutdown
 *   pgv_host_pool_publish_index / _index_handle   the owner of the mirror leaves its pgv_index_export handle in the
 *                                         segment, other serving processes fetch it (waiting up to wait_ms)
 *   pgv_host_pool_shutdown                wakes everybody; searches fail with PGV_ERR_STATE from then on
 * pgv_host_pool_create / _destroy wrap all of it for one process: an anonymous shared segment + one serving thread
 * per lane (context + pgv_index_share view each); children forked afterwards are clients of the same pool.
 *   out_tid / out_dist [k]: the head of GetScanItems + tuplesort for this query, padded with ~0 / +inf
 */
typedef struct pgv_pool pgv_pool;
size_t		pgv_host_pool_shm_bytes(pgv_dtype dtype, int dim, int k, int max_batch, int lanes);
int			pgv_host_pool_shm_init(void *shm, size_t bytes, pgv_dtype dtype, int dim, int probes, int k, int max_batch,
								   int max_wait_us, int lanes);
int			pgv_host_pool_attach(void *shm, size_t bytes, pgv_pool * *out);
void		pgv_host_pool_detach(pgv_pool * pool);
int			pgv_host_pool_publish_index(pgv_pool * pool, const pgv_index_handle * handle);
int			pgv_host_pool_index_handle(pgv_pool * pool, int wait_ms, pgv_index_handle * out);
int			pgv_host_pool_serve(pgv_pool * pool, int lane, pgv_index * view);
/* one serving thread per lane in the calling process, which owns `index` (a context + pgv_index_share view each) */
int			pgv_host_pool_start_threads(pgv_pool * pool, pgv_index * index, int device);
void		pgv_host_pool_shutdown(pgv_pool * pool);
int			pgv_host_pool_is_shut_down(pgv_pool * pool);
int			pgv_host_pool_create(pgv_index * index, int device, pgv_dtype dtype, int dim, int probes, int k,
								 int max_batch, int max_wait_us, int lanes, pgv_pool * *out);
int			pgv_host_pool_search(pgv_pool * pool, const void *query, uint64_t *out_tid, float *out_dist);
void		pgv_host_pool_stats(pgv_pool * pool, int64_t *batches, int64_t *queries);
void		pgv_host_pool_destroy(pgv_pool * pool);

void		pgv_host_ivf_endscan(pgv_ivf_scan * scan);

/*
 * BuildIndex (src/ivfbuild.c:1040-1058) end to end on the GPU: k-means on the given
 * samples (ComputeCenters), assignment of every heap row in batches (AssignTuples /
 * AddTupleToSort), sort by list (tuplesort on Int4LessOperator, stable here) and the
 * page writers above.  rows/tids are the non-NULL heap tuples in heap order, samples the
 * rows SampleRows drew; both are taken AS STORED IN THE HEAP: for the cosine opclass
 * zero-norm rows are skipped and the rest stored normalised (BuildCallback,
 * src/ivfbuild.c:174-180), for the inner-product and cosine opclasses zero-norm samples
 * are skipped and the rest normalised before k-means (SampleCallback, :148-156) -- done
 * here, not by the caller.  Already normalised input passes through unchanged up to one
 * rounding.
 */
int			pgv_host_ivf_build(pgv_ctx * ctx, pgv_ops ops, pgv_dtype dtype, int dim, int lists,
							   const void *rows, const uint64_t *tids, int64_t n,
							   const void *samples, int nsamples, const pgv_rng * rng, pgv_rel * out_rel);
/*
 * The same, and the device mirror of the new index with it: the rows never leave HBM between the assignment and the
 * mirror (pgv_builder_*), the pages are written from the mirror's list-major rows as they come back
 * (pgv_index_drain -> pgv_host_ivf_writer_fill) -- no host sort, no staging pass, no second upload.  *out_index is
 * owned by the caller (pgv_index_free); it equals what pgv_host_ivf_stage + pgv_index_upload of out_rel give.
 */
int			pgv_host_ivf_build_mirror(pgv_ctx * ctx, pgv_ops ops, pgv_dtype dtype, int dim, int lists,
									  const void *rows, const uint64_t *tids, int64_t n,
									  const void *samples, int nsamples, const pgv_rng * rng, pgv_rel * out_rel,
									  pgv_index * *out_index);
/* seconds the last pgv_host_ivf_build on this thread spent per phase:
 * [0] normalise [1] k-means [2] copy to the device + assignment [3] order by list on the device (the mirror)
 * [4] page writer (layout + the rows coming back + tuples) */
void		pgv_host_ivf_build_phases(double out_secs[5]);
/* l2_normalize / halfvec_l2_normalize of one value (src/vector.c:785-819, src/halfvec.c:724-759);
 * returns 0 when the norm is zero (the value is then all zeros) */
int			pgv_host_normalize_value(pgv_dtype dtype, int dim, const void *in, void *out);

#ifdef __cplusplus
}
#endif
#endif
