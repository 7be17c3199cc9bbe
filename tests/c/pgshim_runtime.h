/*
 * pgshim_runtime.h -- what tests/c/ext_driver.c uses to play the server around the glue of ext/: a postmaster that
 * owns one shared mapping (shmem structs, LWLocks, latches, the relation catalog with its 8 KB pages, a bump arena the
 * mock device allocates from), forks backends and background workers, and gives every function of ext/shim/pgshim.h a
 * body.  TEST INFRASTRUCTURE: not PostgreSQL, not part of the product.
 */
#ifndef PGSHIM_RUNTIME_H
#define PGSHIM_RUNTIME_H

#include "pgshim.h"

#define SHIM_BLCKSZ 8192
#define SHIM_MAX_RELS 24
#define SHIM_MAX_PROCS 96

/* which opclass an emulated index relation was "created" with: what IvfflatGetTypeInfo / IvfflatOptionalProcInfo /
 * index_getprocinfo answer for it */
typedef struct ShimOpclass
{
	int			am;				/* 0 ivfflat, 1 hnsw */
	int			maxDimensions;	/* IVFFLAT_MAX_DIM (vector), x 2 (halfvec), x 32 (bit) */
	bool		hasNormProc;	/* FUNCTION 2 (cosine: rows stored normalised) */
	bool		hasKmeansNormProc;	/* FUNCTION 4 (spherical k-means) */
	int			distanceFn;		/* 0 l2 squared, 1 negative inner product, 2 l1 */
	int			halfvec;		/* the halfvec opclasses: FUNCTION 1-4 are halfvec_*, FUNCTION 5 (ivfflat) / 3 (hnsw) the type-info function */
}			ShimOpclass;

/* ---- postmaster ---- */
void		shim_postmaster_init(size_t page_store_bytes, size_t arena_bytes);
/* the arena the mock device carves its "device memory" from, so that an exported index is visible in every process
 * forked from the postmaster (NULL / 0 without one) */
void	   *shim_arena_base(size_t *bytes);
/* bytes from a pool inside the postmaster's mapping: zeroed, at the same address in every process, never given back (the
 * stand-in's dynamic shared memory segments; rows a test wants every process to see) */
void	   *shim_shared_alloc(size_t bytes);
void	  **shim_shared_slot(int i);	/* eight pointer-sized words of shared memory for whoever needs to publish an address */
/* runs the shmem request / startup hooks a preloaded library installed */
void		shim_postmaster_run_shmem_hooks(void);
void		shim_register_bgworker_function(const char *name, void (*fn) (Datum));
/* fork a backend that runs fn(arg) under the top-level error handler; its exit code is fn's return value (or 100 for
 * an ERROR that reached the top level, 101 for a FATAL exit) */
int			shim_fork_backend(int (*fn) (void *), void *arg);
/* serve background-worker requests and reap children until every pid in pids[] has exited (their exit codes go to
 * codes[]) or timeout_s passes; returns the number still running (they are killed) */
int			shim_postmaster_wait(const int *pids, int npids, int *codes, double timeout_s);
/* ends every background worker (SIGTERM -> their CHECK_FOR_INTERRUPTS proc_exit) and reaps them */
void		shim_postmaster_shutdown(void);
int			shim_live_bgworkers(void);
/* kill the background workers the hard way (SIGKILL): what a crashed worker looks like to the backends */
void		shim_kill_bgworkers(void);

/* ---- catalog ---- */
/* registers relation `oid` over `nblocks` pages copied from `pages` into the shared page store */
void		shim_create_relation(Oid oid, const ShimOpclass * opclass, const void *pages, uint32_t nblocks, int dimensions);
/* replaces the page image of a relation (insert / vacuum happened) under the relation's exclusive lock */
void		shim_replace_pages(Oid oid, const void *pages, uint32_t nblocks);
const void *shim_relation_pages(Oid oid, uint32_t *nblocks);
Relation	shim_open_relation(Oid oid);	/* this process's Relation for it (NULL when unknown) */
void		shim_drop_relation(Oid oid);	/* + a relcache invalidation to every process */
/* a relcache invalidation through the shared queue: every process sees it at its next AcceptInvalidationMessages or
 * transaction start (shim_relcache_invalidate below fires the callbacks of THIS process at once) */
void		shim_broadcast_relcache_invalidate(Oid relid);
/* every ReadBufferExtended of a background worker sleeps this long: staging a cold, large index takes seconds */
void		shim_set_bgworker_read_delay_us(uint32_t us);

/* reloptions of an index relation: lists (ivfflat) / m, ef_construction (hnsw); 0 = the access method's default.  What
 * the reference's IvfflatGetLists / HnswGetM / HnswGetEfConstruction read through rd_options. */
void		shim_set_reloptions(Oid oid, int a, int b);
/* the tuple descriptor of this runtime: one entry per column (index relations: the indexed column, typmod = dimensions) */
struct TupleDescData
{
	int			natts;
	Oid			types[4];
	int32_t		typmods[4];
};
#define SHIM_VECTOR_TYPE_OID 16385	/* "vector" got some OID at CREATE EXTENSION: anything that is not a built-in type's */

/* ---- the table under CREATE INDEX (reference-linked builds only: pgshim_ref_runtime.c) ----
 * A heap of `nrows` rows, `rows_per_block` to a block; fetch() hands out row r's column value (palloc'd in the current
 * context, toasted or NULL as the test likes) and its TID.  table_index_build_scan / table_index_build_range_scan walk
 * it and call the access method's IndexBuildCallback the way heapam_index_build_range_scan does: every live row, NULLs
 * included, a per-tuple context reset in between. */
typedef struct ShimHeapDef
{
	int64_t		nrows;
	int			rows_per_block;
	void		(*fetch) (int64_t row, Datum *value, bool *isnull, ItemPointerData *tid, void *arg);
	void	   *arg;
}			ShimHeapDef;
#define SHIM_HEAP_OID 999
Relation	shim_heap_relation(const ShimHeapDef * def);
Relation	shim_heap_attach(void);	/* the heap another process defined (a parallel worker's table_open); NULL: none */
/* how many workers plan_create_index_workers grants the next CREATE INDEX of this process (0: a serial build) */
void		shim_set_parallel_workers(int n);
void		ParallelWorkerMain(Datum main_arg);	/* register with shim_register_bgworker_function("ParallelWorkerMain", ..) */
/* input tuple i of a build's tuplesort (before or after tuplesort_performsort: the order they were put in) */
struct Tuplesortstate;
int64_t		shim_tuplesort_inputs(struct Tuplesortstate *state);
void		shim_tuplesort_input(struct Tuplesortstate *state, int64_t i, int32_t *list, ItemPointerData *tid, const void **value);

int64_t		shim_progress_param(int index);	/* the last pgstat_progress_update_param value of a counter */

/* ---- per-process ---- */
/* run fn(arg) with a top-level handler: an ERROR that no PG_TRY caught aborts the "transaction" (buffer pins released,
 * the query context reset -- reset callbacks fire) and makes this return -1 with the message in shim_last_error() */
int			shim_run_toplevel(int (*fn) (void *), void *arg, int *result);
const char *shim_last_error(void);
/* messages below ERROR (NOTICE, WARNING, LOG) this process raised so far whose text starts with `prefix` (the last 32 are kept) */
int			shim_notices_raised(const char *prefix);
/* a fresh child of TopMemoryContext made current (what the executor's per-query context is to ivfflatbeginscan) */
MemoryContext shim_query_context_begin(void);
void		shim_query_context_end(MemoryContext ctx);	/* reset (callbacks fire) + delete, CurrentMemoryContext = Top */
void		shim_context_reset(MemoryContext ctx);
size_t		shim_context_bytes(MemoryContext ctx);
/* the next CHECK_FOR_INTERRUPTS after `after_checks` more calls raises "canceling statement due to user request" */
void		shim_cancel_after(int after_checks);
int			shim_pinned_buffers(void);	/* buffers this process holds pinned right now */
MemoryContext shim_context_create(void);
MemoryContext shim_context_create_generation(void);	/* chunks at increasing addresses (GenerationContextCreate) */
void		shim_context_delete(MemoryContext ctx);
void		shim_set_guc_bool(const char *name, bool value);
void		shim_set_guc_int(const char *name, int value);
int			shim_guc_snapshot(int *out, int cap);	/* every registered GUC of this process, in registration order */
void		shim_guc_restore(const int *in, int n);
void		shim_relcache_invalidate(Oid relid);
void		shim_run_proc_exit(int code);	/* before_shmem_exit + on_proc_exit callbacks (a clean backend exit) */
double		shim_now(void);
/* the List of this runtime (lappend of pgshim_runtime.c; the list_* of pgshim_ref_runtime.c work on the same cells) */
struct List
{
	int			length,
				cap;
	void	  **elems;
};
int			shim_list_length(const List *l);
void	   *shim_list_nth(const List *l, int n);
int			shim_pin_leaks(void);
long		shim_buffer_reads(void);	/* ReadBuffer calls of this process so far */
void		shim_seed_random(uint64 seed);
/* (reference-linked builds only) pg_prng_double / pg_prng_uint32 draw from the caller's generator; NULLs restore the runtime's */
void		shim_prng_hook(double (*next_double) (void *), uint32_t (*next_u32) (void *), void *state);

#endif
