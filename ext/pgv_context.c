/*
 * pgv_context.c -- GUCs, the per-backend GPU context, and the device mirrors shared by every backend.
 *
 * A Postgres backend is a process (src/ivfscan.c:252-296 runs in each).  The device mirror of an index must not
 * be uploaded per connection (6 GB and ~0.35 s for the headline index), and it must outlive the backend that
 * first asked for it.  So mirrors are owned by a BACKGROUND WORKER per database ("pgvector gpu"): it stages the
 * index through the buffer manager (the walks of src/ivfscan.c:58-111,139-179), uploads it with pgv_index_upload,
 * and publishes the pgv_index_export handle in a registry in shared memory -- the DSM/shared-state pattern of the
 * reference's parallel build (src/ivfbuild.c:830-966).  Backends map the SAME HBM with pgv_index_import and scan
 * it on their own context; the index occupies HBM once however many backends there are.
 *
 * Staleness: ivfflatinsert / ivfflatbulkdelete / hnswinsert / hnswbulkdelete change pages without a relcache
 * invalidation.  They call PgvNoteIndexChange (one atomic increment of the entry's generation); a scan that
 * finds generation != stagedGeneration asks the worker to restage and meanwhile runs on the reference's CPU
 * path (PgvIvfflatGetMirror returns NULL, the hook leaves so->gpu NULL) -- results are always those of the
 * current pages, the GPU serves whenever its mirror is current.  The worker restages at most once per
 * vector.gpu_restage_delay_ms, so an insert-heavy index does not restage per insert.
 *
 * Twin over the emulated page image: pgvector_amd/host/ivf_mirror.c + the stager of ivf_pages.c; the
 * cross-process export/import is exercised by tests/test_gpu_round3.py with real processes.
 */
#include "pgv_gpu.h"

#include "access/genam.h"
#include "access/xact.h"
#include "miscadmin.h"
#include "port/atomics.h"
#include "postmaster/bgworker.h"
#include "storage/bufmgr.h"
#include "storage/ipc.h"
#include "storage/latch.h"
#include "storage/lwlock.h"
#include "storage/shmem.h"
#include "utils/guc.h"
#include "utils/inval.h"
#include "utils/memutils.h"
#include "utils/timestamp.h"

#include <errno.h>
#include <signal.h>

bool		vector_gpu = false;
int			vector_gpu_device = -1;
int			vector_gpu_build_devices = 0;
int			vector_gpu_stage_wait_ms = 0;
int			vector_gpu_restage_delay_ms = 1000;
bool		vector_gpu_pooled = false;
int			vector_gpu_max_own_contexts = 4;	/* GUC vector.gpu_max_own_contexts */
int			vector_gpu_hnsw_build_batch = 1024;	/* GUC vector.gpu_hnsw_build_batch */
bool		vector_gpu_kmeans = true;	/* GUC vector.gpu_kmeans */

#define PGV_MAX_MIRRORS 64

typedef enum PgvMirrorState
{
	PGV_MIRROR_EMPTY = 0,
	PGV_MIRROR_REQUESTED,		/* a backend wants it (re)staged */
	PGV_MIRROR_STAGING,
	PGV_MIRROR_READY,
	PGV_MIRROR_FAILED			/* staging raised an error: stays on the CPU path until the next change */
}			PgvMirrorState;

/* one index's entry in the registry; everything but `generation` is guarded by PgvShared->lock */
typedef struct PgvSharedMirror
{
	Oid			dboid;
	Oid			relid;
	int			kind;			/* PGV_KIND_IVFFLAT / PGV_KIND_HNSW: which staging the worker runs */
	int			state;
	pg_atomic_uint64 generation;	/* bumped by PgvNoteIndexChange */
	uint64		stagedGeneration;	/* value of `generation` the published mirror was staged at */
	TimestampTz stagedAt;
	pgv_index_handle handle;
	int			lists;
	int			dimensions;
	int			dtype;
	int			metric;
	int64		ntuples;
}			PgvSharedMirror;

/*
 * The pooler inside the server (twin over plain processes: pgvector_amd/host/ivf_pool.c + tools/pgv_backend.c).
 * With vector.gpu_pooled a backend needs no GPU context: it puts its query into a slot, sets the worker's latch and
 * sleeps on its own; the worker -- the owner of the mirrors -- takes every filled slot of one (index, probes) it
 * finds, answers them with ONE pgv_search_batch and wakes their backends.  Measured with plain processes (DESIGN
 * 4.8b): own-context backends top out at 4 (50 k QPS) and then lower the total, pooled clients reach 75 k at 64
 * and 120 k at 256.  The batch window is the duration of the previous batch: nothing waits for stragglers.
 */
#define PGV_POOL_SLOTS 256
#define PGV_POOL_MAX_BATCH 256
#define PGV_POOL_ROW_BYTES (IVFFLAT_MAX_DIM * sizeof(float))	/* vector 2000 x 4 = halfvec 4000 x 2 */

typedef enum PgvSlotState
{
	PGV_SLOT_FREE = 0,
	PGV_SLOT_CLAIMED,			/* a backend is filling it */
	PGV_SLOT_FILLED,			/* waiting for the worker */
	PGV_SLOT_TAKEN,				/* part of the batch being answered */
	PGV_SLOT_DONE,				/* results are in */
	PGV_SLOT_UNSERVED,			/* no current mirror (stale, failed): the backend runs the reference's path */
	PGV_SLOT_ABANDONED			/* its backend was cancelled while the worker had it: the worker frees it */
}			PgvSlotState;

typedef struct PgvPoolSlot
{
	pg_atomic_uint32 state;
	Oid			dboid;
	Oid			relid;
	int			probes;
	int			nullQuery;		/* ZeroDistance (src/ivfscan.c:192-196): not pooled, the backend is told so */
	Latch	   *latch;			/* the backend's (in shared memory: &MyProc->procLatch) */
	int			count;			/* results: min(PGV_POOL_HEAD, tuples of the probed lists) */
	int64		total;			/* tuples of the probed lists */
	uint64		staged;			/* the staging (generation + 1) of the mirror that answered */
	float		dist[PGV_POOL_HEAD];
	uint64		tid[PGV_POOL_HEAD];
	char		payload[PGV_POOL_ROW_BYTES];
}			PgvPoolSlot;

/* a worker whose process exists but whose heartbeat is older than this is hung (it beats every 200 ms in its loop and
 * once per page while staging; the longest synchronous call, the upload of an index that fills the HBM, takes seconds) */
#define PGV_WORKER_HUNG_MS 60000
/* a pooled query the worker has not TAKEN after this long (it is staging some index: seconds) is taken back by its
 * backend, which scans on its own context or on the reference's path */
#define PGV_POOL_PATIENCE_MS 500
#define PGV_SPAWN_GAP_MS 500	/* at most one RegisterDynamicBackgroundWorker per this long */

typedef struct PgvSharedState
{
	LWLock	   *lock;
	Latch	   *workerLatch[PGV_MAX_MIRRORS];	/* per database: index = slot of the first entry of that database */
	Oid			workerDb[PGV_MAX_MIRRORS];
	int			workerPid[PGV_MAX_MIRRORS];
	pg_atomic_uint64 workerBeat[PGV_MAX_MIRRORS];	/* GetCurrentTimestamp() of the worker's last loop turn */
	pg_atomic_uint64 lastSpawn;	/* when a backend last asked the postmaster for a worker (any database) */
	pg_atomic_uint64 ownContexts;	/* backends that hold a device context of their own (PgvTryGetContext) */
	PgvSharedMirror mirrors[PGV_MAX_MIRRORS];
	PgvPoolSlot pool[PGV_POOL_SLOTS];
}			PgvSharedState;

static PgvSharedState * PgvShared = NULL;
static shmem_request_hook_type prev_shmem_request_hook = NULL;
static shmem_startup_hook_type prev_shmem_startup_hook = NULL;

static pgv_ctx *backend_ctx = NULL;
static bool backend_ctx_counted = false;	/* this backend's context is in PgvShared->ownContexts */
static int	workerSlot = -1;	/* this process's registration when it is a GPU worker */
static PgvIvfMirror *mirrors = NULL;	/* this backend's imported views */

/* ------------------------------------------------------------------ shared memory */

static void
PgvShmemRequest(void)
{
	if (prev_shmem_request_hook)
		prev_shmem_request_hook();
	RequestAddinShmemSpace(sizeof(PgvSharedState));
	RequestNamedLWLockTranche("pgvector_gpu", 1);
}

static void
PgvShmemStartup(void)
{
	bool		found;

	if (prev_shmem_startup_hook)
		prev_shmem_startup_hook();
	LWLockAcquire(AddinShmemInitLock, LW_EXCLUSIVE);
	PgvShared = ShmemInitStruct("pgvector gpu mirrors", sizeof(PgvSharedState), &found);
	if (!found)
	{
		memset(PgvShared, 0, sizeof(PgvSharedState));
		PgvShared->lock = &(GetNamedLWLockTranche("pgvector_gpu"))->lock;
		for (int i = 0; i < PGV_MAX_MIRRORS; i++)
		{
			pg_atomic_init_u64(&PgvShared->mirrors[i].generation, 0);
			pg_atomic_init_u64(&PgvShared->workerBeat[i], 0);
		}
		pg_atomic_init_u64(&PgvShared->lastSpawn, 0);
		for (int i = 0; i < PGV_POOL_SLOTS; i++)
			pg_atomic_init_u32(&PgvShared->pool[i].state, PGV_SLOT_FREE);
	}
	LWLockRelease(AddinShmemInitLock);
}

/* the entry of (MyDatabaseId, relid), created on demand; NULL when the registry is full or not configured */
static PgvSharedMirror *
PgvFindEntryKind(Oid relid, bool create, int kind)
{
	PgvSharedMirror *free_entry = NULL;

	if (PgvShared == NULL)
		return NULL;
	for (int i = 0; i < PGV_MAX_MIRRORS; i++)
	{
		PgvSharedMirror *e = &PgvShared->mirrors[i];

		if (e->relid == relid && e->dboid == MyDatabaseId)
			return e;
		if (e->relid == 0 && free_entry == NULL)
			free_entry = e;
	}
	if (create && free_entry)
	{
		free_entry->dboid = MyDatabaseId;
		free_entry->relid = relid;
		free_entry->kind = kind;
		free_entry->state = PGV_MIRROR_EMPTY;
		free_entry->stagedGeneration = 0;
	}
	return create ? free_entry : NULL;
}

static PgvSharedMirror *
PgvFindEntry(Oid relid, bool create)
{
	return PgvFindEntryKind(relid, create, PGV_KIND_IVFFLAT);
}

/* entries of the registry in use, all databases (monitoring; 64 for the cluster: PGV_MAX_MIRRORS) */
int
PgvRegistryEntries(void)
{
	int			used = 0;

	if (PgvShared == NULL)
		return 0;
	LWLockAcquire(PgvShared->lock, LW_SHARED);
	for (int i = 0; i < PGV_MAX_MIRRORS; i++)
		used += PgvShared->mirrors[i].relid != InvalidOid;
	LWLockRelease(PgvShared->lock);
	return used;
}

void		PgvReleaseIdleContext(void);

/* ivfflatinsert (src/ivfinsert.c:72-181), ivfflatbulkdelete (src/ivfvacuum.c:18-143), hnswinsert, hnswbulkdelete and
 * ambuild call this after changing pages: mirrors staged before now are stale */
void
PgvNoteIndexChange(Relation index)
{
	PgvSharedMirror *e;

	/* ambuild ends here too (the last statement of ivfflatbuild / hnswbuild): the build's device context goes unless this
	 * backend's scans hold imports on it (PgvReleaseIdleContext; an insert or vacuum in a scanning backend keeps it) */
	PgvReleaseIdleContext();
	if (PgvShared == NULL)
		return;
	LWLockAcquire(PgvShared->lock, LW_SHARED);
	e = PgvFindEntry(RelationGetRelid(index), false);
	if (e)
		pg_atomic_fetch_add_u64(&e->generation, 1);
	LWLockRelease(PgvShared->lock);
}

/* DROP INDEX / REINDEX reach every backend as a relcache invalidation: forget the local view.  The worker is a backend
 * too: it looks whether the indexes of the mirrors it owns still exist (PgvWorkerDropGone) -- a dropped index must not
 * keep its mirror in HBM, nor its entry in the registry, for the worker's lifetime. */
static bool ownedCheck = false;	/* (worker) some relation changed or went: see whether the owned mirrors' indexes still exist */

static void
PgvRelcacheCallback(Datum arg, Oid relid)
{
	(void) arg;
	ownedCheck = true;
	for (PgvIvfMirror * m = mirrors; m != NULL; m = m->next)
		if (m->relid == relid || relid == 0)
			m->valid = false;	/* (an import open scans hold is retired, not unmapped, at the next PgvIvfflatGetMirror) */
	PgvHnswInvalidate(relid);
}

static PgvPoolSlot * myPoolSlot = NULL;	/* the slot this backend has claimed and not given back */

/* a backend that exits with a query in the pool (FATAL skips PG_CATCH) gives its slot back: an unclaimed one is free at
 * once, one the worker holds is freed by the worker (PGV_SLOT_ABANDONED) */
static void
PgvReleaseMyPoolSlot(int code, Datum arg)
{
	(void) code;
	(void) arg;
	if (myPoolSlot != NULL)
	{
		uint32		filled = PGV_SLOT_FILLED,
					taken = PGV_SLOT_TAKEN;

		/* the sequence of PgvPoolSearch's PG_CATCH: a slot the worker has not taken is free at once (FILLED -> FREE by
		 * CAS: the worker may be taking it this instant), one it holds is left for it to free (TAKEN -> ABANDONED),
		 * anything else (CLAIMED, DONE, UNSERVED) is this backend's alone */
		if (!pg_atomic_compare_exchange_u32(&myPoolSlot->state, &filled, PGV_SLOT_FREE) &&
			!pg_atomic_compare_exchange_u32(&myPoolSlot->state, &taken, PGV_SLOT_ABANDONED))
			pg_atomic_write_u32(&myPoolSlot->state, PGV_SLOT_FREE);
		myPoolSlot = NULL;
	}
}

static void
PgvAtExit(int code, Datum arg)
{
	(void) code;
	(void) arg;
	for (PgvIvfMirror * m = mirrors; m != NULL; m = m->next)
		if (m->index)
			pgv_index_free(m->index);	/* unmaps the import; the worker's allocation stays */
	if (backend_ctx)
		pgv_ctx_destroy(backend_ctx);
	backend_ctx = NULL;
	if (backend_ctx_counted && PgvShared != NULL)
		pg_atomic_fetch_add_u64(&PgvShared->ownContexts, -1);
	backend_ctx_counted = false;
}

void
PgvGpuInit(void)
{
	DefineCustomBoolVariable("vector.gpu", "Runs the distance hot path on the GPU (libpgv_hip)", NULL,
							 &vector_gpu, false, PGC_USERSET, 0, NULL, NULL, NULL);
	DefineCustomIntVariable("vector.gpu_device", "HIP device of this backend (-1: device 0, and parallel build worker w takes device (w + 1) mod the device count)", NULL,
							&vector_gpu_device, -1, -1, 63, PGC_USERSET, 0, NULL, NULL, NULL);
	DefineCustomIntVariable("vector.gpu_build_devices", "Devices an ivfflat build's k-means is sharded over (0: every device of the node, 1: this backend's only)", NULL,
							&vector_gpu_build_devices, 0, 0, 16, PGC_USERSET, 0, NULL, NULL, NULL);
	DefineCustomIntVariable("vector.gpu_stage_wait_ms", "How long a scan waits for the GPU worker to stage a mirror before it runs on the CPU", NULL,
							&vector_gpu_stage_wait_ms, 0, 0, 600000, PGC_USERSET, 0, NULL, NULL, NULL);
	DefineCustomIntVariable("vector.gpu_restage_delay_ms", "Minimum time between two stagings of one index", NULL,
							&vector_gpu_restage_delay_ms, 1000, 0, 3600000, PGC_USERSET, 0, NULL, NULL, NULL);
	DefineCustomBoolVariable("vector.gpu_pooled", "Index scans hand their query to the GPU worker, which batches the queries of all backends", NULL,
							 &vector_gpu_pooled, false, PGC_USERSET, 0, NULL, NULL, NULL);
	DefineCustomIntVariable("vector.gpu_max_own_contexts", "Backends that may scan on a device context of their own; the others go through the pooler", NULL,
							&vector_gpu_max_own_contexts, 4, 0, 64, PGC_USERSET, 0, NULL, NULL, NULL);
	DefineCustomBoolVariable("vector.gpu_kmeans", "An ivfflat build computes its centers on the GPU (off: the CPU build's Elkan k-means and its exact centers; the rows are still assigned on the GPU)", NULL,
							 &vector_gpu_kmeans, true, PGC_USERSET, 0, NULL, NULL, NULL);
	DefineCustomIntVariable("vector.gpu_hnsw_build_batch", "Elements an hnsw build links at once on the GPU (1: the serial build's insertion order exactly)", NULL,
							&vector_gpu_hnsw_build_batch, 1024, 1, 16384, PGC_USERSET, 0, NULL, NULL, NULL);
	CacheRegisterRelcacheCallback(PgvRelcacheCallback, (Datum) 0);
	on_proc_exit(PgvAtExit, (Datum) 0);
	before_shmem_exit(PgvReleaseMyPoolSlot, (Datum) 0);
	/* the registry needs shared memory: effective when the library is in shared_preload_libraries; otherwise
	 * PgvShared stays NULL and every scan stays on the CPU path */
	if (process_shared_preload_libraries_in_progress)
	{
		prev_shmem_request_hook = shmem_request_hook;
		shmem_request_hook = PgvShmemRequest;
		prev_shmem_startup_hook = shmem_startup_hook;
		shmem_startup_hook = PgvShmemStartup;
	}
}

/*
 * The device of this backend.  vector.gpu_device pins it; -1 (the default) means device 0 -- except in the workers of a
 * parallel CREATE INDEX (src/ivfbuild.c:830-966: the leader scans and assigns too), which spread over the node:
 * worker w takes device (w + 1) mod the device count, the leader keeps 0.  Assignment, the build's dominant cost, needs
 * no exchange between participants: every participant assigns the heap rows it scans on ITS device and
 * feeds the shared tuplesort like the reference's workers do.  The worker that stages mirrors and the scans stay on
 * device 0 (one mirror per index, on one device).
 */
int
PgvMyDevice(void)
{
	int			n;

	if (vector_gpu_device >= 0)
		return vector_gpu_device;
	if (ParallelWorkerNumber < 0)
		return 0;
	n = pgv_device_count();
	return n > 1 ? (ParallelWorkerNumber + 1) % n : 0;
}

pgv_ctx *
PgvGetContext(void)
{
	if (backend_ctx == NULL && pgv_ctx_create(PgvMyDevice(), NULL, &backend_ctx) != PGV_OK)
		ereport(ERROR, (errmsg("pgvector GPU path: %s", pgv_last_error())));
	return backend_ctx;
}

/*
 * The same for callers that have the reference's code to go back to (every scan and build hook): no device, a device
 * that was lost, a driver that will not initialise -- `vector.gpu = on` must not turn that into failing queries.  NULL,
 * one WARNING per backend, and no new attempt for PGV_CTX_RETRY_MS.
 */
#define PGV_CTX_RETRY_MS 10000
static pgv_ctx *
PgvTryGetContextInternal(bool capped)
{
	static TimestampTz failedAt = 0;
	static bool warned = false;

	if (backend_ctx != NULL)
		return backend_ctx;
	if (failedAt != 0 && !TimestampDifferenceExceeds(failedAt, GetCurrentTimestamp(), PGV_CTX_RETRY_MS))
		return NULL;
	/*
	 * The device runs the queues of FOUR processes side by side; from the fifth on they are time-sliced and ALL of
	 * them slow down (MI355X, 1 M x 1536, one query at a time per process: 4 processes 52 k QPS, 5: 31 k, 8: 31 k,
	 * 16: 27 k -- profiles/r05/own_context_process_sweep.md; threads of one process share its four queues and do not
	 * show this).  Every backend that holds a context is COUNTED, but only a caller that has a pooled alternative is
	 * CAPPED: an ivfflat scan beyond vector.gpu_max_own_contexts gets NULL here, quietly, and goes through
	 * the GPU worker's pooler (PgvOwnContextsExhausted), which needs no context.  Builds (k-means, assignment, hnsw
	 * linking) and hnsw scans have no pooled form -- denied a context they would silently fall back to the CPU, 10-100 x
	 * slower -- so they are never denied; a build gives its context back when it ends (PgvReleaseIdleContext).
	 */
	if (workerSlot < 0 && PgvShared != NULL && !backend_ctx_counted)
	{
		if (pg_atomic_fetch_add_u64(&PgvShared->ownContexts, 1) >= (uint64) vector_gpu_max_own_contexts && capped)
		{
			pg_atomic_fetch_add_u64(&PgvShared->ownContexts, -1);
			ereport(DEBUG1, (errmsg("pgvector GPU path: %d backends hold a device context (vector.gpu_max_own_contexts): this backend's ivfflat scans pool",
									vector_gpu_max_own_contexts)));
			return NULL;
		}
		backend_ctx_counted = true;
	}
	if (pgv_ctx_create(PgvMyDevice(), NULL, &backend_ctx) == PGV_OK)
	{
		failedAt = 0;
		return backend_ctx;
	}
	backend_ctx = NULL;
	if (backend_ctx_counted)
		pg_atomic_fetch_add_u64(&PgvShared->ownContexts, -1);
	backend_ctx_counted = false;
	failedAt = GetCurrentTimestamp();
	if (!warned)
		ereport(WARNING, (errmsg("pgvector GPU path unavailable (%s): using the CPU path", pgv_last_error())));
	warned = true;
	return NULL;
}

/* builds and hnsw scans: no pooled alternative, never denied by vector.gpu_max_own_contexts */
pgv_ctx *
PgvTryGetContext(void)
{
	return PgvTryGetContextInternal(false);
}

/* ivfflat scans: beyond the cap they pool instead */
pgv_ctx *
PgvTryGetScanContext(void)
{
	return PgvTryGetContextInternal(true);
}

/*
 * A build is over: its backend may never touch the device again, and a context held until backend exit pins one of the
 * device's four fast queues (four idle pooled connections that once built an index would push every other session's
 * scans to the pooler).  Given back unless this backend's scans use it (an imported mirror) -- they would only make it
 * again.
 */
void
PgvReleaseIdleContext(void)
{
	if (backend_ctx == NULL || workerSlot >= 0)
		return;
	for (PgvIvfMirror * m = mirrors; m != NULL; m = m->next)
		if (m->index)
			return;
	if (PgvHnswHoldsImports())
		return;
	pgv_ctx_destroy(backend_ctx);
	backend_ctx = NULL;
	if (backend_ctx_counted && PgvShared != NULL)
		pg_atomic_fetch_add_u64(&PgvShared->ownContexts, -1);
	backend_ctx_counted = false;
}

/* this backend has no context and would not be given one: its scans take the pooled path whatever vector.gpu_pooled says */
bool
PgvOwnContextsExhausted(void)
{
	return backend_ctx == NULL && workerSlot < 0 && PgvShared != NULL &&
		pg_atomic_read_u64(&PgvShared->ownContexts) >= (uint64) vector_gpu_max_own_contexts;
}

/*
 * Which kernel family serves this opclass, or false when none does.  The element type is identified EXACTLY:
 * vector reports IVFFLAT_MAX_DIM, halfvec twice that (src/ivfutils.c:382-404); the bit opclass reports 32 x and
 * (for hnsw) sparsevec SPARSEVEC_MAX_DIM -- their index tuples are not dense float rows and stay on the CPU path.
 */
bool
PgvIvfflatOpclass(Relation index, pgv_metric * metric, pgv_dtype * dtype, pgv_ops * ops)
{
	int			maxDimensions = IvfflatGetTypeInfo(index)->maxDimensions;
	bool		spherical = IvfflatOptionalProcInfo(index, IVFFLAT_KMEANS_NORM_PROC) != NULL;
	bool		normalized = IvfflatOptionalProcInfo(index, IVFFLAT_NORM_PROC) != NULL;

	if (maxDimensions == IVFFLAT_MAX_DIM)
		*dtype = PGV_F32;
	else if (maxDimensions == IVFFLAT_MAX_DIM * 2)
		*dtype = PGV_F16;
	else
		return false;			/* bit_hamming_ops, or a type added later */
	*metric = spherical ? PGV_NEG_IP : PGV_L2SQ;
	*ops = normalized ? PGV_OPS_COSINE : (spherical ? PGV_OPS_IP : PGV_OPS_L2);
	return true;
}

/* ------------------------------------------------------------------ the worker's side */

typedef struct PgvOwned
{
	Oid			relid;
	pgv_index  *index;			/* an ivfflat index's mirror, or */
	pgv_hnsw   *hnsw;			/* an hnsw index's */
}			PgvOwned;

static PgvOwned owned[PGV_MAX_MIRRORS];


/*
 * The worker's heartbeat.  Backends take a worker whose process is gone for dead at once, and one whose beat is older
 * than PGV_WORKER_HUNG_MS for hung; everything in the worker that can take long -- staging a large index through the
 * buffer manager -- beats on its way (once per page walked; a no-op in a backend).
 */
void
PgvWorkerBeat(void)
{
	if (workerSlot >= 0 && PgvShared != NULL)
		pg_atomic_write_u64(&PgvShared->workerBeat[workerSlot], (uint64) GetCurrentTimestamp());
}

/*
 * Stage the index out of its pages: the walks of GetScanLists (src/ivfscan.c:58-111) and GetScanItems
 * (:139-179), once per mirror instead of once per query.  Centers, list-major vectors, list offsets and
 * heap TIDs go to the device with pgv_index_upload (TIDs stay on the device: scans get them back with their
 * results, so no backend keeps a TID table).
 */
static pgv_index *
PgvStage(Relation index, pgv_metric metric, pgv_dtype dtype, int lists, int dimensions, int64 *ntuples)
{
	Size		esize = dtype == PGV_F32 ? sizeof(float) : sizeof(uint16);
	Size		rowBytes = esize * (Size) dimensions;
	char	   *centers = palloc(rowBytes * (Size) lists);
	BlockNumber *startPages = palloc(sizeof(BlockNumber) * (Size) lists);
	int64	   *offsets = palloc(sizeof(int64) * ((Size) lists + 1));
	int64		cap = 1024,
				n = 0;
	char	   *vectors = palloc_extended(rowBytes * (Size) cap, MCXT_ALLOC_HUGE);
	uint64	   *tids = palloc_extended(sizeof(uint64) * (Size) cap, MCXT_ALLOC_HUGE);
	BlockNumber nextblkno = IVFFLAT_HEAD_BLKNO;
	pgv_index  *result = NULL;
	int			l = 0;

	/* list pages */
	while (BlockNumberIsValid(nextblkno))
	{
		Buffer		buf = ReadBufferExtended(index, MAIN_FORKNUM, nextblkno, RBM_NORMAL, NULL);
		Page		page;
		OffsetNumber maxoffno;

		PgvWorkerBeat();
		LockBuffer(buf, BUFFER_LOCK_SHARE);
		page = BufferGetPage(buf);
		maxoffno = PageGetMaxOffsetNumber(page);
		for (OffsetNumber offno = FirstOffsetNumber; offno <= maxoffno && l < lists; offno = OffsetNumberNext(offno))
		{
			IvfflatList list = (IvfflatList) PageGetItem(page, PageGetItemId(page, offno));

			startPages[l] = list->startPage;
			memcpy(centers + rowBytes * (Size) l, list->center.x, rowBytes);
			l++;
		}
		nextblkno = IvfflatPageGetOpaque(page)->nextblkno;
		UnlockReleaseBuffer(buf);
	}
	if (l != lists)
		elog(ERROR, "ivfflat index is not valid");

	/* entry pages of every list, in page-chain order: the order the reference feeds its tuplesort */
	for (l = 0; l < lists; l++)
	{
		offsets[l] = n;
		nextblkno = startPages[l];
		while (BlockNumberIsValid(nextblkno))
		{
			Buffer		buf = ReadBufferExtended(index, MAIN_FORKNUM, nextblkno, RBM_NORMAL, NULL);
			Page		page;
			OffsetNumber maxoffno;

			CHECK_FOR_INTERRUPTS();
			PgvWorkerBeat();
			LockBuffer(buf, BUFFER_LOCK_SHARE);
			page = BufferGetPage(buf);
			maxoffno = PageGetMaxOffsetNumber(page);
			if (n + maxoffno > cap)
			{
				cap = (n + maxoffno) * 2;
				vectors = repalloc_huge(vectors, rowBytes * (Size) cap);
				tids = repalloc_huge(tids, sizeof(uint64) * (Size) cap);
			}
			for (OffsetNumber offno = FirstOffsetNumber; offno <= maxoffno; offno = OffsetNumberNext(offno))
			{
				IndexTuple	itup = (IndexTuple) PageGetItem(page, PageGetItemId(page, offno));
				bool		isnull;
				Datum		datum = index_getattr(itup, 1, RelationGetDescr(index), &isnull);
				Vector	   *vec = (Vector *) PG_DETOAST_DATUM(datum);	/* expands a short varlena header */

				/* Vector and HalfVector share the header; the payload starts at ->x */
				memcpy(vectors + rowBytes * (Size) n, vec->x, rowBytes);
				/* ItemPointerData widened to 64 bits: (block << 16) | offset, what the library hands back */
				tids[n] = ((uint64) (((uint32) itup->t_tid.ip_blkid.bi_hi << 16) | itup->t_tid.ip_blkid.bi_lo) << 16) | itup->t_tid.ip_posid;
				if ((Pointer) vec != DatumGetPointer(datum))
					pfree(vec);
				n++;
			}
			nextblkno = IvfflatPageGetOpaque(page)->nextblkno;
			UnlockReleaseBuffer(buf);
		}
	}
	offsets[lists] = n;
	PgvWorkerBeat();			/* (the upload itself: ~20 ms per GB) */
	if (pgv_index_upload(PgvGetContext(), metric, dtype, dimensions, lists, centers, offsets, vectors, tids, &result) != PGV_OK)
		ereport(ERROR, (errmsg("pgvector GPU path: %s", pgv_last_error())));
	*ntuples = n;
	pfree(tids);
	pfree(vectors);
	pfree(offsets);
	pfree(startPages);
	pfree(centers);
	return result;
}

/* an index that is gone gives its registry entry back (PgvShared->lock held exclusively) */
static void
PgvForgetEntry(PgvSharedMirror * e)
{
	e->relid = InvalidOid;
	e->dboid = InvalidOid;
	e->state = PGV_MIRROR_EMPTY;
	e->stagedGeneration = 0;
	e->stagedAt = 0;
	pg_atomic_write_u64(&e->generation, 0);
}

/*
 * A relcache invalidation arrived: DROP INDEX (or DROP TABLE, DROP DATABASE ...) may have taken an index whose mirror
 * this worker owns.  Nobody will ever ask for that index again, so nothing else would free its HBM or its registry
 * entry (64 of them for the cluster).
 */
static void
PgvWorkerDropGone(void)
{
	StartTransactionCommand();
	PG_TRY();
	{
		for (int i = 0; i < PGV_MAX_MIRRORS; i++)
		{
			Relation	index;

			if (owned[i].relid == InvalidOid)
				continue;
			index = try_index_open(owned[i].relid, AccessShareLock);
			if (index != NULL)
			{
				index_close(index, AccessShareLock);
				continue;
			}
			LWLockAcquire(PgvShared->lock, LW_EXCLUSIVE);
			{
				PgvSharedMirror *e = PgvFindEntry(owned[i].relid, false);

				if (e)
					PgvForgetEntry(e);
			}
			LWLockRelease(PgvShared->lock);
			/* backends that still hold an import keep their mapping until their scans end (the driver keeps the
			 * memory until the last mapping closes) */
			if (owned[i].index)
				pgv_index_free(owned[i].index);
			if (owned[i].hnsw)
				pgv_hnsw_free(owned[i].hnsw);
			memset(&owned[i], 0, sizeof(owned[i]));
		}
		CommitTransactionCommand();
	}
	PG_CATCH();
	{
		EmitErrorReport();
		FlushErrorState();
		AbortCurrentTransaction();
	}
	PG_END_TRY();
}

/* stage one requested entry inside a transaction of the worker; publishes READY, FAILED or (index gone) nothing */
static void
PgvWorkerStageEntry(PgvSharedMirror * e)
{
	Oid			relid = e->relid;
	uint64		generation = pg_atomic_read_u64(&e->generation);
	int			kind = e->kind;
	volatile bool ok = false;
	volatile bool gone = false;	/* the relation does not exist (any more): its entry is given back */
	pgv_index  *volatile fresh = NULL;
	pgv_hnsw   *volatile freshHnsw = NULL;
	pgv_index_handle handle;
	int			lists = 0,
				dimensions = 0;
	int64		ntuples = 0;
	pgv_metric	metric = PGV_L2SQ;
	pgv_dtype	dtype = PGV_F32;
	pgv_ops		ops;

	StartTransactionCommand();
	PG_TRY();
	{
		Relation	index = try_index_open(relid, AccessShareLock);

		if (index != NULL)
		{
			if (kind == PGV_KIND_HNSW)
			{
				/* the graph out of its pages (hnswscan_gpu.c), heap TIDs as the elements' payload: lists = m */
				freshHnsw = PgvHnswStage(index, &lists, &dimensions, &ntuples);
				if (freshHnsw != NULL)
				{
					if (pgv_hnsw_export(freshHnsw, &handle) != PGV_OK)
					{
						/* (as for ivfflat below: once more, on a second allocation) */
						pgv_hnsw   *again;

						elog(LOG, "pgvector GPU path: %s -- staging index %u once more", pgv_last_error(), relid);
						again = PgvHnswStage(index, &lists, &dimensions, &ntuples);
						pgv_hnsw_free(freshHnsw);
						freshHnsw = again;
						if (freshHnsw == NULL || pgv_hnsw_export(freshHnsw, &handle) != PGV_OK)
							ereport(ERROR, (errmsg("pgvector GPU path: %s", pgv_last_error())));
					}
					ok = true;
				}
			}
			else if (PgvIvfflatOpclass(index, &metric, &dtype, &ops))
			{
				IvfflatGetMetaPageInfo(index, &lists, &dimensions);
				fresh = PgvStage(index, metric, dtype, lists, dimensions, &ntuples);
				if (pgv_index_export(fresh, &handle) != PGV_OK)
				{
					/* seen once in ~350 exports of small mirrors that were allocated, exported and freed in quick
					 * succession (hipIpcGetMemHandle: invalid argument): a second allocation, made while the first is
					 * still held, gets a handle */
					pgv_index  *again;

					elog(LOG, "pgvector GPU path: %s -- staging index %u once more", pgv_last_error(), relid);
					again = PgvStage(index, metric, dtype, lists, dimensions, &ntuples);
					pgv_index_free(fresh);
					fresh = again;
					if (pgv_index_export(fresh, &handle) != PGV_OK)
						ereport(ERROR, (errmsg("pgvector GPU path: %s", pgv_last_error())));
				}
				ok = true;
			}
			index_close(index, AccessShareLock);
		}
		else
			gone = true;
		CommitTransactionCommand();
	}
	PG_CATCH();
	{
		/* a failed staging must not take the worker down: log, leave the index on the CPU path */
		EmitErrorReport();
		FlushErrorState();
		AbortCurrentTransaction();
		if (fresh)
			pgv_index_free(fresh);
		if (freshHnsw)
			pgv_hnsw_free(freshHnsw);
		fresh = NULL;
		freshHnsw = NULL;
		ok = false;
	}
	PG_END_TRY();

	LWLockAcquire(PgvShared->lock, LW_EXCLUSIVE);
	if (ok)
	{
		e->handle = handle;
		e->lists = lists;
		e->dimensions = dimensions;
		e->dtype = dtype;
		e->metric = metric;
		e->ntuples = ntuples;
		e->stagedGeneration = generation;
		e->stagedAt = GetCurrentTimestamp();
		e->state = PGV_MIRROR_READY;
	}
	else if (gone)
		PgvForgetEntry(e);
	else
		e->state = PGV_MIRROR_FAILED;
	LWLockRelease(PgvShared->lock);

	/* the previous mirror of this index goes once the new handle is out: backends holding an import of it keep
	 * their mapping (the driver keeps the memory until the last mapping closes) and re-import at their next scan.
	 * The index's OWN slot first (a hole in front of it -- a failed staging -- must not take the new mirror and leave
	 * the stale one behind it), a free slot only when the index has none. */
	{
		int			at = -1;

		for (int i = 0; i < PGV_MAX_MIRRORS && at < 0; i++)
			if (owned[i].relid == relid)
				at = i;
		for (int i = 0; ok && i < PGV_MAX_MIRRORS && at < 0; i++)
			if (owned[i].relid == 0)
				at = i;
		if (at >= 0)
		{
			if (owned[at].relid == relid && owned[at].index)
				pgv_index_free(owned[at].index);
			if (owned[at].relid == relid && owned[at].hnsw)
				pgv_hnsw_free(owned[at].hnsw);
			owned[at].relid = ok ? relid : 0;
			owned[at].index = ok ? fresh : NULL;
			owned[at].hnsw = ok ? freshHnsw : NULL;
		}
		else if (ok)
		{
			/* more indexes than slots: this one cannot be kept -- and must not be advertised */
			if (fresh)
				pgv_index_free(fresh);
			if (freshHnsw)
				pgv_hnsw_free(freshHnsw);
			LWLockAcquire(PgvShared->lock, LW_EXCLUSIVE);
			e->state = PGV_MIRROR_FAILED;
			LWLockRelease(PgvShared->lock);
		}
	}
}

/* hand a TAKEN slot back to its backend -- or to the free list when that backend has gone (PGV_SLOT_ABANDONED) */
static void
PgvSlotFinish(PgvPoolSlot * slot, PgvSlotState outcome)
{
	uint32		taken = PGV_SLOT_TAKEN;

	if (pg_atomic_compare_exchange_u32(&slot->state, &taken, (uint32) outcome))
		SetLatch(slot->latch);
	else
		pg_atomic_write_u32(&slot->state, PGV_SLOT_FREE);
}

/*
 * One round of the pooler: every FILLED slot of this database that asks the same (index, probes) as the first one
 * found becomes one pgv_search_batch -- GetScanLists + GetScanItems + the head of the sorted stream
 * (src/ivfscan.c:47-187) for all of them at once.  Returns whether anything was answered.
 */
static bool
PgvWorkerServePool(Oid dboid)
{
	static char *queries = NULL;	/* [PGV_POOL_MAX_BATCH x row bytes], the worker's lifetime */
	static float *dist = NULL;
	static uint64 *tid = NULL;
	PgvPoolSlot *batch[PGV_POOL_MAX_BATCH];
	int			n = 0;
	Oid			relid = InvalidOid;
	int			probes = 0;
	pgv_index  *index = NULL;
	Size		rowBytes = 0;
	bool		current = false;
	uint64		staged = 0;

	for (int i = 0; i < PGV_POOL_SLOTS && n < PGV_POOL_MAX_BATCH; i++)
	{
		PgvPoolSlot *slot = &PgvShared->pool[i];
		uint32		filled = PGV_SLOT_FILLED;

		if (pg_atomic_read_u32(&slot->state) != PGV_SLOT_FILLED || slot->dboid != dboid)
			continue;
		if (n > 0 && (slot->relid != relid || slot->probes != probes))
			continue;			/* another index or another probes setting: the next round's batch */
		if (!pg_atomic_compare_exchange_u32(&slot->state, &filled, PGV_SLOT_TAKEN))
			continue;
		relid = slot->relid;
		probes = slot->probes;
		batch[n++] = slot;
	}
	if (n == 0)
		return false;

	/* the mirror this worker owns for the index, if it is the current one */
	LWLockAcquire(PgvShared->lock, LW_SHARED);
	{
		PgvSharedMirror *e = PgvFindEntry(relid, false);

		if (e && e->state == PGV_MIRROR_READY && e->stagedGeneration == pg_atomic_read_u64(&e->generation))
		{
			current = true;
			staged = e->stagedGeneration + 1;
			rowBytes = (e->dtype == PGV_F32 ? sizeof(float) : sizeof(uint16)) * (Size) e->dimensions;
		}
	}
	LWLockRelease(PgvShared->lock);
	for (int i = 0; current && i < PGV_MAX_MIRRORS; i++)
		if (owned[i].relid == relid)
		{
			index = owned[i].index;
			break;
		}
	if (index == NULL || probes < 1 || probes > pgv_index_lists(index))
	{
		for (int i = 0; i < n; i++)
			PgvSlotFinish(batch[i], PGV_SLOT_UNSERVED);
		return true;
	}
	if (queries == NULL)
	{
		queries = MemoryContextAlloc(TopMemoryContext, PGV_POOL_ROW_BYTES * (Size) PGV_POOL_MAX_BATCH);
		dist = MemoryContextAlloc(TopMemoryContext, sizeof(float) * PGV_POOL_HEAD * (Size) PGV_POOL_MAX_BATCH);
		tid = MemoryContextAlloc(TopMemoryContext, sizeof(uint64) * PGV_POOL_HEAD * (Size) PGV_POOL_MAX_BATCH);
	}
	for (int i = 0; i < n; i++)
		memcpy(queries + rowBytes * (Size) i, batch[i]->payload, rowBytes);
	if (pgv_search_batch(index, queries, n, probes, PGV_POOL_HEAD, dist, NULL, tid) != PGV_OK)
	{
		/* the backends fall back to the reference's path; the worker stays up */
		elog(LOG, "pgvector GPU path: %s", pgv_last_error());
		for (int i = 0; i < n; i++)
			PgvSlotFinish(batch[i], PGV_SLOT_UNSERVED);
		return true;
	}
	for (int i = 0; i < n; i++)
	{
		PgvPoolSlot *slot = batch[i];
		int			count = 0;

		/* the head comes padded with +inf / ~0 past the tuples there are */
		while (count < PGV_POOL_HEAD && tid[(Size) i * PGV_POOL_HEAD + count] != ~(uint64) 0)
			count++;
		memcpy(slot->dist, dist + (Size) i * PGV_POOL_HEAD, sizeof(float) * (Size) count);
		memcpy(slot->tid, tid + (Size) i * PGV_POOL_HEAD, sizeof(uint64) * (Size) count);
		slot->count = count;
		slot->total = count < PGV_POOL_HEAD ? count : -1;	/* -1: at least PGV_POOL_HEAD, the exact number unknown */
		slot->staged = staged;
		PgvSlotFinish(slot, PGV_SLOT_DONE);
	}
	return true;
}

/*
 * The worker leaves (ERROR in its loop -- there is no handler round it, so that is FATAL --, SIGTERM, DROP DATABASE):
 * nobody must think it is still there.  Its registration goes, so that PgvEnsureWorker starts another; the mirrors it
 * owned die with the process, so the registry forgets them; queries waiting in the pool are told to run elsewhere.
 */
static void
PgvWorkerExit(int code, Datum arg)
{
	Oid			dboid = DatumGetObjectId(arg);

	(void) code;
	if (PgvShared == NULL || workerSlot < 0)
		return;
	LWLockAcquire(PgvShared->lock, LW_EXCLUSIVE);
	PgvShared->workerLatch[workerSlot] = NULL;
	PgvShared->workerDb[workerSlot] = 0;
	PgvShared->workerPid[workerSlot] = 0;
	pg_atomic_write_u64(&PgvShared->workerBeat[workerSlot], 0);
	for (int i = 0; i < PGV_MAX_MIRRORS; i++)
		if (PgvShared->mirrors[i].dboid == dboid && PgvShared->mirrors[i].relid != 0)
			PgvShared->mirrors[i].state = PGV_MIRROR_EMPTY;
	LWLockRelease(PgvShared->lock);
	for (int i = 0; i < PGV_POOL_SLOTS; i++)
	{
		PgvPoolSlot *slot = &PgvShared->pool[i];
		uint32		state = pg_atomic_read_u32(&slot->state);

		if (slot->dboid != dboid)
			continue;
		if (state == PGV_SLOT_ABANDONED)
		{
			/* its backend is gone and this worker was to free it: nobody else will (the pool must not shrink by a slot
			 * per cancelled query over the workers' lifetimes) */
			pg_atomic_compare_exchange_u32(&slot->state, &state, PGV_SLOT_FREE);
			continue;
		}
		if (state != PGV_SLOT_FILLED && state != PGV_SLOT_TAKEN)
			continue;
		if (pg_atomic_compare_exchange_u32(&slot->state, &state, PGV_SLOT_UNSERVED))
			SetLatch(slot->latch);
	}
	workerSlot = -1;
}

/*
 * Is the registered worker of this database alive?  A worker that was killed (no exit hook ran) is known by its
 * process being gone -- at once, so that no backend tries to import the handles of an exporter that no longer exists
 * (hipIpcOpenMemHandle on those takes seconds to fail) --; one whose process is there but whose beat is older than
 * PGV_WORKER_HUNG_MS is hung, or the pid has been given to somebody else.  A worker busy in a long synchronous call
 * (the upload of a very large index) only stops beating for a while: alive.
 */
static bool
PgvWorkerAlive(int slot)
{
	uint64		beat = pg_atomic_read_u64(&PgvShared->workerBeat[slot]);
	int			pid = PgvShared->workerPid[slot];

	if (beat == 0 || pid <= 0)
		return false;
	if (kill(pid, 0) != 0 && errno == ESRCH)
		return false;
	return !TimestampDifferenceExceeds((TimestampTz) beat, GetCurrentTimestamp(), PGV_WORKER_HUNG_MS);
}

/* bgw_main of the per-database worker; bgw_main_arg = the database's oid */
void
PgvWorkerMain(Datum main_arg)
{
	Oid			dboid = DatumGetObjectId(main_arg);
	int			slot = -1;

	BackgroundWorkerUnblockSignals();
	BackgroundWorkerInitializeConnectionByOid(dboid, InvalidOid, 0);
	LWLockAcquire(PgvShared->lock, LW_EXCLUSIVE);
	/* this database's registration first (a free slot in front of it must not hide it: two workers would serve one
	 * database), a free slot only when it has none */
	for (int i = 0; i < PGV_MAX_MIRRORS && slot < 0; i++)
		if (PgvShared->workerDb[i] == dboid)
			slot = i;
	for (int i = 0; i < PGV_MAX_MIRRORS && slot < 0; i++)
		if (PgvShared->workerDb[i] == 0)
			slot = i;
	if (slot >= 0 && PgvShared->workerDb[slot] == dboid && PgvShared->workerLatch[slot] != NULL)
	{
		/* two backends asked for a worker at the same moment: the first one to get here serves, this one leaves */
		if (PgvWorkerAlive(slot))
			slot = -1;
	}
	if (slot >= 0)
	{
		PgvShared->workerDb[slot] = dboid;
		PgvShared->workerPid[slot] = MyProcPid;
		PgvShared->workerLatch[slot] = MyLatch;
		pg_atomic_write_u64(&PgvShared->workerBeat[slot], (uint64) GetCurrentTimestamp());
	}
	LWLockRelease(PgvShared->lock);
	if (slot < 0)
		proc_exit(0);
	workerSlot = slot;
	before_shmem_exit(PgvWorkerExit, main_arg);

	for (;;)
	{
		PgvSharedMirror *todo = NULL;

		CHECK_FOR_INTERRUPTS();
		PgvWorkerBeat();
		/* an idle worker runs no transactions: invalidations (DROP INDEX) are taken in here */
		AcceptInvalidationMessages();
		if (ownedCheck)
		{
			ownedCheck = false;
			PgvWorkerDropGone();
		}
		LWLockAcquire(PgvShared->lock, LW_EXCLUSIVE);
		for (int i = 0; i < PGV_MAX_MIRRORS && todo == NULL; i++)
		{
			PgvSharedMirror *e = &PgvShared->mirrors[i];

			if (e->dboid == dboid && e->state == PGV_MIRROR_REQUESTED &&
				(e->stagedAt == 0 ||
				 TimestampDifferenceExceeds(e->stagedAt, GetCurrentTimestamp(), vector_gpu_restage_delay_ms)))
			{
				e->state = PGV_MIRROR_STAGING;
				todo = e;
			}
		}
		LWLockRelease(PgvShared->lock);
		if (todo)
		{
			PgvWorkerStageEntry(todo);
			continue;
		}
		if (PgvWorkerServePool(dboid))
			continue;			/* the next batch has been filling up meanwhile */
		(void) WaitLatch(MyLatch, WL_LATCH_SET | WL_TIMEOUT | WL_EXIT_ON_PM_DEATH, 200L, PG_WAIT_EXTENSION);
		ResetLatch(MyLatch);
	}
}


/* start the worker of this database unless one is registered and alive; returns whether one is there to be woken */
static bool
PgvEnsureWorker(void)
{
	BackgroundWorker worker;
	BackgroundWorkerHandle *handle;
	bool		running = false;
	int			dead = -1;

	LWLockAcquire(PgvShared->lock, LW_SHARED);
	for (int i = 0; i < PGV_MAX_MIRRORS; i++)
		if (PgvShared->workerDb[i] == MyDatabaseId && PgvShared->workerLatch[i] != NULL)
		{
			if (PgvWorkerAlive(i))
			{
				SetLatch(PgvShared->workerLatch[i]);
				running = true;
			}
			else
				dead = i;
		}
	LWLockRelease(PgvShared->lock);
	if (running)
		return true;
	if (dead >= 0)
	{
		/* it went without saying goodbye (SIGKILL, a crash): what PgvWorkerExit would have done */
		LWLockAcquire(PgvShared->lock, LW_EXCLUSIVE);
		if (PgvShared->workerDb[dead] == MyDatabaseId && !PgvWorkerAlive(dead))
		{
			PgvShared->workerLatch[dead] = NULL;
			PgvShared->workerDb[dead] = 0;
			PgvShared->workerPid[dead] = 0;
			pg_atomic_write_u64(&PgvShared->workerBeat[dead], 0);
			for (int i = 0; i < PGV_MAX_MIRRORS; i++)
				if (PgvShared->mirrors[i].dboid == MyDatabaseId && PgvShared->mirrors[i].relid != 0 &&
					PgvShared->mirrors[i].state != PGV_MIRROR_REQUESTED)
					PgvShared->mirrors[i].state = PGV_MIRROR_EMPTY;
		}
		LWLockRelease(PgvShared->lock);
	}
	/* A worker needs a moment to come up and register.  Backends that find none meanwhile (every waiting scan polls,
	 * every new scan asks) must not each have the postmaster fork another: one request per PGV_SPAWN_GAP_MS for the
	 * cluster; the duplicates that still slip through find the registration taken and leave (PgvWorkerMain). */
	{
		uint64		last = pg_atomic_read_u64(&PgvShared->lastSpawn);
		uint64		now = (uint64) GetCurrentTimestamp();

		if (last != 0 && !TimestampDifferenceExceeds((TimestampTz) last, (TimestampTz) now, PGV_SPAWN_GAP_MS))
			return false;
		if (!pg_atomic_compare_exchange_u64(&PgvShared->lastSpawn, &last, now))
			return false;		/* somebody else is asking right now */
	}
	memset(&worker, 0, sizeof(worker));
	worker.bgw_flags = BGWORKER_SHMEM_ACCESS | BGWORKER_BACKEND_DATABASE_CONNECTION;
	worker.bgw_start_time = BgWorkerStart_RecoveryFinished;
	worker.bgw_restart_time = BGW_NEVER_RESTART;
	snprintf(worker.bgw_library_name, sizeof(worker.bgw_library_name), "vector");
	snprintf(worker.bgw_function_name, sizeof(worker.bgw_function_name), "PgvWorkerMain");
	snprintf(worker.bgw_name, sizeof(worker.bgw_name), "pgvector gpu");
	snprintf(worker.bgw_type, sizeof(worker.bgw_type), "pgvector gpu");
	worker.bgw_main_arg = ObjectIdGetDatum(MyDatabaseId);
	worker.bgw_notify_pid = 0;
	(void) RegisterDynamicBackgroundWorker(&worker, &handle);
	return false;
}

/* a worker of this database is registered and alive (PgvShared->lock held) */
static bool
PgvDbWorkerAlive(void)
{
	for (int i = 0; i < PGV_MAX_MIRRORS; i++)
		if (PgvShared->workerDb[i] == MyDatabaseId && PgvShared->workerLatch[i] != NULL && PgvWorkerAlive(i))
			return true;
	return false;
}

static bool
PgvWorkerIsThere(void)
{
	bool		there;

	LWLockAcquire(PgvShared->lock, LW_SHARED);
	there = PgvDbWorkerAlive();
	LWLockRelease(PgvShared->lock);
	return there;
}

/* ------------------------------------------------------------------ the backend's side */

/*
 * The current mirror of an index as this backend sees it: an import of the worker's export, cached until the
 * registry shows a newer staging.  NULL = not available now (not staged yet, stale, unsupported opclass, no shared
 * memory): the scan stays on the reference's CPU path and a (re)staging has been requested.
 */
/* the registry's current staging of the index, asking the worker for one when there is none; false = none now */
static bool
PgvMirrorReadyKind(Relation index, int kind, pgv_index_handle * handle, uint64 *staged, int *lists, int64 *ntuples)
{
	PgvSharedMirror *e;
	bool		ready = false;
	TimestampTz waitUntil;

	/* The usual case -- a current mirror, a living worker -- changes nothing in the registry: a SHARED lock, so that
	 * the scans of all backends (every pooled query comes through here) do not queue up behind one another. */
	LWLockAcquire(PgvShared->lock, LW_SHARED);
	e = PgvFindEntryKind(RelationGetRelid(index), false, kind);
	if (e != NULL && e->state == PGV_MIRROR_READY && e->stagedGeneration == pg_atomic_read_u64(&e->generation) &&
		PgvDbWorkerAlive())
	{
		*handle = e->handle;
		*staged = e->stagedGeneration + 1;	/* 0 = none */
		if (lists)
			*lists = e->lists;
		if (ntuples)
			*ntuples = e->ntuples;
		ready = true;
	}
	LWLockRelease(PgvShared->lock);
	if (ready)
		return true;

	waitUntil = TimestampTzPlusMilliseconds(GetCurrentTimestamp(), vector_gpu_stage_wait_ms);
	for (;;)
	{
		bool		request = false;

		LWLockAcquire(PgvShared->lock, LW_EXCLUSIVE);
		e = PgvFindEntryKind(RelationGetRelid(index), true, kind);
		if (e != NULL)
		{
			uint64		generation = pg_atomic_read_u64(&e->generation);

			if (e->state == PGV_MIRROR_READY && e->stagedGeneration == generation && !PgvDbWorkerAlive())
			{
				/* the handle's exporter is gone (killed: no exit hook cleared the registry): importing it would take
				 * seconds to fail.  PgvEnsureWorker does what the exit hook would have done and starts another. */
				request = true;
			}
			else if (e->state == PGV_MIRROR_READY && e->stagedGeneration == generation)
			{
				*handle = e->handle;
				*staged = e->stagedGeneration + 1;	/* 0 = none */
				if (lists)
					*lists = e->lists;
				if (ntuples)
					*ntuples = e->ntuples;
				ready = true;
			}
			else if (e->state == PGV_MIRROR_EMPTY || e->state == PGV_MIRROR_READY ||
					 (e->state == PGV_MIRROR_FAILED && e->stagedGeneration != generation))
			{
				e->state = PGV_MIRROR_REQUESTED;
				request = true;
			}
			else if (e->state == PGV_MIRROR_REQUESTED)
				request = true;
		}
		LWLockRelease(PgvShared->lock);
		if (ready || e == NULL)
			break;
		if (request)
			PgvEnsureWorker();
		if (GetCurrentTimestamp() >= waitUntil)
			break;
		CHECK_FOR_INTERRUPTS();
		pg_usleep(1000L);
	}
	return ready;
}

static bool
PgvMirrorReady(Relation index, pgv_index_handle * handle, uint64 *staged)
{
	return PgvMirrorReadyKind(index, PGV_KIND_IVFFLAT, handle, staged, NULL, NULL);
}

/* hnswscan_gpu.c: the worker's export of an hnsw index's mirror (m, element count with it); false = none now */
bool
PgvHnswMirrorHandle(Relation index, pgv_index_handle * handle, uint64 *staged, int *m, int64 *nelements)
{
	if (PgvShared == NULL)
		return false;
	return PgvMirrorReadyKind(index, PGV_KIND_HNSW, handle, staged, m, nelements);
}

/* the pooled path's question: does the worker hold a current mirror of this index (asks for one otherwise) */
bool
PgvIvfflatMirrorIsCurrent(Relation index)
{
	pgv_index_handle handle;
	uint64		staged;
	pgv_metric	metric;
	pgv_dtype	dtype;
	pgv_ops		ops;

	if (PgvShared == NULL || !PgvIvfflatOpclass(index, &metric, &dtype, &ops))
		return false;
	return PgvMirrorReady(index, &handle, &staged);
}

/*
 * A published handle could not be imported: its exporter is gone (a worker that died takes its allocations with it).
 * Not the query's error: the registry forgets the staging, the next request has the index staged again, this scan runs
 * on the reference's path.
 */
void
PgvMirrorImportFailed(Relation index, uint64 staged)
{
	PgvSharedMirror *e;

	elog(LOG, "pgvector GPU path: mirror of index %u cannot be imported (%s): restaging", RelationGetRelid(index),
		 pgv_last_error());
	LWLockAcquire(PgvShared->lock, LW_EXCLUSIVE);
	e = PgvFindEntry(RelationGetRelid(index), false);
	if (e && e->state == PGV_MIRROR_READY && e->stagedGeneration + 1 == staged)
		e->state = PGV_MIRROR_EMPTY;
	LWLockRelease(PgvShared->lock);
}

/* unmap an import nobody uses any more and take it off the list */
static void
PgvDropMirror(PgvIvfMirror * m)
{
	PgvIvfMirror **link = &mirrors;

	if (m->index)
		pgv_index_free(m->index);
	m->index = NULL;
	while (*link != NULL && *link != m)
		link = &(*link)->next;
	if (*link == m)
		*link = m->next;
	pfree(m);
}

void
PgvIvfflatReleaseMirror(PgvIvfMirror * mirror)
{
	if (mirror == NULL)
		return;
	mirror->users--;
	/* the last scan on a staging that has been replaced meanwhile: its mapping goes now */
	if (mirror->users <= 0 && mirror->retired)
		PgvDropMirror(mirror);
}

PgvIvfMirror *
PgvIvfflatGetMirror(Relation index, uint64 wantStaged)
{
	PgvIvfMirror *m;
	pgv_index_handle handle;
	uint64		staged = 0;
	bool		ready;
	pgv_metric	metric;
	pgv_dtype	dtype;
	pgv_ops		ops;

	if (PgvShared == NULL || !PgvIvfflatOpclass(index, &metric, &dtype, &ops))
		return NULL;
	/* a staging this backend still holds (an open scan's) serves a scan that must stay on it */
	if (wantStaged != 0)
		for (m = mirrors; m != NULL; m = m->next)
			if (m->relid == RelationGetRelid(index) && m->index != NULL && m->staged == wantStaged)
			{
				m->users++;
				return m;
			}
	ready = PgvMirrorReady(index, &handle, &staged);
	if (!ready || (wantStaged != 0 && staged != wantStaged))
		return NULL;

	for (m = mirrors; m != NULL; m = m->next)
		if (m->relid == RelationGetRelid(index) && !m->retired)
			break;
	if (m != NULL && (!m->valid || m->staged != staged))
	{
		/* a newer staging.  Open scans of this backend (a cursor, a PL/pgSQL loop's outer query) keep the import they
		 * started on -- their pgv_query points into it --; it is unmapped when the last of them ends */
		if (m->users > 0)
		{
			m->retired = true;
			m = NULL;
		}
		else
		{
			if (m->index)
				pgv_index_free(m->index);
			m->index = NULL;
			m->valid = false;
		}
	}
	if (m == NULL)
	{
		m = MemoryContextAllocZero(TopMemoryContext, sizeof(PgvIvfMirror));
		m->relid = RelationGetRelid(index);
		m->next = mirrors;
		mirrors = m;
	}
	if (!m->valid)
	{
		pgv_ctx    *ctx = PgvTryGetScanContext();

		if (ctx == NULL)
			return NULL;		/* no device in this backend (a WARNING has said so), or beyond the cap: CPU path / pooler */
		if (pgv_index_import(ctx, &handle, &m->index) != PGV_OK)
		{
			/* the exporter is gone (a worker that died takes its allocations with it): not this query's error.  The
			 * registry forgets the staging, the next request has it staged again; this scan runs on the CPU path. */
			m->index = NULL;
			PgvMirrorImportFailed(index, staged);
			return NULL;
		}
		IvfflatGetMetaPageInfo(index, &m->lists, &m->dimensions);
		m->metric = metric;
		m->dtype = dtype;
		m->ntuples = pgv_index_rows(m->index);
		m->staged = staged;
		m->valid = true;
	}
	m->users++;
	return m;
}

/*
 * The backend's side of the pooler: one query in, the head of its sorted stream out (PGV_POOL_HEAD heap TIDs at
 * most).  false: not served (no shared memory, no worker, mirror not current, unsupported opclass, a NULL query) --
 * the caller runs this scan on its own context or on the reference's path.
 */
bool
PgvPoolSearch(Relation index, const void *payload, int probes, float *outDist, uint64 *outTid, int *outCount, bool *outComplete,
			  uint64 *outStaged)
{
	PgvPoolSlot *slot = NULL;
	pgv_metric	metric;
	pgv_dtype	dtype;
	pgv_ops		ops;
	int			lists,
				dimensions;
	Size		rowBytes;
	uint32		state;
	volatile bool abandoned = false;
	TimestampTz filledAt;

	if (PgvShared == NULL || payload == NULL || !PgvIvfflatOpclass(index, &metric, &dtype, &ops))
		return false;
	/* asks the worker for a (re)staging when there is no current mirror; NULL then */
	if (PgvIvfflatMirrorIsCurrent(index) == false)
		return false;
	IvfflatGetMetaPageInfo(index, &lists, &dimensions);
	rowBytes = (dtype == PGV_F32 ? sizeof(float) : sizeof(uint16)) * (Size) dimensions;
	if (rowBytes > PGV_POOL_ROW_BYTES)
		return false;
	for (int i = 0; i < PGV_POOL_SLOTS && slot == NULL; i++)
	{
		uint32		free_state = PGV_SLOT_FREE;

		/* start at a slot of this backend's own: no two backends fight over slot 0 */
		PgvPoolSlot *s = &PgvShared->pool[(i + MyProcPid) % PGV_POOL_SLOTS];

		if (pg_atomic_compare_exchange_u32(&s->state, &free_state, PGV_SLOT_CLAIMED))
			slot = s;
	}
	if (slot == NULL)
		return false;			/* more waiting backends than slots: this one scans for itself */
	slot->dboid = MyDatabaseId;
	slot->relid = RelationGetRelid(index);
	slot->probes = probes;
	slot->nullQuery = 0;
	slot->latch = MyLatch;
	memcpy(slot->payload, payload, rowBytes);
	myPoolSlot = slot;
	pg_atomic_write_u32(&slot->state, PGV_SLOT_FILLED);
	PgvEnsureWorker();			/* sets the worker's latch */
	filledAt = GetCurrentTimestamp();
	PG_TRY();
	{
		for (;;)
		{
			state = pg_atomic_read_u32(&slot->state);
			if (state == PGV_SLOT_DONE || state == PGV_SLOT_UNSERVED)
				break;
			(void) WaitLatch(MyLatch, WL_LATCH_SET | WL_TIMEOUT | WL_EXIT_ON_PM_DEATH, 100L, PG_WAIT_EXTENSION);
			ResetLatch(MyLatch);
			CHECK_FOR_INTERRUPTS();
			/* the worker answers a batch in about a millisecond; one that has not even TAKEN the query after
			 * PGV_POOL_PATIENCE_MS is staging some index (seconds): take the query back and scan for ourselves */
			if (TimestampDifferenceExceeds(filledAt, GetCurrentTimestamp(), PGV_POOL_PATIENCE_MS))
			{
				uint32		filled = PGV_SLOT_FILLED;

				if (pg_atomic_compare_exchange_u32(&slot->state, &filled, PGV_SLOT_UNSERVED))
				{
					state = PGV_SLOT_UNSERVED;
					break;
				}
			}
			/* the worker may have died without a word (its exit hook tells us when it can): a query must not wait
			 * for an answer nobody will give.  Take the slot back -- or leave it to a worker that holds it after all --
			 * and scan for ourselves. */
			if (!PgvWorkerIsThere())
			{
				uint32		filled = PGV_SLOT_FILLED;

				state = pg_atomic_read_u32(&slot->state);
				if (state == PGV_SLOT_DONE || state == PGV_SLOT_UNSERVED)
					break;
				if (!pg_atomic_compare_exchange_u32(&slot->state, &filled, PGV_SLOT_UNSERVED))
				{
					uint32		taken = PGV_SLOT_TAKEN;

					if (pg_atomic_compare_exchange_u32(&slot->state, &taken, PGV_SLOT_ABANDONED))
					{
						/* (no return out of a PG_TRY block.)  A TAKEN slot is the dead worker's to free: nobody will
						 * -- one slot of 256 */
						abandoned = true;
						break;
					}
				}
				state = pg_atomic_read_u32(&slot->state);
				if (state == PGV_SLOT_DONE || state == PGV_SLOT_UNSERVED)
					break;
			}
		}
	}
	PG_CATCH();
	{
		/* cancelled while waiting: a slot the worker has not taken yet is free again; one it is answering is marked so
		 * that the worker frees it instead of publishing results nobody reads */
		uint32		filled = PGV_SLOT_FILLED;

		if (!pg_atomic_compare_exchange_u32(&slot->state, &filled, PGV_SLOT_FREE))
		{
			uint32		taken = PGV_SLOT_TAKEN;

			if (!pg_atomic_compare_exchange_u32(&slot->state, &taken, PGV_SLOT_ABANDONED))
				pg_atomic_write_u32(&slot->state, PGV_SLOT_FREE);	/* DONE / UNSERVED already */
		}
		myPoolSlot = NULL;
		PG_RE_THROW();
	}
	PG_END_TRY();
	if (abandoned)
	{
		myPoolSlot = NULL;
		return false;
	}
	if (state == PGV_SLOT_DONE)
	{
		*outCount = slot->count;
		*outComplete = slot->total >= 0;
		*outStaged = slot->staged;
		memcpy(outDist, slot->dist, sizeof(float) * (Size) slot->count);
		memcpy(outTid, slot->tid, sizeof(uint64) * (Size) slot->count);
	}
	myPoolSlot = NULL;
	pg_atomic_write_u32(&slot->state, PGV_SLOT_FREE);
	return state == PGV_SLOT_DONE;
}
