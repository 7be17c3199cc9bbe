/*
 * ivfbuild_gpu.c -- the two distance loops of CREATE INDEX on the device: IvfflatKmeans
 * (src/ivfkmeans.c:553-570 -> pgv_kmeans) and the argmin of AddTupleToSort (src/ivfbuild.c:161-219 ->
 * pgv_assign on batches of heap rows).  Twin over the emulated page image: pgvector_amd/host/ivf_build.c.
 */
#include "pgv_gpu.h"

#include <pthread.h>
#include <signal.h>
#include <stdlib.h>

#include "miscadmin.h"

#define PGV_ASSIGN_BATCH (1 << 18)	/* heap rows handed to the GPU at a time, at most */
#define PGV_ASSIGN_BATCH_BYTES ((Size) 512 << 20)	/* ... and at most this much of them (2000-d rows: 65 536) */

typedef struct PgvIvfBuild
{
	pgv_metric	metric;
	pgv_dtype	dtype;
	pgv_ops		ops;
	Size		rowBytes;
	char	   *centers;		/* [lists x dimensions] payloads, densely packed */
	int			count;			/* rows buffered */
	int			batch;			/* rows per pgv_assign call: min(PGV_ASSIGN_BATCH, PGV_ASSIGN_BATCH_BYTES / row bytes) */
	char	   *rows;			/* [batch x dimensions] */
	ItemPointerData *tids;
	int32	   *lists;
	char	   *value;			/* one Vector / HalfVector varlena, rebuilt from a buffered payload for the tuplesort */
	Size		valueSize;
}			PgvIvfBuild;

static double
PgvRandomDouble(void *state)
{
	(void) state;
	return RandomDouble();		/* pg_prng on pg_global_prng_state, src/ivfflat.h:86-94 */
}

static uint32
PgvRandomInt(void *state)
{
	(void) state;
	return (uint32) RandomInt();
}

/*
 * k-means over the devices of the node -- the multi-GPU half of the parallel build (src/ivfbuild.c:830-966).
 *
 * In the reference the centers are computed by the LEADER alone, before any parallel worker exists (ComputeCenters runs
 * ahead of AssignTuples / IvfflatBeginParallel, src/ivfbuild.c:954-1006), so the exchange cannot ride on the workers'
 * DSM segment: the leader drives the other devices itself, one helper thread per device, each with a context of its
 * own.  Rank r gets rows [r n / R, (r + 1) n / R) of the sample; pgv_comm_unique_id + pgv_comm_create form the group
 * (RCCL over xGMI in libpgv_hip), pgv_kmeans_sharded runs k-means++ and Lloyd with ONE all-reduce of
 * sums | counts | changes per iteration on each context's stream.  The helper threads never touch PostgreSQL state (no
 * palloc, no ereport: they report through their record) and start with every signal blocked, like the hnsw build's.
 * All ranks draw from the same seeded stream (the library's own generator; the seed comes from the backend's
 * RandomInt()): the draws steer replicated decisions.
 * false: not applicable (one device, vector.gpu_build_devices = 1, too few samples) or the group did not form --
 * the caller runs pgv_kmeans on its own device.
 */
typedef struct PgvKmeansRank
{
	pthread_t	thread;
	int			rank,
				nranks,
				device;
	pgv_ctx    *ctx;			/* rank 0: the backend's; the others make (and destroy) their own */
	const unsigned char *id;
	pgv_ops		ops;
	pgv_dtype	dtype;
	int			dim,
				k;
	const char *rows;
	int			n;
	uint64		seed;
	char	   *out;			/* [k x dim] */
	int			iterations;
	pthread_barrier_t *ready;	/* every rank has its context (or not): nobody enters pgv_comm_create alone */
	volatile int *ctx_failed;
	int			rc;
	char		err[256];
}			PgvKmeansRank;

static void *
PgvKmeansRankMain(void *arg)
{
	PgvKmeansRank *r = (PgvKmeansRank *) arg;
	pgv_comm   *comm = NULL;
	pgv_rng		rng;
	bool		own = false;

	r->rc = PGV_OK;
	if (r->ctx == NULL)
	{
		own = true;
		if (pgv_ctx_create(r->device, NULL, &r->ctx) != PGV_OK)
		{
			r->rc = PGV_ERR_DEVICE;
			snprintf(r->err, sizeof(r->err), "device %d: %s", r->device, pgv_last_error());
			r->ctx = NULL;
			__atomic_store_n(r->ctx_failed, 1, __ATOMIC_SEQ_CST);
		}
	}
	pthread_barrier_wait(r->ready);
	if (__atomic_load_n(r->ctx_failed, __ATOMIC_SEQ_CST))
	{
		if (r->rc == PGV_OK)
			r->rc = PGV_ERR_STATE;	/* somebody else has no device: the group is not formed */
	}
	else
	{
		memset(&rng, 0, sizeof(rng));
		rng.seed = r->seed;
		if (pgv_comm_create(r->ctx, r->nranks, r->rank, r->id, &comm) != PGV_OK ||
			pgv_kmeans_sharded(comm, r->ops, r->dtype, r->dim, r->rows, r->n, r->k, 500, &rng, r->out, NULL, &r->iterations) != PGV_OK)
		{
			r->rc = PGV_ERR_DEVICE;
			snprintf(r->err, sizeof(r->err), "rank %d (device %d): %s", r->rank, r->device, pgv_last_error());
		}
		if (comm)
			pgv_comm_destroy(comm);
	}
	if (own && r->ctx)
		pgv_ctx_destroy(r->ctx);
	return NULL;
}

static bool
PgvKmeansOnDevices(pgv_ctx * ctx, pgv_ops ops, pgv_dtype dtype, int dim, const char *in, int n, int k, char *out, int *iterations)
{
	int			ndev = pgv_device_count();
	int			R = vector_gpu_build_devices > 0 ? Min(vector_gpu_build_devices, ndev) : ndev;
	Size		rowBytes = (dtype == PGV_F32 ? sizeof(float) : sizeof(uint16)) * (Size) dim;
	unsigned char id[PGV_COMM_ID_BYTES];
	PgvKmeansRank *ranks;
	pthread_barrier_t ready;
	volatile int ctxFailed = 0;
	sigset_t	all,
				old;
	int			mine = PgvMyDevice();
	int			started = 0;
	bool		ok = true;

	R = Min(R, 16);
	/* a rank wants at least a few tiles of rows: below that one device is faster than the exchanges */
	while (R > 1 && n / R < Max(4 * k / R, 1024))
		R--;
	if (R < 2 || pgv_comm_unique_id(id) != PGV_OK)
		return false;
	ranks = palloc0(sizeof(PgvKmeansRank) * (Size) R);
	pthread_barrier_init(&ready, NULL, (unsigned) R);
	for (int r = 0; r < R; r++)
	{
		int			lo = (int) ((int64) n * r / R),
					hi = (int) ((int64) n * (r + 1) / R);

		ranks[r].rank = r;
		ranks[r].nranks = R;
		/* rank 0 is this backend on its own device; the others take the node's remaining devices in order */
		ranks[r].device = r == 0 ? mine : (mine + r) % ndev;
		ranks[r].ctx = r == 0 ? ctx : NULL;
		ranks[r].id = id;
		ranks[r].ops = ops;
		ranks[r].dtype = dtype;
		ranks[r].dim = dim;
		ranks[r].k = k;
		ranks[r].rows = in + rowBytes * (Size) lo;
		ranks[r].n = hi - lo;
		ranks[r].seed = ((uint64) (uint32) RandomInt() << 32) | (uint32) r * 0;	/* (one draw per rank keeps the backend's stream moving; the seed is rank 0's) */
		ranks[r].out = r == 0 ? out : malloc(rowBytes * (Size) k);
		ranks[r].ready = &ready;
		ranks[r].ctx_failed = &ctxFailed;
	}
	for (int r = 1; r < R; r++)
		ranks[r].seed = ranks[0].seed;
	/* helper threads start with every signal blocked: SIGINT / SIGTERM / SIGUSR1 stay this thread's (the backend's) */
	sigfillset(&all);
	pthread_sigmask(SIG_BLOCK, &all, &old);
	for (int r = 1; r < R; r++)
	{
		if (ranks[r].out == NULL || pthread_create(&ranks[r].thread, NULL, PgvKmeansRankMain, &ranks[r]) != 0)
			break;
		started++;
	}
	pthread_sigmask(SIG_SETMASK, &old, NULL);
	if (started != R - 1)
	{
		/* (cannot happen short of resource exhaustion; the started ones must not wait for ever at the barrier) */
		__atomic_store_n(&ctxFailed, 1, __ATOMIC_SEQ_CST);
		for (int r = started + 1; r < R; r++)
			pthread_barrier_wait(&ready);	/* stand in for the ranks that never started */
	}
	PgvKmeansRankMain(&ranks[0]);	/* this backend is rank 0 */
	for (int r = 1; r <= started; r++)
		pthread_join(ranks[r].thread, NULL);
	pthread_barrier_destroy(&ready);
	for (int r = 0; r < R; r++)
		if (ranks[r].rc != PGV_OK)
		{
			if (ok && ranks[r].err[0])
				ereport(LOG, (errmsg("pgvector GPU path: k-means over %d devices not possible (%s): one device", R, ranks[r].err)));
			ok = false;
		}
	if (ok)
	{
		*iterations = ranks[0].iterations;
		ereport(DEBUG1, (errmsg("pgvector GPU path: k-means sharded over %d devices (%d samples, %d lists): %d iterations", R, n, k,
								ranks[0].iterations)));
	}
	for (int r = 1; r < R; r++)
		free(ranks[r].out);
	pfree(ranks);
	return ok;
}

/*
 * IvfflatKmeans.  samples were normalised by SampleCallback for opclasses with a KMEANS_NORM proc
 * (src/ivfbuild.c:148-156); the centers come back as payloads and are re-wrapped as Vector / HalfVector.
 */
bool
PgvIvfflatKmeans(Relation index, VectorArray samples, VectorArray centers, const IvfflatTypeInfo * typeInfo)
{
	pgv_metric	metric;
	pgv_dtype	dtype;
	pgv_ops		ops;
	pgv_rng		rng = {PgvRandomDouble, PgvRandomInt, NULL, 0};
	Size		rowBytes;
	char	   *in,
			   *out;
	int			iterations;

	pgv_ctx    *ctx;

	(void) typeInfo;
	if (!vector_gpu || !vector_gpu_kmeans || !PgvIvfflatOpclass(index, &metric, &dtype, &ops))
		return false;			/* (vector.gpu_kmeans = off: the reference's ElkanKmeans, its centers to the bit) */
	if ((ctx = PgvTryGetContext()) == NULL)
		return false;			/* no device: the reference's ElkanKmeans */
	rowBytes = (dtype == PGV_F32 ? sizeof(float) : sizeof(uint16)) * (Size) samples->dim;
	/* (50 samples per list: past palloc's 1 GB at 4096 lists of 1536-d rows -- the reference's own array is a
	 * MCXT_ALLOC_HUGE allocation too, src/ivfutils.c VectorArrayInit) */
	in = palloc_extended(rowBytes * (Size) Max(samples->length, 1), MCXT_ALLOC_HUGE);
	out = palloc(rowBytes * (Size) centers->maxlen);
	for (int i = 0; i < samples->length; i++)
		memcpy(in + rowBytes * (Size) i, ((Vector *) VectorArrayGet(samples, i))->x, rowBytes);
	/* several devices on the node: the samples are sharded by row over them and the Lloyd iterations exchange one fused
	 * all-reduce each (PgvKmeansOnDevices below); otherwise, or when the group cannot form, this backend's device alone.
	 * Same error texts as CheckCenters (src/ivfkmeans.c:507-533) either way */
	if (!PgvKmeansOnDevices(ctx, ops, dtype, samples->dim, in, samples->length, centers->maxlen, out, &iterations) &&
		pgv_kmeans(ctx, ops, dtype, samples->dim, in, samples->length, centers->maxlen, 500, &rng,
				   out, NULL, &iterations) != PGV_OK)
		elog(ERROR, "%s", pgv_last_error());
	for (int i = 0; i < centers->maxlen; i++)
	{
		Vector	   *c = (Vector *) VectorArrayGet(centers, i);

		memset(c, 0, centers->itemsize);
		c->vl_len_ = (int32) (centers->itemsize << 2);	/* SET_VARSIZE */
		c->dim = (int16) samples->dim;
		memcpy(c->x, out + rowBytes * (Size) i, rowBytes);
	}
	centers->length = centers->maxlen;
	pfree(in);
	pfree(out);
	return true;
}

void
PgvIvfflatBuildBegin(IvfflatBuildState * buildstate)
{
	PgvIvfBuild *gb;

	buildstate->gpu = NULL;
	if (!vector_gpu || PgvTryGetContext() == NULL)
		return;					/* (no device: the reference's AddTupleToSort) */
	gb = palloc0(sizeof(PgvIvfBuild));
	if (!PgvIvfflatOpclass(buildstate->index, &gb->metric, &gb->dtype, &gb->ops))
	{
		pfree(gb);
		return;					/* bit opclass: the reference's AddTupleToSort */
	}
	gb->rowBytes = (gb->dtype == PGV_F32 ? sizeof(float) : sizeof(uint16)) * (Size) buildstate->dimensions;
	gb->centers = palloc(gb->rowBytes * (Size) buildstate->lists);
	for (int i = 0; i < buildstate->lists; i++)
		memcpy(gb->centers + gb->rowBytes * (Size) i, ((Vector *) VectorArrayGet(buildstate->centers, i))->x, gb->rowBytes);
	gb->batch = (int) Min((Size) PGV_ASSIGN_BATCH, Max(PGV_ASSIGN_BATCH_BYTES / gb->rowBytes, (Size) 1));
	gb->rows = palloc_extended(gb->rowBytes * (Size) gb->batch, MCXT_ALLOC_HUGE);
	gb->tids = palloc(sizeof(ItemPointerData) * (Size) gb->batch);
	/* Vector and HalfVector share the 8-byte header (vl_len_, dim, unused); src/vector.h:18-24, src/halfvec.h:68-74 */
	gb->valueSize = offsetof(Vector, x) + gb->rowBytes;
	gb->value = palloc0(gb->valueSize);
	gb->lists = palloc(sizeof(int32) * (Size) gb->batch);
	buildstate->gpu = gb;
}

/* the argmin loop of AddTupleToSort for the buffered rows, then the reference's own tuplesort feed */
void
PgvIvfflatBuildFlush(IvfflatBuildState * buildstate)
{
	PgvIvfBuild *gb = (PgvIvfBuild *) buildstate->gpu;

	if (gb == NULL || gb->count == 0)
		return;
	CHECK_FOR_INTERRUPTS();
	if (pgv_assign(PgvGetContext(), gb->metric, gb->dtype, buildstate->dimensions, gb->centers, buildstate->lists,
				   gb->rows, gb->count, gb->lists, NULL) != PGV_OK)
		ereport(ERROR, (errmsg("pgvector GPU path: %s", pgv_last_error())));
	/* (beside the reference's own "leader / worker processed N tuples": every participant of a parallel build flushes its own) */
	ereport(DEBUG1, (errmsg("pgvector GPU path: %d rows assigned on the device", gb->count)));
	for (int i = 0; i < gb->count; i++)
	{
		/* The caller's Datum is gone by now: BuildCallback detoasts and normalises inside buildstate->tmpCtx and
		 * resets it after every row, and a non-toasted value points into the heap scan's buffer
		 * (src/ivfbuild.c:238-249).  The tuplesort copies what it is given, so the value is rebuilt from the
		 * payload this file kept. */
		Vector	   *v = (Vector *) gb->value;

		v->vl_len_ = (int32) (gb->valueSize << 2);	/* SET_VARSIZE */
		v->dim = (int16) buildstate->dimensions;
		v->unused = 0;
		memcpy(v->x, gb->rows + gb->rowBytes * (Size) i, gb->rowBytes);
		IvfflatAddToSort(buildstate, gb->lists[i], &gb->tids[i], PointerGetDatum(v));
	}
	gb->count = 0;
}

/* BuildCallback after its NULL / norm handling (src/ivfbuild.c:236-263): `value` is detoasted and normalised */
void
PgvIvfflatBuildAdd(IvfflatBuildState * buildstate, ItemPointer tid, Datum value)
{
	PgvIvfBuild *gb = (PgvIvfBuild *) buildstate->gpu;

	/* only the payload is kept: `value` lives in a context the caller resets after this row */
	memcpy(gb->rows + gb->rowBytes * (Size) gb->count, ((Vector *) DatumGetPointer(value))->x, gb->rowBytes);
	gb->tids[gb->count] = *tid;
	if (++gb->count == gb->batch)
		PgvIvfflatBuildFlush(buildstate);
}
