/*
 * ivfbuild_gpu.c -- the two distance loops of CREATE INDEX on the device: IvfflatKmeans
 * (src/ivfkmeans.c:553-570 -> pgv_kmeans) and the argmin of AddTupleToSort (src/ivfbuild.c:161-219 ->
 * pgv_assign on batches of heap rows).  Twin over the emulated page image: pgvector_amd/host/ivf_build.c.
 */
#include "pgv_gpu.h"

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
	/* same error texts as CheckCenters (src/ivfkmeans.c:507-533) */
	if (pgv_kmeans(ctx, ops, dtype, samples->dim, in, samples->length, centers->maxlen, 500, &rng,
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
