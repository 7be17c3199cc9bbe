/*
 * pgv_context.c -- GUCs, the per-backend GPU context and the per-index device mirror.
 * Mirrors pgvector_amd/host/ivf_mirror.c + the stager of ivf_pages.c, over the buffer manager.
 */
#include "pgv_gpu.h"

#include "miscadmin.h"
#include "storage/bufmgr.h"
#include "storage/ipc.h"
#include "utils/guc.h"
#include "utils/inval.h"
#include "utils/memutils.h"

bool		vector_gpu = false;
int			vector_gpu_device = 0;

static pgv_ctx *backend_ctx = NULL;
static PgvIvfMirror *mirrors = NULL;

static void
PgvAtExit(int code, Datum arg)
{
	(void) code;
	(void) arg;
	for (PgvIvfMirror * m = mirrors; m != NULL; m = m->next)
		if (m->index)
			pgv_index_free(m->index);
	if (backend_ctx)
		pgv_ctx_destroy(backend_ctx);
	backend_ctx = NULL;
}

/* insert / vacuum / REINDEX change the pages: the next scan restages (src/ivfinsert.c, src/ivfvacuum.c) */
static void
PgvRelcacheCallback(Datum arg, Oid relid)
{
	(void) arg;
	for (PgvIvfMirror * m = mirrors; m != NULL; m = m->next)
		if (m->relid == relid || relid == 0)
			m->valid = false;
	PgvHnswInvalidate(relid);
}

void
PgvGpuInit(void)
{
	DefineCustomBoolVariable("vector.gpu", "Runs the distance hot path on the GPU (libpgv_hip)", NULL,
							 &vector_gpu, false, PGC_USERSET, 0, NULL, NULL, NULL);
	DefineCustomIntVariable("vector.gpu_device", "HIP device of this backend", NULL,
							&vector_gpu_device, 0, 0, 63, PGC_USERSET, 0, NULL, NULL, NULL);
	CacheRegisterRelcacheCallback(PgvRelcacheCallback, (Datum) 0);
	on_proc_exit(PgvAtExit, (Datum) 0);
}

pgv_ctx *
PgvGetContext(void)
{
	if (backend_ctx == NULL && pgv_ctx_create(vector_gpu_device, NULL, &backend_ctx) != PGV_OK)
		ereport(ERROR, (errmsg("pgvector GPU path: %s", pgv_last_error())));
	return backend_ctx;
}

void
PgvIvfflatOpclass(Relation index, pgv_metric * metric, pgv_dtype * dtype, pgv_ops * ops)
{
	bool		spherical = IvfflatOptionalProcInfo(index, IVFFLAT_KMEANS_NORM_PROC) != NULL;
	bool		normalized = IvfflatOptionalProcInfo(index, IVFFLAT_NORM_PROC) != NULL;

	/* vector: 2000 dimensions, halfvec: 4000 (src/ivfflat.h:37, src/ivfutils.c:401) */
	*dtype = IvfflatGetTypeInfo(index)->maxDimensions > 2000 ? PGV_F16 : PGV_F32;
	*metric = spherical ? PGV_NEG_IP : PGV_L2SQ;
	*ops = normalized ? PGV_OPS_COSINE : (spherical ? PGV_OPS_IP : PGV_OPS_L2);
}

/*
 * Stage the index out of its pages: the walks of GetScanLists (src/ivfscan.c:58-111) and GetScanItems
 * (:139-179), once per mirror instead of once per query.  Centers, list-major vectors, list offsets and
 * heap TIDs go to the device with pgv_index_upload.
 */
static void
PgvStage(Relation index, PgvIvfMirror * m)
{
	Size		esize = m->dtype == PGV_F32 ? sizeof(float) : sizeof(uint16);
	Size		rowBytes = esize * (Size) m->dimensions;
	char	   *centers = palloc(rowBytes * (Size) m->lists);
	BlockNumber *startPages = palloc(sizeof(BlockNumber) * (Size) m->lists);
	int64	   *offsets = palloc(sizeof(int64) * ((Size) m->lists + 1));
	int64		cap = 1024,
				n = 0;
	char	   *vectors = palloc(rowBytes * (Size) cap);
	ItemPointerData *tids = MemoryContextAlloc(TopMemoryContext, sizeof(ItemPointerData) * (Size) cap);
	uint64	   *tids64;
	BlockNumber nextblkno = IVFFLAT_HEAD_BLKNO;
	int			l = 0;

	/* list pages */
	while (BlockNumberIsValid(nextblkno))
	{
		Buffer		buf = ReadBufferExtended(index, MAIN_FORKNUM, nextblkno, RBM_NORMAL, NULL);
		Page		page;
		OffsetNumber maxoffno;

		LockBuffer(buf, BUFFER_LOCK_SHARE);
		page = BufferGetPage(buf);
		maxoffno = PageGetMaxOffsetNumber(page);
		for (OffsetNumber offno = FirstOffsetNumber; offno <= maxoffno && l < m->lists; offno = OffsetNumberNext(offno))
		{
			IvfflatList list = (IvfflatList) PageGetItem(page, PageGetItemId(page, offno));

			startPages[l] = list->startPage;
			memcpy(centers + rowBytes * (Size) l, list->center.x, rowBytes);
			l++;
		}
		nextblkno = IvfflatPageGetOpaque(page)->nextblkno;
		UnlockReleaseBuffer(buf);
	}
	if (l != m->lists)
		elog(ERROR, "ivfflat index is not valid");

	/* entry pages of every list, in page-chain order: the order the reference feeds its tuplesort */
	for (l = 0; l < m->lists; l++)
	{
		offsets[l] = n;
		nextblkno = startPages[l];
		while (BlockNumberIsValid(nextblkno))
		{
			Buffer		buf = ReadBufferExtended(index, MAIN_FORKNUM, nextblkno, RBM_NORMAL, NULL);
			Page		page;
			OffsetNumber maxoffno;

			CHECK_FOR_INTERRUPTS();
			LockBuffer(buf, BUFFER_LOCK_SHARE);
			page = BufferGetPage(buf);
			maxoffno = PageGetMaxOffsetNumber(page);
			if (n + maxoffno > cap)
			{
				cap = (n + maxoffno) * 2;
				vectors = repalloc(vectors, rowBytes * (Size) cap);
				tids = repalloc(tids, sizeof(ItemPointerData) * (Size) cap);
			}
			for (OffsetNumber offno = FirstOffsetNumber; offno <= maxoffno; offno = OffsetNumberNext(offno))
			{
				IndexTuple	itup = (IndexTuple) PageGetItem(page, PageGetItemId(page, offno));
				bool		isnull;
				Datum		datum = index_getattr(itup, 1, RelationGetDescr(index), &isnull);
				Vector	   *vec = (Vector *) PG_DETOAST_DATUM(datum);	/* expands a short varlena header */

				/* Vector and HalfVector share the header; the payload starts at ->x */
				memcpy(vectors + rowBytes * (Size) n, vec->x, rowBytes);
				tids[n] = itup->t_tid;
				n++;
			}
			nextblkno = IvfflatPageGetOpaque(page)->nextblkno;
			UnlockReleaseBuffer(buf);
		}
	}
	offsets[m->lists] = n;

	/* ItemPointerData widened to 64 bits: (block << 16) | offset, what the library hands back */
	tids64 = palloc(sizeof(uint64) * (Size) Max(n, 1));
	for (int64 i = 0; i < n; i++)
		tids64[i] = ((uint64) (((uint32) tids[i].ip_blkid.bi_hi << 16) | tids[i].ip_blkid.bi_lo) << 16) | tids[i].ip_posid;
	if (pgv_index_upload(PgvGetContext(), m->metric, m->dtype, m->dimensions, m->lists, centers, offsets,
						 vectors, tids64, &m->index) != PGV_OK)
		ereport(ERROR, (errmsg("pgvector GPU path: %s", pgv_last_error())));
	m->tids = tids;
	m->ntuples = n;
	m->valid = true;
	pfree(tids64);
	pfree(vectors);
	pfree(offsets);
	pfree(startPages);
	pfree(centers);
}

PgvIvfMirror *
PgvIvfflatGetMirror(Relation index)
{
	PgvIvfMirror *m;
	pgv_ops		ops;

	for (m = mirrors; m != NULL; m = m->next)
		if (m->relid == RelationGetRelid(index))
			break;
	if (m == NULL)
	{
		m = MemoryContextAllocZero(TopMemoryContext, sizeof(PgvIvfMirror));
		m->relid = RelationGetRelid(index);
		m->next = mirrors;
		mirrors = m;
	}
	if (!m->valid)
	{
		if (m->index)
			pgv_index_free(m->index);
		if (m->tids)
			pfree(m->tids);
		m->index = NULL;
		m->tids = NULL;
		IvfflatGetMetaPageInfo(index, &m->lists, &m->dimensions);
		PgvIvfflatOpclass(index, &m->metric, &m->dtype, &ops);
		PgvStage(index, m);
	}
	return m;
}
