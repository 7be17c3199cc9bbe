/*
 * pgv_gpu.h -- what the pgvector extension gains to run its distance hot path on libpgv_hip.so.
 *
 * The files in ext/ are the glue INTEGRATION.md describes, as C a maintainer drops into src/ and
 * adds to OBJS: they call the C ABI of include/pgv_hip.h and nothing else of this repository.  No
 * PostgreSQL headers exist where this repository is built, so they are type-checked against the
 * stand-in declarations of ext/shim/ (tests/test_ext_glue_cpu.py) and exercised functionally through
 * their twins over the emulated page image (pgvector_amd/host/, tests/).
 *
 * Hook points in the reference (one line each; the reference code stays as the `vector.gpu = off` path):
 *   _PG_init            src/vector.c:57-65      PgvGpuInit();   (shared_preload_libraries = 'vector' for the mirror registry)
 *   ivfflatbeginscan    src/ivfscan.c:252-317   so->gpu = PgvIvfflatBeginScan(index, so);
 *   ivfflatrescan       src/ivfscan.c:322-356   PgvIvfflatRescan(so->gpu);
 *   ivfflatgettuple     src/ivfscan.c:361-414   if (so->gpu) { int r = PgvIvfflatGetTuple(scan); if (r >= 0) return r != 0; }
 *                                               (-1: this scan goes on in the reference's code -- from its first tuple)
 *                       src/ivfscan.c:408-413   if (so->gpu && PgvIvfflatAlreadyReturned(so->gpu, heaptid)) continue;
 *                                               (in a loop round the tuplesort pull: a pooled scan whose mirror was restaged
 *                                               under it restarts on the CPU path and must not return a tuple twice)
 *   ivfflatendscan      src/ivfscan.c:419-431   PgvIvfflatEndScan(so->gpu);
 *   hnswbeginscan       src/hnswscan.c:121-146  so->gpu = PgvHnswBeginScan(index);
 *   hnswgettuple        src/hnswscan.c:228      if (!(so->gpu && PgvHnswGetScanItems(scan, value, &so->w))) so->w = GetScanItems(scan, value);
 *                                               (false: NULL query or hnsw.iterative_scan -- the reference's scan)
 *   hnswendscan         src/hnswscan.c:337-349  PgvHnswEndScan(so->gpu);
 *   InitBuildState      src/hnswbuild.c:660-720 buildstate->gpu = PgvHnswBuildBegin(buildstate);
 *   InsertTuple         src/hnswbuild.c:575     if (buildstate->gpu) PgvHnswBuildDefer(buildstate, element); else InsertTupleInMemory(buildstate, element);
 *   FlushPages          src/hnswbuild.c:304     PgvHnswBuildLink(buildstate);   (first statement: links what was deferred)
 *   IvfflatKmeans       src/ivfkmeans.c:553-570 if (PgvIvfflatKmeans(index, samples, centers, typeInfo)) return;
 *   BuildCallback       src/ivfbuild.c:224-266  if (buildstate->gpu) { PgvIvfflatBuildAdd(buildstate, tid, value); return; }
 *   AssignTuples        src/ivfbuild.c:600-636  PgvIvfflatBuildFlush(buildstate) after the heap scan
 *   ivfflatinsert       src/ivfinsert.c:186-205 PgvNoteIndexChange(index);   after InsertTuple
 *   ivfflatbulkdelete   src/ivfvacuum.c:18-143  PgvNoteIndexChange(index);   when tuples were removed
 *   ivfflatbuild        src/ivfbuild.c:1063     PgvNoteIndexChange(index);   a rebuilt index is a new image
 *   hnswinsert / hnswbulkdelete / hnswbuild     PgvNoteIndexChange(index);   likewise
 * Page changes do not send relcache invalidations; PgvNoteIndexChange bumps the index's generation in the shared
 * registry (one atomic add), scans compare it with the generation their mirror was staged at (pgv_context.c).
 */
#ifndef PGV_GPU_H
#define PGV_GPU_H

#include "postgres.h"

#include "access/relscan.h"
#include "utils/rel.h"

#include "ivfflat.h"
#include "pgv_hip.h"

extern bool vector_gpu;			/* GUC vector.gpu */
extern int	vector_gpu_device;	/* GUC vector.gpu_device */
extern bool vector_gpu_pooled;	/* GUC vector.gpu_pooled: scans hand their query to the GPU worker's pooler */
extern int	vector_gpu_max_own_contexts;	/* GUC: backends with a device context of their own (default 4: the device runs
											 * four processes' queues side by side and time-slices the rest) */
/* no context for this backend and none to be had: its scans take the pooled path whatever vector.gpu_pooled says */
bool		PgvOwnContextsExhausted(void);

void		PgvGpuInit(void);
pgv_ctx    *PgvGetContext(void);	/* ERROR when there is no device (the worker's staging: caught, the index stays on the CPU path) */
pgv_ctx    *PgvTryGetContext(void);	/* NULL + one WARNING instead: the scan and build hooks go back to the reference's code */

/* FUNCTION 1 / element type of an ivfflat opclass (sql/vector.sql:406-425, :819-841); false for opclasses whose
 * index tuples are not dense float rows (bit_hamming_ops): they stay on the CPU path */
bool		PgvIvfflatOpclass(Relation index, pgv_metric * metric, pgv_dtype * dtype, pgv_ops * ops);

/* this backend's view of an index's device mirror: an import of what the GPU worker staged and exported */
typedef struct PgvIvfMirror
{
	Oid			relid;
	bool		valid;
	uint64		staged;			/* the staging (registry generation + 1) this import belongs to */
	pgv_index  *index;
	int			lists;
	int			dimensions;
	pgv_dtype	dtype;
	pgv_metric	metric;
	int64		ntuples;
	int			users;			/* open scans holding this import: it is unmapped when the last one lets go */
	bool		retired;		/* a newer staging has replaced it for new scans */
	struct PgvIvfMirror *next;
}			PgvIvfMirror;

/*
 * The current mirror with a reference taken for the caller (PgvIvfflatReleaseMirror when the scan ends), or NULL: no
 * current mirror (being staged, stale, unsupported opclass) -- the scan stays on the CPU path.  wantStaged != 0 asks for
 * exactly that staging (a pooled scan going on by itself must continue on the image its head came from).
 */
PgvIvfMirror *PgvIvfflatGetMirror(Relation index, uint64 wantStaged);
void		PgvIvfflatReleaseMirror(PgvIvfMirror * mirror);
/* the worker holds a current mirror of the index (a staging is requested otherwise) */
bool		PgvIvfflatMirrorIsCurrent(Relation index);
/*
 * The pooler: one query to the worker, the head of its sorted stream back (at most PGV_POOL_HEAD heap TIDs as
 * (block << 16) | offset; *outComplete = the probed lists hold no more tuples than came back).  false = not served.
 */
#define PGV_POOL_HEAD 64
bool		PgvPoolSearch(Relation index, const void *payload, int probes, float *outDist, uint64 *outTid, int *outCount,
						  bool *outComplete, uint64 *outStaged);
/* insert / vacuum / build changed the index's pages: mirrors staged before now are stale */
void		PgvNoteIndexChange(Relation index);
/* entries of the shared mirror registry in use (monitoring; dropped indexes give theirs back) */
int			PgvRegistryEntries(void);
/* bgw_main of the per-database worker that owns the mirrors */
void		PgvWorkerMain(Datum main_arg);
/* the worker's heartbeat, for everything of its that takes long (the stagers beat once per page; a no-op in a backend) */
void		PgvWorkerBeat(void);
/* a published handle could not be imported (its exporter is gone): forget the staging, have the index staged again */
void		PgvMirrorImportFailed(Relation index, uint64 staged);

/* scan side (ivfscan_gpu.c) */
void	   *PgvIvfflatBeginScan(Relation index, IvfflatScanOpaque so);
void		PgvIvfflatRescan(void *gpu);
int			PgvIvfflatGetTuple(IndexScanDesc scan);	/* 1 a tuple, 0 no more, -1 not served: the reference's path */
bool		PgvIvfflatAlreadyReturned(void *gpu, ItemPointer heaptid);	/* after a -1 in mid-scan: skip what the GPU path gave out */
void		PgvIvfflatEndScan(void *gpu);

/* registry kinds (pgv_context.c) */
#define PGV_KIND_IVFFLAT 0
#define PGV_KIND_HNSW 1
/* the worker's staging of an hnsw index (hnswscan_gpu.c): elements, graph, heap TIDs as the elements' payload;
 * NULL = not a vector / halfvec opclass.  *m, *dimensions, *nelements describe it. */
pgv_hnsw   *PgvHnswStage(Relation index, int *m, int *dimensions, int64 *nelements);
/* a backend's way to the worker's export of it (pgv_context.c); false = none now (a staging has been requested) */
bool		PgvHnswMirrorHandle(Relation index, pgv_index_handle * handle, uint64 *staged, int *m, int64 *nelements);

/* HNSW scan side (hnswscan_gpu.c); List as in nodes/pg_list.h */
void	   *PgvHnswBeginScan(Relation index);
bool		PgvHnswGetScanItems(IndexScanDesc scan, Datum value, List **w);	/* false: not served, the reference's GetScanItems */
void		PgvHnswEndScan(void *gpu);
void		PgvHnswInvalidate(Oid relid);
/* vector / halfvec element type of an hnsw opclass; false for bit and sparsevec opclasses (CPU path) */
bool		PgvHnswElementType(Relation index, pgv_dtype * dtype);
/* FUNCTION 1 of an hnsw opclass that has no FUNCTION 2: L2 or inner product (by the support function's oid) */
pgv_metric	PgvHnswMetricOf(Relation index);

/* HNSW build side (hnswbuild_gpu.c): the in-memory insertions deferred and linked at once before FlushPages
 * (HnswBuildState / HnswElement of src/hnsw.h, which this header does not include) */
struct HnswBuildState;
struct HnswElementData;
void	   *PgvHnswBuildBegin(struct HnswBuildState *buildstate);	/* NULL: vector.gpu off, no device, a bit / sparsevec opclass */
bool		PgvHnswBuildDefer(struct HnswBuildState *buildstate, struct HnswElementData *element);	/* false: the caller inserts it */
void		PgvHnswBuildLink(struct HnswBuildState *buildstate);

/* build side (ivfbuild_gpu.c) */
bool		PgvIvfflatKmeans(Relation index, VectorArray samples, VectorArray centers, const IvfflatTypeInfo * typeInfo);
void		PgvIvfflatBuildBegin(IvfflatBuildState * buildstate);
void		PgvIvfflatBuildAdd(IvfflatBuildState * buildstate, ItemPointer tid, Datum value);
void		PgvIvfflatBuildFlush(IvfflatBuildState * buildstate);

/* the tuplesort feed of AddTupleToSort (src/ivfbuild.c:203-216), left in the reference's file */
void		IvfflatAddToSort(IvfflatBuildState * buildstate, int list, ItemPointer tid, Datum value);

#endif							/* PGV_GPU_H */
