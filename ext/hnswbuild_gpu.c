/*
 * hnswbuild_gpu.c -- the in-memory phase of CREATE INDEX ... USING hnsw with every distance on the device.
 *
 * The reference inserts one element at a time while the graph fits maintenance_work_mem: InsertTuple
 * (src/hnswbuild.c:486-579) allocates the element in graph memory (HnswInitElement draws its level, the value is
 * copied behind it) and InsertTupleInMemory (:436-476) searches the graph for it, selects its neighbors and links it
 * into theirs; FlushPages (:304-316) then serialises the element list.  Here the second half is DEFERRED: elements are
 * allocated exactly as before -- same allocator, same memory accounting, same RandomDouble() per heap tuple, so the
 * levels and the "graph no longer fits" decision are the reference's own --, and right before the list is serialised
 * the whole batch is linked at once by pgv_host_hnsw_build (pgvector_amd/host/hnsw_build.c: the reference's insertion
 * loop replayed over distances the GPU computes in bulk; batches of elements inserted "concurrently" are a legal
 * interleaving of the reference's parallel build workers, max_batch = 1 IS its serial loop).  The result is written
 * into the very structures FlushPages reads: the element list (newest first, :366-374), the neighbor arrays, heap TIDs
 * of duplicates on the element that took them (:318-364), the entry point.
 *
 * Hook points (ext/pgvector-0.8.6-gpu.patch; `vector.gpu = off`, parallel builds and bit / sparsevec opclasses keep the
 * reference's code):
 *   InitBuildState       src/hnswbuild.c:660-742  buildstate->gpu = PgvHnswBuildBegin(buildstate);
 *   InsertTuple          src/hnswbuild.c:575      if (!PgvHnswBuildDefer(buildstate, element)) InsertTupleInMemory(buildstate, element);
 *   FlushPages           src/hnswbuild.c:304      PgvHnswBuildLink(buildstate);   (first statement; a no-op without deferred elements)
 * A PARALLEL build is recognised where it can be: InitBuildState runs before HnswParallelScanAndInsert sets
 * buildstate.hnswarea (:803-805), for every participant, so Begin cannot know; Defer sees hnswarea set on the first
 * tuple, gives the GPU state up and returns false from then on -- the participants insert under the reference's
 * locks, nothing is deferred, and the leader's FlushPages finds nothing to link.
 * Tuples that arrive after a flush go through HnswInsertTupleOnDisk as before (they search pages, not this graph).
 *
 * Unlike the other files of ext/, this one calls ONE function that is not in include/pgv_hip.h: pgv_host_hnsw_build,
 * plain C over the build-side ABI calls (pgv_hnsw_build_neighbors, pgv_hnsw_link_begin / prepare / apply / end: searches,
 * SelectNeighbors and HnswUpdateConnection all run on the device, the host keeps the batch order and the duplicates)
 * that a maintainer adds to OBJS with this file (or links as libpgv_host); INTEGRATION.md 5c.
 */
#include "pgv_gpu.h"

#include "hnsw.h"
#include "miscadmin.h"
#include "utils/memutils.h"

#include <math.h>

/* pgvector_amd/host/pgv_host.h: the declarations this file needs of it */
typedef struct pgv_hnsw_built
{
	int64_t		n;
	int			m;
	int32_t		entry;
	int32_t    *levels;
	int64_t    *nbr_start;
	int32_t    *nbr;
	int32_t    *dup_of;
	int64_t		nelements;
	int64_t		device_pairs;
	int64_t		batches;
	int64_t		deferred_updates;
	double		phase_secs[8];
}			pgv_hnsw_built;
extern int	pgv_host_hnsw_build(pgv_hnsw * mirror, pgv_dtype dtype, int dim, const void *rows, int64_t n, int m,
								int ef_construction, const pgv_rng * rng, int max_batch, pgv_hnsw_built * out);
extern void pgv_host_hnsw_built_free(pgv_hnsw_built * built);
typedef int (*pgv_host_cancel_check) (void *arg);
extern void pgv_host_hnsw_set_cancel_check(pgv_host_cancel_check check, void *arg);
extern const char *pgv_host_last_error(void);

/* elements inserted "at once": the GUC vector.gpu_hnsw_build_batch (default 1024; pgv_host_hnsw_build falls back to smaller
 * batches while the graph is small: it must hold 16 x as many elements) */

typedef struct PgvHnswBuild
{
	pgv_metric	metric;
	pgv_dtype	dtype;
	Size		rowBytes;
	HnswElement *elements;		/* the deferred elements, in heap order */
	int64		count,
				capacity;
}			PgvHnswBuild;

void *
PgvHnswBuildBegin(HnswBuildState * buildstate)
{
	PgvHnswBuild *gb;
	pgv_dtype	dtype;

	/* (hnswarea is still NULL here even for a parallel build's participants: PgvHnswBuildDefer decides that) */
	if (!vector_gpu || buildstate->hnswarea != NULL || !PgvHnswElementType(buildstate->index, &dtype))
		return NULL;
	if (PgvTryGetContext() == NULL)
		return NULL;			/* no device: the reference's InsertTupleInMemory */
	gb = palloc0(sizeof(PgvHnswBuild));
	gb->dtype = dtype;
	/* FUNCTION 1 of the opclass; rows that FUNCTION 2 normalised (cosine) are compared by inner product */
	gb->metric = HnswOptionalProcInfo(buildstate->index, HNSW_NORM_PROC) != NULL ? PGV_NEG_IP : PgvHnswMetricOf(buildstate->index);
	gb->rowBytes = (dtype == PGV_F32 ? sizeof(float) : sizeof(uint16)) * (Size) buildstate->dimensions;
	/* maintenance_work_mem covers the link phase too: PgvHnswBuildLink copies every element's value into dense rows and
	 * pgv_host_hnsw_build keeps its own neighbor tables -- about the graph's size again, held while the graph still is.
	 * InitGraph (src/hnswbuild.c:615-634) has just given the graph the whole budget: it gets half, the "no longer fits"
	 * flush comes at half the tuples, and the peak stays inside what the user allowed.  (The participants of a
	 * parallel build switch to the shared graph after this and never link: their budget is untouched.) */
	{
		/* per element the graph holds the element struct, its value (header + payload) and m (2 on layer 0) HnswCandidate
		 * slots of 16 bytes per layer -- about rowBytes + 700 bytes at m = 16; the link phase adds the dense row and 4 bytes
		 * per neighbor slot, levels, duplicates: about rowBytes + 200.  The graph's share of the budget is therefore
		 * (rowBytes + 700) / (2 rowBytes + 900): one half for long rows, 0.63 at 128 dimensions (not a fixed half) */
		double		share = ((double) gb->rowBytes + 700.0) / (2.0 * (double) gb->rowBytes + 900.0);

		buildstate->graphData.memoryTotal = (Size) ((double) buildstate->graphData.memoryTotal * share);
		ereport(DEBUG1, (errmsg("pgvector GPU path: hnsw build keeps %.0f %% of maintenance_work_mem for the graph, the rest for the device link phase",
								share * 100.0)));
	}
	return gb;
}

/* InsertTuple has allocated `element` (level drawn, value copied, lock initialised): remember it instead of searching
 * the graph for it now.  false: not deferred, the caller inserts it (InsertTupleInMemory) -- the CPU path, and every
 * participant of a parallel build: they share one graph in hnswarea under the reference's locks */
bool
PgvHnswBuildDefer(HnswBuildState * buildstate, HnswElement element)
{
	PgvHnswBuild *gb = (PgvHnswBuild *) buildstate->gpu;
	MemoryContext old;

	if (gb == NULL)
		return false;
	if (buildstate->hnswarea != NULL)
	{
		/* set after InitBuildState by HnswParallelScanAndInsert (src/hnswbuild.c:803-805): a parallel build */
		ereport(DEBUG1, (errmsg("pgvector GPU path: parallel hnsw build, this participant inserts on the CPU path")));
		pfree(gb);
		buildstate->gpu = NULL;
		/* the context PgvHnswBuildBegin made for this participant is of no use to it: give it (and its slot of
		 * vector.gpu_max_own_contexts) back now, not at backend exit */
		PgvReleaseIdleContext();
		return false;
	}

	if (gb->count == gb->capacity)
	{
		/* (the array outlives the per-tuple context BuildCallback runs in) */
		old = MemoryContextSwitchTo(buildstate->graphCtx);
		gb->capacity = gb->capacity ? gb->capacity * 2 : 65536;
		gb->elements = gb->elements ? repalloc_huge(gb->elements, sizeof(HnswElement) * (Size) gb->capacity)
			: palloc_extended(sizeof(HnswElement) * (Size) gb->capacity, MCXT_ALLOC_HUGE);
		MemoryContextSwitchTo(old);
	}
	gb->elements[gb->count++] = element;
	return true;
}

/* the levels HnswInitElement drew, handed to pgv_host_hnsw_build as the uniform draws that give them back:
 * (int) (-log(u) * ml) == level for u = exp(-(level + 0.5) / ml) */
typedef struct PgvLevelReplay
{
	HnswElement *elements;
	int64		next;
	double		ml;
}			PgvLevelReplay;

/* polled by pgv_host_hnsw_build between batches, on this thread: a pending cancel / terminate ends the link phase
 * cleanly (helpers joined, memory freed); the ERROR itself is raised here afterwards, by CHECK_FOR_INTERRUPTS() */
static int
PgvBuildCancelPending(void *arg)
{
	(void) arg;
	/* (not InterruptPending: that is also raised for catch-up and procsignal work, which CHECK_FOR_INTERRUPTS serves without
	 * an error -- the build must not give up for those) */
	return QueryCancelPending || ProcDiePending;
}

static double
PgvReplayLevel(void *state)
{
	PgvLevelReplay *r = (PgvLevelReplay *) state;

	return exp(-((double) r->elements[r->next++]->level + 0.5) / r->ml);
}

/*
 * Link everything that was deferred: called as the first statement of FlushPages, i.e. when the heap scan is over or
 * when the graph stops fitting into maintenance_work_mem -- both times the element list is about to be serialised and
 * the in-memory graph is not searched again.
 */
void
PgvHnswBuildLink(HnswBuildState * buildstate)
{
	PgvHnswBuild *gb = (PgvHnswBuild *) buildstate->gpu;
	char	   *base = buildstate->hnswarea;
	HnswGraph  *graph = buildstate->graph;
	int64		n;
	char	   *rows;
	pgv_hnsw   *mirror = NULL;
	pgv_hnsw_built built;
	PgvLevelReplay replay;
	pgv_rng		rng;
	int			rc;

	if (gb == NULL || gb->count == 0)
		return;
	n = gb->count;
	/* the values as dense rows, in heap order (Vector and HalfVector share the header; the payload starts at ->x) */
	rows = palloc_extended(gb->rowBytes * (Size) n, MCXT_ALLOC_HUGE);
	for (int64 e = 0; e < n; e++)
	{
		if ((e & 0xffff) == 0)
			CHECK_FOR_INTERRUPTS();
		memcpy(rows + gb->rowBytes * (Size) e, ((Vector *) HnswPtrAccess(base, gb->elements[e]->value))->x, gb->rowBytes);
	}
	if (pgv_hnsw_upload(PgvGetContext(), gb->metric, gb->dtype, buildstate->dimensions, rows, n, &mirror) != PGV_OK)
		ereport(ERROR, (errmsg("pgvector GPU path: %s", pgv_last_error())));

	replay.elements = gb->elements;
	replay.next = 0;
	replay.ml = buildstate->ml;
	memset(&rng, 0, sizeof(rng));
	rng.next_double = PgvReplayLevel;
	rng.state = &replay;
	pgv_host_hnsw_set_cancel_check(PgvBuildCancelPending, NULL);
	rc = pgv_host_hnsw_build(mirror, gb->dtype, buildstate->dimensions, rows, n, buildstate->m, buildstate->efConstruction,
							 &rng, vector_gpu_hnsw_build_batch, &built);
	pgv_host_hnsw_set_cancel_check(NULL, NULL);
	pgv_hnsw_free(mirror);
	pfree(rows);
	/* a cancel that arrived after the last between-batch poll: the interrupt's ERROR longjmps out of here, and `built`
	 * (O(n m) of malloc'ed tables) would stay with the backend */
	if (rc == PGV_OK && PgvBuildCancelPending(NULL))
	{
		pgv_host_hnsw_built_free(&built);
		rc = PGV_ERR_STATE;
	}
	CHECK_FOR_INTERRUPTS();		/* a cancelled build: the interrupt's own ERROR, not ours */
	if (rc != PGV_OK)
		ereport(ERROR, (errmsg("pgvector GPU path: %s", pgv_host_last_error())));

	/* ---- the graph, into the structures FlushPages serialises */
	for (int64 e = 0; e < n; e++)
	{
		HnswElement element = gb->elements[e];

		if (built.levels[e] != (int32) element->level)
		{
			pgv_host_hnsw_built_free(&built);
			elog(ERROR, "pgvector GPU path: level replay out of step at element " INT64_FORMAT, e);
		}
		if (built.dup_of[e] >= 0)
		{
			/* FindDuplicateInMemory + AddDuplicateInMemory (:318-364): the row's heap TID went to an element with
			 * the same value; this element is not part of the graph */
			HnswAddHeapTid(gb->elements[built.dup_of[e]], &element->heaptids[0]);
			continue;
		}
		/* AddElementInMemory (:366-374): head insertion, the list is newest first */
		element->next = graph->head;
		HnswPtrStore(base, graph->head, element);
		/* the neighbor tuple's layout: layers from the element's top one down, HnswGetLayerM slots each, -1 = empty */
		for (int lc = element->level; lc >= 0; lc--)
		{
			HnswNeighborArray *a = HnswGetNeighbors(base, element, lc);
			int			lm = HnswGetLayerM(buildstate->m, lc);
			const int32 *slots = built.nbr + built.nbr_start[e] + (int64) (element->level - lc) * buildstate->m;

			a->length = 0;
			a->closerSet = false;
			for (int i = 0; i < lm && slots[i] >= 0; i++)
			{
				HnswCandidate *hc = &a->items[a->length++];

				HnswPtrStore(base, hc->element, gb->elements[slots[i]]);
				hc->distance = 0;	/* (not serialised: HnswSetNeighborTuple writes the elements' TIDs only) */
				hc->closer = false;
			}
		}
	}
	if (built.entry >= 0)
		HnswPtrStore(base, graph->entryPoint, gb->elements[built.entry]);
	ereport(DEBUG1, (errmsg("pgvector GPU path: " INT64_FORMAT " deferred elements linked on the device (" INT64_FORMAT " batches)", n,
							(int64) built.batches)));
	pgv_host_hnsw_built_free(&built);
	/* the list is about to be written out and graphCtx reset with it: nothing is deferred any more */
	gb->elements = NULL;
	gb->count = gb->capacity = 0;
}
