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

#ifdef __cplusplus
}
#endif
#endif
