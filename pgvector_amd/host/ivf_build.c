/*
 * ivf_build.c -- BuildIndex (src/ivfbuild.c:1040-1058) with the distance loops on
 * the GPU: ComputeCenters -> pgv_kmeans, AssignTuples/AddTupleToSort -> pgv_assign
 * in batches, tuplesort by list -> stable counting sort, then the page writers.
 */
#include "pgv_host.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

extern int	pgv_host_fail(int code, const char *fmt,...);

#define ASSIGN_BATCH 65536		/* rows handed to the GPU per BuildCallback batch */

int
pgv_host_ivf_build(pgv_ctx * ctx, pgv_ops ops, pgv_dtype dtype, int dim, int lists,
				   const void *rows, const uint64_t *tids, int64_t n,
				   const void *samples, int nsamples, const pgv_rng * rng, pgv_rel * out_rel)
{
	size_t		es = dtype == PGV_F32 ? 4 : 2;
	size_t		row_bytes = (size_t) dim * es;
	void	   *centers = malloc(row_bytes * (size_t) lists);
	int32_t    *list_of = malloc(sizeof(int32_t) * (size_t) (n > 0 ? n : 1));
	int64_t    *offsets = calloc((size_t) lists + 1, sizeof(int64_t));
	int64_t    *fill = NULL;
	void	   *sorted = NULL;
	uint64_t   *sorted_tids = NULL;
	pgv_metric	metric = ops == PGV_OPS_L2 ? PGV_L2SQ : PGV_NEG_IP;
	int			rc;

	/* ComputeCenters, src/ivfbuild.c:434-480 */
	rc = pgv_kmeans(ctx, ops, dtype, dim, samples, nsamples, lists, 500, rng, centers, NULL, NULL);
	if (rc != PGV_OK)
	{
		pgv_host_fail(rc, "%s", pgv_last_error());
		goto out;
	}
	/* AssignTuples: every heap row to its nearest center, in callback-sized batches */
	for (int64_t r0 = 0; r0 < n; r0 += ASSIGN_BATCH)
	{
		int64_t		cnt = n - r0 < ASSIGN_BATCH ? n - r0 : ASSIGN_BATCH;

		rc = pgv_assign(ctx, metric, dtype, dim, centers, lists, (const char *) rows + (size_t) r0 * row_bytes, cnt,
						list_of + r0, NULL);
		if (rc != PGV_OK)
		{
			pgv_host_fail(rc, "%s", pgv_last_error());
			goto out;
		}
	}
	/* tuplesort on the list id (src/ivfbuild.c:606-615), heap order kept inside a list */
	for (int64_t r = 0; r < n; r++)
		offsets[list_of[r] + 1]++;
	for (int l = 0; l < lists; l++)
		offsets[l + 1] += offsets[l];
	fill = malloc(sizeof(int64_t) * (size_t) lists);
	memcpy(fill, offsets, sizeof(int64_t) * (size_t) lists);
	sorted = malloc(row_bytes * (size_t) (n > 0 ? n : 1));
	sorted_tids = malloc(sizeof(uint64_t) * (size_t) (n > 0 ? n : 1));
	for (int64_t r = 0; r < n; r++)
	{
		int64_t		at = fill[list_of[r]]++;

		memcpy((char *) sorted + (size_t) at * row_bytes, (const char *) rows + (size_t) r * row_bytes, row_bytes);
		sorted_tids[at] = tids[r];
	}
	rc = pgv_host_ivf_write_index(out_rel, dtype, dim, lists, centers, offsets, sorted, sorted_tids);
out:
	free(fill);
	free(sorted);
	free(sorted_tids);
	free(offsets);
	free(list_of);
	free(centers);
	return rc;
}
