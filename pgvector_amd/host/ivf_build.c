/*
 * ivf_build.c -- BuildIndex (src/ivfbuild.c:1040-1058) with the distance loops on
 * the GPU: ComputeCenters -> pgv_kmeans, AssignTuples/AddTupleToSort -> pgv_assign
 * in batches, tuplesort by list -> stable counting sort, then the page writers.
 */
#include "pgv_host.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

extern int	pgv_host_fail(int code, const char *fmt,...);
extern void *pgv_host_big_alloc(size_t bytes);

#define ASSIGN_BATCH (1 << 18)	/* rows handed to the GPU per BuildCallback batch */

static __thread double build_phase_secs[5];

static double
now_secs(void)
{
	struct timespec ts;

	clock_gettime(CLOCK_MONOTONIC, &ts);
	return (double) ts.tv_sec + 1e-9 * (double) ts.tv_nsec;
}

void
pgv_host_ivf_build_phases(double out_secs[5])
{
	memcpy(out_secs, build_phase_secs, sizeof(build_phase_secs));
}

/*
 * Normalise `n` values into a new array, dropping the zero-norm ones (IvfflatCheckNorm +
 * IvfflatNormValue, src/ivfbuild.c:148-156, :174-180); keep[i] = source index of output row i.
 */
static void *
normalize_all(pgv_dtype dtype, int dim, const void *in, int64_t n, int64_t *out_n, int64_t **out_keep)
{
	size_t		row_bytes = (size_t) dim * (dtype == PGV_F32 ? 4 : 2);
	char	   *out = malloc(row_bytes * (size_t) (n > 0 ? n : 1));
	uint8_t    *ok = malloc((size_t) (n > 0 ? n : 1));
	int64_t    *keep = malloc(sizeof(int64_t) * (size_t) (n > 0 ? n : 1));
	int64_t		m = 0;

	if (!out || !ok || !keep)
	{
		free(out);
		free(ok);
		free(keep);
		return NULL;
	}
	/* every value on its own: in place first, compacted afterwards (order kept) */
#pragma omp parallel for schedule(static)
	for (int64_t i = 0; i < n; i++)
		ok[i] = (uint8_t) pgv_host_normalize_value(dtype, dim, (const char *) in + (size_t) i * row_bytes, out + (size_t) i * row_bytes);
	for (int64_t i = 0; i < n; i++)
		if (ok[i])
		{
			if (m != i)
				memmove(out + (size_t) m * row_bytes, out + (size_t) i * row_bytes, row_bytes);
			keep[m++] = i;
		}
	free(ok);
	*out_n = m;
	*out_keep = keep;
	return out;
}

int
pgv_host_ivf_build(pgv_ctx * ctx, pgv_ops ops, pgv_dtype dtype, int dim, int lists,
				   const void *rows, const uint64_t *tids, int64_t n,
				   const void *samples, int nsamples, const pgv_rng * rng, pgv_rel * out_rel)
{
	size_t		es = dtype == PGV_F32 ? 4 : 2;
	size_t		row_bytes = (size_t) dim * es;
	void	   *centers = NULL;
	int32_t    *list_of = NULL;
	int64_t    *offsets = NULL;
	int64_t    *dest = NULL;
	void	   *sorted = NULL;
	uint64_t   *sorted_tids = NULL;
	void	   *norm_rows = NULL,
			   *norm_samples = NULL;
	int64_t    *row_keep = NULL,
			   *sample_keep = NULL;
	pgv_metric	metric = ops == PGV_OPS_L2 ? PGV_L2SQ : PGV_NEG_IP;
	int			rc = PGV_OK;
	double		t0 = now_secs(),
				t1;

	memset(build_phase_secs, 0, sizeof(build_phase_secs));
	/* SampleCallback / BuildCallback normalisation for opclasses with the NORM procs */
	if (ops != PGV_OPS_L2 && nsamples > 0)
	{
		int64_t		m;

		norm_samples = normalize_all(dtype, dim, samples, nsamples, &m, &sample_keep);
		if (!norm_samples)
		{
			rc = pgv_host_fail(PGV_ERR_NOMEM, "out of memory");
			goto out;
		}
		samples = norm_samples;
		nsamples = (int) m;
	}
	if (ops == PGV_OPS_COSINE && n > 0)
	{
		int64_t		m;

		norm_rows = normalize_all(dtype, dim, rows, n, &m, &row_keep);
		if (!norm_rows)
		{
			rc = pgv_host_fail(PGV_ERR_NOMEM, "out of memory");
			goto out;
		}
		rows = norm_rows;
		n = m;
	}
	t1 = now_secs();
	build_phase_secs[0] = t1 - t0;
	t0 = t1;

	centers = malloc(row_bytes * (size_t) lists);
	list_of = malloc(sizeof(int32_t) * (size_t) (n > 0 ? n : 1));
	offsets = calloc((size_t) lists + 1, sizeof(int64_t));
	dest = malloc(sizeof(int64_t) * (size_t) (n > 0 ? n : 1));
	sorted = pgv_host_big_alloc(row_bytes * (size_t) (n > 0 ? n : 1));
	sorted_tids = malloc(sizeof(uint64_t) * (size_t) (n > 0 ? n : 1));
	if (!centers || !list_of || !offsets || !dest || !sorted || !sorted_tids)
	{
		rc = pgv_host_fail(PGV_ERR_NOMEM, "out of memory");
		goto out;
	}

	/* ComputeCenters, src/ivfbuild.c:434-480 */
	rc = pgv_kmeans(ctx, ops, dtype, dim, samples, nsamples, lists, 500, rng, centers, NULL, NULL);
	if (rc != PGV_OK)
	{
		pgv_host_fail(rc, "%s", pgv_last_error());
		goto out;
	}
	t1 = now_secs();
	build_phase_secs[1] = t1 - t0;
	t0 = t1;
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
	t1 = now_secs();
	build_phase_secs[2] = t1 - t0;
	t0 = t1;
	/* tuplesort on the list id (src/ivfbuild.c:606-615), heap order kept inside a list: destinations by a
	 * serial counting pass (8 bytes per row), the 3-6 KB payload copies in parallel */
	for (int64_t r = 0; r < n; r++)
		offsets[list_of[r] + 1]++;
	for (int l = 0; l < lists; l++)
		offsets[l + 1] += offsets[l];
	{
		int64_t    *fill = malloc(sizeof(int64_t) * (size_t) lists);

		if (!fill)
		{
			rc = pgv_host_fail(PGV_ERR_NOMEM, "out of memory");
			goto out;
		}
		memcpy(fill, offsets, sizeof(int64_t) * (size_t) lists);
		for (int64_t r = 0; r < n; r++)
			dest[r] = fill[list_of[r]]++;
		free(fill);
	}
#pragma omp parallel for schedule(static)
	for (int64_t r = 0; r < n; r++)
	{
		memcpy((char *) sorted + (size_t) dest[r] * row_bytes, (const char *) rows + (size_t) r * row_bytes, row_bytes);
		sorted_tids[dest[r]] = tids[row_keep ? row_keep[r] : r];
	}
	t1 = now_secs();
	build_phase_secs[3] = t1 - t0;
	t0 = t1;
	rc = pgv_host_ivf_write_index(out_rel, dtype, dim, lists, centers, offsets, sorted, sorted_tids);
	build_phase_secs[4] = now_secs() - t0;
out:
	free(dest);
	free(sorted);
	free(sorted_tids);
	free(offsets);
	free(list_of);
	free(centers);
	free(norm_rows);
	free(norm_samples);
	free(row_keep);
	free(sample_keep);
	return rc;
}
