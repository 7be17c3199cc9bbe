/*
 * ivf_build.c -- BuildIndex (src/ivfbuild.c:1040-1058) with the distance loops and the sort on
 * the GPU: ComputeCenters -> pgv_kmeans, AssignTuples/AddTupleToSort -> pgv_builder_add in batches,
 * tuplesort by list -> pgv_builder_finish (its result is the device mirror), the page writers fed by pgv_index_drain.
 */
#include "pgv_host.h"

#include <math.h>
#include <pthread.h>
#include <stdio.h>
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

typedef struct
{
	pgv_ivf_writer *writer;
}			drain_sink;

static int
write_rows(void *arg, int64_t first_slot, int64_t count, const void *vectors, const uint64_t *tids)
{
	drain_sink *s = arg;

	return pgv_host_ivf_writer_fill(s->writer, first_slot, count, vectors, tids) == PGV_OK ? 0 : 1;
}

/* pgv_kmeans on a thread of its own; the library's last error is per thread, so it is carried out by hand */
typedef struct kmeans_job
{
	pgv_ctx    *ctx;
	pgv_ops		ops;
	pgv_dtype	dtype;
	int			dim;
	const void *samples;
	int			nsamples;
	int			lists;
	const pgv_rng *rng;
	void	   *centers;
	int			rc;
	char		err[256];
}			kmeans_job;

static void *
kmeans_main(void *arg)
{
	kmeans_job *job = arg;

	job->rc = pgv_kmeans(job->ctx, job->ops, job->dtype, job->dim, job->samples, job->nsamples, job->lists, 500, job->rng,
						 job->centers, NULL, NULL);
	if (job->rc != PGV_OK)
		snprintf(job->err, sizeof(job->err), "%s", pgv_last_error());
	return NULL;
}

int
pgv_host_ivf_build_mirror(pgv_ctx * ctx, pgv_ops ops, pgv_dtype dtype, int dim, int lists,
						  const void *rows, const uint64_t *tids, int64_t n,
						  const void *samples, int nsamples, const pgv_rng * rng, pgv_rel * out_rel,
						  pgv_index * *out_index)
{
	size_t		es = dtype == PGV_F32 ? 4 : 2;
	size_t		row_bytes = (size_t) dim * es;
	void	   *centers = NULL;
	int64_t    *offsets = NULL;
	uint64_t   *kept_tids = NULL;
	void	   *norm_rows = NULL,
			   *norm_samples = NULL;
	int64_t    *row_keep = NULL,
			   *sample_keep = NULL;
	pgv_ivf_writer *writer = NULL;
	pgv_builder *builder = NULL;
	pgv_index  *index = NULL;
	pgv_metric	metric = ops == PGV_OPS_L2 ? PGV_L2SQ : PGV_NEG_IP;
	int			rc = PGV_OK;
	double		t0 = now_secs(),
				t1;

	if (out_index)
		*out_index = NULL;
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
		/* rows with a zero norm are not indexed (src/ivfbuild.c:232-236): the TIDs of the kept ones */
		kept_tids = malloc(sizeof(uint64_t) * (size_t) (n > 0 ? n : 1));
		if (!kept_tids)
		{
			rc = pgv_host_fail(PGV_ERR_NOMEM, "out of memory");
			goto out;
		}
		for (int64_t r = 0; r < n; r++)
			kept_tids[r] = tids[row_keep[r]];
		tids = kept_tids;
	}
	t1 = now_secs();
	build_phase_secs[0] = t1 - t0;
	t0 = t1;

	centers = malloc(row_bytes * (size_t) lists);
	offsets = calloc((size_t) lists + 1, sizeof(int64_t));
	if (!centers || !offsets)
	{
		rc = pgv_host_fail(PGV_ERR_NOMEM, "out of memory");
		goto out;
	}
	/* the page array is allocated now and zeroed on background threads while the device works */
	rc = pgv_host_ivf_writer_begin(out_rel, dtype, dim, lists, n, &writer);
	if (rc != PGV_OK)
		goto out;

	/*
	 * ComputeCenters (src/ivfbuild.c:434-480) on a helper thread, and meanwhile the second pass over the heap: every
	 * row to the device in callback-sized batches (a builder begun without centers only copies, on a stream of its
	 * own).  The two overlap: k-means keeps the compute units busy, the upload the PCIe link.
	 */
	{
		kmeans_job	job = {ctx, ops, dtype, dim, samples, nsamples, lists, rng, centers, PGV_OK, {0}};
		pthread_t	th;
		int			threaded;

		rc = pgv_builder_begin(ctx, metric, dtype, dim, lists, NULL, n, &builder);
		if (rc != PGV_OK)
		{
			pgv_host_fail(rc, "%s", pgv_last_error());
			goto out;
		}
		threaded = pthread_create(&th, NULL, kmeans_main, &job) == 0;
		if (!threaded)
			kmeans_main(&job);
		for (int64_t r0 = 0; r0 < n && rc == PGV_OK; r0 += ASSIGN_BATCH)
		{
			int64_t		cnt = n - r0 < ASSIGN_BATCH ? n - r0 : ASSIGN_BATCH;

			rc = pgv_builder_add(builder, (const char *) rows + (size_t) r0 * row_bytes, tids + r0, cnt);
		}
		if (rc != PGV_OK)
			pgv_host_fail(rc, "%s", pgv_last_error());
		t1 = now_secs();
		build_phase_secs[2] = t1 - t0;	/* the upload (k-means running beside it) */
		t0 = t1;
		if (threaded)
			pthread_join(th, NULL);
		if (rc == PGV_OK && job.rc != PGV_OK)
			rc = pgv_host_fail(job.rc, "%s", job.err);
		if (rc != PGV_OK)
			goto out;
		t1 = now_secs();
		build_phase_secs[1] = t1 - t0;	/* what of k-means was left when the upload had finished */
		t0 = t1;
	}
	/* AssignTuples: every row to its nearest center, where it now is (with the next call) */
	rc = pgv_builder_set_centers(builder, centers);
	if (rc != PGV_OK)
	{
		pgv_host_fail(rc, "%s", pgv_last_error());
		goto out;
	}
	/* tuplesort on the list id (src/ivfbuild.c:606-615), heap order kept inside a list: a gather on the device,
	 * whose result is the mirror */
	rc = pgv_builder_finish(builder, &index, offsets, NULL);
	if (rc != PGV_OK)
	{
		pgv_host_fail(rc, "%s", pgv_last_error());
		goto out;
	}
	t1 = now_secs();
	build_phase_secs[3] = t1 - t0;
	t0 = t1;
	/* InsertTuples (:271-331): page headers from the list lengths, tuples as the sorted rows come back */
	rc = pgv_host_ivf_writer_layout(writer, centers, offsets);
	if (rc == PGV_OK && n > 0)
	{
		drain_sink	sink = {writer};

		rc = pgv_index_drain(index, 0, write_rows, &sink);
		if (rc != PGV_OK)
			pgv_host_fail(rc, "%s | %s", pgv_last_error(), pgv_host_last_error());
	}
	build_phase_secs[4] = now_secs() - t0;
out:
	pgv_host_ivf_writer_end(writer);
	if (builder)
		pgv_builder_free(builder);
	if (rc != PGV_OK)
	{
		pgv_rel_free(out_rel);
		if (index)
			pgv_index_free(index);
		index = NULL;
	}
	if (out_index)
		*out_index = index;
	else if (index)
		pgv_index_free(index);
	free(offsets);
	free(centers);
	free(kept_tids);
	free(norm_rows);
	free(norm_samples);
	free(row_keep);
	free(sample_keep);
	return rc;
}

int
pgv_host_ivf_build(pgv_ctx * ctx, pgv_ops ops, pgv_dtype dtype, int dim, int lists,
				   const void *rows, const uint64_t *tids, int64_t n,
				   const void *samples, int nsamples, const pgv_rng * rng, pgv_rel * out_rel)
{
	return pgv_host_ivf_build_mirror(ctx, ops, dtype, dim, lists, rows, tids, n, samples, nsamples, rng, out_rel, NULL);
}
