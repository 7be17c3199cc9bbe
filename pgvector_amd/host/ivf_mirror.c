/*
 * ivf_mirror.c -- lifecycle of an IVFFlat index's device mirror.
 *
 * The GPU scans a contiguous list-major image of the index (pgv_index), not its
 * pages.  Inside a server that image belongs in backend-local memory keyed by the
 * relfilenode and is dropped by the relcache invalidation callback whenever an
 * insert (src/ivfinsert.c:72-181), a vacuum (src/ivfvacuum.c:18-143) or a rebuild
 * changed pages.  Here the relation is an array of pages with a generation counter
 * that every page-changing call bumps; the mirror restages (the page walk of
 * src/ivfscan.c:58-111, :139-179) and re-uploads when the counter has moved.
 */
#include "pgv_host.h"

#include <stdlib.h>
#include <string.h>

extern int	pgv_host_fail(int code, const char *fmt,...);

struct pgv_ivf_mirror
{
	pgv_ctx    *ctx;
	pgv_metric	metric;
	pgv_dtype	dtype;
	const pgv_rel *rel;			/* what the image was staged from */
	uint64_t	generation;
	int			valid;
	int64_t		restages;
	pgv_ivf_image img;
	pgv_index  *index;
};

int
pgv_host_ivf_mirror_open(pgv_ctx * ctx, pgv_metric metric, pgv_dtype dtype, pgv_ivf_mirror * *out)
{
	pgv_ivf_mirror *m;

	if (!ctx || !out)
		return pgv_host_fail(PGV_ERR_ARG, "pgv_host_ivf_mirror_open: ctx/out is NULL");
	m = calloc(1, sizeof(*m));
	if (!m)
		return pgv_host_fail(PGV_ERR_NOMEM, "out of memory");
	m->ctx = ctx;
	m->metric = metric;
	m->dtype = dtype;
	*out = m;
	return PGV_OK;
}

static void
mirror_drop(pgv_ivf_mirror * m)
{
	if (m->index)
		pgv_index_free(m->index);
	m->index = NULL;
	if (m->valid)
		pgv_host_ivf_image_free(&m->img);
	m->valid = 0;
}

int
pgv_host_ivf_mirror_get(pgv_ivf_mirror * m, const pgv_rel * rel, pgv_index * *out_index,
						const pgv_ivf_image * *out_image)
{
	if (!m || !rel)
		return pgv_host_fail(PGV_ERR_ARG, "pgv_host_ivf_mirror_get: mirror/rel is NULL");
	if (!m->valid || m->rel != rel || m->generation != rel->generation)
	{
		int			rc;

		mirror_drop(m);
		rc = pgv_host_ivf_stage(rel, m->dtype, &m->img);
		if (rc != PGV_OK)
			return rc;
		m->valid = 1;
		rc = pgv_index_upload(m->ctx, m->metric, m->dtype, m->img.dim, m->img.lists, m->img.centers,
							  m->img.list_offsets, m->img.vectors, m->img.tids, &m->index);
		if (rc != PGV_OK)
		{
			mirror_drop(m);
			return pgv_host_fail(rc, "mirror upload failed: %s", pgv_last_error());
		}
		m->rel = rel;
		m->generation = rel->generation;
		m->restages++;
	}
	if (out_index)
		*out_index = m->index;
	if (out_image)
		*out_image = &m->img;
	return PGV_OK;
}

int64_t
pgv_host_ivf_mirror_restages(const pgv_ivf_mirror * m)
{
	return m ? m->restages : 0;
}

void
pgv_host_ivf_mirror_close(pgv_ivf_mirror * m)
{
	if (!m)
		return;
	mirror_drop(m);
	free(m);
}
