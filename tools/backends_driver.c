/*
 * backends_driver.c -- N backends on one device mirror, in plain C over the C ABI (no Python in the timed
 * region): every thread is a backend with its own pgv_ctx (stream + scratch), a pgv_index_share view of the
 * uploaded index and a pgv_query handle, issuing one query at a time like ivfflatgettuple does
 * (pgv_query_rank + pgv_query_scan, src/ivfscan.c:361-414).  bench.py loads this as a shared object and hands it the
 * index it built.   gcc -O2 -shared -fPIC -pthread -Iinclude tools/backends_driver.c -o build/tools/libbackends.so
 */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "pgv_hip.h"
#include "pgv_host.h"

typedef struct
{
	pgv_index  *index;
	int			device;
	int			id;
	int			per_thread;
	const char *queries;
	int			nq;
	size_t		query_bytes;
	int			probes;
	int			k;
	pthread_barrier_t *start;
	double	   *lat;			/* [per_thread] seconds */
	double		t0,
				t1;
	int			rc;
}			backend;

static double
now(void)
{
	struct timespec ts;

	clock_gettime(CLOCK_MONOTONIC, &ts);
	return (double) ts.tv_sec + 1e-9 * (double) ts.tv_nsec;
}

static void *
backend_main(void *arg)
{
	backend    *b = arg;
	pgv_ctx    *ctx = NULL;
	pgv_index  *view = NULL;
	pgv_query  *q = NULL;
	float		dist[64];
	int64_t		slot[64];
	uint64_t	tid[64];
	int			count;
	int64_t		total;

	b->rc = pgv_ctx_create(b->device, NULL, &ctx);
	if (b->rc == PGV_OK)
		b->rc = pgv_index_share(b->index, ctx, &view);
	if (b->rc == PGV_OK)
		b->rc = pgv_query_begin(view, &q);
	for (int j = 0; j < 20 && b->rc == PGV_OK; j++)
	{
		b->rc = pgv_query_rank(q, b->queries + (size_t) ((b->id * 7 + j) % b->nq) * b->query_bytes, b->probes);
		if (b->rc == PGV_OK)
			b->rc = pgv_query_scan(q, 0, b->probes, b->k, dist, slot, tid, &count, &total);
	}
	pthread_barrier_wait(b->start);
	b->t0 = now();
	for (int j = 0; j < b->per_thread && b->rc == PGV_OK; j++)
	{
		double		t = now();

		b->rc = pgv_query_rank(q, b->queries + (size_t) ((b->id * 31 + j) % b->nq) * b->query_bytes, b->probes);
		if (b->rc == PGV_OK)
			b->rc = pgv_query_scan(q, 0, b->probes, b->k, dist, slot, tid, &count, &total);
		b->lat[j] = now() - t;
	}
	b->t1 = now();
	if (q)
		pgv_query_end(q);
	if (view)
		pgv_index_free(view);
	if (ctx)
		pgv_ctx_destroy(ctx);
	return NULL;
}

static int
cmp_double(const void *a, const void *b)
{
	double		x = *(const double *) a,
				y = *(const double *) b;

	return x < y ? -1 : (x > y ? 1 : 0);
}

/* returns PGV_OK or the first backend's error; out[0] = queries/s over all backends, out[1] = p50, out[2] = p90 latency (us) */
int
backends_run(pgv_index * index, int device, int nbackends, int per_thread, const void *queries, int nq,
			 size_t query_bytes, int probes, int k, double *out)
{
	backend    *b = calloc((size_t) nbackends, sizeof(backend));
	pthread_t  *th = calloc((size_t) nbackends, sizeof(pthread_t));
	double	   *lat = malloc(sizeof(double) * (size_t) nbackends * per_thread);
	pthread_barrier_t start;
	double		first = 1e300,
				last = 0;
	int			rc = PGV_OK;

	if (k > 64)
		k = 64;
	pthread_barrier_init(&start, NULL, (unsigned) nbackends);
	for (int i = 0; i < nbackends; i++)
	{
		b[i].index = index;
		b[i].device = device;
		b[i].id = i;
		b[i].per_thread = per_thread;
		b[i].queries = queries;
		b[i].nq = nq;
		b[i].query_bytes = query_bytes;
		b[i].probes = probes;
		b[i].k = k;
		b[i].start = &start;
		b[i].lat = lat + (size_t) i * per_thread;
		pthread_create(&th[i], NULL, backend_main, &b[i]);
	}
	for (int i = 0; i < nbackends; i++)
	{
		pthread_join(th[i], NULL);
		if (b[i].rc != PGV_OK && rc == PGV_OK)
			rc = b[i].rc;
		if (b[i].t0 < first)
			first = b[i].t0;
		if (b[i].t1 > last)
			last = b[i].t1;
	}
	pthread_barrier_destroy(&start);
	if (rc == PGV_OK)
	{
		size_t		n = (size_t) nbackends * per_thread;

		qsort(lat, n, sizeof(double), cmp_double);
		out[0] = (double) n / (last - first);
		out[1] = lat[n / 2] * 1e6;
		out[2] = lat[n * 9 / 10] * 1e6;
	}
	free(lat);
	free(th);
	free(b);
	return rc;
}


/* ---------------------------------------------------------------- the same clients behind the pooler (ivf_pool.c) */
typedef struct
{
	pgv_pool   *pool;
	int			id;
	int			per_thread;
	const char *queries;
	int			nq;
	size_t		query_bytes;
	int			k;
	pthread_barrier_t *start;
	double	   *lat;
	double		t0,
				t1;
	int			rc;
}			client;

static void *
client_main(void *arg)
{
	client	   *c = arg;
	uint64_t	tid[64];
	float		dist[64];

	for (int j = 0; j < 5 && c->rc == PGV_OK; j++)
		c->rc = pgv_host_pool_search(c->pool, c->queries + (size_t) ((c->id * 7 + j) % c->nq) * c->query_bytes, tid, dist);
	pthread_barrier_wait(c->start);
	c->t0 = now();
	for (int j = 0; j < c->per_thread && c->rc == PGV_OK; j++)
	{
		double		t = now();

		c->rc = pgv_host_pool_search(c->pool, c->queries + (size_t) ((c->id * 31 + j) % c->nq) * c->query_bytes, tid, dist);
		c->lat[j] = now() - t;
	}
	c->t1 = now();
	return NULL;
}

/* out[0] = queries/s over all clients, out[1] = p50, out[2] = p90 latency (us), out[3] = mean batch size */
int
pool_run(pgv_index * index, int device, int dtype, int dim, int nclients, int per_thread, const void *queries, int nq,
		 size_t query_bytes, int probes, int k, int max_batch, int max_wait_us, int lanes, double *out)
{
	pgv_pool   *pool = NULL;
	client	   *c;
	pthread_t  *th;
	double	   *lat;
	pthread_barrier_t start;
	pthread_attr_t attr;
	double		first = 1e300,
				last = 0;
	int64_t		batches0,
				queries0,
				batches1,
				queries1;
	int			rc;

	if (k > 64)
		k = 64;
	rc = pgv_host_pool_create(index, device, (pgv_dtype) dtype, dim, probes, k, max_batch, max_wait_us, lanes, &pool);
	if (rc != PGV_OK)
		return rc;
	c = calloc((size_t) nclients, sizeof(client));
	th = calloc((size_t) nclients, sizeof(pthread_t));
	lat = malloc(sizeof(double) * (size_t) nclients * per_thread);
	pthread_barrier_init(&start, NULL, (unsigned) nclients + 1);
	pthread_attr_init(&attr);
	pthread_attr_setstacksize(&attr, 256 * 1024);
	for (int i = 0; i < nclients; i++)
	{
		c[i].pool = pool;
		c[i].id = i;
		c[i].per_thread = per_thread;
		c[i].queries = queries;
		c[i].nq = nq;
		c[i].query_bytes = query_bytes;
		c[i].k = k;
		c[i].start = &start;
		c[i].lat = lat + (size_t) i * per_thread;
		pthread_create(&th[i], &attr, client_main, &c[i]);
	}
	pgv_host_pool_stats(pool, &batches0, &queries0);	/* (the warm-up rounds are still running: an estimate) */
	pthread_barrier_wait(&start);
	pgv_host_pool_stats(pool, &batches0, &queries0);
	for (int i = 0; i < nclients; i++)
	{
		pthread_join(th[i], NULL);
		if (c[i].rc != PGV_OK && rc == PGV_OK)
			rc = c[i].rc;
		if (c[i].t0 < first)
			first = c[i].t0;
		if (c[i].t1 > last)
			last = c[i].t1;
	}
	pgv_host_pool_stats(pool, &batches1, &queries1);
	pthread_attr_destroy(&attr);
	pthread_barrier_destroy(&start);
	if (rc == PGV_OK)
	{
		size_t		n = (size_t) nclients * per_thread;

		qsort(lat, n, sizeof(double), cmp_double);
		out[0] = (double) n / (last - first);
		out[1] = lat[n / 2] * 1e6;
		out[2] = lat[n * 9 / 10] * 1e6;
		out[3] = batches1 > batches0 ? (double) (queries1 - queries0) / (double) (batches1 - batches0) : 0.0;
	}
	pgv_host_pool_destroy(pool);
	free(lat);
	free(th);
	free(c);
	return rc;
}
