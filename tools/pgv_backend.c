/*
 * pgv_backend.c -- one PROCESS of a multi-backend run over the C ABI: what a Postgres backend, or one of the
 * extension's background workers, is to the device (src/ivfscan.c:252-296 runs in every backend process).
 * tools/backends_driver.c spawns these and gathers their results from shared memory; tests/ use it to prove that
 * processes -- not threads -- share ONE device mirror (pgv_index_export / pgv_index_import) and one pooler
 * (pgv_host_pool_* over a shared segment).
 *
 *   pgv_backend owner  <pool-shm> <data-shm> <device>   stage an index image out of <data-shm>, upload it, publish
 *                                                       its export handle in the pool segment, stay until shutdown
 *   pgv_backend serve  <pool-shm> <lane> <device>       import the mirror, lead one lane of the pooler
 *   pgv_backend client <pool-shm> <bank-shm> <id>       NO GPU context: one query at a time through the pooler
 *   pgv_backend query  <pool-shm> <bank-shm> <id>       import the mirror, one query at a time on its own stream
 *                                                       (pgv_query_rank + pgv_query_scan, what ivfflatgettuple issues)
 *
 * gcc -O2 -Iinclude -Ipgvector_amd/host tools/pgv_backend.c -o build/tools/pgv_backend -Lpgvector_amd/lib -lpgv_host -lpgv_hip
 */
#define _GNU_SOURCE
#include <fcntl.h>
#include <limits.h>
#include <linux/futex.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/resource.h>
#include <sys/syscall.h>
#include <time.h>
#include <unistd.h>

#include "pgv_hip.h"
#include "pgv_host.h"
#include "pgv_backend_shm.h"

static double
now(void)
{
	struct timespec ts;

	clock_gettime(CLOCK_MONOTONIC, &ts);
	return (double) ts.tv_sec + 1e-9 * (double) ts.tv_nsec;
}

static void *
map_shm(const char *name, size_t *bytes)
{
	int			fd = shm_open(name, O_RDWR, 0600);
	struct stat st;
	void	   *p;

	if (fd < 0 || fstat(fd, &st) != 0)
	{
		fprintf(stderr, "pgv_backend: cannot open shared memory %s\n", name);
		exit(10);
	}
	p = mmap(NULL, (size_t) st.st_size, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
	close(fd);
	if (p == MAP_FAILED)
	{
		fprintf(stderr, "pgv_backend: cannot map %s\n", name);
		exit(11);
	}
	*bytes = (size_t) st.st_size;
	return p;
}

static void
die(const char *what, int rc)
{
	fprintf(stderr, "pgv_backend: %s failed (%d): %s | %s\n", what, rc, pgv_last_error(), pgv_host_last_error());
	exit(20 + rc);
}

/* experiment (PGV_BACKEND_GATE = width): at most `width` own-context queries in flight over ALL processes */
static void
gate_enter(pgvb_bank * bank, int width)
{
	for (;;)
	{
		uint32_t	cur = __atomic_load_n(&bank->gate, __ATOMIC_RELAXED);

		if ((int) cur < width)
		{
			if (__atomic_compare_exchange_n(&bank->gate, &cur, cur + 1, 1, __ATOMIC_ACQUIRE, __ATOMIC_RELAXED))
				return;
			continue;
		}
		syscall(SYS_futex, &bank->gate, FUTEX_WAIT, cur, NULL, NULL, 0);
	}
}

static void
gate_leave(pgvb_bank * bank, int width)
{
	uint32_t	before = __atomic_fetch_sub(&bank->gate, 1, __ATOMIC_RELEASE);

	if ((int) before >= width)
		syscall(SYS_futex, &bank->gate, FUTEX_WAKE, 1, NULL, NULL, 0);
}

/* arrive at the start line, sleep until the driver fires */
static void
start_line(pgvb_bank * bank)
{
	__atomic_add_fetch(&bank->ready, 1, __ATOMIC_RELEASE);
	syscall(SYS_futex, &bank->ready, FUTEX_WAKE, INT_MAX, NULL, NULL, 0);
	while (__atomic_load_n(&bank->go, __ATOMIC_ACQUIRE) == 0)
	{
		struct timespec rel = {0, 50000000};

		syscall(SYS_futex, &bank->go, FUTEX_WAIT, 0, &rel, NULL, 0);
	}
}

static int
run_owner(const char *pool_name, const char *data_name, int device)
{
	size_t		pool_bytes,
				data_bytes;
	void	   *pool_shm = map_shm(pool_name, &pool_bytes);
	pgvb_image *img = map_shm(data_name, &data_bytes);
	char	   *base = (char *) img;
	pgv_pool   *pool;
	pgv_ctx    *ctx;
	pgv_index  *ix;
	pgv_index_handle h;
	int			rc;

	if (img->magic != PGVB_IMAGE_MAGIC)
		die("image magic", PGV_ERR_ARG);
	if ((rc = pgv_host_pool_attach(pool_shm, pool_bytes, &pool)) != PGV_OK)
		die("pgv_host_pool_attach", rc);
	if ((rc = pgv_ctx_create(device, NULL, &ctx)) != PGV_OK)
		die("pgv_ctx_create", rc);
	rc = pgv_index_upload(ctx, (pgv_metric) img->metric, (pgv_dtype) img->dtype, img->dim, img->nlists,
						  base + img->centers_off, (const int64_t *) (base + img->offsets_off), base + img->vectors_off,
						  img->tids_off ? (const uint64_t *) (base + img->tids_off) : NULL, &ix);
	if (rc != PGV_OK)
		die("pgv_index_upload", rc);
	if ((rc = pgv_index_export(ix, &h)) != PGV_OK)
		die("pgv_index_export", rc);
	if ((rc = pgv_host_pool_publish_index(pool, &h)) != PGV_OK)
		die("pgv_host_pool_publish_index", rc);
	/* the mirror lives as long as this process: stay until the pool is shut down */
	while (!pgv_host_pool_is_shut_down(pool))
		usleep(2000);
	pgv_index_free(ix);
	pgv_ctx_destroy(ctx);
	return 0;
}

static int
run_serve(const char *pool_name, int lane, int device)
{
	size_t		pool_bytes;
	void	   *pool_shm = map_shm(pool_name, &pool_bytes);
	pgv_pool   *pool;
	pgv_ctx    *ctx;
	pgv_index  *view;
	pgv_index_handle h;
	int			rc;

	if ((rc = pgv_host_pool_attach(pool_shm, pool_bytes, &pool)) != PGV_OK)
		die("pgv_host_pool_attach", rc);
	if ((rc = pgv_ctx_create(device, NULL, &ctx)) != PGV_OK)
		die("pgv_ctx_create", rc);
	if ((rc = pgv_host_pool_index_handle(pool, 120000, &h)) != PGV_OK)
		die("pgv_host_pool_index_handle", rc);
	if ((rc = pgv_index_import(ctx, &h, &view)) != PGV_OK)
		die("pgv_index_import", rc);
	rc = pgv_host_pool_serve(pool, lane, view);
	pgv_index_free(view);
	pgv_ctx_destroy(ctx);
	return rc == PGV_OK ? 0 : 20 + rc;
}

static int
run_client(const char *pool_name, const char *bank_name, int id, int independent)
{
	size_t		pool_bytes,
				bank_bytes;
	void	   *pool_shm = map_shm(pool_name, &pool_bytes);
	pgvb_bank  *bank = map_shm(bank_name, &bank_bytes);
	char	   *bbase = (char *) bank;
	pgvb_client *me = (pgvb_client *) (bbase + bank->clients_off) + id;
	struct rusage ru0 = {0};
	double	   *lat = (double *) (bbase + bank->lat_off) + (size_t) id * bank->per_client;
	uint64_t   *ans_t = bank->verify ? (uint64_t *) (bbase + bank->tid_off) + (size_t) id * bank->per_client * bank->k : NULL;
	float	   *ans_d = bank->verify ? (float *) (bbase + bank->dist_off) + (size_t) id * bank->per_client * bank->k : NULL;
	const char *queries = bbase + bank->queries_off;
	pgv_pool   *pool;
	pgv_ctx    *ctx = NULL;
	pgv_index  *view = NULL;
	pgv_query  *q = NULL;
	uint64_t	tid[PGVB_MAX_K];
	float		dist[PGVB_MAX_K];
	int64_t		slot[PGVB_MAX_K];
	int			k = bank->k,
				rc;
	int			gate_width = getenv("PGV_BACKEND_GATE") ? atoi(getenv("PGV_BACKEND_GATE")) : 0;

	if (bank->magic != PGVB_BANK_MAGIC || id < 0 || id >= bank->nclients || k > PGVB_MAX_K)
		die("bank", PGV_ERR_ARG);
	if ((rc = pgv_host_pool_attach(pool_shm, pool_bytes, &pool)) != PGV_OK)
		die("pgv_host_pool_attach", rc);
	if (independent)
	{
		pgv_index_handle h;

		if ((rc = pgv_ctx_create(bank->device, NULL, &ctx)) != PGV_OK)
			die("pgv_ctx_create", rc);
		if ((rc = pgv_host_pool_index_handle(pool, 120000, &h)) != PGV_OK)
			die("pgv_host_pool_index_handle", rc);
		if ((rc = pgv_index_import(ctx, &h, &view)) != PGV_OK)
			die("pgv_index_import", rc);
		if ((rc = pgv_query_begin(view, &q)) != PGV_OK)
			die("pgv_query_begin", rc);
	}
	rc = PGV_OK;
	for (int phase = 0; phase < 2 && rc == PGV_OK; phase++)
	{
		int			n = phase == 0 ? bank->warmup : bank->per_client;

		if (phase == 1)
		{
			start_line(bank);
			getrusage(RUSAGE_SELF, &ru0);
			me->t0 = now();
		}
		for (int j = 0; j < n && rc == PGV_OK; j++)
		{
			const void *query = queries + (size_t) ((id * 31 + j) % bank->nq) * bank->query_bytes;
			double		t = now();

			if (independent)
			{
				int			count;
				int64_t		total;

				if (gate_width > 0)
					gate_enter(bank, gate_width);
				rc = pgv_query_rank(q, query, bank->probes);
				if (rc == PGV_OK)
					rc = pgv_query_scan(q, 0, bank->probes, k, dist, slot, tid, &count, &total);
				if (gate_width > 0)
					gate_leave(bank, gate_width);
				for (int i = count; rc == PGV_OK && i < k; i++)
				{
					tid[i] = ~(uint64_t) 0;
					dist[i] = __builtin_inff();
				}
			}
			else
				rc = pgv_host_pool_search(pool, query, tid, dist);
			if (phase == 1)
			{
				lat[j] = now() - t;
				if (ans_t && rc == PGV_OK)
				{
					memcpy(ans_t + (size_t) j * k, tid, sizeof(uint64_t) * (size_t) k);
					memcpy(ans_d + (size_t) j * k, dist, sizeof(float) * (size_t) k);
				}
			}
		}
	}
	me->t1 = now();
	{
		struct rusage ru1;

		getrusage(RUSAGE_SELF, &ru1);
		me->utime = (ru1.ru_utime.tv_sec - ru0.ru_utime.tv_sec) + (ru1.ru_utime.tv_usec - ru0.ru_utime.tv_usec) * 1e-6;
		me->stime = (ru1.ru_stime.tv_sec - ru0.ru_stime.tv_sec) + (ru1.ru_stime.tv_usec - ru0.ru_stime.tv_usec) * 1e-6;
		me->nvcsw = ru1.ru_nvcsw - ru0.ru_nvcsw;
		me->nivcsw = ru1.ru_nivcsw - ru0.ru_nivcsw;
	}
	me->rc = rc;
	if (rc != PGV_OK)
		snprintf(me->err, sizeof(me->err), "%s | %s", pgv_last_error(), pgv_host_last_error());
	__atomic_add_fetch(&bank->finished, 1, __ATOMIC_RELEASE);
	if (q)
		pgv_query_end(q);
	if (view)
		pgv_index_free(view);
	if (ctx)
		pgv_ctx_destroy(ctx);
	return rc == PGV_OK ? 0 : 20 + rc;
}

int
main(int argc, char **argv)
{
	if (argc == 5 && strcmp(argv[1], "owner") == 0)
		return run_owner(argv[2], argv[3], atoi(argv[4]));
	if (argc == 5 && strcmp(argv[1], "serve") == 0)
		return run_serve(argv[2], atoi(argv[3]), atoi(argv[4]));
	if (argc == 5 && strcmp(argv[1], "client") == 0)
		return run_client(argv[2], argv[3], atoi(argv[4]), 0);
	if (argc == 5 && strcmp(argv[1], "query") == 0)
		return run_client(argv[2], argv[3], atoi(argv[4]), 1);
	fprintf(stderr, "usage: pgv_backend owner|serve|client|query ... (see the header of tools/pgv_backend.c)\n");
	return 2;
}
