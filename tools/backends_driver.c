/*
 * backends_driver.c -- N backends on one device mirror, in plain C over the C ABI (no Python in the timed
 * region): every thread is a backend with its own pgv_ctx (stream + scratch), a pgv_index_share view of the
 * uploaded index and a pgv_query handle, issuing one query at a time like ivfflatgettuple does
 * (pgv_query_rank + pgv_query_scan, src/ivfscan.c:361-414).  bench.py loads this as a shared object and hands it the
 * index it built.   gcc -O2 -shared -fPIC -pthread -Iinclude tools/backends_driver.c -o build/tools/libbackends.so
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
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
	volatile int phase;			/* where this backend is (PH_*): what a deadline reports */
	volatile int done;			/* queries finished */
	char		err[200];		/* pgv_last_error() of this thread, when rc != PGV_OK */
}			backend;

enum
{
	PH_NEW, PH_CTX, PH_SHARE, PH_BEGIN, PH_WARM_RANK, PH_WARM_SCAN, PH_START_LINE, PH_RANK, PH_SCAN, PH_CLEANUP, PH_OVER
};
static const char *const phase_name[] = {"not started", "pgv_ctx_create", "pgv_index_share", "pgv_query_begin", "warm-up pgv_query_rank",
	"warm-up pgv_query_scan", "start line", "pgv_query_rank", "pgv_query_scan", "cleanup", "over"};

/* join with a deadline (CLOCK_REALTIME seconds); 0 = joined */
static int
join_until(pthread_t th, double deadline_rt)
{
	struct timespec ts;

	ts.tv_sec = (time_t) deadline_rt;
	ts.tv_nsec = (long) ((deadline_rt - (double) ts.tv_sec) * 1e9);
	return pthread_timedjoin_np(th, NULL, &ts);
}

static double
now_rt(void)
{
	struct timespec ts;

	clock_gettime(CLOCK_REALTIME, &ts);
	return (double) ts.tv_sec + 1e-9 * (double) ts.tv_nsec;
}

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

	b->phase = PH_CTX;
	b->rc = pgv_ctx_create(b->device, NULL, &ctx);
	b->phase = PH_SHARE;
	if (b->rc == PGV_OK)
		b->rc = pgv_index_share(b->index, ctx, &view);
	b->phase = PH_BEGIN;
	if (b->rc == PGV_OK)
		b->rc = pgv_query_begin(view, &q);
	for (int j = 0; j < 20 && b->rc == PGV_OK; j++)
	{
		b->phase = PH_WARM_RANK;
		b->rc = pgv_query_rank(q, b->queries + (size_t) ((b->id * 7 + j) % b->nq) * b->query_bytes, b->probes);
		b->phase = PH_WARM_SCAN;
		if (b->rc == PGV_OK)
			b->rc = pgv_query_scan(q, 0, b->probes, b->k, dist, slot, tid, &count, &total);
	}
	b->phase = PH_START_LINE;
	pthread_barrier_wait(b->start);
	b->t0 = now();
	for (int j = 0; j < b->per_thread && b->rc == PGV_OK; j++)
	{
		double		t = now();

		b->phase = PH_RANK;
		b->rc = pgv_query_rank(q, b->queries + (size_t) ((b->id * 31 + j) % b->nq) * b->query_bytes, b->probes);
		b->phase = PH_SCAN;
		if (b->rc == PGV_OK)
			b->rc = pgv_query_scan(q, 0, b->probes, b->k, dist, slot, tid, &count, &total);
		b->lat[j] = now() - t;
		b->done = j + 1;
	}
	b->t1 = now();
	if (b->rc != PGV_OK)
		snprintf(b->err, sizeof(b->err), "%s", pgv_last_error());
	b->phase = PH_CLEANUP;
	if (q)
		pgv_query_end(q);
	if (view)
		pgv_index_free(view);
	if (ctx)
		pgv_ctx_destroy(ctx);
	b->phase = PH_OVER;
	return NULL;
}


/* PGVB_CPU_STAT=1: the cgroup's CPU use and throttling over the TIMED phase only (the clients' start-up -- a thousand
 * execs that each load the HIP runtime -- is not part of it), to stderr */
static void
read_cpu_stat(long long *usage, long long *nthr, long long *thr_usec)
{
	FILE	   *f = fopen("/sys/fs/cgroup/cpu.stat", "r");
	char		key[64];
	long long	v;

	*usage = *nthr = *thr_usec = 0;
	if (!f)
		return;
	while (fscanf(f, "%63s %lld", key, &v) == 2)
	{
		if (strcmp(key, "usage_usec") == 0)
			*usage = v;
		else if (strcmp(key, "nr_throttled") == 0)
			*nthr = v;
		else if (strcmp(key, "throttled_usec") == 0)
			*thr_usec = v;
	}
	fclose(f);
}
static int
cmp_double(const void *a, const void *b)
{
	double		x = *(const double *) a,
				y = *(const double *) b;

	return x < y ? -1 : (x > y ? 1 : 0);
}

/* returns PGV_OK or the first backend's error; out[0] = queries/s over all backends, out[1] = p50, out[2] = p90 latency (us).
 * A backend that has not come back `deadline_s` after the start is reported in errbuf with the call it sits in
 * (PGV_ERR_STATE; its thread and arrays are left behind -- the caller is a measurement process that ends soon). */
int
backends_run(pgv_index * index, int device, int nbackends, int per_thread, const void *queries, int nq,
			 size_t query_bytes, int probes, int k, double deadline_s, double *out, char *errbuf, size_t errcap)
{
	backend    *b = calloc((size_t) nbackends, sizeof(backend));
	pthread_t  *th = calloc((size_t) nbackends, sizeof(pthread_t));
	double	   *lat = malloc(sizeof(double) * (size_t) nbackends * per_thread);
	pthread_barrier_t *start = malloc(sizeof(pthread_barrier_t));
	double		first = 1e300,
				last = 0,
				deadline = now_rt() + (deadline_s > 0 ? deadline_s : 60.0);
	int			rc = PGV_OK;

	if (errbuf && errcap)
		errbuf[0] = 0;
	if (k > 64)
		k = 64;
	pthread_barrier_init(start, NULL, (unsigned) nbackends);
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
		b[i].start = start;
		b[i].lat = lat + (size_t) i * per_thread;
		pthread_create(&th[i], NULL, backend_main, &b[i]);
	}
	for (int i = 0; i < nbackends; i++)
	{
		if (join_until(th[i], deadline) != 0)
		{
			/* somebody is stuck: say who and where, and leave everything they may still touch alone */
			size_t		at = 0;

			if (errbuf && errcap)
			{
				at += (size_t) snprintf(errbuf + at, errcap - at, "%d backends, %.0f s: not back:", nbackends, deadline_s);
				for (int j = i; j < nbackends && at + 64 < errcap; j++)
					if (b[j].phase != PH_OVER)
						at += (size_t) snprintf(errbuf + at, errcap - at, " [#%d in %s after %d queries]", j, phase_name[b[j].phase], b[j].done);
			}
			return PGV_ERR_STATE;
		}
		if (b[i].rc != PGV_OK && rc == PGV_OK)
		{
			rc = b[i].rc;
			if (errbuf && errcap)
				snprintf(errbuf, errcap, "backend %d: %s", i, b[i].err);
		}
		if (b[i].t0 < first)
			first = b[i].t0;
		if (b[i].t1 > last)
			last = b[i].t1;
	}
	pthread_barrier_destroy(start);
	free(start);
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
	volatile int done;			/* -1: warming up; else queries finished */
	volatile int over;
	char		err[200];
}			client;

static void *
client_main(void *arg)
{
	client	   *c = arg;
	uint64_t	tid[64];
	float		dist[64];

	c->done = -1;
	for (int j = 0; j < 5 && c->rc == PGV_OK; j++)
		c->rc = pgv_host_pool_search(c->pool, c->queries + (size_t) ((c->id * 7 + j) % c->nq) * c->query_bytes, tid, dist);
	pthread_barrier_wait(c->start);
	c->done = 0;
	c->t0 = now();
	for (int j = 0; j < c->per_thread && c->rc == PGV_OK; j++)
	{
		double		t = now();

		c->rc = pgv_host_pool_search(c->pool, c->queries + (size_t) ((c->id * 31 + j) % c->nq) * c->query_bytes, tid, dist);
		c->lat[j] = now() - t;
		c->done = j + 1;
	}
	c->t1 = now();
	if (c->rc != PGV_OK)
		snprintf(c->err, sizeof(c->err), "%s", pgv_host_last_error());
	c->over = 1;
	return NULL;
}

/* out[0] = queries/s over all clients, out[1] = p50, out[2] = p90 latency (us), out[3] = mean batch size.  Clients not back
 * `deadline_s` after the start are reported in errbuf (PGV_ERR_STATE; the pool and its threads are then left behind). */
int
pool_run(pgv_index * index, int device, int dtype, int dim, int nclients, int per_thread, const void *queries, int nq,
		 size_t query_bytes, int probes, int k, int max_batch, int max_wait_us, int lanes, double deadline_s, double *out,
		 char *errbuf, size_t errcap)
{
	pgv_pool   *pool = NULL;
	client	   *c;
	pthread_t  *th;
	double	   *lat;
	pthread_barrier_t *start;
	pthread_attr_t attr;
	double		first = 1e300,
				last = 0,
				deadline = now_rt() + (deadline_s > 0 ? deadline_s : 60.0);
	int64_t		batches0,
				queries0,
				batches1,
				queries1;
	int			rc;

	if (errbuf && errcap)
		errbuf[0] = 0;
	if (k > 64)
		k = 64;
	rc = pgv_host_pool_create(index, device, (pgv_dtype) dtype, dim, probes, k, max_batch, max_wait_us, lanes, &pool);
	if (rc != PGV_OK)
	{
		if (errbuf && errcap)
			snprintf(errbuf, errcap, "pgv_host_pool_create: %s", pgv_host_last_error());
		return rc;
	}
	c = calloc((size_t) nclients, sizeof(client));
	th = calloc((size_t) nclients, sizeof(pthread_t));
	lat = malloc(sizeof(double) * (size_t) nclients * per_thread);
	start = malloc(sizeof(pthread_barrier_t));
	pthread_barrier_init(start, NULL, (unsigned) nclients);
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
		c[i].start = start;
		c[i].lat = lat + (size_t) i * per_thread;
		pthread_create(&th[i], &attr, client_main, &c[i]);
	}
	pthread_attr_destroy(&attr);
	/* (the warm-up rounds may still be running: the batch statistics are an estimate at the front edge) */
	pgv_host_pool_stats(pool, &batches0, &queries0);
	for (int i = 0; i < nclients; i++)
	{
		if (join_until(th[i], deadline) != 0)
		{
			size_t		at = 0;
			int			stuck = 0;

			for (int j = i; j < nclients; j++)
				stuck += !c[j].over;
			if (errbuf && errcap)
			{
				at += (size_t) snprintf(errbuf + at, errcap - at, "%d pooled clients, %.0f s: %d not back:", nclients, deadline_s, stuck);
				for (int j = i, shown = 0; j < nclients && shown < 8 && at + 64 < errcap; j++)
					if (!c[j].over)
					{
						at += (size_t) snprintf(errbuf + at, errcap - at, " [#%d %s %d]", j, c[j].done < 0 ? "warming up" : "after queries:", c[j].done);
						shown++;
					}
			}
			pgv_host_pool_shutdown(pool);	/* wakes every sleeper; whoever is stuck elsewhere stays */
			return PGV_ERR_STATE;
		}
		if (c[i].rc != PGV_OK && rc == PGV_OK)
		{
			rc = c[i].rc;
			if (errbuf && errcap)
				snprintf(errbuf, errcap, "client %d: %s", i, c[i].err);
		}
		if (c[i].t0 < first)
			first = c[i].t0;
		if (c[i].t1 > last)
			last = c[i].t1;
	}
	pgv_host_pool_stats(pool, &batches1, &queries1);
	pthread_barrier_destroy(start);
	free(start);
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


/* ============================================================================================================
 * The same two experiments with backends that are PROCESSES (tools/pgv_backend.c), which is what a Postgres
 * backend is (src/ivfscan.c:252-296 runs in each): ONE device mirror, exported by its owner and imported by every
 * other process (pgv_index_export / pgv_index_import); ONE pooler whose state lives in a shared segment.
 *
 *   mode 0  every client process imports the mirror and issues pgv_query_rank + pgv_query_scan on its own context
 *   mode 1  client processes have NO GPU context: pgv_host_pool_search over the shared segment; the lanes are led
 *           by `lanes` server processes that import the mirror (server_processes != 0) or by threads of the
 *           calling process (pgv_index_share views)
 *   owner   who uploads and exports the mirror: the calling process (`index`), or -- when `image_shm` names a
 *           pgvb_image segment -- a separate owner process (then every participant runs the same HIP runtime)
 *
 * chaos (tests): bit 0 = SIGKILL client 0 20 ms into the run, bit 1 = SIGKILL the first lane server 20 ms into the run.
 * Nobody may hang: the other clients finish, or -- those whose batch sat in the dead server's lane -- fail with
 * PGV_ERR_STATE; out[6] = clients that answered every query, out[7] = clients that failed or were killed.
 *
 * out[0] queries/s, [1] p50 us, [2] p90 us, [3] mean batch (mode 1), [4] bytes of HBM that went away between "before
 * any child" and "all children at the start line" (contexts + scratch of every process, the mirror NOT among them
 * unless a separate owner uploaded it), [5] processes spawned.  ans_tid / ans_dist [nclients x per_client x k] or NULL.
 */
#include <errno.h>
#include <fcntl.h>
#include <limits.h>
#include <linux/futex.h>
#include <signal.h>
#include <sched.h>
#include <spawn.h>
#include <stdio.h>
#include <sys/mman.h>
#include <sys/syscall.h>
#include <sys/wait.h>
#include <unistd.h>

#include "pgv_backend_shm.h"

extern char **environ;

static void *
make_shm(const char *name, size_t bytes)
{
	int			fd;
	void	   *p;

	shm_unlink(name);
	fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
	if (fd < 0)
		return NULL;
	if (ftruncate(fd, (off_t) bytes) != 0)
	{
		close(fd);
		shm_unlink(name);
		return NULL;
	}
	p = mmap(NULL, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
	close(fd);
	return p == MAP_FAILED ? NULL : p;
}

/*
 * Where the backend processes run.  A container's CPU QUOTA (cgroup cpu.max: 16 CPUs' worth on the GPU boxes) without a
 * cpuset lets a few hundred processes spread over all 256 hardware threads of the host, and then CFS bandwidth control is
 * the bottleneck: every wake-up on another CPU has to fetch a runtime slice from the group's one pool under one spinlock
 * (256 pooled clients: 107-196 us of SYSTEM time per query in the clients' own getrusage, 21-34 throttled periods in a 2 s
 * run; profiles/r06/pool_baton_wake.md).  The harness therefore gives the backends the cpuset the quota implies: as many CPUs
 * (the first ones of the allowed mask) as the quota pays for.  PGVB_PIN=0 leaves them unpinned.
 */
static int
pinned_cpus(cpu_set_t *set)
{
	cpu_set_t	allowed;
	long		want = 0;
	FILE	   *f;
	int			got = 0;

	CPU_ZERO(set);
	if (getenv("PGVB_PIN") && atoi(getenv("PGVB_PIN")) == 0)
		return 0;
	f = fopen("/sys/fs/cgroup/cpu.max", "r");
	if (f)
	{
		char		first[32];
		long long	period = 0;

		if (fscanf(f, "%31s %lld", first, &period) == 2 && strcmp(first, "max") != 0 && period > 0)
			want = (long) ((atoll(first) + period - 1) / period);
		fclose(f);
	}
	if (want < 1 || sched_getaffinity(0, sizeof(allowed), &allowed) != 0 || CPU_COUNT(&allowed) <= want)
		return 0;				/* no quota, or not more CPUs than it pays for: nothing to confine */
	for (int c = 0; c < CPU_SETSIZE && got < want; c++)
		if (CPU_ISSET(c, &allowed))
		{
			CPU_SET(c, set);
			got++;
		}
	return got;
}

static pid_t
spawn(const char *exe, const char *a1, const char *a2, const char *a3, const char *a4)
{
	char	   *argv[] = {(char *) exe, (char *) a1, (char *) a2, (char *) a3, (char *) a4, NULL};
	pid_t		pid = -1;
	cpu_set_t	pin,
				mine;
	int			npin = pinned_cpus(&pin);
	int			rc;

	/* the child inherits the mask of the thread that spawns it */
	if (npin > 0 && (sched_getaffinity(0, sizeof(mine), &mine) != 0 || sched_setaffinity(0, sizeof(pin), &pin) != 0))
		npin = 0;
	rc = posix_spawn(&pid, exe, NULL, NULL, argv, environ);
	if (npin > 0)
		sched_setaffinity(0, sizeof(mine), &mine);
	return rc != 0 ? -1 : pid;
}

int
backends_run_processes(pgv_index * index, const char *image_shm, int device, int mode, int nclients, int per_client,
					   int warmup, const void *queries, int nq, size_t query_bytes, int dtype, int dim, int probes, int k,
					   int max_batch, int max_wait_us, int lanes, int server_processes, const char *exe, int verify,
					   double deadline_s, int chaos, uint64_t *ans_tid, float *ans_dist, double *out, char *errbuf, size_t errcap)
{
	char		pool_name[64],
				bank_name[64],
				num[16],
				devs[16];
	size_t		pool_bytes,
				bank_bytes,
				at;
	void	   *pool_shm = NULL;
	pgvb_bank  *bank = NULL;
	pgv_pool   *pool = NULL;
	pid_t	   *pids = NULL;
	int			npids = 0,
				rc = PGV_OK;
	uint64_t	free0 = 0,
				free1 = 0,
				total = 0;
	long long	cpu0[3] = {0, 0, 0};
	double		t_go = 0;
	int64_t		batches0 = 0,
				queries0 = 0,
				batches1 = 0,
				queries1 = 0;
	int			completed = 0,
				failed = 0;

	if (errbuf && errcap)
		errbuf[0] = 0;
#define FAILP(code, ...) do { rc = (code); if (errbuf) snprintf(errbuf, errcap, __VA_ARGS__); goto out; } while (0)
	if (k > PGVB_MAX_K || nclients < 1 || per_client < 1 || lanes < 1 || lanes > 8)
		FAILP(PGV_ERR_ARG, "bad k / nclients / per_client / lanes");
	if (!getenv("HSA_ENABLE_IPC_MODE_LEGACY"))
		setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 1);	/* dmabuf IPC: the children inherit it */
	snprintf(pool_name, sizeof(pool_name), "/pgv_pool_%d", (int) getpid());
	snprintf(bank_name, sizeof(bank_name), "/pgv_bank_%d", (int) getpid());
	snprintf(devs, sizeof(devs), "%d", device);
	pgv_device_memory(device, &free0, &total);

	/* the pool segment: the handle registry in both modes, the pooler's lanes in mode 1 */
	pool_bytes = pgv_host_pool_shm_bytes((pgv_dtype) dtype, dim, k, max_batch, lanes);
	pool_shm = make_shm(pool_name, pool_bytes);
	if (!pool_shm)
		FAILP(PGV_ERR_NOMEM, "shm_open(%s, %zu bytes) failed: %s", pool_name, pool_bytes, strerror(errno));
	rc = pgv_host_pool_shm_init(pool_shm, pool_bytes, (pgv_dtype) dtype, dim, probes, k, max_batch, max_wait_us, lanes);
	if (rc == PGV_OK)
		rc = pgv_host_pool_attach(pool_shm, pool_bytes, &pool);
	if (rc != PGV_OK)
		FAILP(rc, "pool segment: %s", pgv_host_last_error());

	/* the bank: queries in, latencies / answers out */
	at = (sizeof(pgvb_bank) + 63) & ~(size_t) 63;
	bank_bytes = at + sizeof(pgvb_client) * (size_t) nclients + sizeof(double) * (size_t) nclients * per_client +
		(verify ? (sizeof(uint64_t) + sizeof(float)) * (size_t) nclients * per_client * k : 0) + query_bytes * (size_t) nq + 256;
	bank = make_shm(bank_name, bank_bytes);
	if (!bank)
		FAILP(PGV_ERR_NOMEM, "shm_open(%s, %zu bytes) failed: %s", bank_name, bank_bytes, strerror(errno));
	memset(bank, 0, at);
	bank->nq = nq;
	bank->probes = probes;
	bank->k = k;
	bank->per_client = per_client;
	bank->warmup = warmup;
	bank->nclients = nclients;
	bank->device = device;
	bank->verify = verify;
	bank->query_bytes = query_bytes;
	bank->clients_off = at;
	at += sizeof(pgvb_client) * (size_t) nclients;
	bank->lat_off = at;
	at += sizeof(double) * (size_t) nclients * per_client;
	if (verify)
	{
		bank->tid_off = at;
		at += sizeof(uint64_t) * (size_t) nclients * per_client * k;
		bank->dist_off = at;
		at += sizeof(float) * (size_t) nclients * per_client * k;
	}
	at = (at + 63) & ~(size_t) 63;
	bank->queries_off = at;
	memcpy((char *) bank + at, queries, query_bytes * (size_t) nq);
	memset((char *) bank + bank->clients_off, 0, sizeof(pgvb_client) * (size_t) nclients);
	bank->magic = PGVB_BANK_MAGIC;

	pids = calloc((size_t) nclients + 16, sizeof(pid_t));
	/* who owns the mirror */
	if (image_shm)
	{
		if ((pids[npids] = spawn(exe, "owner", pool_name, image_shm, devs)) < 0)
			FAILP(PGV_ERR_STATE, "cannot spawn %s owner", exe);
		npids++;
	}
	else
	{
		pgv_index_handle h;

		if ((rc = pgv_index_export(index, &h)) != PGV_OK)
			FAILP(rc, "pgv_index_export: %s", pgv_last_error());
		pgv_host_pool_publish_index(pool, &h);
	}
	/* who leads the lanes (mode 1) */
	if (mode == 1 && server_processes)
	{
		for (int l = 0; l < lanes; l++)
		{
			snprintf(num, sizeof(num), "%d", l);
			if ((pids[npids] = spawn(exe, "serve", pool_name, num, devs)) < 0)
				FAILP(PGV_ERR_STATE, "cannot spawn %s serve", exe);
			npids++;
		}
	}
	else if (mode == 1)
	{
		/* threads of this process over the SAME shared segment (pgv_index_share views): only possible when this
		 * process owns the mirror */
		if (image_shm)
			FAILP(PGV_ERR_ARG, "lane threads need the mirror in this process");
		rc = pgv_host_pool_start_threads(pool, index, device);
		if (rc != PGV_OK)
			FAILP(rc, "lane threads: %s", pgv_host_last_error());
	}
	for (int c = 0; c < nclients; c++)
	{
		snprintf(num, sizeof(num), "%d", c);
		if ((pids[npids] = spawn(exe, mode == 1 ? "client" : "query", pool_name, bank_name, num)) < 0)
			FAILP(PGV_ERR_STATE, "cannot spawn %s client %d", exe, c);
		npids++;
	}
	/* everybody warmed up and at the start line (or somebody died on the way) */
	{
		double		deadline = now() + (deadline_s > 0 ? deadline_s : 90.0);

		while (__atomic_load_n(&bank->ready, __ATOMIC_ACQUIRE) < (uint32_t) nclients)
		{
			struct timespec rel = {0, 20000000};

			for (int i = 0; i < npids; i++)
			{
				int			st;

				if (pids[i] > 0 && waitpid(pids[i], &st, WNOHANG) == pids[i])
				{
					pids[i] = 0;
					FAILP(PGV_ERR_STATE, "child %d exited before the start (status %d, exit code %d)", i, st,
						  WIFEXITED(st) ? WEXITSTATUS(st) : -1);
				}
			}
			if (now() > deadline)
				FAILP(PGV_ERR_STATE, "%u of %d clients at the start line after %.0f s", __atomic_load_n(&bank->ready, __ATOMIC_ACQUIRE),
					  nclients, deadline_s > 0 ? deadline_s : 90.0);
			syscall(SYS_futex, &bank->ready, FUTEX_WAIT, __atomic_load_n(&bank->ready, __ATOMIC_ACQUIRE), &rel, NULL, 0);
		}
	}
	pgv_device_memory(device, &free1, &total);
	pgv_host_pool_stats(pool, &batches0, &queries0);
	read_cpu_stat(&cpu0[0], &cpu0[1], &cpu0[2]);
	t_go = now();
	__atomic_store_n(&bank->go, 1, __ATOMIC_RELEASE);
	syscall(SYS_futex, &bank->go, FUTEX_WAKE, INT_MAX, NULL, NULL, 0);
	if (chaos)
	{
		usleep(20000);
		if (chaos & 1)
			kill(pids[npids - nclients], SIGKILL);	/* a backend dies in mid-query */
		if ((chaos & 2) && mode == 1 && server_processes)
			kill(pids[image_shm ? 1 : 0], SIGKILL);	/* the leader of lane 0 dies under its batch */
	}
	/* the clients are the last nclients pids; a run that does not end in 90 s is given up (the children are killed
	 * below) -- a measurement harness must not be able to hang its caller */
	{
		double		deadline = now() + (deadline_s > 0 ? deadline_s : 90.0);
		int			left = nclients;

		completed = failed = 0;

		while (left > 0)
		{
			left = 0;
			for (int i = npids - nclients; i < npids; i++)
			{
				int			st = 0;

				if (pids[i] <= 0)
					continue;
				if (waitpid(pids[i], &st, WNOHANG) == pids[i])
				{
					pids[i] = 0;
					if (WIFEXITED(st) && WEXITSTATUS(st) == 0)
						completed++;
					else
						failed++;
					/* under chaos a killed client and clients that were told PGV_ERR_STATE are what is expected */
					if (chaos && ((WIFSIGNALED(st) && i == npids - nclients && (chaos & 1)) ||
								  (WIFEXITED(st) && WEXITSTATUS(st) == 20 + PGV_ERR_STATE && (chaos & 2))))
						continue;
					if ((!WIFEXITED(st) || WEXITSTATUS(st) != 0) && rc == PGV_OK)
					{
						pgvb_client *cl = (pgvb_client *) ((char *) bank + bank->clients_off) + (i - (npids - nclients));

						rc = cl->rc ? cl->rc : PGV_ERR_STATE;
						if (errbuf)
							snprintf(errbuf, errcap, "client %d: status %d: %s", i - (npids - nclients), st, cl->err);
					}
				}
				else
					left++;
			}
			if (left > 0 && now() > deadline)
				FAILP(PGV_ERR_STATE, "%d of %d clients had not finished after %.0f s", left, nclients, deadline_s > 0 ? deadline_s : 90.0);
			if (left > 0)
				usleep(2000);
		}
	}
	pgv_host_pool_stats(pool, &batches1, &queries1);
	if (getenv("PGVB_CPU_STAT"))
	{
		long long	cpu1[3];

		read_cpu_stat(&cpu1[0], &cpu1[1], &cpu1[2]);
		fprintf(stderr, "timed phase: %.3f s wall, cgroup CPU %.3f s (%.1f us per query), throttled %lld times for %.3f s\n", now() - t_go,
				(cpu1[0] - cpu0[0]) / 1e6, (double) (cpu1[0] - cpu0[0]) / ((double) nclients * per_client), cpu1[1] - cpu0[1],
				(cpu1[2] - cpu0[2]) / 1e6);
		{
			pgvb_client *cl = (pgvb_client *) ((char *) bank + bank->clients_off);
			double		ut = 0,
						st = 0,
						v = 0,
						iv = 0,
						nq = (double) nclients * per_client;

			for (int c = 0; c < nclients; c++)
			{
				ut += cl[c].utime;
				st += cl[c].stime;
				v += (double) cl[c].nvcsw;
				iv += (double) cl[c].nivcsw;
			}
			fprintf(stderr, "clients' own getrusage over their timed loops: user %.1f us, system %.1f us, %.2f voluntary and %.2f involuntary "
					"context switches per query\n", ut / nq * 1e6, st / nq * 1e6, v / nq, iv / nq);
		}
	}
	if (rc == PGV_OK)
	{
		pgvb_client *cl = (pgvb_client *) ((char *) bank + bank->clients_off);
		double	   *lat = (double *) ((char *) bank + bank->lat_off);
		size_t		n = (size_t) nclients * per_client;
		double		first = 1e300,
					last = 0;

		for (int c = 0; c < nclients; c++)
		{
			if (cl[c].t0 < first)
				first = cl[c].t0;
			if (cl[c].t1 > last)
				last = cl[c].t1;
		}
		if (verify && ans_tid && ans_dist)
		{
			memcpy(ans_tid, (char *) bank + bank->tid_off, sizeof(uint64_t) * n * k);
			memcpy(ans_dist, (char *) bank + bank->dist_off, sizeof(float) * n * k);
		}
		qsort(lat, n, sizeof(double), cmp_double);
		out[0] = (double) n / (last - first);
		out[1] = lat[n / 2] * 1e6;
		out[2] = lat[n * 9 / 10] * 1e6;
		out[3] = batches1 > batches0 ? (double) (queries1 - queries0) / (double) (batches1 - batches0) : 0.0;
		out[4] = (double) free0 - (double) free1;
		out[5] = (double) npids;
		out[6] = (double) completed;
		out[7] = (double) failed;
	}
out:
	if (pool)
		pgv_host_pool_shutdown(pool);
	{
		/* servers and owner leave on the pool's shutdown; 10 s for all of them together after a clean run, none
		 * after a failed one */
		double		deadline = now() + (rc == PGV_OK ? 10.0 : 0.0);

		for (int i = 0; pids && i < npids; i++)
			if (pids[i] > 0)
			{
				int			st;

				while (waitpid(pids[i], &st, WNOHANG) == 0)
				{
					if (now() > deadline)
					{
						/* a killed child is reaped when it goes -- but one that sits in an uninterruptible driver call
						 * does not go, and this harness does not wait for it (two seconds, then init inherits it) */
						double		give_up = now() + 2.0;

						kill(pids[i], SIGKILL);
						while (waitpid(pids[i], &st, WNOHANG) == 0 && now() < give_up)
							usleep(2000);
						break;
					}
					usleep(2000);
				}
			}
	}
	free(pids);
	if (pool)
		pgv_host_pool_detach(pool);
	if (pool_shm)
		munmap(pool_shm, pool_bytes);
	if (bank)
		munmap(bank, bank_bytes);
	shm_unlink(pool_name);
	shm_unlink(bank_name);
	return rc;
#undef FAILP
}
