/*
 * oracle_bench.c -- how bench.py times the CPU restatement on a many-core, multi-socket host.
 *
 * TEST INFRASTRUCTURE ONLY (see pgv_oracle.h).  Nothing here restates the reference's arithmetic: these are the
 * thread runners around ora_ivf_search / ora_pages_search / ora_ivf_assign, i.e. "N backends" (one pinned thread
 * each, like N Postgres backends of src/ivfscan.c:252-296 on N cores) and "N parallel build workers"
 * (src/ivfbuild.c:830-966 splits the heap scan between workers; every worker runs the argmin loop of
 * src/ivfbuild.c:183-192 on its share).
 *
 * Placement matters for a baseline that streams 6-8 GB per few hundred queries: an array that one thread
 * allocated and filled sits on ONE NUMA node, and 64 threads reading it scale 5 x, not 64 x.  ora_bench_spread_copy
 * re-homes an array in 2 MB pieces, piece p touched first by thread p % nthreads, with the threads pinned round
 * the machine -- the placement `numactl --interleave=all` gives a Postgres shared_buffers.
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <sched.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>

#include "pgv_oracle.h"

#define SPREAD_PIECE ((size_t) 2 << 20)

static double
bench_now(void)
{
	struct timespec ts;

	clock_gettime(CLOCK_MONOTONIC, &ts);
	return (double) ts.tv_sec + 1e-9 * (double) ts.tv_nsec;
}

/* the CPUs this process may run on, in order */
static int
allowed_cpus(int *out, int cap)
{
	cpu_set_t	set;
	int			n = 0;

	if (sched_getaffinity(0, sizeof(set), &set) != 0)
		return 0;
	for (int c = 0; c < CPU_SETSIZE && n < cap; c++)
		if (CPU_ISSET(c, &set))
			out[n++] = c;
	return n;
}

int
ora_bench_cpus(void)
{
	int			cpus[CPU_SETSIZE];

	return allowed_cpus(cpus, CPU_SETSIZE);
}

/* thread t of nthreads runs on allowed CPU (t * ncpu / nthreads): spread over the whole machine */
static void
pin_thread(int t, int nthreads)
{
	int			cpus[CPU_SETSIZE];
	int			ncpu = allowed_cpus(cpus, CPU_SETSIZE);
	cpu_set_t	one;

	if (ncpu <= 0)
		return;
	CPU_ZERO(&one);
	CPU_SET(cpus[(int) ((long) t * ncpu / nthreads) % ncpu], &one);
	(void) pthread_setaffinity_np(pthread_self(), sizeof(one), &one);
}

void *
ora_bench_alloc(size_t bytes)
{
	void	   *p = mmap(NULL, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);

	if (p == MAP_FAILED)
		return NULL;
	(void) madvise(p, bytes, MADV_HUGEPAGE);
	return p;
}

void
ora_bench_free(void *p, size_t bytes)
{
	if (p)
		munmap(p, bytes);
}

typedef struct
{
	char	   *dst;
	const char *src;
	size_t		bytes;
	int			t,
				nthreads;
}			spread_job;

static void *
spread_main(void *arg)
{
	spread_job *j = arg;

	pin_thread(j->t, j->nthreads);
	for (size_t off = (size_t) j->t * SPREAD_PIECE; off < j->bytes; off += (size_t) j->nthreads * SPREAD_PIECE)
	{
		size_t		len = j->bytes - off < SPREAD_PIECE ? j->bytes - off : SPREAD_PIECE;

		memcpy(j->dst + off, j->src + off, len);	/* first touch: the piece lands on this thread's node */
	}
	return NULL;
}

/* dst = ora_bench_alloc(bytes), not touched yet */
int
ora_bench_spread_copy(void *dst, const void *src, size_t bytes, int nthreads)
{
	pthread_t  *th;
	spread_job *jobs;

	if (nthreads < 1)
		nthreads = 1;
	th = calloc((size_t) nthreads, sizeof(pthread_t));
	jobs = calloc((size_t) nthreads, sizeof(spread_job));
	if (!th || !jobs)
		return -1;
	for (int t = 0; t < nthreads; t++)
	{
		jobs[t] = (spread_job) {dst, src, bytes, t, nthreads};
		pthread_create(&th[t], NULL, spread_main, &jobs[t]);
	}
	for (int t = 0; t < nthreads; t++)
		pthread_join(th[t], NULL);
	free(jobs);
	free(th);
	return 0;
}

/* ---------------------------------------------------------------- N backends answering queries */
typedef struct
{
	const ora_ivf_index *ix;	/* or NULL: the page image */
	const uint8_t *pages;
	uint32_t	nblocks;
	int			ops,
				dtype;
	const char *queries;
	size_t		query_bytes;
	int			nq,
				probes,
				k;
	double		seconds;
	int			t,
				nthreads;
	uint64_t   *out_tids;		/* [nq x k] */
	double	   *out_dist;		/* [nq x k] */
	int		   *out_count;		/* [nq] */
	pthread_barrier_t *start;
	long		done;
	double		t0,
				t1;
}			search_job;

static void *
search_main(void *arg)
{
	search_job *j = arg;
	uint64_t   *tids = malloc(sizeof(uint64_t) * (size_t) j->k);
	double	   *dist = malloc(sizeof(double) * (size_t) j->k);
	double		t_end;

	pin_thread(j->t, j->nthreads);
	pthread_barrier_wait(j->start);
	j->t0 = bench_now();
	t_end = j->t0 + j->seconds;
	/* every query is answered at least once (the parity check reads the answers), then until the time is up */
	for (long i = j->t; i < j->nq || bench_now() < t_end; i += j->nthreads)
	{
		int			qi = (int) (i % j->nq);
		const void *q = j->queries + (size_t) qi * j->query_bytes;
		int			n;

		if (j->ix)
			n = ora_ivf_search(j->ix, q, j->probes, j->k, tids, dist);
		else
			n = ora_pages_search(j->pages, j->nblocks, j->ops, j->dtype, q, j->probes, j->k, tids, dist, NULL);
		if (i < j->nq)
		{
			j->out_count[qi] = n;
			memcpy(j->out_tids + (size_t) qi * j->k, tids, sizeof(uint64_t) * (size_t) (n > 0 ? n : 0));
			memcpy(j->out_dist + (size_t) qi * j->k, dist, sizeof(double) * (size_t) (n > 0 ? n : 0));
		}
		j->done++;
	}
	j->t1 = bench_now();
	free(dist);
	free(tids);
	return NULL;
}

/*
 * nthreads pinned threads, each one query at a time, for `seconds` (and at least once round the nq queries).
 * ix != NULL: ora_ivf_search over the list-major arrays; else ora_pages_search over the page image.
 * out_stats[0] = queries answered, [1] = wall seconds (first start to last finish).  Returns 0.
 */
int
ora_bench_search(const ora_ivf_index * ix, const uint8_t *pages, uint32_t nblocks, int ops, int dtype,
				 const void *queries, size_t query_bytes, int nq, int probes, int k, int nthreads, double seconds,
				 uint64_t *out_tids, double *out_dist, int *out_count, double *out_stats)
{
	pthread_t  *th;
	search_job *jobs;
	pthread_barrier_t start;
	double		first = 1e300,
				last = 0;
	long		total = 0;

	if (nthreads < 1 || nq < 1 || k < 1)
		return -1;
	th = calloc((size_t) nthreads, sizeof(pthread_t));
	jobs = calloc((size_t) nthreads, sizeof(search_job));
	if (!th || !jobs)
		return -1;
	pthread_barrier_init(&start, NULL, (unsigned) nthreads);
	for (int t = 0; t < nthreads; t++)
	{
		jobs[t] = (search_job) {ix, pages, nblocks, ops, dtype, queries, query_bytes, nq, probes, k, seconds, t, nthreads,
								out_tids, out_dist, out_count, &start, 0, 0, 0};
		pthread_create(&th[t], NULL, search_main, &jobs[t]);
	}
	for (int t = 0; t < nthreads; t++)
	{
		pthread_join(th[t], NULL);
		total += jobs[t].done;
		if (jobs[t].t0 < first)
			first = jobs[t].t0;
		if (jobs[t].t1 > last)
			last = jobs[t].t1;
	}
	pthread_barrier_destroy(&start);
	out_stats[0] = (double) total;
	out_stats[1] = last - first;
	free(jobs);
	free(th);
	return 0;
}

/* ---------------------------------------------------------------- N build workers assigning rows */
typedef struct
{
	int			ops,
				dtype,
				dim,
				k;
	const void *centers;
	const char *rows;
	size_t		row_bytes;
	int64_t		lo,
				hi;
	int32_t    *out_list;
	int			t,
				nthreads;
	pthread_barrier_t *start;
	double		t0,
				t1;
}			assign_job;

static void *
assign_main(void *arg)
{
	assign_job *j = arg;

	pin_thread(j->t, j->nthreads);
	pthread_barrier_wait(j->start);
	j->t0 = bench_now();
	if (j->hi > j->lo)
		ora_ivf_assign(j->ops, j->dtype, j->dim, j->centers, j->k, j->rows + (size_t) j->lo * j->row_bytes, j->hi - j->lo,
					   j->out_list + j->lo, NULL);
	j->t1 = bench_now();
	return NULL;
}

/* rows [0, n) split evenly between nthreads pinned workers; returns the wall seconds in *out_seconds */
int
ora_bench_assign(int ops, int dtype, int dim, const void *centers, int k, const void *rows, int64_t n, int nthreads,
				 int32_t *out_list, double *out_seconds)
{
	pthread_t  *th;
	assign_job *jobs;
	pthread_barrier_t start;
	size_t		row_bytes = (size_t) dim * (dtype == ORA_F32 ? 4 : 2);
	double		first = 1e300,
				last = 0;

	if (nthreads < 1)
		return -1;
	th = calloc((size_t) nthreads, sizeof(pthread_t));
	jobs = calloc((size_t) nthreads, sizeof(assign_job));
	if (!th || !jobs)
		return -1;
	pthread_barrier_init(&start, NULL, (unsigned) nthreads);
	for (int t = 0; t < nthreads; t++)
	{
		jobs[t] = (assign_job) {ops, dtype, dim, k, centers, rows, row_bytes, n * t / nthreads, n * (t + 1) / nthreads,
								out_list, t, nthreads, &start, 0, 0};
		pthread_create(&th[t], NULL, assign_main, &jobs[t]);
	}
	for (int t = 0; t < nthreads; t++)
	{
		pthread_join(th[t], NULL);
		if (jobs[t].t0 < first)
			first = jobs[t].t0;
		if (jobs[t].t1 > last)
			last = jobs[t].t1;
	}
	pthread_barrier_destroy(&start);
	*out_seconds = last - first;
	free(jobs);
	free(th);
	return 0;
}
