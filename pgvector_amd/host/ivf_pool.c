/*
 * ivf_pool.c -- a pooler in front of the batched IVFFlat scan.
 *
 * The reference answers one query per backend at a time (ivfflatgettuple, src/ivfscan.c:361-414); on the GPU that
 * shape streams each query's lists alone (pgv_query_*: ~20 k queries/s per backend, ~47 k/s for many backends on
 * their own streams), while the same queries taken together share every pass over a probed list
 * (pgv_search_batch: 750 k/s).  This is the piece between the two: backends (threads here; a background worker
 * fed through shared memory inside the server) hand in ONE query each and block; whatever arrives within
 * max_wait_us of the first query of a batch -- or until max_batch are waiting -- goes to the device as one
 * pgv_search_batch.  Several lanes (a context = stream + scratch, a pgv_index_share view and pinned buffers each)
 * take batches in turn, so the next batch collects and runs its planning while the previous one scans.
 *
 * The first backend to enter an idle lane leads its batch (no extra thread): it waits for the batch to close,
 * runs the scan, publishes the results and wakes the others.  Results are exactly pgv_search_batch's: the head
 * of GetScanItems + tuplesort for each query (src/ivfscan.c:123-187), heap TIDs and FUNCTION 1 distances.
 */
#define _GNU_SOURCE
#include "pgv_host.h"

#include <limits.h>
#include <linux/futex.h>
#include <pthread.h>
#include <sys/syscall.h>
#include <unistd.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

extern int	pgv_host_fail(int code, const char *fmt,...);

typedef struct
{
	pgv_ctx    *ctx;
	pgv_index  *view;
	char	   *queries;		/* pinned [max_batch x row_bytes] */
	uint64_t   *tids;			/* pinned [max_batch x k] */
	float	   *dist;			/* pinned [max_batch x k] */
	int			state;			/* LANE_* */
	int			count;			/* queries of the batch being collected / run */
	int			ready;			/* ... whose payload has been copied into the lane (atomic) */
	int			readers;		/* followers that still have to copy their answer */
	int			rc;
	uint32_t	gen,			/* batch number of this lane */
				done_gen;		/* last batch whose results are published (futex word) */
	uint32_t	fill;			/* bumped when the leader may stop waiting (futex word) */
}			lane;

enum
{
	LANE_FREE, LANE_COLLECTING, LANE_RUNNING
};

struct pgv_pool
{
	int			probes,
				k,
				max_batch,
				max_wait_us,
				nlanes;
	size_t		row_bytes;
	lane	   *lanes;
	int			collecting;		/* lane that takes arrivals, or -1 */
	int			arriving;		/* backends inside pgv_host_pool_search that have not joined a batch yet */
	pthread_mutex_t lock;		/* guards the few words below and the lanes' count / state: tens of nanoseconds.
								 * Adaptive (spins briefly, then sleeps): a pure spinlock collapses once there are more
								 * backends than cores (1024 threads: p90 latency 180 ms, measured) */
	uint32_t	free_epoch;		/* bumped when a lane comes free (futex word) */
	int64_t		batches,
				queries;
};

/* Hundreds of backends wait for one word (their batch's results, a lane coming free).  A condition variable
 * wakes them one futex call and one mutex hand-over at a time -- milliseconds per batch at 256 waiters (measured);
 * spinning starves the HIP runtime's own threads.  A bare futex: sleep until the word changes, wake all at once. */
static void
word_wait(uint32_t *word, uint32_t seen)
{
	syscall(SYS_futex, word, FUTEX_WAIT_PRIVATE, seen, NULL, NULL, 0);
}

static void
word_wait_us(uint32_t *word, uint32_t seen, long us)
{
	struct timespec rel = {us / 1000000L, (us % 1000000L) * 1000L};

	syscall(SYS_futex, word, FUTEX_WAIT_PRIVATE, seen, &rel, NULL, 0);
}

static void
word_wake_all(uint32_t *word)
{
	syscall(SYS_futex, word, FUTEX_WAKE_PRIVATE, INT_MAX, NULL, NULL, 0);
}

static int64_t
now_us(void)
{
	struct timespec ts;

	clock_gettime(CLOCK_MONOTONIC, &ts);
	return (int64_t) ts.tv_sec * 1000000 + ts.tv_nsec / 1000;
}

void
pgv_host_pool_destroy(pgv_pool * pool)
{
	if (!pool)
		return;
	for (int i = 0; i < pool->nlanes; i++)
	{
		lane	   *l = &pool->lanes[i];

		if (l->view)
			pgv_index_free(l->view);
		if (l->ctx)
			pgv_ctx_destroy(l->ctx);
		pgv_pinned_free(l->queries);
		pgv_pinned_free(l->tids);
		pgv_pinned_free(l->dist);
	}
	pthread_mutex_destroy(&pool->lock);
	free(pool->lanes);
	free(pool);
}

int
pgv_host_pool_create(pgv_index * index, int device, pgv_dtype dtype, int dim, int probes, int k, int max_batch,
					 int max_wait_us, int lanes, pgv_pool * *out)
{
	pgv_pool   *pool;

	if (!index || !out)
		return pgv_host_fail(PGV_ERR_ARG, "pgv_host_pool_create: index/out is NULL");
	*out = NULL;
	if (probes < 1 || k < 1 || max_batch < 1 || max_batch > 65536 || max_wait_us < 0 || lanes < 1 || lanes > 8 || dim < 1)
		return pgv_host_fail(PGV_ERR_ARG, "pgv_host_pool_create: bad probes / k / max_batch / max_wait_us / lanes");
	pool = calloc(1, sizeof(pgv_pool));
	if (!pool)
		return pgv_host_fail(PGV_ERR_NOMEM, "out of memory");
	pool->probes = probes;
	pool->k = k;
	pool->max_batch = max_batch;
	pool->max_wait_us = max_wait_us;
	pool->nlanes = lanes;
	pool->row_bytes = (size_t) dim * (dtype == PGV_F32 ? 4 : 2);
	pool->collecting = -1;
	pool->lanes = calloc((size_t) lanes, sizeof(lane));
	{
		pthread_mutexattr_t ma;

		pthread_mutexattr_init(&ma);
		pthread_mutexattr_settype(&ma, PTHREAD_MUTEX_ADAPTIVE_NP);
		pthread_mutex_init(&pool->lock, &ma);
		pthread_mutexattr_destroy(&ma);
	}
	if (!pool->lanes)
	{
		pgv_host_pool_destroy(pool);
		return pgv_host_fail(PGV_ERR_NOMEM, "out of memory");
	}
	for (int i = 0; i < lanes; i++)
	{
		lane	   *l = &pool->lanes[i];
		int			rc = pgv_ctx_create(device, NULL, &l->ctx);

		if (rc == PGV_OK)
			rc = pgv_index_share(index, l->ctx, &l->view);
		if (rc == PGV_OK)
			rc = pgv_pinned_alloc(pool->row_bytes * (size_t) max_batch, (void **) &l->queries);
		if (rc == PGV_OK)
			rc = pgv_pinned_alloc(sizeof(uint64_t) * (size_t) max_batch * k, (void **) &l->tids);
		if (rc == PGV_OK)
			rc = pgv_pinned_alloc(sizeof(float) * (size_t) max_batch * k, (void **) &l->dist);
		if (rc != PGV_OK)
		{
			pgv_host_pool_destroy(pool);
			return pgv_host_fail(rc, "%s", pgv_last_error());
		}
	}
	*out = pool;
	return PGV_OK;
}

/* One backend's query: blocks until its batch has been scanned.  out_tid / out_dist [k]: ascending, padded with
 * ~0 / +inf when the probed lists hold fewer than k tuples (exactly pgv_search_batch's row). */
int
pgv_host_pool_search(pgv_pool * pool, const void *query, uint64_t *out_tid, float *out_dist)
{
	lane	   *l;
	int			slot;
	uint32_t	gen;
	int			rc,
				kick;

	if (!pool || !query || !out_tid || !out_dist)
		return pgv_host_fail(PGV_ERR_ARG, "pgv_host_pool_search: pool/query/out is NULL");
	pthread_mutex_lock(&pool->lock);
	pool->arriving++;
	/* the batch that is collecting, or a new one in a free lane */
	for (;;)
	{
		uint32_t	epoch;

		if (pool->collecting >= 0)
			break;
		for (int i = 0; i < pool->nlanes; i++)
			if (pool->lanes[i].state == LANE_FREE)
			{
				pool->collecting = i;
				pool->lanes[i].state = LANE_COLLECTING;
				pool->lanes[i].count = 0;
				pool->lanes[i].ready = 0;
				pool->lanes[i].gen++;
				break;
			}
		if (pool->collecting >= 0)
			break;
		epoch = pool->free_epoch;
		pthread_mutex_unlock(&pool->lock);
		word_wait(&pool->free_epoch, epoch);
		pthread_mutex_lock(&pool->lock);
	}
	pool->arriving--;
	l = &pool->lanes[pool->collecting];
	slot = l->count++;
	gen = l->gen;
	if (l->count == pool->max_batch)
		pool->collecting = -1;	/* closed: the next arrival opens another lane */
	/* full, or the last of those who were queueing: the leader need not wait longer */
	kick = slot != 0 && (l->count == pool->max_batch || pool->arriving == 0);
	pthread_mutex_unlock(&pool->lock);

	/* the payload goes in outside the lock; the leader waits for `ready` to reach `count` */
	memcpy(l->queries + (size_t) slot * pool->row_bytes, query, pool->row_bytes);
	__atomic_add_fetch(&l->ready, 1, __ATOMIC_RELEASE);
	if (kick)
	{
		__atomic_add_fetch(&l->fill, 1, __ATOMIC_RELEASE);
		word_wake_all(&l->fill);
	}
	if (slot == 0)
	{
		/* the leader: everyone who queued while the lanes were busy joins (they are on their way: `arriving`),
		 * later arrivals get max_wait_us; then scan */
		int			n;
		int64_t		deadline = now_us() + pool->max_wait_us;

		for (;;)
		{
			int64_t		t = now_us();
			uint32_t	seen;

			pthread_mutex_lock(&pool->lock);
			if (l->count >= pool->max_batch || (pool->arriving == 0 && t >= deadline))
			{
				if (pool->collecting >= 0 && &pool->lanes[pool->collecting] == l)
					pool->collecting = -1;
				l->state = LANE_RUNNING;
				n = l->count;
				pool->batches++;
				pool->queries += n;
				pthread_mutex_unlock(&pool->lock);
				break;
			}
			seen = __atomic_load_n(&l->fill, __ATOMIC_ACQUIRE);
			pthread_mutex_unlock(&pool->lock);
			/* stragglers still on their way past the deadline: short naps, bounded by their own progress */
			word_wait_us(&l->fill, seen, t < deadline ? deadline - t : 20);
		}
		while (__atomic_load_n(&l->ready, __ATOMIC_ACQUIRE) < n)
			__builtin_ia32_pause();	/* a follower between its slot and the end of its 6 KB memcpy */

		rc = pgv_search_batch(l->view, l->queries, n, pool->probes, pool->k, l->dist, NULL, l->tids);

		/* publish: the followers sleep on done_gen */
		l->rc = rc;
		__atomic_store_n(&l->readers, n, __ATOMIC_RELAXED);
		__atomic_store_n(&l->done_gen, gen, __ATOMIC_RELEASE);
		if (n > 1)
			word_wake_all(&l->done_gen);
	}
	else
	{
		uint32_t	seen;

		while ((seen = __atomic_load_n(&l->done_gen, __ATOMIC_ACQUIRE)) != gen)
			word_wait(&l->done_gen, seen);
	}
	rc = l->rc;
	if (rc == PGV_OK)
	{
		memcpy(out_tid, l->tids + (size_t) slot * pool->k, sizeof(uint64_t) * (size_t) pool->k);
		memcpy(out_dist, l->dist + (size_t) slot * pool->k, sizeof(float) * (size_t) pool->k);
	}
	/* the lane is free again when its last reader has its answer */
	if (__atomic_sub_fetch(&l->readers, 1, __ATOMIC_ACQ_REL) == 0)
	{
		pthread_mutex_lock(&pool->lock);
		l->state = LANE_FREE;
		__atomic_add_fetch(&pool->free_epoch, 1, __ATOMIC_RELEASE);
		pthread_mutex_unlock(&pool->lock);
		word_wake_all(&pool->free_epoch);	/* everyone queued joins the batch the first of them opens */
	}
	if (rc != PGV_OK)
		return pgv_host_fail(rc, "batch failed: %s", slot == 0 ? pgv_last_error() : "see the leading backend's error");
	return PGV_OK;
}

void
pgv_host_pool_stats(pgv_pool * pool, int64_t *batches, int64_t *queries)
{
	pthread_mutex_lock(&pool->lock);
	if (batches)
		*batches = pool->batches;
	if (queries)
		*queries = pool->queries;
	pthread_mutex_unlock(&pool->lock);
}
