/*
 * ivf_pool.c -- a pooler in front of the batched IVFFlat scan, for backends that are PROCESSES.
 *
 * The reference answers one query per backend at a time (ivfflatgettuple, src/ivfscan.c:361-414), and a backend
 * is a single-threaded process (src/ivfscan.c:252-296 runs in each).  On the GPU that shape streams each query's
 * lists alone (pgv_query_*: ~20 k queries/s per backend), while the same queries taken together share every pass
 * over a probed list (pgv_search_batch: 750 k/s).  This is the piece between the two.
 *
 * Everything the backends and the GPU side exchange lives in ONE position-independent shared-memory segment
 * (inside the server: a DSM segment or a ShmemInitStruct block, the pattern of the reference's parallel build,
 * src/ivfbuild.c:830-966; here: any MAP_SHARED mapping):
 *
 *     header   parameters, a process-shared robust mutex over a few words, futex words, the exported mirror
 *     lane[i]  batch state + query payloads [max_batch x row] + answers [max_batch x k] (tids, distances)
 *
 *   clients  (backends; no GPU context, they never call libpgv_hip) hand in ONE query each with
 *            pgv_host_pool_search and sleep on a futex word of their lane;
 *   servers  (one per lane; a background worker inside the server, a thread or a process here) own a pgv_ctx and a
 *            view of the device mirror (pgv_index_share in the owner's process, pgv_index_import in any other),
 *            close a batch when max_batch queries are waiting or max_wait_us after its first one, run ONE
 *            pgv_search_batch over the lane's payload area (page-locked with pgv_pinned_register) and wake the
 *            batch's clients.  Several lanes take batches in turn, so the next batch collects while the
 *            previous one scans.
 *
 * ONE SCAN AT A TIME (round 5).  A batch of b queries streams the lists its queries probe once: its time grows far
 * slower than b (1 M x 1536, probes 10 on MI355X: 16 queries 0.26 ms, 64: 0.71, 256: 1.10, 1024: 1.27), and two scans
 * that run side by side only share the HBM.  So a lane does not start its scan while another lane's is in flight: it
 * goes on collecting (everything that arrives during a scan shares the next one).  Closed-loop clients then settle into
 * two groups that alternate -- one scanning while the other is on its way back -- instead of two batches scanned side
 * by side (256 client processes, round 4: two batches of ~122 in flight together, 73-120 k QPS, p50 1.7-1.9 ms; now
 * 104-136 k).  From POOL_LINGER_MIN (128) queries on, the batch that collected lingers up to linger_us after the scan's
 * end for that scan's clients to come back and join it: same throughput, a quarter less latency (p50 1.63 vs 2.05 ms).
 * PGV_POOL_OVERLAP=1 in the environment of the process that initialises the segment brings the old behaviour back
 * (A/B measurements); PGV_POOL_LINGER_US overrides the 300 us.
 *
 * Answers are exactly pgv_search_batch's: the head of GetScanItems + tuplesort for each query
 * (src/ivfscan.c:123-187), heap TIDs and FUNCTION 1 distances.
 *
 * Futexes are the shared kind (no FUTEX_PRIVATE_FLAG); the mutex is PTHREAD_PROCESS_SHARED + ROBUST (a client that
 * dies inside the critical section does not wedge the pool); a lane whose clients vanished is reclaimed by its
 * server after POOL_RECLAIM_US, a batch waits for a straggler that never arrives POOL_READY_US at most, and a client
 * that outlives its batch's reclaim gets PGV_ERR_STATE instead of another batch's answers.
 */
#define _GNU_SOURCE
#include "pgv_host.h"

#include <errno.h>
#include <limits.h>
#include <linux/futex.h>
#include <pthread.h>
#include <sys/mman.h>
#include <sys/syscall.h>
#include <unistd.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

extern int	pgv_host_fail(int code, const char *fmt,...);
void		pgv_host_pool_destroy(pgv_pool * pool);

#define POOL_MAGIC 0x7067765f706f6f6cull	/* "pgv_pool" */
#define POOL_VERSION 7
#define POOL_MAX_LANES 8
#define POOL_ALIGN 4096
#define POOL_RECLAIM_US 2000000		/* a published batch nobody finished reading: its clients are gone */
#define POOL_READY_US 200000		/* a client between taking its slot and the end of its payload copy */
#define POOL_LEADER_DEAD_US 3000000	/* a lane whose server has not looked for this long has lost it (it looks every 50 ms) */
#define POOL_LINGER_US 300			/* after a scan: how long the next batch waits for that scan's clients to come back (round 6: 120 -> 300, see profiles/r06/pool_baton_wake.md) */
#define POOL_LINGER_MIN 128
#define POOL_WAKE_STRIDE 4096		/* one page per slot's wake word: see word_of_slot */			/* ... when the two groups together are at least this many queries */

enum
{
	LANE_FREE, LANE_COLLECTING, LANE_RUNNING, LANE_PUBLISHED
};

typedef struct
{
	uint32_t	state;			/* LANE_* (under the lock) */
	int32_t		count;			/* queries of the batch being collected / run (under the lock) */
	int32_t		ready;			/* ... whose payload has been copied into the lane (atomic) */
	int32_t		rc;
	uint64_t	readers;		/* (batch number << 32) | clients that still have to copy their answer (atomic): a
								 * client whose batch was reclaimed under it finds another batch number here and
								 * neither reads the lane nor counts itself out of somebody else's batch */
	uint32_t	gen,			/* batch number of this lane */
				done_gen;		/* last batch whose results are published (futex word) */
	uint32_t	fill;			/* bumped when the lane's server should look again (futex word) */
	int64_t		t_open;			/* now_us() of the batch's first query */
	int32_t		expect;			/* collecting while a scan ran: the count at which everyone that scan answered is back
								 * (count when it ended + its batch size); 0 = nobody is awaited (under the lock) */
	int32_t		pad0;
	int64_t		t_published;
	int64_t		beat;			/* now_us() of the server's last turn of its loop: 0 = no server has led this lane yet;
								 * stale = its server died (kill -9 runs no exit path): clients neither join nor wait */
	uint64_t	q_off,			/* offsets from the segment's base */
				tid_off,
				dist_off;
	char		errmsg[160];
	uint64_t	wake_off;		/* this lane's per-slot wake words, POOL_WAKE_STRIDE apart (a page each) */
	char		pad[8];
}			shm_lane;

typedef struct
{
	uint64_t	magic;
	uint32_t	version;
	uint32_t	nlanes;
	uint64_t	bytes;
	int32_t		probes,
				k,
				max_batch,
				max_wait_us;
	uint64_t	row_bytes;
	pthread_mutex_t lock;		/* guards collecting / arriving and the lanes' count / state: tens of nanoseconds.
								 * Adaptive (spins briefly, then sleeps): a pure spinlock collapses once there are more
								 * backends than cores (1024 clients: p90 latency 180 ms, measured) */
	int32_t		collecting;		/* lane that takes arrivals, or -1 */
	int32_t		arriving;		/* clients inside pgv_host_pool_search that have not joined a batch yet */
	int32_t		scan_lane;		/* lane whose scan is in flight, or -1 (under the lock; only with exclusive) */
	int32_t		exclusive;		/* one scan at a time (default); 0: lanes scan side by side (PGV_POOL_OVERLAP=1) */
	int32_t		linger_us;
	int32_t		wake_fanout;	/* clients a server wakes itself when it publishes; each of them wakes one more (0: all at once) */
	int64_t		t_scan_done;	/* now_us() when the last scan ended (under the lock) */
	int32_t		last_n;			/* ... and how many queries it answered */
	int32_t		pad2;
	uint32_t	free_epoch;		/* bumped when a lane comes free (futex word) */
	uint32_t	shutdown;
	uint32_t	servers;		/* lanes with a server attached (futex word: clients of an unserved pool fail fast) */
	uint32_t	has_index;		/* futex word: the owner published the mirror's handle */
	int64_t		batches,
				queries;
	pgv_index_handle index;
	shm_lane	lanes[POOL_MAX_LANES];
}			shm_pool;

/* a process's handle on the segment */
struct pgv_pool
{
	shm_pool   *s;
	char	   *base;
	/* pgv_host_pool_create only: the segment and the lane servers belong to this handle */
	int			owned;
	int			nthreads;
	pthread_t	threads[POOL_MAX_LANES];
	pgv_ctx    *ctxs[POOL_MAX_LANES];
	pgv_index  *views[POOL_MAX_LANES];
	int			thread_rc[POOL_MAX_LANES];
};

/* Hundreds of backends wait for one word (their batch's results, a lane coming free).  A condition variable
 * wakes them one futex call and one mutex hand-over at a time -- milliseconds per batch at 256 waiters (measured);
 * spinning starves the HIP runtime's own threads.  A bare futex: sleep until the word changes, wake all at once. */
static void
word_wait_us(uint32_t *word, uint32_t seen, long us)
{
	struct timespec rel = {us / 1000000L, (us % 1000000L) * 1000L};

	syscall(SYS_futex, word, FUTEX_WAIT, seen, &rel, NULL, 0);
}

static void
word_wake_all(uint32_t *word)
{
	syscall(SYS_futex, word, FUTEX_WAKE, INT_MAX, NULL, NULL, 0);
}

static void
word_wake_n(uint32_t *word, int n)
{
	syscall(SYS_futex, word, FUTEX_WAKE, n, NULL, NULL, 0);
}

/*
 * Where the client in slot `slot` of a lane sleeps.  Every futex call on a MAP_SHARED word pins the word's PAGE for the
 * key lookup (get_user_pages_fast: an atomic on the page's reference count): 256 backends doing their FUTEX_WAIT /
 * FUTEX_WAKE on words of ONE page spent 107 us of SYSTEM time per query on that cache line (getrusage of the clients,
 * profiles/r06/pool_baton_wake.md; 28 us with 64 clients).  One page per slot: nobody shares a futex page.
 */
static uint32_t *
word_of_slot(char *base, const shm_lane * l, int slot)
{
	return (uint32_t *) (base + l->wake_off + (size_t) POOL_WAKE_STRIDE * (size_t) slot);
}

/* CPUs this process may really use: the affinity mask, capped by the cgroup's CPU quota (cpu.max) */
static int
usable_cpus(void)
{
	long		n = sysconf(_SC_NPROCESSORS_ONLN);
	FILE	   *f = fopen("/sys/fs/cgroup/cpu.max", "r");

	if (f)
	{
		long long	quota = 0,
					period = 0;
		char		first[32];

		if (fscanf(f, "%31s %lld", first, &period) == 2 && strcmp(first, "max") != 0 && period > 0)
		{
			quota = atoll(first);
			if (quota > 0 && (quota + period - 1) / period < n)
				n = (long) ((quota + period - 1) / period);
		}
		fclose(f);
	}
	return n < 1 ? 1 : (int) n;
}

static int64_t
now_us(void)
{
	struct timespec ts;

	clock_gettime(CLOCK_MONOTONIC, &ts);
	return (int64_t) ts.tv_sec * 1000000 + ts.tv_nsec / 1000;
}

static void
pool_lock(shm_pool * s)
{
	if (pthread_mutex_lock(&s->lock) == EOWNERDEAD)
		pthread_mutex_consistent(&s->lock); /* its holder died: the words it guards are single stores, nothing is torn */
}

static void
pool_unlock(shm_pool * s)
{
	pthread_mutex_unlock(&s->lock);
}

static size_t
align_up(size_t v)
{
	return (v + POOL_ALIGN - 1) & ~(size_t) (POOL_ALIGN - 1);
}

size_t
pgv_host_pool_shm_bytes(pgv_dtype dtype, int dim, int k, int max_batch, int lanes)
{
	size_t		row = (size_t) dim * (dtype == PGV_F32 ? 4 : 2);
	size_t		lane = align_up(row * (size_t) max_batch) + align_up(sizeof(uint64_t) * (size_t) max_batch * k) +
		align_up(sizeof(float) * (size_t) max_batch * k) + (size_t) POOL_WAKE_STRIDE * (size_t) max_batch;

	if (dim < 1 || k < 1 || max_batch < 1 || lanes < 1 || lanes > POOL_MAX_LANES)
		return 0;
	return align_up(sizeof(shm_pool)) + lane * (size_t) lanes;
}

int
pgv_host_pool_shm_init(void *shm, size_t bytes, pgv_dtype dtype, int dim, int probes, int k, int max_batch,
					   int max_wait_us, int lanes)
{
	shm_pool   *s = shm;
	size_t		need = pgv_host_pool_shm_bytes(dtype, dim, k, max_batch, lanes);
	size_t		at;
	pthread_mutexattr_t ma;

	if (!shm)
		return pgv_host_fail(PGV_ERR_ARG, "pgv_host_pool_shm_init: segment is NULL");
	if (probes < 1 || k < 1 || max_batch < 1 || max_batch > 65536 || max_wait_us < 0 || lanes < 1 || lanes > POOL_MAX_LANES || dim < 1)
		return pgv_host_fail(PGV_ERR_ARG, "pgv_host_pool_shm_init: bad probes / k / max_batch / max_wait_us / lanes");
	if (need == 0 || bytes < need)
		return pgv_host_fail(PGV_ERR_ARG, "pgv_host_pool_shm_init: segment of %zu bytes, %zu needed", bytes, need);
	if (((uintptr_t) shm & 63) != 0)
		return pgv_host_fail(PGV_ERR_ARG, "pgv_host_pool_shm_init: segment must be 64-byte aligned");
	memset(s, 0, sizeof(shm_pool));
	s->version = POOL_VERSION;
	s->nlanes = (uint32_t) lanes;
	s->bytes = need;
	s->probes = probes;
	s->k = k;
	s->max_batch = max_batch;
	s->max_wait_us = max_wait_us;
	s->row_bytes = (uint64_t) dim * (dtype == PGV_F32 ? 4 : 2);
	s->collecting = -1;
	s->scan_lane = -1;
	s->exclusive = !(getenv("PGV_POOL_OVERLAP") && atoi(getenv("PGV_POOL_OVERLAP")) != 0);
	s->linger_us = getenv("PGV_POOL_LINGER_US") ? atoi(getenv("PGV_POOL_LINGER_US")) : POOL_LINGER_US;
	if (s->linger_us < 0)
		s->linger_us = 0;
	/* BATON WAKE (round 6).  A published batch used to wake all its clients at once (FUTEX_WAKE INT_MAX): a few hundred
	 * processes made runnable in one syscall -- ~1 us each inside the SERVER's critical path, a thundering herd on the pool
	 * mutex, and under a CPU quota (a container's cpu.max: 16 CPUs here) a burst that spends the period's budget at once and
	 * gets the whole group throttled for the rest of it (1024 clients: p90 75 ms).  Now the server wakes as many clients as
	 * there are usable CPUs, and every client, once it has copied its answer, wakes ONE more: the runnable set stays at the
	 * number of CPUs, the herd becomes a pipeline.  PGV_POOL_WAKE_FANOUT overrides (0: all at once, the old behaviour). */
	s->wake_fanout = getenv("PGV_POOL_WAKE_FANOUT") ? atoi(getenv("PGV_POOL_WAKE_FANOUT")) : usable_cpus();
	if (s->wake_fanout < 0)
		s->wake_fanout = 0;
	pthread_mutexattr_init(&ma);
	pthread_mutexattr_setpshared(&ma, PTHREAD_PROCESS_SHARED);
	pthread_mutexattr_setrobust(&ma, PTHREAD_MUTEX_ROBUST);
	pthread_mutexattr_settype(&ma, PTHREAD_MUTEX_ADAPTIVE_NP);
	if (pthread_mutex_init(&s->lock, &ma) != 0)
	{
		pthread_mutexattr_destroy(&ma);
		return pgv_host_fail(PGV_ERR_STATE, "pgv_host_pool_shm_init: process-shared mutex unavailable");
	}
	pthread_mutexattr_destroy(&ma);
	at = align_up(sizeof(shm_pool));
	for (int i = 0; i < lanes; i++)
	{
		shm_lane   *l = &s->lanes[i];

		l->state = LANE_FREE;
		l->q_off = at;
		at += align_up(s->row_bytes * (size_t) max_batch);
		l->tid_off = at;
		at += align_up(sizeof(uint64_t) * (size_t) max_batch * k);
		l->dist_off = at;
		at += align_up(sizeof(float) * (size_t) max_batch * k);
		l->wake_off = at;
		at += (size_t) POOL_WAKE_STRIDE * (size_t) max_batch;
	}
	__atomic_store_n(&s->magic, POOL_MAGIC, __ATOMIC_RELEASE);	/* last: attachers check it */
	return PGV_OK;
}

int
pgv_host_pool_attach(void *shm, size_t bytes, pgv_pool * *out)
{
	shm_pool   *s = shm;
	pgv_pool   *pool;

	if (!shm || !out)
		return pgv_host_fail(PGV_ERR_ARG, "pgv_host_pool_attach: segment/out is NULL");
	*out = NULL;
	if (bytes < sizeof(shm_pool) || __atomic_load_n(&s->magic, __ATOMIC_ACQUIRE) != POOL_MAGIC || s->version != POOL_VERSION)
		return pgv_host_fail(PGV_ERR_STATE, "pgv_host_pool_attach: not an initialised pool segment of this version");
	if (s->bytes > bytes)
		return pgv_host_fail(PGV_ERR_ARG, "pgv_host_pool_attach: mapped %zu of the segment's %llu bytes", bytes,
							 (unsigned long long) s->bytes);
	pool = calloc(1, sizeof(pgv_pool));
	if (!pool)
		return pgv_host_fail(PGV_ERR_NOMEM, "out of memory");
	pool->s = s;
	pool->base = shm;
	*out = pool;
	return PGV_OK;
}

void
pgv_host_pool_detach(pgv_pool * pool)
{
	pgv_host_pool_destroy(pool);	/* a handle that leads lanes shuts the pool down; a plain one is just freed */
}

int
pgv_host_pool_publish_index(pgv_pool * pool, const pgv_index_handle * handle)
{
	if (!pool || !handle)
		return pgv_host_fail(PGV_ERR_ARG, "pgv_host_pool_publish_index: pool/handle is NULL");
	memcpy(&pool->s->index, handle, sizeof(*handle));
	__atomic_store_n(&pool->s->has_index, 1, __ATOMIC_RELEASE);
	word_wake_all(&pool->s->has_index);
	return PGV_OK;
}

int
pgv_host_pool_index_handle(pgv_pool * pool, int wait_ms, pgv_index_handle * out)
{
	int64_t		deadline = now_us() + (int64_t) wait_ms * 1000;

	if (!pool || !out)
		return pgv_host_fail(PGV_ERR_ARG, "pgv_host_pool_index_handle: pool/out is NULL");
	while (__atomic_load_n(&pool->s->has_index, __ATOMIC_ACQUIRE) == 0)
	{
		int64_t		t = now_us();

		if (t >= deadline || __atomic_load_n(&pool->s->shutdown, __ATOMIC_ACQUIRE))
			return pgv_host_fail(PGV_ERR_STATE, "pgv_host_pool_index_handle: no mirror has been published");
		word_wait_us(&pool->s->has_index, 0, deadline - t < 50000 ? deadline - t : 50000);
	}
	memcpy(out, &pool->s->index, sizeof(*out));
	return PGV_OK;
}

void
pgv_host_pool_shutdown(pgv_pool * pool)
{
	shm_pool   *s;

	if (!pool)
		return;
	s = pool->s;
	__atomic_store_n(&s->shutdown, 1, __ATOMIC_RELEASE);
	for (uint32_t i = 0; i < s->nlanes; i++)
	{
		__atomic_add_fetch(&s->lanes[i].fill, 1, __ATOMIC_RELEASE);
		word_wake_all(&s->lanes[i].fill);
		word_wake_all(&s->lanes[i].done_gen);
		if (s->wake_fanout > 0)
			for (int j = 0; j < s->max_batch; j++)	/* the clients asleep on their slots' words */
			{
				uint32_t   *w = word_of_slot(pool->base, &s->lanes[i], j);

				__atomic_add_fetch(w, 1, __ATOMIC_RELEASE);
				word_wake_n(w, 1);
			}
	}
	__atomic_add_fetch(&s->free_epoch, 1, __ATOMIC_RELEASE);
	word_wake_all(&s->free_epoch);
	word_wake_all(&s->has_index);
}

int
pgv_host_pool_is_shut_down(pgv_pool * pool)
{
	return pool == NULL || __atomic_load_n(&pool->s->shutdown, __ATOMIC_ACQUIRE) != 0;
}

/* does somebody lead this lane?  (A server that was killed stops beating; one that left cleanly zeroed the word.) */
static int
lane_is_led(const shm_lane * l, int64_t now)
{
	int64_t		beat = __atomic_load_n(&l->beat, __ATOMIC_ACQUIRE);

	return beat != 0 && now - beat < POOL_LEADER_DEAD_US;
}

/* the last reader (or the lane's server, for a batch whose clients vanished) hands the lane back */
static void
lane_release(shm_pool * s, shm_lane * l)
{
	pool_lock(s);
	l->state = LANE_FREE;
	__atomic_add_fetch(&s->free_epoch, 1, __ATOMIC_RELEASE);
	pool_unlock(s);
	word_wake_all(&s->free_epoch);	/* everyone queued joins the batch the first of them opens */
}

/*
 * The leader loop of one lane: wait for a batch to open, let it fill, scan it, publish.  Returns PGV_OK at
 * pgv_host_pool_shutdown, or the first error that is not a batch's own (a failed batch is reported to its clients
 * and the lane goes on).  `view` is this process's handle on the mirror; its context's stream runs the scans.
 */
int
pgv_host_pool_serve(pgv_pool * pool, int lane, pgv_index * view)
{
	shm_pool   *s;
	shm_lane   *l;
	char	   *q;
	uint64_t   *tids;
	float	   *dist;
	int			pinned_q,
				pinned_t,
				pinned_d;

	if (!pool || !view)
		return pgv_host_fail(PGV_ERR_ARG, "pgv_host_pool_serve: pool/view is NULL");
	s = pool->s;
	if (lane < 0 || lane >= (int) s->nlanes)
		return pgv_host_fail(PGV_ERR_ARG, "pgv_host_pool_serve: lane %d of %u", lane, s->nlanes);
	l = &s->lanes[lane];
	q = pool->base + l->q_off;
	tids = (uint64_t *) (pool->base + l->tid_off);
	dist = (float *) (pool->base + l->dist_off);
	/* page-lock this lane's areas for this process's GPU context: copies in and out become asynchronous DMA.  A
	 * mapping the driver cannot pin still works, staged through the library's own pinned scratch. */
	pinned_q = pgv_pinned_register(q, align_up(s->row_bytes * (size_t) s->max_batch)) == PGV_OK;
	pinned_t = pgv_pinned_register(tids, align_up(sizeof(uint64_t) * (size_t) s->max_batch * s->k)) == PGV_OK;
	pinned_d = pgv_pinned_register(dist, align_up(sizeof(float) * (size_t) s->max_batch * s->k)) == PGV_OK;
	/* a lane whose previous leader died in mid-batch is in whatever state it left: its clients have been told (or will
	 * time out on the old batch number); the lane starts over */
	pool_lock(s);
	if (l->state != LANE_FREE)
	{
		if (s->collecting == lane)
			s->collecting = -1;
		l->state = LANE_FREE;
		l->count = 0;
		__atomic_store_n(&l->done_gen, l->gen, __ATOMIC_RELEASE);	/* nobody waits for a batch that will not come */
		__atomic_store_n(&l->readers, (uint64_t) l->gen << 32, __ATOMIC_RELEASE);
		__atomic_add_fetch(&s->free_epoch, 1, __ATOMIC_RELEASE);
	}
	pool_unlock(s);
	__atomic_store_n(&l->beat, now_us(), __ATOMIC_RELEASE);	/* led from now on (clients join led lanes only) */
	__atomic_add_fetch(&s->servers, 1, __ATOMIC_RELEASE);
	word_wake_all(&s->servers);

	for (;;)
	{
		int			n = 0;
		uint32_t	gen = 0;
		int			rc;

		/* a batch opens in this lane (its first client bumps `fill`), then fills until max_batch or the deadline */
		for (;;)
		{
			uint32_t	seen = __atomic_load_n(&l->fill, __ATOMIC_ACQUIRE);
			int64_t		t;
			long		nap = 50000;

			if (__atomic_load_n(&s->shutdown, __ATOMIC_ACQUIRE))
				goto done;
			t = now_us();
			__atomic_store_n(&l->beat, t, __ATOMIC_RELEASE);
			pool_lock(s);
			if (l->state == LANE_COLLECTING && l->count > 0)
			{
				int64_t		deadline = l->t_open + s->max_wait_us;
				int			busy = 0;

				if (s->exclusive && s->scan_lane >= 0 && s->scan_lane != lane)
				{
					/* another lane's scan is in flight: this batch goes on collecting (its server is woken when that
					 * scan ends) -- unless that lane's server died under its scan */
					if (lane_is_led(&s->lanes[s->scan_lane], t))
						busy = 1;
					else
						s->scan_lane = -1;
				}
				/* the clients of the scan that has just ended are on their way back: give them linger_us to join -- when the
				 * merged batch is large enough for the scan's time to dominate.  Fewer clients do better as two groups that
				 * alternate, one scanning while the other is on its way back (64 client processes, same box: 71.8 k QPS in two
				 * groups of 32 against 64.3 k merged; 256: 104 k / p50 2.05 ms in two groups against 105 k / p50 1.63 ms
				 * merged): no linger below POOL_LINGER_MIN queries */
				if (!busy && s->exclusive && l->expect >= POOL_LINGER_MIN && l->count < l->expect &&
					deadline < s->t_scan_done + s->linger_us)
					deadline = s->t_scan_done + s->linger_us;
				/* everyone who queued while the lanes were busy joins (they are on their way: `arriving`), later
				 * arrivals get max_wait_us */
				/* ... but not for ever: a client that died between arriving++ and its slot leaves `arriving` above
				 * zero for good, and a batch must not wait for it longer than a payload copy takes */
				if (!busy &&
					(l->count >= s->max_batch || (s->arriving == 0 && t >= deadline) || t >= deadline + POOL_READY_US))
				{
					if (s->collecting == lane)
						s->collecting = -1;
					l->state = LANE_RUNNING;
					l->expect = 0;
					if (s->exclusive)
						s->scan_lane = lane;
					n = l->count;
					gen = l->gen;
					s->batches++;
					s->queries += n;
					pool_unlock(s);
					break;
				}
				/* stragglers still on their way past the deadline: short naps, bounded by their own progress */
				nap = busy ? 2000 : (t < deadline ? (long) (deadline - t) : 20);
			}
			else if (l->state == LANE_PUBLISHED && t - l->t_published > POOL_RECLAIM_US)
			{
				/* nobody finished reading for two seconds: the batch's clients are gone */
				uint64_t	w;

				pool_unlock(s);
				/* take the readers that are left out of the count in one step; the last real reader may be doing the
				 * same this instant, and only one of us releases the lane */
				w = __atomic_load_n(&l->readers, __ATOMIC_ACQUIRE);
				while ((uint32_t) w != 0 &&
					   !__atomic_compare_exchange_n(&l->readers, &w, w & ~(uint64_t) 0xffffffffu, 0, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE))
					;
				if ((uint32_t) w != 0)
					lane_release(s, l);
				continue;
			}
			pool_unlock(s);
			if (nap > 100)
			{
				/* a short spin first: the wake-up of a sleeping server costs more than the batch's planning */
				int64_t		until = now_us() + 30;

				while (__atomic_load_n(&l->fill, __ATOMIC_ACQUIRE) == seen && now_us() < until)
					__builtin_ia32_pause();
			}
			if (__atomic_load_n(&l->fill, __ATOMIC_ACQUIRE) == seen)
				word_wait_us(&l->fill, seen, nap);
		}
		{
			int64_t		until = now_us() + POOL_READY_US;

			while (__atomic_load_n(&l->ready, __ATOMIC_ACQUIRE) < n && now_us() < until)
				__builtin_ia32_pause();	/* a client between its slot and the end of its 6 KB memcpy */
		}

		__atomic_store_n(&l->beat, now_us(), __ATOMIC_RELEASE);
		rc = pgv_search_batch(view, q, n, s->probes, s->k, dist, NULL, tids);
		__atomic_store_n(&l->beat, now_us(), __ATOMIC_RELEASE);
		if (s->exclusive)
		{
			/* the device is free: the batch that collected meanwhile may go -- once this scan's clients have had
			 * linger_us to join it */
			int			next;

			pool_lock(s);
			if (s->scan_lane == lane)
				s->scan_lane = -1;
			s->t_scan_done = now_us();
			s->last_n = n;
			next = s->collecting;
			if (next >= 0 && next != lane)
				s->lanes[next].expect = s->lanes[next].count + n;
			pool_unlock(s);
			for (uint32_t i = 0; i < s->nlanes; i++)
				if ((int) i != lane)
				{
					__atomic_add_fetch(&s->lanes[i].fill, 1, __ATOMIC_RELEASE);
					word_wake_all(&s->lanes[i].fill);
				}
		}

		/* publish: the clients sleep on done_gen */
		l->rc = rc;
		if (rc != PGV_OK)
		{
			strncpy(l->errmsg, pgv_last_error(), sizeof(l->errmsg) - 1);
			l->errmsg[sizeof(l->errmsg) - 1] = 0;
		}
		l->t_published = now_us();
		pool_lock(s);
		l->state = LANE_PUBLISHED;
		pool_unlock(s);
		__atomic_store_n(&l->readers, ((uint64_t) gen << 32) | (uint32_t) n, __ATOMIC_RELEASE);
		__atomic_store_n(&l->done_gen, gen, __ATOMIC_RELEASE);
		if (s->wake_fanout > 0)
		{
			/* the first wake_fanout slots; slot i wakes slot i + wake_fanout when it has its answer */
			for (int i = 0; i < n && i < s->wake_fanout; i++)
			{
				uint32_t   *w = word_of_slot(pool->base, l, i);

				__atomic_add_fetch(w, 1, __ATOMIC_RELEASE);
				word_wake_n(w, 1);
			}
		}
		else
			word_wake_all(&l->done_gen);
	}
done:
	__atomic_store_n(&l->beat, 0, __ATOMIC_RELEASE);
	__atomic_sub_fetch(&s->servers, 1, __ATOMIC_RELEASE);
	if (pinned_q)
		pgv_pinned_unregister(q);
	if (pinned_t)
		pgv_pinned_unregister(tids);
	if (pinned_d)
		pgv_pinned_unregister(dist);
	return PGV_OK;
}

/* One backend's query: blocks until its batch has been scanned.  out_tid / out_dist [k]: ascending, padded with
 * ~0 / +inf when the probed lists hold fewer than k tuples (exactly pgv_search_batch's row).  Touches shared
 * memory only -- the calling process needs no GPU context. */
int
pgv_host_pool_search(pgv_pool * pool, const void *query, uint64_t *out_tid, float *out_dist)
{
	shm_pool   *s;
	shm_lane   *l;
	int			lane,
				slot;
	uint32_t	gen,
				seen;
	int			rc,
				kick;
	int64_t		t_arrived = now_us();

	if (!pool || !query || !out_tid || !out_dist)
		return pgv_host_fail(PGV_ERR_ARG, "pgv_host_pool_search: pool/query/out is NULL");
	s = pool->s;
	if (__atomic_load_n(&s->shutdown, __ATOMIC_ACQUIRE))
		return pgv_host_fail(PGV_ERR_STATE, "pgv_host_pool_search: the pool is shut down");
	pool_lock(s);
	s->arriving++;
	/* the batch that is collecting, or a new one in a free lane */
	for (;;)
	{
		uint32_t	epoch;

		/* a batch that collects in a lane whose leader has died will never run: nobody else joins it */
		if (s->collecting >= 0 && !lane_is_led(&s->lanes[s->collecting], now_us()))
			s->collecting = -1;
		if (s->collecting >= 0)
			break;
		for (uint32_t i = 0; i < s->nlanes; i++)
			if (s->lanes[i].state == LANE_FREE && lane_is_led(&s->lanes[i], now_us()))
			{
				s->collecting = (int32_t) i;
				s->lanes[i].state = LANE_COLLECTING;
				s->lanes[i].count = 0;
				__atomic_store_n(&s->lanes[i].ready, 0, __ATOMIC_RELAXED);
				s->lanes[i].gen++;
				s->lanes[i].t_open = now_us();
				/* opened by the first client back from a scan that has just ended: the others are right behind */
				s->lanes[i].expect = (s->exclusive && s->scan_lane < 0 && s->lanes[i].t_open - s->t_scan_done < s->linger_us)
					? s->last_n : 0;
				break;
			}
		if (s->collecting >= 0)
			break;
		epoch = __atomic_load_n(&s->free_epoch, __ATOMIC_ACQUIRE);
		if (__atomic_load_n(&s->shutdown, __ATOMIC_ACQUIRE))
		{
			s->arriving--;
			pool_unlock(s);
			return pgv_host_fail(PGV_ERR_STATE, "pgv_host_pool_search: the pool is shut down");
		}
		{
			/* no lane to join: busy ones come free.  But when NO lane has a live server (none attached for five
			 * seconds, or every leader dead) there is nothing to wait for */
			int64_t		t = now_us();
			int			led = 0;

			for (uint32_t i = 0; i < s->nlanes; i++)
				led += lane_is_led(&s->lanes[i], t);
			if (led == 0 && t - t_arrived > 5000000)
			{
				s->arriving--;
				pool_unlock(s);
				return pgv_host_fail(PGV_ERR_STATE, "pgv_host_pool_search: no server is attached to the pool");
			}
		}
		pool_unlock(s);
		word_wait_us(&s->free_epoch, epoch, 100000);
		pool_lock(s);
	}
	s->arriving--;
	lane = s->collecting;
	l = &s->lanes[lane];
	slot = l->count++;
	gen = l->gen;
	if (l->count == s->max_batch)
		s->collecting = -1;		/* closed: the next arrival opens another lane */
	/* the first query opens the batch, a full batch or the last of those who were queueing closes it: the lane's
	 * server should look */
	kick = slot == 0 || l->count == s->max_batch || s->arriving == 0 || (l->expect > 0 && l->count >= l->expect);
	pool_unlock(s);

	/* the payload goes in outside the lock; the server waits for `ready` to reach `count` */
	memcpy(pool->base + l->q_off + (size_t) slot * s->row_bytes, query, s->row_bytes);
	__atomic_add_fetch(&l->ready, 1, __ATOMIC_RELEASE);
	if (kick)
	{
		__atomic_add_fetch(&l->fill, 1, __ATOMIC_RELEASE);
		word_wake_all(&l->fill);
	}
	/* (signed distance: a client that stalled past its batch's reclaim sees LATER batch numbers here and must not
	 * wait for its own to come round again) */
	while ((int32_t) ((seen = __atomic_load_n(&l->done_gen, __ATOMIC_ACQUIRE)) - gen) < 0)
	{
		if (__atomic_load_n(&s->shutdown, __ATOMIC_ACQUIRE))
			return pgv_host_fail(PGV_ERR_STATE, "pgv_host_pool_search: the pool was shut down under a waiting query");
		if (__atomic_load_n(&s->servers, __ATOMIC_ACQUIRE) == 0 && now_us() - l->t_open > 5000000)
			return pgv_host_fail(PGV_ERR_STATE, "pgv_host_pool_search: no server is attached to the pool");
		/* the lane's leader died under the batch (kill -9, a crash): nobody will publish it */
		if (!lane_is_led(l, now_us()) && now_us() - l->t_open > POOL_LEADER_DEAD_US)
		{
			pool_lock(s);
			if (s->collecting == lane)
				s->collecting = -1;	/* (the lane itself stays out of use: it is never FREE again) */
			pool_unlock(s);
			return pgv_host_fail(PGV_ERR_STATE, "pgv_host_pool_search: the lane's server is gone");
		}
		if (s->wake_fanout > 0)
		{
			/* sleep on this slot's own word: whoever wakes it bumps it first, and done_gen was published before any
			 * wake -- read the word, look at done_gen once more, then wait for the word to move */
			uint32_t   *mine = word_of_slot(pool->base, l, slot);
			uint32_t	w = __atomic_load_n(mine, __ATOMIC_ACQUIRE);

			if ((int32_t) (__atomic_load_n(&l->done_gen, __ATOMIC_ACQUIRE) - gen) >= 0)
				continue;
			word_wait_us(mine, w, 100000);
		}
		else
			word_wait_us(&l->done_gen, seen, 100000);
	}
	if (seen != gen)
		return pgv_host_fail(PGV_ERR_STATE, "pgv_host_pool_search: the batch was reclaimed before this client read its answer");
	rc = l->rc;
	if (rc == PGV_OK)
	{
		memcpy(out_tid, pool->base + l->tid_off + sizeof(uint64_t) * (size_t) slot * s->k, sizeof(uint64_t) * (size_t) s->k);
		memcpy(out_dist, pool->base + l->dist_off + sizeof(float) * (size_t) slot * s->k, sizeof(float) * (size_t) s->k);
	}
	else
		pgv_host_fail(rc, "batch failed: %s", l->errmsg);
	/* the baton: the client wake_fanout slots further on */
	if (s->wake_fanout > 0 && slot + s->wake_fanout < l->count)
	{
		uint32_t   *w = word_of_slot(pool->base, l, slot + s->wake_fanout);

		__atomic_add_fetch(w, 1, __ATOMIC_RELEASE);
		word_wake_n(w, 1);
	}
	/* count this reader out -- of ITS batch only; the lane is free again when its last reader has its answer.  A
	 * reader that finds another batch number (or no readers left) was given up on by the lane's server
	 * (POOL_RECLAIM_US): what it copied may belong to the next batch */
	for (;;)
	{
		uint64_t	w = __atomic_load_n(&l->readers, __ATOMIC_ACQUIRE);

		if ((uint32_t) (w >> 32) != gen || (uint32_t) w == 0)
			return pgv_host_fail(PGV_ERR_STATE, "pgv_host_pool_search: the batch was reclaimed while this client read its answer");
		if (__atomic_compare_exchange_n(&l->readers, &w, w - 1, 0, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE))
		{
			if ((uint32_t) (w - 1) == 0)
				lane_release(s, l);
			break;
		}
	}
	return rc;
}

void
pgv_host_pool_stats(pgv_pool * pool, int64_t *batches, int64_t *queries)
{
	pool_lock(pool->s);
	if (batches)
		*batches = pool->s->batches;
	if (queries)
		*queries = pool->s->queries;
	pool_unlock(pool->s);
}

/* -------------------------------------------------------------- one process: the segment and lane threads in it */

typedef struct
{
	pgv_pool   *pool;
	int			lane;
}			serve_arg;

static void *
serve_thread(void *p)
{
	serve_arg  *a = p;
	pgv_pool   *pool = a->pool;
	int			lane = a->lane;

	free(a);
	pool->thread_rc[lane] = pgv_host_pool_serve(pool, lane, pool->views[lane]);
	return NULL;
}

/* join the lane threads this handle started (the pool must be shut down), drop their views and contexts */
static void
stop_threads(pgv_pool * pool)
{
	for (int i = 0; i < pool->nthreads; i++)
		pthread_join(pool->threads[i], NULL);
	pool->nthreads = 0;
	for (int i = 0; i < POOL_MAX_LANES; i++)
	{
		if (pool->views[i])
			pgv_index_free(pool->views[i]);
		if (pool->ctxs[i])
			pgv_ctx_destroy(pool->ctxs[i]);
		pool->views[i] = NULL;
		pool->ctxs[i] = NULL;
	}
}

void
pgv_host_pool_destroy(pgv_pool * pool)
{
	if (!pool)
		return;
	if (pool->nthreads > 0 || pool->owned)
		pgv_host_pool_shutdown(pool);
	stop_threads(pool);
	if (pool->owned)
	{
		pthread_mutex_destroy(&pool->s->lock);
		munmap(pool->s, pool->s->bytes);
	}
	free(pool);
}

/*
 * Lead every lane of an attached pool with a thread of the calling process, which owns `index` (one context + one
 * pgv_index_share view per lane).  The threads end at pgv_host_pool_shutdown and are joined by
 * pgv_host_pool_destroy / _detach of this handle.
 */
int
pgv_host_pool_start_threads(pgv_pool * pool, pgv_index * index, int device)
{
	int			lanes,
				rc = PGV_OK;

	if (!pool || !index)
		return pgv_host_fail(PGV_ERR_ARG, "pgv_host_pool_start_threads: pool/index is NULL");
	if (pool->nthreads > 0)
		return pgv_host_fail(PGV_ERR_STATE, "pgv_host_pool_start_threads: this handle already leads the lanes");
	lanes = (int) pool->s->nlanes;
	for (int i = 0; i < lanes && rc == PGV_OK; i++)
	{
		rc = pgv_ctx_create(device, NULL, &pool->ctxs[i]);
		if (rc == PGV_OK)
			rc = pgv_index_share(index, pool->ctxs[i], &pool->views[i]);
		if (rc != PGV_OK)
			pgv_host_fail(rc, "%s", pgv_last_error());
	}
	for (int i = 0; i < lanes && rc == PGV_OK; i++)
	{
		serve_arg  *a = malloc(sizeof(serve_arg));

		if (!a)
		{
			rc = pgv_host_fail(PGV_ERR_NOMEM, "out of memory");
			break;
		}
		a->pool = pool;
		a->lane = i;
		if (pthread_create(&pool->threads[i], NULL, serve_thread, a) != 0)
		{
			free(a);
			rc = pgv_host_fail(PGV_ERR_STATE, "pgv_host_pool_start_threads: cannot start the lane's thread");
			break;
		}
		pool->nthreads++;
	}
	if (rc != PGV_OK)
	{
		if (pool->nthreads > 0)
			pgv_host_pool_shutdown(pool);
		stop_threads(pool);
		return rc;
	}
	/* the lanes are serving before the first client can arrive */
	while (__atomic_load_n(&pool->s->servers, __ATOMIC_ACQUIRE) < (uint32_t) lanes)
		word_wait_us(&pool->s->servers, __atomic_load_n(&pool->s->servers, __ATOMIC_ACQUIRE), 1000);
	return PGV_OK;
}

/*
 * Segment, lane contexts and lane servers (threads) all in the calling process, which owns `index`.  The segment
 * is an anonymous MAP_SHARED mapping: children forked afterwards are clients of the same pool (they must not
 * touch the GPU -- a forked HIP runtime is unusable -- and pgv_host_pool_search does not).
 */
int
pgv_host_pool_create(pgv_index * index, int device, pgv_dtype dtype, int dim, int probes, int k, int max_batch,
					 int max_wait_us, int lanes, pgv_pool * *out)
{
	pgv_pool   *pool;
	size_t		bytes;
	void	   *shm;
	int			rc;

	if (!index || !out)
		return pgv_host_fail(PGV_ERR_ARG, "pgv_host_pool_create: index/out is NULL");
	*out = NULL;
	bytes = pgv_host_pool_shm_bytes(dtype, dim, k, max_batch, lanes);
	if (bytes == 0 || probes < 1 || max_batch > 65536 || max_wait_us < 0)
		return pgv_host_fail(PGV_ERR_ARG, "pgv_host_pool_create: bad probes / k / max_batch / max_wait_us / lanes");
	shm = mmap(NULL, bytes, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
	if (shm == MAP_FAILED)
		return pgv_host_fail(PGV_ERR_NOMEM, "pgv_host_pool_create: mmap of %zu shared bytes failed", bytes);
	rc = pgv_host_pool_shm_init(shm, bytes, dtype, dim, probes, k, max_batch, max_wait_us, lanes);
	if (rc == PGV_OK)
		rc = pgv_host_pool_attach(shm, bytes, &pool);
	if (rc != PGV_OK)
	{
		munmap(shm, bytes);
		return rc;
	}
	pool->owned = 1;
	rc = pgv_host_pool_start_threads(pool, index, device);
	if (rc != PGV_OK)
	{
		pgv_host_pool_destroy(pool);
		return rc;
	}
	*out = pool;
	return PGV_OK;
}
