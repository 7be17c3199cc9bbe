/*
 * host_logic_driver.c -- the C host glue (libpgv_host.so) exercised on the CPU against the
 * stand-in device of mock_hip.c: what is tested is the HOST LOGIC around the distance calls --
 * the HNSW build loop against the oracle's graph, the IVFFlat build / stage / scan / iterative
 * scan drivers, vacuum and the self-invalidating mirror -- none of which needs a GPU to be wrong.
 *
 * TEST INFRASTRUCTURE (compiled and run by tests/test_host_logic_cpu.py).  Exit status 0 and
 * "HOST-LOGIC OK" on success.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pgv_hip.h"
#include "pgv_host.h"
#include "pgv_oracle.h"

#define CHECK(call) do { int rc_ = (call); if (rc_ != PGV_OK) { \
	fprintf(stderr, "%s:%d: %s -> %d: %s / %s\n", __FILE__, __LINE__, #call, rc_, pgv_last_error(), pgv_host_last_error()); \
	return 1; } } while (0)
#define EXPECT(cond) do { if (!(cond)) { fprintf(stderr, "%s:%d: EXPECT(%s) failed\n", __FILE__, __LINE__, #cond); return 1; } } while (0)

/* ---- signals during pgv_host_hnsw_build: where do the handlers run? */
#include <pthread.h>
#include <signal.h>
#include <unistd.h>
static pthread_t main_thread;
static volatile int signals_seen,
			signals_off_main,
			storm_on,
			cancel_after,
			cancel_polls;

static void
on_sigusr1(int sig)
{
	(void) sig;
	signals_seen++;
	if (!pthread_equal(pthread_self(), main_thread))
		signals_off_main++;
}

/* process-directed signals: the kernel delivers each to ANY thread that has it unblocked */
static void *
storm_main(void *arg)
{
	sigset_t	all;

	(void) arg;
	sigfillset(&all);
	pthread_sigmask(SIG_BLOCK, &all, NULL);	/* (not to this thread) */
	while (storm_on)
	{
		kill(getpid(), SIGUSR1);
		usleep(200);
	}
	return NULL;
}

static int
cancel_cb(void *arg)
{
	(void) arg;
	return cancel_polls++ >= cancel_after;
}

static uint64_t lcg = 12345;
static uint32_t
urand(void)
{
	lcg = lcg * 6364136223846793005ull + 1442695040888963407ull;
	return (uint32_t) (lcg >> 33);
}

static float
l2sq(const float *a, const float *b, int dim)
{
	float		s = 0.0f;

	for (int i = 0; i < dim; i++)
		s += (a[i] - b[i]) * (a[i] - b[i]);
	return s;
}

/* ---- HNSW: the serial build reproduces the oracle's graph; a batched build searches as well */
static int
test_hnsw_build(void)
{
	enum { N = 900, DIM = 8, M = 6, EFC = 24 };
	float	   *data = malloc(sizeof(float) * N * DIM);
	pgv_ctx    *ctx;
	pgv_hnsw   *mirror;
	pgv_hnsw_built built;
	ora_hnsw   *g;
	ora_prng	st;
	pgv_rng		rng;
	int			same = 0,
				entry_level;
	int32_t		buf[2 * M];

	/* integer coordinates < 1024 in 8 dimensions: every distance is exact in fp32, ties are rare */
	for (int i = 0; i < N * DIM; i++)
		data[i] = (float) (urand() % 1024);
	CHECK(pgv_ctx_create(0, NULL, &ctx));
	CHECK(pgv_hnsw_upload(ctx, PGV_L2SQ, PGV_F32, DIM, data, N, &mirror));
	g = ora_hnsw_build(ORA_OPS_L2, ORA_F32, DIM, data, N, M, EFC, 21);
	EXPECT(ora_hnsw_num_elements(g) == N);
	ora_prng_seed(&st, 21);
	rng.next_double = ora_prng_double_cb;
	rng.next_u32 = ora_prng_u32_cb;
	rng.state = &st;
	rng.seed = 0;
	CHECK(pgv_host_hnsw_build(mirror, PGV_F32, DIM, data, N, M, EFC, &rng, 1, &built));
	EXPECT(built.nelements == N && built.batches == N);
	EXPECT(built.entry == ora_hnsw_entry_point(g, &entry_level));
	for (int e = 0; e < N; e++)
	{
		int			ok = built.levels[e] == ora_hnsw_level(g, e);

		EXPECT(ok);
		for (int lc = 0; lc <= built.levels[e] && ok; lc++)
		{
			int			lm = lc == 0 ? 2 * M : M;
			int			cnt = ora_hnsw_neighbors(g, e, lc, buf);
			const int32_t *mine = built.nbr + built.nbr_start[e] + (int64_t) (built.levels[e] - lc) * M;

			for (int i = 0; i < lm && ok; i++)
				ok = mine[i] == (i < cnt ? buf[i] : -1);
		}
		same += ok;
	}
	/* only the order in which tied candidates are walked may differ (unspecified in the reference) */
	EXPECT(same >= N * 99 / 100);
	pgv_host_hnsw_built_free(&built);

	/* batched inserts: structurally sound and every element reachable enough to find itself */
	CHECK(pgv_host_hnsw_build(mirror, PGV_F32, DIM, data, N, M, EFC, NULL, 32, &built));
	EXPECT(built.nelements == N && built.batches < N / 4);
	for (int e = 0; e < N; e++)
		for (int lc = 0; lc <= built.levels[e]; lc++)
		{
			int			lm = lc == 0 ? 2 * M : M;
			const int32_t *mine = built.nbr + built.nbr_start[e] + (int64_t) (built.levels[e] - lc) * M;

			for (int i = 0; i < lm; i++)
			{
				EXPECT(mine[i] >= -1 && mine[i] < N && mine[i] != e);
				if (mine[i] >= 0)
					EXPECT(built.levels[mine[i]] >= lc);
				for (int j = 0; j < i; j++)
					EXPECT(mine[i] < 0 || mine[i] != mine[j]);
			}
		}
	{
		/* search the built graph with the host-driven walk: a stored vector finds itself first */
		pgv_hnsw_graph graph = {N, M, built.entry, built.levels, built.nbr_start, built.nbr};
		int64_t		elem[16 * 5],
					scored[16];
		float		d[16 * 5];
		int			found = 0;

		CHECK(pgv_host_hnsw_search(mirror, &graph, PGV_F32, DIM, data + 100 * DIM, 16, 40, 5, elem, d, scored));
		for (int q = 0; q < 16; q++)
			found += elem[q * 5] == 100 + q && d[q * 5] == 0.0f;
		EXPECT(found >= 15);
	}
	/* the graph updates on the "device" (csrc/hnsw_link_core.h through the stand-in: the source the GPU kernel is compiled
	 * from) and replayed on the host build the SAME graph, at batch 32 and with the pipeline running (batch 64 from 1024
	 * linked elements on: searches, selection and linking on three threads) -- whichever of the two this process runs by
	 * default, the other one is asked for here */
	{
		const char *was = getenv("PGV_HNSW_HOST_LINK");
		const int	host_default = (was && atoi(was) != 0) || (getenv("PGV_HNSW_HOST_SELECT") && atoi(getenv("PGV_HNSW_HOST_SELECT")) != 0);

		for (int mb = 32; mb <= 64; mb += 32)
		{
			pgv_hnsw_built a,
						b;

			unsetenv("PGV_HNSW_HOST_LINK");
			if (host_default)
				unsetenv("PGV_HNSW_HOST_SELECT");
			CHECK(pgv_host_hnsw_build(mirror, PGV_F32, DIM, data, N, M, EFC, NULL, mb, &a));
			setenv("PGV_HNSW_HOST_LINK", "1", 1);
			CHECK(pgv_host_hnsw_build(mirror, PGV_F32, DIM, data, N, M, EFC, NULL, mb, &b));
			EXPECT(a.entry == b.entry && a.batches == b.batches && a.nelements == b.nelements);
			EXPECT(memcmp(a.levels, b.levels, sizeof(int32_t) * N) == 0);
			EXPECT(memcmp(a.dup_of, b.dup_of, sizeof(int32_t) * N) == 0);
			EXPECT(memcmp(a.nbr, b.nbr, sizeof(int32_t) * (size_t) a.nbr_start[N]) == 0);
			/* (both forms met lists whose replay needed the member-member distances: the second round ran) */
			EXPECT(a.deferred_updates > 0 && b.deferred_updates > 0);
			if (mb == 32)
				EXPECT(memcmp(a.nbr, built.nbr, sizeof(int32_t) * (size_t) a.nbr_start[N]) == 0);
			pgv_host_hnsw_built_free(&a);
			pgv_host_hnsw_built_free(&b);
		}
		/* a device without room for the graph state: the build goes on with the host-side replay, same graph */
		{
			pgv_hnsw_built c;

			unsetenv("PGV_HNSW_HOST_LINK");
			setenv("MOCK_HIP_LINK_NOMEM", "1", 1);
			CHECK(pgv_host_hnsw_build(mirror, PGV_F32, DIM, data, N, M, EFC, NULL, 32, &c));
			unsetenv("MOCK_HIP_LINK_NOMEM");
			EXPECT(c.nelements == built.nelements && memcmp(c.nbr, built.nbr, sizeof(int32_t) * (size_t) c.nbr_start[N]) == 0);
			pgv_host_hnsw_built_free(&c);
		}
		if (was)
			setenv("PGV_HNSW_HOST_LINK", was, 1);
		else
			unsetenv("PGV_HNSW_HOST_LINK");
		if (host_default && !was)
			setenv("PGV_HNSW_HOST_SELECT", "1", 1);
	}
	pgv_host_hnsw_built_free(&built);

	/* inside a server: the helpers and the OpenMP team take no signals (a handler that ran anywhere but on the calling
	 * thread is counted), and a cancel check that fires ends the build cleanly with PGV_ERR_STATE */
	{
		struct sigaction sa,
					old;
		int			rc;

		memset(&sa, 0, sizeof(sa));
		sa.sa_handler = on_sigusr1;
		sigaction(SIGUSR1, &sa, &old);
		main_thread = pthread_self();
		cancel_after = 1 << 30;
		storm_on = 1;
		{
			pthread_t	stormer;

			pthread_create(&stormer, NULL, storm_main, NULL);
			pgv_host_hnsw_set_cancel_check(cancel_cb, NULL);
			CHECK(pgv_host_hnsw_build(mirror, PGV_F32, DIM, data, N, M, EFC, NULL, 32, &built));
			EXPECT(built.nelements == N);
			pgv_host_hnsw_built_free(&built);
			cancel_after = 5;		/* the sixth poll says "cancel" */
			cancel_polls = 0;
			rc = pgv_host_hnsw_build(mirror, PGV_F32, DIM, data, N, M, EFC, NULL, 32, &built);
			EXPECT(rc == PGV_ERR_STATE && strstr(pgv_host_last_error(), "cancelled") != NULL);
			EXPECT(built.levels == NULL && built.nbr == NULL);	/* freed on the way out */
			pgv_host_hnsw_set_cancel_check(NULL, NULL);
			storm_on = 0;
			pthread_join(stormer, NULL);
		}
		EXPECT(signals_seen > 0);
		EXPECT(signals_off_main == 0);
		sigaction(SIGUSR1, &old, NULL);
	}

	/* the on-disk form and back: same graph under the slot renumbering */
	{
		pgv_rel		rel;
		pgv_hnsw_image img;
		uint64_t   *tids = malloc(sizeof(uint64_t) * N);

		for (int i = 0; i < N; i++)
			tids[i] = ((uint64_t) (i + 1) << 16) | 1;
		CHECK(pgv_host_hnsw_build(mirror, PGV_F32, DIM, data, N, M, EFC, NULL, 8, &built));
		pgv_rel_init(&rel);
		CHECK(pgv_host_hnsw_write_index(&rel, PGV_F32, DIM, M, EFC, N, data, tids, built.levels, built.nbr_start,
										built.nbr, built.dup_of, built.entry));
		CHECK(pgv_host_hnsw_stage(&rel, PGV_F32, &img));
		EXPECT(img.n == N && img.m == M && img.dim == DIM);
		EXPECT(img.entry == N - 1 - built.entry);	/* slots are page order: newest element first */
		for (int s = 0; s < N; s++)
		{
			int			e = N - 1 - s;

			EXPECT(img.levels[s] == built.levels[e] && img.heaptids[(size_t) s * 10] == tids[e]);
			EXPECT(memcmp((float *) img.vectors + (size_t) s * DIM, data + (size_t) e * DIM, sizeof(float) * DIM) == 0);
			for (int64_t j = 0; j < img.nbr_start[s + 1] - img.nbr_start[s]; j++)
			{
				int32_t		want = built.nbr[built.nbr_start[e] + j];

				EXPECT(img.nbr[img.nbr_start[s] + j] == (want < 0 ? -1 : N - 1 - want));
			}
		}
		pgv_host_hnsw_image_free(&img);
		pgv_rel_free(&rel);
		pgv_host_hnsw_built_free(&built);
		free(tids);
	}
	/* legal but large parameters (the reference allows m up to 100): a batch of 256 inserts with m = 64
	 * touches more than 32768 (owner, layer) lists, which outgrows the batch's initial list table */
	{
		const int	bigm = 64;

		CHECK(pgv_host_hnsw_build(mirror, PGV_F32, DIM, data, N, bigm, 128, NULL, 256, &built));
		EXPECT(built.nelements == N && built.m == bigm);
		for (int e = 0; e < N; e++)
			for (int lc = 0; lc <= built.levels[e]; lc++)
			{
				int			lm = lc == 0 ? 2 * bigm : bigm;
				const int32_t *mine = built.nbr + built.nbr_start[e] + (int64_t) (built.levels[e] - lc) * bigm;

				for (int i = 0; i < lm; i++)
					EXPECT(mine[i] >= -1 && mine[i] < N && mine[i] != e);
			}
		pgv_host_hnsw_built_free(&built);
	}
	ora_hnsw_free(g);
	pgv_hnsw_free(mirror);
	/* batches that run AHEAD (round 4): full batches of 64 once 1024 elements are linked -- the next batch's searches
	 * and candidate pairs are computed by a helper thread on a view of the mirror while this batch is replayed.  The
	 * graph must be as sound and as searchable as one built batch by batch, and the batch count the same. */
	{
		enum { N2 = 12000 };
		float	   *d2 = malloc(sizeof(float) * N2 * DIM);
		pgv_hnsw   *m2;
		pgv_hnsw_graph graph;
		int64_t		elem[64 * 3],
					scored[64];
		float		dd[64 * 3];
		int			found = 0;

		for (int i = 0; i < N2 * DIM; i++)
			d2[i] = (float) (urand() % 4096);
		CHECK(pgv_hnsw_upload(ctx, PGV_L2SQ, PGV_F32, DIM, d2, N2, &m2));
		CHECK(pgv_host_hnsw_build(m2, PGV_F32, DIM, d2, N2, M, EFC, NULL, 64, &built));
		EXPECT(built.nelements + 0 <= N2 && built.nelements >= N2 - 8);	/* (a few exact duplicates of 4096^8 draws at most) */
		EXPECT(built.batches >= N2 / 64 && built.batches < N2 / 64 + 400);
		for (int e = 0; e < N2; e++)
			for (int lc = 0; lc <= built.levels[e] && built.dup_of[e] < 0; lc++)
			{
				int			lm = lc == 0 ? 2 * M : M;
				const int32_t *mine = built.nbr + built.nbr_start[e] + (int64_t) (built.levels[e] - lc) * M;

				for (int i = 0; i < lm; i++)
				{
					EXPECT(mine[i] >= -1 && mine[i] < N2 && mine[i] != e);
					if (mine[i] >= 0)
						EXPECT(built.levels[mine[i]] >= lc);
				}
			}
		graph.nelements = N2;
		graph.m = M;
		graph.entry = built.entry;
		graph.levels = built.levels;
		graph.nbr_start = built.nbr_start;
		graph.nbr = built.nbr;
		CHECK(pgv_host_hnsw_search(m2, &graph, PGV_F32, DIM, d2 + 5000 * DIM, 64, 40, 3, elem, dd, scored));
		for (int q = 0; q < 64; q++)
			found += dd[q * 3] == 0.0f;
		EXPECT(found >= 62);	/* a stored vector finds itself (or its duplicate) */
		pgv_host_hnsw_built_free(&built);
		pgv_hnsw_free(m2);
		free(d2);
	}
	pgv_ctx_destroy(ctx);
	free(data);
	return 0;
}

/* ---- IVFFlat: build -> pages -> mirror -> amgettuple, iterative scan, insert, vacuum */
/* ---- the same over halfvec elements (PGV_F16): the serial build against the oracle's fp16 graph */
static int
test_hnsw_build_f16(void)
{
	enum { N = 700, DIM = 8, M = 6, EFC = 24 };
	ora_half   *data = malloc(sizeof(ora_half) * N * DIM);
	pgv_ctx    *ctx;
	pgv_hnsw   *mirror;
	pgv_hnsw_built built;
	ora_hnsw   *g;
	ora_prng	st;
	pgv_rng		rng;
	int			same = 0,
				entry_level;
	int32_t		buf[2 * M];

	/* integer coordinates < 512: exact as halves, every squared distance exact in fp32 whatever the order of the sum */
	for (int i = 0; i < N * DIM; i++)
		data[i] = ora_float_to_half((float) (urand() % 512));
	CHECK(pgv_ctx_create(0, NULL, &ctx));
	CHECK(pgv_hnsw_upload(ctx, PGV_L2SQ, PGV_F16, DIM, data, N, &mirror));
	g = ora_hnsw_build(ORA_OPS_L2, ORA_F16, DIM, data, N, M, EFC, 23);
	EXPECT(ora_hnsw_num_elements(g) == N);
	ora_prng_seed(&st, 23);
	rng.next_double = ora_prng_double_cb;
	rng.next_u32 = ora_prng_u32_cb;
	rng.state = &st;
	rng.seed = 0;
	CHECK(pgv_host_hnsw_build(mirror, PGV_F16, DIM, data, N, M, EFC, &rng, 1, &built));
	EXPECT(built.nelements == N && built.batches == N);
	EXPECT(built.entry == ora_hnsw_entry_point(g, &entry_level));
	for (int e = 0; e < N; e++)
	{
		int			ok = built.levels[e] == ora_hnsw_level(g, e);

		EXPECT(ok);
		for (int lc = 0; lc <= built.levels[e] && ok; lc++)
		{
			int			lm = lc == 0 ? 2 * M : M;
			int			cnt = ora_hnsw_neighbors(g, e, lc, buf);
			const int32_t *mine = built.nbr + built.nbr_start[e] + (int64_t) (built.levels[e] - lc) * M;

			for (int i = 0; i < lm && ok; i++)
				ok = mine[i] == (i < cnt ? buf[i] : -1);
		}
		same += ok;
	}
	EXPECT(same >= N * 99 / 100);	/* (tied candidates may be walked in another order: unspecified in the reference) */
	pgv_host_hnsw_built_free(&built);
	/* a batched build of the same elements: sound, and every element finds itself */
	CHECK(pgv_host_hnsw_build(mirror, PGV_F16, DIM, data, N, M, EFC, NULL, 32, &built));
	EXPECT(built.nelements == N && built.batches < N / 4);
	{
		pgv_hnsw_graph graph = {N, M, built.entry, built.levels, built.nbr_start, built.nbr};
		int64_t		elem[8 * 3],
					scored[8];
		float		d[8 * 3];

		CHECK(pgv_host_hnsw_search(mirror, &graph, PGV_F16, DIM, data + 200 * DIM, 8, 40, 3, elem, d, scored));
		for (int q = 0; q < 8; q++)
			EXPECT(elem[q * 3] == 200 + q && d[q * 3] == 0.0f);
	}
	pgv_host_hnsw_built_free(&built);
	ora_hnsw_free(g);
	pgv_hnsw_free(mirror);
	pgv_ctx_destroy(ctx);
	free(data);
	return 0;
}

static int		dead_set[16];
static int		ndead;
static int
is_dead(uint64_t tid, void *state)
{
	(void) state;
	for (int i = 0; i < ndead; i++)
		if ((uint64_t) dead_set[i] == tid)
			return 1;
	return 0;
}

static int
test_ivf(void)
{
	enum { N = 3000, DIM = 24, LISTS = 12, K = 8 };
	float	   *rows = malloc(sizeof(float) * N * DIM);
	uint64_t   *tids = malloc(sizeof(uint64_t) * N);
	pgv_ctx    *ctx;
	pgv_rel		rel;
	pgv_ivf_mirror *mirror;
	pgv_index  *ix;
	const pgv_ivf_image *img;
	pgv_ivf_scan *scan;
	float		q[DIM];

	for (int i = 0; i < N; i++)
	{
		for (int d = 0; d < DIM; d++)
			rows[i * DIM + d] = (float) ((i % 12) * 3 + (d % 4)) + (float) (urand() % 1000) / 2000.0f;
		tids[i] = (uint64_t) (i + 1);
	}
	CHECK(pgv_ctx_create(0, NULL, &ctx));
	pgv_rel_init(&rel);
	CHECK(pgv_host_ivf_build(ctx, PGV_OPS_L2, PGV_F32, DIM, LISTS, rows, tids, N, rows, N, NULL, &rel));
	{
		/* k-means runs on a helper thread while the rows go up: its failure (the stand-in's, on request) must come back as
		 * the build's error, text included, and leave no relation behind */
		pgv_rel		bad;
		int			rc;

		pgv_rel_init(&bad);
		setenv("MOCK_HIP_KMEANS_FAIL", "1", 1);
		rc = pgv_host_ivf_build(ctx, PGV_OPS_L2, PGV_F32, DIM, LISTS, rows, tids, N, rows, N, NULL, &bad);
		unsetenv("MOCK_HIP_KMEANS_FAIL");
		EXPECT(rc != PGV_OK && strstr(pgv_host_last_error(), "was asked to fail") != NULL);
		EXPECT(bad.pages == NULL && bad.nblocks == 0);
	}
	CHECK(pgv_host_ivf_mirror_open(ctx, PGV_L2SQ, PGV_F32, &mirror));
	CHECK(pgv_host_ivf_mirror_get(mirror, &rel, &ix, &img));
	EXPECT(img->nrows == N && img->lists == LISTS && pgv_host_ivf_mirror_restages(mirror) == 1);
	/* every row landed in the list of its nearest center (AddTupleToSort) */
	for (int l = 0; l < LISTS; l++)
		for (int64_t r = img->list_offsets[l]; r < img->list_offsets[l + 1]; r += 97)
		{
			const float *v = (const float *) img->vectors + (size_t) r * DIM;
			float		mine = l2sq(v, (const float *) img->centers + (size_t) l * DIM, DIM);

			for (int c = 0; c < LISTS; c++)
				EXPECT(l2sq(v, (const float *) img->centers + (size_t) c * DIM, DIM) >= mine);
		}

	/* probes = lists: the scan is exact; ascending; every TID once */
	for (int d = 0; d < DIM; d++)
		q[d] = rows[77 * DIM + d] + 0.01f;
	CHECK(pgv_host_ivf_beginscan(ix, img, LISTS, 0, 0, 0, &scan));
	CHECK(pgv_host_ivf_rescan(scan, q));
	{
		double		prev = -1;
		int			count = 0;
		uint64_t	tid;
		double		dist;
		float		best = INFINITY;

		for (int i = 0; i < N; i++)
			if (l2sq(rows + i * DIM, q, DIM) < best)
				best = l2sq(rows + i * DIM, q, DIM);
		while (pgv_host_ivf_gettuple(scan, &tid, &dist) == 1)
		{
			EXPECT(dist >= prev);
			if (count == 0)
				EXPECT(tid == 78 && fabs(dist - (double) best) <= 1e-5 * best);
			prev = dist;
			count++;
		}
		EXPECT(count == N);
	}
	pgv_host_ivf_endscan(scan);

	/* iterative scan (relaxed order): probes = 1, max_probes = lists -> still every tuple, in batches */
	CHECK(pgv_host_ivf_beginscan(ix, img, 1, LISTS, 1, 0, &scan));
	CHECK(pgv_host_ivf_rescan(scan, q));
	{
		int			count = 0;
		uint64_t	tid;
		double		dist;

		while (pgv_host_ivf_gettuple(scan, &tid, &dist) == 1)
			count++;
		EXPECT(count == N);
	}
	/* a NULL query: every tuple of the probed lists at distance 0 (ZeroDistance, src/ivfscan.c:192-196) */
	CHECK(pgv_host_ivf_rescan(scan, NULL));
	{
		uint64_t	tid;
		double		dist;

		EXPECT(pgv_host_ivf_gettuple(scan, &tid, &dist) == 1 && dist == 0.0);
	}
	pgv_host_ivf_endscan(scan);

	/* an opclass with a NORM_PROC (cosine): the scan normalises the query (src/ivfscan.c:222-229) and ranks by
	 * negative inner product; with unit-length stored rows that is the cosine order */
	{
		pgv_index  *ipx;
		float	   *unit = malloc(sizeof(float) * (size_t) img->nrows * DIM);
		float	   *ucen = malloc(sizeof(float) * LISTS * DIM);
		float		big[DIM];
		uint64_t	tid;
		double		dist,
					best = -2.0;
		int64_t		best_row = -1;

		for (int64_t r = 0; r < img->nrows; r++)
		{
			double		nrm = 0;

			for (int d = 0; d < DIM; d++)
				nrm += (double) ((const float *) img->vectors)[r * DIM + d] * ((const float *) img->vectors)[r * DIM + d];
			for (int d = 0; d < DIM; d++)
				unit[r * DIM + d] = (float) (((const float *) img->vectors)[r * DIM + d] / sqrt(nrm));
		}
		for (int l = 0; l < LISTS; l++)
		{
			double		nrm = 0;

			for (int d = 0; d < DIM; d++)
				nrm += (double) ((const float *) img->centers)[l * DIM + d] * ((const float *) img->centers)[l * DIM + d];
			for (int d = 0; d < DIM; d++)
				ucen[l * DIM + d] = (float) (((const float *) img->centers)[l * DIM + d] / sqrt(nrm));
		}
		CHECK(pgv_index_upload(ctx, PGV_NEG_IP, PGV_F32, DIM, LISTS, ucen, img->list_offsets, unit, img->tids, &ipx));
		for (int d = 0; d < DIM; d++)
			big[d] = 1000.0f * q[d];	/* any positive multiple of q has the same cosine order */
		CHECK(pgv_host_ivf_beginscan(ipx, img, LISTS, 0, 0, 1, &scan));
		CHECK(pgv_host_ivf_rescan(scan, big));
		EXPECT(pgv_host_ivf_gettuple(scan, &tid, &dist) == 1);
		{
			double		qn = 0;

			for (int d = 0; d < DIM; d++)
				qn += (double) q[d] * q[d];
			for (int64_t r = 0; r < img->nrows; r++)
			{
				double		ip = 0;

				for (int d = 0; d < DIM; d++)
					ip += (double) unit[r * DIM + d] * q[d];
				ip /= sqrt(qn);
				if (ip > best)
				{
					best = ip;
					best_row = r;
				}
			}
		}
		EXPECT(tid == img->tids[best_row] && fabs(dist + best) <= 1e-4);
		pgv_host_ivf_endscan(scan);
		pgv_index_free(ipx);
		free(unit);
		free(ucen);
	}

	/* insert + vacuum invalidate the mirror; an unchanged relation does not */
	CHECK(pgv_host_ivf_mirror_get(mirror, &rel, &ix, &img));
	EXPECT(pgv_host_ivf_mirror_restages(mirror) == 1);
	{
		int32_t		lst;

		CHECK(pgv_assign(ctx, PGV_L2SQ, PGV_F32, DIM, img->centers, LISTS, q, 1, &lst, NULL));
		CHECK(pgv_host_ivf_insert(&rel, PGV_F32, lst, q, 999999));
	}
	CHECK(pgv_host_ivf_mirror_get(mirror, &rel, &ix, &img));
	EXPECT(pgv_host_ivf_mirror_restages(mirror) == 2 && img->nrows == N + 1);
	CHECK(pgv_host_ivf_beginscan(ix, img, 3, 0, 0, 0, &scan));
	CHECK(pgv_host_ivf_rescan(scan, q));
	{
		uint64_t	tid;
		double		dist;

		EXPECT(pgv_host_ivf_gettuple(scan, &tid, &dist) == 1 && tid == 999999 && dist == 0.0);
		EXPECT(pgv_host_ivf_gettuple(scan, &tid, &dist) == 1 && tid == 78);
	}
	pgv_host_ivf_endscan(scan);
	{
		int64_t		removed,
					remaining;

		dead_set[0] = 999999;
		dead_set[1] = 78;
		ndead = 2;
		CHECK(pgv_host_ivf_bulkdelete(&rel, is_dead, NULL, &removed, &remaining));
		EXPECT(removed == 2 && remaining == N - 1);
	}
	CHECK(pgv_host_ivf_mirror_get(mirror, &rel, &ix, &img));
	EXPECT(pgv_host_ivf_mirror_restages(mirror) == 3 && img->nrows == N - 1);
	CHECK(pgv_host_ivf_beginscan(ix, img, 3, 0, 0, 0, &scan));
	CHECK(pgv_host_ivf_rescan(scan, q));
	{
		uint64_t	tid;
		double		dist;

		EXPECT(pgv_host_ivf_gettuple(scan, &tid, &dist) == 1 && tid != 999999 && tid != 78 && dist > 0.0);
	}
	pgv_host_ivf_endscan(scan);
	pgv_host_ivf_mirror_close(mirror);
	pgv_rel_free(&rel);
	pgv_ctx_destroy(ctx);
	free(rows);
	free(tids);
	return 0;
}

/* ------------------------------------------------------------------ the pooler (ivf_pool.c) under threads */
#include <pthread.h>

typedef struct
{
	pgv_pool   *pool;
	const float *queries;
	int			dim,
				nq,
				k,
				id,
				nthreads;
	uint64_t   *tid;			/* [nq x k] */
	float	   *dist;
	int			rc;
}			pool_client;

static void *
pool_client_main(void *arg)
{
	pool_client *c = arg;

	for (int round = 0; round < 3 && c->rc == PGV_OK; round++)
		for (int j = c->id; j < c->nq && c->rc == PGV_OK; j += c->nthreads)
			c->rc = pgv_host_pool_search(c->pool, c->queries + (size_t) j * c->dim, c->tid + (size_t) j * c->k,
										 c->dist + (size_t) j * c->k);
	return NULL;
}

extern int	mock_hip_search_delay_us;
extern int	mock_hip_search_peak(int reset);

static int
test_pool(void)
{
	enum
	{
		N = 600, DIM = 8, LISTS = 6, NQ = 96, K = 5, THREADS = 12
	};
	float	   *rows = malloc(sizeof(float) * N * DIM),
			   *centers = malloc(sizeof(float) * LISTS * DIM),
			   *queries = malloc(sizeof(float) * NQ * DIM);
	int64_t		off[LISTS + 1];
	uint64_t   *tids = malloc(sizeof(uint64_t) * N);
	uint64_t   *want_t = malloc(sizeof(uint64_t) * NQ * K),
			   *got_t = malloc(sizeof(uint64_t) * NQ * K);
	float	   *want_d = malloc(sizeof(float) * NQ * K),
			   *got_d = malloc(sizeof(float) * NQ * K);
	pgv_ctx    *ctx;
	pgv_index  *ix;
	pgv_pool   *pool;
	pthread_t	th[THREADS];
	pool_client cl[THREADS];
	int64_t		batches,
				nqueries;
	unsigned	seed = 12345;

	for (int i = 0; i < N * DIM; i++)
		rows[i] = (float) (rand_r(&seed) % 1000) / 100.0f;
	for (int i = 0; i < LISTS * DIM; i++)
		centers[i] = (float) (rand_r(&seed) % 1000) / 100.0f;
	for (int i = 0; i < NQ * DIM; i++)
		queries[i] = (float) (rand_r(&seed) % 1000) / 100.0f;
	for (int l = 0; l <= LISTS; l++)
		off[l] = (int64_t) l * (N / LISTS);
	for (int i = 0; i < N; i++)
		tids[i] = 7000 + (uint64_t) i;
	CHECK(pgv_ctx_create(0, NULL, &ctx));
	CHECK(pgv_index_upload(ctx, PGV_L2SQ, PGV_F32, DIM, LISTS, centers, off, rows, tids, &ix));
	CHECK(pgv_search_batch(ix, queries, NQ, 2, K, want_d, NULL, want_t));
	/* argument checks: nothing is created, the error is reported */
	EXPECT(pgv_host_pool_create(NULL, 0, PGV_F32, DIM, 2, K, 5, 300, 2, &pool) != PGV_OK);
	EXPECT(pgv_host_pool_create(ix, 0, PGV_F32, DIM, 2, K, 0, 300, 2, &pool) != PGV_OK && pool == NULL);
	EXPECT(pgv_host_pool_create(ix, 0, PGV_F32, DIM, 2, K, 5, 300, 9, &pool) != PGV_OK && pool == NULL);
	EXPECT(pgv_host_pool_search(NULL, queries, got_t, got_d) != PGV_OK);
	/* small batches, short wait, two lanes: leaders, followers, full and timed-out batches, lanes handed on */
	CHECK(pgv_host_pool_create(ix, 0, PGV_F32, DIM, 2, K, 5, 300, 2, &pool));
	for (int t = 0; t < THREADS; t++)
	{
		cl[t] = (pool_client) {pool, queries, DIM, NQ, K, t, THREADS, got_t, got_d, PGV_OK};
		pthread_create(&th[t], NULL, pool_client_main, &cl[t]);
	}
	for (int t = 0; t < THREADS; t++)
	{
		pthread_join(th[t], NULL);
		EXPECT(cl[t].rc == PGV_OK);
	}
	pgv_host_pool_stats(pool, &batches, &nqueries);
	EXPECT(nqueries == 3 * NQ && batches >= 3 * NQ / 5 && batches < 3 * NQ);
	EXPECT(memcmp(got_t, want_t, sizeof(uint64_t) * NQ * K) == 0);
	EXPECT(memcmp(got_d, want_d, sizeof(float) * NQ * K) == 0);
	/* one backend alone: a batch of one after the wait */
	CHECK(pgv_host_pool_search(pool, queries, got_t, got_d));
	EXPECT(memcmp(got_t, want_t, sizeof(uint64_t) * K) == 0);
	/* ONE SCAN AT A TIME (round 5): three lanes, scans stretched to 2 ms, twelve clients -- the lanes' pgv_search_batch
	 * calls never overlap, the batches that collect during a scan are larger for it, answers unchanged; with
	 * PGV_POOL_OVERLAP=1 (round 4's policy, read when the segment is initialised) the lanes do scan side by side */
	for (int mode = 0; mode < 2; mode++)
	{
		pgv_pool   *p3;
		int64_t		b3,
					q3;

		if (mode == 1)
			setenv("PGV_POOL_OVERLAP", "1", 1);
		/* (mode 1: batches of four at most, so that twelve clients NEED three lanes at once -- with 64 they could fall
		 * into step and share one batch after another, and no two scans would ever be in flight) */
		CHECK(pgv_host_pool_create(ix, 0, PGV_F32, DIM, 2, K, mode == 1 ? 4 : 64, 100, 3, &p3));
		unsetenv("PGV_POOL_OVERLAP");
		mock_hip_search_delay_us = 2000;
		(void) mock_hip_search_peak(1);
		memset(got_t, 0, sizeof(uint64_t) * NQ * K);
		for (int t = 0; t < THREADS; t++)
		{
			cl[t] = (pool_client) {p3, queries, DIM, NQ, K, t, THREADS, got_t, got_d, PGV_OK};
			pthread_create(&th[t], NULL, pool_client_main, &cl[t]);
		}
		for (int t = 0; t < THREADS; t++)
		{
			pthread_join(th[t], NULL);
			EXPECT(cl[t].rc == PGV_OK);
		}
		mock_hip_search_delay_us = 0;
		pgv_host_pool_stats(p3, &b3, &q3);
		EXPECT(q3 == 3 * NQ);
		EXPECT(memcmp(got_t, want_t, sizeof(uint64_t) * NQ * K) == 0);
		if (mode == 0)
		{
			EXPECT(mock_hip_search_peak(1) == 1);
			EXPECT(b3 <= 3 * NQ / 4);	/* twelve closed-loop clients in two alternating groups: ~6 queries a batch */
		}
		else
			EXPECT(mock_hip_search_peak(1) >= 2);
		pgv_host_pool_destroy(p3);
	}
	/* the views keep the arrays alive: drop the uploaded handle first */
	pgv_index_free(ix);
	CHECK(pgv_host_pool_search(pool, queries + DIM, got_t, got_d));
	EXPECT(memcmp(got_t, want_t + K, sizeof(uint64_t) * K) == 0);
	pgv_host_pool_destroy(pool);
	pgv_ctx_destroy(ctx);
	free(rows);
	free(centers);
	free(queries);
	free(tids);
	free(want_t);
	free(got_t);
	free(want_d);
	free(got_d);
	return 0;
}

/* ------------------------------------------------------- the pooler with backends that are PROCESSES
 * The owner formats a shared segment and publishes its mirror's handle; two forked processes import the mirror and
 * serve one lane each (the background workers' role); eight forked clients -- which never call the device library --
 * hand in one query at a time; every answer is pgv_search_batch's row. */
#include <sys/mman.h>
#include <sys/wait.h>

static int
test_pool_processes(void)
{
	enum
	{
		N = 600, DIM = 8, LISTS = 6, NQ = 96, K = 5, CLIENTS = 8, LANES = 2
	};
	float	   *rows = malloc(sizeof(float) * N * DIM),
			   *centers = malloc(sizeof(float) * LISTS * DIM),
			   *queries = malloc(sizeof(float) * NQ * DIM);
	int64_t		off[LISTS + 1];
	uint64_t   *tids = malloc(sizeof(uint64_t) * N);
	uint64_t   *want_t = malloc(sizeof(uint64_t) * NQ * K);
	float	   *want_d = malloc(sizeof(float) * NQ * K);
	size_t		shm_bytes = pgv_host_pool_shm_bytes(PGV_F32, DIM, K, 5, LANES);
	size_t		res_bytes = (sizeof(uint64_t) + sizeof(float)) * NQ * K + sizeof(int) * (CLIENTS + LANES);
	void	   *shm = mmap(NULL, shm_bytes, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
	char	   *res = mmap(NULL, res_bytes, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
	uint64_t   *got_t = (uint64_t *) res;
	float	   *got_d = (float *) (res + sizeof(uint64_t) * NQ * K);
	int		   *rcs = (int *) (res + (sizeof(uint64_t) + sizeof(float)) * NQ * K);
	pgv_ctx    *ctx;
	pgv_index  *ix,
			   *self;
	pgv_pool   *pool;
	pgv_index_handle handle;
	pid_t		servers[LANES],
				clients[CLIENTS];
	int64_t		batches,
				nqueries;
	unsigned	seed = 4321;

	EXPECT(shm != MAP_FAILED && res != MAP_FAILED && shm_bytes > 0);
	for (int i = 0; i < N * DIM; i++)
		rows[i] = (float) (rand_r(&seed) % 1000) / 100.0f;
	for (int i = 0; i < LISTS * DIM; i++)
		centers[i] = (float) (rand_r(&seed) % 1000) / 100.0f;
	for (int i = 0; i < NQ * DIM; i++)
		queries[i] = (float) (rand_r(&seed) % 1000) / 100.0f;
	for (int l = 0; l <= LISTS; l++)
		off[l] = (int64_t) l * (N / LISTS);
	for (int i = 0; i < N; i++)
		tids[i] = 9000 + (uint64_t) i;
	CHECK(pgv_ctx_create(0, NULL, &ctx));
	CHECK(pgv_index_upload(ctx, PGV_L2SQ, PGV_F32, DIM, LISTS, centers, off, rows, tids, &ix));
	CHECK(pgv_search_batch(ix, queries, NQ, 2, K, want_d, NULL, want_t));
	/* segment checks */
	EXPECT(pgv_host_pool_shm_bytes(PGV_F32, DIM, K, 5, 9) == 0);
	EXPECT(pgv_host_pool_attach(shm, shm_bytes, &pool) != PGV_OK);	/* not formatted yet */
	EXPECT(pgv_host_pool_shm_init(shm, shm_bytes - 1, PGV_F32, DIM, 2, K, 5, 300, LANES) != PGV_OK);
	CHECK(pgv_host_pool_shm_init(shm, shm_bytes, PGV_F32, DIM, 2, K, 5, 300, LANES));
	CHECK(pgv_host_pool_attach(shm, shm_bytes, &pool));
	EXPECT(pgv_host_pool_index_handle(pool, 1, &handle) != PGV_OK);	/* nothing published yet */
	CHECK(pgv_index_export(ix, &handle));
	EXPECT(pgv_index_import(ctx, &handle, &self) != PGV_OK);	/* the exporter itself shares, it does not import */
	CHECK(pgv_host_pool_publish_index(pool, &handle));
	memset(res, 0xff, res_bytes);
	for (int s = 0; s < LANES; s++)
	{
		servers[s] = fork();
		if (servers[s] == 0)
		{
			pgv_ctx    *sctx;
			pgv_index  *view;
			pgv_pool   *mine;
			pgv_index_handle h;
			int			rc = pgv_ctx_create(0, NULL, &sctx);

			if (rc == PGV_OK)
				rc = pgv_host_pool_attach(shm, shm_bytes, &mine);
			if (rc == PGV_OK)
				rc = pgv_host_pool_index_handle(mine, 1000, &h);
			if (rc == PGV_OK)
				rc = pgv_index_import(sctx, &h, &view);
			if (rc == PGV_OK)
				rc = pgv_host_pool_serve(mine, s, view);
			rcs[CLIENTS + s] = rc;
			_exit(rc == PGV_OK ? 0 : 3);
		}
	}
	for (int c = 0; c < CLIENTS; c++)
	{
		clients[c] = fork();
		if (clients[c] == 0)
		{
			pgv_pool   *mine;
			int			rc = pgv_host_pool_attach(shm, shm_bytes, &mine);

			for (int round = 0; round < 3 && rc == PGV_OK; round++)
				for (int j = c; j < NQ && rc == PGV_OK; j += CLIENTS)
					rc = pgv_host_pool_search(mine, queries + (size_t) j * DIM, got_t + (size_t) j * K, got_d + (size_t) j * K);
			rcs[c] = rc;
			_exit(rc == PGV_OK ? 0 : 4);
		}
	}
	for (int c = 0; c < CLIENTS; c++)
	{
		int			st = -1;

		EXPECT(waitpid(clients[c], &st, 0) == clients[c] && WIFEXITED(st) && WEXITSTATUS(st) == 0 && rcs[c] == PGV_OK);
	}
	pgv_host_pool_stats(pool, &batches, &nqueries);
	EXPECT(nqueries == 3 * NQ && batches >= 3 * NQ / 5 && batches <= 3 * NQ);
	EXPECT(memcmp(got_t, want_t, sizeof(uint64_t) * NQ * K) == 0);
	EXPECT(memcmp(got_d, want_d, sizeof(float) * NQ * K) == 0);
	/* the owner is a client too */
	CHECK(pgv_host_pool_search(pool, queries + 2 * DIM, got_t, got_d));
	EXPECT(memcmp(got_t, want_t + 2 * K, sizeof(uint64_t) * K) == 0);
	pgv_host_pool_shutdown(pool);
	for (int s = 0; s < LANES; s++)
	{
		int			st = -1;

		EXPECT(waitpid(servers[s], &st, 0) == servers[s] && WIFEXITED(st) && WEXITSTATUS(st) == 0 && rcs[CLIENTS + s] == PGV_OK);
	}
	EXPECT(pgv_host_pool_search(pool, queries, got_t, got_d) != PGV_OK);	/* shut down */
	pgv_host_pool_detach(pool);
	pgv_index_free(ix);
	pgv_ctx_destroy(ctx);
	munmap(shm, shm_bytes);
	munmap(res, res_bytes);
	free(rows);
	free(centers);
	free(queries);
	free(tids);
	free(want_t);
	free(want_d);
	return 0;
}

/* ---- the same drivers over halfvec rows (PGV_F16: 2-byte elements in the tuples, the list pages, the image and the
 * stand-in device's door): build -> pages -> stage -> scans (exact, iterative, cosine-style normalised query) -> insert
 * -> vacuum.  Values are multiples of 1/8 below 64: exact as halves, so a float brute force is the expectation. */
static int
test_ivf_f16(void)
{
	enum { N = 2500, DIM = 40, LISTS = 10 };	/* 8 + 80 bytes a value: under 127, so the index tuples carry SHORT varlena headers */
	float	   *frows = malloc(sizeof(float) * N * DIM);
	ora_half   *rows = malloc(sizeof(ora_half) * N * DIM);
	uint64_t   *tids = malloc(sizeof(uint64_t) * N);
	pgv_ctx    *ctx;
	pgv_rel		rel;
	pgv_ivf_mirror *mirror;
	pgv_index  *ix;
	const pgv_ivf_image *img;
	pgv_ivf_scan *scan;
	float		fq[DIM];
	ora_half	q[DIM];

	for (int i = 0; i < N; i++)
	{
		for (int d = 0; d < DIM; d++)
		{
			frows[i * DIM + d] = (float) ((i % 10) * 5 + (d % 3)) + (float) (urand() % 16) / 8.0f;
			rows[i * DIM + d] = ora_float_to_half(frows[i * DIM + d]);
			EXPECT(ora_half_to_float(rows[i * DIM + d]) == frows[i * DIM + d]);
		}
		tids[i] = (uint64_t) (i + 1);
	}
	CHECK(pgv_ctx_create(0, NULL, &ctx));
	pgv_rel_init(&rel);
	CHECK(pgv_host_ivf_build(ctx, PGV_OPS_L2, PGV_F16, DIM, LISTS, rows, tids, N, rows, N, NULL, &rel));
	CHECK(pgv_host_ivf_mirror_open(ctx, PGV_L2SQ, PGV_F16, &mirror));
	CHECK(pgv_host_ivf_mirror_get(mirror, &rel, &ix, &img));
	EXPECT(img->nrows == N && img->lists == LISTS && img->dim == DIM && img->dtype == PGV_F16);
	/* the image holds the rows' halves bit for bit, each in the list of its nearest center */
	{
		char	   *seen = calloc(N + 1, 1);

		for (int l = 0; l < LISTS; l++)
			for (int64_t r = img->list_offsets[l]; r < img->list_offsets[l + 1]; r++)
			{
				const ora_half *v = (const ora_half *) img->vectors + (size_t) r * DIM;
				const int	row = (int) img->tids[r] - 1;
				float		fv[DIM],
							fc[DIM],
							mine;

				EXPECT(row >= 0 && row < N && !seen[row]);
				seen[row] = 1;
				EXPECT(memcmp(v, rows + (size_t) row * DIM, sizeof(ora_half) * DIM) == 0);
				if (r % 53 != 0)
					continue;
				for (int d = 0; d < DIM; d++)
				{
					fv[d] = ora_half_to_float(v[d]);
					fc[d] = ora_half_to_float(((const ora_half *) img->centers)[(size_t) l * DIM + d]);
				}
				mine = l2sq(fv, fc, DIM);
				for (int c = 0; c < LISTS; c++)
				{
					for (int d = 0; d < DIM; d++)
						fc[d] = ora_half_to_float(((const ora_half *) img->centers)[(size_t) c * DIM + d]);
					EXPECT(l2sq(fv, fc, DIM) >= mine * (1.0f - 1e-5f));
				}
			}
		free(seen);
	}
	/* probes = lists: exact, ascending, every TID once; the nearest is row 321 itself */
	for (int d = 0; d < DIM; d++)
	{
		q[d] = rows[321 * DIM + d];
		fq[d] = frows[321 * DIM + d];
	}
	CHECK(pgv_host_ivf_beginscan(ix, img, LISTS, 0, 0, 0, &scan));
	CHECK(pgv_host_ivf_rescan(scan, q));
	{
		double		prev = -1;
		int			count = 0;
		uint64_t	tid;
		double		dist;

		while (pgv_host_ivf_gettuple(scan, &tid, &dist) == 1)
		{
			EXPECT(dist >= prev);
			EXPECT(tid >= 1 && tid <= N);
			EXPECT(fabs(dist - (double) l2sq(frows + (tid - 1) * DIM, fq, DIM)) <= 1e-4 * (1.0 + dist));
			if (count == 0)
				EXPECT(dist == 0.0);
			prev = dist;
			count++;
		}
		EXPECT(count == N);
	}
	pgv_host_ivf_endscan(scan);
	/* iterative: one probe at a time up to all lists */
	CHECK(pgv_host_ivf_beginscan(ix, img, 1, LISTS, 1, 0, &scan));
	CHECK(pgv_host_ivf_rescan(scan, q));
	{
		int			count = 0;
		uint64_t	tid;
		double		dist;

		while (pgv_host_ivf_gettuple(scan, &tid, &dist) == 1)
			count++;
		EXPECT(count == N);
	}
	pgv_host_ivf_endscan(scan);
	/* insert (one more half row into its nearest list), restage, find it; vacuum it away again */
	{
		int32_t		lst;
		ora_half	nv[DIM];
		uint64_t	tid;
		double		dist;
		int64_t		removed = 0,
					remaining = 0;

		for (int d = 0; d < DIM; d++)
			nv[d] = ora_float_to_half(frows[1000 * DIM + d] + 0.125f);
		CHECK(pgv_assign(ctx, PGV_L2SQ, PGV_F16, DIM, img->centers, LISTS, nv, 1, &lst, NULL));
		CHECK(pgv_host_ivf_insert(&rel, PGV_F16, lst, nv, 777777));
		CHECK(pgv_host_ivf_mirror_get(mirror, &rel, &ix, &img));
		EXPECT(img->nrows == N + 1 && pgv_host_ivf_mirror_restages(mirror) == 2);
		CHECK(pgv_host_ivf_beginscan(ix, img, 3, 0, 0, 0, &scan));
		CHECK(pgv_host_ivf_rescan(scan, nv));
		EXPECT(pgv_host_ivf_gettuple(scan, &tid, &dist) == 1 && tid == 777777 && dist == 0.0);
		pgv_host_ivf_endscan(scan);
		ndead = 1;
		dead_set[0] = 777777;
		CHECK(pgv_host_ivf_bulkdelete(&rel, is_dead, NULL, &removed, &remaining));
		EXPECT(removed == 1 && remaining == N);
		CHECK(pgv_host_ivf_mirror_get(mirror, &rel, &ix, &img));
		EXPECT(img->nrows == N);
		CHECK(pgv_host_ivf_beginscan(ix, img, 3, 0, 0, 0, &scan));
		CHECK(pgv_host_ivf_rescan(scan, nv));
		EXPECT(pgv_host_ivf_gettuple(scan, &tid, &dist) == 1 && tid != 777777 && dist > 0.0);
		pgv_host_ivf_endscan(scan);
	}
	pgv_host_ivf_mirror_close(mirror);
	pgv_rel_free(&rel);
	pgv_ctx_destroy(ctx);
	free(frows);
	free(rows);
	free(tids);
	return 0;
}

int
main(void)
{
	/* first: it forks, and the later tests leave OpenMP worker threads behind */
	if (test_pool_processes())
		return 1;
	if (test_hnsw_build())
		return 1;
	if (test_hnsw_build_f16())
		return 1;
	if (test_ivf())
		return 1;
	if (test_ivf_f16())
		return 1;
	if (test_pool())
		return 1;
	printf("HOST-LOGIC OK\n");
	return 0;
}
