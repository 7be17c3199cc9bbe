/*
 * A C caller of the drop-in boundary, no Python and no torch in the process: the way the
 * pgvector extension itself would use libpgv_hip.so / libpgv_host.so.  Builds an IVFFlat
 * index from synthetic heap rows (BuildIndex, src/ivfbuild.c:1040-1058), stages its pages,
 * scans it (ivfflatbeginscan/gettuple, src/ivfscan.c:252-431) and checks every answer
 * against a brute-force scan written out right here in plain C; then the batched entry
 * points and the on-device HNSW search on a hand-made graph.
 *
 * Test infrastructure (compiled and run by tests/test_gpu_parity.py::test_c_driver).
 * Exit status 0 and "C-DRIVER OK" on success.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pgv_hip.h"
#include "pgv_host.h"

#define CHECK(call) do { int rc_ = (call); if (rc_ != PGV_OK) { \
	fprintf(stderr, "%s:%d: %s -> %d: %s / %s\n", __FILE__, __LINE__, #call, rc_, pgv_last_error(), pgv_host_last_error()); \
	return 1; } } while (0)
#define EXPECT(cond) do { if (!(cond)) { fprintf(stderr, "%s:%d: EXPECT(%s) failed\n", __FILE__, __LINE__, #cond); return 1; } } while (0)

static uint64_t lcg = 0x9E3779B97F4A7C15ull;
static float
frand(void)
{
	lcg = lcg * 6364136223846793005ull + 1442695040888963407ull;
	return (float) ((lcg >> 40) & 0xFFFF) / 65536.0f;
}

/* FUNCTION 1 of vector_l2_ops the plain way (src/vector.c:560-574) */
static float
l2sq(const float *a, const float *b, int dim)
{
	float		s = 0.0f;

	for (int i = 0; i < dim; i++)
		s += (a[i] - b[i]) * (a[i] - b[i]);
	return s;
}

static int
close_enough(double got, double want)
{
	return fabs(got - want) <= 1e-5 * fabs(want) + 1e-6;
}

int
main(void)
{
	enum { N = 6000, DIM = 96, LISTS = 24, NQ = 40, K = 10 };
	float	   *rows = malloc(sizeof(float) * N * DIM);
	float	   *queries = malloc(sizeof(float) * NQ * DIM);
	uint64_t   *tids = malloc(sizeof(uint64_t) * N);
	pgv_ctx    *ctx;
	pgv_rel		rel;
	pgv_ivf_image img;
	pgv_index  *mirror;
	pgv_rng		rng = {NULL, NULL, NULL, 42};

	if (pgv_device_count() < 1)
	{
		fprintf(stderr, "no GPU: %s\n", pgv_last_error());
		return 2;
	}
	for (int i = 0; i < N; i++)
	{
		int			c = i % 12;

		for (int d = 0; d < DIM; d++)
			rows[i * DIM + d] = (float) ((c * 7 + d) % 5) + 0.25f * frand();
		tids[i] = ((uint64_t) (i / 50) << 16) | (uint64_t) (i % 50 + 1);	/* (block, offset) */
	}
	for (int i = 0; i < NQ * DIM; i++)
		queries[i] = (float) ((i * 3) % 5) + 0.25f * frand();

	CHECK(pgv_ctx_create(0, NULL, &ctx));

	/* CREATE INDEX ... USING ivfflat (v vector_l2_ops) WITH (lists = 24): samples = all rows */
	pgv_rel_init(&rel);
	CHECK(pgv_host_ivf_build(ctx, PGV_OPS_L2, PGV_F32, DIM, LISTS, rows, tids, N, rows, N, &rng, &rel));
	CHECK(pgv_host_ivf_stage(&rel, PGV_F32, &img));
	EXPECT(img.nrows == N && img.lists == LISTS && img.dim == DIM);
	CHECK(pgv_index_upload(ctx, PGV_L2SQ, PGV_F32, DIM, LISTS, img.centers, img.list_offsets, img.vectors,
						   img.tids, &mirror));
	EXPECT(pgv_index_rows(mirror) == N && pgv_index_lists(mirror) == LISTS);

	/* SET ivfflat.probes = 24: an exact scan; ORDER BY v <-> q LIMIT 10 through amgettuple */
	{
		pgv_ivf_scan *scan;

		CHECK(pgv_host_ivf_beginscan(mirror, &img, LISTS, 0, 0, 0, &scan));
		for (int q = 0; q < NQ; q++)
		{
			const float *qv = queries + q * DIM;
			double		prev = -1.0;
			float		best[K];

			/* brute force: the K smallest distances */
			for (int j = 0; j < K; j++)
				best[j] = INFINITY;
			for (int i = 0; i < N; i++)
			{
				float		d = l2sq(rows + i * DIM, qv, DIM);

				if (d < best[K - 1])
				{
					int			j = K - 1;

					while (j > 0 && best[j - 1] > d)
					{
						best[j] = best[j - 1];
						j--;
					}
					best[j] = d;
				}
			}
			CHECK(pgv_host_ivf_rescan(scan, qv));
			for (int j = 0; j < K; j++)
			{
				uint64_t	tid;
				double		dist;
				int			i;

				EXPECT(pgv_host_ivf_gettuple(scan, &tid, &dist) == 1);
				EXPECT(dist >= prev);
				prev = dist;
				EXPECT(close_enough(dist, (double) best[j]));
				/* the TID names the heap row at that distance */
				i = (int) (tid >> 16) * 50 + (int) (tid & 0xFFFF) - 1;
				EXPECT(i >= 0 && i < N);
				EXPECT(close_enough(dist, (double) l2sq(rows + i * DIM, qv, DIM)));
			}
		}
		pgv_host_ivf_endscan(scan);
	}

	/* the batched entry point: all queries at once, probes = lists */
	{
		float		dist[NQ * K];
		int64_t		slot[NQ * K];
		uint64_t	tid[NQ * K];
		const float *stored = (const float *) img.vectors;

		CHECK(pgv_search_batch(mirror, queries, NQ, LISTS, K, dist, slot, tid));
		for (int q = 0; q < NQ; q++)
			for (int j = 0; j < K; j++)
			{
				int64_t		s = slot[q * K + j];

				EXPECT(s >= 0 && s < N && tid[q * K + j] == img.tids[s]);
				EXPECT(close_enough(dist[q * K + j], l2sq(stored + s * DIM, queries + q * DIM, DIM)));
				EXPECT(j == 0 || dist[q * K + j] >= dist[q * K + j - 1]);
			}
	}

	/* one fmgr call per heap row, batched (src/vector.c:579-589) */
	{
		float	   *out = malloc(sizeof(float) * N);

		CHECK(pgv_distance_batch(ctx, PGV_L2SQ, PGV_F32, DIM, queries, rows, N, out));
		for (int i = 0; i < N; i += 97)
			EXPECT(close_enough(out[i], l2sq(rows + i * DIM, queries, DIM)));
		free(out);
	}

	/* HNSW: a ring of 64 elements, each linked to its 4 nearest ring neighbours on layer 0;
	 * element 0 also lives on layer 1 (alone).  ef_search = 64 visits the whole ring. */
	{
		enum { HN = 64, M = 4 };
		pgv_hnsw   *h;
		int32_t		levels[HN];
		int64_t		nbr_start[HN + 1];
		int32_t    *nbr;
		int64_t		elem[NQ * K],
					scored[NQ];
		float		dist[NQ * K];
		int64_t		off = 0;

		for (int i = 0; i < HN; i++)
		{
			levels[i] = i == 0 ? 1 : 0;
			nbr_start[i] = off;
			off += (levels[i] + 2) * M;
		}
		nbr_start[HN] = off;
		nbr = malloc(sizeof(int32_t) * off);
		for (int64_t i = 0; i < off; i++)
			nbr[i] = -1;
		for (int i = 0; i < HN; i++)
		{
			int32_t    *l0 = nbr + nbr_start[i] + levels[i] * M;	/* layer lc at (level - lc) * m */

			l0[0] = (i + 1) % HN;
			l0[1] = (i + HN - 1) % HN;
			l0[2] = (i + 2) % HN;
			l0[3] = (i + HN - 2) % HN;
		}
		CHECK(pgv_hnsw_upload(ctx, PGV_L2SQ, PGV_F32, DIM, rows, HN, &h));
		CHECK(pgv_hnsw_set_graph(h, M, 0, levels, nbr_start, nbr));
		CHECK(pgv_hnsw_search(h, queries, NQ, HN, K, elem, dist, scored));
		for (int q = 0; q < NQ; q++)
		{
			float		best = INFINITY;

			EXPECT(scored[q] == HN);	/* connected graph, ef = n: everything is scored once */
			for (int i = 0; i < HN; i++)
			{
				float		d = l2sq(rows + i * DIM, queries + q * DIM, DIM);

				if (d < best)
					best = d;
			}
			EXPECT(close_enough(dist[q * K], best));
			for (int j = 0; j < K; j++)
			{
				EXPECT(elem[q * K + j] >= 0 && elem[q * K + j] < HN);
				EXPECT(close_enough(dist[q * K + j], l2sq(rows + elem[q * K + j] * DIM, queries + q * DIM, DIM)));
				EXPECT(j == 0 || dist[q * K + j] >= dist[q * K + j - 1]);
			}
		}
		pgv_hnsw_free(h);
		free(nbr);
	}

	/* error behaviour at the boundary: status + message, no abort */
	EXPECT(pgv_search_batch(mirror, queries, NQ, 0, K, NULL, NULL, NULL) != PGV_OK);
	EXPECT(strlen(pgv_last_error()) > 0);

	pgv_index_free(mirror);
	pgv_host_ivf_image_free(&img);
	pgv_rel_free(&rel);
	pgv_ctx_destroy(ctx);
	free(rows);
	free(queries);
	free(tids);
	printf("C-DRIVER OK\n");
	return 0;
}
