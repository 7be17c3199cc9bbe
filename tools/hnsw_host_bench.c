/*
 * tools/hnsw_host_bench.c -- pgv_host_hnsw_build against whatever implements the ABI it is linked with.  Linked with
 * the stand-in of tests/c/mock_hip.c it shows where the HOST time of the build goes (phase seconds), and
 * tests/test_hnsw_build_tsan_cpu.py runs it under ThreadSanitizer.
 *
 *   gcc -O2 -rdynamic -I include -I pgvector_amd/host tools/hnsw_host_bench.c tests/c/mock_hip.c \
 *       -o hnsw_host_bench -L pgvector_amd/lib -lpgv_host -lm -lpthread -Wl,-rpath,$PWD/pgvector_amd/lib
 *   hnsw_host_bench [rows [max_batch]]        (8-d uniform rows, m 16, ef_construction 64)
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pgv_hip.h"
#include "pgv_host.h"

static uint64_t lcg = 12345;

static uint32_t
urand(void)
{
	lcg = lcg * 6364136223846793005ull + 1442695040888963407ull;
	return (uint32_t) (lcg >> 33);
}

int
main(int argc, char **argv)
{
	const int	n = argc > 1 ? atoi(argv[1]) : 200000,
				dim = 8,
				m = 16,
				efc = 64,
				max_batch = argc > 2 ? atoi(argv[2]) : 1024;
	static const char *names[8] = {"search", "pairs", "select", "records", "update", "patch", "pairlist", "free"};
	float	   *data = malloc(sizeof(float) * (size_t) n * dim);
	pgv_ctx    *ctx;
	pgv_hnsw   *mirror;
	pgv_hnsw_built built;
	int			rc;

	for (size_t i = 0; i < (size_t) n * dim; i++)
		data[i] = (float) (urand() % 100000) / 1000.0f;
	if (pgv_ctx_create(0, NULL, &ctx) != PGV_OK || pgv_hnsw_upload(ctx, PGV_L2SQ, PGV_F32, dim, data, n, &mirror) != PGV_OK)
	{
		fprintf(stderr, "setup: %s\n", pgv_last_error());
		return 2;
	}
	rc = pgv_host_hnsw_build(mirror, PGV_F32, dim, data, n, m, efc, NULL, max_batch, &built);
	printf("rc %d batches %ld pairs %ld\n", rc, (long) built.batches, (long) built.device_pairs);
	for (int i = 0; i < 8; i++)
		printf("%-9s %.3f\n", names[i], built.phase_secs[i]);
	if (rc == PGV_OK)
		pgv_host_hnsw_built_free(&built);
	pgv_hnsw_free(mirror);
	pgv_ctx_destroy(ctx);
	free(data);
	return rc == PGV_OK ? 0 : 1;
}
