/*
 * host_build_bench.c -- where the HOST side of pgv_host_hnsw_build spends its time, measured without
 * a GPU: the device entry points are tests/c/mock_hip.c (their time lands in the "search" / "pairs"
 * phases and is to be ignored), the phases "select", "records", "update", "patch" are pure host work.
 * Build and run:  tools/host_build_bench.sh [rows] [dim] [m] [ef_construction] [max_batch]
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "pgv_hip.h"
#include "pgv_host.h"

int
main(int argc, char **argv)
{
	int			n = argc > 1 ? atoi(argv[1]) : 30000,
				dim = argc > 2 ? atoi(argv[2]) : 16,
				m = argc > 3 ? atoi(argv[3]) : 16,
				efc = argc > 4 ? atoi(argv[4]) : 64,
				mb = argc > 5 ? atoi(argv[5]) : 256;
	float	   *data = malloc(sizeof(float) * (size_t) n * dim);
	uint64_t	lcg = 7;
	pgv_ctx    *ctx;
	pgv_hnsw   *h;
	pgv_hnsw_built b;

	for (size_t i = 0; i < (size_t) n * dim; i++)
	{
		lcg = lcg * 6364136223846793005ull + 1442695040888963407ull;
		data[i] = (float) ((lcg >> 40) & 0xFFFF) / 65536.0f;
	}
	pgv_ctx_create(0, NULL, &ctx);
	pgv_hnsw_upload(ctx, PGV_L2SQ, PGV_F32, dim, data, n, &h);
	if (pgv_host_hnsw_build(h, PGV_F32, dim, data, n, m, efc, NULL, mb, &b) != PGV_OK)
	{
		fprintf(stderr, "build failed: %s\n", pgv_host_last_error());
		return 1;
	}
	printf("rows %d batches %lld pairs %lld deferred %lld\n", n, (long long) b.batches, (long long) b.device_pairs,
		   (long long) b.deferred_updates);
	printf("mock-device phases: search %.2f s  pairs %.2f s\n", b.phase_secs[0], b.phase_secs[1]);
	printf("host phases:        select %.3f s  records %.3f s  update %.3f s  patch %.3f s\n", b.phase_secs[2],
		   b.phase_secs[3], b.phase_secs[4], b.phase_secs[5]);
	return 0;
}
