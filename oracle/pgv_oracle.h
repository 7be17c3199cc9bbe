/*
 * pgv_oracle.h -- CPU restatement of pgvector's distance hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / the timed CPU baseline.  The
 * product (libpgv_hip.so) never links, loads or falls back to it.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * the pgvector v0.8.6 tree).  Build with the reference's own flags
 * (Makefile:30): -O2 -ftree-vectorize -fassociative-math -fno-signed-zeros
 * -fno-trapping-math -ffp-contract=fast, see oracle/Makefile.
 *
 * Parity pinning: the scalar kernels and SQL-level wrappers are checked
 * against the reference's known-answer tests (test/expected/vector_type.out,
 * halfvec.out, ivfflat_vector.out, hnsw_vector.out; the JSON fixtures under tests/golden) and
 * the fp16 AND fp32 kernels, wrappers, norm and normalize additionally, bit
 * for bit, against oracle/_ref (the reference's own src/halfutils.c,
 * src/bitutils.c and src/vector.c compiled unmodified).  The loops above the
 * kernels are pinned by the reference's own access-method files RUNNING beside
 * them (tests/test_ext_runtime_cpu.py: fourteen reference files linked into a
 * stand-in server program): GetScanLists / GetScanItems over the same pages,
 * ElkanKmeans on the same pg_prng stream (centers bit for bit), the WHOLE
 * serial ivfflatbuild (sample, k-means, AddTupleToSort's argmin: every list
 * tuple for tuple) and the WHOLE serial hnswbuild (every level, neighbor and
 * the entry point) for the l2, cosine, ip and halfvec opclasses, and
 * HnswSearchLayer's walks.  pg_prng is PostgreSQL core (not in the reference
 * tree): restated from its published algorithm, "parity unpinned" for that
 * one piece (it only feeds k-means seeding and HNSW level draws), as is the
 * order in which exactly equal distances leave PostgreSQL's pairing heap.
 */
#ifndef PGV_ORACLE_H
#define PGV_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef uint16_t ora_half;		/* IEEE binary16 bit pattern (src/halfvec.h:53-59) */

/* element type of an indexed column */
enum { ORA_F32 = 0, ORA_F16 = 1 };

/* opclass families (sql/vector.sql:406-425, 819-841) */
enum { ORA_OPS_L2 = 0, ORA_OPS_IP = 1, ORA_OPS_COSINE = 2, ORA_OPS_L1 = 3 };

/* ---- L1 kernels: src/vector.c:560-574, 607-617, 649-666, 725-735 ---- */
float		ora_vector_l2_squared(int dim, const float *ax, const float *bx);
float		ora_vector_inner_product(int dim, const float *ax, const float *bx);
double		ora_vector_cosine_similarity(int dim, const float *ax, const float *bx);
float		ora_vector_l1(int dim, const float *ax, const float *bx);

/* ---- fp16 kernels: src/halfutils.c:29-43,81-91,124-144,197-207 (default)
 * and :46-78,94-121,147-194,211-239 (F16C path, used when the CPU has it,
 * mirroring HalfvecInit src/halfutils.c:278-300) ---- */
float		ora_halfvec_l2_squared(int dim, const ora_half *ax, const ora_half *bx);
float		ora_halfvec_inner_product(int dim, const ora_half *ax, const ora_half *bx);
double		ora_halfvec_cosine_similarity(int dim, const ora_half *ax, const ora_half *bx);
float		ora_halfvec_l1(int dim, const ora_half *ax, const ora_half *bx);
float		ora_halfvec_l2_squared_default(int dim, const ora_half *ax, const ora_half *bx);
float		ora_halfvec_inner_product_default(int dim, const ora_half *ax, const ora_half *bx);
int			ora_halfvec_uses_f16c(void);

/* src/halfutils.h:62-141, 146-233 */
float		ora_half_to_float(ora_half h);
ora_half	ora_float_to_half(float f);

/* ---- SQL-callable wrappers (fmgr level).  Return 0, or ORA_ERR_DIMS when the
 * reference would ereport "different vector dimensions %d and %d"
 * (src/vector.c:70-77); the message is in ora_last_error(). ---- */
#define ORA_OK 0
#define ORA_ERR_DIMS 1
#define ORA_ERR_OVERFLOW 2
#define ORA_ERR_ARG 3
const char *ora_last_error(void);

int			ora_l2_distance(int da, const float *a, int db, const float *b, double *out);	/* vector.c:579-589 */
int			ora_l2_squared_distance(int da, const float *a, int db, const float *b, double *out);	/* :595-605 */
int			ora_inner_product(int da, const float *a, int db, const float *b, double *out); /* :622-632 */
int			ora_negative_inner_product(int da, const float *a, int db, const float *b, double *out);	/* :637-647 */
int			ora_cosine_distance(int da, const float *a, int db, const float *b, double *out);	/* :671-696 */
int			ora_spherical_distance(int da, const float *a, int db, const float *b, double *out);	/* :703-722 */
int			ora_l1_distance(int da, const float *a, int db, const float *b, double *out);	/* :740-750 */
double		ora_vector_norm(int dim, const float *a);	/* :767-780 */
int			ora_l2_normalize(int dim, const float *a, float *out);	/* :785-819 */

int			ora_halfvec_l2_distance(int da, const ora_half *a, int db, const ora_half *b, double *out); /* halfvec.c:575-585 */
int			ora_halfvec_l2_squared_distance(int da, const ora_half *a, int db, const ora_half *b, double *out);
int			ora_halfvec_inner_product_f8(int da, const ora_half *a, int db, const ora_half *b, double *out);
int			ora_halfvec_negative_inner_product(int da, const ora_half *a, int db, const ora_half *b, double *out);
int			ora_halfvec_cosine_distance(int da, const ora_half *a, int db, const ora_half *b, double *out);

/* bit vectors: src/bitutils.c:49-73 (BitHammingDistanceDefault), :99-131 (BitJaccardDistanceDefault) and
 * their fmgr wrappers src/bitvec.c:45-70 ("different bit lengths %d and %d", :14-21) */
uint64_t	ora_bit_hamming(uint32_t bytes, const unsigned char *ax, const unsigned char *bx);
double		ora_bit_jaccard(uint32_t bytes, const unsigned char *ax, const unsigned char *bx);
int			ora_hamming_distance(int bits_a, const unsigned char *a, int bits_b, const unsigned char *b, double *out);
int			ora_jaccard_distance(int bits_a, const unsigned char *a, int bits_b, const unsigned char *b, double *out);
int			ora_halfvec_spherical_distance(int da, const ora_half *a, int db, const ora_half *b, double *out);
int			ora_halfvec_l1_distance(int da, const ora_half *a, int db, const ora_half *b, double *out);
double		ora_halfvec_l2_norm(int dim, const ora_half *a);	/* halfvec.c:703-719 */
int			ora_halfvec_l2_normalize(int dim, const ora_half *a, ora_half *out);	/* :724-759 */

/* generic FUNCTION 1 of an opclass (scan + assignment distance):
 * L2 -> squared L2, IP/cosine -> negative inner product (sql/vector.sql:409,415,422) */
double		ora_index_distance(int ops, int dtype, int dim, const void *a, const void *b);
/* FUNCTION 3 (k-means distance): L2 -> l2_distance, else spherical (sql/vector.sql:410,417,424) */
double		ora_kmeans_distance(int ops, int dtype, int dim, const void *a, const void *b);

/* ---- pg_prng (PostgreSQL src/common/pg_prng.c, xoroshiro128**), used through
 * RandomDouble()/RandomInt() (src/ivfflat.h:86-94) ---- */
typedef struct ora_prng
{
	uint64_t	s0,
				s1;
}			ora_prng;
void		ora_prng_seed(ora_prng * st, uint64_t seed);
uint64_t	ora_prng_u64(ora_prng * st);
uint32_t	ora_prng_u32(ora_prng * st);
double		ora_prng_double(ora_prng * st);
/* C-callable thunks with a void* state, for handing to libpgv_hip's pgv_rng */
double		ora_prng_double_cb(void *st);
uint32_t	ora_prng_u32_cb(void *st);

/* ---- IVFFlat scan (src/ivfscan.c) over a list-major in-memory index ---- */
typedef struct ora_ivf_index
{
	int			ops;			/* ORA_OPS_* */
	int			dtype;			/* ORA_F32 / ORA_F16 */
	int			dim;
	int			nlists;
	const void *centers;		/* [nlists x dim] */
	const int64_t *list_offsets;	/* [nlists + 1] */
	const void *vectors;		/* [n x dim], list-major, insertion order inside a list */
	const uint64_t *tids;		/* [n] heap TIDs (opaque 48-bit values) */
}			ora_ivf_index;

/* GetScanLists, src/ivfscan.c:47-118: the maxprobes nearest centers, ascending.
 * query == NULL reproduces ZeroDistance (:192-196).  Returns the list count. */
int			ora_ivf_get_scan_lists(const ora_ivf_index * ix, const void *query, int maxprobes,
								   int32_t *out_lists, double *out_dist);

/* GetScanItems, src/ivfscan.c:123-187: every tuple of the given lists scored
 * and sorted ascending by float8 distance (stable: ties keep insertion order,
 * a refinement of tuplesort's unspecified tie order).  out_* sized by the sum
 * of the list lengths; returns the tuple count. */
int64_t		ora_ivf_get_scan_items(const ora_ivf_index * ix, const void *query,
								   const int32_t *lists, int nlists,
								   double *out_dist, int64_t *out_slot);

/* whole first batch of ivfflatgettuple (src/ivfscan.c:360-414) incl. query
 * normalisation for cosine (:222-229): top-k heap TIDs.  Returns rows written. */
int			ora_ivf_search(const ora_ivf_index * ix, const void *query, int probes, int k,
						   uint64_t *out_tids, double *out_dist);

/* ---- IVFFlat build ---- */
/* AddTupleToSort argmin, src/ivfbuild.c:183-192 (first strictly-smallest wins) */
void		ora_ivf_assign(int ops, int dtype, int dim, const void *centers, int k,
						   const void *rows, int64_t n, int32_t *out_list, double *out_dist);

/* sample count, src/ivfbuild.c:446-455 */
int			ora_ivf_num_samples(int lists, int64_t max_tuples);

/* InitCenters (k-means++), src/ivfkmeans.c:23-91.  lower_bound may be NULL. */
void		ora_kmeans_init_centers(int ops, int dtype, int dim, const void *samples, int n,
									void *centers, int k, float *lower_bound, ora_prng * rng);

/* ComputeNewCenters, src/ivfkmeans.c:179-236 (+ SumCenters/UpdateCenters and
 * the type vtable src/ivfutils.c:301-361) from a given assignment */
void		ora_kmeans_compute_new_centers(int ops, int dtype, int dim, const void *samples, int n,
										   const int32_t *closest, void *new_centers, int k,
										   int32_t *counts, ora_prng * rng);

/* IvfflatKmeans = ElkanKmeans + checks, src/ivfkmeans.c:246-570.
 * Returns the iteration count, or -1 on a CheckCenters failure. */
int			ora_kmeans(int ops, int dtype, int dim, const void *samples, int n,
					   void *centers, int k, ora_prng * rng, int32_t *out_closest);

/* one plain Lloyd iteration from given centers (what Elkan computes exactly,
 * without its pruning): assignment by FUNCTION 3 distance, first minimum wins */
void		ora_kmeans_lloyd_assign(int ops, int dtype, int dim, const void *samples, int n,
									const void *centers, int k, int32_t *out_closest, float *out_dist);

/* ---- HNSW (src/hnswutils.c, src/hnswscan.c) over an in-memory graph ---- */
typedef struct ora_hnsw ora_hnsw;
ora_hnsw   *ora_hnsw_build(int ops, int dtype, int dim, const void *rows, int64_t n,
						   int m, int ef_construction, uint64_t seed);
/* the same algorithm run the way the reference's parallel build runs it (src/hnswbuild.c:366-480, per-element locks):
 * nthreads inserters into one shared graph; one thread gives ora_hnsw_build's graph */
ora_hnsw   *ora_hnsw_build_parallel(int ops, int dtype, int dim, const void *rows, int64_t n,
									int m, int ef_construction, uint64_t seed, int nthreads);
/* a graph built elsewhere, as neighbor tuples (see oracle_hnsw.c); only ora_hnsw_search applies */
ora_hnsw   *ora_hnsw_import(int ops, int dtype, int dim, const void *values, int64_t n, int m,
							const int32_t *levels, const int64_t *nbr_start, const int32_t *nbr, int32_t entry);
void		ora_hnsw_free(ora_hnsw * g);
/* flat export for the device mirror: per element level, neighbor slots */
int64_t		ora_hnsw_num_elements(const ora_hnsw * g);
int			ora_hnsw_entry_point(const ora_hnsw * g, int *level);
int			ora_hnsw_m(const ora_hnsw * g);
int			ora_hnsw_level(const ora_hnsw * g, int64_t e);
/* copies up to lm neighbor ids of element e at layer lc; returns the count */
int			ora_hnsw_neighbors(const ora_hnsw * g, int64_t e, int lc, int32_t *out);
int64_t		ora_hnsw_element_row(const ora_hnsw * g, int64_t e);
/* hnswgettuple first batch (src/hnswscan.c:25-56 + :293-326): top-k rows.
 * out_scored (may be NULL) = so->tuples, the number of scored elements. */
int			ora_hnsw_search(const ora_hnsw * g, const void *query, int ef_search, int k,
							int64_t *out_rows, double *out_dist, int64_t *out_scored);

/* ---- the same scan over the real on-disk layout: an array of 8 KB pages (oracle_pages.c) ---- */
int			ora_pages_meta(const uint8_t *pages, uint32_t nblocks, int *dim, int *lists);
int			ora_pages_search(const uint8_t *pages, uint32_t nblocks, int ops, int dtype, const void *query,
							 int probes, int k, uint64_t *out_tids, double *out_dist, int64_t *out_scanned);

/* ---- bench.py's thread runners round the functions above (oracle_bench.c): pinned threads, spread placement ---- */
int			ora_bench_cpus(void);
void	   *ora_bench_alloc(size_t bytes);
void		ora_bench_free(void *p, size_t bytes);
int			ora_bench_spread_copy(void *dst, const void *src, size_t bytes, int nthreads);
int			ora_bench_search(const ora_ivf_index * ix, const uint8_t *pages, uint32_t nblocks, int ops, int dtype,
							 const void *queries, size_t query_bytes, int nq, int probes, int k, int nthreads,
							 double seconds, uint64_t *out_tids, double *out_dist, int *out_count, double *out_stats);
int			ora_bench_assign(int ops, int dtype, int dim, const void *centers, int k, const void *rows, int64_t n,
							 int nthreads, int32_t *out_list, double *out_seconds);

#ifdef __cplusplus
}
#endif
#endif
