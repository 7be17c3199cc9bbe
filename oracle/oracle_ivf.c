/*
 * oracle_ivf.c -- CPU restatement of pgvector's IVFFlat hot loops: probe
 * selection and list scan (src/ivfscan.c), tuple->list assignment
 * (src/ivfbuild.c) and k-means (src/ivfkmeans.c).  TEST INFRASTRUCTURE ONLY
 * (see pgv_oracle.h).  Postgres services the loops lean on (pairingheap,
 * tuplesort, palloc) are replaced by the plainest equivalent and the
 * replacement is stated where it matters for results (tie order).
 */
#include "pgv_oracle.h"

#include <float.h>
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

static inline size_t
elem_size(int dtype)
{
	return dtype == ORA_F32 ? sizeof(float) : sizeof(ora_half);
}

static inline const void *
row_ptr(const void *base, int dtype, int dim, int64_t i)
{
	return (const char *) base + (size_t) i * (size_t) dim * elem_size(dtype);
}

static inline void *
row_ptr_rw(void *base, int dtype, int dim, int64_t i)
{
	return (char *) base + (size_t) i * (size_t) dim * elem_size(dtype);
}

/* float8 ordering as PostgreSQL's float8_cmp_internal: NaN sorts after everything */
static inline int
float8_cmp(double a, double b)
{
	if (isnan(a))
		return isnan(b) ? 0 : 1;
	if (isnan(b))
		return -1;
	return (a > b) - (a < b);
}

/* ------------------------------------------------------------ GetScanLists */

typedef struct
{
	double		distance;
	int32_t		list;
}			scan_list;

/* max-heap on (distance, list): stands in for the pairing heap of
 * src/ivfscan.c:30-42.  Among equal distances the larger list id counts as
 * "further", so the kept set is the maxprobes smallest by (distance, id) --
 * one of the outcomes the reference allows (its tie order is unspecified). */
static inline int
further(const scan_list * a, const scan_list * b)
{
	int			c = float8_cmp(a->distance, b->distance);

	return c > 0 || (c == 0 && a->list > b->list);
}

static void
heap_sift_down(scan_list * h, int n, int i)
{
	for (;;)
	{
		int			l = 2 * i + 1,
					r = l + 1,
					m = i;

		if (l < n && further(&h[l], &h[m]))
			m = l;
		if (r < n && further(&h[r], &h[m]))
			m = r;
		if (m == i)
			return;
		scan_list	t = h[i];

		h[i] = h[m];
		h[m] = t;
		i = m;
	}
}

static void
heap_sift_up(scan_list * h, int i)
{
	while (i > 0)
	{
		int			p = (i - 1) / 2;

		if (!further(&h[i], &h[p]))
			return;
		scan_list	t = h[i];

		h[i] = h[p];
		h[p] = t;
		i = p;
	}
}

/* src/ivfscan.c:47-118 */
int
ora_ivf_get_scan_lists(const ora_ivf_index * ix, const void *query, int maxprobes,
					   int32_t *out_lists, double *out_dist)
{
	scan_list  *heap;
	int			count = 0;
	double		max_distance = DBL_MAX;

	if (maxprobes > ix->nlists)
		maxprobes = ix->nlists; /* src/ivfscan.c:274-278 */
	if (maxprobes <= 0)
		return 0;
	heap = malloc(sizeof(scan_list) * (size_t) maxprobes);

	/* list pages are visited in list-id order (src/ivfbuild.c:527-551) */
	for (int l = 0; l < ix->nlists; l++)
	{
		/* NULL query -> ZeroDistance (src/ivfscan.c:192-196, 207-211) */
		double		distance = query == NULL ? 0.0 :
			ora_index_distance(ix->ops, ix->dtype, ix->dim, row_ptr(ix->centers, ix->dtype, ix->dim, l), query);

		if (count < maxprobes)
		{
			heap[count].distance = distance;
			heap[count].list = l;
			heap_sift_up(heap, count);
			count++;
			if (count == maxprobes)
				max_distance = heap[0].distance;
		}
		else if (distance < max_distance)	/* strict: src/ivfscan.c:92 */
		{
			heap[0].distance = distance;
			heap[0].list = l;
			heap_sift_down(heap, count, 0);
			max_distance = heap[0].distance;
		}
	}

	/* pop furthest-first into the tail: ascending output (src/ivfscan.c:114-115) */
	for (int n = count; n > 0; n--)
	{
		out_lists[n - 1] = heap[0].list;
		if (out_dist)
			out_dist[n - 1] = heap[0].distance;
		heap[0] = heap[n - 1];
		heap_sift_down(heap, n - 1, 0);
	}
	free(heap);
	return count;
}

/* ------------------------------------------------------------ GetScanItems */

typedef struct
{
	double		distance;
	int64_t		slot;
	int64_t		seq;
}			scan_item;

static int
scan_item_cmp(const void *pa, const void *pb)
{
	const scan_item *a = pa,
			   *b = pb;
	int			c = float8_cmp(a->distance, b->distance);

	if (c)
		return c;
	return (a->seq > b->seq) - (a->seq < b->seq);
}

/* src/ivfscan.c:123-187; the tuplesort (Float8LessOperator ascending, :238-247)
 * becomes a sort on (distance, insertion sequence) */
int64_t
ora_ivf_get_scan_items(const ora_ivf_index * ix, const void *query,
					   const int32_t *lists, int nlists,
					   double *out_dist, int64_t *out_slot)
{
	int64_t		total = 0,
				n = 0;
	scan_item  *items;

	for (int p = 0; p < nlists; p++)
		total += ix->list_offsets[lists[p] + 1] - ix->list_offsets[lists[p]];
	items = malloc(sizeof(scan_item) * (size_t) (total > 0 ? total : 1));

	for (int p = 0; p < nlists; p++)
	{
		int64_t		beg = ix->list_offsets[lists[p]],
					end = ix->list_offsets[lists[p] + 1];

		for (int64_t s = beg; s < end; s++)
		{
			items[n].distance = query == NULL ? 0.0 :
				ora_index_distance(ix->ops, ix->dtype, ix->dim, row_ptr(ix->vectors, ix->dtype, ix->dim, s), query);
			items[n].slot = s;
			items[n].seq = n;
			n++;
		}
	}
	qsort(items, (size_t) n, sizeof(scan_item), scan_item_cmp);
	for (int64_t i = 0; i < n; i++)
	{
		out_dist[i] = items[i].distance;
		out_slot[i] = items[i].slot;
	}
	free(items);
	return n;
}

/* src/ivfscan.c:360-414, first batch only (iterative scan off) */
int
ora_ivf_search(const ora_ivf_index * ix, const void *query, int probes, int k,
			   uint64_t *out_tids, double *out_dist)
{
	void	   *nq = NULL;
	const void *q = query;
	int32_t    *lists;
	int			nl;
	int64_t		total = 0,
				n;
	double	   *d;
	int64_t    *s;
	int			written = 0;

	if (probes > ix->nlists)
		probes = ix->nlists;	/* src/ivfscan.c:271-272 */
	if (probes < 1)
		return 0;

	/* GetScanValue: cosine opclass (NORM_PROC present) normalises the query, :222-229 */
	if (query != NULL && ix->ops == ORA_OPS_COSINE)
	{
		nq = malloc((size_t) ix->dim * elem_size(ix->dtype));
		if (ix->dtype == ORA_F32)
			ora_l2_normalize(ix->dim, query, nq);
		else
			ora_halfvec_l2_normalize(ix->dim, query, nq);
		q = nq;
	}

	lists = malloc(sizeof(int32_t) * (size_t) probes);
	nl = ora_ivf_get_scan_lists(ix, q, probes, lists, NULL);
	for (int p = 0; p < nl; p++)
		total += ix->list_offsets[lists[p] + 1] - ix->list_offsets[lists[p]];
	d = malloc(sizeof(double) * (size_t) (total > 0 ? total : 1));
	s = malloc(sizeof(int64_t) * (size_t) (total > 0 ? total : 1));
	n = ora_ivf_get_scan_items(ix, q, lists, nl, d, s);
	for (int64_t i = 0; i < n && written < k; i++, written++)
	{
		out_tids[written] = ix->tids ? ix->tids[s[i]] : (uint64_t) s[i];
		if (out_dist)
			out_dist[written] = d[i];
	}
	free(s);
	free(d);
	free(lists);
	free(nq);
	return written;
}

/* ------------------------------------------------------------ build: assign */

/* src/ivfbuild.c:183-192 */
void
ora_ivf_assign(int ops, int dtype, int dim, const void *centers, int k,
			   const void *rows, int64_t n, int32_t *out_list, double *out_dist)
{
	for (int64_t r = 0; r < n; r++)
	{
		const void *value = row_ptr(rows, dtype, dim, r);
		double		min_distance = DBL_MAX;
		int			closest = 0;

		for (int c = 0; c < k; c++)
		{
			double		distance = ora_index_distance(ops, dtype, dim, value, row_ptr(centers, dtype, dim, c));

			if (distance < min_distance)
			{
				min_distance = distance;
				closest = c;
			}
		}
		out_list[r] = closest;
		if (out_dist)
			out_dist[r] = min_distance;
	}
}

/* src/ivfbuild.c:446-455 */
int
ora_ivf_num_samples(int lists, int64_t max_tuples)
{
	int64_t		n = (int64_t) lists * 50;

	if (n < 10000)
		n = 10000;
	if (n > max_tuples)
		n = max_tuples;
	if (n < 1)
		n = 1;
	return (int) n;
}

/* ------------------------------------------------------------------ k-means */

static int
is_spherical(int ops)
{
	/* KMEANS_NORM_PROC (FUNCTION 4) exists for ip and cosine: sql/vector.sql:412-425 */
	return ops == ORA_OPS_IP || ops == ORA_OPS_COSINE;
}

static void
copy_item(int dtype, int dim, void *dst, const void *src)
{
	memcpy(dst, src, (size_t) dim * elem_size(dtype));
}

/* src/ivfkmeans.c:23-91 */
void
ora_kmeans_init_centers(int ops, int dtype, int dim, const void *samples, int n,
						void *centers, int k, float *lower_bound, ora_prng * rng)
{
	float	   *weight = malloc(sizeof(float) * (size_t) n);

	/* first center uniformly at random (:36) */
	copy_item(dtype, dim, row_ptr_rw(centers, dtype, dim, 0),
			  row_ptr(samples, dtype, dim, (int) (ora_prng_u32(rng) % (uint32_t) n)));

	for (int j = 0; j < n; j++)
		weight[j] = FLT_MAX;

	for (int i = 0; i < k; i++)
	{
		double		sum = 0.0;
		double		choice;
		int			j;

		for (j = 0; j < n; j++)
		{
			double		distance = ora_kmeans_distance(ops, dtype, dim, row_ptr(samples, dtype, dim, j),
													   row_ptr(centers, dtype, dim, i));

			if (lower_bound)
				lower_bound[(size_t) j * (size_t) k + (size_t) i] = (float) distance;

			/* squared distance drives the D^2 sampling (:64-70) */
			distance *= distance;
			if (distance < weight[j])
				weight[j] = (float) distance;
			sum += weight[j];
		}

		if (i + 1 == k)
			break;

		/* weighted pick (:77-84) */
		choice = sum * ora_prng_double(rng);
		for (j = 0; j < n - 1; j++)
		{
			choice -= weight[j];
			if (choice <= 0)
				break;
		}
		copy_item(dtype, dim, row_ptr_rw(centers, dtype, dim, i + 1), row_ptr(samples, dtype, dim, j));
	}
	free(weight);
}

/* NormCenters, src/ivfkmeans.c:96-105 -> IvfflatNormVectors src/ivfutils.c:100-114 */
static void
norm_centers(int dtype, int dim, void *centers, int k)
{
	void	   *tmp = malloc((size_t) dim * elem_size(dtype));

	for (int c = 0; c < k; c++)
	{
		void	   *v = row_ptr_rw(centers, dtype, dim, c);

		if (dtype == ORA_F32)
			ora_l2_normalize(dim, v, tmp);
		else
			ora_halfvec_l2_normalize(dim, v, tmp);
		copy_item(dtype, dim, v, tmp);
	}
	free(tmp);
}

/* src/ivfkmeans.c:179-236 with SumCenters/UpdateCenters (:151-174) and the
 * per-type sum/update callbacks (src/ivfutils.c:301-361) */
void
ora_kmeans_compute_new_centers(int ops, int dtype, int dim, const void *samples, int n,
							   const int32_t *closest, void *new_centers, int k,
							   int32_t *counts, ora_prng * rng)
{
	float	   *agg = calloc((size_t) k * (size_t) dim, sizeof(float));
	int32_t    *cnt = counts ? counts : malloc(sizeof(int32_t) * (size_t) k);

	for (int c = 0; c < k; c++)
		cnt[c] = 0;

	/* fp32 sums in sample order (VectorSumCenter / HalfvecSumCenter) */
	for (int i = 0; i < n; i++)
	{
		float	   *x = agg + (size_t) closest[i] * (size_t) dim;

		if (dtype == ORA_F32)
		{
			const float *v = row_ptr(samples, dtype, dim, i);

			for (int j = 0; j < dim; j++)
				x[j] += v[j];
		}
		else
		{
			const ora_half *v = row_ptr(samples, dtype, dim, i);

			for (int j = 0; j < dim; j++)
				x[j] += ora_half_to_float(v[j]);
		}
	}
	for (int i = 0; i < n; i++)
		cnt[closest[i]] += 1;

	for (int c = 0; c < k; c++)
	{
		float	   *x = agg + (size_t) c * (size_t) dim;

		if (cnt[c] > 0)
		{
			for (int j = 0; j < dim; j++)
				if (isinf(x[j]))
					x[j] = x[j] > 0 ? FLT_MAX : -FLT_MAX;
			for (int j = 0; j < dim; j++)
				x[j] /= (float) cnt[c];
		}
		else
		{
			/* empty cluster: uniform random point (:222-227) */
			for (int j = 0; j < dim; j++)
				x[j] = (float) ora_prng_double(rng);
		}
	}

	/* VectorUpdateCenter / HalfvecUpdateCenter */
	for (int c = 0; c < k; c++)
	{
		const float *x = agg + (size_t) c * (size_t) dim;

		if (dtype == ORA_F32)
			memcpy(row_ptr_rw(new_centers, dtype, dim, c), x, sizeof(float) * (size_t) dim);
		else
		{
			ora_half   *h = row_ptr_rw(new_centers, dtype, dim, c);

			for (int j = 0; j < dim; j++)
				h[j] = ora_float_to_half(x[j]);
		}
	}

	if (is_spherical(ops))
		norm_centers(dtype, dim, new_centers, k);

	if (!counts)
		free(cnt);
	free(agg);
}

/* CheckCenters, src/ivfkmeans.c:490-547 */
static int
check_centers(int ops, int dtype, int dim, const void *centers, int k)
{
	for (int c = 0; c < k; c++)
	{
		const void *v = row_ptr(centers, dtype, dim, c);

		for (int j = 0; j < dim; j++)
		{
			float		x = dtype == ORA_F32 ? ((const float *) v)[j] : ora_half_to_float(((const ora_half *) v)[j]);

			if (isnan(x) || isinf(x))
				return -1;
		}
		/* CheckNorms looks at NORM_PROC (FUNCTION 2): cosine only */
		if (ops == ORA_OPS_COSINE)
		{
			double		norm = dtype == ORA_F32 ? ora_vector_norm(dim, v) : ora_halfvec_l2_norm(dim, v);

			if (norm == 0)
				return -1;
		}
	}
	return 0;
}

/* RandomCenters, src/ivfkmeans.c:110-133 */
static void
random_centers(int ops, int dtype, int dim, void *centers, int k, ora_prng * rng)
{
	for (int c = 0; c < k; c++)
	{
		void	   *v = row_ptr_rw(centers, dtype, dim, c);

		for (int j = 0; j < dim; j++)
		{
			float		x = (float) ora_prng_double(rng);

			if (dtype == ORA_F32)
				((float *) v)[j] = x;
			else
				((ora_half *) v)[j] = ora_float_to_half(x);
		}
	}
	if (is_spherical(ops))
		norm_centers(dtype, dim, centers, k);
}

#define KD(a, b) ora_kmeans_distance(ops, dtype, dim, (a), (b))
#define SAMPLE(j) row_ptr(samples, dtype, dim, (j))
#define CENTER(c) row_ptr(centers, dtype, dim, (c))

/* ElkanKmeans, src/ivfkmeans.c:246-485, wrapped as IvfflatKmeans :553-570 */
int
ora_kmeans(int ops, int dtype, int dim, const void *samples, int n,
		   void *centers, int k, ora_prng * rng, int32_t *out_closest)
{
	void	   *new_centers;
	int32_t    *counts,
			   *closest;
	float	   *lower,
			   *upper,
			   *s,
			   *halfcdist,
			   *newcdist;
	int			iterations = 0;

	if (n == 0)
	{
		random_centers(ops, dtype, dim, centers, k, rng);
		return check_centers(ops, dtype, dim, centers, k) ? -1 : 0;
	}

	new_centers = malloc((size_t) k * (size_t) dim * elem_size(dtype));
	counts = malloc(sizeof(int32_t) * (size_t) k);
	closest = malloc(sizeof(int32_t) * (size_t) n);
	lower = malloc(sizeof(float) * (size_t) n * (size_t) k);
	upper = malloc(sizeof(float) * (size_t) n);
	s = malloc(sizeof(float) * (size_t) k);
	halfcdist = malloc(sizeof(float) * (size_t) k * (size_t) k);
	newcdist = malloc(sizeof(float) * (size_t) k);

	ora_kmeans_init_centers(ops, dtype, dim, samples, n, centers, k, lower, rng);

	/* initial assignment from the k-means++ lower bounds (:323-344) */
	for (int j = 0; j < n; j++)
	{
		float		min_distance = FLT_MAX;
		int			c_best = 0;

		for (int c = 0; c < k; c++)
		{
			float		distance = lower[(size_t) j * (size_t) k + (size_t) c];

			if (distance < min_distance)
			{
				min_distance = distance;
				c_best = c;
			}
		}
		upper[j] = min_distance;
		closest[j] = c_best;
	}

	for (int iteration = 0; iteration < 500; iteration++)
	{
		int			changes = 0;
		int			rjreset = iteration != 0;

		iterations = iteration + 1;

		/* step 1: half centre-centre distances (:356-367) */
		for (int a = 0; a < k; a++)
			for (int b = a + 1; b < k; b++)
			{
				float		distance = (float) (0.5 * KD(CENTER(a), CENTER(b)));

				halfcdist[(size_t) a * (size_t) k + (size_t) b] = distance;
				halfcdist[(size_t) b * (size_t) k + (size_t) a] = distance;
			}

		/* s(c) (:370-387) */
		for (int a = 0; a < k; a++)
		{
			float		min_distance = FLT_MAX;

			for (int b = 0; b < k; b++)
			{
				float		distance;

				if (a == b)
					continue;
				distance = halfcdist[(size_t) a * (size_t) k + (size_t) b];
				if (distance < min_distance)
					min_distance = distance;
			}
			s[a] = min_distance;
		}

		for (int j = 0; j < n; j++)
		{
			int			rj;

			/* step 2 (:396-397) */
			if (upper[j] <= s[closest[j]])
				continue;

			rj = rjreset;

			for (int c = 0; c < k; c++)
			{
				float		dxcx;

				/* step 3 (:404-411) */
				if (c == closest[j])
					continue;
				if (upper[j] <= lower[(size_t) j * (size_t) k + (size_t) c])
					continue;
				if (upper[j] <= halfcdist[(size_t) closest[j] * (size_t) k + (size_t) c])
					continue;

				/* step 3a (:416-427) */
				if (rj)
				{
					dxcx = (float) KD(SAMPLE(j), CENTER(closest[j]));
					lower[(size_t) j * (size_t) k + (size_t) closest[j]] = dxcx;
					upper[j] = dxcx;
					rj = 0;
				}
				else
					dxcx = upper[j];

				/* step 3b (:430-447) */
				if (dxcx > lower[(size_t) j * (size_t) k + (size_t) c] ||
					dxcx > halfcdist[(size_t) closest[j] * (size_t) k + (size_t) c])
				{
					float		dxc = (float) KD(SAMPLE(j), CENTER(c));

					lower[(size_t) j * (size_t) k + (size_t) c] = dxc;
					if (dxc < dxcx)
					{
						closest[j] = c;
						upper[j] = dxc;
						changes++;
					}
				}
			}
		}

		/* step 4 (:454) */
		ora_kmeans_compute_new_centers(ops, dtype, dim, samples, n, closest, new_centers, k, counts, rng);

		/* step 5 (:457-471) */
		for (int c = 0; c < k; c++)
			newcdist[c] = (float) KD(CENTER(c), row_ptr(new_centers, dtype, dim, c));
		for (int j = 0; j < n; j++)
			for (int c = 0; c < k; c++)
			{
				float		distance = lower[(size_t) j * (size_t) k + (size_t) c] - newcdist[c];

				if (distance < 0)
					distance = 0;
				lower[(size_t) j * (size_t) k + (size_t) c] = distance;
			}

		/* step 6 (:475-476) */
		for (int j = 0; j < n; j++)
			upper[j] += newcdist[closest[j]];

		/* step 7 (:479-480) */
		memcpy(centers, new_centers, (size_t) k * (size_t) dim * elem_size(dtype));

		if (changes == 0 && iteration != 0)
			break;
	}

	if (out_closest)
		memcpy(out_closest, closest, sizeof(int32_t) * (size_t) n);

	free(newcdist);
	free(halfcdist);
	free(s);
	free(upper);
	free(lower);
	free(closest);
	free(counts);
	free(new_centers);

	return check_centers(ops, dtype, dim, centers, k) ? -1 : iterations;
}

/* plain Lloyd assignment from given centers: the exact computation Elkan's
 * bounds prune (src/ivfkmeans.c:391-451 without the skips); first minimum wins,
 * distances compared as float like the reference's dxc/dxcx */
void
ora_kmeans_lloyd_assign(int ops, int dtype, int dim, const void *samples, int n,
						const void *centers, int k, int32_t *out_closest, float *out_dist)
{
	for (int j = 0; j < n; j++)
	{
		float		best = FLT_MAX;
		int			c_best = 0;

		for (int c = 0; c < k; c++)
		{
			float		distance = (float) KD(SAMPLE(j), CENTER(c));

			if (distance < best)
			{
				best = distance;
				c_best = c;
			}
		}
		out_closest[j] = c_best;
		if (out_dist)
			out_dist[j] = best;
	}
}
