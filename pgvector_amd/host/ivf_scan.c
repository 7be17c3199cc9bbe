/*
 * ivf_scan.c -- ivfflatbeginscan / rescan / gettuple / endscan
 * (src/ivfscan.c:252-431) with GetScanLists and GetScanItems (:47-187) running on
 * the GPU through pgv_rank_lists / pgv_scan_lists.  The tuplesort
 * (Float8LessOperator ascending, :238-247) is a stable merge sort on the float8
 * widening of the kernel value.
 */
#include "pgv_host.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

extern int	pgv_host_fail(int code, const char *fmt,...);

#define SCAN_HEAD 64				/* sorted tuples fetched with the scan itself: covers the usual LIMIT */
#define SCAN_REFILL 256				/* tuples per further fetch from the device-resident batch */
#define SCAN_DEVICE_DEPTH 1024		/* deeper than this, the batch comes over whole and is sorted here */

struct pgv_ivf_scan
{
	pgv_index  *mirror;
	const pgv_ivf_image *img;
	pgv_query  *dq;				/* device-resident scan state (NULL: probes beyond the fused path's limits) */
	int			probes,
				max_probes;
	int			iterative;
	int			normalize_query;
	int			first;
	int			is_null;
	void	   *value;			/* query payload (normalised if needed) */
	int32_t    *lists;			/* listPages equivalent: probe order (host copy, fetched when needed) */
	int			have_lists;
	int			nlists;
	int			list_index;
	int			batch_first,	/* the ranked lists of the current batch */
				batch_n;
	/* the current batch's sorted stream: `count` tuples, position `next` is returned next */
	int64_t		count,
				next;
	/* window [win_base, win_base + win_n) of it fetched from the device, already sorted */
	float		win_dist[SCAN_REFILL];
	int64_t		win_slot[SCAN_REFILL];
	int64_t		win_base;
	int			win_n;
	/* whole batch on the host (deep pulls and the legacy path): distances, slots, sort permutation */
	int			whole;
	float	   *dist;
	int64_t    *slot;
	int64_t    *order;			/* permutation: ascending distance, stable */
	int64_t		capacity;
};

/* l2_normalize / halfvec_l2_normalize (src/vector.c:785-819, src/halfvec.c:724-759) are the
 * reference's own CPU functions and stay in the extension; restated for this harness */
static float
half_to_float(uint16_t h)
{
	uint32_t	sign = ((uint32_t) h & 0x8000u) << 16,
				e = (h >> 10) & 0x1F,
				m = h & 0x3FF,
				bits;
	float		f;

	if (e == 31)
		bits = sign | 0x7F800000u | (m << 13);
	else if (e == 0)
	{
		if (m == 0)
			bits = sign;
		else
		{
			int			ex = -14;

			while (!(m & 0x400))
			{
				m <<= 1;
				ex--;
			}
			bits = sign | ((uint32_t) (ex + 127) << 23) | ((m & 0x3FF) << 13);
		}
	}
	else
		bits = sign | ((e + 112) << 23) | (m << 13);
	memcpy(&f, &bits, 4);
	return f;
}

/* round to nearest even, overflow to inf: what _cvtss_sh(x, 0) / Float4ToHalfUnchecked do
 * (src/halfutils.h:146-233); exported for the unit test */
uint16_t
pgv_host_float_to_half(float f)
{
	uint32_t	x,
				sign,
				r,
				rem;

	memcpy(&x, &f, 4);
	sign = (x >> 16) & 0x8000u;
	x &= 0x7FFFFFFFu;
	if (x >= 0x7F800000u)		/* inf / NaN */
		return (uint16_t) (sign | 0x7C00u | (x > 0x7F800000u ? (0x200u | ((x >> 13) & 0x3FFu)) : 0u));
	if (x >= 0x47800000u)		/* >= 65536 */
		return (uint16_t) (sign | 0x7C00u);
	if (x < 0x38800000u)		/* below the smallest normal half: subnormal or zero */
	{
		int			e,
					shift;
		uint32_t	m,
					half;

		if (x < 0x33000000u)	/* < 2^-25 */
			return (uint16_t) sign;
		e = (int) (x >> 23);
		m = (x & 0x7FFFFFu) | 0x800000u;
		shift = 113 - e + 13;
		r = m >> shift;
		rem = m & ((1u << shift) - 1u);
		half = 1u << (shift - 1);
		if (rem > half || (rem == half && (r & 1u)))
			r++;
		return (uint16_t) (sign | r);
	}
	r = (x - 0x38000000u) >> 13;
	rem = x & 0x1FFFu;
	if (rem > 0x1000u || (rem == 0x1000u && (r & 1u)))
		r++;					/* a carry into the exponent is the correct rounding, up to inf */
	return (uint16_t) (sign | r);
}

/* returns 0 for a zero-norm value (IvfflatCheckNorm, src/ivfutils.c:98-108): such a row is not indexed */
int
pgv_host_normalize_value(pgv_dtype t, int dim, const void *in, void *out)
{
	double		norm = 0;

	for (int i = 0; i < dim; i++)
	{
		double		x = t == PGV_F32 ? (double) ((const float *) in)[i] : (double) half_to_float(((const uint16_t *) in)[i]);

		norm += x * x;
	}
	norm = sqrt(norm);
	memset(out, 0, (size_t) dim * (t == PGV_F32 ? 4 : 2));
	if (norm > 0)
		for (int i = 0; i < dim; i++)
		{
			if (t == PGV_F32)
				((float *) out)[i] = (float) (((const float *) in)[i] / norm);
			else
				((uint16_t *) out)[i] = pgv_host_float_to_half((float) (half_to_float(((const uint16_t *) in)[i]) / norm));
		}
	return norm > 0;
}

int
pgv_host_ivf_beginscan(pgv_index * mirror, const pgv_ivf_image * img, int probes, int max_probes,
					   int iterative, int normalize_query, pgv_ivf_scan * *out)
{
	pgv_ivf_scan *so;
	int			lists = img->lists;
	int			maxp;

	if (!mirror || !img || !out || probes < 1)
		return pgv_host_fail(PGV_ERR_ARG, "pgv_host_ivf_beginscan: bad argument");
	/* src/ivfscan.c:266-277 */
	maxp = iterative ? (max_probes > probes ? max_probes : probes) : probes;
	if (probes > lists)
		probes = lists;
	if (maxp > lists)
		maxp = lists;
	so = calloc(1, sizeof(*so));
	if (!so)
		return pgv_host_fail(PGV_ERR_NOMEM, "out of memory");
	so->mirror = mirror;
	so->img = img;
	so->probes = probes;
	so->max_probes = maxp;
	so->iterative = iterative;
	so->normalize_query = normalize_query;
	so->first = 1;
	so->value = malloc((size_t) img->dim * 4);
	so->lists = malloc(sizeof(int32_t) * (size_t) maxp);
	if (!so->value || !so->lists)
	{
		pgv_host_ivf_endscan(so);
		return pgv_host_fail(PGV_ERR_NOMEM, "out of memory");
	}
	/* the fused single-query path: GetScanLists and GetScanItems stay on the device */
	if (probes <= 256 && maxp <= 1024)
	{
		int			rc = pgv_query_begin(mirror, &so->dq);

		if (rc != PGV_OK)
		{
			pgv_host_ivf_endscan(so);
			return pgv_host_fail(rc, "%s", pgv_last_error());
		}
	}
	*out = so;
	return PGV_OK;
}

int
pgv_host_ivf_rescan(pgv_ivf_scan * so, const void *query)
{
	so->first = 1;
	so->list_index = 0;
	so->count = so->next = 0;
	so->win_n = 0;
	so->whole = 0;
	so->have_lists = 0;
	so->is_null = query == NULL;
	if (query)
	{
		/* GetScanValue, src/ivfscan.c:201-233 */
		if (so->normalize_query)
			pgv_host_normalize_value(so->img->dtype, so->img->dim, query, so->value);
		else
			memcpy(so->value, query, (size_t) so->img->dim * (so->img->dtype == PGV_F32 ? 4 : 2));
	}
	return PGV_OK;
}

/* stable merge sort of indices by float8 distance (NaN last, like float8 ordering) */
static inline int
dist_less(float a, float b)
{
	if (isnan(a))
		return 0;
	if (isnan(b))
		return 1;
	return (double) a < (double) b;
}

static void
merge_sort(int64_t *idx, int64_t *tmp, const float *d, int64_t n)
{
	if (n < 2)
		return;
	{
		int64_t		h = n / 2,
					i = 0,
					j = h,
					k = 0;

		merge_sort(idx, tmp, d, h);
		merge_sort(idx + h, tmp, d, n - h);
		while (i < h && j < n)
			tmp[k++] = dist_less(d[idx[j]], d[idx[i]]) ? idx[j++] : idx[i++];
		while (i < h)
			tmp[k++] = idx[i++];
		while (j < n)
			tmp[k++] = idx[j++];
		memcpy(idx, tmp, sizeof(int64_t) * (size_t) n);
	}
}

/* the ranked list ids on the host (the legacy path and deep pulls need them) */
static int
fetch_lists(pgv_ivf_scan * so)
{
	int			rc;

	if (so->have_lists)
		return PGV_OK;
	if (so->is_null)
	{
		/* ZeroDistance: every center at distance 0; the strict `<` keeps the first maxProbes lists */
		for (int i = 0; i < so->max_probes; i++)
			so->lists[i] = i;
	}
	else
	{
		rc = so->dq ? pgv_query_lists(so->dq, so->lists, so->max_probes)
			: pgv_rank_lists(so->mirror, so->value, 1, so->max_probes, so->lists, NULL);
		if (rc != PGV_OK)
			return pgv_host_fail(rc, "%s", pgv_last_error());
	}
	so->have_lists = 1;
	return PGV_OK;
}

/* the whole current batch onto the host, sorted here: what the reference's tuplesort holds */
static int
fetch_whole_batch(pgv_ivf_scan * so)
{
	int64_t		m = 0,
				got;
	int			rc;

	if (fetch_lists(so) != PGV_OK)
		return -1;
	for (int p = 0; p < so->batch_n; p++)
		m += so->img->list_offsets[so->lists[so->batch_first + p] + 1] - so->img->list_offsets[so->lists[so->batch_first + p]];
	if (m > so->capacity)
	{
		int64_t		cap = m * 2 + 64;
		float	   *nd = realloc(so->dist, sizeof(float) * (size_t) cap);
		int64_t    *ns = nd ? realloc(so->slot, sizeof(int64_t) * (size_t) cap) : NULL;
		int64_t    *no = ns ? realloc(so->order, sizeof(int64_t) * (size_t) cap * 2) : NULL;

		if (nd)
			so->dist = nd;
		if (ns)
			so->slot = ns;
		if (no)
			so->order = no;
		if (!nd || !ns || !no)
			return pgv_host_fail(PGV_ERR_NOMEM, "out of memory for a batch of %lld tuples", (long long) m);
		so->capacity = cap;
	}
	rc = pgv_scan_lists(so->mirror, so->is_null ? NULL : so->value, so->lists + so->batch_first, so->batch_n,
						so->dist, so->slot, so->capacity, &got);
	if (rc != PGV_OK)
		return pgv_host_fail(rc, "%s", pgv_last_error());
	for (int64_t i = 0; i < got; i++)
		so->order[i] = i;
	merge_sort(so->order, so->order + so->capacity, so->dist, got);	/* tuplesort_performsort */
	so->count = got;
	so->whole = 1;
	return PGV_OK;
}

/* GetScanItems for the next batch of `probes` lists (src/ivfscan.c:123-187) */
static int
get_scan_items(pgv_ivf_scan * so)
{
	int			nl = so->nlists - so->list_index;

	if (nl > so->probes)
		nl = so->probes;
	so->batch_first = so->list_index;
	so->batch_n = nl;
	so->list_index += nl;
	so->next = 0;
	so->whole = 0;
	so->win_n = 0;
	so->win_base = 0;
	if (so->dq)
	{
		/* scored, sorted and kept on the device; the head of the sorted stream comes back */
		int64_t		total;
		int			rc = pgv_query_scan(so->dq, so->batch_first, nl, SCAN_HEAD, so->win_dist, so->win_slot, NULL,
										&so->win_n, &total);

		if (rc != PGV_OK)
			return pgv_host_fail(rc, "%s", pgv_last_error());
		so->count = total;
		return PGV_OK;
	}
	return fetch_whole_batch(so);
}

int
pgv_host_ivf_gettuple(pgv_ivf_scan * so, uint64_t *out_tid, double *out_distance)
{
	if (so->first)
	{
		/* GetScanLists, src/ivfscan.c:47-118 */
		if (so->dq)
		{
			int			rc = pgv_query_rank(so->dq, so->is_null ? NULL : so->value, so->max_probes);

			if (rc != PGV_OK)
				return pgv_host_fail(rc, "%s", pgv_last_error()), -1;
		}
		else if (fetch_lists(so) != PGV_OK)
			return -1;
		so->nlists = so->max_probes;
		so->list_index = 0;
		if (get_scan_items(so) != PGV_OK)
			return -1;
		so->first = 0;
	}
	while (so->next >= so->count)
	{
		/* src/ivfscan.c:400-406: iterative scan fetches the next `probes` lists */
		if (!so->iterative || so->list_index >= so->nlists)
			return 0;
		if (get_scan_items(so) != PGV_OK)
			return -1;
	}
	if (!so->whole && so->next >= so->win_base + so->win_n)
	{
		/* the executor pulls past what has been fetched: the next window of the device-resident batch,
		 * or, far into it, the whole batch (then the sort is the host's, like the reference's) */
		if (so->next + SCAN_REFILL <= SCAN_DEVICE_DEPTH)
		{
			int			rc = pgv_query_more(so->dq, (int) so->next, SCAN_REFILL, so->win_dist, so->win_slot, NULL, &so->win_n);

			if (rc != PGV_OK)
				return pgv_host_fail(rc, "%s", pgv_last_error()), -1;
			so->win_base = so->next;
			if (so->win_n <= 0)
				return pgv_host_fail(PGV_ERR_STATE, "device batch ended early"), -1;
		}
		else if (fetch_whole_batch(so) != PGV_OK)
			return -1;
	}
	if (so->whole)
	{
		int64_t		i = so->order[so->next++];

		*out_tid = so->img->tids[so->slot[i]];
		if (out_distance)
			*out_distance = (double) so->dist[i];
	}
	else
	{
		int64_t		i = so->next++ - so->win_base;

		*out_tid = so->img->tids[so->win_slot[i]];
		if (out_distance)
			*out_distance = (double) so->win_dist[i];
	}
	return 1;
}

void
pgv_host_ivf_endscan(pgv_ivf_scan * so)
{
	if (!so)
		return;
	if (so->dq)
		pgv_query_end(so->dq);
	free(so->value);
	free(so->lists);
	free(so->dist);
	free(so->slot);
	free(so->order);
	free(so);
}
