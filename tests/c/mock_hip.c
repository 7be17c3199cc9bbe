/*
 * mock_hip.c -- a CPU stand-in for the entry points of libpgv_hip.so that the C host glue
 * (pgvector_amd/host) calls, so that the HOST LOGIC -- page formats, scan driver, build loops,
 * mirror lifecycle -- can be tested where there is no GPU.
 *
 * TEST INFRASTRUCTURE ONLY.  It is linked into tests/c/host_logic_driver ahead of the real
 * library so that its symbols win; it never ships, the product has no CPU path.  Plain loops in
 * fp32, the simplest implementation that honours each entry point's contract in
 * include/pgv_hip.h; parity of the distances themselves is the GPU tests' business.
 */
#include <errno.h>
#include <math.h>
#include <signal.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "pgv_hip.h"

static char mock_err[256];

const char *
pgv_last_error(void)
{
	return mock_err;
}

static int
fail(int code, const char *msg)
{
	snprintf(mock_err, sizeof(mock_err), "%s", msg);
	return code;
}

struct pgv_ctx
{
	int			dummy;
};

/*
 * "Device memory".  By default the heap; with mock_hip_set_arena a bump allocator inside a MAP_SHARED mapping the
 * caller made BEFORE forking its processes: an index "uploaded" by one of them is then addressable in all of them
 * (same address), which is what lets tests/c/ext_driver.c run the worker-stages / backends-import model of
 * ext/pgv_context.c without a GPU.  Arena memory is never reused (a test's worth of uploads fits).
 */
static int	live_queries = 0;	/* pgv_query handles of this process that have not been ended */

int
mock_hip_live_queries(void)
{
	return live_queries;
}

static int	contexts_made[16];	/* contexts this process has created, by device */

int
mock_hip_contexts_made(int device)
{
	return device >= 0 && device < 16 ? contexts_made[device] : 0;
}

/* (MOCK_HIP_DEVICES: the stand-in node has that many devices -- contexts on any of them are the same host memory) */
int
pgv_device_count(void)
{
	const char *e = getenv("MOCK_HIP_DEVICES");
	int			n = e ? atoi(e) : 1;

	return n < 1 ? 1 : (n > 16 ? 16 : n);
}

int
pgv_device_memory(int device, uint64_t *free_bytes, uint64_t *total_bytes)
{
	(void) device;
	*free_bytes = *total_bytes = (uint64_t) 1 << 40;
	return PGV_OK;
}

static char *arena_base = NULL;
static size_t arena_bytes = 0;

void
mock_hip_set_arena(void *base, size_t bytes)
{
	arena_base = base;
	arena_bytes = bytes;
}

static void *
dev_alloc(size_t bytes)
{
	size_t		at;

	if (bytes == 0)
		bytes = 16;
	if (!arena_base)
		return calloc(1, bytes);
	bytes = (bytes + 63) & ~(size_t) 63;
	at = __atomic_fetch_add((size_t *) arena_base, bytes, __ATOMIC_SEQ_CST) + 64;	/* the first 64 bytes hold the cursor */
	if (at + bytes > arena_bytes)
	{
		fprintf(stderr, "mock_hip: arena exhausted\n");
		abort();
	}
	memset(arena_base + at, 0, bytes);
	return arena_base + at;
}

static void
dev_free(void *p)
{
	if (p && !(arena_base && (char *) p >= arena_base && (char *) p < arena_base + arena_bytes))
		free(p);
}

/* ---- fp16 at the door: the stand-in computes in fp32 only.  Halves are widened (exactly) where they come in -- index and
 * element uploads, centers, rows, samples, queries -- and narrowed (round to nearest even, src/halfutils.h:146-233) where
 * vectors go out (k-means centers, drained rows); an index / graph remembers that its callers speak fp16. */
static float
half_to_float(uint16_t h)
{
	uint32_t	sign = (uint32_t) (h & 0x8000u) << 16,
				exp = (h >> 10) & 0x1fu,
				man = h & 0x3ffu,
				bits;
	float		f;

	if (exp == 0)
	{
		if (man == 0)
			bits = sign;
		else
		{
			/* subnormal half: normalise */
			exp = 127 - 15 + 1;
			while (!(man & 0x400u))
			{
				man <<= 1;
				exp--;
			}
			bits = sign | (exp << 23) | ((man & 0x3ffu) << 13);
		}
	}
	else if (exp == 31)
		bits = sign | 0x7f800000u | (man << 13);
	else
		bits = sign | ((exp + 127 - 15) << 23) | (man << 13);
	memcpy(&f, &bits, 4);
	return f;
}

static uint16_t
float_to_half(float f)
{
	uint32_t	bits;
	uint32_t	sign,
				man;
	int			exp;

	memcpy(&bits, &f, 4);
	sign = (bits >> 16) & 0x8000u;
	exp = (int) ((bits >> 23) & 0xffu) - 127 + 15;
	man = bits & 0x7fffffu;
	if (((bits >> 23) & 0xffu) == 0xffu)
		return (uint16_t) (sign | 0x7c00u | (man ? 0x200u | (man >> 13) : 0));	/* inf / nan */
	if (exp >= 31)
		return (uint16_t) (sign | 0x7c00u);	/* overflow: inf */
	if (exp <= 0)
	{
		uint32_t	m;
		int			shift;

		if (exp < -10)
			return (uint16_t) sign;	/* underflow: zero */
		m = man | 0x800000u;
		shift = 14 - exp;		/* to a 10-bit subnormal mantissa */
		{
			uint32_t	q = m >> shift,
						rem = m & ((1u << shift) - 1),
						half = 1u << (shift - 1);

			if (rem > half || (rem == half && (q & 1)))
				q++;
			return (uint16_t) (sign | q);
		}
	}
	{
		uint32_t	q = ((uint32_t) exp << 10) | (man >> 13),
					rem = man & 0x1fffu;

		if (rem > 0x1000u || (rem == 0x1000u && (q & 1)))
			q++;				/* (a carry out of the mantissa bumps the exponent, up to inf: the right answer) */
		return (uint16_t) (sign | q);
	}
}

static float *
widen(const void *halves, size_t count)
{
	float	   *out = malloc(sizeof(float) * (count > 0 ? count : 1));

	for (size_t i = 0; i < count; i++)
		out[i] = half_to_float(((const uint16_t *) halves)[i]);
	return out;
}

struct pgv_index
{
	int			f16;			/* its callers hand in and take out halves */
	pgv_metric	metric;
	int			dim,
				nlists;
	int64_t		n;
	float	   *centers,
			   *vectors;
	int64_t    *offsets;
	uint64_t   *tids;
	int		   *refs;			/* pgv_index_share: handles on the same arrays */
};

struct pgv_hnsw
{
	int			f16;
	pgv_metric	metric;
	int			dim;
	int64_t		n;
	float	   *vectors;
	int			m;
	int32_t		entry;
	int32_t    *levels;
	int64_t    *nbr_start;
	int32_t    *nbr;
	char	   *payload;		/* [n x payload_bytes] (pgv_hnsw_upload_payload) */
	int			payload_bytes;
	/* pgv_hnsw_link_begin .. _end: distance and closer flag (bit 0; bit 1 of a list's first slot: closerSet) per tuple slot,
	 * and the batch handed to pgv_hnsw_link_prepare */
	float	   *nb_dist;
	uint8_t    *nb_flag;
	int			link_nq,
				link_lcap;
	int32_t    *link_elems,
			   *link_sel_ids,
			   *link_sel_cnt;
	uint8_t    *link_linked,
			   *link_sel_closer;
	float	   *link_sel_dist;
	int64_t		link_pairs2,
				link_deferred;	/* sums over the build, handed out by pgv_hnsw_link_end */
	/* pgv_hnsw_build_search_keep: the candidate lists of two batches */
	struct
	{
		int			nq,
					ef,
					lcap;
		int32_t    *ids,
				   *cnt,
				   *levels;
		float	   *dist;
	}			kept[2];
	int			imported;		/* a view made by pgv_hnsw_import / pgv_hnsw_share: frees nothing but itself */
	struct pgv_hnsw *view_of;	/* pgv_hnsw_share: the owner whose graph / entry point it follows */
};

int
pgv_ctx_create(int device, void *stream, pgv_ctx * *out)
{
	(void) device;
	(void) stream;
	*out = NULL;
	/* test knob: a process without a usable device (none installed, lost, driver not initialising) */
	if (getenv("MOCK_HIP_NO_DEVICE"))
		return fail(PGV_ERR_DEVICE, "mock: no HIP device");
	if (device < 0 || device >= pgv_device_count())
		return fail(PGV_ERR_DEVICE, "mock: no such device");
	__atomic_add_fetch(&contexts_made[device], 1, __ATOMIC_SEQ_CST);
	*out = calloc(1, sizeof(pgv_ctx));
	return PGV_OK;
}

void
pgv_ctx_destroy(pgv_ctx * ctx)
{
	free(ctx);
}

static float
dist(pgv_metric metric, int dim, const float *a, const float *b)
{
	float		s = 0.0f;

	for (int i = 0; i < dim; i++)
	{
		if (metric == PGV_L2SQ)
			s += (a[i] - b[i]) * (a[i] - b[i]);
		else if (metric == PGV_NEG_IP)
			s += a[i] * b[i];
		else
			s += fabsf(a[i] - b[i]);
	}
	return metric == PGV_NEG_IP ? -s : s;
}

/* ------------------------------------------------------------------ IVFFlat */

int
pgv_index_upload(pgv_ctx * ctx, pgv_metric metric, pgv_dtype dtype, int dim, int nlists,
				 const void *centers, const int64_t *list_offsets, const void *vectors, const uint64_t *tids,
				 pgv_index * *out)
{
	pgv_index  *ix;
	int64_t		n = list_offsets[nlists];

	(void) ctx;
	if (dtype == PGV_F16)
	{
		float	   *wc = widen(centers, (size_t) nlists * dim),
				   *wv = widen(vectors, (size_t) n * dim);
		int			rc = pgv_index_upload(ctx, metric, PGV_F32, dim, nlists, wc, list_offsets, wv, tids, out);

		free(wc);
		free(wv);
		if (rc == PGV_OK)
			(*out)->f16 = 1;
		return rc;
	}
	ix = dev_alloc(sizeof(*ix));
	ix->refs = dev_alloc(sizeof(int));
	*ix->refs = 1;
	if (tids)
	{
		ix->tids = dev_alloc(sizeof(uint64_t) * (size_t) (n > 0 ? n : 1));
		memcpy(ix->tids, tids, sizeof(uint64_t) * (size_t) n);
	}
	ix->metric = metric;
	ix->dim = dim;
	ix->nlists = nlists;
	ix->n = n;
	ix->centers = dev_alloc(sizeof(float) * (size_t) nlists * dim);
	memcpy(ix->centers, centers, sizeof(float) * (size_t) nlists * dim);
	ix->offsets = dev_alloc(sizeof(int64_t) * (size_t) (nlists + 1));
	memcpy(ix->offsets, list_offsets, sizeof(int64_t) * (size_t) (nlists + 1));
	ix->vectors = dev_alloc(sizeof(float) * (size_t) (n > 0 ? n : 1) * dim);
	memcpy(ix->vectors, vectors, sizeof(float) * (size_t) n * dim);
	*out = ix;
	return PGV_OK;
}

int
pgv_index_share(pgv_index * ix, pgv_ctx * ctx, pgv_index * *out)
{
	pgv_index  *v = malloc(sizeof(*v));

	(void) ctx;
	*v = *ix;
	__atomic_add_fetch(ix->refs, 1, __ATOMIC_RELAXED);
	*out = v;
	return PGV_OK;
}

int
pgv_pinned_alloc(size_t bytes, void **out)
{
	*out = malloc(bytes ? bytes : 16);
	return *out ? PGV_OK : fail(PGV_ERR_NOMEM, "mock: out of memory");
}

void
pgv_pinned_free(void *p)
{
	free(p);
}

int
pgv_pinned_register(void *p, size_t bytes)
{
	return p && bytes ? PGV_OK : fail(PGV_ERR_ARG, "mock: empty range");
}

void
pgv_pinned_unregister(void *p)
{
	(void) p;
}

/* the stand-in's "IPC handle" is the address of the index: valid in a process forked from the exporter (its pages
 * are the child's too), which is how the host-logic tests run the multi-process pooler without a GPU */
typedef struct
{
	uint64_t	magic;
	uint32_t	pid;
	pgv_index  *ix;
}			mock_handle;

int
pgv_index_export(pgv_index * ix, pgv_index_handle * out)
{
	mock_handle h = {0x6d6f636b6978ull, (uint32_t) getpid(), ix};
	static int	exports = 0;
	const char *every = getenv("MOCK_HIP_EXPORT_FAIL_EVERY");

	/* test knob: the export that fails once in a while on the real device (hipIpcGetMemHandle: invalid argument) */
	if (every && atoi(every) > 0 && ++exports % atoi(every) == 0)
		return fail(PGV_ERR_DEVICE, "mock: hipIpcGetMemHandle failed: invalid argument");

	memset(out, 0, sizeof(*out));
	memcpy(out->bytes, &h, sizeof(h));
	return PGV_OK;
}

int
pgv_index_import(pgv_ctx * ctx, const pgv_index_handle * handle, pgv_index * *out)
{
	mock_handle h;

	memcpy(&h, handle->bytes, sizeof(h));
	*out = NULL;
	if (h.magic != 0x6d6f636b6978ull)
		return fail(PGV_ERR_ARG, "mock: not a handle");
	if (h.pid == (uint32_t) getpid())
		return fail(PGV_ERR_STATE, "mock: the handle was exported by this process (use pgv_index_share)");
	/* like hipIpcOpenMemHandle: an exporter that is gone took its allocations with it */
	if (kill((pid_t) h.pid, 0) != 0 && errno == ESRCH)
		return fail(PGV_ERR_DEVICE, "mock: the exporting process is gone");
	return pgv_index_share(h.ix, ctx, out);
}

void
pgv_index_free(pgv_index * ix)
{
	if (!ix)
		return;
	if (__atomic_sub_fetch(ix->refs, 1, __ATOMIC_ACQ_REL) > 0)
	{
		dev_free(ix);			/* a view's struct is this process's own; the owner's may sit in the arena (views live on) */
		return;
	}
	dev_free(ix->refs);
	dev_free(ix->tids);
	dev_free(ix->centers);
	dev_free(ix->vectors);
	dev_free(ix->offsets);
	dev_free(ix);
}

int
pgv_index_tids(pgv_index * ix, const int64_t *slots, int64_t n, uint64_t *out)
{
	if (!ix->tids)
		return fail(PGV_ERR_STATE, "mock: index was uploaded without tids");
	for (int64_t i = 0; i < n; i++)
		out[i] = slots[i] >= 0 && slots[i] < ix->n ? ix->tids[slots[i]] : ~(uint64_t) 0;
	return PGV_OK;
}

int64_t
pgv_index_rows(const pgv_index * ix)
{
	return ix->n;
}

int
pgv_index_lists(const pgv_index * ix)
{
	return ix->nlists;
}

/* GetScanLists: the maxprobes nearest centers, ascending; a later center never displaces an equal one */
static int
rank_lists_f32(pgv_index * ix, const void *queries, int nq, int maxprobes, int32_t *out_lists, float *out_dist)
{
	float	   *d = malloc(sizeof(float) * (size_t) ix->nlists);
	uint8_t    *used = malloc((size_t) ix->nlists);

	for (int q = 0; q < nq; q++)
	{
		const float *qv = (const float *) queries + (size_t) q * ix->dim;

		for (int l = 0; l < ix->nlists; l++)
			d[l] = dist(ix->metric, ix->dim, ix->centers + (size_t) l * ix->dim, qv);
		memset(used, 0, (size_t) ix->nlists);
		for (int p = 0; p < maxprobes; p++)
		{
			int			best = -1;

			for (int l = 0; l < ix->nlists; l++)
				if (!used[l] && (best < 0 || d[l] < d[best]))
					best = l;
			used[best] = 1;
			out_lists[(size_t) q * maxprobes + p] = best;
			if (out_dist)
				out_dist[(size_t) q * maxprobes + p] = d[best];
		}
	}
	free(d);
	free(used);
	return PGV_OK;
}

/* GetScanItems without the sort: every tuple of the given lists, in list order */
static int
scan_lists_f32(pgv_index * ix, const void *query, const int32_t *lists, int nlists, float *out_dist,
			   int64_t *out_slot, int64_t capacity, int64_t *out_count)
{
	int64_t		m = 0;

	for (int i = 0; i < nlists; i++)
		m += ix->offsets[lists[i] + 1] - ix->offsets[lists[i]];
	*out_count = m;
	if (m > capacity)
		return fail(PGV_ERR_ARG, "mock: capacity");
	m = 0;
	for (int i = 0; i < nlists; i++)
		for (int64_t r = ix->offsets[lists[i]]; r < ix->offsets[lists[i] + 1]; r++, m++)
		{
			out_slot[m] = r;
			out_dist[m] = query ? dist(ix->metric, ix->dim, ix->vectors + (size_t) r * ix->dim, (const float *) query) : 0.0f;
		}
	return PGV_OK;
}

/* GetScanLists + GetScanItems + the head of the ascending sort for every query (stable: stream order on ties) */
/* how many pgv_search_batch calls of this process ran at the same time, at most (the pooler's lanes are threads of one
 * process in pgv_host_pool_create: "one scan at a time" is observable here); mock_hip_search_delay_us stretches a call */
static int	searches_now,
			searches_peak;
int			mock_hip_search_delay_us = 0;

int
mock_hip_search_peak(int reset)
{
	int			peak = __atomic_load_n(&searches_peak, __ATOMIC_ACQUIRE);

	if (reset)
		__atomic_store_n(&searches_peak, 0, __ATOMIC_RELEASE);
	return peak;
}

static int
search_batch_f32(pgv_index * ix, const void *queries, int nq, int probes, int k, float *out_dist, int64_t *out_slot,
				 uint64_t *out_tid)
{
	int32_t    *lists = malloc(sizeof(int32_t) * (size_t) probes);
	int			now = __atomic_add_fetch(&searches_now, 1, __ATOMIC_ACQ_REL);
	int			peak = __atomic_load_n(&searches_peak, __ATOMIC_RELAXED);

	while (now > peak && !__atomic_compare_exchange_n(&searches_peak, &peak, now, 0, __ATOMIC_ACQ_REL, __ATOMIC_RELAXED))
		;
	if (mock_hip_search_delay_us > 0)
		usleep((useconds_t) mock_hip_search_delay_us);
	float	   *d = malloc(sizeof(float) * (size_t) (ix->n > 0 ? ix->n : 1));
	int64_t    *s = malloc(sizeof(int64_t) * (size_t) (ix->n > 0 ? ix->n : 1));

	for (int q = 0; q < nq; q++)
	{
		const float *qv = (const float *) queries + (size_t) q * ix->dim;
		int64_t		m;

		rank_lists_f32(ix, qv, 1, probes, lists, NULL);
		scan_lists_f32(ix, qv, lists, probes, d, s, ix->n, &m);
		for (int r = 0; r < k; r++)
		{
			int64_t		best = -1;

			for (int64_t i = 0; i < m; i++)
				if (s[i] >= 0 && (best < 0 || d[i] < d[best]))
					best = i;
			out_dist[(size_t) q * k + r] = best >= 0 ? d[best] : INFINITY;
			if (out_slot)
				out_slot[(size_t) q * k + r] = best >= 0 ? s[best] : -1;
			if (out_tid)
				out_tid[(size_t) q * k + r] = best >= 0 && ix->tids ? ix->tids[s[best]] : ~(uint64_t) 0;
			if (best >= 0)
				s[best] = -1 - s[best];	/* taken */
		}
	}
	free(lists);
	free(d);
	free(s);
	__atomic_sub_fetch(&searches_now, 1, __ATOMIC_ACQ_REL);
	return PGV_OK;
}

/* pgv_query_*: one query's scan state; the sorted stream of the current batch is kept whole */
/* the public three: halves widened at the door */
int
pgv_rank_lists(pgv_index * ix, const void *queries, int nq, int maxprobes, int32_t *out_lists, float *out_dist)
{
	float	   *w = ix->f16 && queries ? widen(queries, (size_t) nq * ix->dim) : NULL;
	int			rc = rank_lists_f32(ix, w ? w : queries, nq, maxprobes, out_lists, out_dist);

	free(w);
	return rc;
}

int
pgv_scan_lists(pgv_index * ix, const void *query, const int32_t *lists, int nlists, float *out_dist,
			   int64_t *out_slot, int64_t capacity, int64_t *out_count)
{
	float	   *w = ix->f16 && query ? widen(query, (size_t) ix->dim) : NULL;
	int			rc = scan_lists_f32(ix, w ? w : query, lists, nlists, out_dist, out_slot, capacity, out_count);

	free(w);
	return rc;
}

int
pgv_search_batch(pgv_index * ix, const void *queries, int nq, int probes, int k, float *out_dist, int64_t *out_slot,
				 uint64_t *out_tid)
{
	float	   *w = ix->f16 && queries ? widen(queries, (size_t) nq * ix->dim) : NULL;
	int			rc = search_batch_f32(ix, w ? w : queries, nq, probes, k, out_dist, out_slot, out_tid);

	free(w);
	return rc;
}

struct pgv_query
{
	pgv_index  *ix;
	float	   *q;
	int			is_null,
				max_probes;
	int32_t    *lists;
	int64_t		m;
	float	   *sd;
	int64_t    *ss;
};

int
pgv_query_begin(pgv_index * ix, pgv_query * *out)
{
	pgv_query  *q = calloc(1, sizeof(*q));

	live_queries++;
	q->ix = ix;
	q->q = malloc(sizeof(float) * (size_t) ix->dim);
	q->lists = malloc(sizeof(int32_t) * (size_t) ix->nlists);
	*out = q;
	return PGV_OK;
}

void
pgv_query_end(pgv_query * q)
{
	if (!q)
		return;
	live_queries--;
	free(q->q);
	free(q->lists);
	free(q->sd);
	free(q->ss);
	free(q);
}

int
pgv_query_rank(pgv_query * q, const void *query, int max_probes)
{
	if (max_probes > 1024)
		return fail(PGV_ERR_ARG, "mock: max_probes");
	q->is_null = query == NULL;
	q->max_probes = max_probes;
	if (!query)
	{
		for (int i = 0; i < max_probes; i++)
			q->lists[i] = i;
		return PGV_OK;
	}
	if (q->ix->f16)
		for (int i = 0; i < q->ix->dim; i++)
			q->q[i] = half_to_float(((const uint16_t *) query)[i]);
	else
		memcpy(q->q, query, sizeof(float) * (size_t) q->ix->dim);
	return rank_lists_f32(q->ix, q->q, 1, max_probes, q->lists, NULL);
}

int
pgv_query_lists(pgv_query * q, int32_t *out_lists, int n)
{
	memcpy(out_lists, q->lists, sizeof(int32_t) * (size_t) n);
	return PGV_OK;
}

static int
emit(pgv_query * q, int skip, int count, float *out_dist, int64_t *out_slot, uint64_t *out_tid, int *out_count)
{
	int			n = 0;

	for (int64_t i = skip; i < q->m && n < count; i++, n++)
	{
		if (out_dist)
			out_dist[n] = q->sd[i];
		if (out_slot)
			out_slot[n] = q->ss[i];
		if (out_tid)
			out_tid[n] = q->ix->tids ? q->ix->tids[q->ss[i]] : ~(uint64_t) 0;
	}
	*out_count = n;
	return PGV_OK;
}

int
pgv_query_scan(pgv_query * q, int first, int nprobes, int head, float *out_dist, int64_t *out_slot,
			   uint64_t *out_tid, int *out_count, int64_t *out_total)
{
	int64_t		cap = 0,
				got;

	if (nprobes > 256 || head > 1024)
		return fail(PGV_ERR_ARG, "mock: limits");
	for (int i = 0; i < nprobes; i++)
		cap += q->ix->offsets[q->lists[first + i] + 1] - q->ix->offsets[q->lists[first + i]];
	free(q->sd);
	free(q->ss);
	q->sd = malloc(sizeof(float) * (size_t) (cap + 1));
	q->ss = malloc(sizeof(int64_t) * (size_t) (cap + 1));
	scan_lists_f32(q->ix, q->is_null ? NULL : q->q, q->lists + first, nprobes, q->sd, q->ss, cap, &got);
	/* stable insertion sort: ascending distance, insertion order on ties */
	for (int64_t i = 1; i < got; i++)
	{
		float		d = q->sd[i];
		int64_t		s = q->ss[i],
					j = i - 1;

		while (j >= 0 && q->sd[j] > d)
		{
			q->sd[j + 1] = q->sd[j];
			q->ss[j + 1] = q->ss[j];
			j--;
		}
		q->sd[j + 1] = d;
		q->ss[j + 1] = s;
	}
	q->m = got;
	if (out_total)
		*out_total = got;
	return emit(q, 0, head, out_dist, out_slot, out_tid, out_count);
}

int
pgv_query_more(pgv_query * q, int skip, int count, float *out_dist, int64_t *out_slot, uint64_t *out_tid,
			   int *out_count)
{
	if (skip + count > 1024)
		return fail(PGV_ERR_ARG, "mock: depth");
	return emit(q, skip, count, out_dist, out_slot, out_tid, out_count);
}

/* AddTupleToSort's argmin: the first strictly smallest distance wins */
int
pgv_assign(pgv_ctx * ctx, pgv_metric metric, pgv_dtype dtype, int dim, const void *centers, int k,
		   const void *rows, int64_t n, int32_t *out_list, float *out_dist)
{
	(void) ctx;
	if (dtype == PGV_F16)
	{
		float	   *wc = widen(centers, (size_t) k * dim),
				   *wr = widen(rows, (size_t) n * dim);
		int			rc = pgv_assign(ctx, metric, PGV_F32, dim, wc, k, wr, n, out_list, out_dist);

		free(wc);
		free(wr);
		return rc;
	}
	for (int64_t i = 0; i < n; i++)
	{
		int			best = 0;
		float		bd = INFINITY;

		for (int c = 0; c < k; c++)
		{
			float		d = dist(metric, dim, (const float *) rows + (size_t) i * dim, (const float *) centers + (size_t) c * dim);

			if (d < bd)
			{
				bd = d;
				best = c;
			}
		}
		out_list[i] = best;
		if (out_dist)
			out_dist[i] = bd;
	}
	return PGV_OK;
}

/* the build's tuplesort on the "device": rows kept in heap order, assigned, then ordered by list (stable) */
struct pgv_builder
{
	pgv_ctx    *ctx;
	pgv_metric	metric;
	int			dim,
				nlists;
	float	   *centers,
			   *rows;
	uint64_t   *tids;
	int32_t    *lists;
	int64_t		n,
				cap;
	int			has_tids;
	int			deferred;
	int			f16;
	int64_t		assigned;
};

int
pgv_builder_begin(pgv_ctx * ctx, pgv_metric metric, pgv_dtype dtype, int dim, int nlists, const void *centers,
				  int64_t expected_rows, pgv_builder * *out)
{
	pgv_builder *b;

	(void) expected_rows;
	b = calloc(1, sizeof(*b));
	b->ctx = ctx;
	b->metric = metric;
	b->dim = dim;
	b->nlists = nlists;
	b->f16 = dtype == PGV_F16;
	b->centers = malloc(sizeof(float) * (size_t) nlists * dim);
	if (centers && b->f16)
		for (size_t i = 0; i < (size_t) nlists * dim; i++)
			b->centers[i] = half_to_float(((const uint16_t *) centers)[i]);
	else if (centers)
		memcpy(b->centers, centers, sizeof(float) * (size_t) nlists * dim);
	else
		b->deferred = 1;
	*out = b;
	return PGV_OK;
}

static void
builder_assign_pending(pgv_builder * b)
{
	if (b->assigned < b->n)
		pgv_assign(b->ctx, b->metric, PGV_F32, b->dim, b->centers, b->nlists, b->rows + (size_t) b->assigned * b->dim,
				   b->n - b->assigned, b->lists + b->assigned, NULL);
	b->assigned = b->n;
}

int
pgv_builder_set_centers(pgv_builder * b, const void *centers)
{
	if (!b->deferred)
		return fail(PGV_ERR_STATE, "mock: the builder has its centers");
	if (b->f16)
		for (size_t i = 0; i < (size_t) b->nlists * b->dim; i++)
			b->centers[i] = half_to_float(((const uint16_t *) centers)[i]);
	else
		memcpy(b->centers, centers, sizeof(float) * (size_t) b->nlists * b->dim);
	b->deferred = 0;
	return PGV_OK;
}

int
pgv_builder_add(pgv_builder * b, const void *rows, const uint64_t *tids, int64_t n)
{
	if (b->n + n > b->cap)
	{
		b->cap = (b->n + n) * 2;
		b->rows = realloc(b->rows, sizeof(float) * (size_t) b->cap * b->dim);
		b->tids = realloc(b->tids, sizeof(uint64_t) * (size_t) b->cap);
		b->lists = realloc(b->lists, sizeof(int32_t) * (size_t) b->cap);
	}
	if (b->f16)
		for (size_t i = 0; i < (size_t) n * b->dim; i++)
			b->rows[(size_t) b->n * b->dim + i] = half_to_float(((const uint16_t *) rows)[i]);
	else
		memcpy(b->rows + (size_t) b->n * b->dim, rows, sizeof(float) * (size_t) n * b->dim);
	for (int64_t i = 0; i < n; i++)
		b->tids[b->n + i] = tids ? tids[i] : (uint64_t) (b->n + i);
	b->has_tids = tids != NULL;
	b->n += n;
	if (!b->deferred)
		builder_assign_pending(b);
	return PGV_OK;
}

int64_t
pgv_builder_rows(const pgv_builder * b)
{
	return b->n;
}

int
pgv_builder_finish(pgv_builder * b, pgv_index * *out_index, int64_t *out_offsets, int32_t *out_lists)
{
	int64_t    *off = calloc((size_t) b->nlists + 1, sizeof(int64_t));
	int64_t    *fill = malloc(sizeof(int64_t) * (size_t) b->nlists);
	float	   *sorted = malloc(sizeof(float) * (size_t) (b->n > 0 ? b->n : 1) * b->dim);
	uint64_t   *stids = malloc(sizeof(uint64_t) * (size_t) (b->n > 0 ? b->n : 1));
	int			rc;

	if (b->deferred)
	{
		free(stids);
		free(sorted);
		free(fill);
		free(off);
		return fail(PGV_ERR_STATE, "mock: no centers");
	}
	builder_assign_pending(b);
	for (int64_t r = 0; r < b->n; r++)
		off[b->lists[r] + 1]++;
	for (int l = 0; l < b->nlists; l++)
		off[l + 1] += off[l];
	memcpy(fill, off, sizeof(int64_t) * (size_t) b->nlists);
	for (int64_t r = 0; r < b->n; r++)
	{
		int64_t		d = fill[b->lists[r]]++;

		memcpy(sorted + (size_t) d * b->dim, b->rows + (size_t) r * b->dim, sizeof(float) * (size_t) b->dim);
		stids[d] = b->tids[r];
	}
	rc = pgv_index_upload(b->ctx, b->metric, PGV_F32, b->dim, b->nlists, b->centers, off, sorted, stids, out_index);
	if (rc == PGV_OK)
		(*out_index)->f16 = b->f16;
	if (rc == PGV_OK && out_offsets)
		memcpy(out_offsets, off, sizeof(int64_t) * ((size_t) b->nlists + 1));
	if (rc == PGV_OK && out_lists)
		memcpy(out_lists, b->lists, sizeof(int32_t) * (size_t) b->n);
	free(stids);
	free(sorted);
	free(fill);
	free(off);
	b->n = b->assigned = 0;
	return rc;
}

void
pgv_builder_free(pgv_builder * b)
{
	if (!b)
		return;
	free(b->centers);
	free(b->rows);
	free(b->tids);
	free(b->lists);
	free(b);
}

int
pgv_index_drain(pgv_index * ix, int64_t chunk_rows, pgv_rows_sink sink, void *arg)
{
	if (chunk_rows <= 0)
		chunk_rows = 1000;		/* several pieces even for the small test indexes */
	for (int64_t r0 = 0; r0 < ix->n; r0 += chunk_rows)
	{
		int64_t		cnt = ix->n - r0 < chunk_rows ? ix->n - r0 : chunk_rows;

		uint16_t   *narrow = NULL;
		int			stop;

		if (ix->f16)
		{
			narrow = malloc(sizeof(uint16_t) * (size_t) cnt * ix->dim);
			for (size_t i = 0; i < (size_t) cnt * ix->dim; i++)
				narrow[i] = float_to_half(ix->vectors[(size_t) r0 * ix->dim + i]);	/* (exact: they came in as halves) */
		}
		stop = sink(arg, r0, cnt, narrow ? (const void *) narrow : (const void *) (ix->vectors + (size_t) r0 * ix->dim),
					ix->tids ? ix->tids + r0 : NULL);
		free(narrow);
		if (stop != 0)
			return fail(PGV_ERR_STATE, "mock: the sink stopped the drain");
	}
	return PGV_OK;
}

/* a plain Lloyd k-means from evenly spaced samples: enough for the build driver's plumbing */
int
pgv_kmeans(pgv_ctx * ctx, pgv_ops ops, pgv_dtype dtype, int dim, const void *samples, int n, int k,
		   int max_iterations, const pgv_rng * rng, void *out_centers, int32_t *out_closest, int *out_iters);

static int
kmeans_f16(pgv_ctx * ctx, pgv_ops ops, int dim, const void *samples, int n, int k, int max_iterations, const pgv_rng * rng,
		   void *out_centers, int32_t *out_closest, int *out_iters)
{
	/* samples widened, the centers rounded to halves on the way out (HalfvecUpdateCenter, src/ivfutils.c:340-361) */
	float	   *ws = widen(samples, (size_t) n * dim),
			   *wc = malloc(sizeof(float) * (size_t) k * dim);
	int			rc = pgv_kmeans(ctx, ops, PGV_F32, dim, ws, n, k, max_iterations, rng, wc, out_closest, out_iters);

	for (size_t i = 0; rc == PGV_OK && i < (size_t) k * dim; i++)
		((uint16_t *) out_centers)[i] = float_to_half(wc[i]);
	free(ws);
	free(wc);
	return rc;
}

int
pgv_kmeans(pgv_ctx * ctx, pgv_ops ops, pgv_dtype dtype, int dim, const void *samples, int n, int k,
		   int max_iterations, const pgv_rng * rng, void *out_centers, int32_t *out_closest, int *out_iters)
{
	float	   *c = out_centers;
	const float *s = samples;
	int32_t    *closest = malloc(sizeof(int32_t) * (size_t) (n > 0 ? n : 1));
	float	   *sum = malloc(sizeof(float) * (size_t) k * dim);
	int		   *cnt = malloc(sizeof(int) * (size_t) k);
	int			it = 0;

	/* inner product / cosine opclasses: spherical k-means (src/ivfkmeans.c:553-570 on samples the caller normalised):
	 * nearest center by inner product, new centers normalised */
	const int	sph = ops != PGV_OPS_L2;

	(void) ctx;
	(void) rng;
	if (dtype == PGV_F16)
	{
		free(closest);
		free(sum);
		free(cnt);
		return kmeans_f16(ctx, ops, dim, samples, n, k, max_iterations, rng, out_centers, out_closest, out_iters);
	}
	/* (PGV_OPS_COSINE: spherical like PGV_OPS_IP -- what differs is what the caller stores, include/pgv_hip.h:81-85) */
	if (getenv("MOCK_HIP_KMEANS_FAIL"))	/* test knob: the device call that fails */
	{
		free(closest);
		free(sum);
		free(cnt);
		return fail(PGV_ERR_DEVICE, "mock: pgv_kmeans was asked to fail");
	}
	for (int j = 0; j < k; j++)
		for (int d = 0; d < dim; d++)
			c[(size_t) j * dim + d] = n > 0 ? s[(size_t) ((int64_t) j * n / k) * dim + d] : (float) j;
	for (it = 0; it < (max_iterations < 20 ? max_iterations : 20) && n > 0; it++)
	{
		pgv_assign(NULL, sph ? PGV_NEG_IP : PGV_L2SQ, PGV_F32, dim, c, k, s, n, closest, NULL);
		memset(sum, 0, sizeof(float) * (size_t) k * dim);
		memset(cnt, 0, sizeof(int) * (size_t) k);
		for (int i = 0; i < n; i++)
		{
			cnt[closest[i]]++;
			for (int d = 0; d < dim; d++)
				sum[(size_t) closest[i] * dim + d] += s[(size_t) i * dim + d];
		}
		for (int j = 0; j < k; j++)
			if (cnt[j] > 0)
			{
				double		norm = 0.0;

				for (int d = 0; d < dim; d++)
				{
					c[(size_t) j * dim + d] = sum[(size_t) j * dim + d] / (float) cnt[j];
					norm += (double) c[(size_t) j * dim + d] * (double) c[(size_t) j * dim + d];
				}
				norm = sqrt(norm);
				for (int d = 0; d < dim && sph && norm > 0.0; d++)
					c[(size_t) j * dim + d] = (float) ((double) c[(size_t) j * dim + d] / norm);
			}
	}
	if (out_closest && n > 0)
		memcpy(out_closest, closest, sizeof(int32_t) * (size_t) n);
	if (out_iters)
		*out_iters = it;
	free(closest);
	free(sum);
	free(cnt);
	return PGV_OK;
}

/* --------------------------------------------------------------------- HNSW */

int
pgv_hnsw_upload(pgv_ctx * ctx, pgv_metric metric, pgv_dtype dtype, int dim, const void *elements, int64_t n,
				pgv_hnsw * *out)
{
	pgv_hnsw   *h;

	(void) ctx;
	h = dev_alloc(sizeof(*h));
	h->f16 = dtype == PGV_F16;
	h->metric = metric;
	h->dim = dim;
	h->n = n;
	h->entry = -1;
	h->vectors = dev_alloc(sizeof(float) * (size_t) (n > 0 ? n : 1) * dim);
	if (h->f16)
		for (size_t i = 0; i < (size_t) n * dim; i++)
			h->vectors[i] = half_to_float(((const uint16_t *) elements)[i]);
	else
		memcpy(h->vectors, elements, sizeof(float) * (size_t) n * dim);
	*out = h;
	return PGV_OK;
}

int
pgv_hnsw_upload_payload(pgv_ctx * ctx, pgv_metric metric, pgv_dtype dtype, int dim, const void *elements, int64_t n,
						const void *payload, int payload_bytes, pgv_hnsw * *out)
{
	int			rc = pgv_hnsw_upload(ctx, metric, dtype, dim, elements, n, out);

	if (rc != PGV_OK)
		return rc;
	(*out)->payload_bytes = payload_bytes;
	(*out)->payload = dev_alloc((size_t) (n > 0 ? n : 1) * (size_t) payload_bytes);
	memcpy((*out)->payload, payload, (size_t) n * (size_t) payload_bytes);
	return PGV_OK;
}

int
pgv_hnsw_get_payload(pgv_hnsw * h, const int64_t *elements, int n, void *out)
{
	if (!h->payload)
		return fail(PGV_ERR_STATE, "mock: mirror was uploaded without a payload");
	for (int i = 0; i < n; i++)
	{
		char	   *o = (char *) out + (size_t) i * (size_t) h->payload_bytes;

		if (elements[i] >= 0 && elements[i] < h->n)
			memcpy(o, h->payload + (size_t) elements[i] * (size_t) h->payload_bytes, (size_t) h->payload_bytes);
		else
			memset(o, 0, (size_t) h->payload_bytes);
	}
	return PGV_OK;
}

typedef struct
{
	uint64_t	magic;
	uint32_t	pid;
	pgv_hnsw   *h;
}			mock_hnsw_handle;

int
pgv_hnsw_export(pgv_hnsw * h, pgv_index_handle * out)
{
	mock_hnsw_handle e = {0x6d6f636b686e73ull, (uint32_t) getpid(), h};

	memset(out, 0, sizeof(*out));
	memcpy(out->bytes, &e, sizeof(e));
	return PGV_OK;
}

int
pgv_hnsw_import(pgv_ctx * ctx, const pgv_index_handle * handle, pgv_hnsw * *out)
{
	mock_hnsw_handle e;
	pgv_hnsw   *v;

	(void) ctx;
	memcpy(&e, handle->bytes, sizeof(e));
	*out = NULL;
	if (e.magic != 0x6d6f636b686e73ull)
		return fail(PGV_ERR_ARG, "mock: not an hnsw handle");
	if (e.pid == (uint32_t) getpid())
		return fail(PGV_ERR_STATE, "mock: the handle was exported by this process");
	if (kill((pid_t) e.pid, 0) != 0 && errno == ESRCH)
		return fail(PGV_ERR_DEVICE, "mock: the exporting process is gone");
	v = malloc(sizeof(*v));
	*v = *e.h;
	v->imported = 1;
	*out = v;
	return PGV_OK;
}

int
pgv_hnsw_device(const pgv_hnsw * h)
{
	return h ? 0 : -1;
}

/* a view for another context of this process: it follows its owner's graph and entry point */
int
pgv_hnsw_share(pgv_hnsw * h, pgv_ctx * ctx, pgv_hnsw * *out)
{
	pgv_hnsw   *v = malloc(sizeof(*v));

	(void) ctx;
	*v = *h;
	v->imported = 1;
	v->view_of = h->view_of ? h->view_of : h;
	*out = v;
	return PGV_OK;
}

static void
view_refresh(pgv_hnsw * h)
{
	if (h->view_of)
	{
		const pgv_hnsw *o = h->view_of;

		h->m = o->m;
		h->entry = o->entry;
		h->levels = o->levels;
		h->nbr_start = o->nbr_start;
		h->nbr = o->nbr;
	}
}

static void
graph_free(pgv_hnsw * h)
{
	dev_free(h->levels);
	dev_free(h->nbr_start);
	dev_free(h->nbr);
	h->levels = NULL;
	h->nbr_start = NULL;
	h->nbr = NULL;
}

void
pgv_hnsw_free(pgv_hnsw * h)
{
	if (!h)
		return;
	if (h->imported)
	{
		free(h);
		return;
	}
	graph_free(h);
	dev_free(h->vectors);
	dev_free(h->payload);
	dev_free(h);
}

int
pgv_hnsw_set_graph(pgv_hnsw * h, int m, int32_t entry, const int32_t *levels, const int64_t *nbr_start,
				   const int32_t *nbr)
{
	graph_free(h);
	h->m = m;
	h->entry = entry;
	h->levels = dev_alloc(sizeof(int32_t) * (size_t) (h->n > 0 ? h->n : 1));
	h->nbr_start = dev_alloc(sizeof(int64_t) * (size_t) (h->n + 1));
	memcpy(h->levels, levels, sizeof(int32_t) * (size_t) h->n);
	memcpy(h->nbr_start, nbr_start, sizeof(int64_t) * (size_t) (h->n + 1));
	h->nbr = dev_alloc(sizeof(int32_t) * (size_t) (nbr_start[h->n] > 0 ? nbr_start[h->n] : 1));
	memcpy(h->nbr, nbr, sizeof(int32_t) * (size_t) nbr_start[h->n]);
	return PGV_OK;
}

int
pgv_hnsw_update_graph(pgv_hnsw * h, int32_t entry, const int32_t *elements, int nupd, const int64_t *tuple_offsets,
					  const int32_t *tuples)
{
	view_refresh(h);
	h->entry = entry;
	if (h->view_of)
		h->view_of->entry = entry;	/* through a view the patch lands in the owner's graph */
	for (int i = 0; i < nupd; i++)
	{
		int32_t		e = elements[i];
		int64_t		len = h->nbr_start[e + 1] - h->nbr_start[e];

		if (tuple_offsets[i + 1] - tuple_offsets[i] != len)
			return fail(PGV_ERR_ARG, "mock: tuple size");
		memcpy(h->nbr + h->nbr_start[e], tuples + tuple_offsets[i], sizeof(int32_t) * (size_t) len);
	}
	return PGV_OK;
}

static int hnsw_score_f32(pgv_hnsw * h, const void *queries, int nq, const int32_t *slot, const int32_t *query_of, int64_t npairs,
						  float *out);

int
pgv_hnsw_score(pgv_hnsw * h, const void *queries, int nq, const int32_t *slot, const int32_t *query_of,
			   int64_t npairs, float *out)
{
	float	   *w = h->f16 && queries ? widen(queries, (size_t) nq * h->dim) : NULL;
	int			rc = hnsw_score_f32(h, w ? w : queries, nq, slot, query_of, npairs, out);

	free(w);
	return rc;
}

static int
hnsw_score_f32(pgv_hnsw * h, const void *queries, int nq, const int32_t *slot, const int32_t *query_of,
			   int64_t npairs, float *out)
{
	(void) nq;
	for (int64_t i = 0; i < npairs; i++)
		out[i] = dist(h->metric, h->dim, h->vectors + (size_t) slot[i] * h->dim,
					  (const float *) queries + (size_t) (query_of ? query_of[i] : 0) * h->dim);
	return PGV_OK;
}

int
pgv_hnsw_score_pairs(pgv_hnsw * h, const int32_t *a, const int32_t *b, int64_t npairs, float *out)
{
	for (int64_t i = 0; i < npairs; i++)
		out[i] = dist(h->metric, h->dim, h->vectors + (size_t) a[i] * h->dim, h->vectors + (size_t) b[i] * h->dim);
	return PGV_OK;
}

int
pgv_hnsw_score_groups(pgv_hnsw * h, const int32_t *ids, const int64_t *ids_start, const int32_t *from,
					  const int64_t *pair_start, int ngroups, int64_t nids, int64_t npairs, float *out)
{
	(void) nids;
	(void) npairs;
	for (int g = 0; g < ngroups; g++)
	{
		const int32_t *gi = ids + ids_start[g];
		int			n = (int) (ids_start[g + 1] - ids_start[g]);
		int64_t		at = pair_start[g];

		for (int u = from[g] < 1 ? 1 : from[g]; u < n; u++)
			for (int v = 0; v < u; v++, at++)
				out[at] = dist(h->metric, h->dim, h->vectors + (size_t) gi[u] * h->dim, h->vectors + (size_t) gi[v] * h->dim);
	}
	return PGV_OK;
}

typedef struct
{
	int32_t		id;
	float		d;
	int			expanded;
}			sc;

/* HnswSearchLayer (src/hnswutils.c:824-987) with plain arrays: w[] ascending, C = the unexpanded entries of w */
static int
search_layer(const pgv_hnsw * h, const float *q, sc * w, int wn, int ef, int lc, uint8_t *visited)
{
	int			lm = lc == 0 ? 2 * h->m : h->m;

	memset(visited, 0, (size_t) h->n);
	for (int i = 0; i < wn; i++)
	{
		visited[w[i].id] = 1;
		w[i].expanded = 0;
	}
	for (;;)
	{
		int			ci = -1;
		int32_t		c;
		int64_t		start;

		for (int i = 0; i < wn; i++)
			if (!w[i].expanded)
			{
				ci = i;
				break;
			}
		if (ci < 0)
			break;
		w[ci].expanded = 1;
		c = w[ci].id;
		start = h->nbr_start[c] + (int64_t) (h->levels[c] - lc) * h->m;
		for (int j = 0; j < lm; j++)
		{
			int32_t		e = h->nbr[start + j];
			float		d;
			int			pos;

			if (e < 0 || visited[e])
				continue;
			visited[e] = 1;
			d = dist(h->metric, h->dim, h->vectors + (size_t) e * h->dim, q);
			if (h->levels[e] < lc)
				continue;
			if (!(wn < ef || d < w[wn - 1].d))
				continue;
			/* stable insertion after the entries that are not farther; drop the furthest on overflow */
			pos = wn;
			while (pos > 0 && w[pos - 1].d > d)
				pos--;
			if (wn < ef)
				wn++;
			for (int t = wn - 1; t > pos; t--)
				w[t] = w[t - 1];
			w[pos].id = e;
			w[pos].d = d;
			w[pos].expanded = 0;
		}
	}
	return wn;
}

int
pgv_hnsw_build_search(pgv_hnsw * h, const int32_t *elements, const int32_t *insert_levels, int nq,
					  int ef_construction, int layer_cap, int32_t *out_ids, float *out_dist, int32_t *out_count)
{
	sc		   *w;
	uint8_t    *visited;

	view_refresh(h);
	w = malloc(sizeof(sc) * (size_t) (ef_construction + 1));
	visited = malloc((size_t) (h->n > 0 ? h->n : 1));
	for (int q = 0; q < nq; q++)
	{
		const float *qv = h->vectors + (size_t) elements[q] * h->dim;
		int			wn = 1;

		for (int l = 0; l < layer_cap; l++)
			out_count[(size_t) q * layer_cap + l] = 0;
		if (h->entry < 0)
			continue;
		w[0].id = h->entry;
		w[0].d = dist(h->metric, h->dim, h->vectors + (size_t) h->entry * h->dim, qv);
		for (int lc = h->levels[h->entry]; lc >= 0; lc--)
		{
			int			ef = lc <= insert_levels[q] ? ef_construction : 1;

			wn = search_layer(h, qv, w, wn, ef, lc, visited);
			if (lc <= insert_levels[q] && lc < layer_cap)
			{
				size_t		o = ((size_t) q * layer_cap + lc) * ef_construction;

				for (int i = 0; i < wn; i++)
				{
					out_ids[o + i] = w[i].id;
					out_dist[o + i] = w[i].d;
				}
				out_count[(size_t) q * layer_cap + lc] = wn;
			}
		}
	}
	free(w);
	free(visited);
	return PGV_OK;
}

/* SelectNeighbors for new elements' lists (src/hnswutils.c:1064-1165 without cached flags) over candidate lists nearest
 * first: chosen while no chosen neighbor is at distance <= the candidate's own; the rejected fill up to lm */
static int64_t
select_from(const pgv_hnsw * h, const int32_t *ids, const float *ds, const int32_t *cnt, const int32_t *insert_levels, int nq,
			int ef_construction, int layer_cap, int32_t *out_ids, float *out_dist, uint8_t *out_closer, int32_t *out_count)
{
	const size_t per = (size_t) nq * layer_cap;
	const int	stride = 2 * h->m;
	int		   *pick = malloc(sizeof(int) * (size_t) ef_construction);
	uint8_t    *chosen = malloc((size_t) ef_construction);
	int64_t		pairs = 0;

	for (size_t g = 0; g < per; g++)
	{
		const int	lc = (int) (g % layer_cap),
					lm = lc == 0 ? 2 * h->m : h->m;
		const int	nw = lc > insert_levels[g / layer_cap] ? 0 : cnt[g];
		const int32_t *gi = ids + g * ef_construction;
		const float *gd = ds + g * ef_construction;
		int			rn = 0,
					nchosen,
					j;

		if (nw <= lm)
		{
			for (int i = 0; i < nw; i++)
			{
				out_ids[g * stride + i] = gi[nw - 1 - i];
				out_dist[g * stride + i] = gd[nw - 1 - i];
				out_closer[g * stride + i] = 0;
			}
			out_count[g] = nw;
			continue;
		}
		pairs += (int64_t) nw * (nw - 1) / 2;
		memset(chosen, 0, (size_t) nw);
		for (j = 0; j < nw && rn < lm; j++)
		{
			int			closer = 1;

			for (int i = 0; i < rn && closer; i++)
				if (dist(h->metric, h->dim, h->vectors + (size_t) gi[j] * h->dim, h->vectors + (size_t) gi[pick[i]] * h->dim) <= gd[j])
					closer = 0;
			if (closer)
			{
				chosen[j] = 1;
				pick[rn++] = j;
			}
		}
		nchosen = rn;
		for (int x = 0; x < j && rn < lm; x++)
			if (!chosen[x])
				pick[rn++] = x;
		for (int i = 0; i < rn; i++)
		{
			out_ids[g * stride + i] = gi[pick[i]];
			out_dist[g * stride + i] = gd[pick[i]];
			out_closer[g * stride + i] = i < nchosen;
		}
		out_count[g] = rn;
	}
	free(pick);
	free(chosen);
	return pairs;
}

int
pgv_hnsw_build_neighbors(pgv_hnsw * h, const int32_t *elements, const int32_t *insert_levels, int nq, int ef_construction,
						 int layer_cap, int32_t *out_ids, float *out_dist, uint8_t *out_closer, int32_t *out_count,
						 int64_t *out_pairs)
{
	const size_t per = (size_t) nq * layer_cap;
	int32_t    *ids = malloc(sizeof(int32_t) * (per ? per : 1) * ef_construction);
	float	   *ds = malloc(sizeof(float) * (per ? per : 1) * ef_construction);
	int32_t    *cnt = malloc(sizeof(int32_t) * (per ? per : 1));
	int			rc = pgv_hnsw_build_search(h, elements, insert_levels, nq, ef_construction, layer_cap, ids, ds, cnt);
	int64_t		pairs = 0;

	if (rc == PGV_OK)
		pairs = select_from(h, ids, ds, cnt, insert_levels, nq, ef_construction, layer_cap, out_ids, out_dist, out_closer, out_count);
	if (out_pairs)
		*out_pairs = pairs;
	free(ids);
	free(ds);
	free(cnt);
	return rc;
}

/* the same in two halves: the searches' lists kept in a slot of the OWNER (a view searches, another view selects) */
int
pgv_hnsw_build_search_keep(pgv_hnsw * h, const int32_t *elements, const int32_t *insert_levels, int nq, int ef_construction,
						   int layer_cap, int slot)
{
	pgv_hnsw   *o = h->view_of ? h->view_of : h;
	const size_t per = (size_t) nq * layer_cap;

	if (!o->nb_dist || slot < 0 || slot > 1)
		return fail(PGV_ERR_STATE, "mock: pgv_hnsw_build_search_keep needs pgv_hnsw_link_begin / slot 0 or 1");
	free(o->kept[slot].ids);
	free(o->kept[slot].dist);
	free(o->kept[slot].cnt);
	free(o->kept[slot].levels);
	o->kept[slot].ids = malloc(sizeof(int32_t) * (per ? per : 1) * ef_construction);
	o->kept[slot].dist = malloc(sizeof(float) * (per ? per : 1) * ef_construction);
	o->kept[slot].cnt = malloc(sizeof(int32_t) * (per ? per : 1));
	o->kept[slot].levels = malloc(sizeof(int32_t) * (size_t) (nq ? nq : 1));
	memcpy(o->kept[slot].levels, insert_levels, sizeof(int32_t) * (size_t) nq);
	o->kept[slot].nq = nq;
	o->kept[slot].ef = ef_construction;
	o->kept[slot].lcap = layer_cap;
	return pgv_hnsw_build_search(h, elements, insert_levels, nq, ef_construction, layer_cap, o->kept[slot].ids, o->kept[slot].dist,
								 o->kept[slot].cnt);
}

int
pgv_hnsw_build_select_kept(pgv_hnsw * h, int slot, int32_t *out_ids, float *out_dist, uint8_t *out_closer, int32_t *out_count,
						   int64_t *out_pairs)
{
	pgv_hnsw   *o = h->view_of ? h->view_of : h;
	int64_t		pairs;

	/* (no view_refresh: the entry point may be moving under pgv_hnsw_link_apply on another thread, and nothing here
	 * looks at the graph -- the lists, the vectors and m, which is the owner's and fixed for the build) */
	if (!o->nb_dist || slot < 0 || slot > 1)
		return fail(PGV_ERR_STATE, "mock: pgv_hnsw_build_select_kept needs pgv_hnsw_link_begin / slot 0 or 1");
	pairs = select_from(o, o->kept[slot].ids, o->kept[slot].dist, o->kept[slot].cnt, o->kept[slot].levels, o->kept[slot].nq,
						o->kept[slot].ef, o->kept[slot].lcap, out_ids, out_dist, out_closer, out_count);
	if (out_pairs)
		*out_pairs = pairs;
	return PGV_OK;
}

/* ---- the build's graph updates (pgv_hnsw_link_*): the replay is csrc/hnsw_link_core.h, the source the device compiles */
#include "../../pgvector_amd/csrc/hnsw_link_core.h"

static void
link_batch_free(pgv_hnsw * h)
{
	free(h->link_elems);
	free(h->link_sel_ids);
	free(h->link_sel_cnt);
	free(h->link_linked);
	free(h->link_sel_closer);
	free(h->link_sel_dist);
	h->link_elems = h->link_sel_ids = h->link_sel_cnt = NULL;
	h->link_linked = h->link_sel_closer = NULL;
	h->link_sel_dist = NULL;
	h->link_nq = 0;
}

int
pgv_hnsw_link_begin(pgv_hnsw * h)
{
	int64_t		total = h->nbr_start ? h->nbr_start[h->n] : 0;

	if (h->view_of || h->m == 0)
		return fail(PGV_ERR_STATE, "mock: pgv_hnsw_link_begin on a view / without a graph");
	/* test knob: a device without room for the build's graph state */
	if (getenv("MOCK_HIP_LINK_NOMEM"))
		return fail(PGV_ERR_NOMEM, "mock: no room for the hnsw build state");
	free(h->nb_dist);
	free(h->nb_flag);
	h->nb_dist = calloc((size_t) (total > 0 ? total : 1), sizeof(float));
	h->nb_flag = calloc((size_t) (total > 0 ? total : 1), 1);
	return PGV_OK;
}

static void *
dup_bytes(const void *p, size_t bytes)
{
	void	   *c = malloc(bytes ? bytes : 1);

	memcpy(c, p, bytes);
	return c;
}

int
pgv_hnsw_link_prepare(pgv_hnsw * h, const int32_t *elements, const uint8_t *linked, int nq, int layer_cap,
					  const int32_t *sel_ids, const float *sel_dist, const uint8_t *sel_closer, const int32_t *sel_count,
					  int64_t *out_pairs)
{
	const size_t per = (size_t) nq * layer_cap,
				stride = 2 * (size_t) h->m;

	if (!h->nb_dist)
		return fail(PGV_ERR_STATE, "mock: pgv_hnsw_link_prepare needs pgv_hnsw_link_begin");
	link_batch_free(h);
	h->link_nq = nq;
	h->link_lcap = layer_cap;
	h->link_elems = dup_bytes(elements, sizeof(int32_t) * (size_t) nq);
	h->link_linked = dup_bytes(linked, (size_t) nq);
	h->link_sel_ids = dup_bytes(sel_ids, sizeof(int32_t) * per * stride);
	h->link_sel_dist = dup_bytes(sel_dist, sizeof(float) * per * stride);
	h->link_sel_closer = dup_bytes(sel_closer, per * stride);
	h->link_sel_cnt = dup_bytes(sel_count, sizeof(int32_t) * per);
	if (out_pairs)
		*out_pairs = 0;
	return PGV_OK;
}

int
pgv_hnsw_link_apply(pgv_hnsw * h, int32_t entry)
{
	const int	m = h->m,
				nq = h->link_nq,
				lcap = h->link_lcap,
				stride = 2 * m;
	const int64_t total = h->nbr_start[h->n];
	const size_t nlists = (size_t) (total / m) + 1;
	int		   *count = calloc(nlists, sizeof(int)),
			   *first = malloc(sizeof(int) * nlists);
	int			nreq = 0,
				nrec = 0;
	int		   *req_list,
			   *req_next,
			   *rec_list;
	int32_t    *req_elem;
	float	   *req_dist;
	int64_t		pairs = 0,
				pairs2 = 0;
	int			deferred = 0;

	/* the link requests in the reference's order: batch elements in heap order, their layers top down, their neighbors
	 * in list order; filed per list (slot position / m), a list's newcomers therefore in heap order */
	req_list = malloc(sizeof(int) * ((size_t) nq * lcap * stride + 1));
	req_next = malloc(sizeof(int) * ((size_t) nq * lcap * stride + 1));
	req_elem = malloc(sizeof(int32_t) * ((size_t) nq * lcap * stride + 1));
	req_dist = malloc(sizeof(float) * ((size_t) nq * lcap * stride + 1));
	rec_list = malloc(sizeof(int) * ((size_t) nq * lcap * stride + 1));
	for (int q = 0; q < nq; q++)
	{
		const int32_t e = h->link_elems[q];

		if (!h->link_linked[q])
			continue;
		for (int lc = (h->levels[e] < lcap - 1 ? h->levels[e] : lcap - 1); lc >= 0; lc--)
		{
			const size_t g = (size_t) q * lcap + lc;

			for (int i = 0; i < h->link_sel_cnt[g]; i++)
			{
				const int32_t owner = h->link_sel_ids[g * stride + i];
				const int	lidx = (int) ((h->nbr_start[owner] + (int64_t) (h->levels[owner] - lc) * m) / m);

				if (count[lidx]++ == 0)
				{
					first[lidx] = nreq;
					rec_list[nrec++] = lidx;
				}
				else
				{
					int			t = first[lidx];

					while (req_next[t] >= 0)
						t = req_next[t];
					req_next[t] = nreq;
				}
				req_list[nreq] = lc;	/* (the layer; the list is rec_list's) */
				req_next[nreq] = -1;
				req_elem[nreq] = e;
				req_dist[nreq] = h->link_sel_dist[g * stride + i];
				nreq++;
			}
		}
	}
	for (int k = 0; k < nrec; k++)
	{
		const int	lidx = rec_list[k];
		const int64_t pos = (int64_t) lidx * m;
		const int	lc = req_list[first[lidx]];
		const int	lm = lc == 0 ? 2 * m : m;
		const int	nnew = count[lidx];
		int32_t		le[PGV_LINK_LMAX + 1];
		float		ld[PGV_LINK_LMAX + 1];
		uint8_t		lf[PGV_LINK_LMAX + 1],
					scratch[5 * (PGV_LINK_LMAX + 1)];
		int16_t		loc[PGV_LINK_LMAX + 1];
		uint64_t	key[PGV_LINK_LMAX + 1];
		int			len = 0,
					nstart,
					nlocal,
					from,
					stop;
		uint8_t		closer_set;
		int32_t    *ids;
		float	   *newdist,
				   *tri = NULL,
				   *mm = NULL;
		pgv_link_pairs ps;

		while (len < lm && h->nbr[pos + len] >= 0)
		{
			le[len] = h->nbr[pos + len];
			ld[len] = h->nb_dist[pos + len];
			lf[len] = h->nb_flag[pos + len] & 1;
			loc[len] = (int16_t) len;
			len++;
		}
		closer_set = (h->nb_flag[pos] >> 1) & 1;
		nstart = len;
		nlocal = nstart + nnew;
		ids = malloc(sizeof(int32_t) * (size_t) nlocal);
		newdist = malloc(sizeof(float) * (size_t) nnew);
		memcpy(ids, le, sizeof(int32_t) * (size_t) nstart);
		for (int t = first[lidx], j = 0; t >= 0; t = req_next[t], j++)
		{
			ids[nstart + j] = req_elem[t];
			newdist[j] = req_dist[t];
		}
		from = closer_set ? nstart : 1;
		if (from < 1)
			from = 1;
		if (nlocal > lm)
		{
			/* the pairs a selection may look up: those with u >= from (host/hnsw_build.c step 4) */
			int64_t		np = ((int64_t) nlocal * (nlocal - 1) - (int64_t) from * (from - 1)) / 2,
						at = 0;

			tri = malloc(sizeof(float) * (size_t) (np > 0 ? np : 1));
			for (int u = from; u < nlocal; u++)
				for (int v = 0; v < u; v++)
					tri[at++] = dist(h->metric, h->dim, h->vectors + (size_t) ids[u] * h->dim, h->vectors + (size_t) ids[v] * h->dim);
			pairs += np;
		}
		ps.tri = tri;
		ps.from = from;
		ps.base = from * (from - 1) / 2;
		ps.mm = NULL;
		stop = pgv_link_replay(le, ld, lf, loc, &len, &closer_set, lm, ids, newdist, nstart, nlocal, nstart, &ps, key, scratch);
		if (stop < nlocal)
		{
			/* a cached closer member lost its flag and the earlier rejects are checked against the whole selection:
			 * the member-member triangle, then the rest of the newcomers */
			int64_t		at = 0;

			mm = malloc(sizeof(float) * (size_t) (nstart > 1 ? (int64_t) nstart * (nstart - 1) / 2 : 1));
			for (int u = 1; u < nstart; u++)
				for (int v = 0; v < u; v++)
					mm[at++] = dist(h->metric, h->dim, h->vectors + (size_t) ids[u] * h->dim, h->vectors + (size_t) ids[v] * h->dim);
			pairs2 += at;
			deferred++;
			ps.mm = mm;
			stop = pgv_link_replay(le, ld, lf, loc, &len, &closer_set, lm, ids, newdist, nstart, nlocal, stop, &ps, key, scratch);
			if (stop < nlocal)
				return fail(PGV_ERR_STATE, "mock: a list still waits for distances");
		}
		for (int j = 0; j < len; j++)
		{
			h->nbr[pos + j] = le[j];
			h->nb_dist[pos + j] = ld[j];
			h->nb_flag[pos + j] = (uint8_t) (lf[j] | (j == 0 ? (closer_set << 1) : 0));
		}
		free(ids);
		free(newdist);
		free(tri);
		free(mm);
	}
	/* the batch's own elements */
	for (int q = 0; q < nq; q++)
	{
		const int32_t e = h->link_elems[q];

		if (!h->link_linked[q])
			continue;
		for (int lc = 0; lc < lcap && lc <= h->levels[e]; lc++)
		{
			const size_t g = (size_t) q * lcap + lc;
			const int64_t pos = h->nbr_start[e] + (int64_t) (h->levels[e] - lc) * m;

			for (int i = 0; i < h->link_sel_cnt[g]; i++)
			{
				h->nbr[pos + i] = h->link_sel_ids[g * stride + i];
				h->nb_dist[pos + i] = h->link_sel_dist[g * stride + i];
				h->nb_flag[pos + i] = h->link_sel_closer[g * stride + i] & 1;
			}
		}
	}
	h->entry = entry;
	free(count);
	free(first);
	free(req_list);
	free(req_next);
	free(req_elem);
	free(req_dist);
	free(rec_list);
	link_batch_free(h);
	(void) pairs;				/* (the first round's are pgv_hnsw_link_prepare's to report; the stand-in scores them here) */
	h->link_pairs2 += pairs2;
	h->link_deferred += deferred;
	return PGV_OK;
}

int
pgv_hnsw_link_end(pgv_hnsw * h, int32_t *out_nbr, int64_t *out_pairs, int64_t *out_deferred)
{
	if (out_pairs)
		*out_pairs = h->link_pairs2;
	if (out_deferred)
		*out_deferred = h->link_deferred;
	h->link_pairs2 = h->link_deferred = 0;
	if (h->nb_dist && out_nbr && h->nbr_start)
		memcpy(out_nbr, h->nbr, sizeof(int32_t) * (size_t) h->nbr_start[h->n]);
	free(h->nb_dist);
	free(h->nb_flag);
	h->nb_dist = NULL;
	h->nb_flag = NULL;
	link_batch_free(h);
	for (int i = 0; i < 2; i++)
	{
		free(h->kept[i].ids);
		free(h->kept[i].dist);
		free(h->kept[i].cnt);
		free(h->kept[i].levels);
		memset(&h->kept[i], 0, sizeof(h->kept[i]));
	}
	return PGV_OK;
}

/* hnswgettuple's first batch (src/hnswscan.c:25-56): greedy descent with ef = 1, then HnswSearchLayer with ef_search
 * on layer 0; the k nearest, ascending, -1 / +inf padded */
static int hnsw_search_f32(pgv_hnsw * h, const void *queries, int nq, int ef_search, int k, int64_t *out_elem, float *out_dist,
						   int64_t *out_scored);

int
pgv_hnsw_search(pgv_hnsw * h, const void *queries, int nq, int ef_search, int k, int64_t *out_elem, float *out_dist,
				int64_t *out_scored)
{
	float	   *w = h->f16 && queries ? widen(queries, (size_t) nq * h->dim) : NULL;
	int			rc = hnsw_search_f32(h, w ? w : queries, nq, ef_search, k, out_elem, out_dist, out_scored);

	free(w);
	return rc;
}

static int
hnsw_search_f32(pgv_hnsw * h, const void *queries, int nq, int ef_search, int k, int64_t *out_elem, float *out_dist,
				int64_t *out_scored)
{
	sc		   *w = malloc(sizeof(sc) * (size_t) (ef_search + 1));
	uint8_t    *visited = malloc((size_t) (h->n > 0 ? h->n : 1));

	for (int q = 0; q < nq; q++)
	{
		const float *qv = (const float *) queries + (size_t) q * h->dim;
		int			wn = 0;

		if (h->entry >= 0)
		{
			w[0].id = h->entry;
			w[0].d = dist(h->metric, h->dim, h->vectors + (size_t) h->entry * h->dim, qv);
			wn = 1;
			for (int lc = h->levels[h->entry]; lc >= 0; lc--)
				wn = search_layer(h, qv, w, wn, lc == 0 ? ef_search : 1, lc, visited);
		}
		for (int i = 0; i < k; i++)
		{
			out_elem[(size_t) q * k + i] = i < wn ? w[i].id : -1;
			out_dist[(size_t) q * k + i] = i < wn ? w[i].d : INFINITY;
		}
		if (out_scored)
		{
			int64_t		scored = 0;

			for (int64_t e = 0; e < h->n; e++)
				scored += visited[e];
			out_scored[q] = scored;
		}
	}
	free(w);
	free(visited);
	return PGV_OK;
}

/* ------------------------------------------------------------------ multi-GPU
 * The stand-in's pgv_comm_* / pgv_kmeans_sharded / pgv_search_batch_sharded issue the SAME collectives in the same
 * order with the same payloads as libpgv_hip (pgv_abi_comm.hip): all-gather of the ranks' sample counts; per
 * k-means++ round an all-gather of the ranks' weight totals and one of the candidate row; per Lloyd iteration ONE
 * all-reduce of the fused record sums[k x dim] | counts[k] | changes (as floats); per search an all-gather of the
 * ranks' probe-list slices and two of their heads.  tests/test_sharded_cpath_gloo.py runs two ranks of it over gloo.
 */
struct pgv_comm
{
	int			nranks,
				rank;
	pgv_collectives coll;
	struct mock_group *group;	/* pgv_comm_create: the in-process group this rank joined */
};

/*
 * pgv_comm_unique_id / pgv_comm_create on the stand-in: the ranks of a group are THREADS of one process (ext/ivfbuild_gpu.c:
 * the leader of a build drives one helper thread per device), found through the id in a process-wide table; the two
 * collectives are barriers + copies between the ranks' buffers (host memory here), the all-reduce adding in rank order.
 * libpgv_hip forms the same group with ncclCommInitRank and runs the same collectives over RCCL.
 */
#include <pthread.h>
#include <stdbool.h>
typedef struct mock_group
{
	unsigned char id[PGV_COMM_ID_BYTES];
	int			nranks,
				joined,
				left;
	pthread_barrier_t bar;
	const void *send[16];
	void	   *recv[16];
}			mock_group;
static pthread_mutex_t group_lock = PTHREAD_MUTEX_INITIALIZER;
static pthread_cond_t group_cond = PTHREAD_COND_INITIALIZER;
static mock_group *groups[8];
static unsigned long long next_group_id = 1;

int
pgv_comm_unique_id(void *out_id)
{
	unsigned long long v;

	if (!out_id)
		return fail(PGV_ERR_ARG, "out_id is NULL");
	pthread_mutex_lock(&group_lock);
	v = next_group_id++;
	pthread_mutex_unlock(&group_lock);
	memset(out_id, 0, PGV_COMM_ID_BYTES);
	memcpy(out_id, &v, sizeof(v));
	((unsigned char *) out_id)[8] = 0x6d;	/* 'm': a stand-in id */
	return PGV_OK;
}

static int
mg_all_gather(void *state, const void *send, void *recv, size_t bytes, void *stream)
{
	pgv_comm   *cm = state;
	mock_group *g = cm->group;

	(void) stream;
	g->send[cm->rank] = send;
	pthread_barrier_wait(&g->bar);
	for (int r = 0; r < g->nranks; r++)
		memmove((char *) recv + (size_t) r * bytes, g->send[r], bytes);
	pthread_barrier_wait(&g->bar);
	return 0;
}

static int
mg_all_reduce(void *state, float *buf, size_t count, void *stream)
{
	pgv_comm   *cm = state;
	mock_group *g = cm->group;
	float	   *sum = malloc(sizeof(float) * (count ? count : 1));

	(void) stream;
	g->send[cm->rank] = buf;
	pthread_barrier_wait(&g->bar);
	for (size_t i = 0; i < count; i++)
	{
		float		acc = 0.f;

		for (int r = 0; r < g->nranks; r++)
			acc += ((const float *) g->send[r])[i];
		sum[i] = acc;
	}
	pthread_barrier_wait(&g->bar);	/* everybody has read everybody's input */
	memcpy(buf, sum, sizeof(float) * count);
	free(sum);
	pthread_barrier_wait(&g->bar);
	return 0;
}

int
pgv_comm_create(pgv_ctx * ctx, int nranks, int rank, const void *unique_id, pgv_comm * *out)
{
	mock_group *g = NULL;
	pgv_comm   *cm;

	(void) ctx;
	if (!out || !unique_id || nranks < 1 || nranks > 16 || rank < 0 || rank >= nranks)
		return fail(PGV_ERR_ARG, "mock: pgv_comm_create arguments");
	pthread_mutex_lock(&group_lock);
	for (int i = 0; i < 8 && g == NULL; i++)
		if (groups[i] && memcmp(groups[i]->id, unique_id, PGV_COMM_ID_BYTES) == 0)
			g = groups[i];
	if (g == NULL)
	{
		g = calloc(1, sizeof(*g));
		memcpy(g->id, unique_id, PGV_COMM_ID_BYTES);
		g->nranks = nranks;
		pthread_barrier_init(&g->bar, NULL, (unsigned) nranks);
		for (int i = 0; i < 8; i++)
			if (groups[i] == NULL)
			{
				groups[i] = g;
				break;
			}
	}
	g->joined++;
	pthread_cond_broadcast(&group_cond);
	while (g->joined < g->nranks)	/* ncclCommInitRank returns when every rank has called it */
		pthread_cond_wait(&group_cond, &group_lock);
	pthread_mutex_unlock(&group_lock);
	cm = calloc(1, sizeof(*cm));
	cm->nranks = nranks;
	cm->rank = rank;
	cm->group = g;
	cm->coll.all_gather = mg_all_gather;
	cm->coll.all_reduce_sum_f32 = mg_all_reduce;
	cm->coll.state = cm;
	*out = cm;
	return PGV_OK;
}

int
pgv_comm_create_custom(pgv_ctx * ctx, int nranks, int rank, const pgv_collectives * coll, pgv_comm * *out)
{
	pgv_comm   *cm;

	(void) ctx;
	if (nranks < 1 || rank < 0 || rank >= nranks)
		return fail(PGV_ERR_ARG, "mock: bad rank");
	if (nranks > 1 && (!coll || !coll->all_reduce_sum_f32 || !coll->all_gather))
		return fail(PGV_ERR_ARG, "pgv_comm_create_custom: both collectives are needed");
	cm = calloc(1, sizeof(*cm));
	cm->nranks = nranks;
	cm->rank = rank;
	if (coll)
		cm->coll = *coll;
	*out = cm;
	return PGV_OK;
}

void
pgv_comm_destroy(pgv_comm * cm)
{
	if (cm && cm->group)
	{
		mock_group *g = cm->group;
		bool		last;

		pthread_mutex_lock(&group_lock);
		last = ++g->left == g->nranks;
		if (last)
			for (int i = 0; i < 8; i++)
				if (groups[i] == g)
					groups[i] = NULL;
		pthread_mutex_unlock(&group_lock);
		if (last)
		{
			pthread_barrier_destroy(&g->bar);
			free(g);
		}
	}
	free(cm);
}

int
pgv_comm_size(const pgv_comm * cm)
{
	return cm ? cm->nranks : 0;
}

int
pgv_comm_rank(const pgv_comm * cm)
{
	return cm ? cm->rank : -1;
}

static int
mc_all_gather(pgv_comm * cm, const void *send, void *recv, size_t bytes)
{
	if (cm->nranks == 1)
	{
		memmove(recv, send, bytes);
		return PGV_OK;
	}
	return cm->coll.all_gather(cm->coll.state, send, recv, bytes, NULL) == 0 ? PGV_OK : fail(PGV_ERR_DEVICE, "all-gather callback failed");
}

static int
mc_all_reduce(pgv_comm * cm, float *buf, size_t count)
{
	if (cm->nranks == 1)
		return PGV_OK;
	return cm->coll.all_reduce_sum_f32(cm->coll.state, buf, count, NULL) == 0 ? PGV_OK : fail(PGV_ERR_DEVICE, "all-reduce callback failed");
}

typedef struct
{
	const pgv_rng *cb;
	uint64_t	s;
}			mock_rng;

static uint64_t
mr_next(mock_rng * r)
{
	uint64_t	z = (r->s += 0x9E3779B97F4A7C15ull);

	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
	return z ^ (z >> 31);
}

static double
mr_double(mock_rng * r)
{
	return r->cb && r->cb->next_double ? r->cb->next_double(r->cb->state) : (double) (mr_next(r) >> 11) / 9007199254740992.0;
}

static uint32_t
mr_u32(mock_rng * r)
{
	return r->cb && r->cb->next_u32 ? r->cb->next_u32(r->cb->state) : (uint32_t) (mr_next(r) >> 32);
}

int
pgv_kmeans_sharded(pgv_comm * cm, pgv_ops ops, pgv_dtype dtype, int dim, const void *samples, int n, int k,
				   int max_iterations, const pgv_rng * rng, void *out_centers, int32_t *out_closest, int *out_iters)
{
	const int	R = cm->nranks;
	const float *s = samples;
	float	   *c = out_centers;
	mock_rng	r = {rng, rng ? rng->seed : 0};
	int64_t    *cnt = malloc(sizeof(int64_t) * (size_t) (R + 1));
	int64_t		n_total = 0,
				mine = n;
	float	   *weight = malloc(sizeof(float) * (size_t) (n > 0 ? n : 1));
	float	   *row = malloc(sizeof(float) * (size_t) dim * (size_t) (R + 1));
	double	   *totals = malloc(sizeof(double) * (size_t) (R + 1));
	size_t		rec = (size_t) k * dim + (size_t) k + 1;
	float	   *record = malloc(sizeof(float) * rec);
	int32_t    *closest = malloc(sizeof(int32_t) * (size_t) (n > 0 ? n : 1));
	int			iters = 0,
				rc = PGV_OK;

	if (dtype != PGV_F32 || ops != PGV_OPS_L2)
		return fail(PGV_ERR_ARG, "mock: fp32 / l2 only");
	if (max_iterations <= 0)
		max_iterations = 500;
	/* every rank's sample count */
	if ((rc = mc_all_gather(cm, &mine, cnt, sizeof(int64_t))) != PGV_OK)
		goto done;
	for (int q = 0; q < R; q++)
		n_total += cnt[q];
	if (n_total == 0)
	{
		for (size_t i = 0; i < (size_t) k * dim; i++)
			c[i] = (float) mr_double(&r);	/* RandomCenters: the same draws on every rank */
		goto done;
	}
	/* k-means++ (src/ivfkmeans.c:23-91) over the sharded sample */
	for (int j = 0; j < n; j++)
		weight[j] = 3.402823466e+38f;
	{
		int64_t		at = (int64_t) (mr_u32(&r) % (uint32_t) n_total);
		int			owner = 0;

		while (owner < R - 1 && at >= cnt[owner])
			at -= cnt[owner++];
		memset(row, 0, sizeof(float) * (size_t) dim);
		if (owner == cm->rank)
			memcpy(row, s + (size_t) at * dim, sizeof(float) * (size_t) dim);
		if ((rc = mc_all_gather(cm, row, row + dim, sizeof(float) * (size_t) dim)) != PGV_OK)
			goto done;
		memcpy(c, row + dim + (size_t) owner * dim, sizeof(float) * (size_t) dim);
	}
	for (int i = 0; i + 1 < k; i++)
	{
		double		my_total = 0.0,
					all = 0.0,
					target;
		int			owner = R - 1;

		for (int j = 0; j < n; j++)
		{
			float		d = dist(PGV_L2SQ, dim, s + (size_t) j * dim, c + (size_t) i * dim);

			if (d < weight[j])
				weight[j] = d;
			my_total += weight[j];
		}
		if ((rc = mc_all_gather(cm, &my_total, totals, sizeof(double))) != PGV_OK)
			goto done;
		for (int q = 0; q < R; q++)
			all += totals[q];
		target = mr_double(&r) * all;
		for (int q = 0; q < R; q++)
		{
			if (target < totals[q] || q == R - 1)
			{
				owner = q;
				break;
			}
			target -= totals[q];
		}
		memset(row, 0, sizeof(float) * (size_t) dim);
		if (owner == cm->rank && n > 0)
		{
			int			pick = n - 1;
			double		run = 0.0;

			for (int j = 0; j < n; j++)
			{
				run += weight[j];
				if (run > target)
				{
					pick = j;
					break;
				}
			}
			memcpy(row, s + (size_t) pick * dim, sizeof(float) * (size_t) dim);
		}
		if ((rc = mc_all_gather(cm, row, row + dim, sizeof(float) * (size_t) dim)) != PGV_OK)
			goto done;
		memcpy(c + (size_t) (i + 1) * dim, row + dim + (size_t) owner * dim, sizeof(float) * (size_t) dim);
	}
	/* Lloyd: one fused all-reduce per iteration (sums | counts | changes) */
	for (int j = 0; j < n; j++)
		closest[j] = -1;
	for (int it = 0; it < max_iterations; it++)
	{
		float		changes = 0.f;

		iters = it + 1;
		memset(record, 0, sizeof(float) * rec);
		for (int j = 0; j < n; j++)
		{
			int			best = 0;
			float		bd = INFINITY;

			for (int q = 0; q < k; q++)
			{
				float		d = dist(PGV_L2SQ, dim, s + (size_t) j * dim, c + (size_t) q * dim);

				if (d < bd)
				{
					bd = d;
					best = q;
				}
			}
			if (best != closest[j])
				changes += 1.f;
			closest[j] = best;
			record[(size_t) k * dim + best] += 1.f;
			for (int d = 0; d < dim; d++)
				record[(size_t) best * dim + d] += s[(size_t) j * dim + d];
		}
		record[rec - 1] = changes;
		if ((rc = mc_all_reduce(cm, record, rec)) != PGV_OK)
			goto done;
		for (int q = 0; q < k; q++)
		{
			float		cn = record[(size_t) k * dim + q];

			for (int d = 0; d < dim; d++)
				c[(size_t) q * dim + d] = cn > 0.f ? record[(size_t) q * dim + d] / cn : (float) mr_double(&r);	/* empty: replicated draws */
		}
		if (record[rec - 1] == 0.f && it != 0)
			break;
	}
	if (out_closest && n > 0)
		memcpy(out_closest, closest, sizeof(int32_t) * (size_t) n);
done:
	if (out_iters)
		*out_iters = iters;
	free(cnt);
	free(weight);
	free(row);
	free(totals);
	free(record);
	free(closest);
	return rc;
}

int
pgv_search_batch_sharded(pgv_comm * cm, pgv_index * ix, const void *queries, int nq, int probes, int k, float *out_dist,
						 uint64_t *out_tid)
{
	const int	R = cm->nranks;
	const int	per = (nq + R - 1) / R;
	const int	lo = cm->rank * per < nq ? cm->rank * per : nq;
	const int	hi = lo + per < nq ? lo + per : nq;
	size_t		slice = (size_t) per * probes;
	int32_t    *lists_mine = calloc(slice * (size_t) (R + 1), sizeof(int32_t));
	int32_t    *lists_all = lists_mine + slice;
	size_t		head = (size_t) nq * k;
	float	   *dist_all = malloc(sizeof(float) * head * (size_t) (R + 1));
	uint64_t   *tid_all = malloc(sizeof(uint64_t) * head * (size_t) (R + 1));
	float	   *dist_mine = dist_all + head * R;
	uint64_t   *tid_mine = tid_all + head * R;
	float	   *d = malloc(sizeof(float) * (size_t) (ix->n > 0 ? ix->n : 1));
	int64_t    *sl = malloc(sizeof(int64_t) * (size_t) (ix->n > 0 ? ix->n : 1));
	int			rc;

	if (!ix->tids)
		return fail(PGV_ERR_STATE, "a sharded index needs heap tids");
	/* GetScanLists: this rank's slice of the batch against the replicated centers */
	if (hi > lo)
		rank_lists_f32(ix, (const float *) queries + (size_t) lo * ix->dim, hi - lo, probes, lists_mine, NULL);
	if ((rc = mc_all_gather(cm, lists_mine, lists_all, sizeof(int32_t) * slice)) != PGV_OK)
		goto done;
	/* GetScanItems: the probed lists this rank owns (foreign lists are empty in the local index), whole batch */
	for (int q = 0; q < nq; q++)
	{
		const float *qv = (const float *) queries + (size_t) q * ix->dim;
		int64_t		m;

		scan_lists_f32(ix, qv, lists_all + (size_t) q * probes, probes, d, sl, ix->n, &m);
		for (int r = 0; r < k; r++)
		{
			int64_t		best = -1;

			for (int64_t i = 0; i < m; i++)
				if (sl[i] >= 0 && (best < 0 || d[i] < d[best]))
					best = i;
			dist_mine[(size_t) q * k + r] = best >= 0 ? d[best] : INFINITY;
			tid_mine[(size_t) q * k + r] = best >= 0 ? ix->tids[sl[best]] : ~(uint64_t) 0;
			if (best >= 0)
				sl[best] = -1 - sl[best];
		}
	}
	if ((rc = mc_all_gather(cm, dist_mine, dist_all, sizeof(float) * head)) != PGV_OK ||
		(rc = mc_all_gather(cm, tid_mine, tid_all, sizeof(uint64_t) * head)) != PGV_OK)
		goto done;
	/* the final top-k merge: ties go to the lower rank */
	for (int q = 0; q < nq; q++)
	{
		int			at[16] = {0};

		for (int r = 0; r < k; r++)
		{
			int			br = -1;

			for (int p = 0; p < R; p++)
				if (at[p] < k && (br < 0 || dist_all[(size_t) p * head + (size_t) q * k + at[p]] < dist_all[(size_t) br * head + (size_t) q * k + at[br]]))
					br = p;
			out_dist[(size_t) q * k + r] = dist_all[(size_t) br * head + (size_t) q * k + at[br]];
			out_tid[(size_t) q * k + r] = tid_all[(size_t) br * head + (size_t) q * k + at[br]];
			at[br]++;
		}
	}
done:
	free(lists_mine);
	free(dist_all);
	free(tid_all);
	free(d);
	free(sl);
	return rc;
}
