/*
 * hnsw_search.c -- HnswSearchLayer / GetScanItems (src/hnswutils.c:824-987,
 * src/hnswscan.c:25-56) for many queries in lock step, with the candidate
 * scoring loop (:908-934) replaced by one pgv_hnsw_score call per step.
 * Host glue: heaps, visited sets and the admission rules stay on the CPU exactly
 * as in the reference; no distance is computed here.
 */
#include "pgv_host.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static __thread char host_err[256];

const char *
pgv_host_last_error(void)
{
	return host_err;
}

int
pgv_host_fail(int code, const char *fmt,...)
{
	va_list		ap;

	va_start(ap, fmt);
	vsnprintf(host_err, sizeof(host_err), fmt, ap);
	va_end(ap);
	return code;
}

/* ------------------------------------------------------------- small heaps */

typedef struct
{
	int32_t		element;
	float		distance;		/* fp32 kernel value; compared as double like HnswSearchCandidate.distance */
}			cand;

typedef struct
{
	cand	   *a;
	int			n,
				cap;
	int			nearest_first;
}			heap;

static inline int
before(const heap * h, const cand * x, const cand * y)
{
	/* CompareNearestCandidates / CompareFurthestCandidates, src/hnswutils.c:626-654;
	 * order among equal distances is unspecified there, element slot here */
	if (x->distance != y->distance)
		return h->nearest_first ? x->distance < y->distance : x->distance > y->distance;
	return x->element < y->element;
}

static void
heap_push(heap * h, cand c)
{
	int			i;

	if (h->n == h->cap)
	{
		h->cap = h->cap ? h->cap * 2 : 64;
		h->a = realloc(h->a, sizeof(cand) * (size_t) h->cap);
	}
	i = h->n++;
	h->a[i] = c;
	while (i > 0)
	{
		int			p = (i - 1) / 2;
		cand		t;

		if (!before(h, &h->a[i], &h->a[p]))
			break;
		t = h->a[i];
		h->a[i] = h->a[p];
		h->a[p] = t;
		i = p;
	}
}

static cand
heap_pop(heap * h)
{
	cand		top = h->a[0];
	int			i = 0;

	h->a[0] = h->a[--h->n];
	for (;;)
	{
		int			l = 2 * i + 1,
					r = l + 1,
					b = i;
		cand		t;

		if (l < h->n && before(h, &h->a[l], &h->a[b]))
			b = l;
		if (r < h->n && before(h, &h->a[r], &h->a[b]))
			b = r;
		if (b == i)
			break;
		t = h->a[i];
		h->a[i] = h->a[b];
		h->a[b] = t;
		i = b;
	}
	return top;
}

/* ---------------------------------------------- visited set (tidhash stand-in) */

typedef struct
{
	int32_t    *slots;			/* -1 = empty */
	int			cap,
				used;
}			visited_set;

static void
visited_reset(visited_set * v, int want)
{
	int			cap = 64;

	while (cap < want * 2)
		cap <<= 1;
	if (cap > v->cap)
	{
		v->slots = realloc(v->slots, sizeof(int32_t) * (size_t) cap);
		v->cap = cap;
	}
	memset(v->slots, 0xff, sizeof(int32_t) * (size_t) v->cap);
	v->used = 0;
}

static void visited_grow(visited_set * v);

/* returns 1 when the element was already present (tidhash_insert's `found`) */
static int
visited_insert(visited_set * v, int32_t e)
{
	uint32_t	h = (uint32_t) e * 2654435761u;
	int			mask = v->cap - 1;
	int			i = (int) (h & (uint32_t) mask);

	while (v->slots[i] != -1)
	{
		if (v->slots[i] == e)
			return 1;
		i = (i + 1) & mask;
	}
	v->slots[i] = e;
	if (++v->used * 2 > v->cap)
		visited_grow(v);
	return 0;
}

static void
visited_grow(visited_set * v)
{
	int32_t    *old = v->slots;
	int			oldcap = v->cap;

	v->cap *= 2;
	v->slots = malloc(sizeof(int32_t) * (size_t) v->cap);
	memset(v->slots, 0xff, sizeof(int32_t) * (size_t) v->cap);
	v->used = 0;
	for (int i = 0; i < oldcap; i++)
		if (old[i] != -1)
			visited_insert(v, old[i]);
	free(old);
}

/* ------------------------------------------------------------- one search */

typedef struct
{
	int			lc;				/* current layer */
	int			ef;
	int			done;
	heap		C,
				W;
	int			wlen;
	visited_set v;
	int32_t    *unvisited;		/* [2m] neighbors of the candidate being expanded */
	int			nun;
	int64_t		pair0;			/* where this search's pairs sit in the step's batch */
	int64_t		scored;
}			search;

/* neighbor TIDs of element e at layer lc: HnswLoadNeighborTids, src/hnswutils.c:761-791 */
static inline const int32_t *
layer_neighbors(const pgv_hnsw_graph * g, int32_t e, int lc, int *lm)
{
	int			level = g->levels[e];

	*lm = lc == 0 ? g->m * 2 : g->m;	/* HnswGetLayerM */
	return g->nbr + g->nbr_start[e] + (int64_t) (level - lc) * g->m;
}

/* begin HnswSearchLayer at layer lc from entry candidates ep (src/hnswutils.c:845-886) */
static void
layer_begin(search * s, int lc, int ef, const cand * ep, int nep, int m)
{
	s->lc = lc;
	s->ef = ef;
	s->C.n = s->W.n = 0;
	s->wlen = 0;
	visited_reset(&s->v, ef * m * 2);
	for (int i = 0; i < nep; i++)
	{
		visited_insert(&s->v, ep[i].element);
		if (lc == 0)
			s->scored++;		/* :872-873, only the layer-0 call passes &so->tuples */
		heap_push(&s->C, ep[i]);
		heap_push(&s->W, ep[i]);
		s->wlen++;
	}
}

/*
 * Advance until the search needs distances (returns 1, s->unvisited filled) or the
 * whole search is finished (returns 0).  Layer changes happen inside.
 */
static int
search_advance(search * s, const pgv_hnsw_graph * g, int ef_search)
{
	for (;;)
	{
		int			finished_layer = 0;

		if (s->C.n == 0)
			finished_layer = 1;
		else
		{
			cand		c = heap_pop(&s->C);
			cand		f = s->W.a[0];

			if (c.distance > f.distance)	/* :894 */
				finished_layer = 1;
			else
			{
				int			lm;
				const int32_t *nb = layer_neighbors(g, c.element, s->lc, &lm);

				/* HnswLoadUnvisitedFromDisk, :796-819 */
				s->nun = 0;
				for (int i = 0; i < lm; i++)
				{
					if (nb[i] < 0)	/* !ItemPointerIsValid */
						break;
					if (!visited_insert(&s->v, nb[i]))
						s->unvisited[s->nun++] = nb[i];
				}
				if (s->lc == 0)
					s->scored += s->nun;	/* :905-906 */
				if (s->nun > 0)
					return 1;
				continue;
			}
		}
		if (finished_layer)
		{
			/* w = everything left in W (:978-984) becomes the next layer's entry points */
			int			nw = s->W.n;
			cand	   *w = malloc(sizeof(cand) * (size_t) (nw > 0 ? nw : 1));

			for (int i = 0; i < nw; i++)
				w[i] = heap_pop(&s->W);
			if (s->lc == 0)
			{
				/* keep the final W for the caller */
				for (int i = 0; i < nw; i++)
					heap_push(&s->W, w[i]);
				free(w);
				s->done = 1;
				return 0;
			}
			layer_begin(s, s->lc - 1, s->lc - 1 == 0 ? ef_search : 1, w, nw, g->m);
			free(w);
		}
	}
}

/* the reference's loop body for the scored neighbors, src/hnswutils.c:908-975 */
static void
search_admit(search * s, const pgv_hnsw_graph * g, const float *dist)
{
	for (int i = 0; i < s->nun; i++)
	{
		int32_t		e = s->unvisited[i];
		int			always_add = s->wlen < s->ef;
		cand		f = s->W.a[0];
		cand		c;

		if (!(dist[i] < f.distance || always_add))	/* :936 */
			continue;
		if (g->levels[e] < s->lc)	/* :947 */
			continue;
		c.element = e;
		c.distance = dist[i];
		heap_push(&s->C, c);
		heap_push(&s->W, c);
		s->wlen++;
		if (s->wlen > s->ef)	/* :967-973 */
			heap_pop(&s->W);
	}
}

int
pgv_host_hnsw_search(pgv_hnsw * mirror, const pgv_hnsw_graph * g, pgv_dtype dtype, int dim,
					 const void *queries, int nq, int ef_search, int k,
					 int64_t *out_elem, float *out_dist, int64_t *out_scored)
{
	search	   *ss;
	int32_t    *slot,
			   *qof;
	float	   *dist;
	int64_t		cap;
	int			rc = PGV_OK;
	int			active;

	(void) dtype;
	(void) dim;
	if (!mirror || !g || !queries || !out_elem || !out_dist || nq < 0 || k < 1 || ef_search < 1)
		return pgv_host_fail(PGV_ERR_ARG, "pgv_host_hnsw_search: bad argument");
	for (int64_t i = 0; i < (int64_t) nq * k; i++)
	{
		out_elem[i] = -1;
		out_dist[i] = INFINITY;
	}
	if (out_scored)
		memset(out_scored, 0, sizeof(int64_t) * (size_t) nq);
	if (nq == 0 || g->entry < 0)
		return PGV_OK;			/* empty index: src/hnswscan.c:46-47 */

	cap = (int64_t) nq * g->m * 2;
	ss = calloc((size_t) nq, sizeof(search));
	slot = malloc(sizeof(int32_t) * (size_t) cap);
	qof = malloc(sizeof(int32_t) * (size_t) cap);
	dist = malloc(sizeof(float) * (size_t) cap);

	/* HnswEntryCandidate (src/hnswutils.c:609-621): the entry point's distance, for every query */
	for (int q = 0; q < nq; q++)
	{
		slot[q] = g->entry;
		qof[q] = q;
	}
	rc = pgv_hnsw_score(mirror, queries, nq, slot, qof, nq, dist);
	if (rc != PGV_OK)
	{
		pgv_host_fail(rc, "%s", pgv_last_error());
		goto out;
	}
	for (int q = 0; q < nq; q++)
	{
		cand		ep = {g->entry, dist[q]};
		int			lvl = g->levels[g->entry];

		ss[q].unvisited = malloc(sizeof(int32_t) * (size_t) g->m * 2);
		ss[q].C.nearest_first = 1;
		ss[q].W.nearest_first = 0;
		layer_begin(&ss[q], lvl, lvl == 0 ? ef_search : 1, &ep, 1, g->m);
	}

	/* lock step: one scoring call per expansion step of all live searches */
	for (;;)
	{
		int64_t		np = 0;

		active = 0;
		for (int q = 0; q < nq; q++)
		{
			search	   *s = &ss[q];

			if (s->done)
				continue;
			if (!search_advance(s, g, ef_search))
				continue;
			active++;
			s->pair0 = np;
			for (int i = 0; i < s->nun; i++)
			{
				slot[np] = s->unvisited[i];
				qof[np] = q;
				np++;
			}
		}
		if (!active)
			break;
		rc = pgv_hnsw_score(mirror, queries, nq, slot, qof, np, dist);
		if (rc != PGV_OK)
		{
			pgv_host_fail(rc, "%s", pgv_last_error());
			goto out;
		}
		for (int q = 0; q < nq; q++)
			if (!ss[q].done && ss[q].nun > 0)
			{
				search_admit(&ss[q], g, dist + ss[q].pair0);
				ss[q].nun = 0;
			}
	}

	/* pop W furthest first, emit from the tail = nearest first (src/hnswscan.c:293-311) */
	for (int q = 0; q < nq; q++)
	{
		search	   *s = &ss[q];
		int			nw = s->W.n;
		cand	   *w = malloc(sizeof(cand) * (size_t) (nw > 0 ? nw : 1));

		for (int i = 0; i < nw; i++)
			w[i] = heap_pop(&s->W);
		for (int i = nw - 1, o = 0; i >= 0 && o < k; i--, o++)
		{
			out_elem[(int64_t) q * k + o] = w[i].element;
			out_dist[(int64_t) q * k + o] = w[i].distance;
		}
		if (out_scored)
			out_scored[q] = s->scored;
		free(w);
	}

out:
	for (int q = 0; q < nq; q++)
	{
		free(ss[q].C.a);
		free(ss[q].W.a);
		free(ss[q].v.slots);
		free(ss[q].unvisited);
	}
	free(ss);
	free(slot);
	free(qof);
	free(dist);
	return rc;
}
