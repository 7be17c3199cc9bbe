/*
 * oracle_hnsw.c -- CPU restatement of pgvector's HNSW search (the hot path:
 * src/hnswutils.c:824-987, src/hnswscan.c:25-56) and of the in-memory build
 * (src/hnswutils.c:1040-1357, src/hnswbuild.c:339-476), the latter only so that
 * tests and benchmarks have a graph the reference would have produced
 * (SURVEY Appendix D).  TEST INFRASTRUCTURE ONLY (see pgv_oracle.h).
 *
 * Postgres services are replaced by the plainest equivalent: pairing heaps by
 * binary heaps, the visited hash by a stamp array, Lists by arrays.  Where a
 * heap's order among EQUAL distances is unspecified in the reference it is
 * here too (documented at the comparison).
 */
#include "pgv_oracle.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define HNSW_HEAPTIDS 10		/* src/hnsw.h:69 */

typedef struct
{
	int32_t		element;
	float		distance;
}			hnsw_candidate;		/* HnswCandidate, src/hnsw.h:135-140 */

typedef struct
{
	hnsw_candidate *items;
	int			length;
	int			cap;
}			neighbor_array;

typedef struct
{
	int			level;
	int64_t		row;			/* representative row (first heap tid) */
	int64_t		heaptids[HNSW_HEAPTIDS];
	int			heaptids_length;
	int			dead;			/* parallel build: a duplicate that was folded into another element (never linked) */
	neighbor_array *neighbors;	/* [level + 1] */
}			hnsw_element;

/* the visited set of one searcher (visited_hash, src/hnswutils.c:37-44): a stamp per element */
typedef struct
{
	uint32_t   *v;
	uint32_t	stamp;
	int64_t		cap;
}			visit_set;

struct ora_hnsw
{
	int			ops,
				dtype,
				dim,
				m,
				ef_construction;
	double		ml;
	int			max_level;
	char	   *values;			/* [n x dim] index values (normalised for cosine) */
	size_t		item_bytes;
	hnsw_element *elements;
	int64_t		nelements,
				cap;
	int32_t		entry_point;	/* -1 = empty */
	/* search scratch of the single-threaded callers */
	visit_set	vs;
	/* the parallel in-memory build only (ora_hnsw_build_parallel): one lock per element, "held when reading or
	 * modifying the element's neighbors or heaptids" (src/hnswbuild.c:20-21); NULL otherwise */
	pthread_rwlock_t *locks;
};

static inline int
layer_m(int m, int lc)
{
	return lc == 0 ? m * 2 : m; /* HnswGetLayerM, src/hnsw.h:127 */
}

static inline const void *
value_of(const ora_hnsw * g, int32_t e)
{
	return g->values + (size_t) g->elements[e].row * g->item_bytes;
}

static inline double
dist_qe(const ora_hnsw * g, const void *q, int32_t e)
{
	/* HnswGetDistance(q, element), src/hnswutils.c:524-528 */
	return ora_index_distance(g->ops, g->dtype, g->dim, q, value_of(g, e));
}

/* ---------------------------------------------------------------- heaps */

typedef struct
{
	int32_t		element;
	double		distance;
}			search_candidate;	/* HnswSearchCandidate, src/hnsw.h:216-222 */

typedef struct
{
	search_candidate *a;
	int			n,
				cap;
	int			nearest_first;
}			heap;

static int
heap_before(const heap * h, const search_candidate * x, const search_candidate * y)
{
	/* CompareNearestCandidates / CompareFurthestCandidates (src/hnswutils.c:626-654);
	 * equal distances: unspecified there, element id here */
	if (x->distance != y->distance)
		return h->nearest_first ? x->distance < y->distance : x->distance > y->distance;
	return x->element < y->element;
}

static void
heap_push(heap * h, search_candidate c)
{
	int			i;

	if (h->n == h->cap)
	{
		h->cap = h->cap ? h->cap * 2 : 64;
		h->a = realloc(h->a, sizeof(search_candidate) * (size_t) h->cap);
	}
	i = h->n++;
	h->a[i] = c;
	while (i > 0)
	{
		int			p = (i - 1) / 2;

		if (!heap_before(h, &h->a[i], &h->a[p]))
			break;
		search_candidate t = h->a[i];

		h->a[i] = h->a[p];
		h->a[p] = t;
		i = p;
	}
}

static search_candidate
heap_pop(heap * h)
{
	search_candidate top = h->a[0];
	int			i = 0;

	h->a[0] = h->a[--h->n];
	for (;;)
	{
		int			l = 2 * i + 1,
					r = l + 1,
					b = i;

		if (l < h->n && heap_before(h, &h->a[l], &h->a[b]))
			b = l;
		if (r < h->n && heap_before(h, &h->a[r], &h->a[b]))
			b = r;
		if (b == i)
			break;
		search_candidate t = h->a[i];

		h->a[i] = h->a[b];
		h->a[b] = t;
		i = b;
	}
	return top;
}

/* ----------------------------------------------------------- SearchLayer */

static void
visited_begin(visit_set * vs, int64_t nelements)
{
	if (vs->cap < nelements + 1)
	{
		vs->cap = (nelements + 1) * 2;
		vs->v = realloc(vs->v, sizeof(uint32_t) * (size_t) vs->cap);
		memset(vs->v, 0, sizeof(uint32_t) * (size_t) vs->cap);
		vs->stamp = 0;
	}
	if (++vs->stamp == 0)
	{
		memset(vs->v, 0, sizeof(uint32_t) * (size_t) vs->cap);
		vs->stamp = 1;
	}
}

/*
 * Algorithm 2, src/hnswutils.c:824-987.  ep/w: arrays of search candidates; the
 * result is ordered furthest first like the reference's list `w` (:978-984).
 * Returns the result count; *scored (may be NULL) accumulates so->tuples (:872-873, :905-906).
 */
static int
search_layer(ora_hnsw * g, visit_set * vs, const void *q, const search_candidate * ep, int nep, int ef, int lc,
			 search_candidate * *out, int64_t *scored)
{
	heap		C = {0},
				W = {0};
	int			wlen = 0;
	int			lm = layer_m(g->m, lc);
	int32_t    *unvisited = malloc(sizeof(int32_t) * (size_t) lm);
	search_candidate *res;
	int			nres;

	C.nearest_first = 1;
	W.nearest_first = 0;
	visited_begin(vs, g->nelements);

	for (int i = 0; i < nep; i++)
	{
		vs->v[ep[i].element] = vs->stamp;
		if (scored)
			(*scored)++;
		heap_push(&C, ep[i]);
		heap_push(&W, ep[i]);
		wlen++;
	}

	while (C.n > 0)
	{
		search_candidate c = heap_pop(&C);
		search_candidate f = W.a[0];
		const hnsw_element *ce;
		int			nun = 0;

		if (c.distance > f.distance)	/* :894 */
			break;

		ce = &g->elements[c.element];
		/* HnswLoadUnvisitedFromMemory (:731-759) / ...FromDisk (:796-819): same order.  In the parallel build the
		 * neighborhood is read under the element's lock (shared), :738-742 */
		if (lc <= ce->level)
		{
			const neighbor_array *na = &ce->neighbors[lc];

			if (g->locks)
				pthread_rwlock_rdlock(&g->locks[c.element]);
			for (int i = 0; i < na->length; i++)
			{
				int32_t		e = na->items[i].element;

				if (vs->v[e] != vs->stamp)
				{
					vs->v[e] = vs->stamp;
					unvisited[nun++] = e;
				}
			}
			if (g->locks)
				pthread_rwlock_unlock(&g->locks[c.element]);
		}
		if (scored)
			*scored += nun;

		for (int i = 0; i < nun; i++)
		{
			int32_t		e = unvisited[i];
			int			always_add = wlen < ef;
			double		e_distance;
			search_candidate sc;

			f = W.a[0];
			e_distance = dist_qe(g, q, e);	/* the candidate-scoring call, :913-930 */

			if (!(e_distance < f.distance || always_add))	/* :936 */
				continue;
			if (g->elements[e].level < lc)	/* :947 */
				continue;

			sc.element = e;
			sc.distance = e_distance;
			heap_push(&C, sc);
			heap_push(&W, sc);
			wlen++;
			if (wlen > ef)		/* :967-973 */
				heap_pop(&W);
		}
	}

	nres = W.n;
	res = malloc(sizeof(search_candidate) * (size_t) (nres > 0 ? nres : 1));
	for (int i = 0; i < nres; i++)
		res[i] = heap_pop(&W);	/* furthest first */
	*out = res;
	free(unvisited);
	free(C.a);
	free(W.a);
	return nres;
}

/* ------------------------------------------------------ SelectNeighbors */

/* CompareCandidateDistances, src/hnswutils.c:992-1010: descending distance, then
 * descending pointer; element ids grow with allocation order and stand in for pointers */
static int
cand_desc_cmp(const void *pa, const void *pb)
{
	const hnsw_candidate *a = *(hnsw_candidate * const *) pa,
			   *b = *(hnsw_candidate * const *) pb;

	if (a->distance < b->distance)
		return 1;
	if (a->distance > b->distance)
		return -1;
	if (a->element < b->element)
		return 1;
	if (a->element > b->element)
		return -1;
	return 0;
}

/* CheckElementCloser, src/hnswutils.c:1040-1059 */
static int
check_element_closer(const ora_hnsw * g, const hnsw_candidate * e, hnsw_candidate * *r, int nr)
{
	const void *ev = value_of(g, e->element);

	for (int i = 0; i < nr; i++)
	{
		float		distance = (float) ora_index_distance(g->ops, g->dtype, g->dim, ev, value_of(g, r[i]->element));

		if (distance <= e->distance)
			return 0;
	}
	return 1;
}

/*
 * Algorithm 4, src/hnswutils.c:1064-1165.  The reference caches `closer` flags
 * between calls (closerSet); that cache is an exact shortcut of recomputing them
 * (same deterministic order, same comparisons), so they are recomputed here.
 * c: candidate pointers, ordered furthest first unless sort != 0.
 * Returns |r|; r (size lm) receives the selection; *pruned the dropped candidate.
 */
static int
select_neighbors(const ora_hnsw * g, hnsw_candidate * *c, int nc, int lm, hnsw_candidate * *r,
				 hnsw_candidate * *pruned, int sort)
{
	hnsw_candidate **w,
			  **wd;
	int			wn = nc,
				rn = 0,
				wdlen = 0,
				wdoff = 0;

	if (nc <= lm)
	{
		for (int i = 0; i < nc; i++)
			r[i] = c[i];
		return nc;
	}
	w = malloc(sizeof(*w) * (size_t) nc);
	wd = malloc(sizeof(*wd) * (size_t) nc);
	memcpy(w, c, sizeof(*w) * (size_t) nc);
	if (sort)
		qsort(w, (size_t) nc, sizeof(*w), cand_desc_cmp);

	while (wn > 0 && rn < lm)
	{
		hnsw_candidate *e = w[--wn];	/* closest remaining */

		if (check_element_closer(g, e, r, rn))
			r[rn++] = e;
		else
			wd[wdlen++] = e;
	}
	/* keep pruned connections (:1148-1150) */
	while (wdoff < wdlen && rn < lm)
		r[rn++] = wd[wdoff++];
	if (pruned)
		*pruned = wdoff < wdlen ? wd[wdoff] : w[0];	/* :1153-1159 */
	free(w);
	free(wd);
	return rn;
}

static void
neighbors_push(neighbor_array * a, hnsw_candidate hc)
{
	if (a->length == a->cap)
	{
		a->cap = a->cap ? a->cap * 2 : 8;
		a->items = realloc(a->items, sizeof(hnsw_candidate) * (size_t) a->cap);
	}
	a->items[a->length++] = hc;
}

/* HnswUpdateConnection, src/hnswutils.c:1183-1231 */
static void
update_connection(ora_hnsw * g, neighbor_array * neighbors, int32_t new_element, float distance, int lm)
{
	hnsw_candidate new_hc = {new_element, distance};

	if (neighbors->length < lm)
	{
		neighbors_push(neighbors, new_hc);
		return;
	}
	{
		int			nc = neighbors->length + 1;
		hnsw_candidate **c = malloc(sizeof(*c) * (size_t) nc);
		hnsw_candidate **r = malloc(sizeof(*r) * (size_t) lm);
		hnsw_candidate *pruned = NULL;

		for (int i = 0; i < neighbors->length; i++)
			c[i] = &neighbors->items[i];
		c[nc - 1] = &new_hc;
		select_neighbors(g, c, nc, lm, r, &pruned, 1);
		if (pruned != NULL)
			for (int i = 0; i < neighbors->length; i++)
				if (neighbors->items[i].element == pruned->element)
				{
					neighbors->items[i] = new_hc;
					break;
				}
		free(r);
		free(c);
	}
}

/* ----------------------------------------------------------------- build */

/* Algorithm 1, src/hnswutils.c:1280-1357 (in-memory, not `existing`) */
static void
find_element_neighbors(ora_hnsw * g, visit_set * vs, int32_t element, int32_t entry_point)
{
	hnsw_element *el = &g->elements[element];
	const void *q = value_of(g, element);
	int			level = el->level;
	int			entry_level;
	search_candidate *ep,
			   *w = NULL;
	int			nep,
				nw = 0;

	if (entry_point < 0)
		return;

	ep = malloc(sizeof(search_candidate));
	ep[0].element = entry_point;
	ep[0].distance = dist_qe(g, q, entry_point);	/* HnswEntryCandidate */
	nep = 1;
	entry_level = g->elements[entry_point].level;

	/* 1st phase: greedy search to the insert level */
	for (int lc = entry_level; lc >= level + 1; lc--)
	{
		nw = search_layer(g, vs, q, ep, nep, 1, lc, &w, NULL);
		free(ep);
		ep = w;
		nep = nw;
	}
	if (level > entry_level)
		level = entry_level;

	/* 2nd phase */
	for (int lc = level; lc >= 0; lc--)
	{
		int			lm = layer_m(g->m, lc);
		hnsw_candidate *lw;
		hnsw_candidate **c,
				  **r;
		int			rn;

		nw = search_layer(g, vs, q, ep, nep, g->ef_construction, lc, &w, NULL);
		lw = malloc(sizeof(hnsw_candidate) * (size_t) (nw > 0 ? nw : 1));
		c = malloc(sizeof(*c) * (size_t) (nw > 0 ? nw : 1));
		r = malloc(sizeof(*r) * (size_t) (lm > nw ? lm : nw + 1));
		for (int i = 0; i < nw; i++)
		{
			lw[i].element = w[i].element;
			lw[i].distance = (float) w[i].distance; /* :1327 */
			c[i] = &lw[i];
		}
		rn = select_neighbors(g, c, nw, lm, r, NULL, 0);
		for (int i = 0; i < rn; i++)	/* AddConnections, :1170-1178 */
			neighbors_push(&g->elements[element].neighbors[lc], *r[i]);
		free(r);
		free(c);
		free(lw);
		free(ep);
		ep = w;
		nep = nw;
	}
	free(ep);
}

/* FindDuplicateInMemory + AddDuplicateInMemory, src/hnswbuild.c:313-364 */
static int
find_duplicate(ora_hnsw * g, int32_t element)
{
	hnsw_element *el = &g->elements[element];
	const neighbor_array *na = &el->neighbors[0];
	const void *v = value_of(g, element);

	for (int i = 0; i < na->length; i++)
	{
		hnsw_element *ne = &g->elements[na->items[i].element];

		int			added = 0;

		if (memcmp(v, g->values + (size_t) ne->row * g->item_bytes, g->item_bytes) != 0)
			return 0;			/* neighbors are ordered by distance: stop at the first different value */
		/* AddDuplicateInMemory, :322-337: under the duplicate's lock in the parallel build */
		if (g->locks)
			pthread_rwlock_wrlock(&g->locks[na->items[i].element]);
		if (ne->heaptids_length < HNSW_HEAPTIDS)
		{
			ne->heaptids[ne->heaptids_length++] = el->heaptids[0];
			added = 1;
		}
		if (g->locks)
			pthread_rwlock_unlock(&g->locks[na->items[i].element]);
		if (added)
			return 1;
	}
	return 0;
}

static void
free_element(hnsw_element * el)
{
	for (int lc = 0; lc <= el->level; lc++)
		free(el->neighbors[lc].items);
	free(el->neighbors);
}

ora_hnsw *
ora_hnsw_build(int ops, int dtype, int dim, const void *rows, int64_t n, int m, int ef_construction, uint64_t seed)
{
	ora_hnsw   *g = calloc(1, sizeof(ora_hnsw));
	ora_prng	rng;
	size_t		es = dtype == ORA_F32 ? sizeof(float) : sizeof(ora_half);
	/* HnswGetMaxLevel, src/hnsw.h:133 with BLCKSZ 8192 */
	int			by_page = (int) ((8192 - 24 - 8 - 4 - 4) / 6 / m) - 2;

	g->ops = ops;
	g->dtype = dtype;
	g->dim = dim;
	g->m = m;
	g->ef_construction = ef_construction;
	g->ml = 1.0 / log((double) m);	/* HnswGetMl */
	g->max_level = by_page < 63 ? by_page : 63;
	g->item_bytes = (size_t) dim * es;
	g->values = malloc(g->item_bytes * (size_t) (n > 0 ? n : 1));
	g->entry_point = -1;
	g->cap = n > 0 ? n : 1;
	g->elements = malloc(sizeof(hnsw_element) * (size_t) g->cap);
	ora_prng_seed(&rng, seed);

	for (int64_t row = 0; row < n; row++)
	{
		const void *src = (const char *) rows + (size_t) row * g->item_bytes;
		void	   *dst = g->values + (size_t) row * g->item_bytes;
		hnsw_element *el;
		int32_t		id;
		int			level;

		/* HnswFormIndexValue, src/hnswutils.c:406-428: cosine normalises, skips zero norm */
		if (ops == ORA_OPS_COSINE)
		{
			double		norm = dtype == ORA_F32 ? ora_vector_norm(dim, src) : ora_halfvec_l2_norm(dim, src);

			if (!(norm > 0))
				continue;
			if (dtype == ORA_F32)
				ora_l2_normalize(dim, src, dst);
			else
				ora_halfvec_l2_normalize(dim, src, dst);
		}
		else
			memcpy(dst, src, g->item_bytes);

		/* HnswInitElement, src/hnswutils.c:243-270 */
		level = (int) (-log(ora_prng_double(&rng)) * g->ml);
		if (level > g->max_level)
			level = g->max_level;

		id = (int32_t) g->nelements;
		el = &g->elements[id];
		el->level = level;
		el->row = row;
		el->heaptids[0] = row;
		el->heaptids_length = 1;
		el->dead = 0;
		el->neighbors = calloc((size_t) level + 1, sizeof(neighbor_array));
		g->nelements++;			/* visible to the search (it is never its own neighbor: not linked yet) */

		/* InsertTupleInMemory, src/hnswbuild.c:436-476 */
		find_element_neighbors(g, &g->vs, id, g->entry_point);

		/* UpdateGraphInMemory, :410-431 */
		if (find_duplicate(g, id))
		{
			free_element(el);
			g->nelements--;
			continue;
		}
		/* UpdateNeighborsInMemory, :376-405 */
		for (int lc = el->level; lc >= 0; lc--)
		{
			int			lm = layer_m(g->m, lc);
			neighbor_array snapshot = g->elements[id].neighbors[lc];

			for (int i = 0; i < snapshot.length; i++)
			{
				hnsw_candidate hc = snapshot.items[i];

				update_connection(g, &g->elements[hc.element].neighbors[lc], id, hc.distance, lm);
			}
		}
		if (g->entry_point < 0 || el->level > g->elements[g->entry_point].level)
			g->entry_point = id;
	}
	return g;
}

/* ------------------------------------------------- the parallel in-memory build
 *
 * src/hnswbuild.c:1-40 (the locking rules), :366-480 (InsertTupleInMemory and what it calls), :700-1100 (workers
 * scan the heap in parallel and insert into ONE shared graph).  Restated with threads: a lock per element taken
 * shared to read a neighborhood (src/hnswutils.c:738-742) and exclusive to update it (:403-405) or to add a heap TID
 * (:324-334); the entry point behind a reader/writer lock that an insert takes exclusive only when it may move the
 * entry point, announced through a gate every insert passes first (:449-470).  Levels are drawn in heap order from
 * the one seeded stream before the threads start, so they are the serial build's; which neighbors an element finds
 * depends on what the other threads have linked by then -- as in the reference, the graph is not deterministic.
 * With one thread the result IS the serial build's graph (tested).
 */
typedef struct
{
	ora_hnsw   *g;
	int64_t		next;			/* next element to insert (atomic) */
	pthread_rwlock_t entry_lock;
	pthread_mutex_t entry_gate;
}			par_build;

static void
insert_element_parallel(par_build * pb, visit_set * vs, int32_t id)
{
	ora_hnsw   *g = pb->g;
	hnsw_element *el = &g->elements[id];
	int32_t		ep;

	/* wait while somebody is about to take the entry lock exclusively (:449-451) */
	pthread_mutex_lock(&pb->entry_gate);
	pthread_mutex_unlock(&pb->entry_gate);
	pthread_rwlock_rdlock(&pb->entry_lock);
	ep = g->entry_point;
	if (ep < 0 || el->level > g->elements[ep].level)
	{
		/* this insert is likely to move the entry point: alone from here on (:457-470) */
		pthread_rwlock_unlock(&pb->entry_lock);
		pthread_mutex_lock(&pb->entry_gate);
		pthread_rwlock_wrlock(&pb->entry_lock);
		pthread_mutex_unlock(&pb->entry_gate);
		ep = g->entry_point;
	}

	find_element_neighbors(g, vs, id, ep);

	/* UpdateGraphInMemory, :410-431 */
	if (find_duplicate(g, id))
		el->dead = 1;
	else
	{
		/* UpdateNeighborsInMemory, :379-407: this element's lists are copied under its own lock (others may be
		 * linking to it by now), every neighbor is updated under the neighbor's lock */
		for (int lc = el->level; lc >= 0; lc--)
		{
			int			lm = layer_m(g->m, lc);
			hnsw_candidate *mine = malloc(sizeof(hnsw_candidate) * (size_t) lm);
			int			nmine;

			pthread_rwlock_rdlock(&g->locks[id]);
			nmine = el->neighbors[lc].length;
			memcpy(mine, el->neighbors[lc].items, sizeof(hnsw_candidate) * (size_t) nmine);
			pthread_rwlock_unlock(&g->locks[id]);
			for (int i = 0; i < nmine; i++)
			{
				int32_t		nb = mine[i].element;

				pthread_rwlock_wrlock(&g->locks[nb]);
				update_connection(g, &g->elements[nb].neighbors[lc], id, mine[i].distance, lm);
				pthread_rwlock_unlock(&g->locks[nb]);
			}
			free(mine);
		}
		/* the entry point moves under the exclusive lock this insert took for that case (:428-430) */
		if (ep < 0 || el->level > g->elements[ep].level)
			g->entry_point = id;
	}
	pthread_rwlock_unlock(&pb->entry_lock);
}

static void *
par_build_worker(void *arg)
{
	par_build  *pb = arg;
	visit_set	vs = {0};

	for (;;)
	{
		int64_t		id = __atomic_fetch_add(&pb->next, 1, __ATOMIC_RELAXED);

		if (id >= pb->g->nelements)
			break;
		insert_element_parallel(pb, &vs, (int32_t) id);
	}
	free(vs.v);
	return NULL;
}

ora_hnsw *
ora_hnsw_build_parallel(int ops, int dtype, int dim, const void *rows, int64_t n, int m, int ef_construction,
						uint64_t seed, int nthreads)
{
	ora_hnsw   *g = calloc(1, sizeof(ora_hnsw));
	ora_prng	rng;
	size_t		es = dtype == ORA_F32 ? sizeof(float) : sizeof(ora_half);
	int			by_page = (int) ((8192 - 24 - 8 - 4 - 4) / 6 / m) - 2;
	par_build	pb;
	pthread_t  *threads;
	int64_t		live = 0;
	int32_t    *new_id;

	if (nthreads < 1)
		nthreads = 1;
	g->ops = ops;
	g->dtype = dtype;
	g->dim = dim;
	g->m = m;
	g->ef_construction = ef_construction;
	g->ml = 1.0 / log((double) m);
	g->max_level = by_page < 63 ? by_page : 63;
	g->item_bytes = (size_t) dim * es;
	g->values = malloc(g->item_bytes * (size_t) (n > 0 ? n : 1));
	g->entry_point = -1;
	g->cap = n > 0 ? n : 1;
	g->elements = malloc(sizeof(hnsw_element) * (size_t) g->cap);
	ora_prng_seed(&rng, seed);

	/* every element formed before the first insert: index value, level (the serial build's stream), neighbor
	 * lists at their final capacity (HnswInitNeighbors, src/hnswutils.c:218-238: no list ever moves) */
	for (int64_t row = 0; row < n; row++)
	{
		const void *src = (const char *) rows + (size_t) row * g->item_bytes;
		void	   *dst = g->values + (size_t) row * g->item_bytes;
		hnsw_element *el;
		int			level;

		if (ops == ORA_OPS_COSINE)
		{
			double		norm = dtype == ORA_F32 ? ora_vector_norm(dim, src) : ora_halfvec_l2_norm(dim, src);

			if (!(norm > 0))
				continue;
			if (dtype == ORA_F32)
				ora_l2_normalize(dim, src, dst);
			else
				ora_halfvec_l2_normalize(dim, src, dst);
		}
		else
			memcpy(dst, src, g->item_bytes);
		level = (int) (-log(ora_prng_double(&rng)) * g->ml);
		if (level > g->max_level)
			level = g->max_level;
		el = &g->elements[g->nelements++];
		el->level = level;
		el->row = row;
		el->heaptids[0] = row;
		el->heaptids_length = 1;
		el->dead = 0;
		el->neighbors = calloc((size_t) level + 1, sizeof(neighbor_array));
		for (int lc = 0; lc <= level; lc++)
		{
			el->neighbors[lc].cap = layer_m(m, lc);
			el->neighbors[lc].items = malloc(sizeof(hnsw_candidate) * (size_t) el->neighbors[lc].cap);
		}
	}

	g->locks = malloc(sizeof(pthread_rwlock_t) * (size_t) (g->nelements > 0 ? g->nelements : 1));
	for (int64_t e = 0; e < g->nelements; e++)
		pthread_rwlock_init(&g->locks[e], NULL);
	pb.g = g;
	pb.next = 0;
	pthread_rwlock_init(&pb.entry_lock, NULL);
	pthread_mutex_init(&pb.entry_gate, NULL);
	threads = malloc(sizeof(pthread_t) * (size_t) nthreads);
	for (int t = 1; t < nthreads; t++)
		if (pthread_create(&threads[t], NULL, par_build_worker, &pb) != 0)
			threads[t] = 0;
	par_build_worker(&pb);
	for (int t = 1; t < nthreads; t++)
		if (threads[t])
			pthread_join(threads[t], NULL);
	free(threads);
	pthread_mutex_destroy(&pb.entry_gate);
	pthread_rwlock_destroy(&pb.entry_lock);
	for (int64_t e = 0; e < g->nelements; e++)
		pthread_rwlock_destroy(&g->locks[e]);
	free(g->locks);
	g->locks = NULL;

	/* duplicates were never linked: drop them and renumber, as the serial build's reuse of their slot does */
	new_id = malloc(sizeof(int32_t) * (size_t) (g->nelements > 0 ? g->nelements : 1));
	for (int64_t e = 0; e < g->nelements; e++)
		new_id[e] = g->elements[e].dead ? -1 : (int32_t) live++;
	if (live != g->nelements)
	{
		for (int64_t e = 0; e < g->nelements; e++)
		{
			hnsw_element *el = &g->elements[e];

			if (el->dead)
			{
				free_element(el);
				continue;
			}
			for (int lc = 0; lc <= el->level; lc++)
				for (int i = 0; i < el->neighbors[lc].length; i++)
					el->neighbors[lc].items[i].element = new_id[el->neighbors[lc].items[i].element];
			g->elements[new_id[e]] = *el;
		}
		if (g->entry_point >= 0)
			g->entry_point = new_id[g->entry_point];
		g->nelements = live;
	}
	free(new_id);
	return g;
}

/*
 * A graph somebody else built (the GPU build, a staged index), given the way the index stores it: per element its
 * level and the neighbor tuple's (level + 2) * m slots, layer lc at (level - lc) * m, an invalid slot (-1) ends a
 * layer's list (HnswLoadNeighborTids, src/hnswutils.c:761-794).  `values` are the INDEX values (already
 * normalised for cosine); element e answers with heap row e.  Only ora_hnsw_search may be used on the result.
 */
ora_hnsw *
ora_hnsw_import(int ops, int dtype, int dim, const void *values, int64_t n, int m, const int32_t *levels,
				const int64_t *nbr_start, const int32_t *nbr, int32_t entry)
{
	ora_hnsw   *g = calloc(1, sizeof(ora_hnsw));
	size_t		es = dtype == ORA_F32 ? sizeof(float) : sizeof(ora_half);

	g->ops = ops;
	g->dtype = dtype;
	g->dim = dim;
	g->m = m;
	g->ml = 1.0 / log((double) m);
	g->max_level = 63;
	g->item_bytes = (size_t) dim * es;
	g->values = malloc(g->item_bytes * (size_t) (n > 0 ? n : 1));
	memcpy(g->values, values, g->item_bytes * (size_t) n);
	g->cap = n > 0 ? n : 1;
	g->elements = calloc((size_t) g->cap, sizeof(hnsw_element));
	g->nelements = n;
	g->entry_point = n > 0 ? entry : -1;
	for (int64_t e = 0; e < n; e++)
	{
		hnsw_element *el = &g->elements[e];

		el->level = levels[e];
		el->row = e;
		el->heaptids[0] = e;
		el->heaptids_length = 1;
		el->dead = 0;
		el->neighbors = calloc((size_t) el->level + 1, sizeof(neighbor_array));
		for (int lc = 0; lc <= el->level; lc++)
		{
			int			lm = layer_m(m, lc);
			const int32_t *src = nbr + nbr_start[e] + (int64_t) (el->level - lc) * m;
			neighbor_array *a = &el->neighbors[lc];

			a->items = malloc(sizeof(hnsw_candidate) * (size_t) lm);
			a->cap = lm;
			for (int i = 0; i < lm && src[i] >= 0; i++)
			{
				a->items[a->length].element = src[i];
				a->items[a->length].distance = 0.f;	/* not used by the search */
				a->length++;
			}
		}
	}
	return g;
}

void
ora_hnsw_free(ora_hnsw * g)
{
	if (!g)
		return;
	for (int64_t e = 0; e < g->nelements; e++)
		free_element(&g->elements[e]);
	free(g->elements);
	free(g->values);
	free(g->vs.v);
	free(g);
}

int64_t
ora_hnsw_num_elements(const ora_hnsw * g)
{
	return g->nelements;
}

int
ora_hnsw_entry_point(const ora_hnsw * g, int *level)
{
	if (level)
		*level = g->entry_point >= 0 ? g->elements[g->entry_point].level : -1;
	return g->entry_point;
}

int
ora_hnsw_m(const ora_hnsw * g)
{
	return g->m;
}

int
ora_hnsw_level(const ora_hnsw * g, int64_t e)
{
	return g->elements[e].level;
}

int
ora_hnsw_neighbors(const ora_hnsw * g, int64_t e, int lc, int32_t *out)
{
	const neighbor_array *na;

	if (lc > g->elements[e].level)
		return 0;
	na = &g->elements[e].neighbors[lc];
	for (int i = 0; i < na->length; i++)
		out[i] = na->items[i].element;
	return na->length;
}

int64_t
ora_hnsw_element_row(const ora_hnsw * g, int64_t e)
{
	return g->elements[e].row;
}

/* GetScanItems (src/hnswscan.c:25-56) + the pop loop of hnswgettuple (:293-326) */
int
ora_hnsw_search(const ora_hnsw * gc, const void *query, int ef_search, int k,
				int64_t *out_rows, double *out_dist, int64_t *out_scored)
{
	ora_hnsw   *g = (ora_hnsw *) gc;	/* scratch only */
	void	   *nq = NULL;
	const void *q = query;
	search_candidate *ep,
			   *w = NULL;
	int			nep,
				nw,
				written = 0;
	int64_t		scored = 0;

	if (g->entry_point < 0)
		return 0;
	/* GetScanValue, src/hnswscan.c:92-114 */
	if (g->ops == ORA_OPS_COSINE)
	{
		nq = malloc(g->item_bytes);
		if (g->dtype == ORA_F32)
			ora_l2_normalize(g->dim, query, nq);
		else
			ora_halfvec_l2_normalize(g->dim, query, nq);
		q = nq;
	}
	ep = malloc(sizeof(search_candidate));
	ep[0].element = g->entry_point;
	ep[0].distance = dist_qe(g, q, g->entry_point);
	nep = 1;
	for (int lc = g->elements[g->entry_point].level; lc >= 1; lc--)
	{
		nw = search_layer(g, &g->vs, q, ep, nep, 1, lc, &w, NULL);
		free(ep);
		ep = w;
		nep = nw;
	}
	nw = search_layer(g, &g->vs, q, ep, nep, ef_search, 0, &w, &scored);
	free(ep);

	/* nearest first = from the tail of w; each element emits its heap tids last to first */
	for (int i = nw - 1; i >= 0 && written < k; i--)
	{
		hnsw_element *el = &g->elements[w[i].element];

		for (int t = el->heaptids_length - 1; t >= 0 && written < k; t--)
		{
			out_rows[written] = el->heaptids[t];
			if (out_dist)
				out_dist[written] = w[i].distance;
			written++;
		}
	}
	if (out_scored)
		*out_scored = scored;
	free(w);
	free(nq);
	return written;
}
