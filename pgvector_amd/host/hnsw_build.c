/*
 * hnsw_build.c -- the in-memory phase of CREATE INDEX ... USING hnsw
 * (src/hnswbuild.c:436-476 InsertTupleInMemory, :376-431 UpdateGraphInMemory) with every
 * distance computed on the GPU.
 *
 * The reference inserts one element at a time: HnswFindElementNeighbors
 * (src/hnswutils.c:1280-1357) searches the graph from the entry point, SelectNeighbors
 * (:1064-1165) thins each layer's candidate list with CheckElementCloser (:1040-1059), then
 * HnswUpdateConnection (:1183-1231) links the element into its neighbors' lists, re-running the
 * selection on a list that is already full.  Its parallel build runs that loop in several
 * workers at once on shared memory.  Here a BATCH of elements is inserted "at once" the same
 * way: all of them search the graph as it stood when the batch began (one pgv_hnsw_build_search
 * launch: none of them is linked yet, so none can find another), every distance the selections
 * can need is fetched in one pgv_hnsw_score_pairs launch per phase, and the graph updates are
 * applied in heap order -- a legal interleaving of the reference's concurrent workers.  With
 * max_batch = 1 this IS the reference's serial loop.  The host replay itself is spread over
 * OpenMP threads where the reference's steps are independent of each other (the selections of
 * different new elements; the updates of different neighbor lists), standing in for its
 * parallel maintenance workers.
 *
 * Which distances a selection needs depends on its outcome, but never the distances
 * themselves, so they are fetched ahead of the replay: for the candidate list of a new element
 * the whole pairwise matrix; for a neighbor list that a batch links into, what the reference's
 * closer-flag cache (:1098-1140) makes it compute -- the whole matrix when the list has no cached
 * flags yet (first overflow), otherwise only the distances that involve the newcomers.  The rare
 * replay that steps outside that (a cached `closer` member losing its flag makes the reference
 * re-check earlier rejects against the whole selection) is put aside, the missing pairs of its
 * list are fetched in a second launch, and it is replayed then; lists are independent, so that
 * changes nothing.
 */
#include "pgv_host.h"

#include <math.h>
#include <omp.h>
#include <pthread.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

extern int	pgv_host_fail(int code, const char *fmt,...);

#define HNSW_HEAPTIDS 10		/* src/hnsw.h:69 */

typedef struct
{
	int32_t		element;
	float		distance;
	int32_t		local;			/* index into the current record's matrix (valid within a batch) */
	uint8_t		closer;			/* HnswCandidate.closer: cached outcome of CheckElementCloser */
}			cand;

typedef struct
{
	int			length;
	uint8_t		closer_set;		/* HnswNeighborArray.closerSet: the cached flags are usable */
	uint8_t		ord_valid;		/* ord[] is the items' order by CompareCandidateDistances (furthest first) */
	cand	   *items;			/* capacity lm; ord[lm] (item indexes) sits behind it in the same allocation */
}			nlist;

/* a list's items and, behind them, the cached sort order (layer_m <= 200 fits a byte) */
static inline cand *
items_alloc(int lm)
{
	return malloc(sizeof(cand) * (size_t) lm + (size_t) lm);
}

static inline uint8_t *
items_ord(const nlist * l, int lm)
{
	return (uint8_t *) (l->items + lm);
}

typedef struct
{
	int32_t		level;
	int32_t		heaptids;		/* heap TIDs attached (duplicates), src/hnsw.h:69 */
	nlist	   *layers;			/* [level + 1], allocated when the element is linked */
}			elem;

/* a neighbor list touched by a batch, with everything its re-selections can look up */
typedef struct
{
	int32_t		owner;
	int32_t		lc;
	int			nstart;			/* members when the batch began: locals 0 .. nstart - 1 */
	int			nlocal;			/* + the batch elements that selected the owner */
	int32_t    *ids;			/* [nlocal] element of each local */
	const void *items;			/* the list's items (for fetching them ahead in step 5) */
	int64_t		pair0;			/* first of its pairs in its slice's request */
	int			slice;			/* the scoring request (of step 4) that holds them */
	int64_t		pair0b;			/* first of its member-member pairs in the second request (blocked lists) */
	int			full;			/* every pair was requested; otherwise only those with a newcomer (local >= nstart) */
	int			from;			/* the pairs (u, v < u) with u >= from were requested (1: all) */
	int			blocked;		/* an update is waiting for the member-member pairs */
	float	   *newdist;		/* [nlocal - nstart] distance of each newcomer to the owner */
	int			wait_from;		/* first local whose update waits for the second launch */
	int			newcap;			/* newcomers of the batch (counted in step 4a) */
}			record;

typedef struct
{
	/* xoroshiro128** like the library's own source (pgv_abi_common.h Xoro) when no callbacks are given */
	uint64_t	s0,
				s1;
	const pgv_rng *user;
}			rng_state;

static uint64_t
rotl64(uint64_t x, int k)
{
	return (x << k) | (x >> (64 - k));
}

static uint64_t
splitmix64(uint64_t *st)
{
	uint64_t	v = (*st += 0x9E3779B97f4A7C15ull);

	v = (v ^ (v >> 30)) * 0xBF58476D1CE4E5B9ull;
	v = (v ^ (v >> 27)) * 0x94D049BB133111EBull;
	return v ^ (v >> 31);
}

static void
rng_init(rng_state * r, const pgv_rng * user)
{
	uint64_t	seed = user ? user->seed : 0;

	r->user = user;
	r->s0 = splitmix64(&seed);
	r->s1 = splitmix64(&seed);
	if (!r->s0 && !r->s1)
		r->s0 = 1;
}

static double
rng_double(rng_state * r)
{
	uint64_t	a,
				x,
				out;

	if (r->user && r->user->next_double)
		return r->user->next_double(r->user->state);
	a = r->s0;
	x = r->s1 ^ a;
	out = rotl64(a * 5, 7) * 9;
	r->s0 = rotl64(a, 24) ^ x ^ (x << 16);
	r->s1 = rotl64(x, 37);
	return ldexp((double) (out >> 12), -52);
}

/* per-thread bump allocator for a batch's records (ids, distance matrices): reset at the end of the batch instead of
 * a malloc/free pair per list -- a 100 k x 1536 build touched 4.3 M lists and spent 15 % of its time in free() */
typedef struct
{
	char	  **chunks;
	int			nchunks,
				cur;
	size_t		used;
	void	  **big;			/* requests larger than a chunk */
	int			nbig,
				bigcap;
}			arena;

#define ARENA_CHUNK ((size_t) 1 << 20)

static void *
arena_alloc(arena * a, size_t bytes)
{
	void	   *p;

	bytes = (bytes + 15) & ~(size_t) 15;
	if (bytes > ARENA_CHUNK)
	{
		if (a->nbig == a->bigcap)
		{
			void	  **grown = realloc(a->big, sizeof(void *) * (size_t) (a->bigcap ? a->bigcap * 2 : 8));

			if (!grown)
				return NULL;
			a->big = grown;
			a->bigcap = a->bigcap ? a->bigcap * 2 : 8;
		}
		p = malloc(bytes);
		if (p && a->big)
			a->big[a->nbig++] = p;
		else
		{
			free(p);
			p = NULL;
		}
		return p;
	}
	if (a->nchunks == 0 || a->used + bytes > ARENA_CHUNK)
	{
		if (a->nchunks > 0 && a->cur + 1 < a->nchunks)
			a->cur++;
		else
		{
			char	  **grown = realloc(a->chunks, sizeof(char *) * (size_t) (a->nchunks + 1));
			char	   *chunk = malloc(ARENA_CHUNK);

			if (grown)
				a->chunks = grown;
			if (!grown || !chunk)
			{
				free(chunk);
				return NULL;
			}
			a->chunks[a->nchunks] = chunk;
			a->cur = a->nchunks++;
		}
		a->used = 0;
	}
	p = a->chunks[a->cur] + a->used;
	a->used += bytes;
	return p;
}

static void
arena_reset(arena * a)
{
	for (int i = 0; i < a->nbig; i++)
		free(a->big[i]);
	a->nbig = 0;
	a->cur = 0;
	a->used = 0;
}

static void
arena_free(arena * a)
{
	arena_reset(a);
	for (int i = 0; i < a->nchunks; i++)
		free(a->chunks[i]);
	free(a->chunks);
	free(a->big);
	memset(a, 0, sizeof(*a));
}

/* a batch element choosing a neighbor: the owner's list (a record) meets the element as a newcomer */
typedef struct
{
	int32_t		rec;
	int32_t		element;
	float		distance;
}			link_req;

/* one thread's share of step 4a: the lists whose owners are dealt to it */
typedef struct
{
	record	   *recs;
	int			nrec,
				cap;
	link_req   *links;
	int64_t		nlinks,
				links_cap;
	int32_t    *tab;			/* hash: (owner, lc) -> own record index + 1 */
	int64_t		tab_cap;
	int32_t    *dirty;			/* owners whose tuples this batch rewrites, first seen by this thread */
	int			ndirty,
				dirty_cap;
	int			oom;
	int64_t		rec_base,		/* where its records / requests start in the batch's arrays */
				link_base;
}			recpart;

static double
now_secs(void)
{
	struct timespec ts;

	clock_gettime(CLOCK_MONOTONIC, &ts);
	return (double) ts.tv_sec + 1e-9 * (double) ts.tv_nsec;
}

/* where the wall time of a build goes: out->phase_secs[] */
enum
{
	PH_SEARCH, PH_PAIRS, PH_SELECT, PH_RECORDS, PH_UPDATE, PH_PATCH, PH_PAIRLIST, PH_FREE, PH_COUNT
};
#define PHASE(p) do { double t_ = now_secs(); out->phase_secs[cur_phase] += t_ - phase_t0; phase_t0 = t_; cur_phase = (p); } while (0)

static inline int
layer_m(int m, int lc)
{
	return lc == 0 ? 2 * m : m; /* HnswGetLayerM, src/hnsw.h:127 */
}

/* ------------------------------------------------------------ pair requests */

/* a request to pgv_hnsw_score_groups: lists of element slots and, per list, which of its pairs are wanted */
typedef struct
{
	int32_t    *ids;
	int64_t    *ids_start;		/* [ngroups + 1] */
	int32_t    *from;			/* [ngroups] */
	int64_t    *pair_start;		/* [ngroups + 1] */
	int64_t		nids,
				ids_cap;
	int			ngroups,
				groups_cap;
}			groupbuf;

static void
groups_reset(groupbuf * gb)
{
	gb->nids = 0;
	gb->ngroups = 0;
}

/* pairs (u, v), v < u, of the locals u >= from of ids[0 .. n): they start at pair0 in the reply.  0 = out of memory */
static int
groups_add(groupbuf * gb, const int32_t *ids, int n, int from, int64_t pair0)
{
	if (gb->ngroups + 1 >= gb->groups_cap)
	{
		int			cap = gb->groups_cap ? gb->groups_cap * 2 : 4096;
		int64_t    *is = realloc(gb->ids_start, sizeof(int64_t) * (size_t) (cap + 1));
		int64_t    *ps = realloc(gb->pair_start, sizeof(int64_t) * (size_t) (cap + 1));
		int32_t    *fr = realloc(gb->from, sizeof(int32_t) * (size_t) cap);

		if (is)
			gb->ids_start = is;
		if (ps)
			gb->pair_start = ps;
		if (fr)
			gb->from = fr;
		if (!is || !ps || !fr)
			return 0;
		gb->groups_cap = cap;
	}
	if (gb->nids + n > gb->ids_cap)
	{
		int64_t		cap = gb->ids_cap ? gb->ids_cap : 65536;
		int32_t    *grown;

		while (cap < gb->nids + n)
			cap *= 2;
		grown = realloc(gb->ids, sizeof(int32_t) * (size_t) cap);
		if (!grown)
			return 0;
		gb->ids = grown;
		gb->ids_cap = cap;
	}
	memcpy(gb->ids + gb->nids, ids, sizeof(int32_t) * (size_t) n);
	gb->ids_start[gb->ngroups] = gb->nids;
	gb->from[gb->ngroups] = from;
	gb->pair_start[gb->ngroups] = pair0;
	gb->nids += n;
	gb->ngroups++;
	gb->ids_start[gb->ngroups] = gb->nids;
	return 1;
}

/* pairs a group (n locals, from) asks for */
static inline int64_t
group_pairs(int n, int from)
{
	if (from < 1)
		from = 1;
	return n > from ? ((int64_t) n * (n - 1) - (int64_t) from * (from - 1)) / 2 : 0;
}

/*
 * Where a selection finds the distance of two locals: in the reply of pgv_hnsw_score_groups, which holds a group's
 * pairs (u, v < u) for u >= from, u ascending, then v ascending -- no matrix is built.  mm: the member-member triangle
 * (u < from) of a second request, or NULL when those pairs were not fetched.
 */
typedef struct
{
	const float *tri;
	int			from;			/* >= 1 */
	int			base;			/* from * (from - 1) / 2 */
	const float *mm;
}			pairsrc;

static inline pairsrc
pairs_of(const float *tri, int from, const float *mm)
{
	pairsrc		ps;

	if (from < 1)
		from = 1;
	ps.tri = tri;
	ps.from = from;
	ps.base = from * (from - 1) / 2;
	ps.mm = mm;
	return ps;
}

/* ---------------------------------------------------------- SelectNeighbors */

/* CompareCandidateDistances, src/hnswutils.c:992-1010: descending distance, then descending
 * pointer; element slots grow with allocation order and stand in for pointers */
static inline int
cand_desc_cmp(const void *pa, const void *pb)
{
	const cand *a = *(cand * const *) pa,
			   *b = *(cand * const *) pb;

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

/* the same order as one integer: larger key = earlier in cand_desc_cmp's order (-0.0 counts as 0.0, as in a float
 * comparison; elements are >= 0) */
#define SORT_KEYS 208			/* > 2 * 100 + 1 (m <= 100, src/hnsw.h:56) */
static inline uint64_t
cand_key(const cand * x)
{
	float		d = x->distance + 0.0f;
	uint32_t	u;

	memcpy(&u, &d, sizeof(u));
	u ^= (u >> 31) ? 0xFFFFFFFFu : 0x80000000u;
	return ((uint64_t) u << 32) | (uint32_t) x->element;
}

/*
 * CheckElementCloser (src/hnswutils.c:1040-1059) of e against the candidates in set[0 .. n); *missing is set when a
 * pair was not fetched (both locals below ps->from, no member-member triangle).
 */
static int
check_closer(const cand * e, cand * *set, int n, const pairsrc * ps, int *missing)
{
	for (int i = 0; i < n; i++)
	{
		const int	a = e->local,
					b = set[i]->local;
		const int	hi = a > b ? a : b,
					lo = a > b ? b : a;
		float		d;

		if (hi >= ps->from)
			d = ps->tri[hi * (hi - 1) / 2 - ps->base + lo];
		else if (ps->mm)
			d = ps->mm[hi * (hi - 1) / 2 + lo];
		else
		{
			*missing = 1;
			return 0;
		}
		if (d <= e->distance)
			return 0;
	}
	return 1;
}

/*
 * Algorithm 4 with the reference's closer-flag cache (src/hnswutils.c:1064-1165).  Pairwise
 * distances come from ps, indexed by cand.local.  c is ordered furthest first unless sort != 0; with sort and ord,
 * c[ord[0 .. nc - 2]] is the sorted order of all but the last candidate.  w is left holding the sorted candidates.
 * Returns |r|, or -1 when a missing pair was needed (nothing has been changed then).
 * *pruned = the candidate that would be dropped.
 */
static int
select_neighbors(cand * *c, int nc, int lm, const pairsrc * ps, uint8_t *closer_set,
				 cand * new_cand, cand * *r, cand * *pruned, int sort, const uint8_t *ord, cand * *w, cand * *wd,
				 cand * *added, uint8_t *flag)
{
	int			wn = nc,
				rn = 0,
				wdlen = 0,
				wdoff = 0,
				nadded = 0;
	int			must_calculate = !(*closer_set);
	int			removed_any = 0;
	int			missing = 0;

	if (nc <= lm)
	{
		for (int i = 0; i < nc; i++)
			r[i] = c[i];
		return nc;
	}
	if (sort)
	{
		/* list_sort(w, CompareCandidateDistances): a total order, so any algorithm gives the reference's result.  A
		 * neighbor list is a few dozen entries: an insertion sort over one 64-bit key per candidate (the distance's
		 * bits made monotonic, then the element) -- this sort is most of a list update's cost. */
		uint64_t	key[SORT_KEYS];

		if (ord)
		{
			/* c[0 .. nc - 2] in sorted order is c[ord[.]] (the list as the last selection left it); c[nc - 1], the
			 * newcomer, is merged in */
			cand	   *nw = c[nc - 1];
			const uint64_t k = cand_key(nw);
			int			pos = 0,
						placed = 0;

			for (int i = 0; i < nc - 1; i++)
			{
				cand	   *x = c[ord[i]];

				if (!placed && cand_key(x) < k)
				{
					w[pos++] = nw;
					placed = 1;
				}
				w[pos++] = x;
			}
			if (!placed)
				w[pos++] = nw;
		}
		else if (nc <= SORT_KEYS)
			for (int i = 0; i < nc; i++)
			{
				cand	   *x = c[i];
				const uint64_t k = cand_key(x);
				int			j = i;

				while (j > 0 && key[j - 1] < k)
				{
					key[j] = key[j - 1];
					w[j] = w[j - 1];
					j--;
				}
				key[j] = k;
				w[j] = x;
			}
		else
			for (int i = 0; i < nc; i++)
			{
				cand	   *x = c[i];
				int			j = i;

				while (j > 0 && cand_desc_cmp(&x, &w[j - 1]) < 0)
				{
					w[j] = w[j - 1];
					j--;
				}
				w[j] = x;
			}
	}
	else
		memcpy(w, c, sizeof(*w) * (size_t) nc);
	while (wn > 0 && rn < lm)
	{
		cand	   *e = w[--wn];	/* closest remaining */
		uint8_t		closer = e->closer;

		/* use the previous state of r and wd to skip work when possible (:1098-1140) */
		if (must_calculate)
			closer = (uint8_t) check_closer(e, r, rn, ps, &missing);
		else if (nadded > 0)
		{
			if (closer)
			{
				closer = (uint8_t) check_closer(e, added, nadded, ps, &missing);
				if (!closer)
					removed_any = 1;
			}
			else if (removed_any)
			{
				closer = (uint8_t) check_closer(e, r, rn, ps, &missing);
				if (closer)
					added[nadded++] = e;
			}
		}
		else if (e == new_cand)
		{
			closer = (uint8_t) check_closer(e, r, rn, ps, &missing);
			if (closer)
				added[nadded++] = e;
		}
		if (missing)
			return -1;
		flag[wn] = closer;		/* committed below: a replay that runs out of distances must leave no trace */
		if (closer)
			r[rn++] = e;
		else
			wd[wdlen++] = e;
	}
	for (int i = wn; i < nc; i++)
		w[i]->closer = flag[i];
	/* cached values can only be used in future if sorted deterministically (:1143-1144) */
	*closer_set = (uint8_t) (sort != 0);
	/* keep pruned connections (:1146-1148) */
	while (wdoff < wdlen && rn < lm)
		r[rn++] = wd[wdoff++];
	if (pruned)
		*pruned = wdoff < wdlen ? wd[wdoff] : w[0];	/* :1150-1157 */
	return rn;
}

/* ------------------------------------------------------------------- build */

void
pgv_host_hnsw_built_free(pgv_hnsw_built * b)
{
	if (!b)
		return;
	free(b->levels);
	free(b->nbr_start);
	free(b->nbr);
	free(b->dup_of);
	memset(b, 0, sizeof(*b));
}

/* rewrite element e's neighbor tuple (HnswNeighborTupleData: layer lc at (level - lc) * m) in the flat image */
static void
write_tuple(const elem * el, int32_t e, int m, const int64_t *nbr_start, int32_t *nbr)
{
	const elem *x = &el[e];
	int64_t		base = nbr_start[e];

	for (int64_t j = base; j < nbr_start[e + 1]; j++)
		nbr[j] = -1;
	if (!x->layers)
		return;
	for (int lc = 0; lc <= x->level; lc++)
	{
		const nlist *l = &x->layers[lc];
		int64_t		o = base + (int64_t) (x->level - lc) * m;

		for (int i = 0; i < l->length; i++)
			nbr[o + i] = l->items[i].element;
	}
}

/* one layer's part of that tuple, from the list as it stands (the replay writes it while the list is in its cache) */
static inline void
write_segment(const elem * x, int32_t e, int lc, int m, const int64_t *nbr_start, int32_t *nbr)
{
	const nlist *l = &x->layers[lc];
	const int	lm = layer_m(m, lc);
	int32_t    *o = nbr + nbr_start[e] + (int64_t) (x->level - lc) * m;
	int			i = 0;

	for (; i < l->length; i++)
		o[i] = l->items[i].element;
	for (; i < lm; i++)
		o[i] = -1;
}

/* ------------------------------------------------------------ stage A of a batch
 * What a batch needs from the device BEFORE any of its host work -- the searches of HnswFindElementNeighbors for all of
 * its elements and the pairwise distances inside every candidate list that has to be thinned -- depends on the graph as
 * the device holds it and on nothing the host replay of the PREVIOUS batch produces.  So it runs one batch ahead, on a
 * helper thread with a context and a view (pgv_hnsw_share) of its own, while the main thread replays the current batch
 * (SelectNeighbors, the lists' records and updates): the searches and the pair scoring, 5.1 of the 12 s of a 1 M x 1536
 * build, leave the critical path.  The batch that runs ahead searches the graph as it stood before the current batch
 * was linked -- the same blindness the elements of one batch have towards each other, one batch longer; the
 * reference's own parallel workers search while others link (src/hnswbuild.c:366-480).  Only batches of the full
 * max_batch run ahead (their composition then does not depend on how many duplicates the current one finds), never
 * across a change of the entry point, and never with max_batch < 64: max_batch = 1 stays the reference's serial loop.
 */
typedef struct
{
	/* the batch */
	int64_t		i0;
	int			B,
				lcap;
	/* what stage A yields */
	int32_t    *sw_ids;
	float	   *sw_dist;
	int32_t    *sw_cnt;
	int64_t    *tri_off;
	/* ... or, with SelectNeighbors on the device (pgv_hnsw_build_neighbors): each searched layer's neighbor list as
	 * AddConnections stores it, [B x lcap x 2m] */
	int			device_select;
	int32_t    *sel_ids;
	float	   *sel_dist;
	uint8_t    *sel_closer;
	int32_t    *sel_cnt;
	groupbuf	gb;
	float	   *pdist;			/* pinned */
	int64_t		pdist_cap;
	int64_t		npairs;
	int			rc;
	char		err[200];
	double		secs[2];		/* search, pairs */
	uint64_t	seq;			/* the helper's job that computes it (run ahead) */
	/* the device-link build's two-step form: searches kept on the device in `slot`, the selection a job of its own */
	int			slot;
	int			s_posted,
				t_posted;
	uint64_t	seq_s,
				seq_t;
}			stage_a;

static void
stage_a_free(stage_a * a)
{
	free(a->sw_ids);
	free(a->sw_dist);
	free(a->sw_cnt);
	free(a->tri_off);
	free(a->sel_ids);
	free(a->sel_dist);
	free(a->sel_closer);
	free(a->sel_cnt);
	free(a->gb.ids);
	free(a->gb.ids_start);
	free(a->gb.from);
	free(a->gb.pair_start);
	pgv_pinned_free(a->pdist);
	memset(a, 0, sizeof(*a));
}

/* steps 1 and 2 of a batch on `handle` (the mirror itself, or the helper's view of it) */
static int
run_stage_a(pgv_hnsw * handle, stage_a * a, const elem * el, int m, int ef_construction)
{
	const int	B = a->B,
				lcap = a->lcap;
	size_t		per = (size_t) B * lcap;
	int32_t    *ids = malloc(sizeof(int32_t) * (size_t) B);
	int32_t    *lv = malloc(sizeof(int32_t) * (size_t) B);
	int64_t		total = 0;
	double		t0 = now_secs();
	int			rc;

	a->rc = PGV_OK;
	a->err[0] = 0;
	if (a->device_select)
	{
		/* searches + SelectNeighbors in one call: what comes back is the lists themselves */
		const size_t stride = 2 * (size_t) m;
		int64_t		pairs = 0;

		a->sel_ids = realloc(a->sel_ids, sizeof(int32_t) * per * stride);
		a->sel_dist = realloc(a->sel_dist, sizeof(float) * per * stride);
		a->sel_closer = realloc(a->sel_closer, per * stride);
		a->sel_cnt = realloc(a->sel_cnt, sizeof(int32_t) * per);
		if (!ids || !lv || !a->sel_ids || !a->sel_dist || !a->sel_closer || !a->sel_cnt)
		{
			free(ids);
			free(lv);
			snprintf(a->err, sizeof(a->err), "out of memory");
			return a->rc = PGV_ERR_NOMEM;
		}
		for (int b = 0; b < B; b++)
		{
			ids[b] = (int32_t) (a->i0 + b);
			lv[b] = el[a->i0 + b].level;
		}
		rc = pgv_hnsw_build_neighbors(handle, ids, lv, B, ef_construction, lcap, a->sel_ids, a->sel_dist, a->sel_closer,
									  a->sel_cnt, &pairs);
		free(ids);
		free(lv);
		a->npairs = pairs;
		a->secs[0] = now_secs() - t0;
		a->secs[1] = 0;
		if (rc != PGV_OK)
			goto dev_fail;
		return PGV_OK;
	}
	a->sw_ids = realloc(a->sw_ids, sizeof(int32_t) * per * ef_construction);
	a->sw_dist = realloc(a->sw_dist, sizeof(float) * per * ef_construction);
	a->sw_cnt = realloc(a->sw_cnt, sizeof(int32_t) * per);
	a->tri_off = realloc(a->tri_off, sizeof(int64_t) * per);
	if (!ids || !lv || !a->sw_ids || !a->sw_dist || !a->sw_cnt || !a->tri_off)
	{
		free(ids);
		free(lv);
		snprintf(a->err, sizeof(a->err), "out of memory");
		return a->rc = PGV_ERR_NOMEM;
	}
	/* ---- 1. the searches of HnswFindElementNeighbors for the whole batch */
	for (int b = 0; b < B; b++)
	{
		ids[b] = (int32_t) (a->i0 + b);
		lv[b] = el[a->i0 + b].level;
	}
	rc = pgv_hnsw_build_search(handle, ids, lv, B, ef_construction, lcap, a->sw_ids, a->sw_dist, a->sw_cnt);
	free(ids);
	free(lv);
	a->secs[0] = now_secs() - t0;
	t0 = now_secs();
	if (rc != PGV_OK)
		goto dev_fail;
	/* ---- 2. pairwise distances inside every candidate list that has to be thinned */
	groups_reset(&a->gb);
	for (int b = 0; b < B; b++)
		for (int lc = 0; lc < lcap; lc++)
		{
			int			nw = a->sw_cnt[(size_t) b * lcap + lc];

			a->tri_off[(size_t) b * lcap + lc] = total;
			if (nw > layer_m(m, lc))
			{
				/* the candidates as one group, the whole triangle: expanded on the device */
				if (!groups_add(&a->gb, a->sw_ids + ((size_t) b * lcap + lc) * ef_construction, nw, 1, total))
				{
					snprintf(a->err, sizeof(a->err), "out of memory");
					return a->rc = PGV_ERR_NOMEM;
				}
				total += group_pairs(nw, 1);
			}
		}
	a->npairs = total;
	if (a->gb.ngroups > 0)
		a->gb.pair_start[a->gb.ngroups] = total;
	if (total > a->pdist_cap)
	{
		pgv_pinned_free(a->pdist);
		a->pdist = NULL;
		a->pdist_cap = total * 2;
		if ((rc = pgv_pinned_alloc(sizeof(float) * (size_t) a->pdist_cap, (void **) &a->pdist)) != PGV_OK)
		{
			a->pdist_cap = 0;
			goto dev_fail;
		}
	}
	if (total > 0)
	{
		rc = pgv_hnsw_score_groups(handle, a->gb.ids, a->gb.ids_start, a->gb.from, a->gb.pair_start, a->gb.ngroups, a->gb.nids,
								   total, a->pdist);
		if (rc != PGV_OK)
			goto dev_fail;
	}
	a->secs[1] = now_secs() - t0;
	return PGV_OK;
dev_fail:
	snprintf(a->err, sizeof(a->err), "%s", pgv_last_error());
	return a->rc = rc;
}

/* stage A in two jobs (the device-link build): the searches, their candidate lists kept on the device ... */
static void
run_search_keep(pgv_hnsw * handle, stage_a * a, const elem * el, int ef_construction)
{
	const int	B = a->B;
	int32_t    *ids = malloc(sizeof(int32_t) * (size_t) B * 2);
	double		t0 = now_secs();

	a->rc = PGV_OK;
	a->err[0] = 0;
	if (!ids)
	{
		snprintf(a->err, sizeof(a->err), "out of memory");
		a->rc = PGV_ERR_NOMEM;
		return;
	}
	for (int b = 0; b < B; b++)
	{
		ids[b] = (int32_t) (a->i0 + b);
		ids[B + b] = el[a->i0 + b].level;
	}
	a->rc = pgv_hnsw_build_search_keep(handle, ids, ids + B, B, ef_construction, a->lcap, a->slot);
	if (a->rc != PGV_OK)
		snprintf(a->err, sizeof(a->err), "%s", pgv_last_error());
	free(ids);
	a->secs[0] = now_secs() - t0;
}

/* ... and SelectNeighbors over them: the lists as AddConnections stores them come back */
static void
run_select_kept(pgv_hnsw * handle, stage_a * a, int m)
{
	const size_t per = (size_t) a->B * a->lcap,
				stride = 2 * (size_t) m;
	int64_t		pairs = 0;
	double		t0 = now_secs();

	if (a->rc != PGV_OK)		/* (the searches failed: nothing to select from) */
		return;
	a->sel_ids = realloc(a->sel_ids, sizeof(int32_t) * per * stride);
	a->sel_dist = realloc(a->sel_dist, sizeof(float) * per * stride);
	a->sel_closer = realloc(a->sel_closer, per * stride);
	a->sel_cnt = realloc(a->sel_cnt, sizeof(int32_t) * per);
	if (!a->sel_ids || !a->sel_dist || !a->sel_closer || !a->sel_cnt)
	{
		snprintf(a->err, sizeof(a->err), "out of memory");
		a->rc = PGV_ERR_NOMEM;
		return;
	}
	a->rc = pgv_hnsw_build_select_kept(handle, a->slot, a->sel_ids, a->sel_dist, a->sel_closer, a->sel_cnt, &pairs);
	if (a->rc != PGV_OK)
		snprintf(a->err, sizeof(a->err), "%s", pgv_last_error());
	a->npairs = pairs;
	a->secs[1] = now_secs() - t0;
}

/* a graph patch handed to the helper (step 6): its arrays stay untouched until the job is over */
typedef struct
{
	int32_t		entry;
	int32_t    *dirty;
	int64_t    *packed_off;
	int32_t    *packed;
	int			ndirty;
	int64_t		dirty_cap,
				packed_cap;
	uint64_t	seq;			/* the job that last used the arrays */
	int			rc;
	char		err[200];
}			patch_job;

/* a slice of the batch's list records whose pair distances are wanted (step 4): offsets relative to `out` */
typedef struct
{
	groupbuf	gb;
	int64_t		npairs;
	float	   *out;
	int			klo,
				khi;			/* records [klo, khi) */
	uint64_t	seq;
	int			rc;
	char		err[200];
}			score_job;

typedef enum
{
	JOB_STAGE_A, JOB_PATCH, JOB_SCORE, JOB_SEARCH_KEEP, JOB_SELECT_KEPT
}			job_kind;

/*
 * A helper thread with a context, a stream and a view of the mirror of its own (pgv_hnsw_share): jobs run one after
 * the other in the order they were posted, so a patch posted after the stage A that runs ahead cannot overtake it, and
 * a stage A posted after a patch searches the patched graph.
 */
#define WORKER_RING 8
#define SCORE_SLICES 4
typedef struct
{
	pthread_t	thread;
	pthread_mutex_t lock;
	pthread_cond_t wake;
	pgv_ctx    *ctx;
	pgv_hnsw   *view;
	const elem *el;
	int			m,
				ef_construction;
	struct
	{
		job_kind	kind;
		void	   *arg;
	}			ring[WORKER_RING];
	uint64_t	posted,
				completed;
	int			quit;
	int			started;
}			worker;

static void *
worker_main(void *arg)
{
	worker	   *w = arg;

	pthread_mutex_lock(&w->lock);
	for (;;)
	{
		job_kind	kind;
		void	   *ja;

		while (!w->quit && w->completed == w->posted)
			pthread_cond_wait(&w->wake, &w->lock);
		if (w->completed == w->posted)	/* quit, nothing pending */
			break;
		kind = w->ring[w->completed % WORKER_RING].kind;
		ja = w->ring[w->completed % WORKER_RING].arg;
		pthread_mutex_unlock(&w->lock);
		if (kind == JOB_STAGE_A)
			run_stage_a(w->view, ja, w->el, w->m, w->ef_construction);
		else if (kind == JOB_SEARCH_KEEP)
			run_search_keep(w->view, ja, w->el, w->ef_construction);
		else if (kind == JOB_SELECT_KEPT)
			run_select_kept(w->view, ja, w->m);
		else if (kind == JOB_PATCH)
		{
			patch_job  *pj = ja;

			pj->rc = pgv_hnsw_update_graph(w->view, pj->entry, pj->dirty, pj->ndirty, pj->packed_off, pj->packed);
			if (pj->rc != PGV_OK)
				snprintf(pj->err, sizeof(pj->err), "%s", pgv_last_error());
		}
		else
		{
			score_job  *sj = ja;

			sj->rc = sj->npairs > 0 ? pgv_hnsw_score_groups(w->view, sj->gb.ids, sj->gb.ids_start, sj->gb.from, sj->gb.pair_start,
															  sj->gb.ngroups, sj->gb.nids, sj->npairs, sj->out) : PGV_OK;
			if (sj->rc != PGV_OK)
				snprintf(sj->err, sizeof(sj->err), "%s", pgv_last_error());
		}
		pthread_mutex_lock(&w->lock);
		w->completed++;
		pthread_cond_broadcast(&w->wake);
	}
	pthread_mutex_unlock(&w->lock);
	return NULL;
}

/*
 * Inside a PostgreSQL backend (ext/hnswbuild_gpu.c calls this file from FlushPages) two things are not this code's to
 * decide: which thread runs the backend's signal handlers, and whether the statement may be cancelled.
 *   - every thread made here (the two helpers, the OpenMP team) starts with ALL signals blocked: SIGINT / SIGTERM /
 *     SIGUSR1 stay with the backend's main thread, where its handlers expect to run;
 *   - pgv_host_hnsw_set_cancel_check installs a callback the build polls between batches (on the calling thread, never
 *     from a helper): nonzero ends the build cleanly with PGV_ERR_STATE "cancelled" -- helpers joined, memory freed --
 *     and the caller raises its own error (the glue: CHECK_FOR_INTERRUPTS()).  The callback must not longjmp.
 */
static __thread pgv_host_cancel_check cancel_check = NULL;
static __thread void *cancel_arg = NULL;

void
pgv_host_hnsw_set_cancel_check(pgv_host_cancel_check check, void *arg)
{
	cancel_check = check;
	cancel_arg = arg;
}

static void
block_signals(sigset_t *old)
{
	sigset_t	all;

	sigfillset(&all);
	pthread_sigmask(SIG_BLOCK, &all, old);
}

/* a context, a stream and a view of the mirror for a helper; 0 when it could not be had (the build goes on without) */
static int
worker_start(worker * w, pgv_hnsw * mirror, const elem * el, int m, int ef_construction)
{
	if (w->started)
		return 1;
	if (pgv_ctx_create(pgv_hnsw_device(mirror), NULL, &w->ctx) == PGV_OK && pgv_hnsw_share(mirror, w->ctx, &w->view) == PGV_OK)
	{
		w->el = el;
		w->m = m;
		w->ef_construction = ef_construction;
		pthread_mutex_init(&w->lock, NULL);
		pthread_cond_init(&w->wake, NULL);
		{
			sigset_t	old;
			int			err;

			block_signals(&old);	/* the helper inherits the mask: no handler of the backend's ever runs there */
			err = pthread_create(&w->thread, NULL, worker_main, w);
			pthread_sigmask(SIG_SETMASK, &old, NULL);
			if (err == 0)
				return w->started = 1;
		}
		pthread_mutex_destroy(&w->lock);
		pthread_cond_destroy(&w->wake);
	}
	if (w->view)
		pgv_hnsw_free(w->view);
	if (w->ctx)
		pgv_ctx_destroy(w->ctx);
	w->view = NULL;
	w->ctx = NULL;
	return 0;
}

/* the job's number: worker_wait(w, that) returns once it is over */
static uint64_t
worker_post(worker * w, job_kind kind, void *arg)
{
	uint64_t	seq;

	pthread_mutex_lock(&w->lock);
	while (w->posted - w->completed >= WORKER_RING)
		pthread_cond_wait(&w->wake, &w->lock);
	w->ring[w->posted % WORKER_RING].kind = kind;
	w->ring[w->posted % WORKER_RING].arg = arg;
	seq = ++w->posted;
	pthread_cond_broadcast(&w->wake);
	pthread_mutex_unlock(&w->lock);
	return seq;
}

static void
worker_wait(worker * w, uint64_t seq)
{
	if (!w->started)
		return;
	pthread_mutex_lock(&w->lock);
	while (w->completed < seq)
		pthread_cond_wait(&w->wake, &w->lock);
	pthread_mutex_unlock(&w->lock);
}

static void
worker_drain(worker * w)
{
	if (w->started)
		worker_wait(w, w->posted);	/* (posted is only written by this, the posting, thread) */
}

static void
worker_stop(worker * w)
{
	if (!w->started)
		return;
	pthread_mutex_lock(&w->lock);
	w->quit = 1;
	pthread_cond_broadcast(&w->wake);
	pthread_mutex_unlock(&w->lock);
	pthread_join(w->thread, NULL);	/* pending jobs are run first */
	if (w->view)
		pgv_hnsw_free(w->view);
	if (w->ctx)
		pgv_ctx_destroy(w->ctx);
	pthread_mutex_destroy(&w->lock);
	pthread_cond_destroy(&w->wake);
	w->started = 0;
}

/*
 * The batches with the graph updates ON THE DEVICE (pgv_hnsw_link_*): the host keeps the order of things and
 * FindDuplicateInMemory; SelectNeighbors for the new elements runs on the device behind their searches, the lists they
 * chose are replayed by GPU wavefronts (csrc/hnsw_link_core.h: the reference's HnswUpdateConnection), and the tuples the
 * next batch searches are rewritten in place -- no list state, no pair distance and no neighbor tuple crosses PCIe until
 * the finished graph comes back.
 *
 * Three streams share the GPU once batches are full (each a context of its own; S and T are helper threads):
 *   S  searches of batch n + 2            (pgv_hnsw_build_search_keep: posted when batch n has been applied, so they see
 *                                          the graph without batch n + 1 -- the blindness of the host-side form)
 *   T  SelectNeighbors of batch n + 1     (pgv_hnsw_build_select_kept, over the lists S kept on the device)
 *   main  duplicates of batch n, pgv_hnsw_link_prepare (n) -- requests grouped by list, the lists' pair distances --,
 *         then, once the searches that read the old tuples are over (S of n + 1), pgv_hnsw_link_apply (n)
 * The searches are the longest step and run back to back but for the apply in between.  Batches that are not full, or
 * that move the entry point, are done in turn (pgv_hnsw_build_neighbors on the mirror itself); max_batch = 1 is the
 * reference's serial loop.
 */
typedef struct
{
	int64_t		i0;
	int			B,
				lcap;
}			batch_plan;

/* the batch that starts at i0 when `linked` elements are in the graph and the entry point stands at entry_level */
static batch_plan
plan_batch(const elem * el, int64_t n, int64_t i0, int64_t linked, int entry_level, int max_batch)
{
	batch_plan	p;
	int			B = (int) (linked / 16);

	if (B < 1)
		B = 1;
	if (B > max_batch)
		B = max_batch;
	if (B > n - i0)
		B = (int) (n - i0);
	/* the batch ends at (and includes) the first element taller than the entry point (src/hnswbuild.c:398-431) */
	for (int b = 0; b < B; b++)
		if (el[i0 + b].level > entry_level)
		{
			B = b + 1;
			break;
		}
	p.i0 = i0;
	p.B = B;
	p.lcap = 1;
	for (int b = 0; b < B; b++)
	{
		int			l = el[i0 + b].level < entry_level ? el[i0 + b].level : entry_level;

		if (l + 1 > p.lcap)
			p.lcap = l + 1;
	}
	return p;
}

static int
batch_has_tall(const elem * el, const batch_plan * p, int entry_level)
{
	for (int b = 0; b < p->B; b++)
		if (el[p->i0 + b].level > entry_level)
			return 1;
	return 0;
}

static int
build_linked_on_device(pgv_hnsw * mirror, size_t item_bytes, const void *rows, int64_t n, int m, int ef_construction,
					   int max_batch, elem * el, pgv_hnsw_built * out, int32_t *entry_io, int64_t *linked_io)
{
	stage_a		stages[3];		/* batch number % 3 */
	worker		srch,
				sel;
	int64_t		seqno = 0;		/* batches begun (the first element aside) */
	int32_t		entry = -1;
	int64_t		linked = 0;
	int32_t    *ids = malloc(sizeof(int32_t) * (size_t) max_batch);
	uint8_t    *is_linked = malloc((size_t) max_batch);
	int			rc = PGV_OK;
	double		phase_t0 = now_secs();
	int			cur_phase = PH_RECORDS;
	const int	pipelined = max_batch >= 64;

	memset(stages, 0, sizeof(stages));
	for (int i = 0; i < 3; i++)
		stages[i].device_select = 1;
	memset(&srch, 0, sizeof(srch));
	memset(&sel, 0, sizeof(sel));
	if (!ids || !is_linked)
	{
		rc = pgv_host_fail(PGV_ERR_NOMEM, "out of memory");
		goto done;
	}
	for (int64_t i0 = 0; i0 < n;)
	{
		batch_plan	p;
		int			entry_level;
		int64_t		pairs = 0;
		stage_a    *a,
				   *nx;

		if (cancel_check != NULL && cancel_check(cancel_arg))
		{
			rc = pgv_host_fail(PGV_ERR_STATE, "pgv_host_hnsw_build: cancelled after %lld of %lld rows", (long long) i0, (long long) n);
			goto done;
		}
		if (entry < 0)
		{
			/* the first element has nothing to search: it becomes the entry point */
			entry = (int32_t) i0;
			linked = 1;
			if ((rc = pgv_hnsw_update_graph(mirror, entry, NULL, 0, NULL, NULL)) != PGV_OK)
				goto dev_fail;
			i0++;
			out->batches++;
			continue;
		}
		entry_level = el[entry].level;
		p = plan_batch(el, n, i0, linked, entry_level, max_batch);
		a = &stages[seqno % 3];
		nx = &stages[(seqno + 1) % 3];

		PHASE(PH_SEARCH);
		if (a->s_posted && a->i0 == p.i0 && a->B == p.B && a->lcap == p.lcap)
		{
			/* its searches ran ahead; so did its selection unless the batch before was the first of the pipeline */
			worker_wait(&srch, a->seq_s);
			if (!a->t_posted)
			{
				a->seq_t = worker_post(&sel, JOB_SELECT_KEPT, a);
				a->t_posted = 1;
			}
			worker_wait(&sel, a->seq_t);
		}
		else
		{
			/* in turn, on the mirror itself (nothing runs ahead by construction when the plan does not match) */
			worker_drain(&srch);
			worker_drain(&sel);
			for (int i = 0; i < 3; i++)
				stages[i].s_posted = stages[i].t_posted = 0;
			a->i0 = p.i0;
			a->B = p.B;
			a->lcap = p.lcap;
			run_stage_a(mirror, a, el, m, ef_construction);
		}
		a->s_posted = a->t_posted = 0;
		if (a->rc != PGV_OK)
		{
			rc = pgv_host_fail(a->rc, "%s", a->err);
			goto done;
		}
		out->device_pairs += a->npairs;
		/* the searches of the NEXT batch, if they are not under way and its composition is certain: full batches from
		 * here on, no change of the entry point in this one */
		if (pipelined && !nx->s_posted && p.B == max_batch && linked / 16 >= max_batch && p.i0 + p.B < n &&
			!batch_has_tall(el, &p, entry_level))
		{
			batch_plan	q = plan_batch(el, n, p.i0 + p.B, linked, entry_level, max_batch);

			if (!srch.started && worker_start(&srch, mirror, el, m, ef_construction))
				worker_start(&sel, mirror, el, m, ef_construction);
			if (srch.started && sel.started)
			{
				nx->i0 = q.i0;
				nx->B = q.B;
				nx->lcap = q.lcap;
				nx->slot = (int) ((seqno + 1) & 1);
				nx->seq_s = worker_post(&srch, JOB_SEARCH_KEEP, nx);
				nx->s_posted = 1;
				nx->t_posted = 0;
			}
		}

		PHASE(PH_SELECT);
		/* FindDuplicateInMemory, src/hnswbuild.c:313-364, in heap order: an element's layer-0 neighbors are ordered by
		 * distance, the identical ones first */
		for (int b = 0; b < p.B; b++)
		{
			int32_t		e = (int32_t) (i0 + b);
			const size_t g = (size_t) b * p.lcap;
			const char *v = (const char *) rows + (size_t) e * item_bytes;

			ids[b] = e;
			is_linked[b] = 1;
			for (int i = 0; i < a->sel_cnt[g]; i++)
			{
				int32_t		ne = a->sel_ids[g * 2 * (size_t) m + i];

				if (memcmp(v, (const char *) rows + (size_t) ne * item_bytes, item_bytes) != 0)
					break;
				if (el[ne].heaptids < HNSW_HEAPTIDS)
				{
					el[ne].heaptids++;
					out->dup_of[e] = ne;
					is_linked[b] = 0;
					break;
				}
			}
			if (is_linked[b])
			{
				linked++;
				/* the entry point moves up with the tallest element (src/hnswbuild.c:425-430) */
				if (el[e].level > el[entry].level)
					entry = e;
			}
		}

		PHASE(PH_RECORDS);
		if ((rc = pgv_hnsw_link_prepare(mirror, ids, is_linked, p.B, p.lcap, a->sel_ids, a->sel_dist, a->sel_closer, a->sel_cnt,
										&pairs)) != PGV_OK)
			goto dev_fail;
		PHASE(PH_PAIRS);
		/* the searches that run ahead read the tuples this batch is about to rewrite: they have to be over; their
		 * selection does not touch the graph and runs on */
		if (nx->s_posted)
		{
			worker_wait(&srch, nx->seq_s);
			if (!nx->t_posted)
			{
				nx->seq_t = worker_post(&sel, JOB_SELECT_KEPT, nx);
				nx->t_posted = 1;
			}
		}
		PHASE(PH_UPDATE);
		/* (returns with everything enqueued: the searches posted below start on the device when it is through) */
		if ((rc = pgv_hnsw_link_apply(mirror, entry)) != PGV_OK)
			goto dev_fail;
		out->device_pairs += pairs;
		PHASE(PH_RECORDS);
		/* the searches of the batch after the next: they see the graph with this batch in it, without the next (whose
		 * composition has to be certain, full and without a change of the entry point, for the one behind it to be) */
		if (nx->s_posted && nx->B == max_batch && !batch_has_tall(el, &(batch_plan){nx->i0, nx->B, nx->lcap}, el[entry].level) &&
			linked / 16 >= max_batch && nx->i0 + nx->B < n && entry_level == el[entry].level)
		{
			stage_a    *n2 = &stages[(seqno + 2) % 3];
			batch_plan	q = plan_batch(el, n, nx->i0 + nx->B, linked, entry_level, max_batch);

			n2->i0 = q.i0;
			n2->B = q.B;
			n2->lcap = q.lcap;
			n2->slot = (int) ((seqno + 2) & 1);
			n2->seq_s = worker_post(&srch, JOB_SEARCH_KEEP, n2);
			n2->s_posted = 1;
			n2->t_posted = 0;
		}
		i0 += p.B;
		seqno++;
		out->batches++;
	}
	worker_drain(&srch);
	worker_drain(&sel);
	PHASE(PH_PATCH);
	{
		int64_t		pairs2 = 0,
					deferred = 0;

		if ((rc = pgv_hnsw_link_end(mirror, out->nbr, &pairs2, &deferred)) != PGV_OK)
			goto dev_fail;
		out->device_pairs += pairs2;
		out->deferred_updates += deferred;
	}
	PHASE(PH_RECORDS);
	goto done;

dev_fail:
	rc = pgv_host_fail(rc, "%s", pgv_last_error());
done:
	worker_stop(&srch);
	worker_stop(&sel);
	if (rc != PGV_OK)
		pgv_hnsw_link_end(mirror, NULL, NULL, NULL);
	for (int i = 0; i < 3; i++)
		stage_a_free(&stages[i]);
	free(ids);
	free(is_linked);
	*entry_io = entry;
	*linked_io = linked;
	return rc;
}

int
pgv_host_hnsw_build(pgv_hnsw * mirror, pgv_dtype dtype, int dim, const void *rows, int64_t n, int m,
					int ef_construction, const pgv_rng * rng, int max_batch, pgv_hnsw_built * out)
{
	const size_t item_bytes = (size_t) dim * (dtype == PGV_F32 ? 4 : 2);
	/* HnswGetMaxLevel, src/hnsw.h:133 with BLCKSZ 8192 */
	const int	by_page = (int) ((8192 - 24 - 8 - 4 - 4) / 6 / m) - 2;
	const int	max_level = by_page < 63 ? by_page : 63;
	const double ml = 1.0 / log((double) m);	/* HnswGetMl */
	rng_state	rs;
	elem	   *el = NULL;
	int32_t		entry = -1;
	int64_t		linked = 0;
	int			rc = PGV_OK;
	groupbuf	gb = {0};
	int64_t		npairs = 0;
	float	   *pdist = NULL,	/* the list records' pairs (first request) */
			   *pdist2 = NULL;	/* the member-member pairs of the lists whose cached flags did not suffice (second request) */
	int64_t		pdist_cap = 0,
				pdist2_cap = 0;
	int32_t    *sw_ids = NULL,	/* (aliases of the current batch's stage A) */
			   *sw_cnt = NULL;
	float	   *sw_dist = NULL;
	const float *cdist = NULL;	/* the candidate lists' pair distances of the current batch (stage A's) */
	stage_a		stages[2];
	int			cur = 0;		/* stages[cur]: the current batch's; stages[cur ^ 1]: the one running ahead */
	int			ahead_valid = 0;	/* stages[cur] was computed ahead for exactly this batch */
	worker		helper,			/* stage A of the batch that runs ahead and the graph patches, in posting order */
				scorer;			/* the pair distances of the batch's list records, slice by slice */
	patch_job	patches[2];
	int			pcur = 0;
	score_job	slices[SCORE_SLICES];
	int			nslices = 1;
	uint8_t    *is_dirty = NULL;
	record	   *recs = NULL;
	int			recs_cap = 0;
	recpart    *parts = NULL;	/* [nthreads] step 4a's per-thread records and requests */
	int64_t		links_cap = 0;
	int32_t    *grp_elem = NULL;
	float	   *grp_dist = NULL;
	int64_t    *grp_off = NULL;
	arena	   *arenas = NULL;	/* [nthreads] */
	int			oom = 0;
	int64_t    *tri_off = NULL;
	int			nrec = 0;		/* records of the batch in flight (freed at its end, or on the way out) */
	int			nthreads = omp_get_max_threads() < 16 ? omp_get_max_threads() : 16;	/* 32: twice as slow (measured) */
	double		phase_t0 = now_secs();
	int			cur_phase = PH_RECORDS;
	/* SelectNeighbors of the new elements' candidate lists: on the device with the searches (the default), or here from
	 * the lists and their pair triangles (PGV_HNSW_HOST_SELECT=1: the form the device's is tested against) */
	const int	device_select = !(getenv("PGV_HNSW_HOST_SELECT") && atoi(getenv("PGV_HNSW_HOST_SELECT")) != 0);
	/* the graph updates (HnswUpdateConnection for every list a batch links into): on the device (the default;
	 * build_linked_on_device), or replayed here on OpenMP threads (PGV_HNSW_HOST_LINK=1, and whenever the new elements'
	 * selection is the host's) -- the two build the same graph */
	const int	device_link = device_select && !(getenv("PGV_HNSW_HOST_LINK") && atoi(getenv("PGV_HNSW_HOST_LINK")) != 0);

	memset(stages, 0, sizeof(stages));
	stages[0].device_select = stages[1].device_select = device_select;
	memset(&helper, 0, sizeof(helper));
	memset(&scorer, 0, sizeof(scorer));
	memset(patches, 0, sizeof(patches));
	memset(slices, 0, sizeof(slices));
	if (!mirror || !out || (n > 0 && !rows))
		return pgv_host_fail(PGV_ERR_ARG, "pgv_host_hnsw_build: mirror/rows/out is NULL");
	if (m < 2 || m > 100 || ef_construction < 4 || ef_construction > 1000 || ef_construction < 2 * m)
		return pgv_host_fail(PGV_ERR_ARG, "m must be 2..100, ef_construction 4..1000 and >= 2 * m (src/hnsw.c:114-125)");
	if (n >= INT32_MAX)
		return pgv_host_fail(PGV_ERR_ARG, "too many elements");
	if (max_batch < 1)
		max_batch = 1;
	/* a list can receive every element of a batch as a newcomer (degenerate data: thousands of identical rows): the link
	 * replay's record locals are int16 and its pair triangle is indexed in int (csrc/hnsw_link_core.h) -- a batch never
	 * brings more than 16384 newcomers to one list */
	if (max_batch > 16384)
		max_batch = 16384;
	if (max_batch > 2048)
		max_batch = 2048;
	memset(out, 0, sizeof(*out));
	out->n = n;
	out->m = m;
	out->entry = -1;
	out->levels = malloc(sizeof(int32_t) * (size_t) (n > 0 ? n : 1));
	out->nbr_start = malloc(sizeof(int64_t) * (size_t) (n + 1));
	out->dup_of = malloc(sizeof(int32_t) * (size_t) (n > 0 ? n : 1));
	el = calloc((size_t) (n > 0 ? n : 1), sizeof(elem));
	if (!out->levels || !out->nbr_start || !out->dup_of || !el)
	{
		rc = pgv_host_fail(PGV_ERR_NOMEM, "out of memory");
		goto done;
	}

	/* HnswInitElement, src/hnswutils.c:243-270: one draw per heap tuple, in heap order */
	rng_init(&rs, rng);
	out->nbr_start[0] = 0;
	for (int64_t i = 0; i < n; i++)
	{
		int			level = (int) (-log(rng_double(&rs)) * ml);

		if (level > max_level)
			level = max_level;
		el[i].level = level;
		el[i].heaptids = 1;
		out->levels[i] = level;
		out->dup_of[i] = -1;
		out->nbr_start[i + 1] = out->nbr_start[i] + (int64_t) (level + 2) * m;
	}
	out->nbr = malloc(sizeof(int32_t) * (size_t) (out->nbr_start[n] > 0 ? out->nbr_start[n] : 1));
	if (!out->nbr)
	{
		rc = pgv_host_fail(PGV_ERR_NOMEM, "out of memory");
		goto done;
	}
	for (int64_t j = 0; j < out->nbr_start[n]; j++)
		out->nbr[j] = -1;
	if (n == 0)
		goto done;
	rc = pgv_hnsw_set_graph(mirror, m, -1, out->levels, out->nbr_start, out->nbr);
	if (rc != PGV_OK)
	{
		rc = pgv_host_fail(rc, "%s", pgv_last_error());
		goto done;
	}
	if (device_link)
	{
		/* the graph state beside the tuples (5 bytes a slot + a counter pair a list): when the device has no room for it
		 * the lists are replayed on the host instead -- distances still the GPU's, the same graph */
		rc = pgv_hnsw_link_begin(mirror);
		if (rc == PGV_OK)
		{
			PHASE(PH_RECORDS);
			rc = build_linked_on_device(mirror, item_bytes, rows, n, m, ef_construction, max_batch, el, out, &entry, &linked);
			goto done;
		}
		if (rc != PGV_ERR_NOMEM)
			goto dev_fail;
		rc = PGV_OK;
	}
	is_dirty = calloc((size_t) n, 1);
	parts = calloc((size_t) nthreads, sizeof(recpart));
	arenas = calloc((size_t) nthreads, sizeof(arena));
	{
		/* the OpenMP team is created by the first parallel region and keeps the mask of the thread that opened it:
		 * open one now, with every signal blocked, so that the pool's threads never take the backend's signals */
		sigset_t	old;
		int			team = 0;

		block_signals(&old);
#pragma omp parallel num_threads(nthreads) reduction(+:team)
		team += 1;
		pthread_sigmask(SIG_SETMASK, &old, NULL);
		(void) team;
	}

	for (int64_t i0 = 0; i0 < n;)
	{
		int			B;
		int			lcap = 1;
		int			entry_level;
		int			ndirty = 0;
		int64_t		ntuple = 0;

		/* CREATE INDEX can be cancelled between batches (a batch is milliseconds) */
		if (cancel_check != NULL && cancel_check(cancel_arg))
		{
			rc = pgv_host_fail(PGV_ERR_STATE, "pgv_host_hnsw_build: cancelled after %lld of %lld rows", (long long) i0, (long long) n);
			goto done;
		}
		if (entry < 0)
		{
			/* the first element has nothing to search: it becomes the entry point */
			el[i0].layers = calloc((size_t) el[i0].level + 1, sizeof(nlist));
			for (int lc = 0; lc <= el[i0].level; lc++)
				el[i0].layers[lc].items = items_alloc(layer_m(m, lc));
			entry = (int32_t) i0;
			linked = 1;
			rc = pgv_hnsw_update_graph(mirror, entry, NULL, 0, NULL, NULL);
			if (rc != PGV_OK)
				goto dev_fail;
			i0++;
			out->batches++;
			continue;
		}
		/* as many concurrent inserts as the graph can absorb without the newcomers missing each other much */
		B = (int) (linked / 16);
		if (B < 1)
			B = 1;
		if (B > max_batch)
			B = max_batch;
		if (B > n - i0)
			B = (int) (n - i0);
		entry_level = el[entry].level;
		/* An element taller than the entry point becomes the new entry under the reference's exclusive entry
		 * lock (src/hnswbuild.c:398-431): the next tall element links to it on the new top layers.  The batch
		 * therefore ends at (and includes) the first such element -- later ones search from the new entry. */
		for (int b = 0; b < B; b++)
			if (el[i0 + b].level > entry_level)
			{
				B = b + 1;
				break;
			}
		for (int b = 0; b < B; b++)
		{
			int			l = el[i0 + b].level < entry_level ? el[i0 + b].level : entry_level;

			if (l + 1 > lcap)
				lcap = l + 1;
		}

		PHASE(PH_SEARCH);
		/* ---- 1 + 2. stage A: the batch's searches and its candidate lists' pair distances -- computed ahead by the
		 * helper while the previous batch was replayed, or here */
		{
			stage_a    *a = &stages[cur];

			if (ahead_valid && a->i0 == i0 && a->B == B && a->lcap == lcap)
				worker_wait(&helper, a->seq);	/* (usually long over) */
			else
			{
				/* every patch handed to the helper has been issued (a batch that ran ahead for nothing cannot happen
				 * by construction); the device orders this search behind the patch */
				worker_drain(&helper);
				a->i0 = i0;
				a->B = B;
				a->lcap = lcap;
				run_stage_a(mirror, a, el, m, ef_construction);
			}
			ahead_valid = 0;
			if (a->rc != PGV_OK)
			{
				rc = pgv_host_fail(a->rc, "%s", a->err);
				goto done;
			}
			out->device_pairs += a->npairs;
			sw_ids = a->sw_ids;
			sw_dist = a->sw_dist;
			sw_cnt = a->sw_cnt;
			tri_off = a->tri_off;
			cdist = a->pdist;
			/* the NEXT batch runs ahead when its composition is already certain: a full batch follows a full batch
			 * (linked / 16 >= max_batch holds from here on), the entry point does not change in this one, and batches
			 * are large enough for the overlap to matter */
			if (max_batch >= 64 && B == max_batch && linked / 16 >= max_batch && i0 + B < n)
			{
				int			tall = 0;

				for (int b = 0; b < B; b++)
					tall |= el[i0 + b].level > entry_level;
				if (!tall)
				{
					stage_a    *nx = &stages[cur ^ 1];
					int64_t		n0 = i0 + B;
					int			nB = (int) (n - n0 < max_batch ? n - n0 : max_batch);
					int			nlcap = 1;

					for (int b = 0; b < nB; b++)
						if (el[n0 + b].level > entry_level)
						{
							nB = b + 1;
							break;
						}
					for (int b = 0; b < nB; b++)
					{
						int			l = el[n0 + b].level < entry_level ? el[n0 + b].level : entry_level;

						if (l + 1 > nlcap)
							nlcap = l + 1;
					}
					if (!helper.started && worker_start(&helper, mirror, el, m, ef_construction))
						worker_start(&scorer, mirror, el, m, ef_construction);
					if (helper.started)
					{
						nx->i0 = n0;
						nx->B = nB;
						nx->lcap = nlcap;
						/* posted at once: queued behind this batch's own scoring requests instead, the build was 1.7 s
						 * slower (measured, profiles/r04/hnsw_build_pipeline.md) */
						nx->seq = worker_post(&helper, JOB_STAGE_A, nx);
						ahead_valid = 1;
					}
				}
			}
		}

		PHASE(PH_SELECT);
		/* ---- 3. SelectNeighbors + AddConnections per element and layer (independent: one thread each) */
#pragma omp parallel if (B >= 8) num_threads(nthreads)
		{
			cand	   *lw = malloc(sizeof(cand) * (size_t) ef_construction);
			cand	  **c = malloc(sizeof(cand *) * (size_t) ef_construction * 5);
			cand	  **r = c + ef_construction,
					  **w = c + 2 * ef_construction,
					  **wd = c + 3 * ef_construction,
					  **added = c + 4 * ef_construction;
			uint8_t    *flag = malloc((size_t) ef_construction);

#pragma omp for schedule(dynamic, 4)
			for (int b = 0; b < B; b++)
			{
				int32_t		e = (int32_t) (i0 + b);
				elem	   *x = &el[e];

				x->layers = calloc((size_t) x->level + 1, sizeof(nlist));
				if (device_select)
				{
					/* SelectNeighbors ran on the device (pgv_hnsw_build_neighbors): the lists are AddConnections' to store */
					const stage_a *a = &stages[cur];

					for (int lc = lcap - 1; lc >= 0; lc--)
					{
						const size_t g = (size_t) b * lcap + lc;
						const int	rn = a->sel_cnt[g];
						const int	lm = layer_m(m, lc);

						if (rn == 0 || lc > x->level)
							continue;
						x->layers[lc].closer_set = 0;	/* not sorted deterministically (:1143-1144) */
						x->layers[lc].items = items_alloc(lm);
						x->layers[lc].length = rn;
						for (int i = 0; i < rn; i++)
						{
							cand	   *it = &x->layers[lc].items[i];

							it->element = a->sel_ids[g * 2 * (size_t) m + i];
							it->distance = a->sel_dist[g * 2 * (size_t) m + i];
							it->local = 0;
							it->closer = a->sel_closer[g * 2 * (size_t) m + i];
						}
					}
					for (int lc = 0; lc <= x->level; lc++)
						if (!x->layers[lc].items)
							x->layers[lc].items = items_alloc(layer_m(m, lc));
					continue;
				}
				for (int lc = lcap - 1; lc >= 0; lc--)
				{
					int			nw = sw_cnt[(size_t) b * lcap + lc];
					int			lm = layer_m(m, lc);
					const int32_t *wi = sw_ids + ((size_t) b * lcap + lc) * ef_construction;
					const float *wdist = sw_dist + ((size_t) b * lcap + lc) * ef_construction;
					int			rn;
					/* not sorted deterministically: the flags of this selection are not reusable (:1143-1144) */
					uint8_t		closer_set = 0;

					if (nw == 0 || lc > x->level)
						continue;
					/* the reference's list `w` is furthest first; the device returns nearest first */
					for (int i = 0; i < nw; i++)
					{
						lw[i].element = wi[nw - 1 - i];
						lw[i].distance = wdist[nw - 1 - i];
						lw[i].local = nw - 1 - i;
						lw[i].closer = 0;
						c[i] = &lw[i];
					}
					{
						/* (the triangle is only there, and only looked at, when the list has to be thinned) */
						pairsrc		ps = pairs_of(cdist + tri_off[(size_t) b * lcap + lc], 1, NULL);

						rn = select_neighbors(c, nw, lm, &ps, &closer_set, NULL, r, NULL, 0, NULL, w, wd, added, flag);
					}
					x->layers[lc].closer_set = closer_set;
					x->layers[lc].items = items_alloc(lm);
					x->layers[lc].length = rn;
					for (int i = 0; i < rn; i++)
						x->layers[lc].items[i] = *r[i];
				}
				for (int lc = 0; lc <= x->level; lc++)
					if (!x->layers[lc].items)
						x->layers[lc].items = items_alloc(layer_m(m, lc));
			}
			free(lw);
			free(c);
			free(flag);
		}
		/* FindDuplicateInMemory, src/hnswbuild.c:313-364, in heap order: neighbors are ordered by distance */
		for (int b = 0; b < B; b++)
		{
			int32_t		e = (int32_t) (i0 + b);
			elem	   *x = &el[e];
			const nlist *l0 = &x->layers[0];
			const char *v = (const char *) rows + (size_t) e * item_bytes;

			for (int i = 0; i < l0->length; i++)
			{
				int32_t		ne = l0->items[i].element;

				if (memcmp(v, (const char *) rows + (size_t) ne * item_bytes, item_bytes) != 0)
					break;
				if (el[ne].heaptids < HNSW_HEAPTIDS)
				{
					el[ne].heaptids++;
					out->dup_of[e] = ne;
					break;
				}
			}
			if (out->dup_of[e] >= 0)
			{
				for (int lc = 0; lc <= x->level; lc++)
					free(x->layers[lc].items);
				free(x->layers);
				x->layers = NULL;
			}
		}

		PHASE(PH_RECORDS);
		/* ---- 4. the lists this batch links into, and every distance their re-selections can look up.
		 * 4a: one request per (batch element, chosen neighbor, layer) in the order the reference's loop would link
		 * them, and a bare record per distinct list.  The lists are dealt to the threads by owner: every thread walks
		 * all requests (they are the batch's fresh lists, a few hundred KB) and keeps its own, in its own small hash
		 * table -- the order inside a list is the walk's, whatever the number of threads. */
		{
			int64_t		nlinks = 0;

			for (int b = 0; b < B; b++)
			{
				int32_t		e = (int32_t) (i0 + b);

				if (!el[e].layers)
					continue;
				linked++;
				if (!is_dirty[e])
				{
					is_dirty[e] = 1;
					ndirty++;
				}
				/* the entry point moves up with the tallest element (src/hnswbuild.c:425-430) */
				if (el[e].level > el[entry].level)
					entry = e;
			}
			for (int t = 0; t < nthreads; t++)	/* (a short batch runs the walk on one thread: the others' stay empty) */
				parts[t].ndirty = 0;
#pragma omp parallel if (B >= 8) num_threads(nthreads)
			{
				const int	T = omp_get_num_threads(),
							me = omp_get_thread_num();
				recpart    *rp = &parts[me];

				rp->nrec = 0;
				rp->nlinks = 0;
				rp->ndirty = 0;
				if (rp->tab_cap == 0)
				{
					rp->tab_cap = 1 << 10;	/* grows by doubling at load factor 1/2 */
					rp->tab = malloc(sizeof(int32_t) * (size_t) rp->tab_cap);
				}
				if (!rp->tab)
					rp->oom = 1;
				else
					memset(rp->tab, 0, sizeof(int32_t) * (size_t) rp->tab_cap);
				for (int b = 0; b < B && !rp->oom; b++)
				{
					int32_t		e = (int32_t) (i0 + b);
					elem	   *x = &el[e];

					if (!x->layers)
						continue;
					for (int lc = x->level; lc >= 0; lc--)
						for (int i = 0; i < x->layers[lc].length; i++)
						{
							int32_t		owner = x->layers[lc].items[i].element;
							uint64_t	key,
										hv;
							int64_t		h;
							int32_t		ri;		/* record index + 1 (the table may be rebuilt below: h is not kept) */

							if ((int) (((uint64_t) ((uint32_t) owner * 2654435761u) * (uint64_t) T) >> 32) != me)
								continue;
							key = ((uint64_t) owner << 6) | (uint64_t) lc;
							hv = key * 0x9E3779B97F4A7C15ull;
							h = (int64_t) (hv >> 40) & (rp->tab_cap - 1);
							while (rp->tab[h] != 0 && !(rp->recs[rp->tab[h] - 1].owner == owner && rp->recs[rp->tab[h] - 1].lc == lc))
								h = (h + 1) & (rp->tab_cap - 1);
							ri = rp->tab[h];
							if (ri == 0)
							{
								record	   *rcd;

								if (rp->nrec == rp->cap)
								{
									int			cap = rp->cap ? rp->cap * 2 : 1024;
									record	   *grown = realloc(rp->recs, sizeof(record) * (size_t) cap);

									if (!grown)
									{
										rp->oom = 1;
										break;
									}
									rp->recs = grown;
									rp->cap = cap;
								}
								rcd = &rp->recs[rp->nrec];
								rcd->owner = owner;
								rcd->lc = lc;
								rcd->newcap = 0;	/* newcomers, counted here */
								rp->tab[h] = ++rp->nrec;
								ri = rp->nrec;
								if ((int64_t) rp->nrec * 2 > rp->tab_cap)
								{
									/* load factor 1/2 reached (a batch touches up to B * (2m + level * m) lists; m up to
									 * 100, src/hnsw.h:50): double the table and rehash the records made so far */
									int64_t		ncap = rp->tab_cap * 2;
									int32_t    *nt = calloc((size_t) ncap, sizeof(int32_t));

									if (!nt)
									{
										rp->oom = 1;
										break;
									}
									for (int r = 0; r < rp->nrec; r++)
									{
										uint64_t	k2 = ((uint64_t) rp->recs[r].owner << 6) | (uint64_t) rp->recs[r].lc;
										int64_t		h2 = (int64_t) ((k2 * 0x9E3779B97F4A7C15ull) >> 40) & (ncap - 1);

										while (nt[h2] != 0)
											h2 = (h2 + 1) & (ncap - 1);
										nt[h2] = r + 1;
									}
									free(rp->tab);
									rp->tab = nt;
									rp->tab_cap = ncap;
								}
								if (!is_dirty[owner])	/* (owners are dealt to threads: no two write one flag) */
								{
									if (rp->ndirty == rp->dirty_cap)
									{
										int			cap = rp->dirty_cap ? rp->dirty_cap * 2 : 1024;
										int32_t    *grown = realloc(rp->dirty, sizeof(int32_t) * (size_t) cap);

										if (!grown)
										{
											rp->oom = 1;
											break;
										}
										rp->dirty = grown;
										rp->dirty_cap = cap;
									}
									is_dirty[owner] = 1;
									rp->dirty[rp->ndirty++] = owner;
								}
							}
							if (rp->nlinks == rp->links_cap)
							{
								int64_t		cap = rp->links_cap ? rp->links_cap * 2 : 4096;
								link_req   *grown = realloc(rp->links, sizeof(link_req) * (size_t) cap);

								if (!grown)
								{
									rp->oom = 1;
									break;
								}
								rp->links = grown;
								rp->links_cap = cap;
							}
							rp->links[rp->nlinks].rec = ri - 1;
							rp->links[rp->nlinks].element = e;
							rp->links[rp->nlinks].distance = x->layers[lc].items[i].distance;
							rp->nlinks++;
							rp->recs[ri - 1].newcap++;
						}
				}
#pragma omp barrier
#pragma omp single
				{
					/* where each thread's records and requests go in the batch's arrays */
					int64_t		nr = 0,
								nl = 0;

					for (int t = 0; t < T; t++)
					{
						parts[t].rec_base = nr;
						parts[t].link_base = nl;
						nr += parts[t].nrec;
						nl += parts[t].nlinks;
						ndirty += parts[t].ndirty;
						oom |= parts[t].oom;
					}
					if (nr + 1 > recs_cap)
					{
						recs_cap = (int) (nr + 1) * 2;
						recs = realloc(recs, sizeof(record) * (size_t) recs_cap);
						grp_off = realloc(grp_off, sizeof(int64_t) * (size_t) (recs_cap + 1));
					}
					if (nl + 1 > links_cap)
					{
						links_cap = (nl + 1) * 2;
						grp_elem = realloc(grp_elem, sizeof(int32_t) * (size_t) links_cap);
						grp_dist = realloc(grp_dist, sizeof(float) * (size_t) links_cap);
					}
					if (!recs || !grp_off || !grp_elem || !grp_dist)
						oom = 1;
					else
						grp_off[nr] = nl;
					nrec = (int) nr;
					nlinks = nl;
				}
				/* 4b: the requests grouped by list, link order kept inside a list (counting sort, thread by thread) */
				if (!oom)
				{
					int64_t		at = rp->link_base;

					for (int k = 0; k < rp->nrec; k++)
					{
						grp_off[rp->rec_base + k] = at;
						at += rp->recs[k].newcap;
						rp->recs[k].nlocal = 0;		/* fill cursor */
					}
					for (int64_t t = 0; t < rp->nlinks; t++)
					{
						record	   *rcd = &rp->recs[rp->links[t].rec];
						int64_t		to = grp_off[rp->rec_base + rp->links[t].rec] + rcd->nlocal++;

						grp_elem[to] = rp->links[t].element;
						grp_dist[to] = rp->links[t].distance;
					}
					if (rp->nrec > 0)
						memcpy(recs + rp->rec_base, rp->recs, sizeof(record) * (size_t) rp->nrec);
				}
			}
			(void) nlinks;
			if (oom)
			{
				rc = pgv_host_fail(PGV_ERR_NOMEM, "out of memory for the batch's list records");
				goto done;
			}
			/* 4c (parallel: this is where the owners' lists, cold in the cache, are read): members + newcomers */
#pragma omp parallel for if (B >= 8) num_threads(nthreads) schedule(static)
			for (int k = 0; k < nrec; k++)
			{
				record	   *rcd = &recs[k];
				nlist	   *l = &el[rcd->owner].layers[rcd->lc];
				int			nnew = rcd->newcap;

				/* element -> its layers -> the list's items: three dependent misses per record, fetched ahead (a
				 * neighbouring thread's records at the chunk's end: harmless) */
				if (k + 12 < nrec)
					__builtin_prefetch(&el[recs[k + 12].owner]);
				if (k + 8 < nrec)
					__builtin_prefetch(&el[recs[k + 8].owner].layers[recs[k + 8].lc]);
				if (k + 4 < nrec)
				{
					const char *it = (const char *) el[recs[k + 4].owner].layers[recs[k + 4].lc].items;

					__builtin_prefetch(it);
					__builtin_prefetch(it + 64);
					__builtin_prefetch(it + 128);
					__builtin_prefetch(it + 192);
				}

				rcd->nstart = l->length;
				rcd->nlocal = l->length + nnew;
				rcd->ids = arena_alloc(&arenas[omp_get_thread_num()], sizeof(int32_t) * (size_t) rcd->nlocal);
				if (!rcd->ids)
				{
#pragma omp atomic write
					oom = 1;
					rcd->nlocal = rcd->nstart = 0;
					continue;
				}
				rcd->newdist = grp_dist + grp_off[k];
				rcd->items = l->items;
				rcd->full = !l->closer_set;	/* no cached flags: its next selection computes everything */
				rcd->blocked = 0;
				rcd->wait_from = -1;
				for (int j = 0; j < l->length; j++)
				{
					rcd->ids[j] = l->items[j].element;
					l->items[j].local = j;
				}
				memcpy(rcd->ids + l->length, grp_elem + grp_off[k], sizeof(int32_t) * (size_t) nnew);
			}
			if (oom)
			{
				rc = pgv_host_fail(PGV_ERR_NOMEM, "out of memory for the batch's list records");
				goto done;
			}
		}
		PHASE(PH_PAIRLIST);
		/* the records in slices: with the scorer running, slice s + 1's pairs are scored while slice s is replayed */
		nslices = scorer.started && nrec >= 64 * SCORE_SLICES ? SCORE_SLICES : 1;
		{
			int64_t		total = 0;

			/* (one thread per slice; pair offsets are the slice's own: its reply starts at slices[.].out) */
#pragma omp parallel for if (nslices > 1) num_threads(nslices) schedule(static, 1)
			for (int sl = 0; sl < nslices; sl++)
			{
				score_job  *sj = &slices[sl];
				int64_t		mine = 0;

				/* a short first slice (it is the one waited for with nothing to do), the rest in equal parts */
				sj->klo = sl == 0 ? 0 : (int) (((int64_t) nrec * (1 + (sl - 1) * 7 / (nslices - 1))) / 8);
				sj->khi = sl == nslices - 1 ? nrec : (int) (((int64_t) nrec * (1 + sl * 7 / (nslices - 1))) / 8);
				sj->rc = PGV_OK;
				sj->seq = 0;
				groups_reset(&sj->gb);
				for (int k = sj->klo; k < sj->khi; k++)
				{
					record	   *rcd = &recs[k];
					int			from = rcd->full ? 1 : rcd->nstart;	/* cached flags: only the pairs that involve a newcomer */

					rcd->pair0 = mine;
					rcd->slice = sl;
					rcd->from = from < 1 ? 1 : from;
					/* a list that cannot overflow in this batch never runs a selection */
					if (rcd->nlocal <= layer_m(m, rcd->lc))
						continue;
					if (!groups_add(&sj->gb, rcd->ids, rcd->nlocal, from, mine))
					{
						sj->rc = PGV_ERR_NOMEM;
						break;
					}
					mine += group_pairs(rcd->nlocal, from);
				}
				sj->npairs = mine;
				if (sj->gb.ngroups > 0)
					sj->gb.pair_start[sj->gb.ngroups] = mine;
			}
			for (int sl = 0; sl < nslices; sl++)
			{
				if (slices[sl].rc != PGV_OK)
				{
					rc = pgv_host_fail(PGV_ERR_NOMEM, "out of memory");
					goto done;
				}
				total += slices[sl].npairs;
			}
			npairs = total;
		}
		PHASE(PH_PAIRS);
		if (npairs > pdist_cap)
		{
			pgv_pinned_free(pdist);
			pdist = NULL;
			pdist_cap = npairs * 2;
			if ((rc = pgv_pinned_alloc(sizeof(float) * (size_t) pdist_cap, (void **) &pdist)) != PGV_OK)
			{
				pdist_cap = 0;
				goto dev_fail;
			}
		}
		{
			int64_t		base = 0;

			for (int sl = 0; sl < nslices; sl++)
			{
				score_job  *sj = &slices[sl];

				sj->out = pdist + base;
				base += sj->npairs;
				if (sj->npairs == 0)
					continue;
				if (nslices > 1)
					sj->seq = worker_post(&scorer, JOB_SCORE, sj);
				else
				{
					rc = pgv_hnsw_score_groups(mirror, sj->gb.ids, sj->gb.ids_start, sj->gb.from, sj->gb.pair_start, sj->gb.ngroups,
											   sj->gb.nids, sj->npairs, sj->out);
					if (rc != PGV_OK)
						goto dev_fail;
				}
			}
			out->device_pairs += npairs;
		}
		PHASE(PH_UPDATE);

		/* ---- 5. HnswUpdateNeighborsInMemory (src/hnswbuild.c:376-405).  The reference links one element after
		 * the other; updates of different lists do not see each other, so the lists are replayed in parallel,
		 * each with its newcomers in heap order. */
		{
			int			nblocked = 0;

			/* steps 0 .. nslices - 1: the first replay of slice `step`'s lists; step nslices: the second pass over all */
			for (int step = 0; step <= nslices; step++)
			{
				const int	pass = step == nslices;
				const int	klo = pass ? 0 : slices[step].klo,
							khi = pass ? nrec : slices[step].khi;

				if (!pass && slices[step].seq != 0)
				{
					PHASE(PH_PAIRS);
					worker_wait(&scorer, slices[step].seq);
					PHASE(PH_UPDATE);
					if (slices[step].rc != PGV_OK)
					{
						rc = pgv_host_fail(slices[step].rc, "%s", slices[step].err);
						goto done;
					}
				}
				if (pass == 1)
				{
					/* ---- 5b. the updates that were put aside: their lists' member-member pairs, then the replay */
					if (nblocked == 0)
						break;
					groups_reset(&gb);
					npairs = 0;
					for (int k = 0; k < nrec; k++)
						if (recs[k].blocked)
						{
							recs[k].pair0b = npairs;
							if (!groups_add(&gb, recs[k].ids, recs[k].nstart, 1, npairs))
							{
								rc = pgv_host_fail(PGV_ERR_NOMEM, "out of memory");
								goto done;
							}
							npairs += group_pairs(recs[k].nstart, 1);
						}
					gb.pair_start[gb.ngroups] = npairs;
					/* (the first request's reply stays: the newcomers' pairs are looked up there) */
					if (npairs > pdist2_cap)
					{
						pgv_pinned_free(pdist2);
						pdist2 = NULL;
						pdist2_cap = npairs * 2;
						if ((rc = pgv_pinned_alloc(sizeof(float) * (size_t) pdist2_cap, (void **) &pdist2)) != PGV_OK)
						{
							pdist2_cap = 0;
							goto dev_fail;
						}
					}
					rc = pgv_hnsw_score_groups(mirror, gb.ids, gb.ids_start, gb.from, gb.pair_start, gb.ngroups, gb.nids, npairs, pdist2);
					if (rc != PGV_OK)
						goto dev_fail;
					out->device_pairs += npairs;
					out->deferred_updates += nblocked;
				}
#pragma omp parallel if (B >= 8) num_threads(nthreads)
				{
					int			lm0 = layer_m(m, 0);
					cand	  **c = malloc(sizeof(cand *) * (size_t) (lm0 + 1) * 5);
					cand	  **r = c + (lm0 + 1),
							  **w = c + 2 * (lm0 + 1),
							  **wd = c + 3 * (lm0 + 1),
							  **added = c + 4 * (lm0 + 1);
					uint8_t    *flag = malloc((size_t) lm0 + 1);

#pragma omp for schedule(dynamic, 16) reduction(+:nblocked)
					for (int k = klo; k < khi; k++)
					{
						record	   *rcd = &recs[k];
						nlist	   *l = &el[rcd->owner].layers[rcd->lc];
						int			lm = layer_m(m, rcd->lc);
						int			from = rcd->nstart;
						pairsrc		ps;

						if (pass == 1)
						{
							if (!rcd->blocked)
								continue;
							from = rcd->wait_from;
						}
						else if (k + 4 < khi)
						{
							/* a later record's list (last touched by another thread in step 4) and its pairs (fresh
							 * from the device: in no cache) */
							const record *nx = &recs[k + 4];
							const char *it = nx->items,
									   *tr = (const char *) (slices[nx->slice].out + nx->pair0);
							const int	nlm = layer_m(m, nx->lc);

							for (int o = 0; o < nlm * (int) sizeof(cand); o += 64)
								__builtin_prefetch(it + o);
							__builtin_prefetch(tr);
							__builtin_prefetch(tr + 64);
							/* the cached sort order behind the items, and where the list's tuple part goes */
							__builtin_prefetch(it + nlm * (int) sizeof(cand));
							__builtin_prefetch(out->nbr + out->nbr_start[nx->owner] + (int64_t) (el[nx->owner].level - nx->lc) * m, 1);
							__builtin_prefetch(nx->newdist);
						}
						ps = pairs_of(slices[rcd->slice].out + rcd->pair0, rcd->from, pass ? pdist2 + rcd->pair0b : NULL);
						for (int u = from; u < rcd->nlocal; u++)
						{
							cand		new_hc;
							cand	   *pruned = NULL;
							int			nc,
										rn;

							new_hc.element = rcd->ids[u];
							new_hc.distance = rcd->newdist[u - rcd->nstart];
							new_hc.local = u;
							new_hc.closer = 0;
							/* HnswUpdateConnection, src/hnswutils.c:1183-1231 */
							if (l->length < lm)
							{
								l->items[l->length++] = new_hc;
								l->ord_valid = 0;
								continue;
							}
							nc = l->length + 1;
							for (int j = 0; j < l->length; j++)
								c[j] = &l->items[j];
							c[nc - 1] = &new_hc;
							rn = select_neighbors(c, nc, lm, &ps, &l->closer_set, &new_hc, r, &pruned, 1,
												  l->ord_valid ? items_ord(l, lm) : NULL, w, wd, added, flag);
							if (rn < 0)
							{
								/* needs member-member distances that were not fetched: this and every later
								 * update of the list wait for the second launch */
								rcd->blocked = 1;
								rcd->wait_from = u;
								nblocked++;
								break;
							}
							{
								/* the list keeps its members' places, the newcomer takes the dropped one's
								 * (src/hnswutils.c:1211-1227); the sorted order w, less the dropped candidate, is kept
								 * for the list's next selection */
								uint8_t    *ord = items_ord(l, lm);
								int			slot = -1,
											o = 0;

								if (pruned != NULL && pruned != &new_hc)
									slot = (int) (pruned - l->items);
								for (int i = 0; i < nc; i++)
									if (w[i] != pruned)
										ord[o++] = (uint8_t) (w[i] == &new_hc ? slot : (int) (w[i] - l->items));
								l->ord_valid = 1;
								if (slot >= 0)
									l->items[slot] = new_hc;
							}
						}
						/* the list's part of its owner's neighbor tuple, for step 6 */
						write_segment(&el[rcd->owner], rcd->owner, rcd->lc, m, out->nbr_start, out->nbr);
					}
					free(c);
					free(flag);
				}
			}
		}

		if (oom)
		{
			rc = pgv_host_fail(PGV_ERR_NOMEM, "out of memory for the batch's list records");
			goto done;
		}
		PHASE(PH_PATCH);
		/* ---- 6. the graph the next batch searches.  With the helper running, the patch is its job: posted behind the
		 * stage A that runs ahead (whose kernels read the tuples this step overwrites) and in front of the next one;
		 * the arrays of the job before last are free again by now. */
		{
			patch_job  *pj = &patches[pcur];
			int			k = 0;

			worker_wait(&helper, pj->seq);
			if (pj->rc != PGV_OK)
			{
				rc = pgv_host_fail(pj->rc, "%s", pj->err);
				goto done;
			}
			if (ndirty + 1 > pj->dirty_cap)
			{
				pj->dirty_cap = (int64_t) (ndirty + 1) * 2;
				pj->dirty = realloc(pj->dirty, sizeof(int32_t) * (size_t) pj->dirty_cap);
				pj->packed_off = realloc(pj->packed_off, sizeof(int64_t) * (size_t) pj->dirty_cap);
			}
			if (!pj->dirty || !pj->packed_off)
			{
				rc = pgv_host_fail(PGV_ERR_NOMEM, "out of memory");
				goto done;
			}
			/* dirty elements: the batch itself and the owners of the touched lists */
			for (int b = 0; b < B; b++)
				if (is_dirty[i0 + b])
				{
					pj->dirty[k++] = (int32_t) (i0 + b);
					is_dirty[i0 + b] = 0;
				}
			for (int t = 0; t < nthreads; t++)	/* the owners, as step 4a's threads met them */
				for (int q = 0; q < parts[t].ndirty; q++)
				{
					pj->dirty[k++] = parts[t].dirty[q];
					is_dirty[parts[t].dirty[q]] = 0;
				}
			ndirty = k;
			ntuple = 0;
			for (int q = 0; q < ndirty; q++)
			{
				pj->packed_off[q] = ntuple;
				ntuple += out->nbr_start[pj->dirty[q] + 1] - out->nbr_start[pj->dirty[q]];
			}
			pj->packed_off[ndirty] = ntuple;
			if (ntuple + 1 > pj->packed_cap)
			{
				pj->packed_cap = (ntuple + 1) * 2;
				free(pj->packed);
				pj->packed = malloc(sizeof(int32_t) * (size_t) pj->packed_cap);
				if (!pj->packed)
				{
					rc = pgv_host_fail(PGV_ERR_NOMEM, "out of memory");
					goto done;
				}
			}
#pragma omp parallel for if (B >= 8) num_threads(nthreads) schedule(static)
			for (int q = 0; q < ndirty; q++)
			{
				/* (the lists touched by the replay wrote their parts in step 5; the batch's own elements are new) */
				if (pj->dirty[q] >= i0)
					write_tuple(el, pj->dirty[q], m, out->nbr_start, out->nbr);
				memcpy(pj->packed + pj->packed_off[q], out->nbr + out->nbr_start[pj->dirty[q]],
					   sizeof(int32_t) * (size_t) (pj->packed_off[q + 1] - pj->packed_off[q]));
			}
			pj->entry = entry;
			pj->ndirty = ndirty;
			if (helper.started)
			{
				pj->seq = worker_post(&helper, JOB_PATCH, pj);
				pcur ^= 1;
			}
			else
			{
				rc = pgv_hnsw_update_graph(mirror, entry, pj->dirty, ndirty, pj->packed_off, pj->packed);
				if (rc != PGV_OK)
					goto dev_fail;
			}
		}

		PHASE(PH_FREE);
		/* forget the batch's records */
		for (int t = 0; t < nthreads; t++)
			arena_reset(&arenas[t]);
		nrec = 0;
		PHASE(PH_RECORDS);
		i0 += B;
		out->batches++;
		if (ahead_valid)
			cur ^= 1;
	}
	/* the last patches are on the device before the caller searches the mirror */
	worker_drain(&helper);
	for (int i = 0; i < 2; i++)
		if (patches[i].rc != PGV_OK)
		{
			rc = pgv_host_fail(patches[i].rc, "%s", patches[i].err);
			break;
		}
	goto done;

dev_fail:
	rc = pgv_host_fail(rc, "%s", pgv_last_error());
done:
	worker_stop(&helper);
	worker_stop(&scorer);
	for (int i = 0; i < 2; i++)
	{
		free(patches[i].dirty);
		free(patches[i].packed_off);
		free(patches[i].packed);
	}
	for (int i = 0; i < SCORE_SLICES; i++)
	{
		free(slices[i].gb.ids);
		free(slices[i].gb.ids_start);
		free(slices[i].gb.from);
		free(slices[i].gb.pair_start);
	}
	stage_a_free(&stages[0]);
	stage_a_free(&stages[1]);
	out->entry = entry;
	out->nelements = linked;
	if (el)
	{
		for (int64_t i = 0; i < n; i++)
			if (el[i].layers)
			{
				for (int lc = 0; lc <= el[i].level; lc++)
					free(el[i].layers[lc].items);
				free(el[i].layers);
			}
		free(el);
	}
	if (arenas)					/* incl. a batch cut short by an error */
	{
		for (int t = 0; t < nthreads; t++)
			arena_free(&arenas[t]);
		free(arenas);
	}
	free(grp_elem);
	free(grp_dist);
	free(grp_off);
	free(gb.ids);
	free(gb.ids_start);
	free(gb.from);
	free(gb.pair_start);
	pgv_pinned_free(pdist);
	pgv_pinned_free(pdist2);
	free(is_dirty);
	free(recs);
	if (parts)
	{
		for (int t = 0; t < nthreads; t++)
		{
			free(parts[t].recs);
			free(parts[t].dirty);
			free(parts[t].links);
			free(parts[t].tab);
		}
		free(parts);
	}
	if (rc != PGV_OK)
		pgv_host_hnsw_built_free(out);
	return rc;
}
