/*
 * hnsw_link_core.h -- HnswUpdateConnection for ONE neighbor list and the newcomers a batch links into it
 * (src/hnswutils.c:1183-1231), with SelectNeighbors as the reference runs it on a list that is already full: the
 * candidates sorted by CompareCandidateDistances (:992-1010), the cached `closer` flags of the list's last selection
 * reused where the reference reuses them (:1098-1140), CheckElementCloser (:1040-1059) for the rest, pruned connections
 * kept (:1146-1148), the dropped candidate's place handed to the newcomer (:1211-1227).
 *
 * Plain C, no pointers into a graph: the list is three arrays (element, distance, closer flag) plus each item's "local"
 * -- its index in the record's id list, which is how the pair distances are found (a triangle (u, v < u), u >= from, the
 * layout pgv_hnsw_score_groups / hnsw_link_pairs_kernel write).  ONE source for two compilations:
 *   - kernels_hnsw.hip includes it with PGV_LINK_FN = __device__: hnsw_link_kernel runs it, one lane per list;
 *   - tests/c/mock_hip.c includes it as host C, so that the CPU tests of the host build exercise this very code.
 * pgvector_amd/host/hnsw_build.c holds the older host-side replay of the same steps (pointer-based, with a cached sort
 * order); the GPU tests build the same graphs both ways.
 */
#ifndef PGV_HNSW_LINK_CORE_H
#define PGV_HNSW_LINK_CORE_H

#include <stdint.h>

#ifndef PGV_LINK_FN
#define PGV_LINK_FN static inline
#endif

#define PGV_LINK_LMAX 200		/* layer 0 of m = 100 (src/hnsw.h:56) */

/* where the distance of two locals is found (pairs_of / check_closer of host/hnsw_build.c) */
typedef struct
{
	const float *tri;			/* pairs (u, v < u) for u >= from, u ascending then v */
	int			from;			/* >= 1 */
	int			base;			/* from * (from - 1) / 2 */
	const float *mm;			/* the member-member triangle (u < from) of a second request, or NULL */
}			pgv_link_pairs;

/* CompareCandidateDistances as one integer: larger = earlier in the reference's (descending) order */
PGV_LINK_FN uint64_t
pgv_link_key(float distance, int32_t element)
{
	union
	{
		float		f;
		uint32_t	u;
	}			c;

	c.f = distance + 0.0f;		/* (-0.0 counts as 0.0, as in a float comparison) */
	c.u ^= (c.u >> 31) ? 0xFFFFFFFFu : 0x80000000u;
	return ((uint64_t) c.u << 32) | (uint32_t) element;
}

/* CheckElementCloser of candidate (local a, distance da) against the candidates set[0 .. n) (indexes into loc[]);
 * *missing is set when a pair was not fetched */
PGV_LINK_FN int
pgv_link_check_closer(int a, float da, const uint8_t *set, int n, const int16_t *loc, const pgv_link_pairs * ps, int *missing)
{
	for (int i = 0; i < n; i++)
	{
		const int	b = loc[set[i]];
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
		if (d <= da)
			return 0;
	}
	return 1;
}

/*
 * SelectNeighbors over candidates 0 .. nc - 1 (nc = lm + 1: the full list and, as candidate nc - 1, the newcomer).
 * ce / cd / cf / loc: element, distance, cached closer flag, local of each candidate.  On success the flags of the
 * candidates looked at are updated in cf, *closer_set = 1, and *pruned is the candidate to drop; returns |r| (always lm
 * here).  Returns -1 when a pair was needed that has not been fetched: nothing has been changed then.
 * Scratch (the caller's, so that it is as large as ITS lists need and no larger): key, w, r, wd, added, flag [nc] each.
 */
PGV_LINK_FN int
pgv_link_select(int nc, int lm, const int32_t *ce, const float *cd, uint8_t *cf, const int16_t *loc,
				const pgv_link_pairs * ps, uint8_t *closer_set, int *pruned,
				uint64_t *key, uint8_t *w, uint8_t *r, uint8_t *wd, uint8_t *added, uint8_t *flag)
{
	const int	new_cand = nc - 1;
	int			wn = nc,
				rn = 0,
				wdlen = 0,
				wdoff = 0,
				nadded = 0;
	const int	must_calculate = !(*closer_set);
	int			removed_any = 0;
	int			missing = 0;

	/* list_sort(w, CompareCandidateDistances): furthest first, w[wn - 1] the closest (a total order: an insertion sort) */
	for (int i = 0; i < nc; i++)
	{
		const uint64_t k = pgv_link_key(cd[i], ce[i]);
		int			j = i;

		while (j > 0 && key[j - 1] < k)
		{
			key[j] = key[j - 1];
			w[j] = w[j - 1];
			j--;
		}
		key[j] = k;
		w[j] = (uint8_t) i;
	}
	while (wn > 0 && rn < lm)
	{
		const int	e = w[--wn];	/* closest remaining */
		uint8_t		closer = cf[e];

		/* use the previous state of r and wd to skip work when possible (:1098-1140) */
		if (must_calculate)
			closer = (uint8_t) pgv_link_check_closer(loc[e], cd[e], r, rn, loc, ps, &missing);
		else if (nadded > 0)
		{
			if (closer)
			{
				closer = (uint8_t) pgv_link_check_closer(loc[e], cd[e], added, nadded, loc, ps, &missing);
				if (!closer)
					removed_any = 1;
			}
			else if (removed_any)
			{
				closer = (uint8_t) pgv_link_check_closer(loc[e], cd[e], r, rn, loc, ps, &missing);
				if (closer)
					added[nadded++] = (uint8_t) e;
			}
		}
		else if (e == new_cand)
		{
			closer = (uint8_t) pgv_link_check_closer(loc[e], cd[e], r, rn, loc, ps, &missing);
			if (closer)
				added[nadded++] = (uint8_t) e;
		}
		if (missing)
			return -1;
		flag[wn] = closer;		/* committed below: a replay that runs out of distances must leave no trace */
		if (closer)
			r[rn++] = (uint8_t) e;
		else
			wd[wdlen++] = (uint8_t) e;
	}
	for (int i = wn; i < nc; i++)
		cf[w[i]] = flag[i];
	*closer_set = 1;			/* sorted deterministically: the flags are reusable (:1143-1144) */
	/* keep pruned connections (:1146-1148) */
	while (wdoff < wdlen && rn < lm)
		r[rn++] = wd[wdoff++];
	*pruned = wdoff < wdlen ? wd[wdoff] : w[0];	/* :1150-1157 */
	return rn;
}

/*
 * The newcomers u = first .. nlocal - 1 of one record linked into its list, in order (HnswUpdateConnection each).
 *   le / ld / lf [lm]: the list's elements, distances, closer flags (in place); *len its length, *closer_set its flag
 *   loc [lm]: the local of each item (members: their slot when the batch began; newcomers: nstart + their rank)
 *   ids [nlocal], newdist [nlocal - nstart]: the record's id list and each newcomer's distance to the owner
 *   key [lm + 1], scratch [5 x (lm + 1)]: SelectNeighbors' working arrays
 * Returns nlocal when every newcomer is linked, or the u at which a missing pair stopped the replay (that newcomer and
 * the later ones wait for the member-member distances).
 */
PGV_LINK_FN int
pgv_link_replay(int32_t *le, float *ld, uint8_t *lf, int16_t *loc, int *len, uint8_t *closer_set, int lm,
				const int32_t *ids, const float *newdist, int nstart, int nlocal, int first, const pgv_link_pairs * ps,
				uint64_t *key, uint8_t *scratch)
{
	uint8_t    *w = scratch,
			   *r = scratch + (lm + 1),
			   *wd = scratch + 2 * (lm + 1),
			   *added = scratch + 3 * (lm + 1),
			   *flag = scratch + 4 * (lm + 1);

	for (int u = first; u < nlocal; u++)
	{
		const int32_t ne = ids[u];
		const float nd = newdist[u - nstart];
		int			pruned = 0,
					rn;

		if (*len < lm)
		{
			le[*len] = ne;
			ld[*len] = nd;
			lf[*len] = 0;
			loc[*len] = (int16_t) u;
			(*len)++;
			continue;
		}
		/* the newcomer as candidate lm, behind the list (the arrays have lm + 1 places) */
		le[lm] = ne;
		ld[lm] = nd;
		lf[lm] = 0;
		loc[lm] = (int16_t) u;
		rn = pgv_link_select(lm + 1, lm, le, ld, lf, loc, ps, closer_set, &pruned, key, w, r, wd, added, flag);
		if (rn < 0)
			return u;
		/* the list keeps its members' places, the newcomer takes the dropped one's (:1211-1227) */
		if (pruned != lm)
		{
			le[pruned] = le[lm];
			ld[pruned] = ld[lm];
			lf[pruned] = lf[lm];
			loc[pruned] = loc[lm];
		}
	}
	return nlocal;
}

#endif
