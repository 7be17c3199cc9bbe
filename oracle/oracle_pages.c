/*
 * oracle_pages.c -- the reference's IVFFlat scan restated over the REAL on-disk layout: an
 * array of 8 KB PostgreSQL pages holding pgvector's meta page, list pages and entry pages.
 *
 * TEST INFRASTRUCTURE (CPU checker and cpu_baseline of bench.py); never linked into the product.
 *
 * oracle_ivf.c walks contiguous arrays, which flatters a CPU: the reference touches a page
 * header, a line pointer and an IndexTuple header per tuple, follows nextblkno chains, and at
 * 1536 dimensions finds one tuple per 8 KB page with 25 % of it padding.  This
 * file follows
 *   GetScanLists   src/ivfscan.c:47-118   (list pages from IVFFLAT_HEAD_BLKNO, PageGetItem of every
 *                                          IvfflatList, bounded selection of maxProbes, strict `<`)
 *   GetScanItems   src/ivfscan.c:123-187  (entry-page chain of every probed list, index_getattr of
 *                                          every IndexTuple, distance, tuplesort input)
 *   the tuplesort  src/ivfscan.c:182, :238-247 (ascending float8; ties keep insertion order here)
 * over the page layout of PostgreSQL's bufpage.h / itup.h (24-byte page header, 4-byte line
 * pointers lp_off:15 lp_flags:2 lp_len:15, 8-byte IndexTupleData, varlena with 4-byte or 1-byte
 * header) and pgvector's src/ivfflat.h:46-52, :251-275 (magic 0x14FF1A7, page id 0xFF84,
 * IvfflatMetaPageData, IvfflatListData {startPage, insertPage, center}, IvfflatPageOpaqueData).
 * Buffer pins and locks, fmgr and tuplesort's own copies are NOT reproduced: still an upper
 * bound of a real server's speed, a much closer one.
 */
#include "pgv_oracle.h"

#include <float.h>
#include <stdlib.h>
#include <string.h>

#define ORA_BLCKSZ 8192
#define ORA_PAGE_HEADER 24
#define ORA_INVALID_BLOCK 0xFFFFFFFFu
#define ORA_IVF_MAGIC 0x14FF1A7u
#define ORA_IVF_PAGE_ID 0xFF84u
#define ORA_IVF_HEAD_BLKNO 1

static inline const uint8_t *
page_of(const uint8_t *pages, uint32_t blk)
{
	return pages + (size_t) blk * ORA_BLCKSZ;
}

static inline uint16_t
rd16(const uint8_t *p)
{
	uint16_t	v;

	memcpy(&v, p, 2);
	return v;
}

static inline uint32_t
rd32(const uint8_t *p)
{
	uint32_t	v;

	memcpy(&v, p, 4);
	return v;
}

/* PageGetMaxOffsetNumber */
static inline int
max_offset(const uint8_t *page)
{
	int			lower = rd16(page + 12);	/* pd_lower */

	return lower <= ORA_PAGE_HEADER ? 0 : (lower - ORA_PAGE_HEADER) / 4;
}

/* PageGetItem(page, PageGetItemId(page, offno)) */
static inline const uint8_t *
get_item(const uint8_t *page, int offno)
{
	uint32_t	lp = rd32(page + ORA_PAGE_HEADER + (size_t) (offno - 1) * 4);

	return page + (lp & 0x7FFFu);
}

/* IvfflatPageGetOpaque(page)->nextblkno: the special space starts at pd_special */
static inline uint32_t
next_block(const uint8_t *page)
{
	return rd32(page + rd16(page + 16));
}

/* the element payload of a vector / halfvec varlena (PG_DETOAST_DATUM's view): skips the
 * 4-byte header, or the 1-byte short header index_form_tuple gives values under 127 bytes,
 * then int16 dim + int16 unused */
static inline const void *
vector_payload(const uint8_t *datum, int *dim)
{
	const uint8_t *p = (datum[0] & 0x01) ? datum + 1 : datum + 4;	/* VARATT_IS_1B (little endian) */

	*dim = (int) rd16(p);
	return p + 4;
}

typedef struct
{
	double		distance;
	uint32_t	start_page;
}			page_list;

static void
sift_down(page_list * h, int n, int i)
{
	for (;;)
	{
		int			l = 2 * i + 1,
					r = l + 1,
					m = i;

		if (l < n && h[l].distance > h[m].distance)
			m = l;
		if (r < n && h[r].distance > h[m].distance)
			m = r;
		if (m == i)
			return;
		{
			page_list	t = h[i];

			h[i] = h[m];
			h[m] = t;
		}
		i = m;
	}
}

typedef struct
{
	double		distance;
	uint64_t	tid;
	int64_t		seq;
}			page_item;

static int
item_cmp(const void *pa, const void *pb)
{
	const page_item *a = pa,
			   *b = pb;

	/* float8 ordering: NaN after every number */
	if (a->distance != a->distance || b->distance != b->distance)
	{
		int			an = a->distance != a->distance,
					bn = b->distance != b->distance;

		if (an != bn)
			return an - bn;
	}
	else if (a->distance != b->distance)
		return a->distance < b->distance ? -1 : 1;
	return (a->seq > b->seq) - (a->seq < b->seq);
}

/* meta page (block 0): dimensions and lists; 0 on a bad image */
int
ora_pages_meta(const uint8_t *pages, uint32_t nblocks, int *dim, int *lists)
{
	const uint8_t *meta;

	if (nblocks < 1)
		return 0;
	meta = page_of(pages, 0) + ORA_PAGE_HEADER;	/* PageGetContents: MAXALIGN(SizeOfPageHeaderData) */
	if (rd32(meta) != ORA_IVF_MAGIC)
		return 0;
	*dim = rd16(meta + 8);
	*lists = rd16(meta + 10);
	return 1;
}

/*
 * ivfflatgettuple's first batch over the page image: the k nearest tuples of the `probes`
 * nearest lists, ascending.  query NULL = ZeroDistance.  Returns the number of results
 * (<= k); *out_scanned = tuples the tuplesort held.
 */
int
ora_pages_search(const uint8_t *pages, uint32_t nblocks, int ops, int dtype, const void *query,
				 int probes, int k, uint64_t *out_tids, double *out_dist, int64_t *out_scanned)
{
	int			dim,
				lists,
				count = 0;
	double		max_distance = DBL_MAX;
	page_list  *heap;
	uint32_t	blk;
	page_item  *items = NULL;
	int64_t		nitems = 0,
				cap = 0;
	int			nout;

	if (!ora_pages_meta(pages, nblocks, &dim, &lists))
		return -1;
	if (probes > lists)
		probes = lists;			/* src/ivfscan.c:274-278 */
	heap = malloc(sizeof(page_list) * (size_t) (probes > 0 ? probes : 1));

	/* GetScanLists: every list page, every IvfflatList on it */
	for (blk = ORA_IVF_HEAD_BLKNO; blk != ORA_INVALID_BLOCK && blk < nblocks;)
	{
		const uint8_t *page = page_of(pages, blk);
		int			maxoff = max_offset(page);

		for (int off = 1; off <= maxoff; off++)
		{
			const uint8_t *list = get_item(page, off);	/* {startPage, insertPage, center} */
			int			cdim;
			const void *center = vector_payload(list + 8, &cdim);
			double		distance = query == NULL ? 0.0 : ora_index_distance(ops, dtype, dim, center, query);

			if (count < probes)
			{
				/* sift up */
				int			i = count++;

				heap[i].distance = distance;
				heap[i].start_page = rd32(list);
				while (i > 0 && heap[(i - 1) / 2].distance < heap[i].distance)
				{
					page_list	t = heap[i];

					heap[i] = heap[(i - 1) / 2];
					heap[(i - 1) / 2] = t;
					i = (i - 1) / 2;
				}
				if (count == probes)
					max_distance = heap[0].distance;
			}
			else if (distance < max_distance)	/* strict: src/ivfscan.c:92 */
			{
				heap[0].distance = distance;
				heap[0].start_page = rd32(list);
				sift_down(heap, count, 0);
				max_distance = heap[0].distance;
			}
		}
		blk = next_block(page);
	}
	/* ascending probe order (src/ivfscan.c:114-115) */
	{
		page_list  *sorted = malloc(sizeof(page_list) * (size_t) (count > 0 ? count : 1));

		for (int n = count; n > 0; n--)
		{
			sorted[n - 1] = heap[0];
			heap[0] = heap[n - 1];
			sift_down(heap, n - 1, 0);
		}
		free(heap);
		heap = sorted;
	}

	/* GetScanItems: the entry-page chain of every probed list */
	for (int p = 0; p < count; p++)
		for (blk = heap[p].start_page; blk != ORA_INVALID_BLOCK && blk < nblocks;)
		{
			const uint8_t *page = page_of(pages, blk);
			int			maxoff = max_offset(page);

			for (int off = 1; off <= maxoff; off++)
			{
				const uint8_t *itup = get_item(page, off);	/* IndexTupleData: t_tid (6 bytes), t_info (2) */
				int			vdim;
				const void *vec = vector_payload(itup + 8, &vdim);	/* index_getattr(itup, 1, ...) */

				if (nitems == cap)
				{
					cap = cap ? cap * 2 : 4096;
					items = realloc(items, sizeof(page_item) * (size_t) cap);
				}
				items[nitems].distance = query == NULL ? 0.0 : ora_index_distance(ops, dtype, dim, vec, query);
				/* ItemPointerData: bi_hi, bi_lo, ip_posid -> (block << 16) | offset */
				items[nitems].tid = ((uint64_t) (((uint32_t) rd16(itup) << 16) | rd16(itup + 2)) << 16) | rd16(itup + 4);
				items[nitems].seq = nitems;
				nitems++;
			}
			blk = next_block(page);
		}
	free(heap);
	if (out_scanned)
		*out_scanned = nitems;
	qsort(items, (size_t) nitems, sizeof(page_item), item_cmp);	/* tuplesort_performsort */
	nout = nitems < k ? (int) nitems : k;
	for (int i = 0; i < nout; i++)
	{
		out_tids[i] = items[i].tid;
		out_dist[i] = items[i].distance;
	}
	free(items);
	return nout;
}
