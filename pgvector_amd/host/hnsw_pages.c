/*
 * hnsw_pages.c -- the HNSW index in its on-disk form, either side of the device path:
 *
 *   pgv_host_hnsw_write_index   FlushPages (src/hnswbuild.c:300-312): CreateMetaPage (:88-117),
 *                               CreateGraphPages (:150-250) and WriteNeighborTuples (:252-296)
 *                               from a built graph (pgv_host_hnsw_build's arrays)
 *   pgv_host_hnsw_stage         the walk a scan does one tuple at a time (HnswLoadElement,
 *                               src/hnswutils.c:533-571; HnswLoadNeighborTids, :761-794), done
 *                               once: element pages -> dense slots, vectors, levels, heap TIDs
 *                               and the neighbor table pgv_hnsw_set_graph takes
 *
 * Layout (src/hnsw.h:40-47, 334-392): block 0 = meta page {magic 0xA953A953, version 1,
 * dimensions, m, efConstruction, entry (blkno, offno, level), insertPage}; blocks >= 1 = element
 * pages chained by nextblkno, page id 0xFF90; an element tuple is {type 1, level, deleted,
 * version, 10 heap TIDs, neighbor TID, unused, Vector/HalfVector varlena at byte 72}, a neighbor
 * tuple {type 2, version, count, (level + 2) * m index TIDs}; both MAXALIGNed; an element and
 * its neighbor tuple share a page when they fit together.
 */
#include "pgv_host.h"

#include <stdlib.h>
#include <string.h>

extern int	pgv_host_fail(int code, const char *fmt,...);

#define PAGE_HEADER_SIZE 24		/* SizeOfPageHeaderData */
#define ITEMID_SIZE 4
#define SPECIAL_SIZE 8			/* MAXALIGN(sizeof(HnswPageOpaqueData)) */
#define MAXALIGN8(x) (((size_t) (x) + 7) & ~(size_t) 7)
#define HNSW_MAGIC_NUMBER 0xA953A953u
#define HNSW_VERSION 1
#define HNSW_PAGE_ID 0xFF90
#define HNSW_HEAD_BLKNO 1
#define HNSW_HEAPTIDS 10
#define HNSW_ELEMENT_TUPLE_TYPE 1
#define HNSW_NEIGHBOR_TUPLE_TYPE 2
#define ELEMENT_DATA_OFFSET 72	/* offsetof(HnswElementTupleData, data) */
#define NEIGHBOR_TIDS_OFFSET 4	/* offsetof(HnswNeighborTupleData, indextids) */
#define TID_SIZE 6				/* sizeof(ItemPointerData) */
#define HNSW_MAX_SIZE (PGV_BLCKSZ - PAGE_HEADER_SIZE - SPECIAL_SIZE - ITEMID_SIZE)
#define LP_NORMAL 1

typedef struct
{
	uint64_t	pd_lsn;
	uint16_t	pd_checksum;
	uint16_t	pd_flags;
	uint16_t	pd_lower;
	uint16_t	pd_upper;
	uint16_t	pd_special;
	uint16_t	pd_pagesize_version;
	uint32_t	pd_prune_xid;
}			page_header;

typedef struct
{
	uint32_t	nextblkno;
	uint16_t	unused;
	uint16_t	page_id;
}			hnsw_opaque;		/* HnswPageOpaqueData */

typedef struct
{
	uint32_t	magicNumber;
	uint32_t	version;
	uint32_t	dimensions;
	uint16_t	m;
	uint16_t	efConstruction;
	uint32_t	entryBlkno;
	uint16_t	entryOffno;
	int16_t		entryLevel;
	uint32_t	insertPage;
}			hnsw_meta;			/* HnswMetaPageData */

static inline uint8_t *
page_at(const pgv_rel * rel, uint32_t blk)
{
	return rel->pages + (size_t) blk * PGV_BLCKSZ;
}

static inline hnsw_opaque *
page_opaque(uint8_t *page)
{
	return (hnsw_opaque *) (page + ((page_header *) page)->pd_special);
}

static inline int
page_max_offset(const uint8_t *page)
{
	int			lower = ((const page_header *) page)->pd_lower;

	return lower <= PAGE_HEADER_SIZE ? 0 : (lower - PAGE_HEADER_SIZE) / ITEMID_SIZE;
}

static inline size_t
page_free_space(const uint8_t *page)
{
	const page_header *h = (const page_header *) page;
	int			space = (int) h->pd_upper - (int) h->pd_lower;

	return space < ITEMID_SIZE ? 0 : (size_t) (space - ITEMID_SIZE);	/* PageGetFreeSpace */
}

static inline uint8_t *
page_item(const uint8_t *page, int offno, int *len)
{
	uint32_t	lp;

	memcpy(&lp, page + PAGE_HEADER_SIZE + (size_t) (offno - 1) * ITEMID_SIZE, 4);
	if (len)
		*len = (int) (lp >> 17);
	return (uint8_t *) page + (lp & 0x7FFF);
}

static int
page_add_item(uint8_t *page, const void *item, size_t size)
{
	page_header *h = (page_header *) page;
	size_t		aligned = MAXALIGN8(size);
	int			offno = page_max_offset(page) + 1;
	uint32_t	lp;

	if ((size_t) h->pd_lower + ITEMID_SIZE > (size_t) h->pd_upper - aligned)
		return 0;
	h->pd_upper = (uint16_t) (h->pd_upper - aligned);
	memcpy(page + h->pd_upper, item, size);
	lp = (uint32_t) h->pd_upper | ((uint32_t) LP_NORMAL << 15) | ((uint32_t) size << 17);
	memcpy(page + h->pd_lower, &lp, 4);
	h->pd_lower = (uint16_t) (h->pd_lower + ITEMID_SIZE);
	return offno;
}

/* HnswNewBuffer + HnswInitPage, src/hnswutils.c:174-199 */
static uint32_t
rel_new_page(pgv_rel * rel)
{
	uint8_t    *page;
	page_header *h;

	if (rel->nblocks == rel->cap)
	{
		rel->cap = rel->cap ? rel->cap * 2 : 64;
		rel->pages = realloc(rel->pages, (size_t) rel->cap * PGV_BLCKSZ);
	}
	page = page_at(rel, rel->nblocks);
	memset(page, 0, PGV_BLCKSZ);
	h = (page_header *) page;
	h->pd_lower = PAGE_HEADER_SIZE;
	h->pd_special = PGV_BLCKSZ - SPECIAL_SIZE;
	h->pd_upper = h->pd_special;
	h->pd_pagesize_version = PGV_BLCKSZ | 4;
	page_opaque(page)->nextblkno = PGV_INVALID_BLOCK;
	page_opaque(page)->page_id = HNSW_PAGE_ID;
	return rel->nblocks++;
}

static inline void
tid_set(uint8_t *dst, uint32_t blkno, uint16_t offno)
{
	uint16_t	hi = (uint16_t) (blkno >> 16),
				lo = (uint16_t) blkno;

	memcpy(dst + 0, &hi, 2);
	memcpy(dst + 2, &lo, 2);
	memcpy(dst + 4, &offno, 2);
}

static inline void
tid_get(const uint8_t *src, uint32_t *blkno, uint16_t *offno)
{
	uint16_t	hi,
				lo;

	memcpy(&hi, src + 0, 2);
	memcpy(&lo, src + 2, 2);
	memcpy(offno, src + 4, 2);
	*blkno = ((uint32_t) hi << 16) | lo;
}

/* heap TID as the host glue passes it around: (block << 16) | offset */
static inline void
heaptid_set(uint8_t *dst, uint64_t tid)
{
	tid_set(dst, (uint32_t) (tid >> 16), (uint16_t) tid);
}

static inline size_t
elem_bytes(pgv_dtype t)
{
	return t == PGV_F32 ? 4 : 2;
}

/* ------------------------------------------------------------------ writer */

int
pgv_host_hnsw_write_index(pgv_rel * rel, pgv_dtype dtype, int dim, int m, int ef_construction, int64_t n,
						  const void *vectors, const uint64_t *tids, const int32_t *levels,
						  const int64_t *nbr_start, const int32_t *nbr, const int32_t *dup_of, int32_t entry)
{
	const size_t es = elem_bytes(dtype);
	const size_t varsize = 8 + (size_t) dim * es;	/* VECTOR_SIZE / HALFVEC_SIZE */
	const size_t etup_size = MAXALIGN8(ELEMENT_DATA_OFFSET + varsize);
	uint8_t    *etup,
			   *ntup;
	uint32_t   *blkno = NULL,
			   *nblkno = NULL;
	uint16_t   *offno = NULL,
			   *noffno = NULL;
	uint8_t    *nheap = NULL;
	uint32_t	blk;
	hnsw_meta	meta;

	if (!rel || (n > 0 && (!vectors || !tids || !levels || !nbr_start || !nbr)))
		return pgv_host_fail(PGV_ERR_ARG, "pgv_host_hnsw_write_index: NULL argument");
	if (etup_size > PGV_BLCKSZ || etup_size > HNSW_MAX_SIZE)
		return pgv_host_fail(PGV_ERR_DIMS, "index tuple too large (max 2000 / 4000 dimensions, src/hnsw.h:33)");
	pgv_rel_free(rel);
	etup = calloc(1, PGV_BLCKSZ);
	ntup = calloc(1, PGV_BLCKSZ);
	blkno = malloc(sizeof(uint32_t) * (size_t) (n > 0 ? n : 1));
	offno = malloc(sizeof(uint16_t) * (size_t) (n > 0 ? n : 1));
	nblkno = malloc(sizeof(uint32_t) * (size_t) (n > 0 ? n : 1));
	noffno = malloc(sizeof(uint16_t) * (size_t) (n > 0 ? n : 1));
	nheap = calloc((size_t) (n > 0 ? n : 1), 1);

	/* CreateMetaPage, src/hnswbuild.c:88-117 */
	blk = rel_new_page(rel);
	memset(&meta, 0, sizeof(meta));
	meta.magicNumber = HNSW_MAGIC_NUMBER;
	meta.version = HNSW_VERSION;
	meta.dimensions = (uint32_t) dim;
	meta.m = (uint16_t) m;
	meta.efConstruction = (uint16_t) ef_construction;
	meta.entryBlkno = PGV_INVALID_BLOCK;
	meta.entryOffno = 0;
	meta.entryLevel = -1;
	meta.insertPage = PGV_INVALID_BLOCK;
	((page_header *) page_at(rel, 0))->pd_lower = (uint16_t) (PAGE_HEADER_SIZE + sizeof(meta));

	/* CreateGraphPages, :150-250: the in-memory list is newest first (head insertion, :373-374) */
	blk = rel_new_page(rel);
	for (int64_t e = n - 1; e >= 0; e--)
	{
		size_t		ntup_size;
		size_t		combined;
		uint8_t    *page;
		int			heap_n = 0;

		if (dup_of && dup_of[e] >= 0)
			continue;			/* its heap TID lives in another element's tuple */
		ntup_size = MAXALIGN8(NEIGHBOR_TIDS_OFFSET + (size_t) TID_SIZE * (size_t) (levels[e] + 2) * m);
		combined = etup_size + ntup_size + ITEMID_SIZE;

		/* HnswSetElementTuple, src/hnswutils.c:434-450 */
		memset(etup, 0, etup_size);
		etup[0] = HNSW_ELEMENT_TUPLE_TYPE;
		etup[1] = (uint8_t) levels[e];
		etup[2] = 0;			/* deleted */
		etup[3] = 1;			/* version, HnswInitElement :261 */
		heaptid_set(etup + 4, tids[e]);
		heap_n = 1;
		for (int i = heap_n; i < HNSW_HEAPTIDS; i++)
			tid_set(etup + 4 + (size_t) i * TID_SIZE, PGV_INVALID_BLOCK, 0);	/* ItemPointerSetInvalid */
		{
			uint32_t	vl_len = (uint32_t) (varsize << 2);	/* SET_VARSIZE, little endian 4-byte header */
			int16_t		d = (int16_t) dim,
						unused = 0;
			uint8_t    *v = etup + ELEMENT_DATA_OFFSET;

			memcpy(v, &vl_len, 4);
			memcpy(v + 4, &d, 2);
			memcpy(v + 6, &unused, 2);
			memcpy(v + 8, (const char *) vectors + (size_t) e * dim * es, (size_t) dim * es);
		}

		/* keep element and neighbors on the same page if possible (:198-200) */
		page = page_at(rel, blk);
		if (page_free_space(page) < etup_size || (combined <= HNSW_MAX_SIZE && page_free_space(page) < combined))
		{
			uint32_t	nb = rel_new_page(rel);

			page_opaque(page_at(rel, blk))->nextblkno = nb;
			blk = nb;
			page = page_at(rel, blk);
		}
		blkno[e] = blk;
		offno[e] = (uint16_t) (page_max_offset(page) + 1);
		if (combined <= HNSW_MAX_SIZE)
		{
			nblkno[e] = blk;
			noffno[e] = (uint16_t) (offno[e] + 1);
		}
		else
		{
			nblkno[e] = blk + 1;
			noffno[e] = 1;
		}
		tid_set(etup + 4 + HNSW_HEAPTIDS * TID_SIZE, nblkno[e], noffno[e]);
		if (page_add_item(page, etup, etup_size) != offno[e])
			goto fail_item;
		if (page_free_space(page) < ntup_size)
		{
			uint32_t	nb = rel_new_page(rel);

			page_opaque(page_at(rel, blk))->nextblkno = nb;
			blk = nb;
			page = page_at(rel, blk);
		}
		/* placeholder for the neighbors (:243-245) */
		memset(ntup, 0, ntup_size);
		if (page_add_item(page, ntup, ntup_size) != noffno[e])
			goto fail_item;
		nheap[e] = (uint8_t) heap_n;
	}

	/* the heap TIDs of duplicates join their element's tuple (AddDuplicateInMemory, :313-331) */
	if (dup_of)
		for (int64_t r = 0; r < n; r++)
			if (dup_of[r] >= 0)
			{
				int32_t		e = dup_of[r];
				uint8_t    *tup = page_item(page_at(rel, blkno[e]), offno[e], NULL);

				if (nheap[e] >= HNSW_HEAPTIDS)
				{
					free(etup);
					free(ntup);
					free(blkno);
					free(offno);
					free(nblkno);
					free(noffno);
					free(nheap);
					return pgv_host_fail(PGV_ERR_ARG, "element %d has more than %d heap TIDs", (int) e, HNSW_HEAPTIDS);
				}
				heaptid_set(tup + 4 + (size_t) nheap[e] * TID_SIZE, tids[r]);
				nheap[e]++;
			}

	/* WriteNeighborTuples, :252-296 with HnswSetNeighborTuple, src/hnswutils.c:456-485 */
	for (int64_t e = n - 1; e >= 0; e--)
	{
		int			count;
		uint8_t    *tup;
		uint16_t	c16;

		if (dup_of && dup_of[e] >= 0)
			continue;
		count = (levels[e] + 2) * m;
		tup = page_item(page_at(rel, nblkno[e]), noffno[e], NULL);
		tup[0] = HNSW_NEIGHBOR_TUPLE_TYPE;
		tup[1] = 1;				/* version */
		c16 = (uint16_t) count;
		memcpy(tup + 2, &c16, 2);
		for (int i = 0; i < count; i++)
		{
			int32_t		nb = nbr[nbr_start[e] + i];

			if (nb >= 0)
				tid_set(tup + NEIGHBOR_TIDS_OFFSET + (size_t) i * TID_SIZE, blkno[nb], offno[nb]);
			else
				tid_set(tup + NEIGHBOR_TIDS_OFFSET + (size_t) i * TID_SIZE, PGV_INVALID_BLOCK, 0);
		}
	}

	/* HnswUpdateMetaPage(HNSW_UPDATE_ENTRY_ALWAYS, entryPoint, insertPage), :247 */
	if (entry >= 0)
	{
		meta.entryBlkno = blkno[entry];
		meta.entryOffno = offno[entry];
		meta.entryLevel = (int16_t) levels[entry];
	}
	meta.insertPage = blk;
	memcpy(page_at(rel, 0) + PAGE_HEADER_SIZE, &meta, sizeof(meta));
	rel->generation++;
	free(etup);
	free(ntup);
	free(blkno);
	free(offno);
	free(nblkno);
	free(noffno);
	free(nheap);
	return PGV_OK;

fail_item:
	free(etup);
	free(ntup);
	free(blkno);
	free(offno);
	free(nblkno);
	free(noffno);
	free(nheap);
	return pgv_host_fail(PGV_ERR_STATE, "failed to add index item");
}

/* ------------------------------------------------------------------ stager */

void
pgv_host_hnsw_image_free(pgv_hnsw_image * img)
{
	if (!img)
		return;
	free(img->vectors);
	free(img->levels);
	free(img->nbr_start);
	free(img->nbr);
	free(img->heaptids);
	free(img->element_tids);
	memset(img, 0, sizeof(*img));
}

int
pgv_host_hnsw_stage(const pgv_rel * rel, pgv_dtype dtype, pgv_hnsw_image * out)
{
	const size_t es = elem_bytes(dtype);
	hnsw_meta	meta;
	int64_t    *first = NULL;	/* per block: index of its first item in slot_of */
	int32_t    *slot_of = NULL;	/* per (block, offset): element slot or -1 */
	int64_t		nitems = 0,
				n = 0,
				total = 0;
	int			rc = PGV_OK;

	if (!rel || !out)
		return pgv_host_fail(PGV_ERR_ARG, "pgv_host_hnsw_stage: rel/out is NULL");
	memset(out, 0, sizeof(*out));
	if (rel->nblocks < 1)
		return pgv_host_fail(PGV_ERR_STATE, "not an hnsw index");
	memcpy(&meta, page_at(rel, 0) + PAGE_HEADER_SIZE, sizeof(meta));
	if (meta.magicNumber != HNSW_MAGIC_NUMBER)
		return pgv_host_fail(PGV_ERR_STATE, "hnsw index is not valid");	/* src/hnswutils.c:308-309 */
	out->dtype = dtype;
	out->dim = (int) meta.dimensions;
	out->m = meta.m;
	out->ef_construction = meta.efConstruction;
	out->entry = -1;

	/* pass 1: every element tuple gets a dense slot in page order */
	first = malloc(sizeof(int64_t) * ((size_t) rel->nblocks + 1));
	for (uint32_t b = 0; b < rel->nblocks; b++)
	{
		first[b] = nitems;
		nitems += b == 0 ? 0 : page_max_offset(page_at(rel, b));
	}
	first[rel->nblocks] = nitems;
	slot_of = malloc(sizeof(int32_t) * (size_t) (nitems > 0 ? nitems : 1));
	for (uint32_t b = HNSW_HEAD_BLKNO; b < rel->nblocks; b++)
	{
		const uint8_t *page = page_at(rel, b);
		int			maxoff = page_max_offset(page);

		if (page_opaque((uint8_t *) page)->page_id != HNSW_PAGE_ID)
		{
			rc = pgv_host_fail(PGV_ERR_STATE, "block %u is not an hnsw page", b);
			goto done;
		}
		for (int off = 1; off <= maxoff; off++)
		{
			const uint8_t *tup = page_item(page, off, NULL);

			slot_of[first[b] + off - 1] = -1;
			if (tup[0] == HNSW_ELEMENT_TUPLE_TYPE && !tup[2])	/* not deleted */
			{
				slot_of[first[b] + off - 1] = (int32_t) n++;
				total += (int64_t) (tup[1] + 2) * out->m;
			}
		}
	}
	out->n = n;
	out->vectors = malloc((size_t) (n > 0 ? n : 1) * out->dim * es);
	out->levels = malloc(sizeof(int32_t) * (size_t) (n > 0 ? n : 1));
	out->nbr_start = malloc(sizeof(int64_t) * (size_t) (n + 1));
	out->nbr = malloc(sizeof(int32_t) * (size_t) (total > 0 ? total : 1));
	out->heaptids = malloc(sizeof(uint64_t) * (size_t) (n > 0 ? n : 1) * HNSW_HEAPTIDS);
	out->element_tids = malloc(sizeof(uint64_t) * (size_t) (n > 0 ? n : 1));

	/* pass 2: payloads and neighbor tuples */
	{
		int64_t		slot = 0,
					o = 0;

		for (uint32_t b = HNSW_HEAD_BLKNO; b < rel->nblocks; b++)
		{
			const uint8_t *page = page_at(rel, b);
			int			maxoff = page_max_offset(page);

			for (int off = 1; off <= maxoff; off++)
			{
				const uint8_t *tup = page_item(page, off, NULL);
				const uint8_t *v,
						   *ntup;
				uint32_t	nblk;
				uint16_t	noff;
				int16_t		d;
				int			level,
							count;
				uint16_t	c16;

				if (slot_of[first[b] + off - 1] < 0)
					continue;
				level = tup[1];
				/* HnswLoadElementFromTuple, src/hnswutils.c:490-520 */
				for (int i = 0; i < HNSW_HEAPTIDS; i++)
				{
					uint32_t	hb;
					uint16_t	ho;

					tid_get(tup + 4 + (size_t) i * TID_SIZE, &hb, &ho);
					out->heaptids[slot * HNSW_HEAPTIDS + i] = ho == 0 ? UINT64_MAX : (((uint64_t) hb << 16) | ho);
				}
				tid_get(tup + 4 + HNSW_HEAPTIDS * TID_SIZE, &nblk, &noff);
				v = tup + ELEMENT_DATA_OFFSET;
				memcpy(&d, v + 4, 2);
				if (d != out->dim)
				{
					rc = pgv_host_fail(PGV_ERR_STATE, "element has %d dimensions, the index %d", (int) d, out->dim);
					goto done;
				}
				memcpy((char *) out->vectors + (size_t) slot * out->dim * es, v + 8, (size_t) out->dim * es);
				out->levels[slot] = level;
				out->element_tids[slot] = ((uint64_t) b << 16) | (uint64_t) off;
				out->nbr_start[slot] = o;
				/* HnswLoadNeighborTids, :761-794 (all layers at once) */
				if (nblk >= rel->nblocks || noff < 1 || noff > page_max_offset(page_at(rel, nblk)))
				{
					rc = pgv_host_fail(PGV_ERR_STATE, "dangling neighbor tuple pointer");
					goto done;
				}
				ntup = page_item(page_at(rel, nblk), noff, NULL);
				memcpy(&c16, ntup + 2, 2);
				count = (level + 2) * out->m;
				if (ntup[0] != HNSW_NEIGHBOR_TUPLE_TYPE || c16 != count)
				{
					/* neighbors being rewritten by a concurrent insert: treated as none (:779-782) */
					for (int i = 0; i < count; i++)
						out->nbr[o + i] = -1;
				}
				else
				{
					/* layer slices of the tuple: level .. 1 with m entries each, then layer 0 with 2 m */
					for (int s0 = 0; s0 < count;)
					{
						int			len = s0 < level * out->m ? out->m : 2 * out->m;
						int			kept = 0,
									ended = 0;

						for (int i = 0; i < len; i++)
						{
							uint32_t	eb;
							uint16_t	eo;
							int32_t		nb = -1;

							tid_get(ntup + NEIGHBOR_TIDS_OFFSET + (size_t) (s0 + i) * TID_SIZE, &eb, &eo);
							if (ended || eo == 0 || eb >= rel->nblocks || eo > page_max_offset(page_at(rel, eb)))
								ended = 1;	/* ItemPointerIsValid fails: end of the layer's list (:785-786) */
							else
								nb = slot_of[first[eb] + eo - 1];
							/* a valid TID whose tuple is no live element (vacuumed away) is dropped and the rest
							 * moves up: a -1 ends the slice for the host walk and the device kernel alike */
							if (nb >= 0)
								out->nbr[o + s0 + kept++] = nb;
						}
						for (int i = kept; i < len; i++)
							out->nbr[o + s0 + i] = -1;
						s0 += len;
					}
				}
				o += count;
				if (meta.entryBlkno == b && meta.entryOffno == off)
					out->entry = (int32_t) slot;
				slot++;
			}
		}
		out->nbr_start[n] = o;
	}

done:
	free(first);
	free(slot_of);
	if (rc != PGV_OK)
		pgv_host_hnsw_image_free(out);
	return rc;
}
