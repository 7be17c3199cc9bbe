/*
 * ivf_pages.c -- IVFFlat on-disk format, page writers, staging and the scan/build
 * drivers that sit between Postgres pages and the libpgv_hip ABI.
 *
 * The page layout is PostgreSQL's (bufpage.h: 24-byte header, 4-byte line
 * pointers growing up, tuples growing down, special space at the end) with
 * pgvector's special area and tuples (src/ivfflat.h:251-275).  Everything a
 * maintainer needs inside the extension is here in plain C; Postgres itself is
 * replaced by a pgv_rel (array of pages).  No distance is computed on the CPU.
 */
#include "pgv_host.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

extern int	pgv_host_fail(int code, const char *fmt,...);

#include <sys/mman.h>
#include <time.h>

/*
 * Gigabytes that are about to be written once (the page image, the rows sorted by list): 2 MB-aligned and
 * marked for transparent huge pages, so that the first touch costs one fault per 2 MB instead of per 4 KB --
 * where THP is off (or the request is small) this is plain malloc.  Released with free().
 */
void *
pgv_host_big_alloc(size_t bytes)
{
	void	   *p = NULL;

	if (bytes < ((size_t) 64 << 20))
		return malloc(bytes);
	if (posix_memalign(&p, (size_t) 2 << 20, bytes) != 0)
		return NULL;
	(void) madvise(p, bytes, MADV_HUGEPAGE);
	return p;
}

/* ------------------------------------------------------------- page layout */

#define PAGE_HEADER_SIZE 24		/* SizeOfPageHeaderData */
#define ITEMID_SIZE 4			/* sizeof(ItemIdData) */
#define SPECIAL_SIZE 8			/* MAXALIGN(sizeof(IvfflatPageOpaqueData)) */
#define MAXALIGN8(x) (((size_t) (x) + 7) & ~(size_t) 7)
#define INDEX_TUPLE_HEADER 8	/* sizeof(IndexTupleData): 6-byte t_tid + 2-byte t_info */
#define IVFFLAT_MAGIC_NUMBER 0x14FF1A7	/* src/ivfflat.h:47 */
#define IVFFLAT_VERSION 1
#define IVFFLAT_PAGE_ID 0xFF84	/* src/ivfflat.h:48 */
#define IVFFLAT_HEAD_BLKNO 1
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
}			ivf_opaque;			/* IvfflatPageOpaqueData */

typedef struct
{
	uint32_t	magicNumber;
	uint32_t	version;
	uint16_t	dimensions;
	uint16_t	lists;
}			ivf_meta;			/* IvfflatMetaPageData */

static inline size_t
elem_bytes(pgv_dtype t)
{
	return t == PGV_F32 ? 4 : 2;
}

/* VECTOR_SIZE / HALFVEC_SIZE: 4-byte varlena header + int16 dim + int16 unused + payload */
static inline size_t
varlena_size(pgv_dtype t, int dim)
{
	return 8 + (size_t) dim * elem_bytes(t);
}

static inline uint8_t *
page_at(const pgv_rel * rel, uint32_t blk)
{
	return rel->pages + (size_t) blk * PGV_BLCKSZ;
}

static inline ivf_opaque *
page_opaque(uint8_t *page)
{
	return (ivf_opaque *) (page + ((page_header *) page)->pd_special);
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
		*len = (int) (lp >> 17);	/* lp_off:15 | lp_flags:2 | lp_len:15 */
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
		return 0;				/* InvalidOffsetNumber */
	h->pd_upper = (uint16_t) (h->pd_upper - aligned);
	memcpy(page + h->pd_upper, item, size);
	lp = (uint32_t) h->pd_upper | ((uint32_t) LP_NORMAL << 15) | ((uint32_t) size << 17);
	memcpy(page + h->pd_lower, &lp, 4);
	h->pd_lower = (uint16_t) (h->pd_lower + ITEMID_SIZE);
	return offno;
}

void
pgv_rel_init(pgv_rel * rel)
{
	rel->generation = 0;
	rel->pages = NULL;
	rel->nblocks = rel->cap = 0;
}

void
pgv_rel_free(pgv_rel * rel)
{
	uint64_t	generation = rel->generation;

	free(rel->pages);
	pgv_rel_init(rel);
	rel->generation = generation + 1;	/* whatever was staged from the old pages is stale */
}

/* IvfflatInitPage, src/ivfutils.c:148-156 */
static void
init_page(uint8_t *page)
{
	page_header *h = (page_header *) page;

	memset(page, 0, PGV_BLCKSZ);
	h->pd_lower = PAGE_HEADER_SIZE;
	h->pd_special = PGV_BLCKSZ - SPECIAL_SIZE;
	h->pd_upper = h->pd_special;
	h->pd_pagesize_version = PGV_BLCKSZ | 4;	/* PG_PAGE_LAYOUT_VERSION */
	page_opaque(page)->nextblkno = PGV_INVALID_BLOCK;
	page_opaque(page)->page_id = IVFFLAT_PAGE_ID;
}

/* IvfflatNewBuffer + IvfflatInitPage, src/ivfutils.c:135-156 */
static uint32_t
rel_new_page(pgv_rel * rel)
{
	if (rel->nblocks == rel->cap)
	{
		rel->cap = rel->cap ? rel->cap * 2 : 64;
		rel->pages = realloc(rel->pages, (size_t) rel->cap * PGV_BLCKSZ);
	}
	init_page(page_at(rel, rel->nblocks));
	return rel->nblocks++;
}

/* IvfflatAppendPage, src/ivfutils.c:177-203 */
static uint32_t
rel_append_page(pgv_rel * rel, uint32_t cur)
{
	uint32_t	blk = rel_new_page(rel);

	page_opaque(page_at(rel, cur))->nextblkno = blk;
	return blk;
}

/* a Vector / HalfVector varlena as stored (4-byte header form) */
static void
fill_varlena(uint8_t *dst, pgv_dtype t, int dim, const void *payload)
{
	uint32_t	vl = (uint32_t) varlena_size(t, dim) << 2;	/* SET_VARSIZE, little endian */
	int16_t		d = (int16_t) dim,
				unused = 0;

	memcpy(dst, &vl, 4);
	memcpy(dst + 4, &d, 2);
	memcpy(dst + 6, &unused, 2);
	memcpy(dst + 8, payload, (size_t) dim * elem_bytes(t));
}

/*
 * index_form_tuple for (vector): header + datum.  Values whose total size fits a
 * 1-byte header (<= 127 bytes incl. header, i.e. dim <= 29 for float4) are stored
 * short and unaligned (SURVEY A.3 staging gotcha).
 */
static size_t
form_index_tuple(uint8_t *dst, pgv_dtype t, int dim, const void *payload, uint64_t tid)
{
	size_t		full = varlena_size(t, dim);
	size_t		short_size = full - 4 + 1;
	size_t		size;
	uint16_t	hi = (uint16_t) (tid >> 32),
				lo = (uint16_t) (tid >> 16),
				pos = (uint16_t) tid;
	uint16_t	info;

	memset(dst, 0, INDEX_TUPLE_HEADER);
	memcpy(dst + 0, &hi, 2);
	memcpy(dst + 2, &lo, 2);
	memcpy(dst + 4, &pos, 2);
	if (short_size <= 127)
	{
		int16_t		d = (int16_t) dim,
					unused = 0;

		dst[INDEX_TUPLE_HEADER] = (uint8_t) ((short_size << 1) | 1);
		memcpy(dst + INDEX_TUPLE_HEADER + 1, &d, 2);
		memcpy(dst + INDEX_TUPLE_HEADER + 3, &unused, 2);
		memcpy(dst + INDEX_TUPLE_HEADER + 5, payload, (size_t) dim * elem_bytes(t));
		size = INDEX_TUPLE_HEADER + short_size;
	}
	else
	{
		fill_varlena(dst + INDEX_TUPLE_HEADER, t, dim, payload);
		size = INDEX_TUPLE_HEADER + full;
	}
	size = MAXALIGN8(size);		/* index_form_tuple MAXALIGNs the tuple size */
	info = (uint16_t) (size & 0x1FFF) | 0x4000;	/* INDEX_VAR_MASK: has a varwidth attribute */
	memcpy(dst + 6, &info, 2);
	return size;
}

/* ------------------------------------------------------------------ writers */

/*
 * The index's pages in three steps, so that a build can overlap them with the device's work:
 *
 *   pgv_host_ivf_writer_begin   (before the rows are assigned) the page array for at most max_rows tuples, being
 *                               zeroed -- the first touch of 8 GB of fresh memory (1 M x 1536) is page-fault
 *                               bound -- by background threads while the GPU runs k-means and the assignment
 *   pgv_host_ivf_writer_layout  (list lengths known) CreateMetaPage, CreateListPages (src/ivfbuild.c:485-556), and
 *                               every entry page stamped with its FINAL header, line pointers and chain link: every
 *                               index tuple of one index has the same size, so how many fit a page, each list's page
 *                               range and each tuple's place are known from the lengths alone
 *   pgv_host_ivf_writer_fill    InsertTuples (:271-331) for a run of slots of the sorted stream: the tuples
 *                               themselves, any run in any order, rows in parallel
 *
 * The pages come out as the reference's leader writes them one list after another.
 */
#include <omp.h>
#include <pthread.h>
#include <stdio.h>

/*
 * Threads for the copy loops of this file: what OpenMP would use, capped by the container's CPU quota (cgroup v2
 * cpu.max / v1 cfs quota: a box that shows 256 CPUs but grants 16 makes a 256-thread parallel region cost
 * milliseconds of throttled barrier per region) and by 64 (a memcpy-bound loop gains nothing beyond).
 */
int
pgv_host_threads(void)
{
	static int	cached = 0;
	int			t;
	FILE	   *f;

	if (cached)
		return cached;
	t = omp_get_max_threads();
	if ((f = fopen("/sys/fs/cgroup/cpu.max", "r")) != NULL)
	{
		char		quota[32];
		double		period = 0;

		if (fscanf(f, "%31s %lf", quota, &period) == 2 && quota[0] != 'm' && period > 0)
		{
			int			q = (int) (atof(quota) / period);

			if (q >= 1 && q < t)
				t = q;
		}
		fclose(f);
	}
	else if ((f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) != NULL)
	{
		double		q = -1,
					period = 100000;
		FILE	   *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r");

		if (fscanf(f, "%lf", &q) != 1)
			q = -1;
		if (g)
		{
			if (fscanf(g, "%lf", &period) != 1)
				period = 100000;
			fclose(g);
		}
		fclose(f);
		if (q > 0 && period > 0 && (int) (q / period) >= 1 && (int) (q / period) < t)
			t = (int) (q / period);
	}
	if (t > 64)
		t = 64;
	if (t < 1)
		t = 1;
	cached = t;
	return t;
}

#define PREFAULT_PIECE ((size_t) 2 << 20)
#define PREFAULT_THREADS 16

struct pgv_ivf_writer
{
	pgv_rel    *rel;
	pgv_dtype	dtype;
	int			dim,
				lists;
	size_t		tuple_size;		/* MAXALIGNed size of every index tuple */
	int			per_page;
	uint32_t	cap_blocks;		/* pages allocated */
	uint32_t   *first_blk;		/* [lists + 1] page range of every list (after layout) */
	int64_t    *offsets;		/* [lists + 1] */
	/* prefault */
	pthread_t	threads[PREFAULT_THREADS];
	int			nthreads;
	int			joined;
};

typedef struct
{
	uint8_t    *base;
	size_t		bytes;
	int			t,
				nthreads;
}			prefault_job;

static void *
prefault_main(void *arg)
{
	prefault_job *j = arg;

	for (size_t off = (size_t) j->t * PREFAULT_PIECE; off < j->bytes; off += (size_t) j->nthreads * PREFAULT_PIECE)
		memset(j->base + off, 0, j->bytes - off < PREFAULT_PIECE ? j->bytes - off : PREFAULT_PIECE);
	free(j);
	return NULL;
}

static double writer_wait_secs;

/* how long the last writer's layout step waited for the background zeroing of the page array (diagnostics) */
double
pgv_host_ivf_writer_wait_secs(void)
{
	return writer_wait_secs;
}

static void
writer_join(pgv_ivf_writer * w)
{
	if (!w->joined)
	{
		struct timespec a,
					b;

		clock_gettime(CLOCK_MONOTONIC, &a);
		for (int t = 0; t < w->nthreads; t++)
			pthread_join(w->threads[t], NULL);
		clock_gettime(CLOCK_MONOTONIC, &b);
		writer_wait_secs = (double) (b.tv_sec - a.tv_sec) + 1e-9 * (double) (b.tv_nsec - a.tv_nsec);
	}
	w->joined = 1;
}

/* how many index tuples of this index fit a page (PageAddItem until it refuses) and their size */
static int
tuples_per_page(pgv_dtype dtype, int dim, size_t *tuple_size)
{
	uint8_t    *probe = calloc(1, PGV_BLCKSZ);
	uint8_t    *item = calloc(1, PGV_BLCKSZ);
	page_header *h = (page_header *) probe;
	int			per_page = 0;
	void	   *payload = calloc((size_t) dim, elem_bytes(dtype));

	*tuple_size = form_index_tuple(item, dtype, dim, payload, 0);
	h->pd_lower = PAGE_HEADER_SIZE;
	h->pd_special = PGV_BLCKSZ - SPECIAL_SIZE;
	h->pd_upper = h->pd_special;
	while (page_free_space(probe) >= *tuple_size && page_add_item(probe, item, *tuple_size))
		per_page++;
	free(payload);
	free(item);
	free(probe);
	return per_page;
}

/* pages the list tuples need (CreateListPages appends a page whenever the next list tuple does not fit) */
static uint32_t
list_pages_needed(pgv_dtype dtype, int dim, int lists)
{
	size_t		list_size = MAXALIGN8(8 + varlena_size(dtype, dim));
	size_t		room = PGV_BLCKSZ - PAGE_HEADER_SIZE - SPECIAL_SIZE;
	int			per = (int) (room / (list_size + ITEMID_SIZE));

	return per < 1 ? 0 : (uint32_t) ((lists + per - 1) / per);
}

int
pgv_host_ivf_writer_begin(pgv_rel * rel, pgv_dtype dtype, int dim, int lists, int64_t max_rows, pgv_ivf_writer * *out)
{
	pgv_ivf_writer *w;
	size_t		list_size = MAXALIGN8(8 + varlena_size(dtype, dim));
	uint64_t	blocks;
	size_t		bytes;

	*out = NULL;
	pgv_rel_free(rel);
	if (list_size + ITEMID_SIZE > PGV_BLCKSZ - PAGE_HEADER_SIZE - SPECIAL_SIZE)
		return pgv_host_fail(PGV_ERR_DIMS, "vector does not fit an 8 KB page (max 2000 / 4000 dimensions)");
	w = calloc(1, sizeof(*w));
	if (!w)
		return pgv_host_fail(PGV_ERR_NOMEM, "out of memory");
	w->rel = rel;
	w->dtype = dtype;
	w->dim = dim;
	w->lists = lists;
	w->per_page = tuples_per_page(dtype, dim, &w->tuple_size);
	if (w->per_page < 1)
	{
		free(w);
		return pgv_host_fail(PGV_ERR_STATE, "failed to add index item");
	}
	/* meta + list pages + at most one partly filled page per list beyond the full ones */
	blocks = 1 + (uint64_t) list_pages_needed(dtype, dim, lists) + (uint64_t) (max_rows / w->per_page) + (uint64_t) lists + 1;
	if (blocks >= PGV_INVALID_BLOCK)
	{
		free(w);
		return pgv_host_fail(PGV_ERR_ARG, "index too large for 32-bit block numbers");
	}
	bytes = (size_t) blocks * PGV_BLCKSZ;
	rel->pages = pgv_host_big_alloc(bytes);
	if (!rel->pages)
	{
		free(w);
		return pgv_host_fail(PGV_ERR_NOMEM, "out of memory for %llu pages", (unsigned long long) blocks);
	}
	rel->cap = (uint32_t) blocks;
	rel->nblocks = 0;
	w->cap_blocks = (uint32_t) blocks;
	/* zero the array (what init_page's memset would do page by page) on threads of its own */
	{
		int			want = bytes >= ((size_t) 64 << 20) ? (pgv_host_threads() < PREFAULT_THREADS ? pgv_host_threads() : PREFAULT_THREADS) : 1;

		w->nthreads = 0;
		for (int t = 0; t < want; t++)
		{
			prefault_job *j = malloc(sizeof(*j));

			if (!j)
			{
				writer_join(w);
				free(w);
				pgv_rel_free(rel);
				return pgv_host_fail(PGV_ERR_NOMEM, "out of memory");
			}
			j->base = rel->pages;
			j->bytes = bytes;
			j->t = t;
			j->nthreads = want;
			if (pthread_create(&w->threads[w->nthreads], NULL, prefault_main, j) == 0)
				w->nthreads++;
			else
				prefault_main(j);	/* no thread to be had: this share is done here */
		}
	}
	*out = w;
	return PGV_OK;
}

/* IvfflatInitPage on zeroed memory: the header and the special space */
static void
stamp_page(uint8_t *page, int count, size_t tuple_size, uint32_t nextblkno)
{
	page_header *h = (page_header *) page;

	h->pd_special = PGV_BLCKSZ - SPECIAL_SIZE;
	h->pd_lower = (uint16_t) (PAGE_HEADER_SIZE + ITEMID_SIZE * count);
	h->pd_upper = (uint16_t) (h->pd_special - (size_t) count * tuple_size);
	h->pd_pagesize_version = PGV_BLCKSZ | 4;	/* PG_PAGE_LAYOUT_VERSION */
	page_opaque(page)->nextblkno = nextblkno;
	page_opaque(page)->page_id = IVFFLAT_PAGE_ID;
	for (int j = 0; j < count; j++)
	{
		uint32_t	upper = (uint32_t) (h->pd_special - (size_t) (j + 1) * tuple_size);
		uint32_t	lp = upper | ((uint32_t) LP_NORMAL << 15) | ((uint32_t) tuple_size << 17);

		memcpy(page + PAGE_HEADER_SIZE + (size_t) j * ITEMID_SIZE, &lp, 4);
	}
}

int
pgv_host_ivf_writer_layout(pgv_ivf_writer * w, const void *centers, const int64_t *list_offsets)
{
	pgv_rel    *rel = w->rel;
	pgv_dtype	dtype = w->dtype;
	int			dim = w->dim,
				lists = w->lists;
	size_t		es = elem_bytes(dtype);
	size_t		list_size = MAXALIGN8(8 + varlena_size(dtype, dim));
	uint8_t    *item = calloc(1, list_size > PGV_BLCKSZ ? list_size : PGV_BLCKSZ);
	uint32_t   *list_blk = malloc(sizeof(uint32_t) * (size_t) lists);
	int		   *list_off = malloc(sizeof(int) * (size_t) lists);
	uint32_t	blk;
	ivf_meta	meta;

	writer_join(w);
	w->first_blk = malloc(sizeof(uint32_t) * ((size_t) lists + 1));
	w->offsets = malloc(sizeof(int64_t) * ((size_t) lists + 1));
	if (!item || !list_blk || !list_off || !w->first_blk || !w->offsets)
	{
		free(item);
		free(list_blk);
		free(list_off);
		return pgv_host_fail(PGV_ERR_NOMEM, "out of memory");
	}
	memcpy(w->offsets, list_offsets, sizeof(int64_t) * ((size_t) lists + 1));

	/* CreateMetaPage, src/ivfbuild.c:485-506 */
	blk = rel_new_page(rel);
	meta.magicNumber = IVFFLAT_MAGIC_NUMBER;
	meta.version = IVFFLAT_VERSION;
	meta.dimensions = (uint16_t) dim;
	meta.lists = (uint16_t) lists;
	memcpy(page_at(rel, blk) + PAGE_HEADER_SIZE, &meta, sizeof(meta));
	((page_header *) page_at(rel, blk))->pd_lower = (uint16_t) (PAGE_HEADER_SIZE + sizeof(meta));

	/* CreateListPages, :511-556 */
	blk = rel_new_page(rel);
	for (int i = 0; i < lists; i++)
	{
		uint32_t	invalid = PGV_INVALID_BLOCK;

		memset(item, 0, list_size);
		memcpy(item + 0, &invalid, 4);	/* startPage */
		memcpy(item + 4, &invalid, 4);	/* insertPage */
		fill_varlena(item + 8, dtype, dim, (const char *) centers + (size_t) i * dim * es);
		if (page_free_space(page_at(rel, blk)) < list_size)
			blk = rel_append_page(rel, blk);
		list_off[i] = page_add_item(page_at(rel, blk), item, list_size);
		list_blk[i] = blk;
	}

	/* every list's page range (an empty list still owns its start page) */
	w->first_blk[0] = rel->nblocks;
	for (int i = 0; i < lists; i++)
	{
		int64_t		len = list_offsets[i + 1] - list_offsets[i];
		int64_t		np = len > 0 ? (len + w->per_page - 1) / w->per_page : 1;

		w->first_blk[i + 1] = w->first_blk[i] + (uint32_t) np;
	}
	if (w->first_blk[lists] > w->cap_blocks)
	{
		free(item);
		free(list_blk);
		free(list_off);
		return pgv_host_fail(PGV_ERR_ARG, "more rows than pgv_host_ivf_writer_begin was told");
	}
	rel->nblocks = w->first_blk[lists];
#pragma omp parallel for schedule(static) num_threads(pgv_host_threads())
	for (int i = 0; i < lists; i++)
	{
		int64_t		len = list_offsets[i + 1] - list_offsets[i];
		uint32_t	last = w->first_blk[i + 1] - 1;
		uint8_t    *list_item;

		for (uint32_t b = w->first_blk[i]; b <= last; b++)
		{
			int64_t		left = len - (int64_t) (b - w->first_blk[i]) * w->per_page;
			int			count = (int) (left > w->per_page ? w->per_page : (left > 0 ? left : 0));

			stamp_page(page_at(rel, b), count, w->tuple_size, b < last ? b + 1 : PGV_INVALID_BLOCK);	/* IvfflatAppendPage links */
		}
		/* IvfflatUpdateList: start and insert page in the list tuple */
		list_item = page_item(page_at(rel, list_blk[i]), list_off[i], NULL);
		memcpy(list_item + 0, &w->first_blk[i], 4);
		memcpy(list_item + 4, &last, 4);
	}
	free(item);
	free(list_blk);
	free(list_off);
	return PGV_OK;
}

int
pgv_host_ivf_writer_fill(pgv_ivf_writer * w, int64_t first_slot, int64_t count, const void *vectors, const uint64_t *tids)
{
	pgv_rel    *rel = w->rel;
	size_t		row_bytes = (size_t) w->dim * elem_bytes(w->dtype);
	int64_t		n = w->offsets ? w->offsets[w->lists] : -1;
	int			lists = w->lists;

	if (n < 0)
		return pgv_host_fail(PGV_ERR_STATE, "pgv_host_ivf_writer_fill before pgv_host_ivf_writer_layout");
	if (first_slot < 0 || count < 0 || first_slot + count > n)
		return pgv_host_fail(PGV_ERR_ARG, "slots [%lld, %lld) outside the %lld rows laid out", (long long) first_slot,
							 (long long) (first_slot + count), (long long) n);
#pragma omp parallel num_threads(pgv_host_threads())
	{
		int			l = 0;

#pragma omp for schedule(static)
		for (int64_t r = 0; r < count; r++)
		{
			int64_t		slot = first_slot + r;
			int64_t		in_list;
			uint8_t    *page;
			int			j;
			size_t		sz;

			/* the list of this slot: slots arrive ascending within a thread's share, so walk; else bisect */
			if (!(w->offsets[l] <= slot && slot < w->offsets[l + 1]))
			{
				int			lo = 0,
							hi = lists - 1;

				while (lo < hi)
				{
					int			mid = (lo + hi + 1) >> 1;

					if (w->offsets[mid] <= slot)
						lo = mid;
					else
						hi = mid - 1;
				}
				l = lo;
				while (w->offsets[l + 1] <= slot)	/* empty lists share their offset with the next one */
					l++;
			}
			in_list = slot - w->offsets[l];
			page = page_at(rel, w->first_blk[l] + (uint32_t) (in_list / w->per_page));
			j = (int) (in_list % w->per_page);
			/* formed in place (the page is zeroed: the tuple's alignment padding stays zero) -- one copy of the payload,
			 * out of the pinned piece the drain handed over, instead of two */
			sz = form_index_tuple(page + (PGV_BLCKSZ - SPECIAL_SIZE) - (size_t) (j + 1) * w->tuple_size, w->dtype, w->dim,
								  (const char *) vectors + (size_t) r * row_bytes, tids ? tids[r] : 0);
			(void) sz;
		}
	}
	return PGV_OK;
}

int
pgv_host_ivf_writer_end(pgv_ivf_writer * w)
{
	if (!w)
		return PGV_OK;
	writer_join(w);
	if (w->rel)
		w->rel->generation++;
	free(w->first_blk);
	free(w->offsets);
	free(w);
	return PGV_OK;
}

int
pgv_host_ivf_write_index(pgv_rel * rel, pgv_dtype dtype, int dim, int lists,
						 const void *centers, const int64_t *list_offsets,
						 const void *vectors, const uint64_t *tids)
{
	pgv_ivf_writer *w = NULL;
	int			rc = pgv_host_ivf_writer_begin(rel, dtype, dim, lists, list_offsets[lists], &w);

	if (rc == PGV_OK)
		rc = pgv_host_ivf_writer_layout(w, centers, list_offsets);
	if (rc == PGV_OK && list_offsets[lists] > 0)
		rc = pgv_host_ivf_writer_fill(w, 0, list_offsets[lists], vectors, tids);
	pgv_host_ivf_writer_end(w);
	if (rc != PGV_OK)
		pgv_rel_free(rel);
	return rc;
}

/* locate list tuple `list` (list pages hold them in id order, src/ivfbuild.c:527-551) */
static uint8_t *
find_list_item(const pgv_rel * rel, int list)
{
	uint32_t	blk = IVFFLAT_HEAD_BLKNO;
	int			seen = 0;

	while (blk != PGV_INVALID_BLOCK && blk < rel->nblocks)
	{
		uint8_t    *page = page_at(rel, blk);
		int			maxoff = page_max_offset(page);

		if (list < seen + maxoff)
			return page_item(page, list - seen + 1, NULL);
		seen += maxoff;
		blk = page_opaque(page)->nextblkno;
	}
	return NULL;
}

/* the append half of ivfflatinsert (src/ivfinsert.c:107-175), list already chosen */
int
pgv_host_ivf_insert(pgv_rel * rel, pgv_dtype dtype, int list, const void *vector, uint64_t tid)
{
	ivf_meta	meta;
	uint8_t    *li;
	uint8_t		item[PGV_BLCKSZ];
	uint32_t	insert_page;
	size_t		sz;

	if (rel->nblocks < 2)
		return pgv_host_fail(PGV_ERR_STATE, "not an ivfflat index");
	memcpy(&meta, page_at(rel, 0) + PAGE_HEADER_SIZE, sizeof(meta));
	li = find_list_item(rel, list);
	if (!li)
		return pgv_host_fail(PGV_ERR_ARG, "list %d not found", list);
	memcpy(&insert_page, li + 4, 4);
	sz = form_index_tuple(item, dtype, meta.dimensions, vector, tid);
	while (page_free_space(page_at(rel, insert_page)) < sz)
	{
		uint32_t	next = page_opaque(page_at(rel, insert_page))->nextblkno;

		if (next == PGV_INVALID_BLOCK)
		{
			ptrdiff_t	delta = li - rel->pages;	/* pages may move when the relation grows */

			next = rel_append_page(rel, insert_page);
			li = rel->pages + delta;
		}
		insert_page = next;
	}
	page_add_item(page_at(rel, insert_page), item, sz);
	memcpy(li + 4, &insert_page, 4);
	rel->generation++;
	return PGV_OK;
}

/* ------------------------------------------------------------------- vacuum */

/* PageIndexMultiDelete: drop the given (ascending) offsets; survivors keep their order */
static void
page_multi_delete(uint8_t *page, const int *deletable, int ndeletable)
{
	uint8_t		copy[PGV_BLCKSZ];
	page_header *h = (page_header *) page;
	int			maxoff = page_max_offset(page);
	int			d = 0;

	memcpy(copy, page, PGV_BLCKSZ);
	h->pd_lower = PAGE_HEADER_SIZE;
	h->pd_upper = h->pd_special;
	for (int offno = 1; offno <= maxoff; offno++)
	{
		int			len;
		uint8_t    *item;

		if (d < ndeletable && deletable[d] == offno)
		{
			d++;
			continue;
		}
		item = page_item(copy, offno, &len);
		page_add_item(page, item, (size_t) len);
	}
}

int
pgv_host_ivf_bulkdelete(pgv_rel * rel, pgv_host_dead_fn dead, void *state,
						int64_t *tuples_removed, int64_t *num_index_tuples)
{
	uint32_t	blkno = IVFFLAT_HEAD_BLKNO;
	int64_t		removed = 0,
				remaining = 0;

	if (rel->nblocks < 2 || !dead)
		return pgv_host_fail(PGV_ERR_STATE, "not an ivfflat index");
	/* iterate over list pages (src/ivfvacuum.c:31-56) */
	while (blkno != PGV_INVALID_BLOCK && blkno < rel->nblocks)
	{
		uint8_t    *cpage = page_at(rel, blkno);
		int			cmaxoff = page_max_offset(cpage);

		for (int coffno = 1; coffno <= cmaxoff; coffno++)
		{
			uint8_t    *list = page_item(cpage, coffno, NULL);
			uint32_t	search_page,
						insert_page = PGV_INVALID_BLOCK;

			memcpy(&search_page, list + 0, 4);
			/* iterate over entry pages (:63-124) */
			while (search_page != PGV_INVALID_BLOCK && search_page < rel->nblocks)
			{
				uint8_t    *page = page_at(rel, search_page);
				int			maxoff = page_max_offset(page);
				int			deletable[PGV_BLCKSZ / ITEMID_SIZE];
				int			ndeletable = 0;

				for (int offno = 1; offno <= maxoff; offno++)
				{
					const uint8_t *itup = page_item(page, offno, NULL);
					uint16_t	bi_hi,
								bi_lo,
								posid;
					uint64_t	tid;

					memcpy(&bi_hi, itup + 0, 2);
					memcpy(&bi_lo, itup + 2, 2);
					memcpy(&posid, itup + 4, 2);
					tid = ((uint64_t) (((uint32_t) bi_hi << 16) | bi_lo) << 16) | posid;
					if (dead(tid, state))
					{
						deletable[ndeletable++] = offno;
						removed++;
					}
					else
						remaining++;
				}
				/* set to first free page; must be set before searchPage is updated (:110-113) */
				if (insert_page == PGV_INVALID_BLOCK && ndeletable > 0)
					insert_page = search_page;
				search_page = page_opaque(page)->nextblkno;
				if (ndeletable > 0)
					page_multi_delete(page, deletable, ndeletable);
			}
			/* IvfflatUpdateList(index, listInfo, insertPage, Invalid, Invalid) (:132-137) */
			if (insert_page != PGV_INVALID_BLOCK)
				memcpy(list + 4, &insert_page, 4);
		}
		blkno = page_opaque(cpage)->nextblkno;
	}
	if (removed > 0)
		rel->generation++;
	if (tuples_removed)
		*tuples_removed = removed;
	if (num_index_tuples)
		*num_index_tuples = remaining;
	return PGV_OK;
}

/* ------------------------------------------------------------------ staging */

/* payload of the vector attribute of an index tuple, either header form */
static const uint8_t *
tuple_vector_payload(const uint8_t *itup, int *dim)
{
	const uint8_t *datum = itup + INDEX_TUPLE_HEADER;
	int16_t		d;

	if (datum[0] & 1)			/* VARATT_IS_1B: short header, unaligned body */
	{
		memcpy(&d, datum + 1, 2);
		*dim = d;
		return datum + 5;
	}
	memcpy(&d, datum + 4, 2);
	*dim = d;
	return datum + 8;
}

int
pgv_host_ivf_stage(const pgv_rel * rel, pgv_dtype dtype, pgv_ivf_image * out)
{
	ivf_meta	meta;
	size_t		es = elem_bytes(dtype);
	size_t		row_bytes;
	int			l = 0;
	int64_t		n = 0,
				cap = 0;
	uint32_t	blk;

	memset(out, 0, sizeof(*out));
	if (rel->nblocks < 2)
		return pgv_host_fail(PGV_ERR_STATE, "not an ivfflat index");
	memcpy(&meta, page_at(rel, 0) + PAGE_HEADER_SIZE, sizeof(meta));
	if (meta.magicNumber != IVFFLAT_MAGIC_NUMBER)
		return pgv_host_fail(PGV_ERR_STATE, "ivfflat index is not valid");	/* src/ivfutils.c:221-222 */
	out->dtype = dtype;
	out->dim = meta.dimensions;
	out->lists = meta.lists;
	row_bytes = (size_t) out->dim * es;
	out->centers = malloc(row_bytes * (size_t) out->lists);
	out->list_offsets = calloc((size_t) out->lists + 1, sizeof(int64_t));
	out->start_pages = malloc(sizeof(uint32_t) * (size_t) out->lists);
	if (!out->centers || !out->list_offsets || !out->start_pages)
	{
		pgv_host_ivf_image_free(out);
		return pgv_host_fail(PGV_ERR_NOMEM, "out of memory staging %d lists", out->lists);
	}

	/* pass 1: the list pages (GetScanLists' walk, src/ivfscan.c:58-111) */
	for (blk = IVFFLAT_HEAD_BLKNO; blk != PGV_INVALID_BLOCK && l < out->lists; blk = page_opaque(page_at(rel, blk))->nextblkno)
	{
		const uint8_t *page = page_at(rel, blk);
		int			maxoff = page_max_offset(page);

		for (int off = 1; off <= maxoff && l < out->lists; off++, l++)
		{
			const uint8_t *li = page_item(page, off, NULL);

			memcpy(&out->start_pages[l], li, 4);
			memcpy((char *) out->centers + (size_t) l * row_bytes, li + 8 + 8, row_bytes);	/* skip varlena hdr + dim */
		}
	}
	if (l != out->lists)
	{
		pgv_host_ivf_image_free(out);
		return pgv_host_fail(PGV_ERR_STATE, "list pages hold %d of %d lists", l, meta.lists);
	}

	/*
	 * pass 2: every list's entry-page chain (GetScanItems' walk, :139-179).  First the chains are walked for
	 * their line-pointer counts only (a few bytes per page), which fixes every list's row range; then the
	 * lists, which share nothing, are copied out in parallel.
	 */
	(void) cap;
	for (l = 0; l < out->lists; l++)
	{
		out->list_offsets[l] = n;
		for (blk = out->start_pages[l]; blk != PGV_INVALID_BLOCK; blk = page_opaque(page_at(rel, blk))->nextblkno)
			n += page_max_offset(page_at(rel, blk));
	}
	out->vectors = pgv_host_big_alloc(row_bytes * (size_t) (n > 0 ? n : 1));
	out->tids = malloc(sizeof(uint64_t) * (size_t) (n > 0 ? n : 1));
	if (!out->vectors || !out->tids)
	{
		pgv_host_ivf_image_free(out);
		return pgv_host_fail(PGV_ERR_NOMEM, "out of memory staging %lld rows of %zu bytes", (long long) n, row_bytes);
	}
	{
		int			bad_dim = 0;

#pragma omp parallel for schedule(dynamic, 4) num_threads(pgv_host_threads())
		for (int li = 0; li < out->lists; li++)
		{
			int64_t		at = out->list_offsets[li];

			for (uint32_t b = out->start_pages[li]; b != PGV_INVALID_BLOCK; b = page_opaque(page_at(rel, b))->nextblkno)
			{
				const uint8_t *page = page_at(rel, b);
				int			maxoff = page_max_offset(page);

				for (int off = 1; off <= maxoff; off++)
				{
					const uint8_t *itup = page_item(page, off, NULL);
					int			dim;
					const uint8_t *payload = tuple_vector_payload(itup, &dim);
					uint16_t	hi,
								lo,
								pos;

					if (dim != out->dim)
					{
#pragma omp atomic write
						bad_dim = dim;
						continue;
					}
					memcpy((char *) out->vectors + (size_t) at * row_bytes, payload, row_bytes);
					memcpy(&hi, itup + 0, 2);
					memcpy(&lo, itup + 2, 2);
					memcpy(&pos, itup + 4, 2);
					out->tids[at] = ((uint64_t) hi << 32) | ((uint64_t) lo << 16) | pos;
					at++;
				}
			}
		}
		if (bad_dim)
		{
			pgv_host_ivf_image_free(out);
			return pgv_host_fail(PGV_ERR_DIMS, "different vector dimensions %d and %d", bad_dim, meta.dimensions);
		}
	}
	out->list_offsets[out->lists] = n;
	out->nrows = n;
	return PGV_OK;
}

void
pgv_host_ivf_image_free(pgv_ivf_image * img)
{
	free(img->centers);
	free(img->list_offsets);
	free(img->vectors);
	free(img->tids);
	free(img->start_pages);
	memset(img, 0, sizeof(*img));
}
