/*
 * ext_driver.c -- the glue of ext/ (what a maintainer pastes into pgvector's src/) EXECUTED: a postmaster, backends
 * and the "pgvector gpu" background worker as real processes over the stand-in server of pgshim_runtime.c, on the
 * real library (tests/test_ext_runtime_gpu.py) or on tests/c/mock_hip.c (tests/test_ext_runtime_cpu.py).
 *
 *   build     IvfflatKmeans + BuildCallback + AssignTuples through PgvIvfflatKmeans / BuildAdd / BuildFlush, heap
 *             values arriving plain and toasted-style, the caller's memory poisoned after every row (src/ivfbuild.c:238-249)
 *   scans     ivfflatbeginscan / rescan / gettuple / endscan through the hooks: own-context and pooled, heads, deep
 *             pulls (device windows, whole-batch fetch), iterative scans, NULL queries, an ERROR in mid-scan (device
 *             state must go with the memory context), cancel while waiting for the pooler
 *   mirrors   staged by the worker through the buffer manager, imported by backends; insert -> stale -> CPU path ->
 *             restaged; an open scan keeps the import it began on (use-after-free); a pooled scan whose
 *             mirror is restaged under it restarts on the CPU path without returning a tuple twice
 *   worker    killed without a word (its process is gone), ended politely (exit hook): backends never hang, a new one
 *             starts; a staging of several seconds (no beat-based verdict while the process lives), DROP INDEX x 70
 *   hnsw      PgvHnswStage / BeginScan / GetScanItems against the oracle's walk of the same graph
 *
 * Expected answers: the oracle walking the very same pages (ora_pages_search), tie-tolerant.  TEST INFRASTRUCTURE.
 * Prints "EXT-RUNTIME OK" and exits 0 when every scenario passed.
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <unistd.h>

#include "pgv_gpu.h"
#include "hnsw.h"
#include "pgshim_runtime.h"

#include "pgv_host.h"
#include "pgv_oracle.h"

void		mock_hip_set_arena(void *base, size_t bytes) __attribute__((weak));
int			mock_hip_live_queries(void) __attribute__((weak));
int			mock_hip_contexts_made(int device) __attribute__((weak));

#define EXPECT(cond) do { if (!(cond)) { fprintf(stderr, "%s:%d: [%s] EXPECT(%s) failed\n", __FILE__, __LINE__, scenario, #cond); return 1; } } while (0)
static const char *scenario = "setup";
static int	cur_ops = ORA_OPS_L2;	/* which opclass the oracle answers for (the relation under test) */

#define REL_IVF 1001
#define REL_BATCH 1002
#define REL_HNSW 2001
#define DIM 32
#define NROWS 20000
#define LISTS 20
#define PROBES 3

/* what the processes of one run tell each other (a mapping of the driver's own, made before any fork) */
typedef struct Board
{
	volatile int step;			/* scenario-specific hand-shakes */
	volatile int ack;
	volatile uint64 new_tid;
	float		new_row[DIM];
}			Board;
static Board * board;

/* ------------------------------------------------------------------------------------------------ data */
static uint64 lcg = 99;
static double
urand(void)
{
	lcg = lcg * 6364136223846793005ull + 1442695040888963407ull;
	return (double) (lcg >> 11) / 9007199254740992.0;
}

static float
gauss(void)
{
	double		u = urand() + 1e-12,
				v = urand();

	return (float) (sqrt(-2.0 * log(u)) * cos(6.283185307179586 * v));
}

/* clustered rows: LISTS centers in [0,1)^dim, sigma 0.05 */
static void
gen_rows(float *out, int n, int dim, uint64 seed)
{
	float	   *centers = malloc(sizeof(float) * LISTS * dim);

	lcg = 4242;
	for (int i = 0; i < LISTS * dim; i++)
		centers[i] = (float) urand();
	lcg = seed;
	for (int r = 0; r < n; r++)
	{
		int			c = (int) (urand() * LISTS) % LISTS;

		for (int d = 0; d < dim; d++)
			out[(size_t) r * dim + d] = centers[c * dim + d] + 0.05f * gauss();
	}
	free(centers);
}

static uint64
tid_of_row(int r)
{
	return ((uint64) (r / 50 + 1) << 16) | (uint64) (r % 50 + 1);	/* 50 heap tuples per heap page */
}

static ItemPointerData
itemptr(uint64 tid)
{
	ItemPointerData p;

	p.ip_blkid.bi_hi = (uint16) (tid >> 32);
	p.ip_blkid.bi_lo = (uint16) (tid >> 16);
	p.ip_posid = (OffsetNumber) (tid & 0xffff);
	return p;
}

static uint64
tid_key(const ItemPointerData *p)
{
	return ((uint64) (((uint32) p->ip_blkid.bi_hi << 16) | p->ip_blkid.bi_lo) << 16) | p->ip_posid;
}

/* a Vector varlena in the current memory context */
static Vector *
make_vector(const float *x, int dim)
{
	Size		size = offsetof(Vector, x) + sizeof(float) * (Size) dim;
	Vector	   *v = palloc0(size);

	v->vl_len_ = (int32) (size << 2);
	v->dim = (int16) dim;
	memcpy(v->x, x, sizeof(float) * (Size) dim);
	return v;
}

/* the runtime's stand-in for a toasted (compressed) heap value: header with the "compressed" bits, raw size, payload */
static Datum
make_toasted(const Vector *v)
{
	Size		raw = ((uint32) v->vl_len_ >> 2) - 4;
	char	   *t = palloc(8 + raw);
	uint32		hdr = (uint32) ((8 + raw) << 2) | 0x02;
	uint32		rawsz = (uint32) raw;

	memcpy(t, &hdr, 4);
	memcpy(t + 4, &rawsz, 4);
	memcpy(t + 8, (const char *) v + 4, raw);
	return PointerGetDatum(t);
}

/* ------------------------------------------------------------------------------------------------ oracle answers */
typedef struct Expected
{
	int			n;
	uint64	   *tids;
	double	   *dist;
	uint64	   *sorted_tids;	/* for lookups */
	double	   *sorted_dist;
}			Expected;

static int
cmp_u64_pair(const void *a, const void *b)
{
	uint64		x = *(const uint64 *) a,
				y = *(const uint64 *) b;

	return x < y ? -1 : (x > y ? 1 : 0);
}

/* every tuple of the `probes` nearest lists, ascending by distance (ivfflatgettuple's whole first batch) */
static Expected
expected_batch(Oid relid, const float *query, int probes)
{
	Expected	e;
	uint32_t	nblocks;
	const uint8_t *pages = shim_relation_pages(relid, &nblocks);
	int			cap = 1 << 20;
	int64_t		scanned = 0;
	struct
	{
		uint64		t;
		double		d;
	}		   *pairs;

	e.tids = malloc(sizeof(uint64) * (size_t) cap);
	e.dist = malloc(sizeof(double) * (size_t) cap);
	e.n = ora_pages_search(pages, nblocks, cur_ops, ORA_F32, query, probes, cap, e.tids, e.dist, &scanned);
	pairs = malloc(sizeof(*pairs) * (size_t) (e.n > 0 ? e.n : 1));
	for (int i = 0; i < e.n; i++)
	{
		pairs[i].t = e.tids[i];
		pairs[i].d = e.dist[i];
	}
	qsort(pairs, (size_t) e.n, sizeof(*pairs), cmp_u64_pair);
	e.sorted_tids = malloc(sizeof(uint64) * (size_t) (e.n > 0 ? e.n : 1));
	e.sorted_dist = malloc(sizeof(double) * (size_t) (e.n > 0 ? e.n : 1));
	for (int i = 0; i < e.n; i++)
	{
		e.sorted_tids[i] = pairs[i].t;
		e.sorted_dist[i] = pairs[i].d;
	}
	free(pairs);
	return e;
}

static void
expected_free(Expected * e)
{
	free(e->tids);
	free(e->dist);
	free(e->sorted_tids);
	free(e->sorted_dist);
}

static int
expected_lookup(const Expected * e, uint64 tid, double *dist)
{
	int			lo = 0,
				hi = e->n - 1;

	while (lo <= hi)
	{
		int			mid = (lo + hi) / 2;

		if (e->sorted_tids[mid] == tid)
		{
			*dist = e->sorted_dist[mid];
			return 1;
		}
		if (e->sorted_tids[mid] < tid)
			lo = mid + 1;
		else
			hi = mid - 1;
	}
	return 0;
}

/* got[0..n) must be the head of e's stream: distinct tuples of the probed lists whose distances are e's, position by
 * position, within the float tolerance (ties may come in either order) */
static int
check_stream(const Expected * e, const uint64 *got, int n, int offset, const char *what)
{
	for (int i = 0; i < n; i++)
	{
		double		d;

		if (offset + i >= e->n || !expected_lookup(e, got[i], &d))
		{
			fprintf(stderr, "[%s] %s: tuple %d (tid %llx) is not in the probed lists (expected %d tuples)\n", scenario, what, offset + i,
					(unsigned long long) got[i], e->n);
			return 1;
		}
		if (fabs(d - e->dist[offset + i]) > 1e-4 * fabs(e->dist[offset + i]) + 1e-6)
		{
			fprintf(stderr, "[%s] %s: position %d: distance %.9g, the reference's stream has %.9g there\n", scenario, what, offset + i, d,
					e->dist[offset + i]);
			return 1;
		}
		for (int j = 0; j < i; j++)
			if (got[j] == got[i])
			{
				fprintf(stderr, "[%s] %s: tid %llx returned twice (positions %d and %d)\n", scenario, what, (unsigned long long) got[i], j, i);
				return 1;
			}
	}
	return 0;
}

/* ------------------------------------------------------------------------------------------------ the AM's scan, emulated */
typedef struct Scan
{
	IndexScanDescData desc;
	IvfflatScanOpaqueData so;
	ScanKeyData orderby;
	MemoryContext ctx;
	/* the reference's own path, stood in for by the oracle over the current pages */
	Expected	cpu;
	bool		cpu_open;
	int			cpu_next;
	float		query[DIM];
	int			gpu_tuples,
				cpu_tuples;
}			Scan;

/* ivfflatbeginscan (src/ivfscan.c:252-317) + ivfflatrescan with a query */
static void
scan_begin(Scan * s, Relation index, const float *query, int probes, int max_probes)
{
	memset(s, 0, sizeof(*s));
	s->ctx = shim_query_context_begin();
	s->desc.indexRelation = index;
	s->desc.numberOfOrderBys = 1;
	s->desc.orderByData = &s->orderby;
	s->desc.opaque = &s->so;
	s->so.probes = probes;
	s->so.maxProbes = max_probes;
	s->so.dimensions = DIM;
	s->so.first = true;
	if (query)
	{
		memcpy(s->query, query, sizeof(float) * DIM);
		s->so.value = PointerGetDatum(make_vector(query, DIM));	/* GetScanValue stays in the reference's file */
	}
	else
		s->so.value = PointerGetDatum(NULL);
	s->so.gpu = PgvIvfflatBeginScan(index, &s->so);
	PgvIvfflatRescan(s->so.gpu);
}

/* ivfflatgettuple (src/ivfscan.c:361-414) with the two hook lines of ext/pgv_gpu.h */
static bool
scan_gettuple(Scan * s)
{
	if (s->so.gpu)
	{
		int			r = PgvIvfflatGetTuple(&s->desc);

		if (r >= 0)
		{
			s->gpu_tuples += r;
			return r != 0;
		}
	}
	/* the reference's code: first batch only (what the oracle restates over the pages) */
	if (s->so.first)
	{
		s->cpu = expected_batch(RelationGetRelid(s->desc.indexRelation), s->query, s->so.probes);
		s->cpu_open = true;
		s->cpu_next = 0;
		s->so.first = false;
	}
	for (;;)
	{
		ItemPointerData heaptid;

		if (s->cpu_next >= s->cpu.n)
			return false;
		heaptid = itemptr(s->cpu.tids[s->cpu_next++]);
		if (s->so.gpu && PgvIvfflatAlreadyReturned(s->so.gpu, &heaptid))
			continue;
		s->desc.xs_heaptid = heaptid;
		s->cpu_tuples++;
		return true;
	}
}

static void
scan_end(Scan * s)
{
	PgvIvfflatEndScan(s->so.gpu);
	if (s->cpu_open)
		expected_free(&s->cpu);
	shim_query_context_end(s->ctx);
}

static void
make_query(float *q, int i)
{
	float		tmp[DIM * 4];

	gen_rows(tmp, 4, DIM, 777 + (uint64) i * 13);
	memcpy(q, tmp + DIM * (i % 4), sizeof(float) * DIM);
}

/* ------------------------------------------------------------------------------------------------ CREATE INDEX, emulated */
typedef struct Sorted
{
	int			n,
				cap;
	int32	   *list;
	uint64	   *tid;
	float	   *rows;
	int			dim;
}			Sorted;
static Sorted sorted;

static void
sorted_append(int list, uint64 tid, const float *x)
{
	if (sorted.n == sorted.cap)
	{
		sorted.cap = sorted.cap ? sorted.cap * 2 : 4096;
		sorted.list = realloc(sorted.list, sizeof(int32) * (size_t) sorted.cap);
		sorted.tid = realloc(sorted.tid, sizeof(uint64) * (size_t) sorted.cap);
		sorted.rows = realloc(sorted.rows, sizeof(float) * (size_t) sorted.cap * (size_t) sorted.dim);
	}
	sorted.list[sorted.n] = list;
	sorted.tid[sorted.n] = tid;
	memcpy(sorted.rows + (size_t) sorted.n * sorted.dim, x, sizeof(float) * (size_t) sorted.dim);
	sorted.n++;
}

#ifndef PGV_HAVE_REF_IVFBUILD
/* the tuplesort feed of AddTupleToSort (src/ivfbuild.c:203-216): the reference's file keeps it; here it collects.  (With the
 * reference's src/ivfbuild.c in the program, -DPGV_HAVE_REF_IVFBUILD, the function is the REFERENCE'S and the flushes of
 * the build hooks land in the build's tuplesort, which build_index reads back.) */
void
IvfflatAddToSort(IvfflatBuildState * buildstate, int list, ItemPointer tid, Datum value)
{
	(void) buildstate;
	sorted_append(list, tid_key(tid), ((Vector *) DatumGetPointer(value))->x);
}
#endif

/* With the reference's src/ivfbuild.c in the program the flushes of the build hooks go through ITS IvfflatAddToSort into the
 * build's tuplesort: set up as InitBuildState / AssignTuples do (src/ivfbuild.c:340-352, :380-390, :1040), read back in
 * the order the tuples were put */
static void
build_sort_begin(IvfflatBuildState * bs)
{
#ifdef PGV_HAVE_REF_IVFBUILD
	AttrNumber	attNums[] = {1};
	Oid			sortOperators[] = {Int4LessOperator};
	Oid			sortCollations[] = {InvalidOid};
	bool		nullsFirstFlags[] = {false};

	bs->sortdesc = CreateTemplateTupleDesc(3);
	TupleDescInitEntry(bs->sortdesc, (AttrNumber) 1, "list", INT4OID, -1, 0);
	TupleDescInitEntry(bs->sortdesc, (AttrNumber) 2, "tid", TIDOID, -1, 0);
	TupleDescInitEntry(bs->sortdesc, (AttrNumber) 3, "vector", SHIM_VECTOR_TYPE_OID, -1, 0);
	bs->slot = MakeSingleTupleTableSlot(bs->sortdesc, &TTSOpsVirtual);
	bs->sortstate = tuplesort_begin_heap(bs->sortdesc, 1, attNums, sortOperators, sortCollations, nullsFirstFlags, 65536, NULL, 0);
#else
	(void) bs;
#endif
}

static int
build_sort_drain(IvfflatBuildState * bs)
{
#ifdef PGV_HAVE_REF_IVFBUILD
	int			n = (int) shim_tuplesort_inputs(bs->sortstate);

	if ((int) bs->indtuples != n)
		return -1;
	for (int i = 0; i < n; i++)
	{
		int32_t		list;
		ItemPointerData tid;
		const void *value;

		shim_tuplesort_input(bs->sortstate, i, &list, &tid, &value);
		sorted_append(list, tid_key(&tid), ((const Vector *) value)->x);
	}
	tuplesort_end(bs->sortstate);
#else
	(void) bs;
#endif
	return sorted.n;
}

/* ivfflatbuild's GPU-relevant skeleton: samples -> IvfflatKmeans -> heap scan with BuildCallback -> sort by list ->
 * pages (src/ivfbuild.c:1008-1070) */
static int
build_index(Oid relid, const float *rows, int n, int dim, int lists, int toast_every)
{
	Relation	index = shim_open_relation(relid);
	IvfflatBuildState bs;
	VectorArrayData samples,
				centers;
	Size		itemsize = offsetof(Vector, x) + sizeof(float) * (Size) dim;
	int			nsamples = n < 50 * lists ? n : 50 * lists;
	MemoryContext tmp;
	int64_t    *offsets;
	float	   *packed_centers;
	pgv_rel		rel;
	int		   *order;

	EXPECT(index != NULL);
	memset(&bs, 0, sizeof(bs));
	bs.index = index;
	bs.typeInfo = IvfflatGetTypeInfo(index);
	bs.dimensions = dim;
	bs.lists = lists;
	samples.length = nsamples;
	samples.maxlen = nsamples;
	samples.dim = dim;
	samples.itemsize = itemsize;
	samples.items = palloc0(itemsize * (Size) nsamples);
	for (int i = 0; i < nsamples; i++)
	{
		Vector	   *v = (Vector *) VectorArrayGet(&samples, i);
		int			r = (int) (((int64) i * n) / nsamples);

		v->vl_len_ = (int32) (itemsize << 2);
		v->dim = (int16) dim;
		memcpy(v->x, rows + (size_t) r * dim, sizeof(float) * (Size) dim);
		if (cur_ops != ORA_OPS_L2)
		{
			/* SampleCallback normalises the samples of opclasses with a KMEANS_NORM proc (src/ivfbuild.c:148-156) */
			double		norm = 0.0;

			for (int d = 0; d < dim; d++)
				norm += (double) v->x[d] * (double) v->x[d];
			norm = sqrt(norm);
			for (int d = 0; d < dim && norm > 0.0; d++)
				v->x[d] = (float) ((double) v->x[d] / norm);
		}
	}
	centers.length = 0;
	centers.maxlen = lists;
	centers.dim = dim;
	centers.itemsize = itemsize;
	centers.items = palloc0(itemsize * (Size) lists);
	bs.samples = &samples;
	bs.centers = &centers;
	EXPECT(PgvIvfflatKmeans(index, &samples, &centers, bs.typeInfo));
	EXPECT(centers.length == lists);

	sorted.n = 0;
	sorted.dim = dim;
	build_sort_begin(&bs);
	PgvIvfflatBuildBegin(&bs);
	EXPECT(bs.gpu != NULL);
	/* BuildCallback (src/ivfbuild.c:224-266): the value is detoasted inside tmpCtx, which is reset after every row */
	tmp = shim_query_context_begin();
	for (int r = 0; r < n; r++)
	{
		Vector	   *plain = make_vector(rows + (size_t) r * dim, dim);
		Datum		heap_value = (toast_every && r % toast_every == 0) ? make_toasted(plain) : PointerGetDatum(plain);
		Datum		value = PointerGetDatum(PG_DETOAST_DATUM(heap_value));
		ItemPointerData tid = itemptr(tid_of_row(r));

		PgvIvfflatBuildAdd(&bs, &tid, value);
		shim_context_reset(tmp);	/* (pfree poisons: whoever kept a pointer into the row reads 0xDE from now on) */
	}
	shim_query_context_end(tmp);
	PgvIvfflatBuildFlush(&bs);
	EXPECT(build_sort_drain(&bs) == n);
	EXPECT(sorted.n == n);

	/* every row went to its nearest center (the oracle's FUNCTION 1 value decides; float-level ties excepted) */
	packed_centers = malloc(sizeof(float) * (size_t) lists * dim);
	for (int c = 0; c < lists; c++)
		memcpy(packed_centers + (size_t) c * dim, ((Vector *) VectorArrayGet(&centers, c))->x, sizeof(float) * (Size) dim);
	for (int i = 0; i < sorted.n; i += (sorted.n > 50000 ? 37 : 1))
	{
		double		best = INFINITY,
					mine;

		EXPECT(sorted.list[i] >= 0 && sorted.list[i] < lists);
		for (int c = 0; c < lists; c++)
		{
			double		d = ora_index_distance(cur_ops, ORA_F32, dim, sorted.rows + (size_t) i * dim, packed_centers + (size_t) c * dim);

			if (d < best)
				best = d;
		}
		mine = ora_index_distance(cur_ops, ORA_F32, dim, sorted.rows + (size_t) i * dim, packed_centers + (size_t) sorted.list[i] * dim);
		EXPECT(mine <= best + 1e-5 * fabs(best) + 1e-9);
		EXPECT(sorted.tid[i] == tid_of_row(i));	/* rows come out of the flushes in heap order, none lost, none twice */
		EXPECT(memcmp(sorted.rows + (size_t) i * dim, rows + (size_t) i * dim, sizeof(float) * (size_t) dim) == 0);
	}

	/* tuplesort by list (stable: heap order inside a list), then the page writer */
	order = malloc(sizeof(int) * (size_t) n);
	offsets = calloc((size_t) lists + 1, sizeof(int64_t));
	for (int i = 0; i < n; i++)
		offsets[sorted.list[i] + 1]++;
	for (int l = 0; l < lists; l++)
		offsets[l + 1] += offsets[l];
	{
		int64_t    *at = malloc(sizeof(int64_t) * (size_t) lists);
		float	   *v = malloc(sizeof(float) * (size_t) n * dim);
		uint64	   *t = malloc(sizeof(uint64) * (size_t) n);

		memcpy(at, offsets, sizeof(int64_t) * (size_t) lists);
		for (int i = 0; i < n; i++)
		{
			int64_t		p = at[sorted.list[i]]++;

			memcpy(v + (size_t) p * dim, sorted.rows + (size_t) i * dim, sizeof(float) * (size_t) dim);
			t[p] = sorted.tid[i];
		}
		pgv_rel_init(&rel);
		EXPECT(pgv_host_ivf_write_index(&rel, PGV_F32, dim, lists, packed_centers, offsets, v, t) == PGV_OK);
		shim_replace_pages(relid, rel.pages, rel.nblocks);
		pgv_rel_free(&rel);
		free(at);
		free(v);
		free(t);
	}
	PgvNoteIndexChange(index);
	free(order);
	free(offsets);
	free(packed_centers);
	EXPECT(shim_pinned_buffers() == 0);
	return 0;
}

static int
backend_build(void *arg)
{
	float	   *rows = malloc(sizeof(float) * NROWS * DIM);

	(void) arg;
	scenario = "build";
	shim_set_guc_bool("vector.gpu", true);
	shim_seed_random(11);
	gen_rows(rows, NROWS, DIM, 1);
	if (build_index(REL_IVF, rows, NROWS, DIM, LISTS, 7))
		return 1;
	free(rows);
	{
		/* more rows than one assignment batch holds (PGV_ASSIGN_BATCH = 2^18): a flush in mid-scan, the rest at the end */
		int			n = (1 << 18) + 5000,
					dim = 8;
		float	   *big = malloc(sizeof(float) * (size_t) n * dim);

		scenario = "build across an assignment batch";
		gen_rows(big, n, dim, 5);
		if (build_index(REL_BATCH, big, n, dim, 16, 0))
			return 1;
		free(big);
	}
	return 0;
}

/* ------------------------------------------------------------------------------------------------ scans */
static int
pull(Scan * s, uint64 *out, int want)
{
	int			n = 0;

	while (n < want && scan_gettuple(s))
		out[n++] = tid_key(&s->desc.xs_heaptid);
	return n;
}

/* wait until scans are served by the GPU path (the worker has staged the index); 0 when they are */
static int
wait_for_gpu(Relation index, double timeout_s)
{
	double		until = shim_now() + timeout_s;
	float		q[DIM];

	make_query(q, 0);
	while (shim_now() < until)
	{
		Scan		s;
		uint64		t;
		int			got_gpu;

		scan_begin(&s, index, q, PROBES, PROBES);
		pull(&s, &t, 1);
		got_gpu = s.gpu_tuples;
		scan_end(&s);
		if (got_gpu)
			return 0;
		usleep(20000);
	}
	return 1;
}

static int
backend_scan_own(void *arg)
{
	Relation	index = shim_open_relation(REL_IVF);
	uint64	   *got = malloc(sizeof(uint64) * 30000);
	float		q[DIM];

	(void) arg;
	scenario = "own-context scans";
	shim_set_guc_bool("vector.gpu", true);
	/* the first scan finds no mirror: CPU path, and the worker is asked to stage */
	{
		Scan		s;

		make_query(q, 1);
		scan_begin(&s, index, q, PROBES, PROBES);
		EXPECT(pull(&s, got, 5) == 5);
		EXPECT(s.gpu_tuples == 0 && s.cpu_tuples == 5);
		scan_end(&s);
	}
	EXPECT(wait_for_gpu(index, 30.0) == 0);
	for (int i = 0; i < 40; i++)
	{
		Scan		s;
		Expected	e;
		int			want = i < 30 ? 10 : (i < 36 ? 300 : 30000);	/* LIMIT 10; device windows; the whole batch */
		int			n;

		make_query(q, i);
		e = expected_batch(REL_IVF, q, PROBES);
		scan_begin(&s, index, q, PROBES, PROBES);
		n = pull(&s, got, want);
		EXPECT(n == (want < e.n ? want : e.n));
		EXPECT(s.cpu_tuples == 0);
		if (check_stream(&e, got, n, 0, "own context"))
			return 1;
		/* rescan with another query on the same scan */
		make_query(q, i + 100);
		expected_free(&e);
		e = expected_batch(REL_IVF, q, PROBES);
		memcpy(s.query, q, sizeof(q));
		s.so.value = PointerGetDatum(make_vector(q, DIM));
		s.so.first = true;
		PgvIvfflatRescan(s.so.gpu);
		n = pull(&s, got, 10);
		EXPECT(n == 10);
		if (check_stream(&e, got, n, 0, "rescan"))
			return 1;
		scan_end(&s);
		expected_free(&e);
	}
	/* iterative scan (ivfflat.iterative_scan, src/ivfscan.c:400-406): maxProbes 9, three lists at a time; every batch is
	 * sorted by itself */
	{
		Scan		s;
		Expected	e3,
					e6;
		int			n;

		make_query(q, 7);
		e3 = expected_batch(REL_IVF, q, 3);
		e6 = expected_batch(REL_IVF, q, 6);
		scan_begin(&s, index, q, 3, 9);
		n = pull(&s, got, 30000);
		EXPECT(n > e6.n);
		if (check_stream(&e3, got, e3.n, 0, "iterative, first batch"))
			return 1;
		/* the second batch: the tuples of lists 4..6, ascending */
		{
			double		prev = -1.0;

			for (int i = e3.n; i < e6.n; i++)
			{
				double		d,
							unused;

				EXPECT(expected_lookup(&e6, got[i], &d) && !expected_lookup(&e3, got[i], &unused));
				EXPECT(d >= prev - 1e-4 * fabs(prev) - 1e-6);
				prev = d;
			}
		}
		scan_end(&s);
		expected_free(&e3);
		expected_free(&e6);
	}
	/* a NULL query (ZeroDistance, src/ivfscan.c:192-196): every tuple of the probed lists, any order */
	{
		Scan		s;
		int			n;

		scan_begin(&s, index, NULL, PROBES, PROBES);
		n = pull(&s, got, 30000);
		EXPECT(n > 0 && s.gpu_tuples == n);
		for (int i = 1; i < n && i < 2000; i++)
			EXPECT(got[i] != got[0]);
		scan_end(&s);
	}
	/* an ERROR in mid-scan longjmps past ivfflatendscan: the scan's device state must go with its memory context */
	{
		int			before = mock_hip_live_queries ? mock_hip_live_queries() : 0;
		uint64_t	free0 = 0,
					free1 = 0,
					total = 0;

		pgv_device_memory(0, &free0, &total);
		for (int i = 0; i < 200; i++)
		{
			volatile int caught = 0;

			PG_TRY();
			{
				Scan		s;

				make_query(q, i);
				scan_begin(&s, index, q, PROBES, PROBES);
				pull(&s, got, 3);
				elog(ERROR, "division by zero");	/* some other part of the query fails */
				scan_end(&s);
			}
			PG_CATCH();
			{
				caught = 1;
				/* AbortTransaction: the query's context is reset, its reset callbacks fire */
				shim_context_reset(CurrentMemoryContext);
				CurrentMemoryContext = TopMemoryContext;
				FlushErrorState();
			}
			PG_END_TRY();
			EXPECT(caught);
		}
		if (mock_hip_live_queries)
			EXPECT(mock_hip_live_queries() == before);
		pgv_device_memory(0, &free1, &total);
		EXPECT(free0 - free1 < (64u << 20) || free1 >= free0);	/* 200 leaked pgv_query would be ~200 x 1 MB */
	}
	EXPECT(shim_pinned_buffers() == 0);
	free(got);
	return 0;
}

static int
backend_scan_pooled(void *arg)
{
	int			id = (int) (intptr_t) arg;
	Relation	index = shim_open_relation(REL_IVF);
	uint64	   *got = malloc(sizeof(uint64) * 30000);
	float		q[DIM];
	int			served = 0;

	scenario = "pooled scans";
	shim_set_guc_bool("vector.gpu", true);
	shim_set_guc_bool("vector.gpu_pooled", true);
	/* six backends pull past the pooler's head here and each continues on a context of its own (the default lets four
	 * have one; the fifth would continue in the reference's code -- correct, but not what this scenario looks at) */
	shim_set_guc_int("vector.gpu_max_own_contexts", 8);
	EXPECT(wait_for_gpu(index, 30.0) == 0);
	for (int i = 0; i < 40; i++)
	{
		Scan		s;
		Expected	e;
		int			want = i % 8 == 5 ? 150 : (i % 8 == 6 ? 2500 : 10);	/* past the pooler's head of 64: own context takes over */
		int			n;

		make_query(q, id * 100 + i);
		e = expected_batch(REL_IVF, q, PROBES);
		scan_begin(&s, index, q, PROBES, PROBES);
		n = pull(&s, got, want);
		EXPECT(n == (want < e.n ? want : e.n));
		EXPECT(s.cpu_tuples == 0);
		served += s.gpu_tuples > 0;
		if (check_stream(&e, got, n, 0, "pooled"))
			return 1;
		scan_end(&s);
		expected_free(&e);
	}
	EXPECT(served == 40);
	/* cancelled while waiting for the pooler: the ERROR comes out, the slot goes back, the next query is served */
	for (int round = 0; round < 300; round++)
	{
		volatile int caught = 0;

		PG_TRY();
		{
			float		dist[PGV_POOL_HEAD];
			uint64		tid[PGV_POOL_HEAD];
			int			count;
			bool		complete;
			uint64		staged;

			make_query(q, round);
			shim_cancel_after(0);	/* the first CHECK_FOR_INTERRUPTS inside the wait */
			(void) PgvPoolSearch(index, q, PROBES, dist, tid, &count, &complete, &staged);
			shim_cancel_after(-1);	/* (answered before the first check: fine) */
		}
		PG_CATCH();
		{
			caught = 1;
			FlushErrorState();
		}
		PG_END_TRY();
		(void) caught;
	}
	{
		/* 300 cancelled queries later the 256 slots are not used up: the pooler still answers */
		float		dist[PGV_POOL_HEAD];
		uint64		tid[PGV_POOL_HEAD];
		int			count = 0;
		bool		complete;
		uint64		staged = 0;
		Expected	e;

		make_query(q, 3);
		e = expected_batch(REL_IVF, q, PROBES);
		EXPECT(PgvPoolSearch(index, q, PROBES, dist, tid, &count, &complete, &staged));
		EXPECT(count == PGV_POOL_HEAD && staged != 0 && !complete);
		if (check_stream(&e, tid, count, 0, "pool after cancellations"))
			return 1;
		expected_free(&e);
	}
	free(got);
	return 0;
}

/* ------------------------------------------------------------------------------------------------ insert -> stale -> restaged */
static int
wait_step(int step, double timeout_s)
{
	double		until = shim_now() + timeout_s;

	while (board->step < step && shim_now() < until)
		usleep(2000);
	return board->step >= step ? 0 : 1;
}

/* backend A: holds a scan open across the restage (a cursor): it keeps the import it began on */
static int
backend_cursor(void *arg)
{
	Relation	index = shim_open_relation(REL_IVF);
	uint64		got[4000];
	float		q[DIM];
	Scan		s;
	Expected	e;
	int			n;

	(void) arg;
	scenario = "open scan across a restage";
	shim_set_guc_bool("vector.gpu", true);
	EXPECT(wait_for_gpu(index, 30.0) == 0);
	make_query(q, 21);
	e = expected_batch(REL_IVF, q, PROBES);	/* the image the scan starts on */
	scan_begin(&s, index, q, PROBES, PROBES);
	n = pull(&s, got, 5);
	EXPECT(n == 5 && s.gpu_tuples == 5);
	board->ack = 1;				/* the cursor is open */
	EXPECT(wait_step(2, 60.0) == 0);	/* ... the index has been changed and restaged, other scans run on the new import */
	/* a second scan of this backend picks up the NEW staging while the first one is still open */
	{
		Scan		s2;
		uint64		t;

		scan_begin(&s2, index, board->new_row, PROBES, PROBES);
		EXPECT(pull(&s2, &t, 1) == 1 && s2.gpu_tuples == 1);
		EXPECT(t == board->new_tid);	/* the inserted row is its own nearest neighbour */
		scan_end(&s2);
	}
	/* the open scan goes on, on the image it began with: deep into the batch (device windows + whole-batch fetch) */
	n += pull(&s, got + 5, 3000);
	EXPECT(s.cpu_tuples == 0 && n == (e.n < 3005 ? e.n : 3005));
	if (check_stream(&e, got, n, 0, "cursor on the old import"))
		return 1;
	scan_end(&s);
	expected_free(&e);
	return 0;
}

/* backend B: inserts a row (ivfflatinsert), sees its scans fall to the CPU path, then served again with the new row */
static int
backend_insert(void *arg)
{
	Relation	index = shim_open_relation(REL_IVF);
	float		row[DIM];
	uint64		tid = ((uint64) 900000 << 16) | 7;
	double		until;
	int			saw_cpu = 0;

	(void) arg;
	scenario = "insert -> stale -> restaged";
	shim_set_guc_bool("vector.gpu", true);
	shim_set_guc_int("vector.gpu_restage_delay_ms", 0);
	EXPECT(wait_for_gpu(index, 30.0) == 0);
	until = shim_now() + 30.0;
	while (!board->ack && shim_now() < until)
		usleep(2000);
	EXPECT(board->ack);
	/* InsertTuple (src/ivfinsert.c:72-181) on the pages, then the hook */
	make_query(row, 55);
	{
		pgv_rel		rel;
		uint32_t	nblocks;
		const void *pages = shim_relation_pages(REL_IVF, &nblocks);
		uint32_t	list = 0;
		int			dim,
					lists;

		pgv_rel_init(&rel);
		rel.pages = malloc((size_t) (nblocks + 8) * PGV_BLCKSZ);
		memcpy(rel.pages, pages, (size_t) nblocks * PGV_BLCKSZ);
		rel.nblocks = nblocks;
		rel.cap = nblocks + 8;
		EXPECT(ora_pages_meta(rel.pages, rel.nblocks, &dim, &lists) && dim == DIM && lists == LISTS);
		/* FindInsertPage: the nearest list (the oracle's single-probe search tells which tuples are its) */
		{
			uint64		t1;
			double		d1;
			int64_t		scanned;

			EXPECT(ora_pages_search(rel.pages, rel.nblocks, ORA_OPS_L2, ORA_F32, row, 1, 1, &t1, &d1, &scanned) == 1);
			/* which list holds t1: try lists until the insert lands where a 1-probe scan finds it */
			for (list = 0; list < (uint32_t) lists; list++)
			{
				pgv_rel		trial = rel;
				uint64		tt[2];
				double		dd[2];

				trial.pages = malloc((size_t) rel.cap * PGV_BLCKSZ);
				memcpy(trial.pages, rel.pages, (size_t) rel.nblocks * PGV_BLCKSZ);
				if (pgv_host_ivf_insert(&trial, PGV_F32, (int) list, row, tid) == PGV_OK &&
					ora_pages_search(trial.pages, trial.nblocks, ORA_OPS_L2, ORA_F32, row, 1, 2, tt, dd, &scanned) >= 1 && tt[0] == tid)
				{
					free(rel.pages);
					rel = trial;
					break;
				}
				free(trial.pages);
			}
			EXPECT(list < (uint32_t) lists);
		}
		shim_replace_pages(REL_IVF, rel.pages, rel.nblocks);
		free(rel.pages);
	}
	memcpy(board->new_row, row, sizeof(row));
	board->new_tid = tid;
	PgvNoteIndexChange(index);
	/* scans now: the mirror is stale -> the reference's path (current pages, the new row included) until the worker has
	 * restaged; then the GPU path with the new row */
	until = shim_now() + 30.0;
	for (;;)
	{
		Scan		s;
		uint64		t;
		int			n;

		scan_begin(&s, index, row, PROBES, PROBES);
		n = pull(&s, &t, 1);
		EXPECT(n == 1 && t == tid);	/* whichever path answers, it answers from the current pages */
		if (s.cpu_tuples)
			saw_cpu = 1;
		scan_end(&s);
		if (s.gpu_tuples)
			break;
		EXPECT(shim_now() < until);
		usleep(5000);
	}
	EXPECT(saw_cpu);
	board->step = 2;
	return 0;
}

/* a pooled scan that outlives its mirror: head from the pool, then the index changes, then it wants more */
static int
backend_pooled_restage(void *arg)
{
	Relation	index = shim_open_relation(REL_IVF);
	uint64	   *got = malloc(sizeof(uint64) * 30000);
	float		q[DIM];
	Scan		s;
	Expected	before,
				after;
	int			n;

	(void) arg;
	scenario = "pooled scan across a restage";
	shim_set_guc_bool("vector.gpu", true);
	shim_set_guc_bool("vector.gpu_pooled", true);
	shim_set_guc_int("vector.gpu_restage_delay_ms", 0);
	EXPECT(wait_for_gpu(index, 30.0) == 0);
	make_query(q, 31);
	before = expected_batch(REL_IVF, q, PROBES);
	scan_begin(&s, index, q, PROBES, PROBES);
	n = pull(&s, got, 40);
	EXPECT(n == 40 && s.gpu_tuples == 40);
	if (check_stream(&before, got, 40, 0, "pooled head"))
		return 1;
	/* the index changes under the open scan (no new row needed: the generation is what counts) */
	PgvNoteIndexChange(index);
	n += pull(&s, got + n, 30000);
	after = expected_batch(REL_IVF, q, PROBES);
	EXPECT(n == after.n);		/* every tuple of the probed lists, once: not truncated, none twice */
	EXPECT(s.cpu_tuples > 0);	/* past the pooler's head the scan went on in the reference's code */
	for (int i = 0; i < n; i++)
	{
		double		d;

		EXPECT(expected_lookup(&after, got[i], &d));
	}
	{
		uint64	   *copy = malloc(sizeof(uint64) * (size_t) n);

		memcpy(copy, got, sizeof(uint64) * (size_t) n);
		qsort(copy, (size_t) n, sizeof(uint64), cmp_u64_pair);
		for (int i = 1; i < n; i++)
			EXPECT(copy[i] != copy[i - 1]);
		free(copy);
	}
	scan_end(&s);
	expected_free(&before);
	expected_free(&after);
	free(got);
	return 0;
}

/* ------------------------------------------------------------------------------------------------ the worker goes away */
static int
backend_after_worker_loss(void *arg)
{
	int			pooled = (int) (intptr_t) arg;
	Relation	index = shim_open_relation(REL_IVF);
	float		q[DIM];
	double		t0 = shim_now(),
				until = t0 + 40.0;
	int			saw_gpu = 0;

	scenario = pooled ? "pooled query after the worker was killed" : "own-context query after the worker was ended";
	shim_set_guc_bool("vector.gpu", true);
	shim_set_guc_bool("vector.gpu_pooled", pooled != 0);
	make_query(q, 61);
	while (shim_now() < until && !saw_gpu)
	{
		Scan		s;
		uint64		got[10];
		Expected	e = expected_batch(REL_IVF, q, PROBES);
		double		t1 = shim_now();

		scan_begin(&s, index, q, PROBES, PROBES);
		EXPECT(pull(&s, got, 10) == 10);	/* answered -- by whichever path -- and never hung */
		EXPECT(shim_now() - t1 < 8.0);
		if (check_stream(&e, got, 10, 0, "after worker loss"))
			return 1;
		/* the GPU serves again AND a worker is there: the old one's mirror died with it (an import of its handle fails:
		 * CPU path, the registry forgets the staging), so a GPU answer means a new worker staged the index again */
		saw_gpu = s.gpu_tuples > 0 && shim_live_bgworkers() == 1;
		scan_end(&s);
		expected_free(&e);
		if (!saw_gpu)
			usleep(50000);
	}
	EXPECT(saw_gpu);			/* a new worker was started and staged the index again */
	fprintf(stderr, "   the GPU path was back %.1f s after the worker had gone\n", shim_now() - t0);
	return 0;
}

/* ------------------------------------------------------------------------------------------------ hnsw */
#define HN 3000
#define HM 8
static int
backend_hnsw_build(void *arg)
{
	float	   *data = malloc(sizeof(float) * HN * DIM);
	ora_hnsw   *g;
	int32_t    *levels = malloc(sizeof(int32_t) * HN);
	int64_t    *nbr_start = malloc(sizeof(int64_t) * (HN + 1));
	int32_t    *nbr,
			   *dup_of = malloc(sizeof(int32_t) * HN);
	uint64_t   *tids = malloc(sizeof(uint64_t) * HN);
	int			entry_level;
	pgv_rel		rel;

	scenario = "hnsw pages";
	gen_rows(data, HN, DIM, 3);
	g = ora_hnsw_build(ORA_OPS_L2, ORA_F32, DIM, data, HN, HM, 32, 9);
	EXPECT(ora_hnsw_num_elements(g) == HN);
	nbr_start[0] = 0;
	for (int e = 0; e < HN; e++)
	{
		levels[e] = ora_hnsw_level(g, e);
		nbr_start[e + 1] = nbr_start[e] + (int64_t) (levels[e] + 2) * HM;
		dup_of[e] = -1;
		tids[e] = tid_of_row(e);
		EXPECT(ora_hnsw_element_row(g, e) == e);
	}
	nbr = malloc(sizeof(int32_t) * (size_t) nbr_start[HN]);
	for (int e = 0; e < HN; e++)
		for (int lc = 0; lc <= levels[e]; lc++)
		{
			int			lm = lc == 0 ? 2 * HM : HM;
			int32_t		buf[2 * HM];
			int			cnt = ora_hnsw_neighbors(g, e, lc, buf);
			int32_t    *out = nbr + nbr_start[e] + (int64_t) (levels[e] - lc) * HM;

			for (int i = 0; i < lm; i++)
				out[i] = i < cnt ? buf[i] : -1;
		}
	pgv_rel_init(&rel);
	EXPECT(pgv_host_hnsw_write_index(&rel, PGV_F32, DIM, HM, 32, HN, data, tids, levels, nbr_start, nbr, dup_of,
									 ora_hnsw_entry_point(g, &entry_level)) == PGV_OK);
	shim_replace_pages(arg ? (Oid) (uintptr_t) arg : REL_HNSW, rel.pages, rel.nblocks);	/* (arg: another relation to lay the same graph into) */
	pgv_rel_free(&rel);
	ora_hnsw_free(g);
	return 0;
}

static int
backend_hnsw_scan(void *arg)
{
	Relation	index = shim_open_relation(REL_HNSW);
	float	   *data = malloc(sizeof(float) * HN * DIM);
	ora_hnsw   *g;
	double		until = shim_now() + 30.0;
	void	   *gpu = NULL;
	MemoryContext ctx;

	(void) arg;
	scenario = "hnsw scans";
	shim_set_guc_bool("vector.gpu", true);
	hnsw_ef_search = 40;
	gen_rows(data, HN, DIM, 3);
	g = ora_hnsw_build(ORA_OPS_L2, ORA_F32, DIM, data, HN, HM, 32, 9);	/* the graph the pages hold */
	/* hnswbeginscan: NULL (CPU path) until the worker has staged the graph */
	ctx = shim_query_context_begin();
	while ((gpu = PgvHnswBeginScan(index)) == NULL && shim_now() < until)
		usleep(20000);
	EXPECT(gpu != NULL);
	PgvHnswEndScan(gpu);
	shim_query_context_end(ctx);
	for (int i = 0; i < 30; i++)
	{
		IndexScanDescData desc;
		HnswScanOpaqueData so;
		float		q[DIM];
		List	   *w;
		int64_t		rows[40];
		double		dist[40];
		int64_t		scored;
		int			want,
					n;

		make_query(q, i);
		want = ora_hnsw_search(g, q, 40, 40, rows, dist, &scored);
		ctx = shim_query_context_begin();
		memset(&desc, 0, sizeof(desc));
		memset(&so, 0, sizeof(so));
		desc.indexRelation = index;
		desc.opaque = &so;
		so.first = true;
		so.gpu = PgvHnswBeginScan(index);
		EXPECT(so.gpu != NULL);
		w = NIL;
		EXPECT(PgvHnswGetScanItems(&desc, PointerGetDatum(make_vector(q, DIM)), &w));
		n = shim_list_length(w);
		EXPECT(n == want && so.m == HM);
		/* furthest first (hnswgettuple takes llast): position n - 1 - i is the i-th nearest */
		for (int j = 0; j < n; j++)
		{
			HnswSearchCandidate *sc = shim_list_nth(w, n - 1 - j);
			HnswElement el = sc->element.ptr;

			EXPECT(fabs(sc->distance - dist[j]) <= 1e-4 * fabs(dist[j]) + 1e-6);
			EXPECT(el->heaptidsLength == 1);
			if (j + 1 < n && fabs(dist[j + 1] - dist[j]) > 1e-4 * fabs(dist[j]) && (j == 0 || fabs(dist[j] - dist[j - 1]) > 1e-4 * fabs(dist[j])))
				EXPECT(tid_key(&el->heaptids[0]) == tid_of_row((int) rows[j]));
		}
		PgvHnswEndScan(so.gpu);
		shim_query_context_end(ctx);
	}
	/* scans the device walk does not serve: they are handed back untouched (the hook then calls GetScanItems).  A NULL
	 * query (ORDER BY embedding <-> NULL: every distance 0, src/hnswutils.c:555) used to be dereferenced; an iterative
	 * scan's later batches need the first batch's visited set and discarded candidates (src/hnswscan.c:61-88). */
	{
		IndexScanDescData desc;
		HnswScanOpaqueData so;
		float		q[DIM];
		List	   *w = (List *) &desc;	/* must stay as it is */

		make_query(q, 99);
		ctx = shim_query_context_begin();
		memset(&desc, 0, sizeof(desc));
		memset(&so, 0, sizeof(so));
		desc.indexRelation = index;
		desc.opaque = &so;
		so.first = true;
		so.m = -7;
		so.gpu = PgvHnswBeginScan(index);
		EXPECT(so.gpu != NULL);
		EXPECT(!PgvHnswGetScanItems(&desc, PointerGetDatum(NULL), &w));
		EXPECT(w == (List *) &desc && so.m == -7);
		hnsw_iterative_scan = 1;	/* HNSW_ITERATIVE_SCAN_RELAXED */
		EXPECT(!PgvHnswGetScanItems(&desc, PointerGetDatum(make_vector(q, DIM)), &w));
		EXPECT(w == (List *) &desc && so.m == -7);
		hnsw_iterative_scan = HNSW_ITERATIVE_SCAN_OFF;
		EXPECT(PgvHnswGetScanItems(&desc, PointerGetDatum(make_vector(q, DIM)), &w));
		EXPECT(shim_list_length(w) == 40 && so.m == HM);
		PgvHnswEndScan(so.gpu);
		shim_query_context_end(ctx);
	}
	ora_hnsw_free(g);
	free(data);
	return 0;
}

/* ------------------------------------------------------------------------------------------------ vector_ip_ops */
#define REL_IP 1006

/*
 * An inner-product opclass through the same hooks: FUNCTION 4 (KMEANS_NORM) makes the k-means spherical
 * (src/ivfkmeans.c:553-570 on normalised samples), FUNCTION 1 is vector_negative_inner_product, rows are stored as
 * they are (no FUNCTION 2).  PgvIvfflatOpclass must map it to PGV_OPS_IP / PGV_NEG_IP for the build, the worker's
 * staging and the scans; the oracle walks the pages with the same opclass.
 */
static int
backend_ip_opclass(void *arg)
{
	Relation	index = shim_open_relation(REL_IP);
	const int	n = 6000,
				lists = 12;
	float	   *rows = malloc(sizeof(float) * (size_t) n * DIM);
	uint64	   *got = malloc(sizeof(uint64) * 30000);
	float		q[DIM];

	(void) arg;
	scenario = "vector_ip_ops: build";
	cur_ops = ORA_OPS_IP;
	shim_set_guc_bool("vector.gpu", true);
	shim_set_guc_bool("vector.gpu_pooled", false);
	shim_seed_random(17);
	gen_rows(rows, n, DIM, 12);
	if (build_index(REL_IP, rows, n, DIM, lists, 11))
		return 1;
	scenario = "vector_ip_ops: scans";
	EXPECT(wait_for_gpu(index, 30.0) == 0);
	for (int i = 0; i < 12; i++)
	{
		Scan		s;
		Expected	e;
		int			want = i < 8 ? 10 : 400,
					got_n;

		make_query(q, 300 + i);
		e = expected_batch(REL_IP, q, PROBES);
		scan_begin(&s, index, q, PROBES, PROBES);
		got_n = pull(&s, got, want);
		EXPECT(got_n == (want < e.n ? want : e.n));
		EXPECT(s.cpu_tuples == 0 && s.gpu_tuples == got_n);
		if (check_stream(&e, got, got_n, 0, "inner product, own context"))
			return 1;
		scan_end(&s);
		expected_free(&e);
	}
	/* and through the pooler */
	shim_set_guc_bool("vector.gpu_pooled", true);
	for (int i = 0; i < 6; i++)
	{
		Scan		s;
		Expected	e;
		int			got_n;

		make_query(q, 340 + i);
		e = expected_batch(REL_IP, q, PROBES);
		scan_begin(&s, index, q, PROBES, PROBES);
		got_n = pull(&s, got, 10);
		EXPECT(got_n == 10 && s.cpu_tuples == 0);
		if (check_stream(&e, got, got_n, 0, "inner product, pooled"))
			return 1;
		scan_end(&s);
		expected_free(&e);
	}
	free(rows);
	free(got);
	return 0;
}

/* ------------------------------------------------------------------------------------------------ a backend without a device */
/*
 * `vector.gpu = on` in a backend that cannot have a device context (none installed, lost, the driver not initialising):
 * every hook goes back to the reference's code -- one WARNING, no failing query, no failing CREATE INDEX --, and the
 * pooled path still serves it (the worker owns the device; a pooled backend needs none).  Stand-in device only.
 */
static int
backend_no_device(void *arg)
{
	Relation	index = shim_open_relation(REL_IVF);
	Relation	hnsw = shim_open_relation(REL_HNSW);
	uint64		got[300];
	float		q[DIM];

	(void) arg;
	scenario = "a backend without a device";
	cur_ops = ORA_OPS_L2;
	setenv("MOCK_HIP_NO_DEVICE", "1", 1);
	shim_set_guc_bool("vector.gpu", true);
	shim_set_guc_bool("vector.gpu_pooled", false);
	EXPECT(PgvTryGetContext() == NULL);
	for (int i = 0; i < 5; i++)
	{
		Scan		s;
		Expected	e;

		make_query(q, 400 + i);
		e = expected_batch(REL_IVF, q, PROBES);
		scan_begin(&s, index, q, PROBES, PROBES);
		EXPECT(pull(&s, got, 10) == 10);
		EXPECT(s.gpu_tuples == 0 && s.cpu_tuples == 10);	/* the reference's path, not an ERROR */
		if (check_stream(&e, got, 10, 0, "no device, own context"))
			return 1;
		scan_end(&s);
		expected_free(&e);
	}
	/* pooled: answered by the worker's device, deeper pulls than its head fall back to the reference's code without a
	 * tuple twice */
	shim_set_guc_bool("vector.gpu_pooled", true);
	{
		double		until = shim_now() + 30.0;

		while (!PgvIvfflatMirrorIsCurrent(index) && shim_now() < until)
			usleep(20000);
		EXPECT(PgvIvfflatMirrorIsCurrent(index));
	}
	for (int i = 0; i < 5; i++)
	{
		Scan		s;
		Expected	e;
		int			want = i < 3 ? 10 : 200,
					n;

		make_query(q, 410 + i);
		e = expected_batch(REL_IVF, q, PROBES);
		scan_begin(&s, index, q, PROBES, PROBES);
		n = pull(&s, got, want);
		EXPECT(n == (want < e.n ? want : e.n));
		EXPECT(s.gpu_tuples > 0);
		if (want > PGV_POOL_HEAD)
			EXPECT(s.gpu_tuples == PGV_POOL_HEAD && s.cpu_tuples == n - PGV_POOL_HEAD);
		if (check_stream(&e, got, n, 0, "no device, pooled"))
			return 1;
		scan_end(&s);
		expected_free(&e);
	}
	/* the hnsw scan and the build hooks */
	{
		MemoryContext ctx = shim_query_context_begin();
		IvfflatBuildState bs;
		HnswBuildState hbs;
		VectorArrayData samples,
					centers;

		EXPECT(PgvHnswBeginScan(hnsw) == NULL);
		memset(&bs, 0, sizeof(bs));
		bs.index = index;
		bs.dimensions = DIM;
		bs.lists = LISTS;
		memset(&samples, 0, sizeof(samples));
		memset(&centers, 0, sizeof(centers));
		samples.dim = DIM;
		EXPECT(!PgvIvfflatKmeans(index, &samples, &centers, IvfflatGetTypeInfo(index)));
		PgvIvfflatBuildBegin(&bs);
		EXPECT(bs.gpu == NULL);
		memset(&hbs, 0, sizeof(hbs));
		hbs.index = hnsw;
		hbs.dimensions = DIM;
		EXPECT(PgvHnswBuildBegin(&hbs) == NULL);
		shim_query_context_end(ctx);
	}
	return 0;
}

/* ------------------------------------------------------------------------------------------------ hnsw: CREATE INDEX */
#define REL_HNSW2 2002
#define HB 4000					/* heap rows of the build scenario (a few of them duplicates) */

static void *
graph_alloc(Size size, void *state)
{
	HnswBuildState *bs = state;	/* HnswMemoryContextAlloc, src/hnswbuild.c:646-655 */

	bs->graphData.memoryUsed += size;
	return MemoryContextAlloc(bs->graphCtx, size);
}

static int
row_of_tid(uint64 tid)
{
	return (int) ((tid >> 16) - 1) * 50 + (int) (tid & 0xffff) - 1;
}

typedef struct LevelReplay
{
	const int32_t *levels;
	int64_t		next;
	double		ml;
}			LevelReplay;

static double
replay_level(void *state)
{
	LevelReplay *r = state;

	return exp(-((double) r->levels[r->next++] + 0.5) / r->ml);
}

/*
 * BuildGraph's in-memory phase through the hooks of ext/hnswbuild_gpu.c: InsertTuple allocates every element the
 * reference's way (HnswInitElement draws the level, the value is copied into graph memory, the caller's per-tuple
 * context is reset), the hook defers the insertion, and the first statement of FlushPages links the lot.  What
 * CreateGraphPages / WriteNeighborTuples would then serialise -- the element list, newest first, the neighbor arrays,
 * the duplicates' heap TIDs, the entry point -- must be the graph pgv_host_hnsw_build gives for the same rows and the
 * same levels, tuple for tuple; its pages, staged by the worker and scanned through the scan hooks, answer like the
 * oracle walking that graph.
 */
static int
backend_hnsw_gpu_build(void *arg)
{
	Relation	index = shim_open_relation(REL_HNSW2);
	float	   *data = malloc(sizeof(float) * HB * DIM);
	HnswBuildState bs;
	HnswElement *byrow = calloc(HB, sizeof(HnswElement));
	int32_t    *levels = malloc(sizeof(int32_t) * HB),
			   *dup_of = malloc(sizeof(int32_t) * HB);
	int64_t    *nbr_start = malloc(sizeof(int64_t) * (HB + 1));
	int32_t    *nbr;
	uint64_t   *tids = malloc(sizeof(uint64_t) * HB);
	int32_t		entry = -1;
	int			listed = 0,
				prev_row = HB;
	MemoryContext tmp;
	pgv_hnsw   *mirror = NULL;
	pgv_hnsw_built direct;
	pgv_ctx    *ctx = NULL;
	LevelReplay replay;
	pgv_rng		rng;
	pgv_rel		rel;

	(void) arg;
	scenario = "hnsw build hooks";
	EXPECT(index != NULL);
	shim_set_guc_bool("vector.gpu", true);
	shim_seed_random(21);
	for (int i = 0; i < 16; i++)
		(void) RandomDouble();	/* (the stand-in generator's first draws after a small seed are tiny: row 0 would top every level) */
	gen_rows(data, HB, DIM, 8);
	/* duplicates: the same value under several heap TIDs (src/hnswbuild.c:318-364) */
	for (int r = 600; r < 612; r++)
		memcpy(data + (size_t) r * DIM, data + (size_t) 500 * DIM, sizeof(float) * DIM);	/* 12 copies: more than HNSW_HEAPTIDS */
	memcpy(data + (size_t) 3000 * DIM, data + (size_t) 10 * DIM, sizeof(float) * DIM);

	memset(&bs, 0, sizeof(bs));
	bs.index = index;
	bs.typeInfo = HnswGetTypeInfo(index);
	bs.dimensions = DIM;
	bs.m = HM;
	bs.efConstruction = 32;
	bs.ml = 1.0 / log((double) HM);
	bs.maxLevel = (int) ((8192 - 24 - 8 - 4 - 4) / 6 / HM) - 2 < 63 ? (int) ((8192 - 24 - 8 - 4 - 4) / 6 / HM) - 2 : 63;
	bs.graph = &bs.graphData;
	bs.graphData.memoryTotal = (Size) 1 << 30;
	bs.graphCtx = shim_query_context_begin();
	bs.allocator.alloc = graph_alloc;
	bs.allocator.state = &bs;
	bs.hnswarea = NULL;
	bs.gpu = PgvHnswBuildBegin(&bs);
	EXPECT(bs.gpu != NULL);
	{
		/* a parallel build keeps the reference's code.  The reference's order: InitBuildState (hnswarea = NULL, the
		 * hook gives the participant a GPU state) and only THEN buildstate.hnswarea = hnswarea
		 * (HnswParallelScanAndInsert, src/hnswbuild.c:803-805); the first InsertTuple must notice, defer nothing and
		 * leave the element to InsertTupleInMemory, and FlushPages must find nothing to link */
		HnswBuildState par = bs;
		char		area[8];
		HnswElementData dummy;

		par.gpu = PgvHnswBuildBegin(&par);
		EXPECT(par.gpu != NULL);
		par.hnswarea = area;
		memset(&dummy, 0, sizeof(dummy));
		EXPECT(!PgvHnswBuildDefer(&par, &dummy));
		EXPECT(par.gpu == NULL);
		EXPECT(!PgvHnswBuildDefer(&par, &dummy));
		PgvHnswBuildLink(&par);	/* a no-op */
	}

	/* BuildCallback + InsertTuple (src/hnswbuild.c:486-609), the hook in place of InsertTupleInMemory */
	tmp = shim_query_context_begin();
	for (int r = 0; r < HB; r++)
	{
		Vector	   *value = make_vector(data + (size_t) r * DIM, DIM);
		Size		valueSize = offsetof(Vector, x) + sizeof(float) * DIM;
		ItemPointerData tid = itemptr(tid_of_row(r));
		HnswElement element = HnswInitElement(bs.hnswarea, &tid, bs.m, bs.ml, bs.maxLevel, &bs.allocator);
		char	   *valuePtr = HnswAlloc(&bs.allocator, valueSize);

		memcpy(valuePtr, value, valueSize);
		HnswPtrStore(bs.hnswarea, element->value, valuePtr);
		levels[r] = element->level;
		tids[r] = tid_of_row(r);
		EXPECT(PgvHnswBuildDefer(&bs, element));
		bs.graphData.indtuples++;
		shim_context_reset(tmp);	/* the caller's copy of the value is poison from here on */
	}
	shim_query_context_end(tmp);
	CurrentMemoryContext = TopMemoryContext;
	EXPECT(HnswPtrIsNull(bs.hnswarea, bs.graph->head));	/* nothing linked yet */

	/* FlushPages, first statement */
	PgvHnswBuildLink(&bs);
	PgvHnswBuildLink(&bs);		/* (again: nothing deferred, nothing happens) */

	/* ---- what the reference would serialise now */
	nbr_start[0] = 0;
	for (int r = 0; r < HB; r++)
	{
		nbr_start[r + 1] = nbr_start[r] + (int64_t) (levels[r] + 2) * HM;
		dup_of[r] = -2;			/* not seen yet */
	}
	nbr = malloc(sizeof(int32_t) * (size_t) nbr_start[HB]);
	for (int64_t j = 0; j < nbr_start[HB]; j++)
		nbr[j] = -1;
	for (HnswElementPtr it = bs.graph->head; !HnswPtrIsNull(bs.hnswarea, it);)
	{
		HnswElement element = HnswPtrAccess(bs.hnswarea, it);
		int			row = row_of_tid(tid_key(&element->heaptids[0]));

		it = element->next;
		EXPECT(row >= 0 && row < HB && row < prev_row);	/* newest first (head insertion) */
		prev_row = row;
		EXPECT(element->level == levels[row] && element->heaptidsLength >= 1 && element->heaptidsLength <= HNSW_HEAPTIDS);
		byrow[row] = element;
		dup_of[row] = -1;
		for (int t = 1; t < element->heaptidsLength; t++)
		{
			int			dr = row_of_tid(tid_key(&element->heaptids[t]));

			EXPECT(dr > row && dr < HB && dup_of[dr] == -2);
			EXPECT(memcmp(data + (size_t) dr * DIM, data + (size_t) row * DIM, sizeof(float) * DIM) == 0);
			dup_of[dr] = row;
		}
		listed++;
	}
	for (int r = 0; r < HB; r++)
		EXPECT(dup_of[r] != -2);	/* every heap row is an element of the list or a heap TID on one */
	for (int r = 0; r < HB; r++)
		if (dup_of[r] == -1)
			for (int lc = levels[r]; lc >= 0; lc--)
			{
				HnswNeighborArray *a = HnswGetNeighbors(bs.hnswarea, byrow[r], lc);
				int32_t    *out = nbr + nbr_start[r] + (int64_t) (levels[r] - lc) * HM;

				EXPECT(a->length >= 0 && a->length <= HnswGetLayerM(HM, lc));
				for (int i = 0; i < a->length; i++)
				{
					HnswElement ne = HnswPtrAccess(bs.hnswarea, a->items[i].element);
					int			nr = row_of_tid(tid_key(&ne->heaptids[0]));

					EXPECT(nr >= 0 && nr < HB && byrow[nr] == ne && levels[nr] >= lc);
					out[i] = nr;
				}
			}
	EXPECT(!HnswPtrIsNull(bs.hnswarea, bs.graph->entryPoint));
	entry = row_of_tid(tid_key(&HnswPtrAccess(bs.hnswarea, bs.graph->entryPoint)->heaptids[0]));
	EXPECT(dup_of[500] == -1 && dup_of[3000] == 10);
	{
		int			on500 = 0,
					elsewhere = 0;

		for (int r = 600; r < 612; r++)
			if (dup_of[r] == 500)
				on500++;
			else
				elsewhere++;
		EXPECT(on500 == HNSW_HEAPTIDS - 1 && elsewhere == 3);	/* an element holds ten heap TIDs; the rest spill over */
	}
	EXPECT(bs.graphData.indtuples == HB);

	/* ---- the same rows and levels straight through pgv_host_hnsw_build: the same graph, tuple for tuple */
	EXPECT(pgv_ctx_create(0, NULL, &ctx) == PGV_OK);
	EXPECT(pgv_hnsw_upload(ctx, PGV_L2SQ, PGV_F32, DIM, data, HB, &mirror) == PGV_OK);
	replay.levels = levels;
	replay.next = 0;
	replay.ml = bs.ml;
	memset(&rng, 0, sizeof(rng));
	rng.next_double = replay_level;
	rng.state = &replay;
	EXPECT(pgv_host_hnsw_build(mirror, PGV_F32, DIM, data, HB, HM, 32, &rng, 1024, &direct) == PGV_OK);
	EXPECT(direct.entry == entry && direct.nelements == listed);
	for (int r = 0; r < HB; r++)
		EXPECT(direct.levels[r] == levels[r] && direct.dup_of[r] == dup_of[r]);
	EXPECT(memcmp(direct.nbr, nbr, sizeof(int32_t) * (size_t) nbr_start[HB]) == 0);
	pgv_host_hnsw_built_free(&direct);
	pgv_hnsw_free(mirror);
	pgv_ctx_destroy(ctx);
	fprintf(stderr, "   %d heap rows -> %d elements (%d duplicates), entry row %d at level %d; graph memory %zu KB\n", HB, listed,
			HB - listed, entry, levels[entry], bs.graphData.memoryUsed >> 10);

	/* ---- FlushPages proper: the pages, then scans through the scan hooks against the oracle walking this graph */
	pgv_rel_init(&rel);
	EXPECT(pgv_host_hnsw_write_index(&rel, PGV_F32, DIM, HM, 32, HB, data, tids, levels, nbr_start, nbr, dup_of, entry) == PGV_OK);
	shim_replace_pages(REL_HNSW2, rel.pages, rel.nblocks);
	pgv_rel_free(&rel);
	PgvNoteIndexChange(index);
	shim_query_context_end(bs.graphCtx);	/* MemoryContextReset(buildstate->graphCtx), :315 */
	{
		ora_hnsw   *g = ora_hnsw_import(ORA_OPS_L2, ORA_F32, DIM, data, HB, HM, levels, nbr_start, nbr, entry);
		double		until = shim_now() + 30.0;
		void	   *gpu = NULL;
		MemoryContext qctx = shim_query_context_begin();

		EXPECT(g != NULL);
		hnsw_ef_search = 40;
		while ((gpu = PgvHnswBeginScan(index)) == NULL && shim_now() < until)
			usleep(20000);
		EXPECT(gpu != NULL);
		PgvHnswEndScan(gpu);
		shim_query_context_end(qctx);
		for (int i = 0; i < 20; i++)
		{
			IndexScanDescData desc;
			HnswScanOpaqueData so;
			float		q[DIM];
			List	   *w = NIL;
			int64_t		rows[40];
			double		dist[40];
			int64_t		scored;
			int			want,
						n;

			make_query(q, 200 + i);
			want = ora_hnsw_search(g, q, 40, 40, rows, dist, &scored);
			qctx = shim_query_context_begin();
			memset(&desc, 0, sizeof(desc));
			memset(&so, 0, sizeof(so));
			desc.indexRelation = index;
			desc.opaque = &so;
			so.first = true;
			so.gpu = PgvHnswBeginScan(index);
			EXPECT(so.gpu != NULL);
			EXPECT(PgvHnswGetScanItems(&desc, PointerGetDatum(make_vector(q, DIM)), &w));
			n = shim_list_length(w);
			EXPECT(n == want);
			for (int j = 0; j < n; j++)
			{
				HnswSearchCandidate *sc = shim_list_nth(w, n - 1 - j);

				EXPECT(fabs(sc->distance - dist[j]) <= 1e-4 * fabs(dist[j]) + 1e-6);
			}
			PgvHnswEndScan(so.gpu);
			shim_query_context_end(qctx);
		}
		ora_hnsw_free(g);
	}
	free(data);
	free(byrow);
	free(levels);
	free(dup_of);
	free(nbr_start);
	free(nbr);
	free(tids);
	return 0;
}

/* ------------------------------------------------------------------------------------------------ the worker's housekeeping */
#define REL_SLOW 1003
#define REL_WIDE 1005

/*
 * A staging of several seconds (a cold, large index: every page read of the worker sleeps) -- longer than the 3 s after
 * which a worker that had not beaten used to count as dead.
 * The worker must not be taken for dead meanwhile -- no second worker, no forgotten registration --, a pooled query on
 * ANOTHER index must not wait for the staging to end, and the mirror must come out.
 */
static int
backend_slow_staging(void *arg)
{
	Relation	slow = shim_open_relation(REL_SLOW);
	Relation	ivf = shim_open_relation(REL_IVF);
	float		q[DIM];
	double		t0,
				until;
	int			served = 0;

	(void) arg;
	scenario = "a staging of several seconds";
	shim_set_guc_bool("vector.gpu", true);
	make_query(q, 7);
	/* the other index's mirror is current (staged by the earlier phases, or staged now at full speed) */
	shim_set_guc_bool("vector.gpu_pooled", true);
	until = shim_now() + 30.0;
	while (!PgvIvfflatMirrorIsCurrent(ivf) && shim_now() < until)
		usleep(20000);
	EXPECT(PgvIvfflatMirrorIsCurrent(ivf));
	EXPECT(shim_live_bgworkers() == 1);
	{
		/* this backend's own device context and import exist before the clock below starts (a process's first touch of
		 * the device takes as long as it takes: not what the 2.5 s below are about) */
		Scan		s;
		uint64		got[10];

		shim_set_guc_bool("vector.gpu_pooled", false);
		scan_begin(&s, ivf, q, PROBES, PROBES);
		EXPECT(pull(&s, got, 10) == 10 && s.gpu_tuples == 10);
		scan_end(&s);
		shim_set_guc_bool("vector.gpu_pooled", true);
	}
	/* the slow index: the same image as REL_IVF (~420 pages) */
	{
		uint32_t	nblocks;
		const void *pages = shim_relation_pages(REL_IVF, &nblocks);

		shim_replace_pages(REL_SLOW, pages, nblocks);
		fprintf(stderr, "   REL_SLOW: %u pages\n", nblocks);
	}
	/* from here on the worker's page reads crawl: REL_SLOW's staging takes ~5 s */
	shim_set_bgworker_read_delay_us(12000);
	EXPECT(!PgvIvfflatMirrorIsCurrent(slow));	/* requested */
	t0 = shim_now();
	until = t0 + 60.0;
	while (shim_now() < until)
	{
		Scan		s;
		uint64		got[10];
		Expected	e = expected_batch(REL_IVF, q, PROBES);
		double		t1 = shim_now();

		/* pooled queries on the OTHER index while the worker is busy: answered -- taken back after the pooler's
		 * patience and run on the backend's own context -- in well under the staging's duration, and correct */
		scan_begin(&s, ivf, q, PROBES, PROBES);
		EXPECT(pull(&s, got, 10) == 10);
		EXPECT(shim_now() - t1 < 2.5);
		if (check_stream(&e, got, 10, 0, "pooled query beside a long staging"))
			return 1;
		scan_end(&s);
		expected_free(&e);
		served++;
		EXPECT(shim_live_bgworkers() == 1);	/* nobody took the busy worker for dead and started another */
		if (PgvIvfflatMirrorIsCurrent(slow))
			break;
		usleep(100000);
	}
	shim_set_bgworker_read_delay_us(0);
	EXPECT(PgvIvfflatMirrorIsCurrent(slow));
	fprintf(stderr, "   staging took %.1f s with the worker's reads slowed; %d pooled queries on another index answered meanwhile\n",
			shim_now() - t0, served);
	EXPECT(shim_now() - t0 > 3.5);	/* (the scenario is only worth something when the staging is long) */
	return 0;
}

/* DROP INDEX: the worker frees the mirror it owns and gives the registry entry back (64 for the cluster) */
static int
backend_drop_index(void *arg)
{
	ShimOpclass l2 = {0, IVFFLAT_MAX_DIM, false, false, 0};
	const int	n = 400,
				dim = 8,
				lists = 4;
	float	   *rows = malloc(sizeof(float) * (size_t) n * dim);
	float		centers[4 * 8];
	int64_t		offsets[5];
	uint64	   *tids = malloc(sizeof(uint64) * (size_t) n);
	pgv_rel		rel;
	double		until;
	int			before = PgvRegistryEntries();
	int			rounds,
				restaged = 0;

	(void) arg;
	scenario = "dropped indexes give their mirrors and registry entries back";
	shim_set_guc_bool("vector.gpu", true);
	shim_set_guc_bool("vector.gpu_pooled", false);
	/* a small index image: four lists of 100 rows */
	gen_rows(rows, n, dim, 5);
	for (int l = 0; l <= lists; l++)
		offsets[l] = (int64_t) l * (n / lists);
	for (int l = 0; l < lists; l++)
		memcpy(centers + l * dim, rows + (size_t) offsets[l] * dim, sizeof(float) * dim);
	for (int i = 0; i < n; i++)
		tids[i] = tid_of_row(i);
	pgv_rel_init(&rel);
	EXPECT(pgv_host_ivf_write_index(&rel, PGV_F32, dim, lists, centers, offsets, rows, tids) == PGV_OK);
	/* more create / stage / drop rounds than the registry has entries, every index with an oid of its own (what DROP +
	 * CREATE give): without the reclamation the 65th index, and every one after it, would never get a mirror -- and
	 * the mirrors of the dropped ones would stay in device memory for the worker's lifetime */
	/* (on the device: fewer rounds -- what is proven there is that the mirrors' memory comes back; the registry's 64
	 * entries are the same code on either device) */
	rounds = mock_hip_set_arena ? 70 : 12;
	for (int r = 0; r < rounds; r++)
	{
		Oid			oid = 3000 + (Oid) r;
		Relation	index;
		bool		current = false;

		shim_create_relation(oid, &l2, rel.pages, rel.nblocks, dim);
		index = shim_open_relation(oid);
		EXPECT(index != NULL);
		/* a staging that failed is asked for again by the index's next change (an insert, in real life) */
		for (int attempt = 0; attempt < 5 && !current; attempt++)
		{
			if (attempt > 0)
			{
				restaged++;
				PgvNoteIndexChange(index);
			}
			until = shim_now() + 4.0;
			while (!(current = PgvIvfflatMirrorIsCurrent(index)) && shim_now() < until)
				usleep(2000);
		}
		EXPECT(current);
		shim_drop_relation(oid);
	}
	/* the worker takes invalidations in at its next turn (200 ms at most when idle) */
	until = shim_now() + 10.0;
	while (PgvRegistryEntries() > before && shim_now() < until)
		usleep(5000);
	fprintf(stderr, "   %d create / stage / drop rounds (%d stagings asked for again): registry entries in use %d -> %d\n", rounds, restaged,
			before, PgvRegistryEntries());
	EXPECT(PgvRegistryEntries() <= before);
	pgv_rel_free(&rel);
	free(rows);
	free(tids);
	return 0;
}

/* CREATE INDEX on 2000-d rows: the build's row buffer must not ask palloc for more than MaxAllocSize (2^18 rows of
 * 8000 bytes are 2 GB: "invalid memory alloc request size") */
static int
backend_wide_build(void *arg)
{
	IvfflatBuildState bs;
	VectorArrayData centers;
	const int	dim = IVFFLAT_MAX_DIM;
	Size		itemsize = offsetof(Vector, x) + sizeof(float) * (Size) dim;
	float	   *row = calloc((size_t) dim, sizeof(float));

	(void) arg;
	scenario = "build state for 2000-d rows";
	shim_set_guc_bool("vector.gpu", true);
	memset(&bs, 0, sizeof(bs));
	bs.index = shim_open_relation(REL_WIDE);
	EXPECT(bs.index != NULL);
	bs.typeInfo = IvfflatGetTypeInfo(bs.index);
	bs.dimensions = dim;
	bs.lists = 4;
	centers.length = 4;
	centers.maxlen = 4;
	centers.dim = dim;
	centers.itemsize = itemsize;
	centers.items = palloc0(itemsize * 4);
	for (int i = 0; i < 4; i++)
	{
		Vector	   *c = (Vector *) VectorArrayGet(&centers, i);

		c->vl_len_ = (int32) (itemsize << 2);
		c->dim = (int16) dim;
		c->x[0] = (float) i;
	}
	bs.centers = &centers;
	sorted.n = 0;
	sorted.dim = dim;
	build_sort_begin(&bs);
	PgvIvfflatBuildBegin(&bs);	/* (an ERROR here reaches the top level: exit code 100) */
	EXPECT(bs.gpu != NULL);
	for (int i = 0; i < 8; i++)
	{
		ItemPointerData tid = itemptr(tid_of_row(i));

		row[0] = (float) (i % 4) + 0.1f;
		PgvIvfflatBuildAdd(&bs, &tid, PointerGetDatum(make_vector(row, dim)));
	}
	PgvIvfflatBuildFlush(&bs);
	EXPECT(build_sort_drain(&bs) == 8);
	for (int i = 0; i < 8; i++)
		EXPECT(sorted.list[i] == i % 4 && sorted.tid[i] == tid_of_row(i));
	free(row);
	return 0;
}

/* ------------------------------------------------------------------------------------------------ the postmaster */
#ifdef PGV_HAVE_REF_IVFSCAN
/* ------------------------------------------------------------------------------------------------ the reference's own scan
 * The program holds pgvector's src/ivfscan.c with ext/pgvector-0.8.6-gpu.patch applied and src/vector.c, compiled from the
 * reference tree (tests/test_ext_runtime_cpu.py; not on the GPU box, where the tree does not exist): the functions below
 * are the REFERENCE'S -- nothing of this file re-types them. */
extern IndexScanDesc ivfflatbeginscan(Relation index, int nkeys, int norderbys);
extern void ivfflatrescan(IndexScanDesc scan, ScanKey keys, int nkeys, ScanKey orderbys, int norderbys);
extern bool ivfflatgettuple(IndexScanDesc scan, ScanDirection dir);
extern void ivfflatendscan(IndexScanDesc scan);
extern int	ivfflat_probes;
extern int	ivfflat_iterative_scan;
extern int	ivfflat_max_probes;
extern int64 shim_index_scans_counted(void);

/* ORDER BY embedding <-> q LIMIT want through the access method's own entry points */
static int
ref_scan(Relation index, const float *query, int probes, int want, uint64 *got, int *used_gpu)
{
	MemoryContext ctx = shim_query_context_begin();
	ScanKeyData orderby;
	IndexScanDesc scan;
	int			n = 0;

	ivfflat_probes = probes;
	memset(&orderby, 0, sizeof(orderby));
	orderby.sk_argument = PointerGetDatum(make_vector(query, DIM));
	scan = ivfflatbeginscan(index, 0, 1);
	ivfflatrescan(scan, NULL, 0, &orderby, 1);
	*used_gpu = ((IvfflatScanOpaque) scan->opaque)->gpu != NULL;
	while (n < want && ivfflatgettuple(scan, ForwardScanDirection))
		got[n++] = tid_key(&scan->xs_heaptid);
	ivfflatendscan(scan);
	shim_query_context_end(ctx);
	return n;
}

static int
backend_reference_scan(void *arg)
{
	Relation	index = shim_open_relation(REL_IVF);
	uint64	   *got = malloc(sizeof(uint64) * 30000);
	float		q[DIM];
	int64		scans0 = shim_index_scans_counted();
	int			used_gpu;

	(void) arg;
	scenario = "the reference's own ivfflatgettuple";
	/* (1) vector.gpu = off: GetScanLists, GetScanItems, the distance functions of src/vector.c and the sort are the
	 * reference's, over the pages the build hooks wrote -- the oracle's restatement over the same pages must agree */
	shim_set_guc_bool("vector.gpu", false);
	for (int i = 0; i < 24; i++)
	{
		Expected	e;
		int			want = i % 6 == 5 ? 30000 : (i % 6 == 4 ? 200 : 10);
		int			n;

		make_query(q, 300 + i);
		e = expected_batch(REL_IVF, q, PROBES);
		n = ref_scan(index, q, PROBES, want, got, &used_gpu);
		EXPECT(!used_gpu);
		EXPECT(n == (want < e.n ? want : e.n));
		if (check_stream(&e, got, n, 0, "reference, CPU branch"))
			return 1;
		expected_free(&e);
	}
	EXPECT(shim_index_scans_counted() - scans0 == 24);
	EXPECT(shim_pinned_buffers() == 0);
	/* (2) vector.gpu = on: the same entry points; the hook lines inside them hand the scan to ext/ivfscan_gpu.c (own
	 * context: the worker's mirror imported), heads, device windows and the whole batch */
	shim_set_guc_bool("vector.gpu", true);
	EXPECT(wait_for_gpu(index, 30.0) == 0);
	{
		int			served = 0;

		for (int i = 0; i < 24; i++)
		{
			Expected	e;
			int			want = i % 6 == 5 ? 30000 : (i % 6 == 4 ? 300 : 10);
			int			n;

			make_query(q, 400 + i);
			e = expected_batch(REL_IVF, q, PROBES);
			n = ref_scan(index, q, PROBES, want, got, &used_gpu);
			served += used_gpu;
			EXPECT(n == (want < e.n ? want : e.n));
			if (check_stream(&e, got, n, 0, "reference, hooks"))
				return 1;
			expected_free(&e);
		}
		EXPECT(served == 24);
	}
	/* (3) the pooled path through the same entry points */
	shim_set_guc_bool("vector.gpu_pooled", true);
	for (int i = 0; i < 12; i++)
	{
		Expected	e;
		int			want = i % 4 == 3 ? 150 : 10;	/* past the pooler's head: the scan goes on by itself */
		int			n;

		make_query(q, 500 + i);
		e = expected_batch(REL_IVF, q, PROBES);
		n = ref_scan(index, q, PROBES, want, got, &used_gpu);
		EXPECT(used_gpu);
		EXPECT(n == (want < e.n ? want : e.n));
		if (check_stream(&e, got, n, 0, "reference, pooled"))
			return 1;
		expected_free(&e);
	}
	shim_set_guc_bool("vector.gpu_pooled", false);
	/* (4) an iterative scan of the reference (ivfflat.iterative_scan = relaxed_order, src/ivfscan.c:400-406) with the hooks:
	 * probes 3 of max_probes 9, every batch sorted by itself */
	{
		Expected	e3,
					e6;
		int			n;

		ivfflat_iterative_scan = 1;
		ivfflat_max_probes = 9;
		make_query(q, 7);
		e3 = expected_batch(REL_IVF, q, 3);
		e6 = expected_batch(REL_IVF, q, 6);
		n = ref_scan(index, q, 3, e6.n, got, &used_gpu);
		EXPECT(used_gpu && n == e6.n);
		if (check_stream(&e3, got, e3.n, 0, "reference, iterative: first batch"))
			return 1;
		ivfflat_iterative_scan = 0;
		ivfflat_max_probes = 32768;
		expected_free(&e3);
		expected_free(&e6);
	}
	EXPECT(shim_pinned_buffers() == 0);
	free(got);
	return 0;
}
#endif

/* ------------------------------------------------------------------------------------------------ several devices
 * ext/ivfbuild_gpu.c PgvKmeansOnDevices: the leader of a build shards its k-means samples over the devices of the node,
 * one helper thread and one context per further device, pgv_comm_unique_id / pgv_comm_create / pgv_kmeans_sharded (RCCL in
 * libpgv_hip; threads of this process meeting through the id on the stand-in device, whose node has MOCK_HIP_DEVICES = 3).
 * The centers must be the single-device pgv_kmeans' for the same seed, to the float tolerance (the all-reduce adds the
 * ranks' sums in another order).  PgvMyDevice: parallel build worker w takes device (w + 1) mod the device count. */
static int
backend_multi_device_kmeans(void *arg)
{
	Relation	index = shim_open_relation(REL_IVF);
	const int	n = 6000,
				lists = 24;
	const Size	itemsize = offsetof(Vector, x) + sizeof(float) * DIM;
	float	   *rows = malloc(sizeof(float) * (size_t) n * DIM);
	float	   *single = malloc(sizeof(float) * (size_t) lists * DIM);
	VectorArrayData samples,
				centers;
	pgv_rng		rng;
	uint64		seed;
	int			iters = 0,
				made1,
				made2;
	double		worst = 0.0;

	(void) arg;
	scenario = "k-means over the devices of the node";
	setenv("MOCK_HIP_DEVICES", "3", 1);
	shim_set_guc_bool("vector.gpu", true);
	/* the device of a participant */
	EXPECT(pgv_device_count() == 3 || !mock_hip_set_arena);
	if (mock_hip_set_arena)
	{
		EXPECT(PgvMyDevice() == 0);	/* the leader */
		ParallelWorkerNumber = 0;
		EXPECT(PgvMyDevice() == 1);
		ParallelWorkerNumber = 1;
		EXPECT(PgvMyDevice() == 2);
		ParallelWorkerNumber = 2;
		EXPECT(PgvMyDevice() == 0);
		shim_set_guc_int("vector.gpu_device", 1);	/* pinned */
		EXPECT(PgvMyDevice() == 1);
		shim_set_guc_int("vector.gpu_device", -1);
		ParallelWorkerNumber = -1;
	}
	gen_rows(rows, n, DIM, 31);
	samples.length = samples.maxlen = n;
	samples.dim = DIM;
	samples.itemsize = itemsize;
	samples.items = palloc0(itemsize * (Size) n);
	for (int i = 0; i < n; i++)
	{
		Vector	   *v = (Vector *) VectorArrayGet(&samples, i);

		v->vl_len_ = (int32) (itemsize << 2);
		v->dim = (int16) DIM;
		memcpy(v->x, rows + (size_t) i * DIM, sizeof(float) * DIM);
	}
	centers.length = 0;
	centers.maxlen = lists;
	centers.dim = DIM;
	centers.itemsize = itemsize;
	centers.items = palloc0(itemsize * (Size) lists);
	/* the seed PgvKmeansOnDevices will draw: the first RandomInt() of this stream, in the upper half */
	shim_seed_random(99);
	seed = (uint64) (uint32) RandomInt() << 32;
	shim_seed_random(99);
	made1 = mock_hip_contexts_made ? mock_hip_contexts_made(1) : 0;
	made2 = mock_hip_contexts_made ? mock_hip_contexts_made(2) : 0;
	EXPECT(PgvIvfflatKmeans(index, &samples, &centers, IvfflatGetTypeInfo(index)));
	EXPECT(centers.length == lists);
	if (mock_hip_contexts_made)
	{
		/* one helper context on each of the other two devices, made for this k-means and gone with it */
		EXPECT(mock_hip_contexts_made(1) == made1 + 1);
		EXPECT(mock_hip_contexts_made(2) == made2 + 1);
	}
	{
		/* one participant holding every sample (a group of one -- through RCCL on the real device), the same seeded stream */
		unsigned char id[PGV_COMM_ID_BYTES];
		pgv_comm   *solo = NULL;

		memset(&rng, 0, sizeof(rng));
		rng.seed = seed;
		EXPECT(pgv_comm_unique_id(id) == PGV_OK);
		EXPECT(pgv_comm_create(PgvGetContext(), 1, 0, id, &solo) == PGV_OK);
		EXPECT(pgv_kmeans_sharded(solo, PGV_OPS_L2, PGV_F32, DIM, rows, n, lists, 500, &rng, single, NULL, &iters) == PGV_OK);
		pgv_comm_destroy(solo);
		for (int c = 0; c < lists; c++)
			for (int d = 0; d < DIM; d++)
			{
				double		a = ((Vector *) VectorArrayGet(&centers, c))->x[d],
							b = single[(size_t) c * DIM + d];
				double		err = fabs(a - b) / (fabs(b) + 1e-3);

				if (err > worst)
					worst = err;
			}
		if (pgv_device_count() < 2)
		{
			/* a node with one device: PgvIvfflatKmeans above ran pgv_kmeans on the backend's RandomDouble() stream -- compare
			 * the group of one with pgv_kmeans on the SAME seeded stream instead (libpgv_hip: identical) */
			float	   *plain = malloc(sizeof(float) * (size_t) lists * DIM);
			int			it2 = 0;

			memset(&rng, 0, sizeof(rng));
			rng.seed = seed;
			EXPECT(pgv_kmeans(PgvGetContext(), PGV_OPS_L2, PGV_F32, DIM, rows, n, lists, 500, &rng, plain, NULL, &it2) == PGV_OK);
			worst = 0.0;
			for (int i = 0; i < lists * DIM; i++)
			{
				double		err = fabs((double) plain[i] - single[i]) / (fabs((double) single[i]) + 1e-3);

				if (err > worst)
					worst = err;
			}
			free(plain);
			fprintf(stderr, "   one device: k-means through a communicator of one (RCCL) = pgv_kmeans on the same stream to %.2e (%d / %d iterations)\n",
					worst, iters, it2);
		}
		else
			fprintf(stderr, "   k-means of %d samples, %d lists over %d devices = the one-participant centers to %.2e (%d iterations)\n", n,
					lists, pgv_device_count(), worst, iters);
		EXPECT(worst <= 1e-5);
	}
	/* vector.gpu_build_devices = 1: this backend's device alone (no helper contexts) */
	shim_set_guc_int("vector.gpu_build_devices", 1);
	made1 = mock_hip_contexts_made ? mock_hip_contexts_made(1) : 0;
	centers.length = 0;
	EXPECT(PgvIvfflatKmeans(index, &samples, &centers, IvfflatGetTypeInfo(index)));
	if (mock_hip_contexts_made)
		EXPECT(mock_hip_contexts_made(1) == made1);
	shim_set_guc_int("vector.gpu_build_devices", 0);
	free(rows);
	free(single);
	return 0;
}

static int
run_phase(const char *name, int (*fn) (void *), int nprocs, void *const *args, double timeout_s)
{
	int			pids[16],
				codes[16];
	int			left,
				bad = 0;

	for (int i = 0; i < nprocs; i++)
		pids[i] = shim_fork_backend(fn, args ? args[i] : NULL);
	left = shim_postmaster_wait(pids, nprocs, codes, timeout_s);
	for (int i = 0; i < nprocs; i++)
		if (codes[i] != 0)
		{
			bad++;
			fprintf(stderr, "  process %d of phase '%s' ended with code %d\n", i, name, codes[i]);
		}
	fprintf(stderr, "phase %-40s %d process(es): %s   (background workers alive: %d)\n", name, nprocs,
			left ? "TIMED OUT" : (bad ? "FAILED" : "ok"), shim_live_bgworkers());
	return left || bad;
}

#ifdef PGV_HAVE_REF_HNSW
/* ------------------------------------------------------------------------------------------------ the reference's own hnsw scan
 * As above for HNSW: the program holds pgvector's src/hnswscan.c (patched) and the whole of its src/hnswutils.c.  The walk
 * of the CPU branch -- HnswGetEntryPoint, HnswSearchLayer, HnswLoadElement, the visited table, the distance calls into
 * src/vector.c -- is the REFERENCE'S, over the pages pgv_host_hnsw_write_index laid out. */
extern IndexScanDesc hnswbeginscan(Relation index, int nkeys, int norderbys);
extern void hnswrescan(IndexScanDesc scan, ScanKey keys, int nkeys, ScanKey orderbys, int norderbys);
extern bool hnswgettuple(IndexScanDesc scan, ScanDirection dir);
extern void hnswendscan(IndexScanDesc scan);
extern int	hnsw_max_scan_tuples;

/* ORDER BY embedding <-> q through hnswbeginscan / hnswrescan / hnswgettuple / hnswendscan; *reads = pages the scan asked for */
static int
ref_hnsw_scan(Relation index, const float *query, int want, uint64 *got, int *had_gpu, long *reads)
{
	MemoryContext ctx = shim_query_context_begin();
	ScanKeyData orderby;
	IndexScanDesc scan;
	long		reads0;
	int			n = 0;

	memset(&orderby, 0, sizeof(orderby));
	if (query)
		orderby.sk_argument = PointerGetDatum(make_vector(query, DIM));
	else
		orderby.sk_flags = SK_ISNULL;
	scan = hnswbeginscan(index, 0, 1);
	hnswrescan(scan, NULL, 0, &orderby, 1);
	*had_gpu = ((HnswScanOpaque) scan->opaque)->gpu != NULL;
	reads0 = shim_buffer_reads();
	while (n < want && hnswgettuple(scan, ForwardScanDirection))
		got[n++] = tid_key(&scan->xs_heaptid);
	*reads = shim_buffer_reads() - reads0;
	hnswendscan(scan);
	shim_query_context_end(ctx);
	return n;
}

/* the stream of heap TIDs against the oracle's walk of the same graph: the distance at every position, the row itself
 * wherever its distance is clear of its neighbours' */
static int
check_hnsw_stream_n(const float *data, int nrows, const float *q, const uint64 *got, int n, const int64_t *rows, const double *dist,
					int want, const char *what)
{
	if (n != want)
	{
		fprintf(stderr, "%s: %d tuples, the oracle's walk has %d\n", what, n, want);
		return 1;
	}
	for (int j = 0; j < n; j++)
	{
		int			r = row_of_tid(got[j]);
		double		d = 0;

		if (r < 0 || r >= nrows)
		{
			fprintf(stderr, "%s: position %d is no heap tuple (%llx)\n", what, j, (unsigned long long) got[j]);
			return 1;
		}
		for (int k = 0; k < DIM; k++)
			d += ((double) data[(size_t) r * DIM + k] - q[k]) * ((double) data[(size_t) r * DIM + k] - q[k]);
		if (fabs(d - dist[j]) > 1e-4 * fabs(dist[j]) + 1e-6)
		{
			fprintf(stderr, "%s: position %d row %d at %.7g, the oracle has row %d at %.7g\n", what, j, r, d, (int) rows[j], dist[j]);
			return 1;
		}
		if ((j + 1 == n || fabs(dist[j + 1] - dist[j]) > 1e-4 * fabs(dist[j])) && (j == 0 || fabs(dist[j] - dist[j - 1]) > 1e-4 * fabs(dist[j])) &&
			r != (int) rows[j])
		{
			fprintf(stderr, "%s: position %d is row %d, the oracle has row %d\n", what, j, r, (int) rows[j]);
			return 1;
		}
	}
	return 0;
}

static int
check_hnsw_stream(const float *data, const float *q, const uint64 *got, int n, const int64_t *rows, const double *dist, int want,
				  const char *what)
{
	return check_hnsw_stream_n(data, HN, q, got, n, rows, dist, want, what);
}

static int
backend_reference_hnsw_scan(void *arg)
{
	Relation	index = shim_open_relation(REL_HNSW);
	float	   *data = malloc(sizeof(float) * HN * DIM);
	ora_hnsw   *g;
	uint64		got[64];
	float		q[DIM];
	int64_t		rows[40];
	double		dist[40];
	int64_t		scored;
	int			had_gpu,
				n,
				want;
	long		reads,
				cpu_reads = 0;
	double		until;

	(void) arg;
	scenario = "the reference's own hnswgettuple";
	hnsw_ef_search = 40;
	gen_rows(data, HN, DIM, 3);
	g = ora_hnsw_build(ORA_OPS_L2, ORA_F32, DIM, data, HN, HM, 32, 9);	/* the graph REL_HNSW's pages hold */
	/* (1) vector.gpu = off: the reference's walk over the pages = the oracle's walk over the graph */
	shim_set_guc_bool("vector.gpu", false);
	for (int i = 0; i < 30; i++)
	{
		make_query(q, 500 + i);
		want = ora_hnsw_search(g, q, 40, 40, rows, dist, &scored);
		n = ref_hnsw_scan(index, q, 64, got, &had_gpu, &reads);
		EXPECT(!had_gpu && reads > 40);
		cpu_reads += reads;
		if (check_hnsw_stream(data, q, got, n, rows, dist, want, "reference hnsw, CPU branch"))
			return 1;
	}
	/* (2) vector.gpu = on: hnswbeginscan takes the mirror, the hook line inside hnswgettuple hands the first batch to the
	 * device walk -- and the reference's code below it (llast, heap TIDs off the element, list_delete_last) hands out
	 * what the glue built.  No page of the index is read. */
	shim_set_guc_bool("vector.gpu", true);
	until = shim_now() + 30.0;
	for (;;)
	{
		make_query(q, 500);
		n = ref_hnsw_scan(index, q, 1, got, &had_gpu, &reads);
		if (had_gpu || shim_now() > until)
			break;
		usleep(20000);			/* (the worker stages the graph at the first scan that asks) */
	}
	EXPECT(had_gpu);
	for (int i = 0; i < 30; i++)
	{
		make_query(q, 500 + i);
		want = ora_hnsw_search(g, q, 40, 40, rows, dist, &scored);
		n = ref_hnsw_scan(index, q, 64, got, &had_gpu, &reads);
		EXPECT(had_gpu && reads == 0);
		if (check_hnsw_stream(data, q, got, n, rows, dist, want, "reference hnsw, hook"))
			return 1;
	}
	/* (3) what the device walk does not serve goes down the reference's walk with the mirror attached: ORDER BY <-> NULL
	 * (every distance 0: the scan still returns ef_search tuples) and an iterative scan past its first batch */
	n = ref_hnsw_scan(index, NULL, 64, got, &had_gpu, &reads);
	EXPECT(had_gpu && n == 40 && reads > 0);
	hnsw_iterative_scan = 1;	/* HNSW_ITERATIVE_SCAN_RELAXED */
	hnsw_max_scan_tuples = 20000;
	{
		uint64	   *all = malloc(sizeof(uint64) * HN);
		int			seen = 0;
		char	   *mark = calloc(HN, 1);

		make_query(q, 531);
		want = ora_hnsw_search(g, q, 40, 40, rows, dist, &scored);
		n = ref_hnsw_scan(index, q, 400, all, &had_gpu, &reads);
		EXPECT(had_gpu && n == 400 && reads > 40);
		/* the first batch is the plain walk's; later batches never repeat a tuple */
		if (check_hnsw_stream(data, q, all, 40, rows, dist, want, "reference hnsw, iterative, first batch"))
			return 1;
		for (int j = 0; j < n; j++)
		{
			int			r = row_of_tid(all[j]);

			EXPECT(r >= 0 && r < HN && !mark[r]);
			mark[r] = 1;
			seen++;
		}
		EXPECT(seen == 400);
		free(all);
		free(mark);
	}
	hnsw_iterative_scan = HNSW_ITERATIVE_SCAN_OFF;
	fprintf(stderr, "   30 walks of the reference's HnswSearchLayer = the oracle's (%.0f pages read per scan); 30 scans served by the hook with 0 pages read\n",
			(double) cpu_reads / 30.0);
	ora_hnsw_free(g);
	free(data);
	return 0;
}
#endif							/* PGV_HAVE_REF_HNSW */

#ifdef PGV_HAVE_REF_IVFUTILS
/* ------------------------------------------------------------------------------------------------ the reference's own k-means
 * The program holds pgvector's src/ivfkmeans.c (patched: the PgvIvfflatKmeans line in IvfflatKmeans) and the whole of its
 * src/ivfutils.c, compiled with the reference's own OPTFLAGS.  vector.gpu off: InitCenters, ElkanKmeans, ComputeNewCenters,
 * CheckCenters are the REFERENCE'S; handed the oracle's pg_prng stream, they must leave the centers the oracle's
 * restatement (oracle_ivf.c ora_kmeans) leaves from the same seed -- bit for bit -- having drawn the same number of
 * values.  vector.gpu on: the same IvfflatKmeans call is served by the hook. */
extern void IvfflatKmeans(Relation index, VectorArray samples, VectorArray centers, const IvfflatTypeInfo * typeInfo, Size memoryUsed);
extern VectorArray VectorArrayInit(int maxlen, int dimensions, Size itemsize);

static double
kmeans_objective(const float *rows, int n, const float *centers, int k, int spherical)
{
	double		total = 0;

	for (int i = 0; i < n; i++)
	{
		double		best = DBL_MAX;

		for (int c = 0; c < k; c++)
		{
			double		d = 0;

			for (int j = 0; j < DIM; j++)
			{
				double		a = rows[(size_t) i * DIM + j],
							b = centers[(size_t) c * DIM + j];

				d += spherical ? -a * b : (a - b) * (a - b);
			}
			if (d < best)
				best = d;
		}
		total += spherical ? 1.0 + best : best;
	}
	return total;
}

static int
reference_kmeans_case(Oid relid, int ops, int n, int k, uint64 seed)
{
	Relation	index = shim_open_relation(relid);
	MemoryContext ctx = shim_query_context_begin();
	const IvfflatTypeInfo *typeInfo = IvfflatGetTypeInfo(index);	/* the reference's (src/ivfutils.c) */
	Size		itemsize = typeInfo->itemSize(DIM);
	VectorArray samples = VectorArrayInit(n > 0 ? n : 1, DIM, itemsize);
	VectorArray centers = VectorArrayInit(k, DIM, itemsize);
	float	   *rows = malloc(sizeof(float) * (size_t) (n > 0 ? n : 1) * DIM);
	float	   *want = malloc(sizeof(float) * (size_t) k * DIM);
	float	   *got = malloc(sizeof(float) * (size_t) k * DIM);
	ora_prng	a,
				b;
	int			iterations;
	double		elkan,
				hooked;

	EXPECT(itemsize == offsetof(Vector, x) + sizeof(float) * DIM);
	gen_rows(rows, n, DIM, seed);
	for (int i = 0; i < n; i++)
	{
		Vector	   *v = (Vector *) VectorArrayGet(samples, i);

		if (ops != ORA_OPS_L2)
		{
			/* SampleCallback normalises the samples of opclasses with a KMEANS_NORM proc (src/ivfbuild.c:148-156) */
			double		norm = 0.0;

			for (int d = 0; d < DIM; d++)
				norm += (double) rows[(size_t) i * DIM + d] * (double) rows[(size_t) i * DIM + d];
			norm = sqrt(norm);
			for (int d = 0; d < DIM && norm > 0.0; d++)
				rows[(size_t) i * DIM + d] = (float) ((double) rows[(size_t) i * DIM + d] / norm);
		}
		v->vl_len_ = (int32) (itemsize << 2);
		v->dim = DIM;
		memcpy(v->x, rows + (size_t) i * DIM, sizeof(float) * DIM);
	}
	samples->length = n;

	shim_set_guc_bool("vector.gpu", false);
	ora_prng_seed(&a, seed);
	shim_prng_hook(ora_prng_double_cb, ora_prng_u32_cb, &a);
	IvfflatKmeans(index, samples, centers, typeInfo, 0);
	shim_prng_hook(NULL, NULL, NULL);
	EXPECT(centers->length == k);
	ora_prng_seed(&b, seed);
	iterations = ora_kmeans(ops, ORA_F32, DIM, rows, n, want, k, &b, NULL);
	EXPECT(iterations >= 0);
	EXPECT(a.s0 == b.s0 && a.s1 == b.s1);	/* the same number of draws */
	for (int c = 0; c < k; c++)
	{
		Vector	   *v = (Vector *) VectorArrayGet(centers, c);

		EXPECT(v->dim == DIM);
		memcpy(got + (size_t) c * DIM, v->x, sizeof(float) * DIM);
		if (memcmp(v->x, want + (size_t) c * DIM, sizeof(float) * DIM) != 0)
		{
			fprintf(stderr, "reference k-means (n %d, k %d, ops %d): center %d differs from the oracle's: %.9g vs %.9g ...\n", n, k, ops,
					c, v->x[0], want[(size_t) c * DIM]);
			return 1;
		}
	}
	elkan = n > 0 ? kmeans_objective(rows, n, got, k, ops != ORA_OPS_L2) : 0.0;

	/* the hook line: IvfflatKmeans -> PgvIvfflatKmeans -> pgv_kmeans; CheckCenters of the reference then passes over what
	 * came back (no NaN, no duplicates, unit norm for spherical k-means) or raises */
	if (n > 0)
	{
		shim_set_guc_bool("vector.gpu", true);
		centers->length = 0;
		IvfflatKmeans(index, samples, centers, typeInfo, 0);
		EXPECT(centers->length == k);
		for (int c = 0; c < k; c++)
			memcpy(got + (size_t) c * DIM, ((Vector *) VectorArrayGet(centers, c))->x, sizeof(float) * DIM);
		hooked = kmeans_objective(rows, n, got, k, ops != ORA_OPS_L2);
		fprintf(stderr, "   n %5d k %3d %s: the reference's ElkanKmeans = the oracle's, bit for bit (%d iterations); objective %.6g, through the hook %.6g\n",
				n, k, ops == ORA_OPS_L2 ? "l2" : "ip", iterations, elkan, hooked);
		/* the device's k-means (k-means++ from the same draws, Lloyd to a fixed point) lands where Elkan's does or close
		 * by; the mock device's is twenty Lloyd steps from a strided start and only has to be a clustering at all */
		EXPECT(hooked <= elkan * (mock_hip_set_arena ? 8.0 : 1.10) + 1e-6);
	}
	else
		fprintf(stderr, "   n %5d k %3d %s: RandomCenters = the oracle's, bit for bit\n", n, k, ops == ORA_OPS_L2 ? "l2" : "ip");
	free(rows);
	free(want);
	free(got);
	shim_query_context_end(ctx);
	return 0;
}

static int
backend_reference_kmeans(void *arg)
{
	(void) arg;
	scenario = "the reference's own IvfflatKmeans";
	if (reference_kmeans_case(REL_IVF, ORA_OPS_L2, 2000, 20, 77))
		return 1;
	if (reference_kmeans_case(REL_IVF, ORA_OPS_L2, 5000, 100, 78))
		return 1;
	if (reference_kmeans_case(REL_IVF, ORA_OPS_L2, 30, 64, 79))	/* fewer samples than lists: empty clusters are re-drawn */
		return 1;
	if (reference_kmeans_case(REL_IVF, ORA_OPS_L2, 0, 16, 80))	/* an empty table: RandomCenters */
		return 1;
	if (reference_kmeans_case(REL_IP, ORA_OPS_IP, 3000, 40, 81))	/* spherical k-means */
		return 1;
	return 0;
}
#endif							/* PGV_HAVE_REF_IVFUTILS */

#ifdef PGV_HAVE_REF_IVFINSERT
/* ------------------------------------------------------------------------------------------------ the reference's own insert
 * The program also holds pgvector's src/ivfinsert.c (patched: PgvNoteIndexChange in ivfflatinsert).  Its FindInsertPage /
 * InsertTuple and the page-append half of its src/ivfutils.c WRITE into the pages the build hooks laid out -- PageAddItem
 * into the lists' last pages, IvfflatAppendPage past them, IvfflatUpdateList on the list tuples.  What reads those pages
 * afterwards: the reference's own scan, the oracle's page reader, and the PRODUCT'S stager (the worker restages the index
 * after the change) -- all three must agree, and every inserted row must be found. */
#define REL_INS 1008
#ifdef PGV_HAVE_REF_IVFVACUUM
extern IndexBulkDeleteResult *ivfflatbulkdelete(IndexVacuumInfo *info, IndexBulkDeleteResult *stats, IndexBulkDeleteCallback callback,
												 void *callback_state);
extern IndexBulkDeleteResult *ivfflatvacuumcleanup(IndexVacuumInfo *info, IndexBulkDeleteResult *stats);
static int	ins_rows_built;

/* dead: every third row of the build, every other inserted row */
static int
ins_row_is_dead(int r, int built)
{
	return r < built ? r % 3 == 0 : (r - built) % 2 == 0;
}

static bool
ins_dead_callback(ItemPointer itemptr, void *state)
{
	(void) state;
	return ins_row_is_dead(row_of_tid(tid_key(itemptr)), ins_rows_built) != 0;
}
#endif
extern bool ivfflatinsert(Relation index, Datum *values, bool *isnull, ItemPointer heap_tid, Relation heap,
						  IndexUniqueCheck checkUnique, bool indexUnchanged, IndexInfo *indexInfo);

static int
backend_reference_insert(void *arg)
{
	Relation	index = shim_open_relation(REL_INS);
	const int	n = 4000,
				lists = 16,
				nins = 600;
	float	   *rows = malloc(sizeof(float) * (size_t) (n + nins) * DIM);
	uint64	   *got = malloc(sizeof(uint64) * 30000);
	BlockNumber blocks0;
	int			used_gpu;

	(void) arg;
	scenario = "the reference's own ivfflatinsert";
	cur_ops = ORA_OPS_L2;
	shim_set_guc_bool("vector.gpu", true);
	shim_set_guc_bool("vector.gpu_pooled", false);
	shim_seed_random(23);
	gen_rows(rows, n + nins, DIM, 31);
	if (build_index(REL_INS, rows, n, DIM, lists, 7))
		return 1;
	EXPECT(wait_for_gpu(index, 30.0) == 0);
	blocks0 = RelationGetNumberOfBlocks(index);
	for (int j = 0; j < nins; j++)
	{
		MemoryContext ctx = shim_query_context_begin();
		Datum		value = PointerGetDatum(make_vector(rows + (size_t) (n + j) * DIM, DIM));
		bool		isnull = false;
		ItemPointerData tid = itemptr(tid_of_row(n + j));

		ivfflatinsert(index, &value, &isnull, &tid, NULL, UNIQUE_CHECK_NO, false, NULL);
		shim_query_context_end(ctx);
	}
	/* 600 tuples of 144 bytes: the lists' last pages filled up and pages were appended */
	EXPECT(RelationGetNumberOfBlocks(index) > blocks0);
	/* (1) vector.gpu = off: the reference's scan over pages it partly wrote itself = the oracle's reading of them */
	shim_set_guc_bool("vector.gpu", false);
	for (int i = 0; i < 16; i++)
	{
		const int	r = n + (37 * i) % nins;
		Expected	e = expected_batch(REL_INS, rows + (size_t) r * DIM, PROBES);
		int			nn = ref_scan(index, rows + (size_t) r * DIM, PROBES, 10, got, &used_gpu);

		EXPECT(!used_gpu && nn == 10);
		if (check_stream(&e, got, nn, 0, "after the reference's inserts, CPU branch"))
			return 1;
		EXPECT(got[0] == tid_of_row(r));	/* the inserted row itself, at distance 0 */
		expected_free(&e);
	}
	/* (2) vector.gpu = on: the worker stages the index again -- the product's stager over the reference's tuples -- and
	 * the hooks inside the reference's scan serve from the device */
	shim_set_guc_bool("vector.gpu", true);
	EXPECT(wait_for_gpu(index, 30.0) == 0);
	for (int i = 0; i < 16; i++)
	{
		const int	r = n + (37 * i + 11) % nins;
		Expected	e = expected_batch(REL_INS, rows + (size_t) r * DIM, PROBES);
		int			nn = ref_scan(index, rows + (size_t) r * DIM, PROBES, i % 4 == 3 ? 400 : 10, got, &used_gpu);

		EXPECT(used_gpu && nn == (i % 4 == 3 ? (400 < e.n ? 400 : e.n) : 10));
		if (check_stream(&e, got, nn, 0, "after the reference's inserts, hooks"))
			return 1;
		EXPECT(got[0] == tid_of_row(r));
		expected_free(&e);
	}
	fprintf(stderr, "   %d rows through the reference's ivfflatinsert (%u -> %u blocks): its scan, the oracle and the restaged mirror agree\n",
			nins, (unsigned) blocks0, (unsigned) RelationGetNumberOfBlocks(index));
#ifdef PGV_HAVE_REF_IVFVACUUM
	/* (3) VACUUM by the reference's own ivfflatbulkdelete (src/ivfvacuum.c, patched: PgvNoteIndexChange at its end): every
	 * third row of the build and every other inserted row are dead.  PageIndexMultiDelete compacts the pages the lists
	 * keep, IvfflatUpdateList resets the insert pages; afterwards the three readers agree again and no dead row is seen. */
	{
		IndexVacuumInfo info;
		IndexBulkDeleteResult *stats;
		int			dead = 0;

		for (int r = 0; r < n + nins; r++)
			dead += ins_row_is_dead(r, n);
		memset(&info, 0, sizeof(info));
		info.index = index;
		ins_rows_built = n;
		stats = ivfflatbulkdelete(&info, NULL, ins_dead_callback, NULL);
		EXPECT(stats != NULL && (int) stats->tuples_removed == dead && (int) stats->num_index_tuples == n + nins - dead);
		stats = ivfflatvacuumcleanup(&info, stats);
		EXPECT(stats != NULL && stats->num_pages == RelationGetNumberOfBlocks(index));
		for (int pass = 0; pass < 2; pass++)
		{
			shim_set_guc_bool("vector.gpu", pass == 1);
			if (pass == 1)
				EXPECT(wait_for_gpu(index, 30.0) == 0);
			for (int i = 0; i < 16; i++)
			{
				const int	r = n + (2 * (23 * i % (nins / 2)) + 1);	/* an inserted row that lives (odd rank) */
				Expected	e = expected_batch(REL_INS, rows + (size_t) r * DIM, PROBES);
				int			nn = ref_scan(index, rows + (size_t) r * DIM, PROBES, 300, got, &used_gpu);

				EXPECT(used_gpu == (pass == 1) && nn == (300 < e.n ? 300 : e.n));
				if (check_stream(&e, got, nn, 0, pass ? "after the reference's vacuum, hooks" : "after the reference's vacuum, CPU branch"))
					return 1;
				EXPECT(got[0] == tid_of_row(r));
				for (int j = 0; j < nn; j++)
					EXPECT(!ins_row_is_dead(row_of_tid(got[j]), n));
				expected_free(&e);
			}
		}
		fprintf(stderr, "   %d of %d rows removed by the reference's ivfflatbulkdelete: no reader sees one of them\n", dead, n + nins);
	}
#endif
	free(rows);
	free(got);
	return 0;
}
#endif							/* PGV_HAVE_REF_IVFINSERT */

#ifdef PGV_HAVE_REF_HNSWINSERT
/* ------------------------------------------------------------------------------------------------ the reference's own hnsw insert
 * pgvector's src/hnswinsert.c (patched: PgvNoteIndexChange in hnswinsert) is in the program too.  HnswInsertTupleOnDisk
 * searches the pages (src/hnswutils.c), adds the element and its neighbor tuple (PageAddItem, HnswInsertAppendPage),
 * rewrites the neighbors' tuples (PageIndexTupleOverwrite), moves the entry point in the meta page -- into the pages
 * pgv_host_hnsw_write_index laid out.  Afterwards the reference's scan and the device walk over the mirror the worker
 * restages (the product's stager over the reference's tuples) must return the same streams, every inserted row first
 * for its own vector, a duplicate's heap TID beside the original's. */
#define REL_HINS 2003
#ifdef PGV_HAVE_REF_HNSWVACUUM
extern IndexBulkDeleteResult *hnswbulkdelete(IndexVacuumInfo *info, IndexBulkDeleteResult *stats, IndexBulkDeleteCallback callback,
											  void *callback_state);
extern IndexBulkDeleteResult *hnswvacuumcleanup(IndexVacuumInfo *info, IndexBulkDeleteResult *stats);

static bool
hins_dead_callback(ItemPointer itemptr, void *state)
{
	(void) state;
	return row_of_tid(tid_key(itemptr)) % 4 == 0;
}
#endif
extern bool hnswinsert(Relation index, Datum *values, bool *isnull, ItemPointer heap_tid, Relation heap,
					   IndexUniqueCheck checkUnique, bool indexUnchanged, IndexInfo *indexInfo);

static int
backend_reference_hnsw_insert(void *arg)
{
	Relation	index = shim_open_relation(REL_HINS);
	const int	nins = 300,
				ndup = 5;
	float	   *data = malloc(sizeof(float) * (size_t) (HN + nins) * DIM);
	uint64		cpu[64],
				gpu[64];
	int			had_gpu;
	long		reads;
	BlockNumber blocks0 = RelationGetNumberOfBlocks(index);
	double		until;

	(void) arg;
	scenario = "the reference's own hnswinsert";
	hnsw_ef_search = 40;
	shim_seed_random(29);
	gen_rows(data, HN, DIM, 3);	/* the rows of the graph the pages hold */
	{
		float	   *more = malloc(sizeof(float) * (size_t) (HN + nins) * DIM);

		gen_rows(more, HN + nins, DIM, 41);
		memcpy(data + (size_t) HN * DIM, more + (size_t) HN * DIM, sizeof(float) * (size_t) nins * DIM);
		free(more);
	}
	for (int j = 0; j < ndup; j++)	/* the last few are copies of rows already in the graph */
		memcpy(data + (size_t) (HN + nins - 1 - j) * DIM, data + (size_t) (100 + 7 * j) * DIM, sizeof(float) * DIM);
	shim_set_guc_bool("vector.gpu", true);
	for (int j = 0; j < nins; j++)
	{
		MemoryContext ctx = shim_query_context_begin();
		Datum		value = PointerGetDatum(make_vector(data + (size_t) (HN + j) * DIM, DIM));
		bool		isnull = false;
		ItemPointerData tid = itemptr(tid_of_row(HN + j));

		hnswinsert(index, &value, &isnull, &tid, NULL, UNIQUE_CHECK_NO, false, NULL);
		shim_query_context_end(ctx);
	}
	EXPECT(RelationGetNumberOfBlocks(index) > blocks0);
	/* the mirror of the changed index */
	until = shim_now() + 30.0;
	for (;;)
	{
		(void) ref_hnsw_scan(index, data, 1, gpu, &had_gpu, &reads);
		if ((had_gpu && reads == 0) || shim_now() > until)
			break;
		usleep(20000);
	}
	EXPECT(had_gpu && reads == 0);
	for (int i = 0; i < 40; i++)
	{
		const int	r = i < 30 ? HN + (11 * i) % (nins - ndup) : (i < 35 ? HN + nins - 1 - (i - 30) : 37 * i);
		const float *q = data + (size_t) r * DIM;
		int			nc,
					ng;
		int64_t		rows[64];
		double		dist[64];

		shim_set_guc_bool("vector.gpu", false);
		nc = ref_hnsw_scan(index, q, 64, cpu, &had_gpu, &reads);
		EXPECT(!had_gpu && reads > 0 && nc >= 40);
		shim_set_guc_bool("vector.gpu", true);
		ng = ref_hnsw_scan(index, q, 64, gpu, &had_gpu, &reads);
		EXPECT(had_gpu && reads == 0);
		/* the device walk's stream against the reference's walk of the same pages: position by position the distance,
		 * the row wherever its distance is clear of its neighbours' */
		for (int j = 0; j < nc; j++)
		{
			const int	row = row_of_tid(cpu[j]);
			double		d = 0;

			EXPECT(row >= 0 && row < HN + nins);
			for (int k = 0; k < DIM; k++)
				d += ((double) data[(size_t) row * DIM + k] - q[k]) * ((double) data[(size_t) row * DIM + k] - q[k]);
			rows[j] = row;
			dist[j] = d;
		}
		if (check_hnsw_stream_n(data, HN + nins, q, gpu, ng, rows, dist, nc, "after the reference's hnsw inserts"))
			return 1;
		if (i < 30)
			EXPECT(cpu[0] == tid_of_row(r));	/* the inserted row, at distance 0 */
		else if (i < 35)
		{
			/* a duplicate: its heap TID sits on the element of the row it copies -- both come back, at distance 0 */
			const uint64 orig = tid_of_row(100 + 7 * (i - 30)),
						dup = tid_of_row(r);

			EXPECT((cpu[0] == orig && cpu[1] == dup) || (cpu[0] == dup && cpu[1] == orig));
			EXPECT((gpu[0] == orig && gpu[1] == dup) || (gpu[0] == dup && gpu[1] == orig));
		}
	}
	fprintf(stderr, "   %d rows (%d of them duplicates) through the reference's hnswinsert (%u -> %u blocks): its walk and the device's over the restaged pages agree\n",
			nins, ndup, (unsigned) blocks0, (unsigned) RelationGetNumberOfBlocks(index));
#ifdef PGV_HAVE_REF_HNSWVACUUM
	/* VACUUM by the reference's own hnswbulkdelete (src/hnswvacuum.c, patched: PgvNoteIndexChange at its end): every
	 * fourth row is dead.  RemoveHeapTids strips their heap TIDs, RepairGraph links the neighbors of the elements that
	 * go anew (the on-disk search and HnswUpdateConnection of src/hnswutils.c), MarkDeleted blanks their tuples.  The
	 * worker stages the index again; the two walks agree and no dead row comes back. */
	{
		IndexVacuumInfo info;
		IndexBulkDeleteResult *stats;
		int			dead = 0;

		for (int r = 0; r < HN + nins; r++)
			dead += r % 4 == 0;
		memset(&info, 0, sizeof(info));
		info.index = index;
		stats = hnswbulkdelete(&info, NULL, hins_dead_callback, NULL);
		EXPECT(stats != NULL && (int) stats->tuples_removed == dead);
		stats = hnswvacuumcleanup(&info, stats);
		EXPECT(stats != NULL && stats->num_pages == RelationGetNumberOfBlocks(index));
		until = shim_now() + 30.0;
		for (;;)
		{
			(void) ref_hnsw_scan(index, data + DIM, 1, gpu, &had_gpu, &reads);
			if ((had_gpu && reads == 0) || shim_now() > until)
				break;
			usleep(20000);
		}
		EXPECT(had_gpu && reads == 0);
		for (int i = 0; i < 30; i++)
		{
			const int	r = 4 * ((29 * i) % ((HN + nins) / 4)) + 1 + i % 3;	/* a row that lives */
			const float *q = data + (size_t) r * DIM;
			int			nc,
						ng;
			int64_t		rows[64];
			double		dist[64];

			shim_set_guc_bool("vector.gpu", false);
			nc = ref_hnsw_scan(index, q, 64, cpu, &had_gpu, &reads);
			EXPECT(!had_gpu && reads > 0 && nc >= 30);
			shim_set_guc_bool("vector.gpu", true);
			ng = ref_hnsw_scan(index, q, 64, gpu, &had_gpu, &reads);
			EXPECT(had_gpu && reads == 0);
			for (int j = 0; j < nc; j++)
			{
				const int	row = row_of_tid(cpu[j]);
				double		d = 0;

				EXPECT(row >= 0 && row < HN + nins && row % 4 != 0);
				for (int k = 0; k < DIM; k++)
					d += ((double) data[(size_t) row * DIM + k] - q[k]) * ((double) data[(size_t) row * DIM + k] - q[k]);
				rows[j] = row;
				dist[j] = d;
			}
			for (int j = 0; j < ng; j++)
				EXPECT(row_of_tid(gpu[j]) % 4 != 0);
			if (check_hnsw_stream_n(data, HN + nins, q, gpu, ng, rows, dist, nc, "after the reference's hnsw vacuum"))
				return 1;
		}
		fprintf(stderr, "   %d of %d rows removed by the reference's hnswbulkdelete: neither walk returns one of them\n", dead, HN + nins);
	}
#endif
	free(data);
	return 0;
}
#endif							/* PGV_HAVE_REF_HNSWINSERT */

#if defined(PGV_HAVE_REF_IVFBUILD) || defined(PGV_HAVE_REF_HNSWBUILD)
/* ------------------------------------------------------------------------------------------------ the table under CREATE INDEX
 * rows of the stand-in heap (pgshim_ref_runtime.c walks them for table_index_build_scan): some NULL, some toasted */
typedef struct HeapRows
{
	const float *rows;
	int			dim;
	int			toast_every,
				null_every;
}			HeapRows;

static int
heap_row_is_null(const HeapRows * h, int64_t row)
{
	return h->null_every > 0 && row % h->null_every == h->null_every - 1;
}

static void
heap_fetch(int64_t row, Datum *value, bool *isnull, ItemPointerData *tid, void *arg)
{
	const HeapRows *h = arg;

	*tid = itemptr(tid_of_row((int) row));
	*isnull = heap_row_is_null(h, row) != 0;
	*value = (Datum) 0;
	if (!*isnull)
	{
		Vector	   *plain = make_vector(h->rows + (size_t) row * h->dim, h->dim);

		*value = (h->toast_every && row % h->toast_every == 0) ? make_toasted(plain) : PointerGetDatum(plain);
	}
}

static Relation
heap_of(HeapRows * h, int64_t nrows)
{
	ShimHeapDef def;

	def.nrows = nrows;
	def.rows_per_block = 50;	/* tid_of_row */
	def.fetch = heap_fetch;
	def.arg = h;
	return shim_heap_relation(&def);
}
#endif

#ifdef PGV_HAVE_REF_IVFBUILD
/* ------------------------------------------------------------------------------------------------ the reference's own CREATE INDEX (ivfflat)
 * The program holds pgvector's src/ivfbuild.c as well (patched: the GPU branch in AddTupleToSort, IvfflatAddToSort split
 * off, PgvIvfflatBuildBegin / Flush round the heap scan, PgvNoteIndexChange at the end of ivfflatbuild).  ivfflatbuild()
 * itself runs: InitBuildState, ComputeCenters (SampleRows over the stand-in heap -> IvfflatKmeans), CreateMetaPage,
 * CreateListPages, CreateEntryPages (table_index_build_scan -> BuildCallback -> AddTupleToSort, tuplesort, InsertTuples)
 * -- every page of the index is written by the reference.
 *   vector.gpu = off, the oracle's pg_prng stream: the reference's whole serial build against the oracle's restatement of
 *     it (ora_kmeans from the same draws over the same samples, ora_ivf_assign): the centers in the list pages bit for bit,
 *     every list's tuples -- TID and vector, in heap order -- exactly the oracle's.
 *   vector.gpu = on: the same call; k-means, the argmin of every row and nothing else come from the device, the reference
 *     sorts and writes.  Every row is in a nearest list, once; the reference's scan, the oracle's page reader and the
 *     device scan over the mirror the worker stages (the product's stager over the reference's pages) agree. */
#define REL_RBUILD 1009
#define REL_RBUILD_COS 1011
#define REL_RBUILD_IP 1012
extern IndexBuildResult *ivfflatbuild(Relation heap, Relation index, IndexInfo *indexInfo);

static int
staged_image(Oid relid, pgv_ivf_image * img)
{
	pgv_rel		rel;
	uint32_t	nblocks;
	const void *pages = shim_relation_pages(relid, &nblocks);

	pgv_rel_init(&rel);
	rel.pages = (uint8_t *) pages;	/* (read only: the stager never writes) */
	rel.nblocks = nblocks;
	rel.cap = nblocks;
	return pgv_host_ivf_stage(&rel, PGV_F32, img);
}

static int
reference_ivfbuild_case(Oid relid, int ops, const char *opclass, uint64 seed)
{
	Relation	index = shim_open_relation(relid);
	const int	n = 6000,
				lists = 24;
	float	   *rows = malloc(sizeof(float) * (size_t) n * DIM);
	float	   *live = malloc(sizeof(float) * (size_t) n * DIM);	/* what the index stores, row by row */
	float	   *samples = malloc(sizeof(float) * (size_t) n * DIM);	/* what k-means is given */
	int		   *live_row = malloc(sizeof(int) * (size_t) n);
	int			nsamples = 0;
	uint64	   *got = malloc(sizeof(uint64) * 30000);
	HeapRows	h;
	Relation	heap;
	IndexInfo	info;
	IndexBuildResult *res;
	MemoryContext ctx;
	pgv_ivf_image img;
	int			nlive = 0,
				used_gpu;
	uint8_t		empty[1] = {0};

	scenario = "the reference's own ivfflatbuild";
	cur_ops = ops;
	EXPECT(index != NULL);
	gen_rows(rows, n, DIM, 51);
	if (ops != ORA_OPS_L2)
	{
		/* rows without a direction.  vector_cosine_ops (FUNCTION 2 and 4): AddSample and AddTupleToSort leave them out
		 * (IvfflatCheckNorm, src/ivfbuild.c:69-73, :174-180), every other row is stored normalised.  vector_ip_ops
		 * (FUNCTION 4 only): the sample leaves them out and is normalised for the spherical k-means, the index stores
		 * every row as it is */
		memset(rows + (size_t) 40 * DIM, 0, sizeof(float) * DIM);
		memset(rows + (size_t) 2041 * DIM, 0, sizeof(float) * DIM);
	}
	h.rows = rows;
	h.dim = DIM;
	h.toast_every = 7;
	h.null_every = 13;
	heap = heap_of(&h, n);
	for (int r = 0; r < n; r++)
		if (!heap_row_is_null(&h, r))
		{
			const bool	zero = ops != ORA_OPS_L2 && (r == 40 || r == 2041);

			if (!zero)
			{
				if (ops == ORA_OPS_L2)
					memcpy(samples + (size_t) nsamples * DIM, rows + (size_t) r * DIM, sizeof(float) * DIM);
				else
					ora_l2_normalize(DIM, rows + (size_t) r * DIM, samples + (size_t) nsamples * DIM);
				nsamples++;
			}
			if (zero && ops == ORA_OPS_COSINE)
				continue;
			if (ops == ORA_OPS_COSINE)
				ora_l2_normalize(DIM, rows + (size_t) r * DIM, live + (size_t) nlive * DIM);
			else
				memcpy(live + (size_t) nlive * DIM, rows + (size_t) r * DIM, sizeof(float) * DIM);
			live_row[nlive++] = r;
		}
	memset(&info, 0, sizeof(info));

	/* ---- (1) the reference's serial CPU build = the oracle's */
	{
		ora_prng	a,
					b;
		float	   *want_centers = malloc(sizeof(float) * (size_t) lists * DIM);
		int32_t    *want_list = malloc(sizeof(int32_t) * (size_t) nlive);
		int			iterations,
					at = 0;

		shim_set_guc_bool("vector.gpu", false);
		ora_prng_seed(&a, seed);
		shim_prng_hook(ora_prng_double_cb, ora_prng_u32_cb, &a);
		ctx = shim_query_context_begin();
		res = ivfflatbuild(heap, index, &info);
		shim_prng_hook(NULL, NULL, NULL);
		EXPECT(res != NULL && (int) res->heap_tuples == n && (int) res->index_tuples == nlive);
		shim_query_context_end(ctx);
		EXPECT(shim_pinned_buffers() == 0);

		/* the oracle: 6000 rows are fewer than the 10 000 samples asked for, so the sample IS the table's non-NULL rows
		 * in heap order (src/ivfbuild.c:446-455; every block is taken, the reservoir never replaces); SampleRows draws
		 * its block sampler's seed and the reservoir's from the global stream before k-means starts */
		ora_prng_seed(&b, seed);
		(void) ora_prng_u32(&b);
		(void) ora_prng_u32(&b);
		iterations = ora_kmeans(ops, ORA_F32, DIM, samples, nsamples, want_centers, lists, &b, NULL);
		EXPECT(iterations >= 0);
		EXPECT(a.s0 == b.s0 && a.s1 == b.s1);	/* the same number of draws */
		ora_ivf_assign(ops, ORA_F32, DIM, want_centers, lists, live, nlive, want_list, NULL);

		EXPECT(staged_image(relid, &img) == PGV_OK);
		EXPECT(img.dim == DIM && img.lists == lists && img.nrows == nlive);
		if (memcmp(img.centers, want_centers, sizeof(float) * (size_t) lists * DIM) != 0)
		{
#if defined(__SANITIZE_ADDRESS__)
			/* the sanitizer build instruments the reference's loops too: they are vectorised (and, under -fassociative-math,
			 * summed) differently from the oracle library's, a low bit differs somewhere, and a k-means whose assignments
			 * hang on such a bit takes another path.  The plain build is the one that holds the two to the bit; this one
			 * is here for the memory errors, and goes on without the comparison */
			fprintf(stderr, "   %s (sanitizer build): the instrumented k-means took another path than the oracle's; comparison skipped\n", opclass);
			iterations = -1;
			goto lists_done;
#else
			fprintf(stderr, "the centers in the reference's list pages are not the oracle's\n");
			return 1;
#endif
		}
		for (int l = 0; l < lists; l++)
		{
			int64_t		p = img.list_offsets[l];

			/* tuplesort by list, heap order inside (the stand-in's sort is stable): the oracle's rows of list l in order */
			for (int i = 0; i < nlive; i++)
				if (want_list[i] == l)
				{
					if (p >= img.list_offsets[l + 1] || img.tids[p] != tid_of_row(live_row[i]) ||
						memcmp((const float *) img.vectors + (size_t) p * DIM, live + (size_t) i * DIM, sizeof(float) * DIM) != 0)
					{
						fprintf(stderr, "list %d of the reference's build: position %lld is not the oracle's row %d\n", l,
								(long long) (p - img.list_offsets[l]), live_row[i]);
						return 1;
					}
					p++;
					at++;
				}
			EXPECT(p == img.list_offsets[l + 1]);
		}
		EXPECT(at == nlive);
#if defined(__SANITIZE_ADDRESS__)
lists_done:
#endif
		if (iterations >= 0)
			fprintf(stderr, "   %s: the reference's serial ivfflatbuild (%d rows, %d NULL or zero, %d lists, %d Elkan iterations, %u blocks) = the oracle's build: centers bit for bit, every list's tuples in order\n",
					opclass, n, n - nlive, lists, iterations, (unsigned) RelationGetNumberOfBlocks(index));
		pgv_host_ivf_image_free(&img);
		/* and its own scan reads what it wrote */
		for (int i = 0; i < 8; i++)
		{
			const int	r = live_row[(97 * i) % nlive];
			Expected	e = expected_batch(relid, rows + (size_t) r * DIM, PROBES);
			int			nn = ref_scan(index, rows + (size_t) r * DIM, PROBES, 10, got, &used_gpu);

			EXPECT(!used_gpu && nn == 10 && (ops == ORA_OPS_IP || got[0] == tid_of_row(r)));	/* (the largest inner product need not be the row's own) */
			if (check_stream(&e, got, nn, 0, "the reference's CPU build, CPU branch"))
				return 1;
			expected_free(&e);
		}
		free(want_centers);
		free(want_list);
	}

	/* ---- (1b) the same CREATE INDEX with vector.gpu = on and vector.gpu_kmeans = off: the reference's own ElkanKmeans
	 * computes the centers (same sample, same draws: the CPU build's centers to the bit), the device assigns the rows.
	 * Product against reference with nothing in between: the index must be the CPU build's -- page for page, but for a
	 * row that is equally near two centers to the last bit of a float (the stand-in device sums in another order than
	 * the reference's vectorised kernel; such a row may go either way) */
	{
		uint32_t	nb_cpu,
					nb_gpu;
		const uint8_t *pg = shim_relation_pages(relid, &nb_cpu);
		uint8_t    *cpu_pages = malloc((size_t) nb_cpu * 8192);
		pgv_ivf_image ic,
					ig;
		ora_prng	a;
		int			moved = 0;

		memcpy(cpu_pages, pg, (size_t) nb_cpu * 8192);
		EXPECT(staged_image(relid, &ic) == PGV_OK);
		shim_replace_pages(relid, empty, 0);
		shim_set_guc_bool("vector.gpu", true);
		shim_set_guc_bool("vector.gpu_kmeans", false);
		ora_prng_seed(&a, seed);
		shim_prng_hook(ora_prng_double_cb, ora_prng_u32_cb, &a);
		ctx = shim_query_context_begin();
		res = ivfflatbuild(heap, index, &info);
		shim_prng_hook(NULL, NULL, NULL);
		shim_set_guc_bool("vector.gpu_kmeans", true);
		EXPECT(res != NULL && (int) res->heap_tuples == n && (int) res->index_tuples == nlive);
		shim_query_context_end(ctx);
		EXPECT(staged_image(relid, &ig) == PGV_OK);
		EXPECT(ig.lists == ic.lists && ig.nrows == ic.nrows);
		EXPECT(memcmp(ig.centers, ic.centers, sizeof(float) * (size_t) lists * DIM) == 0);	/* Elkan's centers, to the bit */
		{
			/* which list each TID is in, CPU build against device-assigned build */
			int		   *list_cpu = malloc(sizeof(int) * (size_t) n);

			for (int r = 0; r < n; r++)
				list_cpu[r] = -1;
			for (int l = 0; l < lists; l++)
				for (int64_t p2 = ic.list_offsets[l]; p2 < ic.list_offsets[l + 1]; p2++)
					list_cpu[row_of_tid(ic.tids[p2])] = l;
			for (int l = 0; l < lists; l++)
				for (int64_t p2 = ig.list_offsets[l]; p2 < ig.list_offsets[l + 1]; p2++)
					moved += list_cpu[row_of_tid(ig.tids[p2])] != l;
			free(list_cpu);
		}
		pg = shim_relation_pages(relid, &nb_gpu);
		if (getenv("PGV_TEST_TRACE") && nb_gpu == nb_cpu)
			for (size_t o = 0; o < (size_t) nb_cpu * 8192; o++)
				if (pg[o] != cpu_pages[o])
				{
					fprintf(stderr, "      first differing byte: block %zu offset %zu: %02x vs %02x\n", o / 8192, o % 8192, pg[o], cpu_pages[o]);
					break;
				}
		EXPECT(moved * 1000 <= nlive);	/* at most one row in a thousand on the fence */
		fprintf(stderr, "   %s: vector.gpu_kmeans = off -- Elkan's centers to the bit, the device's argmins: %d of %d rows in another list than the CPU build's; the index %s\n",
				opclass, moved, nlive, nb_gpu == nb_cpu && memcmp(pg, cpu_pages, (size_t) nb_cpu * 8192) == 0 ? "is the CPU build's, byte for byte" : "differs in bytes");
		pgv_host_ivf_image_free(&ic);
		pgv_host_ivf_image_free(&ig);
		free(cpu_pages);
	}

	/* ---- (2) DROP + CREATE INDEX with vector.gpu = on: the hooks inside the reference's build */
	shim_replace_pages(relid, empty, 0);
	shim_set_guc_bool("vector.gpu", true);
	shim_set_guc_bool("vector.gpu_pooled", false);
	shim_seed_random(53);
	ctx = shim_query_context_begin();
	res = ivfflatbuild(heap, index, &info);
	EXPECT(res != NULL && (int) res->heap_tuples == n && (int) res->index_tuples == nlive);
	shim_query_context_end(ctx);
	EXPECT(shim_pinned_buffers() == 0);
	EXPECT(staged_image(relid, &img) == PGV_OK);
	EXPECT(img.dim == DIM && img.lists == lists && img.nrows == nlive);
	{
		char	   *seen = calloc((size_t) n, 1);

		for (int l = 0; l < lists; l++)
		{
			int			prev = -1;

			for (int64_t p = img.list_offsets[l]; p < img.list_offsets[l + 1]; p++)
			{
				const int	r = row_of_tid(img.tids[p]);
				const float *x = (const float *) img.vectors + (size_t) p * DIM;
				double		best = INFINITY,
							mine;

				EXPECT(r > prev && r < n && !seen[r] && !heap_row_is_null(&h, r));	/* heap order inside a list, no row twice */
				prev = r;
				seen[r] = 1;
				if (ops == ORA_OPS_COSINE)
				{
					float		unit[DIM];

					ora_l2_normalize(DIM, rows + (size_t) r * DIM, unit);
					for (int d = 0; d < DIM; d++)
						EXPECT(fabsf(x[d] - unit[d]) <= 1e-6f);	/* (the hook's rows were normalised by the reference before it saw them) */
				}
				else
					EXPECT(memcmp(x, rows + (size_t) r * DIM, sizeof(float) * DIM) == 0);
				for (int c = 0; c < lists; c++)
				{
					double		d = ora_index_distance(ops, ORA_F32, DIM, x, (const float *) img.centers + (size_t) c * DIM);

					if (d < best)
						best = d;
				}
				mine = ora_index_distance(ops, ORA_F32, DIM, x, (const float *) img.centers + (size_t) l * DIM);
				EXPECT(mine <= best + 1e-5 * fabs(best) + 1e-9);	/* its nearest center (float-level ties excepted) */
			}
		}
		free(seen);
	}
	pgv_host_ivf_image_free(&img);
	/* the three readers over pages the reference wrote from the device's answers */
	shim_set_guc_bool("vector.gpu", false);
	for (int i = 0; i < 12; i++)
	{
		const int	r = live_row[(131 * i + 5) % nlive];
		Expected	e = expected_batch(relid, rows + (size_t) r * DIM, PROBES);
		int			nn = ref_scan(index, rows + (size_t) r * DIM, PROBES, i % 4 == 3 ? 300 : 10, got, &used_gpu);

		EXPECT(!used_gpu && nn == (i % 4 == 3 ? (300 < e.n ? 300 : e.n) : 10) && (ops == ORA_OPS_IP || got[0] == tid_of_row(r)));
		if (check_stream(&e, got, nn, 0, "the reference's build with the hooks, CPU branch"))
			return 1;
		expected_free(&e);
	}
	shim_set_guc_bool("vector.gpu", true);
	EXPECT(wait_for_gpu(index, 30.0) == 0);
	for (int i = 0; i < 12; i++)
	{
		const int	r = live_row[(131 * i + 71) % nlive];
		Expected	e = expected_batch(relid, rows + (size_t) r * DIM, PROBES);
		int			nn = ref_scan(index, rows + (size_t) r * DIM, PROBES, i % 4 == 3 ? 300 : 10, got, &used_gpu);

		EXPECT(used_gpu && nn == (i % 4 == 3 ? (300 < e.n ? 300 : e.n) : 10) && (ops == ORA_OPS_IP || got[0] == tid_of_row(r)));
		if (check_stream(&e, got, nn, 0, "the reference's build with the hooks, hooks"))
			return 1;
		expected_free(&e);
	}
	fprintf(stderr, "   %s: the reference's ivfflatbuild with the hooks (k-means and %d argmins from the device, pages by the reference): every row in a nearest list; its scan, the oracle and the staged mirror agree\n",
			opclass, nlive);
	free(rows);
	free(live);
	free(samples);
	free(live_row);
	free(got);
	return 0;
}

static int
backend_reference_ivfbuild(void *arg)
{
	(void) arg;
	if (reference_ivfbuild_case(REL_RBUILD, ORA_OPS_L2, "vector_l2_ops", 91))
		return 1;
	/* vector_cosine_ops: FUNCTION 2 (vector_norm) makes the reference normalise what it stores and what it is asked for,
	 * FUNCTION 4 makes the k-means spherical; the hooks get PGV_OPS_COSINE / PGV_NEG_IP */
	if (reference_ivfbuild_case(REL_RBUILD_COS, ORA_OPS_COSINE, "vector_cosine_ops", 93))
		return 1;
	/* vector_ip_ops (BASELINE configs[2]'s opclass): FUNCTION 4 only -- spherical k-means over a normalised sample, the
	 * rows stored as they are, FUNCTION 1 the negative inner product; the hooks get PGV_OPS_IP */
	return reference_ivfbuild_case(REL_RBUILD_IP, ORA_OPS_IP, "vector_ip_ops", 95);
}
#endif							/* PGV_HAVE_REF_IVFBUILD */

#ifdef PGV_HAVE_REF_HNSWBUILD
/* ------------------------------------------------------------------------------------------------ the reference's own CREATE INDEX (hnsw)
 * pgvector's src/hnswbuild.c is in the program too (patched: PgvHnswBuildBegin at the end of InitBuildState,
 * PgvHnswBuildDefer in InsertTuple, PgvHnswBuildLink first in FlushPages, PgvNoteIndexChange at the end of hnswbuild).
 * hnswbuild() itself runs over the stand-in heap: InitBuildState, BuildGraph (table_index_build_scan -> BuildCallback ->
 * InsertTuple: HnswFormIndexValue, HnswInitElement, the allocator and its memory accounting), FlushPages (CreateMetaPage,
 * CreateGraphPages, WriteNeighborTuples) -- every page is the reference's.
 *   vector.gpu = off, the oracle's pg_prng stream: the reference's serial in-memory build (InsertTupleInMemory,
 *     HnswFindElementNeighbors, UpdateNeighborsInMemory, the duplicates) against the oracle's ora_hnsw_build from the same
 *     seed: levels, every neighbor array slot for slot, the heap TIDs of duplicates, the entry point -- read back from
 *     the reference's pages by the product's stager.
 *   vector.gpu = on: the hooks defer every element and link them in FlushPages; the reference serialises what the device
 *     linked.  The reference's walk of those pages and the device walk over the staged mirror return the same streams. */
#define REL_HRBUILD 2004
#define REL_HRBUILD_COS 2006
extern IndexBuildResult *hnswbuild(Relation heap, Relation index, IndexInfo *indexInfo);

static int
staged_hnsw_image(Oid relid, pgv_hnsw_image * img)
{
	pgv_rel		rel;
	uint32_t	nblocks;
	const void *pages = shim_relation_pages(relid, &nblocks);

	pgv_rel_init(&rel);
	rel.pages = (uint8_t *) pages;
	rel.nblocks = nblocks;
	rel.cap = nblocks;
	return pgv_host_hnsw_stage(&rel, PGV_F32, img);
}

static int
reference_hnswbuild_case(Oid relid, int ops, const char *opclass, uint64 seed)
{
	Relation	index = shim_open_relation(relid);
	const int	n = 2500,
				m = HM,
				efc = 32;
	float	   *rows = malloc(sizeof(float) * (size_t) n * DIM);
	float	   *live = malloc(sizeof(float) * (size_t) n * DIM);
	float	   *stored = malloc(sizeof(float) * (size_t) n * DIM);	/* what the index holds of each row: the row, or the row normalised */
	int		   *live_row = malloc(sizeof(int) * (size_t) n);
	int			nzero = 0;
	HeapRows	h;
	Relation	heap;
	IndexInfo	info;
	IndexBuildResult *res;
	MemoryContext ctx;
	pgv_hnsw_image img;
	int			nlive = 0;
	uint8_t		empty[1] = {0};
	uint64		cpu[64],
				gpu[64];
	int			had_gpu;
	long		reads;

	scenario = "the reference's own hnswbuild";
	EXPECT(index != NULL);
	hnsw_ef_search = 40;
	gen_rows(rows, n, DIM, 61);
	/* duplicates: one value under ten heap TIDs (what an element holds), another under two.  (An eleventh copy would become
	 * a second element with the same vector: every distance to the two is an exact tie, and which of them a neighbor list
	 * keeps is decided by the order equal keys leave the server's pairing heap in -- PostgreSQL core, not pinned by the
	 * oracle (order among equal distances).  The hook phase above covers that case tie-tolerantly.) */
	for (int r = 700; r < 709; r++)
		memcpy(rows + (size_t) r * DIM, rows + (size_t) 300 * DIM, sizeof(float) * DIM);
	memcpy(rows + (size_t) 2000 * DIM, rows + (size_t) 10 * DIM, sizeof(float) * DIM);
	if (ops == ORA_OPS_COSINE)
	{
		/* vector_cosine_ops: HnswFormIndexValue normalises every value and leaves rows without a direction out
		 * (src/hnswutils.c:406-428) */
		memset(rows + (size_t) 55 * DIM, 0, sizeof(float) * DIM);
		memset(rows + (size_t) 1500 * DIM, 0, sizeof(float) * DIM);
		nzero = 2;
	}
	for (int r = 0; r < n; r++)
		if (ops == ORA_OPS_COSINE)
			ora_l2_normalize(DIM, rows + (size_t) r * DIM, stored + (size_t) r * DIM);
		else
			memcpy(stored + (size_t) r * DIM, rows + (size_t) r * DIM, sizeof(float) * DIM);
	h.rows = rows;
	h.dim = DIM;
	h.toast_every = 9;
	h.null_every = 17;
	heap = heap_of(&h, n);
	for (int r = 0; r < n; r++)
		if (!heap_row_is_null(&h, r))
		{
			memcpy(live + (size_t) nlive * DIM, rows + (size_t) r * DIM, sizeof(float) * DIM);
			live_row[nlive++] = r;
		}
	memset(&info, 0, sizeof(info));

	/* ---- (1) the reference's serial in-memory build = the oracle's */
	{
		ora_prng	a;
		ora_hnsw   *g;
		int64_t		ne;
		int		   *slot_of_element;
		int			entry_level;
		int			lists_total = 0,
					lists_reordered = 0,
					lists_different = 0,
					first_bad_layer = -1;
		int64_t		first_bad_element = -1;

		shim_set_guc_bool("vector.gpu", false);
		ora_prng_seed(&a, seed);
		shim_prng_hook(ora_prng_double_cb, ora_prng_u32_cb, &a);
		ctx = shim_query_context_begin();
		res = hnswbuild(heap, index, &info);
		shim_prng_hook(NULL, NULL, NULL);
		EXPECT(res != NULL && (int) res->heap_tuples == n && (int) res->index_tuples == nlive - nzero);
		shim_query_context_end(ctx);
		EXPECT(shim_pinned_buffers() == 0);

		g = ora_hnsw_build(ops, ORA_F32, DIM, live, nlive, m, efc, seed);	/* one level draw per non-NULL row, from seed 97 */
		EXPECT(g != NULL);
		ne = ora_hnsw_num_elements(g);
		EXPECT(staged_hnsw_image(relid, &img) == PGV_OK);
		EXPECT(img.dim == DIM && img.m == m && img.ef_construction == efc);
		if (img.n != ne)
		{
			fprintf(stderr, "the reference's pages hold %lld elements, the oracle's graph %lld\n", (long long) img.n, (long long) ne);
			return 1;
		}
		/* slots are in page order; an element is known by its first heap TID */
		slot_of_element = malloc(sizeof(int) * (size_t) ne);
		{
			int		   *slot_of_row = malloc(sizeof(int) * (size_t) n);

			for (int r = 0; r < n; r++)
				slot_of_row[r] = -1;
			for (int64_t s = 0; s < img.n; s++)
			{
				int			r = row_of_tid(img.heaptids[(size_t) s * 10]);

				EXPECT(r >= 0 && r < n && slot_of_row[r] == -1);
				slot_of_row[r] = (int) s;
			}
			for (int64_t e = 0; e < ne; e++)
			{
				const int	r = live_row[ora_hnsw_element_row(g, e)];

				if (slot_of_row[r] < 0)
				{
					fprintf(stderr, "the oracle's element %lld (heap row %d) is no element of the reference's pages\n", (long long) e, r);
					return 1;
				}
				slot_of_element[e] = slot_of_row[r];
			}
			free(slot_of_row);
		}
		for (int64_t e = 0; e < ne; e++)
		{
			const int	s = slot_of_element[e];
			const int	level = ora_hnsw_level(g, e);

			if (img.levels[s] != level)
			{
				fprintf(stderr, "element %lld: level %d in the reference's pages, %d in the oracle's graph\n", (long long) e, img.levels[s], level);
				return 1;
			}
			EXPECT(memcmp((const float *) img.vectors + (size_t) s * DIM, stored + (size_t) live_row[ora_hnsw_element_row(g, e)] * DIM, sizeof(float) * DIM) == 0);
			for (int lc = level; lc >= 0; lc--)
			{
				int32_t		want[2 * HM];
				const int	lm = lc == 0 ? 2 * m : m;
				const int	nw = ora_hnsw_neighbors(g, e, lc, want);
				const int32_t *have = img.nbr + img.nbr_start[s] + (int64_t) (level - lc) * m;

				{
					/* slot for slot; where that fails, as sets (the same neighbors in another order) */
					int			same_order = 1,
								same_set = 1;

					for (int i = 0; i < lm; i++)
						same_order &= have[i] == (i < nw ? slot_of_element[want[i]] : -1);
					for (int i = 0; i < lm && !same_order; i++)
					{
						int			found = have[i] < 0;

						for (int j = 0; j < nw && !found; j++)
							found = have[i] == slot_of_element[want[j]];
						same_set &= found;
					}
					lists_total++;
					lists_reordered += !same_order && same_set;
					lists_different += !same_order && !same_set;
					if (!same_order && first_bad_element < 0)
					{
						first_bad_element = e;
						first_bad_layer = lc;
					}
				}
			}
		}
		/* vector_l2_ops: every slot of every list.  vector_cosine_ops: the distances of a cluster's members to one another
		 * sit in a band a few thousand floats wide, so EXACT TIES between candidates are common, and what the reference
		 * does with equal keys is PostgreSQL's pairing heap's business (which of two equally far candidates leaves W first,
		 * src/hnswutils.c:866-975 over lib/pairingheap.c) -- unpinned.  There every list must hold the oracle's
		 * NEIGHBORS (the same graph), and all but a handful of lists in the oracle's slot order too. */
#if defined(__SANITIZE_ADDRESS__)
		/* (the sanitizer build's instrumented kernels sum in another order: a distance may differ in its last bit, and with
		 * it a decision between two nearly equally far candidates -- the plain build is the one held to the bit) */
		if ((lists_reordered + lists_different) * 50 > lists_total)
#else
		if (lists_different != 0 || (ops == ORA_OPS_L2 ? lists_reordered != 0 : lists_reordered * 50 > lists_total))
#endif
		{
			fprintf(stderr, "%s: %d of %d neighbor lists differ from the oracle's (%d as sets); first: element %lld layer %d\n", opclass,
					lists_reordered + lists_different, lists_total, lists_different, (long long) first_bad_element, first_bad_layer);
			return 1;
		}
		EXPECT(img.entry == slot_of_element[ora_hnsw_entry_point(g, &entry_level)]);
		/* the duplicates' heap TIDs sit on the elements that took them */
		{
			int			on300 = 0,
						rest = 0;

			for (int64_t s = 0; s < img.n; s++)
			{
				const int	first = row_of_tid(img.heaptids[(size_t) s * 10]);

				for (int t = 1; t < 10 && img.heaptids[(size_t) s * 10 + t] != UINT64_MAX; t++)
				{
					const int	dr = row_of_tid(img.heaptids[(size_t) s * 10 + t]);

					EXPECT(memcmp(rows + (size_t) dr * DIM, rows + (size_t) first * DIM, sizeof(float) * DIM) == 0);
					if (first == 300)
						on300++;
					else
						rest++;
				}
			}
			EXPECT(on300 == 9 && rest == 1);	/* rows 700-708 on row 300's element (ten heap TIDs: full), row 2000 on row 10's */
		}
		fprintf(stderr, "   %s: the reference's serial hnswbuild (%d rows, %d NULL or zero, m %d, ef_construction %d) = the oracle's graph: %lld elements, levels, every neighbor %s, the entry point (level %d), %u blocks; %d of %d neighbor lists hold the oracle's neighbors in another slot order, %d differ as sets\n",
				opclass, n, n - nlive + nzero, m, efc, (long long) ne, ops == ORA_OPS_L2 ? "slot" : "list", entry_level,
				(unsigned) RelationGetNumberOfBlocks(index), lists_reordered, lists_total, lists_different);
		/* and the reference's walk of the pages it wrote (hnswgettuple -> GetScanItems -> HnswSearchLayer over
		 * HnswLoadElement, vector.gpu still off) against the oracle's walk of ITS graph: the same stream, position by
		 * position the same distance */
		{
			int			walks = 0;

			for (int i = 0; i < 24; i++)
			{
				const int	r = live_row[(89 * i + 17) % nlive];
				int64_t		orows[64],
							scored;
				double		odist[64],
							wdist[64];
				int			want,
							nc;

				if (ops != ORA_OPS_L2 && (r == 55 || r == 1500))
					continue;
				want = ora_hnsw_search(g, rows + (size_t) r * DIM, 40, 40, orows, odist, &scored);
				nc = ref_hnsw_scan(index, rows + (size_t) r * DIM, 40, cpu, &had_gpu, &reads);
				EXPECT(!had_gpu && reads > 0);
				for (int j = 0; j < want; j++)
				{
					double		d = 0;

					orows[j] = live_row[orows[j]];
					for (int k = 0; k < DIM; k++)
						d += ((double) stored[(size_t) orows[j] * DIM + k] - stored[(size_t) r * DIM + k]) *
							((double) stored[(size_t) orows[j] * DIM + k] - stored[(size_t) r * DIM + k]);
					wdist[j] = d;
				}
				if (check_hnsw_stream_n(stored, n, stored + (size_t) r * DIM, cpu, nc, orows, wdist, want, "the reference's walk of its own build"))
					return 1;
				walks++;
			}
			fprintf(stderr, "   %s: %d walks of the reference's HnswSearchLayer over its own pages = the oracle's walks of its graph\n", opclass, walks);
		}
		free(slot_of_element);
		pgv_host_hnsw_image_free(&img);
		ora_hnsw_free(g);
	}

	/* ---- (1b) the same CREATE INDEX with the hooks and vector.gpu_hnsw_build_batch = 1: every element deferred and linked
	 * on the device one at a time -- the serial build's insertion order.  Same heap, same level draws (the reference's own
	 * HnswInitElement on the same stream): the product must hand FlushPages the reference's serial graph, and the
	 * reference then writes the same index -- compared page for page with what its CPU build wrote above */
	{
		uint32_t	nb_cpu,
					nb_gpu;
		const uint8_t *pg = shim_relation_pages(relid, &nb_cpu);
		uint8_t    *cpu_pages = malloc((size_t) nb_cpu * 8192);
		ora_prng	a;
		pgv_hnsw_image ic,
					ig;
		int			same_bytes;

		memcpy(cpu_pages, pg, (size_t) nb_cpu * 8192);
		EXPECT(staged_hnsw_image(relid, &ic) == PGV_OK);
		shim_replace_pages(relid, empty, 0);
		shim_set_guc_bool("vector.gpu", true);
		shim_set_guc_int("vector.gpu_hnsw_build_batch", 1);
		ora_prng_seed(&a, seed);
		shim_prng_hook(ora_prng_double_cb, ora_prng_u32_cb, &a);
		ctx = shim_query_context_begin();
		res = hnswbuild(heap, index, &info);
		shim_prng_hook(NULL, NULL, NULL);
		shim_set_guc_int("vector.gpu_hnsw_build_batch", 1024);
		EXPECT(res != NULL && (int) res->heap_tuples == n && (int) res->index_tuples == nlive - nzero);
		shim_query_context_end(ctx);
		pg = shim_relation_pages(relid, &nb_gpu);
		EXPECT(nb_gpu == nb_cpu);
		same_bytes = memcmp(pg, cpu_pages, (size_t) nb_cpu * 8192) == 0;
		EXPECT(staged_hnsw_image(relid, &ig) == PGV_OK);
		EXPECT(ig.n == ic.n && ig.entry == ic.entry && memcmp(ig.levels, ic.levels, sizeof(int32_t) * (size_t) ic.n) == 0);
		EXPECT(memcmp(ig.heaptids, ic.heaptids, sizeof(uint64_t) * 10 * (size_t) ic.n) == 0);
		{
			int64_t		slots = ic.nbr_start[ic.n],
						differing = 0;

			for (int64_t j = 0; j < slots; j++)
				differing += ig.nbr[j] != ic.nbr[j];
			/* vector_l2_ops: every slot.  vector_cosine_ops: exact ties inside a cluster are common and the device walk
			 * and the reference's pairing heap order equal keys each in their way (a few slots) */
			if (ops == ORA_OPS_L2 ? differing != 0 : differing * 200 > slots)
			{
				fprintf(stderr, "%s: the hooks at batch 1 leave %lld of %lld neighbor slots different from the reference's serial build\n",
						opclass, (long long) differing, (long long) slots);
				return 1;
			}
			fprintf(stderr, "   %s: the hooks at vector.gpu_hnsw_build_batch = 1 hand FlushPages the reference's serial graph: %lld of %lld neighbor slots differ; the index the reference writes from it %s\n",
					opclass, (long long) differing, (long long) slots, same_bytes ? "is the CPU build's, byte for byte" : "differs in bytes");
			if (ops == ORA_OPS_L2)
				EXPECT(same_bytes);
		}
		pgv_host_hnsw_image_free(&ic);
		pgv_host_hnsw_image_free(&ig);
		free(cpu_pages);
	}

	/* ---- (2) DROP + CREATE INDEX with vector.gpu = on */
	shim_replace_pages(relid, empty, 0);
	shim_set_guc_bool("vector.gpu", true);
	shim_seed_random(59);
	for (int i = 0; i < 16; i++)
		(void) RandomDouble();
	ctx = shim_query_context_begin();
	res = hnswbuild(heap, index, &info);
	EXPECT(res != NULL && (int) res->heap_tuples == n && (int) res->index_tuples == nlive - nzero);
	shim_query_context_end(ctx);
	EXPECT(shim_pinned_buffers() == 0);
	EXPECT(staged_hnsw_image(relid, &img) == PGV_OK);
	EXPECT(img.n > nlive - 20 - nzero && img.n < nlive && img.entry >= 0);
	{
		int64_t		links = 0;

		for (int64_t s = 0; s < img.n; s++)
			for (int64_t j = img.nbr_start[s]; j < img.nbr_start[s + 1]; j++)
			{
				EXPECT(img.nbr[j] >= -1 && img.nbr[j] < img.n && img.nbr[j] != s);
				links += img.nbr[j] >= 0;
			}
		EXPECT(links > img.n * m);	/* a linked graph, not a list of lonely elements */
	}
	pgv_host_hnsw_image_free(&img);
	{
		double		until = shim_now() + 30.0;

		for (;;)
		{
			(void) ref_hnsw_scan(index, rows, 1, gpu, &had_gpu, &reads);
			if ((had_gpu && reads == 0) || shim_now() > until)
				break;
			usleep(20000);
		}
		EXPECT(had_gpu && reads == 0);
	}
	for (int i = 0; i < 30; i++)
	{
		const int	r = live_row[(83 * i + 9) % nlive] == 55 || live_row[(83 * i + 9) % nlive] == 1500 ? 56 : live_row[(83 * i + 9) % nlive];
		const float *q = rows + (size_t) r * DIM,
				   *sq = stored + (size_t) r * DIM;
		int			nc,
					ng;
		int64_t		wrows[64];
		double		wdist[64];

		shim_set_guc_bool("vector.gpu", false);
		nc = ref_hnsw_scan(index, q, 64, cpu, &had_gpu, &reads);
		EXPECT(!had_gpu && reads > 0 && nc >= 40);
		shim_set_guc_bool("vector.gpu", true);
		ng = ref_hnsw_scan(index, q, 64, gpu, &had_gpu, &reads);
		EXPECT(had_gpu && reads == 0);
		for (int j = 0; j < nc; j++)
		{
			const int	row = row_of_tid(cpu[j]);
			double		d = 0;

			EXPECT(row >= 0 && row < n && !heap_row_is_null(&h, row));
			for (int k = 0; k < DIM; k++)
				d += ((double) stored[(size_t) row * DIM + k] - sq[k]) * ((double) stored[(size_t) row * DIM + k] - sq[k]);
			wrows[j] = row;
			wdist[j] = d;
		}
		EXPECT(wdist[0] == 0.0);	/* the row itself (or a copy of it) comes first */
		if (check_hnsw_stream_n(stored, n, sq, gpu, ng, wrows, wdist, nc, "the reference's hnswbuild with the hooks"))
			return 1;
	}
	fprintf(stderr, "   %s: the reference's hnswbuild with the hooks (every element deferred, linked on the device in FlushPages, pages by the reference): its walk and the device walk agree\n", opclass);

	/* ---- (3) the graph outgrows maintenance_work_mem halfway through the heap scan: InsertTuple raises the NOTICE and calls
	 * FlushPages there and then -- the hook links what was deferred so far, the reference writes those pages --, and every
	 * later row goes through the reference's HnswInsertTupleOnDisk into them (src/hnswbuild.c:520-541) */
	{
		extern int	maintenance_work_mem;
		const int	saved = maintenance_work_mem;
		const int	notices0 = shim_notices_raised("hnsw graph no longer fits into maintenance_work_mem");
		int			found = 0;

		shim_replace_pages(relid, empty, 0);
		maintenance_work_mem = 2300;	/* kB: the graph gets half of it (the hook reserves the rest for the link phase), the deferred-elements array takes 512 kB: room for about half of the elements */
		shim_set_guc_bool("vector.gpu", true);
		ctx = shim_query_context_begin();
		res = hnswbuild(heap, index, &info);
		maintenance_work_mem = saved;
		EXPECT(res != NULL && (int) res->heap_tuples == n && (int) res->index_tuples == nlive - nzero);
		shim_query_context_end(ctx);
		EXPECT(shim_pinned_buffers() == 0);
		EXPECT(staged_hnsw_image(relid, &img) == PGV_OK);
		EXPECT(img.n > nlive - 20 - nzero && img.n < nlive && img.entry >= 0);
		pgv_host_hnsw_image_free(&img);
		{
			double		until = shim_now() + 30.0;

			for (;;)
			{
				(void) ref_hnsw_scan(index, rows, 1, gpu, &had_gpu, &reads);
				if ((had_gpu && reads == 0) || shim_now() > until)
					break;
				usleep(20000);
			}
			EXPECT(had_gpu && reads == 0);
		}
		for (int i = 0; i < 40; i++)
		{
			const int	r = live_row[(59 * i + 3) % nlive] == 55 || live_row[(59 * i + 3) % nlive] == 1500 ? 56 : live_row[(59 * i + 3) % nlive];	/* rows from before and after the flush */
			const float *q = rows + (size_t) r * DIM,
				   *sq = stored + (size_t) r * DIM;
			int			nc,
						ng;
			int64_t		wrows[64];
			double		wdist[64];

			shim_set_guc_bool("vector.gpu", false);
			nc = ref_hnsw_scan(index, q, 64, cpu, &had_gpu, &reads);
			EXPECT(!had_gpu && reads > 0 && nc >= 40);
			shim_set_guc_bool("vector.gpu", true);
			ng = ref_hnsw_scan(index, q, 64, gpu, &had_gpu, &reads);
			EXPECT(had_gpu && reads == 0);
			for (int j = 0; j < nc; j++)
			{
				const int	row = row_of_tid(cpu[j]);
				double		d = 0;

				EXPECT(row >= 0 && row < n && !heap_row_is_null(&h, row));
				for (int k = 0; k < DIM; k++)
					d += ((double) stored[(size_t) row * DIM + k] - sq[k]) * ((double) stored[(size_t) row * DIM + k] - sq[k]);
				wrows[j] = row;
				wdist[j] = d;
			}
			found += wdist[0] == 0.0;
			if (check_hnsw_stream_n(stored, n, sq, gpu, ng, wrows, wdist, nc, "the reference's hnswbuild, flushed halfway"))
				return 1;
		}
		EXPECT(found >= 39);	/* (an approximate index: a row may miss itself once in a while, not often) */
		EXPECT(shim_notices_raised("hnsw graph no longer fits into maintenance_work_mem") == notices0 + 1);
		fprintf(stderr, "   maintenance_work_mem too small: the deferred half linked at the NOTICE's FlushPages, the rest through HnswInsertTupleOnDisk; %d of 40 rows found first for their own vector, both walks agree\n",
				found);
	}
	free(rows);
	free(live);
	free(stored);
	free(live_row);
	return 0;
}

static int
backend_reference_hnswbuild(void *arg)
{
	(void) arg;
	if (reference_hnswbuild_case(REL_HRBUILD, ORA_OPS_L2, "vector_l2_ops", 97))
		return 1;
	/* vector_cosine_ops (BASELINE configs[3]'s opclass): FUNCTION 2 present -- values normalised on the way in, the scan's
	 * query too; FUNCTION 1 is the negative inner product */
	return reference_hnswbuild_case(REL_HRBUILD_COS, ORA_OPS_COSINE, "vector_cosine_ops", 99);
}
#endif							/* PGV_HAVE_REF_HNSWBUILD */

#if defined(PGV_HAVE_REF_IVFBUILD) && defined(PGV_HAVE_REF_HNSWBUILD)
/* ------------------------------------------------------------------------------------------------ the reference's own PARALLEL CREATE INDEX
 * amcanbuildparallel = true (src/ivfflat.c:203, src/hnsw.c:203): a table of more than a few MB is built by a leader and
 * max_parallel_maintenance_workers workers.  The stand-in server launches them (pgshim_ref_runtime.c: a piece of shared
 * memory as the DSM segment, workers forked by the postmaster that run the entry point the reference names, a shared
 * block counter for the table scan, the workers' sorted runs merged by the leader), and the reference's own
 * IvfflatBeginParallel / IvfflatParallelBuildMain / IvfflatParallelScanAndSort and HnswBeginParallel /
 * HnswParallelBuildMain / HnswParallelScanAndInsert run, two workers and the leader.
 *   ivfflat, vector.gpu = off: the lists a parallel build writes are the serial build's = the oracle's (k-means runs in
 *     the leader before the workers start; every participant's argmin is the same function of the row).
 *   ivfflat, vector.gpu = on: EVERY PARTICIPANT runs the build hooks -- PgvIvfflatBuildBegin / Add / Flush inside
 *     IvfflatParallelScanAndSort, each process with its own device context -- and the index is complete and correct.
 *   hnsw, vector.gpu = on: the participants share one graph in the DSM segment under the reference's locks; the hook gives
 *     up on the first tuple (PgvHnswBuildDefer sees hnswarea set: ADVICE r4 high) and nothing is deferred -- the index
 *     holds every row and both walks agree.  (A build that deferred in the participants would leave an EMPTY index: the
 *     leader's FlushPages links only what the leader's own state holds.) */
static int
backend_reference_parallel_build(void *arg)
{
	Relation	index = shim_open_relation(REL_RBUILD);
	Relation	hindex = shim_open_relation(REL_HRBUILD);
	const int	n = 6000,
				lists = 24,
				hn = 2500;
	float	   *rows = shim_shared_alloc(sizeof(float) * (size_t) n * DIM);	/* every process sees the table */
	HeapRows   *h = shim_shared_alloc(sizeof(HeapRows));
	float	   *live = malloc(sizeof(float) * (size_t) n * DIM);
	int		   *live_row = malloc(sizeof(int) * (size_t) n);
	uint64	   *got = malloc(sizeof(uint64) * 30000);
	Relation	heap;
	IndexInfo	info;
	IndexBuildResult *res;
	MemoryContext ctx;
	pgv_ivf_image img;
	pgv_hnsw_image himg;
	int			nlive = 0,
				used_gpu;
	uint8_t		empty[1] = {0};

	(void) arg;
	scenario = "the reference's own parallel CREATE INDEX";
	cur_ops = ORA_OPS_L2;
	EXPECT(index != NULL && hindex != NULL);
	gen_rows(rows, n, DIM, 51);
	h->rows = rows;
	h->dim = DIM;
	h->toast_every = 7;
	h->null_every = 13;
	heap = heap_of(h, n);
	for (int r = 0; r < n; r++)
		if (!heap_row_is_null(h, r))
		{
			memcpy(live + (size_t) nlive * DIM, rows + (size_t) r * DIM, sizeof(float) * DIM);
			live_row[nlive++] = r;
		}
	memset(&info, 0, sizeof(info));
	shim_set_parallel_workers(2);

	/* ---- ivfflat, the reference's CPU path in three processes */
	{
		ora_prng	a,
					b;
		float	   *want_centers = malloc(sizeof(float) * (size_t) lists * DIM);
		int32_t    *want_list = malloc(sizeof(int32_t) * (size_t) nlive);

		shim_replace_pages(REL_RBUILD, empty, 0);
		shim_set_guc_bool("vector.gpu", false);
		ora_prng_seed(&a, 91);
		shim_prng_hook(ora_prng_double_cb, ora_prng_u32_cb, &a);
		ctx = shim_query_context_begin();
		res = ivfflatbuild(heap, index, &info);
		shim_prng_hook(NULL, NULL, NULL);
		EXPECT(res != NULL && (int) res->heap_tuples == n && (int) res->index_tuples == nlive);
		shim_query_context_end(ctx);
		EXPECT(shim_pinned_buffers() == 0);
		EXPECT(shim_notices_raised("using 2 parallel workers") == 1);	/* (DEBUG1 in the server; the stand-in prints every level) */
		ora_prng_seed(&b, 91);
		(void) ora_prng_u32(&b);
		(void) ora_prng_u32(&b);
		EXPECT(ora_kmeans(ORA_OPS_L2, ORA_F32, DIM, live, nlive, want_centers, lists, &b, NULL) >= 0);
		ora_ivf_assign(ORA_OPS_L2, ORA_F32, DIM, want_centers, lists, live, nlive, want_list, NULL);
		EXPECT(staged_image(REL_RBUILD, &img) == PGV_OK);
		EXPECT(img.lists == lists && img.nrows == nlive);
		EXPECT(memcmp(img.centers, want_centers, sizeof(float) * (size_t) lists * DIM) == 0);
		for (int l = 0; l < lists; l++)
		{
			int64_t		p = img.list_offsets[l];

			for (int i = 0; i < nlive; i++)
				if (want_list[i] == l)
				{
					EXPECT(p < img.list_offsets[l + 1] && img.tids[p] == tid_of_row(live_row[i]));
					EXPECT(memcmp((const float *) img.vectors + (size_t) p * DIM, live + (size_t) i * DIM, sizeof(float) * DIM) == 0);
					p++;
				}
			EXPECT(p == img.list_offsets[l + 1]);
		}
		pgv_host_ivf_image_free(&img);
		fprintf(stderr, "   ivfflat, leader + 2 workers, the reference's CPU path: %d tuples in the serial build's = the oracle's lists\n", nlive);
		free(want_centers);
		free(want_list);
	}

	/* ---- ivfflat, the hooks in every participant */
	shim_replace_pages(REL_RBUILD, empty, 0);
	shim_set_guc_bool("vector.gpu", true);
	shim_set_guc_bool("vector.gpu_pooled", false);
	shim_seed_random(57);
	ctx = shim_query_context_begin();
	res = ivfflatbuild(heap, index, &info);
	EXPECT(res != NULL && (int) res->heap_tuples == n && (int) res->index_tuples == nlive);
	shim_query_context_end(ctx);
	EXPECT(shim_pinned_buffers() == 0);
	EXPECT(staged_image(REL_RBUILD, &img) == PGV_OK);
	EXPECT(img.lists == lists && img.nrows == nlive);
	{
		char	   *seen = calloc((size_t) n, 1);

		for (int l = 0; l < lists; l++)
		{
			int			prev = -1;

			for (int64_t p = img.list_offsets[l]; p < img.list_offsets[l + 1]; p++)
			{
				const int	r = row_of_tid(img.tids[p]);
				const float *x = (const float *) img.vectors + (size_t) p * DIM;
				double		best = INFINITY,
							mine;

				EXPECT(r > prev && r < n && !seen[r] && !heap_row_is_null(h, r));
				prev = r;
				seen[r] = 1;
				EXPECT(memcmp(x, rows + (size_t) r * DIM, sizeof(float) * DIM) == 0);
				for (int c = 0; c < lists; c++)
				{
					double		d = ora_index_distance(ORA_OPS_L2, ORA_F32, DIM, x, (const float *) img.centers + (size_t) c * DIM);

					if (d < best)
						best = d;
				}
				mine = ora_index_distance(ORA_OPS_L2, ORA_F32, DIM, x, (const float *) img.centers + (size_t) l * DIM);
				EXPECT(mine <= best + 1e-5 * fabs(best) + 1e-9);
			}
		}
		free(seen);
	}
	pgv_host_ivf_image_free(&img);
	EXPECT(wait_for_gpu(index, 30.0) == 0);
	for (int i = 0; i < 12; i++)
	{
		const int	r = live_row[(151 * i + 29) % nlive];
		Expected	e = expected_batch(REL_RBUILD, rows + (size_t) r * DIM, PROBES);
		int			nn = ref_scan(index, rows + (size_t) r * DIM, PROBES, i % 4 == 3 ? 300 : 10, got, &used_gpu);

		EXPECT(used_gpu && nn == (i % 4 == 3 ? (300 < e.n ? 300 : e.n) : 10) && got[0] == tid_of_row(r));
		if (check_stream(&e, got, nn, 0, "parallel build with the hooks"))
			return 1;
		expected_free(&e);
	}
	fprintf(stderr, "   ivfflat, leader + 2 workers, the hooks in every participant (three device contexts): every row in a nearest list, once; scans agree\n");

	/* ---- hnsw: the participants keep the reference's path */
	{
		extern int	maintenance_work_mem;
		const int	saved = maintenance_work_mem;
		uint64		cpu[64],
					gpu[64];
		int			had_gpu,
					hlive = 0;
		long		reads;
		double		until;

		for (int r = 0; r < hn; r++)
			hlive += !heap_row_is_null(h, r);
		heap = heap_of(h, hn);
		shim_replace_pages(REL_HRBUILD, empty, 0);
		hnsw_ef_search = 40;
		maintenance_work_mem = 16384;	/* the shared graph area is maintenance_work_mem less 3 MB (src/hnswbuild.c:955-961) */
		shim_set_guc_bool("vector.gpu", true);
		ctx = shim_query_context_begin();
		res = hnswbuild(heap, hindex, &info);
		maintenance_work_mem = saved;
		EXPECT(res != NULL && (int) res->heap_tuples == hn && (int) res->index_tuples == hlive);
		shim_query_context_end(ctx);
		EXPECT(shim_pinned_buffers() == 0);
		EXPECT(staged_hnsw_image(REL_HRBUILD, &himg) == PGV_OK);
		EXPECT(himg.n == hlive && himg.entry >= 0);	/* (no two rows of this table are equal: every row an element) */
		{
			int64_t		links = 0;

			for (int64_t s = 0; s < himg.n; s++)
				for (int64_t j = himg.nbr_start[s]; j < himg.nbr_start[s + 1]; j++)
					links += himg.nbr[j] >= 0;
			EXPECT(links > himg.n * HM);
		}
		pgv_host_hnsw_image_free(&himg);
		until = shim_now() + 30.0;
		for (;;)
		{
			(void) ref_hnsw_scan(hindex, rows, 1, gpu, &had_gpu, &reads);
			if ((had_gpu && reads == 0) || shim_now() > until)
				break;
			usleep(20000);
		}
		EXPECT(had_gpu && reads == 0);
		for (int i = 0; i < 30; i++)
		{
			int			r = (83 * i + 9) % hn;
			const float *q;
			int			nc,
						ng,
						found = 0;
			int64_t		wrows[64];
			double		wdist[64];

			while (heap_row_is_null(h, r))
				r++;
			q = rows + (size_t) r * DIM;
			shim_set_guc_bool("vector.gpu", false);
			nc = ref_hnsw_scan(hindex, q, 64, cpu, &had_gpu, &reads);
			EXPECT(!had_gpu && reads > 0 && nc >= 40);
			shim_set_guc_bool("vector.gpu", true);
			ng = ref_hnsw_scan(hindex, q, 64, gpu, &had_gpu, &reads);
			EXPECT(had_gpu && reads == 0);
			for (int j = 0; j < nc; j++)
			{
				const int	row = row_of_tid(cpu[j]);
				double		d = 0;

				EXPECT(row >= 0 && row < hn && !heap_row_is_null(h, row));
				for (int k = 0; k < DIM; k++)
					d += ((double) rows[(size_t) row * DIM + k] - q[k]) * ((double) rows[(size_t) row * DIM + k] - q[k]);
				wrows[j] = row;
				wdist[j] = d;
				found |= row == r;
			}
			EXPECT(found);
			if (check_hnsw_stream_n(rows, hn, q, gpu, ng, wrows, wdist, nc, "parallel hnsw build"))
				return 1;
		}
		fprintf(stderr, "   hnsw, leader + 2 workers with vector.gpu on: nothing deferred, the shared graph built under the reference's locks holds all %d rows; both walks agree\n",
				hlive);
	}
	shim_set_parallel_workers(0);
	free(live);
	free(live_row);
	free(got);
	return 0;
}
#endif

#if defined(PGV_HAVE_REF_HALFVEC) && defined(PGV_HAVE_REF_IVFBUILD) && defined(PGV_HAVE_REF_HNSWBUILD)
/* ------------------------------------------------------------------------------------------------ the reference's own halfvec opclasses
 * halfvec_l2_ops on both access methods (sql/vector.sql:819-866): the program also holds pgvector's src/halfvec.c and
 * src/halfutils.c (HalfvecInit has picked the F16C kernels where the CPU has them), and two relations are created with the
 * halfvec opclass -- FUNCTION 1-4 are halfvec_l2_squared_distance / halfvec_l2_norm / halfvec_l2_distance, FUNCTION 5 / 3
 * the type-info functions ivfflat_halfvec_support / hnsw_halfvec_support of src/ivfutils.c / src/hnswutils.c
 * (HalfvecSumCenter, HalfvecUpdateCenter with its round-to-half, item size 8 + 2 d, 2 x the dimensions).  BASELINE's
 * configs[4] is this type.
 *   vector.gpu = off, the oracle's pg_prng stream: the reference's serial ivfflatbuild over fp16 rows = the oracle's
 *     (ora_kmeans / ora_ivf_assign with ORA_F16: centers bit for bit as halves, every list in order); its scan of the pages
 *     = the oracle's page reader; its serial hnswbuild = ora_hnsw_build's graph, slot for slot.
 *   vector.gpu = on: the hooks see maxDimensions 2 x IVFFLAT_MAX_DIM / HNSW_MAX_DIM and take the PGV_F16 kernels: build,
 *     staged mirror, scans and walks against the reference's CPU branch. */
#define REL_HVIVF 1010
#define REL_HVHNSW 2005
#define HDIM 64					/* 8 + 128 bytes a value: a 4-byte varlena header in the index tuple, like every real halfvec index row */

static void *
make_halfvec(const uint16 *x, int dim)
{
	Size		size = offsetof(Vector, x) + sizeof(uint16) * (Size) dim;	/* HalfVector: the same 8-byte header (src/halfvec.h:68-74) */
	Vector	   *v = palloc0(size);

	v->vl_len_ = (int32) (size << 2);
	v->dim = (int16) dim;
	memcpy(v->x, x, sizeof(uint16) * (Size) dim);
	return v;
}

typedef struct HalfRows
{
	const uint16 *rows;
	int			dim;
	int			null_every;
}			HalfRows;

static void
half_fetch(int64_t row, Datum *value, bool *isnull, ItemPointerData *tid, void *arg)
{
	const HalfRows *h = arg;

	*tid = itemptr(tid_of_row((int) row));
	*isnull = h->null_every > 0 && row % h->null_every == h->null_every - 1;
	*value = *isnull ? (Datum) 0 : PointerGetDatum(make_halfvec(h->rows + (size_t) row * h->dim, h->dim));
}

static int
ref_scan_half(Relation index, const uint16 *query, int probes, int want, uint64 *got, int *used_gpu)
{
	MemoryContext ctx = shim_query_context_begin();
	ScanKeyData orderby;
	IndexScanDesc scan;
	int			n = 0;

	ivfflat_probes = probes;
	memset(&orderby, 0, sizeof(orderby));
	orderby.sk_argument = PointerGetDatum(make_halfvec(query, HDIM));
	scan = ivfflatbeginscan(index, 0, 1);
	ivfflatrescan(scan, NULL, 0, &orderby, 1);
	*used_gpu = ((IvfflatScanOpaque) scan->opaque)->gpu != NULL;
	while (n < want && ivfflatgettuple(scan, ForwardScanDirection))
		got[n++] = tid_key(&scan->xs_heaptid);
	ivfflatendscan(scan);
	shim_query_context_end(ctx);
	return n;
}

static int
ref_hnsw_scan_half(Relation index, const uint16 *query, int want, uint64 *got, int *had_gpu, long *reads)
{
	MemoryContext ctx = shim_query_context_begin();
	ScanKeyData orderby;
	IndexScanDesc scan;
	long		reads0;
	int			n = 0;

	memset(&orderby, 0, sizeof(orderby));
	orderby.sk_argument = PointerGetDatum(make_halfvec(query, HDIM));
	scan = hnswbeginscan(index, 0, 1);
	hnswrescan(scan, NULL, 0, &orderby, 1);
	*had_gpu = ((HnswScanOpaque) scan->opaque)->gpu != NULL;
	reads0 = shim_buffer_reads();
	while (n < want && hnswgettuple(scan, ForwardScanDirection))
		got[n++] = tid_key(&scan->xs_heaptid);
	*reads = shim_buffer_reads() - reads0;
	hnswendscan(scan);
	shim_query_context_end(ctx);
	return n;
}

/* got[0..n) against the oracle's reading of the relation's pages for a halfvec query (the stream of the probed lists) */
static int
check_half_scan(Oid relid, const uint16 *q, int probes, const uint64 *got, int n, int want, const char *what)
{
	uint32_t	nblocks;
	const uint8_t *pages = shim_relation_pages(relid, &nblocks);
	const int	cap = 1 << 16;
	uint64	   *tids = malloc(sizeof(uint64) * (size_t) cap);
	double	   *dist = malloc(sizeof(double) * (size_t) cap);
	int64_t		scanned = 0;
	int			en = ora_pages_search(pages, nblocks, ORA_OPS_L2, ORA_F16, q, probes, cap, tids, dist, &scanned);
	int			bad = 0;

	if (n != (want < en ? want : en))
	{
		fprintf(stderr, "[%s] %s: %d tuples, the oracle's stream has %d (asked for %d)\n", scenario, what, n, en, want);
		bad = 1;
	}
	for (int i = 0; i < n && !bad; i++)
	{
		double		d = -1;

		for (int j = 0; j < en; j++)
			if (tids[j] == got[i])
			{
				d = dist[j];
				break;
			}
		if (d < 0 || fabs(d - dist[i]) > 1e-4 * fabs(dist[i]) + 1e-6)
		{
			fprintf(stderr, "[%s] %s: position %d (tid %llx) at %.9g, the oracle's stream has %.9g there\n", scenario, what, i,
					(unsigned long long) got[i], d, dist[i]);
			bad = 1;
		}
	}
	free(tids);
	free(dist);
	return bad;
}

static int
backend_reference_halfvec(void *arg)
{
	Relation	index = shim_open_relation(REL_HVIVF);
	Relation	hindex = shim_open_relation(REL_HVHNSW);
	const int	n = 4000,
				lists = 16,
				hn = 2000,
				m = HM,
				efc = 32;
	float	   *frows = malloc(sizeof(float) * (size_t) n * HDIM);
	uint16	   *rows = malloc(sizeof(uint16) * (size_t) n * HDIM);
	uint16	   *live = malloc(sizeof(uint16) * (size_t) n * HDIM);
	int		   *live_row = malloc(sizeof(int) * (size_t) n);
	uint64	   *got = malloc(sizeof(uint64) * 30000);
	HalfRows	h;
	ShimHeapDef def;
	Relation	heap;
	IndexInfo	info;
	IndexBuildResult *res;
	MemoryContext ctx;
	int			nlive = 0,
				used_gpu;
	uint8_t		empty[1] = {0};

	(void) arg;
	scenario = "the reference's own halfvec opclasses";
	EXPECT(index != NULL && hindex != NULL);
	EXPECT(IvfflatGetTypeInfo(index)->maxDimensions == IVFFLAT_MAX_DIM * 2 && HnswGetTypeInfo(hindex)->maxDimensions == HNSW_MAX_DIM * 2);
	gen_rows(frows, n, HDIM, 71);
	for (size_t i = 0; i < (size_t) n * HDIM; i++)
		rows[i] = ora_float_to_half(frows[i]);
	h.rows = rows;
	h.dim = HDIM;
	h.null_every = 19;
	def.nrows = n;
	def.rows_per_block = 50;
	def.fetch = half_fetch;
	def.arg = &h;
	heap = shim_heap_relation(&def);
	for (int r = 0; r < n; r++)
		if (r % 19 != 18)
		{
			memcpy(live + (size_t) nlive * HDIM, rows + (size_t) r * HDIM, sizeof(uint16) * HDIM);
			live_row[nlive++] = r;
		}
	memset(&info, 0, sizeof(info));

	/* ---- ivfflat (halfvec_l2_ops): the reference's serial build = the oracle's, in halves */
	{
		ora_prng	a,
					b;
		uint16	   *want_centers = malloc(sizeof(uint16) * (size_t) lists * HDIM);
		int32_t    *want_list = malloc(sizeof(int32_t) * (size_t) nlive);
		pgv_ivf_image img;
		pgv_rel		rel;
		uint32_t	nblocks;
		int			iterations;

		shim_set_guc_bool("vector.gpu", false);
		ora_prng_seed(&a, 101);
		shim_prng_hook(ora_prng_double_cb, ora_prng_u32_cb, &a);
		ctx = shim_query_context_begin();
		res = ivfflatbuild(heap, index, &info);
		shim_prng_hook(NULL, NULL, NULL);
		EXPECT(res != NULL && (int) res->heap_tuples == n && (int) res->index_tuples == nlive);
		shim_query_context_end(ctx);
		EXPECT(shim_pinned_buffers() == 0);
		ora_prng_seed(&b, 101);
		(void) ora_prng_u32(&b);
		(void) ora_prng_u32(&b);
		iterations = ora_kmeans(ORA_OPS_L2, ORA_F16, HDIM, live, nlive, want_centers, lists, &b, NULL);
		EXPECT(iterations >= 0 && a.s0 == b.s0 && a.s1 == b.s1);
		ora_ivf_assign(ORA_OPS_L2, ORA_F16, HDIM, want_centers, lists, live, nlive, want_list, NULL);
		pgv_rel_init(&rel);
		rel.pages = (uint8_t *) shim_relation_pages(REL_HVIVF, &nblocks);
		rel.nblocks = rel.cap = nblocks;
		EXPECT(pgv_host_ivf_stage(&rel, PGV_F16, &img) == PGV_OK);
		EXPECT(img.dim == HDIM && img.lists == lists && img.nrows == nlive);
		if (memcmp(img.centers, want_centers, sizeof(uint16) * (size_t) lists * HDIM) != 0)
		{
			fprintf(stderr, "halfvec: the centers in the reference's list pages are not the oracle's\n");
			return 1;
		}
		for (int l = 0; l < lists; l++)
		{
			int64_t		p = img.list_offsets[l];

			for (int i = 0; i < nlive; i++)
				if (want_list[i] == l)
				{
					if (p >= img.list_offsets[l + 1] || img.tids[p] != tid_of_row(live_row[i]) ||
						memcmp((const uint16 *) img.vectors + (size_t) p * HDIM, live + (size_t) i * HDIM, sizeof(uint16) * HDIM) != 0)
					{
						fprintf(stderr, "halfvec: list %d of the reference's build: position %lld is not the oracle's row %d\n", l,
								(long long) (p - img.list_offsets[l]), live_row[i]);
						return 1;
					}
					p++;
				}
			EXPECT(p == img.list_offsets[l + 1]);
		}
		pgv_host_ivf_image_free(&img);
		/* the reference's scan (halfvec_l2_squared_distance through the F16C or the default kernel) = the oracle's reader */
		for (int i = 0; i < 10; i++)
		{
			const uint16 *q = rows + (size_t) live_row[(113 * i) % nlive] * HDIM;
			int			want = i % 3 == 2 ? 250 : 10;
			int			nn = ref_scan_half(index, q, PROBES, want, got, &used_gpu);

			EXPECT(!used_gpu && got[0] == tid_of_row(live_row[(113 * i) % nlive]));
			if (check_half_scan(REL_HVIVF, q, PROBES, got, nn, want, "halfvec, the reference's CPU build and scan"))
				return 1;
		}
		fprintf(stderr, "   halfvec_l2_ops ivfflat: the reference's serial build (%d rows x %d halves, %d lists, %d Elkan iterations) = the oracle's: centers bit for bit, every list in order; its scan = the oracle's page reader\n",
				n, HDIM, lists, iterations);
		free(want_centers);
		free(want_list);
	}

	/* ---- vector.gpu_kmeans = off: the reference's Elkan centers (as halves, to the bit), the fp16 argmin kernel's lists
	 * against the CPU build's */
	{
		pgv_ivf_image ic,
					ig;
		pgv_rel		rel;
		uint32_t	nb_cpu,
					nb_gpu;
		uint8_t    *cpu_pages;
		ora_prng	a;
		int			moved = 0;
		int		   *list_cpu = malloc(sizeof(int) * (size_t) n);

		pgv_rel_init(&rel);
		rel.pages = (uint8_t *) shim_relation_pages(REL_HVIVF, &nb_cpu);
		rel.nblocks = rel.cap = nb_cpu;
		EXPECT(pgv_host_ivf_stage(&rel, PGV_F16, &ic) == PGV_OK);
		cpu_pages = malloc((size_t) nb_cpu * 8192);
		memcpy(cpu_pages, rel.pages, (size_t) nb_cpu * 8192);
		shim_replace_pages(REL_HVIVF, empty, 0);
		shim_set_guc_bool("vector.gpu", true);
		shim_set_guc_bool("vector.gpu_kmeans", false);
		ora_prng_seed(&a, 101);
		shim_prng_hook(ora_prng_double_cb, ora_prng_u32_cb, &a);
		ctx = shim_query_context_begin();
		res = ivfflatbuild(heap, index, &info);
		shim_prng_hook(NULL, NULL, NULL);
		shim_set_guc_bool("vector.gpu_kmeans", true);
		EXPECT(res != NULL && (int) res->index_tuples == nlive);
		shim_query_context_end(ctx);
		rel.pages = (uint8_t *) shim_relation_pages(REL_HVIVF, &nb_gpu);
		rel.nblocks = rel.cap = nb_gpu;
		EXPECT(pgv_host_ivf_stage(&rel, PGV_F16, &ig) == PGV_OK);
		EXPECT(ig.lists == ic.lists && ig.nrows == ic.nrows);
		EXPECT(memcmp(ig.centers, ic.centers, sizeof(uint16) * (size_t) lists * HDIM) == 0);
		for (int r = 0; r < n; r++)
			list_cpu[r] = -1;
		for (int l = 0; l < lists; l++)
			for (int64_t p2 = ic.list_offsets[l]; p2 < ic.list_offsets[l + 1]; p2++)
				list_cpu[row_of_tid(ic.tids[p2])] = l;
		for (int l = 0; l < lists; l++)
			for (int64_t p2 = ig.list_offsets[l]; p2 < ig.list_offsets[l + 1]; p2++)
				moved += list_cpu[row_of_tid(ig.tids[p2])] != l;
		EXPECT(moved * 1000 <= nlive);
		fprintf(stderr, "   halfvec_l2_ops ivfflat: vector.gpu_kmeans = off -- Elkan's centers to the bit, the device's argmins: %d of %d rows in another list than the CPU build's; the index %s\n",
				moved, nlive, nb_gpu == nb_cpu && memcmp(rel.pages, cpu_pages, (size_t) nb_cpu * 8192) == 0 ? "is the CPU build's, byte for byte" : "differs in bytes");
		free(list_cpu);
		free(cpu_pages);
		pgv_host_ivf_image_free(&ic);
		pgv_host_ivf_image_free(&ig);
	}

	/* ---- ivfflat with the hooks: PGV_F16 k-means, argmins, mirror, scans */
	shim_replace_pages(REL_HVIVF, empty, 0);
	shim_set_guc_bool("vector.gpu", true);
	shim_set_guc_bool("vector.gpu_pooled", false);
	shim_seed_random(103);
	ctx = shim_query_context_begin();
	res = ivfflatbuild(heap, index, &info);
	EXPECT(res != NULL && (int) res->heap_tuples == n && (int) res->index_tuples == nlive);
	shim_query_context_end(ctx);
	EXPECT(shim_notices_raised("pgvector GPU path: 3790 rows assigned on the device") == 2);	/* 4000 less 210 NULLs; this build and the one before */
	{
		pgv_ivf_image img;
		pgv_rel		rel;
		uint32_t	nblocks;
		char	   *seen = calloc((size_t) n, 1);

		pgv_rel_init(&rel);
		rel.pages = (uint8_t *) shim_relation_pages(REL_HVIVF, &nblocks);
		rel.nblocks = rel.cap = nblocks;
		EXPECT(pgv_host_ivf_stage(&rel, PGV_F16, &img) == PGV_OK);
		EXPECT(img.lists == lists && img.nrows == nlive);
		for (int l = 0; l < lists; l++)
			for (int64_t p = img.list_offsets[l]; p < img.list_offsets[l + 1]; p++)
			{
				const int	r = row_of_tid(img.tids[p]);
				const uint16 *x = (const uint16 *) img.vectors + (size_t) p * HDIM;
				double		best = INFINITY,
							mine;

				EXPECT(r >= 0 && r < n && !seen[r] && r % 19 != 18);
				seen[r] = 1;
				EXPECT(memcmp(x, rows + (size_t) r * HDIM, sizeof(uint16) * HDIM) == 0);
				for (int c = 0; c < lists; c++)
				{
					double		d = ora_index_distance(ORA_OPS_L2, ORA_F16, HDIM, x, (const uint16 *) img.centers + (size_t) c * HDIM);

					if (d < best)
						best = d;
				}
				mine = ora_index_distance(ORA_OPS_L2, ORA_F16, HDIM, x, (const uint16 *) img.centers + (size_t) l * HDIM);
				EXPECT(mine <= best + 1e-5 * fabs(best) + 1e-9);
			}
		pgv_host_ivf_image_free(&img);
		free(seen);
	}
	{
		double		until = shim_now() + 30.0;

		for (;;)
		{
			(void) ref_scan_half(index, rows, PROBES, 1, got, &used_gpu);
			if (used_gpu || shim_now() > until)
				break;
			usleep(20000);
		}
		EXPECT(used_gpu);
	}
	for (int i = 0; i < 12; i++)
	{
		const uint16 *q = rows + (size_t) live_row[(127 * i + 3) % nlive] * HDIM;
		int			want = i % 3 == 2 ? 250 : 10;
		int			nn = ref_scan_half(index, q, PROBES, want, got, &used_gpu);

		EXPECT(used_gpu && got[0] == tid_of_row(live_row[(127 * i + 3) % nlive]));
		if (check_half_scan(REL_HVIVF, q, PROBES, got, nn, want, "halfvec, build and scan through the hooks"))
			return 1;
	}
	fprintf(stderr, "   halfvec_l2_ops ivfflat with the hooks (PGV_F16 k-means, %d argmins, mirror, scans): every row in a nearest list; device scans = the oracle's page reader\n",
			nlive);

	/* ---- hnsw (halfvec_l2_ops): the reference's serial in-memory build = the oracle's graph */
	{
		ora_prng	a;
		ora_hnsw   *g;
		pgv_hnsw_image img;
		pgv_rel		rel;
		uint32_t	nblocks;
		int64_t		ne;
		int			hlive = 0,
					entry_level;
		int		   *slot_of_row = malloc(sizeof(int) * (size_t) hn);
		int		   *slot_of_element;
		uint64		cpu[64],
					gpu[64];
		int			had_gpu;
		long		reads;

		for (int r = 0; r < hn; r++)
			hlive += r % 19 != 18;
		def.nrows = hn;
		heap = shim_heap_relation(&def);
		hnsw_ef_search = 40;
		shim_set_guc_bool("vector.gpu", false);
		ora_prng_seed(&a, 107);
		shim_prng_hook(ora_prng_double_cb, ora_prng_u32_cb, &a);
		ctx = shim_query_context_begin();
		res = hnswbuild(heap, hindex, &info);
		shim_prng_hook(NULL, NULL, NULL);
		EXPECT(res != NULL && (int) res->heap_tuples == hn && (int) res->index_tuples == hlive);
		shim_query_context_end(ctx);
		g = ora_hnsw_build(ORA_OPS_L2, ORA_F16, HDIM, live, hlive, m, efc, 107);
		EXPECT(g != NULL);
		ne = ora_hnsw_num_elements(g);
		pgv_rel_init(&rel);
		rel.pages = (uint8_t *) shim_relation_pages(REL_HVHNSW, &nblocks);
		rel.nblocks = rel.cap = nblocks;
		EXPECT(pgv_host_hnsw_stage(&rel, PGV_F16, &img) == PGV_OK);
		EXPECT(img.dim == HDIM && img.m == m && img.n == ne);
		for (int r = 0; r < hn; r++)
			slot_of_row[r] = -1;
		for (int64_t s = 0; s < img.n; s++)
		{
			int			r = row_of_tid(img.heaptids[(size_t) s * 10]);

			EXPECT(r >= 0 && r < hn && slot_of_row[r] == -1);
			slot_of_row[r] = (int) s;
		}
		slot_of_element = malloc(sizeof(int) * (size_t) ne);
		for (int64_t e = 0; e < ne; e++)
		{
			slot_of_element[e] = slot_of_row[live_row[ora_hnsw_element_row(g, e)]];
			EXPECT(slot_of_element[e] >= 0);
		}
		for (int64_t e = 0; e < ne; e++)
		{
			const int	s = slot_of_element[e];
			const int	level = ora_hnsw_level(g, e);

			EXPECT(img.levels[s] == level);
			EXPECT(memcmp((const uint16 *) img.vectors + (size_t) s * HDIM, live + (size_t) ora_hnsw_element_row(g, e) * HDIM, sizeof(uint16) * HDIM) == 0);
			for (int lc = level; lc >= 0; lc--)
			{
				int32_t		want[2 * HM];
				const int	lm = lc == 0 ? 2 * m : m;
				const int	nw = ora_hnsw_neighbors(g, e, lc, want);
				const int32_t *have = img.nbr + img.nbr_start[s] + (int64_t) (level - lc) * m;

				for (int i = 0; i < lm; i++)
					if (have[i] != (i < nw ? slot_of_element[want[i]] : -1))
					{
						fprintf(stderr, "halfvec hnsw: element %lld layer %d slot %d: the reference's neighbor tuple has %d, the oracle's array %d\n",
								(long long) e, lc, i, have[i], i < nw ? slot_of_element[want[i]] : -1);
						return 1;
					}
			}
		}
		EXPECT(img.entry == slot_of_element[ora_hnsw_entry_point(g, &entry_level)]);
		fprintf(stderr, "   halfvec_l2_ops hnsw: the reference's serial build (%d rows, m %d) = the oracle's graph: %lld elements, levels, every neighbor slot, the entry point\n",
				hn, m, (long long) ne);
		pgv_host_hnsw_image_free(&img);
		ora_hnsw_free(g);
		free(slot_of_element);
		free(slot_of_row);

		/* vector.gpu_hnsw_build_batch = 1 on the same level draws: the fp16 kernels' graph against the reference's serial one */
		{
			pgv_hnsw_image ic,
						ig;
			uint32_t	nb_cpu,
						nb_gpu;
			uint8_t    *cpu_pages;
			int64_t		slots,
						differing = 0;

			rel.pages = (uint8_t *) shim_relation_pages(REL_HVHNSW, &nb_cpu);
			rel.nblocks = rel.cap = nb_cpu;
			EXPECT(pgv_host_hnsw_stage(&rel, PGV_F16, &ic) == PGV_OK);
			cpu_pages = malloc((size_t) nb_cpu * 8192);
			memcpy(cpu_pages, rel.pages, (size_t) nb_cpu * 8192);
			shim_replace_pages(REL_HVHNSW, empty, 0);
			shim_set_guc_bool("vector.gpu", true);
			shim_set_guc_int("vector.gpu_hnsw_build_batch", 1);
			ora_prng_seed(&a, 107);
			shim_prng_hook(ora_prng_double_cb, ora_prng_u32_cb, &a);
			ctx = shim_query_context_begin();
			res = hnswbuild(heap, hindex, &info);
			shim_prng_hook(NULL, NULL, NULL);
			shim_set_guc_int("vector.gpu_hnsw_build_batch", 1024);
			EXPECT(res != NULL && (int) res->index_tuples == hlive);
			shim_query_context_end(ctx);
			rel.pages = (uint8_t *) shim_relation_pages(REL_HVHNSW, &nb_gpu);
			rel.nblocks = rel.cap = nb_gpu;
			EXPECT(pgv_host_hnsw_stage(&rel, PGV_F16, &ig) == PGV_OK);
			EXPECT(ig.n == ic.n && memcmp(ig.levels, ic.levels, sizeof(int32_t) * (size_t) ic.n) == 0);
			slots = ic.nbr_start[ic.n];
			for (int64_t j = 0; j < slots; j++)
				differing += ig.nbr[j] != ic.nbr[j];
			EXPECT(differing * 200 <= slots);	/* (halves: equal distances are likelier than among floats; a few slots at most) */
			fprintf(stderr, "   halfvec_l2_ops hnsw: the hooks at vector.gpu_hnsw_build_batch = 1: %lld of %lld neighbor slots differ from the reference's serial build; the index %s\n",
					(long long) differing, (long long) slots,
					nb_gpu == nb_cpu && memcmp(rel.pages, cpu_pages, (size_t) nb_cpu * 8192) == 0 ? "is the CPU build's, byte for byte" : "differs in bytes");
			free(cpu_pages);
			pgv_host_hnsw_image_free(&ic);
			pgv_host_hnsw_image_free(&ig);
		}

		/* the hooks: every element deferred, linked by the PGV_F16 kernels; the reference's walk of its own pages against
		 * the device walk over the staged mirror */
		shim_replace_pages(REL_HVHNSW, empty, 0);
		shim_set_guc_bool("vector.gpu", true);
		shim_seed_random(109);
		for (int i = 0; i < 16; i++)
			(void) RandomDouble();
		ctx = shim_query_context_begin();
		res = hnswbuild(heap, hindex, &info);
		EXPECT(res != NULL && (int) res->heap_tuples == hn && (int) res->index_tuples == hlive);
		shim_query_context_end(ctx);
		{
			double		until = shim_now() + 30.0;

			for (;;)
			{
				(void) ref_hnsw_scan_half(hindex, rows, 1, gpu, &had_gpu, &reads);
				if ((had_gpu && reads == 0) || shim_now() > until)
					break;
				usleep(20000);
			}
			EXPECT(had_gpu && reads == 0);
		}
		for (int i = 0; i < 24; i++)
		{
			int			r = (71 * i + 5) % hn;
			const uint16 *q;
			int			nc,
						ng;

			while (r % 19 == 18)
				r++;
			q = rows + (size_t) r * HDIM;
			shim_set_guc_bool("vector.gpu", false);
			nc = ref_hnsw_scan_half(hindex, q, 48, cpu, &had_gpu, &reads);
			EXPECT(!had_gpu && reads > 0 && nc >= 40);
			shim_set_guc_bool("vector.gpu", true);
			ng = ref_hnsw_scan_half(hindex, q, 48, gpu, &had_gpu, &reads);
			EXPECT(had_gpu && reads == 0 && ng == nc);
			EXPECT(cpu[0] == tid_of_row(r) && gpu[0] == tid_of_row(r));
			/* position by position the same distance (rows at float-equal distances may swap) */
			for (int j = 0; j < nc; j++)
			{
				const int	rc = row_of_tid(cpu[j]),
							rg = row_of_tid(gpu[j]);
				double		dc = ora_index_distance(ORA_OPS_L2, ORA_F16, HDIM, q, rows + (size_t) rc * HDIM),
							dg = ora_index_distance(ORA_OPS_L2, ORA_F16, HDIM, q, rows + (size_t) rg * HDIM);

				if (fabs(dc - dg) > 1e-4 * fabs(dc) + 1e-6)
				{
					fprintf(stderr, "halfvec hnsw: position %d: the reference's walk has row %d at %.7g, the device's row %d at %.7g\n", j, rc, dc, rg, dg);
					return 1;
				}
			}
		}
		fprintf(stderr, "   halfvec_l2_ops hnsw with the hooks: every element linked by the fp16 kernels; the reference's walk and the device walk agree\n");
	}
	free(frows);
	free(rows);
	free(live);
	free(live_row);
	free(got);
	return 0;
}
#endif

#if defined(PGV_HAVE_REF_IVFBUILD) && defined(PGV_HAVE_REF_HNSWBUILD)
/* ------------------------------------------------------------------------------------------------ the product's page writers against the reference's pages
 * (stand-in device only: nothing here touches a device, and the GPU box's run has nothing to add.)  The reference's own
 * CREATE INDEX writes an index; the product's stager reads it into the list-major image; the PRODUCT'S page writer
 * (pgvector_amd/host/ivf_pages.c: the meta page, list pages, entry pages, line pointers, chain links it lays out on its
 * own) writes that image back -- and must arrive at the reference's pages byte for byte.  That is the on-disk format of
 * src/ivfflat.h:46-52, :251-275 pinned to what the reference's code actually writes, page headers and padding included. */
static int
backend_writer_against_reference(void *arg)
{
	static const struct
	{
		Oid			relid;
		const char *opclass;
	}			cases[] = {{REL_RBUILD_COS, "vector_cosine_ops"}, {REL_RBUILD_IP, "vector_ip_ops"}, {REL_HVIVF, "halfvec_l2_ops"}};

	(void) arg;
	scenario = "the product's page writer against the reference's pages";
	if (!mock_hip_set_arena)
		return 0;
	for (int c = 0; c < 3; c++)
	{
		const pgv_dtype dtype = cases[c].relid == REL_HVIVF ? PGV_F16 : PGV_F32;
		uint32_t	nblocks;
		const uint8_t *pages = shim_relation_pages(cases[c].relid, &nblocks);
		pgv_rel		rel,
					mine;
		pgv_ivf_image img;
		size_t		differing = 0,
					first = 0;

		pgv_rel_init(&rel);
		rel.pages = (uint8_t *) pages;
		rel.nblocks = rel.cap = nblocks;
		EXPECT(nblocks > 2 && pgv_host_ivf_stage(&rel, dtype, &img) == PGV_OK);
		pgv_rel_init(&mine);
		EXPECT(pgv_host_ivf_write_index(&mine, dtype, img.dim, img.lists, img.centers, img.list_offsets, img.vectors, img.tids) == PGV_OK);
		EXPECT(mine.nblocks == nblocks);
		for (size_t o = 0; o < (size_t) nblocks * 8192; o++)
			if (mine.pages[o] != pages[o])
			{
				if (differing++ == 0)
					first = o;
			}
		if (differing)
		{
			fprintf(stderr, "%s: the product's writer differs from the reference's pages in %zu bytes; first: block %zu offset %zu (%02x, the reference wrote %02x)\n",
					cases[c].opclass, differing, first / 8192, first % 8192, mine.pages[first], pages[first]);
			return 1;
		}
		fprintf(stderr, "   %s: the image staged from the reference's index, written back by the product's page writer = the reference's %u pages, byte for byte\n",
				cases[c].opclass, (unsigned) nblocks);
		pgv_host_ivf_image_free(&img);
		pgv_rel_free(&mine);
	}
	/* the same for hnsw (src/hnsw.h:372-392: element tuples with their heap TIDs and neighbor TID, neighbor tuples, the meta
	 * page's entry point): the halfvec index the reference's FlushPages wrote last, staged, written back by
	 * pgvector_amd/host/hnsw_pages.c */
	{
		uint32_t	nblocks;
		const uint8_t *pages = shim_relation_pages(REL_HVHNSW, &nblocks);
		pgv_rel		rel,
					mine;
		pgv_hnsw_image img;
		uint64_t   *tids;
		size_t		differing = 0,
					first = 0;

		pgv_rel_init(&rel);
		rel.pages = (uint8_t *) pages;
		rel.nblocks = rel.cap = nblocks;
		EXPECT(nblocks > 2 && pgv_host_hnsw_stage(&rel, PGV_F16, &img) == PGV_OK);
		/* the writer takes the elements in insertion order and, like CreateGraphPages walking the list from its head
		 * (src/hnswbuild.c:150-243: newest first), writes them last to first: the staged slots are in page order, so they
		 * go in reversed, element s as n - 1 - s */
		{
			const int64_t n = img.n;
			uint16	   *vec = malloc(sizeof(uint16) * (size_t) n * img.dim);
			int32_t    *levels = malloc(sizeof(int32_t) * (size_t) n);
			int64_t    *nbr_start = malloc(sizeof(int64_t) * (size_t) (n + 1));
			int32_t    *nbr = malloc(sizeof(int32_t) * (size_t) img.nbr_start[n]);

			tids = malloc(sizeof(uint64_t) * (size_t) n);
			nbr_start[0] = 0;
			for (int64_t r = 0; r < n; r++)
			{
				const int64_t s2 = n - 1 - r;
				const int64_t len = img.nbr_start[s2 + 1] - img.nbr_start[s2];

				EXPECT(img.heaptids[(size_t) s2 * 10 + 1] == UINT64_MAX);	/* (no two rows of that table are equal: one heap TID each) */
				tids[r] = img.heaptids[(size_t) s2 * 10];
				levels[r] = img.levels[s2];
				memcpy(vec + (size_t) r * img.dim, (const uint16 *) img.vectors + (size_t) s2 * img.dim, sizeof(uint16) * (size_t) img.dim);
				nbr_start[r + 1] = nbr_start[r] + len;
				for (int64_t j = 0; j < len; j++)
				{
					const int32_t v = img.nbr[img.nbr_start[s2] + j];

					nbr[nbr_start[r] + j] = v < 0 ? -1 : (int32_t) (n - 1 - v);
				}
			}
			pgv_rel_init(&mine);
			EXPECT(pgv_host_hnsw_write_index(&mine, PGV_F16, img.dim, img.m, img.ef_construction, n, vec, tids, levels, nbr_start, nbr, NULL,
											 img.entry < 0 ? -1 : (int32_t) (n - 1 - img.entry)) == PGV_OK);
			free(vec);
			free(levels);
			free(nbr_start);
			free(nbr);
		}
		EXPECT(mine.nblocks == nblocks);
		for (size_t o = 0; o < (size_t) nblocks * 8192; o++)
			if (mine.pages[o] != pages[o])
			{
				if (differing++ == 0)
					first = o;
			}
		if (differing)
		{
			fprintf(stderr, "halfvec_l2_ops hnsw: the product's writer differs from the reference's pages in %zu bytes; first: block %zu offset %zu (%02x, the reference wrote %02x)\n",
					differing, first / 8192, first % 8192, mine.pages[first], pages[first]);
			return 1;
		}
		fprintf(stderr, "   halfvec_l2_ops hnsw: the graph staged from the reference's index, written back by the product's page writer = the reference's %u pages, byte for byte\n",
				(unsigned) nblocks);
		free(tids);
		pgv_host_hnsw_image_free(&img);
		pgv_rel_free(&mine);
	}
	return 0;
}
#endif

int
main(void)
{
	ShimOpclass l2 = {0, IVFFLAT_MAX_DIM, false, false, 0};
	ShimOpclass hnsw_l2 = {1, HNSW_MAX_DIM, false, false, 0};
	ShimOpclass ip = {0, IVFFLAT_MAX_DIM, false, true, 1};	/* vector_ip_ops: FUNCTION 4 (spherical k-means), no FUNCTION 2 */
	uint8_t		empty[1] = {0};
	size_t		arena_bytes = 0;
	void	   *arena;
	int			failed = 0;

	/* (stand-in device only: every 9th export of a worker fails like hipIpcGetMemHandle did once on the device; the
	 * worker's second try must make that invisible to every scenario below) */
	setenv("MOCK_HIP_EXPORT_FAIL_EVERY", "9", 1);
	board = mmap(NULL, sizeof(Board), PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
	memset((void *) board, 0, sizeof(Board));
	shim_postmaster_init((size_t) 832 << 20, mock_hip_set_arena ? (size_t) 512 << 20 : 0);
	arena = shim_arena_base(&arena_bytes);
	if (mock_hip_set_arena && arena)
		mock_hip_set_arena(arena, arena_bytes);
	/* shared_preload_libraries = 'vector': _PG_init in the postmaster */
	process_shared_preload_libraries_in_progress = true;
#ifdef PGV_HAVE_REF_HALFVEC
	{
		extern void HalfvecInit(void);

		HalfvecInit();			/* (_PG_init, src/vector.c:59: the reference's src/halfutils.c picks its kernels by CPUID) */
	}
#endif
	PgvGpuInit();
	shim_postmaster_run_shmem_hooks();
	shim_register_bgworker_function("PgvWorkerMain", PgvWorkerMain);
#if defined(PGV_HAVE_REF_IVFBUILD) || defined(PGV_HAVE_REF_HNSWBUILD)
	shim_register_bgworker_function("ParallelWorkerMain", ParallelWorkerMain);
#endif
	/* empty relations: their pages come from the build */
	shim_create_relation(REL_IVF, &l2, empty, 0, DIM);
	shim_create_relation(REL_BATCH, &l2, empty, 0, 8);
	shim_create_relation(REL_HNSW, &hnsw_l2, empty, 0, DIM);
	shim_create_relation(REL_SLOW, &l2, empty, 0, DIM);
	shim_create_relation(REL_HNSW2, &hnsw_l2, empty, 0, DIM);
	shim_create_relation(REL_IP, &ip, empty, 0, DIM);
	shim_create_relation(REL_WIDE, &l2, empty, 0, IVFFLAT_MAX_DIM);
#ifdef PGV_HAVE_REF_IVFINSERT
	shim_create_relation(REL_INS, &l2, empty, 0, DIM);
#endif
#ifdef PGV_HAVE_REF_HNSWINSERT
	shim_create_relation(REL_HINS, &hnsw_l2, empty, 0, DIM);
#endif
#ifdef PGV_HAVE_REF_IVFBUILD
	shim_create_relation(REL_RBUILD, &l2, empty, 0, DIM);
	shim_set_reloptions(REL_RBUILD, 24, 0);	/* WITH (lists = 24) */
	{
		ShimOpclass cosine = {0, IVFFLAT_MAX_DIM, true, true, 1, 0};	/* vector_cosine_ops: FUNCTION 1 negative inner product, 2 and 4 vector_norm */

		shim_create_relation(REL_RBUILD_COS, &cosine, empty, 0, DIM);
		shim_set_reloptions(REL_RBUILD_COS, 24, 0);
		shim_create_relation(REL_RBUILD_IP, &ip, empty, 0, DIM);
		shim_set_reloptions(REL_RBUILD_IP, 24, 0);
	}
#endif
#ifdef PGV_HAVE_REF_HNSWBUILD
	shim_create_relation(REL_HRBUILD, &hnsw_l2, empty, 0, DIM);
	shim_set_reloptions(REL_HRBUILD, HM, 32);	/* WITH (m = 8, ef_construction = 32) */
	{
		ShimOpclass hnsw_cosine = {1, HNSW_MAX_DIM, true, false, 1, 0};	/* vector_cosine_ops: FUNCTION 1 negative inner product, 2 vector_norm */

		shim_create_relation(REL_HRBUILD_COS, &hnsw_cosine, empty, 0, DIM);
		shim_set_reloptions(REL_HRBUILD_COS, HM, 32);
	}
#endif
#if defined(PGV_HAVE_REF_HALFVEC) && defined(PGV_HAVE_REF_IVFBUILD) && defined(PGV_HAVE_REF_HNSWBUILD)
	{
		ShimOpclass half_l2 = {0, IVFFLAT_MAX_DIM * 2, false, false, 0, 1};	/* halfvec_l2_ops */
		ShimOpclass hnsw_half_l2 = {1, HNSW_MAX_DIM * 2, false, false, 0, 1};

		shim_create_relation(REL_HVIVF, &half_l2, empty, 0, HDIM);
		shim_set_reloptions(REL_HVIVF, 16, 0);
		shim_create_relation(REL_HVHNSW, &hnsw_half_l2, empty, 0, HDIM);
		shim_set_reloptions(REL_HVHNSW, HM, 32);
	}
#endif

	failed |= run_phase("CREATE INDEX through the build hooks", backend_build, 1, NULL, 300.0);
	if (!failed)
		failed |= run_phase("own-context scans", backend_scan_own, 1, NULL, 300.0);
#ifdef PGV_HAVE_REF_IVFSCAN
	if (!failed)
		failed |= run_phase("the reference's own ivfflatgettuple", backend_reference_scan, 1, NULL, 300.0);
#endif
	if (!failed)
	{
		void	   *ids[6] = {(void *) 1, (void *) 2, (void *) 3, (void *) 4, (void *) 5, (void *) 6};

		failed |= run_phase("six pooled backends", backend_scan_pooled, 6, ids, 300.0);
	}
	if (!failed)
	{
		int			pids[2],
					codes[2];

		pids[0] = shim_fork_backend(backend_cursor, NULL);
		pids[1] = shim_fork_backend(backend_insert, NULL);
		failed |= shim_postmaster_wait(pids, 2, codes, 120.0) || codes[0] || codes[1];
		fprintf(stderr, "phase %-40s 2 process(es): %s\n", "insert / restage under an open scan", failed ? "FAILED" : "ok");
	}
	if (!failed)
		failed |= run_phase("pooled scan across a restage", backend_pooled_restage, 1, NULL, 120.0);
	if (!failed)
		failed |= run_phase("a staging of several seconds", backend_slow_staging, 1, NULL, 180.0);
	if (!failed)
		failed |= run_phase("DROP INDEX x 70", backend_drop_index, 1, NULL, 300.0);
	if (!failed)
		failed |= run_phase("build state for 2000-d rows", backend_wide_build, 1, NULL, 120.0);
	if (!failed)
		failed |= run_phase("k-means over the devices of the node", backend_multi_device_kmeans, 1, NULL, 120.0);
	if (!failed)
	{
		void	   *pooled[1] = {(void *) 1};

		/* the worker dies without running any exit hook */
		if (shim_live_bgworkers() != 1)
		{
			fprintf(stderr, "expected ONE background worker, found %d\n", shim_live_bgworkers());
			failed = 1;
		}
		shim_kill_bgworkers();
		failed |= run_phase("worker killed (SIGKILL), pooled client", backend_after_worker_loss, 1, pooled, 120.0);
	}
	if (!failed)
	{
		void	   *own[1] = {(void *) 0};

		/* the worker is ended politely (SIGTERM): its exit hook deregisters it */
		shim_postmaster_shutdown();
		failed |= run_phase("worker ended (SIGTERM), own-context client", backend_after_worker_loss, 1, own, 120.0);
	}
	if (!failed)
		failed |= run_phase("hnsw: pages from the oracle's graph", backend_hnsw_build, 1, NULL, 120.0);
	if (!failed)
		failed |= run_phase("hnsw scans", backend_hnsw_scan, 1, NULL, 120.0);
#ifdef PGV_HAVE_REF_HNSW
	if (!failed)
		failed |= run_phase("the reference's own hnswgettuple", backend_reference_hnsw_scan, 1, NULL, 300.0);
#endif
	if (!failed)
		failed |= run_phase("hnsw: CREATE INDEX through the build hooks", backend_hnsw_gpu_build, 1, NULL, 300.0);
	if (!failed)
		failed |= run_phase("vector_ip_ops: build + scans", backend_ip_opclass, 1, NULL, 300.0);
#ifdef PGV_HAVE_REF_IVFUTILS
	if (!failed)
		failed |= run_phase("the reference's own IvfflatKmeans", backend_reference_kmeans, 1, NULL, 300.0);
#endif
#ifdef PGV_HAVE_REF_IVFINSERT
	if (!failed)
		failed |= run_phase("the reference's own ivfflatinsert", backend_reference_insert, 1, NULL, 300.0);
#endif
#ifdef PGV_HAVE_REF_HNSWINSERT
	if (!failed)
	{
		void	   *rel[1] = {(void *) (uintptr_t) REL_HINS};

		failed |= run_phase("hnsw: the oracle's graph into a second relation", backend_hnsw_build, 1, rel, 120.0);
	}
	if (!failed)
		failed |= run_phase("the reference's own hnswinsert", backend_reference_hnsw_insert, 1, NULL, 300.0);
#endif
#ifdef PGV_HAVE_REF_IVFBUILD
	if (!failed)
		failed |= run_phase("the reference's own ivfflatbuild", backend_reference_ivfbuild, 1, NULL, 300.0);
#endif
#ifdef PGV_HAVE_REF_HNSWBUILD
	if (!failed)
		failed |= run_phase("the reference's own hnswbuild", backend_reference_hnswbuild, 1, NULL, 300.0);
#endif
#if defined(PGV_HAVE_REF_HALFVEC) && defined(PGV_HAVE_REF_IVFBUILD) && defined(PGV_HAVE_REF_HNSWBUILD)
	if (!failed)
		failed |= run_phase("the reference's own halfvec opclasses", backend_reference_halfvec, 1, NULL, 300.0);
#endif
#if defined(PGV_HAVE_REF_IVFBUILD) && defined(PGV_HAVE_REF_HNSWBUILD)
	if (!failed)
		failed |= run_phase("the reference's own parallel CREATE INDEX", backend_reference_parallel_build, 1, NULL, 300.0);
#endif
#if defined(PGV_HAVE_REF_IVFBUILD) && defined(PGV_HAVE_REF_HNSWBUILD) && defined(PGV_HAVE_REF_HALFVEC)
	if (!failed && mock_hip_set_arena)
		failed |= run_phase("the product's page writer against the reference's pages", backend_writer_against_reference, 1, NULL, 120.0);
#endif
	if (!failed && mock_hip_set_arena)
		failed |= run_phase("a backend without a device", backend_no_device, 1, NULL, 120.0);
	shim_postmaster_shutdown();
	if (failed)
	{
		fprintf(stderr, "EXT-RUNTIME FAILED\n");
		return 1;
	}
	printf("EXT-RUNTIME OK\n");
	return 0;
}
