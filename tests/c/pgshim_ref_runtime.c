/*
 * pgshim_ref_runtime.c -- NOT PostgreSQL.  The server functions the REFERENCE'S OWN src/ivfscan.c (with
 * ext/pgvector-0.8.6-gpu.patch applied) and src/vector.c reach when they RUN: tuplesort over (float8, tid) slots, virtual
 * tuple slots, the pairing heap, fmgr calls, scan descriptors, child memory contexts.  Together with
 * tests/c/pgshim_runtime.c (palloc, ereport, the buffer manager over the emulated 8 KB pages, relations, opclasses) this
 * is enough for the reference's ivfflatbeginscan / ivfflatrescan / ivfflatgettuple / ivfflatendscan to execute unmodified:
 * tests/c/ext_driver.c (phase "the reference's own ivfflatgettuple", -DPGV_HAVE_REF_IVFSCAN) runs them with vector.gpu off
 * -- GetScanLists / GetScanItems / the sort of the REFERENCE against the oracle over the same pages -- and with vector.gpu
 * on, where the hook lines inside them hand the scan to ext/ivfscan_gpu.c.
 *
 * Compiled only where the reference tree is mounted (tests/test_ext_runtime_cpu.py), against the patched reference's own
 * ivfflat.h / hnsw.h -- and so are pgshim_runtime.c, ext_driver.c and ext/ in that build: one set of struct layouts.
 * TEST INFRASTRUCTURE ONLY.
 */
#define _GNU_SOURCE				/* RTLD_DEFAULT */
#include "pgshim_runtime.h"
#include "pgshim_ref.h"

#include "ivfflat.h"

#include <dlfcn.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

/* ------------------------------------------------------------------------------------------------ fmgr */
struct FunctionCallInfoBaseData
{
	Datum		args[4];
	int			nargs;
};

Datum
pgshim_getarg(FunctionCallInfo fcinfo, int n)
{
	return fcinfo->args[n];
}

bool
pgshim_argisnull(FunctionCallInfo fcinfo, int n)
{
	(void) fcinfo;
	(void) n;
	return false;
}

Datum
FunctionCall1Coll(FmgrInfo *flinfo, Oid collation, Datum arg1)
{
	struct FunctionCallInfoBaseData fc = {{arg1, 0, 0, 0}, 1};

	(void) collation;
	return flinfo->fn_addr(&fc);
}

Datum
FunctionCall2Coll(FmgrInfo *flinfo, Oid collation, Datum arg1, Datum arg2)
{
	struct FunctionCallInfoBaseData fc = {{arg1, arg2, 0, 0}, 2};

	(void) collation;
	return flinfo->fn_addr(&fc);
}

Datum
DirectFunctionCall1Coll(PGFunction func, Oid collation, Datum arg1)
{
	struct FunctionCallInfoBaseData fc = {{arg1, 0, 0, 0}, 1};

	(void) collation;
	return func(&fc);
}

Datum
DirectFunctionCall2Coll(PGFunction func, Oid collation, Datum arg1, Datum arg2)
{
	struct FunctionCallInfoBaseData fc = {{arg1, arg2, 0, 0}, 2};

	(void) collation;
	return func(&fc);
}

float8
DatumGetFloat8(Datum x)
{
	double		d;

	memcpy(&d, &x, sizeof(d));
	return d;
}

Datum
Float4GetDatum(float4 x)
{
	Datum		d = 0;

	memcpy(&d, &x, sizeof(x));
	return d;
}

float4
DatumGetFloat4(Datum x)
{
	float		f;

	memcpy(&f, &x, sizeof(f));
	return f;
}

int
pgshim_errcode(int sqlstate)
{
	(void) sqlstate;
	return 0;
}

void
float_overflow_error(void)
{
	elog(ERROR, "value out of range: overflow");
	abort();
}

void
float_underflow_error(void)
{
	elog(ERROR, "value out of range: underflow");
	abort();
}

Size
VARSIZE_ANY(const void *p)
{
	return (Size) (*(const uint32 *) p >> 2);	/* 4-byte headers only (SET_VARSIZE of pgshim_ref.h) */
}

/* ------------------------------------------------------------------------------------------------ memory */
Size
add_size(Size s1, Size s2)
{
	return s1 + s2;
}

Size
mul_size(Size s1, Size s2)
{
	return s1 * s2;
}

MemoryContext
AllocSetContextCreate(MemoryContext parent, const char *name, Size minContextSize, Size initBlockSize, Size maxBlockSize)
{
	(void) parent;
	(void) name;
	(void) minContextSize;
	(void) initBlockSize;
	(void) maxBlockSize;
	return shim_context_create();
}

void
MemoryContextDelete(MemoryContext context)
{
	shim_context_delete(context);
}

/* ------------------------------------------------------------------------------------------------ tuple descriptors, slots */
/* (struct TupleDescData: pgshim_runtime.h -- the relations' descriptors of pgshim_runtime.c are the same thing) */
TupleDesc
CreateTemplateTupleDesc(int natts)
{
	TupleDesc	d = palloc0(sizeof(struct TupleDescData));

	d->natts = natts;
	return d;
}

void
TupleDescInitEntry(TupleDesc desc, AttrNumber attributeNumber, const char *attributeName, Oid oidtypeid, int32 typmod, int attdim)
{
	(void) attributeName;
	(void) attdim;
	desc->types[attributeNumber - 1] = oidtypeid;
	desc->typmods[attributeNumber - 1] = typmod;
}

/* access/tupdesc.h: pg_attribute row i of a descriptor, as far as the build's InitBuildState reads it (type, typmod) */
Form_pg_attribute
TupleDescAttr(TupleDesc tupdesc, int i)
{
	static FormData_pg_attribute att[4];

	att[i].atttypid = tupdesc->types[i];
	att[i].atttypmod = tupdesc->typmods[i];
	return &att[i];
}

struct TupleTableSlotOps
{
	int			kind;
};
const TupleTableSlotOps TTSOpsVirtual = {1};
const TupleTableSlotOps TTSOpsMinimalTuple = {2};

TupleTableSlot *
MakeSingleTupleTableSlot(TupleDesc tupdesc, const TupleTableSlotOps *tts_ops)
{
	TupleTableSlot *slot = palloc0(sizeof(TupleTableSlot));

	(void) tts_ops;
	slot->tts_values = palloc0(sizeof(Datum) * 4);
	slot->tts_isnull = palloc0(sizeof(bool) * 4);
	(void) tupdesc;
	return slot;
}

TupleTableSlot *
ExecClearTuple(TupleTableSlot *slot)
{
	return slot;
}

TupleTableSlot *
ExecStoreVirtualTuple(TupleTableSlot *slot)
{
	return slot;
}

Datum
slot_getattr(TupleTableSlot *slot, int attnum, bool *isnull)
{
	*isnull = slot->tts_isnull[attnum - 1];
	return slot->tts_values[attnum - 1];
}

/* ------------------------------------------------------------------------------------------------ tuplesort
 * Two shapes.  The scan's sort (InitScanSortState, src/ivfscan.c:238-247): attribute 1 a float8 key ascending, attribute 2
 * a TID by reference.  The build's sort (InitBuildSortState, src/ivfbuild.c:340-352): attribute 1 the int4 list number
 * ascending, attribute 2 the heap TID, attribute 3 the vector (a varlena by reference).  "Input data is always copied":
 * TID and vector are copied into the sort's own memory context (the one current at tuplesort_begin_heap -- the callers
 * put tuples from inside per-row contexts they reset).  Equal keys keep their input order here (the server leaves it
 * unspecified; for the build that is heap order inside a list, what a serial CREATE INDEX produces in practice). */
typedef struct SortEntry
{
	double		key;
	ItemPointerData tid;
	void	   *value;			/* build shape: the copied varlena */
	int64		seq;
}			SortEntry;

#define SHIM_SORT_RUNS 8
struct Sharedsort
{
	uint32		nruns;
	struct
	{
		char	   *data;		/* packed entries: int32 list, ItemPointerData tid, uint32 value size, value bytes (MAXALIGNed) */
		int64		n;
	}			runs[SHIM_SORT_RUNS];
};

struct Tuplesortstate
{
	MemoryContext ctx;
	bool		build_shape;
	bool		parallel;		/* a participant's or the leader's sort of a parallel build */
	SortCoordinateData coordinate;
	SortEntry  *e;
	int64		n,
				cap,
				pos;
};

Tuplesortstate *
tuplesort_begin_heap(TupleDesc tupDesc, int nkeys, AttrNumber *attNums, Oid *sortOperators, Oid *sortCollations,
					 bool *nullsFirstFlags, int workMem, SortCoordinate coordinate, int sortopt)
{
	Tuplesortstate *st = palloc0(sizeof(Tuplesortstate));

	(void) sortCollations;
	(void) nullsFirstFlags;
	(void) workMem;
	(void) sortopt;
	if (coordinate != NULL)
	{
		st->parallel = true;
		st->coordinate = *coordinate;
	}
	if (nkeys != 1 || attNums[0] != 1 || (sortOperators[0] != Float8LessOperator && sortOperators[0] != Int4LessOperator))
		elog(ERROR, "stand-in tuplesort: float8 or int4 ascending on attribute 1 only");
	st->build_shape = sortOperators[0] == Int4LessOperator;
	if (st->build_shape && (tupDesc->natts != 3 || tupDesc->types[0] != INT4OID || tupDesc->types[1] != TIDOID))
		elog(ERROR, "stand-in tuplesort: the build's sort is (int4 list, tid, vector)");
	st->ctx = CurrentMemoryContext;
	return st;
}

void
tuplesort_puttupleslot(Tuplesortstate *state, TupleTableSlot *slot)
{
	MemoryContext old = MemoryContextSwitchTo(state->ctx);

	if (state->n == state->cap)
	{
		state->cap = state->cap ? state->cap * 2 : 1024;
		state->e = state->e ? repalloc_huge(state->e, sizeof(SortEntry) * (Size) state->cap)
			: palloc_extended(sizeof(SortEntry) * (Size) state->cap, MCXT_ALLOC_HUGE);
	}
	state->e[state->n].tid = *(ItemPointer) DatumGetPointer(slot->tts_values[1]);
	state->e[state->n].seq = state->n;
	if (state->build_shape)
	{
		const void *v = DatumGetPointer(slot->tts_values[2]);
		Size		size = VARSIZE_ANY(v);

		state->e[state->n].key = (double) DatumGetInt32(slot->tts_values[0]);
		state->e[state->n].value = palloc(size);
		memcpy(state->e[state->n].value, v, size);
	}
	else
	{
		state->e[state->n].key = DatumGetFloat8(slot->tts_values[0]);
		state->e[state->n].value = NULL;
	}
	state->n++;
	MemoryContextSwitchTo(old);
}

static int
sort_entry_cmp(const void *a, const void *b)
{
	const SortEntry *x = a,
			   *y = b;

	/* float8 ordering: NaN after everything */
	if (isnan(x->key) || isnan(y->key))
	{
		if (isnan(x->key) != isnan(y->key))
			return isnan(x->key) ? 1 : -1;
	}
	else if (x->key != y->key)
		return x->key < y->key ? -1 : 1;
	return x->seq < y->seq ? -1 : (x->seq > y->seq ? 1 : 0);
}

static int
sort_entry_cmp_tid(const void *a, const void *b)
{
	const SortEntry *x = a,
			   *y = b;
	uint64		tx = ((uint64) x->tid.ip_blkid.bi_hi << 32) | ((uint64) x->tid.ip_blkid.bi_lo << 16) | x->tid.ip_posid,
				ty = ((uint64) y->tid.ip_blkid.bi_hi << 32) | ((uint64) y->tid.ip_blkid.bi_lo << 16) | y->tid.ip_posid;

	if (x->key != y->key)
		return x->key < y->key ? -1 : 1;
	return tx < ty ? -1 : (tx > ty ? 1 : 0);
}

/* a run in shared memory: per tuple int32 list, ItemPointerData, uint32 size of the value, the value; 8-byte aligned */
#define RUN_ENTRY_BYTES(vsize) (((Size) 16 + (vsize) + 7) & ~(Size) 7)

void
tuplesort_performsort(Tuplesortstate *state)
{
	if (state->parallel && !state->coordinate.isWorker)
	{
		/* the leader: nothing was put into THIS sort (the leader's share went through a participant's sort of its own);
		 * its result is every participant's run merged */
		Sharedsort *shared = state->coordinate.sharedsort;
		MemoryContext old = MemoryContextSwitchTo(state->ctx);
		int64		total = 0,
					at = 0;

		if (state->n != 0 || (int) shared->nruns != state->coordinate.nParticipants)
			elog(ERROR, "stand-in tuplesort: %d runs published, %d participants", (int) shared->nruns, state->coordinate.nParticipants);
		for (uint32 r = 0; r < shared->nruns; r++)
			total += shared->runs[r].n;
		state->cap = total > 0 ? total : 1;
		state->e = palloc_extended(sizeof(SortEntry) * (Size) state->cap, MCXT_ALLOC_HUGE);
		for (uint32 r = 0; r < shared->nruns; r++)
		{
			const char *p = shared->runs[r].data;

			for (int64 i = 0; i < shared->runs[r].n; i++)
			{
				int32		list;
				uint32		vsize;

				memcpy(&list, p, 4);
				memcpy(&state->e[at].tid, p + 4, sizeof(ItemPointerData));
				memcpy(&vsize, p + 12, 4);
				state->e[at].key = (double) list;
				state->e[at].value = palloc(vsize);
				memcpy(state->e[at].value, p + 16, vsize);
				p += RUN_ENTRY_BYTES(vsize);
				at++;
			}
		}
		state->n = total;
		MemoryContextSwitchTo(old);
		if (state->n > 1)
			qsort(state->e, (size_t) state->n, sizeof(SortEntry), sort_entry_cmp_tid);
		for (int64 i = 0; i < state->n; i++)
			state->e[i].seq = i;
		state->pos = 0;
		return;
	}
	if (state->n > 1)
		qsort(state->e, (size_t) state->n, sizeof(SortEntry), sort_entry_cmp);
	state->pos = 0;
	if (state->parallel)
	{
		/* a participant: the sorted run goes where the leader finds it */
		Sharedsort *shared = state->coordinate.sharedsort;
		Size		bytes = 8;
		char	   *data,
				   *p;
		uint32		slot;

		if (!state->build_shape)
			elog(ERROR, "stand-in tuplesort: parallel sorts are the build's");
		for (int64 i = 0; i < state->n; i++)
			bytes += RUN_ENTRY_BYTES(VARSIZE_ANY(state->e[i].value));
		data = p = shim_shared_alloc(bytes);
		for (int64 i = 0; i < state->n; i++)
		{
			int32		list = (int32) state->e[i].key;
			uint32		vsize = (uint32) VARSIZE_ANY(state->e[i].value);

			memcpy(p, &list, 4);
			memcpy(p + 4, &state->e[i].tid, sizeof(ItemPointerData));
			memcpy(p + 12, &vsize, 4);
			memcpy(p + 16, state->e[i].value, vsize);
			p += RUN_ENTRY_BYTES(vsize);
		}
		slot = __atomic_fetch_add(&shared->nruns, 1, __ATOMIC_SEQ_CST);
		if (slot >= SHIM_SORT_RUNS)
			elog(ERROR, "stand-in tuplesort: too many participants");
		shared->runs[slot].data = data;
		shared->runs[slot].n = state->n;
		__atomic_thread_fence(__ATOMIC_RELEASE);
	}
}

bool
tuplesort_gettupleslot(Tuplesortstate *state, bool forward, bool copy, TupleTableSlot *slot, Datum *abbrev)
{
	(void) forward;
	(void) copy;
	(void) abbrev;
	if (state->pos >= state->n)
		return false;
	slot->tts_values[0] = state->build_shape ? Int32GetDatum((int32) state->e[state->pos].key) : Float8GetDatum(state->e[state->pos].key);
	slot->tts_isnull[0] = false;
	slot->tts_values[1] = PointerGetDatum(&state->e[state->pos].tid);
	slot->tts_isnull[1] = false;
	if (state->build_shape)
	{
		slot->tts_values[2] = PointerGetDatum(state->e[state->pos].value);
		slot->tts_isnull[2] = false;
	}
	state->pos++;
	return true;
}

/* test access: the tuples in the order they were put (seq), wherever the sort has moved them */
int64_t
shim_tuplesort_inputs(Tuplesortstate *state)
{
	return state->n;
}

void
shim_tuplesort_input(Tuplesortstate *state, int64_t i, int32_t *list, ItemPointerData *tid, const void **value)
{
	const SortEntry *e = NULL;

	if (i >= 0 && i < state->n && state->e[i].seq == i)
		e = &state->e[i];		/* not sorted yet (or left in place) */
	for (int64 j = 0; e == NULL && j < state->n; j++)
		if (state->e[j].seq == i)
			e = &state->e[j];
	if (e == NULL)
		elog(ERROR, "stand-in tuplesort: no input tuple " INT64_FORMAT, (int64) i);
	*list = (int32_t) e->key;
	*tid = e->tid;
	*value = e->value;
}

void
tuplesort_reset(Tuplesortstate *state)
{
	for (int64 i = 0; i < state->n; i++)
		if (state->e[i].value)
			pfree(state->e[i].value);
	state->n = 0;
	state->pos = 0;
}

void
tuplesort_end(Tuplesortstate *state)
{
	tuplesort_reset(state);
	if (state->e)
		pfree(state->e);
	state->e = NULL;
	state->n = state->cap = state->pos = 0;
}

/* ------------------------------------------------------------------------------------------------ lib/pairingheap.c
 * a max-heap by the caller's comparator (CompareLists of src/ivfscan.c:32-41 puts the LARGEST distance first) */
pairingheap *
pairingheap_allocate(pairingheap_comparator compare, void *arg)
{
	pairingheap *h = palloc0(sizeof(pairingheap));

	h->ph_compare = compare;
	h->ph_arg = arg;
	return h;
}

static pairingheap_node *
ph_merge(pairingheap *h, pairingheap_node *a, pairingheap_node *b)
{
	if (a == NULL)
		return b;
	if (b == NULL)
		return a;
	if (h->ph_compare(a, b, h->ph_arg) < 0)
	{
		pairingheap_node *t = a;

		a = b;
		b = t;
	}
	/* b becomes the first child of a */
	b->next_sibling = a->first_child;
	a->first_child = b;
	return a;
}

void
pairingheap_add(pairingheap *heap, pairingheap_node *node)
{
	node->first_child = node->next_sibling = node->prev_or_parent = NULL;
	heap->ph_root = ph_merge(heap, heap->ph_root, node);
}

pairingheap_node *
pairingheap_first(pairingheap *heap)
{
	return heap->ph_root;
}

pairingheap_node *
pairingheap_remove_first(pairingheap *heap)
{
	pairingheap_node *top = heap->ph_root,
			   *list,
			   *acc = NULL;

	if (top == NULL)
		return NULL;
	/* pair the children up left to right, then fold the pairs right to left */
	list = top->first_child;
	{
		pairingheap_node *pairs = NULL;

		while (list)
		{
			pairingheap_node *a = list,
					   *b = a->next_sibling;

			list = b ? b->next_sibling : NULL;
			a->next_sibling = NULL;
			if (b)
				b->next_sibling = NULL;
			a = ph_merge(heap, a, b);
			a->next_sibling = pairs;	/* (a stack of the merged pairs) */
			pairs = a;
		}
		while (pairs)
		{
			pairingheap_node *next = pairs->next_sibling;

			pairs->next_sibling = NULL;
			acc = ph_merge(heap, acc, pairs);
			pairs = next;
		}
	}
	heap->ph_root = acc;
	top->first_child = top->next_sibling = NULL;
	return top;
}

/* ------------------------------------------------------------------------------------------------ scans, buffers, stats */
static struct SnapshotDataStandIn
{
	int			mvcc;
}			the_snapshot = {1};

IndexScanDesc
RelationGetIndexScan(Relation indexRelation, int nkeys, int norderbys)
{
	IndexScanDesc scan = palloc0(sizeof(IndexScanDescData));

	scan->indexRelation = indexRelation;
	scan->xs_snapshot = (struct SnapshotData *) &the_snapshot;
	scan->numberOfKeys = nkeys;
	scan->numberOfOrderBys = norderbys;
	scan->keyData = nkeys > 0 ? palloc0(sizeof(ScanKeyData) * (Size) nkeys) : NULL;
	scan->orderByData = norderbys > 0 ? palloc0(sizeof(ScanKeyData) * (Size) norderbys) : NULL;
	return scan;
}

bool
IsMVCCSnapshot(Snapshot snapshot)
{
	return snapshot != NULL;
}

Buffer
ReadBuffer(Relation reln, BlockNumber blockNum)
{
	return ReadBufferExtended(reln, MAIN_FORKNUM, blockNum, RBM_NORMAL, NULL);
}

BufferAccessStrategy
GetAccessStrategy(BufferAccessStrategyType btype)
{
	(void) btype;
	return NULL;
}

void
FreeAccessStrategy(BufferAccessStrategy strategy)
{
	(void) strategy;
}

static int64 index_scans_counted;

void
pgstat_count_index_scan(Relation rel)
{
	(void) rel;
	index_scans_counted++;
}

int64
shim_index_scans_counted(void)
{
	return index_scans_counted;
}

/* ------------------------------------------------------------------------------------------------ the rest of pgvector
 * globals of src/ivfflat.c and the one function of src/ivfutils.c the scan calls that pgshim_runtime.c does not have */
int			work_mem = 4096;
int			ivfflat_probes = 1;
int			ivfflat_iterative_scan = 0;
int			ivfflat_max_probes = 32768;

#ifndef PGV_HAVE_REF_IVFUTILS
/* src/ivfutils.c:72-80 */
Datum
IvfflatNormValue(const IvfflatTypeInfo * typeInfo, Oid collation, Datum value)
{
	return DirectFunctionCall1Coll(typeInfo->normalize, collation, value);
}
#else
/* with the reference's src/ivfutils.c + src/ivfkmeans.c linked in (-DPGV_HAVE_REF_IVFUTILS) */
int			maintenance_work_mem = 65536;

Size
VARSIZE(const void *p)
{
	return VARSIZE_ANY(p);
}
#endif

/* src/vector.c's _PG_init names them; nothing here calls _PG_init */
void
BitvecInit(void)
{
}

#ifndef PGV_HAVE_REF_HALFVEC		/* (the reference's src/halfutils.c has the real one: the program's main calls it) */
void
HalfvecInit(void)
{
}
#endif

void
HnswInit(void)
{
}

void
IvfflatInit(void)
{
}

#ifdef PGV_HAVE_REF_HNSW
/* ------------------------------------------------------------------------------------------------ for src/hnswscan.c + src/hnswutils.c
 * (-DPGV_HAVE_REF_HNSW: the reference's own hnswbeginscan / hnswrescan / hnswgettuple / hnswendscan and the whole of its
 * src/hnswutils.c -- HnswSearchLayer, HnswLoadElement, the visited tables -- are linked into the program.)  What they reach
 * of the server and pgshim_runtime.c does not have: the pg_list calls over the runtime's List, page locks (nothing vacuums
 * here), the two murmur finalizers simplehash is keyed with, datumCopy of a varlena, context reset / accounting. */
int			hnsw_max_scan_tuples = 20000;
double		hnsw_scan_mem_multiplier = 1;

Datum
FunctionCall0Coll(FmgrInfo *flinfo, Oid collation)
{
	struct FunctionCallInfoBaseData fc = {{0, 0, 0, 0}, 0};

	(void) collation;
	return flinfo->fn_addr(&fc);
}

int
list_length(const List *l)
{
	return l ? l->length : 0;
}

void *
linitial(const List *l)
{
	return l->elems[0];
}

void *
llast(const List *l)
{
	return l->elems[l->length - 1];
}

List *
list_delete_last(List *list)
{
	if (list == NIL || list->length <= 1)
		return NIL;				/* (the cells go with their memory context) */
	list->length--;
	return list;
}

List *
list_copy(const List *list)
{
	List	   *copy = NIL;

	for (int i = 0; i < list_length(list); i++)
		copy = lappend(copy, list->elems[i]);
	return copy;
}

/* cells are the pointers themselves: a ListCell is one void * wide */
void
list_sort(List *list, int (*cmp) (const ListCell *a, const ListCell *b))
{
	_Static_assert(sizeof(ListCell) == sizeof(void *), "ListCell");
	if (list_length(list) > 1)
		qsort(list->elems, (size_t) list->length, sizeof(void *), (int (*) (const void *, const void *)) cmp);
}

ListCell *
list_head(const List *l)
{
	return l && l->length > 0 ? (ListCell *) &l->elems[0] : NULL;
}

ListCell *
lnext(const List *l, const ListCell *c)
{
	void	  **next = (void **) c + 1;

	return next < l->elems + l->length ? (ListCell *) next : NULL;
}

void *
pgshim_lfirst(const ListCell *lc)
{
	return lc->ptr_value;
}

void
ItemPointerSet(ItemPointerData *pointer, BlockNumber blockNumber, OffsetNumber offNum)
{
	pointer->ip_blkid.bi_hi = (uint16) (blockNumber >> 16);
	pointer->ip_blkid.bi_lo = (uint16) (blockNumber & 0xffff);
	pointer->ip_posid = offNum;
}

void
ItemPointerSetInvalid(ItemPointerData *pointer)
{
	ItemPointerSet(pointer, InvalidBlockNumber, InvalidOffsetNumber);
}

bool
ItemPointerEquals(ItemPointer pointer1, ItemPointer pointer2)
{
	return pointer1->ip_blkid.bi_hi == pointer2->ip_blkid.bi_hi && pointer1->ip_blkid.bi_lo == pointer2->ip_blkid.bi_lo &&
		pointer1->ip_posid == pointer2->ip_posid;
}

/* HNSW_SCAN_LOCK keeps vacuum's second pass away from a walk in progress; no vacuum runs in this program */
void
LockPage(Relation relation, BlockNumber blkno, LOCKMODE lockmode)
{
	(void) relation;
	(void) blkno;
	(void) lockmode;
}

void
UnlockPage(Relation relation, BlockNumber blkno, LOCKMODE lockmode)
{
	(void) relation;
	(void) blkno;
	(void) lockmode;
}

Size
BufferGetPageSize(Buffer buffer)
{
	(void) buffer;
	return BLCKSZ;
}

float8
get_float8_infinity(void)
{
	return (float8) INFINITY;
}

/* common/hashfn.h: the 32- and 64-bit murmur3 finalizers */
uint32
murmurhash32(uint32 data)
{
	uint32		h = data;

	h ^= h >> 16;
	h *= 0x85ebca6b;
	h ^= h >> 13;
	h *= 0xc2b2ae35;
	h ^= h >> 16;
	return h;
}

uint64
murmurhash64(uint64 data)
{
	uint64		h = data;

	h ^= h >> 33;
	h *= 0xff51afd7ed558ccdull;
	h ^= h >> 33;
	h *= 0xc4ceb9fe1a85ec53ull;
	h ^= h >> 33;
	return h;
}

/* utils/datum.h: by-reference varlena values only (typByVal false, typLen -1: what hnswrescan copies the query with) */
Datum
datumCopy(Datum value, bool typByVal, int typLen)
{
	Size		size;
	void	   *copy;

	if (typByVal)
		return value;
	if (typLen != -1)
		elog(ERROR, "datumCopy: typLen %d", typLen);
	size = VARSIZE_ANY(DatumGetPointer(value));
	copy = palloc(size);
	memcpy(copy, DatumGetPointer(value), size);
	return PointerGetDatum(copy);
}

void
MemoryContextReset(MemoryContext context)
{
	shim_context_reset(context);
}

Size
MemoryContextMemAllocated(MemoryContext context, bool recurse)
{
	(void) recurse;
	return shim_context_bytes(context);
}
#endif							/* PGV_HAVE_REF_HNSW */

#ifdef PGV_HAVE_REF_IVFINSERT
/* ------------------------------------------------------------------------------------------------ for src/ivfinsert.c
 * (-DPGV_HAVE_REF_IVFINSERT: the reference's own ivfflatinsert -- FindInsertPage, InsertTuple -- and, reached through it,
 * the page-append half of its src/ivfutils.c WRITE into the emulated pages.)  storage/bufpage.c's PageInit / PageAddItem /
 * PageGetFreeSpace over the real page layout, index_form_tuple for the one varlena attribute these indexes have, and a
 * generic-xlog that hands out the page itself: nothing here crashes mid-record, so there is nothing to replay. */
struct GenericXLogState
{
	int			unused;
};
static struct GenericXLogState the_xlog_state;

GenericXLogState *
GenericXLogStart(Relation relation)
{
	(void) relation;
	return &the_xlog_state;
}

Page
GenericXLogRegisterBuffer(GenericXLogState *state, Buffer buffer, int flags)
{
	(void) state;
	(void) flags;
	return BufferGetPage(buffer);
}

XLogRecPtr
GenericXLogFinish(GenericXLogState *state)
{
	(void) state;
	return 0;
}

void
GenericXLogAbort(GenericXLogState *state)
{
	(void) state;				/* (the reference aborts only records that changed nothing: src/ivfutils.c:272-276) */
}

void
MarkBufferDirty(Buffer buffer)
{
	(void) buffer;
}

void
LockRelationForExtension(Relation relation, LOCKMODE lockmode)
{
	(void) relation;
	(void) lockmode;
}

void
UnlockRelationForExtension(Relation relation, LOCKMODE lockmode)
{
	(void) relation;
	(void) lockmode;
}

const char *
RelationGetRelationName(Relation relation)
{
	(void) relation;
	return "index";
}

/* storage/bufpage.c PageInit: an empty page with `specialSize` bytes of special space at its end */
void
PageInit(Page page, Size pageSize, Size specialSize)
{
	PageHeader	p = (PageHeader) page;

	specialSize = MAXALIGN(specialSize);
	memset(page, 0, pageSize);
	p->pd_lower = SizeOfPageHeaderData;
	p->pd_upper = (LocationIndex) (pageSize - specialSize);
	p->pd_special = (LocationIndex) (pageSize - specialSize);
	p->pd_pagesize_version = (uint16) (pageSize | 4);	/* PG_PAGE_LAYOUT_VERSION */
}

/* PageGetFreeSpace: what fits between the line pointers and the tuples, less one new line pointer */
Size
PageGetFreeSpace(Page page)
{
	const PageHeader p = (PageHeader) page;
	int			space = (int) p->pd_upper - (int) p->pd_lower;

	if (space < 4)
		return 0;
	return (Size) (space - 4);
}

/* PageAddItemExtended, the append case (offsetNumber invalid, no overwrite): the tuple below pd_upper, a line pointer
 * (lp_off:15 | lp_flags:2 = LP_NORMAL | lp_len:15) at pd_lower */
OffsetNumber
PageAddItem(Page page, Item item, Size size, OffsetNumber offsetNumber, bool overwrite, bool is_heap)
{
	PageHeader	p = (PageHeader) page;
	const OffsetNumber limit = (OffsetNumber) (PageGetMaxOffsetNumber(page) + 1);
	const int	lower = (int) p->pd_lower + 4;
	const int	upper = (int) p->pd_upper - (int) MAXALIGN(size);
	uint32		lp;

	(void) is_heap;
	if (overwrite || (offsetNumber != InvalidOffsetNumber && offsetNumber != limit))
		elog(ERROR, "pgshim: PageAddItem: only appends");
	if (lower > upper)
		return InvalidOffsetNumber;
	lp = (uint32) upper | (1u << 15) | ((uint32) size << 17);
	memcpy(page + SizeOfPageHeaderData + (size_t) (limit - 1) * 4, &lp, 4);
	memcpy(page + upper, item, size);
	p->pd_lower = (LocationIndex) lower;
	p->pd_upper = (LocationIndex) upper;
	return limit;
}

/* access/common/indextuple.c for one by-reference varlena attribute with a 4-byte header (a vector of more than 30
 * dimensions is never packed into a 1-byte header): header 8 bytes, the value behind it, t_info = size | INDEX_VAR_MASK */
IndexTuple
index_form_tuple(TupleDesc tupleDescriptor, const Datum *values, const bool *isnull)
{
	const void *v = DatumGetPointer(values[0]);
	const Size	vsize = VARSIZE_ANY(v);
	const Size	size = MAXALIGN(sizeof(IndexTupleData) + vsize);
	IndexTuple	tup;

	(void) tupleDescriptor;
	if (isnull[0] || vsize < 128)
		elog(ERROR, "pgshim: index_form_tuple: one non-null varlena of 128 bytes or more");
	tup = palloc0(size);
	memcpy((char *) tup + sizeof(IndexTupleData), v, vsize);
	tup->t_info = (unsigned short) (size | 0x4000);
	return tup;
}

Size
IndexTupleSize(IndexTuple itup)
{
	return (Size) (itup->t_info & 0x1FFF);
}
#endif							/* PGV_HAVE_REF_IVFINSERT */

#ifdef PGV_HAVE_REF_IVFVACUUM
/* ------------------------------------------------------------------------------------------------ for src/ivfvacuum.c
 * (the reference's own ivfflatbulkdelete / ivfflatvacuumcleanup over the emulated pages) */
void
vacuum_delay_point(void)
{
}

void
LockBufferForCleanup(Buffer buffer)
{
	LockBuffer(buffer, BUFFER_LOCK_EXCLUSIVE);	/* (nobody else pins the page in this program while a vacuum runs) */
}

/* storage/bufpage.c PageIndexMultiDelete: the listed items (ascending) go, the others keep their order -- line pointers
 * renumbered, tuples packed against the special space again */
void
PageIndexMultiDelete(Page page, OffsetNumber *itemnos, int nitems)
{
	PageHeader	p = (PageHeader) page;
	const int	nline = PageGetMaxOffsetNumber(page);
	char		old[BLCKSZ];
	int			next = 0,
				kept = 0,
				upper = p->pd_special;

	memcpy(old, page, BLCKSZ);
	for (int off = 1; off <= nline; off++)
	{
		uint32		lp;
		Size		size;

		if (next < nitems && itemnos[next] == off)
		{
			next++;
			continue;
		}
		memcpy(&lp, old + SizeOfPageHeaderData + (size_t) (off - 1) * 4, 4);
		size = lp >> 17;
		upper -= (int) MAXALIGN(size);
		memcpy(page + upper, old + (lp & 0x7FFFu), size);
		lp = (uint32) upper | (lp & (3u << 15)) | ((uint32) size << 17);
		memcpy(page + SizeOfPageHeaderData + (size_t) kept * 4, &lp, 4);
		kept++;
	}
	if (next != nitems)
		elog(ERROR, "pgshim: PageIndexMultiDelete: item numbers not ascending / out of range");
	p->pd_lower = (LocationIndex) (SizeOfPageHeaderData + 4 * kept);
	p->pd_upper = (LocationIndex) upper;
}
#endif							/* PGV_HAVE_REF_IVFVACUUM */

#ifdef PGV_HAVE_REF_HNSWINSERT
/* ------------------------------------------------------------------------------------------------ for src/hnswinsert.c
 * (the reference's own hnswinsert -> HnswInsertTupleOnDisk: the on-disk search of src/hnswutils.c, then element and
 * neighbor tuples added to / overwritten in the emulated pages) */
MemoryContext
GenerationContextCreate(MemoryContext parent, const char *name, Size minContextSize, Size initBlockSize, Size maxBlockSize)
{
	(void) parent;
	(void) name;
	(void) minContextSize;
	(void) initBlockSize;
	(void) maxBlockSize;
	return shim_context_create_generation();
}

Size
ItemIdGetLength(ItemId itemId)
{
	uint32		lp;

	memcpy(&lp, itemId, 4);
	return (Size) (lp >> 17);
}

Size
PageGetExactFreeSpace(Page page)
{
	const PageHeader p = (PageHeader) page;
	int			space = (int) p->pd_upper - (int) p->pd_lower;

	return space < 0 ? 0 : (Size) space;
}

/* storage/bufpage.c PageIndexTupleOverwrite: the tuple at offnum replaced by one of possibly another size; the tuples
 * between pd_upper and it slide, their line pointers follow */
bool
PageIndexTupleOverwrite(Page page, OffsetNumber offnum, Item newtup, Size newsize)
{
	PageHeader	p = (PageHeader) page;
	const int	nline = PageGetMaxOffsetNumber(page);
	uint32		lp;
	int			offset,
				oldsize,
				alignednew = (int) MAXALIGN(newsize),
				diff;

	if (offnum < 1 || offnum > nline)
		elog(ERROR, "pgshim: PageIndexTupleOverwrite: invalid index offnum %u", (unsigned) offnum);
	memcpy(&lp, page + SizeOfPageHeaderData + (size_t) (offnum - 1) * 4, 4);
	offset = (int) (lp & 0x7FFFu);
	oldsize = (int) MAXALIGN(lp >> 17);
	if (alignednew > oldsize + ((int) p->pd_upper - (int) p->pd_lower))
		return false;
	diff = oldsize - alignednew;
	if (diff != 0)
	{
		memmove(page + p->pd_upper + diff, page + p->pd_upper, (size_t) (offset - p->pd_upper));
		p->pd_upper = (LocationIndex) (p->pd_upper + diff);
		for (int i = 1; i <= nline; i++)
		{
			uint32		o;

			memcpy(&o, page + SizeOfPageHeaderData + (size_t) (i - 1) * 4, 4);
			if ((o >> 15 & 3u) != 0 && (int) (o & 0x7FFFu) <= offset)
			{
				o = (o & ~0x7FFFu) | (uint32) ((int) (o & 0x7FFFu) + diff);
				memcpy(page + SizeOfPageHeaderData + (size_t) (i - 1) * 4, &o, 4);
			}
		}
		memcpy(&lp, page + SizeOfPageHeaderData + (size_t) (offnum - 1) * 4, 4);	/* (moved with the others) */
	}
	lp = (lp & 0x1FFFFu) | ((uint32) newsize << 17);
	memcpy(page + SizeOfPageHeaderData + (size_t) (offnum - 1) * 4, &lp, 4);
	memcpy(page + (lp & 0x7FFFu), newtup, newsize);
	return true;
}

/* utils/datum.c: by-reference varlena values (typLen -1) equal byte for byte */
bool
datumIsEqual(Datum value1, Datum value2, bool typByVal, int typLen)
{
	Size		s1,
				s2;

	if (typByVal)
		return value1 == value2;
	if (typLen != -1)
		elog(ERROR, "datumIsEqual: typLen %d", typLen);
	s1 = VARSIZE_ANY(DatumGetPointer(value1));
	s2 = VARSIZE_ANY(DatumGetPointer(value2));
	return s1 == s2 && memcmp(DatumGetPointer(value1), DatumGetPointer(value2), s1) == 0;
}
#endif							/* PGV_HAVE_REF_HNSWINSERT */

#if defined(PGV_HAVE_REF_IVFBUILD) || defined(PGV_HAVE_REF_HNSWBUILD)
/* ------------------------------------------------------------------------------------------------ for src/ivfbuild.c + src/hnswbuild.c
 * (the reference's own ivfflatbuild / hnswbuild: CREATE INDEX from the first sample to the last page.)  What a SERIAL
 * build reaches of the server and the files above do not have: the table it scans (a stand-in heap the test defines:
 * rows, NULLs, toasted values, TIDs), ANALYZE's block and reservoir samplers, the progress counters, spinlocks and lock
 * initialisation.  plan_create_index_workers answers 0 -- no parallel workers in this program --, so everything behind
 * IvfflatBeginParallel / HnswBeginParallel only has to LINK: those stand-ins raise an ERROR if they are ever reached. */
struct ParallelTableScanDescData
{
	uint32		next_block;
	uint32		attached;		/* participants that have begun their scan */
	uint32		mode;			/* 0 undecided, 1 blocks dealt round-robin, 2 first come first served */
	uint32		participants;
};

struct TableScanDescData
{
	struct ParallelTableScanDescData *pscan;
	uint32		my_index,
				my_next;
};

/* the definition sits in shared memory (slot 0 of the runtime's shared words) so that parallel workers -- forked from the
 * postmaster, not from the backend that runs CREATE INDEX -- find it: table_open in a worker attaches to it.  The rows and
 * the callback's argument must be visible there too (shim_shared_alloc; the callback is code, the same in every process). */
static ShimHeapDef *heap_def_p = NULL;
#define heap_def (*heap_def_p)
static struct RelationData heap_rel;
static struct TupleDescData heap_desc;
extern BlockNumber (*shim_heap_blocks_hook) (void);

static BlockNumber
heap_blocks(void)
{
	return (BlockNumber) ((heap_def.nrows + heap_def.rows_per_block - 1) / heap_def.rows_per_block);
}

Relation
shim_heap_attach(void)
{
	heap_def_p = *(ShimHeapDef **) shim_shared_slot(0);
	if (heap_def_p == NULL)
		return NULL;
	memset(&heap_rel, 0, sizeof(heap_rel));
	heap_rel.rd_id = SHIM_HEAP_OID;
	heap_desc.natts = 1;
	heap_desc.types[0] = SHIM_VECTOR_TYPE_OID;
	heap_desc.typmods[0] = -1;
	heap_rel.rd_att = &heap_desc;
	shim_heap_blocks_hook = heap_blocks;
	return &heap_rel;
}

Relation
shim_heap_relation(const ShimHeapDef * def)
{
	ShimHeapDef *shared = shim_shared_alloc(sizeof(ShimHeapDef));

	*shared = *def;
	__atomic_store_n((ShimHeapDef **) shim_shared_slot(0), shared, __ATOMIC_RELEASE);
	return shim_heap_attach();
}

/* heapam_index_build_range_scan (access/heap/heapam_handler.c) as far as an index build sees it: blocks [start, start +
 * numblocks) in physical order, every live row handed to the callback -- NULLs too, the access method skips them --, the
 * column value living in a per-tuple context that is reset before the next row; the return value is the number of live
 * rows scanned */
double
table_index_build_range_scan(Relation table_rel, Relation index_rel, IndexInfo *index_info, bool allow_sync, bool anyvisible,
							 bool progress, BlockNumber start_blockno, BlockNumber numblocks, IndexBuildCallback callback,
							 void *callback_state, TableScanDesc scan)
{
	MemoryContext per_tuple;
	int64		first = (int64) start_blockno * heap_def.rows_per_block;
	int64		end = numblocks == InvalidBlockNumber ? heap_def.nrows : (int64) (start_blockno + numblocks) * heap_def.rows_per_block;
	double		reltuples = 0;

	(void) index_info;
	(void) allow_sync;
	(void) anyvisible;
	(void) progress;
	if (table_rel != &heap_rel)
		elog(ERROR, "stand-in heap: one table");
	if (end > heap_def.nrows)
		end = heap_def.nrows;
	per_tuple = AllocSetContextCreate(CurrentMemoryContext, "index build per-tuple", 0, 0, 0);
	for (;;)
	{
		if (scan != NULL)
		{
			/* a participant of a parallel scan: the next block nobody has taken yet (table_block_parallelscan_nextpage) */
			BlockNumber blk;

			if (__atomic_load_n(&scan->pscan->mode, __ATOMIC_ACQUIRE) == 1 && scan->my_index < scan->pscan->participants)
			{
				blk = scan->my_next;
				scan->my_next += scan->pscan->participants;
			}
			else
				blk = __atomic_fetch_add(&scan->pscan->next_block, 1, __ATOMIC_SEQ_CST);
			if (blk >= heap_blocks())
				break;
			first = (int64) blk * heap_def.rows_per_block;
			end = first + heap_def.rows_per_block < heap_def.nrows ? first + heap_def.rows_per_block : heap_def.nrows;
		}
		for (int64 row = first; row < end; row++)
		{
			MemoryContext old = MemoryContextSwitchTo(per_tuple);
			Datum		values[1];
			bool		isnull[1];
			ItemPointerData tid;

			CHECK_FOR_INTERRUPTS();
			heap_def.fetch(row, &values[0], &isnull[0], &tid, heap_def.arg);
			MemoryContextSwitchTo(old);
			callback(index_rel, &tid, values, isnull, true, callback_state);
			reltuples += 1;
			MemoryContextReset(per_tuple);
		}
		if (scan == NULL)
			break;
	}
	MemoryContextDelete(per_tuple);
	return reltuples;
}

double
table_index_build_scan(Relation table_rel, Relation index_rel, IndexInfo *index_info, bool allow_sync, bool progress,
					   IndexBuildCallback callback, void *callback_state, TableScanDesc scan)
{
	return table_index_build_range_scan(table_rel, index_rel, index_info, allow_sync, false, progress, 0, InvalidBlockNumber,
										callback, callback_state, scan);
}

/* utils/misc/sampling.c.  The block sampler is Knuth's Algorithm S (choose n of N blocks in order, each subset equally
 * likely); the row reservoir is Vitter's: the server switches from his Algorithm X to Z once t > 22 n, here X serves
 * throughout -- the same distribution, more draws.  Each sampler owns its generator (pg_prng_state inside the struct,
 * seeded from the global stream), as in the server. */
double
sampler_random_fract(pg_prng_state *randstate)
{
	double		r;

	do
		r = pg_prng_double(randstate);
	while (r == 0.0);			/* (0, 1) */
	return r;
}

BlockNumber
BlockSampler_Init(BlockSampler bs, BlockNumber nblocks, int samplesize, uint32 randseed)
{
	bs->N = nblocks;
	bs->n = samplesize;
	bs->t = 0;
	bs->m = 0;
	pg_prng_seed(&bs->randstate, (uint64) randseed);
	return (BlockNumber) bs->n < bs->N ? (BlockNumber) bs->n : bs->N;
}

bool
BlockSampler_HasMore(BlockSampler bs)
{
	return bs->t < bs->N && bs->m < bs->n;
}

BlockNumber
BlockSampler_Next(BlockSampler bs)
{
	BlockNumber left = bs->N - bs->t;	/* blocks not looked at yet */
	int			want = bs->n - bs->m;	/* blocks still to choose */

	if ((BlockNumber) want < left)
	{
		/* skip blocks while the draw says "not this one": block t is taken with probability want / left */
		double		v = sampler_random_fract(&bs->randstate);
		double		skip = 1.0 - (double) want / (double) left;

		while (v < skip)
		{
			bs->t++;
			left--;
			skip *= 1.0 - (double) want / (double) left;
		}
	}
	/* (want >= left: every remaining block is needed, no draw) */
	bs->m++;
	return bs->t++;
}

void
reservoir_init_selection_state(ReservoirState rs, int n)
{
	pg_prng_seed(&rs->randstate, (uint64) pg_prng_uint32(&pg_global_prng_state));
	rs->W = exp(-log(sampler_random_fract(&rs->randstate)) / n);	/* (Algorithm Z's state: drawn, never used here) */
}

/* how many of the rows after the t-th to pass over before one replaces a reservoir slot */
double
reservoir_get_next_S(ReservoirState rs, double t, int n)
{
	double		v = sampler_random_fract(&rs->randstate);
	double		s = 0;
	double		q;

	t += 1;
	q = (t - (double) n) / t;
	while (q > v)
	{
		s += 1;
		t += 1;
		q *= (t - (double) n) / t;
	}
	return s;
}

/* ---- locks of a single process, counters, the WAL that is not written */
void
SpinLockInit(volatile slock_t *lock)
{
	*lock = 0;
}

void
SpinLockAcquire(volatile slock_t *lock)
{
	while (__atomic_exchange_n(lock, 1, __ATOMIC_ACQUIRE))
		;
}

void
SpinLockRelease(volatile slock_t *lock)
{
	__atomic_store_n(lock, 0, __ATOMIC_RELEASE);
}

void
LWLockInitialize(LWLock *lock, int tranche_id)
{
	lock->tranche = (uint16) tranche_id;
	lock->state = 0;
}

int			hnsw_lock_tranche_id = 0;	/* src/hnsw.c:31 (the handler file is not in the program) */

void
HnswInitLockTranche(void)
{
	hnsw_lock_tranche_id = 77;	/* src/hnsw.c:36-50 registers a named tranche in shared memory */
}

static int64 progress_params[32];

void
pgstat_progress_update_param(int index, int64 val)
{
	if (index >= 0 && index < 32)
		progress_params[index] = val;
}

int64_t
shim_progress_param(int index)
{
	return progress_params[index];
}

void
pgstat_report_activity(BackendState state, const char *cmd_str)
{
	(void) state;
	(void) cmd_str;
}

void
log_newpage_range(Relation rel, ForkNumber forknum, BlockNumber startblk, BlockNumber endblk, bool page_std)
{
	(void) rel;
	(void) forknum;
	(void) startblk;
	(void) endblk;
	(void) page_std;
}

BlockNumber
RelationGetNumberOfBlocksInFork(Relation relation, ForkNumber forkNum)
{
	(void) forkNum;
	return RelationGetNumberOfBlocks(relation);
}

bool
RelationNeedsWAL(Relation relation)
{
	(void) relation;
	return true;
}

IndexInfo *
BuildIndexInfo(Relation index)
{
	(void) index;
	return palloc0(sizeof(IndexInfo));
}

/* ---- parallel CREATE INDEX (src/ivfbuild.c:600-990, src/hnswbuild.c:760-1100) on the stand-in server
 *
 * What access/transam/parallel.c, storage/ipc/dsm.c + shm_toc.c, the parallel half of utils/sort/tuplesort.c and
 * access/table/tableam.c's parallel block scan do for a parallel index build, as small as it can be made:
 *   - a "dynamic shared memory segment" is a piece of the postmaster's shared pool (shim_shared_alloc): every process of
 *     this program was forked from the postmaster, so it sits at the same address in all of them; the table of contents
 *     is a few (key, pointer) pairs at its head;
 *   - LaunchParallelWorkers asks the postmaster for background workers running "ParallelWorkerMain", which finds the
 *     entry point the leader named ("IvfflatParallelBuildMain", "HnswParallelBuildMain": dlsym, the program is linked
 *     -rdynamic) and calls it; a worker that ends with an ERROR is counted, and the leader's waits raise it;
 *   - the parallel table scan hands out the heap's blocks one at a time from a shared counter;
 *   - a worker's tuplesort_performsort publishes its sorted run in shared memory, the leader's merges the runs (by list
 *     number, then heap TID: inside a list the tuples come out in heap order whoever scanned them);
 *   - condition variables are polled (their users loop on a predicate anyway).
 * plan_create_index_workers answers what the test asked for (shim_set_parallel_workers; 0 = a serial build). */
int			max_parallel_maintenance_workers = 2;
const char *debug_query_string = NULL;
struct SnapshotData
{
	int			unused;
}			SnapshotAnyData;

static int	planned_workers = 0;

void
shim_set_parallel_workers(int n)
{
	planned_workers = n;
}

int
plan_create_index_workers(Oid tableOid, Oid indexOid)
{
	(void) tableOid;
	(void) indexOid;
	return planned_workers;
}

int
RelationGetParallelWorkers(Relation relation, int defaultpw)
{
	(void) relation;
	return defaultpw;
}

#define SHIM_TOC_KEYS 8
struct shm_toc
{
	char		entry[64];		/* the worker entry point's name */
	uint32		launched,
				finished,
				failed;
	int			nkeys;
	uint64		keys[SHIM_TOC_KEYS];
	void	   *addr[SHIM_TOC_KEYS];
	int			ngucs;			/* the leader's settings, restored in every worker (RestoreGUCState) */
	int			gucs[16];
	int			maintenance_work_mem;
	uint32		next_worker_number;	/* ParallelWorkerNumber of the next worker to attach (access/parallel.c) */
};

struct dsm_segment
{
	shm_toc    *toc;
};

static char pending_entry[64];
static shm_toc *leader_toc = NULL;	/* the parallel context this process leads (for the polled waits) */

void
EnterParallelMode(void)
{
}

void
ExitParallelMode(void)
{
}

ParallelContext *
CreateParallelContext(const char *library_name, const char *function_name, int nworkers)
{
	ParallelContext *pcxt = palloc0(sizeof(ParallelContext));

	(void) library_name;
	snprintf(pending_entry, sizeof(pending_entry), "%s", function_name);
	pcxt->nworkers = nworkers;
	return pcxt;
}

void
InitializeParallelDSM(ParallelContext *pcxt)
{
	shm_toc    *toc = shim_shared_alloc(sizeof(shm_toc));

	/* (the estimator's totals are not needed: chunks come from the pool one by one) */
	snprintf(toc->entry, sizeof(toc->entry), "%s", pending_entry);
	toc->ngucs = shim_guc_snapshot(toc->gucs, 16);
	toc->maintenance_work_mem = maintenance_work_mem;
	pcxt->seg = palloc0(sizeof(dsm_segment));
	pcxt->seg->toc = toc;
	pcxt->toc = toc;
}

void *
shm_toc_allocate(shm_toc *toc, Size nbytes)
{
	(void) toc;
	return shim_shared_alloc(nbytes);
}

void
shm_toc_insert(shm_toc *toc, uint64 key, void *address)
{
	if (toc->nkeys == SHIM_TOC_KEYS)
		elog(ERROR, "stand-in shm_toc: too many keys");
	toc->keys[toc->nkeys] = key;
	toc->addr[toc->nkeys] = address;
	__atomic_add_fetch(&toc->nkeys, 1, __ATOMIC_RELEASE);
}

void *
shm_toc_lookup(shm_toc *toc, uint64 key, bool noError)
{
	int			n = __atomic_load_n(&toc->nkeys, __ATOMIC_ACQUIRE);

	for (int i = 0; i < n; i++)
		if (toc->keys[i] == key)
			return toc->addr[i];
	if (!noError)
		elog(ERROR, "could not find key " UINT64_FORMAT " in shm TOC", key);
	return NULL;
}

/* what a parallel worker process runs (bgw_function_name "ParallelWorkerMain"; the test registers it with the postmaster) */
void
ParallelWorkerMain(Datum main_arg)
{
	shm_toc    *toc = (shm_toc *) DatumGetPointer(main_arg);
	void		(*entry) (dsm_segment *, shm_toc *) = (void (*) (dsm_segment *, shm_toc *)) dlsym(RTLD_DEFAULT, toc->entry);
	dsm_segment seg;

	seg.toc = toc;
	ParallelWorkerNumber = (int) __atomic_fetch_add(&toc->next_worker_number, 1, __ATOMIC_SEQ_CST);
	shim_guc_restore(toc->gucs, toc->ngucs);
	maintenance_work_mem = toc->maintenance_work_mem;
	if (entry == NULL)
	{
		__atomic_add_fetch(&toc->failed, 1, __ATOMIC_SEQ_CST);
		elog(ERROR, "parallel worker: no function \"%s\" in this program", toc->entry);
	}
	PG_TRY();
	{
		entry(&seg, toc);
	}
	PG_CATCH();
	{
		__atomic_add_fetch(&toc->failed, 1, __ATOMIC_SEQ_CST);
		PG_RE_THROW();
	}
	PG_END_TRY();
	__atomic_add_fetch(&toc->finished, 1, __ATOMIC_SEQ_CST);
}

void
LaunchParallelWorkers(ParallelContext *pcxt)
{
	BackgroundWorker worker;

	memset(&worker, 0, sizeof(worker));
	snprintf(worker.bgw_function_name, BGW_MAXLEN, "ParallelWorkerMain");
	worker.bgw_main_arg = PointerGetDatum(pcxt->toc);
	pcxt->nworkers_launched = 0;
	/* (the participants of the scan to come, the leader among them: known to the workers before the first of them starts) */
	__atomic_store_n(shim_shared_slot(1), (void *) (uintptr_t) (pcxt->nworkers + 1), __ATOMIC_RELEASE);
	for (int i = 0; i < pcxt->nworkers; i++)
		if (RegisterDynamicBackgroundWorker(&worker, NULL))
			pcxt->nworkers_launched++;
	pcxt->toc->launched = (uint32) pcxt->nworkers_launched;
	__atomic_store_n(shim_shared_slot(1), (void *) (uintptr_t) (pcxt->nworkers_launched + 1), __ATOMIC_RELEASE);
	leader_toc = pcxt->toc;
}

static void
check_workers(void)
{
	CHECK_FOR_INTERRUPTS();
	if (leader_toc && __atomic_load_n(&leader_toc->failed, __ATOMIC_ACQUIRE) > 0)
	{
		leader_toc = NULL;
		elog(ERROR, "a parallel worker of this build ended with an error");
	}
}

void
WaitForParallelWorkersToAttach(ParallelContext *pcxt)
{
	(void) pcxt;
	check_workers();
}

void
WaitForParallelWorkersToFinish(ParallelContext *pcxt)
{
	while (__atomic_load_n(&pcxt->toc->finished, __ATOMIC_ACQUIRE) < pcxt->toc->launched)
	{
		check_workers();
		usleep(200);
	}
}

void
DestroyParallelContext(ParallelContext *pcxt)
{
	leader_toc = NULL;
	pfree(pcxt->seg);
	pfree(pcxt);
}

void
ConditionVariableInit(ConditionVariable *cv)
{
	memset(cv, 0, sizeof(*cv));
}

void
ConditionVariableSleep(ConditionVariable *cv, uint32 wait_event_info)
{
	(void) cv;
	(void) wait_event_info;
	check_workers();
	usleep(200);				/* (callers re-test their predicate: a wakeup with nothing changed is allowed) */
}

void
ConditionVariableCancelSleep(void)
{
}

void
ConditionVariableSignal(ConditionVariable *cv)
{
	(void) cv;
}

/* ---- the parallel block scan of the stand-in heap (heap_def lives in shared memory: see shim_heap_relation) */
Size
table_parallelscan_estimate(Relation rel, Snapshot snapshot)
{
	(void) rel;
	(void) snapshot;
	return sizeof(struct ParallelTableScanDescData);
}

void
table_parallelscan_initialize(Relation rel, ParallelTableScanDesc pscan, Snapshot snapshot)
{
	(void) rel;
	(void) snapshot;
	pscan->next_block = 0;
	pscan->attached = 0;
	pscan->mode = 0;
	pscan->participants = 0;
}

TableScanDesc
table_beginscan_parallel(Relation relation, ParallelTableScanDesc pscan)
{
	TableScanDesc scan = palloc0(sizeof(struct TableScanDescData));

	(void) relation;
	scan->pscan = pscan;
	/* (test determinism, not the server's behaviour: the server hands out blocks first come first served -- here the
	 * first process would be through a small test table before the others have set up.  So the participants wait for
	 * each other, a few seconds at most, and the blocks are then dealt round-robin in the order they arrived: a split of
	 * the table the server could produce too, with a share for everyone.  If somebody does not turn up in time the scan
	 * falls back to the shared counter.) */
	{
		const uint32 expected = (uint32) (uintptr_t) __atomic_load_n(shim_shared_slot(1), __ATOMIC_ACQUIRE);
		double		until = shim_now() + 5.0;
		uint32		undecided = 0;

		scan->my_index = __atomic_fetch_add(&pscan->attached, 1, __ATOMIC_SEQ_CST);
		while (__atomic_load_n(&pscan->attached, __ATOMIC_ACQUIRE) < expected && __atomic_load_n(&pscan->mode, __ATOMIC_ACQUIRE) == 0 &&
			   shim_now() < until)
			__builtin_ia32_pause();
		if (__atomic_load_n(&pscan->attached, __ATOMIC_ACQUIRE) >= expected && expected > 0)
		{
			pscan->participants = expected;
			__atomic_compare_exchange_n(&pscan->mode, &undecided, 1u, 0, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST);
		}
		else
			__atomic_compare_exchange_n(&pscan->mode, &undecided, 2u, 0, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST);
		scan->my_next = scan->my_index;
	}
	return scan;
}

Relation
table_open(Oid relationId, LOCKMODE lockmode)
{
	(void) lockmode;
	if (relationId != SHIM_HEAP_OID || shim_heap_attach() == NULL)
		elog(ERROR, "could not open relation with OID %u", relationId);
	return shim_heap_attach();
}

void
table_close(Relation relation, LOCKMODE lockmode)
{
	(void) relation;
	(void) lockmode;
}

Relation
index_open(Oid relationId, LOCKMODE lockmode)
{
	Relation	r = shim_open_relation(relationId);

	(void) lockmode;
	if (r == NULL)
		elog(ERROR, "could not open relation with OID %u", relationId);
	return r;
}

Snapshot
GetTransactionSnapshot(void)
{
	return SnapshotAny;
}

Snapshot
RegisterSnapshot(Snapshot snapshot)
{
	return snapshot;
}

void
UnregisterSnapshot(Snapshot snapshot)
{
	(void) snapshot;
}

/* ---- the shared half of the build's tuplesort */
Size
tuplesort_estimate_shared(int nWorkers)
{
	(void) nWorkers;
	return sizeof(Sharedsort);
}

void
tuplesort_initialize_shared(Sharedsort *shared, int nWorkers, struct dsm_segment *seg)
{
	(void) nWorkers;
	(void) seg;
	memset(shared, 0, sizeof(*shared));
}

void
tuplesort_attach_shared(Sharedsort *shared, struct dsm_segment *seg)
{
	(void) shared;
	(void) seg;
}
#endif							/* PGV_HAVE_REF_IVFBUILD || PGV_HAVE_REF_HNSWBUILD */
