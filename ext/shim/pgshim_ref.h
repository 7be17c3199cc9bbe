/*
 * pgshim_ref.h -- NOT PostgreSQL.  Declarations only (no runtime stands behind them): what the REFERENCE's own files
 * name once ext/pgvector-0.8.6-gpu.patch is applied to them, so that tests/test_ext_patch_cpu.py can compile the patched
 * src/ivfscan.c, ivfbuild.c, ivfkmeans.c, ivfinsert.c, ivfvacuum.c, hnswscan.c, hnswbuild.c, hnswinsert.c, hnswvacuum.c
 * (-fsyntax-only) together with the glue of ext/ where no server headers exist: the hook lines are type-checked in the code
 * they are inserted into, against the patched ivfflat.h / hnsw.h themselves.  Spelled after the PostgreSQL 17 headers
 * named in each section, just precisely enough for that.  pgshim.h is the part that also has a runtime
 * (tests/c/pgshim_runtime.c); everything here is syntax only.
 */
#ifndef PGSHIM_REF_H
#define PGSHIM_REF_H

#include "pgshim.h"

#include <float.h>
#include <limits.h>
#include <math.h>
#include <stdlib.h>

/* c.h */
typedef int8_t int8;
typedef float float4;
typedef double float8;
typedef uint32 bits32;
typedef size_t Index_unused_;
typedef unsigned int Index;
typedef int16 AttrNumber;
typedef uint32 TransactionId;
typedef uint64 XLogRecPtr;
#define FLEXIBLE_ARRAY_MEMBER
#define PGDLLEXPORT
#define pg_attribute_noreturn() __attribute__((noreturn))
#define pg_attribute_unused() __attribute__((unused))
#define pg_unreachable() __builtin_unreachable()
#define likely(x) __builtin_expect((x) != 0, 1)
#define unlikely(x) __builtin_expect((x) != 0, 0)
#define Assert(x) ((void) 0)
#define StaticAssertDecl(c, m) _Static_assert(c, m)
#define StaticAssertStmt(c, m) _Static_assert(c, m)
#define lengthof(a) (sizeof(a) / sizeof((a)[0]))
#define MAXIMUM_ALIGNOF 8
#define TYPEALIGN(a, len) (((uintptr_t) (len) + ((a) - 1)) & ~((uintptr_t) ((a) - 1)))
#define MAXALIGN(len) TYPEALIGN(MAXIMUM_ALIGNOF, (len))
#define MAXALIGN_DOWN(len) (((uintptr_t) (len)) & ~((uintptr_t) (MAXIMUM_ALIGNOF - 1)))
#define BLCKSZ 8192
#define PG_INT32_MAX INT32_MAX
#define PG_UINT32_MAX UINT32_MAX
#define PG_UINT64_MAX UINT64_MAX
#define INT64CONST(x) (x##L)
#define UINT64CONST(x) (x##UL)
#define UINT64_FORMAT "%lu"
#define HAVE_LONG_INT_64 1
#define PG_USED_FOR_ASSERTS_ONLY __attribute__((unused))
#define PG_BINARY_R "r"
#define CppAsString(x) #x
#define CppConcat(x, y) x##y
#define MemSet(start, val, len) memset(start, val, len)
#define VARHDRSZ ((int32) sizeof(int32))

/* postgres.h: Datum conversions */
#define BoolGetDatum(x) ((Datum) ((x) ? 1 : 0))
#define DatumGetBool(x) ((bool) ((x) != 0))
#define DatumGetInt32(x) ((int32) (x))
#define DatumGetInt16(x) ((int16) (x))
#define Int16GetDatum(x) ((Datum) (x))
#define UInt32GetDatum(x) ((Datum) (x))
#define DatumGetUInt32(x) ((uint32) (x))
#define Int64GetDatum(x) ((Datum) (x))
#define DatumGetInt64(x) ((int64) (x))
#define CStringGetDatum(x) PointerGetDatum(x)
#define DatumGetCString(x) ((char *) DatumGetPointer(x))
float8		DatumGetFloat8(Datum x);
float4		DatumGetFloat4(Datum x);
Datum		Float4GetDatum(float4 x);

/* varatt.h */
#define SET_VARSIZE(p, len) (((struct varlena *) (p))->vl_len_[0] = 0, *(int32 *) (p) = (int32) ((len) << 2))
Size		VARSIZE_ANY(const void *p);
Size		VARSIZE(const void *p);
Size		VARSIZE_ANY_EXHDR(const void *p);
char	   *VARDATA(void *p);
char	   *VARDATA_ANY(const void *p);
bool		VARATT_IS_COMPRESSED(const void *p);
bool		VARATT_IS_EXTENDED(const void *p);
bool		VARATT_IS_SHORT(const void *p);

/* utils/elog.h */
#define DEBUG2 13
#define INFO 17
#define NOTICE 18
#define FATAL 22
#define PANIC 23
int			pgshim_errcode(int sqlstate);
#define errcode(c) pgshim_errcode(c)
int			pgshim_errmore(const char *fmt,...) __attribute__((format(printf, 1, 2)));	/* detail / hint: the primary message stays */
#define errdetail(...) pgshim_errmore(__VA_ARGS__)
#define errhint(...) pgshim_errmore(__VA_ARGS__)
#define ERRCODE_FEATURE_NOT_SUPPORTED 1
#define ERRCODE_INVALID_PARAMETER_VALUE 2
#define ERRCODE_PROGRAM_LIMIT_EXCEEDED 3
#define ERRCODE_DATA_EXCEPTION 4
#define ERRCODE_NUMERIC_VALUE_OUT_OF_RANGE 5
#define ERRCODE_INVALID_TEXT_REPRESENTATION 6
#define ERRCODE_INVALID_BINARY_REPRESENTATION 7
#define ERRCODE_OUT_OF_MEMORY 8
#define ERRCODE_INTERNAL_ERROR 9
#define ERRCODE_ARRAY_ELEMENT_ERROR 10
#define ERRCODE_NULL_VALUE_NOT_ALLOWED 11

/* utils/palloc.h, utils/memutils.h */
#define MaxAllocSize ((Size) 0x3fffffff)
#define MaxAllocHugeSize (SIZE_MAX / 2)
#define MCXT_ALLOC_NO_OOM 0x02
#define MCXT_ALLOC_ZERO 0x04
#define ALLOCSET_DEFAULT_SIZES 0, 8 * 1024, 8 * 1024 * 1024
#define ALLOCSET_SMALL_SIZES 0, 1024, 8 * 1024
#define palloc_object(type) ((type *) palloc(sizeof(type)))
#define palloc0_object(type) ((type *) palloc0(sizeof(type)))
#define palloc_array(type, count) ((type *) palloc(sizeof(type) * (count)))
#define palloc0_array(type, count) ((type *) palloc0(sizeof(type) * (count)))
MemoryContext AllocSetContextCreate(MemoryContext parent, const char *name, Size minContextSize, Size initBlockSize,
									Size maxBlockSize);
MemoryContext GenerationContextCreate(MemoryContext parent, const char *name, Size minContextSize, Size initBlockSize,
									  Size maxBlockSize);
void		MemoryContextDelete(MemoryContext context);
void		MemoryContextReset(MemoryContext context);
Size		MemoryContextMemAllocated(MemoryContext context, bool recurse);
void	   *MemoryContextAllocExtended(MemoryContext context, Size size, int flags);
char	   *pstrdup(const char *in);
char	   *psprintf(const char *fmt,...);
Size		add_size(Size s1, Size s2);
Size		mul_size(Size s1, Size s2);

/* miscadmin.h, utils/guc.h */
extern int	work_mem;
extern int	maintenance_work_mem;
extern int	max_parallel_maintenance_workers;
extern const char *debug_query_string;
typedef enum
{
	PGC_POSTMASTER_ = 0
}			GucContextMore_;
struct config_enum_entry
{
	const char *name;
	int			val;
	bool		hidden;
};
void		DefineCustomEnumVariable(const char *name, const char *short_desc, const char *long_desc, int *valueAddr,
									 int bootValue, const struct config_enum_entry *options, GucContext context, int flags,
									 void *check, void *assign, void *show);
void		DefineCustomRealVariable(const char *name, const char *short_desc, const char *long_desc, double *valueAddr,
									 double bootValue, double minValue, double maxValue, GucContext context, int flags,
									 void *check, void *assign, void *show);
void		MarkGUCPrefixReserved(const char *className);

/* fmgr.h */
#define PG_FUNCTION_INFO_V1(f) extern Datum f(PG_FUNCTION_ARGS)
#define PG_MODULE_MAGIC extern int pgshim_module_magic_
Datum		FunctionCall1Coll(FmgrInfo *flinfo, Oid collation, Datum arg1);
Datum		FunctionCall2Coll(FmgrInfo *flinfo, Oid collation, Datum arg1, Datum arg2);
Datum		DirectFunctionCall1Coll(PGFunction func, Oid collation, Datum arg1);
Datum		DirectFunctionCall2Coll(PGFunction func, Oid collation, Datum arg1, Datum arg2);
#define DirectFunctionCall1(func, arg1) DirectFunctionCall1Coll((PGFunction) (func), InvalidOid, arg1)
#define DirectFunctionCall2(func, arg1, arg2) DirectFunctionCall2Coll((PGFunction) (func), InvalidOid, arg1, arg2)
Datum		pgshim_getarg(FunctionCallInfo fcinfo, int n);
bool		pgshim_argisnull(FunctionCallInfo fcinfo, int n);
#define PG_GETARG_DATUM(n) pgshim_getarg(fcinfo, n)
#define PG_GETARG_POINTER(n) DatumGetPointer(PG_GETARG_DATUM(n))
#define PG_GETARG_INT32(n) DatumGetInt32(PG_GETARG_DATUM(n))
#define PG_GETARG_INT16(n) DatumGetInt16(PG_GETARG_DATUM(n))
#define PG_GETARG_BOOL(n) DatumGetBool(PG_GETARG_DATUM(n))
#define PG_GETARG_OID(n) DatumGetObjectId(PG_GETARG_DATUM(n))
#define PG_GETARG_FLOAT8(n) DatumGetFloat8(PG_GETARG_DATUM(n))
#define PG_GETARG_FLOAT4(n) DatumGetFloat4(PG_GETARG_DATUM(n))
#define PG_GETARG_CSTRING(n) DatumGetCString(PG_GETARG_DATUM(n))
#define PG_ARGISNULL(n) pgshim_argisnull(fcinfo, n)
#define PG_RETURN_DATUM(x) return (x)
#define PG_RETURN_POINTER(x) return PointerGetDatum(x)
#define PG_RETURN_INT32(x) return Int32GetDatum(x)
#define PG_RETURN_BOOL(x) return BoolGetDatum(x)
#define PG_RETURN_FLOAT8(x) return Float8GetDatum(x)
#define PG_RETURN_FLOAT4(x) return Float4GetDatum(x)
#define PG_RETURN_CSTRING(x) return CStringGetDatum(x)
#define PG_RETURN_NULL() return (Datum) 0
#define PG_RETURN_VOID() return (Datum) 0
#define PG_GET_COLLATION() InvalidOid
#define PG_FREE_IF_COPY(ptr, n) ((void) 0)
#define PG_DETOAST_DATUM_COPY(d) PG_DETOAST_DATUM(d)

/* access/attnum.h, access/tupdesc.h, catalog/pg_attribute.h */
typedef struct FormData_pg_attribute
{
	Oid			atttypid;
	int32		atttypmod;
}			FormData_pg_attribute;
typedef FormData_pg_attribute *Form_pg_attribute;
Form_pg_attribute TupleDescAttr(TupleDesc tupdesc, int i);
TupleDesc	CreateTemplateTupleDesc(int natts);
void		TupleDescInitEntry(TupleDesc desc, AttrNumber attributeNumber, const char *attributeName, Oid oidtypeid,
							   int32 typmod, int attdim);

/* catalog/pg_type_d.h, catalog/pg_operator_d.h */
#define INT4OID 23
#define TIDOID 27
#define FLOAT4OID 700
#define FLOAT8OID 701
#define VARBITOID 1562
#define Int4LessOperator 97
#define Float8LessOperator 672

/* executor/tuptable.h */
typedef struct TupleTableSlotOps TupleTableSlotOps;
typedef struct TupleTableSlot
{
	Datum	   *tts_values;
	bool	   *tts_isnull;
}			TupleTableSlot;
extern const TupleTableSlotOps TTSOpsVirtual;
extern const TupleTableSlotOps TTSOpsMinimalTuple;
TupleTableSlot *MakeSingleTupleTableSlot(TupleDesc tupdesc, const TupleTableSlotOps *tts_ops);
TupleTableSlot *ExecClearTuple(TupleTableSlot *slot);
TupleTableSlot *ExecStoreVirtualTuple(TupleTableSlot *slot);
Datum		slot_getattr(TupleTableSlot *slot, int attnum, bool *isnull);

/* utils/tuplesort.h, utils/sortsupport.h */
struct dsm_segment;
typedef struct Tuplesortstate Tuplesortstate;
typedef struct Sharedsort Sharedsort;
typedef struct SortCoordinateData
{
	bool		isWorker;
	int			nParticipants;
	Sharedsort *sharedsort;
}			SortCoordinateData;
typedef SortCoordinateData *SortCoordinate;
Tuplesortstate *tuplesort_begin_heap(TupleDesc tupDesc, int nkeys, AttrNumber *attNums, Oid *sortOperators,
									 Oid *sortCollations, bool *nullsFirstFlags, int workMem, SortCoordinate coordinate,
									 int sortopt);
void		tuplesort_puttupleslot(Tuplesortstate *state, TupleTableSlot *slot);
void		tuplesort_performsort(Tuplesortstate *state);
bool		tuplesort_gettupleslot(Tuplesortstate *state, bool forward, bool copy, TupleTableSlot *slot, Datum *abbrev);
void		tuplesort_reset(Tuplesortstate *state);
void		tuplesort_end(Tuplesortstate *state);
Size		tuplesort_estimate_shared(int nWorkers);
void		tuplesort_initialize_shared(Sharedsort *shared, int nWorkers, struct dsm_segment *seg);
void		tuplesort_attach_shared(Sharedsort *shared, struct dsm_segment *seg);

/* lib/pairingheap.h */
typedef struct pairingheap_node
{
	struct pairingheap_node *first_child;
	struct pairingheap_node *next_sibling;
	struct pairingheap_node *prev_or_parent;
}			pairingheap_node;
typedef int (*pairingheap_comparator) (const pairingheap_node *a, const pairingheap_node *b, void *arg);
typedef struct pairingheap
{
	pairingheap_comparator ph_compare;
	void	   *ph_arg;
	pairingheap_node *ph_root;
}			pairingheap;
#define pairingheap_container(type, membername, ptr) ((type *) ((char *) (ptr) - offsetof(type, membername)))
#define pairingheap_const_container(type, membername, ptr) ((const type *) ((const char *) (ptr) - offsetof(type, membername)))
#define pairingheap_is_empty(h) ((h)->ph_root == NULL)
#define pairingheap_reset(h) ((h)->ph_root = NULL)
pairingheap *pairingheap_allocate(pairingheap_comparator compare, void *arg);
void		pairingheap_add(pairingheap *heap, pairingheap_node *node);
pairingheap_node *pairingheap_first(pairingheap *heap);
pairingheap_node *pairingheap_remove_first(pairingheap *heap);

/* nodes/pg_list.h */
typedef union ListCell
{
	void	   *ptr_value;
	int			int_value;
}			ListCell;
int			list_length(const List *l);
void	   *linitial(const List *l);
void	   *llast(const List *l);
List	   *list_delete_last(List *list);
List	   *list_delete_first(List *list);
List	   *lcons(void *datum, List *list);
List	   *list_concat_unique_ptr(List *list1, const List *list2);
void		list_free(List *list);
void		list_free_deep(List *list);
void		list_sort(List *list, int (*cmp) (const ListCell *a, const ListCell *b));
void	   *list_nth(const List *list, int n);
ListCell   *list_head(const List *l);
ListCell   *lnext(const List *l, const ListCell *c);
void	   *pgshim_lfirst(const ListCell *lc);
#define lfirst(lc) pgshim_lfirst(lc)
#define foreach(cell, lst) for ((cell) = list_head(lst); (cell) != NULL; (cell) = lnext(lst, cell))
#define list_make1(x) lappend(NIL, x)

/* storage/bufmgr.h, storage/bufpage.h, storage/off.h, storage/item.h */
#define InvalidBuffer 0
#define BufferIsValid(b) ((b) != InvalidBuffer)
#define P_NEW InvalidBlockNumber
#define BUFFER_LOCK_UNLOCK 0
#define BUFFER_LOCK_EXCLUSIVE 2
#define InvalidOffsetNumber ((OffsetNumber) 0)
#define OffsetNumberIsValid(o) ((o) != InvalidOffsetNumber)
#define MaxOffsetNumber ((OffsetNumber) (BLCKSZ / sizeof(uint32)))
#define SizeOfPageHeaderData 24
#define MaxItemSize_unused_ 0
typedef enum
{
	BAS_NORMAL_,
	BAS_BULKREAD,
	BAS_BULKWRITE,
	BAS_VACUUM
}			BufferAccessStrategyType;
typedef struct ItemIdData
{
	unsigned	lp_off:15,
				lp_flags:2,
				lp_len:15;
}			ItemIdData;
Buffer		ReadBuffer(Relation reln, BlockNumber blockNum);
void		ReleaseBuffer(Buffer buffer);
void		MarkBufferDirty(Buffer buffer);
BlockNumber BufferGetBlockNumber(Buffer buffer);
BufferAccessStrategy GetAccessStrategy(BufferAccessStrategyType btype);
void		FreeAccessStrategy(BufferAccessStrategy strategy);
BlockNumber RelationGetNumberOfBlocksInFork(Relation relation, ForkNumber forkNum);
void		LockRelationForExtension(Relation relation, LOCKMODE lockmode);
void		UnlockRelationForExtension(Relation relation, LOCKMODE lockmode);
void		PageInit(Page page, Size pageSize, Size specialSize);
Size		PageGetFreeSpace(Page page);
Size		PageGetExactFreeSpace(Page page);
Size		PageGetPageSize(Page page);
OffsetNumber PageAddItem(Page page, Item item, Size size, OffsetNumber offsetNumber, bool overwrite, bool is_heap);
bool		PageIndexTupleOverwrite(Page page, OffsetNumber offnum, Item newtup, Size newsize);
void		PageIndexMultiDelete(Page page, OffsetNumber *itemnos, int nitems);
bool		PageIsNew(Page page);
Size		ItemIdGetLength(ItemId itemId);
bool		ItemIdIsUsed(ItemId itemId);
bool		ItemIdIsDead(ItemId itemId);

/* storage/itemptr.h */
void		ItemPointerSet(ItemPointerData *pointer, BlockNumber blockNumber, OffsetNumber offNum);
void		ItemPointerSetInvalid(ItemPointerData *pointer);
bool		ItemPointerEquals(ItemPointer pointer1, ItemPointer pointer2);
void		ItemPointerCopy(const ItemPointerData *fromPointer, ItemPointerData *toPointer);
int32		ItemPointerCompare(ItemPointer arg1, ItemPointer arg2);

/* access/itup.h */
IndexTuple	index_form_tuple(TupleDesc tupleDescriptor, const Datum *values, const bool *isnull);
Size		IndexTupleSize(IndexTuple itup);

/* access/generic_xlog.h, access/xloginsert.h, access/xlogdefs.h */
typedef struct GenericXLogState GenericXLogState;
#define GENERIC_XLOG_FULL_IMAGE 0x0001
GenericXLogState *GenericXLogStart(Relation relation);
Page		GenericXLogRegisterBuffer(GenericXLogState *state, Buffer buffer, int flags);
XLogRecPtr	GenericXLogFinish(GenericXLogState *state);
void		GenericXLogAbort(GenericXLogState *state);
void		log_newpage_range(Relation rel, ForkNumber forknum, BlockNumber startblk, BlockNumber endblk, bool page_std);

/* storage/lmgr.h, storage/lockdefs.h */
#define RowExclusiveLock 3
#define ShareUpdateExclusiveLock 4
#define ShareLock 5
#define ExclusiveLock 7
#define AccessExclusiveLock 8
void		LockPage(Relation relation, BlockNumber blkno, LOCKMODE lockmode);
void		UnlockPage(Relation relation, BlockNumber blkno, LOCKMODE lockmode);

/* storage/lwlock.h, storage/s_lock.h, storage/spin.h, storage/condition_variable.h */
typedef unsigned char slock_t;
void		SpinLockInit(volatile slock_t *lock);
void		SpinLockAcquire(volatile slock_t *lock);
void		SpinLockRelease(volatile slock_t *lock);
typedef struct ConditionVariable
{
	slock_t		mutex;
	void	   *wakeup;
}			ConditionVariable;
void		ConditionVariableInit(ConditionVariable *cv);
void		ConditionVariableSleep(ConditionVariable *cv, uint32 wait_event_info);
void		ConditionVariableCancelSleep(void);
void		ConditionVariableSignal(ConditionVariable *cv);
void		LWLockInitialize(LWLock *lock, int tranche_id);
int			LWLockNewTrancheId(void);
void		LWLockRegisterTranche(int tranche_id, const char *tranche_name);
bool		LWLockConditionalAcquire(LWLock *lock, LWLockMode mode);
bool		LWLockHeldByMe(LWLock *lock);
#define WAIT_EVENT_PARALLEL_CREATE_INDEX_SCAN 0x0A000001U

/* utils/snapmgr.h, utils/snapshot.h */
typedef struct SnapshotData *Snapshot;
extern struct SnapshotData SnapshotAnyData;
#define SnapshotAny (&SnapshotAnyData)
bool		IsMVCCSnapshot(Snapshot snapshot);
Snapshot	GetTransactionSnapshot(void);
Snapshot	RegisterSnapshot(Snapshot snapshot);
void		UnregisterSnapshot(Snapshot snapshot);

/* pgstat.h, utils/backend_status.h, utils/backend_progress.h, commands/progress.h */
void		pgstat_count_index_scan(Relation rel);
typedef enum
{
	STATE_RUNNING = 2
}			BackendState;
void		pgstat_report_activity(BackendState state, const char *cmd_str);
void		pgstat_progress_update_param(int index, int64 val);
#define PROGRESS_CREATEIDX_SUBPHASE 9
#define PROGRESS_CREATEIDX_TUPLES_TOTAL 11
#define PROGRESS_CREATEIDX_TUPLES_DONE 12
#define PROGRESS_CREATEIDX_SUBPHASE_INITIALIZE 1

/* access/sdir.h, access/skey.h, access/genam.h, access/amapi.h, nodes/execnodes.h, commands/vacuum.h */
#define ScanDirectionIsForward(d) ((d) == ForwardScanDirection)
typedef ScanKeyData *ScanKey;
typedef struct IndexInfo
{
	bool		ii_Concurrent;
	int			ii_ParallelWorkers;
}			IndexInfo;
typedef struct IndexBuildResult
{
	double		heap_tuples;
	double		index_tuples;
}			IndexBuildResult;
typedef struct IndexVacuumInfo
{
	Relation	index;
	Relation	heaprel;
	bool		analyze_only;
	BufferAccessStrategy strategy;
}			IndexVacuumInfo;
typedef struct IndexBulkDeleteResult
{
	BlockNumber num_pages;
	double		num_index_tuples;
	double		tuples_removed;
}			IndexBulkDeleteResult;
typedef bool (*IndexBulkDeleteCallback) (ItemPointer itemptr, void *state);
typedef enum IndexUniqueCheck
{
	UNIQUE_CHECK_NO
}			IndexUniqueCheck;
IndexScanDesc RelationGetIndexScan(Relation indexRelation, int nkeys, int norderbys);
IndexInfo  *BuildIndexInfo(Relation index);
void		vacuum_delay_point(void);

/* utils/rel.h, utils/relcache.h, access/reloptions.h, catalog/index.h, access/table.h, access/tableam.h */
typedef struct HeapTupleData *HeapTuple;
typedef struct TableScanDescData *TableScanDesc;
typedef struct ParallelTableScanDescData *ParallelTableScanDesc;
typedef void (*IndexBuildCallback) (Relation index, ItemPointer tid, Datum *values, bool *isnull, bool tupleIsAlive,
									void *state);
Relation	table_open(Oid relationId, LOCKMODE lockmode);
void		table_close(Relation relation, LOCKMODE lockmode);
Relation	index_open(Oid relationId, LOCKMODE lockmode);
double		table_index_build_scan(Relation table_rel, Relation index_rel, IndexInfo *index_info, bool allow_sync,
								   bool progress, IndexBuildCallback callback, void *callback_state, TableScanDesc scan);
TableScanDesc table_beginscan_parallel(Relation relation, ParallelTableScanDesc pscan);
Size		table_parallelscan_estimate(Relation rel, Snapshot snapshot);
void		table_parallelscan_initialize(Relation rel, ParallelTableScanDesc pscan, Snapshot snapshot);
int			plan_create_index_workers(Oid tableOid, Oid indexOid);
bool		RelationNeedsWAL(Relation relation);
void	   *pgshim_rd_options(Relation relation);

/* access/parallel.h, storage/dsm.h, storage/shm_toc.h */
typedef struct dsm_segment dsm_segment;
typedef struct shm_toc shm_toc;
typedef struct shm_toc_estimator
{
	Size		space_for_chunks;
	Size		number_of_keys;
}			shm_toc_estimator;
typedef struct ParallelContext
{
	int			nworkers;
	int			nworkers_launched;
	shm_toc_estimator estimator;
	dsm_segment *seg;
	shm_toc    *toc;
}			ParallelContext;
#define shm_toc_estimate_chunk(e, sz) ((e)->space_for_chunks = add_size((e)->space_for_chunks, (sz)))
#define shm_toc_estimate_keys(e, cnt) ((e)->number_of_keys = add_size((e)->number_of_keys, (cnt)))
#define BUFFERALIGN(len) TYPEALIGN(32, (len))
void	   *shm_toc_allocate(shm_toc *toc, Size nbytes);
void		shm_toc_insert(shm_toc *toc, uint64 key, void *address);
void	   *shm_toc_lookup(shm_toc *toc, uint64 key, bool noError);
void		EnterParallelMode(void);
void		ExitParallelMode(void);
ParallelContext *CreateParallelContext(const char *library_name, const char *function_name, int nworkers);
void		InitializeParallelDSM(ParallelContext *pcxt);
void		LaunchParallelWorkers(ParallelContext *pcxt);
void		WaitForParallelWorkersToAttach(ParallelContext *pcxt);
void		WaitForParallelWorkersToFinish(ParallelContext *pcxt);
void		DestroyParallelContext(ParallelContext *pcxt);
bool		IsParallelWorker(void);

/* utils/sampling.h, common/pg_prng.h */
typedef struct pg_prng_state
{
	uint64		s0,
				s1;
}			pg_prng_state;
extern pg_prng_state pg_global_prng_state;
double		pg_prng_double(pg_prng_state *state);
uint32		pg_prng_uint32(pg_prng_state *state);
void		pg_prng_seed(pg_prng_state *state, uint64 seed);
typedef struct BlockSamplerData
{
	BlockNumber N;
	int			n;
	BlockNumber t;
	int			m;
	pg_prng_state randstate;
}			BlockSamplerData;
typedef BlockSamplerData *BlockSampler;
typedef struct ReservoirStateData
{
	double		W;
	pg_prng_state randstate;
}			ReservoirStateData;
typedef ReservoirStateData *ReservoirState;
BlockNumber BlockSampler_Init(BlockSampler bs, BlockNumber nblocks, int samplesize, uint32 randseed);
bool		BlockSampler_HasMore(BlockSampler bs);
BlockNumber BlockSampler_Next(BlockSampler bs);
void		reservoir_init_selection_state(ReservoirState rs, int n);
double		reservoir_get_next_S(ReservoirState rs, double t, int n);
double		sampler_random_fract(pg_prng_state *randstate);

/* portability/instr_time.h */
typedef struct instr_time
{
	int64		ticks;
}			instr_time;
#define INSTR_TIME_SET_CURRENT(t) ((t).ticks = 0)
#define INSTR_TIME_SUBTRACT(x, y) ((x).ticks -= (y).ticks)
#define INSTR_TIME_GET_MILLISEC(t) ((double) (t).ticks / 1000000.0)

/* utils/float.h */
float8		get_float8_infinity(void);
float4		get_float4_infinity(void);
float8		get_float8_nan(void);
void		float_overflow_error(void) pg_attribute_noreturn();
void		float_underflow_error(void) pg_attribute_noreturn();

/* utils/relptr.h: a pointer relative to a shared segment's base, 0 = NULL */
#define relptr(type) union { type *relptr_type; Size relptr_off; }
#define relptr_declare(type, relptrtype) typedef relptr(type) relptrtype
#define relptr_access(base, rp) ((rp).relptr_off == 0 ? (__typeof__((rp).relptr_type)) NULL : (__typeof__((rp).relptr_type)) ((base) + (rp).relptr_off - 1))
#define relptr_is_null(rp) ((rp).relptr_off == 0)
#define relptr_offset(rp) ((rp).relptr_off - 1)
#define relptr_store(base, rp, val) ((rp).relptr_off = ((val) == NULL ? 0 : (Size) (((char *) (val)) - (base)) + 1))

/* more of utils/rel.h, storage/bufmgr.h, access/htup_details.h, storage/bufpage.h */
const char *RelationGetRelationName(Relation relation);
void		LockBufferForCleanup(Buffer buffer);
bool		ConditionalLockBufferForCleanup(Buffer buffer);
double		table_index_build_range_scan(Relation table_rel, Relation index_rel, IndexInfo *index_info, bool allow_sync,
										 bool anyvisible, bool progress, BlockNumber start_blockno, BlockNumber numblocks,
										 IndexBuildCallback callback, void *callback_state, TableScanDesc scan);
#define MaxHeapTuplesPerPage 291
typedef uint16 LocationIndex;
typedef struct PageHeaderData
{
	uint64		pd_lsn;
	uint16		pd_checksum;
	uint16		pd_flags;
	LocationIndex pd_lower;
	LocationIndex pd_upper;
	LocationIndex pd_special;
	uint16		pd_pagesize_version;
	TransactionId pd_prune_xid;
	ItemIdData	pd_linp[];
}			PageHeaderData;
typedef PageHeaderData *PageHeader;

/* lib/stringinfo.h, libpq/pqformat.h */
typedef struct StringInfoData
{
	char	   *data;
	int			len;
	int			maxlen;
	int			cursor;
}			StringInfoData;
typedef StringInfoData *StringInfo;
typedef struct varlena bytea;
void		pq_begintypsend(StringInfo buf);
bytea	   *pq_endtypsend(StringInfo buf);
void		pq_sendint16(StringInfo buf, uint16 i);
void		pq_sendint32(StringInfo buf, uint32 i);
void		pq_sendint(StringInfo buf, uint32 i, int b);
void		pq_sendfloat4(StringInfo buf, float4 f);
void		pq_sendbytes(StringInfo buf, const void *data, int datalen);
unsigned int pq_getmsgint(StringInfo msg, int b);
float4		pq_getmsgfloat4(StringInfo msg);
const char *pq_getmsgbytes(StringInfo msg, int datalen);
#define PG_RETURN_BYTEA_P(x) PG_RETURN_POINTER(x)

/* utils/array.h, utils/lsyscache.h, catalog/pg_type.h, utils/fmgrprotos.h */
typedef struct ArrayType ArrayType;
int			ARR_NDIM(const ArrayType *a);
bool		ARR_HASNULL(const ArrayType *a);
Oid			ARR_ELEMTYPE(const ArrayType *a);
int		   *ARR_DIMS(const ArrayType *a);
char	   *ARR_DATA_PTR(const ArrayType *a);
#define PG_GETARG_ARRAYTYPE_P(n) ((ArrayType *) PG_DETOAST_DATUM(PG_GETARG_DATUM(n)))
#define PG_RETURN_ARRAYTYPE_P(x) PG_RETURN_POINTER(x)
#define DatumGetArrayTypeP(d) ((ArrayType *) PG_DETOAST_DATUM(d))
bool		array_contains_nulls(ArrayType *array);
void		deconstruct_array(ArrayType *array, Oid elmtype, int elmlen, bool elmbyval, char elmalign, Datum **elemsp,
							  bool **nullsp, int *nelemsp);
ArrayType  *construct_array(Datum *elems, int nelems, Oid elmtype, int elmlen, bool elmbyval, char elmalign);
int32	   *ArrayGetIntegerTypmods(ArrayType *arr, int *n);
void		get_typlenbyvalalign(Oid typid, int16 *typlen, bool *typbyval, char *typalign);
#define TYPALIGN_INT 'i'
#define TYPALIGN_DOUBLE 'd'
#define FLOAT8PASSBYVAL true
#define FLOAT4PASSBYVAL true
#define NUMERICOID 1700
#define INT2OID 21
Datum		numeric_float4(PG_FUNCTION_ARGS);
Datum		float8_numeric_unused_(PG_FUNCTION_ARGS);

/* utils/varbit.h */
typedef struct VarBit
{
	int32		vl_len_;
	int32		bit_len;
	uint8		bit_dat[];
}			VarBit;
#define VARBITS(v) ((v)->bit_dat)
#define VARBITLEN(v) ((v)->bit_len)
#define VARBITBYTES(v) (VARSIZE(v) - VARHDRSZ - sizeof(int32))
#define VARBITTOTALLEN(bitlen) (((bitlen) + 7) / 8 + VARHDRSZ + sizeof(int32))
#define PG_GETARG_VARBIT_P(n) ((VarBit *) PG_DETOAST_DATUM(PG_GETARG_DATUM(n)))
#define PG_RETURN_VARBIT_P(x) PG_RETURN_POINTER(x)

/* common/shortest_dec.h, parser/scansup.h, port.h */
#define FLOAT_SHORTEST_DECIMAL_LEN 16
int			float_to_shortest_decimal_buf(float f, char *result);
int			float_to_shortest_decimal_bufn(float f, char *result);
bool		scanner_isspace(char ch);
char	   *pnstrdup(const char *in, Size len);
#include <errno.h>

/* common/hashfn.h, utils/memdebug.h */
uint32		murmurhash32(uint32 data);
uint64		murmurhash64(uint64 data);
uint32		hash_bytes(const unsigned char *k, int keylen);
#define VALGRIND_MAKE_MEM_DEFINED(addr, size) ((void) 0)
#define VALGRIND_MAKE_MEM_NOACCESS(addr, size) ((void) 0)
#define VALGRIND_MAKE_MEM_UNDEFINED(addr, size) ((void) 0)

/* more of postgres_ext.h, access/genam.h, storage/bufmgr.h, port/atomics.h, nodes/pg_list.h, fmgr.h */
#define OidIsValid(objectId) ((bool) ((objectId) != InvalidOid))
Oid			index_getprocid(Relation irel, AttrNumber attnum, uint16 procnum);
Size		BufferGetPageSize(Buffer buffer);
#define pg_memory_barrier() __sync_synchronize()
List	   *list_copy(const List *list);
Datum		FunctionCall0Coll(FmgrInfo *flinfo, Oid collation);

/* utils/datum.h */
Datum		datumCopy(Datum value, bool typByVal, int typLen);
bool		datumIsEqual(Datum value1, Datum value2, bool typByVal, int typLen);
int			RelationGetParallelWorkers(Relation relation, int defaultpw);

#endif							/* PGSHIM_REF_H */
