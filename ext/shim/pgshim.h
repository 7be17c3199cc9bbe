/*
 * pgshim.h -- NOT PostgreSQL.  The handful of server declarations the files in ext/ touch, spelled just
 * precisely enough for the glue to be type-checked (tests/test_ext_glue_cpu.py) AND EXECUTED where no server headers
 * exist: tests/c/pgshim_runtime.c gives every function declared here a small body (palloc over resettable contexts
 * with reset callbacks, ereport as a longjmp to the innermost PG_TRY, the buffer manager over an emulated page image,
 * LWLocks / latches / atomics over futexes in a shared mapping, background workers as forked processes), and
 * tests/c/ext_driver.c runs the glue's scans, build hooks, worker and pooler on it (tests/test_ext_runtime_*.py).
 * Inside a real PGXS build these come from the server's own headers and this directory is not on the include path.
 * Names and argument orders follow the PostgreSQL 13-17 headers named in each section.
 */
#ifndef PGSHIM_H
#define PGSHIM_H

#ifndef PG_VERSION_NUM
#define PG_VERSION_NUM 170000	/* the branches of the reference's #if ladders that are type-checked */
#endif

#include <setjmp.h>
#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>

/* c.h / postgres.h */
typedef uint8_t uint8;
typedef uint16_t uint16;
typedef uint32_t uint32;
typedef uint64_t uint64;
typedef int16_t int16;
typedef int32_t int32;
typedef int64_t int64;
typedef size_t Size;
typedef uintptr_t Datum;
typedef unsigned int Oid;
typedef char *Pointer;
#define PG_INT64_MAX INT64_MAX
#define INT64_FORMAT "%ld"
#define Min(a, b) ((a) < (b) ? (a) : (b))
#define Max(a, b) ((a) > (b) ? (a) : (b))
#define PointerGetDatum(p) ((Datum) (p))
#define DatumGetPointer(d) ((Pointer) (d))
#define Int32GetDatum(x) ((Datum) (x))
#define Float8GetDatum(x) float8_as_datum(x)
Datum		float8_as_datum(double x);
#define PG_DETOAST_DATUM(d) ((struct varlena *) pg_detoast_datum((struct varlena *) DatumGetPointer(d)))
struct varlena
{
	char		vl_len_[4];
	char		vl_dat[];
};
struct varlena *pg_detoast_datum(struct varlena *datum);

/* utils/elog.h */
#define DEBUG1 14
#define LOG 15
#define WARNING 19
#define ERROR 21
#define ereport(level, rest) pgshim_ereport(level, rest)
#define errmsg(...) pgshim_errmsg(__VA_ARGS__)
#define elog(level, ...) pgshim_elog(level, __VA_ARGS__)
void		pgshim_ereport(int level, int dummy);
int			pgshim_errmsg(const char *fmt,...) __attribute__((format(printf, 1, 2)));
void		pgshim_elog(int level, const char *fmt,...);

/* utils/palloc.h, utils/memutils.h */
typedef struct MemoryContextData *MemoryContext;
extern MemoryContext TopMemoryContext;
extern MemoryContext CurrentMemoryContext;
void	   *palloc(Size size);
void	   *palloc0(Size size);
void	   *repalloc(void *p, Size size);
void		pfree(void *p);
void	   *MemoryContextAlloc(MemoryContext ctx, Size size);
MemoryContext MemoryContextSwitchTo(MemoryContext ctx);
void	   *MemoryContextAllocZero(MemoryContext ctx, Size size);
typedef void (*MemoryContextCallbackFunction) (void *arg);
typedef struct MemoryContextCallback
{
	MemoryContextCallbackFunction func;
	void	   *arg;
	struct MemoryContextCallback *next;
}			MemoryContextCallback;
void		MemoryContextRegisterResetCallback(MemoryContext ctx, MemoryContextCallback *cb);

/* miscadmin.h, storage/ipc.h, utils/guc.h */
#define CHECK_FOR_INTERRUPTS() pgshim_check_interrupts()
void		pgshim_check_interrupts(void);
#include <signal.h>
extern volatile sig_atomic_t InterruptPending;	/* set by the signal handlers: something for CHECK_FOR_INTERRUPTS to do */
extern volatile sig_atomic_t QueryCancelPending;	/* ... a statement cancel (SIGINT) */
extern volatile sig_atomic_t ProcDiePending;	/* ... a termination request (SIGTERM) */
typedef void (*pg_on_exit_callback) (int code, Datum arg);
void		on_proc_exit(pg_on_exit_callback function, Datum arg);
void		before_shmem_exit(pg_on_exit_callback function, Datum arg);
typedef enum
{
	PGC_USERSET = 6
}			GucContext;
void		DefineCustomBoolVariable(const char *name, const char *short_desc, const char *long_desc, bool *valueAddr,
									 bool bootValue, GucContext context, int flags, void *check, void *assign, void *show);
void		DefineCustomIntVariable(const char *name, const char *short_desc, const char *long_desc, int *valueAddr,
									int bootValue, int minValue, int maxValue, GucContext context, int flags,
									void *check, void *assign, void *show);

/* storage/block.h, storage/off.h, storage/itemptr.h */
typedef uint32 BlockNumber;
typedef uint16 OffsetNumber;
#define InvalidBlockNumber ((BlockNumber) 0xFFFFFFFF)
#define BlockNumberIsValid(b) ((b) != InvalidBlockNumber)
#define FirstOffsetNumber ((OffsetNumber) 1)
#define OffsetNumberNext(o) ((OffsetNumber) (1 + (o)))
typedef struct BlockIdData
{
	uint16		bi_hi;
	uint16		bi_lo;
}			BlockIdData;
typedef struct ItemPointerData
{
	BlockIdData ip_blkid;
	OffsetNumber ip_posid;
}			ItemPointerData;
typedef ItemPointerData *ItemPointer;

/* storage/buf.h, storage/bufmgr.h, storage/bufpage.h */
typedef int Buffer;
typedef char *Page;
typedef struct ItemIdData *ItemId;
typedef Pointer Item;
typedef struct BufferAccessStrategyData *BufferAccessStrategy;
typedef enum
{
	MAIN_FORKNUM = 0,
	INIT_FORKNUM = 3
}			ForkNumber;
typedef enum
{
	RBM_NORMAL = 0
}			ReadBufferMode;
#define BUFFER_LOCK_SHARE 1
struct RelationData;
typedef struct RelationData *Relation;
Buffer		ReadBufferExtended(Relation reln, ForkNumber forkNum, BlockNumber blockNum, ReadBufferMode mode,
							   BufferAccessStrategy strategy);
void		LockBuffer(Buffer buffer, int mode);
void		UnlockReleaseBuffer(Buffer buffer);
Page		BufferGetPage(Buffer buffer);
OffsetNumber PageGetMaxOffsetNumber(Page page);
ItemId		PageGetItemId(Page page, OffsetNumber offsetNumber);
Item		PageGetItem(Page page, ItemId itemId);
char	   *PageGetSpecialPointer(Page page);
char	   *PageGetContents(Page page);
BlockNumber RelationGetNumberOfBlocks(Relation reln);
#define ItemPointerIsValid(p) ((p) != NULL && (p)->ip_posid != 0)
BlockNumber ItemPointerGetBlockNumber(const ItemPointerData *p);
OffsetNumber ItemPointerGetOffsetNumber(const ItemPointerData *p);

/* nodes/pg_list.h */
typedef struct List List;
#define NIL ((List *) NULL)
List	   *lappend(List *list, void *datum);

/* access/itup.h, access/tupdesc.h */
typedef struct IndexTupleData
{
	ItemPointerData t_tid;
	unsigned short t_info;
}			IndexTupleData;
typedef IndexTupleData *IndexTuple;
typedef struct TupleDescData *TupleDesc;
Datum		index_getattr(IndexTuple tup, int attnum, TupleDesc tupleDesc, bool *isnull);

/* access/parallel.h */
extern int	ParallelWorkerNumber;

/* utils/rel.h, utils/inval.h */
struct RelationData
{
	Oid			rd_id;
	TupleDesc	rd_att;
	Oid		   *rd_indcollation;
	struct varlena *rd_options;
};
#define RelationGetRelid(relation) ((relation)->rd_id)
#define RelationGetDescr(relation) ((relation)->rd_att)
typedef void (*RelcacheCallbackFunction) (Datum arg, Oid relid);
void		CacheRegisterRelcacheCallback(RelcacheCallbackFunction func, Datum arg);
void		AcceptInvalidationMessages(void);

/* access/relscan.h, access/sdir.h, access/skey.h */
typedef enum
{
	ForwardScanDirection = 1
}			ScanDirection;
typedef struct ScanKeyData
{
	int			sk_flags;
	Datum		sk_argument;
}			ScanKeyData;
#define SK_ISNULL 0x0001
typedef struct IndexScanDescData
{
	Relation	indexRelation;
	struct SnapshotData *xs_snapshot;
	int			numberOfKeys;
	int			numberOfOrderBys;
	ScanKeyData *keyData;
	ScanKeyData *orderByData;
	void	   *opaque;
	ItemPointerData xs_heaptid;
	bool		xs_recheck;
	bool		xs_recheckorderby;
}			IndexScanDescData;
typedef IndexScanDescData *IndexScanDesc;

/* fmgr.h (only what the glue names) */
typedef struct FunctionCallInfoBaseData *FunctionCallInfo;
#define PG_FUNCTION_ARGS FunctionCallInfo fcinfo
typedef Datum (*PGFunction) (FunctionCallInfo fcinfo);
typedef struct FmgrInfo
{
	PGFunction	fn_addr;
	Oid			fn_oid;
}			FmgrInfo;
FmgrInfo   *index_getprocinfo(Relation irel, int attnum, uint16 procnum);

/* ---- declarations the cross-process mirror registry touches (pgv_context.c) ---- */
#include <stdio.h>

/* postgres.h, postgres_ext.h */
#define InvalidOid ((Oid) 0)
#define ObjectIdGetDatum(x) ((Datum) (x))
#define DatumGetObjectId(x) ((Oid) (x))

/* utils/elog.h: an ERROR longjmps to the innermost PG_TRY (or to the top-level handler of the process) */
extern sigjmp_buf *PG_exception_stack;
#define PG_TRY() \
	do { \
		sigjmp_buf *pgshim_saved_stack = PG_exception_stack; \
		sigjmp_buf	pgshim_local_jmp; \
		if (sigsetjmp(pgshim_local_jmp, 0) == 0) \
		{ \
			PG_exception_stack = &pgshim_local_jmp
#define PG_CATCH() \
		} \
		else \
		{ \
			PG_exception_stack = pgshim_saved_stack
#define PG_END_TRY() \
		} \
		PG_exception_stack = pgshim_saved_stack; \
	} while (0)
#define PG_RE_THROW() pgshim_rethrow()
void		pgshim_rethrow(void) __attribute__((noreturn));
void		EmitErrorReport(void);
void		FlushErrorState(void);

/* utils/palloc.h */
#define MCXT_ALLOC_HUGE 0x01
void	   *palloc_extended(Size size, int flags);
void	   *repalloc_huge(void *pointer, Size size);

/* miscadmin.h, storage/ipc.h */
extern Oid	MyDatabaseId;
extern int	MyProcPid;
extern bool process_shared_preload_libraries_in_progress;
typedef void (*shmem_request_hook_type) (void);
typedef void (*shmem_startup_hook_type) (void);
extern shmem_request_hook_type shmem_request_hook;
extern shmem_startup_hook_type shmem_startup_hook;
void		proc_exit(int code) __attribute__((noreturn));
void		pg_usleep(long microsec);

/* storage/shmem.h, storage/lwlock.h */
typedef struct LWLock
{
	uint16		tranche;
	uint32		state;			/* (the stand-in runtime's lock word) */
}			LWLock;
typedef union LWLockPadded
{
	LWLock		lock;
	char		pad[128];
}			LWLockPadded;
typedef enum LWLockMode
{
	LW_EXCLUSIVE,
	LW_SHARED
}			LWLockMode;
extern LWLock *AddinShmemInitLock;
bool		LWLockAcquire(LWLock *lock, LWLockMode mode);
void		LWLockRelease(LWLock *lock);
void		RequestAddinShmemSpace(Size size);
void		RequestNamedLWLockTranche(const char *tranche_name, int num_lwlocks);
LWLockPadded *GetNamedLWLockTranche(const char *tranche_name);
void	   *ShmemInitStruct(const char *name, Size size, bool *foundPtr);

/* port/atomics.h */
typedef struct pg_atomic_uint64
{
	volatile uint64 value;
}			pg_atomic_uint64;
void		pg_atomic_init_u64(volatile pg_atomic_uint64 *ptr, uint64 val);
uint64		pg_atomic_read_u64(volatile pg_atomic_uint64 *ptr);
void		pg_atomic_write_u64(volatile pg_atomic_uint64 *ptr, uint64 val);
uint64		pg_atomic_fetch_add_u64(volatile pg_atomic_uint64 *ptr, int64 add_);
bool		pg_atomic_compare_exchange_u64(volatile pg_atomic_uint64 *ptr, uint64 *expected, uint64 newval);
typedef struct pg_atomic_uint32
{
	volatile uint32 value;
}			pg_atomic_uint32;
void		pg_atomic_init_u32(volatile pg_atomic_uint32 *ptr, uint32 val);
uint32		pg_atomic_read_u32(volatile pg_atomic_uint32 *ptr);
void		pg_atomic_write_u32(volatile pg_atomic_uint32 *ptr, uint32 val);
bool		pg_atomic_compare_exchange_u32(volatile pg_atomic_uint32 *ptr, uint32 *expected, uint32 newval);

/* storage/latch.h, utils/wait_event.h */
typedef struct Latch Latch;
extern Latch *MyLatch;
#define WL_LATCH_SET (1 << 0)
#define WL_TIMEOUT (1 << 3)
#define WL_EXIT_ON_PM_DEATH (1 << 5)
#define PG_WAIT_EXTENSION 0x07000000U
int			WaitLatch(Latch *latch, int wakeEvents, long timeout, uint32 wait_event_info);
void		SetLatch(Latch *latch);
void		ResetLatch(Latch *latch);

/* utils/timestamp.h, datatype/timestamp.h */
typedef int64 TimestampTz;
TimestampTz GetCurrentTimestamp(void);
bool		TimestampDifferenceExceeds(TimestampTz start_time, TimestampTz stop_time, int msec);
#define TimestampTzPlusMilliseconds(tz, ms) ((tz) + ((ms) * (int64) 1000))

/* access/xact.h, access/genam.h, storage/lockdefs.h */
typedef int LOCKMODE;
#define AccessShareLock 1
void		StartTransactionCommand(void);
void		CommitTransactionCommand(void);
void		AbortCurrentTransaction(void);
Relation	try_index_open(Oid relationId, LOCKMODE lockmode);
void		index_close(Relation relation, LOCKMODE lockmode);

/* postmaster/bgworker.h */
#define BGWORKER_SHMEM_ACCESS 0x0001
#define BGWORKER_BACKEND_DATABASE_CONNECTION 0x0002
#define BGW_NEVER_RESTART -1
#define BGW_MAXLEN 96
typedef enum
{
	BgWorkerStart_PostmasterStart,
	BgWorkerStart_ConsistentState,
	BgWorkerStart_RecoveryFinished
}			BgWorkerStartTime;
typedef struct BackgroundWorker
{
	char		bgw_name[BGW_MAXLEN];
	char		bgw_type[BGW_MAXLEN];
	int			bgw_flags;
	BgWorkerStartTime bgw_start_time;
	int			bgw_restart_time;
	char		bgw_library_name[BGW_MAXLEN];
	char		bgw_function_name[BGW_MAXLEN];
	Datum		bgw_main_arg;
	int			bgw_notify_pid;
}			BackgroundWorker;
typedef struct BackgroundWorkerHandle BackgroundWorkerHandle;
bool		RegisterDynamicBackgroundWorker(BackgroundWorker *worker, BackgroundWorkerHandle **handle);
void		BackgroundWorkerUnblockSignals(void);
void		BackgroundWorkerInitializeConnectionByOid(Oid dboid, Oid useroid, uint32 flags);

#endif							/* PGSHIM_H */
