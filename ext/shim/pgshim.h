/*
 * pgshim.h -- NOT PostgreSQL.  The handful of server declarations the files in ext/ touch, spelled just
 * precisely enough for `gcc -fsyntax-only` to type-check the glue where no server headers exist
 * (tests/test_ext_glue_cpu.py).  Inside a real PGXS build these come from the server's own headers and
 * this directory is not on the include path.  Names and argument orders follow the PostgreSQL 13-17
 * headers named in each section; nothing here has a body.
 */
#ifndef PGSHIM_H
#define PGSHIM_H

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
typedef void (*pg_on_exit_callback) (int code, Datum arg);
void		on_proc_exit(pg_on_exit_callback function, Datum arg);
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
	MAIN_FORKNUM = 0
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

/* utils/rel.h, utils/inval.h */
struct RelationData
{
	Oid			rd_id;
	TupleDesc	rd_att;
};
#define RelationGetRelid(relation) ((relation)->rd_id)
#define RelationGetDescr(relation) ((relation)->rd_att)
typedef void (*RelcacheCallbackFunction) (Datum arg, Oid relid);
void		CacheRegisterRelcacheCallback(RelcacheCallbackFunction func, Datum arg);

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
	int			numberOfOrderBys;
	ScanKeyData *orderByData;
	void	   *opaque;
	ItemPointerData xs_heaptid;
	bool		xs_recheck;
	bool		xs_recheckorderby;
}			IndexScanDescData;
typedef IndexScanDescData *IndexScanDesc;

/* fmgr.h (only what the glue names) */
typedef Datum (*PGFunction) (void *fcinfo);
typedef struct FmgrInfo
{
	PGFunction	fn_addr;
	Oid			fn_oid;
}			FmgrInfo;
FmgrInfo   *index_getprocinfo(Relation irel, int attnum, uint16 procnum);

#endif							/* PGSHIM_H */
