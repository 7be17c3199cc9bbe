/*
 * pgshim_runtime.c -- bodies for the server functions ext/shim/pgshim.h declares, so that the glue of ext/ RUNS:
 *
 *   memory     palloc & co over contexts that can be reset; reset callbacks fire first (utils/mmgr/mcxt.c's order)
 *   errors     ereport(ERROR) longjmps to the innermost PG_TRY, or to the top-level handler of the process, which
 *              aborts the "transaction": buffer pins and locks released, the query context reset
 *   buffers    ReadBufferExtended / LockBuffer / page accessors over relations that are arrays of 8 KB pages in the real
 *              on-disk layout (PageHeaderData, ItemIdData, special space), kept in a shared mapping; pins are counted
 *   shmem      ShmemInitStruct / named LWLock tranches / pg_atomic / latches (futexes) in that same mapping
 *   processes  a postmaster that forks backends and, on RegisterDynamicBackgroundWorker, background workers; proc_exit
 *              runs before_shmem_exit and on_proc_exit callbacks
 *   pgvector   the few functions of the extension itself the glue calls (IvfflatGetMetaPageInfo, type info, optional
 *              support procs, HnswInitElementFromBlock ...), answered from the emulated catalog
 *
 * TEST INFRASTRUCTURE ONLY (tests/test_ext_runtime_*.py): it is neither PostgreSQL nor part of the product.  Where a
 * behaviour of the server matters to the glue it is cited; everything else is the simplest thing that is correct.
 */
#define _GNU_SOURCE
#include "pgshim_runtime.h"

#include "hnsw.h"
#include "ivfflat.h"

#include <errno.h>
#include <limits.h>
#include <math.h>
#include <linux/futex.h>
#include <signal.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <sys/mman.h>
#include <sys/syscall.h>
#include <sys/time.h>
#include <sys/wait.h>
#include <time.h>
#include <unistd.h>

/* ------------------------------------------------------------------------------------------------ shared state */
struct Latch
{
	uint32		is_set;
};

#define SHIM_LOCK_STRIPES 64
typedef struct ShimRel
{
	Oid			oid;
	ShimOpclass opc;
	int			dimensions;
	size_t		pages_off;
	uint32		nblocks;
	uint32		cap_blocks;
	uint32		lock;			/* 0x80000000: a writer holds or waits for the relation */
	int			reloptions[2];	/* lists | m, ef_construction (0: the access method's default) */
	/* readers by stripe (a process's pid picks its stripe), a cache line each: PostgreSQL's content locks are per BUFFER,
	 * so backends scanning one index do not share a lock word; one reader count per relation would put every page lock
	 * of every backend on ONE line and bound a many-process scan timing (oracle/ref_scan_bench.c) by that line */
	struct
	{
		uint32		n;
		uint32		pad[15];
	}			readers[SHIM_LOCK_STRIPES];
}			ShimRel;

typedef struct ShimBgw
{
	uint32		state;			/* 0 free, 1 requested, 2 running */
	char		fn[BGW_MAXLEN];
	Datum		arg;
	int			pid;
}			ShimBgw;

#define SHIM_MAX_STRUCTS 8
#define SHIM_MAX_BGW 8
#define SHIM_INVAL_RING 64

typedef struct ShimShared
{
	uint64		magic;
	int			postmaster_pid;
	struct
	{
		char		name[64];
		size_t		off,
					size;
	}			structs[SHIM_MAX_STRUCTS];
	int			nstructs;
	size_t		shmem_off,
				shmem_end;
	LWLock		addin_lock;
	LWLockPadded tranche[4];
	struct Latch latches[SHIM_MAX_PROCS];
	uint32		next_proc;
	ShimBgw		bgw[SHIM_MAX_BGW];
	uint32		bgw_kick;
	ShimRel		rels[SHIM_MAX_RELS];
	uint32		catalog_lock;
	size_t		page_off,
				page_end,
				page_used;
	size_t		arena_off,
				arena_bytes;
	/* a pool the stand-in's dynamic shared memory segments (parallel CREATE INDEX) and the tests' shared data are carved
	 * from: part of the postmaster's mapping, so at the SAME address in every process; never given back */
	size_t		dsm_off,
				dsm_bytes;
	uint64		dsm_used;
	void	   *shared_slots[8];
	/* shared invalidation queue (sinval): relcache invalidations reach every process at its next
	 * AcceptInvalidationMessages / transaction start */
	uint64		inval_seq;
	Oid			inval_relid[SHIM_INVAL_RING];
	/* test knob: every ReadBufferExtended of a background worker sleeps this long (a cold, large index) */
	uint32		bgw_read_delay_us;
}			ShimShared;

static ShimShared * S = NULL;
static char *Sbase = NULL;

/* ------------------------------------------------------------------------------------------------ process state */
Oid			MyDatabaseId = 5;
int			MyProcPid = 0;
int			ParallelWorkerNumber = -1;	/* access/parallel.h: -1 in the leader, 0 .. n-1 in its workers (set by ParallelWorkerMain) */
bool		process_shared_preload_libraries_in_progress = false;
shmem_request_hook_type shmem_request_hook = NULL;
shmem_startup_hook_type shmem_startup_hook = NULL;
LWLock	   *AddinShmemInitLock = NULL;
Latch	   *MyLatch = NULL;
sigjmp_buf *PG_exception_stack = NULL;
int			hnsw_ef_search = 40;
int			hnsw_iterative_scan = 0;	/* HNSW_ITERATIVE_SCAN_OFF */

static char last_error[512];
static int	last_error_level = 0;
static volatile sig_atomic_t proc_die_pending = 0;
static int	cancel_countdown = -1;
static int	is_bgworker = 0;

double
shim_now(void)
{
	struct timespec ts;

	clock_gettime(CLOCK_MONOTONIC, &ts);
	return (double) ts.tv_sec + 1e-9 * (double) ts.tv_nsec;
}

static void
futex_wait(uint32 *word, uint32 seen, long ms)
{
	struct timespec rel = {ms / 1000, (ms % 1000) * 1000000L};

	syscall(SYS_futex, word, FUTEX_WAIT, seen, ms >= 0 ? &rel : NULL, NULL, 0);
}

static void
futex_wake(uint32 *word)
{
	syscall(SYS_futex, word, FUTEX_WAKE, INT_MAX, NULL, NULL, 0);
}

/* ------------------------------------------------------------------------------------------------ memory contexts */
typedef struct ShimChunk
{
	struct MemoryContextData *ctx;
	struct ShimChunk *prev,
			   *next;
	Size		size;
	uint64		guard;
}			ShimChunk;

struct MemoryContextData
{
	const char *name;
	ShimChunk  *chunks;
	MemoryContextCallback *callbacks;
	size_t		bytes;
	/* a generation context (utils/mmgr/generation.c: chunks laid out one behind the other, freed all at once): chunks
	 * are carved from ONE reserved address range, so a later allocation has a higher address -- which is what the
	 * reference's CompareCandidateDistances falls back on when two candidates are equally far (src/hnswutils.c:992-1010
	 * compares the elements' pointers): ties then go by insertion order, as they do inside a generation block */
	char	   *gen_base;
	size_t		gen_used,
				gen_cap;
};

static struct MemoryContextData top_context = {"TopMemoryContext", NULL, NULL, 0};
MemoryContext TopMemoryContext = &top_context;
MemoryContext CurrentMemoryContext = &top_context;

#define CHUNK_GUARD 0x70616c6c6f636b21ull
#define SHIM_MAX_ALLOC ((Size) 0x3fffffff)	/* MaxAllocSize, utils/memutils.h: 1 GB - 1 */

static void *
context_alloc(MemoryContext ctx, Size size, bool huge)
{
	ShimChunk  *c;

	/* palloc / repalloc refuse requests past MaxAllocSize (mcxt.c: "invalid memory alloc request size"); only the
	 * _huge / MCXT_ALLOC_HUGE forms take more */
	if (!huge && size > SHIM_MAX_ALLOC)
		ereport(ERROR, (errmsg("invalid memory alloc request size %zu", (size_t) size)));
	if (ctx->gen_base)
	{
		size_t		need = (sizeof(ShimChunk) + (size ? size : 1) + 15) & ~(size_t) 15;

		if (ctx->gen_used + need > ctx->gen_cap)
			ereport(ERROR, (errmsg("out of memory")));
		c = (ShimChunk *) (ctx->gen_base + ctx->gen_used);
		ctx->gen_used += need;
	}
	else
		c = malloc(sizeof(ShimChunk) + (size ? size : 1));
	if (c == NULL)
		ereport(ERROR, (errmsg("out of memory")));
	/* fresh memory is not zero in the server either: a fixed pattern (what a MEMORY_CONTEXT_CHECKING build's
	 * randomize_mem is for) -- whoever relies on palloc zeroing shows up, and bytes nobody sets (the `unused` field of a
	 * k-means center, src/vector.h:22, goes to the list page as it is) are the same in every build of an index */
	memset((char *) c + sizeof(ShimChunk), 0x7F, size);
	c->ctx = ctx;
	c->size = size;
	c->guard = CHUNK_GUARD;
	c->prev = NULL;
	c->next = ctx->chunks;
	if (ctx->chunks)
		ctx->chunks->prev = c;
	ctx->chunks = c;
	ctx->bytes += size;
	return (char *) c + sizeof(ShimChunk);
}

void *
MemoryContextAlloc(MemoryContext ctx, Size size)
{
	return context_alloc(ctx, size, false);
}

MemoryContext
MemoryContextSwitchTo(MemoryContext ctx)
{
	MemoryContext old = CurrentMemoryContext;

	CurrentMemoryContext = ctx;
	return old;
}

void *
MemoryContextAllocZero(MemoryContext ctx, Size size)
{
	void	   *p = MemoryContextAlloc(ctx, size);

	memset(p, 0, size);
	return p;
}

void *
palloc(Size size)
{
	return MemoryContextAlloc(CurrentMemoryContext, size);
}

void *
palloc0(Size size)
{
	return MemoryContextAllocZero(CurrentMemoryContext, size);
}

void *
palloc_extended(Size size, int flags)
{
	return context_alloc(CurrentMemoryContext, size, (flags & MCXT_ALLOC_HUGE) != 0);
}

static ShimChunk *
chunk_of(void *p)
{
	ShimChunk  *c = (ShimChunk *) ((char *) p - sizeof(ShimChunk));

	if (c->guard != CHUNK_GUARD)
	{
		fprintf(stderr, "pgshim: pfree/repalloc of a pointer palloc did not return (or freed twice)\n");
		abort();
	}
	return c;
}

void
pfree(void *p)
{
	ShimChunk  *c = chunk_of(p);

	if (c->prev)
		c->prev->next = c->next;
	else
		c->ctx->chunks = c->next;
	if (c->next)
		c->next->prev = c->prev;
	c->ctx->bytes -= c->size;
	c->guard = 0;
	memset(p, 0xDE, c->size);	/* whoever still reads it reads garbage */
	if (c->ctx->gen_base == NULL)
		free(c);				/* (a generation context gives its range back at reset) */
}

static void *
context_realloc(void *p, Size size, bool huge)
{
	ShimChunk  *c = chunk_of(p);
	void	   *n = context_alloc(c->ctx, size, huge);

	memcpy(n, p, c->size < size ? c->size : size);
	pfree(p);
	return n;
}

void *
repalloc(void *p, Size size)
{
	return context_realloc(p, size, false);
}

void *
repalloc_huge(void *p, Size size)
{
	return context_realloc(p, size, true);
}

void
MemoryContextRegisterResetCallback(MemoryContext ctx, MemoryContextCallback *cb)
{
	cb->next = ctx->callbacks;
	ctx->callbacks = cb;
}

void
shim_context_reset(MemoryContext ctx)
{
	/* callbacks first, then the memory (MemoryContextResetOnly -> MemoryContextCallResetCallbacks) */
	while (ctx->callbacks)
	{
		MemoryContextCallback *cb = ctx->callbacks;

		ctx->callbacks = cb->next;
		cb->func(cb->arg);
	}
	while (ctx->chunks)
		pfree((char *) ctx->chunks + sizeof(ShimChunk));
	if (ctx->gen_base)
	{
		madvise(ctx->gen_base, ctx->gen_used, MADV_DONTNEED);
		ctx->gen_used = 0;
	}
}

size_t
shim_context_bytes(MemoryContext ctx)
{
	return ctx->bytes;
}

static MemoryContext query_context = NULL;

MemoryContext
shim_query_context_begin(void)
{
	MemoryContext ctx = calloc(1, sizeof(struct MemoryContextData));

	ctx->name = "query";
	query_context = ctx;
	CurrentMemoryContext = ctx;
	return ctx;
}

/* a context of its own for whoever asks (AllocSetContextCreate of tests/c/pgshim_ref_runtime.c) */
MemoryContext
shim_context_create(void)
{
	MemoryContext ctx = calloc(1, sizeof(struct MemoryContextData));

	ctx->name = "child";
	return ctx;
}

MemoryContext
shim_context_create_generation(void)
{
	MemoryContext ctx = shim_context_create();

	ctx->name = "generation";
	ctx->gen_cap = (size_t) 4 << 30;	/* address space, not memory */
	ctx->gen_base = mmap(NULL, ctx->gen_cap, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
	if (ctx->gen_base == MAP_FAILED)
	{
		perror("pgshim: mmap (generation context)");
		abort();
	}
	return ctx;
}

void
shim_context_delete(MemoryContext ctx)
{
	shim_context_reset(ctx);
	if (CurrentMemoryContext == ctx)
		CurrentMemoryContext = TopMemoryContext;
	if (ctx->gen_base)
		munmap(ctx->gen_base, ctx->gen_cap);
	free(ctx);
}

void
shim_query_context_end(MemoryContext ctx)
{
	shim_context_reset(ctx);
	if (query_context == ctx)
		query_context = NULL;
	CurrentMemoryContext = TopMemoryContext;
	free(ctx);
}

/* ------------------------------------------------------------------------------------------------ errors */
static int
pending_level(int level)
{
	last_error_level = level;
	return level;
}

int
pgshim_errmsg(const char *fmt,...)
{
	va_list		ap;

	va_start(ap, fmt);
	vsnprintf(last_error, sizeof(last_error), fmt, ap);
	va_end(ap);
	return 0;
}

/* errdetail / errhint of the reference's ereport calls: checked for their format, not kept (the primary message is what
 * shim_last_error and shim_notices_raised report) */
int
pgshim_errmore(const char *fmt,...)
{
	(void) fmt;
	return 0;
}

/* the last messages below ERROR, for tests that expect a NOTICE or a WARNING */
#define SHIM_NOTICE_RING 32
static char notice_ring[SHIM_NOTICE_RING][160];
static int	notices_total = 0;

int
shim_notices_raised(const char *prefix)
{
	int			n = 0;

	for (int i = 0; i < (notices_total < SHIM_NOTICE_RING ? notices_total : SHIM_NOTICE_RING); i++)
		n += strncmp(notice_ring[i], prefix, strlen(prefix)) == 0;
	return n;
}

static void throw_error(void) __attribute__((noreturn));
static void
throw_error(void)
{
	if (PG_exception_stack != NULL)
		siglongjmp(*PG_exception_stack, 1);
	fprintf(stderr, "pgshim: ERROR with no handler installed: %s\n", last_error);
	abort();
}

void
pgshim_ereport(int level, int dummy)
{
	(void) dummy;
	pending_level(level);
	if (level >= ERROR)
		throw_error();
	snprintf(notice_ring[notices_total++ % SHIM_NOTICE_RING], sizeof(notice_ring[0]), "%.159s", last_error);
	fprintf(stderr, "%s:  %s\n", level == WARNING ? "WARNING" : (level == 18 ? "NOTICE" : "LOG"), last_error);
}

void
pgshim_elog(int level, const char *fmt,...)
{
	va_list		ap;

	va_start(ap, fmt);
	vsnprintf(last_error, sizeof(last_error), fmt, ap);
	va_end(ap);
	pgshim_ereport(level, 0);
}

void
pgshim_rethrow(void)
{
	throw_error();
}

void
EmitErrorReport(void)
{
	fprintf(stderr, "ERROR:  %s\n", last_error);
}

void
FlushErrorState(void)
{
	last_error_level = 0;
}

const char *
shim_last_error(void)
{
	return last_error;
}

/* ------------------------------------------------------------------------------------------------ exit callbacks */
typedef struct ExitCb
{
	pg_on_exit_callback fn;
	Datum		arg;
}			ExitCb;
static ExitCb on_exit_cbs[16],
			before_exit_cbs[16];
static int	n_on_exit = 0,
			n_before_exit = 0;

void
on_proc_exit(pg_on_exit_callback function, Datum arg)
{
	on_exit_cbs[n_on_exit].fn = function;
	on_exit_cbs[n_on_exit++].arg = arg;
}

void
before_shmem_exit(pg_on_exit_callback function, Datum arg)
{
	before_exit_cbs[n_before_exit].fn = function;
	before_exit_cbs[n_before_exit++].arg = arg;
}

void
shim_run_proc_exit(int code)
{
	/* shmem_exit, then proc_exit's own list, each last-registered first (storage/ipc/ipc.c) */
	while (n_before_exit > 0)
	{
		n_before_exit--;
		before_exit_cbs[n_before_exit].fn(code, before_exit_cbs[n_before_exit].arg);
	}
	while (n_on_exit > 0)
	{
		n_on_exit--;
		on_exit_cbs[n_on_exit].fn(code, on_exit_cbs[n_on_exit].arg);
	}
}

void
proc_exit(int code)
{
	if (is_bgworker)
		fprintf(stderr, "LOG:  background worker %d exits with code %d%s%s\n", (int) getpid(), code, code ? ": " : "", code ? last_error : "");
	shim_run_proc_exit(code);
	fflush(NULL);
	_exit(code);
}

/* ------------------------------------------------------------------------------------------------ interrupts */
void
shim_cancel_after(int after_checks)
{
	cancel_countdown = after_checks;
}

volatile sig_atomic_t InterruptPending = 0;
volatile sig_atomic_t QueryCancelPending = 0;
volatile sig_atomic_t ProcDiePending = 0;

void
pgshim_check_interrupts(void)
{
	if (proc_die_pending)
	{
		/* FATAL: "terminating background worker due to administrator command": the exit callbacks run */
		fprintf(stderr, "FATAL:  terminating process %d due to administrator command\n", (int) getpid());
		proc_exit(1);
	}
	if (cancel_countdown >= 0 && cancel_countdown-- == 0)
		ereport(ERROR, (errmsg("canceling statement due to user request")));
}

static void
handle_sigterm(int sig)
{
	(void) sig;
	proc_die_pending = 1;
	InterruptPending = 1;
	ProcDiePending = 1;
	if (MyLatch)
	{
		__atomic_store_n(&MyLatch->is_set, 1, __ATOMIC_RELEASE);
		futex_wake(&MyLatch->is_set);
	}
}

void
pg_usleep(long microsec)
{
	usleep((useconds_t) microsec);
}

/* ------------------------------------------------------------------------------------------------ GUCs */
typedef struct Guc
{
	const char *name;
	bool	   *b;
	int		   *i;
}			Guc;
static Guc	gucs[16];
static int	ngucs = 0;

void
DefineCustomBoolVariable(const char *name, const char *short_desc, const char *long_desc, bool *valueAddr, bool bootValue,
						 GucContext context, int flags, void *check, void *assign, void *show)
{
	(void) short_desc, (void) long_desc, (void) context, (void) flags, (void) check, (void) assign, (void) show;
	*valueAddr = bootValue;
	gucs[ngucs].name = name;
	gucs[ngucs].b = valueAddr;
	gucs[ngucs++].i = NULL;
}

void
DefineCustomIntVariable(const char *name, const char *short_desc, const char *long_desc, int *valueAddr, int bootValue,
						int minValue, int maxValue, GucContext context, int flags, void *check, void *assign, void *show)
{
	(void) short_desc, (void) long_desc, (void) minValue, (void) maxValue, (void) context, (void) flags, (void) check,
		(void) assign, (void) show;
	*valueAddr = bootValue;
	gucs[ngucs].name = name;
	gucs[ngucs].b = NULL;
	gucs[ngucs++].i = valueAddr;
}

/* the GUC values of this process, for a parallel worker to start with (SerializeGUCState / RestoreGUCState): the
 * variables were registered in the postmaster, so every process has them in the same order */
int
shim_guc_snapshot(int *out, int cap)
{
	for (int g = 0; g < ngucs && g < cap; g++)
		out[g] = gucs[g].b ? (int) *gucs[g].b : *gucs[g].i;
	return ngucs < cap ? ngucs : cap;
}

void
shim_guc_restore(const int *in, int n)
{
	for (int g = 0; g < ngucs && g < n; g++)
	{
		if (gucs[g].b)
			*gucs[g].b = in[g] != 0;
		else
			*gucs[g].i = in[g];
	}
}

void
shim_set_guc_bool(const char *name, bool value)
{
	for (int g = 0; g < ngucs; g++)
		if (strcmp(gucs[g].name, name) == 0 && gucs[g].b)
		{
			*gucs[g].b = value;
			return;
		}
	fprintf(stderr, "pgshim: no bool GUC %s\n", name);
	abort();
}

void
shim_set_guc_int(const char *name, int value)
{
	for (int g = 0; g < ngucs; g++)
		if (strcmp(gucs[g].name, name) == 0 && gucs[g].i)
		{
			*gucs[g].i = value;
			return;
		}
	fprintf(stderr, "pgshim: no int GUC %s\n", name);
	abort();
}

/* ------------------------------------------------------------------------------------------------ shared memory */
static size_t requested_shmem = 0;

void
RequestAddinShmemSpace(Size size)
{
	requested_shmem += size;
}

void
RequestNamedLWLockTranche(const char *tranche_name, int num_lwlocks)
{
	(void) tranche_name, (void) num_lwlocks;
}

LWLockPadded *
GetNamedLWLockTranche(const char *tranche_name)
{
	(void) tranche_name;
	return S->tranche;
}

void *
ShmemInitStruct(const char *name, Size size, bool *foundPtr)
{
	for (int i = 0; i < S->nstructs; i++)
		if (strcmp(S->structs[i].name, name) == 0)
		{
			*foundPtr = true;
			return Sbase + S->structs[i].off;
		}
	if (S->nstructs == SHIM_MAX_STRUCTS || S->shmem_off + size > S->shmem_end)
		ereport(ERROR, (errmsg("out of shared memory")));
	snprintf(S->structs[S->nstructs].name, sizeof(S->structs[0].name), "%s", name);
	S->structs[S->nstructs].off = S->shmem_off;
	S->structs[S->nstructs].size = size;
	S->nstructs++;
	S->shmem_off += (size + 127) & ~(size_t) 127;
	*foundPtr = false;
	return Sbase + S->structs[S->nstructs - 1].off;
}

/* a reader-writer lock over the state word: readers count, 0x80000000 = a writer holds it.  (No queue, no fairness:
 * writers can be kept waiting by a stream of readers -- the tests' lock traffic is light.) */
#define LW_WRITER 0x80000000u

bool
LWLockAcquire(LWLock *lock, LWLockMode mode)
{
	for (;;)
	{
		uint32		v = __atomic_load_n(&lock->state, __ATOMIC_RELAXED);

		if (mode == LW_EXCLUSIVE ? v == 0 : !(v & LW_WRITER))
		{
			if (__atomic_compare_exchange_n(&lock->state, &v, mode == LW_EXCLUSIVE ? LW_WRITER : v + 1, 0, __ATOMIC_ACQUIRE,
											__ATOMIC_RELAXED))
				return true;
			continue;
		}
		futex_wait(&lock->state, v, 1);
	}
}

void
LWLockRelease(LWLock *lock)
{
	if (__atomic_load_n(&lock->state, __ATOMIC_RELAXED) & LW_WRITER)
		__atomic_store_n(&lock->state, 0u, __ATOMIC_RELEASE);
	else
		__atomic_sub_fetch(&lock->state, 1u, __ATOMIC_RELEASE);
	futex_wake(&lock->state);
}

void
pg_atomic_init_u64(volatile pg_atomic_uint64 *ptr, uint64 val)
{
	ptr->value = val;
}

uint64
pg_atomic_read_u64(volatile pg_atomic_uint64 *ptr)
{
	return __atomic_load_n(&ptr->value, __ATOMIC_SEQ_CST);
}

void
pg_atomic_write_u64(volatile pg_atomic_uint64 *ptr, uint64 val)
{
	__atomic_store_n(&ptr->value, val, __ATOMIC_RELAXED);
}

bool
pg_atomic_compare_exchange_u64(volatile pg_atomic_uint64 *ptr, uint64 *expected, uint64 newval)
{
	return __atomic_compare_exchange_n(&ptr->value, expected, newval, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST);
}

uint64
pg_atomic_fetch_add_u64(volatile pg_atomic_uint64 *ptr, int64 add_)
{
	return __atomic_fetch_add(&ptr->value, (uint64) add_, __ATOMIC_SEQ_CST);
}

void
pg_atomic_init_u32(volatile pg_atomic_uint32 *ptr, uint32 val)
{
	ptr->value = val;
}

uint32
pg_atomic_read_u32(volatile pg_atomic_uint32 *ptr)
{
	return __atomic_load_n(&ptr->value, __ATOMIC_SEQ_CST);
}

void
pg_atomic_write_u32(volatile pg_atomic_uint32 *ptr, uint32 val)
{
	__atomic_store_n(&ptr->value, val, __ATOMIC_SEQ_CST);
}

bool
pg_atomic_compare_exchange_u32(volatile pg_atomic_uint32 *ptr, uint32 *expected, uint32 newval)
{
	return __atomic_compare_exchange_n(&ptr->value, expected, newval, 0, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST);
}

/* ------------------------------------------------------------------------------------------------ latches, time */
int
WaitLatch(Latch *latch, int wakeEvents, long timeout, uint32 wait_event_info)
{
	double		until = shim_now() + (double) timeout / 1000.0;

	(void) wait_event_info;
	for (;;)
	{
		double		left;

		if ((wakeEvents & WL_LATCH_SET) && __atomic_load_n(&latch->is_set, __ATOMIC_ACQUIRE))
			return WL_LATCH_SET;
		if ((wakeEvents & WL_EXIT_ON_PM_DEATH) && getppid() != S->postmaster_pid)
			proc_exit(1);
		if (!(wakeEvents & WL_TIMEOUT))
		{
			futex_wait(&latch->is_set, 0, 100);
			continue;
		}
		left = until - shim_now();
		if (left <= 0)
			return WL_TIMEOUT;
		futex_wait(&latch->is_set, 0, (long) (left * 1000.0) + 1);
	}
}

void
SetLatch(Latch *latch)
{
	__atomic_store_n(&latch->is_set, 1, __ATOMIC_RELEASE);
	futex_wake(&latch->is_set);
}

void
ResetLatch(Latch *latch)
{
	__atomic_store_n(&latch->is_set, 0, __ATOMIC_RELEASE);
}

TimestampTz
GetCurrentTimestamp(void)
{
	struct timeval tv;

	gettimeofday(&tv, NULL);
	return (TimestampTz) tv.tv_sec * 1000000 + tv.tv_usec;
}

bool
TimestampDifferenceExceeds(TimestampTz start_time, TimestampTz stop_time, int msec)
{
	return stop_time - start_time >= (int64) msec * 1000;
}

Datum
float8_as_datum(double x)
{
	Datum		d;

	memcpy(&d, &x, sizeof(d));
	return d;
}

/* ------------------------------------------------------------------------------------------------ the catalog */
static struct TupleDescData rel_descs[SHIM_MAX_RELS];
static struct RelationData rel_objs[SHIM_MAX_RELS];
/* IvfflatOptions / HnswOptions (src/ivfflat.h:138-142, src/hnsw.h:225-230): a varlena header and the ints behind it */
static struct
{
	int32		vl_len_;
	int			a,
				b;
}			rel_options[SHIM_MAX_RELS];
/* (reference-linked builds) the heap stand-in of pgshim_ref_runtime.c answers for its own relation */
BlockNumber (*shim_heap_blocks_hook) (void) = NULL;

static void
spin_lock(uint32 *w)
{
	for (;;)
	{
		uint32		zero = 0;

		if (__atomic_compare_exchange_n(w, &zero, 1u, 0, __ATOMIC_ACQUIRE, __ATOMIC_RELAXED))
			return;
		usleep(20);
	}
}

static void
spin_unlock(uint32 *w)
{
	__atomic_store_n(w, 0u, __ATOMIC_RELEASE);
}

static ShimRel *
find_rel(Oid oid)
{
	for (int i = 0; i < SHIM_MAX_RELS; i++)
		if (S->rels[i].oid == oid && oid != 0)
			return &S->rels[i];
	return NULL;
}

static uint32 *
rel_my_readers(ShimRel * r)
{
	return &r->readers[(uint32) MyProcPid % SHIM_LOCK_STRIPES].n;
}

static void
rel_lock_shared(ShimRel * r)
{
	uint32	   *mine = rel_my_readers(r);

	for (;;)
	{
		__atomic_add_fetch(mine, 1, __ATOMIC_SEQ_CST);
		if (!(__atomic_load_n(&r->lock, __ATOMIC_SEQ_CST) & 0x80000000u))
			return;
		__atomic_sub_fetch(mine, 1, __ATOMIC_SEQ_CST);	/* a writer is in or waiting: out of its way */
		while (__atomic_load_n(&r->lock, __ATOMIC_ACQUIRE) & 0x80000000u)
			usleep(20);
	}
}

static void
rel_unlock_shared(ShimRel * r)
{
	__atomic_sub_fetch(rel_my_readers(r), 1, __ATOMIC_RELEASE);
}

static void
rel_lock_exclusive(ShimRel * r)
{
	for (;;)
	{
		uint32		zero = 0;

		if (__atomic_compare_exchange_n(&r->lock, &zero, 0x80000000u, 0, __ATOMIC_SEQ_CST, __ATOMIC_RELAXED))
			break;
		usleep(50);
	}
	for (int i = 0; i < SHIM_LOCK_STRIPES; i++)	/* the readers that were in before the flag went up */
		while (__atomic_load_n(&r->readers[i].n, __ATOMIC_SEQ_CST) != 0)
			usleep(20);
}

static void
rel_unlock_exclusive(ShimRel * r)
{
	__atomic_store_n(&r->lock, 0u, __ATOMIC_RELEASE);
}

void
shim_create_relation(Oid oid, const ShimOpclass * opclass, const void *pages, uint32_t nblocks, int dimensions)
{
	ShimRel    *r = NULL;
	uint32		cap = nblocks * 2 + 4096;	/* room for the pages a build or inserts bring later */

	spin_lock(&S->catalog_lock);
	/* the storage of a dropped relation is used again when it is large enough (DROP INDEX / CREATE INDEX in a loop) */
	for (int i = 0; i < SHIM_MAX_RELS && r == NULL; i++)
		if (S->rels[i].oid == 0 && S->rels[i].cap_blocks >= cap)
			r = &S->rels[i];
	if (r != NULL)
		cap = r->cap_blocks;
	else
	{
		for (int i = 0; i < SHIM_MAX_RELS && r == NULL; i++)
			if (S->rels[i].oid == 0 && S->rels[i].cap_blocks == 0)
				r = &S->rels[i];
		if (r == NULL || S->page_used + (size_t) cap * SHIM_BLCKSZ > S->page_end - S->page_off)
		{
			fprintf(stderr, "pgshim: catalog or page store full\n");
			abort();
		}
		r->pages_off = S->page_off + S->page_used;
		S->page_used += (size_t) cap * SHIM_BLCKSZ;
	}
	r->opc = *opclass;
	r->dimensions = dimensions;
	r->cap_blocks = cap;
	r->nblocks = nblocks;
	r->lock = 0;
	memset(r->readers, 0, sizeof(r->readers));
	r->reloptions[0] = r->reloptions[1] = 0;
	memcpy(Sbase + r->pages_off, pages, (size_t) nblocks * SHIM_BLCKSZ);
	__atomic_store_n(&r->oid, oid, __ATOMIC_RELEASE);
	spin_unlock(&S->catalog_lock);
}

void
shim_replace_pages(Oid oid, const void *pages, uint32_t nblocks)
{
	ShimRel    *r = find_rel(oid);

	if (r == NULL || nblocks > r->cap_blocks)
	{
		fprintf(stderr, "pgshim: shim_replace_pages: unknown relation or too many blocks\n");
		abort();
	}
	rel_lock_exclusive(r);
	memcpy(Sbase + r->pages_off, pages, (size_t) nblocks * SHIM_BLCKSZ);
	r->nblocks = nblocks;
	rel_unlock_exclusive(r);
}

const void *
shim_relation_pages(Oid oid, uint32_t *nblocks)
{
	ShimRel    *r = find_rel(oid);

	if (r == NULL)
		return NULL;
	*nblocks = r->nblocks;
	return Sbase + r->pages_off;
}

void
shim_drop_relation(Oid oid)
{
	ShimRel    *r = find_rel(oid);

	if (r)
		__atomic_store_n(&r->oid, 0, __ATOMIC_RELEASE);
	shim_broadcast_relcache_invalidate(oid);	/* DROP INDEX reaches every backend as a relcache invalidation */
}

Relation
shim_open_relation(Oid oid)
{
	ShimRel    *r = find_rel(oid);

	if (r == NULL)
		return NULL;
	rel_objs[r - S->rels].rd_id = oid;
	rel_descs[r - S->rels].natts = 1;
	rel_descs[r - S->rels].types[0] = SHIM_VECTOR_TYPE_OID;
	rel_descs[r - S->rels].typmods[0] = r->dimensions;
	rel_objs[r - S->rels].rd_att = &rel_descs[r - S->rels];
	if (r->reloptions[0] != 0)
	{
		/* WITH (lists = ..) / WITH (m = .., ef_construction = ..); rd_options stays NULL for the defaults */
		rel_options[r - S->rels].a = r->reloptions[0];
		rel_options[r - S->rels].b = r->reloptions[1];
		rel_objs[r - S->rels].rd_options = (struct varlena *) &rel_options[r - S->rels];
	}
	else
		rel_objs[r - S->rels].rd_options = NULL;
	{
		static Oid	no_collation[1] = {InvalidOid};

		rel_objs[r - S->rels].rd_indcollation = no_collation;	/* (the reference's beginscan reads rd_indcollation[0]) */
	}
	return &rel_objs[r - S->rels];
}

void
shim_set_reloptions(Oid oid, int a, int b)
{
	ShimRel    *r = find_rel(oid);

	if (r == NULL)
	{
		fprintf(stderr, "pgshim: shim_set_reloptions: unknown relation\n");
		abort();
	}
	r->reloptions[0] = a;
	r->reloptions[1] = b;
}

static ShimRel *
rel_of(Relation rel)
{
	ShimRel    *r = find_rel(rel->rd_id);

	if (r == NULL)
		ereport(ERROR, (errmsg("could not open relation with OID %u", rel->rd_id)));
	return r;
}

Relation
try_index_open(Oid relationId, LOCKMODE lockmode)
{
	(void) lockmode;
	return shim_open_relation(relationId);
}

void
index_close(Relation relation, LOCKMODE lockmode)
{
	(void) relation, (void) lockmode;
}

typedef struct RelcacheCb
{
	RelcacheCallbackFunction fn;
	Datum		arg;
}			RelcacheCb;
static RelcacheCb relcache_cbs[8];
static int	n_relcache_cbs = 0;

void
CacheRegisterRelcacheCallback(RelcacheCallbackFunction func, Datum arg)
{
	relcache_cbs[n_relcache_cbs].fn = func;
	relcache_cbs[n_relcache_cbs++].arg = arg;
}

void
shim_relcache_invalidate(Oid relid)
{
	for (int i = 0; i < n_relcache_cbs; i++)
		relcache_cbs[i].fn(relcache_cbs[i].arg, relid);
}

/* the shared invalidation queue: a message per relation, read by every process at its own pace; a reader that fell a
 * whole ring behind gets "everything" (relid 0), like sinval's reset */
static uint64 inval_seen = 0;

void
shim_broadcast_relcache_invalidate(Oid relid)
{
	uint64		at = __atomic_fetch_add(&S->inval_seq, 1, __ATOMIC_SEQ_CST);

	__atomic_store_n(&S->inval_relid[at % SHIM_INVAL_RING], relid, __ATOMIC_RELEASE);
}

void
AcceptInvalidationMessages(void)
{
	uint64		now = __atomic_load_n(&S->inval_seq, __ATOMIC_ACQUIRE);

	if (now == inval_seen)
		return;
	if (now - inval_seen > SHIM_INVAL_RING / 2)
		shim_relcache_invalidate(0);
	else
		for (uint64 i = inval_seen; i < now; i++)
			shim_relcache_invalidate(__atomic_load_n(&S->inval_relid[i % SHIM_INVAL_RING], __ATOMIC_ACQUIRE));
	inval_seen = now;
}

void
shim_set_bgworker_read_delay_us(uint32_t us)
{
	__atomic_store_n(&S->bgw_read_delay_us, us, __ATOMIC_RELEASE);
}

/* ------------------------------------------------------------------------------------------------ buffers and pages */
#define MAX_PINS 64
static struct
{
	Buffer		buf;
	bool		locked;
}			pins[MAX_PINS];
static int	npins = 0;

#define BUF_REL(b) (((b) - 1) >> 24)
#define BUF_BLK(b) ((uint32) (((b) - 1) & 0xFFFFFF))

static long buffer_reads;		/* pages this process asked the buffer manager for */

long
shim_buffer_reads(void)
{
	return buffer_reads;
}

Buffer
ReadBufferExtended(Relation reln, ForkNumber forkNum, BlockNumber blockNum, ReadBufferMode mode, BufferAccessStrategy strategy)
{
	ShimRel    *r = rel_of(reln);
	Buffer		b;

	(void) forkNum, (void) mode, (void) strategy;
	if (is_bgworker && __atomic_load_n(&S->bgw_read_delay_us, __ATOMIC_ACQUIRE) > 0)
		usleep(__atomic_load_n(&S->bgw_read_delay_us, __ATOMIC_ACQUIRE));
	if (blockNum == 0xFFFFFFFFu)	/* P_NEW: the relation grows by one zeroed page (under the caller's extension lock) */
	{
		uint32		nb = __atomic_load_n(&r->nblocks, __ATOMIC_ACQUIRE);

		if (nb >= r->cap_blocks)
			ereport(ERROR, (errmsg("could not extend relation %u: page store full", reln->rd_id)));
		memset(Sbase + r->pages_off + (size_t) nb * SHIM_BLCKSZ, 0, SHIM_BLCKSZ);
		__atomic_store_n(&r->nblocks, nb + 1, __ATOMIC_RELEASE);
		blockNum = nb;
	}
	if (blockNum >= __atomic_load_n(&r->nblocks, __ATOMIC_ACQUIRE))
		ereport(ERROR, (errmsg("could not read block %u of relation %u: read only 0 of 8192 bytes", blockNum, reln->rd_id)));
	if (npins == MAX_PINS)
		ereport(ERROR, (errmsg("too many buffers pinned")));
	b = (Buffer) (((int) (r - S->rels) << 24) | (int) blockNum) + 1;
	buffer_reads++;
	pins[npins].buf = b;
	pins[npins++].locked = false;
	return b;
}

static int
pin_index(Buffer buffer)
{
	for (int i = npins - 1; i >= 0; i--)
		if (pins[i].buf == buffer)
			return i;
	fprintf(stderr, "pgshim: buffer %d is not pinned by this process\n", buffer);
	abort();
}

#ifndef BUFFER_LOCK_UNLOCK
#define BUFFER_LOCK_UNLOCK 0	/* storage/bufmgr.h */
#endif

void
LockBuffer(Buffer buffer, int mode)
{
	int			i = pin_index(buffer);

	if (mode == BUFFER_LOCK_UNLOCK)
	{
		/* (src/hnswbuild.c:138: a fresh page is unlocked and locked again before it is filled) */
		if (pins[i].locked)
			rel_unlock_shared(&S->rels[BUF_REL(buffer)]);
		pins[i].locked = false;
		return;
	}
	if (pins[i].locked)
		return;					/* (shared and exclusive are one lock here) */
	rel_lock_shared(&S->rels[BUF_REL(buffer)]);
	pins[i].locked = true;
}

void
UnlockReleaseBuffer(Buffer buffer)
{
	int			i = pin_index(buffer);

	if (pins[i].locked)
		rel_unlock_shared(&S->rels[BUF_REL(buffer)]);
	pins[i] = pins[--npins];
}

static void
release_all_buffers(void)
{
	/* what the resource owner does at transaction abort */
	while (npins > 0)
		UnlockReleaseBuffer(pins[npins - 1].buf);
}

int
shim_pinned_buffers(void)
{
	return npins;
}

Page
BufferGetPage(Buffer buffer)
{
	(void) pin_index(buffer);
	return Sbase + S->rels[BUF_REL(buffer)].pages_off + (size_t) BUF_BLK(buffer) * SHIM_BLCKSZ;
}

BlockNumber
BufferGetBlockNumber(Buffer buffer)
{
	(void) pin_index(buffer);
	return BUF_BLK(buffer);
}

BlockNumber
RelationGetNumberOfBlocks(Relation reln)
{
	if (reln->rd_id == SHIM_HEAP_OID && shim_heap_blocks_hook)
		return shim_heap_blocks_hook();
	return __atomic_load_n(&rel_of(reln)->nblocks, __ATOMIC_ACQUIRE);
}

/* PageHeaderData: pd_lsn 8, pd_checksum 2, pd_flags 2, pd_lower 2, pd_upper 2, pd_special 2, pd_pagesize_version 2,
 * pd_prune_xid 4 = 24 bytes, then ItemIdData[] (lp_off:15, lp_flags:2, lp_len:15) -- storage/bufpage.h, storage/itemid.h */
#define PAGE_HEADER_SIZE 24

static uint16
page_u16(Page page, int off)
{
	uint16		v;

	memcpy(&v, page + off, 2);
	return v;
}

OffsetNumber
PageGetMaxOffsetNumber(Page page)
{
	uint16		lower = page_u16(page, 12);

	return lower <= PAGE_HEADER_SIZE ? 0 : (OffsetNumber) ((lower - PAGE_HEADER_SIZE) / 4);
}

ItemId
PageGetItemId(Page page, OffsetNumber offsetNumber)
{
	return (ItemId) (page + PAGE_HEADER_SIZE + (size_t) (offsetNumber - 1) * 4);
}

Item
PageGetItem(Page page, ItemId itemId)
{
	uint32		lp;

	memcpy(&lp, itemId, 4);
	return page + (lp & 0x7FFFu);
}

char *
PageGetSpecialPointer(Page page)
{
	return page + page_u16(page, 16);
}

char *
PageGetContents(Page page)
{
	return page + PAGE_HEADER_SIZE;	/* MAXALIGN(SizeOfPageHeaderData) */
}

BlockNumber
ItemPointerGetBlockNumber(const ItemPointerData *p)
{
	return ((BlockNumber) p->ip_blkid.bi_hi << 16) | p->ip_blkid.bi_lo;
}

OffsetNumber
ItemPointerGetOffsetNumber(const ItemPointerData *p)
{
	return p->ip_posid;
}

/* the one attribute of an index tuple without NULLs: right behind the 8-byte header (a short varlena needs no
 * alignment padding, a 4-byte-header one sits MAXALIGNed there anyway) -- access/itup.h, access/tupmacs.h */
Datum
index_getattr(IndexTuple tup, int attnum, TupleDesc tupleDesc, bool *isnull)
{
	(void) attnum, (void) tupleDesc;
	*isnull = false;
	return PointerGetDatum((char *) tup + sizeof(IndexTupleData));
}

/*
 * varlena forms a Datum can arrive in (postgres.h / varatt.h, little endian):
 *   xxxxxx00  4-byte header, length << 2: plain -- returned as is
 *   xxxxxxx1  1-byte header, length << 1 | 1 (values under 127 bytes: what index_form_tuple and heap_form_tuple
 *             store) -- expanded into a palloc'd copy with a 4-byte header
 *   xxxxxx10  4-byte header of a COMPRESSED value; this runtime's stand-in for "toasted": header, uint32 raw size,
 *             then the payload uncompressed -- expanded into a palloc'd plain copy, like a detoast does
 */
struct varlena *
pg_detoast_datum(struct varlena *datum)
{
	const unsigned char *p = (const unsigned char *) datum;

	if (p[0] & 0x01)
	{
		Size		len = p[0] >> 1;	/* includes the header byte */
		struct varlena *out = palloc(len - 1 + 4);
		uint32		hdr = (uint32) (len - 1 + 4) << 2;

		memcpy(out, &hdr, 4);
		memcpy((char *) out + 4, p + 1, len - 1);
		return out;
	}
	if ((p[0] & 0x03) == 0x02)
	{
		uint32		raw;
		struct varlena *out;
		uint32		hdr;

		memcpy(&raw, p + 4, 4);
		out = palloc(raw + 4);
		hdr = (raw + 4) << 2;
		memcpy(out, &hdr, 4);
		memcpy((char *) out + 4, p + 8, raw);
		return out;
	}
	return datum;
}

/* ------------------------------------------------------------------------------------------------ lists */
List *
lappend(List *list, void *datum)
{
	if (list == NIL)
	{
		list = palloc0(sizeof(List));
		list->cap = 16;
		list->elems = palloc(sizeof(void *) * 16);
	}
	if (list->length == list->cap)
	{
		list->cap *= 2;
		list->elems = repalloc(list->elems, sizeof(void *) * (Size) list->cap);
	}
	list->elems[list->length++] = datum;
	return list;
}

int
shim_list_length(const List *l)
{
	return l ? l->length : 0;
}

void *
shim_list_nth(const List *l, int n)
{
	return l->elems[n];
}

/* ------------------------------------------------------------------------------------------------ transactions */
static MemoryContext xact_context = NULL;
static MemoryContext xact_saved = NULL;

void
StartTransactionCommand(void)
{
	AcceptInvalidationMessages();
	xact_context = calloc(1, sizeof(struct MemoryContextData));
	xact_context->name = "transaction";
	xact_saved = CurrentMemoryContext;
	CurrentMemoryContext = xact_context;
}

static void
end_transaction(void)
{
	release_all_buffers();
	if (xact_context)
	{
		CurrentMemoryContext = xact_saved ? xact_saved : TopMemoryContext;
		shim_context_reset(xact_context);
		free(xact_context);
		xact_context = NULL;
	}
}

static int	pin_leaks = 0;

void
CommitTransactionCommand(void)
{
	if (npins > 0)
	{
		fprintf(stderr, "WARNING:  buffer refcount leak: %d buffers still pinned at commit\n", npins);
		pin_leaks += npins;
	}
	end_transaction();
}

void
AbortCurrentTransaction(void)
{
	end_transaction();
}

int
shim_pin_leaks(void)
{
	return pin_leaks;
}

int
shim_run_toplevel(int (*fn) (void *), void *arg, int *result)
{
	sigjmp_buf	top;
	sigjmp_buf *saved = PG_exception_stack;

	if (sigsetjmp(top, 0) == 0)
	{
		PG_exception_stack = &top;
		*result = fn(arg);
		PG_exception_stack = saved;
		return 0;
	}
	/* PostgresMain's error recovery: report, abort the transaction (pins, locks), drop the query's memory */
	PG_exception_stack = saved;
	EmitErrorReport();
	AbortCurrentTransaction();
	if (query_context)
		shim_query_context_end(query_context);
	CurrentMemoryContext = TopMemoryContext;
	cancel_countdown = -1;
	return -1;
}

/* ------------------------------------------------------------------------------------------------ processes */
static struct
{
	char		name[BGW_MAXLEN];
	void		(*fn) (Datum);
}			bgw_fns[4];
static int	n_bgw_fns = 0;
static int	my_proc_slot = -1;

void
shim_register_bgworker_function(const char *name, void (*fn) (Datum))
{
	snprintf(bgw_fns[n_bgw_fns].name, BGW_MAXLEN, "%s", name);
	bgw_fns[n_bgw_fns++].fn = fn;
}

void
shim_postmaster_init(size_t page_store_bytes, size_t arena_bytes)
{
	size_t		shmem_bytes = 8u << 20;
	size_t		dsm_bytes = (size_t) 192 << 20;	/* (MAP_NORESERVE: only what is touched costs memory) */
	size_t		total = ((sizeof(ShimShared) + 4095) & ~(size_t) 4095) + shmem_bytes + page_store_bytes + arena_bytes + dsm_bytes;

	Sbase = mmap(NULL, total, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
	if (Sbase == MAP_FAILED)
	{
		perror("pgshim: mmap");
		abort();
	}
	S = (ShimShared *) Sbase;
	memset(S, 0, sizeof(ShimShared));
	S->magic = 0x70677368696dull;
	S->postmaster_pid = (int) getpid();
	S->shmem_off = (sizeof(ShimShared) + 4095) & ~(size_t) 4095;
	S->shmem_end = S->shmem_off + shmem_bytes;
	S->page_off = S->shmem_end;
	S->page_end = S->page_off + page_store_bytes;
	S->arena_off = S->page_end;
	S->arena_bytes = arena_bytes;
	S->dsm_off = S->arena_off + arena_bytes;
	S->dsm_bytes = dsm_bytes;
	AddinShmemInitLock = &S->addin_lock;
	MyProcPid = (int) getpid();
}

void *
shim_shared_alloc(size_t bytes)
{
	uint64		at;

	bytes = (bytes + 63) & ~(size_t) 63;
	at = __atomic_fetch_add(&S->dsm_used, bytes, __ATOMIC_SEQ_CST);
	if (at + bytes > S->dsm_bytes)
	{
		fprintf(stderr, "pgshim: shared pool exhausted (%zu more bytes asked for)\n", bytes);
		abort();
	}
	return Sbase + S->dsm_off + at;	/* (fresh anonymous pages: zero) */
}

void **
shim_shared_slot(int i)
{
	return &S->shared_slots[i];
}

void *
shim_arena_base(size_t *bytes)
{
	*bytes = S->arena_bytes;
	return S->arena_bytes ? Sbase + S->arena_off : NULL;
}

void
shim_postmaster_run_shmem_hooks(void)
{
	if (shmem_request_hook)
		shmem_request_hook();
	if (shmem_startup_hook)
		shmem_startup_hook();
	process_shared_preload_libraries_in_progress = false;
}

static void
child_init(void)
{
	my_proc_slot = (int) __atomic_fetch_add(&S->next_proc, 1, __ATOMIC_SEQ_CST);
	if (my_proc_slot >= SHIM_MAX_PROCS)
	{
		fprintf(stderr, "pgshim: too many processes\n");
		_exit(99);
	}
	MyLatch = &S->latches[my_proc_slot];
	MyLatch->is_set = 0;
	MyProcPid = (int) getpid();
	PG_exception_stack = NULL;
	inval_seen = __atomic_load_n(&S->inval_seq, __ATOMIC_ACQUIRE);	/* a new process starts with empty caches */
}

int
shim_fork_backend(int (*fn) (void *), void *arg)
{
	pid_t		pid;

	fflush(NULL);
	pid = fork();
	if (pid == 0)
	{
		int			result = 0;

		child_init();
		if (shim_run_toplevel(fn, arg, &result) != 0)
		{
			fprintf(stderr, "backend %d: ERROR reached the top level: %s\n", (int) getpid(), last_error);
			result = 100;
		}
		proc_exit(result);
	}
	return (int) pid;
}

static int
run_bgworker(void *arg)
{
	ShimBgw    *w = arg;

	for (int i = 0; i < n_bgw_fns; i++)
		if (strcmp(bgw_fns[i].name, w->fn) == 0)
		{
			bgw_fns[i].fn(w->arg);
			return 0;
		}
	fprintf(stderr, "pgshim: no background worker function %s\n", w->fn);
	return 98;
}

static void
start_requested_bgworkers(void)
{
	for (int i = 0; i < SHIM_MAX_BGW; i++)
	{
		ShimBgw    *w = &S->bgw[i];
		pid_t		pid;

		if (__atomic_load_n(&w->state, __ATOMIC_ACQUIRE) != 1)
			continue;

		fflush(NULL);
		pid = fork();			/* (w is shared memory: only the parent may write the pid into it) */
		if (pid == 0)
		{
			int			result = 0;

			child_init();
			is_bgworker = 1;
			/* StartBackgroundWorker: an ERROR that reaches here is reported and the worker exits with code 1 (the
			 * exit callbacks run) */
			if (shim_run_toplevel(run_bgworker, w, &result) != 0)
				result = 1;
			proc_exit(result);
		}
		w->pid = (int) pid;
		__atomic_store_n(&w->state, 2, __ATOMIC_RELEASE);
	}
}

bool
RegisterDynamicBackgroundWorker(BackgroundWorker *worker, BackgroundWorkerHandle **handle)
{
	for (int i = 0; i < SHIM_MAX_BGW; i++)
	{
		ShimBgw    *w = &S->bgw[i];
		uint32		zero = 0;

		if (!__atomic_compare_exchange_n(&w->state, &zero, 3u, 0, __ATOMIC_ACQUIRE, __ATOMIC_RELAXED))
			continue;
		snprintf(w->fn, BGW_MAXLEN, "%s", worker->bgw_function_name);
		w->arg = worker->bgw_main_arg;
		__atomic_store_n(&w->state, 1, __ATOMIC_RELEASE);
		__atomic_add_fetch(&S->bgw_kick, 1, __ATOMIC_RELEASE);
		futex_wake(&S->bgw_kick);
		if (handle)
			*handle = NULL;
		return true;
	}
	return false;				/* max_worker_processes reached */
}

void
BackgroundWorkerUnblockSignals(void)
{
	struct sigaction sa;

	memset(&sa, 0, sizeof(sa));
	sa.sa_handler = handle_sigterm;
	sigaction(SIGTERM, &sa, NULL);
}

void
BackgroundWorkerInitializeConnectionByOid(Oid dboid, Oid useroid, uint32 flags)
{
	(void) useroid, (void) flags;
	MyDatabaseId = dboid;
}

static void
reap_bgworkers(void)
{
	for (int i = 0; i < SHIM_MAX_BGW; i++)
	{
		ShimBgw    *w = &S->bgw[i];
		int			st;

		if (__atomic_load_n(&w->state, __ATOMIC_ACQUIRE) == 2 && waitpid(w->pid, &st, WNOHANG) == w->pid)
		{
			w->pid = 0;
			__atomic_store_n(&w->state, 0, __ATOMIC_RELEASE);
		}
	}
}

int
shim_live_bgworkers(void)
{
	int			n = 0;

	reap_bgworkers();
	for (int i = 0; i < SHIM_MAX_BGW; i++)
		n += __atomic_load_n(&S->bgw[i].state, __ATOMIC_ACQUIRE) == 2;
	return n;
}

int
shim_postmaster_wait(const int *pids, int npids, int *codes, double timeout_s)
{
	double		until = shim_now() + timeout_s;
	int			left = npids;
	int		   *done = calloc((size_t) (npids > 0 ? npids : 1), sizeof(int));

	for (int i = 0; i < npids; i++)
		codes[i] = -1;
	while (left > 0 && shim_now() < until)
	{
		uint32		kick = __atomic_load_n(&S->bgw_kick, __ATOMIC_ACQUIRE);

		start_requested_bgworkers();
		reap_bgworkers();
		for (int i = 0; i < npids; i++)
		{
			int			st;

			if (!done[i] && waitpid(pids[i], &st, WNOHANG) == pids[i])
			{
				done[i] = 1;
				left--;
				codes[i] = WIFEXITED(st) ? WEXITSTATUS(st) : 128 + WTERMSIG(st);
			}
		}
		if (left > 0)
			futex_wait(&S->bgw_kick, kick, 2);
	}
	for (int i = 0; i < npids; i++)
		if (!done[i])
		{
			kill(pids[i], SIGKILL);
			waitpid(pids[i], NULL, 0);
		}
	free(done);
	return left;
}

void
shim_kill_bgworkers(void)
{
	for (int i = 0; i < SHIM_MAX_BGW; i++)
		if (__atomic_load_n(&S->bgw[i].state, __ATOMIC_ACQUIRE) == 2)
		{
			kill(S->bgw[i].pid, SIGKILL);
			waitpid(S->bgw[i].pid, NULL, 0);
			S->bgw[i].pid = 0;
			__atomic_store_n(&S->bgw[i].state, 0, __ATOMIC_RELEASE);
		}
}

void
shim_postmaster_shutdown(void)
{
	double		until = shim_now() + 10.0;

	for (int i = 0; i < SHIM_MAX_BGW; i++)
		if (__atomic_load_n(&S->bgw[i].state, __ATOMIC_ACQUIRE) == 2)
			kill(S->bgw[i].pid, SIGTERM);
	while (shim_live_bgworkers() > 0 && shim_now() < until)
		usleep(2000);
	shim_kill_bgworkers();
}

/* ------------------------------------------------------------------------------------------------ pgvector's own */
#ifndef PGV_HAVE_REF_IVFUTILS
static Size
vector_item_size(int dimensions)
{
	return offsetof(Vector, x) + sizeof(float) * (Size) dimensions;
}

static Size
halfvec_item_size(int dimensions)
{
	return offsetof(Vector, x) + sizeof(uint16) * (Size) dimensions;
}
#endif

#ifndef PGV_HAVE_REF_IVFUTILS
static IvfflatTypeInfo ivf_type_infos[SHIM_MAX_RELS];
#endif
#ifndef PGV_HAVE_REF_HNSW
static HnswTypeInfo hnsw_type_infos[SHIM_MAX_RELS];
#endif
static FmgrInfo proc_infos[SHIM_MAX_RELS][8];

#ifndef PGV_HAVE_REF_IVFUTILS	/* (with -DPGV_HAVE_REF_IVFUTILS the reference's own src/ivfutils.c is linked in and has them) */
const IvfflatTypeInfo *
IvfflatGetTypeInfo(Relation index)
{
	ShimRel    *r = rel_of(index);
	IvfflatTypeInfo *t = &ivf_type_infos[r - S->rels];

	t->maxDimensions = r->opc.maxDimensions;
	t->itemSize = r->opc.maxDimensions == IVFFLAT_MAX_DIM * 2 ? halfvec_item_size : vector_item_size;
	return t;
}
#endif

#ifndef PGV_HAVE_REF_HNSW		/* (with -DPGV_HAVE_REF_HNSW the reference's own src/hnswutils.c is linked in and has them) */
const HnswTypeInfo *
HnswGetTypeInfo(Relation index)
{
	ShimRel    *r = rel_of(index);

	hnsw_type_infos[r - S->rels].maxDimensions = r->opc.maxDimensions;
	return &hnsw_type_infos[r - S->rels];
}
#endif

#ifndef PGV_HAVE_REF_IVFUTILS
/* src/ivfutils.c:46-52: NULL when the opclass has no such support function */
FmgrInfo *
IvfflatOptionalProcInfo(Relation index, uint16 procnum)
{
	ShimRel    *r = rel_of(index);

	if (procnum == IVFFLAT_NORM_PROC && !r->opc.hasNormProc)
		return NULL;
	if (procnum == IVFFLAT_KMEANS_NORM_PROC && !r->opc.hasKmeansNormProc)
		return NULL;
	return &proc_infos[r - S->rels][procnum & 7];
}
#endif

#ifndef PGV_HAVE_REF_HNSW
FmgrInfo *
HnswOptionalProcInfo(Relation index, uint16 procnum)
{
	ShimRel    *r = rel_of(index);

	if (procnum == HNSW_NORM_PROC && !r->opc.hasNormProc)
		return NULL;
	return &proc_infos[r - S->rels][procnum & 7];
}
#endif

__attribute__((weak)) Datum
vector_negative_inner_product(PG_FUNCTION_ARGS)
{
	(void) fcinfo;
	return 0;					/* (a stand-in: never called; weak, so that the reference's src/vector.c can be linked in) */
}

__attribute__((weak)) Datum
halfvec_negative_inner_product(PG_FUNCTION_ARGS)
{
	(void) fcinfo;
	return 0;					/* (weak like the above: the reference's src/halfvec.c has the real one) */
}

__attribute__((weak)) Datum
l1_distance(PG_FUNCTION_ARGS)
{
	(void) fcinfo;
	return 0;					/* (a stand-in: never called; weak, so that the reference's src/vector.c can be linked in) */
}

__attribute__((weak)) Datum
halfvec_l1_distance(PG_FUNCTION_ARGS)
{
	(void) fcinfo;
	return 0;
}

/* the reference's own FUNCTION 1 of vector_l2_ops when src/vector.c is part of the program (tests/c/pgshim_ref_runtime.c) */
extern Datum vector_l2_squared_distance(PG_FUNCTION_ARGS) __attribute__((weak));

static Datum
vector_l2_squared_distance_stub(PG_FUNCTION_ARGS)
{
	(void) fcinfo;
	return 0;
}

/* FUNCTION 3 / 4 / 2 of the ivfflat vector opclasses (sql/vector.sql: l2_distance or vector_spherical_distance, vector_norm):
 * there when the reference's src/vector.c is linked in, NULL in the plain build -- whose glue never calls through them */
extern Datum l2_distance(PG_FUNCTION_ARGS) __attribute__((weak));
extern Datum vector_spherical_distance(PG_FUNCTION_ARGS) __attribute__((weak));
extern Datum vector_norm(PG_FUNCTION_ARGS) __attribute__((weak));

/* the halfvec opclasses' support functions (sql/vector.sql:819-866), there when the reference's src/halfvec.c and
 * src/ivfutils.c / src/hnswutils.c are part of the program */
extern Datum halfvec_l2_squared_distance(PG_FUNCTION_ARGS) __attribute__((weak));
extern Datum halfvec_l2_distance(PG_FUNCTION_ARGS) __attribute__((weak));
extern Datum halfvec_spherical_distance(PG_FUNCTION_ARGS) __attribute__((weak));
extern Datum halfvec_l2_norm(PG_FUNCTION_ARGS) __attribute__((weak));
extern Datum ivfflat_halfvec_support(PG_FUNCTION_ARGS) __attribute__((weak));
extern Datum hnsw_halfvec_support(PG_FUNCTION_ARGS) __attribute__((weak));

FmgrInfo *
index_getprocinfo(Relation irel, int attnum, uint16 procnum)
{
	ShimRel    *r = rel_of(irel);
	FmgrInfo   *f = &proc_infos[r - S->rels][procnum & 7];

	(void) attnum;
	if (r->opc.halfvec)
	{
		if (procnum == 1)
			f->fn_addr = r->opc.distanceFn == 1 ? halfvec_negative_inner_product :
				(r->opc.distanceFn == 2 ? halfvec_l1_distance : halfvec_l2_squared_distance);
		else if (r->opc.am == 0 && procnum == 3)
			f->fn_addr = r->opc.distanceFn == 0 ? halfvec_l2_distance : halfvec_spherical_distance;
		else if (r->opc.am == 0 && procnum == 5)
			f->fn_addr = ivfflat_halfvec_support;
		else if (r->opc.am == 1 && procnum == 3)
			f->fn_addr = hnsw_halfvec_support;
		else if (procnum == 2 || (r->opc.am == 0 && procnum == 4))
			f->fn_addr = halfvec_l2_norm;
		return f;
	}
	if (procnum == 1)
		f->fn_addr = r->opc.distanceFn == 1 ? vector_negative_inner_product :
			(r->opc.distanceFn == 2 ? l1_distance :
			 (vector_l2_squared_distance ? vector_l2_squared_distance : vector_l2_squared_distance_stub));
	else if (r->opc.am == 0 && procnum == 3)
		f->fn_addr = r->opc.distanceFn == 0 ? l2_distance : vector_spherical_distance;
	else if (procnum == 2 || (r->opc.am == 0 && procnum == 4))
		f->fn_addr = vector_norm;
	return f;
}

/* what the reference's IvfflatOptionalProcInfo / HnswOptionalProcInfo (src/ivfutils.c:46-52, src/hnswutils.c:103-110) ask
 * before index_getprocinfo.  ivfflat: 1 distance, 2 norm, 3 k-means distance, 4 k-means norm, 5 type info; hnsw: 1 distance,
 * 2 norm, 3 type info.  The vector opclasses have no type-info function. */
Oid
index_getprocid(Relation irel, int16 attnum, uint16 procnum)
{
	ShimRel    *r = rel_of(irel);

	(void) attnum;
	if (procnum == 1)
		return 1;
	if (procnum == 2)
		return r->opc.hasNormProc ? 2 : InvalidOid;
	if (r->opc.am == 0 && procnum == 3)
		return 3;
	if (r->opc.am == 0 && procnum == 4)
		return r->opc.hasKmeansNormProc ? 4 : InvalidOid;
	if (r->opc.halfvec && procnum == (r->opc.am == 0 ? 5 : 3))
		return 5;				/* ivfflat_halfvec_support / hnsw_halfvec_support */
	return InvalidOid;
}

#ifndef PGV_HAVE_REF_IVFUTILS
/* src/ivfutils.c:150-175: dimensions and lists off the meta page (block 0) */
void
IvfflatGetMetaPageInfo(Relation index, int *lists, int *dimensions)
{
	Buffer		buf = ReadBufferExtended(index, MAIN_FORKNUM, 0, RBM_NORMAL, NULL);
	Page		page;
	uint32		magic;
	uint16		d,
				l;

	LockBuffer(buf, BUFFER_LOCK_SHARE);
	page = BufferGetPage(buf);
	memcpy(&magic, PageGetContents(page), 4);
	memcpy(&d, PageGetContents(page) + 8, 2);
	memcpy(&l, PageGetContents(page) + 10, 2);
	UnlockReleaseBuffer(buf);
	if (magic != 0x14FF1A7)
		elog(ERROR, "ivfflat index is not valid");
	if (lists)
		*lists = l;
	if (dimensions)
		*dimensions = d;
}
#endif

#ifndef PGV_HAVE_REF_HNSW		/* (the reference's own src/hnswutils.c otherwise) */
HnswElement
HnswInitElementFromBlock(BlockNumber blkno, OffsetNumber offno)
{
	HnswElement e = palloc(sizeof(HnswElementData));
	char	   *base = NULL;

	/* src/hnswutils.c:282-293 sets these four fields of a palloc'd (NOT zeroed) element; the rest is the caller's */
	memset(e, 0x5a, sizeof(HnswElementData));
	e->blkno = blkno;
	e->offno = offno;
	HnswPtrStore(base, e->neighbors, (HnswNeighborArrayPtr *) NULL);
	HnswPtrStore(base, e->value, (char *) NULL);
	return e;
}

void
HnswAddHeapTid(HnswElement element, ItemPointer heaptid)
{
	element->heaptids[element->heaptidsLength++] = *heaptid;
}

/* src/hnswutils.c:229-236 */
void *
HnswAlloc(HnswAllocator * allocator, Size size)
{
	if (allocator)
		return (*(allocator)->alloc) (size, (allocator)->state);
	return palloc(size);
}

/* HnswInitElement + HnswInitNeighbors + HnswInitNeighborArray, src/hnswutils.c:201-266: the level is drawn here, one
 * RandomDouble() per heap tuple */
HnswElement
HnswInitElement(char *base, ItemPointer heaptid, int m, double ml, int maxLevel, HnswAllocator * allocator)
{
	HnswElement element = HnswAlloc(allocator, sizeof(HnswElementData));
	int			level = (int) (-log(RandomDouble()) * ml);
	HnswNeighborArrayPtr *neighborList;

	if (level > maxLevel)
		level = maxLevel;
	memset(element, 0x5a, sizeof(HnswElementData));	/* (the server's allocator does not zero either) */
	element->heaptidsLength = 0;
	HnswAddHeapTid(element, heaptid);
	element->level = (uint8) level;
	element->deleted = 0;
	element->version = 1;
	neighborList = HnswAlloc(allocator, sizeof(HnswNeighborArrayPtr) * (Size) (level + 1));
	HnswPtrStore(base, element->neighbors, neighborList);
	for (int lc = 0; lc <= level; lc++)
	{
		int			lm = HnswGetLayerM(m, lc);
		HnswNeighborArray *a = HnswAlloc(allocator, offsetof(HnswNeighborArray, items) + sizeof(HnswCandidate) * (Size) lm);

		a->length = 0;
		a->closerSet = false;
		HnswPtrStore(base, neighborList[lc], a);
	}
	HnswPtrStore(base, element->value, (char *) NULL);
	return element;
}

#endif

static uint64 rng_state = 0x9E3779B97F4A7C15ull;

static uint64
rng_next(void)
{
	rng_state ^= rng_state << 13;
	rng_state ^= rng_state >> 7;
	rng_state ^= rng_state << 17;
	return rng_state;
}

void
shim_seed_random(uint64 seed)
{
	rng_state = seed ? seed : 1;
}

#ifndef RandomDouble				/* (compiled against the reference's own ivfflat.h / hnsw.h they are macros over pg_prng) */
double
RandomDouble(void)
{
	return (double) (rng_next() >> 11) / 9007199254740992.0;
}

int
RandomInt(void)
{
	return (int) (rng_next() >> 33);
}
#else
/* common/pg_prng.h, as far as RandomDouble() / RandomInt() of the reference's headers go */
pg_prng_state pg_global_prng_state;
static double (*prng_double_hook) (void *);
static uint32_t (*prng_u32_hook) (void *);
static void *prng_hook_state;

/* draws come from the caller's generator from now on (NULLs: back to this file's): a test that runs the reference's
 * k-means beside the oracle's hands both the same pg_prng stream */
void
shim_prng_hook(double (*next_double) (void *), uint32_t (*next_u32) (void *), void *state)
{
	prng_double_hook = next_double;
	prng_u32_hook = next_u32;
	prng_hook_state = state;
}

/* a generator of the caller's own (utils/sampling.h keeps one per BlockSampler / reservoir): a stream apart from the
 * global one, like the server's -- xorshift128+ over the state words, nothing the tests pin */
static uint64
private_next(pg_prng_state *state)
{
	uint64		a = state->s0,
				b = state->s1;

	state->s0 = b;
	a ^= a << 23;
	state->s1 = a ^ b ^ (a >> 17) ^ (b >> 26);
	return state->s1 + b;
}

void
pg_prng_seed(pg_prng_state *state, uint64 seed)
{
	state->s0 = seed * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull;
	state->s1 = (seed ^ 0xD1B54A32D192ED03ull) * 0xBF58476D1CE4E5B9ull + 1;
}

double
pg_prng_double(pg_prng_state *state)
{
	if (state != &pg_global_prng_state)
		return (double) (private_next(state) >> 11) / 9007199254740992.0;
	if (prng_double_hook)
		return prng_double_hook(prng_hook_state);
	return (double) (rng_next() >> 11) / 9007199254740992.0;
}

uint32
pg_prng_uint32(pg_prng_state *state)
{
	if (state != &pg_global_prng_state)
		return (uint32) (private_next(state) >> 32);
	if (prng_u32_hook)
		return prng_u32_hook(prng_hook_state);
	return (uint32) (rng_next() >> 32);
}
#endif
