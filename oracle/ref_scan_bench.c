/*
 * ref_scan_bench.c -- times the REFERENCE'S OWN compiled index scan.  TEST INFRASTRUCTURE ONLY (bench.py's
 * `cpu_baseline` leg and tests/ may run it; nothing of the product links or calls it).
 *
 * What runs: ivfflatbeginscan / ivfflatrescan / ivfflatgettuple / ivfflatendscan of /root/reference/src/ivfscan.c
 * (GetScanLists :47-118, GetScanItems :123-187, ivfflatgettuple :361-414), its src/ivfutils.c, and the distance
 * functions of its src/vector.c (VectorL2SquaredDistance :560-574 behind vector_l2_squared_distance :595-605) and
 * src/halfvec.c / src/halfutils.c -- UNMODIFIED and UNPATCHED (no hook lines: this program has no device), compiled
 * where they lie by oracle/Makefile `refbench` with the reference's own flags (its Makefile:30 + PGXS's -O2).  They run
 * over the 8 KB page image of an index (ivfflat's on-disk format, src/ivfflat.h:251-275) exactly as they do inside a
 * server: one single-threaded backend PROCESS per connection (index scans are not parallel: src/ivfflat.c:207,266),
 * every page through ReadBuffer / LockBuffer / UnlockReleaseBuffer, every tuple through index_getattr, the fmgr call,
 * a virtual slot and tuplesort_puttupleslot, the result through tuplesort_performsort / gettupleslot.
 *
 * What stands in for PostgreSQL (absent from this image): tests/c/pgshim_runtime.c + pgshim_ref_runtime.c -- a buffer
 * manager over one shared mapping (no hash lookup, no clock sweep, a striped reader count instead of per-buffer content
 * locks), palloc contexts, fmgr, slots, a tuplesort that copies each tuple and qsorts.  All of it is LIGHTER than the
 * real thing, so the numbers printed here are an upper bound on what the reference does inside a real server.
 *
 * usage: ref_scan_bench PAGES QUERIES DIM NQ PROBES K PROCS SECS SECS_SINGLE ANSWERS [f32|f16] [l2|ip]
 *   PAGES    file of 8 KB pages (the index relation, block 0 first)
 *   QUERIES  NQ x DIM elements (float, or IEEE half with f16)
 *   ANSWERS  written: NQ x K uint64 heap TIDs ((block << 16) | offset, ~0 where the scan ended early), in scan order
 * prints ONE JSON object on stdout.
 */
#include "postgres.h"

#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include "ivfflat.h"
#include "vector.h"
#include "halfvec.h"
#include "pgshim_runtime.h"

#define REL_BENCH 7000
#define MAX_PROCS 40			/* the stand-in runtime has 96 process slots for a program's life: A + B + C */

extern int	ivfflat_probes;
extern void HalfvecInit(void);

typedef struct Board
{
	volatile int ready;			/* backends attached and warmed up */
	volatile int go;			/* 1: the timed phase runs, 2: stop */
	volatile int64_t done[MAX_PROCS];	/* queries each backend finished inside the timed phase */
	volatile double busy[MAX_PROCS];	/* seconds each backend spent in them */
}			Board;

static Board *board;
static uint64_t *answers;		/* [nq x k], shared */
static const void *queries;
static int	dim,
			nq,
			probes,
			k,
			is_half;
static size_t qbytes;

static double
now(void)
{
	struct timespec ts;

	clock_gettime(CLOCK_MONOTONIC, &ts);
	return ts.tv_sec + ts.tv_nsec * 1e-9;
}

static uint64_t
tid_key(const ItemPointerData *p)
{
	return ((uint64_t) (((uint32_t) p->ip_blkid.bi_hi << 16) | p->ip_blkid.bi_lo) << 16) | p->ip_posid;
}

/* ORDER BY embedding <-> $1 LIMIT k through the access method's own entry points; out may be NULL */
static int
one_query(Relation index, int qi, uint64_t *out)
{
	MemoryContext ctx = shim_query_context_begin();
	ScanKeyData orderby;
	IndexScanDesc scan;
	int			n = 0;

	memset(&orderby, 0, sizeof(orderby));
	if (is_half)
	{
		HalfVector *v = InitHalfVector(dim);

		memcpy(v->x, (const char *) queries + (size_t) qi * qbytes, qbytes);
		orderby.sk_argument = PointerGetDatum(v);
	}
	else
	{
		Vector	   *v = InitVector(dim);

		memcpy(v->x, (const char *) queries + (size_t) qi * qbytes, qbytes);
		orderby.sk_argument = PointerGetDatum(v);
	}
	scan = ivfflatbeginscan(index, 0, 1);
	ivfflatrescan(scan, NULL, 0, &orderby, 1);
	while (n < k && ivfflatgettuple(scan, ForwardScanDirection))
	{
		if (out)
			out[n] = tid_key(&scan->xs_heaptid);
		n++;
	}
	ivfflatendscan(scan);
	shim_query_context_end(ctx);
	return n;
}

/* phase A: the answers (and first touch of every page a later phase reads); backend p takes queries p, p + P, ... */
typedef struct Share
{
	int			p,
				nprocs;
}			Share;

static int
backend_answers(void *arg)
{
	Share	   *s = arg;
	Relation	index = shim_open_relation(REL_BENCH);

	ivfflat_probes = probes;
	for (int qi = s->p; qi < nq; qi += s->nprocs)
	{
		uint64_t   *out = answers + (size_t) qi * k;
		int			n = one_query(index, qi, out);

		for (; n < k; n++)
			out[n] = ~(uint64_t) 0;
	}
	return shim_pinned_buffers() != 0;
}

/* phases B / C: whole queries back to back until told to stop; backend p starts at its own place in the query file */
static int
backend_timed(void *arg)
{
	Share	   *s = arg;
	Relation	index = shim_open_relation(REL_BENCH);
	int			qi = (int) ((int64_t) s->p * nq / s->nprocs);
	int64_t		done = 0;
	double		t0;

	ivfflat_probes = probes;
	one_query(index, qi, NULL);	/* relcache, fmgr lookups, first palloc blocks */
	__atomic_add_fetch(&board->ready, 1, __ATOMIC_SEQ_CST);
	while (__atomic_load_n(&board->go, __ATOMIC_ACQUIRE) == 0)
		usleep(200);
	t0 = now();
	while (__atomic_load_n(&board->go, __ATOMIC_ACQUIRE) == 1)
	{
		one_query(index, qi, NULL);
		qi = qi + 1 == nq ? 0 : qi + 1;
		done++;
		board->done[s->p] = done;
		board->busy[s->p] = now() - t0;
	}
	return 0;
}

static double
run_timed(int nprocs, double secs, int64_t *total)
{
	static Share shares[MAX_PROCS];
	int			pids[MAX_PROCS],
				codes[MAX_PROCS];
	double		t0,
				el;
	int64_t		sum = 0;

	memset((void *) board, 0, sizeof(*board));
	for (int p = 0; p < nprocs; p++)
	{
		shares[p].p = p;
		shares[p].nprocs = nprocs;
		pids[p] = shim_fork_backend(backend_timed, &shares[p]);
	}
	while (__atomic_load_n(&board->ready, __ATOMIC_ACQUIRE) < nprocs)
		usleep(500);
	t0 = now();
	__atomic_store_n(&board->go, 1, __ATOMIC_RELEASE);
	while (now() - t0 < secs)
		usleep(1000);
	/* what was FINISHED inside the window counts (a backend's query in flight at the end does not) */
	for (int p = 0; p < nprocs; p++)
		sum += board->done[p];
	el = now() - t0;
	__atomic_store_n(&board->go, 2, __ATOMIC_RELEASE);
	if (shim_postmaster_wait(pids, nprocs, codes, 120.0) != 0)
		fprintf(stderr, "ref_scan_bench: backends did not end\n");
	*total = sum;
	return el;
}

int
main(int argc, char **argv)
{
	ShimOpclass opc = {0, IVFFLAT_MAX_DIM, false, false, 0, 0};
	struct stat st;
	int			fd,
				procs;
	double		secs,
				secs1;
	void	   *pages;
	uint32_t	nblocks;
	int64_t		total = 0,
				single = 0;
	double		el = 0,
				el1 = 0,
				t_load,
				t_answers;
	FILE	   *f;

	if (argc < 11)
	{
		fprintf(stderr, "usage: ref_scan_bench PAGES QUERIES DIM NQ PROBES K PROCS SECS SECS_SINGLE ANSWERS [f32|f16] [l2|ip]\n");
		return 2;
	}
	dim = atoi(argv[3]);
	nq = atoi(argv[4]);
	probes = atoi(argv[5]);
	k = atoi(argv[6]);
	procs = atoi(argv[7]);
	secs = atof(argv[8]);
	secs1 = atof(argv[9]);
	is_half = argc > 11 && strcmp(argv[11], "f16") == 0;
	if (argc > 12 && strcmp(argv[12], "ip") == 0)
	{
		opc.hasKmeansNormProc = true;	/* vector_ip_ops: FUNCTION 4, no FUNCTION 2 */
		opc.distanceFn = 1;
	}
	if (is_half)
	{
		opc.maxDimensions = IVFFLAT_MAX_DIM * 2;
		opc.halfvec = 1;
	}
	if (procs < 1)
		procs = 1;
	if (procs > MAX_PROCS)
		procs = MAX_PROCS;
	qbytes = (size_t) dim * (is_half ? 2 : 4);

	t_load = now();
	fd = open(argv[1], O_RDONLY);
	if (fd < 0 || fstat(fd, &st) != 0 || st.st_size % 8192 != 0)
	{
		fprintf(stderr, "ref_scan_bench: cannot read %s as 8 KB pages\n", argv[1]);
		return 2;
	}
	nblocks = (uint32_t) (st.st_size / 8192);
	pages = mmap(NULL, (size_t) st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
	if (pages == MAP_FAILED)
	{
		perror("mmap");
		return 2;
	}
	/* the page store: shim_create_relation keeps room for twice the blocks (MAP_NORESERVE: untouched room is free) */
	shim_postmaster_init(((size_t) nblocks * 2 + 8192) * 8192, 0);
	HalfvecInit();				/* (_PG_init, src/vector.c:59) */
	shim_create_relation(REL_BENCH, &opc, pages, nblocks, dim);
	munmap(pages, (size_t) st.st_size);
	close(fd);
	{
		size_t		bytes = (size_t) nq * qbytes;
		void	   *q = shim_shared_alloc(bytes);

		f = fopen(argv[2], "rb");
		if (f == NULL || fread(q, 1, bytes, f) != bytes)
		{
			fprintf(stderr, "ref_scan_bench: cannot read %d queries from %s\n", nq, argv[2]);
			return 2;
		}
		fclose(f);
		queries = q;
	}
	answers = shim_shared_alloc(sizeof(uint64_t) * (size_t) nq * k);
	board = shim_shared_alloc(sizeof(Board));
	t_load = now() - t_load;

	/* A: answers */
	t_answers = now();
	{
		static Share shares[MAX_PROCS];
		int			pids[MAX_PROCS],
					codes[MAX_PROCS];
		int			np = procs < nq ? procs : nq;

		for (int p = 0; p < np; p++)
		{
			shares[p].p = p;
			shares[p].nprocs = np;
			pids[p] = shim_fork_backend(backend_answers, &shares[p]);
		}
		if (shim_postmaster_wait(pids, np, codes, 1800.0) != 0)
		{
			fprintf(stderr, "ref_scan_bench: the answer pass did not end\n");
			return 1;
		}
		for (int p = 0; p < np; p++)
			if (codes[p] != 0)
			{
				fprintf(stderr, "ref_scan_bench: backend %d of the answer pass ended with code %d: %s\n", p, codes[p], shim_last_error());
				return 1;
			}
	}
	t_answers = now() - t_answers;
	f = fopen(argv[10], "wb");
	if (f == NULL || fwrite(answers, sizeof(uint64_t), (size_t) nq * k, f) != (size_t) nq * k)
	{
		fprintf(stderr, "ref_scan_bench: cannot write %s\n", argv[10]);
		return 2;
	}
	fclose(f);

	/* B: `procs` backends, C: one */
	if (secs > 0)
		el = run_timed(procs, secs, &total);
	if (secs1 > 0)
		el1 = run_timed(1, secs1, &single);

	printf("{\"qps\": %.3f, \"procs\": %d, \"queries\": %lld, \"secs\": %.3f, \"single_qps\": %.3f, \"single_queries\": %lld, "
		   "\"single_secs\": %.3f, \"answer_pass_secs\": %.3f, \"load_secs\": %.3f, \"blocks\": %u, \"dim\": %d, \"probes\": %d, "
		   "\"k\": %d, \"type\": \"%s\", \"isa\": \"%s\"}\n",
		   el > 0 ? total / el : 0.0, procs, (long long) total, el, el1 > 0 ? single / el1 : 0.0, (long long) single, el1,
		   t_answers, t_load, nblocks, dim, probes, k, is_half ? "halfvec" : "vector",
#ifdef REF_BENCH_ISA
		   REF_BENCH_ISA
#else
		   "?"
#endif
		);
	return 0;
}
