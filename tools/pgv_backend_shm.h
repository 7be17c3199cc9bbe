/*
 * pgv_backend_shm.h -- the shared-memory records tools/backends_driver.c (the parent) and tools/pgv_backend.c (its
 * child processes) exchange.  Offsets are relative to the start of the segment (position independent).
 */
#ifndef PGV_BACKEND_SHM_H
#define PGV_BACKEND_SHM_H

#include <stdint.h>

#define PGVB_BANK_MAGIC 0x7067765f62616e6bull	/* "pgv_bank" */
#define PGVB_IMAGE_MAGIC 0x7067765f696d6167ull	/* "pgv_imag" */
#define PGVB_MAX_K 64

/* one client's outcome */
typedef struct
{
	double		t0,
				t1;
	int32_t		rc;
	char		err[204];
	/* getrusage of the timed phase (PGVB_CPU_STAT): where a client's CPU time goes */
	double		utime,
				stime;
	int64_t		nvcsw,
				nivcsw;
}			pgvb_client;

/* queries in, latencies and (verify) answers out */
typedef struct
{
	uint64_t	magic;
	int32_t		nq,
				probes,
				k,
				per_client,
				warmup,
				nclients,
				device,
				verify;
	uint64_t	query_bytes;
	uint32_t	ready,			/* futex word: clients at the start line */
				go,				/* futex word: the driver fires */
				finished;
	uint32_t	gate;			/* futex word: own-context queries in flight over all processes (PGV_BACKEND_GATE) */
	uint64_t	clients_off,	/* pgvb_client [nclients] */
				lat_off,		/* double [nclients x per_client] seconds */
				tid_off,		/* uint64 [nclients x per_client x k] (verify) */
				dist_off,		/* float  [nclients x per_client x k] (verify) */
				queries_off;	/* [nq x query_bytes] */
}			pgvb_bank;

/* a list-major index image for the owner process to upload (pgv_index_upload's arguments) */
typedef struct
{
	uint64_t	magic;
	int32_t		metric,
				dtype,
				dim,
				nlists;
	int64_t		nrows;
	uint64_t	centers_off,
				offsets_off,
				vectors_off,
				tids_off;		/* 0 = none */
}			pgvb_image;

#endif
