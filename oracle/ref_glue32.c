/*
 * ref_glue32.c -- what the reference's src/vector.c needs of a server to RUN its fp32 distance functions, and plain-C
 * entry points onto them for ctypes.  src/vector.c is compiled UNMODIFIED, where it lies under $(REFERENCE), against the
 * declaration-only stand-in headers of ../ext/shim (pgshim.h + pgshim_ref.h); this file gives the handful of those
 * declarations that the distance path reaches a body: the fmgr argument accessors, palloc, ereport (a longjmp back to
 * the entry point, the message kept -- so the reference's own error texts can be compared), float8 Datums.  Everything
 * else the object references (array / pq / typmod input-output, the other types' init functions) aborts: nothing here
 * calls into it.  Linked into oracle/_ref/libpgvref32.so.  TEST INFRASTRUCTURE ONLY.
 *
 * VectorL2SquaredDistance, VectorInnerProduct, VectorCosineSimilarity, VectorL1Distance (src/vector.c:560-735) are
 * `static` there; they are reached through the SQL-callable wrappers around them (l2_distance :579,
 * vector_l2_squared_distance :595, inner_product :622, vector_negative_inner_product :637, cosine_distance :671,
 * vector_spherical_distance :703, l1_distance :740, vector_norm :767, l2_normalize :785).
 */
#include "postgres.h"
#include "fmgr.h"
#include "vector.h"

#include <setjmp.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

struct FunctionCallInfoBaseData
{
	Datum		args[4];
	int			nargs;
};

static jmp_buf ref_jmp;
static bool ref_jmp_set;
static char ref_errtext[512];

/* ---- fmgr / Datum / varlena */
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
struct varlena *
pg_detoast_datum(struct varlena *datum)
{
	return datum;
}
Datum
float8_as_datum(double x)
{
	Datum		d;

	memcpy(&d, &x, sizeof(d));
	return d;
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

/* ---- memory */
void *
palloc(Size size)
{
	void	   *p = malloc(size ? size : 1);

	if (!p)
		abort();
	return p;
}
void *
palloc0(Size size)
{
	void	   *p = calloc(1, size ? size : 1);

	if (!p)
		abort();
	return p;
}
void
pfree(void *p)
{
	free(p);
}
Size
add_size(Size a, Size b)
{
	return a + b;
}
Size
mul_size(Size a, Size b)
{
	return a * b;
}

/* ---- ereport(ERROR, (errcode(..), errmsg(..))): the message is formatted first, then the report longjmps */
int
pgshim_errcode(int sqlstate)
{
	(void) sqlstate;
	return 0;
}
int
pgshim_errmsg(const char *fmt,...)
{
	va_list		ap;

	va_start(ap, fmt);
	vsnprintf(ref_errtext, sizeof(ref_errtext), fmt, ap);
	va_end(ap);
	return 0;
}
/* errdetail / errhint (ext/shim/pgshim_ref.h): the primary message is the one the tests compare */
int
pgshim_errmore(const char *fmt,...)
{
	(void) fmt;
	return 0;
}
static void
ref_raise(void)
{
	if (!ref_jmp_set)
	{
		fprintf(stderr, "pgvref32: ERROR outside an entry point: %s\n", ref_errtext);
		abort();
	}
	longjmp(ref_jmp, 1);
}
void
pgshim_ereport(int level, int dummy)
{
	(void) dummy;
	if (level >= ERROR)
		ref_raise();
}
void
pgshim_elog(int level, const char *fmt,...)
{
	va_list		ap;

	va_start(ap, fmt);
	vsnprintf(ref_errtext, sizeof(ref_errtext), fmt, ap);
	va_end(ap);
	if (level >= ERROR)
		ref_raise();
}
void
float_overflow_error(void)
{
	snprintf(ref_errtext, sizeof(ref_errtext), "value out of range: overflow");
	ref_raise();
	abort();
}
void
float_underflow_error(void)
{
	snprintf(ref_errtext, sizeof(ref_errtext), "value out of range: underflow");
	ref_raise();
	abort();
}

/* ---- entry points */
extern Datum l2_distance(PG_FUNCTION_ARGS);
extern Datum vector_l2_squared_distance(PG_FUNCTION_ARGS);
extern Datum inner_product(PG_FUNCTION_ARGS);
extern Datum vector_negative_inner_product(PG_FUNCTION_ARGS);
extern Datum cosine_distance(PG_FUNCTION_ARGS);
extern Datum vector_spherical_distance(PG_FUNCTION_ARGS);
extern Datum l1_distance(PG_FUNCTION_ARGS);
extern Datum vector_norm(PG_FUNCTION_ARGS);
extern Datum l2_normalize(PG_FUNCTION_ARGS);

static const struct
{
	const char *name;
	PGFunction	fn;
}			ref_fns[] = {
	{"l2_distance", l2_distance},
	{"vector_l2_squared_distance", vector_l2_squared_distance},
	{"inner_product", inner_product},
	{"vector_negative_inner_product", vector_negative_inner_product},
	{"cosine_distance", cosine_distance},
	{"vector_spherical_distance", vector_spherical_distance},
	{"l1_distance", l1_distance},
	{"vector_norm", vector_norm},
};

static Vector *
make_vector(int dim, const float *x)
{
	Size		size = VECTOR_SIZE(dim);
	Vector	   *v = palloc0(size);

	SET_VARSIZE(v, size);
	v->dim = (int16) dim;
	memcpy(v->x, x, sizeof(float) * (size_t) dim);
	return v;
}

const char *
pgvref32_last_error(void)
{
	return ref_errtext;
}

/* a float8-valued function of the table above on (a) or (a, b): 0 and *out, or 1 and pgvref32_last_error() */
int
pgvref32_call(const char *name, int dim_a, const float *a, int dim_b, const float *b, double *out)
{
	struct FunctionCallInfoBaseData fc;
	PGFunction	fn = NULL;
	Vector	   *va,
			   *vb = NULL;
	volatile int rc = 0;

	for (size_t i = 0; i < sizeof(ref_fns) / sizeof(ref_fns[0]); i++)
		if (strcmp(ref_fns[i].name, name) == 0)
			fn = ref_fns[i].fn;
	if (fn == NULL)
	{
		snprintf(ref_errtext, sizeof(ref_errtext), "pgvref32: no function %s", name);
		return 2;
	}
	va = make_vector(dim_a, a);
	if (b)
		vb = make_vector(dim_b, b);
	memset(&fc, 0, sizeof(fc));
	fc.args[0] = PointerGetDatum(va);
	fc.args[1] = PointerGetDatum(vb);
	fc.nargs = b ? 2 : 1;
	ref_errtext[0] = 0;
	ref_jmp_set = true;
	if (setjmp(ref_jmp) == 0)
		*out = DatumGetFloat8(fn(&fc));
	else
		rc = 1;
	ref_jmp_set = false;
	free(va);
	free(vb);
	return rc;
}

/* rows [n x dim] against one query: the loop a scan makes around FUNCTION 1, for bulk comparisons */
int
pgvref32_rows(const char *name, int dim, const float *query, const float *rows, long n, double *out)
{
	for (long i = 0; i < n; i++)
	{
		int			rc = pgvref32_call(name, dim, rows + (size_t) i * dim, dim, query, &out[i]);

		if (rc != 0)
			return rc;
	}
	return 0;
}

/* l2_normalize (src/vector.c:785-819) -> out[dim] */
int
pgvref32_l2_normalize(int dim, const float *a, float *out)
{
	struct FunctionCallInfoBaseData fc;
	Vector	   *va = make_vector(dim, a);
	volatile int rc = 0;

	memset(&fc, 0, sizeof(fc));
	fc.args[0] = PointerGetDatum(va);
	fc.nargs = 1;
	ref_errtext[0] = 0;
	ref_jmp_set = true;
	if (setjmp(ref_jmp) == 0)
	{
		Vector	   *r = (Vector *) DatumGetPointer(l2_normalize(&fc));

		memcpy(out, r->x, sizeof(float) * (size_t) dim);
		free(r);
	}
	else
		rc = 1;
	ref_jmp_set = false;
	free(va);
	return rc;
}
