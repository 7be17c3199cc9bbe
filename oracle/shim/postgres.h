/*
 * Minimal stand-in for PostgreSQL's postgres.h, just enough for the
 * reference's src/halfutils.c to compile UNMODIFIED into oracle/_ref.
 * TEST INFRASTRUCTURE ONLY -- original code, nothing copied from PostgreSQL.
 */
#ifndef PGV_SHIM_POSTGRES_H
#define PGV_SHIM_POSTGRES_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef uint8_t uint8;
typedef uint16_t uint16;
typedef uint32_t uint32;
typedef uint64_t uint64;
typedef int8_t int8;
typedef int16_t int16;
typedef int32_t int32;
typedef int64_t int64;
typedef size_t Size;
typedef uintptr_t Datum;
typedef char *Pointer;

#define PG_VERSION_NUM 170000
#define FLEXIBLE_ARRAY_MEMBER
#define HAVE__GET_CPUID 1
#define HAVE_LONG_INT_64 1
#define unlikely(x) __builtin_expect((x) != 0, 0)
#define likely(x) __builtin_expect((x) != 0, 1)
#define PGDLLEXPORT
#define PG_FUNCTION_ARGS void *fcinfo

#define ERROR 21
#define ERRCODE_NUMERIC_VALUE_OUT_OF_RANGE 1
#define errcode(x) (x)
#define errmsg(...) 0
void		pgv_shim_error(void) __attribute__((noreturn));
#define ereport(level, rest) pgv_shim_error()

static inline void *
palloc(Size n)
{
	return malloc(n);
}

static inline Size
add_size(Size a, Size b)
{
	return a + b;
}

static inline Size
mul_size(Size a, Size b)
{
	return a * b;
}

#endif
