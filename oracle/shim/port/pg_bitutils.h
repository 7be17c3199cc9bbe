/*
 * Stand-in for PostgreSQL's port/pg_bitutils.h: what the reference's src/bitutils.c uses from it.
 * TEST INFRASTRUCTURE ONLY -- original code, nothing copied from PostgreSQL.
 */
#ifndef PGV_SHIM_PG_BITUTILS_H
#define PGV_SHIM_PG_BITUTILS_H

#define HAVE__BUILTIN_POPCOUNT 1
extern const uint8 pg_number_of_ones[256];	/* defined in ref_glue.c */

#endif
