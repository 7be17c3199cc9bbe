/*
 * ref_glue.c -- the few symbols the reference's src/halfutils.c expects from
 * the server, plus plain-C entry points onto its dispatch pointers so ctypes
 * can call them.  Linked with the UNMODIFIED reference object into
 * oracle/_ref/libpgvref.so.  TEST INFRASTRUCTURE ONLY.
 */
#include "postgres.h"
#include "halfutils.h"
#include "bitutils.h"
#include "port/pg_bitutils.h"

#include <stdio.h>

void
pgv_shim_error(void)
{
	fprintf(stderr, "pgvref: ereport(ERROR) reached in shimmed reference code\n");
	abort();
}

int
float_to_shortest_decimal_buf(float f, char *result)
{
	return snprintf(result, FLOAT_SHORTEST_DECIMAL_LEN, "%g", f);
}

/* HalfvecInit (src/halfutils.c:278-300) picks Default or F16C variants */
void
pgvref_init(void)
{
	HalfvecInit();
}

float
pgvref_halfvec_l2_squared(int dim, uint16 *a, uint16 *b)
{
	return HalfvecL2SquaredDistance(dim, (half *) a, (half *) b);
}

float
pgvref_halfvec_inner_product(int dim, uint16 *a, uint16 *b)
{
	return HalfvecInnerProduct(dim, (half *) a, (half *) b);
}

double
pgvref_halfvec_cosine_similarity(int dim, uint16 *a, uint16 *b)
{
	return HalfvecCosineSimilarity(dim, (half *) a, (half *) b);
}

float
pgvref_halfvec_l1(int dim, uint16 *a, uint16 *b)
{
	return HalfvecL1Distance(dim, (half *) a, (half *) b);
}

/* src/halfutils.h:62-141 and :146-233 as compiled into this object */
float
pgvref_half_to_float(uint16 h)
{
	return HalfToFloat4((half) h);
}

uint16
pgvref_float_to_half(float f)
{
	return (uint16) Float4ToHalfUnchecked(f);
}

/* ---- src/bitutils.c, compiled unmodified next to halfutils.c ---- */

/* the byte popcount table the reference's tail loops index (a server symbol) */
const uint8 pg_number_of_ones[256] = {
#define B2(n) n, n + 1, n + 1, n + 2
#define B4(n) B2(n), B2(n + 1), B2(n + 1), B2(n + 2)
#define B6(n) B4(n), B4(n + 1), B4(n + 1), B4(n + 2)
	B6(0), B6(1), B6(1), B6(2)
};

void
pgvref_bit_init(void)
{
	BitvecInit();
}

uint64
pgvref_bit_hamming(uint32 bytes, unsigned char *a, unsigned char *b)
{
	return BitHammingDistance(bytes, a, b, 0);
}

double
pgvref_bit_jaccard(uint32 bytes, unsigned char *a, unsigned char *b)
{
	return BitJaccardDistance(bytes, a, b, 0, 0, 0);
}
