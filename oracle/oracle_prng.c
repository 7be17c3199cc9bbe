/*
 * oracle_prng.c -- restatement of PostgreSQL's pg_prng (src/common/pg_prng.c,
 * PG >= 15): xoroshiro128** seeded through splitmix64.  pgvector reaches it via
 * RandomDouble()/RandomInt() (src/ivfflat.h:86-94, src/hnsw.h:104-109).
 *
 * TEST INFRASTRUCTURE ONLY.  PostgreSQL core is not part of /root/reference,
 * so this piece is restated from the published algorithm (Blackman & Vigna,
 * "Scrambled linear pseudorandom number generators") and PostgreSQL's
 * documented output mapping; parity for it is UNPINNED (no server here to
 * compare streams with).  It only feeds k-means++ seeding, empty-cluster
 * refills and HNSW level draws -- never a distance.
 */
#include "pgv_oracle.h"

#include <math.h>

static inline uint64_t
rotl64(uint64_t x, int bits)
{
	return (x << bits) | (x >> (64 - bits));
}

static uint64_t
splitmix64(uint64_t *state)
{
	uint64_t	v = (*state += UINT64_C(0x9E3779B97f4A7C15));

	v = (v ^ (v >> 30)) * UINT64_C(0xBF58476D1CE4E5B9);
	v = (v ^ (v >> 27)) * UINT64_C(0x94D049BB133111EB);
	return v ^ (v >> 31);
}

void
ora_prng_seed(ora_prng * st, uint64_t seed)
{
	st->s0 = splitmix64(&seed);
	st->s1 = splitmix64(&seed);
	/* all-zero state is the one invalid state of xoroshiro */
	if (st->s0 == 0 && st->s1 == 0)
	{
		st->s0 = UINT64_C(0x5851F42D4C957F2D);
		st->s1 = UINT64_C(0x14057B7EF767814F);
	}
}

uint64_t
ora_prng_u64(ora_prng * st)
{
	uint64_t	s0 = st->s0;
	uint64_t	sx = st->s1 ^ s0;
	uint64_t	out = rotl64(s0 * 5, 7) * 9;

	st->s0 = rotl64(s0, 24) ^ sx ^ (sx << 16);
	st->s1 = rotl64(sx, 37);
	return out;
}

/* pg_prng_uint32: the high half of the 64-bit output */
uint32_t
ora_prng_u32(ora_prng * st)
{
	return (uint32_t) (ora_prng_u64(st) >> 32);
}

/* pg_prng_double: 52 random mantissa bits -> [0, 1) */
double
ora_prng_double(ora_prng * st)
{
	return ldexp((double) (ora_prng_u64(st) >> (64 - 52)), -52);
}

double
ora_prng_double_cb(void *st)
{
	return ora_prng_double((ora_prng *) st);
}

uint32_t
ora_prng_u32_cb(void *st)
{
	return ora_prng_u32((ora_prng *) st);
}
