/*
 * NOT PostgreSQL's lib/simplehash.h.  The reference's hnsw.h instantiates it three times with SH_DECLARE (tidhash,
 * pointerhash, offsethash): for the syntax check of the patched files (tests/test_ext_patch_cpu.py) each instantiation
 * gets its table type and the prototypes the reference's .c files call -- declarations only, nothing is defined.
 * Deliberately without an include guard, as the template it stands in for.
 */
#include "pgshim_ref.h"

#define PGSHIM_SH_CAT_(a, b) a##b
#define PGSHIM_SH_CAT(a, b) PGSHIM_SH_CAT_(a, b)
#define PGSHIM_SH(name) PGSHIM_SH_CAT(SH_PREFIX, PGSHIM_SH_CAT(_, name))

#ifdef SH_DECLARE
typedef struct PGSHIM_SH(hash)
{
	uint64		size;
	uint32		members;
	SH_ELEMENT_TYPE *data;
	MemoryContext ctx;
	void	   *private_data;
}			PGSHIM_SH(hash);

typedef struct PGSHIM_SH(iterator)
{
	uint32		cur;
	uint32		end;
	bool		done;
}			PGSHIM_SH(iterator);

SH_SCOPE	PGSHIM_SH(hash) * PGSHIM_SH(create) (MemoryContext ctx, uint32 nelements, void *private_data);
SH_SCOPE void PGSHIM_SH(destroy) (PGSHIM_SH(hash) * tb);
SH_SCOPE void PGSHIM_SH(reset) (PGSHIM_SH(hash) * tb);
SH_SCOPE	SH_ELEMENT_TYPE *PGSHIM_SH(insert) (PGSHIM_SH(hash) * tb, SH_KEY_TYPE key, bool *found);
SH_SCOPE	SH_ELEMENT_TYPE *PGSHIM_SH(lookup) (PGSHIM_SH(hash) * tb, SH_KEY_TYPE key);
SH_SCOPE bool PGSHIM_SH(delete) (PGSHIM_SH(hash) * tb, SH_KEY_TYPE key);
SH_SCOPE	SH_ELEMENT_TYPE *PGSHIM_SH(insert_hash) (PGSHIM_SH(hash) * tb, SH_KEY_TYPE key, uint32 hash, bool *found);
SH_SCOPE	SH_ELEMENT_TYPE *PGSHIM_SH(lookup_hash) (PGSHIM_SH(hash) * tb, SH_KEY_TYPE key, uint32 hash);
SH_SCOPE void PGSHIM_SH(start_iterate) (PGSHIM_SH(hash) * tb, PGSHIM_SH(iterator) * iter);
SH_SCOPE	SH_ELEMENT_TYPE *PGSHIM_SH(iterate) (PGSHIM_SH(hash) * tb, PGSHIM_SH(iterator) * iter);
#endif

#ifdef SH_DEFINE
/*
 * A working table for the programs that LINK the reference's files (tests/test_ext_runtime_cpu.py: hnswutils.c defines the
 * three tables): open addressing with linear probing over a power-of-two array, grown at 3/4 full.  SH_KEY names the key
 * member, SH_HASH_KEY / SH_EQUAL hash and compare, `status` (0 empty, 1 in use) is the element's own field -- the
 * template's parameters as the reference sets them.  Not PostgreSQL's algorithm (robin hood): only its interface.
 */
static void
PGSHIM_SH(grow_) (PGSHIM_SH(hash) * tb, uint64 newsize)
{
	SH_ELEMENT_TYPE *old = tb->data;
	uint64		oldsize = tb->size;

	tb->data = MemoryContextAllocZero(tb->ctx, sizeof(SH_ELEMENT_TYPE) * newsize);
	tb->size = newsize;
	tb->members = 0;
	for (uint64 i = 0; i < oldsize; i++)
		if (old[i].status)
		{
			uint64		at = SH_HASH_KEY(tb, old[i].SH_KEY) & (newsize - 1);

			while (tb->data[at].status)
				at = (at + 1) & (newsize - 1);
			tb->data[at] = old[i];
			tb->members++;
		}
	if (old)
		pfree(old);
}

SH_SCOPE	PGSHIM_SH(hash) *
PGSHIM_SH(create) (MemoryContext ctx, uint32 nelements, void *private_data)
{
	PGSHIM_SH(hash) * tb = MemoryContextAllocZero(ctx, sizeof(PGSHIM_SH(hash)));
	uint64		size = 16;

	while (size < (uint64) nelements * 2)
		size <<= 1;
	tb->ctx = ctx;
	tb->private_data = private_data;
	PGSHIM_SH(grow_) (tb, size);
	return tb;
}

SH_SCOPE void
PGSHIM_SH(destroy) (PGSHIM_SH(hash) * tb)
{
	pfree(tb->data);
	pfree(tb);
}

SH_SCOPE void
PGSHIM_SH(reset) (PGSHIM_SH(hash) * tb)
{
	memset(tb->data, 0, sizeof(SH_ELEMENT_TYPE) * tb->size);
	tb->members = 0;
}

SH_SCOPE	SH_ELEMENT_TYPE *
PGSHIM_SH(insert) (PGSHIM_SH(hash) * tb, SH_KEY_TYPE key, bool *found)
{
	uint64		at;

	if ((uint64) (tb->members + 1) * 4 > tb->size * 3)
		PGSHIM_SH(grow_) (tb, tb->size * 2);
	at = SH_HASH_KEY(tb, key) & (tb->size - 1);
	while (tb->data[at].status)
	{
		if (SH_EQUAL(tb, tb->data[at].SH_KEY, key))
		{
			*found = true;
			return &tb->data[at];
		}
		at = (at + 1) & (tb->size - 1);
	}
	*found = false;
	tb->data[at].SH_KEY = key;
	tb->data[at].status = 1;
	tb->members++;
	return &tb->data[at];
}

/* (the caller's hash is the one SH_HASH_KEY gives: recomputed here, the argument only has to be accepted) */
SH_SCOPE	SH_ELEMENT_TYPE *
PGSHIM_SH(insert_hash) (PGSHIM_SH(hash) * tb, SH_KEY_TYPE key, uint32 hash, bool *found)
{
	(void) hash;
	return PGSHIM_SH(insert) (tb, key, found);
}

SH_SCOPE	SH_ELEMENT_TYPE *
PGSHIM_SH(lookup) (PGSHIM_SH(hash) * tb, SH_KEY_TYPE key)
{
	uint64		at = SH_HASH_KEY(tb, key) & (tb->size - 1);

	while (tb->data[at].status)
	{
		if (SH_EQUAL(tb, tb->data[at].SH_KEY, key))
			return &tb->data[at];
		at = (at + 1) & (tb->size - 1);
	}
	return NULL;
}
#endif

#undef SH_PREFIX
#undef SH_ELEMENT_TYPE
#undef SH_KEY
#undef SH_HASH_KEY
#undef SH_EQUAL
#undef SH_KEY_TYPE
#undef SH_SCOPE
#undef SH_DECLARE
#undef SH_DEFINE
#undef PGSHIM_SH
#undef PGSHIM_SH_CAT
#undef PGSHIM_SH_CAT_
