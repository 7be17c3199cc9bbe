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
SH_SCOPE void PGSHIM_SH(start_iterate) (PGSHIM_SH(hash) * tb, PGSHIM_SH(iterator) * iter);
SH_SCOPE	SH_ELEMENT_TYPE *PGSHIM_SH(iterate) (PGSHIM_SH(hash) * tb, PGSHIM_SH(iterator) * iter);
#endif

#undef SH_PREFIX
#undef SH_ELEMENT_TYPE
#undef SH_KEY_TYPE
#undef SH_SCOPE
#undef SH_DECLARE
#undef SH_DEFINE
#undef PGSHIM_SH
#undef PGSHIM_SH_CAT
#undef PGSHIM_SH_CAT_
