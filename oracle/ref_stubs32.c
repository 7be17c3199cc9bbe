/*
 * ref_stubs32.c -- symbols the reference's src/vector.c references from the parts of it that oracle/ref_glue32.c never
 * calls (text / binary I/O, casts, aggregates, _PG_init): each aborts if reached.  A file of its own so that it need
 * not agree with the stand-in headers' prototypes.  -DPGV_REF_STUBS_IN_DRIVER (tests/c/ext_driver.c with the reference's
 * ivfscan.c + vector.c linked in) leaves out what that program defines itself.  TEST INFRASTRUCTURE ONLY.
 */
#include <stdio.h>
#include <stdlib.h>

#define NOT_REACHED(name) void name(void) { fprintf(stderr, "pgvref32: " #name " reached\n"); abort(); }
NOT_REACHED(ARR_DATA_PTR)
NOT_REACHED(ARR_DIMS)
NOT_REACHED(ARR_ELEMTYPE)
NOT_REACHED(ARR_HASNULL)
NOT_REACHED(ARR_NDIM)
NOT_REACHED(ArrayGetIntegerTypmods)
#ifndef PGV_REF_STUBS_IN_DRIVER
NOT_REACHED(BitvecInit)
#endif
#ifndef PGV_REF_STUBS_IN_DRIVER
NOT_REACHED(DirectFunctionCall1Coll)
#endif
#ifndef PGV_REF_STUBS_IN_DRIVER
NOT_REACHED(HalfvecInit)
#endif
#ifndef PGV_REF_STUBS_IN_DRIVER
NOT_REACHED(HnswInit)
#endif
NOT_REACHED(InitBitVector)
#ifndef PGV_REF_STUBS_IN_DRIVER
NOT_REACHED(IvfflatInit)
#endif
#ifndef PGV_REF_STUBS_IN_DRIVER
NOT_REACHED(PgvGpuInit)
#endif
NOT_REACHED(array_contains_nulls)
NOT_REACHED(construct_array)
NOT_REACHED(deconstruct_array)
NOT_REACHED(float_to_shortest_decimal_bufn)
NOT_REACHED(get_typlenbyvalalign)
NOT_REACHED(numeric_float4)
NOT_REACHED(pnstrdup)
NOT_REACHED(pq_begintypsend)
NOT_REACHED(pq_endtypsend)
NOT_REACHED(pq_getmsgfloat4)
NOT_REACHED(pq_getmsgint)
NOT_REACHED(pq_sendfloat4)
NOT_REACHED(pq_sendint16)
NOT_REACHED(scanner_isspace)

#if defined(PGV_HAVE_REF_HALFVEC)
/* the reference's src/halfvec.c + src/halfutils.c are in the program too: halfvec_l2_normalize is theirs; what halfvec.c's
 * text output names and no index path reaches */
NOT_REACHED(float_to_shortest_decimal_buf)
NOT_REACHED(sparsevec_l2_normalize)
#elif defined(PGV_HAVE_REF_HNSW) && defined(PGV_HAVE_REF_IVFINSERT)
NOT_REACHED(halfvec_l2_normalize)
NOT_REACHED(sparsevec_l2_normalize)
#elif defined(PGV_HAVE_REF_HNSW)
/* with the reference's src/hnswutils.c linked in as well (tests/c/ext_driver.c, phase "the reference's own hnswgettuple"):
 * what its insert / update-meta-page half names and a scan never reaches */
NOT_REACHED(GenericXLogStart)
NOT_REACHED(GenericXLogRegisterBuffer)
NOT_REACHED(GenericXLogFinish)
NOT_REACHED(MarkBufferDirty)
NOT_REACHED(PageInit)
NOT_REACHED(halfvec_l2_normalize)
NOT_REACHED(sparsevec_l2_normalize)
#endif

#if defined(PGV_HAVE_REF_IVFUTILS) && !defined(PGV_HAVE_REF_IVFINSERT)
/* src/ivfutils.c's page-append half (IvfflatAppendPage, IvfflatUpdateList): no scan or k-means reaches it (with the
 * reference's src/ivfinsert.c linked in -- PGV_HAVE_REF_IVFINSERT -- tests/c/pgshim_ref_runtime.c has the real things) */
NOT_REACHED(GenericXLogAbort)
#ifndef PGV_HAVE_REF_HNSW
NOT_REACHED(GenericXLogStart)
NOT_REACHED(GenericXLogRegisterBuffer)
NOT_REACHED(GenericXLogFinish)
NOT_REACHED(MarkBufferDirty)
NOT_REACHED(PageInit)
#endif
#endif
