/*
 * ext/shim/hnsw.h -- NOT pgvector's src/hnsw.h.  The fields, macros and functions of it that ext/hnswscan_gpu.c
 * touches, declared for the syntax check only (names as in src/hnsw.h:36-70, 182-222, 348-427, 470-475; inside the
 * extension the real header is used and HnswScanOpaqueData gains ONE field, marked "added").
 */
#ifndef EXT_SHIM_HNSW_H
#define EXT_SHIM_HNSW_H
#include "pgshim.h"
#include "ivfflat.h"			/* Vector */

#define HNSW_MAX_DIM 2000
#define HNSW_METAPAGE_BLKNO 0
#define HNSW_HEAD_BLKNO 1
#define HNSW_ELEMENT_TUPLE_TYPE 1
#define HNSW_NEIGHBOR_TUPLE_TYPE 2
#define HNSW_HEAPTIDS 10
#define HNSW_MAX_EF_SEARCH 1000
#define HNSW_DISTANCE_PROC 1
#define HNSW_NORM_PROC 2

extern int	hnsw_ef_search;
extern int	hnsw_iterative_scan;
#define HNSW_ITERATIVE_SCAN_OFF 0	/* src/hnsw.h: typedef enum HnswIterativeScanMode, first member */

typedef struct HnswMetaPageData
{
	uint32		magicNumber;
	uint32		version;
	uint32		dimensions;
	uint16		m;
	uint16		efConstruction;
	BlockNumber entryBlkno;
	OffsetNumber entryOffno;
	int16		entryLevel;
	BlockNumber insertPage;
}			HnswMetaPageData;
#define HnswPageGetMeta(page) ((HnswMetaPageData *) PageGetContents(page))

typedef struct HnswPageOpaqueData
{
	BlockNumber nextblkno;
	uint16		unused;
	uint16		page_id;
}			HnswPageOpaqueData;
typedef HnswPageOpaqueData *HnswPageOpaque;
#define HnswPageGetOpaque(page) ((HnswPageOpaque) PageGetSpecialPointer(page))

typedef struct HnswElementTupleData
{
	uint8		type;
	uint8		level;
	uint8		deleted;
	uint8		version;
	ItemPointerData heaptids[HNSW_HEAPTIDS];
	ItemPointerData neighbortid;
	uint16		unused;
	Vector		data;
}			HnswElementTupleData;
typedef HnswElementTupleData *HnswElementTuple;
#define HnswIsElementTuple(tup) ((tup)->type == HNSW_ELEMENT_TUPLE_TYPE)

typedef struct HnswNeighborTupleData
{
	uint8		type;
	uint8		version;
	uint16		count;
	ItemPointerData indextids[];
}			HnswNeighborTupleData;
typedef HnswNeighborTupleData *HnswNeighborTuple;

typedef struct HnswElementData
{
	ItemPointerData heaptids[HNSW_HEAPTIDS];
	uint8		heaptidsLength;
	uint8		level;
	BlockNumber blkno;
	OffsetNumber offno;
}			HnswElementData;
typedef HnswElementData *HnswElement;
typedef union
{
	HnswElementData *ptr;
}			HnswElementPtr;
#define HnswPtrStore(base, hp, value) ((void) (base), (void) ((hp).ptr = (value)))

typedef struct HnswSearchCandidate
{
	HnswElementPtr element;
	double		distance;
}			HnswSearchCandidate;

typedef struct HnswTypeInfo
{
	int			maxDimensions;
}			HnswTypeInfo;

typedef struct HnswSupport
{
	FmgrInfo   *procinfo;
	FmgrInfo   *normprocinfo;
	Oid			collation;
}			HnswSupport;

typedef struct HnswScanOpaqueData
{
	const HnswTypeInfo *typeInfo;
	bool		first;
	List	   *w;
	int			m;
	int64		tuples;
	HnswSupport support;
	void	   *gpu;			/* added: PgvHnswScan of ext/hnswscan_gpu.c, NULL when the scan stays on the CPU */
}			HnswScanOpaqueData;
typedef HnswScanOpaqueData *HnswScanOpaque;

/* FUNCTION 1 of the inner-product / cosine and L1 opclasses (src/vector.c:632-646, :728-735, src/halfvec.c) */
Datum		vector_negative_inner_product(void *fcinfo);
Datum		halfvec_negative_inner_product(void *fcinfo);
Datum		l1_distance(void *fcinfo);
Datum		halfvec_l1_distance(void *fcinfo);

HnswElement HnswInitElementFromBlock(BlockNumber blkno, OffsetNumber offno);
void		HnswAddHeapTid(HnswElement element, ItemPointer heaptid);
FmgrInfo   *HnswOptionalProcInfo(Relation index, uint16 procnum);
const HnswTypeInfo *HnswGetTypeInfo(Relation index);
#endif
